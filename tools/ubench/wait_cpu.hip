// Micro-benchmark: how much CPU does a host thread burn while it waits for the device?  hipStreamSynchronize under the default device flags and under
// hipDeviceScheduleBlockingSync, hipEventSynchronize on a default event and on a hipEventBlockingSync event: thread CPU time per 200 ms of waiting, and the wake-up
// latency of a SHORT wait (a 20 us kernel).  The consumer services wait in worker threads and in the lookahead's callers while the encode is CPU-bound.
// Build: hipcc --offload-arch=gfx950 -O3 wait_cpu.hip -o wait_cpu ; run: ./wait_cpu [blocking]   (blocking: hipSetDeviceFlags(hipDeviceScheduleBlockingSync) first)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <ctime>

__global__ void spin(long long cycles, int* sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (cycles < 0) *sink = 1;
}
static double now(clockid_t c) { timespec t; clock_gettime(c, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char** argv)
{
    if (argc > 1 && !strcmp(argv[1], "blocking")) printf("hipSetDeviceFlags(hipDeviceScheduleBlockingSync) -> %d\n", (int)hipSetDeviceFlags(hipDeviceScheduleBlockingSync));
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int* sink; hipMalloc(&sink, 4);
    hipEvent_t plain, blocking;
    hipEventCreateWithFlags(&plain, hipEventDisableTiming);
    hipEventCreateWithFlags(&blocking, hipEventDisableTiming | hipEventBlockingSync);
    const long long longK = 20000000LL, shortK = 2000LL;          // wall_clock64 ticks at 100 MHz: 200 ms, 20 us
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, shortK, sink); hipStreamSynchronize(s);
    for (int mode = 0; mode < 3; mode++)
    {
        const char* name = mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "hipEventSynchronize(plain event)" : "hipEventSynchronize(hipEventBlockingSync event)";
        auto wait = [&]() { if (mode == 0) hipStreamSynchronize(s); else { hipEvent_t e = mode == 1 ? plain : blocking; hipEventRecord(e, s); hipEventSynchronize(e); } };
        double c0 = now(CLOCK_THREAD_CPUTIME_ID), w0 = now(CLOCK_MONOTONIC);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, longK, sink);
        wait();
        const double cpu = now(CLOCK_THREAD_CPUTIME_ID) - c0, wall = now(CLOCK_MONOTONIC) - w0;
        w0 = now(CLOCK_MONOTONIC);
        for (int i = 0; i < 200; i++) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, shortK, sink); wait(); }
        const double shortUs = 1e6 * (now(CLOCK_MONOTONIC) - w0) / 200;
        printf("%-52s long wait: wall %.1f ms, thread CPU %.1f ms (%.0f %%); launch + 20 us kernel + wait: %.1f us\n", name, 1e3 * wall, 1e3 * cpu, 100 * cpu / wall, shortUs);
    }
    return 0;
}
