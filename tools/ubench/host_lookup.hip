// Micro-benchmark behind DESIGN.md 5.1 (round 4): what does ONE random lookup into a surface buffer cost a host thread, by how the
// buffer was allocated?  The T3 seams answer pu[].sad calls from SAD surfaces the device wrote into pinned host memory; with the seams
// on, MotionEstimate::motionEstimate spent ~1 us per lookup (profiles/r04_encoder_family_profile_seams.txt) - a DRAM miss is 0.1 us.
//   a  hipHostMalloc(default)                       what csrc/me_stream.hip used through round 3
//   b  hipHostMalloc(hipHostMallocNonCoherent)
//   c  aligned_alloc + madvise(MADV_HUGEPAGE) + hipHostRegister      transparent huge pages under the pinning
//   d  plain malloc (no pinning)                    the floor
// Pattern: T threads, each N dependent-free random 4-byte reads at 208-byte record granularity inside a window of W MiB (W = the
// surfaces one search touches vs. everything resident), plus the same with a DMA write stream (hipMemcpyAsync D2H) running beside it.
// Build: hipcc --offload-arch=gfx950 -O3 host_lookup.hip -o host_lookup -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static double run(const uint8_t* buf, size_t bytes, size_t windowBytes, int threads, size_t n)
{
    std::vector<std::thread> th;
    std::vector<uint64_t> sums(threads);
    const double t0 = now();
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            uint64_t x = 88172645463325252ull + t * 7919, s = 0;
            const size_t base = (bytes - windowBytes) / threads * t;
            const size_t recs = windowBytes / 208;
            for (size_t i = 0; i < n; i++)
            {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                s += *(const uint32_t*)(buf + base + (x % recs) * 208 + 128);
            }
            sums[t] = s;
        });
    for (auto& t : th) t.join();
    const double dt = now() - t0;
    uint64_t s = 0; for (auto v : sums) s += v;
    if (s == 42) printf("!");
    return 1e9 * dt / n;          // ns per lookup per thread
}

int main(int argc, char** argv)
{
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 2048) << 20;
    const int threads = argc > 2 ? atoi(argv[2]) : 16;
    const size_t n = 4000000;
    void* a = nullptr; void* b = nullptr;
    if (hipHostMalloc(&a, bytes, hipHostMallocDefault) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; }
    if (hipHostMalloc(&b, bytes, hipHostMallocNonCoherent) != hipSuccess) b = nullptr;
    void* c = aligned_alloc(2 << 20, bytes);
    madvise(c, bytes, MADV_HUGEPAGE);
    memset(c, 1, bytes);
    const bool creg = hipHostRegister(c, bytes, hipHostRegisterDefault) == hipSuccess;
    void* d = malloc(bytes); memset(d, 1, bytes);
    memset(a, 1, bytes); if (b) memset(b, 1, bytes);
    void* dev = nullptr; hipMalloc(&dev, 256 << 20);
    hipStream_t st; hipStreamCreate(&st);
    FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    char thp[128] = "?"; if (f) { fgets(thp, sizeof(thp), f); fclose(f); }
    printf("# buffer %zu MiB, %d threads, %zu lookups each; THP: %s", bytes >> 20, threads, n, thp);
    const char* names[4] = { "hipHostMalloc default", "hipHostMalloc non-coherent", creg ? "aligned_alloc + MADV_HUGEPAGE + hipHostRegister" : "aligned_alloc + MADV_HUGEPAGE (register FAILED)", "malloc" };
    void* bufs[4] = { a, b, c, d };
    for (int i = 0; i < 4; i++)
    {
        if (!bufs[i]) continue;
        for (size_t w : { (size_t)1 << 20, (size_t)64 << 20, bytes / threads })
        {
            const double ns1 = run((const uint8_t*)bufs[i], bytes, w, 1, n);
            const double nsT = run((const uint8_t*)bufs[i], bytes, w, threads, n);
            printf("%-52s window %5zu MiB: %7.1f ns / lookup (1 thread)  %7.1f ns (%d threads)\n", names[i], w >> 20, ns1, nsT, threads);
        }
        if (i < 3)
        {
            // the same with the device writing into the buffer beside the readers (16 x 256 MiB D2H copies in flight)
            for (int k = 0; k < 16; k++) hipMemcpyAsync((uint8_t*)bufs[i] + (size_t)(k % (bytes >> 28 ? bytes >> 28 : 1)) * (256 << 20), dev, 256 << 20, hipMemcpyDeviceToHost, st);
            const double nsT = run((const uint8_t*)bufs[i], bytes, bytes / threads, threads, n);
            hipStreamSynchronize(st);
            printf("%-52s under D2H traffic      : %7.1f ns (%d threads)\n", names[i], nsT, threads);
        }
    }
    return 0;
}
