// runtime.hip - process-level state of libx265hip.so: device selection, error text.
// There is deliberately no CPU fallback here: if HIP cannot give us a gfx950 device every
// entry point returns X265HIP_ENODEV and x265hip_last_error() says why.
#include "common.h"

#include <map>
#include <vector>
#include <mutex>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>

namespace x265hip {

static thread_local char t_err[512] = "";
static std::once_flag g_initOnce;
static int g_initRc = X265HIP_ENODEV;
static int g_device = -1;

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof(t_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char* what)
{
    if (e == hipSuccess)
        return 0;
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return X265HIP_ENODEV;
}

// WAIT POLICY (round 6).  Every host-side wait of this runtime - hipStreamSynchronize, hipEventSynchronize, with or without hipEventBlockingSync - SPINS under the default
// device flags: a thread that waits 200 ms for a kernel burns 200 ms of a core (tools/ubench/wait_cpu.hip, profiles/r06_wait_cpu.txt); with hipDeviceScheduleBlockingSync it
// burns 1.5 ms and a SHORT wait (launch + 20 us kernel + wait) costs the same 31 - 32 us either way.  The consumer services wait in worker threads and in the lookahead's /
// AQ's / weightAnalyse's callers.  Measured in the real encode (tools/r6_wait_ab.sh, one box, interleaved): blocking waits take 8 % off the process's CPU seconds
// (cfg3 74.3 -> 68.3 s, cfg4 169 -> 159 s) and leave the fps where it was or a little below (8.46 -> 8.25, 2.08 -> 2.08): the encode is bound by its row dependencies, not by
// free cores, and a worker that sleeps wakes a little later (more records arrive late).  So the library does NOT touch the process-wide flag by itself; a host that
// shares the machine asks for it: X265HIP_WAIT=block in the environment (read once) or x265hip_set_wait_policy(X265HIP_WAIT_BLOCK).
std::atomic<int> g_waitPolicy{-1};                 // -1: not decided yet (environment), else X265HIP_WAIT_*
std::atomic<unsigned long long> g_waitApplied{0};  // bit d: the policy has been applied to device d
std::atomic<bool> g_waitTouched{false};            // this library changed a device's scheduling flag at least once

static int wait_policy()
{
    int p = g_waitPolicy.load(std::memory_order_relaxed);
    if (p < 0)
    {
        const char* e = getenv("X265HIP_WAIT");
        p = (e && !strcmp(e, "block")) ? X265HIP_WAIT_BLOCK : X265HIP_WAIT_SPIN;
        g_waitPolicy.store(p, std::memory_order_relaxed);
    }
    return p;
}

void apply_wait_policy(int device)
{
    if (device < 0 || device >= 64) return;
    const unsigned long long bit = 1ull << device;
    if (g_waitApplied.fetch_or(bit, std::memory_order_relaxed) & bit) return;
    if (wait_policy() == X265HIP_WAIT_SPIN && !g_waitTouched.load(std::memory_order_relaxed)) return;      // the runtime's default, never touched: leave the flags alone
    g_waitTouched.store(true, std::memory_order_relaxed);
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) cur = -1;
    if (cur != device && hipSetDevice(device) != hipSuccess) return;
    (void)hipSetDeviceFlags(wait_policy() == X265HIP_WAIT_BLOCK ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto);       // a runtime that refuses keeps its default: slower host, same results
    (void)hipGetLastError();
    if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
}

static void do_init(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
    {
        set_error("no HIP device available (hipGetDeviceCount: %s, count %d); libx265hip has no CPU fallback",
                  hipGetErrorString(e), n);
        g_initRc = X265HIP_ENODEV;
        return;
    }
    if (device >= n)
    {
        set_error("x265hip_init: device %d requested, %d present", device, n);
        g_initRc = X265HIP_EINVAL;
        return;
    }
    if (device < 0 && hipGetDevice(&device) != hipSuccess)      // lazy init: validate the calling thread's current device
        device = 0;
    hipDeviceProp_t prop;
    if (check_hip(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties")) { g_initRc = X265HIP_ENODEV; return; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    {
        set_error("device %d is %s; this library only carries gfx950 (MI355X) code objects", device, prop.gcnArchName);
        g_initRc = X265HIP_ENODEV;
        return;
    }
    g_device = device;
    g_initRc = 0;
    apply_wait_policy(device);
}

int ensure_device()
{
    std::call_once(g_initOnce, do_init, -1);
    if (g_initRc)
    {
        if (!t_err[0])
            set_error("libx265hip: no usable gfx950 device (initialisation failed earlier)");
        return g_initRc;
    }
    return 0;
}

// A grow-only device buffer per (device, stream, slot) for entry points that need a few KB of stream-ordered scratch on every call.  Calls
// on one stream are ordered, so the next call may reuse the buffer; hipMallocAsync / hipFreeAsync per call kept the host from running
// ahead of the device on this runtime (the pattern-search step took twice its stage sum, round-2 verdict).  Returns NULL on failure
// (set_error holds the reason).  Round 4 (round-3 advisor):
//   * the ENQUEUE SEQUENCE of such an entry (clear the scratch, launch the kernels that use it) must not interleave with another
//     thread's sequence on the same stream - two threads on the NULL stream could queue memset A, memset B, kernel A, kernel B.  The
//     entries hold stream_sequence_lock(stream) while they enqueue; stream order then serialises the users of the buffer.
//   * a buffer that was handed out is never freed: growing retires the old one (it may be baked into a captured graph);
//   * a stream that is CAPTURING (torch.cuda.graph captures on a stream of its own, and allocation is prohibited there) adopts a spare
//     buffer: every allocation made outside capture leaves one spare of its size behind, so the launch-by-launch warm-up that precedes a
//     capture provides it.  No spare = an error that says so, not an allocation inside the capture.
namespace {
struct ScratchKey { int dev; hipStream_t s; int slot; bool operator<(const ScratchKey& o) const { return dev != o.dev ? dev < o.dev : (s != o.s ? s < o.s : slot < o.slot); } };
struct ScratchBuf { void* p; size_t n; };
std::mutex g_scratchMu;
std::map<ScratchKey, ScratchBuf> g_scratch;
std::map<std::pair<int, int>, std::vector<ScratchBuf>> g_scratchSpare;      // (device, slot) -> buffers no stream owns yet
std::map<std::pair<int, hipStream_t>, std::mutex> g_seqMu;
}

std::unique_lock<std::mutex> stream_sequence_lock(hipStream_t s)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::mutex* m;
    {
        std::lock_guard<std::mutex> lk(g_scratchMu);
        m = &g_seqMu[std::make_pair(dev, s)];              // std::map never moves its nodes
    }
    return std::unique_lock<std::mutex>(*m);
}

void* stream_scratch(hipStream_t s, int slot, size_t bytes)
{
    int dev = 0;
    if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (s) (void)hipStreamIsCapturing(s, &cap);
    const bool capturing = cap == hipStreamCaptureStatusActive;
    std::lock_guard<std::mutex> lk(g_scratchMu);
    ScratchBuf& b = g_scratch[ScratchKey{ dev, s, slot }];
    if (b.n >= bytes) return b.p;
    const size_t want = (bytes + 4095) & ~(size_t)4095;
    std::vector<ScratchBuf>& spare = g_scratchSpare[std::make_pair(dev, slot)];
    for (size_t i = 0; i < spare.size(); i++)
        if (spare[i].n >= want)
        {
            b = spare[i];                                    // the old buffer of this key (if any) stays allocated: graphs may hold it
            spare.erase(spare.begin() + i);
            if (!capturing && spare.empty())
            {
                ScratchBuf extra = { nullptr, b.n };         // leave a spare behind for the next stream that starts inside a capture (only when the pool ran dry:
                                                             // streams handed back with x265hip_stream_release refill it, so the footprint stays bounded)
                if (hipMalloc(&extra.p, extra.n) == hipSuccess) spare.push_back(extra);
            }
            return b.p;
        }
    if (capturing)
    {
        set_error("stream scratch: the stream is capturing and no buffer of %zu bytes was prepared - run the entry once outside the capture first (a warm-up on any stream leaves a spare)", want);
        return nullptr;
    }
    ScratchBuf fresh = { nullptr, want }, extra = { nullptr, want };
    if (check_hip(hipMalloc(&fresh.p, want), "hipMalloc(stream scratch)")) return nullptr;
    if (hipMalloc(&extra.p, want) == hipSuccess) spare.push_back(extra);
    b = fresh;                                               // a smaller buffer this key held before is retired, not freed
    return b.p;
}

} // namespace x265hip

using namespace x265hip;

extern "C" {

const char* x265hip_version(void) { return "x265hip 0.1 (gfx950, x265 3.5 EncoderPrimitives ABI, X265_BUILD 199)"; }

const char* x265hip_last_error(void) { return t_err; }

/* A host that destroys a stream tells the library first (round-4 advisor, low: per-stream state otherwise only ever grows).  The scratch buffers the stream's calls
 * used (x265hip_me_search, split x265hip_lowres_cost) go back to the spare pool - the next stream that needs one adopts it instead of allocating, so a process that
 * creates and destroys streams keeps a bounded footprint (the stream's enqueue-sequence lock node stays: see below).  Buffers are never freed here: a HIP graph captured on
 * the stream may still hold them, which is also why a stream whose graphs are still replayed must NOT be released (its buffers would be handed to another stream).
 * Call it with the stream idle, before hipStreamDestroy.  Returns the number of buffers returned to the pool, X265HIP_EBUSY while another thread is enqueuing on it. */
int x265hip_stream_release(void* stream)
{
    const hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(g_scratchMu);
    // The stream is looked up on EVERY device (round-5 advisor: keyed on the caller's current device, a stream of another device silently released nothing).
    // Its sequence-lock node is KEPT: stream_sequence_lock() takes the node's address under g_scratchMu and locks it after dropping g_scratchMu, so a node erased
    // here could be locked after its destruction; a mutex per (device, stream) ever seen is a few dozen bytes.  EBUSY while somebody holds it.
    for (auto& m : g_seqMu)
        if (m.first.second == s)
        {
            if (!m.second.try_lock()) { set_error("stream_release: another thread is enqueuing on this stream"); return X265HIP_EBUSY; }
            m.second.unlock();
        }
    int n = 0;
    for (auto it = g_scratch.begin(); it != g_scratch.end();)
        if (it->first.s == s)
        {
            if (it->second.p) { g_scratchSpare[std::make_pair(it->first.dev, it->first.slot)].push_back(it->second); n++; }
            it = g_scratch.erase(it);
        }
        else ++it;
    return n;
}

int x265hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int x265hip_init(int device)
{
    std::call_once(g_initOnce, do_init, device);
    if (g_initRc)
        return g_initRc;
    /* device >= 0: make it the calling thread's current device (every launch, hipMallocAsync and staging
     * buffer of this library goes to the current device of the thread that calls in); device < 0: keep
     * whatever the host process selected (e.g. torch.cuda.set_device) - it was validated as gfx950 above.
     * A second init that names ANOTHER device is honoured per thread after the same arch check. */
    if (device >= 0)
    {
        if (device != g_device)
        {
            hipDeviceProp_t prop;
            if (check_hip(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties")) return X265HIP_ENODEV;
            if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            {
                set_error("device %d is %s; this library only carries gfx950 (MI355X) code objects", device, prop.gcnArchName);
                return X265HIP_ENODEV;
            }
        }
        if (check_hip(hipSetDevice(device), "hipSetDevice")) return X265HIP_ENODEV;
        apply_wait_policy(device);
    }
    return 0;
}

extern "C" int x265hip_set_wait_policy(int policy)
{
    if (policy != X265HIP_WAIT_BLOCK && policy != X265HIP_WAIT_SPIN) { set_error("x265hip_set_wait_policy: policy %d", policy); return X265HIP_EINVAL; }
    g_waitPolicy.store(policy, std::memory_order_relaxed);
    g_waitApplied.store(0, std::memory_order_relaxed);          // devices are visited again
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess) apply_wait_policy(cur);
    return 0;
}

} // extern "C"
