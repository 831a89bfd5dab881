// phase_cache.hip - the picture-granular, host-pointer CONSUMER of x265hip_phase_planes.
//
// A host encoder owns reconstructed pictures in the reference's PicYuv layout (common/picyuv.cpp:87-114).  Once per reference picture
// it submits the three buffers; a worker thread of this library uploads them, interpolates every fractional phase (15 luma planes,
// 63 per chroma plane) and streams the planes into pinned host memory.  The encoder's sub-sample refinement
// (MotionEstimate::subpelCompare, motion.cpp:1571-1664 - 12 to 20 block interpolations per searched PU at --subme 3) then compares the
// source block with a block of the right phase plane IN PLACE instead of interpolating it; while a picture's planes are still on
// their way the host interpolates itself - the samples are the same either way, so the bitstream cannot change.
#include "common.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

using namespace x265hip;

struct x265hip_phase_cache
{
    x265hip_phase_cache_params prm;
    int bpp, device;
    size_t lumaBytes, chromaBytes;         // one source plane
    size_t guardLo, guardHi;               // readable bytes around a device source plane (x265hip_phase_planes' contract)
    uint8_t* dSrc = nullptr;               // [guard | plane | guard] of the plane being interpolated
    uint8_t* dOut = nullptr;               // 63 chroma or 15 luma planes, whichever is larger
    hipStream_t stream = nullptr;
    struct Slot
    {
        uint8_t* stage[3] = { nullptr, nullptr, nullptr };      // pinned copies of the submitted buffers (taken inside submit)
        uint8_t* out[3] = { nullptr, nullptr, nullptr };        // pinned phase planes
        std::atomic<int> ready[2];
        std::atomic<int> generation{0};
    };
    std::vector<Slot> slots;
    struct Job { int slot; int generation; };
    std::deque<Job> queue;
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false;
    std::thread worker;
    std::atomic<uint64_t> fills{0}, failed{0}, usKernel{0}, usDownload{0}, bytesDown{0};
    char workerError[256] = "";
};

namespace {

double pc_now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int run_plane(x265hip_phase_cache* c, x265hip_phase_cache::Slot& s, int plane, double& usK, double& usD)
{
    const bool chroma = plane != 0;
    const size_t srcBytes = chroma ? c->chromaBytes : c->lumaBytes;
    const int nph = chroma ? 63 : 15;
    const double t0 = pc_now_us();
    X265HIP_TRY(hipMemcpyAsync(c->dSrc + c->guardLo, s.stage[plane], srcBytes, hipMemcpyHostToDevice, c->stream));
    x265hip_phase_planes_params p;
    p.depth = c->prm.depth; p.chroma = chroma; p.src = c->dSrc + c->guardLo; p.dst = c->dOut;
    p.stride = chroma ? c->prm.stride_c : c->prm.stride; p.rows = chroma ? c->prm.rows_c : c->prm.rows;
    int rc = x265hip_phase_planes(&p, c->stream);
    if (rc) return rc;
    X265HIP_TRY(hipStreamSynchronize(c->stream));
    const double t1 = pc_now_us();
    X265HIP_TRY(hipMemcpyAsync(s.out[plane], c->dOut, srcBytes * nph, hipMemcpyDeviceToHost, c->stream));
    X265HIP_TRY(hipStreamSynchronize(c->stream));
    usK += t1 - t0; usD += pc_now_us() - t1;
    c->bytesDown += srcBytes * nph;
    return 0;
}

int run_job(x265hip_phase_cache* c, const x265hip_phase_cache::Job& job)
{
    X265HIP_TRY(hipSetDevice(c->device));
    apply_wait_policy(c->device);
    x265hip_phase_cache::Slot& s = c->slots[job.slot];
    double usK = 0, usD = 0;
    int rc = run_plane(c, s, 0, usK, usD);
    if (rc) return rc;
    if (s.generation.load(std::memory_order_acquire) == job.generation) s.ready[0].store(job.generation, std::memory_order_release);
    if (c->prm.rows_c > 0)
    {
        for (int pl = 1; pl <= 2; pl++)
            if ((rc = run_plane(c, s, pl, usK, usD))) return rc;
        if (s.generation.load(std::memory_order_acquire) == job.generation) s.ready[1].store(job.generation, std::memory_order_release);
    }
    c->usKernel += (uint64_t)usK; c->usDownload += (uint64_t)usD;
    c->fills++;
    return 0;
}

void pc_worker(x265hip_phase_cache* c)
{
    for (;;)
    {
        x265hip_phase_cache::Job job;
        {
            std::unique_lock<std::mutex> lk(c->mu);
            c->cv.wait(lk, [c] { return c->stop || !c->queue.empty(); });
            if (c->stop) return;
            job = c->queue.front();
            c->queue.pop_front();
        }
        if (c->slots[job.slot].generation.load() != job.generation) continue;          // superseded before it ran
        if (run_job(c, job))
        {
            c->failed++;
            snprintf(c->workerError, sizeof(c->workerError), "%s", x265hip_last_error());
        }
    }
}

void pc_free(x265hip_phase_cache* c)
{
    for (auto& s : c->slots)
        for (int i = 0; i < 3; i++)
        {
            if (s.stage[i]) (void)hipHostFree(s.stage[i]);
            if (s.out[i]) (void)hipHostFree(s.out[i]);
        }
    if (c->dSrc) (void)hipFree(c->dSrc);
    if (c->dOut) (void)hipFree(c->dOut);
    if (c->stream) (void)hipStreamDestroy(c->stream);
}

} // namespace

extern "C" {

int x265hip_phase_cache_create(x265hip_phase_cache** out, const x265hip_phase_cache_params* p)
{
    if (!out || !p) { set_error("phase_cache_create: NULL argument"); return X265HIP_EINVAL; }
    *out = nullptr;
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("phase_cache_create: depth %d", p->depth); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    if (p->stride <= 0 || p->rows <= 0 || (p->stride & 3) || (p->rows & 3))
    { set_error("phase_cache_create: luma pitch %ld / rows %d must be positive multiples of 4", (long)p->stride, p->rows); return X265HIP_EINVAL; }
    if (p->rows_c < 0 || (p->rows_c > 0 && (p->stride_c <= 0 || (p->stride_c & 3) || (p->rows_c & 3))))
    { set_error("phase_cache_create: chroma pitch %ld / rows %d must be multiples of 4", (long)p->stride_c, p->rows_c); return X265HIP_EINVAL; }
    if (p->slots < 1 || p->slots > 64) { set_error("phase_cache_create: slots %d out of [1,64]", p->slots); return X265HIP_EINVAL; }
    int rc = ensure_device();
    if (rc) return rc;
    x265hip_phase_cache* c = new (std::nothrow) x265hip_phase_cache;
    if (!c) { set_error("phase_cache_create: out of memory"); return X265HIP_EINVAL; }
    c->prm = *p;
    c->bpp = bpp;
    c->lumaBytes = (size_t)p->stride * p->rows * bpp;
    c->chromaBytes = p->rows_c > 0 ? (size_t)p->stride_c * p->rows_c * bpp : 0;
    c->guardLo = ((size_t)4 * p->stride * bpp + 64 + 255) & ~(size_t)255;
    c->guardHi = (size_t)8 * p->stride * bpp + 256;
    if (hipGetDevice(&c->device) != hipSuccess) c->device = 0;
#define PC_TRY(expr) do { if (check_hip((expr), #expr)) { pc_free(c); delete c; return X265HIP_ENODEV; } } while (0)
    PC_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    PC_TRY(hipMalloc((void**)&c->dSrc, c->guardLo + c->lumaBytes + c->guardHi));
    PC_TRY(hipMemset(c->dSrc, 0, c->guardLo + c->lumaBytes + c->guardHi));
    const size_t outBytes = c->lumaBytes * 15 > c->chromaBytes * 63 ? c->lumaBytes * 15 : c->chromaBytes * 63;
    PC_TRY(hipMalloc((void**)&c->dOut, outBytes));
    c->slots = std::vector<x265hip_phase_cache::Slot>(p->slots);
    for (auto& s : c->slots)
    {
        s.ready[0].store(0); s.ready[1].store(0);
        PC_TRY(hipHostMalloc((void**)&s.stage[0], c->lumaBytes, hipHostMallocDefault));
        PC_TRY(hipHostMalloc((void**)&s.out[0], c->lumaBytes * 15, hipHostMallocDefault));
        for (int i = 1; i <= 2 && c->chromaBytes; i++)
        {
            PC_TRY(hipHostMalloc((void**)&s.stage[i], c->chromaBytes, hipHostMallocDefault));
            PC_TRY(hipHostMalloc((void**)&s.out[i], c->chromaBytes * 63, hipHostMallocDefault));
        }
    }
    PC_TRY(hipDeviceSynchronize());          // the zero fill of dSrc is queued on the null stream; the worker's stream does not order with it
#undef PC_TRY
    c->worker = std::thread(pc_worker, c);
    *out = c;
    return 0;
}

void x265hip_phase_cache_destroy(x265hip_phase_cache* c)
{
    if (!c) return;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->stop = true;
    }
    c->cv.notify_all();
    if (c->worker.joinable()) c->worker.join();
    pc_free(c);
    delete c;
}

int x265hip_phase_cache_submit(x265hip_phase_cache* c, int slot, const void* luma_buf, const void* cb_buf, const void* cr_buf)
{
    if (!c || slot < 0 || slot >= (int)c->slots.size() || !luma_buf || (c->chromaBytes && (!cb_buf || !cr_buf)))
    { set_error("phase_cache_submit: bad argument"); return X265HIP_EINVAL; }
    x265hip_phase_cache::Slot& s = c->slots[slot];
    const int gen = s.generation.fetch_add(1) + 1;
    memcpy(s.stage[0], luma_buf, c->lumaBytes);
    if (c->chromaBytes) { memcpy(s.stage[1], cb_buf, c->chromaBytes); memcpy(s.stage[2], cr_buf, c->chromaBytes); }
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->queue.push_back({ slot, gen });
    }
    c->cv.notify_one();
    return gen;
}

const void* x265hip_phase_cache_planes(x265hip_phase_cache* c, int slot, int plane)
{
    return (c && slot >= 0 && slot < (int)c->slots.size() && plane >= 0 && plane < 3) ? c->slots[slot].out[plane] : nullptr;
}

const volatile int* x265hip_phase_cache_ready(x265hip_phase_cache* c, int slot)
{
    return (c && slot >= 0 && slot < (int)c->slots.size()) ? reinterpret_cast<const volatile int*>(c->slots[slot].ready) : nullptr;
}

int x265hip_phase_cache_stats(x265hip_phase_cache* c, x265hip_phase_cache_stats_t* st)
{
    if (!c || !st) { set_error("phase_cache_stats: NULL"); return X265HIP_EINVAL; }
    st->fills = c->fills; st->failed = c->failed; st->us_upload_kernel = c->usKernel; st->us_download = c->usDownload;
    st->bytes_downloaded = c->bytesDown; st->bytes_per_picture = c->lumaBytes * 15 + 2 * c->chromaBytes * 63;
    if (c->failed) set_error("phase_cache worker: %s", c->workerError);
    return 0;
}

} // extern "C"
