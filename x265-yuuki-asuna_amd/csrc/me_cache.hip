// me_cache.hip - the CONSUMER side of the exhaustive search: a frame-granular, host-pointer entry on top of
// x265hip_me_fullsearch ("results cached in device/pinned buffers that the per-call stubs look up").
//
// A host encoder owns pictures in the reference's PicYuv layout (common/picyuv.cpp:87-114).  Per (source picture, reference
// picture) it submits the two luma buffers once; a worker thread of this library uploads them, runs ONE exhaustive-search launch
// for every CTU, and streams the SAD surfaces into pinned host memory one CTU row at a time, raising a per-row flag as each row
// lands - the order in which the reference's wavefront (WPP) rows need them, so the encode overlaps the PCIe transfer.  The
// encoder's sad / sad_x3 / sad_x4 slots then LOOK UP (include/x265hip.h: x265hip_surf_lookup) instead of computing; a row that
// has not arrived, a displacement outside the window or a PU that is not a union of 8x8 blocks falls back to the host's own
// primitive - the values are identical either way (SAD is additive over the 8x8 grid), so the bitstream cannot change.
// Reference call sites served: motion.cpp:228-328 (COST_MV / COST_MV_X3 / X4), :397-604 (STAR), :1069-1083 (UMH), :1397-1445 (FULL).
#include "common.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

using namespace x265hip;

struct x265hip_me_cache
{
    x265hip_me_cache_params prm;
    int bpp, ctusW, ctusH, nc, ng, groupBytes, device;
    size_t planeBytes, surfBytes, rowBytes, orgOffset;     // orgOffset: bytes from the buffer start to sample (0,0)
    void* dFenc = nullptr; void* dRef = nullptr;
    uint64_t fencKeyOnDevice = ~0ull;
    hipStream_t stream = nullptr;
    // Pinned copies of the submitted SOURCE pictures: one per source picture that still has queued pairs (shared by the pairs of a picture,
    // counted by them), so that a submit of the next picture never replaces samples a queued pair has yet to upload.  Grown on demand.
    struct FencStage { uint8_t* buf = nullptr; uint64_t key = ~0ull; int users = 0; uint64_t stamp = 0; };
    std::vector<FencStage> fencStages;
    uint64_t fencStamp = 0;
    std::mutex stageMu;
    struct Slot
    {
        uint8_t* surf = nullptr;           // pinned host surfaces
        void* dSurf = nullptr;             // device surfaces: every pair of a batch keeps its own, the rows are downloaded interleaved
        uint8_t* stageRef = nullptr;       // pinned copy of the submitted reference plane (taken inside submit: the caller's buffer
                                           //   need not outlive the call)
        std::vector<hipEvent_t> rowEvents;
        std::atomic<int>* ready = nullptr; // per CTU row
        std::atomic<int> generation{0};
    };
    std::vector<Slot> slots;
    struct Job { int slot; int generation; int fencStage; uint64_t fencKey; };      // the source travels WITH the job (copied under mu)
    std::deque<Job> queue;
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false;
    std::thread worker;
    // statistics
    std::atomic<uint64_t> fills{0}, failed{0}, batches{0};
    std::atomic<uint64_t> usUpload{0}, usKernel{0}, usDownload{0}, bytesDown{0};
    char workerError[256] = "";
};

namespace {

double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// A batch = the (source, reference) pairs that were queued together (normally all references of one picture): every pair is
// uploaded and searched (one launch each), then the surfaces come down CTU ROW BY CTU ROW ACROSS THE PAIRS - row 0 of every pair,
// row 1 of every pair, ... - the order in which the reference's wavefront rows need them, with a flag raised per (pair, row).
int run_batch(x265hip_me_cache* c, const std::vector<x265hip_me_cache::Job>& batch)
{
    X265HIP_TRY(hipSetDevice(c->device));
    apply_wait_policy(c->device);
    const double t0 = now_us();
    for (const auto& job : batch)
    {
        x265hip_me_cache::Slot& s = c->slots[job.slot];
        if (c->fencKeyOnDevice != job.fencKey)
        {
            const uint8_t* src;
            { std::lock_guard<std::mutex> lk(c->stageMu); src = c->fencStages[job.fencStage].buf; }      // the vector may grow under a submit
            const double u0 = now_us();
            X265HIP_TRY(hipMemcpyAsync(c->dFenc, src, c->planeBytes, hipMemcpyHostToDevice, c->stream));    // stream order: after the searches
            X265HIP_TRY(hipStreamSynchronize(c->stream));                                                   //   that still read the old source
            c->usUpload += (uint64_t)(now_us() - u0);
            c->fencKeyOnDevice = job.fencKey;
        }
        X265HIP_TRY(hipMemcpyAsync(c->dRef, s.stageRef, c->planeBytes, hipMemcpyHostToDevice, c->stream));
        x265hip_me_params p;
        memset(&p, 0, sizeof(p));
        p.depth = c->prm.depth; p.width = c->prm.width; p.height = c->prm.height; p.range = c->prm.range;
        p.fenc = (const uint8_t*)c->dFenc + c->orgOffset; p.fenc_stride = c->prm.stride;
        p.fref = (const uint8_t*)c->dRef + c->orgOffset;  p.fref_stride = c->prm.stride;
        p.surf = (int32_t*)s.dSurf; p.surf_format = c->prm.surf_format;
        int rc = x265hip_me_fullsearch(&p, c->stream);          // stream order: the next pair's upload waits for this launch
        if (rc) return rc;
    }
    X265HIP_TRY(hipStreamSynchronize(c->stream));
    const double t1 = now_us();
    // all row copies are queued at once (the PCIe pipe stays full), each followed by an event; flags are raised as they complete
    for (int r = 0; r < c->ctusH; r++)
        for (const auto& job : batch)
        {
            x265hip_me_cache::Slot& s = c->slots[job.slot];
            X265HIP_TRY(hipMemcpyAsync(s.surf + (size_t)r * c->rowBytes, (const uint8_t*)s.dSurf + (size_t)r * c->rowBytes, c->rowBytes,
                                       hipMemcpyDeviceToHost, c->stream));
            X265HIP_TRY(hipEventRecord(s.rowEvents[r], c->stream));
        }
    for (int r = 0; r < c->ctusH; r++)
        for (const auto& job : batch)
        {
            x265hip_me_cache::Slot& s = c->slots[job.slot];
            X265HIP_TRY(hipEventSynchronize(s.rowEvents[r]));
            if (s.generation.load(std::memory_order_acquire) == job.generation)      // a newer submit owns the flags otherwise
                s.ready[r].store(job.generation, std::memory_order_release);
        }
    const double t2 = now_us();
    c->usKernel += (uint64_t)(t1 - t0); c->usDownload += (uint64_t)(t2 - t1);
    c->bytesDown += c->surfBytes * batch.size();
    c->fills += batch.size();
    c->batches++;
    return 0;
}

void worker_main(x265hip_me_cache* c)
{
    for (;;)
    {
        std::vector<x265hip_me_cache::Job> batch, dropped;
        {
            std::unique_lock<std::mutex> lk(c->mu);
            c->cv.wait(lk, [c] { return c->stop || !c->queue.empty(); });
            if (c->stop) return;
            while (!c->queue.empty())                   // everything queued so far is one batch
            {
                const x265hip_me_cache::Job job = c->queue.front();
                c->queue.pop_front();
                if (c->slots[job.slot].generation.load() == job.generation) batch.push_back(job);
                else dropped.push_back(job);                                          // superseded before it ran
            }
        }
        if (!batch.empty() && run_batch(c, batch))
        {
            c->failed += batch.size();
            snprintf(c->workerError, sizeof(c->workerError), "%s", x265hip_last_error());
        }
        if (!batch.empty() || !dropped.empty())
        {
            std::lock_guard<std::mutex> lk(c->stageMu);          // the pairs are done with (or never needed) their source picture's staging copy
            for (const auto& j : batch) c->fencStages[j.fencStage].users--;
            for (const auto& j : dropped) c->fencStages[j.fencStage].users--;
        }
    }
}

void free_all(x265hip_me_cache* c)
{
    for (auto& s : c->slots)
    {
        if (s.surf) (void)hipHostFree(s.surf);
        if (s.dSurf) (void)hipFree(s.dSurf);
        if (s.stageRef) (void)hipHostFree(s.stageRef);
        for (hipEvent_t e : s.rowEvents) if (e) (void)hipEventDestroy(e);
        delete[] s.ready;
    }
    for (auto& f : c->fencStages) if (f.buf) (void)hipHostFree(f.buf);
    if (c->dFenc) (void)hipFree(c->dFenc);
    if (c->dRef) (void)hipFree(c->dRef);
    if (c->stream) (void)hipStreamDestroy(c->stream);
}

} // namespace

extern "C" {

int x265hip_me_cache_create(x265hip_me_cache** out, const x265hip_me_cache_params* p)
{
    if (!out || !p) { set_error("me_cache_create: NULL argument"); return X265HIP_EINVAL; }
    *out = nullptr;
    int rc = ensure_device();
    if (rc) return rc;
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("me_cache_create: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->width <= 0 || p->height <= 0 || (p->width & 63) || (p->height & 63))
    { set_error("me_cache_create: width/height must be whole CTUs (got %dx%d)", p->width, p->height); return X265HIP_EINVAL; }
    if (p->range < 1 || p->range > 256 || p->margin_x < p->range + 12 || p->margin_y < p->range + 12)
    { set_error("me_cache_create: range %d needs margins >= range + 12 (have %d / %d)", p->range, p->margin_x, p->margin_y); return X265HIP_EINVAL; }
    if (p->stride < p->width + 2 * p->margin_x) { set_error("me_cache_create: stride %ld < width + 2 * margin_x", (long)p->stride); return X265HIP_EINVAL; }
    if (p->slots < 1 || p->slots > 64) { set_error("me_cache_create: slots %d out of [1,64]", p->slots); return X265HIP_EINVAL; }
    if (p->surf_format != X265HIP_SURF_I32 && !((p->surf_format == X265HIP_SURF_PACKED || p->surf_format == X265HIP_SURF_PACKED_T) && p->depth == 8))
    { set_error("me_cache_create: surf_format %d for depth %d", p->surf_format, p->depth); return X265HIP_EINVAL; }
    x265hip_me_cache* c = new (std::nothrow) x265hip_me_cache;
    if (!c) { set_error("me_cache_create: out of memory"); return X265HIP_EINVAL; }
    c->prm = *p;
    c->bpp = p->depth == 8 ? 1 : 2;
    c->ctusW = p->width / 64; c->ctusH = p->height / 64;
    c->nc = 2 * p->range + 1; c->ng = (c->nc + 3) / 4;
    c->groupBytes = p->surf_format == X265HIP_SURF_I32 ? X265HIP_SURF_GROUP_BYTES_I32 : X265HIP_SURF_GROUP_BYTES_PACKED;
    c->planeBytes = (size_t)p->stride * (p->height + 2 * p->margin_y) * c->bpp;
    c->orgOffset = ((size_t)p->margin_y * p->stride + p->margin_x) * c->bpp;
    c->rowBytes = (size_t)c->ctusW * c->nc * c->ng * c->groupBytes;
    c->surfBytes = c->rowBytes * c->ctusH;
    if ((c->orgOffset & 3) || ((p->stride * c->bpp) & 3))
    { set_error("me_cache_create: sample (0,0) and the row pitch must be 4-byte aligned"); delete c; return X265HIP_EINVAL; }
    if (hipGetDevice(&c->device) != hipSuccess) c->device = 0;
#define MC_TRY(expr) do { if (check_hip((expr), #expr)) { free_all(c); delete c; return X265HIP_ENODEV; } } while (0)
    MC_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    MC_TRY(hipMalloc(&c->dFenc, c->planeBytes));
    MC_TRY(hipMalloc(&c->dRef, c->planeBytes));
    c->fencStages.resize(2);
    for (auto& f : c->fencStages) MC_TRY(hipHostMalloc((void**)&f.buf, c->planeBytes, hipHostMallocDefault));
    c->slots = std::vector<x265hip_me_cache::Slot>(p->slots);
    for (auto& s : c->slots)
    {
        MC_TRY(hipHostMalloc((void**)&s.surf, c->surfBytes, hipHostMallocDefault));
        MC_TRY(hipMalloc(&s.dSurf, c->surfBytes));
        MC_TRY(hipHostMalloc((void**)&s.stageRef, c->planeBytes, hipHostMallocDefault));
        s.rowEvents.assign(c->ctusH, nullptr);
        for (int r = 0; r < c->ctusH; r++) MC_TRY(hipEventCreateWithFlags(&s.rowEvents[r], hipEventDisableTiming));
        s.ready = new std::atomic<int>[c->ctusH];
        for (int r = 0; r < c->ctusH; r++) s.ready[r].store(0);
    }
#undef MC_TRY
    c->worker = std::thread(worker_main, c);
    *out = c;
    return 0;
}

void x265hip_me_cache_destroy(x265hip_me_cache* c)
{
    if (!c) return;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->stop = true;
    }
    c->cv.notify_all();
    if (c->worker.joinable()) c->worker.join();
    free_all(c);
    delete c;
}

/* Copies the planes (whole allocated buffers, margins included) and queues the searches as ONE batch; returns at once.  fenc_key names
 * the source picture (e.g. its POC): it is copied / uploaded once per key.  generations[i] receives slot i's new generation. */
int x265hip_me_cache_submit_batch(x265hip_me_cache* c, int n, const int* slots, const void* fenc_buf, uint64_t fenc_key, const void* const* ref_bufs,
                                  int* generations)
{
    if (!c || n < 1 || !slots || !fenc_buf || !ref_bufs || !generations) { set_error("me_cache_submit_batch: bad argument"); return X265HIP_EINVAL; }
    for (int i = 0; i < n; i++)
        if (slots[i] < 0 || slots[i] >= (int)c->slots.size() || !ref_bufs[i]) { set_error("me_cache_submit_batch: bad slot / plane %d", i); return X265HIP_EINVAL; }
    int stage = -1;
    {
        // the source picture's staging copy: the one that already holds this key, else the least recently used copy no queued pair counts
        // on, else a new one (two source pictures in flight is the normal case under frame threads; more only when the worker is behind)
        std::lock_guard<std::mutex> lk(c->stageMu);
        for (size_t k = 0; k < c->fencStages.size(); k++)
            if (c->fencStages[k].key == fenc_key) stage = (int)k;
        if (stage < 0)
        {
            for (size_t k = 0; k < c->fencStages.size(); k++)
                if (c->fencStages[k].users == 0 && (stage < 0 || c->fencStages[k].stamp < c->fencStages[stage].stamp)) stage = (int)k;
            if (stage < 0)
            {
                if (c->fencStages.size() >= 64) { set_error("me_cache_submit_batch: 64 source pictures are waiting for the worker"); return X265HIP_EBUSY; }
                x265hip_me_cache::FencStage f;
                if (ensure_device() || check_hip(hipHostMalloc((void**)&f.buf, c->planeBytes, hipHostMallocDefault), "hipHostMalloc(source staging)")) return X265HIP_ENODEV;
                c->fencStages.push_back(f);
                stage = (int)c->fencStages.size() - 1;
            }
            memcpy(c->fencStages[stage].buf, fenc_buf, c->planeBytes);
            c->fencStages[stage].key = fenc_key;
        }
        c->fencStages[stage].users += n;
        c->fencStages[stage].stamp = ++c->fencStamp;
    }
    std::vector<x265hip_me_cache::Job> jobs;
    for (int i = 0; i < n; i++)
    {
        x265hip_me_cache::Slot& s = c->slots[slots[i]];
        const int gen = s.generation.fetch_add(1) + 1;          // readers compare ready[row] with the generation they were handed
        memcpy(s.stageRef, ref_bufs[i], c->planeBytes);
        generations[i] = gen;
        jobs.push_back({ slots[i], gen, stage, fenc_key });
    }
    {
        std::lock_guard<std::mutex> lk(c->mu);
        for (const auto& j : jobs) c->queue.push_back(j);
    }
    c->cv.notify_one();
    return 0;
}

/* one pair: returns the slot's new GENERATION (> 0) or a negative error */
int x265hip_me_cache_submit(x265hip_me_cache* c, int slot, const void* fenc_buf, uint64_t fenc_key, const void* ref_buf)
{
    int gen = 0;
    const int rc = x265hip_me_cache_submit_batch(c, 1, &slot, fenc_buf, fenc_key, &ref_buf, &gen);
    return rc ? rc : gen;
}

const void* x265hip_me_cache_surface(x265hip_me_cache* c, int slot)
{
    return (c && slot >= 0 && slot < (int)c->slots.size()) ? c->slots[slot].surf : nullptr;
}

/* int [ctu rows]: row r of the slot's surfaces is complete when ready[r] == the generation x265hip_me_cache_submit returned */
const volatile int* x265hip_me_cache_ready(x265hip_me_cache* c, int slot)
{
    return (c && slot >= 0 && slot < (int)c->slots.size()) ? reinterpret_cast<const volatile int*>(c->slots[slot].ready) : nullptr;
}

int x265hip_me_cache_stats(x265hip_me_cache* c, x265hip_me_cache_stats_t* st)
{
    if (!c || !st) { set_error("me_cache_stats: NULL"); return X265HIP_EINVAL; }
    st->fills = c->fills; st->failed = c->failed; st->batches = c->batches; st->us_upload = c->usUpload; st->us_kernel = c->usKernel;
    st->us_download = c->usDownload; st->bytes_downloaded = c->bytesDown; st->surface_bytes = c->surfBytes;
    if (c->failed) set_error("me_cache worker: %s", c->workerError);
    return 0;
}

} // extern "C"
