// lookahead_host.hip - whole host functions behind host pointers: the lookahead's frame cost / intra estimates (rounds 2 - 3) and, at the end of the
// file, LookaheadTLD::calcAdaptiveQuantFrame (x265hip_aq_frame_host) and the frame encoder's weightAnalyse (x265hip_weight_analyse_host, with its
// compensation / weightCost kernels) - round 4.  First: the CONSUMER side of the lookahead's frame cost estimate: a host-pointer, frame-granular entry on top of
// x265hip_lowres_cost, shaped like the loop it replaces.  CostEstimateGroup::estimateFrameCost (encoder/slicetype.cpp:3115-3213)
// spends its time in the estimateCUCost loop over every 8x8 block of the half-resolution picture (:3216-3388: predictor candidates,
// the lowres motionEstimate, bi-directional candidates, intra competition, the frame / row sums); a host encoder whose Lowres
// planes live in host memory hands the (p0, b, p1) triple's planes and per-block arrays over, one launch walks the whole picture
// (the wavefront of dependent rows runs in one workgroup, csrc/lowres_cost_kernels.hip) and the per-block results land in the
// caller's Lowres arrays - lowresMvs, lowresMvCosts, lowresCosts, rowSatds - plus the frame sums.  Stateless and re-entrant: every
// calling thread (the reference scores several triples at once from its pool, slicetype.cpp:2389-2436 batch mode) owns a stream
// and grow-only device scratch, released at thread exit.  Same integers as the loop it replaces (tests: device == oracle == the real
// CostEstimateGroup::singleCost), so the slice-type decisions and the bitstream cannot change.
#include "common.h"
#include <chrono>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include "tile_interp.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

using namespace x265hip;

namespace {

struct LaThread
{
    hipStream_t stream = nullptr;
    uint8_t* dev = nullptr;
    size_t cap = 0;
    uint8_t* pin = nullptr;                    // pinned staging: one upload and one download per call instead of a pageable copy per array
    size_t pinCap = 0;
    ~LaThread()
    {
        if (!stream) return;
        (void)hipStreamSynchronize(stream);
        if (dev) (void)hipFree(dev);
        if (pin) (void)hipHostFree(pin);
        (void)hipStreamDestroy(stream);
    }
    int ensure(size_t need)
    {
        if (!stream)
        {
            // X265HIP_LA_PRIORITY=1 (experiment): the highest stream priority the device offers.  A pool thread of the host WAITS for every call made here, while the
            // services that share the device queue chip-filling searches nobody waits for row by row - yet at cfg3 it measured no gain (three interleaved pairs:
            // 8.91 / 8.88 / 8.17 fps with, 9.06 / 9.10 / 8.92 without, profiles/r06_lookahead_resident.txt): plain streams stay the default.
            int least = 0, greatest = 0;
            if (!getenv("X265HIP_LA_PRIORITY") || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || greatest == least)
                X265HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
            else
                X265HIP_TRY(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, greatest));
        }
        if (need <= cap) return 0;
        if (dev) X265HIP_TRY(hipFree(dev));
        dev = nullptr; cap = 0;
        size_t n = (size_t)1 << 22;
        while (n < need) n <<= 1;
        X265HIP_TRY(hipMalloc((void**)&dev, n));
        cap = n;
        return 0;
    }
    int ensure_pin(size_t need)
    {
        if (need <= pinCap) return 0;
        if (pin) X265HIP_TRY(hipHostFree(pin));
        pin = nullptr; pinCap = 0;
        size_t n = (size_t)1 << 20;
        while (n < need) n <<= 1;
        X265HIP_TRY(hipHostMalloc((void**)&pin, n, hipHostMallocDefault));
        pinCap = n;
        return 0;
    }
};

LaThread& la_thread()
{
    static thread_local LaThread t;
    return t;
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Device copies of host arrays the caller vouches for: plane_key != 0 names the CONTENT of a plane set (a picture's four lowres planes
// do not change while its frame number stays the same), so the dozens of (p0, b, p1) triples the lookahead scores around a picture
// upload each plane once.  Shared by all calling threads; entries are only read by kernels after their upload was synchronised.
// An entry handed out is PINNED until the call that asked for it has synchronised its stream (PlanePins below): neither the
// same-host-buffer replacement nor the LRU eviction may free a buffer another thread's - or this call's own, not yet launched -
// kernel is about to read (round-2 advisor finding).  A replaced but still pinned entry is only retired (host = NULL: it matches
// no later request) and freed by whoever evicts next after its last user let go.
// Round 6: (a) the vectors / vector costs a frame-cost estimate searched (lowresMvs[l][d], lowresMvCosts[l][d]: written once per lifetime of the
// picture, Lowres::init resets them, lowres.cpp:283-284) live here too, keyed by the picture's key: the estimates that reuse a list - three quarters
// of them - read the device copy the search left behind instead of uploading the host's arrays again.  (b) Buffers an entry gives up go to a pool
// and are handed out again for the same size: hipFree waits for the whole device (every other thread's 5 ms walk) with the cache locked.
struct CachedPlane { const void* host; uint64_t key; size_t bytes; void* dev; uint64_t stamp; int pins; };
std::mutex g_planeMu;
std::vector<CachedPlane> g_planes;
struct PooledBuffer { size_t bytes; void* dev; };
std::vector<PooledBuffer> g_bufferPool;
uint64_t g_planeClock = 0;
size_t g_planeBytes = 0;
constexpr size_t PLANE_CACHE_LIMIT = (size_t)6 << 30;          // 6 GiB of HBM at most
constexpr size_t BUFFER_POOL_LIMIT = 512;

// (g_planeMu held) a device buffer of `bytes` + 64: from the pool when one of that size waits there
int pool_take(size_t bytes, void** out)
{
    for (size_t i = 0; i < g_bufferPool.size(); i++)
        if (g_bufferPool[i].bytes == bytes) { *out = g_bufferPool[i].dev; g_bufferPool.erase(g_bufferPool.begin() + i); return 0; }
    X265HIP_TRY(hipMalloc(out, bytes + 64));
    return 0;
}
void pool_give(size_t bytes, void* dev)
{
    if (g_bufferPool.size() < BUFFER_POOL_LIMIT) g_bufferPool.push_back({ bytes, dev });
    else (void)hipFree(dev);
}
// (g_planeMu held) make room for an entry of `host`: an older entry of the same host buffer goes (the picture was replaced), then the least recently used beyond the limit
void cache_evict_for(const void* host, size_t bytes)
{
    for (size_t i = 0; i < g_planes.size();)
    {
        if (g_planes[i].host == host || (!g_planes[i].host && !g_planes[i].pins))
        {
            if (g_planes[i].pins) { g_planes[i].host = nullptr; i++; continue; }             // retired, freed once unpinned
            pool_give(g_planes[i].bytes, g_planes[i].dev); g_planeBytes -= g_planes[i].bytes; g_planes.erase(g_planes.begin() + i);
        }
        else i++;
    }
    while (g_planeBytes + bytes > PLANE_CACHE_LIMIT)
    {
        size_t lru = g_planes.size();
        for (size_t i = 0; i < g_planes.size(); i++)
            if (!g_planes[i].pins && (lru == g_planes.size() || g_planes[i].stamp < g_planes[lru].stamp)) lru = i;
        if (lru == g_planes.size()) break;                      // everything left is in use: exceed the soft limit rather than free it
        (void)hipFree(g_planes[lru].dev); g_planeBytes -= g_planes[lru].bytes; g_planes.erase(g_planes.begin() + lru);
    }
}

// returns the device copy of the host array (allocation start `host`, `bytes` long), uploading it on `s` when it is not cached yet;
// the entry comes back pinned (PlanePins)
int cached_plane(const void* host, uint64_t key, size_t bytes, hipStream_t s, void** out)
{
    std::lock_guard<std::mutex> lk(g_planeMu);
    for (auto& e : g_planes)
        if (e.host == host && e.key == key && e.bytes == bytes) { e.stamp = ++g_planeClock; e.pins++; *out = e.dev; return 0; }
    cache_evict_for(host, bytes);
    void* d = nullptr;
    int rc = pool_take(bytes, &d);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);           // other threads may use the entry as soon as the lock is released
    if (e != hipSuccess) { pool_give(bytes, d); X265HIP_TRY(e); }
    g_planes.push_back({ host, key, bytes, d, ++g_planeClock, 1 });
    g_planeBytes += bytes;
    *out = d;
    return 0;
}

// a device buffer that will BECOME the cache's copy of `host` once the caller has filled it and synchronised (cached_adopt); until then it is the caller's alone
int cached_reserve(size_t bytes, void** out)
{
    std::lock_guard<std::mutex> lk(g_planeMu);
    return pool_take(bytes, out);
}
void cached_adopt(const void* host, uint64_t key, size_t bytes, void* dev)
{
    std::lock_guard<std::mutex> lk(g_planeMu);
    cache_evict_for(host, bytes);
    g_planes.push_back({ host, key, bytes, dev, ++g_planeClock, 0 });
    g_planeBytes += bytes;
}
void cached_abandon(size_t bytes, void* dev)
{
    std::lock_guard<std::mutex> lk(g_planeMu);
    pool_give(bytes, dev);
}

// the entries one call took from the cache; unpinned when the call leaves (after its final stream synchronisation, or on an error path)
struct PlanePins
{
    void* dev[24]; int n = 0;
    hipStream_t stream;
    explicit PlanePins(hipStream_t s) : stream(s) {}
    void add(void* d) { if (n < 24) dev[n++] = d; }
    ~PlanePins()
    {
        if (!n) return;
        (void)hipStreamSynchronize(stream);          // error paths leave with work queued; the normal path is already idle here
        std::lock_guard<std::mutex> lk(g_planeMu);
        for (int i = 0; i < n; i++)
            for (auto& e : g_planes) if (e.dev == dev[i] && e.pins > 0) { e.pins--; break; }
    }
};

} // namespace

extern "C" void x265hip_lowres_planes_forget(void)
{
    std::lock_guard<std::mutex> lk(g_planeMu);
    for (size_t i = 0; i < g_planes.size();)          // entries a running call still reads are retired, not freed
    {
        if (g_planes[i].pins) { g_planes[i].host = nullptr; i++; continue; }
        (void)hipFree(g_planes[i].dev); g_planeBytes -= g_planes[i].bytes; g_planes.erase(g_planes.begin() + i);
    }
    for (auto& b : g_bufferPool) (void)hipFree(b.dev);
    g_bufferPool.clear();
}

namespace {
// diagnostic (X265HIP_LA_STATS=1: printed when the process ends): the estimates by what they had to search and the wall time of the calls
struct LaKinds
{
    std::atomic<uint64_t> n[6], us[6], maxUs[6];          // P: none / list 0;  B: none / list 0 only / list 1 only / both
    ~LaKinds()
    {
        if (!getenv("X265HIP_LA_STATS")) return;
        static const char* const name[6] = { "P, nothing searched", "P, list 0 searched", "B, nothing searched", "B, list 0 searched", "B, list 1 searched", "B, both lists searched" };
        for (int i = 0; i < 6; i++)
            if (n[i]) fprintf(stderr, "libx265hip: lowres_cost_host %-24s %8llu calls %10.3f ms each, longest %.3f ms\n", name[i], (unsigned long long)n[i].load(), 1e-3 * us[i].load() / n[i].load(), 1e-3 * maxUs[i].load());
    }
} g_laKinds;
// X265HIP_LA_RESIDENT_OFF=1 (A/B): the searched vectors are not kept on the device, every array travels by itself from / to pageable memory as before round 6's second half
const bool g_laResidentOff = getenv("X265HIP_LA_RESIDENT_OFF") != nullptr;
}

extern "C" int x265hip_lowres_cost_host(const x265hip_lowres_cost_host_params* p)
{
    int rc = ensure_device();
    if (rc) return rc;
    struct KindTimer
    {
        int kind; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        ~KindTimer()
        {
            if (kind < 0) return;
            g_laKinds.n[kind].fetch_add(1, std::memory_order_relaxed);
            const uint64_t us = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
            g_laKinds.us[kind].fetch_add(us, std::memory_order_relaxed);
            uint64_t m = g_laKinds.maxUs[kind].load(std::memory_order_relaxed);
            while (us > m && !g_laKinds.maxUs[kind].compare_exchange_weak(m, us, std::memory_order_relaxed)) { }
        }
    } kindTimer{ -1 };
    if (p && p->ref1[0]) kindTimer.kind = 2 + (p->do_search[0] ? 1 : 0) + (p->do_search[1] ? 2 : 0);
    else if (p) kindTimer.kind = p->do_search[0] ? 1 : 0;
    if (!p || !p->cur || !p->intra_cost || !p->cost_q || !p->mvs[0] || !p->mv_costs[0] || !p->lowres_costs || !p->row_satds || !p->frame)
    { set_error("lowres_cost_host: NULL operand"); return X265HIP_EINVAL; }
    for (int k = 0; k < 4; k++) if (!p->ref[k]) { set_error("lowres_cost_host: NULL list-0 plane %d", k); return X265HIP_EINVAL; }
    const bool bidir = p->ref1[0] != nullptr;
    if (bidir) for (int k = 0; k < 4; k++) if (!p->ref1[k]) { set_error("lowres_cost_host: NULL list-1 plane %d", k); return X265HIP_EINVAL; }
    const bool wbi = p->ref_bi[0] != nullptr;
    if (wbi && !bidir) { set_error("lowres_cost_host: ref_bi needs a B picture"); return X265HIP_EINVAL; }
    if (bidir && (!p->mvs[1] || !p->mv_costs[1])) { set_error("lowres_cost_host: NULL list-1 output"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("lowres_cost_host: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->width_in_cu <= 0 || p->height_in_cu <= 0 || p->lines <= 0 || p->margin_x < 32 || p->margin_y < 32 || p->stride < p->width_in_cu * 8 + 2 * p->margin_x)
    { set_error("lowres_cost_host: geometry %d x %d blocks, %d lines, margins %d / %d, stride %ld", p->width_in_cu, p->height_in_cu, p->lines, p->margin_x, p->margin_y, (long)p->stride); return X265HIP_EINVAL; }
    if (p->cost_q_half < 64) { set_error("lowres_cost_host: cost_q_half %d", p->cost_q_half); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    const int n = p->width_in_cu * p->height_in_cu;
    const int NL = bidir ? 2 : 1;
    // a plane as the caller holds it: (0,0) sits margin_y rows and margin_x samples into a buffer of lines + 2 * margin_y rows
    const size_t org = ((size_t)p->margin_y * p->stride + p->margin_x) * bpp;
    const size_t planeBytes = (size_t)p->stride * (p->lines + 2 * p->margin_y) * bpp;
    const int nplanes = 1 + 4 + (bidir ? 4 : 0) + (wbi ? 4 : 0);
    const size_t costBytes = (size_t)(2 * p->cost_q_half + 1) * 2;
    const bool searches = p->do_search[0] || (bidir && p->do_search[1]);
    // what the kernels of this request read: the vector-cost table only when a list is searched, the intra costs only on a P picture
    const bool needCost = searches, needIntra = !bidir;
    // the vectors of a list: searched = an output; reused = the device copy the search left behind (keyed pictures), else uploaded
    const uint64_t mvKey = g_laResidentOff ? 0 : p->plane_key_cur;
    // device layout: [planes without a key] [UP: everything this call uploads, one copy] [DOWN: everything it downloads, one copy]
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = align256(off + bytes); return o; };
    // which planes may come from the shared cache: cur (key_cur), the list-0 / list-1 references when they are the pictures' own planes
    uint64_t pkey[13];
    {
        int k = 0;
        pkey[k++] = p->plane_key_cur;
        for (int i = 0; i < 4; i++) pkey[k++] = p->plane_key_ref;
        if (bidir) for (int i = 0; i < 4; i++) pkey[k++] = p->plane_key_ref1;
        if (wbi) for (int i = 0; i < 4; i++) pkey[k++] = p->plane_key_ref_bi;
    }
    size_t oPlane[13];
    for (int i = 0; i < nplanes; i++) oPlane[i] = pkey[i] ? 0 : take(planeBytes + 64);
    const size_t oUp = off;
    const size_t oPair = take(sizeof(x265hip_lowres_cost_pair));
    const size_t oInvq = p->inv_qscale ? take((size_t)n * 4) : 0, oIntra = needIntra ? take((size_t)n * 4) : 0, oCost = needCost ? take(costBytes) : 0;
    size_t oMv[2] = { 0, 0 }, oMc[2] = { 0, 0 };
    for (int li = 0; li < NL; li++)
        if (!p->do_search[li] && !mvKey) { oMv[li] = take((size_t)n * 8); oMc[li] = take((size_t)n * 4); }
    const size_t upBytes = off - oUp;
    const size_t oDown = off;
    const size_t oFrame = take(32), oRows = take((size_t)p->height_in_cu * 4), oLc = take((size_t)n * 2);
    for (int li = 0; li < NL; li++)
        if (p->do_search[li]) { oMv[li] = take((size_t)n * 8); oMc[li] = take((size_t)n * 4); }
    const size_t downBytes = off - oDown;
    LaThread& t = la_thread();
    rc = t.ensure(off);
    if (!rc) rc = t.ensure_pin(upBytes + downBytes);
    if (rc) return rc;
    hipStream_t s = t.stream;
    uint8_t* d = t.dev;
    uint8_t* const pinUp = t.pin;
    uint8_t* const pinDown = t.pin + upBytes;
    // planes: the caller passes sample (0,0); the allocation starts `org` bytes before it
    const void* planes[13];
    int k = 0;
    planes[k++] = p->cur;
    for (int i = 0; i < 4; i++) planes[k++] = p->ref[i];
    if (bidir) for (int i = 0; i < 4; i++) planes[k++] = p->ref1[i];
    if (wbi) for (int i = 0; i < 4; i++) planes[k++] = p->ref_bi[i];
    uint8_t* dPlane[13];
    PlanePins pins(s);                    // released when this call returns: after the final synchronisation below
    for (int i = 0; i < nplanes; i++)
    {
        if (pkey[i])
        {
            void* cp = nullptr;
            rc = cached_plane((const uint8_t*)planes[i] - org, pkey[i], planeBytes, s, &cp);
            if (rc) return rc;
            pins.add(cp);
            dPlane[i] = (uint8_t*)cp;
        }
        else
        {
            if (check_hip(hipMemcpyAsync(d + oPlane[i], (const uint8_t*)planes[i] - org, planeBytes, hipMemcpyHostToDevice, s), "lowres_cost_host upload")) return X265HIP_ENODEV;
            dPlane[i] = d + oPlane[i];
        }
    }
    // a list that is not searched again keeps the mvs / costs it was given (estimateFrameCost's bDoSearch)
    int32_t* dMv[2] = { nullptr, nullptr };
    int32_t* dMc[2] = { nullptr, nullptr };
    for (int li = 0; li < NL; li++)
    {
        if (!p->do_search[li] && mvKey)
        {
            void *cv = nullptr, *cc = nullptr;
            rc = cached_plane(p->mvs[li], mvKey, (size_t)n * 8, s, &cv);
            if (rc) return rc;
            pins.add(cv);
            rc = cached_plane(p->mv_costs[li], mvKey, (size_t)n * 4, s, &cc);
            if (rc) return rc;
            pins.add(cc);
            dMv[li] = (int32_t*)cv; dMc[li] = (int32_t*)cc;
        }
        else
        {
            dMv[li] = (int32_t*)(d + oMv[li]); dMc[li] = (int32_t*)(d + oMc[li]);
            if (!p->do_search[li]) { memcpy(pinUp + (oMv[li] - oUp), p->mvs[li], (size_t)n * 8); memcpy(pinUp + (oMc[li] - oUp), p->mv_costs[li], (size_t)n * 4); }
        }
    }
    if (needCost) memcpy(pinUp + (oCost - oUp), p->cost_q - p->cost_q_half, costBytes);
    if (needIntra) memcpy(pinUp + (oIntra - oUp), p->intra_cost, (size_t)n * 4);
    if (p->inv_qscale) memcpy(pinUp + (oInvq - oUp), p->inv_qscale, (size_t)n * 4);

    x265hip_lowres_cost_pair pr;
    memset(&pr, 0, sizeof(pr));
    pr.cur = dPlane[0] + org;
    for (int i = 0; i < 4; i++) pr.ref[i] = dPlane[1 + i] + org;
    if (bidir) for (int i = 0; i < 4; i++) pr.ref1[i] = dPlane[5 + i] + org;
    if (wbi) for (int i = 0; i < 4; i++) pr.ref_bi[i] = dPlane[9 + i] + org;
    pr.intra_cost = needIntra ? (const int32_t*)(d + oIntra) : (const int32_t*)(d + oLc);          // (a B picture's kernels never read it)
    pr.inv_qscale = p->inv_qscale ? (const int32_t*)(d + oInvq) : nullptr;
    pr.mvs = dMv[0]; pr.mv_costs = dMc[0];
    pr.mvs1 = bidir ? dMv[1] : nullptr; pr.mv_costs1 = bidir ? dMc[1] : nullptr;
    pr.do_search[0] = p->do_search[0]; pr.do_search[1] = p->do_search[1];
    pr.lowres_costs = (uint16_t*)(d + oLc); pr.row_satds = (int32_t*)(d + oRows); pr.frame = (int64_t*)(d + oFrame);
    // the pair record travels in this thread's own scratch: no stream-ordered allocation per call
    memcpy(pinUp + (oPair - oUp), &pr, sizeof(pr));
    if (check_hip(hipMemcpyAsync(d + oUp, pinUp, upBytes, hipMemcpyHostToDevice, s), "lowres_cost_host upload")) return X265HIP_ENODEV;
    x265hip_lowres_cost_params q;
    memset(&q, 0, sizeof(q));
    q.depth = p->depth; q.stride = p->stride; q.width_in_cu = p->width_in_cu; q.height_in_cu = p->height_in_cu;
    q.cost_q = needCost ? (const uint16_t*)(d + oCost) : (const uint16_t*)(d + oLc);              // (never read without a search)
    q.qoff = p->cost_q_half; q.bframe_bias = p->bframe_bias;
    q.pairs = (const x265hip_lowres_cost_pair*)(d + oPair); q.npairs = 1;
    q.pairs_on_device = (bidir ? 2 : 1) | (!searches ? 4 : 0)                                    // | 4: no list is searched again - the dependency-free launch
                      | (p->do_search[0] ? 8 : 0) | ((bidir && p->do_search[1]) ? 16 : 0);        // | 8, | 16: the lists that are (a B estimate walks them side by side)
    rc = x265hip_lowres_cost(&q, s);
    if (rc) return rc;
    if (check_hip(hipMemcpyAsync(pinDown, d + oDown, downBytes, hipMemcpyDeviceToHost, s), "lowres_cost_host download")) return X265HIP_ENODEV;
    // the vectors just searched stay on the device for the estimates that reuse them
    void* keep[2][2] = { { nullptr, nullptr }, { nullptr, nullptr } };
    bool kept = true;
    if (mvKey)
        for (int li = 0; li < NL && kept; li++)
            if (p->do_search[li])
            {
                if (cached_reserve((size_t)n * 8, &keep[li][0]) || cached_reserve((size_t)n * 4, &keep[li][1])) { kept = false; break; }
                if (hipMemcpyAsync(keep[li][0], d + oMv[li], (size_t)n * 8, hipMemcpyDeviceToDevice, s) != hipSuccess ||
                    hipMemcpyAsync(keep[li][1], d + oMc[li], (size_t)n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) kept = false;
            }
    const hipError_t es = hipStreamSynchronize(s);
    if (es != hipSuccess || !kept)
    {
        for (int li = 0; li < 2; li++) { if (keep[li][0]) cached_abandon((size_t)n * 8, keep[li][0]); if (keep[li][1]) cached_abandon((size_t)n * 4, keep[li][1]); }
        (void)hipGetLastError();
        if (es != hipSuccess) X265HIP_TRY(es);
    }
    else
        for (int li = 0; li < NL; li++)
            if (keep[li][0]) { cached_adopt(p->mvs[li], mvKey, (size_t)n * 8, keep[li][0]); cached_adopt(p->mv_costs[li], mvKey, (size_t)n * 4, keep[li][1]); }
    memcpy(p->frame, pinDown + (oFrame - oDown), 32);
    memcpy(p->row_satds, pinDown + (oRows - oDown), (size_t)p->height_in_cu * 4);
    memcpy(p->lowres_costs, pinDown + (oLc - oDown), (size_t)n * 2);
    for (int li = 0; li < NL; li++)
        if (p->do_search[li]) { memcpy(p->mvs[li], pinDown + (oMv[li] - oDown), (size_t)n * 8); memcpy(p->mv_costs[li], pinDown + (oMc[li] - oDown), (size_t)n * 4); }
    return 0;
}

// The intra half of the lookahead behind host pointers: the per-block work of LookaheadTLD::lowresIntraEstimate (slicetype.cpp:696-772)
// for one picture whose lowres plane 0 lives in host memory; the AQ weighting and the row / frame sums of :779-803 stay with the caller
// (they read its invQscaleFactor arrays).  plane_key as in x265hip_lowres_cost_host: the upload is shared with the frame cost estimates.
extern "C" int x265hip_lowres_intra_host(const x265hip_lowres_intra_host_params* p)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !p->plane || !p->intra_cost || !p->intra_mode || !p->lowres_costs) { set_error("lowres_intra_host: NULL operand"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("lowres_intra_host: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->width_in_cu <= 0 || p->height_in_cu <= 0 || p->lines <= 0 || p->margin_x < 8 || p->margin_y < 8 || p->stride < p->width_in_cu * 8 + 2 * p->margin_x)
    { set_error("lowres_intra_host: geometry"); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    const int n = p->width_in_cu * p->height_in_cu;
    const size_t org = ((size_t)p->margin_y * p->stride + p->margin_x) * bpp;
    const size_t planeBytes = (size_t)p->stride * (p->lines + 2 * p->margin_y) * bpp;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = align256(off + bytes); return o; };
    const size_t oPlane = p->plane_key ? 0 : take(planeBytes + 64);
    const size_t oCost = take((size_t)n * 4), oMode = take((size_t)n), oLc = take((size_t)n * 2);
    LaThread& t = la_thread();
    rc = t.ensure(off);
    if (rc) return rc;
    hipStream_t s = t.stream;
    uint8_t* d = t.dev;
    uint8_t* dPlane = d + oPlane;
    PlanePins pins(s);
    if (p->plane_key)
    {
        void* cp = nullptr;
        rc = cached_plane((const uint8_t*)p->plane - org, p->plane_key, planeBytes, s, &cp);
        if (rc) return rc;
        pins.add(cp);
        dPlane = (uint8_t*)cp;
    }
    else
        X265HIP_TRY(hipMemcpyAsync(dPlane, (const uint8_t*)p->plane - org, planeBytes, hipMemcpyHostToDevice, s));
    x265hip_lowres_intra_params q;
    memset(&q, 0, sizeof(q));
    q.depth = p->depth; q.plane = dPlane + org; q.stride = p->stride; q.width_in_cu = p->width_in_cu; q.height_in_cu = p->height_in_cu;
    q.intra_penalty = p->intra_penalty;
    q.intra_cost = (int32_t*)(d + oCost); q.intra_mode = d + oMode; q.lowres_costs = (uint16_t*)(d + oLc);
    rc = x265hip_lowres_intra(&q, s);
    if (rc) return rc;
    X265HIP_TRY(hipMemcpyAsync(p->intra_cost, d + oCost, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    X265HIP_TRY(hipMemcpyAsync(p->intra_mode, d + oMode, (size_t)n, hipMemcpyDeviceToHost, s));
    X265HIP_TRY(hipMemcpyAsync(p->lowres_costs, d + oLc, (size_t)n * 2, hipMemcpyDeviceToHost, s));
    X265HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

// ---- LookaheadTLD::calcAdaptiveQuantFrame behind host pointers (encoder/slicetype.cpp:444-694) --------------------------------------
// The pixel work (acEnergyCu of every block: 1.5 samples read per luma sample) is one launch of x265hip_aq_energy over a device copy
// of the picture; what is left on the calling thread is the reference's own double-precision pass over the block energies.
extern "C" int x265hip_aq_frame_host(const x265hip_aq_frame_host_params* p)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !p->y || !p->qp_aq_offset || !p->qp_cutree_offset || !p->inv_qscale || !p->wp_sum || !p->wp_ssd) { set_error("aq_frame_host: NULL operand"); return X265HIP_EINVAL; }
    if ((p->cb == NULL) != (p->cr == NULL)) { set_error("aq_frame_host: cb and cr go together"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("aq_frame_host: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->qg_size != 16 && p->qg_size != 8) { set_error("aq_frame_host: qg_size %d", p->qg_size); return X265HIP_EINVAL; }
    if (p->aq_mode < 1 || p->aq_mode > 3 || !(p->aq_strength > 0)) { set_error("aq_frame_host: aq_mode %d strength %g (modes 1..3 with a strength)", p->aq_mode, p->aq_strength); return X265HIP_EINVAL; }
    if (p->width <= 0 || p->height <= 0 || p->stride < p->width || (p->cb && p->stride_c < p->width / 2)) { set_error("aq_frame_host: geometry"); return X265HIP_EINVAL; }
    if (p->inv_qscale_8x8 && (p->qg_size != 8 || p->width_in_cu <= 0 || p->height_in_cu <= 0)) { set_error("aq_frame_host: inv_qscale_8x8 goes with qg_size 8 and the lowres grid"); return X265HIP_EINVAL; }
    const int q = p->qg_size, bpp = p->depth == 8 ? 1 : 2;
    const int bw = (p->width + q - 1) / q, bh = (p->height + q - 1) / q, nblk = bw * bh;
    if (p->inv_qscale_8x8 && (2 * p->width_in_cu > bw || 2 * p->height_in_cu > bh)) { set_error("aq_frame_host: the lowres grid reaches past the qg-8 blocks"); return X265HIP_EINVAL; }
    // the blocks' footprint, and the chroma one (4:2:0: half of it)
    const size_t lw = (size_t)bw * q * bpp, lh = (size_t)bh * q, cw = lw / 2, ch = lh / 2;
    const size_t pitchY = align256(lw), pitchC = align256(cw);
    const size_t offCb = align256(pitchY * lh), offCr = offCb + (p->cb ? align256(pitchC * ch) : 0), offE = offCr + (p->cb ? align256(pitchC * ch) : 0);
    const size_t offWp = offE + align256((size_t)nblk * 4), total = offWp + 256;
    LaThread& t = la_thread();
    rc = t.ensure(total);
    if (rc) return rc;
    hipStream_t s = t.stream;
    X265HIP_TRY(hipMemcpy2DAsync(t.dev, pitchY, p->y, (size_t)p->stride * bpp, lw, lh, hipMemcpyHostToDevice, s));
    if (p->cb)
    {
        X265HIP_TRY(hipMemcpy2DAsync(t.dev + offCb, pitchC, p->cb, (size_t)p->stride_c * bpp, cw, ch, hipMemcpyHostToDevice, s));
        X265HIP_TRY(hipMemcpy2DAsync(t.dev + offCr, pitchC, p->cr, (size_t)p->stride_c * bpp, cw, ch, hipMemcpyHostToDevice, s));
    }
    x265hip_aq_energy_params e;
    memset(&e, 0, sizeof(e));
    e.depth = p->depth; e.y = t.dev; e.stride = (intptr_t)(pitchY / bpp);
    if (p->cb) { e.cb = t.dev + offCb; e.cr = t.dev + offCr; e.stride_c = (intptr_t)(pitchC / bpp); }
    e.width = p->width; e.height = p->height; e.qg_size = q;
    e.energy = (uint32_t*)(t.dev + offE); e.wp = (uint64_t*)(t.dev + offWp);
    rc = x265hip_aq_energy(&e, s);
    if (rc) return rc;
    std::vector<uint32_t> own;
    uint32_t* energy = p->energy;
    if (!energy) { own.resize(nblk); energy = own.data(); }
    uint64_t wp[6];
    X265HIP_TRY(hipMemcpyAsync(energy, t.dev + offE, (size_t)nblk * 4, hipMemcpyDeviceToHost, s));
    X265HIP_TRY(hipMemcpyAsync(wp, t.dev + offWp, sizeof(wp), hipMemcpyDeviceToHost, s));
    X265HIP_TRY(hipStreamSynchronize(s));
    x265hip_aq_offsets_params o;
    memset(&o, 0, sizeof(o));
    o.depth = p->depth; o.qg_size = q; o.aq_mode = p->aq_mode; o.aq_strength = p->aq_strength; o.nblocks = nblk;
    o.energy = energy; o.qp_aq_offset = p->qp_aq_offset; o.inv_qscale = p->inv_qscale;
    rc = x265hip_aq_offsets(&o);
    if (rc) return rc;
    if (p->qp_cutree_offset != p->qp_aq_offset) memcpy(p->qp_cutree_offset, p->qp_aq_offset, (size_t)nblk * sizeof(double));
    if (p->inv_qscale_8x8)                                     // slicetype.cpp:626-640: the four qg-8 factors under a lowres block, averaged
        for (int cy = 0; cy < p->height_in_cu; cy++)
            for (int cx = 0; cx < p->width_in_cu; cx++)
            {
                const int32_t* f = p->inv_qscale + cx * 2 + cy * p->width_in_cu * 4;
                p->inv_qscale_8x8[cx + cy * p->width_in_cu] = (f[0] + f[1] + f[bw] + f[bw + 1]) / 4;
            }
    // :662-675 (the reference rounds the picture to 16s here whatever the quantisation group)
    const int w16 = ((p->width + 8) >> 4) << 4, h16 = ((p->height + 8) >> 4) << 4;
    for (int i = 0; i < 3; i++)
    {
        const uint64_t sum = (i && !p->cb) ? 0 : wp[i], ssd = (i && !p->cb) ? 0 : wp[3 + i];
        p->wp_sum[i] = sum;
        if (p->normalise_wp)
        {
            const int wi = i ? w16 >> 1 : w16, hi = i ? h16 >> 1 : h16;          // int arithmetic as the reference's width[i] * height[i]
            p->wp_ssd[i] = ssd - (sum * sum + (wi * hi) / 2) / (wi * hi);
        }
        else
            p->wp_ssd[i] = ssd;
    }
    return 0;
}

// ---- weightAnalyse behind host pointers (encoder/weightPrediction.cpp:222-497) ---------------------------------------------------------
// Pixel work on the device - the motion-compensated copies (mcLuma / mcChroma) and weightCost for the unweighted plane plus every
// (scale, offset) pair the scan could visit, one launch per plane - and the reference's own decision logic replayed on the calling
// thread from the downloaded scores.
namespace {

struct WaMcLumaArgs { const uint8_t* plane[4]; uint8_t* out; long strideB; int width, lines; const int32_t* mvs; };

// mcLuma (:59-92): one thread per sample of an 8x8 block; Lowres::lowresMC (lowres.h:67-92) - a half-sample plane at the integer part of
// the clipped quarter-sample vector, or the rounded average of two such planes when either component is odd
template <typename Px>
__global__ void __launch_bounds__(256) wa_mc_luma_kernel(WaMcLumaArgs a)
{
    const int bw = a.width >> 3;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)bw * (a.lines >> 3) * 64;
    if (i >= total) return;
    const int cu = (int)(i >> 6), px = (int)(i & 7), py = (int)((i >> 3) & 7);
    const int by = cu / bw, bx = cu - by * bw, x = bx * 8, y = by * 8;
    const int mx = clip3((-x - 8) * 4, (a.width - x - 1 + 8) * 4, a.mvs[2 * cu]), my = clip3((-y - 8) * 4, (a.lines - y - 1 + 8) * 4, a.mvs[2 * cu + 1]);
    const long st = a.strideB / (long)sizeof(Px);
    const long at = (long)(y + py) * st + x + px;
    const int hpelA = (my & 2) | ((mx & 2) >> 1);
    int v = (int)reinterpret_cast<const Px*>(a.plane[hpelA])[at + (mx >> 2) + (long)(my >> 2) * st];
    if ((mx | my) & 1)
    {
        const int qx = mx + (mx & 1), qy = my + (my & 1);
        const int hpelB = (qy & 2) | ((qx & 2) >> 1);
        v = (v + (int)reinterpret_cast<const Px*>(a.plane[hpelB])[at + (qx >> 2) + (long)(qy >> 2) * st] + 1) >> 1;      // pixelavg_pp, pixel.cpp:385-397
    }
    reinterpret_cast<Px*>(a.out)[at] = (Px)v;
}

struct WaMcChromaArgs { const uint8_t* src; uint8_t* out; long strideB; int width, height, lowresWidthInCU, lowresHeightInCU, depth; const int32_t* mvs; };

__constant__ int8_t kWaChromaFilter[8][4] = { { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 }, { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };

// mcChroma (:96-166), 4:2:0: 8x8 chroma blocks.  The reference tests the block's SAMPLE position against the lowres CU counts (:121) and
// indexes the vectors with y * lowresWidthInCU + x / 8 (:113,118); the vector is the lowres one (mv << 1 >> 1), its integer part taken with
// >> 2 and its fraction with & 7 (:134-137) - all kept.  Filters: interp_horiz_pp_c / interp_vert_pp_c / interp_horiz_ps_c (row extension) +
// interp_vert_sp_c (ipfilter.cpp), 4 taps.
template <typename Px>
__global__ void __launch_bounds__(256) wa_mc_chroma_kernel(WaMcChromaArgs a)
{
    const int bw = a.width >> 3;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)bw * (a.height >> 3) * 64;
    if (i >= total) return;
    const int blk = (int)(i >> 6), px = (int)(i & 7), py = (int)((i >> 3) & 7);
    const int by = blk / bw, bx = blk - by * bw, x = bx * 8, y = by * 8;
    const long st = a.strideB / (long)sizeof(Px);
    const Px* src = reinterpret_cast<const Px*>(a.src);
    const long at = (long)(y + py) * st + x + px;
    int v;
    if (x < a.lowresWidthInCU && y < a.lowresHeightInCU)
    {
        const int cu = y * a.lowresWidthInCU + bx;
        const int mx = clip3((-x - 8) * 4, (a.width - x - 1 + 8) * 4, a.mvs[2 * cu]), my = clip3((-y - 8) * 4, (a.height - y - 1 + 8) * 4, a.mvs[2 * cu + 1]);
        const Px* t = src + at + (long)(my >> 2) * st + (mx >> 2);
        const int xf = mx & 7, yf = my & 7, maxVal = (1 << a.depth) - 1;
        if (!(xf | yf)) v = (int)t[0];
        else if (!yf)
        {
            int s = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) s += (int)t[k - 1] * kWaChromaFilter[xf][k];
            v = clip3(0, maxVal, (s + 32) >> 6);
        }
        else if (!xf)
        {
            int s = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) s += (int)t[(long)(k - 1) * st] * kWaChromaFilter[yf][k];
            v = clip3(0, maxVal, (s + 32) >> 6);
        }
        else
        {
            const int headRoom = 14 - a.depth, shiftH = 6 - headRoom, offH = -(1 << 13) << shiftH;
            const int shiftV = 6 + headRoom, offV = (1 << (shiftV - 1)) + ((1 << 13) << 6);
            int s = 0;
#pragma unroll
            for (int r = 0; r < 4; r++)
            {
                int h = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) h += (int)t[(long)(r - 1) * st + k - 1] * kWaChromaFilter[xf][k];
                s += (int)(int16_t)((h + offH) >> shiftH) * kWaChromaFilter[yf][r];
            }
            v = clip3(0, maxVal, (s + offV) >> shiftV);
        }
    }
    else
        v = (int)src[at];
    reinterpret_cast<Px*>(a.out)[at] = (Px)v;
}

struct WaCostArgs { const uint8_t* fenc; const uint8_t* ref; long strideB; int width, height, depth; const int32_t* intraCost; const int32_t* cand; uint32_t* cost; };

// weightCost (:172-217): one thread per 8x8 block, blockIdx.y = candidate { present, scale, denom, offset }; weight_pp's arithmetic per
// sample (pixel.cpp:518-543), four 4x4 Hadamards (satd 8x8), luma blocks capped by the intra cost; uint32 sums wrap as the reference's
template <typename Px>
__global__ void __launch_bounds__(256) wa_cost_kernel(WaCostArgs a)
{
    const int bw = a.width >> 3, nblk = bw * (a.height >> 3);
    const int mb = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    uint32_t v = 0;
    if (mb < nblk)
    {
        const int by = mb / bw, bx = mb - by * bw;
        const int present = a.cand[4 * c], scale = a.cand[4 * c + 1], denom = a.cand[4 * c + 2];
        const int correction = 14 - a.depth, maxVal = (1 << a.depth) - 1;
        const int offset = a.cand[4 * c + 3] << (a.depth - 8), round = (denom ? 1 << (denom - 1) : 0) << correction, shift = denom + correction;
        const long st = a.strideB / (long)sizeof(Px);
        const Px* f = reinterpret_cast<const Px*>(a.fenc) + (long)(by * 8) * st + bx * 8;
        const Px* r = reinterpret_cast<const Px*>(a.ref) + (long)(by * 8) * st + bx * 8;
        int satd = 0;
#pragma unroll
        for (int t = 0; t < 4; t++)
        {
            int d[4][4];
            const int ox = (t & 1) * 4, oy = (t >> 1) * 4;
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++)
                {
                    int rv = (int)r[(oy + y) * st + ox + x];
                    if (present) rv = clip3(0, maxVal, ((scale * (int)(int16_t)(rv << correction) + round) >> shift) + offset);
                    d[y][x] = rv - (int)f[(oy + y) * st + ox + x];
                }
            satd += x265hip::tile_satd4(d);
        }
        v = (uint32_t)(a.intraCost ? min(satd, a.intraCost[mb]) : satd);
    }
    v = (uint32_t)group_sum<64>((int)v);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&a.cost[c], v);
}

// ---- the decision half of x265hip_weight_analyse_host --------------------------------------------------------------------------------------
// What the reference decides in weightAnalyse (encoder/weightPrediction.cpp:248-492), organised for a device that scores every trial of a
// plane in ONE launch: the brightness statistics of a plane pair (:268-277) -> a seed gain (:300-330) -> a table of trials around the seed
// (:403-446 visits them one weightCost call at a time; here the table is built first, scored as a batch, then walked in the reference's
// order) -> the verdict (:448-470).  The float expressions are the reference's, operation for operation - the result must be bit-identical.
struct Gain { int on, mul, log2d, add; };                  // sample' = clip(((mul * sample + round) >> log2d) + add); on = signalled in the slice header
inline Gain unity(int log2d) { return { 0, 1 << log2d, log2d, 0 }; }
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct Brightness                                           // of one colour plane of a (picture, reference) pair
{
    float ratio, meanCur, meanRef;                          // sqrt of the AC-energy ratio; mean sample values at 8-bit scale
    void measure(uint64_t ssdCur, uint64_t ssdRef, uint64_t sumCur, uint64_t sumRef, int samples, int depth)
    {
        const uint64_t guard = !ssdRef;                     // :272-273: both energies get + 1 when the REFERENCE's is zero
        ratio = std::sqrt((float)(ssdCur + guard) / (ssdRef + guard));
        meanCur = (float)sumCur / (samples) / (1 << (depth - 8));
        meanRef = (float)sumRef / (samples) / (1 << (depth - 8));
    }
    bool flat() const { return std::fabs(meanRef - meanCur) < 0.5f && std::fabs(1.f - ratio) < 1.f / 128.f; }      // :303-309: nothing to gain
};

// Exp-Golomb code lengths (common/bitstream.h:94-112) and the slice-header price of signalling a gain (weightPrediction.cpp:49-56)
inline int ue_bits(unsigned v) { if (!v) return 1; int n = 0; while (v >> (n + 1)) n++; return 2 * n + 1; }
inline int se_bits(int v) { int t = 1 - v * 2; if (t < 0) t = v * 2; return t < 256 ? ue_bits((unsigned)t) : ue_bits((unsigned)t >> 8) + 16; }
inline int header_price(const Gain& g, int lambda, bool chroma)
{
    return (chroma ? 4 * lambda : lambda) * (10 + ue_bits((unsigned)g.log2d + 1) * (chroma ? 1 : 2) + 2 * (se_bits(g.mul) + se_bits(g.add)));
}

// The trials around a seed multiplier: one ROW per multiplier within +-4 of the seed whose distance from unity fits the header's signed byte;
// a row carries the five offsets around the offset that multiplier implies for the two means (re-deriving the multiplier when that offset
// does not fit a signed byte, :421-431).  Distinct (mul, add) pairs get one slot each in the launch's candidate list; slot 0 is the
// unweighted plane.
struct TrialTable
{
    struct Row { int mul, addLo, addHi; };
    Row row[16];
    int nrows = 0, ncand = 1, log2d = 0;
    int32_t cand[64][4];                                    // { weighted?, mul, log2d, add } per slot: what wa_cost_kernel reads
    int slot(int mul, int add)
    {
        for (int i = 1; i < ncand; i++) if (cand[i][1] == mul && cand[i][3] == add) return i;
        cand[ncand][0] = 1; cand[ncand][1] = mul; cand[ncand][2] = log2d; cand[ncand][3] = add;
        return ncand++;
    }
    void build(int seedMul, int d, const Brightness& b)
    {
        log2d = d;
        cand[0][0] = 0; cand[0][1] = 1; cand[0][2] = 0; cand[0][3] = 0;
        for (int m0 = clampi(seedMul - 4, 0, 127); m0 <= clampi(seedMul + 4, 0, 127); m0++)
        {
            const int fromUnity = m0 - (1 << d);
            if (fromUnity > 127 || fromUnity <= -128) continue;
            int m = m0;
            int a = (int)(b.meanCur - b.meanRef * m / (1 << d) + 0.5f);
            if (a < -128 || a > 127)
            {
                a = clampi(a, -128, 127);
                m = clampi((int)((1 << d) * (b.meanCur - a) / b.meanRef + 0.5f), 0, 127);
            }
            Row& r = row[nrows++];
            r.mul = m; r.addLo = clampi(a - 2, -128, 127); r.addHi = clampi(a + 2, -128, 127);
            for (int k = r.addLo; k <= r.addHi; k++) slot(m, k);
        }
    }
    // The reference's walk over the scored table: rows in order, offsets ascending, strict improvement only; a row is left as soon as the
    // best offset so far is its first one and a later one has been looked at (:440-441 - the offsets are unimodal in practice).
    bool walk(const uint32_t* score, int lambda, bool chroma, int& mul, int& add, uint32_t& best)
    {
        bool improved = false;
        for (int k = 0; k < nrows; k++)
            for (int a = row[k].addLo; a <= row[k].addHi; a++)
            {
                const Gain g = { 1, row[k].mul, log2d, a };
                const uint32_t sc = score[slot(row[k].mul, a)] + (uint32_t)header_price(g, lambda, chroma);
                if (sc < best) { best = sc; mul = row[k].mul; add = a; improved = true; }
                if (add == row[k].addLo && a != row[k].addLo) break;
            }
        return improved;
    }
};

} // namespace

extern "C" int x265hip_weight_analyse_host(const x265hip_weight_analyse_host_params* p)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !p->lowres || !p->cb || !p->cr || !p->intra_cost || !p->weights || !p->denoms) { set_error("weight_analyse_host: NULL operand (4:2:0 pictures only)"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("weight_analyse_host: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->nlists < 1 || p->nlists > 2) { set_error("weight_analyse_host: nlists %d", p->nlists); return X265HIP_EINVAL; }
    if (p->lowres_width <= 0 || p->lowres_lines <= 0 || (p->lowres_width & 7) || (p->lowres_lines & 7) || p->lowres_margin_x < 16 || p->lowres_margin_y < 16 ||
        p->lowres_stride < p->lowres_width + 2 * p->lowres_margin_x)
    { set_error("weight_analyse_host: lowres geometry %d x %d, margins %d / %d, stride %ld (multiples of 8, margins >= 16)", p->lowres_width, p->lowres_lines, p->lowres_margin_x, p->lowres_margin_y, (long)p->lowres_stride); return X265HIP_EINVAL; }
    const int cw = ((p->pic_width >> 4) << 4) >> 1, chh = ((p->pic_height >> 4) << 4) >> 1;          // :377-378: the chroma area weightCost measures
    if (p->pic_width < 16 || p->pic_height < 16 || p->margin_xc < 24 || p->margin_yc < 24 || p->stride_c < cw + 2 * p->margin_xc)
    { set_error("weight_analyse_host: picture %d x %d, chroma margins %d / %d, stride %ld (margins >= 24)", p->pic_width, p->pic_height, p->margin_xc, p->margin_yc, (long)p->stride_c); return X265HIP_EINVAL; }
    for (int l = 0; l < p->nlists; l++)
    {
        for (int k = 0; k < 4; k++) if (!p->ref[l].lowres[k]) { set_error("weight_analyse_host: NULL lowres plane %d of list %d", k, l); return X265HIP_EINVAL; }
        if (!p->ref[l].cb || !p->ref[l].cr) { set_error("weight_analyse_host: NULL chroma plane of list %d", l); return X265HIP_EINVAL; }
    }
    const int bpp = p->depth == 8 ? 1 : 2;
    const int lw = p->lowres_width, lh = p->lowres_lines, n = (lw >> 3) * (lh >> 3);
    const size_t lorg = ((size_t)p->lowres_margin_y * p->lowres_stride + p->lowres_margin_x) * bpp;
    const size_t lplaneBytes = (size_t)p->lowres_stride * (lh + 2 * p->lowres_margin_y) * bpp;
    const int CM = 24;                                                                                 // chroma rows / samples around the measured area that compensation may read
    const size_t corg = ((size_t)CM * p->stride_c + CM) * bpp;
    const size_t cplaneBytes = ((size_t)p->stride_c * (chh + 2 * CM) + 2 * CM) * bpp;
    const size_t mcBytes = std::max((size_t)p->lowres_stride * lh, (size_t)p->stride_c * chh) * bpp;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = align256(off + bytes); return o; };
    const size_t oCur = p->plane_key ? 0 : take(lplaneBytes + 64);
    size_t oRef[2][4] = {}, oRefC[2][2] = {}, oMvs[2] = {};
    for (int l = 0; l < p->nlists; l++)
    {
        for (int k = 0; k < 4; k++) oRef[l][k] = p->ref[l].plane_key ? 0 : take(lplaneBytes + 64);
        oRefC[l][0] = take(cplaneBytes + 64); oRefC[l][1] = take(cplaneBytes + 64);
        oMvs[l] = take((size_t)n * 8);
    }
    const size_t oCb = take(cplaneBytes), oCr = take(cplaneBytes), oIntra = take((size_t)n * 4), oMc = take(mcBytes + 64);
    const size_t oCand = take(64 * 16), oCost = take(64 * 4);
    LaThread& t = la_thread();
    rc = t.ensure(off);
    if (rc) return rc;
    hipStream_t s = t.stream;
    uint8_t* d = t.dev;
    auto up = [&](size_t o, const void* src, size_t bytes) { return check_hip(hipMemcpyAsync(d + o, src, bytes, hipMemcpyHostToDevice, s), "weight_analyse_host upload"); };
    PlanePins pins(s);
    auto lowres_plane = [&](const void* host, uint64_t key, size_t o, uint8_t** out) -> int
    {
        if (key)
        {
            void* cp = nullptr;
            const int r = cached_plane((const uint8_t*)host - lorg, key, lplaneBytes, s, &cp);
            if (r) return r;
            pins.add(cp);
            *out = (uint8_t*)cp + lorg;
            return 0;
        }
        if (up(o, (const uint8_t*)host - lorg, lplaneBytes)) return X265HIP_ENODEV;
        *out = d + o + lorg;
        return 0;
    };
    // nothing is uploaded before a plane gets past the early exits (a picture without a brightness change costs no transfer at all)
    uint8_t* dCur = nullptr;
    bool curChromaUp[2] = { false, false };

    const int lookaheadLambda[3] = { 1, 16, 256 };                 // (int)x265_lambda_tab[X265_LOOKAHEAD_QP] at 8 / 10 / 12 bits (constants.cpp)
    const int lambda = lookaheadLambda[(p->depth - 8) / 2];
    const int lumaSamples = (((p->pic_width + 15) >> 4) << 4) * (((p->pic_height + 15) >> 4) << 4);
    const int samples[3] = { lumaSamples, lumaSamples >> 2, lumaSamples >> 2 };
    int log2dLuma = 7, log2dChroma = 7;                            // the denominators carry over from list 0 to list 1 (:253, :487-488)
    memset(p->weights, 0, 2 * 3 * 4 * sizeof(int32_t));
    memset(p->denoms, 0, 4 * sizeof(int32_t));

    for (int list = 0; list < p->nlists; list++)
    {
        const x265hip_weight_analyse_ref& R = p->ref[list];
        Brightness b[3];
        for (int c = 0; c < 3; c++) b[c].measure(p->wp_ssd[c], R.wp_ssd[c], p->wp_sum[c], R.wp_sum[c], samples[c], p->depth);
        if (!list)      // the chroma denominator both chroma ratios fit 7 bits with (:280-286)
            while (log2dChroma > 0 && !(b[1].ratio < 127.f / (1 << log2dChroma) && b[2].ratio < 127.f / (1 << log2dChroma))) log2dChroma--;
        Gain g[3] = { { 0, 1, 0, 0 }, unity(log2dChroma), unity(log2dChroma) };

        bool refOnDevice = false;
        uint8_t* dRef[4] = {};
        const int32_t* mvs = nullptr;

        // the plane pair of colour plane c on the device - the reference's plane motion-compensated with the lookahead's vectors when it has them
        // (:336-400) - as (source, prediction, byte pitch, measured width / height)
        struct PlanePair { const uint8_t* cur; const uint8_t* pred; long pitchB; int w, h; };
        auto stage = [&](int c, PlanePair& pp) -> int
        {
            if (!c)
            {
                if (!dCur)
                {
                    int r = lowres_plane(p->lowres, p->plane_key, oCur, &dCur);
                    if (r) return r;
                    if (up(oIntra, p->intra_cost, (size_t)n * 4)) return X265HIP_ENODEV;
                }
                if (!refOnDevice)
                {
                    for (int k = 0; k < (mvs ? 4 : 1); k++) { int r = lowres_plane(R.lowres[k], R.plane_key, oRef[list][k], &dRef[k]); if (r) return r; }
                    if (mvs && up(oMvs[list], mvs, (size_t)n * 8)) return X265HIP_ENODEV;
                    refOnDevice = true;
                }
                pp = { dCur, dRef[0], (long)p->lowres_stride * bpp, lw, lh };
                if (mvs)
                {
                    WaMcLumaArgs m;
                    for (int k = 0; k < 4; k++) m.plane[k] = dRef[k];
                    m.out = d + oMc; m.strideB = pp.pitchB; m.width = lw; m.lines = lh; m.mvs = (const int32_t*)(d + oMvs[list]);
                    const unsigned grid = (unsigned)(((long)n * 64 + 255) / 256);
                    if (bpp == 1) hipLaunchKernelGGL(wa_mc_luma_kernel<uint8_t>, dim3(grid), dim3(256), 0, s, m);
                    else hipLaunchKernelGGL(wa_mc_luma_kernel<uint16_t>, dim3(grid), dim3(256), 0, s, m);
                    pp.pred = d + oMc;
                }
                return 0;
            }
            const size_t o = oRefC[list][c - 1];
            if (up(o, (const uint8_t*)(c == 1 ? R.cb : R.cr) - corg, cplaneBytes)) return X265HIP_ENODEV;
            if (!curChromaUp[c - 1])                                // the current picture's chroma: only the measured area is read
            {
                if (up(c == 1 ? oCb : oCr, c == 1 ? p->cb : p->cr, (size_t)p->stride_c * chh * bpp)) return X265HIP_ENODEV;
                curChromaUp[c - 1] = true;
            }
            pp = { d + (c == 1 ? oCb : oCr), d + o + corg, (long)p->stride_c * bpp, cw, chh };
            if (mvs)
            {
                WaMcChromaArgs m;
                m.src = pp.pred; m.out = d + oMc; m.strideB = pp.pitchB; m.width = cw; m.height = chh; m.lowresWidthInCU = lw >> 3; m.lowresHeightInCU = lh >> 3;
                m.depth = p->depth; m.mvs = (const int32_t*)(d + oMvs[list]);
                const unsigned grid = (unsigned)(((long)(cw >> 3) * (chh >> 3) * 64 + 255) / 256);
                if (bpp == 1) hipLaunchKernelGGL(wa_mc_chroma_kernel<uint8_t>, dim3(grid), dim3(256), 0, s, m);
                else hipLaunchKernelGGL(wa_mc_chroma_kernel<uint16_t>, dim3(grid), dim3(256), 0, s, m);
                pp.pred = d + oMc;
            }
            return 0;
        };

        for (int c = 0; c < 3; c++)
        {
            if (c && !g[0].on) break;                               // chroma is only weighted beside a weighted luma (:296-298)
            const int log2dPlane = c ? log2dChroma : log2dLuma;
            if (b[c].flat()) { g[c] = unity(log2dPlane); continue; }

            // the seed: the energy ratio as a multiplier (:311-330)
            int seedMul, seedLog2d = log2dPlane;
            if (c)
            {
                seedMul = clampi((int)(b[c].ratio * (1 << log2dPlane) + 0.5f), 0, 255);
                if (seedMul > 127) continue;                        // does not fit the header: the plane keeps unity
                g[c].mul = seedMul;                                 // (what a later copy between the chroma planes would carry, :473-480)
            }
            else
            {
                seedMul = (int)(b[0].ratio * (1 << log2dPlane) + 0.5f);                     // WeightParam::setFromWeightAndOffset, slice.h:304-316
                while (!list && seedLog2d > 0 && seedMul > 127) { seedLog2d--; seedMul >>= 1; }
                seedMul = seedMul < 127 ? seedMul : 127;
                g[0] = { g[0].on, seedMul, seedLog2d, 0 };
                mvs = R.mvs;                                        // from the luma plane on, the list's lookahead vectors compensate the reference (:332-334)
            }

            PlanePair pp;
            rc = stage(c, pp);
            if (rc) return rc;

            // every trial the walk could visit + the unweighted plane, scored by one launch (weightCost, :172-217, per candidate)
            TrialTable trials;
            trials.build(seedMul, seedLog2d, b[c]);
            if (up(oCand, trials.cand, (size_t)trials.ncand * 16)) return X265HIP_ENODEV;
            X265HIP_TRY(hipMemsetAsync(d + oCost, 0, 64 * 4, s));
            WaCostArgs ca;
            ca.fenc = pp.cur; ca.ref = pp.pred; ca.strideB = pp.pitchB; ca.width = pp.w; ca.height = pp.h; ca.depth = p->depth;
            ca.intraCost = c ? nullptr : (const int32_t*)(d + oIntra); ca.cand = (const int32_t*)(d + oCand); ca.cost = (uint32_t*)(d + oCost);
            const dim3 grid((unsigned)(((pp.w >> 3) * (pp.h >> 3) + 255) / 256), (unsigned)trials.ncand);
            if (bpp == 1) hipLaunchKernelGGL(wa_cost_kernel<uint8_t>, grid, dim3(256), 0, s, ca);
            else hipLaunchKernelGGL(wa_cost_kernel<uint16_t>, grid, dim3(256), 0, s, ca);
            X265HIP_TRY(hipGetLastError());
            uint32_t score[64];
            X265HIP_TRY(hipMemcpyAsync(score, d + oCost, (size_t)trials.ncand * 4, hipMemcpyDeviceToHost, s));
            X265HIP_TRY(hipStreamSynchronize(s));

            // the verdict (:448-470)
            const uint32_t plain = score[0];
            if (!plain) { g[c] = unity(log2dPlane); continue; }     // a perfect prediction already
            int mul = seedMul, add = 0, log2d = seedLog2d;
            uint32_t best = plain;
            const bool improved = trials.walk(score, lambda, c != 0, mul, add, best);
            if (!c && !list && log2d > 0 && !(mul & 1))
            {   // list 0's luma gain in lowest terms: common factors of two leave the multiplier and the denominator (:452-460)
                const int twos = mul ? __builtin_ctz((unsigned)mul) : 32;
                const int drop = twos < log2d ? twos : log2d;
                log2d -= drop;
                mul >>= drop;
            }
            const bool worthIt = improved && !(mul == (1 << log2d) && add == 0) && !((float)best / plain > 0.998f);
            g[c] = worthIt ? Gain{ 1, mul, log2d, add } : unity(log2dPlane);
        }
        if (g[0].on && g[1].on != g[2].on)                          // 4:2:0 signals the two chroma planes together (:472-481)
        {
            if (g[1].on) g[2] = g[1];
            else g[1] = g[2];
        }
        log2dLuma = g[0].log2d;
        log2dChroma = g[1].log2d;
        for (int c = 0; c < 3; c++)
        {
            int32_t* o = p->weights + (list * 3 + c) * 4;
            o[0] = g[c].on; o[1] = g[c].mul; o[2] = g[c].log2d; o[3] = g[c].add;
        }
        p->denoms[list * 2] = log2dLuma; p->denoms[list * 2 + 1] = log2dChroma;
    }
    return 0;
}
