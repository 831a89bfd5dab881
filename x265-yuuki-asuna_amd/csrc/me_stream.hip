// me_stream.hip - the ROW-GRANULAR consumer of the exhaustive search, for hosts that encode several pictures at once.
//
// x265hip_me_cache (csrc/me_cache.hip) takes whole pictures: a reference picture must be complete when the next picture starts, which
// is --frame-threads 1.  The reference's own parallelism is frame threads (5 on 16 cores): picture k + 1 starts while picture k is
// still being reconstructed, CTU row by CTU row, each row published through Frame::m_reconRowFlag (encoder/framefilter.cpp:664) and
// awaited row by row by the consumers (encoder/frameencoder.cpp:852-868, m_refLagRows :161-164).  This service follows that protocol:
//
//   * a host thread hands over CTU ROWS of a picture as they become final (x265hip_me_stream_picture_rows: the rows are copied into
//     pinned staging inside the call - the hook sits right after m_reconRowFlag[row].set(1)); a source picture arrives in one piece;
//   * a (source, reference) pair is OPENED when the first search of that pair is about to run (x265hip_me_stream_pair_open);
//   * the worker thread uploads rows as they arrive and searches every CTU row of every open pair as soon as the rows of the
//     reference its window reaches (row + (63 + range) / 64) are on the device - bands of up to `band_rows` CTU rows per launch of
//     x265hip_me_fullsearch - then downloads the band's SAD surfaces on a copy stream and raises one flag per (pair, CTU row).
//     The window needs 1 row below the CTU row at merange <= 64; the host itself waits for 3 (motion.cpp's sub-pel lag), so the
//     device runs two reference rows ahead of the first host lookup of a row.
//   * min_level = 1 keeps only the 16x16 / 32x32 / 64x64 levels of a record (720 -> 208 bytes packed, 1360 -> 336 bytes int32): a
//     compaction kernel between search and download.  An 8x8 SAD on cached pixels costs a host less than a cache-missing lookup;
//     with the 8x8 level gone a 4K pair at +-32 is 0.47 GB of pinned host memory instead of 1.6 GB.
//
//   * layout = X265HIP_STREAM_PLANES (round 4): what lands in host memory is PU-MAJOR - per CTU, one (2R+1) x pitch raster of SADs per
//     square PU (uint16 for 8x8 / 16x16, saturating; uint32 for 32x32 / 64x64) instead of records that interleave all PUs per
//     displacement.  A search walks ONE PU over neighbouring displacements: in the record layout every probe is a cache and TLB miss of
//     its own (1 us per lookup measured inside the real encoder, profiles/r04_encoder_family_profile_seams.txt), in a PU's own raster the
//     probes of one search share a few lines of a 2 - 5 KB plane.  The transposition runs on the device between search and download.
//   * centre_range > 0: every CTU's window is centred on the displacement an exhaustive minima-only search of +-centre_range finds for
//     its 64x64 block (clamped so the window stays inside the margins): a +-R window around where the picture moved covers what a
//     +-24 window around (0, 0) covered with a fraction of the bytes.  centres[ctu] travels with the row's flag.
//     (Surfaces stay in hipHostMalloc memory: aligned_alloc + MADV_HUGEPAGE + hipHostRegister measured no faster per lookup -
//     tools/ubench/host_lookup.hip, profiles/r04_host_lookup_ubench.txt - and two GPU-suite runs with registered user memory ended in
//     values that did not verify; the runs before and after, on hipHostMalloc, are clean.)
//
// Readers never wait and never lock: surface rows are valid when ready[row] == the pair's generation, checked BEFORE and AFTER the
// read (a slot that was reopened in between has its flags cleared before any of its rows can be rewritten).  Everything a lookup
// cannot serve is answered by the host's own primitive with identical values, so the bitstream cannot change.
#include "common.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include <cstdlib>

using namespace x265hip;

namespace {

// records of `rec16` 16-byte chunks -> their last `tail16` chunks, contiguous
__global__ void surf_tail_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, size_t nrec, int rec16, int tail16)
{
    const size_t n = nrec * (size_t)tail16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    {
        const size_t r = i / (size_t)tail16;
        const int c = (int)(i - r * (size_t)tail16);
        out[i] = in[r * (size_t)rec16 + (size_t)(rec16 - tail16 + c)];
    }
}

// records -> PU-major planes (X265HIP_STREAM_PLANES).  in: the band's records [ctu][row][group] of recBytes (packed 720: u16 [80][4] +
// i32 [5][4]; int32 1360: i32 [85][4]); out: per CTU (ctuBytes apart, the band's first CTU at out) the planes of PUs pu0 .. 84 in
// order, each nc rows of `pitch` = 4 * ng entries: uint16 (saturating) below PU 80, uint32 from there.  One thread moves the 4
// displacements of one (CTU, PU, row, group); the group index runs fastest so that the stores of a plane row are contiguous.
__global__ void __launch_bounds__(256) surf_planes_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int nctu, int nc, int ng, int recBytes, int packed, int pu0,
                                                          size_t ctuBytes)
{
    const int npu = 85 - pu0, pitch = 4 * ng;
    const size_t total = (size_t)nctu * nc * npu * ng;
    const size_t planeS = (size_t)nc * pitch * 2, planeW = (size_t)nc * pitch * 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    {
        const int g = (int)(i % ng);
        size_t t = i / ng;
        const int pu = pu0 + (int)(t % npu); t /= npu;
        const int row = (int)(t % nc);
        const int ctu = (int)(t / nc);
        const uint8_t* rec = in + (((size_t)ctu * nc + row) * ng + g) * recBytes;
        uint32_t v[4];
        if (packed && pu < 80) { const ushort4 q = *reinterpret_cast<const ushort4*>(rec + pu * 8); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
        else
        {
            const uint4 q = *reinterpret_cast<const uint4*>(rec + (packed ? 640 + (pu - 80) * 16 : pu * 16));
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        }
        uint8_t* base = out + (size_t)ctu * ctuBytes;
        if (pu < 80)
        {
            ushort4 o;
            o.x = (unsigned short)min(v[0], 65535u); o.y = (unsigned short)min(v[1], 65535u); o.z = (unsigned short)min(v[2], 65535u); o.w = (unsigned short)min(v[3], 65535u);
            *reinterpret_cast<ushort4*>(base + (size_t)(pu - pu0) * planeS + ((size_t)row * pitch + 4 * g) * 2) = o;
        }
        else
            *reinterpret_cast<uint4*>(base + (size_t)(80 - pu0) * planeS + (size_t)(pu - 80) * planeW + ((size_t)row * pitch + 4 * g) * 4) = make_uint4(v[0], v[1], v[2], v[3]);
    }
}

// centre of every CTU's window = the displacement of its 64x64 block's minimum SAD in the +-big search, clamped to +-maxX / +-maxY
__global__ void centre_kernel(const unsigned long long* __restrict__ best, int16_t* __restrict__ centres, int nctu, int big, int maxX, int maxY)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nctu) return;
    const uint32_t idx = (uint32_t)best[(size_t)i * 85 + 84];
    const int ncb = 2 * big + 1;
    const int mx = (int)(idx % ncb) - big, my = (int)(idx / ncb) - big;
    centres[2 * i] = (int16_t)clip3(-maxX, maxX, mx);
    centres[2 * i + 1] = (int16_t)clip3(-maxY, maxY, my);
}

// primitives.weight_pp (common/pixel.cpp:518-543) over whole buffer lines, margins included: the plane MotionReference::applyWeight
// builds row by row (encoder/reference.cpp:119-178: weight_pp on the picture, then the borders replicated) is the reconstructed
// plane weighted sample by sample - a replicated border sample weights to the replicated weighted sample.
template <typename Px>
__global__ void __launch_bounds__(256) weight_lines_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t ndw, int w0, int round, int shift, int offset,
                                                           int correction, int maxVal)
{
    constexpr int PER = 4 / (int)sizeof(Px), BITS = 8 * (int)sizeof(Px);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ndw; i += (size_t)gridDim.x * blockDim.x)
    {
        const uint32_t v = src[i];
        uint32_t o = 0;
#pragma unroll
        for (int k = 0; k < PER; k++)
        {
            const int px = (int)((v >> (k * BITS)) & ((1u << BITS) - 1));
            const int val = (int)(int16_t)(px << correction);                       // "simulating pixel to short conversion" (pixel.cpp:535)
            o |= (uint32_t)clip3(0, maxVal, ((w0 * val + round) >> shift) + offset) << (k * BITS);
        }
        dst[i] = o;
    }
}

double ms_now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

enum { ROW_NONE = 0, ROW_STAGED = 1, ROW_ON_DEVICE = 2 };

} // namespace

struct x265hip_me_stream
{
    x265hip_me_stream_params prm;
    int bpp, ctusW, ctusH, nc, ng, fullRec, rec, device, bandRows, lagRows, planes, centreRange, maxCx, maxCy;
    size_t planeBytes, rowBytes, fullRowBytes, surfBytes, linePitch, ctuBytes;
    hipStream_t compute = nullptr, copy = nullptr;
    uint8_t* dScratch = nullptr;            // full records of one band (min_level > 0 or planes)
    unsigned long long* dBest = nullptr;    // centre_range: minima of one band's +-centre_range search
    uint16_t* dZeroCost = nullptr;          // ... searched on SAD alone
    struct Pic
    {
        uint64_t key = 0; bool used = false; uint32_t epoch = 0; uint64_t stamp = 0; int busy = 0;
        uint8_t* stage = nullptr;           // pinned copy, rows staged by the host threads
        uint8_t* dev = nullptr;
        std::vector<uint8_t> rows;          // per CTU row: ROW_*
        // a DERIVED picture = picture `parent` weighted on the device (x265hip_me_stream_pair_open_weighted): no key, no staging use
        bool derived = false; int parent = -1; uint32_t parentEpoch = 0; x265hip_weight w = { 0, 0, 0, 0 };
    };
    struct Slot
    {
        uint8_t* surf = nullptr; uint8_t* dSurf = nullptr;
        int16_t* centres = nullptr; int16_t* dCentres = nullptr;      // [ctu][2] (centre_range > 0)
        std::atomic<int>* ready = nullptr;
        int generation = 0;
        int fenc = -1, ref = -1; uint32_t fencEpoch = 0, refEpoch = 0;
        int nextRow = 0; bool active = false;
        hipEvent_t evSearched = nullptr, evDown = nullptr;
    };
    std::vector<Pic> pics;
    std::vector<Slot> slots;
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false, dirty = false;
    uint64_t clock = 0;
    std::thread worker;
    std::atomic<uint64_t> bands{0}, rowsSearched{0}, rowsUploaded{0}, pairsOpened{0}, pairsCompleted{0}, failed{0}, noPicture{0};
    std::atomic<uint64_t> usBusy{0}, bytesDown{0}, bytesUp{0}, rowsWeighted{0}, weightedPairs{0};
    char workerError[256] = "";
};

namespace {

typedef x265hip_me_stream S;

// first / one-past-last buffer line of CTU rows [r0, r0 + n): the margins travel with the first / last row
inline void row_lines(const S* s, int r0, int n, long& y0, long& y1)
{
    y0 = r0 == 0 ? 0 : s->prm.margin_y + (long)r0 * 64;
    y1 = r0 + n == s->ctusH ? (long)s->prm.height + 2 * s->prm.margin_y : s->prm.margin_y + (long)(r0 + n) * 64;
}

struct Band { int slot, gen, r0, r1, fenc, ref; };
struct Upload { int pic, r0, n; };
struct Weigh { int pic, parent, r0, n; };

int run_round(S* s, const std::vector<Upload>& ups, const std::vector<Weigh>& weighs, const std::vector<Band>& bands)
{
    X265HIP_TRY(hipSetDevice(s->device));
    apply_wait_policy(s->device);
    for (const Upload& u : ups)
    {
        long y0, y1;
        row_lines(s, u.r0, u.n, y0, y1);
        const size_t off = (size_t)y0 * s->linePitch, bytes = (size_t)(y1 - y0) * s->linePitch;
        X265HIP_TRY(hipMemcpyAsync(s->pics[u.pic].dev + off, s->pics[u.pic].stage + off, bytes, hipMemcpyHostToDevice, s->compute));
        s->bytesUp += bytes; s->rowsUploaded += u.n;
    }
    for (const Weigh& q : weighs)
    {
        long y0, y1;
        row_lines(s, q.r0, q.n, y0, y1);
        const size_t off = (size_t)y0 * s->linePitch, ndw = (size_t)(y1 - y0) * s->linePitch / 4;
        const x265hip_weight& w = s->pics[q.pic].w;
        const int correction = 14 - s->prm.depth, maxVal = (1 << s->prm.depth) - 1;
        size_t blocks = (ndw + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        const uint32_t* src = (const uint32_t*)(s->pics[q.parent].dev + off);
        uint32_t* dst = (uint32_t*)(s->pics[q.pic].dev + off);
        if (s->bpp == 1) hipLaunchKernelGGL(weight_lines_kernel<uint8_t>, dim3((unsigned)blocks), dim3(256), 0, s->compute, src, dst, ndw, w.w0, w.round, w.shift, w.offset, correction, maxVal);
        else hipLaunchKernelGGL(weight_lines_kernel<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, s->compute, src, dst, ndw, w.w0, w.round, w.shift, w.offset, correction, maxVal);
        X265HIP_TRY(hipGetLastError());
        s->rowsWeighted += q.n;
    }
    const size_t org = ((size_t)s->prm.margin_y * s->prm.stride + s->prm.margin_x) * s->bpp;
    for (const Band& b : bands)
    {
        S::Slot& sl = s->slots[b.slot];
        const int n = b.r1 - b.r0 + 1;
        const size_t bandOff = (size_t)b.r0 * 64 * s->linePitch;
        x265hip_me_params p;
        memset(&p, 0, sizeof(p));
        p.depth = s->prm.depth; p.width = s->prm.width; p.height = n * 64;
        p.fenc = s->pics[b.fenc].dev + org + bandOff; p.fenc_stride = s->prm.stride;
        p.fref = s->pics[b.ref].dev + org + bandOff;  p.fref_stride = s->prm.stride;
        const int nctuBand = n * s->ctusW;
        if (s->centreRange)
        {
            // where did each CTU go?  minima of the +-centre_range search on SAD alone, the 64x64 block's displacement = the window's centre
            int rc = x265hip_me_best_reset((uint64_t*)s->dBest, (size_t)nctuBand * 85, s->compute);
            if (rc) return rc;
            p.range = s->centreRange; p.best = (uint64_t*)s->dBest; p.cost_x = p.cost_y = s->dZeroCost;
            rc = x265hip_me_fullsearch(&p, s->compute);
            if (rc) return rc;
            hipLaunchKernelGGL(centre_kernel, dim3((nctuBand + 63) / 64), dim3(64), 0, s->compute, (const unsigned long long*)s->dBest,
                               sl.dCentres + (size_t)b.r0 * s->ctusW * 2, nctuBand, s->centreRange, s->maxCx, s->maxCy);
            X265HIP_TRY(hipGetLastError());
            p.best = nullptr; p.cost_x = p.cost_y = nullptr;
            p.centres = sl.dCentres + (size_t)b.r0 * s->ctusW * 2;
        }
        p.range = s->prm.range;
        p.surf_format = s->prm.surf_format;
        uint8_t* dst = sl.dSurf + (size_t)b.r0 * s->rowBytes;
        const bool staged = s->prm.min_level || s->planes;
        p.surf = (int32_t*)(staged ? s->dScratch : dst);
        int rc = x265hip_me_fullsearch(&p, s->compute);
        if (rc) return rc;
        if (s->planes)
        {
            const int pu0 = s->prm.min_level > 1 ? 80 : s->prm.min_level ? 64 : 0;
            const size_t total = (size_t)nctuBand * s->nc * (85 - pu0) * s->ng;
            size_t blocks = (total + 255) / 256;
            if (blocks > 16384) blocks = 16384;
            hipLaunchKernelGGL(surf_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, s->compute, (const uint8_t*)s->dScratch, dst, nctuBand, s->nc, s->ng, s->fullRec,
                               s->prm.surf_format == X265HIP_SURF_PACKED ? 1 : 0, pu0, s->ctuBytes);
            X265HIP_TRY(hipGetLastError());
        }
        else if (s->prm.min_level)
        {
            const size_t nrec = (size_t)n * s->ctusW * s->nc * s->ng;
            const int rec16 = s->fullRec / 16, tail16 = s->rec / 16;
            size_t blocks = (nrec * tail16 + 255) / 256;
            if (blocks > 16384) blocks = 16384;
            hipLaunchKernelGGL(surf_tail_kernel, dim3((unsigned)blocks), dim3(256), 0, s->compute, (const uint4*)s->dScratch, (uint4*)dst, nrec, rec16, tail16);
            X265HIP_TRY(hipGetLastError());
        }
        X265HIP_TRY(hipEventRecord(sl.evSearched, s->compute));
        X265HIP_TRY(hipStreamWaitEvent(s->copy, sl.evSearched, 0));
        X265HIP_TRY(hipMemcpyAsync(sl.surf + (size_t)b.r0 * s->rowBytes, dst, (size_t)n * s->rowBytes, hipMemcpyDeviceToHost, s->copy));
        if (s->centreRange)
            X265HIP_TRY(hipMemcpyAsync(sl.centres + (size_t)b.r0 * s->ctusW * 2, sl.dCentres + (size_t)b.r0 * s->ctusW * 2, (size_t)nctuBand * 4, hipMemcpyDeviceToHost, s->copy));
        X265HIP_TRY(hipEventRecord(sl.evDown, s->copy));
        s->bytesDown += (size_t)n * s->rowBytes;
    }
    for (const Band& b : bands)
    {
        S::Slot& sl = s->slots[b.slot];
        X265HIP_TRY(hipEventSynchronize(sl.evDown));
        std::lock_guard<std::mutex> lk(s->mu);                    // against pair_open: a reopened slot keeps its cleared flags
        if (sl.generation == b.gen)
        {
            for (int r = b.r0; r <= b.r1; r++) sl.ready[r].store(b.gen, std::memory_order_release);
            if (b.r1 == s->ctusH - 1) s->pairsCompleted++;
        }
        s->bands++; s->rowsSearched += b.r1 - b.r0 + 1;
    }
    return 0;
}

struct PendingPins { hipEvent_t ev; std::vector<int> pics; };

void worker_main(S* s)
{
    std::vector<PendingPins> pending;
    std::vector<hipEvent_t> spareEvents;
    (void)hipSetDevice(s->device);
    for (;;)
    {
        std::vector<Upload> ups;
        std::vector<Weigh> weighs;
        std::vector<Band> bands;
        {
            std::unique_lock<std::mutex> lk(s->mu);
            // rounds without bands leave their pins to an event (round-5 advisor: a blocking wait there serialised upload and search scheduling while rows arrive one
            // at a time): whatever has completed is unpinned here; while pins are outstanding the wait is bounded, so they never outlive their copies by much
            auto reap = [&] {
                for (size_t i = 0; i < pending.size();)
                    if (hipEventQuery(pending[i].ev) != hipErrorNotReady)
                    {
                        for (int pic : pending[i].pics) s->pics[pic].busy--;
                        spareEvents.push_back(pending[i].ev);
                        pending.erase(pending.begin() + i);
                    }
                    else i++;
            };
            reap();
            while (!s->stop && !s->dirty)
            {
                if (pending.empty()) s->cv.wait(lk, [s] { return s->stop || s->dirty; });
                else { s->cv.wait_for(lk, std::chrono::milliseconds(1), [s] { return s->stop || s->dirty; }); reap(); }
            }
            if (s->stop)
            {
                lk.unlock();
                (void)hipStreamSynchronize(s->compute);
                for (auto& p : pending) (void)hipEventDestroy(p.ev);
                for (auto& e : spareEvents) (void)hipEventDestroy(e);
                return;
            }
            s->dirty = false;
            for (int i = 0; i < (int)s->pics.size(); i++)
            {
                S::Pic& pc = s->pics[i];
                if (!pc.used) continue;
                for (int r = 0; r < s->ctusH;)
                {
                    if (pc.rows[r] != ROW_STAGED) { r++; continue; }
                    int e = r;
                    while (e < s->ctusH && pc.rows[e] == ROW_STAGED) pc.rows[e++] = ROW_ON_DEVICE;      // stream order: every later launch sees them
                    ups.push_back({ i, r, e - r });
                    pc.busy++;                              // pinned until the round is synchronised (see the end of the round)
                    r = e;
                }
            }
            // derived pictures an open pair reads: weight the rows of their parent that are on the device (or on their way: stream order)
            for (int i = 0; i < (int)s->pics.size(); i++)
            {
                S::Pic& pd = s->pics[i];
                if (!pd.used || !pd.derived) continue;
                const S::Pic& pp = s->pics[pd.parent];
                if (!pp.used || pp.epoch != pd.parentEpoch) continue;                     // the parent went away: pairs on pd go stale below
                bool wanted = false;
                for (const auto& sl : s->slots) wanted |= sl.active && sl.ref == i && sl.refEpoch == pd.epoch;
                if (!wanted) continue;
                for (int r = 0; r < s->ctusH;)
                {
                    if (pd.rows[r] == ROW_ON_DEVICE || pp.rows[r] != ROW_ON_DEVICE) { r++; continue; }
                    int e = r;
                    while (e < s->ctusH && pd.rows[e] != ROW_ON_DEVICE && pp.rows[e] == ROW_ON_DEVICE) pd.rows[e++] = ROW_ON_DEVICE;
                    weighs.push_back({ i, pd.parent, r, e - r });
                    pd.busy++; s->pics[pd.parent].busy++;
                    r = e;
                }
            }
            for (int i = 0; i < (int)s->slots.size(); i++)
            {
                S::Slot& sl = s->slots[i];
                if (!sl.active) continue;
                const S::Pic& pf = s->pics[sl.fenc]; const S::Pic& pr = s->pics[sl.ref];
                bool gone = !pf.used || !pr.used || pf.epoch != sl.fencEpoch || pr.epoch != sl.refEpoch;
                if (!gone && pr.derived) gone = !s->pics[pr.parent].used || s->pics[pr.parent].epoch != pr.parentEpoch;
                if (gone) { sl.active = false; s->noPicture++; continue; }
                int r1 = sl.nextRow - 1;
                while (r1 + 1 < s->ctusH && r1 + 1 - sl.nextRow < s->bandRows)
                {
                    const int r = r1 + 1;
                    const int lo = r - s->lagRows < 0 ? 0 : r - s->lagRows, hi = r + s->lagRows >= s->ctusH ? s->ctusH - 1 : r + s->lagRows;
                    bool ok = pf.rows[r] == ROW_ON_DEVICE;
                    for (int k = lo; k <= hi && ok; k++) ok = pr.rows[k] == ROW_ON_DEVICE;
                    if (!ok) break;
                    r1 = r;
                }
                if (r1 < sl.nextRow) continue;
                bands.push_back({ i, sl.generation, sl.nextRow, r1, sl.fenc, sl.ref });
                // a pair that closes with this band (active = false below) no longer holds its pictures: pin them until the launches that
                // read them have run, or a host thread staging a new picture could take the entry while uploads / searches are queued on it
                s->pics[sl.fenc].busy++; s->pics[sl.ref].busy++;
                sl.nextRow = r1 + 1;
                if (sl.nextRow == s->ctusH) sl.active = false;
                else s->dirty = true;                                   // more rows may be searchable at once: come round again
            }
        }
        if (ups.empty() && weighs.empty() && bands.empty()) continue;
        const double t0 = ms_now_us();
        bool failedRound = false;
        if (run_round(s, ups, weighs, bands))
        {
            s->failed += bands.size() + 1;
            snprintf(s->workerError, sizeof(s->workerError), "%s", x265hip_last_error());
            failedRound = true;
        }
        // Bands end with a wait for their downloads (everything queued before them on the compute stream has run by then).  A round without bands leaves its
        // pins to an EVENT recorded behind its uploads / weighted rows and carries on; a round that FAILED may have stopped anywhere, e.g. before its bands'
        // download wait: both streams are drained before its pins go (the uploads still queued would otherwise read an entry a host thread is staging anew).
        bool deferred = false;
        if (failedRound) { (void)hipStreamSynchronize(s->compute); (void)hipStreamSynchronize(s->copy); }
        else if (bands.empty())
        {
            hipEvent_t ev = nullptr;
            if (!spareEvents.empty()) { ev = spareEvents.back(); spareEvents.pop_back(); }
            else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
            if (ev && hipEventRecord(ev, s->compute) == hipSuccess)
            {
                PendingPins pp; pp.ev = ev;
                for (const Upload& u : ups) pp.pics.push_back(u.pic);
                for (const Weigh& q : weighs) { pp.pics.push_back(q.pic); pp.pics.push_back(q.parent); }
                pending.push_back(pp);
                deferred = true;
            }
            else (void)hipStreamSynchronize(s->compute);
        }
        if (!deferred)
        {
            std::lock_guard<std::mutex> lk(s->mu);
            for (const Upload& u : ups) s->pics[u.pic].busy--;
            for (const Weigh& q : weighs) { s->pics[q.pic].busy--; s->pics[q.parent].busy--; }
            for (const Band& b : bands) { s->pics[b.fenc].busy--; s->pics[b.ref].busy--; }
        }
        s->usBusy += (uint64_t)(ms_now_us() - t0);
    }
}

void free_all(S* s)
{
    for (auto& p : s->pics) { if (p.stage) (void)hipHostFree(p.stage); if (p.dev) (void)hipFree(p.dev); }
    for (auto& sl : s->slots)
    {
        if (sl.surf) (void)hipHostFree(sl.surf);
        if (sl.dSurf) (void)hipFree(sl.dSurf);
        if (sl.centres) (void)hipHostFree(sl.centres);
        if (sl.dCentres) (void)hipFree(sl.dCentres);
        if (sl.evSearched) (void)hipEventDestroy(sl.evSearched);
        if (sl.evDown) (void)hipEventDestroy(sl.evDown);
        delete[] sl.ready;
    }
    if (s->dScratch) (void)hipFree(s->dScratch);
    if (s->dBest) (void)hipFree(s->dBest);
    if (s->dZeroCost) (void)hipFree(s->dZeroCost);
    if (s->compute) (void)hipStreamDestroy(s->compute);
    if (s->copy) (void)hipStreamDestroy(s->copy);
}

// index of the picture named `key`, created when it is new (least recently used entry that no open pair reads); -1 when every entry is held
bool pair_reads(const S* s, int i)
{
    const S::Pic& p = s->pics[i];
    for (const auto& sl : s->slots)
        if (sl.active && ((sl.fenc == i && sl.fencEpoch == p.epoch) || (sl.ref == i && sl.refEpoch == p.epoch))) return true;
    return false;
}

// an entry nobody needs: least recently used one that no open pair reads - directly or as the parent of a derived picture a pair reads
int pick_victim(S* s)
{
    int victim = -1;
    for (int i = 0; i < (int)s->pics.size(); i++)
    {
        S::Pic& p = s->pics[i];
        if (!p.used) return i;
        if (p.busy) continue;
        bool held = pair_reads(s, i);
        for (int d = 0; d < (int)s->pics.size() && !held; d++)
        {
            const S::Pic& pd = s->pics[d];
            held = pd.used && pd.derived && pd.parent == i && pd.parentEpoch == p.epoch && pair_reads(s, d);
        }
        if (!held && (victim < 0 || p.stamp < s->pics[victim].stamp)) victim = i;
    }
    return victim;
}

int find_or_make_picture(S* s, uint64_t key)
{
    for (int i = 0; i < (int)s->pics.size(); i++)
        if (s->pics[i].used && !s->pics[i].derived && s->pics[i].key == key) { s->pics[i].stamp = ++s->clock; return i; }
    const int victim = pick_victim(s);
    if (victim < 0) return -1;
    S::Pic& p = s->pics[victim];
    p.used = true; p.key = key; p.epoch++; p.stamp = ++s->clock; p.busy = 0;
    p.derived = false; p.parent = -1;
    std::fill(p.rows.begin(), p.rows.end(), (uint8_t)ROW_NONE);
    return victim;
}

// picture `parent` weighted with w: found among the derived entries or created (its rows are produced by the worker)
int find_or_make_derived(S* s, int parent, const x265hip_weight& w)
{
    const uint32_t pe = s->pics[parent].epoch;
    for (int i = 0; i < (int)s->pics.size(); i++)
    {
        S::Pic& p = s->pics[i];
        if (p.used && p.derived && p.parent == parent && p.parentEpoch == pe && p.w.w0 == w.w0 && p.w.round == w.round && p.w.shift == w.shift && p.w.offset == w.offset)
        { p.stamp = ++s->clock; return i; }
    }
    const int victim = pick_victim(s);
    if (victim < 0) return -1;
    S::Pic& p = s->pics[victim];
    p.used = true; p.key = 0; p.epoch++; p.stamp = ++s->clock; p.busy = 0;
    p.derived = true; p.parent = parent; p.parentEpoch = pe; p.w = w;
    std::fill(p.rows.begin(), p.rows.end(), (uint8_t)ROW_NONE);
    return victim;
}

} // namespace

extern "C" {

int x265hip_me_stream_create(x265hip_me_stream** out, const x265hip_me_stream_params* p)
{
    if (!out || !p) { set_error("me_stream_create: NULL argument"); return X265HIP_EINVAL; }
    *out = nullptr;
    int rc = ensure_device();
    if (rc) return rc;
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("me_stream_create: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->width <= 0 || p->height <= 0 || (p->width & 63) || (p->height & 63))
    { set_error("me_stream_create: width/height must be whole CTUs (got %dx%d)", p->width, p->height); return X265HIP_EINVAL; }
    if (p->range < 1 || p->range > 256 || p->margin_x < p->range + 12 || p->margin_y < p->range + 12)
    { set_error("me_stream_create: range %d needs margins >= range + 12 (have %d / %d)", p->range, p->margin_x, p->margin_y); return X265HIP_EINVAL; }
    if (p->stride < p->width + 2 * p->margin_x) { set_error("me_stream_create: stride %ld < width + 2 * margin_x", (long)p->stride); return X265HIP_EINVAL; }
    if (p->slots < 1 || p->slots > 256 || p->pictures < 2 || p->pictures > 256) { set_error("me_stream_create: slots %d / pictures %d", p->slots, p->pictures); return X265HIP_EINVAL; }
    if (p->surf_format != X265HIP_SURF_I32 && !(p->surf_format == X265HIP_SURF_PACKED && p->depth == 8))
    { set_error("me_stream_create: surf_format %d for depth %d (record-contiguous formats only: X265HIP_SURF_PACKED at 8 bits, X265HIP_SURF_I32)", p->surf_format, p->depth); return X265HIP_EINVAL; }
    if (p->min_level < 0 || p->min_level > (p->layout == X265HIP_STREAM_PLANES ? 2 : 1) || p->band_rows < 0) { set_error("me_stream_create: min_level %d / band_rows %d", p->min_level, p->band_rows); return X265HIP_EINVAL; }
    if (p->layout != X265HIP_STREAM_RECORDS && p->layout != X265HIP_STREAM_PLANES) { set_error("me_stream_create: layout %d", p->layout); return X265HIP_EINVAL; }
    if (p->device_plus_1 < 0 || p->device_plus_1 > x265hip_device_count()) { set_error("me_stream_create: device %d of %d", p->device_plus_1 - 1, x265hip_device_count()); return X265HIP_ENODEV; }
    if (p->centre_range < 0 || (p->centre_range && (p->centre_range < p->range || p->centre_range > 128 || p->margin_x < p->centre_range + 12 || p->margin_y < p->centre_range + 12)))
    { set_error("me_stream_create: centre_range %d (0, or range .. 128 with margins >= centre_range + 12)", p->centre_range); return X265HIP_EINVAL; }
    S* s = new (std::nothrow) S;
    if (!s) { set_error("me_stream_create: out of memory"); return X265HIP_EINVAL; }
    s->prm = *p;
    s->bpp = p->depth == 8 ? 1 : 2;
    s->ctusW = p->width / 64; s->ctusH = p->height / 64;
    s->nc = 2 * p->range + 1; s->ng = (s->nc + 3) / 4;
    s->fullRec = p->surf_format == X265HIP_SURF_I32 ? X265HIP_SURF_GROUP_BYTES_I32 : X265HIP_SURF_GROUP_BYTES_PACKED;
    s->rec = !p->min_level ? s->fullRec : p->surf_format == X265HIP_SURF_I32 ? X265HIP_SURF_TAIL_BYTES_I32 : X265HIP_SURF_TAIL_BYTES_PACKED;
    s->linePitch = (size_t)p->stride * s->bpp;
    s->planeBytes = s->linePitch * (p->height + 2 * p->margin_y);
    s->planes = p->layout == X265HIP_STREAM_PLANES;
    s->ctuBytes = (size_t)s->nc * s->ng * s->rec;
    if (s->planes)
    {
        s->ctuBytes = x265hip_stream_planes_ctu_bytes(p->range, p->min_level);
        s->rec = 0;                                           // no records in this layout
    }
    s->rowBytes = (size_t)s->ctusW * s->ctuBytes;
    s->centreRange = p->centre_range;
    s->maxCx = p->centre_range ? (p->margin_x - p->range - 12 < p->centre_range ? p->margin_x - p->range - 12 : p->centre_range) : 0;
    s->maxCy = p->centre_range ? (p->margin_y - p->range - 12 < p->centre_range ? p->margin_y - p->range - 12 : p->centre_range) : 0;
    s->fullRowBytes = (size_t)s->ctusW * s->nc * s->ng * s->fullRec;
    s->surfBytes = s->rowBytes * s->ctusH;
    s->bandRows = p->band_rows ? p->band_rows : 8;
    if (s->bandRows > s->ctusH) s->bandRows = s->ctusH;
    // CTU rows of the reference a CTU row's launches reach below (and above) itself: the centre search reads 63 + centre_range lines past the
    // row's first line, the window around a centre |cy| <= maxCy another maxCy + range (round-4 advisor: (63 + range) / 64 ignored both, so
    // with progressively arriving rows a band could be searched against stale lines of a recycled picture entry)
    {
        const int reach = s->centreRange ? (s->centreRange > s->maxCy + p->range ? s->centreRange : s->maxCy + p->range) : p->range;
        s->lagRows = (63 + reach) / 64;
    }
    if ((s->linePitch & 3) || ((((size_t)p->margin_y * p->stride + p->margin_x) * s->bpp) & 3))
    { set_error("me_stream_create: sample (0,0) and the row pitch must be 4-byte aligned"); delete s; return X265HIP_EINVAL; }
    if (p->device_plus_1 > 0)
    {
        // an instance pinned to a GPU: validated as gfx950 and made current for this (creating) thread; the worker selects it for itself
        rc = x265hip_init(p->device_plus_1 - 1);
        if (rc) { delete s; return rc; }
        s->device = p->device_plus_1 - 1;
    }
    else if (hipGetDevice(&s->device) != hipSuccess) s->device = 0;
#define MS_TRY(expr) do { if (check_hip((expr), #expr)) { free_all(s); delete s; return X265HIP_ENODEV; } } while (0)
    MS_TRY(hipStreamCreateWithFlags(&s->compute, hipStreamNonBlocking));
    MS_TRY(hipStreamCreateWithFlags(&s->copy, hipStreamNonBlocking));
    if (p->min_level || s->planes) MS_TRY(hipMalloc((void**)&s->dScratch, s->fullRowBytes * s->bandRows));
    if (s->centreRange)
    {
        MS_TRY(hipMalloc((void**)&s->dBest, (size_t)s->bandRows * s->ctusW * 85 * 8));
        MS_TRY(hipMalloc((void**)&s->dZeroCost, (size_t)(2 * s->centreRange + 8) * 2));
        MS_TRY(hipMemset(s->dZeroCost, 0, (size_t)(2 * s->centreRange + 8) * 2));
    }
    s->pics = std::vector<S::Pic>(p->pictures);
    for (auto& pc : s->pics)
    {
        MS_TRY(hipHostMalloc((void**)&pc.stage, s->planeBytes, hipHostMallocDefault));
        MS_TRY(hipMalloc((void**)&pc.dev, s->planeBytes + 256));
        pc.rows.assign(s->ctusH, (uint8_t)ROW_NONE);
    }
    s->slots = std::vector<S::Slot>(p->slots);
    for (auto& sl : s->slots)
    {
        MS_TRY(hipHostMalloc((void**)&sl.surf, s->surfBytes, hipHostMallocDefault));
        MS_TRY(hipMalloc((void**)&sl.dSurf, s->surfBytes));
        if (s->centreRange)
        {
            MS_TRY(hipHostMalloc((void**)&sl.centres, (size_t)s->ctusW * s->ctusH * 4, hipHostMallocDefault));
            MS_TRY(hipMalloc((void**)&sl.dCentres, (size_t)s->ctusW * s->ctusH * 4));
            memset(sl.centres, 0, (size_t)s->ctusW * s->ctusH * 4);
        }
        MS_TRY(hipEventCreateWithFlags(&sl.evSearched, hipEventDisableTiming));
        MS_TRY(hipEventCreateWithFlags(&sl.evDown, hipEventDisableTiming));
        sl.ready = new std::atomic<int>[s->ctusH];
        for (int r = 0; r < s->ctusH; r++) sl.ready[r].store(0);
    }
    MS_TRY(hipDeviceSynchronize());          // the zero fill above is queued on the null stream; the worker's streams do not order with it
#undef MS_TRY
    s->worker = std::thread(worker_main, s);
    *out = s;
    return 0;
}

void x265hip_me_stream_destroy(x265hip_me_stream* s)
{
    if (!s) return;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->stop = true;
    }
    s->cv.notify_all();
    if (s->worker.joinable()) s->worker.join();
    (void)hipStreamSynchronize(s->compute);
    (void)hipStreamSynchronize(s->copy);
    free_all(s);
    delete s;
}

/* CTU rows [ctu_row0, ctu_row0 + ctu_rows) of the picture named `key` are final in `buf` (the whole allocated plane: stride * (height +
 * 2 * margin_y) samples; the top margin belongs to row 0, the bottom margin to the last row): copied before the call returns. */
int x265hip_me_stream_picture_rows(x265hip_me_stream* s, uint64_t key, const void* buf, int ctu_row0, int ctu_rows)
{
    if (!s || !buf || ctu_row0 < 0 || ctu_rows < 1 || ctu_row0 + ctu_rows > s->ctusH) { set_error("me_stream_picture_rows: bad argument"); return X265HIP_EINVAL; }
    int idx;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        idx = find_or_make_picture(s, key);
        if (idx < 0) { set_error("me_stream_picture_rows: every picture entry is held by an open pair (pictures = %d)", s->prm.pictures); return X265HIP_EBUSY; }
        s->pics[idx].busy++;
    }
    long y0, y1;
    row_lines(s, ctu_row0, ctu_rows, y0, y1);
    memcpy(s->pics[idx].stage + (size_t)y0 * s->linePitch, (const uint8_t*)buf + (size_t)y0 * s->linePitch, (size_t)(y1 - y0) * s->linePitch);
    {
        std::lock_guard<std::mutex> lk(s->mu);
        S::Pic& p = s->pics[idx];
        p.busy--;
        if (p.used && p.key == key)
            for (int r = ctu_row0; r < ctu_row0 + ctu_rows; r++) p.rows[r] = ROW_STAGED;
        s->dirty = true;
    }
    s->cv.notify_one();
    return 0;
}

/* Opens pair (source picture fenc_key, reference picture ref_key) in `slot`: its CTU rows are searched as the two pictures' rows
 * arrive (before or after this call).  w != NULL: the pair searches the reference picture WEIGHTED on the device with the arguments
 * of primitives.weight_pp (the plane MotionReference::applyWeight materialises on the host, encoder/reference.cpp:119-178).
 * Returns the slot's new GENERATION (> 0) or a negative error. */
int x265hip_me_stream_pair_open_weighted(x265hip_me_stream* s, int slot, uint64_t fenc_key, uint64_t ref_key, const x265hip_weight* w)
{
    if (!s || slot < 0 || slot >= (int)s->slots.size()) { set_error("me_stream_pair_open: bad slot"); return X265HIP_EINVAL; }
    if (fenc_key == ref_key) { set_error("me_stream_pair_open: source and reference picture carry the same key"); return X265HIP_EINVAL; }
    if (w && (w->shift < 14 - s->prm.depth || w->shift > 31 || w->w0 < -128 * 64 || w->w0 > 128 * 64))
    { set_error("me_stream_pair_open: weight (w0 %d, shift %d) out of range (shift includes the 14 - depth correction of weight_pp)", w->w0, w->shift); return X265HIP_EINVAL; }
    int gen;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        S::Slot& sl = s->slots[slot];
        // every entry this pair needs is looked up (and pinned with busy) before the slot lets go of its previous pair: a failed open
        // changes nothing, and the entry returned for the source can never be the victim chosen for the reference (round-3 advisor)
        const int f = find_or_make_picture(s, fenc_key);
        if (f >= 0) s->pics[f].busy++;
        int r = f < 0 ? -1 : find_or_make_picture(s, ref_key);
        if (r >= 0 && w)
        {
            s->pics[r].busy++;
            const int d = find_or_make_derived(s, r, *w);
            s->pics[r].busy--;
            r = d;
        }
        if (f >= 0) s->pics[f].busy--;
        if (f < 0 || r < 0) { set_error("me_stream_pair_open: no picture entry free (pictures = %d)", s->prm.pictures); return X265HIP_EBUSY; }
        sl.active = false;                                   // the slot's previous pair no longer holds its pictures
        gen = ++sl.generation;
        if (gen <= 0) gen = sl.generation = 1;
        for (int k = 0; k < s->ctusH; k++) sl.ready[k].store(0, std::memory_order_release);      // before any row of the slot can be rewritten
        sl.fenc = f; sl.ref = r; sl.fencEpoch = s->pics[f].epoch; sl.refEpoch = s->pics[r].epoch;
        sl.nextRow = 0; sl.active = true;
        s->dirty = true;
        s->pairsOpened++;
        if (w) s->weightedPairs++;
    }
    s->cv.notify_one();
    return gen;
}

int x265hip_me_stream_pair_open(x265hip_me_stream* s, int slot, uint64_t fenc_key, uint64_t ref_key)
{
    return x265hip_me_stream_pair_open_weighted(s, slot, fenc_key, ref_key, nullptr);
}

const void* x265hip_me_stream_surface(x265hip_me_stream* s, int slot)
{
    return (s && slot >= 0 && slot < (int)s->slots.size()) ? s->slots[slot].surf : nullptr;
}

const volatile int* x265hip_me_stream_ready(x265hip_me_stream* s, int slot)
{
    return (s && slot >= 0 && slot < (int)s->slots.size()) ? reinterpret_cast<const volatile int*>(s->slots[slot].ready) : nullptr;
}

int x265hip_me_stream_record_bytes(x265hip_me_stream* s) { return s ? s->rec : 0; }

/* int16 [ctu][2] of the slot: the displacement each CTU's window is centred on (centre_range > 0; NULL otherwise); row r's entries are
 * valid while ready[r] == the pair's generation, like its surfaces */
const int16_t* x265hip_me_stream_centres(x265hip_me_stream* s, int slot)
{
    return (s && slot >= 0 && slot < (int)s->slots.size()) ? s->slots[slot].centres : nullptr;
}

size_t x265hip_me_stream_ctu_bytes(x265hip_me_stream* s) { return s ? s->ctuBytes : 0; }

int x265hip_me_stream_stats(x265hip_me_stream* s, x265hip_me_stream_stats_t* st)
{
    if (!s || !st) { set_error("me_stream_stats: NULL"); return X265HIP_EINVAL; }
    st->pairs_opened = s->pairsOpened; st->pairs_completed = s->pairsCompleted; st->bands = s->bands; st->rows_searched = s->rowsSearched;
    st->rows_uploaded = s->rowsUploaded; st->failed = s->failed; st->stale_pairs = s->noPicture; st->us_busy = s->usBusy;
    st->bytes_downloaded = s->bytesDown; st->bytes_uploaded = s->bytesUp; st->surface_bytes = s->surfBytes;
    st->rows_weighted = s->rowsWeighted; st->weighted_pairs = s->weightedPairs;
    if (s->failed) set_error("me_stream worker: %s", s->workerError);
    return 0;
}

} // extern "C"
