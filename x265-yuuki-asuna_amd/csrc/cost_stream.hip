// cost_stream.hip - the ROW-GRANULAR service of the sub-sample cost tables (include/x265hip.h, "SUB-SAMPLE COST TABLES"): what a host
// that runs the reference's frame threads plugs behind MotionEstimate::subpelCompare.
//
//   PICTURE  source or reconstructed picture (three planes, PicYuv layout) named by a key; a source picture arrives in one piece, a
//            reconstructed one CTU row by CTU row where the reference raises Frame::m_reconRowFlag (encoder/framefilter.cpp:664).
//   VIEW     (reconstructed picture, weights): the planes a search on that reference reads - MotionReference::applyWeight's weight_pp
//            planes when the slice weights the reference (encoder/reference.cpp:119-178) - and all their fractional phases
//            (x265hip_phase_planes), growing line by line behind the rows.  DEVICE memory only; pairs on the same view share it.
//   PAIR     slot = (source picture, view): per CTU row, as soon as the source row and the view's lines the row's candidates can reach
//            (reference rows <= r + 2: what the host's own frame encoder waits for, encoder/frameencoder.cpp:161-164,852-868) are there:
//            centre search (+-centre_range minima, SAD alone) -> SAD rasters of +-window around each CTU's centre -> candidates of every
//            PU shape -> SATD tables around them (csrc/cost_kernels.hip) -> records to pinned host memory, ready[row] = generation.
// One worker thread, one HIP stream: uploads, weighting, interpolation and the table chain of a round are stream-ordered; the round ends
// with one synchronisation, then the flags of its bands are raised.  Readers never wait and never lock (ready[row] before AND after the
// read); whatever is not served is the host primitive's to compute - the same integers.
#include "common.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

using namespace x265hip;

namespace {

// primitives.weight_pp (common/pixel.cpp:518-543) over whole buffer lines; round / shift include the 14 - depth correction
template <typename Px>
__global__ void __launch_bounds__(256) cs_weight_lines_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t ndw, int w0, int round, int shift,
                                                              int offset, int correction, int maxVal)
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
            const int val = (int)(int16_t)(px << correction);
            o |= (uint32_t)clip3(0, maxVal, ((w0 * val + round) >> shift) + offset) << (k * BITS);
        }
        dst[i] = o;
    }
}

// centre of every CTU's window = the displacement of its 64x64 block's minimum SAD in the +-big search, clamped to +-maxX / [-maxY, maxYDown] (downwards the
// candidates must stay inside the reference rows that exist when the row is computed)
__global__ void cs_centre_kernel(const unsigned long long* __restrict__ best, int16_t* __restrict__ centres, int nctu, int big, int maxX, int maxY, int maxYDown)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nctu) return;
    const uint32_t idx = (uint32_t)best[(size_t)i * 85 + 84];
    const int ncb = 2 * big + 1;
    const int mx = (int)(idx % ncb) - big, my = (int)(idx / ncb) - big;
    centres[2 * i] = (int16_t)clip3(-maxX, maxX, mx);
    centres[2 * i + 1] = (int16_t)clip3(-maxY, maxYDown, my);
}

double cs_now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

} // namespace

struct x265hip_cost_stream
{
    x265hip_cost_stream_params prm;
    int bpp, device, ctusW, ctuRows, nplanes, bandRows, npu, recBytes, maxCx, maxCy, maxCyDown;
    size_t planeBytes[2], pitch[2], ctuBytes, rowBytes;
    int rows[2], margin[2], ctuLines[2], nph[2];
    hipStream_t stream = nullptr;
    struct Pic
    {
        uint64_t key = 0; bool used = false; uint32_t epoch = 0; uint64_t stamp = 0; int busy = 0;
        uint8_t* stage[3] = { nullptr, nullptr, nullptr };      // pinned planes, rows staged by the host threads
        uint8_t* dSrc[3] = { nullptr, nullptr, nullptr };
        std::vector<uint8_t> staged;                            // per CTU row
        int nextRow = 0;                                        // rows [0, nextRow) are uploaded (or queued on the stream)
    };
    struct View
    {
        bool used = false; uint64_t stamp = 0; int busy = 0;          // busy: launches of the running round read it
        int pic = -1; uint32_t picEpoch = 0; bool active = false;      // active: still growing behind its picture
        unsigned mask = 0; x265hip_weight w[3] = {};
        uint8_t* dW[3] = { nullptr, nullptr, nullptr };         // the picture's planes weighted (only the planes of the mask)
        uint8_t* dOut[3] = { nullptr, nullptr, nullptr };       // every phase plane
        int rowsSeen = 0; int done[2] = { 0, 0 };
    };
    struct Slot
    {
        uint8_t* tables = nullptr;                              // pinned host memory
        std::atomic<int>* ready = nullptr;
        int generation = 0;
        int fenc = -1, view = -1; uint32_t fencEpoch = 0; uint64_t viewStamp = 0;
        int nextRow = 0; bool active = false;
        int bandLimit = 1;                                      // rows of the pair's next band: 1, 2, 4 ... band_rows - the first rows of EVERY open pair land before anybody's last ones
        uint16_t* hMvCost = nullptr; uint16_t* dMvCost = nullptr; bool hasCost = false;      // the pair's vector-cost table (pinned staging, device copy); !hasCost: rank by SAD alone
    };
    std::vector<Pic> pics;
    std::vector<View> views;
    std::vector<Slot> slots;
    // one band's scratch
    uint8_t* dSurf = nullptr; unsigned long long* dBest = nullptr; uint16_t* dZeroCost = nullptr; int16_t* dCentres = nullptr; int16_t* dCand = nullptr; uint8_t* dTables = nullptr;
    uint64_t clock = 0;
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false, dirty = false;
    std::thread worker;
    std::atomic<uint64_t> pairsOpened{0}, pairsCompleted{0}, bands{0}, rowsServed{0}, rowsUploaded{0}, failed{0}, stalePairs{0}, viewsOpened{0}, viewsShared{0}, linesWeighted{0};
    std::atomic<uint64_t> usBusy{0}, bytesDown{0}, bytesUp{0};
    char workerError[256] = "";
};

namespace {

typedef x265hip_cost_stream CS;

// buffer lines [y0, y1) of plane kind k (0 luma, 1 chroma) that CTU rows [r0, r0 + n) occupy; margins travel with the first / last row
inline void cs_lines(const CS* s, int k, int r0, int n, int& y0, int& y1)
{
    y0 = r0 == 0 ? 0 : s->margin[k] + r0 * s->ctuLines[k];
    y1 = r0 + n == s->ctuRows ? s->rows[k] : s->margin[k] + (r0 + n) * s->ctuLines[k];
}

struct Upload { int pic, r0, r1; };
struct ViewJob { int view, pic, r0, r1; int done[2]; unsigned mask; x265hip_weight w[3]; };
struct Band { int slot, gen, r0, r1, fenc, view; bool hasCost; };

int run_round(CS* s, const std::vector<Upload>& ups, const std::vector<ViewJob>& jobs, const std::vector<Band>& bands)
{
    X265HIP_TRY(hipSetDevice(s->device));
    apply_wait_policy(s->device);
    for (const Upload& u : ups)
        for (int pl = 0; pl < s->nplanes; pl++)
        {
            const int k = pl ? 1 : 0;
            int y0, y1;
            cs_lines(s, k, u.r0, u.r1 - u.r0, y0, y1);
            X265HIP_TRY(hipMemcpyAsync(s->pics[u.pic].dSrc[pl] + (size_t)y0 * s->pitch[k], s->pics[u.pic].stage[pl] + (size_t)y0 * s->pitch[k],
                                       (size_t)(y1 - y0) * s->pitch[k], hipMemcpyHostToDevice, s->stream));
            s->bytesUp += (size_t)(y1 - y0) * s->pitch[k];
            if (!pl) s->rowsUploaded += u.r1 - u.r0;
        }
    for (const ViewJob& job : jobs)
    {
        CS::View& v = s->views[job.view];
        for (int pl = 0; pl < s->nplanes; pl++)
        {
            const int k = pl ? 1 : 0;
            int y0, y1;
            cs_lines(s, k, job.r0, job.r1 - job.r0, y0, y1);
            const uint8_t* src = s->pics[job.pic].dSrc[pl];
            if (job.mask & (1u << pl))
            {
                const size_t off = (size_t)y0 * s->pitch[k], ndw = (size_t)(y1 - y0) * s->pitch[k] / 4;
                const x265hip_weight& w = job.w[pl];
                const int correction = 14 - s->prm.depth, maxVal = (1 << s->prm.depth) - 1;
                size_t blocks = (ndw + 255) / 256;
                if (blocks > 8192) blocks = 8192;
                if (s->bpp == 1)
                    hipLaunchKernelGGL(cs_weight_lines_kernel<uint8_t>, dim3((unsigned)blocks), dim3(256), 0, s->stream, (const uint32_t*)(src + off), (uint32_t*)(v.dW[pl] + off),
                                       ndw, w.w0, w.round, w.shift, w.offset, correction, maxVal);
                else
                    hipLaunchKernelGGL(cs_weight_lines_kernel<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, s->stream, (const uint32_t*)(src + off), (uint32_t*)(v.dW[pl] + off),
                                       ndw, w.w0, w.round, w.shift, w.offset, correction, maxVal);
                X265HIP_TRY(hipGetLastError());
                s->linesWeighted += (uint64_t)(y1 - y0);
                src = v.dW[pl];
            }
            // producible now: lines [max(done, 4), y1 - 8) - a line needs 3 source lines above and up to 8 below it
            const int b0 = job.done[k] < 4 ? 4 : job.done[k], b1 = y1 - 8;
            if (b1 - b0 < 4) continue;
            const size_t lineOff = (size_t)(b0 - 4) * s->pitch[k];
            int rc = phase_planes_launch(s->prm.depth, k, src + lineOff, v.dOut[pl] + lineOff, k ? s->prm.stride_c : s->prm.stride, b1 - b0 + 12, s->planeBytes[k], s->stream);
            if (rc) return rc;
        }
    }
    const size_t org = ((size_t)s->prm.margin_y * s->prm.stride + s->prm.margin_x) * s->bpp;
    for (const Band& b : bands)
    {
        CS::Slot& sl = s->slots[b.slot];
        const CS::View& v = s->views[b.view];
        const int n = b.r1 - b.r0 + 1, nctuBand = n * s->ctusW;
        const size_t bandOff = (size_t)b.r0 * 64 * s->pitch[0];
        const uint8_t* refL = (v.mask & 1) ? v.dW[0] : s->pics[v.pic].dSrc[0];
        x265hip_me_params p;
        memset(&p, 0, sizeof(p));
        p.depth = s->prm.depth; p.width = s->prm.width; p.height = n * 64;
        p.fenc = s->pics[b.fenc].dSrc[0] + org + bandOff; p.fenc_stride = s->prm.stride;
        p.fref = refL + org + bandOff; p.fref_stride = s->prm.stride;
        int rc;
        if (s->prm.centre_range)
        {
            if ((rc = x265hip_me_best_reset((uint64_t*)s->dBest, (size_t)nctuBand * 85, s->stream))) return rc;
            p.range = s->prm.centre_range; p.best = (uint64_t*)s->dBest; p.cost_x = p.cost_y = s->dZeroCost;
            if ((rc = x265hip_me_fullsearch(&p, s->stream))) return rc;
            hipLaunchKernelGGL(cs_centre_kernel, dim3((nctuBand + 63) / 64), dim3(64), 0, s->stream, (const unsigned long long*)s->dBest, s->dCentres, nctuBand,
                               s->prm.centre_range, s->maxCx, s->maxCy, s->maxCyDown);
            X265HIP_TRY(hipGetLastError());
            p.best = nullptr; p.cost_x = p.cost_y = nullptr;
        }
        else
            X265HIP_TRY(hipMemsetAsync(s->dCentres, 0, (size_t)nctuBand * 4, s->stream));
        p.centres = s->dCentres;
        p.range = s->prm.window; p.surf_format = X265HIP_SURF_I32; p.surf = (int32_t*)s->dSurf;
        if ((rc = x265hip_me_fullsearch(&p, s->stream))) return rc;
        const uint16_t* mvCost = nullptr;
        if (b.hasCost)
        {
            // a slot reopened under this copy belongs to a stale generation: its band is discarded below
            X265HIP_TRY(hipMemcpyAsync(sl.dMvCost, sl.hMvCost, (2 * (size_t)s->prm.window + 1) * sizeof(uint16_t), hipMemcpyHostToDevice, s->stream));
            mvCost = sl.dMvCost;
        }
        x265hip_cost_candidates_params c = { nctuBand, s->prm.window, (const int32_t*)s->dSurf, s->dCentres, s->prm.shapes, s->prm.candidates, s->dCand, mvCost };
        if ((rc = x265hip_cost_candidates(&c, s->stream))) return rc;
        x265hip_cost_tables_params t;
        memset(&t, 0, sizeof(t));
        t.depth = s->prm.depth; t.width = s->prm.width; t.stride = s->prm.stride; t.margin_x = s->prm.margin_x; t.margin_y = s->prm.margin_y;
        t.stride_c = s->prm.stride_c; t.margin_y_c = s->prm.margin_y_c; t.ctu_row0 = b.r0; t.ctu_rows = n;
        for (int pl = 0; pl < s->nplanes; pl++)
        {
            t.fenc[pl] = s->pics[b.fenc].dSrc[pl];
            t.ref[pl] = (v.mask & (1u << pl)) ? v.dW[pl] : s->pics[v.pic].dSrc[pl];
            t.phases[pl] = v.dOut[pl];
        }
        t.plane_bytes = s->planeBytes[0]; t.plane_bytes_c = s->planeBytes[1];
        t.shapes = s->prm.shapes; t.candidates = s->prm.candidates; t.subme = s->prm.subme; t.chroma = s->prm.chroma; t.sad_costs = s->prm.sad_costs;
        t.cand = s->dCand; t.tables = s->dTables;
        if ((rc = x265hip_cost_tables(&t, s->stream))) return rc;
        X265HIP_TRY(hipMemcpyAsync(sl.tables + (size_t)b.r0 * s->rowBytes, s->dTables, (size_t)n * s->rowBytes, hipMemcpyDeviceToHost, s->stream));
        s->bytesDown += (size_t)n * s->rowBytes;
    }
    X265HIP_TRY(hipStreamSynchronize(s->stream));
    if (!bands.empty())
    {
        std::lock_guard<std::mutex> lk(s->mu);                    // against pair_open: a reopened slot keeps its cleared flags
        for (const Band& b : bands)
        {
            CS::Slot& sl = s->slots[b.slot];
            if (sl.generation == b.gen)
            {
                for (int r = b.r0; r <= b.r1; r++) sl.ready[r].store(b.gen, std::memory_order_release);
                if (b.r1 == s->ctuRows - 1) s->pairsCompleted++;
            }
            s->bands++; s->rowsServed += b.r1 - b.r0 + 1;
        }
    }
    return 0;
}

void cs_worker(CS* s)
{
    for (;;)
    {
        std::vector<Upload> ups;
        std::vector<ViewJob> jobs;
        std::vector<Band> bands;
        {
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv.wait(lk, [s] { return s->stop || s->dirty; });
            if (s->stop) return;
            s->dirty = false;
            for (int i = 0; i < (int)s->pics.size(); i++)
            {
                CS::Pic& pc = s->pics[i];
                if (!pc.used) continue;
                int r1 = pc.nextRow;
                while (r1 < s->ctuRows && pc.staged[r1]) r1++;          // views grow top to bottom: only a contiguous prefix is useful
                if (r1 > pc.nextRow) { ups.push_back({ i, pc.nextRow, r1 }); pc.nextRow = r1; pc.busy++; }
            }
            for (int i = 0; i < (int)s->views.size(); i++)
            {
                CS::View& v = s->views[i];
                if (!v.used || !v.active) continue;
                const CS::Pic& pc = s->pics[v.pic];
                if (!pc.used || pc.epoch != v.picEpoch) { v.active = false; continue; }      // the picture went away: what is finished stays valid
                if (pc.nextRow > v.rowsSeen)
                {
                    ViewJob job = { i, v.pic, v.rowsSeen, pc.nextRow, { v.done[0], v.done[1] }, v.mask, { v.w[0], v.w[1], v.w[2] } };
                    jobs.push_back(job);
                    s->pics[v.pic].busy++; v.busy++;
                    // the lines this job will have produced (the launches are stream-ordered before any band of this round)
                    for (int k = 0; k < (s->nplanes > 1 ? 2 : 1); k++)
                    {
                        int y0, y1;
                        cs_lines(s, k, v.rowsSeen, pc.nextRow - v.rowsSeen, y0, y1);
                        const int b0 = v.done[k] < 4 ? 4 : v.done[k], b1 = y1 - 8;
                        if (b1 - b0 >= 4) v.done[k] = b1;
                    }
                    v.rowsSeen = pc.nextRow;
                    if (v.rowsSeen == s->ctuRows) v.active = false;
                }
            }
            for (int i = 0; i < (int)s->slots.size(); i++)
            {
                CS::Slot& sl = s->slots[i];
                if (!sl.active) continue;
                const CS::Pic& pf = s->pics[sl.fenc];
                const CS::View& v = s->views[sl.view];
                if (!pf.used || pf.epoch != sl.fencEpoch || !v.used || v.stamp != sl.viewStamp ||
                    (v.rowsSeen < s->ctuRows && (!s->pics[v.pic].used || s->pics[v.pic].epoch != v.picEpoch)))
                { sl.active = false; s->stalePairs++; continue; }
                int r1 = sl.nextRow - 1;
                while (r1 + 1 < s->ctuRows && r1 + 1 - sl.nextRow < sl.bandLimit)
                {
                    const int r = r1 + 1;
                    const int need = r + 2 > s->ctuRows ? s->ctuRows : r + 2;          // the candidates of row r reach <= 44 luma lines below it (maxCyDown + window + 2)
                    if (pf.nextRow <= r || v.rowsSeen < need) break;
                    r1 = r;
                }
                if (r1 < sl.nextRow) continue;
                bands.push_back({ i, sl.generation, sl.nextRow, r1, sl.fenc, sl.view, sl.hasCost });
                s->pics[sl.fenc].busy++; s->pics[v.pic].busy++; s->views[sl.view].busy++;
                sl.nextRow = r1 + 1;
                sl.bandLimit = sl.bandLimit * 2 > s->bandRows ? s->bandRows : sl.bandLimit * 2;
                if (sl.nextRow == s->ctuRows) sl.active = false;
                else s->dirty = true;
            }
        }
        if (ups.empty() && jobs.empty() && bands.empty()) continue;
        const double t0 = cs_now_us();
        if (run_round(s, ups, jobs, bands))
        {
            s->failed += bands.size() + 1;
            snprintf(s->workerError, sizeof(s->workerError), "%s", x265hip_last_error());
            (void)hipStreamSynchronize(s->stream);                      // nothing queued may still read a picture whose pin goes below
        }
        {
            std::lock_guard<std::mutex> lk(s->mu);
            for (const Upload& u : ups) s->pics[u.pic].busy--;
            for (const ViewJob& j : jobs) { s->pics[j.pic].busy--; s->views[j.view].busy--; }
            for (const Band& b : bands) { s->pics[b.fenc].busy--; s->pics[s->views[b.view].pic].busy--; s->views[b.view].busy--; }
        }
        s->usBusy += (uint64_t)(cs_now_us() - t0);
    }
}

void cs_free(CS* s)
{
    for (auto& pc : s->pics)
        for (int i = 0; i < 3; i++)
        {
            if (pc.stage[i]) (void)hipHostFree(pc.stage[i]);
            if (pc.dSrc[i]) (void)hipFree(pc.dSrc[i]);
        }
    for (auto& v : s->views)
        for (int i = 0; i < 3; i++)
        {
            if (v.dW[i]) (void)hipFree(v.dW[i]);
            if (v.dOut[i]) (void)hipFree(v.dOut[i]);
        }
    for (auto& sl : s->slots) { if (sl.tables) (void)hipHostFree(sl.tables); if (sl.hMvCost) (void)hipHostFree(sl.hMvCost); if (sl.dMvCost) (void)hipFree(sl.dMvCost); delete[] sl.ready; }
    if (s->dSurf) (void)hipFree(s->dSurf);
    if (s->dBest) (void)hipFree(s->dBest);
    if (s->dZeroCost) (void)hipFree(s->dZeroCost);
    if (s->dCentres) (void)hipFree(s->dCentres);
    if (s->dCand) (void)hipFree(s->dCand);
    if (s->dTables) (void)hipFree(s->dTables);
    if (s->stream) (void)hipStreamDestroy(s->stream);
}

bool cs_pic_held(const CS* s, int i)
{
    const CS::Pic& p = s->pics[i];
    for (const auto& v : s->views)
        if (v.used && v.pic == i && v.picEpoch == p.epoch)
        {
            if (v.active) return true;
            // a finished view still reads its picture's planes where it is not weighted (phase 0 = the picture itself)
            for (const auto& sl : s->slots) if (sl.active && sl.view == (int)(&v - &s->views[0]) && sl.viewStamp == v.stamp) return true;
        }
    for (const auto& sl : s->slots) if (sl.active && sl.fenc == i && sl.fencEpoch == p.epoch) return true;
    return false;
}

// index of the picture named `key`, created when it is new: least recently used entry nothing is still fed from; -1 = all held
int cs_find_or_make_picture(CS* s, uint64_t key)
{
    for (int i = 0; i < (int)s->pics.size(); i++)
        if (s->pics[i].used && s->pics[i].key == key) { s->pics[i].stamp = ++s->clock; return i; }
    int victim = -1;
    for (int i = 0; i < (int)s->pics.size(); i++)
    {
        CS::Pic& p = s->pics[i];
        if (!p.used) { victim = i; break; }
        if (p.busy || cs_pic_held(s, i)) continue;
        if (victim < 0 || p.stamp < s->pics[victim].stamp) victim = i;
    }
    if (victim < 0) return -1;
    CS::Pic& p = s->pics[victim];
    // views of the recycled entry are void: their unweighted planes are the picture's own
    for (auto& v : s->views) if (v.used && v.pic == victim && v.picEpoch == p.epoch) { v.used = false; v.active = false; }
    p.used = true; p.key = key; p.epoch++; p.stamp = ++s->clock; p.busy = 0;
    std::fill(p.staged.begin(), p.staged.end(), (uint8_t)0);
    p.nextRow = 0;
    return victim;
}

bool same_w(const x265hip_weight& a, const x265hip_weight& b) { return a.w0 == b.w0 && a.round == b.round && a.shift == b.shift && a.offset == b.offset; }

// the view of (pic, weights): an existing one (shared) or a new one in the least recently used entry no active pair reads; -1 = all held
int cs_find_or_make_view(CS* s, int pic, const x265hip_weight* w, unsigned mask)
{
    const CS::Pic& pc = s->pics[pic];
    for (int i = 0; i < (int)s->views.size(); i++)
    {
        CS::View& v = s->views[i];
        if (!v.used || v.pic != pic || v.picEpoch != pc.epoch || v.mask != mask) continue;
        bool same = true;
        for (int c = 0; c < 3 && same; c++) if (mask & (1u << c)) same = same_w(v.w[c], w[c]);
        if (same) { s->viewsShared++; return i; }
    }
    int victim = -1;
    for (int i = 0; i < (int)s->views.size(); i++)
    {
        CS::View& v = s->views[i];
        if (!v.used) { victim = i; break; }
        bool held = v.busy > 0 || v.active;
        for (const auto& sl : s->slots) held |= sl.active && sl.view == i && sl.viewStamp == v.stamp;
        if (!held && (victim < 0 || v.stamp < s->views[victim].stamp)) victim = i;
    }
    if (victim < 0) return -1;
    CS::View& v = s->views[victim];
    v.used = true; v.stamp = ++s->clock; v.busy = 0; v.pic = pic; v.picEpoch = pc.epoch; v.active = true;
    v.mask = mask;
    for (int c = 0; c < 3; c++) v.w[c] = (mask & (1u << c)) ? w[c] : x265hip_weight{ 0, 0, 0, 0 };
    v.rowsSeen = 0; v.done[0] = v.done[1] = 0;
    s->viewsOpened++;
    return victim;
}

} // namespace

extern "C" {

int x265hip_cost_stream_create(x265hip_cost_stream** out, const x265hip_cost_stream_params* p)
{
    if (!out || !p) { set_error("cost_stream_create: NULL argument"); return X265HIP_EINVAL; }
    *out = nullptr;
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("cost_stream_create: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->width < 64 || (p->width & 63) || p->height < 64 || (p->height & 63) || p->stride < p->width + 2 * p->margin_x || (p->stride & 3) || (p->margin_x & 3) ||
        p->margin_y < 16 || (p->margin_y & 3))
    { set_error("cost_stream_create: luma geometry %dx%d pitch %ld margins %d / %d", p->width, p->height, (long)p->stride, p->margin_x, p->margin_y); return X265HIP_EINVAL; }
    const bool hasC = p->stride_c > 0;
    if (hasC && (p->stride_c < p->width / 2 + 2 * p->margin_x || (p->stride_c & 3) || p->margin_y_c < 8 || (p->margin_y_c & 3) || p->margin_y_c * 2 > p->margin_y))
    { set_error("cost_stream_create: chroma geometry pitch %ld margin %d", (long)p->stride_c, p->margin_y_c); return X265HIP_EINVAL; }
    if (p->chroma && !hasC) { set_error("cost_stream_create: chroma costs need the chroma planes"); return X265HIP_EINVAL; }
    if (p->window < 0 || p->window > 32 || p->centre_range < 0 || p->centre_range + 12 > p->margin_x || p->centre_range + 12 > p->margin_y || p->candidates < 1 || p->candidates > 2 || p->shapes < 0 || p->shapes > 2 ||
        x265hip_cost_record_bytes(p->subme, 0) == 0)
    { set_error("cost_stream_create: window %d / centre range %d / %d candidates / shape set %d / subme %d", p->window, p->centre_range, p->candidates, p->shapes, p->subme); return X265HIP_EINVAL; }
    if (p->slots < 1 || p->slots > 256 || p->pictures < 2 || p->pictures > 256 || p->views < 1 || p->views > 64)
    { set_error("cost_stream_create: %d slots / %d pictures / %d views", p->slots, p->pictures, p->views); return X265HIP_EINVAL; }
    // the candidates of a CTU lie within centre +- window; the fractional positions reach 2 samples further, the phase planes' lines [4, rows - 8) are valid
    // (chroma: the vector halves): |candidate| <= margin_y - 20 vertically, margin_x - 12 horizontally.  DOWNWARDS a row's candidates must stay inside the
    // reference rows <= r + 1 (the row is computed one reference row before the host may start it): the view's lines end 8 above the last row's end, so a luma
    // block may reach 64 - 8 - 2 = 54 lines, a chroma block 32 - 8 - 2 = 22 = luma 42: candidate <= 42 (luma only: 54)
    const int maxCx = p->margin_x - p->window - 12, maxCy = (hasC && p->margin_y_c * 2 < p->margin_y ? p->margin_y_c * 2 : p->margin_y) - p->window - 20;
    const int maxCyDownRaw = (hasC ? 42 : 54) - p->window;
    if (maxCx < 0 || maxCy < 0 || maxCyDownRaw < 0) { set_error("cost_stream_create: margins %d / %d too small for a window of +-%d", p->margin_x, p->margin_y, p->window); return X265HIP_EINVAL; }
    int rc = ensure_device();
    if (rc) return rc;
    if (p->device_plus_1 < 0 || p->device_plus_1 > x265hip_device_count()) { set_error("cost_stream_create: device %d of %d", p->device_plus_1 - 1, x265hip_device_count()); return X265HIP_ENODEV; }
    if (p->device_plus_1 > 0 && (rc = x265hip_init(p->device_plus_1 - 1))) return rc;
    CS* s = new (std::nothrow) CS;
    if (!s) { set_error("cost_stream_create: out of memory"); return X265HIP_EINVAL; }
    s->prm = *p;
    s->bpp = p->depth == 8 ? 1 : 2;
    s->ctusW = p->width / 64; s->ctuRows = p->height / 64;
    s->nplanes = hasC ? 3 : 1;
    s->bandRows = p->band_rows > 0 ? p->band_rows : 8;
    if (s->bandRows > s->ctuRows) s->bandRows = s->ctuRows;
    s->npu = x265hip_cost_pu_count(p->shapes);
    s->recBytes = x265hip_cost_record_bytes(p->subme, p->sad_costs);
    s->ctuBytes = x265hip_cost_ctu_bytes(p->subme, p->shapes, p->candidates, p->sad_costs);
    s->rowBytes = s->ctuBytes * s->ctusW;
    s->maxCx = maxCx; s->maxCy = maxCy;
    if (p->centre_range && (s->maxCx > p->centre_range)) s->maxCx = p->centre_range;
    if (p->centre_range && (s->maxCy > p->centre_range)) s->maxCy = p->centre_range;
    s->maxCyDown = maxCyDownRaw < s->maxCy ? maxCyDownRaw : s->maxCy;
    s->pitch[0] = (size_t)p->stride * s->bpp; s->pitch[1] = (size_t)p->stride_c * s->bpp;
    s->rows[0] = p->height + 2 * p->margin_y; s->rows[1] = hasC ? p->height / 2 + 2 * p->margin_y_c : 0;
    s->planeBytes[0] = s->pitch[0] * s->rows[0]; s->planeBytes[1] = s->pitch[1] * s->rows[1];
    s->margin[0] = p->margin_y; s->margin[1] = p->margin_y_c;
    s->ctuLines[0] = 64; s->ctuLines[1] = 32;
    s->nph[0] = 15; s->nph[1] = 63;
    if (hipGetDevice(&s->device) != hipSuccess) s->device = 0;
#define CS_TRY(expr) do { if (check_hip((expr), #expr)) { cs_free(s); delete s; return X265HIP_ENODEV; } } while (0)
    CS_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    s->pics = std::vector<CS::Pic>(p->pictures);
    for (auto& pc : s->pics)
    {
        pc.staged.assign(s->ctuRows, 0);
        for (int i = 0; i < s->nplanes; i++)
        {
            const int k = i ? 1 : 0;
            CS_TRY(hipHostMalloc((void**)&pc.stage[i], s->planeBytes[k], hipHostMallocDefault));
            CS_TRY(hipMalloc((void**)&pc.dSrc[i], s->planeBytes[k] + 256));
            CS_TRY(hipMemset(pc.dSrc[i], 0, s->planeBytes[k] + 256));
        }
    }
    s->views = std::vector<CS::View>(p->views);
    for (auto& v : s->views)
        for (int i = 0; i < s->nplanes; i++)
        {
            const int k = i ? 1 : 0;
            CS_TRY(hipMalloc((void**)&v.dW[i], s->planeBytes[k] + 256));
            CS_TRY(hipMemset(v.dW[i], 0, s->planeBytes[k] + 256));
            CS_TRY(hipMalloc((void**)&v.dOut[i], s->planeBytes[k] * s->nph[k] + 256));
        }
    s->slots = std::vector<CS::Slot>(p->slots);
    for (auto& sl : s->slots)
    {
        CS_TRY(hipHostMalloc((void**)&sl.tables, s->rowBytes * s->ctuRows, hipHostMallocDefault));
        CS_TRY(hipHostMalloc((void**)&sl.hMvCost, (2 * (size_t)p->window + 1) * sizeof(uint16_t), hipHostMallocDefault));
        CS_TRY(hipMalloc((void**)&sl.dMvCost, (2 * (size_t)p->window + 1) * sizeof(uint16_t)));
        sl.ready = new (std::nothrow) std::atomic<int>[s->ctuRows];
        if (!sl.ready) { set_error("cost_stream_create: out of memory"); cs_free(s); delete s; return X265HIP_EINVAL; }
        for (int r = 0; r < s->ctuRows; r++) sl.ready[r].store(0);
    }
    const size_t nctuBand = (size_t)s->bandRows * s->ctusW;
    CS_TRY(hipMalloc((void**)&s->dSurf, nctuBand * x265hip_surf_ctu_bytes(X265HIP_SURF_I32, p->window) + 256));
    CS_TRY(hipMalloc((void**)&s->dBest, nctuBand * 85 * sizeof(unsigned long long)));
    CS_TRY(hipMalloc((void**)&s->dZeroCost, (2 * (size_t)(p->centre_range ? p->centre_range : 1) + 1) * sizeof(uint16_t)));
    CS_TRY(hipMemset(s->dZeroCost, 0, (2 * (size_t)(p->centre_range ? p->centre_range : 1) + 1) * sizeof(uint16_t)));
    CS_TRY(hipMalloc((void**)&s->dCentres, nctuBand * 4));
    CS_TRY(hipMalloc((void**)&s->dCand, nctuBand * s->npu * p->candidates * 4));
    CS_TRY(hipMalloc((void**)&s->dTables, nctuBand * s->ctuBytes));

    // the device memsets are queued on the null stream and the worker's stream is non-blocking: wait for the fills here, once
    CS_TRY(hipDeviceSynchronize());
#undef CS_TRY
    s->worker = std::thread(cs_worker, s);
    *out = s;
    return 0;
}

void x265hip_cost_stream_destroy(x265hip_cost_stream* s)
{
    if (!s) return;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->stop = true;
    }
    s->cv.notify_all();
    if (s->worker.joinable()) s->worker.join();
    (void)hipStreamSynchronize(s->stream);
    cs_free(s);
    delete s;
}

int x265hip_cost_stream_picture_rows(x265hip_cost_stream* s, uint64_t key, const void* luma_buf, const void* cb_buf, const void* cr_buf, int ctu_row0, int ctu_rows)
{
    if (!s || !luma_buf || (s->nplanes > 1 && (!cb_buf || !cr_buf)) || ctu_row0 < 0 || ctu_rows < 1 || ctu_row0 + ctu_rows > s->ctuRows)
    { set_error("cost_stream_picture_rows: bad argument"); return X265HIP_EINVAL; }
    int idx;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        idx = cs_find_or_make_picture(s, key);
        if (idx < 0) { set_error("cost_stream_picture_rows: every picture entry is still read (pictures = %d)", (int)s->pics.size()); return X265HIP_EBUSY; }
        s->pics[idx].busy++;
    }
    CS::Pic& pc = s->pics[idx];
    const void* bufs[3] = { luma_buf, cb_buf, cr_buf };
    for (int pl = 0; pl < s->nplanes; pl++)
    {
        const int k = pl ? 1 : 0;
        int y0, y1;
        cs_lines(s, k, ctu_row0, ctu_rows, y0, y1);
        memcpy(pc.stage[pl] + (size_t)y0 * s->pitch[k], (const uint8_t*)bufs[pl] + (size_t)y0 * s->pitch[k], (size_t)(y1 - y0) * s->pitch[k]);
    }
    {
        std::lock_guard<std::mutex> lk(s->mu);
        pc.busy--;
        if (pc.used && pc.key == key)
            for (int r = ctu_row0; r < ctu_row0 + ctu_rows; r++) pc.staged[r] = 1;
        s->dirty = true;
    }
    s->cv.notify_one();
    return 0;
}

int x265hip_cost_stream_pair_open(x265hip_cost_stream* s, int slot, uint64_t fenc_key, uint64_t ref_key, const x265hip_weight* w, unsigned planes_weighted, const uint16_t* mv_cost)
{
    if (!s || slot < 0 || slot >= (int)s->slots.size()) { set_error("cost_stream_pair_open: bad slot"); return X265HIP_EINVAL; }
    unsigned mask = w ? planes_weighted & ((1u << s->nplanes) - 1) : 0;
    for (int c = 0; c < s->nplanes; c++)
        if ((mask & (1u << c)) && (w[c].shift < 14 - s->prm.depth || w[c].shift > 31))
        { set_error("cost_stream_pair_open: plane %d shift %d (it includes the 14 - depth correction of weight_pp)", c, w[c].shift); return X265HIP_EINVAL; }
    int gen;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        CS::Slot& sl = s->slots[slot];
        const bool was = sl.active;
        sl.active = false;                                    // the slot's previous pair no longer holds anything
        const int fenc = cs_find_or_make_picture(s, fenc_key);
        if (fenc >= 0) s->pics[fenc].busy++;                  // not the victim of the next line
        const int ref = fenc < 0 ? -1 : cs_find_or_make_picture(s, ref_key);
        if (fenc >= 0) s->pics[fenc].busy--;
        const int view = ref < 0 ? -1 : cs_find_or_make_view(s, ref, w, mask);
        if (view < 0)
        {
            sl.active = was;
            set_error("cost_stream_pair_open: no %s entry free (pictures = %d, views = %d)", ref < 0 ? "picture" : "view", (int)s->pics.size(), (int)s->views.size());
            return X265HIP_EBUSY;
        }
        if (++sl.generation <= 0) sl.generation = 1;
        for (int r = 0; r < s->ctuRows; r++) sl.ready[r].store(0, std::memory_order_release);      // before anything is rewritten
        sl.fenc = fenc; sl.fencEpoch = s->pics[fenc].epoch; sl.view = view; sl.viewStamp = s->views[view].stamp;
        sl.nextRow = 0; sl.active = true; sl.bandLimit = 1;
        sl.hasCost = mv_cost != nullptr;
        if (mv_cost) memcpy(sl.hMvCost, mv_cost, (2 * (size_t)s->prm.window + 1) * sizeof(uint16_t));
        gen = sl.generation;
        s->pairsOpened++;
        s->dirty = true;
    }
    s->cv.notify_one();
    return gen;
}

const void* x265hip_cost_stream_tables(x265hip_cost_stream* s, int slot) { return (s && slot >= 0 && slot < (int)s->slots.size()) ? s->slots[slot].tables : nullptr; }

const volatile int* x265hip_cost_stream_ready(x265hip_cost_stream* s, int slot)
{
    return (s && slot >= 0 && slot < (int)s->slots.size()) ? reinterpret_cast<const volatile int*>(s->slots[slot].ready) : nullptr;
}

int x265hip_cost_stream_stats(x265hip_cost_stream* s, x265hip_cost_stream_stats_t* st)
{
    if (!s || !st) { set_error("cost_stream_stats: NULL"); return X265HIP_EINVAL; }
    st->pairs_opened = s->pairsOpened; st->pairs_completed = s->pairsCompleted; st->bands = s->bands; st->rows_served = s->rowsServed; st->rows_uploaded = s->rowsUploaded;
    st->failed = s->failed; st->stale_pairs = s->stalePairs; st->views_opened = s->viewsOpened; st->views_shared = s->viewsShared; st->lines_weighted = s->linesWeighted;
    st->us_busy = s->usBusy; st->bytes_downloaded = s->bytesDown; st->bytes_uploaded = s->bytesUp; st->table_bytes = s->rowBytes * s->ctuRows;
    if (s->failed) set_error("cost_stream worker: %s", s->workerError);
    return 0;
}

} // extern "C"
