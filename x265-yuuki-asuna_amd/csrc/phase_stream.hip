// phase_stream.hip - the ROW-GRANULAR consumer of x265hip_phase_planes, for hosts that encode several pictures at once.
//
// x265hip_phase_cache (csrc/phase_cache.hip) takes a finished reference picture.  Under the reference's frame threads a picture is
// searched while it is still being reconstructed, CTU row by CTU row (Frame::m_reconRowFlag, encoder/framefilter.cpp:664; consumers wait
// row by row, encoder/frameencoder.cpp:852-868).  Two kinds of object (round 4):
//
//   PICTURE  a reconstructed picture named by a key: the producer hands every CTU row over where it raises the flag
//            (x265hip_phase_stream_picture_rows: copied into pinned staging inside the call); the worker uploads the rows.
//   VIEW     a slot = every fractional phase of ONE picture, optionally WEIGHTED first: x265's default --weightp lets a slice search the
//            plane MotionReference::applyWeight materialises (primitives.weight_pp of every finished row, encoder/reference.cpp:119-178,
//            encoder/frameencoder.cpp:865-866), and interpolating a weighted plane is not weighting an interpolated one - so a weighted
//            reference is a view of its own, keyed by (picture, weight triple).  A consumer opens the view the first time a search refers
//            to it (x265hip_phase_stream_view_open); the worker weights the picture's rows on the device as they arrive (margins included:
//            a replicated border sample weights to the replicated weighted sample), interpolates every fractional phase of the lines that
//            became computable (a line needs 3 source lines above and up to 8 below it, so the last 8 lines of a row wait for the next row)
//            and copies those lines of all 15 luma / 2 x 63 chroma planes into the view's pinned host memory.
//
// progress[0] (luma) and progress[1] (chroma) of a view = generation << 32 | lines finished, counted from the top of the buffer: a block
// whose last line is below that is the host's to interpolate itself - same samples either way (MotionEstimate::subpelCompare,
// motion.cpp:1571-1664; Predict::predInterLumaPixel / predInterChromaPixel, predict.cpp:261-351).  Readers check progress before AND after
// reading (view_open clears it before a slot's planes can be rewritten).
// x265hip_phase_stream_open / _rows (round 3: one anonymous picture per slot, opened by the producer) are kept on top of the two.
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
__global__ void __launch_bounds__(256) ps_weight_lines_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t ndw, int w0, int round, int shift,
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

} // namespace

struct x265hip_phase_stream
{
    x265hip_phase_stream_params prm;
    int bpp, device, ctuRows, nplanes;
    size_t planeBytes[2], pitch[2];         // luma / chroma source plane
    int rows[2], margin[2], ctuLines[2], nph[2];
    hipStream_t stream = nullptr;
    struct Pic
    {
        uint64_t key = 0; bool used = false; uint32_t epoch = 0; uint64_t stamp = 0; int busy = 0;
        uint8_t* stage[3] = { nullptr, nullptr, nullptr };      // pinned source planes, rows staged by the host threads
        uint8_t* dSrc[3] = { nullptr, nullptr, nullptr };
        std::vector<uint8_t> staged;                            // per CTU row
        int nextRow = 0;                                        // rows [0, nextRow) are uploaded (or queued on the stream)
    };
    struct Slot
    {
        uint8_t* dW[3] = { nullptr, nullptr, nullptr };         // the picture's planes weighted (only the planes of the mask)
        uint8_t* dOut[3] = { nullptr, nullptr, nullptr };       // every phase plane of the view on the device
        uint8_t* out[3] = { nullptr, nullptr, nullptr };        // ... and in pinned host memory
        std::atomic<uint64_t> progress[2];
        int generation = 0;
        int pic = -1; uint32_t picEpoch = 0; bool active = false;
        unsigned mask = 0; x265hip_weight w[3] = {};
        int rowsSeen = 0;                                       // rows [0, rowsSeen) of the picture are worked into this view
        int done[2] = { 0, 0 };                                 // buffer lines finished per plane kind
    };
    std::vector<Pic> pics;
    std::vector<Slot> slots;
    uint64_t clock = 0, anon = 0;
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false, dirty = false;
    std::thread worker;
    std::atomic<uint64_t> opened{0}, completed{0}, bands{0}, failed{0}, usBusy{0}, bytesDown{0}, bytesUp{0}, weightedViews{0}, linesWeighted{0};
    char workerError[256] = "";
};

namespace {

typedef x265hip_phase_stream PS;

double ps_now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// buffer lines [y0, y1) of plane kind k (0 luma, 1 chroma) that CTU rows [r0, r0 + n) occupy; margins travel with the first / last row
inline void ps_lines(const PS* s, int k, int r0, int n, int& y0, int& y1)
{
    y0 = r0 == 0 ? 0 : s->margin[k] + r0 * s->ctuLines[k];
    y1 = r0 + n == s->ctuRows ? s->rows[k] : s->margin[k] + (r0 + n) * s->ctuLines[k];
}

struct Upload { int pic, r0, r1; };
struct ViewJob { int slot, gen, pic, r0, r1; int done[2]; unsigned mask; x265hip_weight w[3]; };      // the view's state is snapshot under the lock

int run_round(PS* s, const std::vector<Upload>& ups, const std::vector<ViewJob>& jobs)
{
    X265HIP_TRY(hipSetDevice(s->device));
    apply_wait_policy(s->device);
    for (const Upload& u : ups)
        for (int pl = 0; pl < s->nplanes; pl++)
        {
            const int k = pl ? 1 : 0;
            int y0, y1;
            ps_lines(s, k, u.r0, u.r1 - u.r0, y0, y1);
            X265HIP_TRY(hipMemcpyAsync(s->pics[u.pic].dSrc[pl] + (size_t)y0 * s->pitch[k], s->pics[u.pic].stage[pl] + (size_t)y0 * s->pitch[k],
                                       (size_t)(y1 - y0) * s->pitch[k], hipMemcpyHostToDevice, s->stream));
            s->bytesUp += (size_t)(y1 - y0) * s->pitch[k];
        }
    std::vector<int> newDone(jobs.size() * 2);
    for (size_t j = 0; j < jobs.size(); j++)
    {
        const ViewJob& job = jobs[j];
        PS::Slot& sl = s->slots[job.slot];
        newDone[2 * j] = job.done[0]; newDone[2 * j + 1] = job.done[1];
        for (int pl = 0; pl < s->nplanes; pl++)
        {
            const int k = pl ? 1 : 0;
            int y0, y1;
            ps_lines(s, k, job.r0, job.r1 - job.r0, y0, y1);
            const uint8_t* src = s->pics[job.pic].dSrc[pl];
            if (job.mask & (1u << pl))
            {
                const size_t off = (size_t)y0 * s->pitch[k], ndw = (size_t)(y1 - y0) * s->pitch[k] / 4;
                const x265hip_weight& w = job.w[pl];
                const int correction = 14 - s->prm.depth, maxVal = (1 << s->prm.depth) - 1;
                size_t blocks = (ndw + 255) / 256;
                if (blocks > 8192) blocks = 8192;
                if (s->bpp == 1)
                    hipLaunchKernelGGL(ps_weight_lines_kernel<uint8_t>, dim3((unsigned)blocks), dim3(256), 0, s->stream, (const uint32_t*)(src + off), (uint32_t*)(sl.dW[pl] + off),
                                       ndw, w.w0, w.round, w.shift, w.offset, correction, maxVal);
                else
                    hipLaunchKernelGGL(ps_weight_lines_kernel<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, s->stream, (const uint32_t*)(src + off), (uint32_t*)(sl.dW[pl] + off),
                                       ndw, w.w0, w.round, w.shift, w.offset, correction, maxVal);
                X265HIP_TRY(hipGetLastError());
                s->linesWeighted += (uint64_t)(y1 - y0);
                src = sl.dW[pl];
            }
            // producible now: lines [max(done, 4), y1 - 8) - every source line below y1 is on the device
            const int b0 = job.done[k] < 4 ? 4 : job.done[k], b1 = y1 - 8;
            if (b1 - b0 < 4) continue;
            const size_t lineOff = (size_t)(b0 - 4) * s->pitch[k];
            int rc = phase_planes_launch(s->prm.depth, k, src + lineOff, sl.dOut[pl] + lineOff, k ? s->prm.stride_c : s->prm.stride, b1 - b0 + 12,
                                         s->planeBytes[k], s->stream);
            if (rc) return rc;
            const size_t o = (size_t)b0 * s->pitch[k], w = (size_t)(b1 - b0) * s->pitch[k];
            X265HIP_TRY(hipMemcpy2DAsync(sl.out[pl] + o, s->planeBytes[k], sl.dOut[pl] + o, s->planeBytes[k], w, s->nph[k], hipMemcpyDeviceToHost, s->stream));
            s->bytesDown += w * s->nph[k];
            newDone[2 * j + k] = b1;
        }
    }
    X265HIP_TRY(hipStreamSynchronize(s->stream));
    {
        std::lock_guard<std::mutex> lk(s->mu);               // against view_open(): a reopened slot keeps its cleared progress
        for (size_t j = 0; j < jobs.size(); j++)
        {
            const ViewJob& job = jobs[j];
            PS::Slot& sl = s->slots[job.slot];
            if (sl.generation != job.gen) continue;
            for (int k = 0; k < 2; k++)
            {
                sl.done[k] = newDone[2 * j + k];
                sl.progress[k].store((uint64_t)(uint32_t)job.gen << 32 | (uint32_t)sl.done[k], std::memory_order_release);
            }
            if (job.r1 == s->ctuRows) s->completed++;
        }
    }
    s->bands += jobs.size();
    return 0;
}

void ps_worker(PS* s)
{
    for (;;)
    {
        std::vector<Upload> ups;
        std::vector<ViewJob> jobs;
        {
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv.wait(lk, [s] { return s->stop || s->dirty; });
            if (s->stop) return;
            s->dirty = false;
            for (int i = 0; i < (int)s->pics.size(); i++)
            {
                PS::Pic& pc = s->pics[i];
                if (!pc.used) continue;
                int r1 = pc.nextRow;
                while (r1 < s->ctuRows && pc.staged[r1]) r1++;          // rows are interpolated top to bottom: only a contiguous prefix is useful
                // busy: the entry stays this picture's until the round is synchronised - a view that closes with its last rows (below) no longer
                // holds it, and a host thread staging a NEW picture into the same pinned memory would race the upload still reading it
                if (r1 > pc.nextRow) { ups.push_back({ i, pc.nextRow, r1 }); pc.nextRow = r1; pc.busy++; }
            }
            for (int i = 0; i < (int)s->slots.size(); i++)
            {
                PS::Slot& sl = s->slots[i];
                if (!sl.active) continue;
                const PS::Pic& pc = s->pics[sl.pic];
                if (!pc.used || pc.epoch != sl.picEpoch) { sl.active = false; continue; }      // the picture went away: what is finished stays valid
                if (pc.nextRow > sl.rowsSeen)
                {
                    ViewJob job = { i, sl.generation, sl.pic, sl.rowsSeen, pc.nextRow, { sl.done[0], sl.done[1] }, sl.mask, { sl.w[0], sl.w[1], sl.w[2] } };
                    jobs.push_back(job);
                    s->pics[sl.pic].busy++;
                    sl.rowsSeen = pc.nextRow;
                    if (sl.rowsSeen == s->ctuRows) sl.active = false;
                }
            }
        }
        if (ups.empty() && jobs.empty()) continue;
        const double t0 = ps_now_us();
        if (run_round(s, ups, jobs))
        {
            s->failed++;
            snprintf(s->workerError, sizeof(s->workerError), "%s", x265hip_last_error());
        }
        {
            // the round has been synchronised (or has failed): its pictures may be recycled again
            std::lock_guard<std::mutex> lk(s->mu);
            for (const Upload& u : ups) s->pics[u.pic].busy--;
            for (const ViewJob& j : jobs) s->pics[j.pic].busy--;
        }
        s->usBusy += (uint64_t)(ps_now_us() - t0);
    }
}

void ps_free(PS* s)
{
    for (auto& pc : s->pics)
        for (int i = 0; i < 3; i++)
        {
            if (pc.stage[i]) (void)hipHostFree(pc.stage[i]);
            if (pc.dSrc[i]) (void)hipFree(pc.dSrc[i]);
        }
    for (auto& sl : s->slots)
        for (int i = 0; i < 3; i++)
        {
            if (sl.out[i]) (void)hipHostFree(sl.out[i]);
            if (sl.dW[i]) (void)hipFree(sl.dW[i]);
            if (sl.dOut[i]) (void)hipFree(sl.dOut[i]);
        }
    if (s->stream) (void)hipStreamDestroy(s->stream);
}

// index of the picture named `key`, created when it is new: least recently used entry no view is still being fed from; -1 = all held
int ps_find_or_make_picture(PS* s, uint64_t key)
{
    for (int i = 0; i < (int)s->pics.size(); i++)
        if (s->pics[i].used && s->pics[i].key == key) { s->pics[i].stamp = ++s->clock; return i; }
    int victim = -1;
    for (int i = 0; i < (int)s->pics.size(); i++)
    {
        PS::Pic& p = s->pics[i];
        if (!p.used) { victim = i; break; }
        if (p.busy) continue;
        bool held = false;
        for (const auto& sl : s->slots) held |= sl.active && sl.pic == i && sl.picEpoch == p.epoch;
        if (!held && (victim < 0 || p.stamp < s->pics[victim].stamp)) victim = i;
    }
    if (victim < 0) return -1;
    PS::Pic& p = s->pics[victim];
    p.used = true; p.key = key; p.epoch++; p.stamp = ++s->clock; p.busy = 0;
    std::fill(p.staged.begin(), p.staged.end(), (uint8_t)0);
    p.nextRow = 0;
    return victim;
}

int ps_view_open_locked(PS* s, int slot, int pic, const x265hip_weight* w, unsigned mask)
{
    PS::Slot& sl = s->slots[slot];
    if (++sl.generation <= 0) sl.generation = 1;
    sl.progress[0].store(0, std::memory_order_release); sl.progress[1].store(0, std::memory_order_release);      // before anything is rewritten
    sl.pic = pic; sl.picEpoch = s->pics[pic].epoch;
    sl.mask = w ? mask : 0;
    for (int c = 0; c < 3; c++) sl.w[c] = (w && (mask & (1u << c))) ? w[c] : x265hip_weight{ 0, 0, 0, 0 };
    sl.rowsSeen = 0; sl.done[0] = sl.done[1] = 0;
    sl.active = true;
    s->opened++;
    if (sl.mask) s->weightedViews++;
    s->dirty = true;
    return sl.generation;
}

} // namespace

extern "C" {

int x265hip_phase_stream_create(x265hip_phase_stream** out, const x265hip_phase_stream_params* p)
{
    if (!out || !p) { set_error("phase_stream_create: NULL argument"); return X265HIP_EINVAL; }
    *out = nullptr;
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("phase_stream_create: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->stride <= 0 || (p->stride & 3) || p->ctu_rows < 1 || p->margin_y < 8 || (p->margin_y & 3))
    { set_error("phase_stream_create: luma pitch %ld / %d CTU rows / margin %d", (long)p->stride, p->ctu_rows, p->margin_y); return X265HIP_EINVAL; }
    if (p->rows_c < 0 || (p->rows_c > 0 && (p->stride_c <= 0 || (p->stride_c & 3) || p->margin_y_c < 8 || (p->margin_y_c & 3) ||
                                            p->rows_c != p->ctu_rows * 32 + 2 * p->margin_y_c)))
    { set_error("phase_stream_create: chroma geometry (4:2:0: rows_c = ctu_rows * 32 + 2 * margin_y_c)"); return X265HIP_EINVAL; }
    if (p->rows != p->ctu_rows * 64 + 2 * p->margin_y) { set_error("phase_stream_create: rows %d != ctu_rows * 64 + 2 * margin_y", p->rows); return X265HIP_EINVAL; }
    if (p->slots < 1 || p->slots > 64 || p->pictures < 0 || p->pictures > 256) { set_error("phase_stream_create: slots %d out of [1,64] / pictures %d out of [0,256]", p->slots, p->pictures); return X265HIP_EINVAL; }
    int rc = ensure_device();
    if (rc) return rc;
    if (p->device_plus_1 < 0 || p->device_plus_1 > x265hip_device_count()) { set_error("phase_stream_create: device %d of %d", p->device_plus_1 - 1, x265hip_device_count()); return X265HIP_ENODEV; }
    if (p->device_plus_1 > 0 && (rc = x265hip_init(p->device_plus_1 - 1))) return rc;       // pinned to a GPU: current for the creating thread from here on
    PS* s = new (std::nothrow) PS;
    if (!s) { set_error("phase_stream_create: out of memory"); return X265HIP_EINVAL; }
    s->prm = *p;
    s->bpp = p->depth == 8 ? 1 : 2;
    s->ctuRows = p->ctu_rows;
    s->nplanes = p->rows_c > 0 ? 3 : 1;
    s->pitch[0] = (size_t)p->stride * s->bpp; s->pitch[1] = (size_t)p->stride_c * s->bpp;
    s->rows[0] = p->rows; s->rows[1] = p->rows_c;
    s->planeBytes[0] = s->pitch[0] * p->rows; s->planeBytes[1] = s->pitch[1] * p->rows_c;
    s->margin[0] = p->margin_y; s->margin[1] = p->margin_y_c;
    s->ctuLines[0] = 64; s->ctuLines[1] = 32;
    s->nph[0] = 15; s->nph[1] = 63;
    if (hipGetDevice(&s->device) != hipSuccess) s->device = 0;
#define PS_TRY(expr) do { if (check_hip((expr), #expr)) { ps_free(s); delete s; return X265HIP_ENODEV; } } while (0)
    PS_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    s->pics = std::vector<PS::Pic>(p->pictures ? p->pictures : p->slots);
    for (auto& pc : s->pics)
    {
        pc.staged.assign(s->ctuRows, 0);
        for (int i = 0; i < s->nplanes; i++)
        {
            const int k = i ? 1 : 0;
            PS_TRY(hipHostMalloc((void**)&pc.stage[i], s->planeBytes[k], hipHostMallocDefault));
            PS_TRY(hipMalloc((void**)&pc.dSrc[i], s->planeBytes[k] + 256));
            PS_TRY(hipMemset(pc.dSrc[i], 0, s->planeBytes[k] + 256));
        }
    }
    s->slots = std::vector<PS::Slot>(p->slots);
    for (auto& sl : s->slots)
    {
        sl.progress[0].store(0); sl.progress[1].store(0);
        for (int i = 0; i < s->nplanes; i++)
        {
            const int k = i ? 1 : 0;
            PS_TRY(hipMalloc((void**)&sl.dW[i], s->planeBytes[k] + 256));
            PS_TRY(hipMemset(sl.dW[i], 0, s->planeBytes[k] + 256));
            PS_TRY(hipMalloc((void**)&sl.dOut[i], s->planeBytes[k] * s->nph[k]));
            PS_TRY(hipHostMalloc((void**)&sl.out[i], s->planeBytes[k] * s->nph[k], hipHostMallocDefault));
        }
    }
    // hipMemset returns before a DEVICE memset has run (it is queued on the null stream), and the worker's stream is non-blocking: an
    // upload of the first rows could be overtaken by the zero fill queued before it (seen as planes that intermittently did not equal
    // the whole-picture planes when other work sat in front of the memsets, round 4) - wait for the fills here, once
    PS_TRY(hipDeviceSynchronize());
#undef PS_TRY
    s->worker = std::thread(ps_worker, s);
    *out = s;
    return 0;
}

void x265hip_phase_stream_destroy(x265hip_phase_stream* s)
{
    if (!s) return;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->stop = true;
    }
    s->cv.notify_all();
    if (s->worker.joinable()) s->worker.join();
    (void)hipStreamSynchronize(s->stream);
    ps_free(s);
    delete s;
}

/* CTU rows [ctu_row0, ctu_row0 + ctu_rows) of the picture named `key` are final in the three buffers (whole allocated planes; cb / cr
 * may be NULL when rows_c = 0); copied before the call returns.  X265HIP_EBUSY: every picture entry still feeds a view. */
int x265hip_phase_stream_picture_rows(x265hip_phase_stream* s, uint64_t key, const void* luma_buf, const void* cb_buf, const void* cr_buf, int ctu_row0, int ctu_rows)
{
    if (!s || !luma_buf || (s->prm.rows_c > 0 && (!cb_buf || !cr_buf)) || ctu_row0 < 0 || ctu_rows < 1 || ctu_row0 + ctu_rows > s->ctuRows)
    { set_error("phase_stream_picture_rows: bad argument"); return X265HIP_EINVAL; }
    int idx;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        idx = ps_find_or_make_picture(s, key);
        if (idx < 0) { set_error("phase_stream_picture_rows: every picture entry still feeds a view (pictures = %d)", (int)s->pics.size()); return X265HIP_EBUSY; }
        s->pics[idx].busy++;
    }
    PS::Pic& pc = s->pics[idx];
    const void* bufs[3] = { luma_buf, cb_buf, cr_buf };
    for (int pl = 0; pl < s->nplanes; pl++)
    {
        const int k = pl ? 1 : 0;
        int y0, y1;
        ps_lines(s, k, ctu_row0, ctu_rows, y0, y1);
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

/* `slot` becomes the view of picture `key` - w = NULL: as reconstructed; otherwise plane c (0 luma, 1 Cb, 2 Cr) is weighted with w[c]
 * (the arguments of primitives.weight_pp, round / shift including the 14 - depth correction) when bit c of planes_weighted is set.
 * The picture's rows may arrive before or after.  Returns the slot's new GENERATION (> 0); progress is cleared before anything is
 * rewritten. */
int x265hip_phase_stream_view_open(x265hip_phase_stream* s, int slot, uint64_t key, const x265hip_weight* w, unsigned planes_weighted)
{
    if (!s || slot < 0 || slot >= (int)s->slots.size()) { set_error("phase_stream_view_open: bad slot"); return X265HIP_EINVAL; }
    if (w)
        for (int c = 0; c < s->nplanes; c++)
            if ((planes_weighted & (1u << c)) && (w[c].shift < 14 - s->prm.depth || w[c].shift > 31))
            { set_error("phase_stream_view_open: plane %d shift %d (it includes the 14 - depth correction of weight_pp)", c, w[c].shift); return X265HIP_EINVAL; }
    int gen;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        const bool was = s->slots[slot].active;
        s->slots[slot].active = false;                        // the slot's previous view no longer holds its picture
        const int pic = ps_find_or_make_picture(s, key);
        if (pic < 0) { s->slots[slot].active = was; set_error("phase_stream_view_open: no picture entry free (pictures = %d)", (int)s->pics.size()); return X265HIP_EBUSY; }
        gen = ps_view_open_locked(s, slot, pic, w, planes_weighted & ((1u << s->nplanes) - 1));
    }
    s->cv.notify_one();
    return gen;
}

/* round-3 entry: a new reconstructed picture takes `slot` (an anonymous picture of its own, unweighted view) */
int x265hip_phase_stream_open(x265hip_phase_stream* s, int slot)
{
    if (!s || slot < 0 || slot >= (int)s->slots.size()) { set_error("phase_stream_open: bad slot"); return X265HIP_EINVAL; }
    uint64_t key;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        key = (1ull << 63) | ((uint64_t)slot << 40) | (++s->anon & 0xffffffffffull);
    }
    return x265hip_phase_stream_view_open(s, slot, key, nullptr, 0);
}

int x265hip_phase_stream_rows(x265hip_phase_stream* s, int slot, int gen, const void* luma_buf, const void* cb_buf, const void* cr_buf, int ctu_row0, int ctu_rows)
{
    if (!s || slot < 0 || slot >= (int)s->slots.size()) { set_error("phase_stream_rows: bad argument"); return X265HIP_EINVAL; }
    uint64_t key;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        PS::Slot& sl = s->slots[slot];
        if (sl.generation != gen || sl.pic < 0 || !s->pics[sl.pic].used || s->pics[sl.pic].epoch != sl.picEpoch)
        { set_error("phase_stream_rows: slot %d was reopened (generation %d, not %d)", slot, sl.generation, gen); return X265HIP_EBUSY; }
        key = s->pics[sl.pic].key;
    }
    return x265hip_phase_stream_picture_rows(s, key, luma_buf, cb_buf, cr_buf, ctu_row0, ctu_rows);
}

const void* x265hip_phase_stream_planes(x265hip_phase_stream* s, int slot, int plane)
{
    return (s && slot >= 0 && slot < (int)s->slots.size() && plane >= 0 && plane < 3) ? s->slots[slot].out[plane] : nullptr;
}

const volatile uint64_t* x265hip_phase_stream_progress(x265hip_phase_stream* s, int slot)
{
    return (s && slot >= 0 && slot < (int)s->slots.size()) ? reinterpret_cast<const volatile uint64_t*>(s->slots[slot].progress) : nullptr;
}

int x265hip_phase_stream_stats(x265hip_phase_stream* s, x265hip_phase_stream_stats_t* st)
{
    if (!s || !st) { set_error("phase_stream_stats: NULL"); return X265HIP_EINVAL; }
    st->opened = s->opened; st->completed = s->completed; st->bands = s->bands; st->failed = s->failed; st->us_busy = s->usBusy;
    st->bytes_downloaded = s->bytesDown; st->bytes_uploaded = s->bytesUp;
    st->bytes_per_picture = s->planeBytes[0] * 15 + 2 * s->planeBytes[1] * 63;
    st->weighted_views = s->weightedViews; st->lines_weighted = s->linesWeighted;
    if (s->failed) set_error("phase_stream worker: %s", s->workerError);
    return 0;
}

} // extern "C"
