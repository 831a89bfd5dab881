// phase_stream.hip - the ROW-GRANULAR consumer of x265hip_phase_planes, for hosts that encode several pictures at once.
//
// x265hip_phase_cache (csrc/phase_cache.hip) takes a finished reference picture.  Under the reference's frame threads a picture is
// searched while it is still being reconstructed, CTU row by CTU row (Frame::m_reconRowFlag, encoder/framefilter.cpp:664; consumers wait
// row by row, encoder/frameencoder.cpp:852-868).  Here the producer side OPENS a slot when a reconstructed picture's first row is final and
// hands every further row over where it raises the flag; the worker uploads the rows, interpolates every fractional phase of the lines
// that became computable (a line needs 3 source lines above and up to 8 below it, so the last 8 lines of a row wait for the next row)
// and copies those lines of all 15 luma / 2 x 63 chroma planes into pinned host memory.  progress[0] (luma) and progress[1] (chroma)
// = generation << 32 | lines finished, counted from the top of the buffer: a block whose last line is below that is the host's to
// interpolate itself - same samples either way (MotionEstimate::subpelCompare, motion.cpp:1571-1664; Predict::predInterLumaPixel /
// predInterChromaPixel, predict.cpp:261-351).  Readers check progress before AND after reading (open() clears it before a slot's
// planes can be rewritten).
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

struct x265hip_phase_stream
{
    x265hip_phase_stream_params prm;
    int bpp, device, ctuRows;
    size_t planeBytes[2], pitch[2];         // luma / chroma source plane
    int rows[2], margin[2], ctuLines[2], nph[2];
    hipStream_t stream = nullptr;
    struct Slot
    {
        uint8_t* stage[3] = { nullptr, nullptr, nullptr };      // pinned source planes, rows staged by the host threads
        uint8_t* dSrc[3] = { nullptr, nullptr, nullptr };
        uint8_t* dOut[3] = { nullptr, nullptr, nullptr };       // every phase plane of the picture on the device
        uint8_t* out[3] = { nullptr, nullptr, nullptr };        // ... and in pinned host memory
        std::atomic<uint64_t> progress[2];
        int generation = 0;
        std::vector<uint8_t> staged;                            // per CTU row
        int nextRow = 0;                                        // rows [0, nextRow) are uploaded
        int done[2] = { 0, 0 };                                 // buffer lines finished per plane kind
    };
    std::vector<Slot> slots;
    struct Job { int slot, gen; };
    std::deque<Job> queue;
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false;
    std::thread worker;
    std::atomic<uint64_t> opened{0}, completed{0}, bands{0}, failed{0}, usBusy{0}, bytesDown{0}, bytesUp{0};
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

int run_job(PS* s, const PS::Job& job)
{
    X265HIP_TRY(hipSetDevice(s->device));
    PS::Slot& sl = s->slots[job.slot];
    int r0, r1;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (sl.generation != job.gen) return 0;              // reopened: this picture's remaining rows are dropped
        r0 = sl.nextRow; r1 = r0;
        while (r1 < s->ctuRows && sl.staged[r1]) r1++;
        if (r1 == r0) return 0;
        sl.nextRow = r1;
    }
    const bool chroma = s->prm.rows_c > 0;
    int newDone[2] = { sl.done[0], sl.done[1] };
    for (int pl = 0; pl < (chroma ? 3 : 1); pl++)
    {
        const int k = pl ? 1 : 0;
        int y0, y1;
        ps_lines(s, k, r0, r1 - r0, y0, y1);
        X265HIP_TRY(hipMemcpyAsync(sl.dSrc[pl] + (size_t)y0 * s->pitch[k], sl.stage[pl] + (size_t)y0 * s->pitch[k], (size_t)(y1 - y0) * s->pitch[k],
                                   hipMemcpyHostToDevice, s->stream));
        s->bytesUp += (size_t)(y1 - y0) * s->pitch[k];
        // producible now: lines [max(done, 4), y1 - 8) - every source line below y1 is on the device
        const int b0 = sl.done[k] < 4 ? 4 : sl.done[k], b1 = y1 - 8;
        if (b1 - b0 < 4) continue;
        const size_t lineOff = (size_t)(b0 - 4) * s->pitch[k];
        int rc = phase_planes_launch(s->prm.depth, k, sl.dSrc[pl] + lineOff, sl.dOut[pl] + lineOff, k ? s->prm.stride_c : s->prm.stride, b1 - b0 + 12,
                                     s->planeBytes[k], s->stream);
        if (rc) return rc;
        const size_t o = (size_t)b0 * s->pitch[k], w = (size_t)(b1 - b0) * s->pitch[k];
        X265HIP_TRY(hipMemcpy2DAsync(sl.out[pl] + o, s->planeBytes[k], sl.dOut[pl] + o, s->planeBytes[k], w, s->nph[k], hipMemcpyDeviceToHost, s->stream));
        s->bytesDown += w * s->nph[k];
        newDone[k] = b1;
    }
    X265HIP_TRY(hipStreamSynchronize(s->stream));
    {
        std::lock_guard<std::mutex> lk(s->mu);               // against open(): a reopened slot keeps its cleared progress
        if (sl.generation == job.gen)
        {
            for (int k = 0; k < 2; k++)
            {
                sl.done[k] = newDone[k];
                sl.progress[k].store((uint64_t)(uint32_t)job.gen << 32 | (uint32_t)newDone[k], std::memory_order_release);
            }
            if (r1 == s->ctuRows) s->completed++;
        }
    }
    s->bands++;
    return 0;
}

void ps_worker(PS* s)
{
    for (;;)
    {
        PS::Job job;
        {
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv.wait(lk, [s] { return s->stop || !s->queue.empty(); });
            if (s->stop) return;
            job = s->queue.front();
            s->queue.pop_front();
        }
        const double t0 = ps_now_us();
        if (run_job(s, job))
        {
            s->failed++;
            snprintf(s->workerError, sizeof(s->workerError), "%s", x265hip_last_error());
        }
        s->usBusy += (uint64_t)(ps_now_us() - t0);
    }
}

void ps_free(PS* s)
{
    for (auto& sl : s->slots)
        for (int i = 0; i < 3; i++)
        {
            if (sl.stage[i]) (void)hipHostFree(sl.stage[i]);
            if (sl.out[i]) (void)hipHostFree(sl.out[i]);
            if (sl.dSrc[i]) (void)hipFree(sl.dSrc[i]);
            if (sl.dOut[i]) (void)hipFree(sl.dOut[i]);
        }
    if (s->stream) (void)hipStreamDestroy(s->stream);
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
    if (p->slots < 1 || p->slots > 64) { set_error("phase_stream_create: slots %d out of [1,64]", p->slots); return X265HIP_EINVAL; }
    int rc = ensure_device();
    if (rc) return rc;
    PS* s = new (std::nothrow) PS;
    if (!s) { set_error("phase_stream_create: out of memory"); return X265HIP_EINVAL; }
    s->prm = *p;
    s->bpp = p->depth == 8 ? 1 : 2;
    s->ctuRows = p->ctu_rows;
    s->pitch[0] = (size_t)p->stride * s->bpp; s->pitch[1] = (size_t)p->stride_c * s->bpp;
    s->rows[0] = p->rows; s->rows[1] = p->rows_c;
    s->planeBytes[0] = s->pitch[0] * p->rows; s->planeBytes[1] = s->pitch[1] * p->rows_c;
    s->margin[0] = p->margin_y; s->margin[1] = p->margin_y_c;
    s->ctuLines[0] = 64; s->ctuLines[1] = 32;
    s->nph[0] = 15; s->nph[1] = 63;
    if (hipGetDevice(&s->device) != hipSuccess) s->device = 0;
#define PS_TRY(expr) do { if (check_hip((expr), #expr)) { ps_free(s); delete s; return X265HIP_ENODEV; } } while (0)
    PS_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    s->slots = std::vector<PS::Slot>(p->slots);
    for (auto& sl : s->slots)
    {
        sl.progress[0].store(0); sl.progress[1].store(0);
        sl.staged.assign(s->ctuRows, 0);
        for (int i = 0; i < (p->rows_c > 0 ? 3 : 1); i++)
        {
            const int k = i ? 1 : 0;
            PS_TRY(hipHostMalloc((void**)&sl.stage[i], s->planeBytes[k], hipHostMallocDefault));
            PS_TRY(hipMalloc((void**)&sl.dSrc[i], s->planeBytes[k] + 256));
            PS_TRY(hipMemset(sl.dSrc[i], 0, s->planeBytes[k] + 256));
            PS_TRY(hipMalloc((void**)&sl.dOut[i], s->planeBytes[k] * s->nph[k]));
            PS_TRY(hipHostMalloc((void**)&sl.out[i], s->planeBytes[k] * s->nph[k], hipHostMallocDefault));
        }
    }
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

/* a new reconstructed picture takes `slot`: returns the slot's new GENERATION (> 0); progress is cleared before anything is rewritten */
int x265hip_phase_stream_open(x265hip_phase_stream* s, int slot)
{
    if (!s || slot < 0 || slot >= (int)s->slots.size()) { set_error("phase_stream_open: bad slot"); return X265HIP_EINVAL; }
    std::lock_guard<std::mutex> lk(s->mu);
    PS::Slot& sl = s->slots[slot];
    if (++sl.generation <= 0) sl.generation = 1;
    sl.progress[0].store(0, std::memory_order_release); sl.progress[1].store(0, std::memory_order_release);
    std::fill(sl.staged.begin(), sl.staged.end(), (uint8_t)0);
    sl.nextRow = 0; sl.done[0] = sl.done[1] = 0;
    s->opened++;
    return sl.generation;
}

/* CTU rows [ctu_row0, ctu_row0 + ctu_rows) of the picture that holds `slot` (generation `gen`) are final in the three buffers (whole
 * allocated planes; cb / cr may be NULL when rows_c = 0); copied before the call returns */
int x265hip_phase_stream_rows(x265hip_phase_stream* s, int slot, int gen, const void* luma_buf, const void* cb_buf, const void* cr_buf, int ctu_row0, int ctu_rows)
{
    if (!s || slot < 0 || slot >= (int)s->slots.size() || !luma_buf || (s->prm.rows_c > 0 && (!cb_buf || !cr_buf)) || ctu_row0 < 0 || ctu_rows < 1 ||
        ctu_row0 + ctu_rows > s->ctuRows) { set_error("phase_stream_rows: bad argument"); return X265HIP_EINVAL; }
    PS::Slot& sl = s->slots[slot];
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (sl.generation != gen) { set_error("phase_stream_rows: slot %d was reopened (generation %d, not %d)", slot, sl.generation, gen); return X265HIP_EBUSY; }
    }
    const void* bufs[3] = { luma_buf, cb_buf, cr_buf };
    for (int pl = 0; pl < (s->prm.rows_c > 0 ? 3 : 1); pl++)
    {
        const int k = pl ? 1 : 0;
        int y0, y1;
        ps_lines(s, k, ctu_row0, ctu_rows, y0, y1);
        memcpy(sl.stage[pl] + (size_t)y0 * s->pitch[k], (const uint8_t*)bufs[pl] + (size_t)y0 * s->pitch[k], (size_t)(y1 - y0) * s->pitch[k]);
    }
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (sl.generation != gen) return X265HIP_EBUSY;
        for (int r = ctu_row0; r < ctu_row0 + ctu_rows; r++) sl.staged[r] = 1;
        s->queue.push_back({ slot, gen });
    }
    s->cv.notify_one();
    return 0;
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
    if (s->failed) set_error("phase_stream worker: %s", s->workerError);
    return 0;
}

} // extern "C"
