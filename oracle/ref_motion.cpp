/* oracle/ref_motion.cpp - TEST INFRASTRUCTURE, never part of the product path.
 *
 * C-ABI window onto the REAL reference motion search: builds an x265::MotionEstimate object (encoder/motion.cpp)
 * over caller-owned padded luma planes and runs MotionEstimate::motionEstimate() - integer search pattern + sub-pel
 * refinement - exactly as the lookahead / --pme callers do through the luma-only setSourcePU overload
 * (motion.cpp:171-196).  Compiled by oracle/Makefile into oracle/_ref/libx265ref<depth>.so; the tests use it to pin the
 * oracle's restatement of the search drivers and of the sub-pel refinement against the reference itself.
 */
#include "common.h"
#include "primitives.h"
#include "lowres.h"
#include "motion.h"
#include "mv.h"

#include <cstring>

using namespace X265_NS;

extern "C" void x265ref_encoder_table_reset_c(void);

struct x265ref_me_job
{
    int32_t px, py;            /* PU position in the picture */
    int32_t w, h;              /* PU size (a reference partition size) */
    int32_t qmvpx, qmvpy;      /* quarter-pel motion vector predictor */
    int32_t out_qmvx, out_qmvy, out_cost;
};

extern "C" {

/* fenc / fref: pixel (0,0) of padded planes with the same stride.  method: X265_*_SEARCH (x265.h), subme 0..7,
 * qp selects the BitCost lambda table (bitcost.cpp:40-58).  mvmin / mvmax: integer-pel search bounds applied to every job.
 * Returns the number of jobs run. */
int x265ref_motion_estimate_mvc(const void* fenc, const void* fref, intptr_t stride, int method, int subme, int merange, int qp,
                                int mvminx, int mvminy, int mvmaxx, int mvmaxy, x265ref_me_job* jobs, int njobs,
                                const int32_t* mvc /* [njobs][12][2] quarter-pel, or NULL */, const int32_t* numMvc /* [njobs] */)
{
    static bool tableReady = false;
    if (!tableReady) { x265ref_encoder_table_reset_c(); MotionEstimate::initScales(); tableReady = true; }   /* MotionEstimate reads the global table; Encoder::Encoder() calls initScales (encoder.cpp:123: the SAD_THRESH scales of UMH) */
    MotionEstimate me;
    me.init(X265_CSP_I400);
    me.setQP(qp);
    ReferencePlanes ref;
    ref.fpelPlane[0] = (pixel*)fref;
    ref.lumaStride = stride;
    ref.isLowres = false;
    ref.isWeighted = false;
    const MV mvmin(mvminx, mvminy), mvmax(mvmaxx, mvmaxy);
    for (int i = 0; i < njobs; i++)
    {
        x265ref_me_job& j = jobs[i];
        const intptr_t offset = (intptr_t)j.py * stride + j.px;
        me.setSourcePU((pixel*)fenc, stride, offset, j.w, j.h, method, method, method, subme);
        MV out(0, 0);
        const MV qmvp(j.qmvpx, j.qmvpy);
        MV cand[12];
        const int nc = (mvc && numMvc) ? numMvc[i] : 0;
        for (int k = 0; k < nc; k++) cand[k] = MV(mvc[(i * 12 + k) * 2], mvc[(i * 12 + k) * 2 + 1]);
        j.out_cost = me.motionEstimate(&ref, mvmin, mvmax, qmvp, nc, cand, merange, out, 1);
        j.out_qmvx = out.x;
        j.out_qmvy = out.y;
    }
    return njobs;
}

int x265ref_motion_estimate(const void* fenc, const void* fref, intptr_t stride, int method, int subme, int merange, int qp,
                            int mvminx, int mvminy, int mvmaxx, int mvmaxy, x265ref_me_job* jobs, int njobs)
{
    return x265ref_motion_estimate_mvc(fenc, fref, stride, method, subme, merange, qp, mvminx, mvminy, mvmaxx, mvmaxy, jobs, njobs, NULL, NULL);
}

/* X265_SEA: the twelve integral planes of the reference picture are built the way FrameFilter::processPostRow does
 * (framefilter.cpp:716-823: integral_init*h per row into row y + 1, integral_init*v once bh rows exist), with the library's own
 * primitives, over a padded plane of `width` x `height` samples with margins padX / padY (fref = sample (0,0), stride >= width +
 * 2 * padX); MotionEstimate::integral[] then points at the PU's position in each plane (search.cpp:2264). */
int x265ref_motion_estimate_sea(const void* fenc, const void* fref, intptr_t stride, int width, int height, int padX, int padY,
                                int subme, int merange, int qp, int mvminx, int mvminy, int mvmaxx, int mvmaxy,
                                x265ref_me_job* jobs, int njobs)
{
    static bool tableReady = false;
    if (!tableReady) { x265ref_encoder_table_reset_c(); MotionEstimate::initScales(); tableReady = true; }
    if (stride < width + 2 * padX) return -1;
    static const int planeW[INTEGRAL_PLANE_NUM] = { 32, 32, 32, 24, 16, 16, 16, 12, 8, 8, 4, 4 };      /* framedata.h:171 */
    static const int planeH[INTEGRAL_PLANE_NUM] = { 32, 24, 8, 32, 16, 12, 4, 16, 32, 8, 16, 4 };
    const size_t planeLen = (size_t)stride * (height + 2 * padY);
    uint32_t* buf[INTEGRAL_PLANE_NUM];
    uint32_t* org[INTEGRAL_PLANE_NUM];
    for (int i = 0; i < INTEGRAL_PLANE_NUM; i++)
    {
        buf[i] = (uint32_t*)calloc(planeLen + 64, sizeof(uint32_t));      /* encoder.cpp:2351-2355 */
        org[i] = buf[i] + stride * padY + padX;
    }
    auto hidx = [](int n) { return n == 4 ? INTEGRAL_4 : n == 8 ? INTEGRAL_8 : n == 12 ? INTEGRAL_12 : n == 16 ? INTEGRAL_16 : n == 24 ? INTEGRAL_24 : INTEGRAL_32; };
    for (int y = -padY; y < height + padY - 1; y++)
    {
        pixel* pix = (pixel*)fref + (intptr_t)y * stride - padX;
        for (int i = 0; i < INTEGRAL_PLANE_NUM; i++)
        {
            uint32_t* sum = org[i] + (intptr_t)(y + 1) * stride - padX;
            primitives.integral_inith[hidx(planeW[i])](sum, pix, stride);
            if (y >= planeH[i] - padY)
                primitives.integral_initv[hidx(planeH[i])](sum - (intptr_t)planeH[i] * stride, stride);
        }
    }
    MotionEstimate me;
    me.init(X265_CSP_I400);
    me.setQP(qp);
    ReferencePlanes ref;
    ref.fpelPlane[0] = (pixel*)fref;
    ref.lumaStride = stride;
    ref.isLowres = false;
    ref.isWeighted = false;
    const MV mvmin(mvminx, mvminy), mvmax(mvmaxx, mvmaxy);
    for (int i = 0; i < njobs; i++)
    {
        x265ref_me_job& j = jobs[i];
        const intptr_t offset = (intptr_t)j.py * stride + j.px;
        me.setSourcePU((pixel*)fenc, stride, offset, j.w, j.h, X265_SEA, X265_SEA, X265_SEA, subme);
        for (int k = 0; k < INTEGRAL_PLANE_NUM; k++) me.integral[k] = org[k] + offset;
        MV out(0, 0);
        j.out_cost = me.motionEstimate(&ref, mvmin, mvmax, MV(j.qmvpx, j.qmvpy), 0, NULL, merange, out, 1);
        j.out_qmvx = out.x;
        j.out_qmvy = out.y;
    }
    for (int i = 0; i < INTEGRAL_PLANE_NUM; i++) free(buf[i]);
    return njobs;
}

/* the u16 cost of a quarter-pel mv difference for `qp` (index d + 2 * BC_MAX_MV), for tests that want the exact table */
uint16_t x265ref_mvcost_entry(int qp, int d)
{
    BitCost bc;
    bc.setQP(qp);
    bc.setMVP(MV(0, 0));
    return bc.mvcost(MV(d, 0)) - bc.mvcost(MV(0, 0)) + bc.mvcost(MV(0, 0)) / 2;   /* mvcost(d,0) = cost[d] + cost[0] */
}

} // extern "C"
