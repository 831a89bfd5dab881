/* oracle/ref_lookahead.cpp - TEST INFRASTRUCTURE, never part of the product path.
 *
 * C-ABI window onto the REAL reference lookahead preparation: builds x265::PicYuv / x265::Lowres / x265::LookaheadTLD
 * objects over a caller-supplied padded luma plane and runs Lowres::init() (common/lowres.cpp:245-306) and
 * LookaheadTLD::lowresIntraEstimate() (encoder/slicetype.cpp:696-772), then copies the half-resolution planes and the
 * per-block results out.  The tests use it to pin oracle/x265_oracle_pipeline3.c against the reference classes.
 */
#include "common.h"
#include "primitives.h"
#include "picyuv.h"
#include "lowres.h"
#include "frame.h"
#include "slicetype.h"
#include "x265.h"

#include <cstring>

using namespace X265_NS;

extern "C" void x265ref_encoder_table_reset_c(void);

extern "C" {

/* plane: the ALLOCATION START of a padded luma plane with the reference's PicYuv geometry for `width` x `height`
 * (maxCUSize 64: margins 96 / 80, stride = ceil(width / 64) * 64 + 192); it is copied wholesale.
 * Outputs: geometry[] = { lowres width, lines, lumaStride, blocks per row, blocks per column };
 * planes[i]: room for lumaStride * (lines + 2 * 80) pixels each, receives the whole padded plane i;
 * intraCost / intraMode / lowresCosts: one entry per 8x8 lowres block.  Returns 0 on success. */
int x265ref_lowres_intra(const void* plane, int width, int height, int* geometry,
                         void* plane0, void* plane1, void* plane2, void* plane3,
                         int32_t* intraCost, uint8_t* intraMode, uint16_t* lowresCosts)
{
    static bool tableReady = false;
    if (!tableReady) { x265ref_encoder_table_reset_c(); tableReady = true; }
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width;
    param->sourceHeight = height;
    param->internalCsp = X265_CSP_I400;
    param->maxCUSize = 64;
    param->rc.aqMode = 0;
    param->rc.hevcAq = 0;
    param->bAQMotion = 0;
    param->bEnableHME = 0;
    PicYuv pic;
    pic.m_param = param;
    if (!pic.create(param, true)) return -1;
    const int h64 = (height + 63) / 64 * 64;
    memcpy(pic.m_picOrg[0] - pic.m_lumaMarginY * pic.m_stride - pic.m_lumaMarginX, plane,
           sizeof(pixel) * pic.m_stride * (h64 + 2 * pic.m_lumaMarginY));
    Lowres lr;
    memset((void*)&lr, 0, sizeof(lr));
    const uint32_t qgSize = 32;
    if (!lr.create(param, &pic, qgSize)) return -2;
    lr.init(&pic, 0);
    LookaheadTLD tld;
    tld.init(lr.maxBlocksInRow, lr.maxBlocksInCol, lr.maxBlocksInRow * lr.maxBlocksInCol);
    tld.lowresIntraEstimate(lr, qgSize);

    geometry[0] = lr.width; geometry[1] = lr.lines; geometry[2] = (int)lr.lumaStride;
    geometry[3] = lr.maxBlocksInRow; geometry[4] = lr.maxBlocksInCol;
    const size_t planesize = (size_t)lr.lumaStride * (lr.lines + 2 * pic.m_lumaMarginY);
    void* outs[4] = { plane0, plane1, plane2, plane3 };
    for (int i = 0; i < 4; i++)
        memcpy(outs[i], lr.lowresPlane[i] - pic.m_lumaMarginY * lr.lumaStride - pic.m_lumaMarginX, planesize * sizeof(pixel));
    const int ncu = lr.maxBlocksInRow * lr.maxBlocksInCol;
    memcpy(intraCost, lr.intraCost, ncu * sizeof(int32_t));
    memcpy(intraMode, lr.intraMode, ncu * sizeof(uint8_t));
    memcpy(lowresCosts, lr.lowresCosts[0][0], ncu * sizeof(uint16_t));
    lr.destroy();
    pic.destroy();
    x265_param_free(param);
    return 0;
}

/* The REAL CostEstimateGroup::singleCost(0, 1, 1) (encoder/slicetype.cpp:3021-3388) for a P picture `cur` predicted from `ref`:
 * both planes as in x265ref_lowres_intra.  Runs Lowres::init on both, lowresIntraEstimate on the current picture, then the
 * frame cost estimate with a pool-less Lookahead (no cooperative slices, no HME, no weighted prediction, no AQ).
 * Outputs per 8x8 block: mvs int32 [n][2], mvCosts int32 [n], lowresCosts uint16 [n]; rowSatds int32 [rows];
 * frame int64 [4] = { returned score, costEst, costEstAq, intraMbs }.  Returns 0 on success. */
static int lowres_cost_core(const void* curPlane, const void* refPlane, int width, int height,
                            int32_t* mvs, int32_t* mvCosts, uint16_t* lowresCosts, int32_t* rowSatds, int64_t* frame,
                            const uint64_t* wpSsd, const uint64_t* wpSum, int32_t* isWeighted);

int x265ref_lowres_cost(const void* curPlane, const void* refPlane, int width, int height,
                        int32_t* mvs, int32_t* mvCosts, uint16_t* lowresCosts, int32_t* rowSatds, int64_t* frame)
{
    return lowres_cost_core(curPlane, refPlane, width, height, mvs, mvCosts, lowresCosts, rowSatds, frame, NULL, NULL, NULL);
}

/* The same with --weightp: estimateFrameCost runs weightsAnalyse first (slicetype.cpp:3136-3138) and, when it accepts a weight,
 * searches the WEIGHTED reference planes (:3222,3267).  wpSsd / wpSum: wp_ssd[0] / wp_sum[0] of the current picture ([0]) and the
 * reference ([1]); *isWeighted receives weightedRef.isWeighted. */
int x265ref_lowres_cost_weightp(const void* curPlane, const void* refPlane, int width, int height, const uint64_t* wpSsd, const uint64_t* wpSum,
                                int32_t* mvs, int32_t* mvCosts, uint16_t* lowresCosts, int32_t* rowSatds, int64_t* frame, int32_t* isWeighted)
{
    return lowres_cost_core(curPlane, refPlane, width, height, mvs, mvCosts, lowresCosts, rowSatds, frame, wpSsd, wpSum, isWeighted);
}

static int lowres_cost_core(const void* curPlane, const void* refPlane, int width, int height,
                            int32_t* mvs, int32_t* mvCosts, uint16_t* lowresCosts, int32_t* rowSatds, int64_t* frame,
                            const uint64_t* wpSsd, const uint64_t* wpSum, int32_t* isWeighted)
{
    static bool tableReady = false;
    if (!tableReady) { x265ref_encoder_table_reset_c(); tableReady = true; }
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width;
    param->sourceHeight = height;
    param->internalCsp = X265_CSP_I400;
    param->maxCUSize = 64;
    param->rc.aqMode = 0;
    param->rc.hevcAq = 0;
    param->bAQMotion = 0;
    param->bEnableHME = 0;
    param->bEnableWeightedPred = wpSsd ? 1 : 0;
    param->bEnableWeightedBiPred = 0;
    param->lookaheadSlices = 0;
    const int h64 = (height + 63) / 64 * 64;
    PicYuv pics[2];
    Lowres lrs[2];
    const void* srcs[2] = { refPlane, curPlane };
    const uint32_t qgSize = 32;
    for (int i = 0; i < 2; i++)
    {
        pics[i].m_param = param;
        if (!pics[i].create(param, true)) return -1;
        memcpy(pics[i].m_picOrg[0] - pics[i].m_lumaMarginY * pics[i].m_stride - pics[i].m_lumaMarginX, srcs[i],
               sizeof(pixel) * pics[i].m_stride * (h64 + 2 * pics[i].m_lumaMarginY));
        memset((void*)&lrs[i], 0, sizeof(Lowres));
        if (!lrs[i].create(param, &pics[i], qgSize)) return -2;
        lrs[i].init(&pics[i], i);
        if (wpSsd) { lrs[i].wp_ssd[0] = wpSsd[1 - i]; lrs[i].wp_sum[0] = wpSum[1 - i]; }
    }
    Lookahead la(param, NULL);
    if (!la.create()) return -3;
    la.m_tld[0].lowresIntraEstimate(lrs[1], qgSize);
    Lowres* frames[2] = { &lrs[0], &lrs[1] };
    CostEstimateGroup estGroup(la, frames);
    frame[0] = estGroup.singleCost(0, 1, 1);
    Lowres& fenc = lrs[1];
    const int ncu = fenc.maxBlocksInRow * fenc.maxBlocksInCol;
    for (int i = 0; i < ncu; i++)
    {
        mvs[2 * i] = fenc.lowresMvs[0][1][i].x;
        mvs[2 * i + 1] = fenc.lowresMvs[0][1][i].y;
        mvCosts[i] = fenc.lowresMvCosts[0][1][i];
        lowresCosts[i] = fenc.lowresCosts[1][0][i];
    }
    for (int y = 0; y < fenc.maxBlocksInCol; y++) rowSatds[y] = fenc.rowSatds[1][0][y];
    frame[1] = fenc.costEst[1][0];
    frame[2] = fenc.costEstAq[1][0];
    frame[3] = fenc.intraMbs[1];
    if (isWeighted) *isWeighted = fenc.weightedRef[1].isWeighted ? 1 : 0;
    la.destroy();
    for (int i = 0; i < 2; i++) { lrs[i].destroy(); pics[i].destroy(); }
    x265_param_free(param);
    return 0;
}

/* The REAL CostEstimateGroup::singleCost(0, 2, 1) for a B picture `cur` between `ref0` (list 0) and `ref1` (list 1): same set-up
 * as x265ref_lowres_cost.  Outputs per 8x8 block: mvs0 / mvs1 int32 [n][2], mvCosts0 / mvCosts1 int32 [n], lowresCosts uint16 [n];
 * rowSatds int32 [rows]; frame int64 [4] = { returned score, costEst (as stored: already scaled), costEstAq, intraMbs }. */
static int lowres_cost_b_core(const void* curPlane, const void* ref0Plane, const void* ref1Plane, int width, int height,
                              int32_t* mvs0, int32_t* mvCosts0, int32_t* mvs1, int32_t* mvCosts1, uint16_t* lowresCosts, int32_t* rowSatds,
                              int64_t* frame, const uint64_t* wpSsd, const uint64_t* wpSum, int32_t* isWeighted);

int x265ref_lowres_cost_b(const void* curPlane, const void* ref0Plane, const void* ref1Plane, int width, int height,
                          int32_t* mvs0, int32_t* mvCosts0, int32_t* mvs1, int32_t* mvCosts1, uint16_t* lowresCosts, int32_t* rowSatds,
                          int64_t* frame)
{
    return lowres_cost_b_core(curPlane, ref0Plane, ref1Plane, width, height, mvs0, mvCosts0, mvs1, mvCosts1, lowresCosts, rowSatds, frame, NULL, NULL, NULL);
}

/* B picture with --weightp: the list-0 search (predictor candidates, skip cost, motionEstimate) runs on the weighted list-0 planes,
 * the two bi-directional candidates keep the unweighted ones (slicetype.cpp:3222,3267,3328).  wpSsd / wpSum: current picture ([0]) and
 * the list-0 reference ([1]). */
int x265ref_lowres_cost_b_weightp(const void* curPlane, const void* ref0Plane, const void* ref1Plane, int width, int height,
                                  const uint64_t* wpSsd, const uint64_t* wpSum,
                                  int32_t* mvs0, int32_t* mvCosts0, int32_t* mvs1, int32_t* mvCosts1, uint16_t* lowresCosts, int32_t* rowSatds,
                                  int64_t* frame, int32_t* isWeighted)
{
    return lowres_cost_b_core(curPlane, ref0Plane, ref1Plane, width, height, mvs0, mvCosts0, mvs1, mvCosts1, lowresCosts, rowSatds, frame, wpSsd, wpSum, isWeighted);
}

static int lowres_cost_b_core(const void* curPlane, const void* ref0Plane, const void* ref1Plane, int width, int height,
                              int32_t* mvs0, int32_t* mvCosts0, int32_t* mvs1, int32_t* mvCosts1, uint16_t* lowresCosts, int32_t* rowSatds,
                              int64_t* frame, const uint64_t* wpSsd, const uint64_t* wpSum, int32_t* isWeighted)
{
    static bool tableReady = false;
    if (!tableReady) { x265ref_encoder_table_reset_c(); tableReady = true; }
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width;
    param->sourceHeight = height;
    param->internalCsp = X265_CSP_I400;
    param->maxCUSize = 64;
    param->rc.aqMode = 0;
    param->rc.hevcAq = 0;
    param->bAQMotion = 0;
    param->bEnableHME = 0;
    param->bEnableWeightedPred = wpSsd ? 1 : 0;
    param->bEnableWeightedBiPred = 0;
    param->lookaheadSlices = 0;
    const int h64 = (height + 63) / 64 * 64;
    PicYuv pics[3];
    Lowres lrs[3];
    const void* srcs[3] = { ref0Plane, curPlane, ref1Plane };
    const uint32_t qgSize = 32;
    for (int i = 0; i < 3; i++)
    {
        pics[i].m_param = param;
        if (!pics[i].create(param, true)) return -1;
        memcpy(pics[i].m_picOrg[0] - pics[i].m_lumaMarginY * pics[i].m_stride - pics[i].m_lumaMarginX, srcs[i],
               sizeof(pixel) * pics[i].m_stride * (h64 + 2 * pics[i].m_lumaMarginY));
        memset((void*)&lrs[i], 0, sizeof(Lowres));
        if (!lrs[i].create(param, &pics[i], qgSize)) return -2;
        lrs[i].init(&pics[i], i);
    }
    if (wpSsd) { lrs[1].wp_ssd[0] = wpSsd[0]; lrs[1].wp_sum[0] = wpSum[0]; lrs[0].wp_ssd[0] = wpSsd[1]; lrs[0].wp_sum[0] = wpSum[1]; }
    Lookahead la(param, NULL);
    if (!la.create()) return -3;
    la.m_tld[0].lowresIntraEstimate(lrs[1], qgSize);
    Lowres* frames[3] = { &lrs[0], &lrs[1], &lrs[2] };
    CostEstimateGroup estGroup(la, frames);
    frame[0] = estGroup.singleCost(0, 2, 1);
    Lowres& fenc = lrs[1];
    const int ncu = fenc.maxBlocksInRow * fenc.maxBlocksInCol;
    for (int i = 0; i < ncu; i++)
    {
        mvs0[2 * i] = fenc.lowresMvs[0][1][i].x; mvs0[2 * i + 1] = fenc.lowresMvs[0][1][i].y;
        mvs1[2 * i] = fenc.lowresMvs[1][1][i].x; mvs1[2 * i + 1] = fenc.lowresMvs[1][1][i].y;
        mvCosts0[i] = fenc.lowresMvCosts[0][1][i];
        mvCosts1[i] = fenc.lowresMvCosts[1][1][i];
        lowresCosts[i] = fenc.lowresCosts[1][1][i];
    }
    for (int y = 0; y < fenc.maxBlocksInCol; y++) rowSatds[y] = fenc.rowSatds[1][1][y];
    frame[1] = fenc.costEst[1][1];
    frame[2] = fenc.costEstAq[1][1];
    frame[3] = fenc.intraMbs[1];
    if (isWeighted) *isWeighted = fenc.weightedRef[1].isWeighted ? 1 : 0;
    la.destroy();
    for (int i = 0; i < 3; i++) { lrs[i].destroy(); pics[i].destroy(); }
    x265_param_free(param);
    return 0;
}


/* The REAL LookaheadTLD::weightsAnalyse(fenc, ref) (encoder/slicetype.cpp:860-957) for a current picture / reference pair (planes as
 * in x265ref_lowres_intra; the reference is frame 0, the current picture frame 1).  wpSsd / wpSum: wp_ssd[0] / wp_sum[0] of the
 * current picture ([0]) and of the reference ([1]) - normally left in Lowres by the adaptive-quantisation pass.
 * Outputs: out[0] = weightedRef.isWeighted; weighted: the four weighted planes' whole padded buffers (only meaningful when
 * out[0]); intraCost (int32 per 8x8 block) as computed by lowresIntraEstimate on the current picture.  Returns 0 on success. */
int x265ref_weights_analyse(const void* curPlane, const void* refPlane, int width, int height, const uint64_t* wpSsd, const uint64_t* wpSum,
                            int32_t* out, void* w0, void* w1, void* w2, void* w3, int32_t* intraCost)
{
    static bool tableReady = false;
    if (!tableReady) { x265ref_encoder_table_reset_c(); tableReady = true; }
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width;
    param->sourceHeight = height;
    param->internalCsp = X265_CSP_I400;
    param->maxCUSize = 64;
    param->rc.aqMode = 0;
    param->rc.hevcAq = 0;
    param->bAQMotion = 0;
    param->bEnableHME = 0;
    const int h64 = (height + 63) / 64 * 64;
    PicYuv pics[2];
    Lowres lrs[2];
    const void* srcs[2] = { refPlane, curPlane };
    const uint32_t qgSize = 32;
    for (int i = 0; i < 2; i++)
    {
        pics[i].m_param = param;
        if (!pics[i].create(param, true)) return -1;
        memcpy(pics[i].m_picOrg[0] - pics[i].m_lumaMarginY * pics[i].m_stride - pics[i].m_lumaMarginX, srcs[i],
               sizeof(pixel) * pics[i].m_stride * (h64 + 2 * pics[i].m_lumaMarginY));
        memset((void*)&lrs[i], 0, sizeof(Lowres));
        if (!lrs[i].create(param, &pics[i], qgSize)) return -2;
        lrs[i].init(&pics[i], i);
        lrs[i].wp_ssd[0] = wpSsd[1 - i];
        lrs[i].wp_sum[0] = wpSum[1 - i];
    }
    Lowres& fenc = lrs[1];
    Lowres& ref = lrs[0];
    {
        LookaheadTLD tld;
        tld.init(fenc.maxBlocksInRow, fenc.maxBlocksInCol, fenc.maxBlocksInRow * fenc.maxBlocksInCol);
        tld.lowresIntraEstimate(fenc, qgSize);
        memcpy(intraCost, fenc.intraCost, sizeof(int32_t) * fenc.maxBlocksInRow * fenc.maxBlocksInCol);
        tld.weightsAnalyse(fenc, ref);
        const ReferencePlanes& wr = fenc.weightedRef[1];
        out[0] = wr.isWeighted ? 1 : 0;
        if (tld.wbuffer[0])
        {
            const size_t planesize = fenc.buffer[1] - fenc.buffer[0];
            void* outs[4] = { w0, w1, w2, w3 };
            for (int i = 0; i < 4; i++) memcpy(outs[i], tld.wbuffer[i], planesize * sizeof(pixel));
        }
    }
    for (int i = 0; i < 2; i++) { lrs[i].destroy(); pics[i].destroy(); }
    x265_param_free(param);
    return 0;
}


/* The REAL LookaheadTLD::calcAdaptiveQuantFrame (encoder/slicetype.cpp:439-694) on one picture.  yPlane: allocation start of the
 * padded luma plane (as in x265ref_lowres_intra); cb / cr: compact (width / 2) x (height / 2) chroma planes or NULL for 4:0:0 (then
 * width and height should be multiples of the block size: no chroma padding is synthesised).  aqMode 0..3, hevcAq off.
 * Outputs per qgSize x qgSize block (row-major): qpAqOffset (double), invQscale (int32); wpSum / wpSsd [3] as left in Lowres. */
int x265ref_aq_frame(const void* yPlane, const void* cb, const void* cr, int width, int height, int qgSize, int aqMode, double aqStrength,
                     int weightp, double* qpAqOffset, int32_t* invQscale, uint64_t* wpSum, uint64_t* wpSsd, int32_t* numBlocks)
{
    static bool tableReady = false;
    if (!tableReady) { x265ref_encoder_table_reset_c(); tableReady = true; }
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width;
    param->sourceHeight = height;
    param->internalCsp = cb ? X265_CSP_I420 : X265_CSP_I400;
    param->maxCUSize = 64;
    param->rc.aqMode = aqMode;
    param->rc.aqStrength = aqStrength;
    param->rc.hevcAq = 0;
    param->rc.qgSize = qgSize;
    param->rc.cuTree = 0;
    param->rc.bStatRead = 0;
    param->bAQMotion = 0;
    param->bEnableHME = 0;
    param->bHDR10Opt = 0;
    param->bDynamicRefine = 0;
    param->bEnableFades = 0;
    param->bEnableWeightedPred = weightp;
    param->bEnableWeightedBiPred = 0;
    PicYuv pic;
    pic.m_param = param;
    if (!pic.create(param, true)) return -1;
    const int h64 = (height + 63) / 64 * 64;
    memcpy(pic.m_picOrg[0] - pic.m_lumaMarginY * pic.m_stride - pic.m_lumaMarginX, yPlane,
           sizeof(pixel) * pic.m_stride * (h64 + 2 * pic.m_lumaMarginY));
    if (cb)
    {
        const void* cs[2] = { cb, cr };
        for (int c = 0; c < 2; c++)
            for (int y = 0; y < height / 2; y++)
                memcpy(pic.m_picOrg[1 + c] + (intptr_t)y * pic.m_strideC, (const pixel*)cs[c] + (size_t)y * (width / 2), sizeof(pixel) * (width / 2));
    }
    int rc = 0;
    {
        Frame frame;
        frame.m_param = param;
        frame.m_fencPic = &pic;
        frame.m_quantOffsets = NULL;
        memset((void*)&frame.m_lowres, 0, sizeof(Lowres));
        if (!frame.m_lowres.create(param, &pic, qgSize)) rc = -2;
        else
        {
            LookaheadTLD tld;
            tld.init(frame.m_lowres.maxBlocksInRow, frame.m_lowres.maxBlocksInCol, frame.m_lowres.maxBlocksInRow * frame.m_lowres.maxBlocksInCol);
            tld.calcAdaptiveQuantFrame(&frame, param);
            const int n = qgSize == 8 ? frame.m_lowres.maxBlocksInRowFullRes * frame.m_lowres.maxBlocksInColFullRes
                                      : frame.m_lowres.maxBlocksInRow * frame.m_lowres.maxBlocksInCol;
            *numBlocks = n;
            if (frame.m_lowres.qpAqOffset)
            {
                memcpy(qpAqOffset, frame.m_lowres.qpAqOffset, sizeof(double) * n);
                for (int i = 0; i < n; i++) invQscale[i] = frame.m_lowres.invQscaleFactor[i];
            }
            for (int i = 0; i < 3; i++) { wpSum[i] = frame.m_lowres.wp_sum[i]; wpSsd[i] = frame.m_lowres.wp_ssd[i]; }
            frame.m_lowres.destroy();
        }
        frame.m_fencPic = NULL;
    }
    pic.destroy();
    x265_param_free(param);
    return rc;
}


/* The REAL calcAdaptiveQuantFrame with rc.hevcAq (LookaheadTLD::xPreanalyze / xPreanalyzeQp, slicetype.cpp:293-441) on one picture:
 * planes as in x265ref_aq_frame.  Outputs: layerParts[4] (numAQPartInWidth * numAQPartInHeight of every enabled layer, 0 = off),
 * the enabled layers' dActivity / dQpOffset one after the other, dAvgActivity[4], invQscaleFactor over the deepest layer's partitions,
 * wp_sum / wp_ssd [3]. */
int x265ref_aq_hevc_frame(const void* yPlane, const void* cb, const void* cr, int width, int height, int qgSize, double qpAdaptationRange,
                          int weightp, int32_t* layerParts, double* activity, double* qpOffset, double* avgActivity, int32_t* invQscale,
                          uint64_t* wpSum, uint64_t* wpSsd)
{
    static bool tableReady = false;
    if (!tableReady) { x265ref_encoder_table_reset_c(); tableReady = true; }
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width;
    param->sourceHeight = height;
    param->internalCsp = cb ? X265_CSP_I420 : X265_CSP_I400;
    param->maxCUSize = 64;
    param->rc.aqMode = 2;
    param->rc.aqStrength = 1.0;
    param->rc.hevcAq = 1;
    param->rc.qpAdaptationRange = qpAdaptationRange;
    param->rc.qgSize = qgSize;
    param->rc.cuTree = 0;
    param->rc.bStatRead = 0;
    param->bAQMotion = 0;
    param->bEnableHME = 0;
    param->bHDR10Opt = 0;
    param->bDynamicRefine = 0;
    param->bEnableFades = 0;
    param->bEnableWeightedPred = weightp;
    param->bEnableWeightedBiPred = 0;
    PicYuv pic;
    pic.m_param = param;
    if (!pic.create(param, true)) return -1;
    const int h64 = (height + 63) / 64 * 64;
    memcpy(pic.m_picOrg[0] - pic.m_lumaMarginY * pic.m_stride - pic.m_lumaMarginX, yPlane,
           sizeof(pixel) * pic.m_stride * (h64 + 2 * pic.m_lumaMarginY));
    if (cb)
    {
        const void* cs[2] = { cb, cr };
        for (int c = 0; c < 2; c++)
            for (int y = 0; y < height / 2; y++)
                memcpy(pic.m_picOrg[1 + c] + (intptr_t)y * pic.m_strideC, (const pixel*)cs[c] + (size_t)y * (width / 2), sizeof(pixel) * (width / 2));
    }
    int rc = 0;
    {
        Frame frame;
        frame.m_param = param;
        frame.m_fencPic = &pic;
        frame.m_quantOffsets = NULL;
        memset((void*)&frame.m_lowres, 0, sizeof(Lowres));
        if (!frame.m_lowres.create(param, &pic, qgSize)) rc = -2;
        else
        {
            LookaheadTLD tld;
            tld.init(frame.m_lowres.maxBlocksInRow, frame.m_lowres.maxBlocksInCol, frame.m_lowres.maxBlocksInRow * frame.m_lowres.maxBlocksInCol);
            tld.calcAdaptiveQuantFrame(&frame, param);
            size_t at = 0;
            int deepest = -1;
            for (int d = 0; d < 4; d++)
            {
                layerParts[d] = 0; avgActivity[d] = 0;
                if (!aqLayerDepth[0][6 - (qgSize == 64 ? 6 : qgSize == 32 ? 5 : qgSize == 16 ? 4 : 3)][d]) continue;
                PicQPAdaptationLayer& L = frame.m_lowres.pAQLayer[d];
                const int n = L.numAQPartInWidth * L.numAQPartInHeight;
                memcpy(activity + at, L.dActivity, sizeof(double) * n);
                memcpy(qpOffset + at, L.dQpOffset, sizeof(double) * n);
                avgActivity[d] = L.dAvgActivity;
                layerParts[d] = n; at += n; deepest = d;
            }
            if (deepest >= 0)
                for (int i = 0; i < layerParts[deepest]; i++) invQscale[i] = frame.m_lowres.invQscaleFactor[i];
            if ((int)frame.m_lowres.pAQLayer->minAQDepth != deepest) rc = -3;
            for (int i = 0; i < 3; i++) { wpSum[i] = frame.m_lowres.wp_sum[i]; wpSsd[i] = frame.m_lowres.wp_ssd[i]; }
            frame.m_lowres.destroy();
        }
        frame.m_fencPic = NULL;
    }
    pic.destroy();
    x265_param_free(param);
    return rc;
}


namespace { struct LookaheadProbe : public Lookahead
{
    LookaheadProbe(x265_param* p, ThreadPool* t) : Lookahead(p, t) {}
    using Lookahead::estimateCUPropagate;            /* protected in the class (slicetype.h:196) */
    using Lookahead::cuTreeFinish;
    using Lookahead::frameCostRecalculate;
}; }

/* One cuTree propagation step with the REAL Lookahead::estimateCUPropagate (encoder/slicetype.cpp:2641-2753).  Three pictures:
 * ref0 (frame 0), cur (frame 1), ref1 (frame 2; NULL = a P picture, p1 = b = 1).  The real singleCost fills lowresMvs / lowresCosts of
 * the current picture first; then propagateCost of the three pictures is set from the caller (propCur NULL = not referenced),
 * invQscaleFactor of the current picture from invQscale, and estimateCUPropagate(frames, averageDuration, 0, p1, 1, referenced) runs.
 * Outputs: the inputs the step read (intraCost, lowresCosts, mvs0, mvs1) and the references' propagateCost afterwards. */
int x265ref_cutree_propagate(const void* curPlane, const void* ref0Plane, const void* ref1Plane, int width, int height,
                             const int32_t* invQscale, const uint16_t* propCur, uint16_t* propRef0, uint16_t* propRef1,
                             int fpsNum, int fpsDenom, double averageDuration, int weightedBiPred,
                             int32_t* intraCost, uint16_t* lowresCosts, int32_t* mvs0, int32_t* mvs1)
{
    static bool tableReady = false;
    if (!tableReady) { x265ref_encoder_table_reset_c(); tableReady = true; }
    const int nf = ref1Plane ? 3 : 2, p1 = ref1Plane ? 2 : 1;
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width;
    param->sourceHeight = height;
    param->internalCsp = X265_CSP_I400;
    param->maxCUSize = 64;
    param->rc.aqMode = 2;                       /* Lowres::create allocates invQscaleFactor only with AQ on */
    param->rc.hevcAq = 0;
    param->rc.qgSize = 32;
    param->rc.vbvBufferSize = 0;
    param->bAQMotion = 0;
    param->bEnableHME = 0;
    param->bEnableWeightedPred = 0;
    param->bEnableWeightedBiPred = weightedBiPred;
    param->lookaheadSlices = 0;
    param->fpsNum = fpsNum;
    param->fpsDenom = fpsDenom;
    const int h64 = (height + 63) / 64 * 64;
    PicYuv pics[3];
    Lowres lrs[3];
    const void* srcs[3] = { ref0Plane, curPlane, ref1Plane };
    const uint32_t qgSize = 32;
    for (int i = 0; i < nf; i++)
    {
        pics[i].m_param = param;
        if (!pics[i].create(param, true)) return -1;
        memcpy(pics[i].m_picOrg[0] - pics[i].m_lumaMarginY * pics[i].m_stride - pics[i].m_lumaMarginX, srcs[i],
               sizeof(pixel) * pics[i].m_stride * (h64 + 2 * pics[i].m_lumaMarginY));
        memset((void*)&lrs[i], 0, sizeof(Lowres));
        if (!lrs[i].create(param, &pics[i], qgSize)) return -2;
        lrs[i].init(&pics[i], i);
    }
    Lowres& fenc = lrs[1];
    const int ncu = fenc.maxBlocksInRow * fenc.maxBlocksInCol;
    for (int i = 0; i < nf; i++)
        for (int k = 0; k < ncu; k++) lrs[i].invQscaleFactor[k] = i == 1 ? invQscale[k] : 256;
    int rc = 0;
    {
        LookaheadProbe la(param, NULL);
        if (!la.create()) rc = -3;
        else
        {
            la.m_tld[0].lowresIntraEstimate(fenc, qgSize);
            Lowres* frames[3] = { &lrs[0], &lrs[1], nf == 3 ? &lrs[2] : NULL };
            CostEstimateGroup estGroup(la, frames);
            estGroup.singleCost(0, p1, 1);
            for (int k = 0; k < ncu; k++)
            {
                intraCost[k] = fenc.intraCost[k];
                lowresCosts[k] = fenc.lowresCosts[1][p1 - 1][k];
                mvs0[2 * k] = fenc.lowresMvs[0][1][k].x; mvs0[2 * k + 1] = fenc.lowresMvs[0][1][k].y;
                if (nf == 3) { mvs1[2 * k] = fenc.lowresMvs[1][1][k].x; mvs1[2 * k + 1] = fenc.lowresMvs[1][1][k].y; }
            }
            memcpy(lrs[0].propagateCost, propRef0, sizeof(uint16_t) * ncu);
            if (nf == 3) memcpy(lrs[2].propagateCost, propRef1, sizeof(uint16_t) * ncu);
            if (propCur) memcpy(fenc.propagateCost, propCur, sizeof(uint16_t) * ncu);
            la.estimateCUPropagate(frames, averageDuration, 0, p1, 1, propCur ? 1 : 0);
            memcpy(propRef0, lrs[0].propagateCost, sizeof(uint16_t) * ncu);
            if (nf == 3) memcpy(propRef1, lrs[2].propagateCost, sizeof(uint16_t) * ncu);
            la.destroy();
        }
    }
    for (int i = 0; i < nf; i++) { lrs[i].destroy(); pics[i].destroy(); }
    x265_param_free(param);
    return rc;
}


/* The REAL Lookahead::cuTreeFinish (encoder/slicetype.cpp:2889-2937) on caller-supplied per-block arrays of a width x height picture
 * (qgSize 32, hevcAq off): qpCuTreeOffset is preset to `preset` and updated in place. */
static int cutree_finish_core(int qgSize, int width, int height, const int32_t* intraCost, const int32_t* invQscale, const uint16_t* propagateCost,
                              const double* qpAqOffset, int fpsNum, int fpsDenom, double averageDuration, double qCompress,
                              int ref0Distance, double weightedCostDelta, double* qpCuTreeOffset);
int x265ref_cutree_finish(int width, int height, const int32_t* intraCost, const int32_t* invQscale, const uint16_t* propagateCost,
                          const double* qpAqOffset, int fpsNum, int fpsDenom, double averageDuration, double qCompress,
                          int ref0Distance, double weightedCostDelta, double* qpCuTreeOffset)
{
    return cutree_finish_core(32, width, height, intraCost, invQscale, propagateCost, qpAqOffset, fpsNum, fpsDenom, averageDuration, qCompress, ref0Distance,
                              weightedCostDelta, qpCuTreeOffset);
}
/* --qg-size 8: invQscale = invQscaleFactor8x8 per lowres block, qpAqOffset / qpCuTreeOffset on the full-resolution grid (4 per block) */
int x265ref_cutree_finish_qg8(int width, int height, const int32_t* intraCost, const int32_t* invQscale8x8, const uint16_t* propagateCost,
                              const double* qpAqOffset, int fpsNum, int fpsDenom, double averageDuration, double qCompress,
                              int ref0Distance, double weightedCostDelta, double* qpCuTreeOffset)
{
    return cutree_finish_core(8, width, height, intraCost, invQscale8x8, propagateCost, qpAqOffset, fpsNum, fpsDenom, averageDuration, qCompress, ref0Distance,
                              weightedCostDelta, qpCuTreeOffset);
}
static int cutree_finish_core(int qgSize, int width, int height, const int32_t* intraCost, const int32_t* invQscale, const uint16_t* propagateCost,
                              const double* qpAqOffset, int fpsNum, int fpsDenom, double averageDuration, double qCompress,
                              int ref0Distance, double weightedCostDelta, double* qpCuTreeOffset)
{
    static bool tableReady = false;
    if (!tableReady) { x265ref_encoder_table_reset_c(); tableReady = true; }
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width;
    param->sourceHeight = height;
    param->internalCsp = X265_CSP_I400;
    param->maxCUSize = 64;
    param->rc.aqMode = 2;
    param->rc.hevcAq = 0;
    param->rc.qgSize = qgSize;
    param->rc.qCompress = qCompress;
    param->bEnableHME = 0;
    param->fpsNum = fpsNum;
    param->fpsDenom = fpsDenom;
    PicYuv pic;
    pic.m_param = param;
    if (!pic.create(param, true)) return -1;
    Lowres lr;
    memset((void*)&lr, 0, sizeof(Lowres));
    if (!lr.create(param, &pic, qgSize)) return -2;
    const int ncu = lr.maxBlocksInRow * lr.maxBlocksInCol;
    int rc = 0;
    {
        LookaheadProbe la(param, NULL);
        if (!la.create()) rc = -3;
        else
        {
            const int nq = qgSize == 8 ? 4 * ncu : ncu;
            for (int k = 0; k < ncu; k++)
            {
                lr.intraCost[k] = intraCost[k];
                if (qgSize == 8) lr.invQscaleFactor8x8[k] = invQscale[k]; else lr.invQscaleFactor[k] = invQscale[k];
                lr.propagateCost[k] = propagateCost[k];
            }
            for (int k = 0; k < nq; k++) { lr.qpAqOffset[k] = qpAqOffset[k]; lr.qpCuTreeOffset[k] = qpCuTreeOffset[k]; }
            if (ref0Distance) lr.weightedCostDelta[ref0Distance - 1] = weightedCostDelta;
            la.cuTreeFinish(&lr, averageDuration, ref0Distance);
            for (int k = 0; k < nq; k++) qpCuTreeOffset[k] = lr.qpCuTreeOffset[k];
            la.destroy();
        }
    }
    lr.destroy();
    pic.destroy();
    x265_param_free(param);
    return rc ? rc : ncu;
}


/* The REAL Lookahead::frameCostRecalculate(frames, 0, 1, 1) (encoder/slicetype.cpp:2941-3011) for a P picture of a width x height
 * source whose lowresCosts[1][0] and qpCuTreeOffset are supplied by the caller; rowSatds receives rowSatds[1][0].  Returns the score
 * through *score. */
/* The REAL Lookahead::cuTreeFinish with rc.hevcAq (computeCUTreeQpOffset, slicetype.cpp:2749-2887) on caller-supplied per-block arrays
 * (qgSize 16 or 32): dQpOffset of every enabled layer is preset from qpOffsetIn (layers one after the other, as x265ref_aq_hevc_frame
 * lays them out), the layers' dCuTreeOffset come back the same way.  layerParts[4] receives the partition counts. */
int x265ref_cutree_finish_hevc_aq(int width, int height, int qgSize, const int32_t* intraCost, const int32_t* invQscale, const uint16_t* propagateCost,
                                  int fpsNum, int fpsDenom, double averageDuration, double qCompress, int ref0Distance, double weightedCostDelta,
                                  const double* qpOffsetIn, double* cuTreeOffsetOut, int32_t* layerParts)
{
    static bool tableReady = false;
    if (!tableReady) { x265ref_encoder_table_reset_c(); tableReady = true; }
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width;
    param->sourceHeight = height;
    param->internalCsp = X265_CSP_I400;
    param->maxCUSize = 64;
    param->rc.aqMode = 2;
    param->rc.hevcAq = 1;
    param->rc.qgSize = qgSize;
    param->rc.qCompress = qCompress;
    param->bEnableHME = 0;
    param->fpsNum = fpsNum;
    param->fpsDenom = fpsDenom;
    PicYuv pic;
    pic.m_param = param;
    if (!pic.create(param, true)) return -1;
    Lowres lr;
    memset((void*)&lr, 0, sizeof(Lowres));
    if (!lr.create(param, &pic, qgSize)) return -2;
    const int ncu = lr.maxBlocksInRow * lr.maxBlocksInCol;
    int rc = 0;
    {
        LookaheadProbe la(param, NULL);
        if (!la.create()) rc = -3;
        else
        {
            for (int k = 0; k < ncu; k++) { lr.intraCost[k] = intraCost[k]; lr.invQscaleFactor[k] = invQscale[k]; lr.propagateCost[k] = propagateCost[k]; }
            const int aqDepth = 6 - (qgSize == 64 ? 6 : qgSize == 32 ? 5 : qgSize == 16 ? 4 : 3);
            size_t at = 0;
            for (int d = 0; d < 4; d++)
            {
                layerParts[d] = 0;
                if (!aqLayerDepth[0][aqDepth][d]) continue;
                PicQPAdaptationLayer& L = lr.pAQLayer[d];
                const int n = L.numAQPartInWidth * L.numAQPartInHeight;
                for (int k = 0; k < n; k++) L.dQpOffset[k] = qpOffsetIn[at + k];
                layerParts[d] = n; at += n;
            }
            if (ref0Distance) lr.weightedCostDelta[ref0Distance - 1] = weightedCostDelta;
            la.cuTreeFinish(&lr, averageDuration, ref0Distance);
            at = 0;
            for (int d = 0; d < 4; d++)
            {
                if (!layerParts[d]) continue;
                memcpy(cuTreeOffsetOut + at, lr.pAQLayer[d].dCuTreeOffset, sizeof(double) * layerParts[d]);
                at += layerParts[d];
            }
            la.destroy();
        }
    }
    lr.destroy();
    pic.destroy();
    x265_param_free(param);
    return rc ? rc : ncu;
}

static int frame_cost_recalculate_core(int qgSize, int width, int height, const uint16_t* lowresCosts, const double* qpCuTreeOffset, int32_t* rowSatds, int64_t* score);
int x265ref_frame_cost_recalculate(int width, int height, const uint16_t* lowresCosts, const double* qpCuTreeOffset, int32_t* rowSatds, int64_t* score)
{
    return frame_cost_recalculate_core(32, width, height, lowresCosts, qpCuTreeOffset, rowSatds, score);
}
int x265ref_frame_cost_recalculate_qg8(int width, int height, const uint16_t* lowresCosts, const double* qpCuTreeOffset, int32_t* rowSatds, int64_t* score)
{
    return frame_cost_recalculate_core(8, width, height, lowresCosts, qpCuTreeOffset, rowSatds, score);
}
static int frame_cost_recalculate_core(int qgSize, int width, int height, const uint16_t* lowresCosts, const double* qpCuTreeOffset, int32_t* rowSatds, int64_t* score)
{
    static bool tableReady = false;
    if (!tableReady) { x265ref_encoder_table_reset_c(); tableReady = true; }
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width;
    param->sourceHeight = height;
    param->internalCsp = X265_CSP_I400;
    param->maxCUSize = 64;
    param->rc.aqMode = 2;
    param->rc.hevcAq = 0;
    param->rc.qgSize = qgSize;
    param->bEnableHME = 0;
    PicYuv pic;
    pic.m_param = param;
    if (!pic.create(param, true)) return -1;
    Lowres lrs[2];
    for (int i = 0; i < 2; i++)
    {
        memset((void*)&lrs[i], 0, sizeof(Lowres));
        if (!lrs[i].create(param, &pic, qgSize)) return -2;
    }
    Lowres& fenc = lrs[1];
    fenc.sliceType = X265_TYPE_P;
    const int ncu = fenc.maxBlocksInRow * fenc.maxBlocksInCol;
    int rc = 0;
    {
        LookaheadProbe la(param, NULL);
        if (!la.create()) rc = -3;
        else
        {
            for (int k = 0; k < ncu; k++) fenc.lowresCosts[1][0][k] = lowresCosts[k];
            for (int k = 0; k < (qgSize == 8 ? 4 * ncu : ncu); k++) fenc.qpCuTreeOffset[k] = qpCuTreeOffset[k];
            Lowres* frames[2] = { &lrs[0], &lrs[1] };
            *score = la.frameCostRecalculate(frames, 0, 1, 1);
            for (int y = 0; y < fenc.maxBlocksInCol; y++) rowSatds[y] = fenc.rowSatds[1][0][y];
            la.destroy();
        }
    }
    for (int i = 0; i < 2; i++) lrs[i].destroy();
    pic.destroy();
    x265_param_free(param);
    return rc ? rc : ncu;
}

} // extern "C"
