/* oracle/x265_oracle_pipeline2.c
 *
 * TEST INFRASTRUCTURE - NOT PRODUCT CODE (same rules as x265_oracle.c).
 *
 * Stage: inter prediction + residual coding round trip of one square PU/TU per block.
 * Restates, on top of the oracle's primitive table, the call sequence of
 *   Predict::predInterLumaPixel        (source/common/predict.cpp:245-265: copy_pp / luma_hpp / luma_vpp / luma_hvpp),
 *   calcresidual                        (source/encoder/search.cpp:357 / pixel.cpp:471-483),
 *   Quant::transformNxN, non-RDOQ path  (source/common/quant.cpp:397-480: dct, quant with flat scaling
 *                                        s_quantScales[rem] (scalinglist.cpp:129,386), qbits = 14 + per + transformShift,
 *                                        add = (I ? 171 : 85) << (qbits - 9); sign hiding off),
 *   Quant::invtransformNxN              (quant.cpp:543-605: dequant_normal with s_invQuantScales[rem] << per,
 *                                        the DC-only blockfill shortcut :586-598, else idct),
 *   add_ps / copy when no coefficient survives, and sse_pp (search.cpp:367-375).
 */
#ifndef X265HIP_DEPTH
#error "compile with -DX265HIP_DEPTH=8|10|12"
#endif
#include "x265hip_table.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef x265hip_pixel pixel;
#define CAT_(a, b)   a##b
#define CAT(a, b)    CAT_(a, b)
#define EXPORT(name) CAT(CAT(name, _d), X265HIP_DEPTH)

void EXPORT(x265oracle_setup_primitives)(x265hip_EncoderPrimitives* p);
void EXPORT(x265oracle_prims_once)(x265hip_EncoderPrimitives* p, int* state);

static const int kQuantScales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };   /* scalinglist.cpp:129 */
static const int kInvQuantScales[6] = { 40, 45, 51, 57, 64, 72 };                  /* scalinglist.cpp:130 */
static const int kLvlBase[4] = { 0, 64, 80, 84 };

static void zxy(int z, int* x, int* y)
{
    *x = (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4);
    *y = ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4);
}

/* level: 0..2 -> 8x8, 16x16, 32x32 blocks.  mv: int32 [ctu*85][2] = {cost, qx | qy << 16} from the sub-pel stage.
 * Outputs: recon plane (same geometry as fenc), levels int16 [ctu][npu][n*n], numSig uint32 [ctu][npu],
 * dist uint64 [ctu][npu]. */
/* ---- sign-bit hiding of the non-RDOQ quantiser: Quant::signBitHidingHDQ (quant.cpp:247-395), called by transformNxN when
 * numSig >= 2 and pps.bSignHideEnabled (quant.cpp:471-476) - the x265 default.  Per 4x4 coefficient group in scan order whose first
 * and last non-zero level lie >= SBH_THRESHOLD (4, common.h:281) positions apart, the sign of the first non-zero level is inferred
 * from the parity of the group's sum; when the parity is wrong, the level whose change costs least (deltaU from the quantiser,
 * dct.cpp:664-686) moves by one towards / away from zero.  Scan orders: the up-right diagonal / horizontal / vertical scans of the
 * standard (6.5.3-6.5.5) over 4x4 groups; equal to the reference's g_scanOrder (checked in tests/test_oracle_classes_vs_reference.py). */
enum { ORACLE_SCAN_DIAG = 0, ORACLE_SCAN_HOR = 1, ORACLE_SCAN_VER = 2 };
#define TU_FLAG_INTRA_SLICE 1
#define TU_FLAG_SIGN_HIDE   2

static void scan_xy(int type, int n, int idx, int* x, int* y)      /* idx-th position of an n x n grid in scan `type` */
{
    if (type == ORACLE_SCAN_HOR) { *x = idx % n; *y = idx / n; return; }
    if (type == ORACLE_SCAN_VER) { *x = idx / n; *y = idx % n; return; }
    int k = 0;
    for (int d = 0; d < 2 * n - 1; d++)
        for (int yy = d < n - 1 ? d : n - 1; yy >= 0; yy--)
        {
            const int xx = d - yy;
            if (xx >= n) break;
            if (k++ == idx) { *x = xx; *y = yy; return; }
        }
    *x = *y = 0;
}

void EXPORT(x265oracle_scan_order)(int type, int log2n, uint16_t* out)
{
    const int n = 1 << log2n, g = n >> 2;
    if (log2n > 3) type = ORACLE_SCAN_DIAG;                               /* mode-dependent scans exist for 4x4 and 8x8 only */
    for (int cg = 0; cg < g * g; cg++)
    {
        int cx, cy;
        scan_xy(type, g, cg, &cx, &cy);
        for (int i = 0; i < 16; i++)
        {
            int ix, iy;
            scan_xy(type, 4, i, &ix, &iy);
            out[cg * 16 + i] = (uint16_t)((cy * 4 + iy) * n + cx * 4 + ix);
        }
    }
}

/* CUData::getTUEntropyCodingParameters' scan choice (cudata.cpp:2067-2089) for an intra TU of a 4:2:0 picture */
static int intra_scan_type(int mode, int n, int chroma)
{
    if (!(n == 4 || (!chroma && n == 8))) return ORACLE_SCAN_DIAG;
    return mode >= 22 && mode <= 30 ? ORACLE_SCAN_HOR : (mode >= 6 && mode <= 14 ? ORACLE_SCAN_VER : ORACLE_SCAN_DIAG);
}

static uint32_t sign_hide(int16_t* level, const int32_t* deltaU, const int16_t* dct, uint32_t numSig, int scanType, int log2n)
{
    uint16_t scan[32 * 32];
    EXPORT(x265oracle_scan_order)(scanType, log2n, scan);
    const int ncg = 1 << (2 * log2n - 4);
    int cgLast = -1;
    for (int i = (ncg << 4) - 1; i >= 0 && cgLast < 0; i--) if (level[scan[i]]) cgLast = i >> 4;
    for (int cg = cgLast; cg >= 0; cg--)
    {
        const uint16_t* sc = scan + cg * 16;
        int first = -1, last = -1, sum = 0;
        for (int i = 0; i < 16; i++) if (level[sc[i]]) { if (first < 0) first = i; last = i; sum += level[sc[i]]; }
        if (first < 0 || last - first < 4) continue;
        const int signbit = level[sc[first]] > 0 ? 0 : 1;
        if (signbit == (sum & 1)) continue;
        int minCost = 0x7fffffff, minPos = -1, change = 0;
        for (int i = cg == cgLast ? last : 15; i >= 0; i--)
        {
            const int pos = sc[i];
            int cost = 0x7fffffff, ch = 0;
            if (level[pos])
            {
                if (deltaU[pos] > 0) { cost = -deltaU[pos]; ch = 1; }
                else if (!(i == first && (level[pos] == 1 || level[pos] == -1))) { cost = deltaU[pos]; ch = -1; }
            }
            else if (i > first || (dct[pos] >= 0 ? 0 : 1) == signbit) { cost = -deltaU[pos]; ch = 1; }
            if (cost < minCost) { minCost = cost; minPos = pos; change = ch; }
        }
        if (level[minPos] == 32767 || level[minPos] == -32768) change = -1;
        if (!level[minPos]) numSig++;
        else if (change == -1 && (level[minPos] == 1 || level[minPos] == -1)) numSig--;
        level[minPos] = (int16_t)(level[minPos] + (dct[minPos] < 0 ? -change : change));
    }
    return numSig;
}

/* ---- optional per-coefficient tables of the TU stages (x265hip_tu_tables): the scaling list's quantiser / dequantiser coefficients
 * (quant.cpp:463, :562-567 with dequant_scaling, dct.cpp:612-662) and the denoiser (primitives.denoiseDct before the quantiser,
 * quant.cpp:444-451, dct.cpp:744-755).  Test infrastructure: set once per test, read by the four recon functions below. */
static const int32_t* g_tabQuant; static const int32_t* g_tabDequant; static const uint16_t* g_tabNrOffset; static uint32_t* g_tabNrSum;
void EXPORT(x265oracle_set_tu_tables)(const int32_t* quantCoeff, const int32_t* dequantCoeff, const uint16_t* nrOffset, uint32_t* nrSum)
{
    g_tabQuant = quantCoeff; g_tabDequant = dequantCoeff; g_tabNrOffset = nrOffset; g_tabNrSum = nrSum;
}
/* capture of what a host-side RDOQ pass needs (x265hip_tu_tables.dct_coeff_out / delta_u_out): laid out like the levels */
static int16_t* g_capDct; static int32_t* g_capDeltaU;
void EXPORT(x265oracle_set_tu_capture)(int16_t* dctCoeff, int32_t* deltaU) { g_capDct = dctCoeff; g_capDeltaU = deltaU; }
static void tab_capture(const int16_t* coef, const int32_t* deltaU, size_t elemOff, int num)
{
    if (g_capDct) memcpy(g_capDct + elemOff, coef, (size_t)num * sizeof(int16_t));
    if (g_capDeltaU) memcpy(g_capDeltaU + elemOff, deltaU, (size_t)num * sizeof(int32_t));
}
/* capture of the PREDICTION of every block (before the residual round trip) into an unpadded plane of the stage's geometry - lets
 * tests compare the prediction half of the inter stages with the real Predict::motionCompensation (oracle/ref_predict.cpp) */
static pixel* g_capPred; static intptr_t g_capPredStride;
void EXPORT(x265oracle_set_pred_capture)(pixel* plane, intptr_t stride) { g_capPred = plane; g_capPredStride = stride; }
static void cap_pred(const pixel* pred, int predStride, int px, int py, int n)
{
    if (!g_capPred) return;
    for (int y = 0; y < n; y++) memcpy(g_capPred + (intptr_t)(py + y) * g_capPredStride + px, pred + y * predStride, sizeof(pixel) * n);
}
static void tab_denoise(int16_t* coef, int num)          /* denoiseDct with the sums added atomically (the callers run CTUs in parallel) */
{
    if (!g_tabNrOffset) return;
    for (int i = 0; i < num; i++)
    {
        int level = coef[i];
        const int sign = level >> 31;
        level = (level + sign) ^ sign;
        __atomic_fetch_add(&g_tabNrSum[i], (uint32_t)level, __ATOMIC_RELAXED);
        level -= g_tabNrOffset[i];
        coef[i] = (int16_t)(level < 0 ? 0 : (level ^ sign) - sign);
    }
}
static void tab_dequant(const x265hip_EncoderPrimitives* prim, const int16_t* q, int16_t* coef, int num, int per, int dqScale, int dqShift)
{
    if (g_tabDequant) prim->dequant_scaling(q, g_tabDequant, coef, num, per, dqShift);
    else prim->dequant_normal(q, coef, num, dqScale, dqShift);
}

int EXPORT(x265oracle_inter_recon)(const pixel* fenc, intptr_t fencStride, const pixel* fref, intptr_t frefStride,
                                   pixel* recon, intptr_t reconStride, int width, int height, int level,
                                   const int32_t* mv, int qp, int flags,
                                   int16_t* levels, uint32_t* numSigOut, uint64_t* distOut,
                                   int ctuBegin, int ctuEnd, int nthreads)
{
    static x265hip_EncoderPrimitives prim;
    static int ready = 0;
    EXPORT(x265oracle_prims_once)(&prim, &ready);
    const int ctusW = width / 64;
    const int n = 8 << level, log2n = 3 + level, npu = (64 / n) * (64 / n);
    const int puIdx = level == 0 ? X265HIP_LUMA_8x8 : (level == 1 ? X265HIP_LUMA_16x16 : X265HIP_LUMA_32x32);
    const struct x265hip_PU* pu = &prim.pu[puIdx];
    const struct x265hip_CU* cu = &prim.cu[log2n - 2];
    const int per = qp / 6, rem = qp % 6;
    const int transformShift = 15 - X265HIP_DEPTH - log2n;
    const int qbits = 14 + per + transformShift;
    const int add = ((flags & TU_FLAG_INTRA_SLICE) ? 171 : 85) << (qbits - 9);
    const int dqShift = 20 - 14 - transformShift;
    const int dqScale = kInvQuantScales[rem] << per;
    (void)height;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int ctu = ctuBegin; ctu < ctuEnd; ctu++)
    {
        const int cx = (ctu % ctusW) * 64, cy = (ctu / ctusW) * 64;
        pixel pred[64 * 64] __attribute__((aligned(64)));
        int16_t resi[64 * 64] __attribute__((aligned(64)));
        int16_t coef[32 * 32] __attribute__((aligned(64)));
        int32_t quantCoeff[32 * 32] __attribute__((aligned(64)));
        int32_t deltaU[32 * 32];
        for (int i = 0; i < n * n; i++) quantCoeff[i] = g_tabQuant ? g_tabQuant[i] : kQuantScales[rem];
        for (int z = 0; z < npu; z++)
        {
            int bx, by;
            zxy(z, &bx, &by);
            const int px = cx + bx * n, py = cy + by * n;
            const int32_t packed = mv[((size_t)ctu * 85 + kLvlBase[level] + z) * 2 + 1];
            const int qx = (int16_t)(packed & 0xffff), qy = (int16_t)(packed >> 16);
            const pixel* src = fref + (intptr_t)(py + (qy >> 2)) * frefStride + px + (qx >> 2);
            const pixel* fe = fenc + (intptr_t)py * fencStride + px;
            pixel* rec = recon + (intptr_t)py * reconStride + px;
            const int xf = qx & 3, yf = qy & 3;
            /* predInterLumaPixel */
            if (!(xf | yf)) pu->copy_pp(pred, 64, src, frefStride);
            else if (!yf) pu->luma_hpp(src, frefStride, pred, 64, xf);
            else if (!xf) pu->luma_vpp(src, frefStride, pred, 64, yf);
            else pu->luma_hvpp(src, frefStride, pred, 64, xf, yf);
            /* residual, transform, quantisation */
            cap_pred(pred, 64, px, py, n);
            cu->sub_ps(resi, 64, fe, pred, fencStride, 64);
            cu->dct(resi, coef, 64);
            tab_denoise(coef, n * n);
            int16_t* q = levels + ((size_t)ctu * npu + z) * n * n;
            uint32_t numSig = prim.quant(coef, quantCoeff, deltaU, q, qbits, add, n * n);
            tab_capture(coef, deltaU, (size_t)(q - levels), n * n);
            if ((flags & TU_FLAG_SIGN_HIDE) && numSig >= 2) numSig = sign_hide(q, deltaU, coef, numSig, ORACLE_SCAN_DIAG, log2n);
            numSigOut[(size_t)ctu * npu + z] = numSig;
            if (numSig)
            {
                tab_dequant(&prim, q, coef, n * n, per, dqScale, dqShift);
                if (numSig == 1 && q[0] != 0)
                {
                    const int shift_2nd = 12 - (X265HIP_DEPTH - 8) - 3;
                    const int dc = ((((coef[0] * (64 >> 6) + 1) >> 1) * (64 >> 3)) + (1 << (shift_2nd - 1))) >> shift_2nd;
                    cu->blockfill_s[0](resi, 64, (int16_t)dc);
                }
                else
                    cu->idct(coef, resi, 64);
                cu->add_ps[0](rec, reconStride, pred, resi, 64, 64);
            }
            else
                cu->copy_pp(rec, reconStride, pred, 64);
            distOut[(size_t)ctu * npu + z] = (uint64_t)cu->sse_pp(fe, fencStride, rec, reconStride);
        }
    }
    return 0;
}

/* The same stage with bi-prediction (B pictures): per block `dir` = 1 (list 0 only), 2 (list 1 only) or 3 (both): Predict::
 * motionCompensation (predict.cpp:168-243, no weighted prediction): uni-directional blocks as above; bi-directional blocks take
 * predInterLumaShort of each list (:267-304: convert_p2s / luma_hps / luma_vps / luma_hps with row extension + luma_vss) and
 * combine them with addAvg (pixel.cpp: (a + b + offset) >> shift at 14-bit intermediate precision).  mv0 / mv1: the two lists'
 * records in the sub-pel stage's format; dir: uint8 [ctu][npu] or NULL (all 3). */
/* Explicit weighted prediction of the bi-predictive stage (x265hip_recon_bi_params.weight0 / weight1): the two lists' WeightParam of
 * the plane as { wtPresent OF THE LUMA ENTRY (what the reference tests for every plane), inputWeight, inputOffset, log2WeightDenom },
 * NULL = that list has no table (P slices without
 * pps.bUseWeightPred, B slices without pps.bUseWeightedBiPred).  Predict::motionCompensation (predict.cpp:77-243): a block predicted
 * from one list whose table is present takes predInterLumaShort + addWeightUni (weight_sp, :525-545); a block predicted from both
 * takes addWeightBi (:411-456, weightBidir :52-55) when both tables exist and one is present, otherwise addAvg.  Test infrastructure:
 * set per test. */
static const int32_t* g_predW[2];
void EXPORT(x265oracle_set_pred_weights)(const int32_t* w0, const int32_t* w1) { g_predW[0] = w0; g_predW[1] = w1; }

int EXPORT(x265oracle_inter_recon_bi)(const pixel* fenc, intptr_t fencStride, const pixel* fref0, const pixel* fref1, intptr_t frefStride,
                                      pixel* recon, intptr_t reconStride, int width, int height, int level,
                                      const int32_t* mv0, const int32_t* mv1, const uint8_t* dir, int qp, int flags,
                                      int16_t* levels, uint32_t* numSigOut, uint64_t* distOut, int nthreads)
{
    static x265hip_EncoderPrimitives prim;
    static int ready = 0;
    EXPORT(x265oracle_prims_once)(&prim, &ready);
    const int ctusW = width / 64, nctu = ctusW * (height / 64);
    const int n = 8 << level, log2n = 3 + level, npu = (64 / n) * (64 / n);
    const int puIdx = level == 0 ? X265HIP_LUMA_8x8 : (level == 1 ? X265HIP_LUMA_16x16 : X265HIP_LUMA_32x32);
    const struct x265hip_PU* pu = &prim.pu[puIdx];
    const struct x265hip_CU* cu = &prim.cu[log2n - 2];
    const int per = qp / 6, rem = qp % 6;
    const int transformShift = 15 - X265HIP_DEPTH - log2n;
    const int qbits = 14 + per + transformShift;
    const int add = ((flags & TU_FLAG_INTRA_SLICE) ? 171 : 85) << (qbits - 9);
    const int dqShift = 20 - 14 - transformShift;
    const int dqScale = kInvQuantScales[rem] << per;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int ctu = 0; ctu < nctu; ctu++)
    {
        const int cx = (ctu % ctusW) * 64, cy = (ctu / ctusW) * 64;
        pixel pred[64 * 64] __attribute__((aligned(64)));
        int16_t ps[2][64 * 64] __attribute__((aligned(64)));
        int16_t immed[64 * (64 + 7)] __attribute__((aligned(64)));
        int16_t resi[64 * 64] __attribute__((aligned(64)));
        int16_t coef[32 * 32] __attribute__((aligned(64)));
        int32_t quantCoeff[32 * 32] __attribute__((aligned(64)));
        int32_t deltaU[32 * 32];
        for (int i = 0; i < n * n; i++) quantCoeff[i] = g_tabQuant ? g_tabQuant[i] : kQuantScales[rem];
        for (int z = 0; z < npu; z++)
        {
            int bx, by;
            zxy(z, &bx, &by);
            const int px = cx + bx * n, py = cy + by * n;
            const int d = dir ? dir[(size_t)ctu * npu + z] : 3;
            const pixel* fe = fenc + (intptr_t)py * fencStride + px;
            pixel* rec = recon + (intptr_t)py * reconStride + px;
            for (int l = 0; l < 2; l++)
            {
                if (!(d & (1 << l))) continue;
                const int32_t packed = (l ? mv1 : mv0)[((size_t)ctu * 85 + kLvlBase[level] + z) * 2 + 1];
                const int qx = (int16_t)(packed & 0xffff), qy = (int16_t)(packed >> 16);
                const pixel* src = (l ? fref1 : fref0) + (intptr_t)(py + (qy >> 2)) * frefStride + px + (qx >> 2);
                const int xf = qx & 3, yf = qy & 3;
                const int32_t* wl = g_predW[l];
                if (d != 3 && !(wl && wl[0]))
                {
                    /* predInterLumaPixel */
                    if (!(xf | yf)) pu->copy_pp(pred, 64, src, frefStride);
                    else if (!yf) pu->luma_hpp(src, frefStride, pred, 64, xf);
                    else if (!xf) pu->luma_vpp(src, frefStride, pred, 64, yf);
                    else pu->luma_hvpp(src, frefStride, pred, 64, xf, yf);
                }
                else
                {
                    /* predInterLumaShort */
                    if (!(xf | yf)) pu->convert_p2s[0](src, frefStride, ps[l], 64);
                    else if (!yf) pu->luma_hps(src, frefStride, ps[l], 64, xf, 0);
                    else if (!xf) pu->luma_vps(src, frefStride, ps[l], 64, yf);
                    else
                    {
                        pu->luma_hps(src, frefStride, immed, n, xf, 1);
                        pu->luma_vss(immed + 3 * n, n, ps[l], 64, yf);
                    }
                }
            }
            const int shiftNum = 14 - X265HIP_DEPTH;
            if (d == 3)
            {
                const int32_t* w0 = g_predW[0]; const int32_t* w1 = g_predW[1];
                if (w0 && w1 && (w0[0] || w1[0]))
                {
                    /* addWeightBi: wv.o = inputOffset << (depth - 8), the shift and the rounding of list 0's denominator for both */
                    const int offset = w0[2] * (1 << (X265HIP_DEPTH - 8)) + w1[2] * (1 << (X265HIP_DEPTH - 8));
                    const int shift = w0[3] + shiftNum + 1, round = shift ? (1 << (shift - 1)) : 0;
                    const int maxVal = (1 << X265HIP_DEPTH) - 1;
                    for (int y = 0; y < n; y++)
                        for (int x = 0; x < n; x++)
                        {
                            const int v = (w0[1] * (ps[0][y * 64 + x] + 8192) + w1[1] * (ps[1][y * 64 + x] + 8192) + round + (offset * (1 << (shift - 1)))) >> shift;
                            pred[y * 64 + x] = (pixel)(v < 0 ? 0 : (v > maxVal ? maxVal : v));
                        }
                }
                else
                    pu->addAvg[0](ps[0], ps[1], pred, 64, 64, 64);
            }
            else
            {
                const int l = d == 2;
                const int32_t* wl = g_predW[l];
                if (wl && wl[0])                           /* addWeightUni through the weight_sp primitive */
                {
                    const int shift = wl[3] + shiftNum, round = shift ? (1 << (shift - 1)) : 0;
                    prim.weight_sp(ps[l], pred, 64, 64, n, n, wl[1], round, shift, wl[2] * (1 << (X265HIP_DEPTH - 8)));
                }
            }
            cap_pred(pred, 64, px, py, n);
            cu->sub_ps(resi, 64, fe, pred, fencStride, 64);
            cu->dct(resi, coef, 64);
            tab_denoise(coef, n * n);
            int16_t* q = levels + ((size_t)ctu * npu + z) * n * n;
            uint32_t numSig = prim.quant(coef, quantCoeff, deltaU, q, qbits, add, n * n);
            tab_capture(coef, deltaU, (size_t)(q - levels), n * n);
            if ((flags & TU_FLAG_SIGN_HIDE) && numSig >= 2) numSig = sign_hide(q, deltaU, coef, numSig, ORACLE_SCAN_DIAG, log2n);
            numSigOut[(size_t)ctu * npu + z] = numSig;
            if (numSig)
            {
                tab_dequant(&prim, q, coef, n * n, per, dqScale, dqShift);
                if (numSig == 1 && q[0] != 0)
                {
                    const int shift_2nd = 12 - (X265HIP_DEPTH - 8) - 3;
                    const int dc = ((((coef[0] * (64 >> 6) + 1) >> 1) * (64 >> 3)) + (1 << (shift_2nd - 1))) >> shift_2nd;
                    cu->blockfill_s[0](resi, 64, (int16_t)dc);
                }
                else
                    cu->idct(coef, resi, 64);
                cu->add_ps[0](rec, reconStride, pred, resi, 64, 64);
            }
            else
                cu->copy_pp(rec, reconStride, pred, 64);
            distOut[(size_t)ctu * npu + z] = (uint64_t)cu->sse_pp(fe, fencStride, rec, reconStride);
        }
    }
    return 0;
}

/* One chroma plane of the same stage for 4:2:0 pictures: Predict::predInterChromaPixel (predict.cpp:304-351: the luma mv in 1/8
 * chroma samples, 4-tap filters - copy_pp / filter_hpp / filter_vpp / filter_hps (+3 rows) + filter_vsp of the chroma table) and
 * the residual round trip on (n/2) x (n/2) blocks (DCT also for 4x4: DST-VII is intra luma only).  fenc / fref / recon: sample
 * (0,0) of the chroma planes; width / height: LUMA size; qp: the plane's quantiser QP (chroma mapping and offsets applied by the
 * caller, + QP_BD_OFFSET).  Outputs as x265oracle_inter_recon with (n/2)^2 levels per block. */
int EXPORT(x265oracle_inter_recon_chroma)(const pixel* fenc, intptr_t fencStride, const pixel* fref, intptr_t frefStride,
                                          pixel* recon, intptr_t reconStride, int width, int height, int level,
                                          const int32_t* mv, int qp, int flags,
                                          int16_t* levels, uint32_t* numSigOut, uint64_t* distOut, int nthreads)
{
    static x265hip_EncoderPrimitives prim;
    static int ready = 0;
    EXPORT(x265oracle_prims_once)(&prim, &ready);
    const int ctusW = width / 64, nctu = ctusW * (height / 64);
    const int n = 8 << level, nc = n >> 1, log2nc = 2 + level, npu = (64 / n) * (64 / n);
    const int puIdx = level == 0 ? X265HIP_LUMA_8x8 : (level == 1 ? X265HIP_LUMA_16x16 : X265HIP_LUMA_32x32);
    const struct x265hip_PUChroma* pu = &prim.chroma[1].pu[puIdx];       /* X265_CSP_I420 */
    const struct x265hip_CU* cu = &prim.cu[log2nc - 2];
    const int per = qp / 6, rem = qp % 6;
    const int transformShift = 15 - X265HIP_DEPTH - log2nc;
    const int qbits = 14 + per + transformShift;
    const int add = ((flags & TU_FLAG_INTRA_SLICE) ? 171 : 85) << (qbits - 9);
    const int dqShift = 20 - 14 - transformShift;
    const int dqScale = kInvQuantScales[rem] << per;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int ctu = 0; ctu < nctu; ctu++)
    {
        const int cx = (ctu % ctusW) * 32, cy = (ctu / ctusW) * 32;
        pixel pred[32 * 32] __attribute__((aligned(64)));
        int16_t resi[32 * 32] __attribute__((aligned(64)));
        int16_t coef[16 * 16] __attribute__((aligned(64)));
        int16_t immed[32 * (32 + 3)] __attribute__((aligned(64)));
        int32_t quantCoeff[16 * 16] __attribute__((aligned(64)));
        int32_t deltaU[16 * 16];
        for (int i = 0; i < nc * nc; i++) quantCoeff[i] = g_tabQuant ? g_tabQuant[i] : kQuantScales[rem];
        for (int z = 0; z < npu; z++)
        {
            int bx, by;
            zxy(z, &bx, &by);
            const int px = cx + bx * nc, py = cy + by * nc;
            const int32_t packed = mv[((size_t)ctu * 85 + kLvlBase[level] + z) * 2 + 1];
            const int qx = (int16_t)(packed & 0xffff), qy = (int16_t)(packed >> 16);       /* 1/4 luma = 1/8 chroma samples */
            const pixel* src = fref + (intptr_t)(py + (qy >> 3)) * frefStride + px + (qx >> 3);
            const pixel* fe = fenc + (intptr_t)py * fencStride + px;
            pixel* rec = recon + (intptr_t)py * reconStride + px;
            const int xf = qx & 7, yf = qy & 7;
            if (!(xf | yf)) pu->copy_pp(pred, 32, src, frefStride);
            else if (!yf) pu->filter_hpp(src, frefStride, pred, 32, xf);
            else if (!xf) pu->filter_vpp(src, frefStride, pred, 32, yf);
            else
            {
                pu->filter_hps(src, frefStride, immed, nc, xf, 1);
                pu->filter_vsp(immed + 1 * nc, nc, pred, 32, yf);
            }
            cap_pred(pred, 32, px, py, nc);
            cu->sub_ps(resi, 32, fe, pred, fencStride, 32);
            cu->dct(resi, coef, 32);
            tab_denoise(coef, nc * nc);
            int16_t* q = levels + ((size_t)ctu * npu + z) * nc * nc;
            uint32_t numSig = prim.quant(coef, quantCoeff, deltaU, q, qbits, add, nc * nc);
            tab_capture(coef, deltaU, (size_t)(q - levels), nc * nc);
            if ((flags & TU_FLAG_SIGN_HIDE) && numSig >= 2) numSig = sign_hide(q, deltaU, coef, numSig, ORACLE_SCAN_DIAG, log2nc);
            numSigOut[(size_t)ctu * npu + z] = numSig;
            if (numSig)
            {
                tab_dequant(&prim, q, coef, nc * nc, per, dqScale, dqShift);
                if (numSig == 1 && q[0] != 0)
                {
                    const int shift_2nd = 12 - (X265HIP_DEPTH - 8) - 3;
                    const int dc = ((((coef[0] * (64 >> 6) + 1) >> 1) * (64 >> 3)) + (1 << (shift_2nd - 1))) >> shift_2nd;
                    cu->blockfill_s[0](resi, 32, (int16_t)dc);
                }
                else
                    cu->idct(coef, resi, 32);
                cu->add_ps[0](rec, reconStride, pred, resi, 32, 32);
            }
            else
                cu->copy_pp(rec, reconStride, pred, 32);
            distOut[(size_t)ctu * npu + z] = (uint64_t)cu->sse_pp(fe, fencStride, rec, reconStride);
        }
    }
    return 0;
}

/* One chroma plane of the bi-predictive stage (B pictures, and weighted P pictures): Predict::motionCompensation's chroma half
 * (predict.cpp:77-243) - a block of one list predInterChromaPixel (:304-351), or with a present weight table predInterChromaShort
 * (:355-409: p2s / filter_hps / filter_vps / filter_hps with row extension + filter_vss) + addWeightUni (:547-576); a block of both
 * lists the two short predictions combined by addAvg or, with tables, addWeightBi (:458-520).  The plane's own weights come from
 * x265oracle_set_pred_weights.  Block geometry and outputs as x265oracle_inter_recon_chroma. */
int EXPORT(x265oracle_inter_recon_chroma_bi)(const pixel* fenc, intptr_t fencStride, const pixel* fref0, const pixel* fref1, intptr_t frefStride,
                                             pixel* recon, intptr_t reconStride, int width, int height, int level,
                                             const int32_t* mv0, const int32_t* mv1, const uint8_t* dir, int qp, int flags,
                                             int16_t* levels, uint32_t* numSigOut, uint64_t* distOut, int nthreads)
{
    static x265hip_EncoderPrimitives prim;
    static int ready = 0;
    EXPORT(x265oracle_prims_once)(&prim, &ready);
    const int ctusW = width / 64, nctu = ctusW * (height / 64);
    const int n = 8 << level, nc = n >> 1, log2nc = 2 + level, npu = (64 / n) * (64 / n);
    const int puIdx = level == 0 ? X265HIP_LUMA_8x8 : (level == 1 ? X265HIP_LUMA_16x16 : X265HIP_LUMA_32x32);
    const struct x265hip_PUChroma* pu = &prim.chroma[1].pu[puIdx];       /* X265_CSP_I420 */
    const struct x265hip_CU* cu = &prim.cu[log2nc - 2];
    const int per = qp / 6, rem = qp % 6;
    const int transformShift = 15 - X265HIP_DEPTH - log2nc;
    const int qbits = 14 + per + transformShift;
    const int add = ((flags & TU_FLAG_INTRA_SLICE) ? 171 : 85) << (qbits - 9);
    const int dqShift = 20 - 14 - transformShift;
    const int dqScale = kInvQuantScales[rem] << per;
    const int shiftNum = 14 - X265HIP_DEPTH, maxVal = (1 << X265HIP_DEPTH) - 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int ctu = 0; ctu < nctu; ctu++)
    {
        const int cx = (ctu % ctusW) * 32, cy = (ctu / ctusW) * 32;
        pixel pred[32 * 32] __attribute__((aligned(64)));
        int16_t ps[2][32 * 32] __attribute__((aligned(64)));
        int16_t resi[32 * 32] __attribute__((aligned(64)));
        int16_t coef[16 * 16] __attribute__((aligned(64)));
        int16_t immed[32 * (32 + 3)] __attribute__((aligned(64)));
        int32_t quantCoeff[16 * 16] __attribute__((aligned(64)));
        int32_t deltaU[16 * 16];
        for (int i = 0; i < nc * nc; i++) quantCoeff[i] = g_tabQuant ? g_tabQuant[i] : kQuantScales[rem];
        for (int z = 0; z < npu; z++)
        {
            int bx, by;
            zxy(z, &bx, &by);
            const int px = cx + bx * nc, py = cy + by * nc;
            const int d = dir ? dir[(size_t)ctu * npu + z] : 3;
            const pixel* fe = fenc + (intptr_t)py * fencStride + px;
            pixel* rec = recon + (intptr_t)py * reconStride + px;
            for (int l = 0; l < 2; l++)
            {
                if (!(d & (1 << l))) continue;
                const int32_t packed = (l ? mv1 : mv0)[((size_t)ctu * 85 + kLvlBase[level] + z) * 2 + 1];
                const int qx = (int16_t)(packed & 0xffff), qy = (int16_t)(packed >> 16);       /* 1/4 luma = 1/8 chroma samples */
                const pixel* src = (l ? fref1 : fref0) + (intptr_t)(py + (qy >> 3)) * frefStride + px + (qx >> 3);
                const int xf = qx & 7, yf = qy & 7;
                const int32_t* wl = g_predW[l];
                if (d != 3 && !(wl && wl[0]))
                {
                    if (!(xf | yf)) pu->copy_pp(pred, 32, src, frefStride);
                    else if (!yf) pu->filter_hpp(src, frefStride, pred, 32, xf);
                    else if (!xf) pu->filter_vpp(src, frefStride, pred, 32, yf);
                    else
                    {
                        pu->filter_hps(src, frefStride, immed, nc, xf, 1);
                        pu->filter_vsp(immed + 1 * nc, nc, pred, 32, yf);
                    }
                }
                else
                {
                    if (!(xf | yf)) pu->p2s[0](src, frefStride, ps[l], 32);
                    else if (!yf) pu->filter_hps(src, frefStride, ps[l], 32, xf, 0);
                    else if (!xf) pu->filter_vps(src, frefStride, ps[l], 32, yf);
                    else
                    {
                        pu->filter_hps(src, frefStride, immed, nc, xf, 1);
                        pu->filter_vss(immed + 1 * nc, nc, ps[l], 32, yf);
                    }
                }
            }
            if (d == 3)
            {
                const int32_t* w0 = g_predW[0]; const int32_t* w1 = g_predW[1];
                if (w0 && w1 && (w0[0] || w1[0]))
                {
                    const int offset = w0[2] * (1 << (X265HIP_DEPTH - 8)) + w1[2] * (1 << (X265HIP_DEPTH - 8));
                    const int shift = w0[3] + shiftNum + 1, round = shift ? (1 << (shift - 1)) : 0;
                    for (int y = 0; y < nc; y++)
                        for (int x = 0; x < nc; x++)
                        {
                            const int v = (w0[1] * (ps[0][y * 32 + x] + 8192) + w1[1] * (ps[1][y * 32 + x] + 8192) + round + (offset * (1 << (shift - 1)))) >> shift;
                            pred[y * 32 + x] = (pixel)(v < 0 ? 0 : (v > maxVal ? maxVal : v));
                        }
                }
                else
                    pu->addAvg[0](ps[0], ps[1], pred, 32, 32, 32);
            }
            else
            {
                const int l = d == 2;
                const int32_t* wl = g_predW[l];
                if (wl && wl[0])
                {
                    const int shift = wl[3] + shiftNum, round = shift ? (1 << (shift - 1)) : 0;
                    prim.weight_sp(ps[l], pred, 32, 32, nc, nc, wl[1], round, shift, wl[2] * (1 << (X265HIP_DEPTH - 8)));
                }
            }
            cap_pred(pred, 32, px, py, nc);
            cu->sub_ps(resi, 32, fe, pred, fencStride, 32);
            cu->dct(resi, coef, 32);
            tab_denoise(coef, nc * nc);
            int16_t* q = levels + ((size_t)ctu * npu + z) * nc * nc;
            uint32_t numSig = prim.quant(coef, quantCoeff, deltaU, q, qbits, add, nc * nc);
            tab_capture(coef, deltaU, (size_t)(q - levels), nc * nc);
            if ((flags & TU_FLAG_SIGN_HIDE) && numSig >= 2) numSig = sign_hide(q, deltaU, coef, numSig, ORACLE_SCAN_DIAG, log2nc);
            numSigOut[(size_t)ctu * npu + z] = numSig;
            if (numSig)
            {
                tab_dequant(&prim, q, coef, nc * nc, per, dqScale, dqShift);
                if (numSig == 1 && q[0] != 0)
                {
                    const int shift_2nd = 12 - (X265HIP_DEPTH - 8) - 3;
                    const int dc = ((((coef[0] * (64 >> 6) + 1) >> 1) * (64 >> 3)) + (1 << (shift_2nd - 1))) >> shift_2nd;
                    cu->blockfill_s[0](resi, 32, (int16_t)dc);
                }
                else
                    cu->idct(coef, resi, 32);
                cu->add_ps[0](rec, reconStride, pred, resi, 32, 32);
            }
            else
                cu->copy_pp(rec, reconStride, pred, 32);
            distOut[(size_t)ctu * npu + z] = (uint64_t)cu->sse_pp(fe, fencStride, rec, reconStride);
        }
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Intra TU candidate set: the pixel work of Search::codeIntraLumaQT for one (TU, mode) candidate
 * (source/encoder/search.cpp:335-373): Predict::predIntraLumaAng (predict.cpp:579-588: filtered neighbours per
 * g_intraFilterFlags & size, edge filter for sizes <= 16), calcresidual, Quant::transformNxN (DST for the 4x4 luma
 * intra TU, quant.cpp:426-431), Quant::invtransformNxN (no DC shortcut under DST, quant.cpp:583-603), add_ps / copy_pp
 * into the candidate's reconstruction, sse_pp.  Bit costs (CABAC) stay with the host.
 * jobs: { off[0] fenc block, off[1] unfiltered neighbours, off[2] filtered neighbours, off[3] reconstruction block;
 *         arg[0] intra mode 0..34 } with element offsets into fenc / nb / recon. */
typedef struct { int64_t off[4]; int32_t arg[4]; } intra_job;
static const uint8_t kIntraFilterFlags[35] = {
    0x38, 0x00,
    0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30,
    0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30,
    0x38 };

static int intra_recon_core(const pixel* fenc, intptr_t fencStride, const pixel* nb, pixel* recon, intptr_t reconStride,
                            int n, int qp, int flags, const intra_job* jobs, int njobs,
                            int16_t* levels, uint32_t* numSigOut, uint64_t* distOut, int nthreads, int chroma);

int EXPORT(x265oracle_intra_recon)(const pixel* fenc, intptr_t fencStride, const pixel* nb, pixel* recon, intptr_t reconStride,
                                   int n, int qp, int flags, const intra_job* jobs, int njobs,
                                   int16_t* levels, uint32_t* numSigOut, uint64_t* distOut, int nthreads)
{
    return intra_recon_core(fenc, fencStride, nb, recon, reconStride, n, qp, flags, jobs, njobs, levels, numSigOut, distOut, nthreads, 0);
}

/* The chroma flavour for 4:2:0 (Search::codeIntraChromaQt's pixel work, search.cpp:899-930): Predict::predIntraChromaAng
 * (predict.cpp:590-598) always predicts from the UNFILTERED neighbours with bFilter = 0 (no DC / vertical / horizontal edge
 * smoothing), and the 4x4 TU uses the DCT (useDST needs TEXT_LUMA, quant.cpp:426,583).  qp = the chroma QP the host mapped
 * (Quant::setChromaQP, + QP_BD_OFFSET); fenc / nb / recon are the chroma plane's. */
int EXPORT(x265oracle_intra_recon_chroma)(const pixel* fenc, intptr_t fencStride, const pixel* nb, pixel* recon, intptr_t reconStride,
                                          int n, int qp, int flags, const intra_job* jobs, int njobs,
                                          int16_t* levels, uint32_t* numSigOut, uint64_t* distOut, int nthreads)
{
    return intra_recon_core(fenc, fencStride, nb, recon, reconStride, n, qp, flags, jobs, njobs, levels, numSigOut, distOut, nthreads, 1);
}

static int intra_recon_core(const pixel* fenc, intptr_t fencStride, const pixel* nb, pixel* recon, intptr_t reconStride,
                            int n, int qp, int flags, const intra_job* jobs, int njobs,
                            int16_t* levels, uint32_t* numSigOut, uint64_t* distOut, int nthreads, int chroma)
{
    static x265hip_EncoderPrimitives prim;
    static int ready = 0;
    EXPORT(x265oracle_prims_once)(&prim, &ready);
    const int log2n = n == 4 ? 2 : (n == 8 ? 3 : (n == 16 ? 4 : 5));
    if ((1 << log2n) != n) return -1;
    const struct x265hip_CU* cu = &prim.cu[log2n - 2];
    const int useDST = n == 4 && !chroma;
    const int per = qp / 6, rem = qp % 6;
    const int transformShift = 15 - X265HIP_DEPTH - log2n;
    const int qbits = 14 + per + transformShift;
    const int add = ((flags & TU_FLAG_INTRA_SLICE) ? 171 : 85) << (qbits - 9);
    const int dqShift = 20 - 14 - transformShift;
    const int dqScale = kInvQuantScales[rem] << per;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 16)
#endif
    for (int j = 0; j < njobs; j++)
    {
        pixel pred[32 * 32] __attribute__((aligned(64)));
        int16_t resi[32 * 32] __attribute__((aligned(64)));
        int16_t coef[32 * 32] __attribute__((aligned(64)));
        int32_t quantCoeff[32 * 32] __attribute__((aligned(64)));
        int32_t deltaU[32 * 32];
        for (int i = 0; i < n * n; i++) quantCoeff[i] = g_tabQuant ? g_tabQuant[i] : kQuantScales[rem];
        const intra_job* jb = &jobs[j];
        const int mode = jb->arg[0];
        const pixel* fe = fenc + jb->off[0];
        pixel* rec = recon + jb->off[3];
        const int filter = !chroma && (kIntraFilterFlags[mode] & n);
        cu->intra_pred[mode](pred, n, nb + (filter ? jb->off[2] : jb->off[1]), mode, chroma ? 0 : log2n <= 4);
        if (g_capPred)                    /* prediction capture (x265oracle_set_pred_capture): laid out like the candidates' reconstructions */
            for (int y = 0; y < n; y++) memcpy(g_capPred + jb->off[3] + (intptr_t)y * g_capPredStride, pred + y * n, sizeof(pixel) * n);
        /* calcresidual assumes one stride for fenc / pred / residual (search.cpp:357); restate it for separate strides */
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++) resi[y * n + x] = (int16_t)((int)fe[y * fencStride + x] - (int)pred[y * n + x]);
        if (useDST) prim.dst4x4(resi, coef, n);
        else cu->dct(resi, coef, n);
        tab_denoise(coef, n * n);
        int16_t* q = levels + (size_t)j * n * n;
        uint32_t numSig = prim.quant(coef, quantCoeff, deltaU, q, qbits, add, n * n);
        tab_capture(coef, deltaU, (size_t)(q - levels), n * n);
        if ((flags & TU_FLAG_SIGN_HIDE) && numSig >= 2) numSig = sign_hide(q, deltaU, coef, numSig, intra_scan_type(mode, n, chroma), log2n);
        numSigOut[j] = numSig;
        if (numSig)
        {
            tab_dequant(&prim, q, coef, n * n, per, dqScale, dqShift);
            if (numSig == 1 && q[0] != 0 && !useDST)
            {
                const int shift_2nd = 12 - (X265HIP_DEPTH - 8) - 3;
                const int dc = ((((coef[0] * (64 >> 6) + 1) >> 1) * (64 >> 3)) + (1 << (shift_2nd - 1))) >> shift_2nd;
                cu->blockfill_s[0](resi, n, (int16_t)dc);
            }
            else if (useDST) prim.idst4x4(coef, resi, n);
            else cu->idct(coef, resi, n);
            cu->add_ps[0](rec, reconStride, pred, resi, n, n);
        }
        else
            cu->copy_pp(rec, reconStride, pred, n);
        distOut[j] = (uint64_t)cu->sse_pp(rec, reconStride, fe, fencStride);
    }
    return 0;
}
