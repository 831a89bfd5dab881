/* oracle/x265_oracle_host.c
 *
 * TEST INFRASTRUCTURE - NOT PRODUCT CODE (see x265_oracle.c for the rules).
 *
 * Restatement of the EncoderPrimitives slots that SURVEY.md section 8 keeps on the HOST (rows a9 and a16):
 * the RDOQ cost pre-passes and CABAC bit-cost estimators (reference dct.cpp:757-1069) and the frame-level
 * copies / lowres downscale / SSIM / cutree helpers (reference pixel.cpp:604-701, 864-956).  They are not
 * offloaded (serial, CABAC-state coupled or whole-frame host copies); restating them completes the oracle's
 * table so that every slot the reference's C filler sets is pinned against it (tests/test_oracle_vs_reference.py).
 *
 * CABAC tables: the state-transition table is derived here from the HEVC standard's transIdxLps table
 * (ITU-T H.265 table 9-46); the per-state bit costs are the encoder's own constants and are NOT reproduced:
 * the tests hand them in through x265oracle_set_entropy_bits() (read from the reference build's exported
 * x265_entropyStateBits symbol), so no reference data lives in this file.
 */
#ifndef X265HIP_DEPTH
#error "compile with -DX265HIP_DEPTH=8|10|12"
#endif
#include "x265hip_table.h"

#include <stdlib.h>
#include <string.h>
#include <stdint.h>

typedef x265hip_pixel pixel;

#define DEPTH        X265HIP_DEPTH
#define CAT_(a, b)   a##b
#define CAT(a, b)    CAT_(a, b)
#define EXPORT(name) CAT(CAT(name, _d), X265HIP_DEPTH)

#define CG_SIZE            4        /* common.h:290 MLS_CG_SIZE */
#define CG_COEFFS          16       /* common.h:304 SCAN_SET_SIZE */
#define CG_MAX             64       /* common.h:289 MLS_GRP_NUM */
#define C1FLAGS            8        /* common.h:283 C1FLAG_NUMBER */
#define REMAIN_REDUCTION   3        /* common.h:278 COEF_REMAIN_BIN_REDUCTION */
#define TR_DYNAMIC_RANGE   15       /* common.h:297 */
#define RDOQ_SCALE_BITS    15       /* common.h:296 */
#define LOWRES_COST_MASK   ((1 << 14) - 1)   /* slicetype.h:41 */

/* ------------------------------------------------------------------ CABAC state machine */
static uint32_t stateBits[128];     /* [31:24] next state after coding the bin, [23:0] bit cost (<<15 fixed point) */
static uint8_t nextState[128][2];
static int cabacReady;

/* H.265 table 9-46, transIdxLps */
static const uint8_t kTransIdxLps[64] = {
    0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
    24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63 };

static void init_cabac(void)
{
    if (cabacReady) return;
    /* context byte = pStateIdx << 1 | valMps (contexts.h:116 sbacNext) */
    for (int s = 0; s < 128; s++)
    {
        const int p = s >> 1, mps = s & 1;
        for (int bin = 0; bin < 2; bin++)
        {
            int np, nm = mps;
            if (bin == mps) np = p < 62 ? p + 1 : p;            /* transIdxMps */
            else { np = kTransIdxLps[p]; if (p == 0) nm = 1 - mps; }
            nextState[s][bin] = (uint8_t)(np << 1 | nm);
        }
    }
    /* the terminate-bin pseudo state never moves */
    nextState[126][0] = nextState[126][1] = 126;
    nextState[127][0] = nextState[127][1] = 127;
    cabacReady = 1;
}

void EXPORT(x265oracle_set_entropy_bits)(const uint32_t* bits128)
{
    memcpy(stateBits, bits128, sizeof(stateBits));
}

const uint8_t* EXPORT(x265oracle_next_state_table)(void) { init_cabac(); return &nextState[0][0]; }

static inline uint32_t bin_cost(uint32_t state, uint32_t bin) { return stateBits[state ^ bin] & 0xFFFFFF; }   /* contexts.h:117 */

/* ------------------------------------------------------------------ a9: RDOQ helpers */
/* dct.cpp:757-791 - walk the scan until numSig coefficients were seen; per 16-position group collect the count,
 * the non-zero flag bits (MSB = first in scan) and the sign bits (bit i = sign of the i-th non-zero). */
static int scan_pos_last(const uint16_t* scan, const int16_t* coeff, uint16_t* coeffSign, uint16_t* coeffFlag, uint8_t* coeffNum,
                         int numSig, const uint16_t* scanCG4x4, const int trSize)
{
    (void)scanCG4x4; (void)trSize;
    memset(coeffNum, 0, CG_MAX * sizeof(*coeffNum));
    memset(coeffFlag, 0, CG_MAX * sizeof(*coeffFlag));
    memset(coeffSign, 0, CG_MAX * sizeof(*coeffSign));
    int pos = 0;
    do
    {
        const unsigned cg = (unsigned)pos >> 4;
        const int c = coeff[scan[pos++]];
        const unsigned nz = c != 0;
        numSig -= (int)nz;
        coeffSign[cg] = (uint16_t)(coeffSign[cg] + (uint16_t)(((uint32_t)c >> 31) << coeffNum[cg]));
        coeffFlag[cg] = (uint16_t)((coeffFlag[cg] << 1) + nz);
        coeffNum[cg] = (uint8_t)(coeffNum[cg] + nz);
    }
    while (numSig > 0);
    return pos - 1;
}

/* dct.cpp:794-835 - first / last non-zero scan position inside one 4x4 group and the parity source (sum of the levels
 * between them); an all-zero group returns first = 16 and an unspecified upper part. */
static uint32_t find_pos_first_last(const int16_t* dstCoeff, const intptr_t trSize, const uint16_t scanTbl[16])
{
    int last, first;
    for (last = CG_COEFFS - 1; last >= 0; last--)
        if (dstCoeff[(scanTbl[last] >> 2) * trSize + (scanTbl[last] & 3)]) break;
    for (first = 0; first < CG_COEFFS; first++)
        if (dstCoeff[(scanTbl[first] >> 2) * trSize + (scanTbl[first] & 3)]) break;
    uint32_t sum = 0;
    for (int n = first; n <= last; n++)
        sum += (uint32_t)(int32_t)dstCoeff[(scanTbl[n] >> 2) * trSize + (scanTbl[n] & 3)];
    return (sum << 31) | ((uint32_t)last << 8) | (uint32_t)first;
}

/* dct.cpp:838-894 - significance-flag bit cost of one coefficient group, walking the scan backwards from
 * scanPosSigOff; contexts are updated in place; the absolute levels of the non-zero coefficients are written out
 * in coding order (the buffer pointer is pre-decremented by the "last position already known" slot). */
static uint32_t cost_coeff_nxn(const uint16_t* scan, const int16_t* coeff, intptr_t trSize, uint16_t* absCoeff, const uint8_t* tabSigCtx,
                               uint32_t scanFlagMask, uint8_t* baseCtx, int offset, int scanPosSigOff, int subPosBase)
{
    uint16_t mag[CG_COEFFS];
    uint32_t numNonZero = scanPosSigOff < CG_COEFFS - 1 ? 1 : 0;
    uint32_t sum = 0;
    absCoeff -= numNonZero;
    for (int y = 0; y < CG_SIZE; y++)
        for (int x = 0; x < CG_SIZE; x++)
            mag[y * CG_SIZE + x] = (uint16_t)abs(coeff[y * trSize + x]);
    do
    {
        const uint32_t blkPos = scan[scanPosSigOff];
        const uint32_t sig = scanFlagMask & 1;
        scanFlagMask >>= 1;
        if (scanPosSigOff != 0 || subPosBase == 0 || numNonZero)
        {
            /* the DC position of the whole block always uses context 0 */
            const uint32_t ctxSig = (subPosBase + scanPosSigOff) ? (uint32_t)(tabSigCtx[blkPos] + offset) : 0;
            const uint32_t st = baseCtx[ctxSig];
            const uint32_t packed = stateBits[st ^ sig];
            uint32_t nxt = (packed >> 24) + (st & 1);
            if ((st ^ sig) == 1) nxt = sig;
            baseCtx[ctxSig] = (uint8_t)nxt;
            sum += packed;
        }
        absCoeff[numNonZero] = mag[blkPos];
        numNonZero += sig;
        scanPosSigOff--;
    }
    while (scanPosSigOff >= 0);
    return sum & 0xFFFFFF;
}

static inline int ilog2(uint32_t v) { int r = 0; while (v >>= 1) r++; return r; }

/* dct.cpp:886-931 - Golomb-Rice / exp-Golomb length of the coeff_abs_level_remaining bins, adaptive rice parameter */
static uint32_t cost_coeff_remain(uint16_t* absCoeff, int numNonZero, int idx)
{
    uint32_t rice = 0, sum = 0;
    int baseLevel = 3;
    do
    {
        if (idx >= C1FLAGS) baseLevel = 1;
        int code = absCoeff[idx] - baseLevel;
        if (code >= 0)
        {
            code = (int)((uint32_t)code >> rice) - REMAIN_REDUCTION;
            if (code >= 0) code = 2 * ilog2((uint32_t)code + 1);
            sum += (uint32_t)(REMAIN_REDUCTION + 1 + (int)rice + code);
            if (absCoeff[idx] > (REMAIN_REDUCTION << rice)) rice = (rice + 1) - (rice >> 2);
        }
        baseLevel = 2;
        idx++;
    }
    while (idx < numNonZero);
    return sum;
}

/* dct.cpp:934-987 - greater-than-1 flags (context set walks 1 -> 2 -> 3, drops to 0 after the first level > 1) and the
 * single greater-than-2 flag; returns bits | c1 << 26 | firstC2Idx << 28 */
static uint32_t cost_c1c2_flag(uint16_t* absCoeff, intptr_t numC1Flag, uint8_t* baseCtxMod, intptr_t ctxOffset)
{
    init_cabac();
    uint32_t sum = 0, c1 = 1, firstC2Idx = 8, firstC2Flag = 2, c1Next = 0xFFFFFFFEu;
    int idx = 0;
    do
    {
        const uint32_t gt1 = absCoeff[idx] > 1, gt2 = absCoeff[idx] > 2;
        const uint32_t st = baseCtxMod[c1];
        baseCtxMod[c1] = nextState[st][gt1];
        sum += bin_cost(st, gt1);
        if (gt1) c1Next = 0;
        if (gt1 + firstC2Flag == 3) firstC2Flag = gt2;
        if (gt1 + firstC2Idx == 9) firstC2Idx = (uint32_t)idx;
        c1 = c1Next & 3;
        c1Next >>= 2;
        idx++;
    }
    while (idx < numC1Flag);
    if (!c1)
    {
        baseCtxMod += ctxOffset;
        const uint32_t st = baseCtxMod[0];
        baseCtxMod[0] = nextState[st][firstC2Flag];
        sum += bin_cost(st, firstC2Flag);
    }
    return (sum & 0x00FFFFFF) + (c1 << 26) + (firstC2Idx << 28);
}

/* dct.cpp:988-1062 - distortion of leaving a coefficient group uncoded: coef^2 scaled to the RDOQ fixed point (the
 * reference round-trips through double: exact below 2^53), optionally minus the psy-rd energy term */
static inline int64_t via_double(int64_t v) { return (int64_t)(double)v; }

static void rdoq_uncoded(int16_t* resi, int16_t* fenc, int64_t* costUncoded, int64_t* totalUncoded, int64_t* totalRd,
                         const int64_t* psyScale, uint32_t blkPos, int log2TrSize, int doSquare, int doPsy)
{
    const int transformShift = TR_DYNAMIC_RANGE - DEPTH - log2TrSize;
    const int scaleBits = RDOQ_SCALE_BITS - 2 * transformShift;
    const int psyShift = 2 * transformShift + 1 > 0 ? 2 * transformShift + 1 : 0;
    const uint32_t trSize = 1u << log2TrSize;
    for (int y = 0; y < CG_SIZE; y++, blkPos += trSize)
        for (int x = 0; x < CG_SIZE; x++)
        {
            const int64_t c = resi[blkPos + x];
            if (doSquare) costUncoded[blkPos + x] = via_double((c * c) << scaleBits);
            if (doPsy)
            {
                const int64_t predicted = fenc[blkPos + x] - c;
                costUncoded[blkPos + x] -= via_double((*psyScale * predicted) >> psyShift);
            }
            *totalUncoded += costUncoded[blkPos + x];
            *totalRd += costUncoded[blkPos + x];
        }
}

#define DEF_RDOQ(L2) \
static void nonpsy_##L2(int16_t* r, int64_t* cu, int64_t* tu, int64_t* tr, uint32_t bp) { rdoq_uncoded(r, NULL, cu, tu, tr, NULL, bp, L2, 1, 0); } \
static void psy_##L2(int16_t* r, int16_t* f, int64_t* cu, int64_t* tu, int64_t* tr, int64_t* ps, uint32_t bp) { rdoq_uncoded(r, f, cu, tu, tr, ps, bp, L2, 1, 1); } \
static void psy1_##L2(int16_t* r, int64_t* cu, int64_t* tu, int64_t* tr, uint32_t bp) { rdoq_uncoded(r, NULL, cu, tu, tr, NULL, bp, L2, 1, 0); } \
static void psy2_##L2(int16_t* r, int16_t* f, int64_t* cu, int64_t* tu, int64_t* tr, int64_t* ps, uint32_t bp) { rdoq_uncoded(r, f, cu, tu, tr, ps, bp, L2, 0, 1); }
DEF_RDOQ(2) DEF_RDOQ(3) DEF_RDOQ(4) DEF_RDOQ(5)

/* ------------------------------------------------------------------ a16: frame-level helpers */
/* pixel.cpp:604-629 - half-resolution planes for the lookahead: full-pel, H, V and HV (centre) phases, each sample the
 * rounded average of two rounded vertical averages (matches the asm's pavgb chain, not a plain 4-tap mean) */
static inline int avg2(int a, int b) { return (a + b + 1) >> 1; }
static void frame_init_lowres(const pixel* src0, pixel* dst0, pixel* dsth, pixel* dstv, pixel* dstc,
                              intptr_t srcStride, intptr_t dstStride, int width, int height)
{
    for (int y = 0; y < height; y++)
    {
        const pixel* r0 = src0 + (intptr_t)2 * y * srcStride;
        const pixel* r1 = r0 + srcStride;
        const pixel* r2 = r1 + srcStride;
        for (int x = 0; x < width; x++)
        {
            const int a01 = avg2(r0[2 * x], r1[2 * x]), b01 = avg2(r0[2 * x + 1], r1[2 * x + 1]), c01 = avg2(r0[2 * x + 2], r1[2 * x + 2]);
            const int a12 = avg2(r1[2 * x], r2[2 * x]), b12 = avg2(r1[2 * x + 1], r2[2 * x + 1]), c12 = avg2(r1[2 * x + 2], r2[2 * x + 2]);
            dst0[y * dstStride + x] = (pixel)avg2(a01, b01);
            dsth[y * dstStride + x] = (pixel)avg2(b01, c01);
            dstv[y * dstStride + x] = (pixel)avg2(a12, b12);
            dstc[y * dstStride + x] = (pixel)avg2(b12, c12);
        }
    }
}

/* pixel.cpp:631-657 - raw moments of two horizontally adjacent 4x4 blocks: {sum a, sum b, sum a^2 + b^2, sum ab} */
static void ssim_4x4x2_core(const pixel* pix1, intptr_t stride1, const pixel* pix2, intptr_t stride2, int* sums /* [2][4] */)
{
    for (int z = 0; z < 2; z++)
    {
        uint32_t s1 = 0, s2 = 0, ss = 0, s12 = 0;
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++)
            {
                const int a = pix1[4 * z + x + y * stride1], b = pix2[4 * z + x + y * stride2];
                s1 += a; s2 += b; ss += a * a + b * b; s12 += a * b;
            }
        sums[z * 4 + 0] = (int)s1; sums[z * 4 + 1] = (int)s2; sums[z * 4 + 2] = (int)ss; sums[z * 4 + 3] = (int)s12;
    }
}

/* pixel.cpp:659-686 - SSIM of one 8x8 window from the summed moments; integer arithmetic in the 8-bit build, float in
 * the high-bit-depth build (the 10-bit products overflow int) */
static float ssim_end_1(int s1, int s2, int ss, int s12)
{
    const double pmax = (double)((1 << DEPTH) - 1);
#if DEPTH > 8
    typedef float T;
    const T c1 = (float)(.01 * .01 * pmax * pmax * 64);
    const T c2 = (float)(.03 * .03 * pmax * pmax * 64 * 63);
#else
    typedef int T;
    const T c1 = (int)(.01 * .01 * pmax * pmax * 64 + .5);
    const T c2 = (int)(.03 * .03 * pmax * pmax * 64 * 63 + .5);
#endif
    const T fs1 = (T)s1, fs2 = (T)s2, fss = (T)ss, fs12 = (T)s12;
    const T vars = (T)(fss * 64 - fs1 * fs1 - fs2 * fs2);
    const T covar = (T)(fs12 * 64 - fs1 * fs2);
    return (float)(2 * fs1 * fs2 + c1) * (float)(2 * covar + c2) / ((float)(fs1 * fs1 + fs2 * fs2 + c1) * (float)(vars + c2));
}

/* pixel.cpp:688-701 - sum of the SSIM of `width` overlapping 8x8 windows built from two rows of 4x4 moments */
static float ssim_end_4(int* sum0 /* [5][4] */, int* sum1, int width)
{
    float ssim = 0.0f;
    for (int i = 0; i < width; i++)
    {
        int m[4];
        for (int k = 0; k < 4; k++) m[k] = sum0[i * 4 + k] + sum0[(i + 1) * 4 + k] + sum1[i * 4 + k] + sum1[(i + 1) * 4 + k];
        ssim += ssim_end_1(m[0], m[1], m[2], m[3]);
    }
    return ssim;
}

/* pixel.cpp:864-910 - input-plane conversions */
static void planecopy_cp(const uint8_t* src, intptr_t srcStride, pixel* dst, intptr_t dstStride, int width, int height, int shift)
{
    for (int r = 0; r < height; r++)
        for (int c = 0; c < width; c++) dst[r * dstStride + c] = (pixel)(((pixel)src[r * srcStride + c]) << shift);
}
static void planecopy_sp(const uint16_t* src, intptr_t srcStride, pixel* dst, intptr_t dstStride, int width, int height, int shift, uint16_t mask)
{
    for (int r = 0; r < height; r++)
        for (int c = 0; c < width; c++) dst[r * dstStride + c] = (pixel)((src[r * srcStride + c] >> shift) & mask);
}
static void planecopy_sp_shl(const uint16_t* src, intptr_t srcStride, pixel* dst, intptr_t dstStride, int width, int height, int shift, uint16_t mask)
{
    for (int r = 0; r < height; r++)
        for (int c = 0; c < width; c++) dst[r * dstStride + c] = (pixel)((src[r * srcStride + c] << shift) & mask);
}
static void planecopy_pp_shr(const pixel* src, intptr_t srcStride, pixel* dst, intptr_t dstStride, int width, int height, int shift)
{
    for (int r = 0; r < height; r++)
        for (int c = 0; c < width; c++) dst[r * dstStride + c] = (pixel)(src[r * srcStride + c] >> shift);
}

#if DEPTH > 8
/* pixel.cpp:996-1016 (high bit depth builds only) - clamp the source luma plane in place, return its maximum and sum */
static pixel plane_clip_and_max(pixel* src, intptr_t stride, int width, int height, uint64_t* outsum, const pixel minPix, const pixel maxPix)
{
    pixel maxLevel = 0;
    uint64_t sum = 0;
    for (int r = 0; r < height; r++)
        for (int c = 0; c < width; c++)
        {
            pixel v = src[r * stride + c];
            v = v < minPix ? minPix : (v > maxPix ? maxPix : v);
            src[r * stride + c] = v;
            if (v > maxLevel) maxLevel = v;
            sum += v;
        }
    *outsum = sum;
    return maxLevel;
}
#endif

/* pixel.cpp:912-943 - cutree: how much of a CU's cost propagates to its references (double arithmetic) */
static void propagate_cost(int* dst, const uint16_t* propagateIn, const int32_t* intraCosts, const uint16_t* interCosts,
                           const int32_t* invQscales, const double* fpsFactor, int len)
{
    const double fps = *fpsFactor / 256;
    for (int i = 0; i < len; i++)
    {
        const int intra = intraCosts[i];
        const int inter0 = interCosts[i] & LOWRES_COST_MASK;
        const int inter = intra < inter0 ? intra : inter0;
        const double propagateIntra = intra * invQscales[i];
        const double amount = (double)propagateIn[i] + propagateIntra * fps;
        const double num = (double)(intra - inter);
        dst[i] = (int)(amount * num / (double)intra + 0.5);
    }
}

/* pixel.cpp:945-958 - Q8.8 fixed point <-> double for the cutree offset file */
static void fix8_pack(uint16_t* dst, double* src, int count) { for (int i = 0; i < count; i++) dst[i] = (uint16_t)(int16_t)(src[i] * 256.0); }
static void fix8_unpack(double* dst, uint16_t* src, int count) { for (int i = 0; i < count; i++) dst[i] = (double)(int16_t)src[i] / 256.0; }

/* ------------------------------------------------------------------ table filler (adds to x265oracle_setup_primitives) */
void EXPORT(x265oracle_setup_host_primitives)(x265hip_EncoderPrimitives* p)
{
    init_cabac();
    p->cu[0].nonPsyRdoQuant = nonpsy_2; p->cu[1].nonPsyRdoQuant = nonpsy_3; p->cu[2].nonPsyRdoQuant = nonpsy_4; p->cu[3].nonPsyRdoQuant = nonpsy_5;
    p->cu[0].psyRdoQuant = psy_2; p->cu[1].psyRdoQuant = psy_3; p->cu[2].psyRdoQuant = psy_4; p->cu[3].psyRdoQuant = psy_5;
    p->cu[0].psyRdoQuant_1p = psy1_2; p->cu[1].psyRdoQuant_1p = psy1_3; p->cu[2].psyRdoQuant_1p = psy1_4; p->cu[3].psyRdoQuant_1p = psy1_5;
    p->cu[0].psyRdoQuant_2p = psy2_2; p->cu[1].psyRdoQuant_2p = psy2_3; p->cu[2].psyRdoQuant_2p = psy2_4; p->cu[3].psyRdoQuant_2p = psy2_5;
    p->scanPosLast = scan_pos_last;
    p->findPosFirstLast = find_pos_first_last;
    p->costCoeffNxN = cost_coeff_nxn;
    p->costCoeffRemain = cost_coeff_remain;
    p->costC1C2Flag = cost_c1c2_flag;
    p->frameInitLowres = frame_init_lowres;
    p->frameInitLowerRes = frame_init_lowres;
    p->ssim_4x4x2_core = ssim_4x4x2_core;
    p->ssim_end_4 = ssim_end_4;
    p->planecopy_cp = planecopy_cp;
    p->planecopy_sp = planecopy_sp;
    p->planecopy_sp_shl = planecopy_sp_shl;
    p->planecopy_pp_shr = planecopy_pp_shr;
    p->propagateCost = propagate_cost;
    p->fix8Pack = fix8_pack;
    p->fix8Unpack = fix8_unpack;
#if DEPTH > 8
    p->planeClipAndMax = plane_clip_and_max;
#endif
}
