/* oracle/x265_oracle_pipeline7.c
 *
 * TEST INFRASTRUCTURE - NOT PRODUCT CODE (same rules as x265_oracle.c).
 *
 * Stage: the frame encoder's weighted-prediction analysis, weightAnalyse (source/encoder/weightPrediction.cpp:222-520, called once
 * per P / B slice from FrameEncoder::compressFrame when --weightp / --weightb are on).  Restated on top of the oracle's primitive
 * table, statement by statement, quirks included:
 *   mcLuma      (:59-92)   a motion-compensated copy of the reference's lowres luma from the lookahead's lowres MVs, 8x8 blocks,
 *                          Lowres::lowresMC (common/lowres.h:67-92: a half-sample plane, or pixelavg_pp of two for quarter positions)
 *   mcChroma    (:96-166)  the same for a full-resolution chroma plane, (16 >> shift)-sample blocks, 4-tap filters; the reference
 *                          compares SAMPLE coordinates with the lowres CU counts (:121) and takes the integer part of the eighth-
 *                          sample chroma vector with >> 2 (:134) - both kept
 *   weightCost  (:172-217) weight_pp of the (compensated) reference, then the sum of 8x8 SATDs against the source, luma blocks
 *                          capped by the lookahead's intra cost; uint32 accumulation
 *   weightAnalyse (:222-497) the float guess per plane from Lowres::wp_ssd / wp_sum, the early exits, the scale x offset scan with
 *                          sliceHeaderCost (:49-56), the luma denominator reduction, the 0.998 acceptance, the chroma pairing and the
 *                          denominators carried from list 0 to list 1.
 * 4:2:0 and 4:0:0.  Pinned in situ: oracle/ref_seam.cpp's weightAnalyse seam runs the reference's own function after every served
 * slice and compares the weight tables (tests/test_seam_cpu.py). */
#ifndef X265HIP_DEPTH
#error "compile with -DX265HIP_DEPTH=8|10|12"
#endif
#include "x265hip_table.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef x265hip_pixel pixel;
#define CAT_(a, b)   a##b
#define CAT(a, b)    CAT_(a, b)
#define EXPORT(name) CAT(CAT(name, _d), X265HIP_DEPTH)

void EXPORT(x265oracle_setup_primitives)(x265hip_EncoderPrimitives* p);
void EXPORT(x265oracle_setup_host_primitives)(x265hip_EncoderPrimitives* p);

static x265hip_EncoderPrimitives prim;
static int ready;
static void init(void)
{
    if (__atomic_load_n(&ready, __ATOMIC_ACQUIRE) == 2) return;
    int expected = 0;
    if (__atomic_compare_exchange_n(&ready, &expected, 1, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE))
    {
        EXPORT(x265oracle_setup_primitives)(&prim);
        EXPORT(x265oracle_setup_host_primitives)(&prim);
        __atomic_store_n(&ready, 2, __ATOMIC_RELEASE);
    }
    else
        while (__atomic_load_n(&ready, __ATOMIC_ACQUIRE) != 2) { }
}

#define CSP_I420 1
static inline int clip3i(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

/* one reference list as weightAnalyse sees it (slice.m_refFrameList[list][0]) */
typedef struct x265oracle_wa_list
{
    const pixel* lowres[4];        /* sample (0,0) of the reference's four lowres planes (Lowres::lowresPlane) */
    const pixel* cb; const pixel* cr;   /* sample (0,0) of the reference SOURCE picture's chroma planes, borders extended (:333-343) */
    const int32_t* mvs;            /* fenc.lowresMvs[list][diffPoc] as (x, y) int32 pairs when the lookahead searched this distance
                                    * (diffPoc <= bframes + 1 and mvs[0].x != 0x7FFF, :323-348), else NULL */
    uint64_t wp_ssd[3], wp_sum[3]; /* the reference's Lowres::wp_ssd / wp_sum */
} x265oracle_wa_list;

typedef struct { int present, weight, denom, offset; } WP;

static int bit_size(unsigned v)            /* bitstream.h:94-112: Exp-Golomb length of the code number v */
{
    if (!v) return 1;
    int n = 0;
    while (v >> (n + 1)) n++;
    return 2 * n + 1;
}
static int size_ue(unsigned val) { return bit_size(val + 1); }
static int size_se(int val)
{
    int tmp = 1 - val * 2;
    if (tmp < 0) tmp = val * 2;
    return tmp < 256 ? bit_size((unsigned)tmp) : bit_size((unsigned)tmp >> 8) + 16;
}
/* weightPrediction.cpp:49-56 */
static int slice_header_cost(const WP* w, int lambda, int bChroma)
{
    if (bChroma) lambda *= 4;
    const int denomCost = size_ue((unsigned)w->denom) * (2 - bChroma);
    return lambda * (10 + denomCost + 2 * (size_se(w->weight) + size_se(w->offset)));
}

/* :59-92 */
static void mc_luma(pixel* mcout, const pixel* const plane[4], intptr_t stride, int width, int lines, const int32_t* mvs)
{
    int cu = 0;
    for (int y = 0; y < lines; y += 8)
    {
        intptr_t pixoff = (intptr_t)y * stride;
        const int mvminy = (-y - 8) * 4, mvmaxy = (lines - y - 1 + 8) * 4;
        for (int x = 0; x < width; x += 8, pixoff += 8, cu++)
        {
            const int mvminx = (-x - 8) * 4, mvmaxx = (width - x - 1 + 8) * 4;
            const int mx = clip3i(mvminx, mvmaxx, mvs[2 * cu]), my = clip3i(mvminy, mvmaxy, mvs[2 * cu + 1]);
            if ((mx | my) & 1)                                                   /* lowres.h:75-85 */
            {
                const int hpelA = (my & 2) | ((mx & 2) >> 1);
                const pixel* frefA = plane[hpelA] + pixoff + (mx >> 2) + (intptr_t)(my >> 2) * stride;
                const int qmvx = mx + (mx & 1), qmvy = my + (my & 1);
                const int hpelB = (qmvy & 2) | ((qmvx & 2) >> 1);
                const pixel* frefB = plane[hpelB] + pixoff + (qmvx >> 2) + (intptr_t)(qmvy >> 2) * stride;
                prim.pu[X265HIP_LUMA_8x8].pixelavg_pp[0](mcout + pixoff, stride, frefA, stride, frefB, stride, 32);
            }
            else
            {
                const int hpel = (my & 2) | ((mx & 2) >> 1);
                prim.cu[1].copy_pp(mcout + pixoff, stride, plane[hpel] + pixoff + (mx >> 2) + (intptr_t)(my >> 2) * stride, stride);      /* cu[BLOCK_8x8] */
            }
        }
    }
}

/* :96-166, 4:2:0: 8x8 chroma blocks, chroma[I420].pu[LUMA_16x16] */
static void mc_chroma(pixel* mcout, const pixel* src, intptr_t stride, const int32_t* mvs, int lowresWidthInCU, int lowresHeightInCU, int height, int width)
{
    const int bw = 8, bh = 8;
    for (int y = 0; y < height; y += bh)
    {
        int cu = y * lowresWidthInCU;
        intptr_t pixoff = (intptr_t)y * stride;
        const int mvminy = (-y - 8) * 4, mvmaxy = (height - y - 1 + 8) * 4;
        for (int x = 0; x < width; x += bw, cu++, pixoff += bw)
        {
            if (x < lowresWidthInCU && y < lowresHeightInCU)
            {
                int mx = mvs[2 * cu] * 2, my = mvs[2 * cu + 1] * 2;            /* mv <<= 1 */
                mx >>= 1; my >>= 1;                                             /* >>= hshift / vshift */
                const int mvminx = (-x - 8) * 4, mvmaxx = (width - x - 1 + 8) * 4;
                mx = clip3i(mvminx, mvmaxx, mx); my = clip3i(mvminy, mvmaxy, my);
                const pixel* temp = src + pixoff + (intptr_t)(my >> 2) * stride + (mx >> 2);
                const int xFrac = mx & 7, yFrac = my & 7;
                if (!(yFrac | xFrac))
                    prim.chroma[CSP_I420].pu[X265HIP_LUMA_16x16].copy_pp(mcout + pixoff, stride, temp, stride);
                else if (!yFrac)
                    prim.chroma[CSP_I420].pu[X265HIP_LUMA_16x16].filter_hpp(temp, stride, mcout + pixoff, stride, xFrac);
                else if (!xFrac)
                    prim.chroma[CSP_I420].pu[X265HIP_LUMA_16x16].filter_vpp(temp, stride, mcout + pixoff, stride, yFrac);
                else
                {
                    int16_t immed[16 * (16 + 4 - 1)] __attribute__((aligned(64)));
                    prim.chroma[CSP_I420].pu[X265HIP_LUMA_16x16].filter_hps(temp, stride, immed, bw, xFrac, 1);
                    prim.chroma[CSP_I420].pu[X265HIP_LUMA_16x16].filter_vsp(immed + ((4 >> 1) - 1) * bw, bw, mcout + pixoff, stride, yFrac);
                }
            }
            else
                prim.chroma[CSP_I420].pu[X265HIP_LUMA_16x16].copy_pp(mcout + pixoff, stride, src + pixoff, stride);
        }
    }
}

/* :172-217 (4:2:0 / luma); w = NULL: the unweighted cost */
static uint32_t weight_cost(const pixel* fenc, const pixel* ref, pixel* weightTemp, intptr_t stride, const int32_t* intraCost, int width, int height, const WP* w, int bLuma)
{
    if (w)
    {
        const int offset = w->offset << (X265HIP_DEPTH - 8), denom = w->denom, round = denom ? 1 << (denom - 1) : 0, correction = 14 - X265HIP_DEPTH;
        const int pwidth = ((width + 31) >> 5) << 5;
        prim.weight_pp(ref, weightTemp, stride, pwidth, height, w->weight, round << correction, denom + correction, offset);
        ref = weightTemp;
    }
    uint32_t cost = 0;
    const pixel* f = fenc; const pixel* r = ref;
    int cu = 0;
    for (int y = 0; y < height; y += 8, r += 8 * stride, f += 8 * stride)
        for (int x = 0; x < width; x += 8, cu++)
        {
            const int cmp = prim.pu[X265HIP_LUMA_8x8].satd(r + x, stride, f + x, stride);
            cost += bLuma ? (uint32_t)(cmp < intraCost[cu] ? cmp : intraCost[cu]) : (uint32_t)cmp;
        }
    return cost;
}

/* weightAnalyse.  fencLowres: sample (0,0) of the current picture's lowres plane 0 (stride lowresStride, lowresWidth x lowresLines,
 * multiples of 8); fencCb / fencCr: sample (0,0) of its source chroma planes (NULL: 4:0:0), strideC; picWidth / picHeight: the source
 * picture (PicYuv::m_picWidth / m_picHeight); intraCost: Lowres::intraCost; fencSsd / fencSum: its wp_ssd / wp_sum; nlists 1 (P) or 2 (B).
 * scratch: 2 * max(lowresStride * lowresLines, strideC * picHeight / 2) samples.  out: int32 [2][3][4] = { wtPresent, inputWeight,
 * log2WeightDenom, inputOffset } of reference 0 per (list, plane); denoms: int32 [2][2] = lumaDenom, chromaDenom after each list (what
 * the other references of the list are reset to, :468-474). */
void EXPORT(x265oracle_weight_analyse)(const pixel* fencLowres, intptr_t lowresStride, int lowresWidth, int lowresLines,
                                       const pixel* fencCb, const pixel* fencCr, intptr_t strideC, int picWidth, int picHeight,
                                       const int32_t* intraCost, const uint64_t* fencSsd, const uint64_t* fencSum,
                                       int nlists, const x265oracle_wa_list* lists, pixel* scratch, size_t scratchHalf, int32_t* out, int32_t* denoms)
{
    init();
    static const int lambdaTab[3] = { 1, 16, 256 };                 /* (int)x265_lambda_tab[X265_LOOKAHEAD_QP = 12 + 6 * (depth - 8)] */
    const int lambda = lambdaTab[(X265HIP_DEPTH - 8) / 2];
    const float epsilon = 1.f / 128.f;
    const int nplanes = fencCb ? 3 : 1;
    const int lowresWidthInCU = lowresWidth >> 3, lowresHeightInCU = lowresLines >> 3;
    pixel* mcbuf = scratch; pixel* weightTemp = scratch + scratchHalf;
    int chromaDenom = 7, lumaDenom = 7, denom;
    int numpixels[3];
    const int w16 = ((picWidth + 15) >> 4) << 4, h16 = ((picHeight + 15) >> 4) << 4;
    numpixels[0] = w16 * h16;
    numpixels[1] = numpixels[2] = numpixels[0] >> 2;
    memset(out, 0, 2 * 3 * 4 * sizeof(int32_t));
    memset(denoms, 0, 4 * sizeof(int32_t));

    for (int list = 0; list < nlists; list++)
    {
        WP weights[3];
        const x265oracle_wa_list* L = &lists[list];
        float guessScale[3], fencMean[3], refMean[3];
        for (int plane = 0; plane < nplanes; plane++)
        {
            weights[plane].present = 0; weights[plane].weight = 1; weights[plane].denom = 0; weights[plane].offset = 0;
            const uint64_t fencVar = fencSsd[plane] + !L->wp_ssd[plane];
            const uint64_t refVar = L->wp_ssd[plane] + !L->wp_ssd[plane];
            guessScale[plane] = sqrtf((float)fencVar / refVar);
            fencMean[plane] = (float)fencSum[plane] / (numpixels[plane]) / (1 << (X265HIP_DEPTH - 8));
            refMean[plane] = (float)L->wp_sum[plane] / (numpixels[plane]) / (1 << (X265HIP_DEPTH - 8));
        }
        if (nplanes == 1) { guessScale[1] = guessScale[2] = 0; }      /* uninitialised in the reference for 4:0:0; only chromaDenom (unused there) depends on it */
        while (!list && chromaDenom > 0)
        {
            const float thresh = 127.f / (1 << chromaDenom);
            if (guessScale[1] < thresh && guessScale[2] < thresh) break;
            chromaDenom--;
        }
        for (int p = 1; p < 3; p++) { weights[p].present = 0; weights[p].weight = 1 << chromaDenom; weights[p].denom = chromaDenom; weights[p].offset = 0; }

        const int32_t* mvs = NULL;
        for (int plane = 0; plane < nplanes; plane++)
        {
            denom = plane ? chromaDenom : lumaDenom;
            if (plane && !weights[0].present) break;
            if (fabsf(refMean[plane] - fencMean[plane]) < 0.5f && fabsf(1.f - guessScale[plane]) < epsilon)
            {
                weights[plane].present = 0; weights[plane].weight = 1 << denom; weights[plane].denom = denom; weights[plane].offset = 0;
                continue;
            }
            if (plane)
            {
                const int scale = clip3i(0, 255, (int)(guessScale[plane] * (1 << denom) + 0.5f));
                if (scale > 127) continue;
                weights[plane].weight = scale;
            }
            else
            {
                /* WeightParam::setFromWeightAndOffset(w, 0, denom, !list), slice.h:304-316 */
                weights[0].offset = 0; weights[0].denom = denom; weights[0].weight = (int)(guessScale[0] * (1 << denom) + 0.5f);
                while (!list && weights[0].denom > 0 && weights[0].weight > 127) { weights[0].denom--; weights[0].weight >>= 1; }
                if (weights[0].weight > 127) weights[0].weight = 127;
            }
            int mindenom = weights[plane].denom, minscale = weights[plane].weight, minoff = 0;
            if (!plane) mvs = L->mvs;

            const pixel* orig; const pixel* fref; intptr_t stride; int width, height;
            if (plane == 0)
            {
                orig = fencLowres; stride = lowresStride; width = lowresWidth; height = lowresLines;
                fref = L->lowres[0];
                if (mvs) { mc_luma(mcbuf, L->lowres, stride, width, height, mvs); fref = mcbuf; }
            }
            else
            {
                orig = plane == 1 ? fencCb : fencCr; stride = strideC;
                fref = plane == 1 ? L->cb : L->cr;
                width = ((picWidth >> 4) << 4) >> 1; height = ((picHeight >> 4) << 4) >> 1;
                if (mvs) { mc_chroma(mcbuf, fref, stride, mvs, lowresWidthInCU, lowresHeightInCU, height, width); fref = mcbuf; }
            }
            const uint32_t origscore = weight_cost(orig, fref, weightTemp, stride, intraCost, width, height, NULL, !plane);
            if (!origscore)
            {
                weights[plane].present = 0; weights[plane].weight = 1 << denom; weights[plane].denom = denom; weights[plane].offset = 0;
                continue;
            }
            uint32_t minscore = origscore;
            int bFound = 0;
            const int scaleDist = 4, offsetDist = 2;
            const int startScale = clip3i(0, 127, minscale - scaleDist), endScale = clip3i(0, 127, minscale + scaleDist);
            for (int scale = startScale; scale <= endScale; scale++)
            {
                const int deltaWeight = scale - (1 << mindenom);
                if (deltaWeight > 127 || deltaWeight <= -128) continue;
                int curScale = scale;
                int curOffset = (int)(fencMean[plane] - refMean[plane] * curScale / (1 << mindenom) + 0.5f);
                if (curOffset < -128 || curOffset > 127)
                {
                    curOffset = clip3i(-128, 127, curOffset);
                    curScale = (int)((1 << mindenom) * (fencMean[plane] - curOffset) / refMean[plane] + 0.5f);
                    curScale = clip3i(0, 127, curScale);
                }
                const int startOffset = clip3i(-128, 127, curOffset - offsetDist), endOffset = clip3i(-128, 127, curOffset + offsetDist);
                for (int off = startOffset; off <= endOffset; off++)
                {
                    WP wsp = { 1, curScale, mindenom, off };
                    const uint32_t s = weight_cost(orig, fref, weightTemp, stride, intraCost, width, height, &wsp, !plane) + (uint32_t)slice_header_cost(&wsp, lambda, !!plane);
                    if (s < minscore) { minscore = s; minscale = curScale; minoff = off; bFound = 1; }
                    if (minoff == startOffset && off != startOffset) break;
                }
            }
            if (!(plane || list))
            {
                if (mindenom > 0 && !(minscale & 1))
                {
                    int idx = 0;
                    if (!minscale) idx = 32;                                 /* CTZ(0) */
                    else while (!((minscale >> idx) & 1)) idx++;
                    const int shift = idx < mindenom ? idx : mindenom;
                    mindenom -= shift;
                    minscale >>= shift;
                }
            }
            if (!bFound || (minscale == (1 << mindenom) && minoff == 0) || (float)minscore / origscore > 0.998f)
            { weights[plane].present = 0; weights[plane].weight = 1 << denom; weights[plane].denom = denom; weights[plane].offset = 0; }
            else
            { weights[plane].present = 1; weights[plane].weight = minscale; weights[plane].denom = mindenom; weights[plane].offset = minoff; }
        }
        if (weights[0].present && nplanes == 3)
        {
            if (weights[1].present != weights[2].present)
            {
                if (weights[1].present) weights[2] = weights[1];
                else weights[1] = weights[2];
            }
        }
        lumaDenom = weights[0].denom;
        chromaDenom = weights[1].denom;
        for (int plane = 0; plane < 3; plane++)
        {
            int32_t* o = out + (list * 3 + plane) * 4;
            o[0] = weights[plane].present; o[1] = weights[plane].weight; o[2] = weights[plane].denom; o[3] = weights[plane].offset;
        }
        denoms[list * 2] = lumaDenom; denoms[list * 2 + 1] = chromaDenom;
    }
}
