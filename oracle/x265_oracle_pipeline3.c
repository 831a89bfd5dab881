/* oracle/x265_oracle_pipeline3.c
 *
 * TEST INFRASTRUCTURE - NOT PRODUCT CODE (same rules as x265_oracle.c).
 *
 * Stage: lookahead picture preparation and intra cost estimate (the intra half).
 * Restates, on top of the oracle's primitive table, the call sequences of
 *   Lowres::init          (source/common/lowres.cpp:294-306: frameInitLowres into the four half-resolution planes,
 *                          then extendPicBorder of each, pixel.cpp:1027-1041) and
 *   LookaheadTLD::lowresIntraEstimate (source/encoder/slicetype.cpp:696-772: per 8x8 block - copy_pp, neighbour
 *                          collection from the padded plane, intra_filter, DC / planar / coarse-to-fine angular scan
 *                          with intra_pred[] + satd 8x8, COPY2_IF_LT order, + intraPenalty + lowresPenalty).
 * AQ weighting and the row / frame cost sums that follow in the reference are host bookkeeping and not restated.
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
void EXPORT(x265oracle_setup_host_primitives)(x265hip_EncoderPrimitives* p);

#define LOWRES_CU     8            /* common.h X265_LOWRES_CU_SIZE */
#define LOWRES_CU_L2  3
#define COST_MAX      (1 << 28)    /* motion.h:103 */
#define LOWRES_COST_MASK ((1 << 14) - 1)

static x265hip_EncoderPrimitives prim;
static int ready;
static void init(void)          /* thread-safe: 0 = empty, 1 = being filled, 2 = ready (the seam tests call in from several threads) */
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

static const uint8_t kFilterFlags[35] = {        /* constants.cpp:561 g_intraFilterFlags */
    0x38, 0x00,
    0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30,
    0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30,
    0x38 };

/* pixel.cpp:1027-1041 extendPicBorder: left/right columns of every row, then whole rows above / below */
static void extend_pic_border(pixel* pic, intptr_t stride, int width, int height, int marginX, int marginY)
{
    for (int y = 0; y < height; y++)
    {
        pixel* row = pic + y * stride;
        for (int x = 0; x < marginX; x++) { row[-marginX + x] = row[0]; row[width + x] = row[width - 1]; }
    }
    pixel* top = pic - marginX;
    for (int y = 0; y < marginY; y++) memcpy(top - (y + 1) * stride, top, (size_t)(width + 2 * marginX) * sizeof(pixel));
    pixel* bot = pic + (height - 1) * stride - marginX;
    for (int y = 0; y < marginY; y++) memcpy(bot + (y + 1) * stride, bot, (size_t)(width + 2 * marginX) * sizeof(pixel));
}

/* src: full-resolution padded plane ((0,0) pointer); planes[4]: (0,0) pointers of the four lowres planes, all with
 * `lumaStride` and margins of marginX / marginY pixels; width / lines: lowres size (multiples of 8). */
void EXPORT(x265oracle_lowres_init)(const pixel* src, intptr_t srcStride, pixel* p0, pixel* ph, pixel* pv, pixel* pc,
                                    intptr_t lumaStride, int width, int lines, int marginX, int marginY)
{
    init();
    prim.frameInitLowres(src, p0, ph, pv, pc, srcStride, lumaStride, width, lines);
    extend_pic_border(p0, lumaStride, width, lines, marginX, marginY);
    extend_pic_border(ph, lumaStride, width, lines, marginX, marginY);
    extend_pic_border(pv, lumaStride, width, lines, marginX, marginY);
    extend_pic_border(pc, lumaStride, width, lines, marginX, marginY);
}

/* plane: lowres plane 0 ((0,0) pointer, padded).  Outputs per 8x8 block, raster order: intraCost (int32), intraMode
 * (uint8), lowresCosts (uint16 = min(cost, LOWRES_COST_MASK)). */
void EXPORT(x265oracle_lowres_intra)(const pixel* plane, intptr_t stride, int widthInCU, int heightInCU, int intraPenalty,
                                     int32_t* intraCost, uint8_t* intraMode, uint16_t* lowresCosts, int nthreads)
{
    init();
    const int cuSize = LOWRES_CU, cuSize2 = 2 * LOWRES_CU, sizeIdx = LOWRES_CU_L2 - 2;
    const int lowresPenalty = 4;
    x265hip_pixelcmp_t satd = prim.pu[sizeIdx].satd;          /* pu[1] = LUMA_8x8 (slicetype.cpp:710) */
    const int planar = cuSize >= 8;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
    for (int cuXY = 0; cuXY < widthInCU * heightInCU; cuXY++)
    {
        const int cuX = cuXY % widthInCU, cuY = cuXY / widthInCU;
        pixel prediction[LOWRES_CU * LOWRES_CU], fencIntra[LOWRES_CU * LOWRES_CU];
        pixel neighbours[2][LOWRES_CU * 4 + 1];
        pixel* samples = neighbours[0];
        pixel* filtered = neighbours[1];
        const pixel* pixCur = plane + cuSize * cuX + (intptr_t)cuSize * cuY * stride;
        prim.cu[sizeIdx].copy_pp(fencIntra, cuSize, pixCur, stride);
        pixCur -= stride + 1;
        memcpy(samples, pixCur, (2 * cuSize + 1) * sizeof(pixel));
        for (int i = 1; i <= 2 * cuSize; i++) samples[cuSize2 + i] = pixCur[i * stride];
        prim.cu[sizeIdx].intra_filter(samples, filtered);

        int cost, icost = COST_MAX;
        uint32_t ilowmode = 0;
        prim.cu[sizeIdx].intra_pred[1](prediction, cuSize, samples, 0, cuSize <= 16);
        cost = satd(fencIntra, cuSize, prediction, cuSize);
        if (cost < icost) { icost = cost; ilowmode = 1; }
        prim.cu[sizeIdx].intra_pred[0](prediction, cuSize, neighbours[planar], 0, 0);
        cost = satd(fencIntra, cuSize, prediction, cuSize);
        if (cost < icost) { icost = cost; ilowmode = 0; }

        int acost = COST_MAX;
        uint32_t alowmode = 4;
#define TRY(M) do { const uint32_t mode_ = (M); const int filter = !!(kFilterFlags[mode_] & cuSize); \
        prim.cu[sizeIdx].intra_pred[mode_](prediction, cuSize, neighbours[filter], (int)mode_, cuSize <= 16); \
        cost = satd(fencIntra, cuSize, prediction, cuSize); \
        if (cost < acost) { acost = cost; alowmode = mode_; } } while (0)
        for (uint32_t mode = 5; mode < 35; mode += 5) TRY(mode);
        for (uint32_t dist = 2; dist >= 1; dist--)
        {
            const uint32_t minusmode = alowmode - dist, plusmode = alowmode + dist;
            TRY(minusmode);
            TRY(plusmode);
        }
#undef TRY
        if (acost < icost) { icost = acost; ilowmode = alowmode; }
        icost += intraPenalty + lowresPenalty;
        lowresCosts[cuXY] = (uint16_t)(icost < LOWRES_COST_MASK ? icost : LOWRES_COST_MASK);
        intraCost[cuXY] = icost;
        intraMode[cuXY] = (uint8_t)ilowmode;
    }
}

/* ================================================================ weighted-reference analysis of the lookahead
 * LookaheadTLD::weightCostLuma (slicetype.cpp:807-841): the reference picture's plane 0 weighted by (scale, denom, offset)
 * through primitives.weight_pp - with the 14-bit intermediate of the interpolation filters: round << (14 - depth), shift
 * denom + (14 - depth), offset << (depth - 8) - then the 8x8 SATD against the current picture, block by block, each capped by
 * the block's intra cost.  fenc / ref: sample (0,0) of the lowres planes 0 (the reference weights the whole padded buffer; only
 * the blocks' samples matter, which may reach up to 7 samples into the padding); intraCost: int32 [blocks]. */
uint32_t EXPORT(x265oracle_lowres_weight_cost)(const pixel* fenc, const pixel* ref, intptr_t stride, int width, int lines,
                                               const int32_t* intraCost, int present, int scale, int denom, int inputOffset)
{
    init();
    const int correction = 14 - X265HIP_DEPTH;
    const int offset = inputOffset << (X265HIP_DEPTH - 8), round = denom ? 1 << (denom - 1) : 0;
    uint32_t cost = 0;
    int mb = 0;
    pixel wbuf[8 * 16] __attribute__((aligned(64)));
    for (int y = 0; y < lines; y += 8)
        for (int x = 0; x < width; x += 8, mb++)
        {
            const pixel* src = ref + (intptr_t)y * stride + x;
            intptr_t sstride = stride;
            if (present)
            {
                /* weight_pp works on rows of 16: weight a 16-wide strip, use its left half */
                pixel strip[8 * 16];
                for (int r = 0; r < 8; r++)
                    for (int c = 0; c < 16; c++) strip[r * 16 + c] = c < 8 ? src[r * stride + c] : 0;
                prim.weight_pp(strip, wbuf, 16, 16, 8, scale, round << correction, denom + correction, offset);
                src = wbuf; sstride = 16;
            }
            const int satd = prim.pu[X265HIP_LUMA_8x8].satd(src, sstride, fenc + (intptr_t)y * stride + x, stride);
            cost += (uint32_t)(satd < intraCost[mb] ? satd : intraCost[mb]);
        }
    return cost;
}

/* LookaheadTLD::weightsAnalyse (slicetype.cpp:860-957) for one (current, reference) pair: the float arithmetic exactly as written
 * there (C float: sqrtf, the mean through two divisions, (int)(x + 0.5f)).  wpSsd / wpSum: the pictures' wp_ssd[0] / wp_sum[0]
 * (luma statistics the adaptive-quantisation pass leaves in Lowres, lowres.h:220-221), [0] = current, [1] = reference.
 * out[0] = 1 when a weight is chosen (weightedRef.isWeighted), out[1..3] = scale, denom, offset (input offset, 8-bit domain),
 * out[4] / out[5] = minscore / origscore (0 when the early exits were taken). */
#include <math.h>
void EXPORT(x265oracle_weights_analyse)(const pixel* fenc, const pixel* ref, intptr_t stride, int width, int lines, const int32_t* intraCost,
                                        const uint64_t* wpSsd, const uint64_t* wpSum, int64_t* out)
{
    static const float epsilon = 1.f / 128.f;
    memset(out, 0, 6 * sizeof(int64_t));
    float guessScale, fencMean, refMean;
    if (wpSsd[0] && wpSsd[1]) guessScale = sqrtf((float)wpSsd[0] / wpSsd[1]);
    else guessScale = 1.0f;
    fencMean = (float)wpSum[0] / (lines * width) / (1 << (X265HIP_DEPTH - 8));
    refMean = (float)wpSum[1] / (lines * width) / (1 << (X265HIP_DEPTH - 8));
    if (fabsf(refMean - fencMean) < 0.5f && fabsf(1.f - guessScale) < epsilon) return;
    /* WeightParam::setFromWeightAndOffset(w, 0, 7, true) (slice.h:306-318): halve an even weight while the denominator lasts, clamp to 127 */
    int w = (int)(guessScale * 128 + 0.5f), mindenom = 7;
    while (mindenom > 0 && (w > 127)) { mindenom--; w >>= 1; }
    int minscale = w < 127 ? w : 127, minoff = 0, found = 0;
    unsigned int minscore, origscore;
    /* wp.wtPresent is still 0 here (:866): the first score is the UNWEIGHTED cost, the yardstick of the 0.998 test below */
    origscore = minscore = EXPORT(x265oracle_lowres_weight_cost)(fenc, ref, stride, width, lines, intraCost, 0, minscale, mindenom, 0);
    out[4] = minscore; out[5] = origscore;
    if (!minscore) return;
    int curScale = minscale;
    int curOffset = (int)(fencMean - refMean * curScale / (1 << mindenom) + 0.5f);
    if (curOffset < -128 || curOffset > 127)
    {
        curOffset = curOffset < -128 ? -128 : (curOffset > 127 ? 127 : curOffset);
        curScale = (int)((1 << mindenom) * (fencMean - curOffset) / refMean + 0.5f);
        curScale = curScale < 0 ? 0 : (curScale > 127 ? 127 : curScale);
    }
    const unsigned int s = EXPORT(x265oracle_lowres_weight_cost)(fenc, ref, stride, width, lines, intraCost, 1, curScale, mindenom, curOffset);
    if (s < minscore) { minscore = s; minscale = curScale; minoff = curOffset; found = 1; }
    if (mindenom > 0 && !(minscale & 1))
    {
        int idx = 0;
        if (!minscale) idx = 32;                                     /* CTZ(0): tzcnt's answer */
        else while (!((minscale >> idx) & 1)) idx++;
        const int shift = idx < mindenom ? idx : mindenom;
        mindenom -= shift;
        minscale >>= shift;
    }
    out[4] = minscore;
    if (!found || (minscale == 1 << mindenom && minoff == 0) || (float)minscore / origscore > 0.998f) return;
    out[0] = 1; out[1] = minscale; out[2] = mindenom; out[3] = minoff;
}

/* ================================================================ adaptive quantisation pass of the lookahead
 * LookaheadTLD::calcAdaptiveQuantFrame (slicetype.cpp:439-694) without hevcAq / edge mode / HDR10 / per-block quant offsets:
 * the AC energy of every qgSize x qgSize block (acEnergyCu :256-275: cu[].var of the luma block plus, for 4:2:0, of the two half-size
 * chroma blocks; energy = ssd - (sum^2 >> shift), and every call adds the block's sum / ssd to the picture's wp_sum / wp_ssd), the
 * QP offset per block in double precision (AQ modes 1-3, :508-632) with invQscaleFactor = x265_exp2fix8(offset) (common.cpp:96-103),
 * and the final wp_ssd normalisation (:662-675) when weighted prediction is on.
 * y / cb / cr: sample (0,0) of padded planes (cb = NULL: 4:0:0); blocks run over [0, width) x [0, height) in steps of qgSize and may
 * reach into the padding.  energy: uint32 per block; qpAqOffset: double per block; invQscale: int32 per block. */
static uint32_t aq_block_energy(const pixel* src, intptr_t stride, int n, int shift, uint64_t* wpSum, uint64_t* wpSsd)
{
    uint32_t sum = 0, sqr = 0;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) { const uint32_t v = src[y * stride + x]; sum += v; sqr += v * v; }
    *wpSum += sum; *wpSsd += sqr;
    return sqr - (uint32_t)(((uint64_t)sum * sum) >> shift);
}

static int exp2fix8(double x)
{
    static uint8_t lut[64];
    static int lutReady = 0;
    if (!lutReady) { for (int i = 0; i < 64; i++) lut[i] = (uint8_t)((pow(2.0, i / 64.0) - 1.0) * 256.0 + 0.5); lutReady = 1; }   /* x265_exp2_lut, constants.cpp:552 */
    const int i = (int)(x * (-64.f / 6.f) + 512.5f);
    if (i < 0) return 0;
    if (i > 1023) return 0xffff;
    return (lut[i & 63] + 256) << (i >> 6) >> 8;
}

void EXPORT(x265oracle_aq_frame)(const pixel* y, const pixel* cb, const pixel* cr, intptr_t stride, intptr_t strideC, int width, int height,
                                 int qgSize, int aqMode, double aqStrength, int weightp,
                                 uint32_t* energy, double* qpAqOffset, int32_t* invQscale, uint64_t* wpSum, uint64_t* wpSsd)
{
    const int inc = qgSize == 8 ? 8 : 16, lshift = qgSize == 8 ? 6 : 8, cshift = qgSize == 8 ? 4 : 6;
    const float modeOneConst = qgSize == 8 ? 11.427f : 14.427f, modeTwoConst = qgSize == 8 ? 8.f : 11.f;
    const int bw = (width + inc - 1) / inc, bh = (height + inc - 1) / inc, blockCount = bw * bh;
    for (int i = 0; i < 3; i++) wpSum[i] = wpSsd[i] = 0;
#define ENERGY(BX, BY) (aq_block_energy(y + (BX) + (intptr_t)(BY) * stride, stride, inc, lshift, &wpSum[0], &wpSsd[0]) + \
        (cb ? aq_block_energy(cb + ((BX) >> 1) + (intptr_t)((BY) >> 1) * strideC, strideC, inc >> 1, cshift, &wpSum[1], &wpSsd[1]) + \
              aq_block_energy(cr + ((BX) >> 1) + (intptr_t)((BY) >> 1) * strideC, strideC, inc >> 1, cshift, &wpSum[2], &wpSsd[2]) : 0u))
    if (aqMode == 0 || aqStrength == 0)
    {
        for (int i = 0; i < blockCount; i++) { qpAqOffset[i] = 0; invQscale[i] = 256; }
        if (weightp)
            for (int by = 0, i = 0; by < height; by += inc)
                for (int bx = 0; bx < width; bx += inc, i++) energy[i] = ENERGY(bx, by);
    }
    else
    {
        double avg_adj_pow2 = 0, avg_adj = 0, qp_adj = 0, bias_strength = 0.f, strength = 0.f;
        if (aqMode == 2 || aqMode == 3)
        {
            const double bit_depth_correction = 1.f / (1 << (2 * (X265HIP_DEPTH - 8)));
            for (int by = 0, i = 0; by < height; by += inc)
                for (int bx = 0; bx < width; bx += inc, i++)
                {
                    energy[i] = ENERGY(bx, by);
                    qp_adj = pow(energy[i] * bit_depth_correction + 1, 0.1);
                    qpAqOffset[i] = qp_adj;
                    avg_adj += qp_adj;
                    avg_adj_pow2 += qp_adj * qp_adj;
                }
            avg_adj /= blockCount;
            avg_adj_pow2 /= blockCount;
            strength = aqStrength * avg_adj;
            avg_adj = avg_adj - 0.5f * (avg_adj_pow2 - modeTwoConst) / avg_adj;
            bias_strength = aqStrength;
        }
        else
            strength = aqStrength * 1.0397f;
        for (int by = 0, i = 0; by < height; by += inc)
            for (int bx = 0; bx < width; bx += inc, i++)
            {
                if (aqMode == 3)
                {
                    qp_adj = qpAqOffset[i];
                    qp_adj = strength * (qp_adj - avg_adj) + bias_strength * (1.f - modeTwoConst / (qp_adj * qp_adj));
                }
                else if (aqMode == 2)
                {
                    qp_adj = qpAqOffset[i];
                    qp_adj = strength * (qp_adj - avg_adj);
                }
                else
                {
                    energy[i] = ENERGY(bx, by);
                    qp_adj = strength * (log2((double)(energy[i] > 1 ? energy[i] : 1)) - (modeOneConst + 2 * (X265HIP_DEPTH - 8)));
                }
                qpAqOffset[i] = qp_adj;
                invQscale[i] = exp2fix8(qp_adj);
            }
    }
#undef ENERGY
    if (weightp)
    {
        const int maxCol = ((width + 8) >> 4) << 4, maxRow = ((height + 8) >> 4) << 4;
        const int w[3] = { maxCol, maxCol >> 1, maxCol >> 1 }, h[3] = { maxRow, maxRow >> 1, maxRow >> 1 };
        for (int i = 0; i < 3; i++)
            wpSsd[i] = wpSsd[i] - (wpSum[i] * wpSum[i] + (uint64_t)(w[i] * h[i]) / 2) / (uint64_t)(w[i] * h[i]);
    }
}

/* ================================================================ --hevc-aq: the adaptive-quantisation pass on quadrant variances
 * LookaheadTLD::xPreanalyze + xPreanalyzeQp (slicetype.cpp:293-441), what calcAdaptiveQuantFrame runs instead of the AQ modes when
 * rc.hevcAq is set (:507-511).  For every enabled layer d (aqLayerDepth[ctu size][log2 ctu - log2 qg][d], lowres.h:123-142; partitions of
 * (maxCUSize >> d)^2 samples, clipped at the picture's right / bottom edge): sum and sum of squares of the partition's four quadrants
 * - split at HALF THE CLIPPED size, every quadrant divided by (cw / 2) * (ch / 2) whatever it really holds (:339-389) -, activity =
 * 1 + the smallest quadrant variance, the layer's average activity over ceil(w / P) * ceil(h / P) partitions, then per partition
 * dQpOffset = log2((maxQScale * act + avg) / (act + maxQScale * avg)) * 6 with maxQScale = 2^(qpAdaptationRange / 6).  The deepest
 * enabled layer feeds invQscaleFactor = x265_exp2fix8(dQpOffset), written SEQUENTIALLY in partition order (:432-441), and the same loop
 * gathers the wp_sum / wp_ssd statistics through acEnergyCu.  qpAqOffset / qpCuTreeOffset are not written by this path.
 * layerParts[d] receives the partition count of layer d (0: layer off); activity / qpOffset hold the enabled layers one after the other. */
static const uint8_t kAqLayerDepth[3][4][4] = {
    { { 1, 0, 1, 0 }, { 1, 1, 1, 0 }, { 1, 1, 1, 0 }, { 1, 1, 1, 1 } },      /* ctu 64: qg 64, 32, 16, 8 */
    { { 1, 1, 0, 0 }, { 1, 1, 0, 0 }, { 1, 1, 1, 0 }, { 0, 0, 0, 0 } },      /* ctu 32 */
    { { 1, 0, 0, 0 }, { 1, 1, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 } } };    /* ctu 16 */

void EXPORT(x265oracle_aq_hevc_quadrants)(const pixel* y, intptr_t stride, int width, int height, int part, uint64_t* sums)
{
    /* sums[partition][quadrant][0 = sum, 1 = sum of squares] - the integer half of xPreanalyze, what the device kernel produces */
    for (int py = 0, i = 0; py < height; py += part)
        for (int px = 0; px < width; px += part, i++)
        {
            const int cw = part < width - px ? part : width - px, ch = part < height - py ? part : height - py;
            uint64_t* q = sums + (size_t)i * 8;
            for (int k = 0; k < 8; k++) q[k] = 0;
            for (int by = 0; by < ch; by++)
                for (int bx = 0; bx < cw; bx++)
                {
                    const uint64_t v = y[(intptr_t)(py + by) * stride + px + bx];
                    const int quad = (by >= (ch >> 1)) * 2 + (bx >= (cw >> 1));
                    q[2 * quad] += v; q[2 * quad + 1] += v * v;
                }
        }
}

void EXPORT(x265oracle_aq_hevc_offsets)(int width, int height, int part, double qpAdaptationRange, const uint64_t* sums,
                                        double* activity, double* qpOffset, double* avgActivity)
{
    const int pw = (width + part - 1) / part, ph = (height + part - 1) / part;
    double dSumAct = 0.0;
    for (int py = 0, i = 0; py < height; py += part)
        for (int px = 0; px < width; px += part, i++)
        {
            const int cw = part < width - px ? part : width - px, ch = part < height - py ? part : height - py;
            const uint32_t numPix = (uint32_t)(cw >> 1) * (uint32_t)(ch >> 1);
            double dMinVar = 1.7976931348623158e+308;
            if (numPix)
                for (int k = 0; k < 4; k++)
                {
                    const double dAverage = (double)sums[(size_t)i * 8 + 2 * k] / numPix;
                    const double dVariance = (double)sums[(size_t)i * 8 + 2 * k + 1] / numPix - dAverage * dAverage;
                    dMinVar = dMinVar < dVariance ? dMinVar : dVariance;
                }
            else
                dMinVar = 0.0;
            activity[i] = 1.0 + dMinVar;
            dSumAct += activity[i];
        }
    const double dAvgAct = dSumAct / ((double)pw * ph == 0 ? 1 : (uint32_t)(pw * ph));
    *avgActivity = dAvgAct;
    for (int i = 0; i < pw * ph; i++)
    {
        const double dMaxQScale = pow(2.0, qpAdaptationRange / 6.0);
        const double dNormAct = (dMaxQScale * activity[i] + dAvgAct) / (activity[i] + dMaxQScale * dAvgAct);
        qpOffset[i] = (log2(dNormAct) / log2(2.0)) * 6.0;
    }
}

void EXPORT(x265oracle_aq_hevc_frame)(const pixel* y, const pixel* cb, const pixel* cr, intptr_t stride, intptr_t strideC, int width, int height,
                                      int maxCUSize, int qgSize, double qpAdaptationRange, int weightp,
                                      int32_t* layerParts, double* activity, double* qpOffset, double* avgActivity, int32_t* invQscale,
                                      uint64_t* wpSum, uint64_t* wpSsd)
{
    int lc = 0, lq = 0;
    while ((1 << lc) < maxCUSize) lc++;
    while ((1 << lq) < qgSize) lq++;
    const int ctuIdx = 6 - lc, aqDepth = lc - lq;
    size_t at = 0, deepestAt = 0;
    int deepest = -1;
    for (int i = 0; i < 3; i++) wpSum[i] = wpSsd[i] = 0;
    for (int d = 0; d < 4; d++)
    {
        layerParts[d] = 0; avgActivity[d] = 0;
        if (ctuIdx < 0 || ctuIdx > 2 || aqDepth < 0 || aqDepth > 3 || !kAqLayerDepth[ctuIdx][aqDepth][d]) continue;
        const int part = maxCUSize >> d, n = ((width + part - 1) / part) * ((height + part - 1) / part);
        uint64_t* sums = (uint64_t*)malloc(sizeof(uint64_t) * 8 * (size_t)n);
        EXPORT(x265oracle_aq_hevc_quadrants)(y, stride, width, height, part, sums);
        EXPORT(x265oracle_aq_hevc_offsets)(width, height, part, qpAdaptationRange, sums, activity + at, qpOffset + at, &avgActivity[d]);
        free(sums);
        layerParts[d] = n; deepest = d; deepestAt = at; at += (size_t)n;
    }
    if (deepest < 0) return;
    const int part = maxCUSize >> deepest, lshift = qgSize == 8 ? 6 : 8, cshift = qgSize == 8 ? 4 : 6, inc = qgSize == 8 ? 8 : 16;
    for (int py = 0, i = 0; py < height; py += part)
        for (int px = 0; px < width; px += part, i++)
        {
            invQscale[i] = exp2fix8(qpOffset[deepestAt + i]);
            /* acEnergyCu(curFrame, x, y, csp, qgSize): the block is qgSize-driven (8x8 or 16x16), the position the partition's */
            (void)aq_block_energy(y + px + (intptr_t)py * stride, stride, inc, lshift, &wpSum[0], &wpSsd[0]);
            if (cb)
            {
                (void)aq_block_energy(cb + (px >> 1) + (intptr_t)(py >> 1) * strideC, strideC, inc >> 1, cshift, &wpSum[1], &wpSsd[1]);
                (void)aq_block_energy(cr + (px >> 1) + (intptr_t)(py >> 1) * strideC, strideC, inc >> 1, cshift, &wpSum[2], &wpSsd[2]);
            }
        }
    if (weightp)
    {
        const int maxCol = ((width + 8) >> 4) << 4, maxRow = ((height + 8) >> 4) << 4;
        const int w[3] = { maxCol, maxCol >> 1, maxCol >> 1 }, h[3] = { maxRow, maxRow >> 1, maxRow >> 1 };
        for (int i = 0; i < 3; i++)
            wpSsd[i] = wpSsd[i] - (wpSum[i] * wpSum[i] + (uint64_t)(w[i] * h[i]) / 2) / (uint64_t)(w[i] * h[i]);
    }
}

/* ================================================================ cuTree: one propagation step
 * Lookahead::estimateCUPropagate (slicetype.cpp:2641-2753) with primitives.propagateCost (pixel.cpp:914-940): every 8x8 lowres block
 * of picture b passes on  (propagateIn + intraCost * invQscale * fpsFactor / 256) * (intraCost - min(intraCost, interCost)) / intraCost
 * (double arithmetic, + 0.5, truncated) to the blocks its motion vectors point at in the list-0 / list-1 reference, split bilinearly
 * over the four blocks the displaced block overlaps (weights in 1/32 block units, blocks outside the picture dropped), halved by the
 * bi-prediction weights when both lists are used; the references' propagateCost accumulate with saturation at 65535.
 * propagateIn: uint16 [h][w] = frames[b]->propagateCost, or NULL for a non-referenced picture (zeros); mvs0 / mvs1: int32 [n][2];
 * lowresCosts: uint16 [n] (cost | lists used << 14); refCost0 / refCost1: uint16 [n], updated in place (refCost1 may be NULL for P). */
void EXPORT(x265oracle_cutree_propagate)(int widthInCU, int heightInCU, const uint16_t* propagateIn, const int32_t* intraCost,
                                         const uint16_t* lowresCosts, const int32_t* invQscale, const int32_t* mvs0, const int32_t* mvs1,
                                         double fpsFactor, int bipredWeight, uint16_t* refCost0, uint16_t* refCost1)
{
    const double fps = fpsFactor / 256;
    const int32_t bipredWeights[2] = { bipredWeight, 64 - bipredWeight };
    uint16_t* refCosts[2] = { refCost0, refCost1 };
    const int32_t* mvsL[2] = { mvs0, mvs1 };
#define CLIP_ADD(S, X) (S) = (uint16_t)((int32_t)(S) + (X) < 65535 ? (int32_t)(S) + (X) : 65535)
    for (int blocky = 0; blocky < heightInCU; blocky++)
        for (int blockx = 0; blockx < widthInCU; blockx++)
        {
            const int cuIndex = blocky * widthInCU + blockx;
            const int intra = intraCost[cuIndex];
            const int inter0 = lowresCosts[cuIndex] & LOWRES_COST_MASK, inter = intra < inter0 ? intra : inter0;
            const double propagateIntra = intra * invQscale[cuIndex];
            const double amount = (double)(propagateIn ? propagateIn[cuIndex] : 0) + propagateIntra * fps;
            const double r = amount * (double)(intra - inter) / (double)intra + 0.5;
            const int32_t propagate_amount = (r == r && r < 2147483648.0 && r > -2147483649.0) ? (int32_t)r : INT32_MIN;   /* cvttsd2si's answer for NaN / overflow */
            if (propagate_amount <= 0) continue;
            const int32_t lists_used = lowresCosts[cuIndex] >> 14;
            for (int list = 0; list < 2; list++)
            {
                if (!((lists_used >> list) & 1)) continue;
                int32_t listamount = propagate_amount;
                if (lists_used == 3) listamount = (listamount * bipredWeights[list] + 32) >> 6;
                int32_t x = mvsL[list][2 * cuIndex], y = mvsL[list][2 * cuIndex + 1];
                uint16_t* rc = refCosts[list];
                if (!x && !y) { CLIP_ADD(rc[cuIndex], listamount); continue; }
                const int32_t cux = (x >> 5) + blockx, cuy = (y >> 5) + blocky;
                const int32_t idx0 = cux + cuy * widthInCU;
                x &= 31; y &= 31;
                const int32_t w[4] = { (32 - y) * (32 - x), (32 - y) * x, y * (32 - x), y * x };
                for (int k = 0; k < 4; k++)
                {
                    const int32_t tx = cux + (k & 1), ty = cuy + (k >> 1);
                    if (tx < 0 || ty < 0 || tx >= widthInCU || ty >= heightInCU) continue;      /* :2722-2741: each target checked on its own */
                    CLIP_ADD(rc[idx0 + (k & 1) + (k >> 1) * widthInCU], (listamount * w[k] + 512) >> 10);
                }
            }
        }
#undef CLIP_ADD
}

/* Lookahead::cuTreeFinish (slicetype.cpp:2889-2937, quantisation groups of 16 or more, hevcAq off): the propagated cost of every
 * 8x8 lowres block becomes a QP offset, qpCuTreeOffset = qpAqOffset - strength * (log2(intra + propagate) - log2(intra) + weightDelta)
 * with intra = (intraCost * invQscale + 128) >> 8 and propagate = (propagateCost * fpsFactorQ8 + 128) >> 8; blocks whose scaled intra
 * cost is 0 keep their value.  fpsFactorQ8 = (int)(CLIP_DURATION(averageDuration) / CLIP_DURATION(fpsDenom / fpsNum) * 256);
 * strength = m_cuTreeStrength = 5.0 * (1.0 - qCompress) (:989); weightDelta = 1 - weightedCostDelta[ref0Distance - 1] when that is
 * positive, else 0. */
void EXPORT(x265oracle_cutree_finish)(int n, const int32_t* intraCost, const int32_t* invQscale, const uint16_t* propagateCost,
                                      const double* qpAqOffset, int fpsFactorQ8, double weightDelta, double strength, double* qpCuTreeOffset)
{
    for (int i = 0; i < n; i++)
    {
        const int intracost = (intraCost[i] * invQscale[i] + 128) >> 8;
        if (!intracost) continue;
        const int propagate = (propagateCost[i] * fpsFactorQ8 + 128) >> 8;
        const double log2_ratio = log2((double)(intracost + propagate)) - log2((double)intracost) + weightDelta;
        qpCuTreeOffset[i] = qpAqOffset[i] - strength * log2_ratio;
    }
}

/* cuTreeFinish with --hevc-aq: Lookahead::computeCUTreeQpOffset (slicetype.cpp:2749-2887), the qgSize >= 16 branch, one layer */
void EXPORT(x265oracle_cutree_finish_hevc_aq)(int width, int height, int part, int blocksInRow, const int32_t* intraCost, const int32_t* invQscale,
                                              const uint16_t* propagateCost, int fpsFactorQ8, double weightDelta, double strength,
                                              const double* qpOffset, double* cuTreeOffset)
{
    const unsigned loopIncr = 16, nw = (width + part - 1) / part, nh = (height + part - 1) / part;
    for (unsigned y = 0, i = 0; y < nh; y++)
        for (unsigned x = 0; x < nw; x++, i++)
        {
            const unsigned block_x = x * part, block_y = y * part;
            unsigned blockXY = 0;
            double log2_ratio = 0;
            for (unsigned yy = block_y; yy < block_y + part && yy < (unsigned)height; yy += loopIncr)
                for (unsigned xx = block_x; xx < block_x + part && xx < (unsigned)width; xx += loopIncr)
                {
                    const unsigned idx = ((yy / loopIncr) * blocksInRow) + (xx / loopIncr);
                    int ic = (intraCost[idx] * invQscale[idx] + 128) >> 8;
                    int pc = (propagateCost[idx] * fpsFactorQ8 + 128) >> 8;
                    log2_ratio += (log2((double)(ic + pc)) - log2((double)ic) + weightDelta);
                    blockXY++;
                }
            cuTreeOffset[i] = qpOffset[i] - (strength * log2_ratio) / blockXY;
        }
}

/* the --qg-size 8 branches (slicetype.cpp:2903-2921, 2990-3002): offsets on the full-resolution 8x8 grid, costs on the lowres grid */
void EXPORT(x265oracle_cutree_finish_qg8)(int widthInCU, int heightInCU, const int32_t* intraCost, const int32_t* invQscale8x8, const uint16_t* propagateCost,
                                          const double* qpAqOffset, int fpsFactorQ8, double weightDelta, double strength, double* qpCuTreeOffset)
{
    const int fullRow = 2 * widthInCU;
    for (int cuY = 0; cuY < heightInCU; cuY++)
        for (int cuX = 0; cuX < widthInCU; cuX++)
        {
            const int cuXY = cuX + cuY * widthInCU;
            int intracost = ((intraCost[cuXY]) / 4 * invQscale8x8[cuXY] + 128) >> 8;
            if (intracost)
            {
                int propagate = ((propagateCost[cuXY]) / 4 * fpsFactorQ8 + 128) >> 8;
                double log2_ratio = log2((double)(intracost + propagate)) - log2((double)intracost) + weightDelta;
                qpCuTreeOffset[cuX * 2 + cuY * widthInCU * 4] = qpAqOffset[cuX * 2 + cuY * widthInCU * 4] - strength * (log2_ratio);
                qpCuTreeOffset[cuX * 2 + cuY * widthInCU * 4 + 1] = qpAqOffset[cuX * 2 + cuY * widthInCU * 4 + 1] - strength * (log2_ratio);
                qpCuTreeOffset[cuX * 2 + cuY * widthInCU * 4 + fullRow] = qpAqOffset[cuX * 2 + cuY * widthInCU * 4 + fullRow] - strength * (log2_ratio);
                qpCuTreeOffset[cuX * 2 + cuY * widthInCU * 4 + fullRow + 1] = qpAqOffset[cuX * 2 + cuY * widthInCU * 4 + fullRow + 1] - strength * (log2_ratio);
            }
        }
}

int64_t EXPORT(x265oracle_frame_cost_recalculate_qg8)(int widthInCU, int heightInCU, const uint16_t* lowresCosts, const double* qpCuTreeOffset,
                                                      int32_t* rowSatds)
{
    const int fullRow = 2 * widthInCU;
    int64_t score = 0;
    for (int cuy = heightInCU - 1; cuy >= 0; cuy--)
    {
        rowSatds[cuy] = 0;
        for (int cux = widthInCU - 1; cux >= 0; cux--)
        {
            int cuCost = lowresCosts[cux + cuy * widthInCU] & LOWRES_COST_MASK;
            const double qp_adj = (qpCuTreeOffset[cux * 2 + cuy * widthInCU * 4] + qpCuTreeOffset[cux * 2 + cuy * widthInCU * 4 + 1] +
                                   qpCuTreeOffset[cux * 2 + cuy * widthInCU * 4 + fullRow] + qpCuTreeOffset[cux * 2 + cuy * widthInCU * 4 + fullRow + 1]) / 4;
            cuCost = (cuCost * exp2fix8(qp_adj) + 128) >> 8;
            rowSatds[cuy] += cuCost;
            if ((cuy > 0 && cuy < heightInCU - 1 && cux > 0 && cux < widthInCU - 1) || widthInCU <= 2 || heightInCU <= 2) score += cuCost;
        }
    }
    return score;
}

/* Lookahead::frameCostRecalculate (slicetype.cpp:2941-3011; P pictures, quantisation groups of 16 or more, hevcAq off): the frame
 * cost after cuTree changed the quantisers - every block's lowres cost scaled by x265_exp2fix8(qpCuTreeOffset), summed per row into
 * rowSatds and over the interior blocks (all blocks when the picture is at most two blocks wide or high) into the returned score. */
int64_t EXPORT(x265oracle_frame_cost_recalculate)(int widthInCU, int heightInCU, const uint16_t* lowresCosts, const double* qpCuTreeOffset,
                                                  int32_t* rowSatds)
{
    int64_t score = 0;
    for (int cuy = heightInCU - 1; cuy >= 0; cuy--)
    {
        rowSatds[cuy] = 0;
        for (int cux = widthInCU - 1; cux >= 0; cux--)
        {
            const int cuxy = cux + cuy * widthInCU;
            int cuCost = lowresCosts[cuxy] & LOWRES_COST_MASK;
            cuCost = (cuCost * exp2fix8(qpCuTreeOffset[cuxy]) + 128) >> 8;
            rowSatds[cuy] += cuCost;
            if ((cuy > 0 && cuy < heightInCU - 1 && cux > 0 && cux < widthInCU - 1) || widthInCU <= 2 || heightInCU <= 2) score += cuCost;
        }
    }
    return score;
}
