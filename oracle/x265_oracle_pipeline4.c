/* oracle/x265_oracle_pipeline4.c
 *
 * TEST INFRASTRUCTURE - NOT PRODUCT CODE (same rules as x265_oracle.c).
 *
 * Stage: in-loop deblocking of the luma reconstruction (deblocking half).
 * Restates, on top of the oracle's primitive table,
 *   Deblock::getBoundaryStrength for P pictures with one reference (source/common/deblock.cpp:191-215: Bs 1 when either side
 *     has coded luma coefficients on a transform edge or the motion vectors differ by >= 4 quarter-pels in x or y, else 0;
 *     picture borders are not filtered, :46-70) for a picture cut into square inter blocks of one size, and
 *   Deblock::edgeFilterLuma (:317-415): per 4-sample edge unit beta / tc from the average QP (tables :499-509), the dE /
 *     strong-filter decisions (calcDP / calcDQ / useStrongFiltering :249-265), primitives.pelFilterLumaStrong
 *     (loopfilter.cpp:140-159) or the normal filter pelFilterLuma (:278-315);
 *   all vertical edges of the picture, then all horizontal edges (deblockCTU order, framefilter.cpp; edges of one direction
 *   do not interact: they lie 8 samples apart and a filter changes at most 3 samples per side).
 */
#ifndef X265HIP_DEPTH
#error "compile with -DX265HIP_DEPTH=8|10|12"
#endif
#include "x265hip_table.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef x265hip_pixel pixel;
#define DEPTH        X265HIP_DEPTH
#define CAT_(a, b)   a##b
#define CAT(a, b)    CAT_(a, b)
#define EXPORT(name) CAT(CAT(name, _d), X265HIP_DEPTH)

void EXPORT(x265oracle_setup_primitives)(x265hip_EncoderPrimitives* p);
void EXPORT(x265oracle_prims_once)(x265hip_EncoderPrimitives* p, int* state);

static const uint8_t kTc[54] = {            /* H.265 table 8-12 (deblock.cpp:499-503) */
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2,
    2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24 };
static const uint8_t kBeta[52] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17,
    18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64 };

static int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static pixel clip_px(int v) { return (pixel)clip3(0, (1 << DEPTH) - 1, v); }
static int iabs(int v) { return v < 0 ? -v : v; }

/* Bs maps of a picture made of n x n inter blocks (n = 8 << level), z-order inside each 64x64 CTU like the other stages:
 * mv = int32 [ctu*85][2] {cost, qx | qy << 16}, numSig = uint32 [ctu][npu].
 * bsVer: uint8 [height/4][width/8] (unit r of the vertical edge at x = 8 * ex), bsHor: uint8 [height/8][width/4]. */
void EXPORT(x265oracle_deblock_bs)(int width, int height, int level, const int32_t* mv, const uint32_t* numSig, const uint8_t* intra,
                                   uint8_t* bsVer, uint8_t* bsHor);
void EXPORT(x265oracle_deblock_bs_inter)(int width, int height, int level, const int32_t* mv, const uint32_t* numSig,
                                         uint8_t* bsVer, uint8_t* bsHor)
{
    EXPORT(x265oracle_deblock_bs)(width, height, level, mv, numSig, NULL, bsVer, bsHor);
}

/* intra: optional uint8 [ctu][npu], non-zero = the block is an intra CU: an edge with an intra block on either side gets Bs 2
 * (deblock.cpp:198-199). */
void EXPORT(x265oracle_deblock_bs)(int width, int height, int level, const int32_t* mv, const uint32_t* numSig, const uint8_t* intra,
                                   uint8_t* bsVer, uint8_t* bsHor)
{
    const int n = 8 << level, npu = (64 / n) * (64 / n), ctusW = width / 64;
    const int lbase = level == 0 ? 0 : (level == 1 ? 64 : (level == 2 ? 80 : 84));
    memset(bsVer, 0, (size_t)(height / 4) * (width / 8));
    memset(bsHor, 0, (size_t)(height / 8) * (width / 4));
#define BLK(X, Y, MVX, MVY, CBF) do { \
        const int ctu_ = ((Y) / 64) * ctusW + (X) / 64, bx_ = ((X) & 63) / n, by_ = ((Y) & 63) / n; \
        int z_ = 0; for (int b_ = 0; b_ < 3; b_++) z_ |= (((bx_ >> b_) & 1) << (2 * b_)) | (((by_ >> b_) & 1) << (2 * b_ + 1)); \
        const int32_t pk_ = mv[((size_t)ctu_ * 85 + lbase + z_) * 2 + 1]; \
        MVX = (int16_t)(pk_ & 0xffff); MVY = (int16_t)(pk_ >> 16); CBF = (numSig[(size_t)ctu_ * npu + z_] != 0) | ((intra && intra[(size_t)ctu_ * npu + z_]) ? 2 : 0); } while (0)
    for (int y = 0; y < height; y += 4)
        for (int x = n; x < width; x += n)            /* vertical block edges; x = 0 is the picture border */
        {
            int px, py, pc, qx, qy, qc;
            BLK(x - 1, y, px, py, pc); BLK(x, y, qx, qy, qc);
            bsVer[(size_t)(y / 4) * (width / 8) + x / 8] = ((pc | qc) & 2) ? 2 : ((pc || qc) ? 1 : ((iabs(qx - px) >= 4 || iabs(qy - py) >= 4) ? 1 : 0));
        }
    for (int y = n; y < height; y += n)
        for (int x = 0; x < width; x += 4)
        {
            int px, py, pc, qx, qy, qc;
            BLK(x, y - 1, px, py, pc); BLK(x, y, qx, qy, qc);
            bsHor[(size_t)(y / 8) * (width / 4) + x / 4] = ((pc | qc) & 2) ? 2 : ((pc || qc) ? 1 : ((iabs(qx - px) >= 4 || iabs(qy - py) >= 4) ? 1 : 0));
        }
#undef BLK
}

static void filter_unit(const x265hip_EncoderPrimitives* prim, pixel* src, intptr_t srcStep, intptr_t offset, int dir, int bs, int qp,
                        int betaOffset, int tcOffset)
{
    const int shift = DEPTH - 8;
    const int beta = kBeta[clip3(0, 51, qp + betaOffset)] << shift;
#define DP(S) iabs((int)(S)[-offset * 3] - 2 * (int)(S)[-offset * 2] + (int)(S)[-offset])
#define DQ(S) iabs((int)(S)[0] - 2 * (int)(S)[offset] + (int)(S)[offset * 2])
    const int dp0 = DP(src), dq0 = DQ(src), dp3 = DP(src + srcStep * 3), dq3 = DQ(src + srcStep * 3);
    const int d0 = dp0 + dq0, d3 = dp3 + dq3, d = d0 + d3;
    if (d >= beta) return;
    const int tc = kTc[clip3(0, 53, qp + 2 * (bs - 1) + tcOffset)] << shift;
#define STRONG(S) (iabs((int)(S)[-offset * 4] - (int)(S)[-offset]) + iabs((int)(S)[offset * 3] - (int)(S)[0]) < (beta >> 3) && \
                   iabs((int)(S)[-offset] - (int)(S)[0]) < ((tc * 5 + 1) >> 1))
    const int sw = 2 * d0 < (beta >> 2) && 2 * d3 < (beta >> 2) && STRONG(src) && STRONG(src + srcStep * 3);
    if (sw)
    {
        prim->pelFilterLumaStrong[dir](src, srcStep, offset, 2 * tc, 2 * tc);
        return;
    }
    const int sideThreshold = (beta + (beta >> 1)) >> 3;
    const int maskP1 = (dp0 + dp3) < sideThreshold, maskQ1 = (dq0 + dq3) < sideThreshold;
    const int thrCut = tc * 10, tc2 = tc >> 1;
    for (int i = 0; i < 4; i++, src += srcStep)
    {
        const int m4 = src[0], m3 = src[-offset], m5 = src[offset], m2 = src[-offset * 2];
        int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
        if (iabs(delta) < thrCut)
        {
            delta = clip3(-tc, tc, delta);
            src[-offset] = clip_px(m3 + delta);
            src[0] = clip_px(m4 - delta);
            if (maskP1)
            {
                const int m1 = src[-offset * 3];
                src[-offset * 2] = clip_px(m2 + clip3(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1)));
            }
            if (maskQ1)
            {
                const int m6 = src[offset * 2];
                src[offset] = clip_px(m5 + clip3(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1)));
            }
        }
    }
#undef DP
#undef DQ
#undef STRONG
}

/* rec: pixel (0,0) of the reconstruction (filtered in place).  qpMap: optional int8 [height/8][width/8] (QP of every 8x8 block;
 * the unit's QP is the rounded mean of its two sides), else the uniform `qp`. */
void EXPORT(x265oracle_deblock_luma)(pixel* rec, intptr_t stride, int width, int height, const uint8_t* bsVer, const uint8_t* bsHor,
                                     int qp, const int8_t* qpMap, int betaOffsetDiv2, int tcOffsetDiv2)
{
    static x265hip_EncoderPrimitives prim;
    static int ready = 0;
    EXPORT(x265oracle_prims_once)(&prim, &ready);
    const int bo = betaOffsetDiv2 * 2, to = tcOffsetDiv2 * 2, w8 = width / 8;
    for (int ex = 1; ex < width / 8; ex++)                 /* EDGE_VER: offset 1, srcStep stride */
        for (int u = 0; u < height / 4; u++)
        {
            const int bs = bsVer[(size_t)u * (width / 8) + ex];
            if (!bs) continue;
            const int by = (u * 4) / 8;
            const int q = qpMap ? (qpMap[by * w8 + ex - 1] + qpMap[by * w8 + ex] + 1) >> 1 : qp;
            filter_unit(&prim, rec + (intptr_t)(u * 4) * stride + ex * 8, stride, 1, 0, bs, q, bo, to);
        }
    for (int ey = 1; ey < height / 8; ey++)                /* EDGE_HOR: offset stride, srcStep 1 */
        for (int u = 0; u < width / 4; u++)
        {
            const int bs = bsHor[(size_t)ey * (width / 4) + u];
            if (!bs) continue;
            const int bx = (u * 4) / 8;
            const int q = qpMap ? (qpMap[(ey - 1) * w8 + bx] + qpMap[ey * w8 + bx] + 1) >> 1 : qp;
            filter_unit(&prim, rec + (intptr_t)(ey * 8) * stride + u * 4, 1, stride, 1, bs, q, bo, to);
        }
}

/* Deblock::getBoundaryStrength in full (deblock.cpp:191-247): pictures with several references and B pictures.  Per block and list
 * a reference PICTURE id (int8, -1 = list unused; equal ids = the same picture, whichever list they come from) and an mv record
 * array per list; intra as above.  P logic (isInterP on both sides): Bs 1 when the list-0 pictures differ or the mvs differ by >= 4;
 * B logic: the four-way comparison of (ref0, ref1) x (mv0, mv1) of :231-246. */
static int bs_motion(int sliceB, int rp0, int rp1, int rq0, int rq1, const int* mp0, const int* mp1, const int* mq0, const int* mq1)
{
    static const int zero[2] = { 0, 0 };
    if (rp0 < 0) mp0 = zero;
    if (rq0 < 0) mq0 = zero;
#define FAR(A, B) (iabs((A)[0] - (B)[0]) >= 4 || iabs((A)[1] - (B)[1]) >= 4)
    if (!sliceB) return (rp0 != rq0 || FAR(mq0, mp0)) ? 1 : 0;
    if (rp1 < 0) mp1 = zero;
    if (rq1 < 0) mq1 = zero;
    if ((rp0 == rq0 && rp1 == rq1) || (rp0 == rq1 && rp1 == rq0))
    {
        if (rp0 != rp1)
        {
            if (rp0 == rq0) return (FAR(mq0, mp0) || FAR(mq1, mp1)) ? 1 : 0;
            return (FAR(mq1, mp0) || FAR(mq0, mp1)) ? 1 : 0;
        }
        return ((FAR(mq0, mp0) || FAR(mq1, mp1)) && (FAR(mq1, mp0) || FAR(mq0, mp1))) ? 1 : 0;
    }
    return 1;
#undef FAR
}

void EXPORT(x265oracle_deblock_bs_b)(int width, int height, int level, int sliceB, const int32_t* mv0, const int32_t* mv1,
                                     const int8_t* ref0, const int8_t* ref1, const uint32_t* numSig, const uint8_t* intra,
                                     uint8_t* bsVer, uint8_t* bsHor)
{
    const int n = 8 << level, npu = (64 / n) * (64 / n), ctusW = width / 64;
    const int lbase = level == 0 ? 0 : (level == 1 ? 64 : (level == 2 ? 80 : 84));
    memset(bsVer, 0, (size_t)(height / 4) * (width / 8));
    memset(bsHor, 0, (size_t)(height / 8) * (width / 4));
    for (int dir = 0; dir < 2; dir++)
        for (int y = dir ? n : 0; y < height; y += dir ? n : 4)
            for (int x = dir ? 0 : n; x < width; x += dir ? 4 : n)
            {
                int blk[2], m0[2][2], m1[2][2], r0[2], r1[2], cbf[2], in[2];
                for (int s2 = 0; s2 < 2; s2++)                              /* s2 = 0: P side, 1: Q side */
                {
                    const int xx = dir ? x : x - 1 + s2, yy = dir ? y - 1 + s2 : y;
                    const int ctu = (yy / 64) * ctusW + xx / 64, bx = (xx & 63) / n, by = (yy & 63) / n;
                    int z = 0;
                    for (int b = 0; b < 3; b++) z |= (((bx >> b) & 1) << (2 * b)) | (((by >> b) & 1) << (2 * b + 1));
                    blk[s2] = ctu * npu + z;
                    const int32_t p0 = mv0[((size_t)ctu * 85 + lbase + z) * 2 + 1], p1 = mv1 ? mv1[((size_t)ctu * 85 + lbase + z) * 2 + 1] : 0;
                    m0[s2][0] = (int16_t)(p0 & 0xffff); m0[s2][1] = (int16_t)(p0 >> 16);
                    m1[s2][0] = (int16_t)(p1 & 0xffff); m1[s2][1] = (int16_t)(p1 >> 16);
                    r0[s2] = ref0 ? ref0[blk[s2]] : 0;
                    r1[s2] = ref1 ? ref1[blk[s2]] : -1;
                    cbf[s2] = numSig[blk[s2]] != 0;
                    in[s2] = intra && intra[blk[s2]];
                }
                int bs;
                if (in[0] || in[1]) bs = 2;
                else if (cbf[0] || cbf[1]) bs = 1;
                else bs = bs_motion(sliceB, r0[0], r1[0], r0[1], r1[1], m0[0], m1[0], m0[1], m1[1]);
                if (dir) bsHor[(size_t)(y / 8) * (width / 4) + x / 4] = (uint8_t)bs;
                else bsVer[(size_t)(y / 4) * (width / 8) + x / 8] = (uint8_t)bs;
            }
}

/* Deblock::edgeFilterChroma (deblock.cpp:417-497) for the two chroma planes of a 4:2:0 picture: only edges with Bs 2 (an intra
 * block on either side) on the 8-sample chroma grid (luma positions that are multiples of 16, deblock.cpp:104-113) are filtered,
 * 4 chroma lines per unit with the Bs of the luma unit they start at; tc from the mean QP + the plane's PPS offset through the
 * chroma QP mapping table (constants.cpp:346-350) at index qp + DEFAULT_INTRA_TC_OFFSET (2) + tcOffset; the filter is
 * primitives.pelFilterChroma (loopfilter.cpp:160-180).  cb / cr: sample (0,0) of the planes, strideC their stride; width / height:
 * LUMA size; bs maps and qpMap as for x265oracle_deblock_luma.  Vertical edges first, then horizontal. */
static const uint8_t kChromaScale[70] = {
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 33, 33, 34, 34, 35,
    35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51 };

void EXPORT(x265oracle_deblock_chroma)(pixel* cb, pixel* cr, intptr_t strideC, int width, int height, const uint8_t* bsVer, const uint8_t* bsHor,
                                       int qp, const int8_t* qpMap, int cbQpOffset, int crQpOffset, int tcOffsetDiv2)
{
    static x265hip_EncoderPrimitives prim;
    static int ready = 0;
    EXPORT(x265oracle_prims_once)(&prim, &ready);
    pixel* planes[2] = { cb, cr };
    const int offs[2] = { cbQpOffset, crQpOffset };
    const int to = tcOffsetDiv2 * 2, w8 = width / 8;
    for (int dir = 0; dir < 2; dir++)
        for (int e = 1; e < (dir ? height : width) / 16; e++)          /* edges at luma 16 * e */
            for (int cu = 0; cu < (dir ? width : height) / 8; cu++)   /* chroma units of 4 lines = 8 luma lines along the edge */
            {
                const int bs = dir ? bsHor[(size_t)(2 * e) * (width / 4) + 2 * cu] : bsVer[(size_t)(2 * cu) * w8 + 2 * e];
                if (bs <= 1) continue;
                int qpA = qp;
                if (qpMap)
                    qpA = dir ? (qpMap[(2 * e - 1) * w8 + cu] + qpMap[(2 * e) * w8 + cu] + 1) >> 1
                              : (qpMap[cu * w8 + 2 * e - 1] + qpMap[cu * w8 + 2 * e] + 1) >> 1;
                for (int c = 0; c < 2; c++)
                {
                    int q = qpA + offs[c];
                    if (q >= 30) q = kChromaScale[q];
                    const int tc = kTc[clip3(0, 53, q + 2 + to)] << (DEPTH - 8);
                    pixel* src = dir ? planes[c] + (intptr_t)(8 * e) * strideC + 4 * cu : planes[c] + (intptr_t)(4 * cu) * strideC + 8 * e;
                    prim.pelFilterChroma[dir](src, dir ? 1 : strideC, dir ? strideC : 1, tc, -1, -1);
                }
            }
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * Stage: sample adaptive offset of the deblocked luma picture (SAO half) - the two pixel passes;
 * the rate-distortion choice of the parameters between them (sao.cpp:1225-1605, entropy-coder bit counts) stays host work.
 *
 *   x265oracle_sao_stats: SAO::calcSaoStatsCTU (encoder/sao.cpp:735-917) for every CTU, luma, bSaoNonDeblocked = 0, bLimitSAO = 0,
 *     one slice: difference source - deblocked, then the oracle's saoCuStatsBO / E0 / E1 / E2 / E3 primitives over the reference's
 *     sub-rectangles (the right 5 columns and bottom 4 rows of a CTU wait for the neighbour CTU's deblocking, so they are left
 *     out unless the CTU touches the picture edge; E0 leaves the bottom 4 rows out even there, :835).
 *   x265oracle_sao_apply: SAO::generateLumaOffsets + applyPixelOffsets (:572-630, :274-570) for every CTU.  The reference works
 *     in place but classifies against saved copies of the not yet offset neighbours (m_tmpU / m_tmpL1, framefilter.cpp:300-322),
 *     which is an out-of-place filter: dst = src + offset[class(src neighbourhood)].  Edge classes: sign(c - a) + sign(c - b) + 2
 *     mapped through s_eoTable = { 1, 2, 0, 3, 4 } (sao.cpp:67) to offset[] with offset[0] = 0; picture-border samples without both
 *     neighbours keep their value; band offset: offset[(c >> (depth - 5)) - bandPos mod 32] for the 4 bands from bandPos.
 * stats layout: int32 [numCtu][5][32] with type order EO_0, EO_1, EO_2, EO_3, BO (sao.h:36-44).
 * params layout: int32 [numCtu][7] = { typeIdx (-1 = off), bandPos, offset[4], mergeLeft (ignored here: merged CTUs carry the
 * left CTU's values, as rdoSaoUnitCu copies them) }. */
#ifdef _OPENMP
#include <omp.h>
#endif

static int sao_sign(int x) { return (x > 0) - (x < 0); }

int EXPORT(x265oracle_sao_stats_plane)(const pixel* fenc, const pixel* rec, intptr_t stride, int picWidth, int picHeight,
                                       int ctuW, int ctuH, int planeOffset, int32_t* count, int32_t* offsetOrg, int nthreads);
int EXPORT(x265oracle_sao_stats)(const pixel* fenc, const pixel* rec, intptr_t stride, int picWidth, int picHeight,
                                 int32_t* count, int32_t* offsetOrg, int nthreads)
{
    return EXPORT(x265oracle_sao_stats_plane)(fenc, rec, stride, picWidth, picHeight, 64, 64, 0, count, offsetOrg, nthreads);
}

/* Any plane: ctuW x ctuH = the CTU's footprint in this plane (64x64 luma, 32x32 for 4:2:0 chroma), picWidth / picHeight the plane's
 * size, planeOffset = the reference's plane_offset (0 luma, 2 chroma: chroma deblocking reaches fewer samples, sao.cpp:782). */
int EXPORT(x265oracle_sao_stats_plane)(const pixel* fenc, const pixel* rec, intptr_t stride, int picWidth, int picHeight,
                                       int ctuW, int ctuH, int planeOffset, int32_t* count, int32_t* offsetOrg, int nthreads)
{
    static x265hip_EncoderPrimitives prim;
    static int ready = 0;
    EXPORT(x265oracle_prims_once)(&prim, &ready);
    const int ctusW = (picWidth + ctuW - 1) / ctuW, ctusH = (picHeight + ctuH - 1) / ctuH;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int addr = 0; addr < ctusW * ctusH; addr++)
    {
        const int lpelx = (addr % ctusW) * ctuW, tpely = (addr / ctusW) * ctuH;
        const int rpelx = lpelx + ctuW < picWidth ? lpelx + ctuW : picWidth, bpely = tpely + ctuH < picHeight ? tpely + ctuH : picHeight;
        const int ctuWidth = rpelx - lpelx, ctuHeight = bpely - tpely;
        const int aboveUnavail = !tpely;
        const pixel* fenc0 = fenc + lpelx + (intptr_t)tpely * stride;
        const pixel* rec0 = rec + lpelx + (intptr_t)tpely * stride;
        int32_t* cnt = count + (size_t)addr * 5 * 32;
        int32_t* org = offsetOrg + (size_t)addr * 5 * 32;
        memset(cnt, 0, sizeof(int32_t) * 5 * 32);
        memset(org, 0, sizeof(int32_t) * 5 * 32);
        int16_t diff[64 * 64] __attribute__((aligned(32)));
        int8_t upStore[2 * (64 + 32)], *upBuff1 = upStore + 16, *upBufft = upBuff1 + (64 + 32);
        for (int y = 0; y < ctuHeight; y++)
            for (int x = 0; x < ctuWidth; x++) diff[y * 64 + x] = (int16_t)((int)fenc0[y * stride + x] - (int)rec0[y * stride + x]);
        const int skipB = 4 - planeOffset, skipR = 5 - planeOffset;
        const int atRight = rpelx == picWidth, atBottom = bpely == picHeight;
        /* band offset: everything already deblocked */
        prim.saoCuStatsBO(diff, rec0, stride, atRight ? ctuWidth : ctuWidth - skipR, atBottom ? ctuHeight : ctuHeight - skipB, org + 4 * 32, cnt + 4 * 32);
        /* EO_0 (horizontal) */
        {
            const int startX = !lpelx, endX = atRight ? ctuWidth - 1 : ctuWidth - skipR;
            prim.saoCuStatsE0(diff + startX, rec0 + startX, stride, endX - startX, ctuHeight - skipB, org + 0 * 32, cnt + 0 * 32);
        }
        /* EO_1 (vertical) */
        {
            const int startY = aboveUnavail, endX = atRight ? ctuWidth : ctuWidth - skipR, endY = atBottom ? ctuHeight - 1 : ctuHeight - skipB;
            const pixel* r = rec0 + startY * stride;
            prim.sign(upBuff1, r, r - stride, ctuWidth);
            prim.saoCuStatsE1(diff + startY * 64, rec0 + startY * stride, stride, upBuff1, endX, endY - startY, org + 1 * 32, cnt + 1 * 32);
        }
        /* EO_2 (135 degrees) and EO_3 (45 degrees) */
        {
            const int startX = !lpelx, endX = atRight ? ctuWidth - 1 : ctuWidth - skipR;
            const int startY = aboveUnavail, endY = atBottom ? ctuHeight - 1 : ctuHeight - skipB;
            const pixel* r = rec0 + startY * stride;
            prim.sign(upBuff1, r + startX, r + startX - stride - 1, endX - startX);
            prim.saoCuStatsE2(diff + startX + startY * 64, rec0 + startX + startY * stride, stride, upBuff1, upBufft, endX - startX, endY - startY,
                              org + 2 * 32, cnt + 2 * 32);
            prim.sign(upBuff1, r + startX - 1, r + startX - 1 - stride + 1, endX - startX + 1);
            prim.saoCuStatsE3(diff + startX + startY * 64, rec0 + startX + startY * stride, stride, upBuff1 + 1, endX - startX, endY - startY,
                              org + 3 * 32, cnt + 3 * 32);
        }
    }
    return 0;
}

int EXPORT(x265oracle_sao_apply_plane)(const pixel* src, pixel* dst, intptr_t stride, int picWidth, int picHeight, int ctuW, int ctuH,
                                       const int32_t* params, int nthreads);
int EXPORT(x265oracle_sao_apply)(const pixel* src, pixel* dst, intptr_t stride, int picWidth, int picHeight, const int32_t* params, int nthreads)
{
    return EXPORT(x265oracle_sao_apply_plane)(src, dst, stride, picWidth, picHeight, 64, 64, params, nthreads);
}

int EXPORT(x265oracle_sao_apply_plane)(const pixel* src, pixel* dst, intptr_t stride, int picWidth, int picHeight, int ctuW, int ctuH,
                                       const int32_t* params, int nthreads)
{
    static const int kEoTable[5] = { 1, 2, 0, 3, 4 };
    const int ctusW = (picWidth + ctuW - 1) / ctuW, ctusH = (picHeight + ctuH - 1) / ctuH;
    const int maxVal = (1 << DEPTH) - 1, boShift = DEPTH - 5;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int addr = 0; addr < ctusW * ctusH; addr++)
    {
        const int32_t* p = params + (size_t)addr * 7;
        const int typeIdx = p[0], bandPos = p[1];
        const int lpelx = (addr % ctusW) * ctuW, tpely = (addr / ctusW) * ctuH;
        const int rpelx = lpelx + ctuW < picWidth ? lpelx + ctuW : picWidth, bpely = tpely + ctuH < picHeight ? tpely + ctuH : picHeight;
        int offsetEo[5], offsetBo[32];
        {
            int off[5] = { 0, p[2], p[3], p[4], p[5] };
            for (int e = 0; e < 5; e++) offsetEo[e] = (int8_t)off[kEoTable[e]];
            memset(offsetBo, 0, sizeof(offsetBo));
            for (int i = 0; i < 4; i++) offsetBo[(bandPos + i) & 31] = (int8_t)p[2 + i];
        }
        /* neighbour steps of the four edge classes */
        static const int kDx[4] = { 1, 0, 1, -1 }, kDy[4] = { 0, 1, 1, 1 };
        for (int y = tpely; y < bpely; y++)
            for (int x = lpelx; x < rpelx; x++)
            {
                const pixel* c = src + x + (intptr_t)y * stride;
                int v = *c;
                if (typeIdx == 4)
                    v = clip3(0, maxVal, v + offsetBo[v >> boShift]);
                else if (typeIdx >= 0)
                {
                    const int dx = kDx[typeIdx], dy = kDy[typeIdx];
                    /* a sample is classified only when both neighbours lie inside the picture (startX / endX / startY / endY) */
                    const int okx = !dx || (x > 0 && x < picWidth - 1), oky = !dy || (y > 0 && y < picHeight - 1);
                    if (okx && oky)
                    {
                        const int a = c[-dx - dy * stride], b = c[dx + dy * stride];
                        const int edgeType = sao_sign(v - a) + sao_sign(v - b) + 2;
                        v = clip3(0, maxVal, v + offsetEo[edgeType]);
                    }
                }
                dst[x + (intptr_t)y * stride] = (pixel)v;
            }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * SAO parameters on the device's side of the seam: SAO::saoStatsInitialOffset (sao.cpp:1378-1433, exact: roundIBDI (:34-37) of
 * offsetOrg / count, clipped to +-(OFFSET_THRESH - 1) with OFFSET_THRESH = 1 << min(depth - 5, 5), the sign constraint of the edge
 * classes) followed by a DISTORTION-ONLY choice of the type: the type (EO_0..EO_3, then BO with its best window of four
 * consecutive bands, first minimum wins) whose initial offsets give the smallest sum of estSaoDist (:56-59), off when no sum is
 * negative.  That choice is a documented stand-in for rdoSaoUnitCu / saoLumaComponentParamDist (:1225-1605), which iterate the
 * offsets against CABAC bit counts and try the merge candidates - entropy-coder work that stays with the host.
 *   initOffset: optional int32 [nctu][5][32] = SAO::m_offset[plane] after saoStatsInitialOffset;  params: int32 [nctu][7] */
int EXPORT(x265oracle_sao_decide)(const int32_t* count, const int32_t* offsetOrg, int nctu, int32_t* initOffset, int32_t* params)
{
    const int thresh = 1 << ((DEPTH - 5) < 5 ? (DEPTH - 5) : 5);
    for (int a = 0; a < nctu; a++)
    {
        const int32_t* cnt = count + (size_t)a * 160;
        const int32_t* org = offsetOrg + (size_t)a * 160;
        int32_t off[5][32];
        memset(off, 0, sizeof(off));
        for (int t = 0; t < 5; t++)
        {
            const int c0 = t < 4 ? 1 : 0, c1 = t < 4 ? 5 : 32;
            for (int c = c0; c < c1; c++)
            {
                const int32_t n = cnt[t * 32 + c], e = org[t * 32 + c];
                if (!n) continue;
                int o = e >= 0 ? (e * 2 + n) / (n * 2) : -((-e * 2 + n) / (n * 2));
                o = clip3(-thresh + 1, thresh - 1, o);
                if (t < 4) o = c < 3 ? (o > 0 ? o : 0) : (o < 0 ? o : 0);
                off[t][c] = o;
            }
        }
        if (initOffset) memcpy(initOffset + (size_t)a * 160, off, sizeof(off));
        int64_t best = 0;
        int32_t* p = params + (size_t)a * 7;
        p[0] = -1; p[1] = 0; p[2] = p[3] = p[4] = p[5] = 0; p[6] = 0;
        for (int t = 0; t < 4; t++)
        {
            int64_t d = 0;
            for (int c = 1; c < 5; c++) d += ((int64_t)cnt[t * 32 + c] * off[t][c] - (int64_t)org[t * 32 + c] * 2) * off[t][c];
            if (d < best) { best = d; p[0] = t; p[1] = 0; for (int i = 0; i < 4; i++) p[2 + i] = off[t][1 + i]; }
        }
        int64_t db[32];
        for (int b = 0; b < 32; b++) db[b] = ((int64_t)cnt[128 + b] * off[4][b] - (int64_t)org[128 + b] * 2) * off[4][b];
        int64_t bo = 0; int start = -1;
        for (int s = 0; s <= 28; s++)
        {
            const int64_t d = db[s] + db[s + 1] + db[s + 2] + db[s + 3];
            if (start < 0 || d < bo) { bo = d; start = s; }
        }
        if (bo < best) { best = bo; p[0] = 4; p[1] = start; for (int i = 0; i < 4; i++) p[2 + i] = off[4][start + i]; }
    }
    return 0;
}
