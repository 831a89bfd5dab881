/* oracle/x265_oracle_pipeline.c
 *
 * TEST INFRASTRUCTURE - NOT PRODUCT CODE (same rules as x265_oracle.c).
 *
 * CPU restatement of how the reference's CALLERS drive the primitives for the batched stages of
 * the frame pipeline, written against the oracle table only (so it is "the reference's CPU path":
 * per-PU calls of pu[].sad / sad_x4 exactly as motion.cpp issues them).  Used by tests as the
 * checker for the HIP batch kernels and by bench.py's cpu_baseline leg (OpenMP over CTUs).
 *
 * Stage: exhaustive integer motion search.  Follows MotionEstimate::setSourcePU
 * (source/encoder/motion.cpp:189-222: the PU is copied into a FENC_STRIDE=64 buffer) and the
 * X265_FULL_SEARCH loop (motion.cpp:1397-1445: raster scan of the mv window, sad_x4 on groups of
 * four consecutive x positions, plain sad for the remainder, COPY2_IF_LT = strict less-than), with
 * the search window centred on the PU (mvp = 0) and bcost starting at "infinity".
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

/* z-order index -> (x, y) in units of the PU size inside the CTU */
static void zorder_xy(int z, int* x, int* y)
{
    *x = (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4);
    *y = ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4);
}

static const int kLevelSize[4] = { 8, 16, 32, 64 };
static const int kLevelPU[4] = { X265HIP_LUMA_8x8, X265HIP_LUMA_16x16, X265HIP_LUMA_32x32, X265HIP_LUMA_64x64 };
static const int kLevelBase[4] = { 0, 64, 80, 84 };   /* position of each PU level inside the 85-entry CTU record */
#define PUS_PER_CTU 85

/* surf : int32 [ctu][mvy][mvx/4][85][4]   best : uint64 [ctu][85] = cost << 32 | (mvyi * NC + mvxi)
 * (85 = 64 8x8 PUs, 16 16x16, 4 32x32, 1 64x64, each level in z-order; mv columns are stored in groups of 4,
 * the last group padded; same layout as the HIP ABI).
 * Either output may be NULL; levelMask selects PU levels (bit l).  Processes CTUs [ctuBegin, ctuEnd). */
int EXPORT(x265oracle_me_fullsearch)(const pixel* fenc, intptr_t fencStride, const pixel* fref, intptr_t frefStride,
                                     int width, int height, int range, int ctuBegin, int ctuEnd,
                                     int32_t* surf, uint64_t* best, const uint16_t* costX, const uint16_t* costY,
                                     int levelMask, int nthreads)
{
    static x265hip_EncoderPrimitives prim;
    static int ready = 0;
    EXPORT(x265oracle_prims_once)(&prim, &ready);
    const int ctusW = width / 64;
    const int NC = 2 * range + 1;
    const int NG = (NC + 3) >> 2;
    (void)height;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int ctu = ctuBegin; ctu < ctuEnd; ctu++)
    {
        const int cx = (ctu % ctusW) * 64, cy = (ctu / ctusW) * 64;
        pixel fencPU[64 * 64] __attribute__((aligned(64)));
        for (int l = 0; l < 4; l++)
        {
            if (!(levelMask & (1 << l)) || (!surf && !best))
                continue;
            const int n = kLevelSize[l], npu = (64 / n) * (64 / n);
            const struct x265hip_PU* pu = &prim.pu[kLevelPU[l]];
            for (int z = 0; z < npu; z++)
            {
                int bx, by;
                zorder_xy(z, &bx, &by);
                const int px = cx + bx * n, py = cy + by * n;
                /* setSourcePU: private copy at stride 64 */
                pu->copy_pp(fencPU, 64, fenc + (intptr_t)py * fencStride + px, fencStride);
                const pixel* refPU = fref + (intptr_t)py * frefStride + px;
                uint32_t bcost = 0xFFFFFFFFu; uint32_t bidx = 0xFFFFFFFFu;
                for (int my = -range; my <= range; my++)
                {
                    for (int mx = -range; mx <= range; )
                    {
                        int32_t costs[4];
                        int cnt;
                        const pixel* base = refPU + (intptr_t)my * frefStride + mx;
                        if (mx + 3 <= range)
                        {
                            pu->sad_x4(fencPU, base, base + 1, base + 2, base + 3, frefStride, costs);
                            cnt = 4;
                        }
                        else
                        {
                            costs[0] = pu->sad(fencPU, 64, base, frefStride);
                            cnt = 1;
                        }
                        for (int k = 0; k < cnt; k++, mx++)
                        {
                            const int xi = mx + range, yi = my + range;
                            if (surf)
                                surf[((((size_t)ctu * NC + yi) * NG + (xi >> 2)) * PUS_PER_CTU + kLevelBase[l] + z) * 4 + (xi & 3)] = costs[k];
                            if (best)
                            {
                                const uint32_t c = (uint32_t)costs[k] + costX[xi] + costY[yi];
                                if (c < bcost) { bcost = c; bidx = (uint32_t)(yi * NC + xi); }
                            }
                        }
                    }
                }
                if (best)
                    best[(size_t)ctu * PUS_PER_CTU + kLevelBase[l] + z] = ((uint64_t)bcost << 32) | bidx;
            }
        }
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * Stage: sub-pel refinement of every PU's integer motion vector.
 * Follows MotionEstimate::motionEstimate after the integer search (source/encoder/motion.cpp:1448-1561,
 * non-lowres branch) and MotionEstimate::subpelCompare (:1571-1664, luma part: bChromaSATD is off for
 * subpelRefine <= 2): square1 half-pel then quarter-pel iterations with the SubpelWorkload table (:48-58),
 * COPY2_IF_LT strict-less updates, candidates measured by luma_hpp / luma_vpp / luma_hvpp into a
 * blockwidth-stride buffer + sad or satd against the 64-stride source copy.
 *
 *   bestIn  : uint64 [ctu][85] from the integer stage (cost << 32 | mvyi * NC + mvxi)
 *   costQ   : uint16 quarter-pel mv component cost, indexed by q + qoff  (q = qpel displacement)
 *   out     : per PU { int32 cost; int16 qmvx; int16 qmvy }  (8 bytes, [ctu][85])
 */
typedef struct { int hpel_iters, hpel_dirs, qpel_iters, qpel_dirs, hpel_satd; } SubpelWorkload;
static const SubpelWorkload kWorkload[8] = {
    { 1, 4, 0, 4, 0 }, { 1, 4, 1, 4, 0 }, { 1, 4, 1, 4, 1 }, { 2, 4, 1, 4, 1 },
    { 2, 4, 2, 4, 1 }, { 1, 8, 1, 8, 1 }, { 2, 8, 1, 8, 1 }, { 2, 8, 2, 8, 1 } };
static const int kSquare1[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { -1, 1 }, { 1, -1 }, { 1, 1 } };

typedef struct { int32_t cost; int16_t qx, qy; } SubpelOut;

static int subpel_compare(const struct x265hip_PU* pu, const pixel* fencPU, const pixel* refPU, intptr_t refStride,
                          int qx, int qy, int n, int useSatd, pixel* subpelbuf)
{
    const pixel* fref = refPU + (qx >> 2) + (intptr_t)(qy >> 2) * refStride;
    const int xFrac = qx & 3, yFrac = qy & 3;
    x265hip_pixelcmp_t cmp = useSatd ? pu->satd : pu->sad;
    if (!(xFrac | yFrac))
        return cmp(fencPU, 64, fref, refStride);
    if (!yFrac) pu->luma_hpp(fref, refStride, subpelbuf, n, xFrac);
    else if (!xFrac) pu->luma_vpp(fref, refStride, subpelbuf, n, yFrac);
    else pu->luma_hvpp(fref, refStride, subpelbuf, n, xFrac, yFrac);
    return cmp(fencPU, 64, subpelbuf, n);
}

int EXPORT(x265oracle_subpel_refine)(const pixel* fenc, intptr_t fencStride, const pixel* fref, intptr_t frefStride,
                                     int width, int height, int range, int ctuBegin, int ctuEnd,
                                     const uint64_t* bestIn, const uint16_t* costQ, int qoff, int subme,
                                     void* outv, int nthreads)
{
    static x265hip_EncoderPrimitives prim;
    static int ready = 0;
    EXPORT(x265oracle_prims_once)(&prim, &ready);
    SubpelOut* out = (SubpelOut*)outv;
    const int ctusW = width / 64;
    const int NC = 2 * range + 1;
    const SubpelWorkload wl = kWorkload[subme];
    (void)height;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int ctu = ctuBegin; ctu < ctuEnd; ctu++)
    {
        const int cx = (ctu % ctusW) * 64, cy = (ctu / ctusW) * 64;
        pixel fencPU[64 * 64] __attribute__((aligned(64)));
        pixel subpelbuf[64 * 64] __attribute__((aligned(64)));
        for (int l = 0; l < 4; l++)
        {
            const int n = kLevelSize[l], npu = (64 / n) * (64 / n);
            const struct x265hip_PU* pu = &prim.pu[kLevelPU[l]];
            for (int z = 0; z < npu; z++)
            {
                int bx, by;
                zorder_xy(z, &bx, &by);
                const int px = cx + bx * n, py = cy + by * n;
                pu->copy_pp(fencPU, 64, fenc + (intptr_t)py * fencStride + px, fencStride);
                const pixel* refPU = fref + (intptr_t)py * frefStride + px;
                const uint64_t key = bestIn[(size_t)ctu * PUS_PER_CTU + kLevelBase[l] + z];
                const int idx = (int)(key & 0xffffffffu);
                int bcost = (int)(key >> 32);
                int bx4 = ((idx % NC) - range) * 4, by4 = ((idx / NC) - range) * 4;      /* bmv.toQPel() */
#define MVCOST(qx, qy) ((int)costQ[(qx) + qoff] + (int)costQ[(qy) + qoff])
                if (!bcost)
                    bcost = MVCOST(bx4, by4);
                else
                {
                    int hpelSatd = wl.hpel_satd;
                    if (hpelSatd)
                        bcost = subpel_compare(pu, fencPU, refPU, frefStride, bx4, by4, n, 1, subpelbuf) + MVCOST(bx4, by4);
                    for (int iter = 0; iter < wl.hpel_iters; iter++)
                    {
                        int bdir = 0;
                        for (int i = 1; i <= wl.hpel_dirs; i++)
                        {
                            const int qx = bx4 + kSquare1[i][0] * 2, qy = by4 + kSquare1[i][1] * 2;
                            const int cost = subpel_compare(pu, fencPU, refPU, frefStride, qx, qy, n, hpelSatd, subpelbuf) + MVCOST(qx, qy);
                            if (cost < bcost) { bcost = cost; bdir = i; }
                        }
                        if (bdir) { bx4 += kSquare1[bdir][0] * 2; by4 += kSquare1[bdir][1] * 2; }
                        else break;
                    }
                    if (!hpelSatd)
                        bcost = subpel_compare(pu, fencPU, refPU, frefStride, bx4, by4, n, 1, subpelbuf) + MVCOST(bx4, by4);
                    for (int iter = 0; iter < wl.qpel_iters; iter++)
                    {
                        int bdir = 0;
                        for (int i = 1; i <= wl.qpel_dirs; i++)
                        {
                            const int qx = bx4 + kSquare1[i][0], qy = by4 + kSquare1[i][1];
                            const int cost = subpel_compare(pu, fencPU, refPU, frefStride, qx, qy, n, 1, subpelbuf) + MVCOST(qx, qy);
                            if (cost < bcost) { bcost = cost; bdir = i; }
                        }
                        if (bdir) { bx4 += kSquare1[bdir][0]; by4 += kSquare1[bdir][1]; }
                        else break;
                    }
                }
#undef MVCOST
                SubpelOut* o = &out[(size_t)ctu * PUS_PER_CTU + kLevelBase[l] + z];
                o->cost = bcost; o->qx = (int16_t)bx4; o->qy = (int16_t)by4;
            }
        }
    }
    return 0;
}
