/* oracle/x265_oracle_search.c
 *
 * TEST INFRASTRUCTURE - NOT PRODUCT CODE (same rules as x265_oracle.c).
 *
 * Restatement of MotionEstimate::motionEstimate (source/encoder/motion.cpp:739-1561) for one PU and a general
 * quarter-pel predictor, luma only, no extra candidates: the predictor / zero-mv start (:772-799), the integer search
 * patterns DIA (:827-850), HEX (:852-948 incl. the square refine), STAR (:1138-1240 + StarPatternSearch :362-604,
 * including the raster refinement's `tmv << 3` cost quirk at :1196) and FULL (:1397-1445), the choice between the
 * search result and the measured predictor (:1452-1458), the zero-residual shortcut (:1464-1469) and the sub-pel
 * refinement (:1508-1561) with subpelCompare (:1571-1613).  UMH and SEA are not restated.
 * Pinned against the real class through oracle/ref_motion.cpp (tests/test_oracle_classes_vs_reference.py).
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

enum { ME_DIA = 0, ME_HEX = 1, ME_UMH = 2, ME_STAR = 3, ME_SEA = 4, ME_FULL = 5 };     /* x265.h:492-497 */

typedef struct { int32_t px, py, w, h, qmvpx, qmvpy, out_qmvx, out_qmvy, out_cost; } me_job;
typedef struct { int x, y; } mv_t;

typedef struct { int hpel_iters, hpel_dirs, qpel_iters, qpel_dirs, hpel_satd; } workload_t;
static const workload_t kWorkload[8] = {          /* motion.cpp:48-58 */
    { 1, 4, 0, 4, 0 }, { 1, 4, 1, 4, 0 }, { 1, 4, 1, 4, 1 }, { 2, 4, 1, 4, 1 },
    { 2, 4, 2, 4, 1 }, { 1, 8, 1, 8, 1 }, { 2, 8, 1, 8, 1 }, { 2, 8, 2, 8, 1 } };
static const mv_t kHex2[8] = { { -1, -2 }, { -2, 0 }, { -1, 2 }, { 1, 2 }, { 2, 0 }, { 1, -2 }, { -1, -2 }, { -2, 0 } };
static const uint8_t kMod6m1[8] = { 5, 0, 1, 2, 3, 4, 5, 0 };
static const mv_t kSquare1[9] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { -1, 1 }, { 1, -1 }, { 1, 1 } };
static const mv_t kOffsets[16] = { { -1, 0 }, { 0, -1 }, { -1, -1 }, { 1, -1 }, { -1, 0 }, { 1, 0 }, { -1, 1 }, { -1, -1 },
                                   { 1, -1 }, { 1, 1 }, { -1, 0 }, { 0, 1 }, { -1, 1 }, { 1, 1 }, { 1, 0 }, { 0, 1 } };
static const int kPuDims[25][2] = {
    { 4, 4 }, { 8, 8 }, { 16, 16 }, { 32, 32 }, { 64, 64 }, { 8, 4 }, { 4, 8 }, { 16, 8 }, { 8, 16 }, { 32, 16 }, { 16, 32 },
    { 64, 32 }, { 32, 64 }, { 16, 12 }, { 12, 16 }, { 16, 4 }, { 4, 16 }, { 32, 24 }, { 24, 32 }, { 32, 8 }, { 8, 32 },
    { 64, 48 }, { 48, 64 }, { 64, 16 }, { 16, 64 } };

typedef struct
{
    const struct x265hip_PU* pu;
    pixel fenc[64 * 64];              /* the PU source at stride 64 (motion.cpp:193) */
    const pixel* fref;                /* reference sample under the PU's top-left corner */
    intptr_t stride;
    int w, h;
    const uint16_t* cost;             /* cost[q]: bit cost of a quarter-pel mv DIFFERENCE component q (index 0 = zero) */
    int mvpx, mvpy;
    mv_t mvmin, mvmax;
    const pixel* lowres[4];           /* lowres references only: the four half-pel phase planes under the block (else lowres[0] = NULL) */
} me_ctx;

static inline int mvcost_q(const me_ctx* c, int qx, int qy) { return c->cost[qx - c->mvpx] + c->cost[qy - c->mvpy]; }
static inline int sad_at(const me_ctx* c, int mx, int my) { return c->pu->sad(c->fenc, 64, c->fref + mx + (intptr_t)my * c->stride, c->stride); }
static inline int cost_mv(const me_ctx* c, int mx, int my) { return sad_at(c, mx, my) + mvcost_q(c, mx * 4, my * 4); }
static inline int in_range(const me_ctx* c, int x, int y) { return x >= c->mvmin.x && x <= c->mvmax.x && y >= c->mvmin.y && y <= c->mvmax.y; }

/* motion.cpp:1571-1613 */
static int subpel_compare(const me_ctx* c, int qx, int qy, int useSatd)
{
    const pixel* fref = c->fref + (qx >> 2) + (intptr_t)(qy >> 2) * c->stride;
    const int xf = qx & 3, yf = qy & 3;
    x265hip_pixelcmp_t cmp = useSatd ? c->pu->satd : c->pu->sad;
    if (!(xf | yf)) return cmp(c->fenc, 64, fref, c->stride);
    pixel buf[64 * 64];
    if (!yf) c->pu->luma_hpp(fref, c->stride, buf, c->w, xf);
    else if (!xf) c->pu->luma_vpp(fref, c->stride, buf, c->w, yf);
    else c->pu->luma_hvpp(fref, c->stride, buf, c->w, xf, yf);
    return cmp(c->fenc, 64, buf, c->w);
}

/* lowres.h:95-121 (ReferencePlanes::lowresQPelCost): half-pel positions read a phase plane, quarter-pel positions average two */
static int lowres_qpel_cost(const me_ctx* c, int qx, int qy, int useSatd)
{
    x265hip_pixelcmp_t cmp = useSatd ? c->pu->satd : c->pu->sad;
    if ((qx | qy) & 1)
    {
        pixel buf[8 * 8] __attribute__((aligned(16)));
        const int hpelA = (qy & 2) | ((qx & 2) >> 1);
        const pixel* frefA = c->lowres[hpelA] + (qx >> 2) + (intptr_t)(qy >> 2) * c->stride;
        const int qmvx = qx + (qx & 1), qmvy = qy + (qy & 1);
        const int hpelB = (qmvy & 2) | ((qmvx & 2) >> 1);
        const pixel* frefB = c->lowres[hpelB] + (qmvx >> 2) + (intptr_t)(qmvy >> 2) * c->stride;
        c->pu->pixelavg_pp[0](buf, 8, frefA, c->stride, frefB, c->stride, 32);
        return cmp(c->fenc, 64, buf, 8);
    }
    const int hpel = (qy & 2) | ((qx & 2) >> 1);
    return cmp(c->fenc, 64, c->lowres[hpel] + (qx >> 2) + (intptr_t)(qy >> 2) * c->stride, c->stride);
}

/* motion.cpp:362-604; point numbers and distances as in the reference's diagrams */
static void star_pattern(const me_ctx* c, mv_t* bmv, int* bcost, int* bPointNr, int* bDistance, int earlyExitIters, int merange)
{
    const mv_t omv = *bmv;
    int saved = *bcost, rounds = 0;
#define PT(MX, MY, P, D) do { const int cost_ = cost_mv(c, (MX), (MY)); \
        if (cost_ < *bcost) { *bcost = cost_; bmv->x = (MX); bmv->y = (MY); *bPointNr = (P); *bDistance = (D); } } while (0)
    {
        const int dist = 1;
        const int top = omv.y - dist, bottom = omv.y + dist, left = omv.x - dist, right = omv.x + dist;
        /* inside the bounds the reference scores the same four points with one sad_x4 in this order */
        if (top >= c->mvmin.y) PT(omv.x, top, 2, dist);
        if (left >= c->mvmin.x) PT(left, omv.y, 4, dist);
        if (right <= c->mvmax.x) PT(right, omv.y, 5, dist);
        if (bottom <= c->mvmax.y) PT(omv.x, bottom, 7, dist);
        if (*bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
    for (int dist = 2; dist <= 8; dist <<= 1)
    {
        const int top = omv.y - dist, bottom = omv.y + dist, left = omv.x - dist, right = omv.x + dist;
        const int top2 = omv.y - (dist >> 1), bottom2 = omv.y + (dist >> 1), left2 = omv.x - (dist >> 1), right2 = omv.x + (dist >> 1);
        saved = *bcost;
        if (top >= c->mvmin.y && left >= c->mvmin.x && right <= c->mvmax.x && bottom <= c->mvmax.y)
        {
            /* the x4 order differs from the per-point order of the border branch (:448-455 vs :459-500) */
            PT(omv.x, top, 2, dist); PT(left2, top2, 1, dist >> 1); PT(right2, top2, 3, dist >> 1); PT(left, omv.y, 4, dist);
            PT(right, omv.y, 5, dist); PT(left2, bottom2, 6, dist >> 1); PT(right2, bottom2, 8, dist >> 1); PT(omv.x, bottom, 7, dist);
        }
        else
        {
            if (top >= c->mvmin.y) PT(omv.x, top, 2, dist);
            if (top2 >= c->mvmin.y)
            {
                if (left2 >= c->mvmin.x) PT(left2, top2, 1, dist >> 1);
                if (right2 <= c->mvmax.x) PT(right2, top2, 3, dist >> 1);
            }
            if (left >= c->mvmin.x) PT(left, omv.y, 4, dist);
            if (right <= c->mvmax.x) PT(right, omv.y, 5, dist);
            if (bottom2 <= c->mvmax.y)
            {
                if (left2 >= c->mvmin.x) PT(left2, bottom2, 6, dist >> 1);
                if (right2 <= c->mvmax.x) PT(right2, bottom2, 8, dist >> 1);
            }
            if (bottom <= c->mvmax.y) PT(omv.x, bottom, 7, dist);
        }
        if (*bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
    for (int dist = 16; dist <= (int)(int16_t)merange; dist <<= 1)
    {
        const int top = omv.y - dist, bottom = omv.y + dist, left = omv.x - dist, right = omv.x + dist;
        saved = *bcost;
        if (top >= c->mvmin.y && left >= c->mvmin.x && right <= c->mvmax.x && bottom <= c->mvmax.y)
        {
            PT(omv.x, top, 0, dist); PT(left, omv.y, 0, dist); PT(right, omv.y, 0, dist); PT(omv.x, bottom, 0, dist);
            for (int index = 1; index < 4; index++)
            {
                const int posYT = top + (dist >> 2) * index, posYB = bottom - (dist >> 2) * index;
                const int posXL = omv.x - (dist >> 2) * index, posXR = omv.x + (dist >> 2) * index;
                PT(posXL, posYT, 0, dist); PT(posXR, posYT, 0, dist); PT(posXL, posYB, 0, dist); PT(posXR, posYB, 0, dist);
            }
        }
        else
        {
            if (top >= c->mvmin.y) PT(omv.x, top, 0, dist);
            if (left >= c->mvmin.x) PT(left, omv.y, 0, dist);
            if (right <= c->mvmax.x) PT(right, omv.y, 0, dist);
            if (bottom <= c->mvmax.y) PT(omv.x, bottom, 0, dist);
            for (int index = 1; index < 4; index++)
            {
                const int posYT = top + (dist >> 2) * index, posYB = bottom - (dist >> 2) * index;
                const int posXL = omv.x - (dist >> 2) * index, posXR = omv.x + (dist >> 2) * index;
                if (posYT >= c->mvmin.y)
                {
                    if (posXL >= c->mvmin.x) PT(posXL, posYT, 0, dist);
                    if (posXR <= c->mvmax.x) PT(posXR, posYT, 0, dist);
                }
                if (posYB <= c->mvmax.y)
                {
                    if (posXL >= c->mvmin.x) PT(posXL, posYB, 0, dist);
                    if (posXR <= c->mvmax.x) PT(posXR, posYB, 0, dist);
                }
            }
        }
        if (*bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
#undef PT
}

/* X265_HEX_SEARCH (motion.cpp:847-942, also the tail of UMH through `goto me_hex2`): hexagon of radius 2, half-hexagon walk,
 * square refine */
static void hex_refine(const me_ctx* c, mv_t* pbmv, int* pbcost, int merange)
{
    mv_t bmv = *pbmv;
    int bcost = *pbcost;
    int costs[4];
        /* first full hexagon: six points in two groups of three; out-of-range rows are scored but not accepted */
#define X3(D0, D1, D2) do { costs[0] = cost_mv(c, bmv.x + (D0).x, bmv.y + (D0).y); costs[1] = cost_mv(c, bmv.x + (D1).x, bmv.y + (D1).y); \
                            costs[2] = cost_mv(c, bmv.x + (D2).x, bmv.y + (D2).y); } while (0)
#define YOK(DY) ((bmv.y + (DY) >= c->mvmin.y) & (bmv.y + (DY) <= c->mvmax.y))
#define LT(V) do { if ((V) < bcost) bcost = (V); } while (0)
        { const mv_t a = { -2, 0 }, b = { -1, 2 }, d = { 1, 2 }; X3(a, b, d); }
        bcost <<= 3;
        if (YOK(0)) LT((costs[0] << 3) + 2);
        if (YOK(2)) { LT((costs[1] << 3) + 3); LT((costs[2] << 3) + 4); }
        { const mv_t a = { 2, 0 }, b = { 1, -2 }, d = { -1, -2 }; X3(a, b, d); }
        if (YOK(0)) LT((costs[0] << 3) + 5);
        if (YOK(-2)) { LT((costs[1] << 3) + 6); LT((costs[2] << 3) + 7); }
        if (bcost & 7)
        {
            int dir = (bcost & 7) - 2;
            if (YOK(kHex2[dir + 1].y))
            {
                bmv.x += kHex2[dir + 1].x; bmv.y += kHex2[dir + 1].y;
                /* half hexagons that do not overlap the previous iteration */
                for (int i = (merange >> 1) - 1; i > 0 && in_range(c, bmv.x, bmv.y); i--)
                {
                    X3(kHex2[dir + 0], kHex2[dir + 1], kHex2[dir + 2]);
                    bcost &= ~7;
                    if (YOK(kHex2[dir + 0].y)) LT((costs[0] << 3) + 1);
                    if (YOK(kHex2[dir + 1].y)) LT((costs[1] << 3) + 2);
                    if (YOK(kHex2[dir + 2].y)) LT((costs[2] << 3) + 3);
                    if (!(bcost & 7)) break;
                    dir += (bcost & 7) - 2;
                    dir = kMod6m1[dir + 1];
                    bmv.x += kHex2[dir + 1].x; bmv.y += kHex2[dir + 1].y;
                }
            }
        }
        bcost >>= 3;
        /* square refine */
        int dir = 0;
        costs[0] = cost_mv(c, bmv.x, bmv.y - 1); costs[1] = cost_mv(c, bmv.x, bmv.y + 1);
        costs[2] = cost_mv(c, bmv.x - 1, bmv.y); costs[3] = cost_mv(c, bmv.x + 1, bmv.y);
        if (YOK(-1) && costs[0] < bcost) { bcost = costs[0]; dir = 1; }
        if (YOK(1) && costs[1] < bcost) { bcost = costs[1]; dir = 2; }
        if (costs[2] < bcost) { bcost = costs[2]; dir = 3; }
        if (costs[3] < bcost) { bcost = costs[3]; dir = 4; }
        costs[0] = cost_mv(c, bmv.x - 1, bmv.y - 1); costs[1] = cost_mv(c, bmv.x - 1, bmv.y + 1);
        costs[2] = cost_mv(c, bmv.x + 1, bmv.y - 1); costs[3] = cost_mv(c, bmv.x + 1, bmv.y + 1);
        if (YOK(-1) && costs[0] < bcost) { bcost = costs[0]; dir = 5; }
        if (YOK(1) && costs[1] < bcost) { bcost = costs[1]; dir = 6; }
        if (YOK(-1) && costs[2] < bcost) { bcost = costs[2]; dir = 7; }
        if (YOK(1) && costs[3] < bcost) { bcost = costs[3]; dir = 8; }
        bmv.x += kSquare1[dir].x; bmv.y += kSquare1[dir].y;
#undef X3
#undef YOK
#undef LT
    *pbmv = bmv; *pbcost = bcost;
}

/* motion.cpp:61,123-150: SAD_THRESH(v) = bcost < (v >> 4) * sizeScale[part], sizeScale = (H * H) >> 4 */
static int sad_thresh(int bcost, int v, int h) { return bcost < ((v >> 4) * ((h * h) >> 4)); }

static const mv_t kHex4[16] = { { 0, -4 }, { 0, 4 }, { -2, -3 }, { 2, -3 }, { -4, -2 }, { 4, -2 }, { -4, -1 }, { 4, -1 },
                                { -4, 0 }, { 4, 0 }, { -4, 1 }, { 4, 1 }, { -4, 2 }, { 4, 2 }, { -2, 3 }, { 2, 3 } };

/* X265_UMH_SEARCH (motion.cpp:946-1130); pmv = the clipped predictor rounded to full-pel.  Upstream quirks kept: COST_MV_X4 only tests the candidate's y against the search range (:295-302); the hexagon
 * grid's fast path tests omv.y + dy with the UNscaled hex4 offset (:1087 MIN_MV) and keeps its winner in a packed dx * 16 + (dy & 15);
 * with mv candidates the range is rescaled by range_mul[mvd_ctx][sad_ctx] (:982-1040) and that range also drives the final
 * hexagon refinement. */
static void umh_search(const me_ctx* c, mv_t* pbmv, int* pbcost, int merange, int pmvx, int pmvy, int h, int is64,
                       const int32_t* mvc, int numMvc)
{
    mv_t bmv = *pbmv, omv;
    int bcost = *pbcost;
    int costs[16];
#define X4(M0X, M0Y, M1X, M1Y, M2X, M2Y, M3X, M3Y) do { \
        const int dx_[4] = { (M0X), (M1X), (M2X), (M3X) }, dy_[4] = { (M0Y), (M1Y), (M2Y), (M3Y) }; \
        for (int k_ = 0; k_ < 4; k_++) costs[k_] = cost_mv(c, omv.x + dx_[k_], omv.y + dy_[k_]); \
        for (int k_ = 0; k_ < 4; k_++) \
            if ((omv.y + dy_[k_] >= c->mvmin.y) & (omv.y + dy_[k_] <= c->mvmax.y)) \
                if (costs[k_] < bcost) { bcost = costs[k_]; bmv.x = omv.x + dx_[k_]; bmv.y = omv.y + dy_[k_]; } } while (0)
#define ONE(MX, MY) do { const int cost_ = cost_mv(c, (MX), (MY)); if (cost_ < bcost) { bcost = cost_; bmv.x = (MX); bmv.y = (MY); } } while (0)
#define DIA1(MX, MY) do { omv.x = (MX); omv.y = (MY); X4(0, -1, 0, 1, -1, 0, 1, 0); } while (0)
#define CROSS(START, XMAX, YMAX) do { \
        int16_t i_ = (int16_t)(START); \
        const int xm_ = (XMAX), ym_ = (YMAX); \
        int lim_ = c->mvmax.x - omv.x < omv.x - c->mvmin.x ? c->mvmax.x - omv.x : omv.x - c->mvmin.x; \
        if (xm_ <= lim_) for (; i_ < xm_ - 2; i_ += 4) X4(i_, 0, -i_, 0, i_ + 2, 0, -i_ - 2, 0); \
        for (; i_ < xm_; i_ += 2) { if (omv.x + i_ <= c->mvmax.x) ONE(omv.x + i_, omv.y); if (omv.x - i_ >= c->mvmin.x) ONE(omv.x - i_, omv.y); } \
        i_ = (int16_t)(START); \
        lim_ = c->mvmax.y - omv.y < omv.y - c->mvmin.y ? c->mvmax.y - omv.y : omv.y - c->mvmin.y; \
        if (ym_ <= lim_) for (; i_ < ym_ - 2; i_ += 4) X4(0, i_, 0, -i_, 0, i_ + 2, 0, -i_ - 2); \
        for (; i_ < ym_; i_ += 2) { if (omv.y + i_ <= c->mvmax.y) ONE(omv.x, omv.y + i_); if (omv.y - i_ >= c->mvmin.y) ONE(omv.x, omv.y - i_); } } while (0)
    int16_t cross_start = 1;
    /* refine predictors */
    omv = bmv;
    const int ucost1 = bcost;
    DIA1(pmvx, pmvy);
    if (pmvx | pmvy) DIA1(0, 0);
    const int ucost2 = bcost;
    if ((bmv.x | bmv.y) && (bmv.x != pmvx || bmv.y != pmvy)) DIA1(bmv.x, bmv.y);
    if (bcost == ucost2) cross_start = 3;
    /* early termination */
    omv = bmv;
    int done = 0;
    if (bcost == ucost2 && sad_thresh(bcost, 2000, h))
    {
        X4(0, -2, -1, -1, 1, -1, -2, 0);
        X4(2, 0, -1, 1, 1, 1, 0, 2);
        if (bcost == ucost1 && sad_thresh(bcost, 500, h)) done = 1;
        else if (bcost == ucost2)
        {
            const int16_t range = (int16_t)((merange >> 1) | 1);
            CROSS(3, range, range);
            X4(-1, -2, 1, -2, -2, -1, 2, -1);
            X4(-2, 1, 2, 1, -1, 2, 1, 2);
            if (bcost == ucost2) done = 1;
            else cross_start = (int16_t)(range + 2);
        }
    }
    if (!done)
    {
        /* adaptive search range based on the agreement of the mv candidates (:982-1040) */
        if (numMvc)
        {
            static const uint8_t range_mul[4][4] = { { 3, 3, 4, 4 }, { 3, 4, 4, 4 }, { 4, 4, 4, 5 }, { 4, 4, 5, 6 } };
            int mvd, denom = 1;
            if (numMvc == 1)
                mvd = is64 ? 25 : abs(c->mvpx - mvc[0]) + abs(c->mvpy - mvc[1]);
            else
            {
                denom = numMvc - 1;
                mvd = 0;
                if (!is64) { mvd = abs(c->mvpx - mvc[0]) + abs(c->mvpy - mvc[1]); denom++; }
                for (int i = 0; i < numMvc - 1; i++) mvd += abs(mvc[2 * i] - mvc[2 * i + 2]) + abs(mvc[2 * i + 1] - mvc[2 * i + 3]);
            }
            const int sad_ctx = sad_thresh(bcost, 1000, h) ? 0 : sad_thresh(bcost, 2000, h) ? 1 : sad_thresh(bcost, 4000, h) ? 2 : 3;
            const int mvd_ctx = mvd < 10 * denom ? 0 : mvd < 20 * denom ? 1 : mvd < 40 * denom ? 2 : 3;
            merange = (merange * range_mul[mvd_ctx][sad_ctx]) >> 2;
        }
        CROSS(cross_start, merange, merange >> 1);
        X4(-2, -2, -2, 2, 2, -2, 2, 2);
        /* hexagon grid */
        omv = bmv;
        uint16_t i = 1;
        do
        {
            int m = c->mvmax.x - omv.x;
            if (omv.x - c->mvmin.x < m) m = omv.x - c->mvmin.x;
            if (c->mvmax.y - omv.y < m) m = c->mvmax.y - omv.y;
            if (omv.y - c->mvmin.y < m) m = omv.y - c->mvmin.y;
            if (4 * i > m)
            {
                for (int j = 0; j < 16; j++)
                {
                    const int mx = omv.x + kHex4[j].x * i, my = omv.y + kHex4[j].y * i;
                    if (in_range(c, mx, my)) ONE(mx, my);
                }
            }
            else
            {
                int16_t dir = 0;
                /* the reference's sad_x4 order (:1075-1078): k = 0..15 */
                static const int8_t ox[16] = { 0, 0, -2, 2, -4, 4, -4, 4, -4, 4, -4, 4, -4, 4, -2, 2 };
                static const int8_t oy[16] = { -4, 4, -3, -3, -2, -2, -1, -1, 0, 0, 1, 1, 2, 2, 3, 3 };
                for (int k = 0; k < 16; k++) costs[k] = cost_mv(c, omv.x + ox[k] * i, omv.y + oy[k] * i);
                for (int k = 0; k < 16; k++)
                    if ((omv.y + oy[k] >= c->mvmin.y) & (omv.y + oy[k] <= c->mvmax.y))      /* unscaled dy: upstream */
                        if (costs[k] < bcost) { bcost = costs[k]; dir = (int16_t)(ox[k] * 16 + (oy[k] & 15)); }
                if (dir)
                {
                    bmv.x = omv.x + i * (dir >> 4);
                    bmv.y = omv.y + i * ((int)((uint32_t)(int)dir << 28) >> 28);
                }
            }
        }
        while (++i <= merange >> 2);
        if (in_range(c, bmv.x, bmv.y)) hex_refine(c, &bmv, &bcost, merange);
    }
#undef X4
#undef ONE
#undef DIA1
#undef CROSS
    *pbmv = bmv; *pbcost = bcost;
}

static int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

/* ---- X265_SEA (motion.cpp:1241-1395).  The reference keeps twelve "integral planes" per reference picture
 * (framefilter.cpp:716-823 with the integral_init*h / *v primitives): plane value at (x, y) = the sum of the bw x bh block of
 * reference samples whose top-left corner is (x, y), for (bw, bh) in the order below.  Here the block sums are taken straight
 * from the reference samples; the pin test builds the planes with the real primitives. */
static const int kSeaPlaneDims[12][2] = { { 32, 32 }, { 32, 24 }, { 32, 8 }, { 24, 32 }, { 16, 16 }, { 16, 12 }, { 16, 4 }, { 12, 16 },
                                          { 8, 32 }, { 8, 8 }, { 4, 16 }, { 4, 4 } };     /* framedata.h:171 */

typedef struct
{
    int nAds;                 /* terms of the ADS sum: the PU's ads_x1 / ads_x2 / ads_x4 (pixel.cpp:1105-1129) */
    int plane;                /* index into kSeaPlaneDims */
    int offX[4], offY[4];     /* position of each term's block sum relative to the candidate */
    int encDC[4];
} sea_plan;

/* the choices motion.cpp:1258-1364 makes for a w x h PU; -1 for the sizes whose DC terms read source samples outside the PU
 * (8x4, 4x8, 32x8, 8x32: fenc + deltaX / deltaY lies beyond the block that setSourcePU copied) */
static int sea_make_plan(const me_ctx* c, sea_plan* p)
{
    const int w = c->w, h = c->h;
    if ((w == 8 && h == 4) || (w == 4 && h == 8) || (w == 32 && h == 8) || (w == 8 && h == 32) || (w == 4 && h == 4)) return -1;
    const int vertical = h == 2 * w, horizontal = w == 2 * h, square = w == h;
    const int smallRect = (w == 16 && h == 12) || (w == 12 && h == 16) || (w == 16 && h == 4) || (w == 4 && h == 16);
    const int asymVertical = !square && !vertical && w < h;
    const int deltaX = w <= 8 ? w : w >> 1, deltaY = h <= 8 ? h : h >> 1;
    int tw, th;                                                     /* the block size the source DCs are taken over (:1283-1303) */
    if (vertical) { tw = w; th = h >> 1; }
    else if (horizontal) { tw = w >> 1; th = h; }
    else if (!square) { tw = smallRect ? w : w >> 1; th = smallRect ? h : h >> 1; }
    else { tw = w <= 8 ? w : w >> 1; th = w <= 8 ? h : h >> 1; }
    p->nAds = (square ? w <= 8 : smallRect) ? 1 : ((vertical || horizontal) ? 2 : 4);
    int bw, bh;                                                     /* :1315-1347, keyed on deltaX / deltaY only */
    switch (deltaX)
    {
    case 32: bw = 32; bh = (deltaY % 24 == 0) ? 24 : (deltaY == 8 ? 8 : 32); break;
    case 24: bw = 24; bh = 32; break;
    case 16: bw = 16; bh = (deltaY % 12 == 0) ? 12 : (deltaY == 4 ? 4 : 16); break;
    case 12: bw = 12; bh = 16; break;
    case 8: bw = 8; bh = deltaY == 32 ? 32 : 8; break;
    case 4: bw = 4; bh = deltaY == 16 ? 16 : 4; break;
    default: bw = 4; bh = 4; break;
    }
    p->plane = -1;
    for (int k = 0; k < 12; k++) if (kSeaPlaneDims[k][0] == bw && kSeaPlaneDims[k][1] == bh) p->plane = k;
    int dc[4];
    for (int k = 0; k < 4; k++)
    {
        dc[k] = 0;
        const int ox = (k & 1) ? deltaX : 0, oy = (k & 2) ? deltaY : 0;
        if (ox + tw > w || oy + th > h) continue;                   /* never used by the ADS variant of a supported size */
        for (int y = 0; y < th; y++)
            for (int x = 0; x < tw; x++) dc[k] += c->fenc[(oy + y) * 64 + ox + x];
    }
    /* `delta` of the ads call: deltaY rows for squares, vertical and asymmetric-vertical PUs (:1349-1355), deltaY SAMPLES along the
     * row for the asymmetric-horizontal ones (they are missing from that list), deltaX samples for horizontal PUs (:1360-1361) */
    const int rows = square || vertical || asymVertical;
    const int dx = horizontal ? deltaX : (rows ? 0 : deltaY), dy = (!horizontal && rows) ? deltaY : 0;
    memset(p->offX, 0, sizeof(p->offX)); memset(p->offY, 0, sizeof(p->offY));
    if (p->nAds == 4)
    {
        p->offX[1] = w >> 1;                                        /* ads_x4<lx, ly>: sums[lx >> 1] with the WHOLE PU's lx */
        p->offX[2] = dx; p->offY[2] = dy;
        p->offX[3] = dx + (w >> 1); p->offY[3] = dy;
        memcpy(p->encDC, dc, sizeof(dc));
    }
    else if (p->nAds == 2)
    {
        p->offX[1] = dx; p->offY[1] = dy;
        p->encDC[0] = dc[0]; p->encDC[1] = vertical ? dc[2] : dc[1];    /* :1357-1358 */
    }
    else p->encDC[0] = dc[0];
    return 0;
}

static int sea_block_sum(const me_ctx* c, int plane, int x, int y)
{
    const pixel* r = c->fref + x + (intptr_t)y * c->stride;
    int s = 0;
    for (int j = 0; j < kSeaPlaneDims[plane][1]; j++, r += c->stride)
        for (int i = 0; i < kSeaPlaneDims[plane][0]; i++) s += r[i];
    return s;
}

/* The row loop (:1366-1391).  Three different mv-cost expressions meet here: the ADS filter adds the ordinary x cost
 * (fpelCostMvX), the sad_x3 groups add m_cost[4x - 2 * qmvp.x] (p_cost_mvx is m_cost_mvx, itself already offset by the
 * predictor, offset again) and no y cost, the row's y cost is m_cost[y - 2 * qmvp.y] << 2 (a full-pel y indexing the quarter-pel
 * table), and the 0-2 candidates left over after the groups of three use the ordinary mvcost. */
static int sea_search(const me_ctx* c, mv_t* pbmv, int* pbcost, int merange)
{
    sea_plan p;
    if (sea_make_plan(c, &p)) return -1;
    mv_t bmv = *pbmv;
    int bcost = *pbcost;
    const int ox = bmv.x, oy = bmv.y;
    const int minX = ox - merange > c->mvmin.x ? ox - merange : c->mvmin.x, minY = oy - merange > c->mvmin.y ? oy - merange : c->mvmin.y;
    const int maxX = ox + merange < c->mvmax.x ? ox + merange : c->mvmax.x, maxY = oy + merange < c->mvmax.y ? oy + merange : c->mvmax.y;
    const int width = (maxX - minX + 3) & ~3;                       /* up to three candidates beyond maxX are examined */
    int16_t list[4 * 64 + 8];
    for (int ty = minY; ty <= maxY; ty++)
    {
        const int ycost = c->cost[ty - 2 * c->mvpy] << 2;
        if (bcost <= ycost) continue;
        bcost -= ycost;
        int n = 0;
        for (int i = 0; i < width; i++)
        {
            const int x = minX + i;
            int ads = c->cost[4 * x - c->mvpx];
            for (int k = 0; k < p.nAds; k++) ads += abs(p.encDC[k] - sea_block_sum(c, p.plane, x + p.offX[k], ty + p.offY[k]));
            if (ads < bcost) list[n++] = (int16_t)i;
        }
        int i = 0;
        for (; i < n - 2; i += 3)
            for (int k = 0; k < 3; k++)
            {
                const int x = minX + list[i + k];
                const int cost = sad_at(c, x, ty) + c->cost[4 * x - 2 * c->mvpx];
                if (cost < bcost) { bcost = cost; bmv.x = x; bmv.y = ty; }
            }
        bcost += ycost;
        for (; i < n; i++)
        {
            const int x = minX + list[i];
            const int cost = cost_mv(c, x, ty);
            if (cost < bcost) { bcost = cost; bmv.x = x; bmv.y = ty; }
        }
    }
    *pbmv = bmv; *pbcost = bcost;
    return 0;
}

static int motion_estimate_one(me_ctx* c, int method, int subme, int merange, const int32_t* mvc, int numMvc, int* outQx, int* outQy)
{
    const int qminx = c->mvmin.x * 4, qminy = c->mvmin.y * 4, qmaxx = c->mvmax.x * 4, qmaxy = c->mvmax.y * 4;
    /* measure the clipped quarter-pel predictor (:772-781), re-measure its full-pel rounding (:783-787), try mv 0 (:789-799) */
    int pmvx = clip3(qminx, qmaxx, c->mvpx), pmvy = clip3(qminy, qmaxy, c->mvpy);
    int bestprex = pmvx, bestprey = pmvy;
    int bprecost = c->lowres[0] ? lowres_qpel_cost(c, pmvx, pmvy, 0) /* :775-776 */ : subpel_compare(c, pmvx, pmvy, 0);
    mv_t bmv = { (pmvx + 2) >> 2, (pmvy + 2) >> 2 };
    int bcost = bprecost;
    if ((pmvx | pmvy) & 3) bcost = cost_mv(c, bmv.x, bmv.y);
    if (pmvx | pmvy)
    {
        const int cost = sad_at(c, 0, 0) + mvcost_q(c, 0, 0);
        if (cost < bcost)
        {
            bcost = cost;
            bmv.x = 0;
            const int zy = 0 < c->mvmax.y ? 0 : c->mvmax.y;
            bmv.y = zy > c->mvmin.y ? zy : c->mvmin.y;
        }
    }
    /* extra quarter-pel candidates (:800-812): SAD + mv cost against the measured predictor, not against the search start */
    for (int i = 0; i < numMvc; i++)
    {
        const int mx = clip3(qminx, qmaxx, mvc[2 * i]), my = clip3(qminy, qmaxy, mvc[2 * i + 1]);
        if ((mx | my) && (mx != pmvx || my != pmvy) && (mx != bestprex || my != bestprey))
        {
            const int cost = subpel_compare(c, mx, my, 0) + mvcost_q(c, mx, my);
            if (cost < bprecost) { bprecost = cost; bestprex = mx; bestprey = my; }
        }
    }
    int costs[4];
    switch (method)
    {
    case ME_DIA:
    {
        bcost <<= 4;
        int i = merange;
        do
        {
            costs[0] = cost_mv(c, bmv.x, bmv.y - 1); costs[1] = cost_mv(c, bmv.x, bmv.y + 1);
            costs[2] = cost_mv(c, bmv.x - 1, bmv.y); costs[3] = cost_mv(c, bmv.x + 1, bmv.y);
            if ((bmv.y - 1 >= c->mvmin.y) & (bmv.y - 1 <= c->mvmax.y)) { if ((costs[0] << 4) + 1 < bcost) bcost = (costs[0] << 4) + 1; }
            if ((bmv.y + 1 >= c->mvmin.y) & (bmv.y + 1 <= c->mvmax.y)) { if ((costs[1] << 4) + 3 < bcost) bcost = (costs[1] << 4) + 3; }
            if ((costs[2] << 4) + 4 < bcost) bcost = (costs[2] << 4) + 4;
            if ((costs[3] << 4) + 12 < bcost) bcost = (costs[3] << 4) + 12;
            if (!(bcost & 15)) break;
            bmv.x -= (int)((uint32_t)bcost << 28) >> 30;       /* the direction packed in the low 4 bits */
            bmv.y -= (int)((uint32_t)bcost << 30) >> 30;
            bcost &= ~15;
        }
        while (--i && in_range(c, bmv.x, bmv.y));
        bcost >>= 4;
        break;
    }
    case ME_HEX:
        hex_refine(c, &bmv, &bcost, merange);
        break;
    case ME_UMH:
        /* pmv = pmv.roundToFPel() before the switch (motion.cpp:814) */
        umh_search(c, &bmv, &bcost, merange, (pmvx + 2) >> 2, (pmvy + 2) >> 2, c->h, c->w == 64 && c->h == 64, mvc, numMvc);
        break;
    case ME_STAR:
    {
        int bPointNr = 0, bDistance = 0;
        star_pattern(c, &bmv, &bcost, &bPointNr, &bDistance, 3, merange);
        int done = 0;
        if (bDistance == 1)
        {
            if (bPointNr)
            {
                const int saved = bcost;
                const mv_t m1 = { bmv.x + kOffsets[(bPointNr - 1) * 2].x, bmv.y + kOffsets[(bPointNr - 1) * 2].y };
                const mv_t m2 = { bmv.x + kOffsets[(bPointNr - 1) * 2 + 1].x, bmv.y + kOffsets[(bPointNr - 1) * 2 + 1].y };
                if (in_range(c, m1.x, m1.y)) { const int cost = cost_mv(c, m1.x, m1.y); if (cost < bcost) { bcost = cost; bmv = m1; } }
                if (in_range(c, m2.x, m2.y)) { const int cost = cost_mv(c, m2.x, m2.y); if (cost < bcost) { bcost = cost; bmv = m2; } }
                if (bcost == saved) done = 1;
            }
            else done = 1;
        }
        if (done) break;
        const int RasterDistance = 5;
        if (bDistance > RasterDistance)
        {
            for (int ty = c->mvmin.y; ty <= c->mvmax.y; ty += RasterDistance)
                for (int tx = c->mvmin.x; tx <= c->mvmax.x; tx += RasterDistance)
                {
                    if (tx + RasterDistance * 3 <= c->mvmax.x)
                    {
                        for (int k = 0; k < 4; k++, tx += (k < 4 ? RasterDistance : 0))
                        {
                            /* the fourth candidate of every sad_x4 group is priced with mvcost(tmv << 3) (:1196) */
                            const int cost = sad_at(c, tx, ty) + (k == 3 ? mvcost_q(c, tx * 8, ty * 8) : mvcost_q(c, tx * 4, ty * 4));
                            if (cost < bcost) { bcost = cost; bmv.x = tx; bmv.y = ty; }
                        }
                    }
                    else
                    {
                        const int cost = cost_mv(c, tx, ty);
                        if (cost < bcost) { bcost = cost; bmv.x = tx; bmv.y = ty; }
                    }
                }
        }
        while (bDistance > 0)
        {
            bDistance = 0;
            bPointNr = 0;
            star_pattern(c, &bmv, &bcost, &bPointNr, &bDistance, 32, merange);
            if (bDistance == 1)
            {
                if (!bPointNr) break;
                const mv_t m1 = { bmv.x + kOffsets[(bPointNr - 1) * 2].x, bmv.y + kOffsets[(bPointNr - 1) * 2].y };
                const mv_t m2 = { bmv.x + kOffsets[(bPointNr - 1) * 2 + 1].x, bmv.y + kOffsets[(bPointNr - 1) * 2 + 1].y };
                if (in_range(c, m1.x, m1.y)) { const int cost = cost_mv(c, m1.x, m1.y); if (cost < bcost) { bcost = cost; bmv = m1; } }
                if (in_range(c, m2.x, m2.y)) { const int cost = cost_mv(c, m2.x, m2.y); if (cost < bcost) { bcost = cost; bmv = m2; } }
                break;
            }
        }
        break;
    }
    case ME_SEA:
        if (sea_search(c, &bmv, &bcost, merange)) return -1;
        break;
    case ME_FULL:
        for (int ty = c->mvmin.y; ty <= c->mvmax.y; ty++)
            for (int tx = c->mvmin.x; tx <= c->mvmax.x; tx++)
            {
                const int cost = cost_mv(c, tx, ty);
                if (cost < bcost) { bcost = cost; bmv.x = tx; bmv.y = ty; }
            }
        break;
    default:
        return -1;
    }

    int bx, by;
    if (bprecost < bcost) { bx = bestprex; by = bestprey; bcost = bprecost; }
    else { bx = bmv.x * 4; by = bmv.y * 4; }
    const workload_t wl = kWorkload[subme];
    if (!bcost)
        bcost = mvcost_q(c, bx, by);
    else if (c->lowres[0])
    {
        /* motion.cpp:1471-1503: one half-pel round on SAD, re-measure on SATD, one quarter-pel round on SATD */
        int bdir = 0;
        for (int i = 1; i <= wl.hpel_dirs; i++)
        {
            const int qx = bx + kSquare1[i].x * 2, qy = by + kSquare1[i].y * 2;
            if ((qy < qminy) | (qy > qmaxy)) continue;
            const int cost = lowres_qpel_cost(c, qx, qy, 0) + mvcost_q(c, qx, qy);
            if (cost < bcost) { bcost = cost; bdir = i; }
        }
        bx += kSquare1[bdir].x * 2; by += kSquare1[bdir].y * 2;
        bcost = lowres_qpel_cost(c, bx, by, 1) + mvcost_q(c, bx, by);
        bdir = 0;
        for (int i = 1; i <= wl.qpel_dirs; i++)
        {
            const int qx = bx + kSquare1[i].x, qy = by + kSquare1[i].y;
            if ((qy < qminy) | (qy > qmaxy)) continue;
            const int cost = lowres_qpel_cost(c, qx, qy, 1) + mvcost_q(c, qx, qy);
            if (cost < bcost) { bcost = cost; bdir = i; }
        }
        bx += kSquare1[bdir].x; by += kSquare1[bdir].y;
    }
    else
    {
        int hpelSatd = wl.hpel_satd;
        if (hpelSatd) bcost = subpel_compare(c, bx, by, 1) + mvcost_q(c, bx, by);
        for (int iter = 0; iter < wl.hpel_iters; iter++)
        {
            int bdir = 0;
            for (int i = 1; i <= wl.hpel_dirs; i++)
            {
                const int qx = bx + kSquare1[i].x * 2, qy = by + kSquare1[i].y * 2;
                if ((qy < qminy) | (qy > qmaxy)) continue;
                const int cost = subpel_compare(c, qx, qy, hpelSatd) + mvcost_q(c, qx, qy);
                if (cost < bcost) { bcost = cost; bdir = i; }
            }
            if (bdir) { bx += kSquare1[bdir].x * 2; by += kSquare1[bdir].y * 2; }
            else break;
        }
        if (!hpelSatd) bcost = subpel_compare(c, bx, by, 1) + mvcost_q(c, bx, by);
        for (int iter = 0; iter < wl.qpel_iters; iter++)
        {
            int bdir = 0;
            for (int i = 1; i <= wl.qpel_dirs; i++)
            {
                const int qx = bx + kSquare1[i].x, qy = by + kSquare1[i].y;
                if ((qy < qminy) | (qy > qmaxy)) continue;
                const int cost = subpel_compare(c, qx, qy, 1) + mvcost_q(c, qx, qy);
                if (cost < bcost) { bcost = cost; bdir = i; }
            }
            if (bdir) { bx += kSquare1[bdir].x; by += kSquare1[bdir].y; }
            else break;
        }
    }
    *outQx = bx; *outQy = by;
    return bcost;
}

/* fenc / fref: pixel (0,0) of padded planes of equal stride.  cost: uint16 table, cost[q] for q in [-qoff, qoff]
 * (pointer to the q = 0 entry is cost + qoff).  Runs every job; returns 0, or -1 for an unsupported method / PU size. */
int EXPORT(x265oracle_motion_estimate_mvc)(const pixel* fenc, const pixel* fref, intptr_t stride, int method, int subme, int merange,
                                           const uint16_t* cost, int qoff, int mvminx, int mvminy, int mvmaxx, int mvmaxy,
                                           me_job* jobs, int njobs, int nthreads, const int32_t* mvc, const int32_t* numMvc);

int EXPORT(x265oracle_motion_estimate)(const pixel* fenc, const pixel* fref, intptr_t stride, int method, int subme, int merange,
                                       const uint16_t* cost, int qoff, int mvminx, int mvminy, int mvmaxx, int mvmaxy,
                                       me_job* jobs, int njobs, int nthreads)
{
    return EXPORT(x265oracle_motion_estimate_mvc)(fenc, fref, stride, method, subme, merange, cost, qoff, mvminx, mvminy, mvmaxx, mvmaxy,
                                                  jobs, njobs, nthreads, NULL, NULL);
}

/* mvc: optional int32 [njobs][12][2] quarter-pel candidates, numMvc: int32 [njobs] (the reference passes at most 12, search.cpp:2094) */
int EXPORT(x265oracle_motion_estimate_mvc)(const pixel* fenc, const pixel* fref, intptr_t stride, int method, int subme, int merange,
                                           const uint16_t* cost, int qoff, int mvminx, int mvminy, int mvmaxx, int mvmaxy,
                                           me_job* jobs, int njobs, int nthreads, const int32_t* mvc, const int32_t* numMvc)
{
    static x265hip_EncoderPrimitives prim;
    static int ready = 0;
    EXPORT(x265oracle_prims_once)(&prim, &ready);
    if (subme < 0 || subme > 7) return -1;
    int rc = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < njobs; i++)
    {
        me_job* j = &jobs[i];
        int part = -1;
        for (int k = 1; k < 25; k++) if (kPuDims[k][0] == j->w && kPuDims[k][1] == j->h) part = k;
        if (part < 0) { rc = -1; continue; }
        me_ctx c;
        c.pu = &prim.pu[part];
        c.stride = stride; c.w = j->w; c.h = j->h;
        c.fref = fref + j->px + (intptr_t)j->py * stride;
        c.pu->copy_pp(c.fenc, 64, fenc + j->px + (intptr_t)j->py * stride, stride);
        c.cost = cost + qoff;
        c.mvpx = j->qmvpx; c.mvpy = j->qmvpy;
        c.mvmin.x = mvminx; c.mvmin.y = mvminy; c.mvmax.x = mvmaxx; c.mvmax.y = mvmaxy;
        c.lowres[0] = c.lowres[1] = c.lowres[2] = c.lowres[3] = NULL;
        int qx = 0, qy = 0;
        const int cst = motion_estimate_one(&c, method, subme, merange, mvc ? mvc + (size_t)i * 24 : NULL, (mvc && numMvc) ? numMvc[i] : 0, &qx, &qy);
        if (cst < 0) { rc = -1; continue; }
        j->out_cost = cst; j->out_qmvx = qx; j->out_qmvy = qy;
    }
    return rc;
}


/* lowres.h:66-93 (ReferencePlanes::lowresMC): the prediction itself - a pointer into a phase plane, or the average of two in buf */
static const pixel* lowres_mc(const me_ctx* c, int qx, int qy, pixel* buf, intptr_t* outStride)
{
    if ((qx | qy) & 1)
    {
        const int hpelA = (qy & 2) | ((qx & 2) >> 1);
        const pixel* frefA = c->lowres[hpelA] + (qx >> 2) + (intptr_t)(qy >> 2) * c->stride;
        const int qmvx = qx + (qx & 1), qmvy = qy + (qy & 1);
        const int hpelB = (qmvy & 2) | ((qmvx & 2) >> 1);
        const pixel* frefB = c->lowres[hpelB] + (qmvx >> 2) + (intptr_t)(qmvy >> 2) * c->stride;
        c->pu->pixelavg_pp[0](buf, *outStride, frefA, c->stride, frefB, c->stride, 32);
        return buf;
    }
    *outStride = c->stride;
    const int hpel = (qy & 2) | ((qx & 2) >> 1);
    return c->lowres[hpel] + (qx >> 2) + (intptr_t)(qy >> 2) * c->stride;
}

/* CostEstimateGroup::estimateFrameCost + estimateCUCost (slicetype.cpp:3115-3213, 3216-3388; reverse raster order :3189-3198) for
 * a P picture (refs1 == NULL: b == p1, one list, intra competes) or a B picture (two lists + the two bi-directional candidates,
 * :3322-3343); no HME, no weighted reference.  Every 8x8 block of the half-resolution picture: per searched list the mvs of the
 * already finished right / lower neighbours are tried as predictors (SATD at lowresMC, :3284-3301, which also yields the B
 * picture's skipCost), HEX search with merange 16 and subpelRefine 1 against the reference's four phase planes (:3307), skip
 * shortcut (:3311-3315); then + lowresPenalty and, for P, intra wins if cheaper (:3346-3356); frame sums over the non-edge blocks.
 *   cur: pixel (0,0) of the current picture's plane 0; refs0 / refs1: pixel (0,0) of the four planes of the list-0 / list-1
 *   reference (same stride); doSearch[i] == 0 keeps list i's mvs / mvCosts as given (estimateFrameCost's bDoSearch, :3125-3127).
 *   intraCost: LookaheadTLD::lowresIntraEstimate's output for the current picture; invQscale: fenc->invQscaleFactor or NULL.
 * Outputs: mvs0/1 int32 [n][2] (quarter-pel), mvCosts0/1 int32 [n], lowresCosts uint16 [n], rowSatds int32 [heightInCU],
 * frame int64 [4] = { costEst as accumulated, costEstAq, intraMbs, the returned score (costEst * 100 / (130 + bFrameBias) for B) }.
 * Serial by construction (each block needs its neighbours' mvs). */
int EXPORT(x265oracle_lowres_cost_wp)(const pixel* cur, const pixel* const* refs0, const pixel* const* refs1, intptr_t stride,
                                      int widthInCU, int heightInCU, const uint16_t* cost, int qoff, const int32_t* intraCost,
                                      const int32_t* invQscale, const int* doSearch, int bFrameBias,
                                      int32_t* mvs0, int32_t* mvCosts0, int32_t* mvs1, int32_t* mvCosts1,
                                      uint16_t* lowresCosts, int32_t* rowSatds, int64_t* frame, const pixel* const* refs0Bi);

int EXPORT(x265oracle_lowres_cost)(const pixel* cur, const pixel* const* refs0, const pixel* const* refs1, intptr_t stride,
                                   int widthInCU, int heightInCU, const uint16_t* cost, int qoff, const int32_t* intraCost,
                                   const int32_t* invQscale, const int* doSearch, int bFrameBias,
                                   int32_t* mvs0, int32_t* mvCosts0, int32_t* mvs1, int32_t* mvCosts1,
                                   uint16_t* lowresCosts, int32_t* rowSatds, int64_t* frame)
{
    return EXPORT(x265oracle_lowres_cost_wp)(cur, refs0, refs1, stride, widthInCU, heightInCU, cost, qoff, intraCost, invQscale, doSearch, bFrameBias,
                                             mvs0, mvCosts0, mvs1, mvCosts1, lowresCosts, rowSatds, frame, NULL);
}

/* --weightp: refs0 are then the WEIGHTED list-0 planes (what the list-0 search, its predictor candidates and the skip cost see,
 * slicetype.cpp:3222,3267) and refs0Bi the unweighted ones, which the two bi-directional candidates keep using (:3328); NULL = refs0 */
int EXPORT(x265oracle_lowres_cost_wp)(const pixel* cur, const pixel* const* refs0, const pixel* const* refs1, intptr_t stride,
                                      int widthInCU, int heightInCU, const uint16_t* cost, int qoff, const int32_t* intraCost,
                                      const int32_t* invQscale, const int* doSearch, int bFrameBias,
                                      int32_t* mvs0, int32_t* mvCosts0, int32_t* mvs1, int32_t* mvCosts1,
                                      uint16_t* lowresCosts, int32_t* rowSatds, int64_t* frame, const pixel* const* refs0Bi)
{
    if (!refs0Bi) refs0Bi = refs0;
    /* the seam tests call this from several encoder threads at once: the table is filled exactly once */
    static x265hip_EncoderPrimitives prim;
    static int ready = 0;
    if (!__atomic_load_n(&ready, __ATOMIC_ACQUIRE))
    {
        static int lock = 0;
        while (__atomic_exchange_n(&lock, 1, __ATOMIC_ACQUIRE)) { }
        if (!ready) { EXPORT(x265oracle_setup_primitives)(&prim); __atomic_store_n(&ready, 1, __ATOMIC_RELEASE); }
        __atomic_store_n(&lock, 0, __ATOMIC_RELEASE);
    }
    int part = -1;
    for (int k = 0; k < 25; k++) if (kPuDims[k][0] == 8 && kPuDims[k][1] == 8) part = k;
    const int bBidir = refs1 != NULL;
    const pixel* const* refs[2] = { refs0, refs1 };
    int32_t* mvsL[2] = { mvs0, mvs1 };
    int32_t* mvCostsL[2] = { mvCosts0, mvCosts1 };
    const int cuSize = 8, lowresPenalty = 4, merange = 16;
    int64_t costEst = 0, costEstAq = 0, intraMbs = 0;
    for (int cuY = heightInCU - 1; cuY >= 0; cuY--)
    {
        const int lastRow = cuY == heightInCU - 1;
        rowSatds[cuY] = 0;
        for (int cuX = widthInCU - 1; cuX >= 0; cuX--)
        {
            const int cuXY = cuX + cuY * widthInCU;
            const intptr_t pelOffset = cuSize * cuX + cuSize * cuY * stride;
            me_ctx c;
            c.pu = &prim.pu[part];
            c.stride = stride; c.w = 8; c.h = 8;
            c.pu->copy_pp(c.fenc, 64, cur + pelOffset, stride);
            c.cost = cost + qoff;
            c.mvmin.x = -cuX * cuSize - 8; c.mvmin.y = -cuY * cuSize - 8;
            c.mvmax.x = (widthInCU - cuX - 1) * cuSize + 8; c.mvmax.y = (heightInCU - cuY - 1) * cuSize + 8;
            int bcost = 1 << 28, listused = 0;                        /* MotionEstimate::COST_MAX (motion.h:65) */
            for (int i = 0; i < 1 + bBidir; i++)
            {
                int32_t* fencMV = mvsL[i] + 2 * cuXY;
                int skipCost = 0x7fffffff;
                if (!doSearch[i])
                {
                    if (mvCostsL[i][cuXY] < bcost) { bcost = mvCostsL[i][cuXY]; listused = i + 1; }
                    continue;
                }
                for (int k = 0; k < 4; k++) c.lowres[k] = refs[i][k] + pelOffset;
                c.fref = c.lowres[0];
                /* reverse-order mv prediction (:3266-3282) */
                int numc = 0, mvc[4][2];
#define MVC(IDX) do { mvc[numc][0] = fencMV[2 * (IDX)]; mvc[numc][1] = fencMV[2 * (IDX) + 1]; numc++; } while (0)
                if (cuX < widthInCU - 1) MVC(1);
                if (!lastRow)
                {
                    MVC(widthInCU);
                    if (cuX > 0) MVC(widthInCU - 1);
                    if (cuX < widthInCU - 1) MVC(widthInCU + 1);
                }
#undef MVC
                int mvpx = 0, mvpy = 0;
                if (numc)
                {
                    int mvpcost = 1 << 28;                          /* MotionEstimate::COST_MAX */
                    for (int idx = 0; idx < numc; idx++)
                    {
                        /* bufSATD over lowresMC's prediction = the SATD flavour of lowresQPelCost */
                        const int cst = lowres_qpel_cost(&c, mvc[idx][0], mvc[idx][1], 1);
                        if (cst < mvpcost) { mvpcost = cst; mvpx = mvc[idx][0]; mvpy = mvc[idx][1]; }
                        /* :3297-3299: while the best predictor so far is the zero mv, remember the cost just measured */
                        if (!(mvpx | mvpy) && bBidir) skipCost = cst;
                    }
                }
                c.mvpx = mvpx; c.mvpy = mvpy;
                int qx = 0, qy = 0;
                int fencCost = motion_estimate_one(&c, ME_HEX, 1, merange, NULL, 0, &qx, &qy);
                if (skipCost < 64 && skipCost < fencCost && bBidir) { fencCost = skipCost; qx = qy = 0; }
                fencMV[0] = qx; fencMV[1] = qy;
                mvCostsL[i][cuXY] = fencCost;
                if (fencCost < bcost) { bcost = fencCost; listused = i + 1; }
            }
            if (bBidir)
            {
                /* avg(l0-mv, l1-mv) candidate, then the co-located one (:3322-3343) */
                pixel buf0[64], buf1[64], avg[64];
                intptr_t st0 = 8, st1 = 8;
                for (int k = 0; k < 4; k++) c.lowres[k] = refs0Bi[k] + pelOffset;
                const pixel* src0 = lowres_mc(&c, mvs0[2 * cuXY], mvs0[2 * cuXY + 1], buf0, &st0);
                for (int k = 0; k < 4; k++) c.lowres[k] = refs1[k] + pelOffset;
                const pixel* src1 = lowres_mc(&c, mvs1[2 * cuXY], mvs1[2 * cuXY + 1], buf1, &st1);
                c.pu->pixelavg_pp[0](avg, 8, src0, st0, src1, st1, 32);
                int bicost = c.pu->satd(c.fenc, 64, avg, 8);
                if (bicost < bcost) { bcost = bicost; listused = 3; }
                c.pu->pixelavg_pp[0](avg, 8, refs0Bi[0] + pelOffset, stride, refs1[0] + pelOffset, stride, 32);
                bicost = c.pu->satd(c.fenc, 64, avg, 8);
                if (bicost < bcost) { bcost = bicost; listused = 3; }
                bcost += lowresPenalty;
            }
            else
            {
                bcost += lowresPenalty;
                if (intraCost[cuXY] < bcost) { bcost = intraCost[cuXY]; listused = 0; }
            }
            const int bFrameScoreCU = (cuX > 0 && cuX < widthInCU - 1 && cuY > 0 && cuY < heightInCU - 1) || widthInCU <= 2 || heightInCU <= 2;
            const int bcostAq = (bFrameScoreCU && invQscale) ? ((bcost * invQscale[cuXY] + 128) >> 8) : bcost;
            if (bFrameScoreCU)
            {
                costEst += bcost; costEstAq += bcostAq;
                if (!listused && !bBidir) intraMbs++;
            }
            rowSatds[cuY] += bcostAq;
            lowresCosts[cuXY] = (uint16_t)((bcost < 0x3fff ? bcost : 0x3fff) | (listused << 14));
        }
    }
    frame[0] = costEst; frame[1] = costEstAq; frame[2] = intraMbs;
    frame[3] = bBidir ? costEst * 100 / (130 + bFrameBias) : costEst;
    return 0;
}
