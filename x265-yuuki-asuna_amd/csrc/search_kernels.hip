// search_kernels.hip - MotionEstimate::motionEstimate for a list of PUs on gfx950 (SURVEY.md section 8(f) item 1:
// the search DRIVERS as device code, next to the pixels, instead of a host loop issuing one SAD per candidate).
//
// Reference semantics (source/encoder/motion.cpp): predictor / zero-mv start :772-799 (bprecost is the SAD at the clipped
// quarter-pel predictor WITHOUT its mv cost), integer patterns DIA :827-850, HEX :852-948 (incl. square refine), STAR
// :1138-1240 with StarPatternSearch :362-604 (point numbering, early exit after 3 / 32 idle rounds, two-point refinement,
// raster refinement whose fourth candidate is priced with mvcost(tmv << 3), :1196) and FULL :1397-1445; predictor-vs-search
// choice :1452-1458; zero-residual shortcut :1464-1469; sub-pel refinement :1508-1561 over subpelCompare :1571-1613
// (luma_hpp / luma_vpp / luma_hvpp + sad / satd); UMH :946-1130.  mv costs come from the caller's table, indexed by the quarter-pel
// difference to the predictor (bitcost.h:45).  X265_SEA (:1241-1395) is sea_search() below, on the block-sum planes of x265hip_sea_integral.
//
// Mapping: a lane GROUP per PU - a DPP quad (4 lanes) for PUs of up to 8 tiles of 4x4 (8x8, 8x4, 16x8 ...), a 16-lane row up
// to 32 tiles (16x16 ... 32x16), the whole wavefront above (32x32 ... 64x64: 4 tiles per lane).  The source tiles stay packed
// in registers; a candidate is scored by every lane of the group on its tiles (v_sad_u8 / v_sad_u16 against unaligned reference
// dwords, or packed dot4/dot2 interpolation + 4x4 Hadamard for fractional positions, tile_interp.h) and summed across the group
// with DPP, so every lane of a group holds the same score and runs the reference's serial decision code redundantly; groups of
// one wavefront follow their own decisions through ordinary SIMT divergence (the instruction stream is the same, only the
// motion vectors differ).  A wavefront takes 16 consecutive jobs: the small ones together, the medium ones four at a time; the
// large ones are queued and run by a second kernel with a wavefront each - any job order is correct.
#include "pu_eval.h"

namespace x265hip {

struct SearchArgs
{
    const uint8_t* fenc; long fencStrideB;
    const uint8_t* fref; long frefStrideB;
    const uint16_t* cost;            // cost[q]: pointer to the q = 0 entry
    x265hip_me_search_job* jobs; int njobs;
    int depth, method, subme, merange;
    int mvminx, mvminy, mvmaxx, mvmaxy;
    const int32_t* mvc; const int32_t* numMvc;      // optional [njobs][12][2] quarter-pel candidates / [njobs]
    int* largeCount; int* largeQueue;               // jobs of more than 32 tiles, collected by the first kernel for the second
    const uint32_t* integral[12];                   // X265_SEA: block-sum planes of the reference, entry of sample (0,0)
};

// Point i of one StarPatternSearch round (motion.cpp:362-604) as offsets from the round's origin, in the reference's
// evaluation order, with its point number and distance.  dist 1: the 4 axis neighbours; dist 2..8: 8 points (axis at dist,
// diagonals at dist / 2); dist >= 16: 16 points on the diamond of radius dist.
__constant__ signed char kStarBx[8] = { 0, -1, 1, -1, 1, -1, 1, 0 }, kStarBy[8] = { -1, -1, -1, 0, 0, 1, 1, 1 };
__constant__ unsigned char kStarBhalf[8] = { 0, 1, 1, 0, 0, 1, 1, 0 }, kStarBpt[8] = { 2, 1, 3, 4, 5, 6, 8, 7 };
__constant__ signed char kStarAx[4] = { 0, -1, 1, 0 }, kStarAy[4] = { -1, 0, 0, 1 };
__constant__ unsigned char kStarApt[4] = { 2, 4, 5, 7 };

__device__ __forceinline__ void star_point(int dist, int i, int& dx, int& dy, int& pt, int& d)
{
    if (dist == 1) { dx = kStarAx[i]; dy = kStarAy[i]; pt = kStarApt[i]; d = 1; return; }
    if (dist <= 8)
    {
        const int m = kStarBhalf[i] ? dist >> 1 : dist;
        dx = kStarBx[i] * m; dy = kStarBy[i] * m; pt = kStarBpt[i]; d = m;
        return;
    }
    pt = 0; d = dist;
    if (i < 4) { dx = kStarAx[i] * dist; dy = kStarAy[i] * dist; return; }
    const int step = (dist >> 2) * (i >> 2), q = i & 3;
    dx = (q & 1) ? step : -step;
    dy = q < 2 ? -dist + step : dist - step;
}

// motion.cpp:362-604.  Inside the search bounds a round is scored in groups (the reference's sad_x4 calls, same order) with
// the loads of a group in flight together; a round that touches the bounds is scored point by point with the reference's
// directional border tests.
template <typename Px, int G, int T>
__device__ __forceinline__ void star_pattern(const PuEval<Px, G, T>& c, SMv& bmv, int& bcost, int& bPointNr, int& bDistance, int earlyExitIters, int merange)
{
    const SMv omv = bmv;
    int rounds = 0;
    for (int dist = 1; dist <= 8 || dist <= (int)(int16_t)merange; dist <<= 1)
    {
        const int npts = dist == 1 ? 4 : (dist <= 8 ? 8 : 16);
        const bool inside = omv.y - dist >= c.mvmin.y && omv.x - dist >= c.mvmin.x && omv.x + dist <= c.mvmax.x && omv.y + dist <= c.mvmax.y;
        const int saved = bcost;
        if (inside)
        {
            constexpr int NB = 4;                             // the reference's sad_x4 groups, same order
#pragma unroll 1
            for (int base = 0; base < npts; base += NB)
            {
                int mxs[NB], mys[NB], pts[NB], ds[NB], cs[NB];
#pragma unroll
                for (int n = 0; n < NB; n++) { int dx, dy; star_point(dist, base + n, dx, dy, pts[n], ds[n]); mxs[n] = omv.x + dx; mys[n] = omv.y + dy; }
                c.template cost_mv_n<NB>(mxs, mys, cs);
#pragma unroll
                for (int n = 0; n < NB; n++)
                    if (cs[n] < bcost) { bcost = cs[n]; bmv.x = mxs[n]; bmv.y = mys[n]; bPointNr = pts[n]; bDistance = ds[n]; }
            }
        }
        else
        {
            for (int i = 0; i < npts; i++)
            {
                int dx, dy, pt, d;
                star_point(dist, i, dx, dy, pt, d);
                const int mx = omv.x + dx, my = omv.y + dy;
                // the reference tests a point only against the bound(s) it moves towards
                if ((dx < 0 && mx < c.mvmin.x) || (dx > 0 && mx > c.mvmax.x) || (dy < 0 && my < c.mvmin.y) || (dy > 0 && my > c.mvmax.y)) continue;
                const int cost = c.cost_mv(mx, my);
                if (cost < bcost) { bcost = cost; bmv.x = mx; bmv.y = my; bPointNr = pt; bDistance = d; }
            }
        }
        if (bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
}

// X265_UMH_SEARCH (motion.cpp:946-1130): predictor diamonds, early termination on the SAD_THRESH scales (:61, :123-150), cross and
// octagon probes, optional range adaptation from the mv candidates (:982-1040), the 16-point hexagon grid, then the radius-2 hexagon
// refinement of X265_HEX_SEARCH.  Upstream behaviour kept: COST_MV_X4 only tests a candidate's y against the search range (:295-302);
// the hexagon grid's fast path tests omv.y + dy with the UNscaled offset (:1087).
__constant__ signed char kUmhHexX[16] = { 0, 0, -2, 2, -4, 4, -4, 4, -4, 4, -4, 4, -4, 4, -2, 2 };
__constant__ signed char kUmhHexY[16] = { -4, 4, -3, -3, -2, -2, -1, -1, 0, 0, 1, 1, 2, 2, 3, 3 };
__constant__ unsigned char kUmhRangeMul[4][4] = { { 3, 3, 4, 4 }, { 3, 4, 4, 4 }, { 4, 4, 4, 5 }, { 4, 4, 5, 6 } };

template <typename Px, int G, int T>
__device__ __forceinline__ void umh_search(const PuEval<Px, G, T>& c, SMv& bmv, int& bcost, int merange, const int pmvx, const int pmvy,
                                           const int h, const bool is64, const int32_t* mvc, const int numMvc)
{
    SMv omv = bmv;
    auto thresh = [&](int v) { return bcost < ((v >> 4) * ((h * h) >> 4)); };
    auto x4 = [&](int d0x, int d0y, int d1x, int d1y, int d2x, int d2y, int d3x, int d3y)
    {
        const int mxs[4] = { omv.x + d0x, omv.x + d1x, omv.x + d2x, omv.x + d3x }, mys[4] = { omv.y + d0y, omv.y + d1y, omv.y + d2y, omv.y + d3y };
        int cs[4];
        c.template cost_mv_n<4>(mxs, mys, cs);
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((mys[k] >= c.mvmin.y) & (mys[k] <= c.mvmax.y))
                if (cs[k] < bcost) { bcost = cs[k]; bmv.x = mxs[k]; bmv.y = mys[k]; }
    };
    auto one = [&](int mx, int my) { const int cost = c.cost_mv(mx, my); if (cost < bcost) { bcost = cost; bmv.x = mx; bmv.y = my; } };
    auto dia1 = [&](int mx, int my) { omv.x = mx; omv.y = my; x4(0, -1, 0, 1, -1, 0, 1, 0); };
    auto cross = [&](int start, int xmax, int ymax)
    {
        int i = (int16_t)start;
        if (xmax <= min(c.mvmax.x - omv.x, omv.x - c.mvmin.x))
            for (; i < xmax - 2; i += 4) x4(i, 0, -i, 0, i + 2, 0, -i - 2, 0);
        for (; i < xmax; i += 2)
        {
            if (omv.x + i <= c.mvmax.x) one(omv.x + i, omv.y);
            if (omv.x - i >= c.mvmin.x) one(omv.x - i, omv.y);
        }
        i = (int16_t)start;
        if (ymax <= min(c.mvmax.y - omv.y, omv.y - c.mvmin.y))
            for (; i < ymax - 2; i += 4) x4(0, i, 0, -i, 0, i + 2, 0, -i - 2);
        for (; i < ymax; i += 2)
        {
            if (omv.y + i <= c.mvmax.y) one(omv.x, omv.y + i);
            if (omv.y - i >= c.mvmin.y) one(omv.x, omv.y - i);
        }
    };
    int crossStart = 1;
    const int ucost1 = bcost;
    dia1(pmvx, pmvy);
    if (pmvx | pmvy) dia1(0, 0);
    const int ucost2 = bcost;
    if ((bmv.x | bmv.y) && (bmv.x != pmvx || bmv.y != pmvy)) dia1(bmv.x, bmv.y);
    if (bcost == ucost2) crossStart = 3;
    omv = bmv;
    if (bcost == ucost2 && thresh(2000))
    {
        x4(0, -2, -1, -1, 1, -1, -2, 0);
        x4(2, 0, -1, 1, 1, 1, 0, 2);
        if (bcost == ucost1 && thresh(500)) return;
        if (bcost == ucost2)
        {
            const int range = (int16_t)((merange >> 1) | 1);
            cross(3, range, range);
            x4(-1, -2, 1, -2, -2, -1, 2, -1);
            x4(-2, 1, 2, 1, -1, 2, 1, 2);
            if (bcost == ucost2) return;
            crossStart = (int16_t)(range + 2);
        }
    }
    if (numMvc)
    {
        int mvd, denom = 1;
        if (numMvc == 1)
            mvd = is64 ? 25 : abs(c.mvpx - mvc[0]) + abs(c.mvpy - mvc[1]);
        else
        {
            denom = numMvc - 1;
            mvd = 0;
            if (!is64) { mvd = abs(c.mvpx - mvc[0]) + abs(c.mvpy - mvc[1]); denom++; }
            for (int i = 0; i < numMvc - 1; i++) mvd += abs(mvc[2 * i] - mvc[2 * i + 2]) + abs(mvc[2 * i + 1] - mvc[2 * i + 3]);
        }
        const int sadCtx = thresh(1000) ? 0 : (thresh(2000) ? 1 : (thresh(4000) ? 2 : 3));
        const int mvdCtx = mvd < 10 * denom ? 0 : (mvd < 20 * denom ? 1 : (mvd < 40 * denom ? 2 : 3));
        merange = (merange * kUmhRangeMul[mvdCtx][sadCtx]) >> 2;
    }
    cross(crossStart, merange, merange >> 1);
    x4(-2, -2, -2, 2, 2, -2, 2, 2);
    omv = bmv;
    int i = 1;
    do
    {
        const int m = min(min(c.mvmax.x - omv.x, omv.x - c.mvmin.x), min(c.mvmax.y - omv.y, omv.y - c.mvmin.y));
        if (4 * i > m)
        {
            for (int j = 0; j < 16; j++)
            {
                const int mx = omv.x + kUmhHexX[j] * i, my = omv.y + kUmhHexY[j] * i;
                if (c.in_range(mx, my)) one(mx, my);
            }
        }
        else
        {
            int dir = 0;
#pragma unroll 1
            for (int g4 = 0; g4 < 16; g4 += 4)                        // the reference's four sad_x4 groups, same order
            {
                int mxs[4], mys[4], cs[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { mxs[k] = omv.x + kUmhHexX[g4 + k] * i; mys[k] = omv.y + kUmhHexY[g4 + k] * i; }
                c.template cost_mv_n<4>(mxs, mys, cs);
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const int dy = kUmhHexY[g4 + k];
                    if ((omv.y + dy >= c.mvmin.y) & (omv.y + dy <= c.mvmax.y))
                        if (cs[k] < bcost) { bcost = cs[k]; dir = (int16_t)(kUmhHexX[g4 + k] * 16 + (dy & 15)); }
                }
            }
            if (dir)
            {
                bmv.x = omv.x + i * (dir >> 4);
                bmv.y = omv.y + i * ((int)((uint32_t)dir << 28) >> 28);
            }
        }
    }
    while (++i <= (merange >> 2));
    if (c.in_range(bmv.x, bmv.y)) hex_search<Px, G, T>(c, bmv, bcost, merange);
}

// One PU per lane group: `job` is the group's job (every lane of the group passes the same value).
constexpr int kSeaListLen = 128;      // candidates per SEA row: 2 * merange + 4 <= 128

// the group's lanes that voted true, bit i = lane i of the group (the lanes of a group are always active together)
template <int G> __device__ __forceinline__ unsigned long long group_ballot(bool v)
{
    const unsigned long long m = __ballot(v);
    if (G == 64) return m;
    return (m >> (threadIdx.x & 63 & ~(G - 1))) & ((1ull << G) - 1);
}

// X265_SEA (motion.cpp:1241-1395): row by row, the ADS filter - sum over the PU's 1 / 2 / 4 DC terms of |source DC - block sum of
// the reference| plus the x cost, against the best cost so far - selects the candidates whose SAD is then measured.  The G lanes
// of the group take G consecutive candidates of the row through the filter at a time; the survivors are measured in the
// reference's order and grouping (threes through sad_x3 with a y-less cost, the 0-2 left over with the ordinary cost), because
// the two cost expressions differ and the result depends on which one a candidate meets.  Which ADS variant, which plane and
// which offsets a PU size gets - including the sizes where they do not match the DC blocks - follows :1258-1364 literally.
// Returns false for the PU sizes whose DC terms lie outside the PU.
template <typename Px, int G, int T>
__device__ __forceinline__ bool sea_search(const PuEval<Px, G, T>& c, const SearchArgs& a, const x265hip_me_search_job& jb, SMv& bmv, int& bcost,
                                           const int merange)
{
    constexpr int BPP = sizeof(Px);
    const int gl = threadIdx.x & (G - 1);
    const int w = jb.w, h = jb.h;
    if ((w == 8 && h == 4) || (w == 4 && h == 8) || (w == 32 && h == 8) || (w == 8 && h == 32)) return false;
    const bool vertical = h == 2 * w, horizontal = w == 2 * h, square = w == h;
    const bool smallRect = (w == 16 && h == 12) || (w == 12 && h == 16) || (w == 16 && h == 4) || (w == 4 && h == 16);
    const bool asymVertical = !square && !vertical && w < h;
    const int deltaX = w <= 8 ? w : w >> 1, deltaY = h <= 8 ? h : h >> 1;
    int tw, th;                                                     // the block the source DCs are taken over (:1283-1303)
    if (vertical) { tw = w; th = h >> 1; }
    else if (horizontal) { tw = w >> 1; th = h; }
    else if (!square) { tw = smallRect ? w : w >> 1; th = smallRect ? h : h >> 1; }
    else { tw = w <= 8 ? w : w >> 1; th = w <= 8 ? h : h >> 1; }
    const int nAds = (square ? w <= 8 : smallRect) ? 1 : ((vertical || horizontal) ? 2 : 4);      // pixel.cpp:1105-1129
    int plane;                                                      // :1315-1347, keyed on deltaX / deltaY only
    switch (deltaX)
    {
    case 32: plane = (deltaY % 24 == 0) ? 1 : (deltaY == 8 ? 2 : 0); break;
    case 24: plane = 3; break;
    case 16: plane = (deltaY % 12 == 0) ? 5 : (deltaY == 4 ? 6 : 4); break;
    case 12: plane = 7; break;
    case 8: plane = deltaY == 32 ? 8 : 9; break;
    case 4: plane = deltaY == 16 ? 10 : 11; break;
    default: plane = 11; break;
    }
    // source DCs: every lane adds the tiles it holds to the blocks they fall into (sad_x4 against zeros, :1305-1311)
    int dc[4] = { 0, 0, 0, 0 };
    const int tilesX = w >> 2;
#pragma unroll
    for (int k = 0; k < T; k++)
    {
        if (!c.have[k]) continue;
        const int t = gl + G * k, ty = t / tilesX, x = (t - ty * tilesX) * 4, y = ty * 4;
        uint32_t sum = 0;
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int q = 0; q < BPP; q++) sum = sad_dw<Px>(c.src[k][r][q], 0u, sum);
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int ox = (q & 1) ? deltaX : 0, oy = (q & 2) ? deltaY : 0;
            if (x >= ox && x < ox + tw && y >= oy && y < oy + th) dc[q] += (int)sum;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) dc[q] = group_total<G>(dc[q]);
    // `delta` of the ads call: deltaY rows for squares, vertical and asymmetric-vertical PUs (:1349-1355), deltaY SAMPLES along the
    // row for the asymmetric-horizontal ones (missing from that list), deltaX samples for horizontal PUs (:1360-1361)
    const int ps = c.strideB / BPP;
    const bool rows = square || vertical || asymVertical;
    const int delta = horizontal ? deltaX : (rows ? deltaY * ps : deltaY);
    int off1, off2 = 0, off3 = 0, enc1;
    if (nAds == 4) { off1 = w >> 1; off2 = delta; off3 = delta + (w >> 1); enc1 = dc[1]; }
    else { off1 = delta; enc1 = vertical ? dc[2] : dc[1]; }           // :1357-1358
    const uint32_t* pl = a.integral[plane] + (long)jb.py * ps + jb.px;
    const int minX = max(bmv.x - merange, c.mvmin.x), minY = max(bmv.y - merange, c.mvmin.y);
    const int maxX = min(bmv.x + merange, c.mvmax.x), maxY = min(bmv.y + merange, c.mvmax.y);
    const int width = (maxX - minX + 3) & ~3;                       // up to three candidates beyond maxX are examined
    __shared__ short seaLists[256 / 4][kSeaListLen];                // one survivor list per lane group of the workgroup
    short* list = seaLists[(threadIdx.x / G) * (G / 4)];
    for (int ty = minY; ty <= maxY; ty++)
    {
        // three mv-cost expressions meet here: the filter adds the ordinary x cost; the threes add m_cost[4x - 2 qmvp.x] and no y
        // cost; the row's y cost is m_cost[y - 2 qmvp.y] << 2; the left-over candidates use the ordinary mvcost (:1247-1248,1369,304-306)
        const int ycost = (int)c.cost[ty - 2 * c.mvpy] << 2;
        if (bcost <= ycost) continue;
        const int thr = bcost - ycost;                              // the filter's threshold for the whole row
        int cur = thr;
        const uint32_t* row = pl + (long)ty * ps;
        // the filter: G x U consecutive candidates per step (independent plane reads, issued together); the survivors' column
        // numbers go to the group's list in LDS in ascending order - the reference's mvs[] scratch array
        constexpr int U = G == 4 ? 8 : (G == 16 ? 2 : 1);
        int n = 0;
#pragma unroll 1
        for (int i0 = 0; i0 < width; i0 += G * U)
        {
            bool pass[U];
#pragma unroll
            for (int u = 0; u < U; u++)
            {
                const int i = i0 + u * G + gl;
                pass[u] = false;
                if (i < width)
                {
                    const int x = minX + i;
                    const uint32_t* sp = row + x;
                    int ads = (int)c.cost[4 * x - c.mvpx] + abs(dc[0] - (int)sp[0]);
                    if (nAds >= 2) ads += abs(enc1 - (int)sp[off1]);
                    if (nAds == 4) ads += abs(dc[2] - (int)sp[off2]) + abs(dc[3] - (int)sp[off3]);
                    pass[u] = ads < thr;
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++)
            {
                const unsigned long long m = group_ballot<G>(pass[u]);
                if (pass[u]) list[n + __popcll(m & ((1ull << gl) - 1))] = (short)(minX + i0 + u * G + gl);
                n += __popcll(m);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");       // the list is read back by the lanes of the same wavefront
        // the threes (sad_x3 with the y-less cost), two at a time so that six SADs share a memory round trip
        int i = 0;
#pragma unroll 1
        for (; i + 6 <= n; i += 6)
        {
            int xs[6], ys[6], sads[6];
#pragma unroll
            for (int k = 0; k < 6; k++) { xs[k] = list[i + k]; ys[k] = ty; }
            c.template sad_n<6>(xs, ys, sads);
#pragma unroll
            for (int k = 0; k < 6; k++)
            {
                const int cost = sads[k] + (int)c.cost[4 * xs[k] - 2 * c.mvpx];
                if (cost < cur) { cur = cost; bmv.x = xs[k]; bmv.y = ty; }
            }
        }
        if (i + 3 <= n)
        {
            int xs[3], ys[3], sads[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { xs[k] = list[i + k]; ys[k] = ty; }
            c.template sad_n<3>(xs, ys, sads);
#pragma unroll
            for (int k = 0; k < 3; k++)
            {
                const int cost = sads[k] + (int)c.cost[4 * xs[k] - 2 * c.mvpx];
                if (cost < cur) { cur = cost; bmv.x = xs[k]; bmv.y = ty; }
            }
            i += 3;
        }
        bcost = cur + ycost;
        for (; i < n; i++)                                          // the 0-2 left over: ordinary cost against the restored bcost
        {
            const int x = list[i];
            const int cost = c.cost_mv(x, ty);
            if (cost < bcost) { bcost = cost; bmv.x = x; bmv.y = ty; }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");       // before the next row overwrites the list
    }
    return true;
}

// SEA: the kernels are instantiated twice, with only the SEA pattern or with all the others (the pattern code shares a register
// budget; SEA's row loop would push the 16-bit wavefront-per-PU kernel into scratch)
template <typename Px, int G, int T, bool SEA>
__device__ __forceinline__ void search_job(const SearchArgs& a, const int job)
{
    constexpr int BPP = sizeof(Px);
    const int gl = threadIdx.x & (G - 1);                         // lane inside the group
    x265hip_me_search_job jb = a.jobs[job];
    PuEval<Px, G, T> c;
    // 32-bit lane offsets from one scalar base: the loads take the saddr form and the address math stays in 32 bits
    constexpr uint32_t kBias = 1u << 30;
    c.base = a.fref - kBias;
    c.strideB = (int)a.frefStrideB; c.depth = a.depth; c.cost = a.cost;
    c.mvpx = jb.qmvpx; c.mvpy = jb.qmvpy;
    c.mvmin.x = a.mvminx; c.mvmin.y = a.mvminy; c.mvmax.x = a.mvmaxx; c.mvmax.y = a.mvmaxy;
    const int tilesX = jb.w >> 2, ntiles = tilesX * (jb.h >> 2);
#pragma unroll
    for (int k = 0; k < T; k++)
    {
        const int t = gl + G * k;
        c.have[k] = t < ntiles;
        const int ty = c.have[k] ? t / tilesX : 0, tx = c.have[k] ? t - ty * tilesX : 0;
        const uint8_t* fe = a.fenc + (long)(jb.py + ty * 4) * a.fencStrideB + (long)(jb.px + tx * 4) * BPP;
        c.refOrg[k] = kBias + (uint32_t)((jb.py + ty * 4) * c.strideB + (jb.px + tx * 4) * BPP);
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int q = 0; q < BPP; q++) c.src[k][r][q] = c.have[k] ? ld_u32(fe + r * a.fencStrideB + 4 * q) : 0;
    }
    const int merange = a.merange;
    const int qminx = c.mvmin.x * 4, qminy = c.mvmin.y * 4, qmaxx = c.mvmax.x * 4, qmaxy = c.mvmax.y * 4;
    const int pmvx = s_clip3(qminx, qmaxx, c.mvpx), pmvy = s_clip3(qminy, qmaxy, c.mvpy);
    int bprecost = c.cmp_q(pmvx, pmvy, false);
    int bestprex = pmvx, bestprey = pmvy;
    SMv bmv = { (pmvx + 2) >> 2, (pmvy + 2) >> 2 };
    int bcost = bprecost;
    if ((pmvx | pmvy) & 3) bcost = c.cost_mv(bmv.x, bmv.y);
    if (pmvx | pmvy)
    {
        const int cost = c.sad_at(0, 0) + c.mvcost_q(0, 0);
        if (cost < bcost)
        {
            bcost = cost;
            bmv.x = 0;
            const int zy = 0 < c.mvmax.y ? 0 : c.mvmax.y;
            bmv.y = zy > c.mvmin.y ? zy : c.mvmin.y;
        }
    }
    // extra quarter-pel candidates (motion.cpp:800-812) compete with the measured predictor
    const int numMvc = (a.mvc && a.numMvc) ? a.numMvc[job] : 0;
    for (int i = 0; i < numMvc; i++)
    {
        const int mx = s_clip3(qminx, qmaxx, a.mvc[(job * 12 + i) * 2]), my = s_clip3(qminy, qmaxy, a.mvc[(job * 12 + i) * 2 + 1]);
        if ((mx | my) && (mx != pmvx || my != pmvy) && (mx != bestprex || my != bestprey))
        {
            const int cost = c.cmp_q(mx, my, false) + c.mvcost_q(mx, my);
            if (cost < bprecost) { bprecost = cost; bestprex = mx; bestprey = my; }
        }
    }
    int costs[4];
#define YOK(DY) ((bmv.y + (DY) >= c.mvmin.y) & (bmv.y + (DY) <= c.mvmax.y))
#define LT(V) do { const int v_ = (V); if (v_ < bcost) bcost = v_; } while (0)
    if constexpr (SEA)              // X265_SEA
    {
        if (!sea_search(c, a, jb, bmv, bcost, merange))
        {
            if (gl == 0) { a.jobs[job].out_qmvx = 0; a.jobs[job].out_qmvy = 0; a.jobs[job].out_cost = -1; }
            return;
        }
    }
    else if (a.method == 0)         // X265_DIA_SEARCH
    {
        bcost <<= 4;
        int i = merange;
        do
        {
            { const int mxs[4] = { bmv.x, bmv.x, bmv.x - 1, bmv.x + 1 }, mys[4] = { bmv.y - 1, bmv.y + 1, bmv.y, bmv.y }; c.template cost_mv_n<4>(mxs, mys, costs); }
            if (YOK(-1)) LT((costs[0] << 4) + 1);
            if (YOK(1)) LT((costs[1] << 4) + 3);
            LT((costs[2] << 4) + 4);
            LT((costs[3] << 4) + 12);
            if (!(bcost & 15)) break;
            bmv.x -= (int)((uint32_t)bcost << 28) >> 30;
            bmv.y -= (int)((uint32_t)bcost << 30) >> 30;
            bcost &= ~15;
        }
        while (--i && c.in_range(bmv.x, bmv.y));
        bcost >>= 4;
    }
    else if (a.method == 1)         // X265_HEX_SEARCH
    {
        hex_search<Px, G, T>(c, bmv, bcost, merange);
    }
    else if (a.method == 2)         // X265_UMH_SEARCH; pmv is rounded to full-pel before the pattern switch (motion.cpp:814)
    {
        umh_search<Px, G, T>(c, bmv, bcost, merange, (pmvx + 2) >> 2, (pmvy + 2) >> 2, jb.h, jb.w == 64 && jb.h == 64,
                             a.mvc ? a.mvc + (size_t)job * 24 : nullptr, numMvc);
    }
    else if (a.method == 3)         // X265_STAR_SEARCH (motion.cpp:1138-1240), one StarPatternSearch call site for both phases
    {
        int bPointNr = 0, bDistance = 0;
        auto two_points = [&]()
        {
            const SMv m1 = { bmv.x + kSOffsets[(bPointNr - 1) * 2].x, bmv.y + kSOffsets[(bPointNr - 1) * 2].y };
            const SMv m2 = { bmv.x + kSOffsets[(bPointNr - 1) * 2 + 1].x, bmv.y + kSOffsets[(bPointNr - 1) * 2 + 1].y };
            if (c.in_range(m1.x, m1.y)) { const int cost = c.cost_mv(m1.x, m1.y); if (cost < bcost) { bcost = cost; bmv = m1; } }
            if (c.in_range(m2.x, m2.y)) { const int cost = c.cost_mv(m2.x, m2.y); if (cost < bcost) { bcost = cost; bmv = m2; } }
        };
        for (int iter = 0;; iter = 1)
        {
            if (iter) { bDistance = 0; bPointNr = 0; }
            star_pattern<Px, G, T>(c, bmv, bcost, bPointNr, bDistance, iter ? 32 : 3, merange);
            if (iter == 0)
            {
                if (bDistance == 1)
                {
                    if (!bPointNr) break;
                    const int saved = bcost;
                    two_points();
                    if (bcost == saved) break;
                }
                const int RasterDistance = 5;
                if (bDistance > RasterDistance)
                {
                    for (int ty = c.mvmin.y; ty <= c.mvmax.y; ty += RasterDistance)
                        for (int tx = c.mvmin.x; tx <= c.mvmax.x; tx += RasterDistance)
                        {
                            if (tx + RasterDistance * 3 <= c.mvmax.x)
                            {
                                // one sad_x4 group; its fourth candidate is priced with mvcost(tmv << 3) (:1196)
                                const int mxs[4] = { tx, tx + RasterDistance, tx + 2 * RasterDistance, tx + 3 * RasterDistance }, mys[4] = { ty, ty, ty, ty };
                                int cs[4];
                                c.template cost_mv_n<4>(mxs, mys, cs);
                                cs[3] += c.mvcost_q(mxs[3] * 8, ty * 8) - c.mvcost_q(mxs[3] * 4, ty * 4);
#pragma unroll
                                for (int k = 0; k < 4; k++)
                                    if (cs[k] < bcost) { bcost = cs[k]; bmv.x = mxs[k]; bmv.y = ty; }
                                tx += 3 * RasterDistance;
                            }
                            else
                            {
                                const int cost = c.cost_mv(tx, ty);
                                if (cost < bcost) { bcost = cost; bmv.x = tx; bmv.y = ty; }
                            }
                        }
                }
                if (!(bDistance > 0)) break;
                continue;
            }
            if (bDistance == 1) { if (bPointNr) two_points(); break; }
            if (!(bDistance > 0)) break;
        }
    }
    else                            // X265_FULL_SEARCH
    {
        for (int ty = c.mvmin.y; ty <= c.mvmax.y; ty++)
            for (int tx = c.mvmin.x; tx <= c.mvmax.x; tx++)
            {
                const int cost = c.cost_mv(tx, ty);
                if (cost < bcost) { bcost = cost; bmv.x = tx; bmv.y = ty; }
            }
    }
#undef YOK
#undef LT

    int bx, by;
    if (bprecost < bcost) { bx = bestprex; by = bestprey; bcost = bprecost; }
    else { bx = bmv.x * 4; by = bmv.y * 4; }
    const int hpelIters = kSWorkload[a.subme][0], hpelDirs = kSWorkload[a.subme][1], qpelIters = kSWorkload[a.subme][2],
              qpelDirs = kSWorkload[a.subme][3];
    const bool hpelSatd = kSWorkload[a.subme][4] != 0;
    if (!bcost)
        bcost = c.mvcost_q(bx, by);
    else
    {
        if (hpelSatd) bcost = c.cmp_q(bx, by, true) + c.mvcost_q(bx, by);
        for (int iter = 0; iter < hpelIters; iter++)
        {
            int bdir = 0;
            for (int i = 1; i <= hpelDirs; i++)
            {
                const int qx = bx + sSquare1(i).x * 2, qy = by + sSquare1(i).y * 2;
                if ((qy < qminy) | (qy > qmaxy)) continue;
                const int cost = c.cmp_q(qx, qy, hpelSatd) + c.mvcost_q(qx, qy);
                if (cost < bcost) { bcost = cost; bdir = i; }
            }
            if (bdir) { bx += sSquare1(bdir).x * 2; by += sSquare1(bdir).y * 2; }
            else break;
        }
        if (!hpelSatd) bcost = c.cmp_q(bx, by, true) + c.mvcost_q(bx, by);
        for (int iter = 0; iter < qpelIters; iter++)
        {
            int bdir = 0;
            for (int i = 1; i <= qpelDirs; i++)
            {
                const int qx = bx + sSquare1(i).x, qy = by + sSquare1(i).y;
                if ((qy < qminy) | (qy > qmaxy)) continue;
                const int cost = c.cmp_q(qx, qy, true) + c.mvcost_q(qx, qy);
                if (cost < bcost) { bcost = cost; bdir = i; }
            }
            if (bdir) { bx += sSquare1(bdir).x; by += sSquare1(bdir).y; }
            else break;
        }
    }
    if (gl == 0)
    {
        a.jobs[job].out_qmvx = bx;
        a.jobs[job].out_qmvy = by;
        a.jobs[job].out_cost = bcost;
    }
}

// size class of a job: 0 = quad (<= 8 tiles), 1 = 16-lane row (<= 32 tiles), 2 = whole wavefront
__device__ __forceinline__ int job_class(const x265hip_me_search_job& j)
{
    const int nt = (j.w >> 2) * (j.h >> 2);
    return nt <= 8 ? 0 : (nt <= 32 ? 1 : 2);
}

template <typename Px, bool SEA>
__global__ void __launch_bounds__(256, 3) me_search_kernel(SearchArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int first = wave * 16;                                  // this wavefront's 16 consecutive jobs
    if (first >= a.njobs) return;
    // classes of the 16 jobs, one per lane 0..15, shared through ballots
    int cls = 3;
    if (lane < 16 && first + lane < a.njobs) cls = job_class(a.jobs[first + lane]);
    const unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1), m2 = __ballot(cls == 2);
    // small PUs: quad q takes job q
    {
        const int q = lane >> 2;
        if ((m0 >> q) & 1) search_job<Px, 4, 2, SEA>(a, first + q);
    }
    // medium PUs: four at a time, one per 16-lane row
    {
        unsigned long long m = m1;
        while (m)
        {
            int mine = -1;
#pragma unroll
            for (int r = 0; r < 4; r++)
            {
                if (!m) break;
                const int j = __ffsll((long long)m) - 1;
                m &= m - 1;
                if ((lane >> 4) == r) mine = j;
            }
            if (mine >= 0) search_job<Px, 16, 2, SEA>(a, first + mine);
        }
    }
    // large PUs need the whole wavefront: they are queued for the second kernel, which gives each of them a wavefront of its own
    // (sixteen of them one after the other in this wavefront would leave most of the chip idle)
    if (m2)
    {
        const int n = __popcll(m2);
        int base = 0;
        if (lane == 0) base = atomicAdd(a.largeCount, n);
        base = __shfl(base, 0, 64);
        if (cls == 2) a.largeQueue[base + __popcll(m2 & ((1ull << lane) - 1))] = first + lane;
    }
}

template <typename Px, bool SEA>
__global__ void __launch_bounds__(256, 3) me_search_large_kernel(SearchArgs a)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= *a.largeCount) return;
    search_job<Px, 64, 4, SEA>(a, a.largeQueue[w]);
}

} // namespace x265hip

using namespace x265hip;

extern "C" int x265hip_me_search(const x265hip_me_search_params* p, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !p->fenc || !p->fref || !p->cost_q || !p->jobs) { set_error("me_search: NULL operand"); return X265HIP_EINVAL; }
    if (p->njobs < 0) { set_error("me_search: njobs %d", p->njobs); return X265HIP_EINVAL; }
    if (p->njobs == 0) return 0;
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("me_search: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->method < X265HIP_ME_DIA || p->method > X265HIP_ME_FULL) { set_error("me_search: unknown search method %d", p->method); return X265HIP_EINVAL; }
    if (p->method == X265HIP_ME_SEA && (p->merange < 0 || 2 * p->merange + 4 > 128)) { set_error("me_search: X265_SEA merange %d out of [0,62]", p->merange); return X265HIP_EINVAL; }
    if (p->method == X265HIP_ME_SEA)
        for (int k = 0; k < 12; k++)
            if (!p->integral[k]) { set_error("me_search: X265_SEA needs the twelve block-sum planes (x265hip_sea_integral); integral[%d] is NULL", k); return X265HIP_EINVAL; }
    if (p->subme < 0 || p->subme > 7) { set_error("me_search: subme %d out of [0,7]", p->subme); return X265HIP_EINVAL; }
    if (p->mvmin_x > p->mvmax_x || p->mvmin_y > p->mvmax_y) { set_error("me_search: empty mv range"); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    SearchArgs a;
    a.fenc = (const uint8_t*)p->fenc; a.fencStrideB = (long)p->fenc_stride * bpp;
    a.fref = (const uint8_t*)p->fref; a.frefStrideB = (long)p->fref_stride * bpp;
    a.cost = p->cost_q + p->qoff; a.jobs = p->jobs; a.njobs = p->njobs;
    a.depth = p->depth; a.method = p->method; a.subme = p->subme; a.merange = p->merange;
    a.mvminx = p->mvmin_x; a.mvminy = p->mvmin_y; a.mvmaxx = p->mvmax_x; a.mvmaxy = p->mvmax_y;
    a.mvc = p->mvc; a.numMvc = p->num_mvc;
    for (int k = 0; k < 12; k++) a.integral[k] = p->integral[k];
    hipStream_t s = (hipStream_t)stream;
    // scratch of this stream: [0] = number of large jobs, [1..] = their indices.  Another thread on the same stream must not queue its clear between
    // this call's clear and kernels: the sequence is enqueued under the stream's sequence lock (round-3 advisor)
    const std::unique_lock<std::mutex> seq = stream_sequence_lock(s);
    int* scratch = (int*)stream_scratch(s, 0, sizeof(int) * ((size_t)p->njobs + 1));
    if (!scratch) return X265HIP_ENODEV;
    X265HIP_TRY(hipMemsetAsync(scratch, 0, sizeof(int), s));
    a.largeCount = scratch; a.largeQueue = scratch + 1;
    const int wgs = (p->njobs + 63) / 64;                          // 4 wavefronts x 16 jobs per workgroup
    // the second grid covers the worst case (every job large): surplus wavefronts leave at once
    const int wgsLarge = (p->njobs + 3) / 4;
    const bool sea = p->method == X265HIP_ME_SEA;
#define LAUNCH_SEARCH(PX, SEA) do { \
        hipLaunchKernelGGL((me_search_kernel<PX, SEA>), dim3(wgs), dim3(256), 0, s, a); \
        hipLaunchKernelGGL((me_search_large_kernel<PX, SEA>), dim3(wgsLarge), dim3(256), 0, s, a); } while (0)
    if (bpp == 1) { if (sea) LAUNCH_SEARCH(uint8_t, true); else LAUNCH_SEARCH(uint8_t, false); }
    else { if (sea) LAUNCH_SEARCH(uint16_t, true); else LAUNCH_SEARCH(uint16_t, false); }
#undef LAUNCH_SEARCH
    X265HIP_TRY(hipGetLastError());
    return 0;
}
