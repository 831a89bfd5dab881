// pu_eval.h - a PU held by a lane group (quad / 16-lane row / wavefront) and scored candidate by candidate: shared by the search
// drivers (search_kernels.hip) and the lookahead cost estimate (lowres_cost_kernels.hip).
#pragma once
#include "tile_interp.h"

namespace x265hip {

struct SMv { int x, y; };

// The small direction tables of encoder/motion.cpp:64-66 (hex2, mod6m1, square1) as packed immediates.  They are indexed by the direction a
// lane group has just decided; a __constant__ array indexed per lane is a vector load from memory - one more dependent round trip in every
// round of a search whose time IS its chain of round trips (profiles/r03_lowres_cost_counters.txt).  Four bits per entry, value + 2.
constexpr uint64_t pk_nib(int a0, int a1, int a2, int a3, int a4, int a5, int a6, int a7, int a8 = 0)
{
    return (uint64_t)a0 | (uint64_t)a1 << 4 | (uint64_t)a2 << 8 | (uint64_t)a3 << 12 | (uint64_t)a4 << 16 | (uint64_t)a5 << 20 | (uint64_t)a6 << 24 |
           (uint64_t)a7 << 28 | (uint64_t)a8 << 32;
}
__device__ __forceinline__ SMv sHex2(int i)
{
    constexpr uint32_t X = (uint32_t)pk_nib(-1 + 2, -2 + 2, -1 + 2, 1 + 2, 2 + 2, 1 + 2, -1 + 2, -2 + 2), Y = (uint32_t)pk_nib(-2 + 2, 0 + 2, 2 + 2, 2 + 2, 0 + 2, -2 + 2, -2 + 2, 0 + 2);
    return { (int)((X >> (4 * i)) & 15) - 2, (int)((Y >> (4 * i)) & 15) - 2 };
}
__device__ __forceinline__ int sMod6m1(int i) { return (int)(((uint32_t)pk_nib(5, 0, 1, 2, 3, 4, 5, 0) >> (4 * i)) & 15); }      // (x - 1) % 6
__device__ __forceinline__ SMv sSquare1(int i)
{
    constexpr uint64_t X = pk_nib(0 + 2, 0 + 2, 0 + 2, -1 + 2, 1 + 2, -1 + 2, -1 + 2, 1 + 2, 1 + 2), Y = pk_nib(0 + 2, -1 + 2, 1 + 2, 0 + 2, 0 + 2, -1 + 2, 1 + 2, -1 + 2, 1 + 2);
    return { (int)((X >> (4 * i)) & 15) - 2, (int)((Y >> (4 * i)) & 15) - 2 };
}
static __constant__ SMv kSOffsets[16] = { { -1, 0 }, { 0, -1 }, { -1, -1 }, { 1, -1 }, { -1, 0 }, { 1, 0 }, { -1, 1 }, { -1, -1 },
                                   { 1, -1 }, { 1, 1 }, { -1, 0 }, { 0, 1 }, { -1, 1 }, { 1, 1 }, { 1, 0 }, { 0, 1 } };
static __constant__ int kSWorkload[8][5] = { { 1, 4, 0, 4, 0 }, { 1, 4, 1, 4, 0 }, { 1, 4, 1, 4, 1 }, { 2, 4, 1, 4, 1 },
                                      { 2, 4, 2, 4, 1 }, { 1, 8, 1, 8, 1 }, { 2, 8, 1, 8, 1 }, { 2, 8, 2, 8, 1 } };

// keeps a value (and the loads behind it) where it is computed: an empty asm that reads and writes the register
__device__ __forceinline__ void pin_value(int& v) { asm volatile("" : "+v"(v)); }

// sum over the G lanes of a group, result in every lane of the group
template <int G> __device__ __forceinline__ int group_total(int v)
{
    if (G == 4) return quad_sum(v);
    if (G == 16) return row_sum(v);
    return wave_sum_of_rows(row_sum(v));
}

template <typename Px, int G, int T>
struct PuEval
{
    static constexpr int BPP = sizeof(Px);
    static constexpr int DW = BPP;                 // dwords per 4-sample tile row
    const uint8_t* base;                            // wave-uniform: reference plane origin minus kBias bytes
    uint32_t refOrg[T];                             // byte offset from `base` of the reference under each of the lane's tiles (mv 0)
    uint32_t src[T][4][DW];                         // the lane's source tiles, packed
    bool have[T];
    int strideB;
    int depth;
    const uint16_t* cost;
    int mvpx, mvpy;
    SMv mvmin, mvmax;

    __device__ __forceinline__ int mvcost_q(int qx, int qy) const { return (int)cost[qx - mvpx] + (int)cost[qy - mvpy]; }
    __device__ __forceinline__ bool in_range(int x, int y) const { return x >= mvmin.x && x <= mvmax.x && y >= mvmin.y && y <= mvmax.y; }

    // SAD of the PU at integer displacement (mx, my)
    __device__ __forceinline__ int sad_at(int mx, int my) const
    {
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < T; k++)
        {
            if (!have[k]) continue;
            const uint32_t ro = refOrg[k] + (uint32_t)(my * strideB + mx * BPP);
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int q = 0; q < DW; q++)
                    acc = sad_dw<Px>(ld_u32(base + (ro + (uint32_t)(r * strideB + 4 * q))), src[k][r][q], acc);
        }
        return group_total<G>((int)acc);
    }
    __device__ __forceinline__ int cost_mv(int mx, int my) const { return sad_at(mx, my) + mvcost_q(mx * 4, my * 4); }

    // N candidates at once (the reference's sad_x3 / sad_x4 groups): all reference loads are issued before the first
    // reduction, so one memory round trip serves the group.  sad_n: the bare SADs (SEA adds its own cost terms); cost_mv_n: SAD + mvcost
    template <int N>
    __device__ __forceinline__ void sad_n(const int (&mx)[N], const int (&my)[N], int (&out)[N]) const
    {
        uint32_t acc[N];
#pragma unroll
        for (int n = 0; n < N; n++) acc[n] = 0;
#pragma unroll
        for (int k = 0; k < T; k++)
        {
            if (!have[k]) continue;
#pragma unroll
            for (int n = 0; n < N; n++)
            {
                const uint32_t ro = refOrg[k] + (uint32_t)(my[n] * strideB + mx[n] * BPP);
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int q = 0; q < DW; q++)
                        acc[n] = sad_dw<Px>(ld_u32(base + (ro + (uint32_t)(r * strideB + 4 * q))), src[k][r][q], acc[n]);
            }
            // at most 64 dwords in flight per lane: larger groups go one tile at a time
            if (N * T * 4 * DW > 64) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int n = 0; n < N; n++) out[n] = group_total<G>((int)acc[n]);
    }
    template <int N>
    __device__ __forceinline__ void cost_mv_n(const int (&mx)[N], const int (&my)[N], int (&out)[N]) const
    {
        uint32_t acc[N];
        int mc[N];
        // the mv costs first: their table loads travel with the reference loads (one round trip), not behind the reduction
#pragma unroll
        for (int n = 0; n < N; n++) { acc[n] = 0; mc[n] = mvcost_q(mx[n] * 4, my[n] * 4); }
#pragma unroll
        for (int k = 0; k < T; k++)
        {
            if (!have[k]) continue;
#pragma unroll
            for (int n = 0; n < N; n++)
            {
                const uint32_t ro = refOrg[k] + (uint32_t)(my[n] * strideB + mx[n] * BPP);
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int q = 0; q < DW; q++)
                        acc[n] = sad_dw<Px>(ld_u32(base + (ro + (uint32_t)(r * strideB + 4 * q))), src[k][r][q], acc[n]);
            }
            if (N * T * 4 * DW > 64) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int n = 0; n < N; n++) out[n] = group_total<G>((int)acc[n]) + mc[n];
        // every score of the group exists HERE: without the pin the compiler sinks a candidate's table load and sum into the branch that
        // reads it - a memory round trip per decision instead of one per group
#pragma unroll
        for (int n = 0; n < N; n++) pin_value(out[n]);
    }

    // subpelCompare: SAD or SATD of the PU at quarter-pel displacement (qx, qy)
    __device__ __forceinline__ int cmp_q(int qx, int qy, bool useSatd) const
    {
        if (!((qx | qy) & 3) && !useSatd) return sad_at(qx >> 2, qy >> 2);
        int acc = 0;
#pragma unroll
        for (int k = 0; k < T; k++)
        {
            if (!have[k]) continue;
            int d[4][4];
            tile_predict<BPP>(base + (refOrg[k] + (uint32_t)((qy >> 2) * strideB + (qx >> 2) * BPP)), (long)strideB, qx & 3, qy & 3, depth, d);
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++)
                {
                    const int s = BPP == 1 ? (int)((src[k][y][0] >> (8 * x)) & 0xff) : (int)((src[k][y][x >> 1] >> (16 * (x & 1))) & 0xffff);
                    d[y][x] = s - d[y][x];
                }
            if (useSatd) acc += tile_satd4(d);
            else
            {
#pragma unroll
                for (int y = 0; y < 4; y++)
#pragma unroll
                    for (int x = 0; x < 4; x++) acc += abs(d[y][x]);
            }
            __builtin_amdgcn_sched_barrier(0);        // one tile's interpolation at a time: bounds the register footprint
        }
        return group_total<G>(acc);
    }
};


__device__ __forceinline__ int s_clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

// X265_HEX_SEARCH (motion.cpp:852-948): first hexagon, directional half-hexagon walk, square refine
template <typename Px, int G, int T>
__device__ __forceinline__ void hex_search(const PuEval<Px, G, T>& c, SMv& bmv, int& bcost, int merange)
{
    int costs[4];
#define YOK(DY) ((bmv.y + (DY) >= c.mvmin.y) & (bmv.y + (DY) <= c.mvmax.y))
#define LT(V) do { const int v_ = (V); if (v_ < bcost) bcost = v_; } while (0)
#define X3(D0X, D0Y, D1X, D1Y, D2X, D2Y) do { const int mxs_[3] = { bmv.x + (D0X), bmv.x + (D1X), bmv.x + (D2X) }, mys_[3] = { bmv.y + (D0Y), bmv.y + (D1Y), bmv.y + (D2Y) }; \
                                              int cs_[3]; c.template cost_mv_n<3>(mxs_, mys_, cs_); costs[0] = cs_[0]; costs[1] = cs_[1]; costs[2] = cs_[2]; } while (0)
        // the six points of the first hexagon are independent of each other: one group, one memory round trip; the decisions
        // are then taken in the reference's order (two COST_MV_X3 calls, motion.cpp:868-885)
        {
            const int mxs_[6] = { bmv.x - 2, bmv.x - 1, bmv.x + 1, bmv.x + 2, bmv.x + 1, bmv.x - 1 };
            const int mys_[6] = { bmv.y, bmv.y + 2, bmv.y + 2, bmv.y, bmv.y - 2, bmv.y - 2 };
            int cs_[6];
            c.template cost_mv_n<6>(mxs_, mys_, cs_);
            bcost <<= 3;
            if (YOK(0)) LT((cs_[0] << 3) + 2);
            if (YOK(2)) { LT((cs_[1] << 3) + 3); LT((cs_[2] << 3) + 4); }
            if (YOK(0)) LT((cs_[3] << 3) + 5);
            if (YOK(-2)) { LT((cs_[4] << 3) + 6); LT((cs_[5] << 3) + 7); }
        }
        if (bcost & 7)
        {
            int dir = (bcost & 7) - 2;
            if (YOK(sHex2(dir + 1).y))
            {
                bmv.x += sHex2(dir + 1).x; bmv.y += sHex2(dir + 1).y;
                for (int i = (merange >> 1) - 1; i > 0 && c.in_range(bmv.x, bmv.y); i--)
                {
                    X3(sHex2(dir + 0).x, sHex2(dir + 0).y, sHex2(dir + 1).x, sHex2(dir + 1).y, sHex2(dir + 2).x, sHex2(dir + 2).y);
                    bcost &= ~7;
                    if (YOK(sHex2(dir + 0).y)) LT((costs[0] << 3) + 1);
                    if (YOK(sHex2(dir + 1).y)) LT((costs[1] << 3) + 2);
                    if (YOK(sHex2(dir + 2).y)) LT((costs[2] << 3) + 3);
                    if (!(bcost & 7)) break;
                    dir += (bcost & 7) - 2;
                    dir = sMod6m1(dir + 1);
                    bmv.x += sHex2(dir + 1).x; bmv.y += sHex2(dir + 1).y;
                }
            }
        }
        bcost >>= 3;
        int dir = 0;
        {
            // square refine (:927-947): the eight neighbours of the final hexagon centre in one group, accepted in the reference's order
            const int mxs[8] = { bmv.x, bmv.x, bmv.x - 1, bmv.x + 1, bmv.x - 1, bmv.x - 1, bmv.x + 1, bmv.x + 1 };
            const int mys[8] = { bmv.y - 1, bmv.y + 1, bmv.y, bmv.y, bmv.y - 1, bmv.y + 1, bmv.y - 1, bmv.y + 1 };
            int cs8[8];
            c.template cost_mv_n<8>(mxs, mys, cs8);
            if (YOK(-1) && cs8[0] < bcost) { bcost = cs8[0]; dir = 1; }
            if (YOK(1) && cs8[1] < bcost) { bcost = cs8[1]; dir = 2; }
            if (cs8[2] < bcost) { bcost = cs8[2]; dir = 3; }
            if (cs8[3] < bcost) { bcost = cs8[3]; dir = 4; }
            if (YOK(-1) && cs8[4] < bcost) { bcost = cs8[4]; dir = 5; }
            if (YOK(1) && cs8[5] < bcost) { bcost = cs8[5]; dir = 6; }
            if (YOK(-1) && cs8[6] < bcost) { bcost = cs8[6]; dir = 7; }
            if (YOK(1) && cs8[7] < bcost) { bcost = cs8[7]; dir = 8; }
        }
        bmv.x += sSquare1(dir).x; bmv.y += sSquare1(dir).y;
#undef X3
#undef YOK
#undef LT
}

} // namespace x265hip
