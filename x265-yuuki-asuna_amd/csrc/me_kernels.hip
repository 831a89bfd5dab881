// me_kernels.hip - CTU-tiled exhaustive integer motion search on gfx950.
//
// Reference semantics: pu[LUMA_NxN].sad / sad_x3 / sad_x4 (source/common/pixel.cpp:40-119) as
// issued by the full search of MotionEstimate::motionEstimate (source/encoder/motion.cpp:1397-1445:
// every mv of the [-range,range]^2 window in raster order, COPY2_IF_LT strict-less tie-break,
// cost = sad + mvcost).  SAD is additive over sub-blocks, so the 16x16/32x32/64x64 values are the
// exact sums of the 8x8 ones.
//
// Mapping (one workgroup per 64x64 CTU):
//   * the (64+2R)^2 reference window is staged once in LDS (dword-aligned copy of the rows, so HBM is
//     read with coalesced aligned dwords and each reference pixel leaves HBM/L2 once per CTU);
//   * a wavefront owns one mv column (fixed mvx): lane l holds the l-th 8x8 block (z-order) of the
//     CTU in registers and walks DOWN the window row by row.  A window row contributes to the 8
//     vertical displacements it overlaps, so each lane keeps a ring of 8 running SADs: 3 LDS dwords
//     (12 B) + 2 v_alignbit feed 16 v_sad_u8 - register reuse keeps LDS at ~1/5 of its bandwidth;
//   * every row step completes one mv: 8x8 SADs are summed to 16x16 (DPP quad_perm), 32x32 (DPP
//     row_ror) and 64x64 (v_readlane) without touching LDS, written as one 85-int record per mv
//     ([ctu][mvy][mvx][64 + 16 + 4 + 1]) so a wavefront stores 340 contiguous bytes, and/or folded into a per-PU running minimum of
//     (cost << 32 | raster index) that is merged across wavefronts with one 64-bit atomicMin.
#include "common.h"
#include <mutex>

#include <cstdlib>
#include <type_traits>

namespace x265hip {

struct MEArgs
{
    const uint8_t* fenc;  long fencStrideB;     // byte strides
    const uint8_t* fref;  long frefStrideB;
    int ctusW;
    int range;            // R
    int rowBytes;         // LDS row pitch in bytes (multiple of 128: see lds_row_off)
    int payloadDw;        // dwords copied per window row
    int32_t*  surf;              // [ctu][mvy][mvx/4][85][4]
    unsigned long long* best;     // [ctu][85]
    const uint16_t* costX;
    const uint16_t* costY;
    const int16_t* centres;      // optional [ctu][2]: the window of CTU c is centred on displacement (centres[2c], centres[2c+1]) instead of (0, 0)
    int xcdOrder;                // 1: a whole-picture launch maps workgroup -> CTU so that every XCD holds a contiguous band of CTUs (X265HIP_ME_XCD_OFF=1: raster order, A/B)
};

// lane -> 8x8 block coordinates inside the CTU, z-order (quad = one 16x16, 16 lanes = one 32x32)
__device__ __forceinline__ void zorder_xy(int lane, int& bx, int& by)
{
    bx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4);
    by = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
}

template <typename Px> struct MECfg;
template <> struct MECfg<uint8_t>  { static constexpr int DWPR = 2; static constexpr int NLD = 3; };   // dwords per 8-px row; dwords loaded
template <> struct MECfg<uint16_t> { static constexpr int DWPR = 4; static constexpr int NLD = 5; };

// LDS row placement.  A half-wave reads, for one window row step, dwords by*8*pitch + bx*DWPR + c
// (by = 0..3 block rows, bx = 0..7).  With the pitch a multiple of 32 dwords the four block rows
// would collide on the same banks, so every group of 8 window rows is skewed by {0,1,16,17} dwords
// (u8: the 8 bx values occupy the even banks 0..14, so the four skews tile all 32 banks conflict-free).
__device__ __forceinline__ int lds_skew_bytes(int rowGroup)
{
    const int g = rowGroup & 3;
    return ((g & 1) + ((g & 2) << 3)) * 4;                // 0, 1, 16, 17 dwords
}
__device__ __forceinline__ int lds_row_off(int r, int pitchBytes)
{
    return r * pitchBytes + lds_skew_bytes(r >> 3);
}

// PITCH = LDS row pitch in bytes as a compile-time constant (ds_read offsets become immediates).
template <typename Px, bool SURF, bool BEST, int PITCH>
__global__ void __launch_bounds__(1024, SURF ? 4 : 8) me_ctu_kernel(MEArgs a)
{
    constexpr int BPP  = PxInfo<Px>::BPP;
    constexpr int DWPR = MECfg<Px>::DWPR;
    constexpr int NLD  = MECfg<Px>::NLD;
    extern __shared__ __attribute__((aligned(16))) uint8_t win[];

    const int R = a.range;
    const int NC = 2 * R + 1;                 // mv columns == mv rows
    const int NG = (NC + 3) >> 2;             // column groups of 4 in the surface layout
    const int rows = 64 + 2 * R;
    // XCD-aware order (round 6): a whole-picture launch hands every XCD a contiguous band of CTUs - the windows of neighbouring CTUs overlap by 2 R / (64 + 2 R) and meet in one L2
    const int ctu = (gridDim.y == 1 && a.xcdOrder) ? xcd_swizzle((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int cx = (ctu % a.ctusW) * 64, cy = (ctu / a.ctusW) * 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    constexpr int pitch = PITCH;

    // ---- stage the search window: aligned dword copy of each row -------------------------------
    const int ccx = a.centres ? a.centres[2 * ctu] : 0, ccy = a.centres ? a.centres[2 * ctu + 1] : 0;
    const uint8_t* g0 = a.fref + (long)(cy + ccy - R) * a.frefStrideB + (long)(cx + ccx - R) * BPP;
    const int adj = (int)((uintptr_t)g0 & 3);           // identical for every row (stride % 4 == 0)
    const uint8_t* g0a = g0 - adj;
    const int rowDw = a.payloadDw;
    for (int r = wave; r < rows; r += nwaves)
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(g0a + (long)r * a.frefStrideB);
        uint32_t* dst = reinterpret_cast<uint32_t*>(win + lds_row_off(r, pitch));
        for (int c = lane; c < rowDw; c += 64)
            dst[c] = src[c];
    }

    // ---- this lane's 8x8 source block, kept in registers ---------------------------------------
    int bx, by;
    zorder_xy(lane, bx, by);
    uint32_t F[8][DWPR];
    {
        const uint8_t* fe = a.fenc + (long)(cy + by * 8) * a.fencStrideB + (long)(cx + bx * 8) * BPP;
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int k = 0; k < DWPR; k++)
                F[j][k] = ld_u32(fe + (long)j * a.fencStrideB + 4 * k);
    }
    __syncthreads();

    unsigned long long bk8 = ~0ull, bk16 = ~0ull, bk32 = ~0ull, bk64 = ~0ull;
    // which upper-level value this lane writes into the 85-int record (see the SURF store below)
    const bool uMask = (lane & 3) == 0 || (lane & 15) == 1 || lane == 2;
    const int uSel = (lane & 3) == 0 ? 0 : ((lane & 15) == 1 ? 1 : 2);
    const int uOff = uSel == 0 ? 64 + (lane >> 2) : (uSel == 1 ? 80 + (lane >> 4) : 84);

    const int T = 2 * R + 8;                  // window rows a lane walks through (always even, >= 10)
    for (int mvxi = wave; mvxi < NC; mvxi += nwaves)
    {
        const int xb = adj + (bx * 8 + mvxi) * BPP;                 // byte column inside the LDS row
        const int sh = __builtin_amdgcn_readfirstlane((xb & 3) * 8); // wave-uniform (bx*8*BPP % 4 == 0)
        const int colB = xb & ~3;
        const uint32_t cxv = BEST ? a.costX[mvxi] : 0;

        uint32_t acc[8];
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = 0;

        // rows of one 8-row block share a skew: address = blockBase(t0) + p * pitch (p immediate)
        const uint8_t* colBase = win + (by * 8) * pitch + colB;
        auto block_base = [&](const int t0) { return colBase + t0 * pitch + lds_skew_bytes(by + (t0 >> 3)); };
        auto ldpair = [&](uint32_t (&d)[2][NLD], const uint8_t* bb, const int p)
        {
#pragma unroll
            for (int q = 0; q < 2; q++)
            {
                const uint32_t* lp = reinterpret_cast<const uint32_t*>(bb + (p + q) * pitch);
#pragma unroll
                for (int k = 0; k < NLD; k++) d[q][k] = lp[k];
            }
        };
        auto align_pair = [&](uint32_t (&rr)[2][DWPR], const uint32_t (&d)[2][NLD])
        {
#pragma unroll
            for (int q = 0; q < 2; q++)
#pragma unroll
                for (int k = 0; k < DWPR; k++) rr[q][k] = __builtin_amdgcn_alignbit(d[q][k + 1], d[q][k], sh);
        };
        // software pipeline over PAIRS of window rows: rr = aligned pixels of the pair being consumed,
        // d = raw dwords of the next pair, requested from LDS before the SAD work of the current pair
        uint32_t rr[2][DWPR], d[2][NLD];
        ldpair(d, block_base(0), 0);
        align_pair(rr, d);

        // 8 window rows per call.  FIRST: warm-up rows (displacement index m = t - j < 0 is skipped,
        // only the last row completes an mv).  NROWS < 8 only for the final partial block.
        auto rows8 = [&](auto firstTag, auto nrowsTag, const int t0)
        {
            constexpr bool FIRST = decltype(firstTag)::value;
            constexpr int NROWS = decltype(nrowsTag)::value;
            const uint8_t* bb = block_base(t0);
            const uint8_t* bn = block_base(t0 + 8);
#pragma unroll
            for (int p = 0; p < NROWS; p++)
            {
                if ((p & 1) == 0)                                   // LDS has 2 spare rows past the window
                {
                    if (p + 2 < 8) ldpair(d, bb, p + 2); else ldpair(d, bn, 0);
                }
                // window row t meets source row j at vertical displacement index m = t - j
#pragma unroll
                for (int j = 0; j < 8; j++)
                {
                    if (FIRST && j > p) continue;
                    uint32_t v = acc[(p - j) & 7];
#pragma unroll
                    for (int k = 0; k < DWPR; k++) v = sad_dw<Px>(F[j][k], rr[p & 1][k], v);
                    acc[(p - j) & 7] = v;
                }

                if (!FIRST || p == 7)
                {
                    const int m = t0 + p - 7;                 // completed displacement row
                    const int slot = (p + 1) & 7;
                    const int s8 = (int)acc[slot];
                    acc[slot] = 0;
                    const int s16 = quad_sum(s8);
                    const int s32 = row_sum_of_quads(s16);
                    const int s64 = wave_sum_of_rows(s32);
                    if (SURF)
                    {
                        // uniform group address (scalar math): [ctu][mvy][mvx/4][85 PUs][4 columns]
                        int32_t* grp = a.surf + (((long)ctu * NC + m) * NG + (mvxi >> 2)) * 340 + (mvxi & 3);
                        grp[lane * 4] = s8;
                        // the 21 upper-level values leave in ONE masked store: lane 4q -> 16x16 PU q,
                        // lane 16r+1 -> 32x32 PU r, lane 2 -> the 64x64 PU
                        const int vU = uSel == 0 ? s16 : (uSel == 1 ? s32 : s64);
                        if (uMask) grp[uOff * 4] = vU;
                    }
                    if (BEST)
                    {
                        const uint32_t mvc = cxv + a.costY[m];
                        const uint32_t idx = (uint32_t)(m * NC + mvxi);
                        const unsigned long long k8 = ((unsigned long long)((uint32_t)s8 + mvc) << 32) | idx;
                        const unsigned long long k16 = ((unsigned long long)((uint32_t)s16 + mvc) << 32) | idx;
                        const unsigned long long k32 = ((unsigned long long)((uint32_t)s32 + mvc) << 32) | idx;
                        const unsigned long long k64 = ((unsigned long long)((uint32_t)s64 + mvc) << 32) | idx;
                        bk8 = k8 < bk8 ? k8 : bk8;
                        bk16 = k16 < bk16 ? k16 : bk16;
                        bk32 = k32 < bk32 ? k32 : bk32;
                        bk64 = k64 < bk64 ? k64 : bk64;
                    }
                }
                if (p & 1) align_pair(rr, d);
            }
        };
        using I8 = std::integral_constant<int, 8>;
        rows8(std::true_type{}, I8{}, 0);                       // T >= 10: the first 8 rows always exist
        int t0 = 8;
        for (; t0 + 8 <= T; t0 += 8)
            rows8(std::false_type{}, I8{}, t0);
        switch (T - t0)                                         // T is even: 0, 2, 4 or 6 rows left
        {
        case 2: rows8(std::false_type{}, std::integral_constant<int, 2>{}, t0); break;
        case 4: rows8(std::false_type{}, std::integral_constant<int, 4>{}, t0); break;
        case 6: rows8(std::false_type{}, std::integral_constant<int, 6>{}, t0); break;
        default: break;
        }
    }

    if (BEST)
    {
        unsigned long long* rec = a.best + (size_t)ctu * 85;
        atomicMin(&rec[lane], bk8);
        if ((lane & 3) == 0) atomicMin(&rec[64 + (lane >> 2)], bk16);
        if ((lane & 15) == 0) atomicMin(&rec[80 + (lane >> 4)], bk32);
        if (lane == 0) atomicMin(&rec[84], bk64);
    }
}

// ------------------------------------------------------------------------------------------------
// 8-bit fast path: v_qsad_pk_u16_u8 evaluates one 4-pixel source group against FOUR consecutive
// horizontal displacements in one instruction (24 cycles vs 4 x 8.2 for v_sad_u8, measured), with the
// sliding window taken directly from an aligned pair of LDS dwords - no v_alignbit.  A wavefront owns a
// GROUP of 4 mv columns; the 8-deep ring holds 4 packed u16 SADs per slot (an 8x8 SAD is <= 16320).
// The window is staged so that LDS byte 0 of a row is window column 0 (unaligned global dword loads),
// which makes column group g start on LDS dword 2*bx + g for every CTU.
// VAR (minima-only launches, A/B: X265HIP_ME_BEST_VARIANT): bit 0 = the 8x8 level keeps one running minimum PER COLUMN of `(sad + costY) << 8 | m` and adds a
// column's costX once per group (fewer instructions per row); bit 1 = no scheduling fence between rows.  profiles/r04_me_minima_ab.txt has the timings.
template <bool SURF, bool BEST, int PITCH, bool PACKED = false, int VAR = 0>
__global__ void __launch_bounds__(SURF && BEST ? 768 : 1024, SURF && BEST ? 3 : 4) me_ctu_q_kernel(MEArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t win[];
    typedef unsigned long long u64;
    constexpr bool PIPE = true;

    const int R = a.range;
    const int NC = 2 * R + 1;
    const int NG = (NC + 3) >> 2;
    const int rows = 64 + 2 * R;
    // XCD-aware order (round 6): a whole-picture launch hands every XCD a contiguous band of CTUs - the windows of neighbouring CTUs overlap by 2 R / (64 + 2 R) and meet in one L2
    const int ctu = (gridDim.y == 1 && a.xcdOrder) ? xcd_swizzle((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int cx = (ctu % a.ctusW) * 64, cy = (ctu / a.ctusW) * 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    constexpr int pitch = PITCH;

    const int ccx = a.centres ? a.centres[2 * ctu] : 0, ccy = a.centres ? a.centres[2 * ctu + 1] : 0;
    const uint8_t* g0 = a.fref + (long)(cy + ccy - R) * a.frefStrideB + (long)(cx + ccx - R);
    const int rowDw = a.payloadDw;
    for (int r = wave; r < rows; r += nwaves)
    {
        const uint8_t* src = g0 + (long)r * a.frefStrideB;
        uint32_t* dst = reinterpret_cast<uint32_t*>(win + lds_row_off(r, pitch));
        for (int c = lane; c < rowDw; c += 64)
            dst[c] = ld_u32(src + 4 * c);
    }
    int bx, by;
    zorder_xy(lane, bx, by);
    uint32_t F[8][2];
    {
        const uint8_t* fe = a.fenc + (long)(cy + by * 8) * a.fencStrideB + (long)(cx + bx * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) { F[j][0] = ld_u32(fe + (long)j * a.fencStrideB); F[j][1] = ld_u32(fe + (long)j * a.fencStrideB + 4); }
    }
    __syncthreads();

    // Upper PU levels are reduced "transposed": after the packed quad sums, lane (q = lane >> 2, k = lane & 3) keeps
    // column k of 16x16 PU q; row_ror adds give column k of 32x32 PU (lane >> 4) in every lane of the row, and two
    // v_permlane{16,32}_swap adds (gfx950) give column k of the 64x64 PU in every lane.  So 84 upper values live in 3
    // VGPRs, the 16x16 level is one fully coalesced dword store, and every lane folds a single column per level.
    u64 bk8 = ~0ull, bk16 = ~0ull, bk32 = ~0ull, bk64 = ~0ull;
    const int kcol = lane & 3;
    const uint32_t selK = 0x0c0c0000u | (uint32_t)((2 * kcol + 1) << 8) | (uint32_t)(2 * kcol);   // v_perm: u16 #kcol of {qhi, qlo}
    // one masked dword store covers the 32x32 level (lanes with (lane & 15) < 4) and the 64x64 level (lanes 52..55)
    const bool is64 = lane >= 52 && lane < 56;
    const bool uMask = (lane & 15) < 4 || is64;
    const int uOffDw = is64 ? 84 * 4 + kcol : (80 + (lane >> 4)) * 4 + kcol;

    const int T = 2 * R + 8;
    for (int g = wave; g < NG; g += nwaves)
    {
        const uint8_t* colBase = win + (by * 8) * pitch + bx * 8 + 4 * g;
        auto block_base = [&](const int t0) { return colBase + t0 * pitch + lds_skew_bytes(by + (t0 >> 3)); };
        auto ldpair = [&](uint32_t (&d)[2][3], const uint8_t* bb, const int p)
        {
#pragma unroll
            for (int q = 0; q < 2; q++)
            {
                const uint32_t* lp = reinterpret_cast<const uint32_t*>(bb + (p + q) * pitch);
                d[q][0] = lp[0]; d[q][1] = lp[1]; d[q][2] = lp[2];
            }
        };
        // Row-local 32-bit keys: inside one column group only the row m varies, so `cost << 10 | m << 2 | k` (8x8,
        // cost < 2^17) and `cost << 8 | m` (upper levels, cost < 2^22; m < 256) order candidates exactly like
        // (cost, raster index); they are widened and merged into the 64-bit running minima once per group.
        // Pad columns (>= 2R+1) get a cost offset no real candidate reaches.
        uint32_t cxk4[4] = { 0, 0, 0, 0 }, cxL = 0;      // (costX << 2 | k) per column; this lane's own column cost
        uint32_t r8 = 0xffffffffu, r16 = 0xffffffffu, r32 = 0xffffffffu, r64 = 0xffffffffu;
        uint32_t r8c[4] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu };      // VAR & 1
        if (BEST)
        {
#pragma unroll
            for (int k = 0; k < 4; k++)
                cxk4[k] = ((4 * g + k < NC ? (uint32_t)a.costX[4 * g + k] : (1u << 20)) << 2) | (uint32_t)k;
            cxL = 4 * g + kcol < NC ? (uint32_t)a.costX[4 * g + kcol] : (1u << 23);
        }
        const uint32_t cxL8 = cxL << 8;
        u64 acc[8];
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = 0;
        uint32_t cur[2][3], nxt[2][3];
        ldpair(cur, block_base(0), 0);
        (void)nxt;

        auto rows8 = [&](auto firstTag, auto nrowsTag, const int t0)
        {
            constexpr bool FIRST = decltype(firstTag)::value;
            constexpr int NROWS = decltype(nrowsTag)::value;
            const uint8_t* bb = block_base(t0);
            const uint8_t* bn = block_base(t0 + 8);
#pragma unroll
            for (int p = 0; p < NROWS; p++)
            {
                if ((p & 1) == 0)
                {
                    if (PIPE) { if (p + 2 < 8) ldpair(nxt, bb, p + 2); else ldpair(nxt, bn, 0); }
                    else if (p > 0 || !FIRST) ldpair(cur, bb, p);
                }
                const u64 w0 = ((u64)cur[p & 1][1] << 32) | cur[p & 1][0];   // window bytes 0..7 of this block column
                const u64 w1 = ((u64)cur[p & 1][2] << 32) | cur[p & 1][1];   // bytes 4..11
#pragma unroll
                for (int j = 0; j < 8; j++)
                {
                    if (FIRST && j > p) continue;
                    // j == 0 opens a fresh ring slot (the one emitted a row ago): start from the constant 0 instead of clearing registers
                    u64 v = j == 0 ? 0ull : acc[(p - j) & 7];
                    v = __builtin_amdgcn_qsad_pk_u16_u8(w0, F[j][0], v);
                    v = __builtin_amdgcn_qsad_pk_u16_u8(w1, F[j][1], v);
                    acc[(p - j) & 7] = v;
                }
                if (!FIRST || p == 7)
                {
                    const int m = t0 + p - 7;
                    const int slot = (p + 1) & 7;
                    const u64 A = acc[slot];                                          // the slot is reopened by the next row's j == 0 step
                    const uint32_t lo = (uint32_t)A, hi = (uint32_t)(A >> 32);      // columns {0,1} and {2,3}, u16 each
                    // 16x16 = quad sums, still packed (<= 65280 per half: no carry between the halves)
                    const uint32_t qlo = (uint32_t)quad_sum((int)lo), qhi = (uint32_t)quad_sum((int)hi);
                    const int s8[4] = { (int)(lo & 0xffff), (int)(lo >> 16), (int)(hi & 0xffff), (int)(hi >> 16) };
                    const int v16 = (int)__builtin_amdgcn_perm(qhi, qlo, selK);      // column kcol of 16x16 PU (lane >> 2)
                    const int v32 = row_sum_of_quads(v16);                           // column kcol of 32x32 PU (lane >> 4)
                    typedef unsigned v2u __attribute__((ext_vector_type(2)));
                    v2u sw = __builtin_amdgcn_permlane16_swap((unsigned)v32, (unsigned)v32, false, false);
                    const unsigned h64 = sw.x + sw.y;                                // rows {0,1} / {2,3}
                    sw = __builtin_amdgcn_permlane32_swap(h64, h64, false, false);
                    const int v64 = (int)(sw.x + sw.y);                              // column kcol of the 64x64 PU
                    if (SURF)
                    {
                        typedef int v4i __attribute__((ext_vector_type(4)));
                        // wave-uniform group base kept in SGPRs; lanes add a 32-bit byte offset (saddr addressing)
                        const u64 gofs = (u64)((((long)ctu * NC + m) * NG + g) * (PACKED ? 180 : 340)) * 4;
                        const u64 gsc = ((u64)__builtin_amdgcn_readfirstlane((uint32_t)(gofs >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)gofs);
                        char* grp = reinterpret_cast<char*>(a.surf) + gsc;
                        if (PACKED)
                        {
                            // the accumulators already ARE the u16[4] records of the 8x8 level
                            typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
                            const v2u32 v8 = { lo, hi };
                            *reinterpret_cast<v2u32*>(grp + (uint32_t)(lane * 8)) = v8;                        // 512 B per wavefront
                            *reinterpret_cast<uint16_t*>(grp + (uint32_t)(512 + lane * 2)) = (uint16_t)v16;    // 16x16 level, 128 B
                            if (uMask) *reinterpret_cast<int*>(grp + (uint32_t)(640 - 320 * 4 + uOffDw * 4)) = is64 ? v64 : v32;
                        }
                        else
                        {
                            const v4i v8 = { s8[0], s8[1], s8[2], s8[3] };
                            *reinterpret_cast<v4i*>(grp + (uint32_t)(lane * 16)) = v8;             // 1 KiB per wavefront store
                            *reinterpret_cast<int*>(grp + (uint32_t)(1024 + lane * 4)) = v16;      // records 64..79, 256 contiguous bytes
                            if (uMask) *reinterpret_cast<int*>(grp + (uint32_t)(uOffDw * 4)) = is64 ? v64 : v32;
                        }
                    }
                    if (BEST && (VAR & 1))
                    {
                        const uint32_t rb = ((uint32_t)a.costY[m] << 8) | (uint32_t)m;      // the row's share of every key: costY << 8 | m
                        const uint32_t k0 = ((lo & 0xffffu) << 8) + rb, k1 = ((lo >> 16) << 8) + rb;
                        const uint32_t k2 = ((hi & 0xffffu) << 8) + rb, k3 = ((hi >> 16) << 8) + rb;
                        r8c[0] = k0 < r8c[0] ? k0 : r8c[0];
                        r8c[1] = k1 < r8c[1] ? k1 : r8c[1];
                        r8c[2] = k2 < r8c[2] ? k2 : r8c[2];
                        r8c[3] = k3 < r8c[3] ? k3 : r8c[3];
                        const uint32_t base = cxL8 + rb;
                        const uint32_t k16 = ((uint32_t)v16 << 8) + base;
                        const uint32_t k32 = ((uint32_t)v32 << 8) + base;
                        const uint32_t k64 = ((uint32_t)v64 << 8) + base;
                        r16 = k16 < r16 ? k16 : r16;
                        r32 = k32 < r32 ? k32 : r32;
                        r64 = k64 < r64 ? k64 : r64;
                    }
                    else if (BEST)
                    {
                        const uint32_t cy_ = a.costY[m];
                        {   // 8x8: min over the 4 columns of (sad << 2) + (costX << 2 | k); costY is common to the row
                            const uint32_t k0 = ((uint32_t)s8[0] << 2) + cxk4[0], k1 = ((uint32_t)s8[1] << 2) + cxk4[1];
                            const uint32_t k2 = ((uint32_t)s8[2] << 2) + cxk4[2], k3 = ((uint32_t)s8[3] << 2) + cxk4[3];
                            uint32_t kmin = k0 < k1 ? k0 : k1;
                            kmin = k2 < kmin ? k2 : kmin;
                            kmin = k3 < kmin ? k3 : kmin;
                            const uint32_t rowc = (cy_ << 10) | ((uint32_t)m << 2);
                            uint32_t key = ((kmin & ~3u) << 8) + rowc;                   // cost << 10 | m << 2
                            key = (key & ~3u) | (kmin & 3u);                             // | k  (v_bfi)
                            r8 = key < r8 ? key : r8;
                        }
                        // (v + cxL + cy) << 8 | m  ==  (v << 8) + base with base = (cxL << 8) + ((cy << 8) | m): one v_lshl_add_u32 per level
                        const uint32_t base = cxL8 + ((cy_ << 8) | (uint32_t)m);
                        const uint32_t k16 = ((uint32_t)v16 << 8) + base;
                        const uint32_t k32 = ((uint32_t)v32 << 8) + base;
                        const uint32_t k64 = ((uint32_t)v64 << 8) + base;
                        r16 = k16 < r16 ? k16 : r16;
                        r32 = k32 < r32 ? k32 : r32;
                        r64 = k64 < r64 ? k64 : r64;
                    }
                }
                if (PIPE && (p & 1))
                {
#pragma unroll
                    for (int q = 0; q < 2; q++) { cur[q][0] = nxt[q][0]; cur[q][1] = nxt[q][1]; cur[q][2] = nxt[q][2]; }
                }
                // keep the scheduler from interleaving the emission code of different rows (it would keep
                // several rows' worth of unpacked sums alive and spill)
                if (!(VAR & 2)) __builtin_amdgcn_sched_barrier(0);
            }
        };
        using I8 = std::integral_constant<int, 8>;
        rows8(std::true_type{}, I8{}, 0);
        int t0 = 8;
        for (; t0 + 8 <= T; t0 += 8)
            rows8(std::false_type{}, I8{}, t0);
        switch (T - t0)
        {
        case 2: rows8(std::false_type{}, std::integral_constant<int, 2>{}, t0); break;
        case 4: rows8(std::false_type{}, std::integral_constant<int, 4>{}, t0); break;
        case 6: rows8(std::false_type{}, std::integral_constant<int, 6>{}, t0); break;
        default: break;
        }
        if (BEST)
        {   // widen the group's row-local keys to cost << 32 | raster index and merge
            if (VAR & 1)
            {   // each column's minimum over the rows gets its costX now; `cost << 10 | m << 2 | k` orders the four like (cost, raster index)
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const uint32_t kc = ((r8c[k] >> 8) << 2) + cxk4[k];
                    const uint32_t key = ((kc & ~3u) << 8) | ((r8c[k] & 255u) << 2) | (kc & 3u);
                    r8 = key < r8 ? key : r8;
                }
            }
            const u64 w8 = ((u64)(r8 >> 10) << 32) | (uint32_t)(((r8 >> 2) & 255u) * NC + 4 * g + (r8 & 3u));
            bk8 = w8 < bk8 ? w8 : bk8;
            auto widen = [&](const uint32_t r) { return ((u64)(r >> 8) << 32) | (uint32_t)((r & 255u) * NC + 4 * g + kcol); };
            const u64 w16 = widen(r16), w32 = widen(r32), w64 = widen(r64);
            bk16 = w16 < bk16 ? w16 : bk16;
            bk32 = w32 < bk32 ? w32 : bk32;
            bk64 = w64 < bk64 ? w64 : bk64;
        }
    }

    if (BEST)
    {
        u64* rec = a.best + (size_t)ctu * 85;
        atomicMin(&rec[lane], bk8);
        atomicMin(&rec[64 + (lane >> 2)], bk16);           // the 4 column lanes of a 16x16 PU merge here
        if ((lane & 15) < 4) atomicMin(&rec[80 + (lane >> 4)], bk32);
        if (lane < 4) atomicMin(&rec[84], bk64);
    }
}

// ------------------------------------------------------------------------------------------------
// Round 5: the minima-only 8-bit launch with the OTHER half of its instruction stream reworked (round-4 verdict, weak 2: 44 level-sum / key /
// minimum instructions per window row beside 16 quad-SADs; SQ_WAIT_INST_ANY 1.7 x SQ_ACTIVE).  Same walk, same ring of packed accumulators,
// same keys as me_ctu_q_kernel<best>; each change is a template flag so that it can be measured ALONE on one box (tools/r5_me_ab.sh,
// profiles/r05_me_flags_ab.txt) - FL = 0 is round 4's instruction stream:
//   1  LD64    the window bytes 0..7 / 4..11 of a row are read as two 4-byte-aligned 64-bit LDS loads into the aligned register pairs
//              v_qsad_pk_u16_u8 wants, a block of 8 rows from three address registers (the three-dword load costs a copy per row - gfx950
//              wants even-aligned 64-bit operands - and a register shuffle per pair of rows);
//   2  CTAB    the row's `costY[m] << 8 | m` comes from a table the workgroup builds once in LDS: a block of 8 rows fetches its entries one
//              block AHEAD, every row takes its entry with v_readlane -> row constants in SGPRs (round 4: a global_load_ushort + vmcnt wait
//              in every row's chain and five vector instructions to combine a uniform value);
//   4  PAIR64  the 64x64 level is reduced for TWO window rows at a time: v_permlane16_swap exchanges the odd 16-lane rows of row A's 32x32
//              sums with the even rows of row B's, one add, v_permlane32_swap of the sum with itself, one add -> lanes of rows 0 / 2 hold
//              A's total, rows 1 / 3 hold B's (5 instructions per PAIR of rows where two separate reductions take 14);
//   8  DEFERX  the column's costX joins the 16x16 / 32x32 / 64x64 running minima once per column group (min(x + c) = min(x) + c);
//  16  COLMIN  the 8x8 level keeps one running minimum per column and gives each column its costX once per group;
//  64  RING    the three-dword loads of round 4 without their copies: two pairs of rows in flight in registers whose role alternates (the unrolled
//              rows index them with constants), a block of 8 rows read from three address registers;
// 256  LDA     the window staged twice (the second copy four bytes on), row groups skewed by {0, 16} dwords: both 64-bit values of a row are 8-byte-aligned
//              ds_read_b64 from one copy or the other - LD64 without misaligned loads (ranges up to +-59);
// 128  QUAD64  PAIR64 for FOUR rows: two pair sums {A01,B01,A23,B23} / {C01,D01,C23,D23} change halves in one v_permlane32_swap -> 16-lane row r
//              holds the 64x64 total of the quad's row r (6 instructions and one key per four rows);
//  32  MASK    the low halves of the packed 8x8 sums are extracted with an opaque v_and (the compiler turns `(x & 0xffff) << n` into
//              shift + mask + add; this way it is mask + v_lshl_add_u32).
// Results are identical for every FL (tests/test_gpu_me.py runs them all against the oracle).
template <int PITCH, int FL>
__global__ void __launch_bounds__(1024, 4) me_ctu_q2_kernel(MEArgs a, int ctabOff, int groupsPerWg)
{
    constexpr bool LD64 = (FL & 1) != 0, CTAB = (FL & 2) != 0, PAIR64 = (FL & 4) != 0, DEFERX = (FL & 8) != 0, COLMIN = (FL & 16) != 0, MASK = (FL & 32) != 0;
    constexpr bool RING = (FL & 64) != 0, QUAD64 = (FL & 128) != 0, LDA = (FL & 256) != 0;
    static_assert(!LDA || (!LD64 && !RING), "LDA is a window-load scheme of its own");
    static_assert(!QUAD64 || (PAIR64 && CTAB), "QUAD64 extends PAIR64 and takes its per-lane row constants from the table");
    static_assert(!(RING && LD64), "RING is the three-dword load path without its copies");
    extern __shared__ __attribute__((aligned(16))) uint8_t win[];
    typedef unsigned long long u64;
    typedef u64 __attribute__((aligned(4))) u64a4;
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const int R = a.range;
    const int NC = 2 * R + 1;
    const int NG = (NC + 3) >> 2;
    const int rows = 64 + 2 * R;
    // XCD-aware order (round 6): a whole-picture launch hands every XCD a contiguous band of CTUs - the windows of neighbouring CTUs overlap by 2 R / (64 + 2 R) and meet in one L2
    const int ctu = (gridDim.y == 1 && a.xcdOrder) ? xcd_swizzle((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int cx = (ctu % a.ctusW) * 64, cy = (ctu / a.ctusW) * 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    constexpr int pitch = PITCH;

    const int ccx = a.centres ? a.centres[2 * ctu] : 0, ccy = a.centres ? a.centres[2 * ctu + 1] : 0;
    const uint8_t* g0 = a.fref + (long)(cy + ccy - R) * a.frefStrideB + (long)(cx + ccx - R);
    const int rowDw = a.payloadDw;
    // LDA: the window is staged TWICE - copy B is copy A moved on by four bytes - so that both 64-bit values of a row (window bytes 0..7 and 4..11 of a block column) are
    // 8-byte-ALIGNED loads from one copy or the other: for an even column group w0 comes from A and w1 from B, for an odd one the other way round.  The row pitch is 320
    // bytes (not a power of two) and groups of 8 rows are skewed by {0, 8, 16, 24} dwords: the LDS serves a 64-bit load 16 lanes at a time - 4 block columns x 4 block rows -
    // and the four block rows then land on four disjoint sets of 8 banks (a first version skewed by {0, 16} and lost 5.8e7 cycles per launch to bank conflicts).
    const int ldaB = LDA ? (rows + 2) * pitch : 0;
    for (int r = wave; r < rows; r += nwaves)
    {
        const uint8_t* src = g0 + (long)r * a.frefStrideB;
        uint32_t* dst = reinterpret_cast<uint32_t*>(win + (LDA ? r * pitch + ((r >> 3) & 3) * 32 : lds_row_off(r, pitch)));
        for (int c = lane; c < rowDw; c += 64)
        {
            dst[c] = ld_u32(src + 4 * c);
            if (LDA) dst[c + ldaB / 4] = ld_u32(src + 4 * c + 4);
        }
    }
    // CTAB: the rows' share of every key, costY[m] << 8 | m (16 entries of slack: a block fetches the NEXT block's entries)
    uint32_t* ctab = reinterpret_cast<uint32_t*>(win + ctabOff);
    if (CTAB)
        for (int i = threadIdx.x; i < NC + 16; i += blockDim.x)
            ctab[i] = i < NC ? ((uint32_t)a.costY[i] << 8) | (uint32_t)i : 0xffffff00u;
    int bx, by;
    zorder_xy(lane, bx, by);
    uint32_t F[8][2];
    {
        const uint8_t* fe = a.fenc + (long)(cy + by * 8) * a.fencStrideB + (long)(cx + bx * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) { F[j][0] = ld_u32(fe + (long)j * a.fencStrideB); F[j][1] = ld_u32(fe + (long)j * a.fencStrideB + 4); }
    }
    __syncthreads();

    u64 bk8 = ~0ull, bk16 = ~0ull, bk32 = ~0ull, bk64 = ~0ull;
    const int kcol = lane & 3;
    const uint32_t selK = 0x0c0c0000u | (uint32_t)((2 * kcol + 1) << 8) | (uint32_t)(2 * kcol);   // v_perm: u16 #kcol of {qhi, qlo}
    const int oddRow = (lane >> 4) & 1;                     // PAIR64: 0 = this lane ends up with row A's 64x64 total, 1 = row B's

    const int T = 2 * R + 8;
    // a launch of few CTUs (a band of the frame-parallel ring, a small picture) deals a CTU's column groups over blockIdx.y workgroups - each stages the window for
    // itself, the 64-bit atomic minima merge them like they merge the wavefronts of one workgroup - so that the chip is filled (groupsPerWg = 0: one workgroup per CTU)
    const int gPer = groupsPerWg > 0 ? groupsPerWg : NG;
    const int gFirst = (int)blockIdx.y * gPer, gEnd = gFirst + gPer < NG ? gFirst + gPer : NG;
    for (int g = gFirst + wave; g < gEnd; g += nwaves)
    {
        const uint32_t colOff = (uint32_t)((by * 8) * pitch + bx * 8 + 4 * g);
        // LDS byte offset of window row t0 of this lane's block column.  LD64 keeps it opaque: ds_read reaches 255 dwords past its address
        // register, so a block of 8 rows is read from THREE address registers (rows 0 - 3, rows 4 - 7, the next block's rows 0 - 1)
        auto block_off = [&](const int t0)
        {
            uint32_t o = colOff + (uint32_t)(t0 * pitch + (LDA ? ((by + (t0 >> 3)) & 3) * 32 : lds_skew_bytes(by + (t0 >> 3))));
            if (LD64 || RING || LDA) asm volatile("" : "+v"(o));
            return o;
        };
        // LDA: where this group's w0 / w1 live relative to block_off (which points at copy A, column 4 g): an even group reads w0 = A[4 g], w1 = B[4 g];
        // an odd one w0 = B[4 (g - 1)] (= A[4 g]), w1 = A[4 (g + 1)] - every address a multiple of 8
        const int ldaW0 = LDA ? ((g & 1) ? ldaB - 4 : 0) : 0, ldaW1 = LDA ? ((g & 1) ? 4 : ldaB) : 0;
        auto ldpairA = [&](u64 (&d)[2][2], const uint32_t off, const int p)
        {
#pragma unroll
            for (int q = 0; q < 2; q++)
            {
                d[q][0] = *reinterpret_cast<const u64*>(win + off + ldaW0 + (p + q) * pitch);
                d[q][1] = *reinterpret_cast<const u64*>(win + off + ldaW1 + (p + q) * pitch);
            }
        };
        auto ldpair64 = [&](u64 (&d)[2][2], const uint32_t off, const int p)
        {
#pragma unroll
            for (int q = 0; q < 2; q++)
            {
                d[q][0] = *reinterpret_cast<const u64a4*>(win + off + (p + q) * pitch);
                d[q][1] = *reinterpret_cast<const u64a4*>(win + off + (p + q) * pitch + 4);
            }
        };
        auto ldpair32 = [&](uint32_t (&d)[2][3], const uint32_t off, const int p)
        {
#pragma unroll
            for (int q = 0; q < 2; q++)
            {
                const uint32_t* lp = reinterpret_cast<const uint32_t*>(win + off + (p + q) * pitch);
                d[q][0] = lp[0]; d[q][1] = lp[1]; d[q][2] = lp[2];
            }
        };
        uint32_t cxk4[4], cxL;
#pragma unroll
        for (int k = 0; k < 4; k++)
            cxk4[k] = ((4 * g + k < NC ? (uint32_t)a.costX[4 * g + k] : (1u << 20)) << 2) | (uint32_t)k;
        cxL = 4 * g + kcol < NC ? (uint32_t)a.costX[4 * g + kcol] : (1u << 23);
        const uint32_t cxL8 = cxL << 8;
        uint32_t r8 = 0xffffffffu, r16 = 0xffffffffu, r32 = 0xffffffffu, r64 = 0xffffffffu;
        uint32_t r8c[4] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu };
        u64 acc[8];
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = 0;
        u64 buf[2][2][2];                                   // LD64: [pipeline slot][row of the pair][bytes 0..7 / 4..11]; slot parity returns after 8 rows
        uint32_t cur[2][3], nxt[2][3];                      // !LD64: round 4's three dwords per row
        uint32_t d3[2][2][3];                               // RING: the same three dwords, two pairs of rows in flight, no copies (slot parity as above)
        (void)buf; (void)cur; (void)nxt; (void)d3;
        if (LDA) ldpairA(buf[0], block_off(0), 0); else if (LD64) ldpair64(buf[0], block_off(0), 0); else if (RING) ldpair32(d3[0], block_off(0), 0); else ldpair32(cur, block_off(0), 0);

        // CTAB: a block's row constants, fetched while the block before it runs.  lane l holds the entry of window row t0 + (l & 7)
        // (m = t0 + (l & 7) - 7; the first block only completes m = 0), pr[q] the entry of the row this lane holds after the paired reduction
        // QUAD64: pr[0] / pr[1] = the entries of window rows 0..3 / 4..7 by the lane's 16-lane row, pr[2] / pr[3] = those of a tail pair (rows 0,1 / 4,5)
        uint32_t cbNext = 0, prNext[4] = { 0, 0, 0, 0 };
        auto fetch_consts = [&](const int t0)
        {
            const int i0 = t0 - 7 + (lane & 7);
            cbNext = ctab[i0 < 0 ? 0 : i0];
            if (QUAD64)
            {
                const int a0 = t0 - 7 + (lane >> 4), a1 = t0 - 7 + oddRow;
                prNext[0] = ctab[a0 < 0 ? 0 : a0]; prNext[1] = ctab[a0 + 4 < 0 ? 0 : a0 + 4];
                prNext[2] = ctab[a1 < 0 ? 0 : a1]; prNext[3] = ctab[a1 + 4 < 0 ? 0 : a1 + 4];
            }
            else if (PAIR64)
            {
#pragma unroll
                for (int q = 0; q < 4; q++) { const int i = t0 - 7 + 2 * q + oddRow; prNext[q] = ctab[i < 0 ? 0 : i]; }
            }
        };
        if (CTAB) fetch_consts(0);

        auto rows8 = [&](auto firstTag, auto nrowsTag, const int t0)
        {
            constexpr bool FIRST = decltype(firstTag)::value;
            constexpr int NROWS = decltype(nrowsTag)::value;
            const uint32_t bb = block_off(t0), bn = block_off(t0 + 8);
            uint32_t bb4 = bb + 4 * pitch;
            if (LD64 || RING) asm volatile("" : "+v"(bb4));
            const int mb = t0 - 7;
            uint32_t cb = 0, pr[4] = { 0, 0, 0, 0 };
            if (CTAB)
            {
                cb = cbNext;
#pragma unroll
                for (int q = 0; q < 4; q++) pr[q] = prNext[q];
                fetch_consts(t0 + 8);
            }
            uint32_t v32even = 0, sbEven = 0, pairSum = 0;
#pragma unroll
            for (int p = 0; p < NROWS; p++)
            {
                if ((p & 1) == 0)
                {
                    if (LDA) { if (p + 2 < 8) ldpairA(buf[((p >> 1) + 1) & 1], bb, p + 2); else ldpairA(buf[((p >> 1) + 1) & 1], bn, 0); }
                    else if (LD64)
                    {
                        if (p + 2 < 4) ldpair64(buf[((p >> 1) + 1) & 1], bb, p + 2);
                        else if (p + 2 < 8) ldpair64(buf[((p >> 1) + 1) & 1], bb4, p + 2 - 4);
                        else ldpair64(buf[((p >> 1) + 1) & 1], bn, 0);
                    }
                    else if (RING)
                    {
                        if (p + 2 < 4) ldpair32(d3[((p >> 1) + 1) & 1], bb, p + 2);
                        else if (p + 2 < 8) ldpair32(d3[((p >> 1) + 1) & 1], bb4, p + 2 - 4);
                        else ldpair32(d3[((p >> 1) + 1) & 1], bn, 0);
                    }
                    else { if (p + 2 < 8) ldpair32(nxt, bb, p + 2); else ldpair32(nxt, bn, 0); }
                }
                u64 w0, w1;
                if (LD64 || LDA) { w0 = buf[(p >> 1) & 1][p & 1][0]; w1 = buf[(p >> 1) & 1][p & 1][1]; }
                else if (RING)
                {
                    const uint32_t (&dd)[3] = d3[(p >> 1) & 1][p & 1];
                    w0 = ((u64)dd[1] << 32) | dd[0]; w1 = ((u64)dd[2] << 32) | dd[1];
                }
                else { w0 = ((u64)cur[p & 1][1] << 32) | cur[p & 1][0]; w1 = ((u64)cur[p & 1][2] << 32) | cur[p & 1][1]; }
#pragma unroll
                for (int j = 0; j < 8; j++)
                {
                    if (FIRST && j > p) continue;
                    u64 v = j == 0 ? 0ull : acc[(p - j) & 7];
                    v = __builtin_amdgcn_qsad_pk_u16_u8(w0, F[j][0], v);
                    v = __builtin_amdgcn_qsad_pk_u16_u8(w1, F[j][1], v);
                    acc[(p - j) & 7] = v;
                }
                if (!FIRST || p == 7)
                {
                    const int m = mb + p;
                    const int slot = (p + 1) & 7;
                    const u64 A = acc[slot];
                    const uint32_t lo = (uint32_t)A, hi = (uint32_t)(A >> 32);
                    const uint32_t qlo = (uint32_t)quad_sum((int)lo), qhi = (uint32_t)quad_sum((int)hi);
                    uint32_t s80, s82;
                    if (MASK)
                    {
                        asm("v_and_b32 %0, 0xffff, %1" : "=v"(s80) : "v"(lo));
                        asm("v_and_b32 %0, 0xffff, %1" : "=v"(s82) : "v"(hi));
                    }
                    else { s80 = lo & 0xffffu; s82 = hi & 0xffffu; }
                    const uint32_t s81 = lo >> 16, s83 = hi >> 16;
                    const uint32_t v16 = __builtin_amdgcn_perm(qhi, qlo, selK);
                    const uint32_t v32 = (uint32_t)row_sum_of_quads((int)v16);
                    // costY[m] << 8 | m: an SGPR from the block's table entries, or built from round 4's per-row load
                    const uint32_t sb = CTAB ? (uint32_t)__builtin_amdgcn_readlane((int)cb, p) : (((uint32_t)a.costY[m] << 8) | (uint32_t)m);
                    if (COLMIN)
                    {
                        const uint32_t k0 = (s80 << 8) + sb, k1 = (s81 << 8) + sb, k2 = (s82 << 8) + sb, k3 = (s83 << 8) + sb;
                        r8c[0] = k0 < r8c[0] ? k0 : r8c[0];
                        r8c[1] = k1 < r8c[1] ? k1 : r8c[1];
                        r8c[2] = k2 < r8c[2] ? k2 : r8c[2];
                        r8c[3] = k3 < r8c[3] ? k3 : r8c[3];
                    }
                    else
                    {
                        const uint32_t k0 = (s80 << 2) + cxk4[0], k1 = (s81 << 2) + cxk4[1];
                        const uint32_t k2 = (s82 << 2) + cxk4[2], k3 = (s83 << 2) + cxk4[3];
                        uint32_t kmin = k0 < k1 ? k0 : k1;
                        kmin = k2 < kmin ? k2 : kmin;
                        kmin = k3 < kmin ? k3 : kmin;
                        uint32_t key = ((kmin & ~3u) << 8) + (sb << 2);              // cost << 10 | m << 2
                        key = (key & ~3u) | (kmin & 3u);                             // | k
                        r8 = key < r8 ? key : r8;
                    }
                    const uint32_t base = DEFERX ? sb : cxL8 + sb;
                    const uint32_t k16 = (v16 << 8) + base, k32 = (v32 << 8) + base;
                    r16 = k16 < r16 ? k16 : r16;
                    r32 = k32 < r32 ? k32 : r32;
                    if (!PAIR64 || FIRST)
                    {   // one row on its own (PAIR64: the first block completes a single row, m = 0)
                        v2u sw = __builtin_amdgcn_permlane16_swap(v32, v32, false, false);
                        const unsigned h64 = sw.x + sw.y;
                        sw = __builtin_amdgcn_permlane32_swap(h64, h64, false, false);
                        const uint32_t k64 = ((sw.x + sw.y) << 8) + base;
                        r64 = k64 < r64 ? k64 : r64;
                    }
                    else if ((p & 1) == 0) { v32even = v32; sbEven = sb; }
                    else if (QUAD64)
                    {
                        // rows A (p - 1) and B (p): {A0,B0,A2,B2} + {A1,B1,A3,B3} = {A01, B01, A23, B23}
                        v2u sw = __builtin_amdgcn_permlane16_swap(v32even, v32, false, false);
                        const unsigned h64 = sw.x + sw.y;
                        const bool quadEnd = (p & 3) == 3, tailPair = (p & 3) == 1 && p + 2 >= NROWS;       // constants once the row loop is unrolled
                        if (quadEnd)
                        {   // with the pair before it {C01, D01, C23, D23}: the halves change places -> 16-lane row r holds the total of the quad's row r
                            sw = __builtin_amdgcn_permlane32_swap(pairSum, h64, false, false);
                            uint32_t mine = pr[p >> 2];
                            if (!DEFERX) mine += cxL8;
                            const uint32_t k64 = ((sw.x + sw.y) << 8) + mine;
                            r64 = k64 < r64 ? k64 : r64;
                        }
                        else if (tailPair)
                        {   // the block ends with this pair (2 or 6 rows left): finished on its own, rows 0 / 2 hold A's total, rows 1 / 3 B's
                            sw = __builtin_amdgcn_permlane32_swap(h64, h64, false, false);
                            uint32_t mine = pr[2 + (p >> 2)];
                            if (!DEFERX) mine += cxL8;
                            const uint32_t k64 = ((sw.x + sw.y) << 8) + mine;
                            r64 = k64 < r64 ? k64 : r64;
                        }
                        else pairSum = h64;
                    }
                    else
                    {
                        // rows A (p - 1) and B (p): {A0,B0,A2,B2} + {A1,B1,A3,B3}, then the two halves of that sum
                        v2u sw = __builtin_amdgcn_permlane16_swap(v32even, v32, false, false);
                        const unsigned h64 = sw.x + sw.y;
                        sw = __builtin_amdgcn_permlane32_swap(h64, h64, false, false);
                        const uint32_t tot = sw.x + sw.y;                            // 16-lane rows 0 / 2: A's 64x64 total, rows 1 / 3: B's
                        uint32_t mine = CTAB ? pr[p >> 1] : (oddRow ? sb : sbEven);  // costY << 8 | m of the row this lane holds
                        if (!DEFERX) mine += cxL8;
                        const uint32_t k64 = (tot << 8) + mine;
                        r64 = k64 < r64 ? k64 : r64;
                    }
                }
                if (!LD64 && !RING && !LDA && (p & 1))
                {
#pragma unroll
                    for (int q = 0; q < 2; q++) { cur[q][0] = nxt[q][0]; cur[q][1] = nxt[q][1]; cur[q][2] = nxt[q][2]; }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        using I8 = std::integral_constant<int, 8>;
        rows8(std::true_type{}, I8{}, 0);
        int t0 = 8;
        for (; t0 + 8 <= T; t0 += 8)
            rows8(std::false_type{}, I8{}, t0);
        switch (T - t0)
        {
        case 2: rows8(std::false_type{}, std::integral_constant<int, 2>{}, t0); break;
        case 4: rows8(std::false_type{}, std::integral_constant<int, 4>{}, t0); break;
        case 6: rows8(std::false_type{}, std::integral_constant<int, 6>{}, t0); break;
        default: break;
        }
        {   // widen the group's row-local keys to cost << 32 | raster index and merge
            if (COLMIN)
            {   // each column's minimum over the rows gets its costX now; `cost << 10 | m << 2 | k` orders the four like (cost, raster index)
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const uint32_t kc = ((r8c[k] >> 8) << 2) + cxk4[k];
                    const uint32_t key = ((kc & ~3u) << 8) | ((r8c[k] & 255u) << 2) | (kc & 3u);
                    r8 = key < r8 ? key : r8;
                }
            }
            const u64 w8 = ((u64)(r8 >> 10) << 32) | (uint32_t)(((r8 >> 2) & 255u) * NC + 4 * g + (r8 & 3u));
            bk8 = w8 < bk8 ? w8 : bk8;
            auto widen = [&](const uint32_t r) { const uint32_t rc = DEFERX ? r + cxL8 : r; return ((u64)(rc >> 8) << 32) | (uint32_t)((rc & 255u) * NC + 4 * g + kcol); };
            const u64 w16 = widen(r16), w32 = widen(r32), w64 = widen(r64);
            bk16 = w16 < bk16 ? w16 : bk16;
            bk32 = w32 < bk32 ? w32 : bk32;
            bk64 = w64 < bk64 ? w64 : bk64;
        }
    }

    u64* rec = a.best + (size_t)ctu * 85;
    atomicMin(&rec[lane], bk8);
    atomicMin(&rec[64 + (lane >> 2)], bk16);
    if ((lane & 15) < 4) atomicMin(&rec[80 + (lane >> 4)], bk32);
    // PAIR64: lanes 0..3 saw the even window rows of their column, lanes 16..19 the odd ones; QUAD64: every 16-lane row saw its own rows
    if ((lane & (QUAD64 ? 0x0f : PAIR64 ? 0x2f : 0x3f)) < 4) atomicMin(&rec[84], bk64);
}

// ------------------------------------------------------------------------------------------------
// 10-bit fast path: the column-group organisation of me_ctu_q_kernel for 16-bit pixels.  There is no packed
// multi-displacement SAD for 16-bit samples, so the core is v_sad_u16 (2 pixels, 8.8 cycles): a wavefront owns 4 mv
// columns, each window row is 6 dwords (12 pixels: even columns read pixel pairs as stored, odd columns through
// v_alignbit 16) feeding 8 ring slots x 4 columns x 4 dwords = 128 v_sad_u16; the emission (transposed upper levels,
// 16-byte group stores, row-local 32-bit keys) is the 8-bit kernel's.  Key widths hold for depth <= 10
// (64x64 SAD + mv cost < 2^23); 12-bit pictures use the generic kernel.
template <bool SURF, bool BEST, int PITCH, int VAR = 1>
__global__ void __launch_bounds__(SURF && BEST ? 768 : 1024, SURF && BEST ? 3 : 4) me_ctu_w_kernel(MEArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t win[];
    typedef unsigned long long u64;

    const int R = a.range;
    const int NC = 2 * R + 1;
    const int NG = (NC + 3) >> 2;
    const int rows = 64 + 2 * R;
    // XCD-aware order (round 6): a whole-picture launch hands every XCD a contiguous band of CTUs - the windows of neighbouring CTUs overlap by 2 R / (64 + 2 R) and meet in one L2
    const int ctu = (gridDim.y == 1 && a.xcdOrder) ? xcd_swizzle((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int cx = (ctu % a.ctusW) * 64, cy = (ctu / a.ctusW) * 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    constexpr int pitch = PITCH;

    const int ccx = a.centres ? a.centres[2 * ctu] : 0, ccy = a.centres ? a.centres[2 * ctu + 1] : 0;
    const uint8_t* g0 = a.fref + (long)(cy + ccy - R) * a.frefStrideB + (long)(cx + ccx - R) * 2;
    const int rowDw = a.payloadDw;
    for (int r = wave; r < rows; r += nwaves)
    {
        const uint8_t* src = g0 + (long)r * a.frefStrideB;
        uint32_t* dst = reinterpret_cast<uint32_t*>(win + lds_row_off(r, pitch));
        for (int c = lane; c < rowDw; c += 64)
            dst[c] = ld_u32(src + 4 * c);
    }
    int bx, by;
    zorder_xy(lane, bx, by);
    uint32_t F[8][4];
    {
        const uint8_t* fe = a.fenc + (long)(cy + by * 8) * a.fencStrideB + (long)(cx + bx * 8) * 2;
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int k = 0; k < 4; k++) F[j][k] = ld_u32(fe + (long)j * a.fencStrideB + 4 * k);
    }
    __syncthreads();

    u64 bk8 = ~0ull, bk16 = ~0ull, bk32 = ~0ull, bk64 = ~0ull;
    const int kcol = lane & 3;
    const bool is64 = lane >= 52 && lane < 56;
    const bool uMask = (lane & 15) < 4 || is64;
    const int uOffDw = is64 ? 84 * 4 + kcol : (80 + (lane >> 4)) * 4 + kcol;

    const int T = 2 * R + 8;
    for (int g = wave; g < NG; g += nwaves)
    {
        const uint8_t* colBase = win + (by * 8) * pitch + (bx * 8 + 4 * g) * 2;
        // (round 4) the 8x8 level keeps one running minimum PER COLUMN of `(sad + costY) << 8 | m`: a column's costX is the same for every row of the
        // group, so it joins once, at the end of the group - 8 instead of 12 instructions per row for the level, 18 -> 12 spilled registers; 4K 2.67 -> 2.62 ms,
        // 8K 10.32 -> 10.05 ms.  The same change made the 8-bit kernel SLOWER (1.43 -> 1.50 ms with v_mad_u32_u16 keys - a quarter-rate instruction on
        // gfx950 - and 1.55 ms with and / shift-add keys, 17 % fewer instructions either way): there the key arithmetic was independent filler between the
        // dependent DPP / permlane steps of the level sums, and without it their latency shows (profiles/r04_me_minima_ab.txt).  Left as it was.
        uint32_t cxk4[4] = { 0, 0, 0, 0 }, cxL = 0;
        uint32_t r8c[4] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu };
        uint32_t r8 = 0xffffffffu, r16 = 0xffffffffu, r32 = 0xffffffffu, r64 = 0xffffffffu;
        if (BEST)
        {
#pragma unroll
            for (int k = 0; k < 4; k++)
                cxk4[k] = ((4 * g + k < NC ? (uint32_t)a.costX[4 * g + k] : (1u << 20)) << 2) | (uint32_t)k;
            cxL = 4 * g + kcol < NC ? (uint32_t)a.costX[4 * g + kcol] : (1u << 23);
        }
        const uint32_t cxL8 = cxL << 8;
        uint32_t acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int k = 0; k < 4; k++) acc[i][k] = 0;

        auto row_step = [&](auto firstTag, const int t0, const int p)
        {
            constexpr bool FIRST = decltype(firstTag)::value;
            const int t = t0 + p;
            const uint32_t* lp = reinterpret_cast<const uint32_t*>(colBase + t * pitch + lds_skew_bytes(by + (t >> 3)));
            uint32_t d[6], e[5];
#pragma unroll
            for (int k = 0; k < 6; k++) d[k] = lp[k];
#pragma unroll
            for (int k = 0; k < 5; k++) e[k] = __builtin_amdgcn_alignbit(d[k + 1], d[k], 16);
#pragma unroll
            for (int j = 0; j < 8; j++)
            {
                if (FIRST && j > p) continue;
                const int slot = (p - j) & 7;
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    acc[slot][0] = __builtin_amdgcn_sad_u16(d[k], F[j][k], acc[slot][0]);
                    acc[slot][1] = __builtin_amdgcn_sad_u16(e[k], F[j][k], acc[slot][1]);
                    acc[slot][2] = __builtin_amdgcn_sad_u16(d[k + 1], F[j][k], acc[slot][2]);
                    acc[slot][3] = __builtin_amdgcn_sad_u16(e[k + 1], F[j][k], acc[slot][3]);
                }
            }
            if (!FIRST || p == 7)
            {
                const int m = t - 7;
                const int slot = (p + 1) & 7;
                int s8[4], s16[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { s8[k] = (int)acc[slot][k]; acc[slot][k] = 0; s16[k] = quad_sum(s8[k]); }
                const int v16 = kcol == 0 ? s16[0] : (kcol == 1 ? s16[1] : (kcol == 2 ? s16[2] : s16[3]));   // column kcol of 16x16 PU (lane >> 2)
                const int v32 = row_sum_of_quads(v16);
                typedef unsigned v2u __attribute__((ext_vector_type(2)));
                v2u sw = __builtin_amdgcn_permlane16_swap((unsigned)v32, (unsigned)v32, false, false);
                const unsigned h64 = sw.x + sw.y;
                sw = __builtin_amdgcn_permlane32_swap(h64, h64, false, false);
                const int v64 = (int)(sw.x + sw.y);
                if (SURF)
                {
                    typedef int v4i __attribute__((ext_vector_type(4)));
                    const u64 gofs = (u64)((((long)ctu * NC + m) * NG + g) * 340) * 4;
                    const u64 gsc = ((u64)__builtin_amdgcn_readfirstlane((uint32_t)(gofs >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)gofs);
                    char* grp = reinterpret_cast<char*>(a.surf) + gsc;
                    const v4i v8 = { s8[0], s8[1], s8[2], s8[3] };
                    *reinterpret_cast<v4i*>(grp + (uint32_t)(lane * 16)) = v8;
                    *reinterpret_cast<int*>(grp + (uint32_t)(1024 + lane * 4)) = v16;
                    if (uMask) *reinterpret_cast<int*>(grp + (uint32_t)(uOffDw * 4)) = is64 ? v64 : v32;
                }
                if (BEST && !(VAR & 1))
                {   // round 3's scheme: one key per row
                    const uint32_t cy_ = a.costY[m];
                    {
                        const uint32_t k0 = ((uint32_t)s8[0] << 2) + cxk4[0], k1 = ((uint32_t)s8[1] << 2) + cxk4[1];
                        const uint32_t k2 = ((uint32_t)s8[2] << 2) + cxk4[2], k3 = ((uint32_t)s8[3] << 2) + cxk4[3];
                        uint32_t kmin = k0 < k1 ? k0 : k1;
                        kmin = k2 < kmin ? k2 : kmin;
                        kmin = k3 < kmin ? k3 : kmin;
                        const uint32_t rowc = (cy_ << 10) | ((uint32_t)m << 2);
                        uint32_t key = ((kmin & ~3u) << 8) + rowc;
                        key = (key & ~3u) | (kmin & 3u);
                        r8 = key < r8 ? key : r8;
                    }
                    const uint32_t cxy = cxL + cy_;
                    const uint32_t k16 = (((uint32_t)v16 + cxy) << 8) | (uint32_t)m;
                    const uint32_t k32 = (((uint32_t)v32 + cxy) << 8) | (uint32_t)m;
                    const uint32_t k64 = (((uint32_t)v64 + cxy) << 8) | (uint32_t)m;
                    r16 = k16 < r16 ? k16 : r16;
                    r32 = k32 < r32 ? k32 : r32;
                    r64 = k64 < r64 ? k64 : r64;
                }
                else if (BEST)
                {
                    const uint32_t rb = ((uint32_t)a.costY[m] << 8) | (uint32_t)m;          // the row's share of every key: costY << 8 | m
#pragma unroll
                    for (int k = 0; k < 4; k++)
                    {
                        const uint32_t kk = ((uint32_t)s8[k] << 8) + rb;                     // (sad + costY) << 8 | m (an 8x8 SAD of 12-bit samples is < 2^18)
                        r8c[k] = kk < r8c[k] ? kk : r8c[k];
                    }
                    const uint32_t base = cxL8 + rb;
                    const uint32_t k16 = ((uint32_t)v16 << 8) + base;
                    const uint32_t k32 = ((uint32_t)v32 << 8) + base;
                    const uint32_t k64 = ((uint32_t)v64 << 8) + base;
                    r16 = k16 < r16 ? k16 : r16;
                    r32 = k32 < r32 ? k32 : r32;
                    r64 = k64 < r64 ? k64 : r64;
                }
            }
            if (!(VAR & 2)) __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int p = 0; p < 8; p++) row_step(std::true_type{}, 0, p);
        for (int t0 = 8; t0 < T; t0 += 8)
        {
#pragma unroll
            for (int p = 0; p < 8; p++)
                if (t0 + p < T) row_step(std::false_type{}, t0, p);
        }
        if (BEST)
        {
            if (VAR & 1)
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const uint32_t kc = ((r8c[k] >> 8) << 2) + cxk4[k];                      // (sad + costY + costX) << 2 | k
                const uint32_t key = ((kc & ~3u) << 8) | ((r8c[k] & 255u) << 2) | (kc & 3u);
                r8 = key < r8 ? key : r8;
            }
            const u64 w8 = ((u64)(r8 >> 10) << 32) | (uint32_t)(((r8 >> 2) & 255u) * NC + 4 * g + (r8 & 3u));
            bk8 = w8 < bk8 ? w8 : bk8;
            auto widen = [&](const uint32_t r) { return ((u64)(r >> 8) << 32) | (uint32_t)((r & 255u) * NC + 4 * g + kcol); };
            const u64 w16 = widen(r16), w32 = widen(r32), w64 = widen(r64);
            bk16 = w16 < bk16 ? w16 : bk16;
            bk32 = w32 < bk32 ? w32 : bk32;
            bk64 = w64 < bk64 ? w64 : bk64;
        }
    }
    if (BEST)
    {
        u64* rec = a.best + (size_t)ctu * 85;
        atomicMin(&rec[lane], bk8);
        atomicMin(&rec[64 + (lane >> 2)], bk16);
        if ((lane & 15) < 4) atomicMin(&rec[80 + (lane >> 4)], bk32);
        if (lane < 4) atomicMin(&rec[84], bk64);
    }
}

// Round 5: the minima-only 16-bit launch with what paid off in the 8-bit kernel (me_ctu_q2_kernel flags CTAB, DEFERX, QUAD64; the per-column 8x8 minima
// were this kernel's already): the row's `costY << 8 | m` from an LDS table through v_readlane (fetched a block of 8 rows ahead) instead of a
// global_load_ushort + wait in every row's chain, costX once per column group, the 64x64 level reduced for four window rows at a time, the window rows of a
// block addressed from one register.  Same SADs, same keys, identical results (tests/test_gpu_me.py); X265HIP_ME_W2=0 selects round 4's kernel (A/B).
template <int PITCH>
__global__ void __launch_bounds__(1024, 4) me_ctu_w2_kernel(MEArgs a, int ctabOff, int groupsPerWg)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t win[];
    typedef unsigned long long u64;
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const int R = a.range;
    const int NC = 2 * R + 1;
    const int NG = (NC + 3) >> 2;
    const int rows = 64 + 2 * R;
    // XCD-aware order (round 6): a whole-picture launch hands every XCD a contiguous band of CTUs - the windows of neighbouring CTUs overlap by 2 R / (64 + 2 R) and meet in one L2
    const int ctu = (gridDim.y == 1 && a.xcdOrder) ? xcd_swizzle((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int cx = (ctu % a.ctusW) * 64, cy = (ctu / a.ctusW) * 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    constexpr int pitch = PITCH;

    const int ccx = a.centres ? a.centres[2 * ctu] : 0, ccy = a.centres ? a.centres[2 * ctu + 1] : 0;
    const uint8_t* g0 = a.fref + (long)(cy + ccy - R) * a.frefStrideB + (long)(cx + ccx - R) * 2;
    const int rowDw = a.payloadDw;
    for (int r = wave; r < rows; r += nwaves)
    {
        const uint8_t* src = g0 + (long)r * a.frefStrideB;
        uint32_t* dst = reinterpret_cast<uint32_t*>(win + lds_row_off(r, pitch));
        for (int c = lane; c < rowDw; c += 64)
            dst[c] = ld_u32(src + 4 * c);
    }
    uint32_t* ctab = reinterpret_cast<uint32_t*>(win + ctabOff);
    for (int i = threadIdx.x; i < NC + 16; i += blockDim.x)
        ctab[i] = i < NC ? ((uint32_t)a.costY[i] << 8) | (uint32_t)i : 0xffffff00u;
    int bx, by;
    zorder_xy(lane, bx, by);
    uint32_t F[8][4];
    {
        const uint8_t* fe = a.fenc + (long)(cy + by * 8) * a.fencStrideB + (long)(cx + bx * 8) * 2;
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int k = 0; k < 4; k++) F[j][k] = ld_u32(fe + (long)j * a.fencStrideB + 4 * k);
    }
    __syncthreads();

    u64 bk8 = ~0ull, bk16 = ~0ull, bk32 = ~0ull, bk64 = ~0ull;
    const int kcol = lane & 3;
    const int oddRow = (lane >> 4) & 1;
    const bool lane1 = (lane & 1) != 0, lane2 = (lane & 2) != 0;

    const int T = 2 * R + 8;
    const int gPer = groupsPerWg > 0 ? groupsPerWg : NG;          // (see me_ctu_q2_kernel: a CTU's column groups dealt over blockIdx.y workgroups when the launch has few CTUs)
    const int gFirst = (int)blockIdx.y * gPer, gEnd = gFirst + gPer < NG ? gFirst + gPer : NG;
    for (int g = gFirst + wave; g < gEnd; g += nwaves)
    {
        const uint32_t colOff = (uint32_t)((by * 8) * pitch + (bx * 8 + 4 * g) * 2);
        auto block_off = [&](const int t0) { uint32_t o = colOff + (uint32_t)(t0 * pitch + lds_skew_bytes(by + (t0 >> 3))); asm volatile("" : "+v"(o)); return o; };
        uint32_t cxk4[4], cxL;
#pragma unroll
        for (int k = 0; k < 4; k++)
            cxk4[k] = ((4 * g + k < NC ? (uint32_t)a.costX[4 * g + k] : (1u << 20)) << 2) | (uint32_t)k;
        cxL = 4 * g + kcol < NC ? (uint32_t)a.costX[4 * g + kcol] : (1u << 23);
        const uint32_t cxL8 = cxL << 8;
        uint32_t r8c[4] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu };
        uint32_t r8 = 0xffffffffu, r16 = 0xffffffffu, r32 = 0xffffffffu, r64 = 0xffffffffu;
        uint32_t acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int k = 0; k < 4; k++) acc[i][k] = 0;

        // row constants: lane l takes the table entry of window row t0 + (l & 7) at the top of a block (its first use is a row of 128 SADs away); the entry of
        // the row a lane ends up holding after the four-row 64x64 reduction is read when the quad starts - registers are what this kernel is short of
        auto tab = [&](const int i) { return ctab[i < 0 ? 0 : i]; };
        auto rows8 = [&](auto firstTag, auto nrowsTag, const int t0)
        {
            constexpr bool FIRST = decltype(firstTag)::value;
            constexpr int NROWS = decltype(nrowsTag)::value;
            const uint32_t bb = block_off(t0);
            const uint32_t cb = tab(t0 - 7 + (lane & 7));
            uint32_t v32even = 0, pairSum = 0, prq = 0;
#pragma unroll
            for (int p = 0; p < NROWS; p++)
            {
                if (!FIRST && (p & 3) == 0)          // the quad (or tail pair) that starts here: the entry of the row this lane will hold
                    prq = tab(t0 - 7 + p + (p + 4 <= NROWS ? (lane >> 4) : oddRow));
                const uint32_t* lp = reinterpret_cast<const uint32_t*>(win + bb + p * pitch);
                uint32_t d[6], e[5];
#pragma unroll
                for (int k = 0; k < 6; k++) d[k] = lp[k];
#pragma unroll
                for (int k = 0; k < 5; k++) e[k] = __builtin_amdgcn_alignbit(d[k + 1], d[k], 16);
#pragma unroll
                for (int j = 0; j < 8; j++)
                {
                    if (FIRST && j > p) continue;
                    const int slot = (p - j) & 7;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                    {
                        acc[slot][0] = __builtin_amdgcn_sad_u16(d[k], F[j][k], acc[slot][0]);
                        acc[slot][1] = __builtin_amdgcn_sad_u16(e[k], F[j][k], acc[slot][1]);
                        acc[slot][2] = __builtin_amdgcn_sad_u16(d[k + 1], F[j][k], acc[slot][2]);
                        acc[slot][3] = __builtin_amdgcn_sad_u16(e[k + 1], F[j][k], acc[slot][3]);
                    }
                }
                if (!FIRST || p == 7)
                {
                    const int slot = (p + 1) & 7;
                    int s8[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) { s8[k] = (int)acc[slot][k]; acc[slot][k] = 0; }
                    // 16x16 level, transposed while it is summed: lane (quad q, k = lane & 3) ends with column k of 16x16 PU q.  A butterfly over the quad - each
                    // lane keeps the column whose parity is its own and hands the other one over (quad_perm xor 1), then the same for the pair of pairs (xor 2):
                    // 6 selects + 3 DPP adds where four full quad sums and a four-way select took 8 + 3 (which the compiler turned into divergent branches)
                    const int keep0 = lane1 ? s8[1] : s8[0], give0 = lane1 ? s8[0] : s8[1];
                    const int keep1 = lane1 ? s8[3] : s8[2], give1 = lane1 ? s8[2] : s8[3];
                    const int t0s = keep0 + dpp<0xB1>(give0), t1s = keep1 + dpp<0xB1>(give1);          // columns {0|1} / {2|3} over the lane pair
                    const int v16i = (lane2 ? t1s : t0s) + dpp<0x4E>(lane2 ? t0s : t1s);
                    const uint32_t v16 = (uint32_t)v16i;
                    const uint32_t v32 = (uint32_t)row_sum_of_quads(v16i);
                    const uint32_t sb = (uint32_t)__builtin_amdgcn_readlane((int)cb, p);      // costY[m] << 8 | m, in an SGPR
#pragma unroll
                    for (int k = 0; k < 4; k++)
                    {
                        const uint32_t kk = ((uint32_t)s8[k] << 8) + sb;                               // (sad + costY) << 8 | m (an 8x8 SAD of 10-bit samples is < 2^16)
                        r8c[k] = kk < r8c[k] ? kk : r8c[k];
                    }
                    const uint32_t k16 = (v16 << 8) + sb, k32 = (v32 << 8) + sb;             // costX joins once per group
                    r16 = k16 < r16 ? k16 : r16;
                    r32 = k32 < r32 ? k32 : r32;
                    if (FIRST)
                    {   // the first block completes a single row (m = 0)
                        v2u sw = __builtin_amdgcn_permlane16_swap(v32, v32, false, false);
                        const unsigned h64 = sw.x + sw.y;
                        sw = __builtin_amdgcn_permlane32_swap(h64, h64, false, false);
                        const uint32_t k64 = ((sw.x + sw.y) << 8) + sb;
                        r64 = k64 < r64 ? k64 : r64;
                    }
                    else if ((p & 1) == 0) v32even = v32;
                    else
                    {
                        // rows A (p - 1) and B (p): {A0,B0,A2,B2} + {A1,B1,A3,B3} = {A01, B01, A23, B23}
                        v2u sw = __builtin_amdgcn_permlane16_swap(v32even, v32, false, false);
                        const unsigned h64 = sw.x + sw.y;
                        const bool quadEnd = (p & 3) == 3, tailPair = (p & 3) == 1 && p + 2 >= NROWS;       // constants once the row loop is unrolled
                        if (quadEnd)
                        {   // with the pair before it: the halves change places -> 16-lane row r holds the total of the quad's row r
                            sw = __builtin_amdgcn_permlane32_swap(pairSum, h64, false, false);
                            const uint32_t k64 = ((sw.x + sw.y) << 8) + prq;
                            r64 = k64 < r64 ? k64 : r64;
                        }
                        else if (tailPair)
                        {
                            sw = __builtin_amdgcn_permlane32_swap(h64, h64, false, false);
                            const uint32_t k64 = ((sw.x + sw.y) << 8) + prq;
                            r64 = k64 < r64 ? k64 : r64;
                        }
                        else pairSum = h64;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        using I8 = std::integral_constant<int, 8>;
        rows8(std::true_type{}, I8{}, 0);
        int t0 = 8;
        for (; t0 + 8 <= T; t0 += 8)
            rows8(std::false_type{}, I8{}, t0);
        switch (T - t0)
        {
        case 2: rows8(std::false_type{}, std::integral_constant<int, 2>{}, t0); break;
        case 4: rows8(std::false_type{}, std::integral_constant<int, 4>{}, t0); break;
        case 6: rows8(std::false_type{}, std::integral_constant<int, 6>{}, t0); break;
        default: break;
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const uint32_t kc = ((r8c[k] >> 8) << 2) + cxk4[k];                      // (sad + costY + costX) << 2 | k
            const uint32_t key = ((kc & ~3u) << 8) | ((r8c[k] & 255u) << 2) | (kc & 3u);
            r8 = key < r8 ? key : r8;
        }
        const u64 w8 = ((u64)(r8 >> 10) << 32) | (uint32_t)(((r8 >> 2) & 255u) * NC + 4 * g + (r8 & 3u));
        bk8 = w8 < bk8 ? w8 : bk8;
        auto widen = [&](const uint32_t r) { const uint32_t rc = r + cxL8; return ((u64)(rc >> 8) << 32) | (uint32_t)((rc & 255u) * NC + 4 * g + kcol); };
        const u64 w16 = widen(r16), w32 = widen(r32), w64 = widen(r64);
        bk16 = w16 < bk16 ? w16 : bk16;
        bk32 = w32 < bk32 ? w32 : bk32;
        bk64 = w64 < bk64 ? w64 : bk64;
    }
    u64* rec = a.best + (size_t)ctu * 85;
    atomicMin(&rec[lane], bk8);
    atomicMin(&rec[64 + (lane >> 2)], bk16);
    if ((lane & 15) < 4) { atomicMin(&rec[80 + (lane >> 4)], bk32); atomicMin(&rec[84], bk64); }      // every 16-lane row saw its own window rows of the 64x64 level
}

// wavefronts per workgroup: as many as the column count keeps busy (16 = 1024 threads max); the
// columns are dealt round-robin, so the idle tail is at most one column per wavefront.
int launch_me_cand(const x265hip_me_params* p, hipStream_t s);       // me_cand_kernel.hip: 0 = launched, 1 = not applicable, < 0 = error

// the flag set the minima-only 8-bit launch uses by default (-1 = round 4's kernel): what measured fastest on one box, profiles/r05_me_flags_ab.txt
static const bool W2_DEFAULT = true;          // the 16-bit twin (me_ctu_w2_kernel): 2.60 -> 2.35 ms at 4K, 9.95 -> 8.95 ms at 8K (one box, interleaved: profiles/r05_me10_ab.txt)
static const int Q2_DEFAULT_FLAGS = 254;      // CTAB | PAIR64 | DEFERX | COLMIN | MASK | RING | QUAD64: 1.24 ms against round 4's 1.42 at 4K (three interleaved rounds)

// Every A/B switch of the search launches, read ONCE (round-5 advisor: some were read per launch - getenv in a hot path, not thread-safe against setenv in the host
// process - and x265hip_me_minima_kernel_name re-derived the selection on its own and could disagree with the launch).  x265hip_me_env_refresh() re-reads them: a
// TEST-ONLY hook for the parity tests and soaks that flip a switch between launches of one process.
struct MeEnv
{
    bool split, generic, cand;
    int bestVar;            // X265HIP_ME_BEST_VARIANT & 3, -1 = unset
    int q2Flags;            // X265HIP_ME_Q2_FLAGS, or the default / -1 that follows from bestVar
    int bestWaves;
    bool splitGroups8, splitGroups16;
    bool w2;
    bool xcdOrder;
};
static MeEnv g_meEnv;
static std::once_flag g_meEnvOnce;
static void me_env_read()
{
    MeEnv e;
    e.split = getenv("X265HIP_ME_SPLIT") != nullptr;
    e.generic = getenv("X265HIP_ME_GENERIC") != nullptr;
    const char* which = getenv("X265HIP_ME_KERNEL");
    e.cand = which && which[0] == 'c';
    const char* bv = getenv("X265HIP_ME_BEST_VARIANT");
    e.bestVar = bv ? atoi(bv) & 3 : -1;
    const char* q2 = getenv("X265HIP_ME_Q2_FLAGS");
    e.q2Flags = q2 ? atoi(q2) : (bv ? -1 : Q2_DEFAULT_FLAGS);
    e.xcdOrder = getenv("X265HIP_ME_XCD_OFF") == nullptr;
    e.bestWaves = getenv("X265HIP_ME_BEST_WAVES") ? atoi(getenv("X265HIP_ME_BEST_WAVES")) : 0;
    const char* sg = getenv("X265HIP_ME_SPLIT_GROUPS");
    e.splitGroups8 = !(sg && atoi(sg) == 0);
    e.splitGroups16 = sg && atoi(sg) == 1;
    const char* w2 = getenv("X265HIP_ME_W2");
    e.w2 = w2 ? atoi(w2) != 0 : W2_DEFAULT;
    g_meEnv = e;
}
static const MeEnv& me_env()
{
    std::call_once(g_meEnvOnce, me_env_read);
    return g_meEnv;
}
// LDS row pitch (bytes) of a launch: the power of two that holds the widest payload of the kernels of that sample size + the largest skew (17 dwords)
static int me_pitch_bytes(int bpp, int range)
{
    const int nld = bpp == 1 ? MECfg<uint8_t>::NLD : MECfg<uint16_t>::NLD;
    const int payload = (3 + (56 + 2 * range) * bpp + 4 * nld + 3) >> 2;
    const int payloadW = bpp == 2 ? (3 + (56 + 2 * range) * 2 + 4 * 6 + 3) >> 2 : payload;
    int pitchDw = 64;
    while (pitchDw < (payloadW > payload ? payloadW : payload) + 17) pitchDw <<= 1;
    return pitchDw * 4;
}

static int pick_waves(int ncols)
{
    return ncols >= 16 ? 16 : (ncols < 4 ? 4 : ncols);
}

template <typename Px>
static int launch_me(const x265hip_me_params* p, hipStream_t s)
{
    const MeEnv& env = me_env();
    const bool p_split = env.split;         // surfaces-then-minima pair of launches instead of the fused one (A/B runs)
    const bool p_generic = env.generic;     // force the generic (v_sad) kernel (A/B runs)
    constexpr int BPP = PxInfo<Px>::BPP;
    MEArgs a;
    a.fenc = (const uint8_t*)p->fenc;  a.fencStrideB = (long)p->fenc_stride * BPP;
    a.fref = (const uint8_t*)p->fref;  a.frefStrideB = (long)p->fref_stride * BPP;
    a.ctusW = p->width / 64;
    a.range = p->range;
    const int payload = (3 + (56 + 2 * p->range) * BPP + 4 * MECfg<Px>::NLD + 3) >> 2;   // dwords actually read per row
    a.payloadDw = payload;
    // the 10-bit column-group kernel reads 6 dwords per window row, one more than the generic kernel: the pitch must hold ITS payload (at
    // +-13 and +-77 the generic payload + skew is exactly a power of two and the fast path refused with an "internal" error: found by
    // tools/r3_soak.py, round 3)
    const int payloadW = sizeof(Px) == 2 ? (3 + (56 + 2 * p->range) * 2 + 4 * 6 + 3) >> 2 : payload;
    (void)payloadW;
    a.rowBytes = me_pitch_bytes(BPP, p->range);                // power-of-two pitch >= payload + max skew (17 dwords): the same function names the kernel
    a.surf = p->surf; a.best = (unsigned long long*)p->best;
    const bool anySurf = p->surf != nullptr, anyBest = p->best != nullptr;
    const bool packed = anySurf && p->surf_format == X265HIP_SURF_PACKED;
    a.costX = p->cost_x; a.costY = p->cost_y;
    a.centres = p->centres;
    a.xcdOrder = env.xcdOrder ? 1 : 0;
    const int nctu = a.ctusW * (p->height / 64);
    const size_t lds = (size_t)a.rowBytes * (64 + 2 * p->range + 2);     // + 2 rows the pipeline may prefetch past the window
    if (lds > 160 * 1024) { set_error("me_fullsearch: range %d needs %zu B of LDS (> 160 KiB)", p->range, lds); return X265HIP_EINVAL; }
    const int nw = pick_waves(2 * p->range + 1);
    dim3 grid(nctu), block(nw * 64);
#define LAUNCH_P(SF, BS, PT) do { \
        if (lds > 64 * 1024) X265HIP_TRY(hipFuncSetAttribute((const void*)me_ctu_kernel<Px, SF, BS, PT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((me_ctu_kernel<Px, SF, BS, PT>), grid, block, lds, s, a); } while (0)
#define LAUNCH(SF, BS) do { \
        if (a.rowBytes == 256) LAUNCH_P(SF, BS, 256); else if (a.rowBytes == 512) LAUNCH_P(SF, BS, 512); \
        else if (a.rowBytes == 1024) LAUNCH_P(SF, BS, 1024); \
        else { set_error("me_fullsearch: range %d needs an LDS row pitch of %d bytes (unsupported)", p->range, a.rowBytes); return X265HIP_EINVAL; } } while (0)
    // 8-bit: the record-per-lane kernel (me_cand_kernel.hip) owns the chunk-major surface format X265HIP_SURF_PACKED_T; for the
    // record-contiguous formats and for minima alone the row-walking kernels below are faster (1.76 / 1.34 ms against 3.2 / 1.42 ms at
    // 4K, profiles/r02_me_cand_ab.txt).  X265HIP_ME_KERNEL=cand forces it for every 8-bit launch (parity tests, A/B).
    {
        const bool wantT = p->surf && (p->surf_format == X265HIP_SURF_PACKED_T || p->surf_format == X265HIP_SURF_PACKED_B);
        if (sizeof(Px) == 1 && !p_generic && (wantT || env.cand))
        {
            const int rc = launch_me_cand(p, s);
            if (rc <= 0) return rc;
        }
        if (wantT)
        { set_error("me_fullsearch: X265HIP_SURF_PACKED_T / _PACKED_B are written by the record-per-lane kernel only (depth 8, window within its LDS / step limits)"); return X265HIP_EINVAL; }
    }
    if (sizeof(Px) == 1 && a.rowBytes == 256 && !p_generic)
    {
        // 8-bit fast path (v_qsad_pk_u16_u8); 2 * range + 75 bytes of window row must fit the 256-byte pitch
        const int bestVar = env.bestVar;           // A/B and the parity test of every variant
        const int q2Flags = env.q2Flags;           // round 5's flagged kernel (me_ctu_q2_kernel<256, FL>): A/B and parity tests; unset = the default
        const int bestWaves = env.bestWaves;       // A/B: wavefronts per workgroup of the minima-only launch
        // 4K and up: 12 wavefronts per workgroup instead of 16 - step 1.875 -> 1.823 ms at 4K on one box, three interleaved rounds (8 does the same, 10 and 6 lose;
        // at 1080p 16 stays best): profiles/r04_me_minima_ab.txt
#define LAUNCH_QV(V) do { int nwq = pick_waves((2 * p->range + 4) / 4); if (nwq > 16) nwq = 16; if (nctu >= 1024 && nwq > 12) nwq = 12; \
        if (bestWaves >= 4 && bestWaves <= 16 && bestWaves < pick_waves((2 * p->range + 4) / 4) + 1) nwq = bestWaves; \
        hipLaunchKernelGGL((me_ctu_q_kernel<false, true, 256, false, V>), grid, dim3(nwq * 64), lds, s, a); } while (0)
#define LAUNCH_Q(SF, BS, MAXW) do { int nwq = pick_waves((2 * p->range + 4) / 4); if (nwq > (MAXW)) nwq = (MAXW); \
        if (packed) hipLaunchKernelGGL((me_ctu_q_kernel<SF, BS, 256, SF>), grid, dim3(nwq * 64), lds, s, a); \
        else hipLaunchKernelGGL((me_ctu_q_kernel<SF, BS, 256, false>), grid, dim3(nwq * 64), lds, s, a); } while (0)
        // Both outputs: ONE fused launch.  It needs ~170 VGPRs, so its workgroup is 8 wavefronts (2 per SIMD,
        // 256-VGPR budget) instead of 16; the qsad chains carry enough ILP to keep the VALU busy at that occupancy.
        // X265HIP_ME_SPLIT forces the older surfaces-then-minima pair of launches (A/B measurements).
        if (anySurf && anyBest && !p_split) LAUNCH_Q(true, true, 12);
        else
        {
            if (anySurf) LAUNCH_Q(true, false, 16);
            if (anyBest)
            {
                if (bestVar == 1) LAUNCH_QV(1); else if (bestVar == 2) LAUNCH_QV(2); else if (bestVar == 3) LAUNCH_QV(3);
                else if (q2Flags < 0) LAUNCH_QV(0);          // round 4's kernel
                else
                {   // round 5's flagged kernel: the same launch geometry + the row-constant table behind the window; from 4K up 8 wavefronts per workgroup (two
                    // workgroups per CU) are 0.8 % ahead of 12: 1.217 - 1.219 against 1.228 - 1.239 ms, one box, two interleaved rounds (profiles/r05_me_flags_ab.txt)
                    int nwq = pick_waves((2 * p->range + 4) / 4); if (nwq > 16) nwq = 16; if (nctu >= 1024 && nwq > 8) nwq = 8;
                    if (bestWaves >= 4 && bestWaves <= 16 && bestWaves < pick_waves((2 * p->range + 4) / 4) + 1) nwq = bestWaves;
                    // few CTUs (a band of the ring: 2 CTU rows of a 4K picture = 120; a small picture): one workgroup per CTU would leave most of the 256 CUs idle -
                    // 8 column groups per workgroup of 8 wavefronts instead, ceil(groups / 8) workgroups per CTU (X265HIP_ME_SPLIT_GROUPS=0: off, A/B).  Banded 4K 8-bit steps
                    // (2 / 3 / 4 CTU rows per band): 3.11 / 2.72 / 2.43 -> 2.95 / 2.49 / 2.38 ms, one box, interleaved (profiles/r05_band_tables.txt)
                    const bool splitGroups = env.splitGroups8;
                    const int ngroups = (2 * p->range + 4) / 4;
                    int gPer = 0;
                    dim3 grid2 = grid;
                    if (splitGroups && nctu < 192 && ngroups > 8) { nwq = 8; gPer = 8; grid2 = dim3(nctu, (ngroups + 7) / 8); }
                    const size_t ldsA = (size_t)320 * (64 + 2 * p->range + 2);            // flag 256: two copies of the window at a 320-byte pitch
                    if ((q2Flags & 256) && (a.payloadDw + 24 > 80 || 2 * ldsA > 150 * 1024)) { set_error("me_fullsearch: X265HIP_ME_Q2_FLAGS bit 256 (two window copies) holds +-75 at most"); return X265HIP_EINVAL; }
                    const size_t ctabAt = (q2Flags & 256) ? 2 * ldsA : lds;
                    const size_t lds2 = ctabAt + (size_t)(2 * p->range + 1 + 16) * 4;
#define LAUNCH_Q2P(PT, FLV) case FLV: \
                        if (lds2 > 64 * 1024) X265HIP_TRY(hipFuncSetAttribute((const void*)me_ctu_q2_kernel<PT, FLV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2)); \
                        hipLaunchKernelGGL((me_ctu_q2_kernel<PT, FLV>), grid2, dim3(nwq * 64), lds2, s, a, (int)ctabAt, gPer); break;
#define LAUNCH_Q2(FLV) LAUNCH_Q2P(256, FLV)
                    switch (q2Flags)
                    {
                    /* the single flags, the sets the profile tables name, the default (profiles/r05_me_flags_ab.txt lists more combinations: they were instantiated for the A/B visits) */
                    LAUNCH_Q2(0) LAUNCH_Q2(1) LAUNCH_Q2(2) LAUNCH_Q2(4) LAUNCH_Q2(8) LAUNCH_Q2(32) LAUNCH_Q2(62) LAUNCH_Q2(254) LAUNCH_Q2P(320, 446) LAUNCH_Q2P(320, 256)
                    default: set_error("me_fullsearch: X265HIP_ME_Q2_FLAGS %d is not an instantiated combination", q2Flags); return X265HIP_EINVAL;
                    }
#undef LAUNCH_Q2
#undef LAUNCH_Q2P
                }
            }
        }
#undef LAUNCH_Q
#undef LAUNCH_QV
    }
    else if (packed) { set_error("me_fullsearch: X265HIP_SURF_PACKED needs depth 8 and 2 * range + 75 <= 256"); return X265HIP_EINVAL; }
    else if (sizeof(Px) == 2 && p->depth <= 10 && (a.rowBytes == 512 || a.rowBytes == 256) && !p_generic)
    {
        // 10-bit fast path (column groups, v_sad_u16); 2 * (2 * range + 76) bytes of window row must fit the 512-byte pitch
        const int plw = (3 + (56 + 2 * p->range) * 2 + 4 * 6 + 3) >> 2;                 // the group kernel reads 6 dwords per row
        a.payloadDw = plw > a.payloadDw ? plw : a.payloadDw;
        if (a.payloadDw + 17 > a.rowBytes / 4) { set_error("me_fullsearch: internal: window row does not fit the LDS pitch"); return X265HIP_EINVAL; }
        const int bestVarW = env.bestVar;      // A/B
        const int bestWavesW = env.bestWaves;
#define LAUNCH_WV(V) do { int nwq = pick_waves((2 * p->range + 4) / 4); if (nwq > 16) nwq = 16; if (nctu >= 1024 && nwq > 12) nwq = 12;      /* 4K: step 3.20 -> 3.18 ms, 8K 12.61 -> 12.53 */ \
        if (bestWavesW >= 4 && bestWavesW <= 16) nwq = bestWavesW; \
        if (a.rowBytes == 256) hipLaunchKernelGGL((me_ctu_w_kernel<false, true, 256, V>), grid, dim3(nwq * 64), lds, s, a); \
        else { \
            if (lds > 64 * 1024) X265HIP_TRY(hipFuncSetAttribute((const void*)me_ctu_w_kernel<false, true, 512, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL((me_ctu_w_kernel<false, true, 512, V>), grid, dim3(nwq * 64), lds, s, a); } } while (0)
#define LAUNCH_W(SF, BS, MAXW) do { int nwq = pick_waves((2 * p->range + 4) / 4); if (nwq > (MAXW)) nwq = (MAXW); \
        if (a.rowBytes == 256) hipLaunchKernelGGL((me_ctu_w_kernel<SF, BS, 256>), grid, dim3(nwq * 64), lds, s, a); \
        else { \
            if (lds > 64 * 1024) X265HIP_TRY(hipFuncSetAttribute((const void*)me_ctu_w_kernel<SF, BS, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL((me_ctu_w_kernel<SF, BS, 512>), grid, dim3(nwq * 64), lds, s, a); } } while (0)
        if (anySurf && anyBest && !p_split) LAUNCH_W(true, true, 12);
        else
        {
            if (anySurf) LAUNCH_W(true, false, 16);
            if (anyBest)
            {
                const bool w2 = env.w2 && bestVarW < 0 && p->range <= 120;      // round 5's kernel (me_ctu_w2_kernel): X265HIP_ME_W2=0 = round 4's (A/B)
                if (w2)
                {
                    int nwq = pick_waves((2 * p->range + 4) / 4); if (nwq > 16) nwq = 16; if (nctu >= 1024 && nwq > 12) nwq = 12;
                    if (bestWavesW >= 4 && bestWavesW <= 16) nwq = bestWavesW;
                    // off by default at 16 bits: the 92 KB window leaves room for ONE workgroup per CU, so four small workgroups per CTU run in two rounds at half the
                    // occupancy - banded 4K 10-bit steps 4.45 / 4.05 / 3.80 ms (2 / 3 / 4 CTU rows per band) became 4.50 / 4.25 / 3.80 (profiles/r05_band_tables.txt); 1 = on (A/B)
                    const bool splitGroupsW = env.splitGroups16;
                    const int ngroups = (2 * p->range + 4) / 4;
                    int gPer = 0;
                    dim3 grid2 = grid;
                    if (splitGroupsW && nctu < 192 && ngroups > 8) { nwq = 8; gPer = 8; grid2 = dim3(nctu, (ngroups + 7) / 8); }      // few CTUs: see the 8-bit launch
                    const size_t lds2 = lds + (size_t)(2 * p->range + 1 + 16) * 4;
                    if (a.rowBytes == 256) hipLaunchKernelGGL((me_ctu_w2_kernel<256>), grid2, dim3(nwq * 64), lds2, s, a, (int)lds, gPer);
                    else
                    {
                        if (lds2 > 64 * 1024) X265HIP_TRY(hipFuncSetAttribute((const void*)me_ctu_w2_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
                        hipLaunchKernelGGL((me_ctu_w2_kernel<512>), grid2, dim3(nwq * 64), lds2, s, a, (int)lds, gPer);
                    }
                }
                else if (bestVarW == 0) LAUNCH_WV(0); else if (bestVarW == 2) LAUNCH_WV(2); else if (bestVarW == 3) LAUNCH_WV(3); else LAUNCH_WV(1);
            }
        }
#undef LAUNCH_W
#undef LAUNCH_WV
    }
    else if (anySurf && anyBest) LAUNCH(true, true);
    else if (anySurf) LAUNCH(true, false);
    else LAUNCH(false, true);
#undef LAUNCH
    X265HIP_TRY(hipGetLastError());
    return 0;
}

__global__ void fill_u64_kernel(unsigned long long* p, size_t n, unsigned long long v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) p[i] = v;
}

} // namespace x265hip

using namespace x265hip;

extern "C" const char* x265hip_me_minima_kernel_name(int depth, int range)
{
    // the SAME switches and the SAME pitch function as launch_me: what this returns is what a minima-only launch runs
    static thread_local char name[64];
    const MeEnv& env = me_env();
    const int bpp = depth == 8 ? 1 : 2, pitch = me_pitch_bytes(bpp, range);
    if (depth != 8)
    {
        if (env.generic || depth > 10 || pitch > 512) return "me_ctu_kernel<u16,best>";          // 12 bits, or a window row beyond the 512-byte pitch (+-76 and up at 16 bits)
        return (env.w2 && env.bestVar < 0 && range <= 120) ? "me_ctu_w2_kernel" : "me_ctu_w_kernel<best>";
    }
    if (env.cand && !env.generic) return "me_ctu_c_kernel (record per lane)";
    if (env.generic || pitch != 256) return "me_ctu_kernel<u8,best>";                            // the window row + the largest skew must fit the 256-byte LDS pitch: +-58
    if (env.q2Flags < 0 || range > 120 || env.bestVar > 0) return "me_ctu_q_kernel<best>";
    snprintf(name, sizeof(name), "me_ctu_q2_kernel<256,%d>", env.q2Flags);
    return name;
}

/* TEST-ONLY: re-read the X265HIP_ME_* switches (they are read once per process otherwise).  Not for use while launches are in flight on other threads. */
extern "C" void x265hip_me_env_refresh(void)
{
    (void)me_env();
    me_env_read();
}

extern "C" int x265hip_me_fullsearch(const x265hip_me_params* p, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !p->fenc || !p->fref) { set_error("me_fullsearch: NULL plane"); return X265HIP_EINVAL; }
    if (p->width <= 0 || p->height <= 0 || (p->width & 63) || (p->height & 63))
    { set_error("me_fullsearch: width/height must be positive multiples of 64 (got %dx%d)", p->width, p->height); return X265HIP_EINVAL; }
    if (p->range < 1 || p->range > 256) { set_error("me_fullsearch: range %d out of [1,256]", p->range); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("me_fullsearch: depth %d", p->depth); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    if (((p->fenc_stride * bpp) & 3) || ((p->fref_stride * bpp) & 3) || ((uintptr_t)p->fenc & 3))
    { set_error("me_fullsearch: plane strides must be multiples of 4 bytes and fenc 4-byte aligned"); return X265HIP_EINVAL; }
    const bool anyBest = p->best != nullptr;
    if (!p->surf && !p->best) { set_error("me_fullsearch: no output requested"); return X265HIP_EINVAL; }
    if (p->surf_format != X265HIP_SURF_I32 && p->surf_format != X265HIP_SURF_PACKED && p->surf_format != X265HIP_SURF_PACKED_T && p->surf_format != X265HIP_SURF_PACKED_B) { set_error("me_fullsearch: surf_format %d", p->surf_format); return X265HIP_EINVAL; }
    if (anyBest && (!p->cost_x || !p->cost_y)) { set_error("me_fullsearch: best[] needs cost_x / cost_y"); return X265HIP_EINVAL; }
    if (p->depth == 8) return launch_me<uint8_t>(p, (hipStream_t)stream);
    return launch_me<uint16_t>(p, (hipStream_t)stream);
}

extern "C" int x265hip_me_best_reset(uint64_t* best, size_t count, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!best) { set_error("me_best_reset: NULL"); return X265HIP_EINVAL; }
    if (!count) return 0;
    int blocks = (int)((count + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fill_u64_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (unsigned long long*)best, count, ~0ull);
    X265HIP_TRY(hipGetLastError());
    return 0;
}
