// phase_kernels.hip - every fractional-sample phase of a reference plane in one pass (x265hip_phase_planes).
//
// The reference interpolates a PU-sized block per sub-sample candidate (MotionEstimate::subpelCompare, motion.cpp:1571-1664: luma_hpp /
// luma_vpp / luma_hvpp, chroma filter_hpp / filter_vpp / filter_hps + filter_vsp) and again for the final prediction
// (predict.cpp:261-351).  All of these are position-invariant FIR filters of the reference plane, so the sample a block
// interpolation writes for source position (x, y) and phase (xf, yf) is a function of the plane alone: this kernel evaluates it for
// every position and every phase once per picture; a consumer reads blocks of the phase planes in place.
// Arithmetic (exact, int32 sums, the reference's int16 narrowing and clipping): ipfilter.cpp:79-118 (horizontal pp), :164-203
// (vertical pp), :362-369 = :120-162 (horizontal ps with row extension) + :241-282 (vertical sp); taps constants.cpp:250-268.
//
// Bound: HBM writes (15 + 2 * 63 / 4 = 46.5 output bytes per luma source byte at 4:2:0).  A thread owns a 4x4 tile and produces EVERY
// phase of it: the source samples are loaded once and each horizontally filtered column set is shared by its vertical phases.
#include "common.h"
#include "tile_interp.h"

#include <cstdlib>

namespace x265hip {

struct PhaseArgs
{
    const uint8_t* src;
    uint8_t* dst;
    long strideB;
    int rows, tilesW, depth;
    size_t planeBytes;
};

__constant__ int8_t kPhaseChromaTaps[8][4] = {           // constants.cpp:261-268
    { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
    { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };

template <typename Px> __device__ __forceinline__ void phase_store(uint8_t* out, long strideB, const int (&d)[4][4])
{
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        uint32_t* o = reinterpret_cast<uint32_t*>(out + y * strideB);
        if (sizeof(Px) == 1)
            o[0] = (uint32_t)d[y][0] | ((uint32_t)d[y][1] << 8) | ((uint32_t)d[y][2] << 16) | ((uint32_t)d[y][3] << 24);
        else
        {
            o[0] = (uint32_t)d[y][0] | ((uint32_t)d[y][1] << 16);
            o[1] = (uint32_t)d[y][2] | ((uint32_t)d[y][3] << 16);
        }
    }
}

// grid: x = blocks of 256 tiles along a tile row, y = tile row.  One 4x4 tile per lane, ALL 15 phases of it: the three horizontally
// filtered 11-row columns (xf = 1, 2, 3) are formed once and shared by their four vertical phases (a phase-per-thread kernel filtered
// 111 rows per tile horizontally, this one 33).
template <typename Px>
__global__ void __launch_bounds__(256) phase_luma_kernel(PhaseArgs a)
{
    constexpr int BPP = sizeof(Px);
    int bx, by;
    xcd_swizzle_2d(bx, by);                                                       // a tile row reads 11 sample rows for its 4: vertical neighbours on one XCD's L2
    const int tx = bx * 256 + threadIdx.x, ty = by + 1;                           // the first 4 and the last 8 rows are not produced
    if (tx >= a.tilesW) return;
    const long off = (long)(ty * 4) * a.strideB + (long)tx * 4 * BPP;
    const uint8_t* org = a.src + off;
    const int maxVal = (1 << a.depth) - 1, headRoom = 14 - a.depth;
    const int shiftPS = 6 - headRoom, offPS = -(8192 << shiftPS);
    const int shiftSP = 6 + headRoom, offSP = (1 << (shiftSP - 1)) + (8192 << 6);
    auto vtaps = [](const int yf, uint32_t (&cv)[4])
    {
        cv[0] = tile_sel3(yf, 0x0004ffffu, 0x0004ffffu, 0x00010000u); cv[1] = tile_sel3(yf, 0x003afff6u, 0x0028fff5u, 0x0011fffbu);
        cv[2] = tile_sel3(yf, 0xfffb0011u, 0xfff50028u, 0xfff6003au); cv[3] = tile_sel3(yf, 0x00000001u, 0xffff0004u, 0xffff0004u);
    };
#pragma unroll
    for (int xf = 0; xf < 4; xf++)
    {
        // column data of the tile's 4 columns, rows -3 .. +7: raw samples (xf = 0) or the horizontal sums (no rounding yet)
        int col[11][4];
#pragma unroll
        for (int t = 0; t < 11; t++)
        {
            const uint8_t* rp = org + (long)(t - 3) * a.strideB;
            if (xf == 0)
            {
                if (BPP == 1) { const uint32_t w = ld_u32(rp); col[t][0] = w & 0xff; col[t][1] = (w >> 8) & 0xff; col[t][2] = (w >> 16) & 0xff; col[t][3] = w >> 24; }
                else { const uint32_t w0 = ld_u32(rp), w1 = ld_u32(rp + 4); col[t][0] = w0 & 0xffff; col[t][1] = w0 >> 16; col[t][2] = w1 & 0xffff; col[t][3] = w1 >> 16; }
            }
            else
                tile_hrow<BPP>(rp - 3 * BPP, xf, col[t]);
        }
        int d[4][4];
        if (xf)
        {   // yf = 0: luma_hpp
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++) d[y][x] = tile_clip16((col[y + 3][x] + 32) >> 6, maxVal);
            phase_store<Px>(a.dst + (size_t)(xf - 1) * a.planeBytes + off, a.strideB, d);
        }
        // (row r, row r + 1) pairs of what the vertical filter reads: samples (luma_vpp) or the 16-bit intermediates of luma_hps
        uint32_t pairs[10][4];
#pragma unroll
        for (int t = 0; t < 10; t++)
#pragma unroll
            for (int x = 0; x < 4; x++)
            {
                const uint32_t lo = xf ? (uint32_t)((col[t][x] + offPS) >> shiftPS) : (uint32_t)col[t][x];
                const uint32_t hi = xf ? (uint32_t)((col[t + 1][x] + offPS) >> shiftPS) : (uint32_t)col[t + 1][x];
                pairs[t][x] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
            }
#pragma unroll
        for (int yf = 1; yf < 4; yf++)
        {
            uint32_t cv[4];
            vtaps(yf, cv);
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++)
                {
                    int sum = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) sum = tile_dot2(pairs[y + 2 * j][x], cv[j], sum);
                    d[y][x] = xf ? tile_clip16((sum + offSP) >> shiftSP, maxVal) : tile_clip16((sum + 32) >> 6, maxVal);
                }
            phase_store<Px>(a.dst + (size_t)(yf * 4 + xf - 1) * a.planeBytes + off, a.strideB, d);
        }
    }
}

// ---- round 3: the same arithmetic, reorganised for the memory system (the round-2 kernel above ran at 0.20 of the HBM peak although it is
// a pure streaming writer: its loads were re-issued for every xf because the stores in between might alias them, every re-load waited
// behind the stores queued before it - vmcnt retires in order -, and a lane stored 4 bytes at a time).  Here
//   * a tile's 11 source rows are loaded ONCE, before any store (11 x 12 bytes / 11 x 24 bytes in registers), so after the first
//     wait the kernel only computes and stores;
//   * the 4 lanes of a DPP quad transpose their tiles' rows (two butterfly steps), so a lane stores 16 contiguous bytes of ONE row -
//     one global_store_dwordx4 per phase and lane instead of four dword stores (16-bit samples: lane pairs, two dwordx4 per phase).
// Needs the row pitch to be a multiple of 16 bytes (whole quads); other pitches take the kernel above.
__device__ __forceinline__ uint32_t quad_xor1(uint32_t v) { return (uint32_t)dpp<0xB1>((int)v); }      // quad_perm [1,0,3,2]
__device__ __forceinline__ uint32_t quad_xor2(uint32_t v) { return (uint32_t)dpp<0x4E>((int)v); }      // quad_perm [2,3,0,1]

// p[r] = row r of this lane's tile (one dword: 4 samples of 8 bits) -> p[k] = row (lane & 3) of tile k of the quad
__device__ __forceinline__ void quad_transpose4(uint32_t (&p)[4], int l)
{
    const bool o1 = l & 1, o2 = l & 2;
    {   // exchange across lane ^ 1: even lanes keep rows 0 / 2 and receive the partner's, odd lanes keep rows 1 / 3
        const uint32_t a = quad_xor1(o1 ? p[0] : p[1]), b = quad_xor1(o1 ? p[2] : p[3]);
        if (o1) { p[0] = a; p[2] = b; } else { p[1] = a; p[3] = b; }
    }
    {   // exchange across lane ^ 2: lanes 0 / 1 keep elements 0 / 1, lanes 2 / 3 keep elements 2 / 3
        const uint32_t a = quad_xor2(o2 ? p[0] : p[2]), b = quad_xor2(o2 ? p[1] : p[3]);
        if (o2) { p[0] = a; p[1] = b; } else { p[2] = a; p[3] = b; }
    }
}

// after step 1 lane l holds (own row r0, partner row r0) for r0 = l & 1 in slots (0,1) and r0 + 2 in slots (2,3); after step 2 slot k = row
// (l & 3) of tile k.  Check: lane 0 starts with rows (A0 A1 A2 A3) of tile A; step 1 gives (A0 B0 A2 B2); step 2 with lane 2 (C0 D0 C2 D2)
// gives (A0 B0 C0 D0).
template <typename Px> __device__ __forceinline__ void phase_store_wide(uint8_t* plane, long strideB, int ty, int tx, int lane, const int (&d)[4][4])
{
    if (sizeof(Px) == 1)
    {
        uint32_t p[4];
#pragma unroll
        for (int y = 0; y < 4; y++) p[y] = (uint32_t)d[y][0] | ((uint32_t)d[y][1] << 8) | ((uint32_t)d[y][2] << 16) | ((uint32_t)d[y][3] << 24);
        quad_transpose4(p, lane);
        uint4* o = reinterpret_cast<uint4*>(plane + (long)(ty * 4 + (lane & 3)) * strideB + (long)(tx & ~3) * 4);
        *o = make_uint4(p[0], p[1], p[2], p[3]);
    }
    else
    {
        // 16-bit samples: a row of a tile is 8 bytes; lane pairs exchange so that the even lane stores rows 0 and 2 of both tiles, the odd lane rows 1 and 3
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int y = 0; y < 4; y++) { lo[y] = (uint32_t)d[y][0] | ((uint32_t)d[y][1] << 16); hi[y] = (uint32_t)d[y][2] | ((uint32_t)d[y][3] << 16); }
        const bool odd = lane & 1;
#pragma unroll
        for (int r = 0; r < 4; r += 2)
        {
            const uint32_t sl = quad_xor1(odd ? lo[r] : lo[r + 1]), sh = quad_xor1(odd ? hi[r] : hi[r + 1]);
            // even lane: (own row r | partner's row r); odd lane: (partner's row r + 1 | own row r + 1)
            const uint4 v = odd ? make_uint4(sl, sh, lo[r + 1], hi[r + 1]) : make_uint4(lo[r], hi[r], sl, sh);
            uint4* o = reinterpret_cast<uint4*>(plane + (long)(ty * 4 + r + (odd ? 1 : 0)) * strideB + (long)(tx & ~1) * 8);
            *o = v;
        }
    }
}

// 4 horizontal 8-tap sums (no rounding) from the row's preloaded dwords: w = the bytes from sample (x0 - 3) on (3 dwords for 8-bit, 6 for 16-bit samples)
template <int BPP> __device__ __forceinline__ void tile_hrow_regs(const uint32_t* w, int xf, int (&out)[4])
{
    if (BPP == 1)
    {
        const uint32_t c03 = tile_sel3(xf, 0x3af604ffu, 0x28f504ffu, 0x11fb0100u), c47 = tile_sel3(xf, 0x0001fb11u, 0xff04f528u, 0xff04f63au);
        const uint32_t w0 = w[0] ^ 0x80808080u, w1 = w[1] ^ 0x80808080u, w2 = w[2] ^ 0x80808080u;
#pragma unroll
        for (int x = 0; x < 4; x++)
        {
            const uint32_t lo = x ? __builtin_amdgcn_alignbyte(w1, w0, x) : w0, hi = x ? __builtin_amdgcn_alignbyte(w2, w1, x) : w1;
            out[x] = __builtin_amdgcn_sdot4((int)hi, (int)c47, __builtin_amdgcn_sdot4((int)lo, (int)c03, 8192, false), false);
        }
    }
    else
    {
        const uint32_t cp[4] = { tile_sel3(xf, 0x0004ffffu, 0x0004ffffu, 0x00010000u), tile_sel3(xf, 0x003afff6u, 0x0028fff5u, 0x0011fffbu),
                                 tile_sel3(xf, 0xfffb0011u, 0xfff50028u, 0xfff6003au), tile_sel3(xf, 0x00000001u, 0xffff0004u, 0xffff0004u) };
#pragma unroll
        for (int x = 0; x < 4; x++)
        {
            int sacc = 0;
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                const int k = (x >> 1) + j;
                sacc = tile_dot2((x & 1) ? __builtin_amdgcn_alignbyte(w[k + 1], w[k], 2) : w[k], cp[j], sacc);
            }
            out[x] = sacc;
        }
    }
}

// one xf column set of a tile: its yf = 0 plane (xf > 0: luma_hpp) and its three vertical phases (xf = 0: luma_vpp on the samples, xf > 0:
// luma_hps + luma_vsp = luma_hvpp).  Rows are turned into (row r, row r + 1) pairs as they are formed, so only the 40 pair dwords stay live.
template <typename Px, bool XF0>
__device__ __forceinline__ void phase_column_set(const PhaseArgs& a, const uint32_t (&raw)[11][sizeof(Px) == 1 ? 3 : 6], int xf, int ty, int tx, int lane, bool live)
{
    constexpr int BPP = sizeof(Px);
    const int maxVal = (1 << a.depth) - 1, headRoom = 14 - a.depth;
    const int shiftPS = 6 - headRoom, offPS = -(8192 << shiftPS);
    const int shiftSP = 6 + headRoom, offSP = (1 << (shiftSP - 1)) + (8192 << 6);
    int d[4][4];
    uint32_t pairs[10][4], prev[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int t = 0; t < 11; t++)
    {
        int col[4];
        if (XF0)
        {
            if (BPP == 1)
            {   // the tile's own samples are bytes 3 .. 6 of the 12 loaded
                const uint32_t w = __builtin_amdgcn_alignbyte(raw[t][1], raw[t][0], 3);
                col[0] = w & 0xff; col[1] = (w >> 8) & 0xff; col[2] = (w >> 16) & 0xff; col[3] = w >> 24;
            }
            else
            {   // samples 3 .. 6 of the 12 loaded: upper half of dword 1, dword 2, lower half of dword 3
                col[0] = raw[t][1] >> 16; col[1] = raw[t][2] & 0xffff; col[2] = raw[t][2] >> 16; col[3] = raw[t][3] & 0xffff;
            }
        }
        else
        {
            tile_hrow_regs<BPP>(raw[t], xf, col);
            if (t >= 3 && t < 7)
#pragma unroll
                for (int x = 0; x < 4; x++) d[t - 3][x] = tile_clip16((col[x] + 32) >> 6, maxVal);          // yf = 0: luma_hpp
        }
#pragma unroll
        for (int x = 0; x < 4; x++)
        {
            const uint32_t cur = XF0 ? (uint32_t)col[x] : (uint32_t)((col[x] + offPS) >> shiftPS);
            if (t) pairs[t - 1][x] = __builtin_amdgcn_perm(cur, prev[x], 0x05040100u);
            prev[x] = cur;
        }
    }
    if (!XF0 && live) phase_store_wide<Px>(a.dst + (size_t)(xf - 1) * a.planeBytes, a.strideB, ty, tx, lane, d);
#pragma unroll 1
    for (int yf = 1; yf < 4; yf++)
    {
        const uint32_t cv[4] = { tile_sel3(yf, 0x0004ffffu, 0x0004ffffu, 0x00010000u), tile_sel3(yf, 0x003afff6u, 0x0028fff5u, 0x0011fffbu),
                                 tile_sel3(yf, 0xfffb0011u, 0xfff50028u, 0xfff6003au), tile_sel3(yf, 0x00000001u, 0xffff0004u, 0xffff0004u) };
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++)
            {
                int sum = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) sum = tile_dot2(pairs[y + 2 * j][x], cv[j], sum);
                d[y][x] = XF0 ? tile_clip16((sum + 32) >> 6, maxVal) : tile_clip16((sum + offSP) >> shiftSP, maxVal);
            }
        if (live) phase_store_wide<Px>(a.dst + (size_t)(yf * 4 + xf - 1) * a.planeBytes, a.strideB, ty, tx, lane, d);
    }
}

template <typename Px>
__device__ __forceinline__ void phase_luma_wide_body(const PhaseArgs& a)
{
    constexpr int BPP = sizeof(Px), NW = BPP == 1 ? 3 : 6;
    const int lane = threadIdx.x & 63;
    int bx, by;
    xcd_swizzle_2d(bx, by);                                                       // a tile row reads 11 sample rows for its 4: vertical neighbours on one XCD's L2
    const int tx = bx * 256 + threadIdx.x, ty = by + 1;                           // the first 4 and the last 8 rows are not produced
    // whole quads only (tilesW % 4 == 0); lanes beyond the row repeat its last quad's loads and store nothing
    const bool live = tx < a.tilesW;
    const int txl = live ? tx : a.tilesW - 4 + (tx & 3);
    const uint8_t* org = a.src + (long)(ty * 4) * a.strideB + (long)txl * 4 * BPP;
    uint32_t raw[11][NW];
#pragma unroll
    for (int t = 0; t < 11; t++)
#pragma unroll
        for (int k = 0; k < NW; k++) raw[t][k] = ld_u32(org + (long)(t - 3) * a.strideB - 3 * BPP + 4 * k);
    phase_column_set<Px, true>(a, raw, 0, ty, tx, lane, live);
#pragma unroll 1
    for (int xf = 1; xf < 4; xf++)
        phase_column_set<Px, false>(a, raw, xf, ty, tx, lane, live);
}
// Measured (profiles/r03_phase_kernel_ab.txt, 4K luma, 15 planes): this kernel 0.075 ms at 8 bits / 0.087 ms at 10 bits against 0.071 / 0.096 ms
// for the kernel above; capped at 128 VGPRs (4 wavefronts per SIMD, a little scratch) 0.079 / 0.138 ms.  So the stage was never waiting for
// its loads or stores: ~3700 VALU instructions per tile (11 rows x 3 horizontal sets, 12 vertical phases of 16 samples, rounding, clipping,
// packing) are 60 us of issue time on the 1024 SIMDs - both kernels sit within 20 % of that.  16-bit samples take this kernel (fewer,
// wider stores matter more when a tile is twice the bytes), 8-bit samples the one above; X265HIP_PHASE_KERNEL=1 / 2 force one or the other.
template <typename Px> __global__ void __launch_bounds__(256) phase_luma_wide_kernel(PhaseArgs a) { phase_luma_wide_body<Px>(a); }

// 4-tap chroma set, eighth-sample phases: phase = yf * 8 + xf.  Same structure: one 4x4 tile per lane, the 7 x 7 source samples loaded
// once, the horizontally filtered columns of an xf shared by its eight vertical phases.
template <typename Px>
__global__ void __launch_bounds__(256) phase_chroma_kernel(PhaseArgs a)
{
    constexpr int BPP = sizeof(Px);
    int bx, by;
    xcd_swizzle_2d(bx, by);
    const int tx = bx * 256 + threadIdx.x, ty = by + 1;
    if (tx >= a.tilesW) return;
    const long off = (long)(ty * 4) * a.strideB + (long)tx * 4 * BPP;
    const int maxVal = (1 << a.depth) - 1, headRoom = 14 - a.depth;
    const int shiftPS = 6 - headRoom, offPS = -(8192 << shiftPS);
    const int shiftSP = 6 + headRoom, offSP = (1 << (shiftSP - 1)) + (8192 << 6);
    // rows -1 .. +5, columns -1 .. +5 of the tile
    int s[7][7];
#pragma unroll
    for (int r = 0; r < 7; r++)
    {
        const uint8_t* rp = a.src + off + (long)(r - 1) * a.strideB - BPP;
        if (BPP == 1)
        {
            const uint32_t w0 = ld_u32(rp), w1 = ld_u32(rp + 4);
            s[r][0] = w0 & 0xff; s[r][1] = (w0 >> 8) & 0xff; s[r][2] = (w0 >> 16) & 0xff; s[r][3] = w0 >> 24;
            s[r][4] = w1 & 0xff; s[r][5] = (w1 >> 8) & 0xff; s[r][6] = (w1 >> 16) & 0xff;
        }
        else
        {
            const uint32_t w0 = ld_u32(rp), w1 = ld_u32(rp + 4), w2 = ld_u32(rp + 8), w3 = ld_u32(rp + 12);
            s[r][0] = w0 & 0xffff; s[r][1] = w0 >> 16; s[r][2] = w1 & 0xffff; s[r][3] = w1 >> 16;
            s[r][4] = w2 & 0xffff; s[r][5] = w2 >> 16; s[r][6] = w3 & 0xffff;
        }
    }
#pragma unroll 1
    for (int xf = 0; xf < 8; xf++)
    {
        const int cx0 = kPhaseChromaTaps[xf][0], cx1 = kPhaseChromaTaps[xf][1], cx2 = kPhaseChromaTaps[xf][2], cx3 = kPhaseChromaTaps[xf][3];
        int col[7][4];                       // xf = 0: the samples of columns 0..3; else the horizontal sums (no rounding yet)
#pragma unroll
        for (int r = 0; r < 7; r++)
#pragma unroll
            for (int x = 0; x < 4; x++)
                col[r][x] = xf ? cx0 * s[r][x] + cx1 * s[r][x + 1] + cx2 * s[r][x + 2] + cx3 * s[r][x + 3] : s[r][x + 1];
        int d[4][4];
        if (xf)
        {   // yf = 0: filter_hpp
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++) d[y][x] = tile_clip16((col[y + 1][x] + 32) >> 6, maxVal);
            phase_store<Px>(a.dst + (size_t)(xf - 1) * a.planeBytes + off, a.strideB, d);
#pragma unroll
            for (int r = 0; r < 7; r++)
#pragma unroll
                for (int x = 0; x < 4; x++) col[r][x] = (int16_t)((col[r][x] + offPS) >> shiftPS);          // filter_hps output (int16)
        }
#pragma unroll 1
        for (int yf = 1; yf < 8; yf++)
        {
            const int cy0 = kPhaseChromaTaps[yf][0], cy1 = kPhaseChromaTaps[yf][1], cy2 = kPhaseChromaTaps[yf][2], cy3 = kPhaseChromaTaps[yf][3];
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++)
                {
                    const int sum = cy0 * col[y][x] + cy1 * col[y + 1][x] + cy2 * col[y + 2][x] + cy3 * col[y + 3][x];
                    d[y][x] = xf ? tile_clip16((sum + offSP) >> shiftSP, maxVal) : tile_clip16((sum + 32) >> 6, maxVal);
                }
            phase_store<Px>(a.dst + (size_t)(yf * 8 + xf - 1) * a.planeBytes + off, a.strideB, d);
        }
    }
}

} // namespace x265hip

using namespace x265hip;

// rows of the source (multiple of 4, >= 16) -> lines [4, rows - 8) of every phase plane; plane_bytes = distance between consecutive phase
// planes of dst (the whole plane for a band of a larger picture: csrc/phase_stream.hip runs bands with src / dst moved to the band's
// first line - 4 and the planes' own pitch)
int x265hip::phase_planes_launch(int depth, int chroma, const void* src, void* dst, intptr_t stride, int rows, size_t plane_bytes, hipStream_t s)
{
    const int bpp = depth == 8 ? 1 : 2;
    PhaseArgs a;
    a.src = (const uint8_t*)src; a.dst = (uint8_t*)dst; a.strideB = (long)stride * bpp; a.rows = rows;
    a.tilesW = (int)(stride / 4); a.depth = depth; a.planeBytes = plane_bytes;
    // tile rows 1 .. rows / 4 - 3: a tile reads 3 rows above and 7 below itself (and a few bytes of the neighbouring rows at the row
    // ends), so every access stays inside the plane without any guard memory around it
    const dim3 gridL((a.tilesW + 255) / 256, rows / 4 - 3);               // a thread produces every phase of its tile
    if (chroma)
    {
        if (bpp == 1) hipLaunchKernelGGL(phase_chroma_kernel<uint8_t>, gridL, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(phase_chroma_kernel<uint16_t>, gridL, dim3(256), 0, s, a);
    }
    else
    {
        static const int force = getenv("X265HIP_PHASE_KERNEL") ? atoi(getenv("X265HIP_PHASE_KERNEL")) : 0;
        // the wide-store kernel needs whole quads per row (pitch a multiple of 16 bytes / 4 tiles)
        const bool wideOk = !(a.tilesW & 3) && !(((uintptr_t)dst | plane_bytes) & 15);
        const bool wide = wideOk && (force == 2 || (force != 1 && bpp == 2));
        if (wide)
        {
            if (bpp == 1) hipLaunchKernelGGL(phase_luma_wide_kernel<uint8_t>, gridL, dim3(256), 0, s, a);
            else hipLaunchKernelGGL(phase_luma_wide_kernel<uint16_t>, gridL, dim3(256), 0, s, a);
        }
        else if (bpp == 1) hipLaunchKernelGGL(phase_luma_kernel<uint8_t>, gridL, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(phase_luma_kernel<uint16_t>, gridL, dim3(256), 0, s, a);
    }
    return check_hip(hipGetLastError(), "phase_planes launch");
}

extern "C" int x265hip_phase_planes(const x265hip_phase_planes_params* p, void* stream)
{
    if (!p || !p->src || !p->dst) { set_error("phase_planes: NULL argument"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("phase_planes: depth %d", p->depth); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    if (p->stride <= 0 || p->rows <= 0 || (p->rows & 3) || ((p->stride * bpp) & 3) || (p->stride & 3))
    { set_error("phase_planes: stride %ld / rows %d must be positive multiples of 4", (long)p->stride, p->rows); return X265HIP_EINVAL; }
    if ((uintptr_t)p->dst & 3) { set_error("phase_planes: dst must be 4-byte aligned"); return X265HIP_EINVAL; }
    if (p->rows / 4 > 65535 || p->rows < 16) { set_error("phase_planes: %d rows", p->rows); return X265HIP_EINVAL; }
    int rc = ensure_device();
    if (rc) return rc;
    return phase_planes_launch(p->depth, p->chroma, p->src, p->dst, p->stride, p->rows, (size_t)p->stride * p->rows * bpp, (hipStream_t)stream);
}
