// tile_interp.h - packed HEVC luma interpolation of one 4x4 tile (shared device code).
//
// Reference arithmetic: source/common/ipfilter.cpp - interp_horiz_pp_c :79-118 (h only), interp_vert_pp_c :164-203
// (v only), interp_hv_pp_c :362-369 = interp_horiz_ps_c :120-162 with row extension + interp_vert_sp_c :241-282, taps
// constants.cpp:250-259; every sum is formed exactly (int32) so the reference's rounding / int16 casts apply unchanged.
// Packed forms: horizontal taps by v_dot4_i32_i8 on (pixel - 128) bytes (the bias 128 * sum(taps) = 8192 is the start
// value of the accumulator) for 8-bit pixels or v_dot2_i32_i16 on pixel pairs for 16-bit pixels; vertical taps by
// v_dot2_i32_i16 on (row r, row r + 1) pairs built with one v_perm_b32 per sample.
#pragma once
#include "common.h"

namespace x265hip {

typedef short tile_v2i16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int tile_dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(tile_v2i16, a), __builtin_bit_cast(tile_v2i16, b), c, false);
}
__device__ __forceinline__ uint32_t tile_sel3(int f, uint32_t a1, uint32_t a2, uint32_t a3) { return f == 1 ? a1 : (f == 2 ? a2 : a3); }
__device__ __forceinline__ int tile_clip16(int v, int maxVal)
{
    const int16_t s = (int16_t)v;                   // the reference narrows to int16_t before clipping
    return s < 0 ? 0 : (s > maxVal ? maxVal : s);
}

// 4 horizontal 8-tap sums (no rounding); rp = byte address of sample (x0 - 3) of the row
template <int BPP>
__device__ __forceinline__ void tile_hrow(const uint8_t* rp, int xf, int (&out)[4])
{
    if (BPP == 1)
    {
        const uint32_t c03 = tile_sel3(xf, 0x3af604ffu, 0x28f504ffu, 0x11fb0100u), c47 = tile_sel3(xf, 0x0001fb11u, 0xff04f528u, 0xff04f63au);
        const uint32_t w0 = ld_u32(rp) ^ 0x80808080u, w1 = ld_u32(rp + 4) ^ 0x80808080u, w2 = ld_u32(rp + 8) ^ 0x80808080u;
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
        uint32_t dd[6];
#pragma unroll
        for (int k = 0; k < 6; k++) dd[k] = ld_u32(rp + 4 * k);
#pragma unroll
        for (int x = 0; x < 4; x++)
        {
            int sacc = 0;
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                const int k = (x >> 1) + j;
                sacc = tile_dot2((x & 1) ? __builtin_amdgcn_alignbyte(dd[k + 1], dd[k], 2) : dd[k], cp[j], sacc);
            }
            out[x] = sacc;
        }
    }
}

// Predicted samples of the 4x4 tile whose integer-position top-left sample sits at byte address `org` (row pitch
// strideB bytes), for the fractional offsets xf, yf in [0,3].  depth = bit depth of the pixels.
template <int BPP>
__device__ __forceinline__ void tile_predict(const uint8_t* org, long strideB, int xf, int yf, int depth, int (&d)[4][4])
{
    const int maxVal = (1 << depth) - 1, headRoom = 14 - depth;
    if (!(xf | yf))
    {
#pragma unroll
        for (int y = 0; y < 4; y++)
        {
            const uint8_t* rp = org + y * strideB;
            if (BPP == 1)
            {
                const uint32_t w = ld_u32(rp);
                d[y][0] = w & 0xff; d[y][1] = (w >> 8) & 0xff; d[y][2] = (w >> 16) & 0xff; d[y][3] = w >> 24;
            }
            else
            {
                const uint32_t w0 = ld_u32(rp), w1 = ld_u32(rp + 4);
                d[y][0] = w0 & 0xffff; d[y][1] = w0 >> 16; d[y][2] = w1 & 0xffff; d[y][3] = w1 >> 16;
            }
        }
        return;
    }
    if (!yf)
    {
#pragma unroll
        for (int y = 0; y < 4; y++)
        {
            int hs[4];
            tile_hrow<BPP>(org + y * strideB - 3 * BPP, xf, hs);
#pragma unroll
            for (int x = 0; x < 4; x++) d[y][x] = tile_clip16((hs[x] + 32) >> 6, maxVal);
        }
        return;
    }
    const int shiftPS = 6 - headRoom, offPS = -(8192 << shiftPS);
    const int shiftSP = 6 + headRoom, offSP = (1 << (shiftSP - 1)) + (8192 << 6);
    const uint32_t cv[4] = { tile_sel3(yf, 0x0004ffffu, 0x0004ffffu, 0x00010000u), tile_sel3(yf, 0x003afff6u, 0x0028fff5u, 0x0011fffbu),
                             tile_sel3(yf, 0xfffb0011u, 0xfff50028u, 0xfff6003au), tile_sel3(yf, 0x00000001u, 0xffff0004u, 0xffff0004u) };
    uint32_t pairs[10][4];                           // (row r, row r + 1) at the 4 columns, rows -3 .. +7
    if (!xf)
    {
        uint32_t raw[11][2];
#pragma unroll
        for (int t = 0; t < 11; t++)
        {
            const uint8_t* rp = org + (t - 3) * strideB;
            raw[t][0] = ld_u32(rp);
            raw[t][1] = BPP == 2 ? ld_u32(rp + 4) : 0;
        }
#pragma unroll
        for (int t = 0; t < 10; t++)
#pragma unroll
            for (int x = 0; x < 4; x++)
                pairs[t][x] = BPP == 1 ? __builtin_amdgcn_perm(raw[t + 1][0], raw[t][0], 0x0c000c00u | (uint32_t)x | ((uint32_t)(4 + x) << 16))
                                       : __builtin_amdgcn_perm(raw[t + 1][x >> 1], raw[t][x >> 1], (x & 1) ? 0x07060302u : 0x05040100u);
    }
    else
    {
        int im[11][4];
#pragma unroll
        for (int t = 0; t < 11; t++)
        {
            int hs[4];
            tile_hrow<BPP>(org + (t - 3) * strideB - 3 * BPP, xf, hs);
#pragma unroll
            for (int x = 0; x < 4; x++) im[t][x] = (hs[x] + offPS) >> shiftPS;
        }
#pragma unroll
        for (int t = 0; t < 10; t++)
#pragma unroll
            for (int x = 0; x < 4; x++) pairs[t][x] = __builtin_amdgcn_perm((uint32_t)im[t + 1][x], (uint32_t)im[t][x], 0x05040100u);
    }
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
}

// sum of |H4 d H4^T| / 2 of a 4x4 difference block (always an integer: the 16 coefficients share one parity)
__device__ __forceinline__ int tile_satd4(const int (&d)[4][4])
{
    int t4[4][4], acc = 0;
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        const int s0 = d[y][0] + d[y][1], s1 = d[y][0] - d[y][1], s2 = d[y][2] + d[y][3], s3 = d[y][2] - d[y][3];
        t4[y][0] = s0 + s2; t4[y][1] = s1 + s3; t4[y][2] = s0 - s2; t4[y][3] = s1 - s3;
    }
#pragma unroll
    for (int x = 0; x < 4; x++)
    {
        const int s0 = t4[0][x] + t4[1][x], s1 = t4[0][x] - t4[1][x], s2 = t4[2][x] + t4[3][x], s3 = t4[2][x] - t4[3][x];
        acc += abs(s0 + s2) + abs(s1 + s3) + abs(s0 - s2) + abs(s1 - s3);
    }
    return acc >> 1;
}

} // namespace x265hip
