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
// Bound: HBM writes (15 + 2 * 63 / 4 = 46.5 output bytes per luma source byte at 4:2:0); the source tile and its apron are re-read
// through L2 by the phases of the same tile, which are adjacent in the grid (phase = fastest workgroup index).
#include "common.h"
#include "tile_interp.h"

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

// grid: x = phase - 1 (15), y = blocks of 256 tiles along the rows' tiles, z = tile row.  One 4x4 tile per lane.
template <typename Px>
__global__ void __launch_bounds__(256) phase_luma_kernel(PhaseArgs a)
{
    constexpr int BPP = sizeof(Px);
    const int phase = blockIdx.x + 1, xf = phase & 3, yf = phase >> 2;
    const int tx = blockIdx.y * 256 + threadIdx.x, ty = blockIdx.z + 1;          // the first 4 and the last 8 rows are not produced
    if (tx >= a.tilesW) return;
    const long off = (long)(ty * 4) * a.strideB + (long)tx * 4 * BPP;
    int d[4][4];
    tile_predict<BPP>(a.src + off, a.strideB, xf, yf, a.depth, d);
    phase_store<Px>(a.dst + (size_t)(phase - 1) * a.planeBytes + off, a.strideB, d);
}

// 4-tap chroma set, eighth-sample phases: phase = yf * 8 + xf
template <typename Px>
__global__ void __launch_bounds__(256) phase_chroma_kernel(PhaseArgs a)
{
    constexpr int BPP = sizeof(Px);
    const int phase = blockIdx.x + 1, xf = phase & 7, yf = phase >> 3;
    const int tx = blockIdx.y * 256 + threadIdx.x, ty = blockIdx.z + 1;
    if (tx >= a.tilesW) return;
    const long off = (long)(ty * 4) * a.strideB + (long)tx * 4 * BPP;
    const int maxVal = (1 << a.depth) - 1, headRoom = 14 - a.depth;
    const int cx0 = kPhaseChromaTaps[xf][0], cx1 = kPhaseChromaTaps[xf][1], cx2 = kPhaseChromaTaps[xf][2], cx3 = kPhaseChromaTaps[xf][3];
    const int cy0 = kPhaseChromaTaps[yf][0], cy1 = kPhaseChromaTaps[yf][1], cy2 = kPhaseChromaTaps[yf][2], cy3 = kPhaseChromaTaps[yf][3];
    int d[4][4];
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
    if (!yf)
    {
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++)
                d[y][x] = tile_clip16((cx0 * s[y + 1][x] + cx1 * s[y + 1][x + 1] + cx2 * s[y + 1][x + 2] + cx3 * s[y + 1][x + 3] + 32) >> 6, maxVal);
    }
    else if (!xf)
    {
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++)
                d[y][x] = tile_clip16((cy0 * s[y][x + 1] + cy1 * s[y + 1][x + 1] + cy2 * s[y + 2][x + 1] + cy3 * s[y + 3][x + 1] + 32) >> 6, maxVal);
    }
    else
    {
        const int shiftPS = 6 - headRoom, offPS = -(8192 << shiftPS);
        const int shiftSP = 6 + headRoom, offSP = (1 << (shiftSP - 1)) + (8192 << 6);
        int im[7][4];
#pragma unroll
        for (int r = 0; r < 7; r++)
#pragma unroll
            for (int x = 0; x < 4; x++)
                im[r][x] = (int16_t)((cx0 * s[r][x] + cx1 * s[r][x + 1] + cx2 * s[r][x + 2] + cx3 * s[r][x + 3] + offPS) >> shiftPS);
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++)
                d[y][x] = tile_clip16((cy0 * im[y][x] + cy1 * im[y + 1][x] + cy2 * im[y + 2][x] + cy3 * im[y + 3][x] + offSP) >> shiftSP, maxVal);
    }
    phase_store<Px>(a.dst + (size_t)(phase - 1) * a.planeBytes + off, a.strideB, d);
}

} // namespace x265hip

using namespace x265hip;

extern "C" int x265hip_phase_planes(const x265hip_phase_planes_params* p, void* stream)
{
    if (!p || !p->src || !p->dst) { set_error("phase_planes: NULL argument"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("phase_planes: depth %d", p->depth); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    if (p->stride <= 0 || p->rows <= 0 || (p->rows & 3) || ((p->stride * bpp) & 3) || (p->stride & 3))
    { set_error("phase_planes: stride %ld / rows %d must be positive multiples of 4", (long)p->stride, p->rows); return X265HIP_EINVAL; }
    if ((uintptr_t)p->dst & 3) { set_error("phase_planes: dst must be 4-byte aligned"); return X265HIP_EINVAL; }
    int rc = ensure_device();
    if (rc) return rc;
    PhaseArgs a;
    a.src = (const uint8_t*)p->src; a.dst = (uint8_t*)p->dst; a.strideB = (long)p->stride * bpp; a.rows = p->rows;
    a.tilesW = (int)(p->stride / 4); a.depth = p->depth; a.planeBytes = (size_t)p->stride * p->rows * bpp;
    if (p->rows / 4 > 65535 || p->rows < 16) { set_error("phase_planes: %d rows", p->rows); return X265HIP_EINVAL; }
    // tile rows 1 .. rows / 4 - 3: a tile reads 3 rows above and 7 below itself (and a few bytes of the neighbouring rows at the row
    // ends), so every access stays inside the plane without any guard memory around it
    const dim3 grid(p->chroma ? 63 : 15, (a.tilesW + 255) / 256, p->rows / 4 - 3);
    hipStream_t s = (hipStream_t)stream;
    if (p->chroma)
    {
        if (bpp == 1) hipLaunchKernelGGL(phase_chroma_kernel<uint8_t>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(phase_chroma_kernel<uint16_t>, grid, dim3(256), 0, s, a);
    }
    else
    {
        if (bpp == 1) hipLaunchKernelGGL(phase_luma_kernel<uint8_t>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(phase_luma_kernel<uint16_t>, grid, dim3(256), 0, s, a);
    }
    return check_hip(hipGetLastError(), "phase_planes launch");
}
