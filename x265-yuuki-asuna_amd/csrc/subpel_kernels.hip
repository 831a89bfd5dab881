// subpel_kernels.hip - sub-pel motion refinement for every PU of every CTU, on gfx950.
//
// Reference semantics: MotionEstimate::motionEstimate after the integer search
// (source/encoder/motion.cpp:1448-1561, non-lowres branch: SubpelWorkload table :48-58, square1 pattern :66,
// half-pel then quarter-pel iterations, COPY2_IF_LT strict-less) and MotionEstimate::subpelCompare
// (:1571-1664, luma: pu[].luma_hpp / luma_vpp / luma_hvpp into a blockwidth-stride buffer, then pu[].sad or
// pu[].satd against the PU source copy).  Interpolation arithmetic: source/common/ipfilter.cpp:79-118,
// :164-203, :120-162 + :319-369 (hps with row extension then vertical sp), SATD source/common/pixel.cpp:210-297.
//
// Mapping: one workgroup per PU.  The PU source block and the reference patch around the integer motion
// vector (+-3 pixels of drift + 8-tap apron) are staged in LDS once; every candidate of an iteration
// (4 or 8 directions) is evaluated concurrently, one thread per (candidate, 4x4 tile): the thread
// interpolates its 16 samples straight from the LDS patch (hv: 11 horizontally filtered rows kept in
// registers), takes the 4x4 Hadamard (or SAD) in registers and adds its partial cost to an LDS bin.  The
// serial decision loop of the reference then runs redundantly in every thread on the reduced costs, so no
// host round trip sits between the candidates - this is SURVEY section 8(f) item 1 for the sub-pel part.
#include "common.h"

namespace x265hip {

struct SubpelArgs
{
    const uint8_t* fenc; long fencStrideB;
    const uint8_t* fref; long frefStrideB;
    int ctusW, range, depth;
    int level;                       // 0..3 -> 8, 16, 32, 64
    const unsigned long long* bestIn;
    const uint16_t* costQ; int qoff;
    int hpelIters, hpelDirs, qpelIters, qpelDirs, hpelSatd;
    int2* out;                       // {cost, qx | qy << 16} per PU, [ctu][85]
};

__constant__ int16_t kSpLumaTaps[4][8] = {
    { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
    { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
__constant__ int kSpSquare1[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { -1, 1 }, { 1, -1 }, { 1, 1 } };

constexpr int SP_MARGIN = 7;                        // patch origin = integer mv - 7: 3 px of drift + 3 px of left apron + 1
constexpr int SP_PITCH = 64 + 2 * SP_MARGIN + 2;    // 80 int16 per LDS row

__device__ __forceinline__ int sp_clip16(int v, int maxVal)
{
    const int16_t s = (int16_t)v;
    return s < 0 ? 0 : (s > maxVal ? maxVal : s);
}

template <typename Px>
__global__ void __launch_bounds__(256) subpel_refine_kernel(SubpelArgs a)
{
    __shared__ int16_t patch[(64 + 2 * SP_MARGIN + 2) * SP_PITCH];
    __shared__ int16_t src[64 * 64];
    __shared__ int costs[9];
    constexpr int BPP = sizeof(Px);

    const int n = 8 << a.level, npu = (64 / n) * (64 / n);
    const int lbase = a.level == 0 ? 0 : (a.level == 1 ? 64 : (a.level == 2 ? 80 : 84));
    const int ctu = blockIdx.x / npu, z = blockIdx.x - ctu * npu;
    const int bxz = (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4), byz = ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4);
    const int px = (ctu % a.ctusW) * 64 + bxz * n, py = (ctu / a.ctusW) * 64 + byz * n;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int R = a.range, NC = 2 * R + 1;
    const int maxVal = (1 << a.depth) - 1, headRoom = 14 - a.depth;

    const unsigned long long key = a.bestIn[(size_t)ctu * 85 + lbase + z];
    const int idx = (int)(key & 0xffffffffu);
    int bcost = (int)(key >> 32);
    const int imx = (idx % NC) - R, imy = (idx / NC) - R;                      // integer mv
    int bqx = imx * 4, bqy = imy * 4;

    // ---- stage source block and reference patch ------------------------------------------------------
    const int pw = n + 2 * SP_MARGIN + 1;                                        // columns / rows of the patch actually used
    {
        const Px* fe = reinterpret_cast<const Px*>(a.fenc + (long)py * a.fencStrideB) + px;
        const long fst = a.fencStrideB / BPP;
        for (int i = tid; i < n * n; i += nth) { const int y = i / n, x = i - y * n; src[y * 64 + x] = (int16_t)fe[y * fst + x]; }
        const Px* rf = reinterpret_cast<const Px*>(a.fref + (long)(py + imy - SP_MARGIN) * a.frefStrideB) + (px + imx - SP_MARGIN);
        const long rst = a.frefStrideB / BPP;
        for (int i = tid; i < pw * pw; i += nth) { const int y = i / pw, x = i - y * pw; patch[y * SP_PITCH + x] = (int16_t)rf[y * rst + x]; }
    }
    __syncthreads();

    const int tpr = n >> 2, ntiles = tpr * tpr;
    const int tshift = a.level * 2 + 2;                                          // log2(ntiles)

    // cost of one candidate for one 4x4 tile; (qx, qy) = absolute qpel mv
    auto tile_cost = [&](const int qx, const int qy, const int tile, const bool useSatd) -> int
    {
        const int ty = tile / tpr, tx = tile - ty * tpr;
        const int ox = (qx >> 2) - imx + SP_MARGIN + tx * 4, oy = (qy >> 2) - imy + SP_MARGIN + ty * 4;   // patch coords of the tile's full-pel sample
        const int xf = qx & 3, yf = qy & 3;
        int d[4][4];
        if (!(xf | yf))
        {
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++) d[y][x] = patch[(oy + y) * SP_PITCH + ox + x];
        }
        else if (!yf)
        {
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++)
                {
                    int s = 0;
#pragma unroll
                    for (int t = 0; t < 8; t++) s += (int)patch[(oy + y) * SP_PITCH + ox + x + t - 3] * kSpLumaTaps[xf][t];
                    d[y][x] = sp_clip16((s + 32) >> 6, maxVal);
                }
        }
        else if (!xf)
        {
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++)
                {
                    int s = 0;
#pragma unroll
                    for (int t = 0; t < 8; t++) s += (int)patch[(oy + y + t - 3) * SP_PITCH + ox + x] * kSpLumaTaps[yf][t];
                    d[y][x] = sp_clip16((s + 32) >> 6, maxVal);
                }
        }
        else
        {
            const int shiftPS = 6 - headRoom, offPS = -(8192 << shiftPS);
            const int shiftSP = 6 + headRoom, offSP = (1 << (shiftSP - 1)) + (8192 << 6);
            int16_t im[11][4];
#pragma unroll
            for (int r = 0; r < 11; r++)
#pragma unroll
                for (int x = 0; x < 4; x++)
                {
                    int s = 0;
#pragma unroll
                    for (int t = 0; t < 8; t++) s += (int)patch[(oy + r - 3) * SP_PITCH + ox + x + t - 3] * kSpLumaTaps[xf][t];
                    im[r][x] = (int16_t)((s + offPS) >> shiftPS);
                }
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++)
                {
                    int s = 0;
#pragma unroll
                    for (int t = 0; t < 8; t++) s += (int)im[y + t][x] * kSpLumaTaps[yf][t];
                    d[y][x] = sp_clip16((s + offSP) >> shiftSP, maxVal);
                }
        }
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++) d[y][x] = (int)src[(ty * 4 + y) * 64 + tx * 4 + x] - d[y][x];
        int acc = 0;
        if (!useSatd)
        {
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++) acc += abs(d[y][x]);
            return acc;
        }
        int t4[4][4];
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
    };

    // evaluate candidates base + square1[first..last] * step concurrently; results in costs[first..last]
    auto evaluate = [&](const int first, const int last, const int step, const bool useSatd)
    {
        if (tid < 9) costs[tid] = 0;
        __syncthreads();
        const int ncand = last - first + 1;
        for (int it = tid; it < (ncand << tshift); it += nth)
        {
            const int c = first + (it >> tshift), tile = it & (ntiles - 1);
            const int qx = bqx + kSpSquare1[c][0] * step, qy = bqy + kSpSquare1[c][1] * step;
            atomicAdd(&costs[c], tile_cost(qx, qy, tile, useSatd));
        }
        __syncthreads();
    };
    auto mvcost = [&](const int qx, const int qy) { return (int)a.costQ[qx + a.qoff] + (int)a.costQ[qy + a.qoff]; };

    if (!bcost)
        bcost = mvcost(bqx, bqy);
    else
    {
        const bool hs = a.hpelSatd != 0;
        if (hs) { evaluate(0, 0, 0, true); bcost = costs[0] + mvcost(bqx, bqy); __syncthreads(); }
        for (int iter = 0; iter < a.hpelIters; iter++)
        {
            evaluate(1, a.hpelDirs, 2, hs);
            int bdir = 0;
            for (int i = 1; i <= a.hpelDirs; i++)
            {
                const int c = costs[i] + mvcost(bqx + kSpSquare1[i][0] * 2, bqy + kSpSquare1[i][1] * 2);
                if (c < bcost) { bcost = c; bdir = i; }
            }
            __syncthreads();
            if (bdir) { bqx += kSpSquare1[bdir][0] * 2; bqy += kSpSquare1[bdir][1] * 2; }
            else break;
        }
        if (!hs) { evaluate(0, 0, 0, true); bcost = costs[0] + mvcost(bqx, bqy); __syncthreads(); }
        for (int iter = 0; iter < a.qpelIters; iter++)
        {
            evaluate(1, a.qpelDirs, 1, true);
            int bdir = 0;
            for (int i = 1; i <= a.qpelDirs; i++)
            {
                const int c = costs[i] + mvcost(bqx + kSpSquare1[i][0], bqy + kSpSquare1[i][1]);
                if (c < bcost) { bcost = c; bdir = i; }
            }
            __syncthreads();
            if (bdir) { bqx += kSpSquare1[bdir][0]; bqy += kSpSquare1[bdir][1]; }
            else break;
        }
    }
    if (tid == 0)
        a.out[(size_t)ctu * 85 + lbase + z] = make_int2(bcost, (bqx & 0xffff) | (bqy << 16));
}

} // namespace x265hip

using namespace x265hip;

extern "C" int x265hip_subpel_refine(const x265hip_subpel_params* p, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !p->fenc || !p->fref || !p->best_in || !p->cost_q || !p->out) { set_error("subpel_refine: NULL operand"); return X265HIP_EINVAL; }
    if ((p->width & 63) || (p->height & 63) || p->width <= 0 || p->height <= 0) { set_error("subpel_refine: width/height must be multiples of 64"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("subpel_refine: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->subme < 0 || p->subme > 7) { set_error("subpel_refine: subme %d out of [0,7]", p->subme); return X265HIP_EINVAL; }
    static const int wl[8][5] = { { 1, 4, 0, 4, 0 }, { 1, 4, 1, 4, 0 }, { 1, 4, 1, 4, 1 }, { 2, 4, 1, 4, 1 },
                                  { 2, 4, 2, 4, 1 }, { 1, 8, 1, 8, 1 }, { 2, 8, 1, 8, 1 }, { 2, 8, 2, 8, 1 } };   // motion.cpp:48-58
    const int bpp = p->depth == 8 ? 1 : 2;
    SubpelArgs a;
    a.fenc = (const uint8_t*)p->fenc; a.fencStrideB = (long)p->fenc_stride * bpp;
    a.fref = (const uint8_t*)p->fref; a.frefStrideB = (long)p->fref_stride * bpp;
    a.ctusW = p->width / 64; a.range = p->range; a.depth = p->depth;
    a.bestIn = (const unsigned long long*)p->best_in; a.costQ = p->cost_q; a.qoff = p->qoff;
    a.hpelIters = wl[p->subme][0]; a.hpelDirs = wl[p->subme][1]; a.qpelIters = wl[p->subme][2]; a.qpelDirs = wl[p->subme][3]; a.hpelSatd = wl[p->subme][4];
    a.out = (int2*)p->out;
    const int nctu = a.ctusW * (p->height / 64);
    hipStream_t s = (hipStream_t)stream;
    for (int level = 0; level < 4; level++)
    {
        a.level = level;
        const int npu = 64 >> (2 * level);
        const int threads = level == 0 ? 64 : (level == 1 ? 128 : 256);
        if (p->depth == 8) hipLaunchKernelGGL(subpel_refine_kernel<uint8_t>, dim3(nctu * npu), dim3(threads), 0, s, a);
        else hipLaunchKernelGGL(subpel_refine_kernel<uint16_t>, dim3(nctu * npu), dim3(threads), 0, s, a);
    }
    X265HIP_TRY(hipGetLastError());
    return 0;
}
