// subpel_kernels.hip - sub-pel motion refinement for every PU of every CTU, on gfx950.
//
// Reference semantics: MotionEstimate::motionEstimate after the integer search
// (source/encoder/motion.cpp:1448-1561, non-lowres branch: SubpelWorkload table :48-58, square1 pattern :66,
// half-pel then quarter-pel iterations, COPY2_IF_LT strict-less) and MotionEstimate::subpelCompare
// (:1571-1664, luma: pu[].luma_hpp / luma_vpp / luma_hvpp into a blockwidth-stride buffer, then pu[].sad or
// pu[].satd against the PU source copy).  Interpolation arithmetic: source/common/ipfilter.cpp:79-118,
// :164-203, :120-162 + :319-369 (hps with row extension then vertical sp), SATD source/common/pixel.cpp:210-297.
//
// Mapping: one workgroup per (CTU, PU level).  The CTU source block and, per PU, the reference patch around its
// integer motion vector (+-3 pixels of drift + 8-tap apron) are staged in LDS once; every candidate of an
// iteration (4 or 8 directions) of every PU is evaluated concurrently, one thread per (PU, candidate, 4x4 tile): the thread
// interpolates its 16 samples straight from the LDS patch (hv: 11 horizontally filtered rows kept in
// registers), takes the 4x4 Hadamard (or SAD) in registers and adds its partial cost to an LDS bin.  The
// serial decision loop of the reference then runs in one thread per PU on the reduced costs, so no
// host round trip sits between the candidates - this is SURVEY section 8(f) item 1 for the sub-pel part.
#include "common.h"

namespace x265hip {

struct SubpelArgs
{
    const uint8_t* fenc; long fencStrideB;
    const uint8_t* fref; long frefStrideB;
    int ctusW, range, depth;
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

__device__ __forceinline__ int sp_clip16(int v, int maxVal)
{
    const int16_t s = (int16_t)v;
    return s < 0 ? 0 : (s > maxVal ? maxVal : s);
}

// One workgroup per (CTU, PU level): ALL PUs of the level are refined together, so every evaluation round has
// npu * ndirs * tiles = 1024 (4 directions) or 2048 (8) independent (PU, candidate, 4x4 tile) items for the 256
// threads whatever the PU size.  Per-PU state (best cost / mv, alive flag, candidate costs) lives in LDS; the
// round structure of the reference loop is uniform, PUs that stop early simply contribute no items.
template <typename Px>
__global__ void __launch_bounds__(256) subpel_refine_kernel(SubpelArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smemRaw[];
    Px* smem = reinterpret_cast<Px*>(smemRaw);
    __shared__ int sCost[64][9];
    __shared__ int sB[64], sQx[64], sQy[64], sIx[64], sIy[64], sAlive[64];
    constexpr int BPP = sizeof(Px);

    const int level = blockIdx.y;                          // all four PU levels of a CTU run concurrently
    const int n = 8 << level, npu = 64 >> (2 * level);
    const int lbase = level == 0 ? 0 : (level == 1 ? 64 : (level == 2 ? 80 : 84));
    const int ctu = blockIdx.x;
    const int cx = (ctu % a.ctusW) * 64, cy = (ctu / a.ctusW) * 64;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int R = a.range, NC = 2 * R + 1;
    const int maxVal = (1 << a.depth) - 1, headRoom = 14 - a.depth;
    const int pw = n + 2 * SP_MARGIN + 1;                  // patch width = height
    const int psz = pw * pw;
    Px* src = smem;                                   // 64 x 64 source CTU (all PUs of the level tile it)
    Px* patches = smem + 64 * 64;                     // npu patches of pw x pw

    if (tid < npu)
    {
        const unsigned long long key = a.bestIn[(size_t)ctu * 85 + lbase + tid];
        const int idx = (int)(key & 0xffffffffu);
        const int imx = (idx % NC) - R, imy = (idx / NC) - R;
        sIx[tid] = imx; sIy[tid] = imy;
        sQx[tid] = imx * 4; sQy[tid] = imy * 4;
        const int c = (int)(key >> 32);
        sB[tid] = c;
        sAlive[tid] = c != 0;                              // zero residual: skip refinement (motion.cpp:1464-1469)
    }
    {
        const Px* fe = reinterpret_cast<const Px*>(a.fenc + (long)cy * a.fencStrideB) + cx;
        const long fst = a.fencStrideB / BPP;
        for (int i = tid; i < 64 * 64; i += nth) src[i] = fe[(i >> 6) * fst + (i & 63)];
    }
    __syncthreads();
    {
        const long rst = a.frefStrideB / BPP;
        for (int i = tid; i < npu * psz; i += nth)
        {
            const int pu = i / psz, r = i - pu * psz, y = r / pw, x = r - y * pw;
            const int bxz = (pu & 1) | ((pu >> 1) & 2) | ((pu >> 2) & 4), byz = ((pu >> 1) & 1) | ((pu >> 2) & 2) | ((pu >> 3) & 4);
            const Px* rf = reinterpret_cast<const Px*>(a.fref) + (long)(cy + byz * n + sIy[pu] - SP_MARGIN + y) * rst
                           + (cx + bxz * n + sIx[pu] - SP_MARGIN + x);
            patches[i] = rf[0];
        }
    }
    __syncthreads();

    const int tpr = n >> 2;
    const int tshift = level * 2 + 2;                      // log2(tiles per PU)
    const int ntiles = 1 << tshift;

    auto tile_cost = [&](const int pu, const int qx, const int qy, const int tile, const bool useSatd) -> int
    {
        const Px* patch = patches + pu * psz;
        const int bxz = (pu & 1) | ((pu >> 1) & 2) | ((pu >> 2) & 4), byz = ((pu >> 1) & 1) | ((pu >> 2) & 2) | ((pu >> 3) & 4);
        const int ty = tile / tpr, tx = tile - ty * tpr;
        const int ox = (qx >> 2) - sIx[pu] + SP_MARGIN + tx * 4, oy = (qy >> 2) - sIy[pu] + SP_MARGIN + ty * 4;
        const int xf = qx & 3, yf = qy & 3;
        int d[4][4];
        if (!(xf | yf))
        {
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++) d[y][x] = patch[(oy + y) * pw + ox + x];
        }
        else if (!yf)
        {
#pragma unroll
            for (int y = 0; y < 4; y++)
            {
                int in[11];
#pragma unroll
                for (int t = 0; t < 11; t++) in[t] = patch[(oy + y) * pw + ox + t - 3];
#pragma unroll
                for (int x = 0; x < 4; x++)
                {
                    int s = 0;
#pragma unroll
                    for (int t = 0; t < 8; t++) s += in[x + t] * kSpLumaTaps[xf][t];
                    d[y][x] = sp_clip16((s + 32) >> 6, maxVal);
                }
            }
        }
        else if (!xf)
        {
#pragma unroll
            for (int x = 0; x < 4; x++)
            {
                int in[11];
#pragma unroll
                for (int t = 0; t < 11; t++) in[t] = patch[(oy + t - 3) * pw + ox + x];
#pragma unroll
                for (int y = 0; y < 4; y++)
                {
                    int s = 0;
#pragma unroll
                    for (int t = 0; t < 8; t++) s += in[y + t] * kSpLumaTaps[yf][t];
                    d[y][x] = sp_clip16((s + 32) >> 6, maxVal);
                }
            }
        }
        else
        {
            const int shiftPS = 6 - headRoom, offPS = -(8192 << shiftPS);
            const int shiftSP = 6 + headRoom, offSP = (1 << (shiftSP - 1)) + (8192 << 6);
            int16_t im[11][4];
#pragma unroll
            for (int r = 0; r < 11; r++)
            {
                int in[11];
#pragma unroll
                for (int t = 0; t < 11; t++) in[t] = patch[(oy + r - 3) * pw + ox + t - 3];
#pragma unroll
                for (int x = 0; x < 4; x++)
                {
                    int s = 0;
#pragma unroll
                    for (int t = 0; t < 8; t++) s += in[x + t] * kSpLumaTaps[xf][t];
                    im[r][x] = (int16_t)((s + offPS) >> shiftPS);
                }
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
        const Px* sp = src + (byz * n + ty * 4) * 64 + bxz * n + tx * 4;
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 4; x++) d[y][x] = (int)sp[y * 64 + x] - d[y][x];
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

    // PU state: 0 = zero residual at the integer mv, refinement skipped (motion.cpp:1464-1469); 1 = searching;
    // 2 = the current phase ended for this PU (the reference's `break`), waiting for the next phase.
    // evaluate(): for every searching PU, candidates best + square1[first..last] * step -> sCost[pu][first..last]
    auto evaluate = [&](const int first, const int last, const int step, const bool useSatd)
    {
        for (int i = tid; i < 64 * 9; i += nth) (&sCost[0][0])[i] = 0;
        __syncthreads();
        const int perPu = (last - first + 1) << tshift;
        for (int it = tid; it < npu * perPu; it += nth)
        {
            const int pu = it / perPu, r = it - pu * perPu;
            if (sAlive[pu] != 1) continue;
            const int c = first + (r >> tshift), tile = r & (ntiles - 1);
            atomicAdd(&sCost[pu][c], tile_cost(pu, sQx[pu] + kSpSquare1[c][0] * step, sQy[pu] + kSpSquare1[c][1] * step, tile, useSatd));
        }
        __syncthreads();
    };
    auto mvcost = [&](const int qx, const int qy) { return (int)a.costQ[qx + a.qoff] + (int)a.costQ[qy + a.qoff]; };
    // one iteration's decision for PU `tid` (COPY2_IF_LT: strict less, first direction wins; motion.cpp:1520-1531)
    auto decide = [&](const int ndirs, const int step)
    {
        if (tid < npu && sAlive[tid] == 1)
        {
            int bcost = sB[tid], bdir = 0;
            const int qx0 = sQx[tid], qy0 = sQy[tid];
            for (int i = 1; i <= ndirs; i++)
            {
                const int c = sCost[tid][i] + mvcost(qx0 + kSpSquare1[i][0] * step, qy0 + kSpSquare1[i][1] * step);
                if (c < bcost) { bcost = c; bdir = i; }
            }
            sB[tid] = bcost;
            if (bdir) { sQx[tid] = qx0 + kSpSquare1[bdir][0] * step; sQy[tid] = qy0 + kSpSquare1[bdir][1] * step; }
            else sAlive[tid] = 2;
        }
        __syncthreads();
    };
    auto remeasure = [&]()          // SATD of the current best replaces the SAD-based cost
    {
        evaluate(0, 0, 0, true);
        if (tid < npu && sAlive[tid] == 1) sB[tid] = sCost[tid][0] + mvcost(sQx[tid], sQy[tid]);
        __syncthreads();
    };

    const bool hs = a.hpelSatd != 0;
    if (hs) remeasure();
    for (int iter = 0; iter < a.hpelIters; iter++)
    {
        evaluate(1, a.hpelDirs, 2, hs);
        decide(a.hpelDirs, 2);
    }
    if (tid < npu && sAlive[tid] == 2) sAlive[tid] = 1;          // next phase: every refined PU searches again
    __syncthreads();
    if (!hs) remeasure();
    for (int iter = 0; iter < a.qpelIters; iter++)
    {
        evaluate(1, a.qpelDirs, 1, true);
        decide(a.qpelDirs, 1);
    }
    if (tid < npu)
    {
        int bcost = sB[tid];
        if (sAlive[tid] == 0) bcost = mvcost(sQx[tid], sQy[tid]);     // zero-residual PUs return the mv cost only
        a.out[(size_t)ctu * 85 + lbase + tid] = make_int2(bcost, (sQx[tid] & 0xffff) | (sQy[tid] << 16));
    }
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
    // one launch: grid.y = PU level; LDS sized for level 0 (64 patches of 23 x 23), the largest
    const int pw0 = 8 + 2 * SP_MARGIN + 1;
    const size_t lds = (size_t)(64 * 64 + 64 * pw0 * pw0) * bpp;
    static_assert(64 * (8 + 15) * (8 + 15) >= 1 * (64 + 15) * (64 + 15) && 64 * 23 * 23 >= 16 * 31 * 31 && 64 * 23 * 23 >= 4 * 47 * 47, "level 0 is the largest");
    if (p->depth == 8)
        hipLaunchKernelGGL(subpel_refine_kernel<uint8_t>, dim3(nctu, 4), dim3(256), lds, s, a);
    else
    {
        static bool attr = false;
        if (!attr) { X265HIP_TRY(hipFuncSetAttribute((const void*)subpel_refine_kernel<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
        hipLaunchKernelGGL(subpel_refine_kernel<uint16_t>, dim3(nctu, 4), dim3(256), lds, s, a);
    }
    X265HIP_TRY(hipGetLastError());
    return 0;
}
