// deblock_kernels.hip - in-loop deblocking of a device-resident luma reconstruction on gfx950
// (the deblocking half: the reconstruction stays on the GPU that hands it on as a reference).
//
// Reference semantics (source/common/deblock.cpp): Deblock::getBoundaryStrength :191-215 for P pictures with one reference
// (Bs 1 when either side of a transform edge has coded luma coefficients or the motion vectors differ by >= 4 quarter-pels,
// else 0; picture borders are not filtered, bsCuEdge :46-70); Deblock::edgeFilterLuma :317-415 - per 4-sample unit of an
// edge: beta / tc from the mean QP of the two sides (tables :499-509 = H.265 table 8-12), dE and the strong-filter decision
// (calcDP / calcDQ / useStrongFiltering :249-265), then primitives.pelFilterLumaStrong (loopfilter.cpp:140-159) or the
// file-static normal filter pelFilterLuma (:278-315).  All vertical edges of the picture are filtered before all horizontal
// edges; edges of one direction lie 8 samples apart and a filter changes at most 3 samples per side, so a pass is one
// launch with one thread per 4-sample unit and no ordering inside it.
#include "common.h"

namespace x265hip {

__constant__ unsigned char kDbTc[54] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2,
    2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24 };
__constant__ unsigned char kDbBeta[52] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17,
    18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64 };

struct BsArgs
{
    int width, height, level, sliceB;
    const int2* mv; const int2* mv1; const int8_t* ref0; const int8_t* ref1;
    const uint32_t* numSig; const uint8_t* intra;
    uint8_t* bsVer; uint8_t* bsHor;
};

// what getBoundaryStrength reads of the block on one side of an edge
struct BsSide
{
    int mx0, my0, mx1, my1;     // list-0 / list-1 mv, quarter samples (zero when the list is unused, deblock.cpp:213-214,225-226)
    int r0, r1;                 // reference picture id per list, -1 = unused
    int flags;                  // bit 0: coded coefficients, bit 1: intra CU (Bs 2 on its edges, deblock.cpp:198-199)
};

__device__ __forceinline__ BsSide db_block(const BsArgs& a, int x, int y)
{
    const int n = 8 << a.level, npu = 64 >> (2 * a.level), ctusW = a.width >> 6;
    const int lbase = a.level == 0 ? 0 : (a.level == 1 ? 64 : (a.level == 2 ? 80 : 84));
    const int ctu = (y >> 6) * ctusW + (x >> 6), bx = (x & 63) / n, by = (y & 63) / n;
    const int z = (bx & 1) | ((by & 1) << 1) | ((bx & 2) << 1) | ((by & 2) << 2) | ((bx & 4) << 2) | ((by & 4) << 3);
    const size_t blk = (size_t)ctu * npu + z, rec = (size_t)ctu * 85 + lbase + z;
    BsSide s;
    s.r0 = a.ref0 ? a.ref0[blk] : 0;
    s.r1 = a.ref1 ? a.ref1[blk] : -1;
    const int pk0 = s.r0 >= 0 ? a.mv[rec].y : 0;
    const int pk1 = (s.r1 >= 0 && a.mv1) ? a.mv1[rec].y : 0;
    s.mx0 = (int16_t)(pk0 & 0xffff); s.my0 = pk0 >> 16;
    s.mx1 = (int16_t)(pk1 & 0xffff); s.my1 = pk1 >> 16;
    s.flags = (a.numSig[blk] != 0) | ((a.intra && a.intra[blk]) ? 2 : 0);
    return s;
}

__device__ __forceinline__ bool db_far(int ax, int ay, int bx, int by) { return abs(ax - bx) >= 4 || abs(ay - by) >= 4; }

// Deblock::getBoundaryStrength (deblock.cpp:191-247) for the edge between blocks P and Q
__device__ __forceinline__ int db_strength(const BsArgs& a, const BsSide& p, const BsSide& q)
{
    if ((p.flags | q.flags) & 2) return 2;
    if (p.flags | q.flags) return 1;
    const bool f00 = db_far(q.mx0, q.my0, p.mx0, p.my0);
    if (!a.sliceB) return (p.r0 != q.r0 || f00) ? 1 : 0;
    if (!((p.r0 == q.r0 && p.r1 == q.r1) || (p.r0 == q.r1 && p.r1 == q.r0))) return 1;
    const bool f11 = db_far(q.mx1, q.my1, p.mx1, p.my1);
    const bool f10 = db_far(q.mx1, q.my1, p.mx0, p.my0), f01 = db_far(q.mx0, q.my0, p.mx1, p.my1);
    if (p.r0 != p.r1) return (p.r0 == q.r0 ? (f00 || f11) : (f10 || f01)) ? 1 : 0;
    return ((f00 || f11) && (f10 || f01)) ? 1 : 0;
}

// one thread per 4-sample unit of the 8x8 edge grid, both directions in one launch (blockIdx.y = direction)
__global__ void __launch_bounds__(256) deblock_bs_inter_kernel(BsArgs a)
{
    const int n = 8 << a.level;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.y == 0)
    {
        const int w8 = a.width >> 3, total = (a.height >> 2) * w8;
        if (i >= total) return;
        const int u = i / w8, ex = i - u * w8, x = ex * 8, y = u * 4;
        int bs = 0;
        if (x > 0 && (x % n) == 0) bs = db_strength(a, db_block(a, x - 1, y), db_block(a, x, y));
        a.bsVer[i] = (uint8_t)bs;
    }
    else
    {
        const int w4 = a.width >> 2, total = (a.height >> 3) * w4;
        if (i >= total) return;
        const int ey = i / w4, u = i - ey * w4, x = u * 4, y = ey * 8;
        int bs = 0;
        if (y > 0 && (y % n) == 0) bs = db_strength(a, db_block(a, x, y - 1), db_block(a, x, y));
        a.bsHor[i] = (uint8_t)bs;
    }
}

struct DbArgs
{
    uint8_t* rec; long strideB;
    int width, height, depth;
    const uint8_t* bs; const int8_t* qpMap;
    int qp, betaOffset, tcOffset;
};

// DIR 0: vertical edges (taps along x, 4 rows per unit); DIR 1: horizontal edges (taps along y, 4 columns per unit)
template <typename Px, int DIR>
__global__ void __launch_bounds__(256) deblock_luma_kernel(DbArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int w8 = a.width >> 3, w4 = a.width >> 2;
    const int total = DIR == 0 ? (a.height >> 2) * w8 : (a.height >> 3) * w4;
    if (i >= total) return;
    const int bs = a.bs[i];
    if (!bs) return;
    const long st = a.strideB / (long)sizeof(Px);
    int x, y, qp = a.qp;
    if (DIR == 0)
    {
        const int u = i / w8, ex = i - u * w8;
        x = ex * 8; y = u * 4;
        if (a.qpMap) qp = (a.qpMap[(y >> 3) * w8 + ex - 1] + a.qpMap[(y >> 3) * w8 + ex] + 1) >> 1;
    }
    else
    {
        const int ey = i / w4, u = i - ey * w4;
        x = u * 4; y = ey * 8;
        if (a.qpMap) qp = (a.qpMap[(ey - 1) * w8 + (x >> 3)] + a.qpMap[ey * w8 + (x >> 3)] + 1) >> 1;
    }
    Px* src = reinterpret_cast<Px*>(a.rec) + (long)y * st + x;
    const long offset = DIR == 0 ? 1 : st, srcStep = DIR == 0 ? st : 1;
    const int shift = a.depth - 8, maxVal = (1 << a.depth) - 1;
    // the unit's 4 lines x 8 taps
    int m[4][8];
#pragma unroll
    for (int l = 0; l < 4; l++)
#pragma unroll
        for (int t = 0; t < 8; t++) m[l][t] = src[l * srcStep + (t - 4) * offset];
    const int beta = (int)kDbBeta[clip3(0, 51, qp + a.betaOffset)] << shift;
    auto dP = [&](int l) { return abs(m[l][1] - 2 * m[l][2] + m[l][3]); };
    auto dQ = [&](int l) { return abs(m[l][4] - 2 * m[l][5] + m[l][6]); };
    const int dp0 = dP(0), dq0 = dQ(0), dp3 = dP(3), dq3 = dQ(3);
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    if (d0 + d3 >= beta) return;
    const int tc = (int)kDbTc[clip3(0, 53, qp + 2 * (bs - 1) + a.tcOffset)] << shift;
    auto strongLine = [&](int l) { return abs(m[l][0] - m[l][3]) + abs(m[l][7] - m[l][4]) < (beta >> 3) && abs(m[l][3] - m[l][4]) < ((tc * 5 + 1) >> 1); };
    const bool sw = 2 * d0 < (beta >> 2) && 2 * d3 < (beta >> 2) && strongLine(0) && strongLine(3);
    if (sw)
    {
        const int t2 = 2 * tc;
#pragma unroll
        for (int l = 0; l < 4; l++)
        {
            const int m0 = m[l][0], m1 = m[l][1], m2 = m[l][2], m3 = m[l][3], m4 = m[l][4], m5 = m[l][5], m6 = m[l][6], m7 = m[l][7];
            Px* p = src + l * srcStep;
            p[-3 * offset] = (Px)(clip3(-t2, t2, ((2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3) - m1) + m1);
            p[-2 * offset] = (Px)(clip3(-t2, t2, ((m1 + m2 + m3 + m4 + 2) >> 2) - m2) + m2);
            p[-offset]     = (Px)(clip3(-t2, t2, ((m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3) - m3) + m3);
            p[0]           = (Px)(clip3(-t2, t2, ((m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3) - m4) + m4);
            p[offset]      = (Px)(clip3(-t2, t2, ((m3 + m4 + m5 + m6 + 2) >> 2) - m5) + m5);
            p[2 * offset]  = (Px)(clip3(-t2, t2, ((m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3) - m6) + m6);
        }
        return;
    }
    const int sideThreshold = (beta + (beta >> 1)) >> 3;
    const bool maskP1 = dp0 + dp3 < sideThreshold, maskQ1 = dq0 + dq3 < sideThreshold;
    const int thrCut = tc * 10, tc2 = tc >> 1;
#pragma unroll
    for (int l = 0; l < 4; l++)
    {
        const int m1 = m[l][1], m2 = m[l][2], m3 = m[l][3], m4 = m[l][4], m5 = m[l][5], m6 = m[l][6];
        int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
        if (abs(delta) >= thrCut) continue;
        delta = clip3(-tc, tc, delta);
        Px* p = src + l * srcStep;
        p[-offset] = (Px)clip3(0, maxVal, m3 + delta);
        p[0] = (Px)clip3(0, maxVal, m4 - delta);
        if (maskP1) p[-2 * offset] = (Px)clip3(0, maxVal, m2 + clip3(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1)));
        if (maskQ1) p[offset] = (Px)clip3(0, maxVal, m5 + clip3(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1)));
    }
}

// Deblock::edgeFilterChroma (deblock.cpp:417-497) + pelFilterChroma (loopfilter.cpp:160-180) for one chroma plane of a 4:2:0 picture:
// a thread per 4-line chroma unit of an edge on the 8-sample chroma grid (luma multiples of 16); only Bs 2 units are filtered.
struct DbChromaArgs
{
    uint8_t* plane[2]; long strideB;           // Cb, Cr: one launch per direction filters both (grid.y = plane)
    int width, height, depth;                  // LUMA size
    const uint8_t* bs; const int8_t* qpMap;
    int qp, qpOffset[2], tcOffset;
};

__constant__ uint8_t kDbChromaScale[70] = {
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 33, 33, 34, 34, 35,
    35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51 };

template <typename Px, int DIR>
__global__ void __launch_bounds__(256) deblock_chroma_kernel(DbChromaArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int w8 = a.width >> 3, w4 = a.width >> 2;
    const int nEdges = (DIR == 0 ? a.width : a.height) >> 4, nUnits = (DIR == 0 ? a.height : a.width) >> 3;
    if (i >= nEdges * nUnits) return;
    const int e = i / nUnits, cu = i - e * nUnits;
    if (e == 0) return;                                     // the picture border is not an edge
    const int bs = DIR == 0 ? a.bs[(2 * cu) * w8 + 2 * e] : a.bs[(size_t)(2 * e) * w4 + 2 * cu];
    if (bs <= 1) return;
    int qp = a.qp;
    if (a.qpMap)
        qp = DIR == 0 ? (a.qpMap[cu * w8 + 2 * e - 1] + a.qpMap[cu * w8 + 2 * e] + 1) >> 1
                      : (a.qpMap[(2 * e - 1) * w8 + cu] + a.qpMap[(2 * e) * w8 + cu] + 1) >> 1;
    qp += a.qpOffset[blockIdx.y];
    if (qp >= 30) qp = kDbChromaScale[qp];
    const int tc = (int)kDbTc[clip3(0, 53, qp + 2 + a.tcOffset)] << (a.depth - 8);
    const int maxVal = (1 << a.depth) - 1;
    const long st = a.strideB / (long)sizeof(Px);
    Px* src = reinterpret_cast<Px*>(a.plane[blockIdx.y]) + (DIR == 0 ? (long)(4 * cu) * st + 8 * e : (long)(8 * e) * st + 4 * cu);
    const long offset = DIR == 0 ? 1 : st, srcStep = DIR == 0 ? st : 1;
#pragma unroll
    for (int l = 0; l < 4; l++)
    {
        Px* p = src + l * srcStep;
        const int m2 = p[-2 * offset], m3 = p[-offset], m4 = p[0], m5 = p[offset];
        const int delta = clip3(-tc, tc, (((m4 - m3) * 4) + m2 - m5 + 4) >> 3);
        p[-offset] = (Px)clip3(0, maxVal, m3 + delta);
        p[0] = (Px)clip3(0, maxVal, m4 - delta);
    }
}

} // namespace x265hip

using namespace x265hip;

extern "C" int x265hip_deblock_bs_inter(const x265hip_deblock_bs_params* p, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !p->mv || !p->num_sig || !p->bs_ver || !p->bs_hor) { set_error("deblock_bs_inter: NULL operand"); return X265HIP_EINVAL; }
    if ((p->width & 63) || (p->height & 63) || p->width <= 0 || p->height <= 0) { set_error("deblock_bs_inter: width/height must be multiples of 64"); return X265HIP_EINVAL; }
    if (p->level < 0 || p->level > 3) { set_error("deblock_bs_inter: level %d", p->level); return X265HIP_EINVAL; }
    BsArgs a;
    a.width = p->width; a.height = p->height; a.level = p->level;
    a.mv = (const int2*)p->mv; a.numSig = p->num_sig; a.bsVer = p->bs_ver; a.bsHor = p->bs_hor;
    const int nv = (p->height >> 2) * (p->width >> 3), nh = (p->height >> 3) * (p->width >> 2);
    const int n = nv > nh ? nv : nh;
    a.intra = p->intra;
    a.sliceB = p->slice_b; a.mv1 = (const int2*)p->mv1; a.ref0 = p->ref0; a.ref1 = p->ref1;
    if (p->slice_b && p->ref1 && !p->mv1) { set_error("deblock_bs_inter: ref1 without mv1"); return X265HIP_EINVAL; }
    hipLaunchKernelGGL(deblock_bs_inter_kernel, dim3((n + 255) / 256, 2), dim3(256), 0, (hipStream_t)stream, a);
    X265HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int x265hip_deblock_luma(const x265hip_deblock_params* p, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !p->rec || !p->bs_ver || !p->bs_hor) { set_error("deblock_luma: NULL operand"); return X265HIP_EINVAL; }
    if ((p->width & 7) || (p->height & 7) || p->width <= 0 || p->height <= 0) { set_error("deblock_luma: width/height must be multiples of 8"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("deblock_luma: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->beta_offset_div2 < -6 || p->beta_offset_div2 > 6 || p->tc_offset_div2 < -6 || p->tc_offset_div2 > 6)
    { set_error("deblock_luma: beta / tc offsets out of [-6, 6]"); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    DbArgs a;
    a.rec = (uint8_t*)p->rec; a.strideB = (long)p->stride * bpp;
    a.width = p->width; a.height = p->height; a.depth = p->depth;
    a.qpMap = p->qp_map; a.qp = p->qp; a.betaOffset = p->beta_offset_div2 * 2; a.tcOffset = p->tc_offset_div2 * 2;
    hipStream_t s = (hipStream_t)stream;
    const int nv = (p->height >> 2) * (p->width >> 3), nh = (p->height >> 3) * (p->width >> 2);
    a.bs = p->bs_ver;
    if (bpp == 1) hipLaunchKernelGGL((deblock_luma_kernel<uint8_t, 0>), dim3((nv + 255) / 256), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((deblock_luma_kernel<uint16_t, 0>), dim3((nv + 255) / 256), dim3(256), 0, s, a);
    a.bs = p->bs_hor;
    if (bpp == 1) hipLaunchKernelGGL((deblock_luma_kernel<uint8_t, 1>), dim3((nh + 255) / 256), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((deblock_luma_kernel<uint16_t, 1>), dim3((nh + 255) / 256), dim3(256), 0, s, a);
    X265HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int x265hip_deblock_chroma(const x265hip_deblock_chroma_params* p, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !p->cb || !p->cr || !p->bs_ver || !p->bs_hor) { set_error("deblock_chroma: NULL operand"); return X265HIP_EINVAL; }
    if ((p->width & 15) || (p->height & 15) || p->width <= 0 || p->height <= 0) { set_error("deblock_chroma: width/height must be multiples of 16"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("deblock_chroma: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->tc_offset_div2 < -6 || p->tc_offset_div2 > 6 || p->cb_qp_offset < -12 || p->cb_qp_offset > 12 || p->cr_qp_offset < -12 || p->cr_qp_offset > 12)
    { set_error("deblock_chroma: tc / chroma QP offsets out of range"); return X265HIP_EINVAL; }
    if (p->qp < 0 || p->qp > 51) { set_error("deblock_chroma: qp %d out of [0, 51]", p->qp); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    hipStream_t s = (hipStream_t)stream;
    DbChromaArgs a;
    a.strideB = (long)p->stride * bpp;
    a.width = p->width; a.height = p->height; a.depth = p->depth;
    a.qpMap = p->qp_map; a.qp = p->qp; a.tcOffset = p->tc_offset_div2 * 2;
    const int nv = (p->width >> 4) * (p->height >> 3), nh = (p->height >> 4) * (p->width >> 3);
    a.plane[0] = (uint8_t*)p->cb; a.plane[1] = (uint8_t*)p->cr;
    a.qpOffset[0] = p->cb_qp_offset; a.qpOffset[1] = p->cr_qp_offset;
    for (int dir = 0; dir < 2; dir++)
    {
        a.bs = dir ? p->bs_hor : p->bs_ver;
        const dim3 grid(((dir ? nh : nv) + 255) / 256, 2);
        if (bpp == 1 && dir == 0) hipLaunchKernelGGL((deblock_chroma_kernel<uint8_t, 0>), grid, dim3(256), 0, s, a);
        else if (bpp == 1) hipLaunchKernelGGL((deblock_chroma_kernel<uint8_t, 1>), grid, dim3(256), 0, s, a);
        else if (dir == 0) hipLaunchKernelGGL((deblock_chroma_kernel<uint16_t, 0>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((deblock_chroma_kernel<uint16_t, 1>), grid, dim3(256), 0, s, a);
    }
    X265HIP_TRY(hipGetLastError());
    return 0;
}
