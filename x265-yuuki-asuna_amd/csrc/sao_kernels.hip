// sao_kernels.hip - sample adaptive offset of the device-resident deblocked luma picture (SAO half):
// the two pixel passes.  The rate-distortion choice of the parameters between them (sao.cpp:1225-1605, entropy-coder bit counts
// over ~100 numbers per CTU) stays with the host.
//
// Reference semantics: SAO::calcSaoStatsCTU (encoder/sao.cpp:735-917; saoCuStatsBO / E0..E3, common/loopfilter.cpp) - per CTU,
// type and class the number of samples and the sum of (source - deblocked) - and SAO::generateLumaOffsets / applyPixelOffsets
// (sao.cpp:572-630, 274-570), which filter in place against saved copies of the not yet offset neighbours, i.e. out of place.
// Edge class of a sample c with neighbours a, b along the direction: s_eoTable[sign(c - a) + sign(c - b) + 2], table { 1, 2, 0,
// 3, 4 } (sao.cpp:67); band = c >> (depth - 5).
//
// Statistics: one workgroup per CTU; the deblocked 66x66 neighbourhood is staged in LDS once, a thread walks 16 samples of a
// row with a sliding 3x3 window and keeps the 4 x 5 edge accumulators packed (count << 20 | biased sum) in registers; the 32
// bands go through per-wavefront LDS histograms.  HBM traffic is the two planes read once: 2 samples per pixel.
#include "common.h"

namespace x265hip {

struct SaoStatsArgs
{
    const uint8_t* fenc; long fencStrideB;
    const uint8_t* rec; long recStrideB;
    int width, height, depth, ctusW;
    int ctuW, ctuH, planeOffset;      // the CTU's footprint in this plane (64x64 luma, 32x32 4:2:0 chroma) and the reference's plane_offset
    int nctu;
    int32_t* count; int32_t* offsetOrg;
};

__device__ __forceinline__ int sao_sign(int x) { return max(-1, min(1, x)); }      // one v_med3_i32

// Up to three planes per launch (grid.y = plane): the SAO passes of Y, Cb and Cr are independent, latency-bound launches of one round of
// workgroups each - three of them cost three times the latency, one launch with three times the workgroups costs it once.
struct SaoStatsArgs3 { SaoStatsArgs p[3]; };

template <typename Px>
__global__ void __launch_bounds__(256) sao_stats_kernel(SaoStatsArgs3 aa)
{
    const SaoStatsArgs& a = aa.p[blockIdx.y];
    if ((int)blockIdx.x >= a.nctu) return;
    constexpr int BPP = sizeof(Px);
    constexpr int LW = 68;                                   // LDS row pitch of the 66-wide neighbourhood
    __shared__ uint16_t sRec[66 * LW];
    __shared__ int sBo[4][2][32];
    __shared__ int sEo[2][20];
    const int tid = threadIdx.x;
    const int addr = xcd_swizzle(blockIdx.x, a.nctu);          // neighbouring CTUs (shared halo lines) on one XCD's L2
    const int lpelx = (addr % a.ctusW) * a.ctuW, tpely = (addr / a.ctusW) * a.ctuH;
    const int rpelx = min(lpelx + a.ctuW, a.width), bpely = min(tpely + a.ctuH, a.height);
    const int ctuW = rpelx - lpelx, ctuH = bpely - tpely;
    const bool atRight = rpelx == a.width, atBottom = bpely == a.height;
    // the reference's sub-rectangles (sao.cpp:806-915): the right 5 columns / bottom 4 rows (3 / 2 for chroma) wait for the
    // neighbour's deblocking
    const int skipR = 5 - a.planeOffset, skipB = 4 - a.planeOffset;
    const int boEndX = atRight ? ctuW : ctuW - skipR, boEndY = atBottom ? ctuH : ctuH - skipB;
    const int startX = !lpelx, endX0 = atRight ? ctuW - 1 : ctuW - skipR;
    const int startY = !tpely, endY1 = atBottom ? ctuH - 1 : ctuH - skipB;
    const int e0EndY = ctuH - skipB, e1EndX = boEndX;
    for (int i = tid; i < 4 * 2 * 32; i += 256) (&sBo[0][0][0])[i] = 0;
    if (tid < 40) (&sEo[0][0])[tid] = 0;
    // stage rows -1..ctuH, columns -1..ctuW of the deblocked picture (the planes are padded, so the border reads are legal)
    const Px* rec = reinterpret_cast<const Px*>(a.rec) + lpelx + (long)tpely * (a.recStrideB / BPP);
    const long rst = a.recStrideB / BPP;
    const int lane = tid & 63, wave = tid >> 6;
    {
        // packed loads: a thread fetches 4 samples (one dword, two for 16-bit pixels) of a staged row at a time
        const int qpr = (a.ctuW + 2 + 3) >> 2, nq = qpr * (a.ctuH + 2);
        for (int i = tid; i < nq; i += 256)
        {
            const int r = i / qpr, c = (i - r * qpr) * 4;
            const uint8_t* sp = reinterpret_cast<const uint8_t*>(rec + (long)(r - 1) * rst + (c - 1));
            uint32_t v[4];
            if (BPP == 1) { const uint32_t w = ld_u32(sp); v[0] = w & 0xff; v[1] = (w >> 8) & 0xff; v[2] = (w >> 16) & 0xff; v[3] = w >> 24; }
            else { const uint32_t w0 = ld_u32(sp), w1 = ld_u32(sp + 4); v[0] = w0 & 0xffff; v[1] = w0 >> 16; v[2] = w1 & 0xffff; v[3] = w1 >> 16; }
            uint32_t* d = reinterpret_cast<uint32_t*>(&sRec[r * LW + c]);        // LW and c are multiples of 4: dword aligned
            d[0] = v[0] | (v[1] << 16);
            d[1] = v[2] | (v[3] << 16);
        }
    }
    __syncthreads();
    // a thread owns 16 consecutive samples of one row (four threads per row)
    const int y = tid >> 2, x0 = (tid & 3) * 16;
    const Px* fe = reinterpret_cast<const Px*>(a.fenc) + lpelx + (long)(tpely + y) * (a.fencStrideB / BPP);
    const int bias = 1 << a.depth, boShift = a.depth - 5;
    uint32_t acc[4][5];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int k = 0; k < 5; k++) acc[t][k] = 0;
    if (y < ctuH)
    {
        // sliding window along the row: w[r][0..2] = columns x-1, x, x+1 of rows y-1, y, y+1 (LDS coordinates are shifted by one)
        int w[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++) { w[r][1] = sRec[(y + r) * LW + x0]; w[r][2] = sRec[(y + r) * LW + x0 + 1]; }
        const bool yE0 = y < e0EndY, yE1 = y >= startY && y < endY1, yBo = y < boEndY;
        // the thread's 16 source samples, fetched as dwords up front (a partial CTU reads into the padded margin)
        int fev[16];
        {
            const uint8_t* fp = reinterpret_cast<const uint8_t*>(fe + x0);
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                if (BPP == 1) { const uint32_t w4 = ld_u32(fp + 4 * k); fev[4 * k] = w4 & 0xff; fev[4 * k + 1] = (w4 >> 8) & 0xff; fev[4 * k + 2] = (w4 >> 16) & 0xff; fev[4 * k + 3] = w4 >> 24; }
                else { const uint32_t w0 = ld_u32(fp + 8 * k), w1 = ld_u32(fp + 8 * k + 4); fev[4 * k] = w0 & 0xffff; fev[4 * k + 1] = w0 >> 16; fev[4 * k + 2] = w1 & 0xffff; fev[4 * k + 3] = w1 >> 16; }
            }
        }
        // s_eoTable = { 1, 2, 0, 3, 4 }: class of edgeType e
        auto cls = [](int e) { return (0x43021 >> (4 * e)) & 7; };                // nibble e of 0x43021
        int runBand = 0, runCnt = 0, runSum = 0;
#pragma unroll
        for (int i = 0; i < 16; i++)
        {
            const int x = x0 + i;
#pragma unroll
            for (int r = 0; r < 3; r++) { w[r][0] = w[r][1]; w[r][1] = w[r][2]; w[r][2] = sRec[(y + r) * LW + x + 2]; }
            if (x >= ctuW) break;
            const int c = w[1][1];
            const int d = fev[i] - c;
            const uint32_t unit = (1u << 20) | (uint32_t)(d + bias);
            const bool xE = x >= startX && x < endX0;
            const int sc_l = sao_sign(c - w[1][0]), sc_r = sao_sign(c - w[1][2]);
            const int sc_u = sao_sign(c - w[0][1]), sc_d = sao_sign(c - w[2][1]);
            const int sc_ul = sao_sign(c - w[0][0]), sc_dr = sao_sign(c - w[2][2]);
            const int sc_ur = sao_sign(c - w[0][2]), sc_dl = sao_sign(c - w[2][0]);
            const int k0 = (xE && yE0) ? cls(sc_l + sc_r + 2) : 7;
            const int k1 = (x < e1EndX && yE1) ? cls(sc_u + sc_d + 2) : 7;
            const int k2 = (xE && yE1) ? cls(sc_ul + sc_dr + 2) : 7;
            const int k3 = (xE && yE1) ? cls(sc_ur + sc_dl + 2) : 7;
#pragma unroll
            for (int k = 0; k < 5; k++)
            {
                acc[0][k] += k0 == k ? unit : 0u;
                acc[1][k] += k1 == k ? unit : 0u;
                acc[2][k] += k2 == k ? unit : 0u;
                acc[3][k] += k3 == k ? unit : 0u;
            }
            if (yBo && x < boEndX)
            {
                // neighbouring samples mostly share a band: accumulate the run in registers, touch the LDS histogram on a change
                const int band = c >> boShift;
                if (band != runBand)
                {
                    if (runCnt) { atomicAdd(&sBo[wave][0][runBand], runCnt); atomicAdd(&sBo[wave][1][runBand], runSum); }
                    runBand = band; runCnt = 0; runSum = 0;
                }
                runCnt++; runSum += d;
            }
        }
        if (runCnt) { atomicAdd(&sBo[wave][0][runBand], runCnt); atomicAdd(&sBo[wave][1][runBand], runSum); }
    }
    // unpack, reduce over the wavefront, then over the workgroup
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int k = 0; k < 5; k++)
        {
            const int cnt = (int)(acc[t][k] >> 20);
            const int sum = (int)(acc[t][k] & 0xfffffu) - cnt * bias;
            // DPP row sums + four readlanes: no LDS round trips (the 40 butterfly reductions through ds_bpermute cost 8 us of the 44 us launch;
            // halving the per-thread walk instead - 512 threads x 8 samples - changes nothing: the rest is the latency of one round of workgroups)
            const int cw = wave_sum_of_rows(row_sum(cnt)), sw = wave_sum_of_rows(row_sum(sum));
            if ((tid & 63) == 0 && cw) { atomicAdd(&sEo[0][t * 5 + k], cw); atomicAdd(&sEo[1][t * 5 + k], sw); }
        }
    __syncthreads();
    int32_t* cnt = a.count + (size_t)addr * 160;
    int32_t* org = a.offsetOrg + (size_t)addr * 160;
    if (tid < 160)
    {
        const int t = tid >> 5, k = tid & 31;
        int c = 0, s = 0;
        if (t < 4) { if (k < 5) { c = sEo[0][t * 5 + k]; s = sEo[1][t * 5 + k]; } }
        else
        {
#pragma unroll
            for (int w4 = 0; w4 < 4; w4++) { c += sBo[w4][0][k]; s += sBo[w4][1][k]; }
        }
        cnt[tid] = c; org[tid] = s;
    }
}

struct SaoApplyArgs
{
    const uint8_t* src; long srcStrideB;
    uint8_t* dst; long dstStrideB;
    int width, height, depth, ctusW, ctuW, ctuH;
    const int32_t* params;
    int nctu;
};
struct SaoApplyArgs3 { SaoApplyArgs p[3]; };

template <typename Px>
__global__ void __launch_bounds__(256) sao_apply_kernel(SaoApplyArgs3 aa)
{
    const SaoApplyArgs& a = aa.p[blockIdx.y];
    if ((int)blockIdx.x >= a.nctu) return;
    constexpr int BPP = sizeof(Px);
    const int tid = threadIdx.x, addr = xcd_swizzle(blockIdx.x, a.nctu);
    const int lpelx = (addr % a.ctusW) * a.ctuW, tpely = (addr / a.ctusW) * a.ctuH;
    const int32_t* p = a.params + (size_t)addr * 7;
    const int typeIdx = p[0], bandPos = p[1];
    const int o0 = (int8_t)p[2], o1 = (int8_t)p[3], o2 = (int8_t)p[4], o3 = (int8_t)p[5];
    const int maxVal = (1 << a.depth) - 1, boShift = a.depth - 5;
    const long sst = a.srcStrideB / BPP, dst_st = a.dstStrideB / BPP;
    const Px* src = reinterpret_cast<const Px*>(a.src);
    Px* dst = reinterpret_cast<Px*>(a.dst);
    // round 6 (closing): a lane owns FOUR consecutive samples of a row - one dword (two for 16-bit samples) per load / store where the first version moved
    // one sample per lane and instruction (three loads + one store per sample for an edge class); the neighbours of the four samples come as two more
    // (unaligned) loads of four.  The planes are padded pictures: the sample beyond a picture edge that such a load touches exists, and is never used.
    const int qpr = (a.ctuW + 3) >> 2, rowsPerPass = 256 / qpr;                 // quads per CTU row: 16 (64-wide CTUs) / 8 (32-wide chroma footprints)
    const int xl = (tid % qpr) * 4, x = lpelx + xl;
    if (x >= a.width) return;
    // neighbour step of the edge classes: EO_0 horizontal, EO_1 vertical, EO_2 135 degrees, EO_3 45 degrees
    const int dx = typeIdx == 1 ? 0 : (typeIdx == 3 ? -1 : 1), dy = typeIdx == 0 ? 0 : 1;
    auto load4 = [&](const Px* q, int (&v)[4])
    {
        const uint8_t* b = reinterpret_cast<const uint8_t*>(q);
        if (BPP == 1) { const uint32_t w = ld_u32(b); v[0] = w & 0xff; v[1] = (w >> 8) & 0xff; v[2] = (w >> 16) & 0xff; v[3] = w >> 24; }
        else { const uint32_t w0 = ld_u32(b), w1 = ld_u32(b + 4); v[0] = w0 & 0xffff; v[1] = w0 >> 16; v[2] = w1 & 0xffff; v[3] = w1 >> 16; }
    };
    const bool whole = x + 3 < a.width;
    for (int yl = tid / qpr; yl < a.ctuH; yl += rowsPerPass)
    {
        const int y = tpely + yl;
        if (y >= a.height) break;
        const Px* c = src + x + (long)y * sst;
        int v[4];
        load4(c, v);
        if (typeIdx == 4)
        {
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                const int k = ((v[j] >> boShift) - bandPos) & 31;                  // offset[i] sits on band (bandPos + i) mod 32
                const int off = k == 0 ? o0 : (k == 1 ? o1 : (k == 2 ? o2 : (k == 3 ? o3 : 0)));
                v[j] = clip3(0, maxVal, v[j] + off);
            }
        }
        else if (typeIdx >= 0 && (!dy || (y > 0 && y < a.height - 1)))
        {
            int na[4], nb[4];
            load4(c - dx - dy * sst, na);
            load4(c + dx + dy * sst, nb);
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                if (dx && !(x + j > 0 && x + j < a.width - 1)) continue;
                const int e = sao_sign(v[j] - na[j]) + sao_sign(v[j] - nb[j]) + 2;
                // offsetEo[e] = offset[s_eoTable[e]] with offset[] = { 0, o0, o1, o2, o3 }, s_eoTable = { 1, 2, 0, 3, 4 }
                const int off = e == 0 ? o0 : (e == 1 ? o1 : (e == 2 ? 0 : (e == 3 ? o2 : o3)));
                v[j] = clip3(0, maxVal, v[j] + off);
            }
        }
        Px* d = dst + x + (long)y * dst_st;
        if (whole)
        {
            uint8_t* b = reinterpret_cast<uint8_t*>(d);
            if (BPP == 1) *reinterpret_cast<u32_unaligned*>(b) = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
            else
            {
                reinterpret_cast<u32_unaligned*>(b)[0] = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
                reinterpret_cast<u32_unaligned*>(b)[1] = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
            }
        }
        else
            for (int j = 0; j < 4 && x + j < a.width; j++) d[j] = (Px)v[j];
    }
}


// ------------------------------------------------------------------------------------------------
// SAO parameters without leaving the device: SAO::saoStatsInitialOffset (sao.cpp:1378-1433: roundIBDI of offsetOrg / count,
// clip to +-(OFFSET_THRESH - 1), sign constraint of the edge classes) + a distortion-only choice of the type by estSaoDist
// (sao.cpp:56-59) - the documented stand-in for the entropy-coder-driven rdoSaoUnitCu that lets the closed-loop pipeline hand the
// next picture a reference that went through SAO.
struct SaoDecideArgs { const int32_t* count; const int32_t* offsetOrg; int nctu, depth; int32_t* initOffset; int32_t* params; };
struct SaoDecideArgs3 { SaoDecideArgs p[3]; };

// one wavefront per CTU: lanes 0..15 take the (edge type, class) pairs, lanes 32..63 the 32 bands - the divisions of the initial offsets
// run side by side - and lane 0 makes the small serial choice from LDS
__global__ void __launch_bounds__(64) sao_decide_kernel(SaoDecideArgs3 aa)
{
    const SaoDecideArgs& a = aa.p[blockIdx.y];
    if ((int)blockIdx.x >= a.nctu) return;
    __shared__ int sOff[5][32];
    __shared__ long long sDist[5][32];
    const int ctu = blockIdx.x, lane = threadIdx.x;
    const int32_t* cnt = a.count + (size_t)ctu * 160;
    const int32_t* org = a.offsetOrg + (size_t)ctu * 160;
    const int thresh = 1 << (a.depth - 5 < 5 ? a.depth - 5 : 5);
    int t = -1, c = 0;
    if (lane < 16) { t = lane >> 2; c = 1 + (lane & 3); }
    else if (lane >= 32) { t = 4; c = lane - 32; }
    for (int i = lane; i < 160; i += 64) { sOff[i >> 5][i & 31] = 0; sDist[i >> 5][i & 31] = 0; }
    __syncthreads();
    if (t >= 0)
    {
        const int n = cnt[t * 32 + c], e = org[t * 32 + c];
        int o = 0;
        if (n)
        {
            o = e >= 0 ? (e * 2 + n) / (n * 2) : -((-e * 2 + n) / (n * 2));          // roundIBDI (sao.cpp:34-37)
            o = clip3(-thresh + 1, thresh - 1, o);
            if (t < 4) o = c < 3 ? max(o, 0) : min(o, 0);
        }
        sOff[t][c] = o;
        sDist[t][c] = ((long long)n * o - (long long)e * 2) * o;                       // estSaoDist (sao.cpp:56-59)
    }
    __syncthreads();
    if (a.initOffset)
        for (int i = lane; i < 160; i += 64) a.initOffset[(size_t)ctu * 160 + i] = sOff[i >> 5][i & 31];
    if (lane == 0)
    {
        long long best = 0;
        int type = -1, band = 0;
        for (int ty = 0; ty < 4; ty++)
        {
            const long long d = sDist[ty][1] + sDist[ty][2] + sDist[ty][3] + sDist[ty][4];
            if (d < best) { best = d; type = ty; }
        }
        long long bo = 0; int start = -1;
        for (int sidx = 0; sidx <= 28; sidx++)
        {
            const long long d = sDist[4][sidx] + sDist[4][sidx + 1] + sDist[4][sidx + 2] + sDist[4][sidx + 3];
            if (start < 0 || d < bo) { bo = d; start = sidx; }
        }
        if (bo < best) { type = 4; band = start; }
        int32_t* p = a.params + (size_t)ctu * 7;
        p[0] = type; p[1] = band; p[6] = 0;
        for (int i = 0; i < 4; i++) p[2 + i] = type < 0 ? 0 : (type == 4 ? sOff[4][band + i] : sOff[type][1 + i]);
    }
}

} // namespace x265hip

using namespace x265hip;

static int fill_stats(const x265hip_sao_stats_params* p, SaoStatsArgs& a)
{
    if (!p || !p->fenc || !p->rec || !p->count || !p->offset_org) { set_error("sao_stats: NULL operand"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("sao_stats: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->width <= 0 || p->height <= 0) { set_error("sao_stats: empty picture"); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    a.fenc = (const uint8_t*)p->fenc; a.fencStrideB = (long)p->fenc_stride * bpp;
    a.rec = (const uint8_t*)p->rec; a.recStrideB = (long)p->rec_stride * bpp;
    a.ctuW = p->ctu_width ? p->ctu_width : 64; a.ctuH = p->ctu_height ? p->ctu_height : 64; a.planeOffset = p->plane_offset;
    if (a.ctuW < 8 || a.ctuW > 64 || a.ctuH < 8 || a.ctuH > 64 || a.planeOffset < 0 || a.planeOffset > 2)
    { set_error("sao_stats: CTU footprint %d x %d / plane_offset %d", a.ctuW, a.ctuH, a.planeOffset); return X265HIP_EINVAL; }
    a.width = p->width; a.height = p->height; a.depth = p->depth; a.ctusW = (p->width + a.ctuW - 1) / a.ctuW;
    a.count = p->count; a.offsetOrg = p->offset_org;
    a.nctu = a.ctusW * ((p->height + a.ctuH - 1) / a.ctuH);
    return 0;
}

static int fill_apply(const x265hip_sao_apply_params* p, SaoApplyArgs& a)
{
    if (!p || !p->src || !p->dst || !p->ctu_params) { set_error("sao_apply: NULL operand"); return X265HIP_EINVAL; }
    if (p->src == p->dst) { set_error("sao_apply: the filter is out of place (a sample is classified against unfiltered neighbours)"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("sao_apply: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->width <= 0 || p->height <= 0) { set_error("sao_apply: empty picture"); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    a.src = (const uint8_t*)p->src; a.srcStrideB = (long)p->src_stride * bpp;
    a.dst = (uint8_t*)p->dst; a.dstStrideB = (long)p->dst_stride * bpp;
    a.ctuW = p->ctu_width ? p->ctu_width : 64; a.ctuH = p->ctu_height ? p->ctu_height : 64;
    if (a.ctuW < 8 || a.ctuW > 64 || a.ctuH < 8 || a.ctuH > 64) { set_error("sao_apply: CTU footprint %d x %d", a.ctuW, a.ctuH); return X265HIP_EINVAL; }
    a.width = p->width; a.height = p->height; a.depth = p->depth; a.ctusW = (p->width + a.ctuW - 1) / a.ctuW;
    a.params = p->ctu_params;
    a.nctu = a.ctusW * ((p->height + a.ctuH - 1) / a.ctuH);
    return 0;
}

static int launch_stats(const SaoStatsArgs3& aa, int nplanes, int depth, hipStream_t s)
{
    int nmax = 0;
    for (int i = 0; i < nplanes; i++) nmax = aa.p[i].nctu > nmax ? aa.p[i].nctu : nmax;
    if (depth == 8) hipLaunchKernelGGL(sao_stats_kernel<uint8_t>, dim3(nmax, nplanes), dim3(256), 0, s, aa);
    else hipLaunchKernelGGL(sao_stats_kernel<uint16_t>, dim3(nmax, nplanes), dim3(256), 0, s, aa);
    return check_hip(hipGetLastError(), "sao_stats launch");
}

static int launch_apply(const SaoApplyArgs3& aa, int nplanes, int depth, hipStream_t s)
{
    int nmax = 0;
    for (int i = 0; i < nplanes; i++) nmax = aa.p[i].nctu > nmax ? aa.p[i].nctu : nmax;
    if (depth == 8) hipLaunchKernelGGL(sao_apply_kernel<uint8_t>, dim3(nmax, nplanes), dim3(256), 0, s, aa);
    else hipLaunchKernelGGL(sao_apply_kernel<uint16_t>, dim3(nmax, nplanes), dim3(256), 0, s, aa);
    return check_hip(hipGetLastError(), "sao_apply launch");
}

extern "C" int x265hip_sao_stats(const x265hip_sao_stats_params* p, void* stream)
{
    SaoStatsArgs3 aa = {};
    int rc = fill_stats(p, aa.p[0]);
    if (rc) return rc;
    if ((rc = ensure_device())) return rc;
    return launch_stats(aa, 1, p->depth, (hipStream_t)stream);
}

extern "C" int x265hip_sao_apply(const x265hip_sao_apply_params* p, void* stream)
{
    SaoApplyArgs3 aa = {};
    int rc = fill_apply(p, aa.p[0]);
    if (rc) return rc;
    if ((rc = ensure_device())) return rc;
    return launch_apply(aa, 1, p->depth, (hipStream_t)stream);
}

/* the application of 1..3 planes (Y, Cb, Cr) as ONE launch: what follows x265hip_sao_rdo, whose decision needs all planes' statistics first */
extern "C" int x265hip_sao_apply_planes(int nplanes, const x265hip_sao_apply_params* apply, void* stream)
{
    if (nplanes < 1 || nplanes > 3 || !apply) { set_error("sao_apply_planes: %d planes", nplanes); return X265HIP_EINVAL; }
    SaoApplyArgs3 ap = {};
    for (int i = 0; i < nplanes; i++)
    {
        int rc = fill_apply(&apply[i], ap.p[i]);
        if (rc) return rc;
        if (apply[i].depth != apply[0].depth) { set_error("sao_apply_planes: planes of different bit depths"); return X265HIP_EINVAL; }
    }
    int rc = ensure_device();
    if (rc) return rc;
    return launch_apply(ap, nplanes, apply[0].depth, (hipStream_t)stream);
}

extern "C" int x265hip_sao_decide(int depth, const int32_t* count, const int32_t* offset_org, int nctu, int32_t* init_offset, int32_t* ctu_params, void* stream)
{
    if (!count || !offset_org || !ctu_params) { set_error("sao_decide: NULL operand"); return X265HIP_EINVAL; }
    if (depth != 8 && depth != 10 && depth != 12) { set_error("sao_decide: depth %d", depth); return X265HIP_EINVAL; }
    if (nctu <= 0) { set_error("sao_decide: nctu %d", nctu); return X265HIP_EINVAL; }
    int rc = ensure_device();
    if (rc) return rc;
    SaoDecideArgs3 aa = {};
    aa.p[0] = SaoDecideArgs{ count, offset_org, nctu, depth, init_offset, ctu_params };
    hipLaunchKernelGGL(sao_decide_kernel, dim3(nctu, 1), dim3(64), 0, (hipStream_t)stream, aa);
    return check_hip(hipGetLastError(), "sao_decide launch");
}

/* Y, Cb and Cr (or any 1..3 planes of one bit depth) through the three SAO steps with ONE launch per step instead of one per plane and
 * step: statistics -> (when `apply` is given) parameters on the device -> application.  stats[i] / apply[i] describe plane i exactly as
 * the single-plane entries take them; apply[i].ctu_params receives plane i's parameters. */
extern "C" int x265hip_sao_planes(int nplanes, const x265hip_sao_stats_params* stats, const x265hip_sao_apply_params* apply, void* stream)
{
    if (nplanes < 1 || nplanes > 3 || !stats) { set_error("sao_planes: %d planes", nplanes); return X265HIP_EINVAL; }
    SaoStatsArgs3 sa = {};
    SaoApplyArgs3 ap = {};
    SaoDecideArgs3 da = {};
    int nmax = 0;
    for (int i = 0; i < nplanes; i++)
    {
        int rc = fill_stats(&stats[i], sa.p[i]);
        if (rc) return rc;
        if (stats[i].depth != stats[0].depth) { set_error("sao_planes: planes of different bit depths"); return X265HIP_EINVAL; }
        if (apply)
        {
            if ((rc = fill_apply(&apply[i], ap.p[i]))) return rc;
            if (apply[i].depth != stats[0].depth || ap.p[i].nctu != sa.p[i].nctu) { set_error("sao_planes: statistics / application geometry of plane %d differ", i); return X265HIP_EINVAL; }
            da.p[i] = SaoDecideArgs{ stats[i].count, stats[i].offset_org, sa.p[i].nctu, stats[i].depth, nullptr, const_cast<int32_t*>(apply[i].ctu_params) };
        }
        nmax = sa.p[i].nctu > nmax ? sa.p[i].nctu : nmax;
    }
    int rc = ensure_device();
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if ((rc = launch_stats(sa, nplanes, stats[0].depth, s))) return rc;
    if (!apply) return 0;
    hipLaunchKernelGGL(sao_decide_kernel, dim3(nmax, nplanes), dim3(64), 0, s, da);
    if ((rc = check_hip(hipGetLastError(), "sao_decide launch"))) return rc;
    return launch_apply(ap, nplanes, stats[0].depth, s);
}
