// loopfilter_kernels.hip - SAO apply / statistics, deblocking edge filters, sign, SEA integral rows and
// the SEA ADS pre-filter on gfx950.
//
// Reference semantics: source/common/loopfilter.cpp calSign :39-43, processSaoCUE0 :45-63, E1 :65-78,
// E1_2Rows :80-97, E2 :99-109, E3 :111-123, B0 :125-138, pelFilterLumaStrong_c :140-159, pelFilterChroma_c
// :167-180; source/encoder/sao.cpp saoCuStatsBO_c :1762-1778, E0 :1780-1814, E1 :1816-1850, E2 :1852-1887,
// E3 :1889-1925 (edge classes folded through s_eoTable = {1,2,0,3,4}, sao.cpp:65-72; diff stride is the
// fixed MAX_CU_SIZE = 64); source/encoder/framefilter.cpp integral_init*h/v :39-140; source/common/pixel.cpp
// ads_x4/x2/x1 :121-165.
//
// The serial C loops carry sign buffers from pixel to pixel / row to row.  Every carried value is a
// pure function of ORIGINAL neighbouring pixels (e.g. the left sign of x is sign(rec[x] - rec[x-1])), so
// the kernels evaluate all samples in parallel, and reproduce the buffers' final contents - which the
// callers read back - in closed form.  Operand conventions (planes p0..p3, job off[]/arg[]):
//   SIGN         p0 dst(int8) p1 src1 p2 src2                      arg0 = endX
//   SAO_E0       p0 rec (stride)  p1 offsetEo(int8[5])  p2 signLeft(int8[2])          arg0 = width
//   SAO_E1[_2R]  p0 rec  p1 offsetEo  p2 upBuff1 (in/out)                              arg0 = width
//   SAO_E2       p0 rec  p1 offsetEo  p2 buff1 (in)  p3 bufft (out, written at x+1)    arg0 = width
//   SAO_E3       p0 rec  p1 offsetEo  p2 upBuff1 (in/out, written at x-1)              arg0 = startX arg1 = endX
//   SAO_B0       p0 rec  p1 offset(int8[32])                                           arg0 = ctuWidth arg1 = ctuHeight
//   STATS_*      p0 diff(int16, plane stride; the reference uses 64)  p1 rec (stride)  p2 upBuff1  p3 upBufft    arg0 = endX arg1 = endY
//                result[job*64 + 0..31] += stats, result[job*64 + 32..63] += count
//   DEBLOCK_LUMA_STRONG  p0 src   arg0 = srcStep arg1 = offset arg2 = tcP arg3 = tcQ
//   DEBLOCK_CHROMA       p0 src   arg0 = srcStep arg1 = offset arg2 = tc  off[2] = maskP off[3] = maskQ
//   INTEGRAL_H   p0 sum(uint32) p1 pix   arg0 = distance to the row above (stride) arg1 = N arg2 = count (stride - N)
//   INTEGRAL_V   p0 sum  arg0 = stride arg1 = N arg2 = count (stride)
//   ADS          p0 sums(uint32) p1 costMvX(uint16) p2 mvs(int16 out) p3 encDC(int32[4])
//                arg0 = delta arg1 = width arg2 = thresh arg3 = lx | (nsum << 16);  result[job] = count
#include "common.h"

namespace x265hip {

struct LfArgs { x265hip_plane p[4]; const x265hip_job* jobs; uint32_t* result; int njobs, depth; };

__device__ __forceinline__ int sgn(int v) { return (v > 0) - (v < 0); }

__constant__ int kEoTable[5] = { 1, 2, 0, 3, 4 };

template <typename Px, int KIND>
__global__ void __launch_bounds__(256) lf_kernel(LfArgs a)
{
    const int tid = threadIdx.x, nth = blockDim.x;
    const int maxVal = (1 << a.depth) - 1;

    if (KIND == X265HIP_LF_DEBLOCK_LUMA_STRONG || KIND == X265HIP_LF_DEBLOCK_CHROMA)
    {
        // 4 lines per edge segment: thread = (job, line)
        const long gj = (long)blockIdx.x * (nth >> 2) + (tid >> 2);
        if (gj >= a.njobs) return;
        const x265hip_job jb = a.jobs[gj];
        const long step = jb.arg[0], off = jb.arg[1];
        Px* src = (Px*)a.p[0].base + jb.off[0] + (tid & 3) * step;
        if (KIND == X265HIP_LF_DEBLOCK_LUMA_STRONG)
        {
            const int tcP = jb.arg[2], tcQ = jb.arg[3];
            const int p3 = src[-off * 4], p2 = src[-off * 3], p1 = src[-off * 2], p0 = src[-off];
            const int q0 = src[0], q1 = src[off], q2 = src[off * 2], q3 = src[off * 3];
            src[-off * 3] = (Px)(clip3(-tcP, tcP, ((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3) - p2) + p2);
            src[-off * 2] = (Px)(clip3(-tcP, tcP, ((p2 + p1 + p0 + q0 + 2) >> 2) - p1) + p1);
            src[-off]     = (Px)(clip3(-tcP, tcP, ((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3) - p0) + p0);
            src[0]        = (Px)(clip3(-tcQ, tcQ, ((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3) - q0) + q0);
            src[off]      = (Px)(clip3(-tcQ, tcQ, ((p0 + q0 + q1 + q2 + 2) >> 2) - q1) + q1);
            src[off * 2]  = (Px)(clip3(-tcQ, tcQ, ((p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3) - q2) + q2);
        }
        else
        {
            const int tc = jb.arg[2], maskP = (int)jb.off[2], maskQ = (int)jb.off[3];
            const int p1 = src[-off * 2], p0 = src[-off], q0 = src[0], q1 = src[off];
            const int delta = clip3(-tc, tc, (((q0 - p0) * 4) + p1 - q1 + 4) >> 3);
            src[-off] = (Px)clip3(0, maxVal, p0 + (delta & maskP));
            src[0] = (Px)clip3(0, maxVal, q0 - (delta & maskQ));
        }
        return;
    }

    const x265hip_job jb = a.jobs[blockIdx.x];
    if (KIND == X265HIP_LF_SIGN)
    {
        int8_t* d = (int8_t*)a.p[0].base + jb.off[0];
        const Px* s1 = (const Px*)a.p[1].base + jb.off[1];
        const Px* s2 = (const Px*)a.p[2].base + jb.off[2];
        for (int x = tid; x < jb.arg[0]; x += nth) d[x] = (int8_t)sgn((int)s1[x] - (int)s2[x]);
        return;
    }
    if (KIND == X265HIP_LF_INTEGRAL_H)
    {
        uint32_t* sum = (uint32_t*)a.p[0].base + jb.off[0];
        const Px* pix = (const Px*)a.p[1].base + jb.off[1];
        const int above = jb.arg[0], n = jb.arg[1], count = jb.arg[2];   // count = stride - N in the reference loop
        for (int x = tid; x < count; x += nth)
        {
            uint32_t v = 0;
            for (int i = 0; i < n; i++) v += pix[x + i];
            sum[x] = v + sum[x - above];
        }
        return;
    }
    if (KIND == X265HIP_LF_INTEGRAL_V)
    {
        uint32_t* sum = (uint32_t*)a.p[0].base + jb.off[0];
        const int stride = jb.arg[0], n = jb.arg[1], count = jb.arg[2];
        for (int x = tid; x < count; x += nth) sum[x] = sum[x + (long)n * stride] - sum[x];
        return;
    }
    if (KIND == X265HIP_LF_ADS)
    {
        // one wavefront; ordered compaction with ballot + prefix popcount
        const uint32_t* sums = (const uint32_t*)a.p[0].base + jb.off[0];
        const uint16_t* cost = (const uint16_t*)a.p[1].base + jb.off[1];
        int16_t* mvs = (int16_t*)a.p[2].base + jb.off[2];
        const int* enc = (const int*)a.p[3].base + jb.off[3];
        const int delta = jb.arg[0], width = jb.arg[1], thresh = jb.arg[2];
        const int lx = jb.arg[3] & 0xffff, nsum = jb.arg[3] >> 16;
        int base = 0;
        for (int i0 = 0; i0 < width; i0 += 64)
        {
            const int i = i0 + tid;
            bool hit = false;
            if (i < width)
            {
                long v = labs((long)enc[0] - (long)sums[i]);
                if (nsum == 4)
                    v += labs((long)enc[1] - (long)sums[i + (lx >> 1)]) + labs((long)enc[2] - (long)sums[i + delta])
                       + labs((long)enc[3] - (long)sums[i + delta + (lx >> 1)]);
                else if (nsum == 2)
                    v += labs((long)enc[1] - (long)sums[i + delta]);
                hit = (int)v + (int)cost[i] < thresh;
            }
            const unsigned long long m = __ballot(hit);
            if (hit) mvs[base + __popcll(m & ((1ull << tid) - 1))] = (int16_t)i;
            base += __popcll(m);
        }
        if (tid == 0 && a.result) a.result[blockIdx.x] = (uint32_t)base;
        return;
    }

    // ------------------------------------------------------------------ SAO apply
    if (KIND == X265HIP_LF_SAO_E0)
    {
        Px* rec = (Px*)a.p[0].base + jb.off[0];
        const int8_t* offEo = (const int8_t*)a.p[1].base + jb.off[1];
        const int8_t* signLeft = (const int8_t*)a.p[2].base + jb.off[2];
        const int width = jb.arg[0];
        const long st = a.p[0].stride;
        // two rows, width <= 64 (one CTU): 128 threads read their three pixels, then all write
        const int y = tid >> 6, x = tid & 63;
        const bool on = x < width;
        Px* r = rec + y * st;
        int c = 0, cls = 0;
        if (on)
        {
            c = r[x];
            const int sr = sgn(c - (int)r[x + 1]);
            const int sl = x == 0 ? (int)signLeft[y] : sgn(c - (int)r[x - 1]);
            cls = sr + sl + 2;
        }
        __syncthreads();
        if (on) r[x] = (Px)clip3(0, maxVal, c + offEo[cls]);
        return;
    }
    if (KIND == X265HIP_LF_SAO_E1 || KIND == X265HIP_LF_SAO_E1_2ROWS)
    {
        Px* rec = (Px*)a.p[0].base + jb.off[0];
        const int8_t* offEo = (const int8_t*)a.p[1].base + jb.off[1];
        int8_t* up = (int8_t*)a.p[2].base + jb.off[2];
        const long st = a.p[0].stride;
        const int rows = KIND == X265HIP_LF_SAO_E1 ? 1 : 2;
        for (int x = tid; x < jb.arg[0]; x += nth)
        {
            int u = up[x];
            for (int y = 0; y < rows; y++)
            {
                const int c = rec[y * st + x];
                const int sd = sgn(c - (int)rec[(y + 1) * st + x]);
                rec[y * st + x] = (Px)clip3(0, maxVal, c + offEo[sd + u + 2]);
                u = -sd;
            }
            up[x] = (int8_t)u;
        }
        return;
    }
    if (KIND == X265HIP_LF_SAO_E2)
    {
        Px* rec = (Px*)a.p[0].base + jb.off[0];
        const int8_t* offEo = (const int8_t*)a.p[1].base + jb.off[1];
        const int8_t* b1 = (const int8_t*)a.p[2].base + jb.off[2];
        int8_t* bt = (int8_t*)a.p[3].base + jb.off[3];
        const long st = a.p[0].stride;
        for (int x = tid; x < jb.arg[0]; x += nth)
        {
            const int c = rec[x];
            const int sd = sgn(c - (int)rec[x + st + 1]);
            bt[x + 1] = (int8_t)(-sd);
            rec[x] = (Px)clip3(0, maxVal, c + offEo[sd + b1[x] + 2]);
        }
        return;
    }
    if (KIND == X265HIP_LF_SAO_E3)
    {
        Px* rec = (Px*)a.p[0].base + jb.off[0];
        const int8_t* offEo = (const int8_t*)a.p[1].base + jb.off[1];
        int8_t* up = (int8_t*)a.p[2].base + jb.off[2];
        const long st = a.p[0].stride;
        const int startX = jb.arg[0], endX = jb.arg[1];
        // iteration x reads up[x] and writes up[x-1]: read everything first (endX - startX <= 64 per pass)
        for (int x0 = startX + 1; x0 < endX; x0 += nth)
        {
            const int x = x0 + tid;
            int c = 0, sd = 0, u = 0;
            const bool on = x < endX;
            if (on) { c = rec[x]; sd = sgn(c - (int)rec[x + st]); u = up[x]; }
            __syncthreads();
            if (on) { up[x - 1] = (int8_t)(-sd); rec[x] = (Px)clip3(0, maxVal, c + offEo[(int8_t)(sd + u + 2)]); }
            __syncthreads();
        }
        return;
    }
    if (KIND == X265HIP_LF_SAO_B0)
    {
        Px* rec = (Px*)a.p[0].base + jb.off[0];
        const int8_t* offs = (const int8_t*)a.p[1].base + jb.off[1];
        const int w = jb.arg[0], h = jb.arg[1], sh = a.depth - 5;
        const long st = a.p[0].stride;
        __shared__ int sOff[32];                       // the 32 band offsets: one global read per workgroup instead of one per sample
        if (tid < 32) sOff[tid] = offs[tid];
        __syncthreads();
        if ((w & 3) == 0)
        {
            // 4 samples per thread and step through packed dwords
            const int qpr = w >> 2;
            for (int q = tid; q < qpr * h; q += nth)
            {
                const int y = q / qpr, x = (q - y * qpr) * 4;
                uint8_t* p = reinterpret_cast<uint8_t*>(rec + y * st + x);
                if (sizeof(Px) == 1)
                {
                    const uint32_t v = ld_u32(p);
                    uint32_t r = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                    {
                        const int c = (v >> (8 * k)) & 0xff;
                        r |= (uint32_t)clip3(0, maxVal, c + sOff[c >> sh]) << (8 * k);
                    }
                    *reinterpret_cast<u32_unaligned*>(p) = r;
                }
                else
                {
                    const uint32_t v0 = ld_u32(p), v1 = ld_u32(p + 4);
                    const int c0 = v0 & 0xffff, c1 = v0 >> 16, c2 = v1 & 0xffff, c3 = v1 >> 16;
                    const uint32_t r0 = (uint32_t)clip3(0, maxVal, c0 + sOff[c0 >> sh]) | ((uint32_t)clip3(0, maxVal, c1 + sOff[c1 >> sh]) << 16);
                    const uint32_t r1 = (uint32_t)clip3(0, maxVal, c2 + sOff[c2 >> sh]) | ((uint32_t)clip3(0, maxVal, c3 + sOff[c3 >> sh]) << 16);
                    *reinterpret_cast<u32_unaligned*>(p) = r0;
                    *reinterpret_cast<u32_unaligned*>(p + 4) = r1;
                }
            }
            return;
        }
        for (int i = tid; i < w * h; i += nth)
        {
            const int y = i / w, x = i - y * w;
            const int c = rec[y * st + x];
            rec[y * st + x] = (Px)clip3(0, maxVal, c + sOff[c >> sh]);
        }
        return;
    }

    // ------------------------------------------------------------------ SAO statistics
    {
        __shared__ int sStat[32], sCnt[32];
        const int16_t* diff = (const int16_t*)a.p[0].base + jb.off[0];
        const Px* rec = (const Px*)a.p[1].base + jb.off[1];
        int8_t* up1 = (int8_t*)a.p[2].base + jb.off[2];
        int8_t* upt = (int8_t*)a.p[3].base + jb.off[3];
        const long st = a.p[1].stride;
        const int endX = jb.arg[0], endY = jb.arg[1];
        if (tid < 32) { sStat[tid] = 0; sCnt[tid] = 0; }
        __syncthreads();
        for (int i = tid; i < endX * endY; i += nth)
        {
            const int y = i / endX, x = i - y * endX;
            const Px* r = rec + y * st;
            const int c = r[x];
            int cls;
            if (KIND == X265HIP_LF_STATS_BO) cls = c >> (a.depth - 5);
            else if (KIND == X265HIP_LF_STATS_E0) cls = sgn(c - (int)r[x + 1]) + sgn(c - (int)r[x - 1]) + 2;
            else if (KIND == X265HIP_LF_STATS_E1)
                cls = sgn(c - (int)r[x + st]) + (y == 0 ? (int)up1[x] : sgn(c - (int)r[x - st])) + 2;
            else if (KIND == X265HIP_LF_STATS_E2)
                cls = sgn(c - (int)r[x + st + 1]) + (y == 0 ? (int)up1[x] : sgn(c - (int)r[x - st - 1])) + 2;
            else
                cls = sgn(c - (int)r[x + st - 1]) + (y == 0 ? (int)up1[x] : sgn(c - (int)r[x - st + 1])) + 2;
            atomicAdd(&sStat[cls], (int)diff[y * a.p[0].stride + x]);
            atomicAdd(&sCnt[cls], 1);
        }
        __syncthreads();
        // final contents of the carried sign buffers (read back by the caller)
        if (KIND == X265HIP_LF_STATS_E1)
        {
            const Px* r = rec + (long)(endY - 1) * st;
            for (int x = tid; x < endX; x += nth) up1[x] = (int8_t)sgn((int)r[x + st] - (int)r[x]);
        }
        else if (KIND == X265HIP_LF_STATS_E2)
        {
            // row y writes T_y[0] = sign(rec[y+1][0] - rec[y][-1]), T_y[x+1] = sign(rec[y+1][x+1] - rec[y][x]) into the
            // buffer that plays "upBufft" for that row; the two buffers trade places after every row
            const int ye = (endY - 1) & ~1;                  // last even row -> lands in the caller's upBufft
            const int yo = endY >= 2 ? ((endY - 2) | 1) : -1; // last odd row  -> lands in the caller's upBuff1
            for (int k = tid; k <= endX; k += nth)
            {
                const Px* re = rec + (long)ye * st;
                upt[k] = (int8_t)sgn((int)re[st + k] - (int)re[k - 1]);
                if (yo >= 0)
                {
                    const Px* ro = rec + (long)yo * st;
                    up1[k] = (int8_t)sgn((int)ro[st + k] - (int)ro[k - 1]);
                }
            }
        }
        else if (KIND == X265HIP_LF_STATS_E3)
        {
            const Px* r = rec + (long)(endY - 1) * st;
            for (int k = tid; k <= endX; k += nth)             // indices -1 .. endX-1
                up1[k - 1] = (int8_t)sgn((int)r[st + k - 1] - (int)r[k]);
        }
        int32_t* res = (int32_t*)a.result + (long)blockIdx.x * 64;
        const int bins = KIND == X265HIP_LF_STATS_BO ? 32 : 5;
        if (tid < bins)
        {
            const int o = KIND == X265HIP_LF_STATS_BO ? tid : kEoTable[tid];
            res[o] += sStat[tid];
            res[32 + o] += sCnt[tid];
        }
    }
}

} // namespace x265hip

using namespace x265hip;

extern "C" int x265hip_loopfilter_batch(int kind, int depth, const x265hip_plane planes[4], const x265hip_job* jobs, int njobs,
                                        uint32_t* result, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!planes || !jobs || njobs < 0) { set_error("loopfilter_batch: NULL operand"); return X265HIP_EINVAL; }
    if (njobs == 0) return 0;
    if (depth != 8 && depth != 10 && depth != 12) { set_error("loopfilter_batch: depth %d", depth); return X265HIP_EINVAL; }
    if (kind >= X265HIP_LF_STATS_BO && kind <= X265HIP_LF_STATS_E3 && !result) { set_error("loopfilter_batch: statistics need result"); return X265HIP_EINVAL; }
    LfArgs a;
    for (int i = 0; i < 4; i++) a.p[i] = planes[i];
    a.jobs = jobs; a.result = result; a.njobs = njobs; a.depth = depth;
    hipStream_t s = (hipStream_t)stream;
    int blocks = njobs, threads = 256;
    if (kind == X265HIP_LF_DEBLOCK_LUMA_STRONG || kind == X265HIP_LF_DEBLOCK_CHROMA) blocks = (njobs + 63) / 64;
    if (kind == X265HIP_LF_ADS || kind == X265HIP_LF_SAO_E3) threads = 64;
    if (kind == X265HIP_LF_SAO_E0) threads = 128;
#define CASE(PX, K) case K: hipLaunchKernelGGL((lf_kernel<PX, K>), dim3(blocks), dim3(threads), 0, s, a); break;
#define ALL(PX) switch (kind) { \
        CASE(PX, X265HIP_LF_SIGN) CASE(PX, X265HIP_LF_SAO_E0) CASE(PX, X265HIP_LF_SAO_E1) CASE(PX, X265HIP_LF_SAO_E1_2ROWS) \
        CASE(PX, X265HIP_LF_SAO_E2) CASE(PX, X265HIP_LF_SAO_E3) CASE(PX, X265HIP_LF_SAO_B0) CASE(PX, X265HIP_LF_STATS_BO) \
        CASE(PX, X265HIP_LF_STATS_E0) CASE(PX, X265HIP_LF_STATS_E1) CASE(PX, X265HIP_LF_STATS_E2) CASE(PX, X265HIP_LF_STATS_E3) \
        CASE(PX, X265HIP_LF_DEBLOCK_LUMA_STRONG) CASE(PX, X265HIP_LF_DEBLOCK_CHROMA) CASE(PX, X265HIP_LF_INTEGRAL_H) \
        CASE(PX, X265HIP_LF_INTEGRAL_V) CASE(PX, X265HIP_LF_ADS) \
        default: set_error("loopfilter_batch: unknown kind %d", kind); return X265HIP_EINVAL; }
    if (depth == 8) { ALL(uint8_t) } else { ALL(uint16_t) }
#undef ALL
#undef CASE
    X265HIP_TRY(hipGetLastError());
    return 0;
}
