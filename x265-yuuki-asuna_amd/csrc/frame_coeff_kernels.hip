// frame_coeff_kernels.hip - the last two slot families of the EncoderPrimitives table on the device:
//
//  x265hip_frame_batch      frame-level helpers: planecopy_cp / _sp / _sp_shl / _pp_shr (pixel.cpp:864-910), planeClipAndMax (:996-1016),
//                           ssim_4x4x2_core / ssim_end_4 (:631-701), cuTreeFix8Pack / Unpack (:945-958)
//  x265hip_frame_init_lowres  frameInitLowres / frameInitLowerRes of any size (:604-629; the whole-CTU form with the border extension
//                           fused is x265hip_lowres_init)
//  x265hip_propagate_cost   estimateCUPropagateCost over one row of lowres blocks (:912-943; the fused form is x265hip_cutree_propagate)
//  x265hip_coeff_batch      the RDOQ helpers of Quant::rdoQuant (quant.cpp:609-1300): scanPosLast, findPosFirstLast, costCoeffNxN,
//                           costCoeffRemain, costC1C2Flag and the four uncoded-cost pre-passes (dct.cpp:757-1069)
//
// Mapping.  The plane kernels are HBM-bound byte / word moves: a thread owns 16 consecutive samples of a row (16-byte loads and stores when the row
// pointers allow), a row of the grid per picture row, a grid layer per job, so that a wavefront reads and writes 1 (8-bit) or 2 KiB (16-bit) contiguous bytes.  The coefficient
// helpers are serial in the CABAC context state BY DEFINITION (every bin's cost and next state depend on the previous bin of the same
// context): what is parallel is the batch - thousands of coefficient groups of different TUs, each with its own copy of the contexts -
// so a lane owns one call and walks its <= 16 steps; scanPosLast alone is parallel inside a call (a lane per coefficient group of the
// TU, a wavefront per call, the "stop after numSig non-zeros" rule resolved with a prefix sum over the lanes).
//
// Floating point: ssim_end_4 and propagateCost follow the reference's C operation by operation with explicitly rounded multiplies /
// adds / divides (__fmul_rn & co. are never contracted into FMAs; the reference's C is compiled without FMA contraction).
#include "common.h"

namespace x265hip {
namespace {

struct FrameArgs
{
    int kind, depth, w, h;
    x265hip_plane p0, p1;
    const x265hip_job* jobs;
    int njobs;
    void* out;
    float c1f, c2f; int c1i, c2i;                           // ssim_end_1's constants for this depth, evaluated by the launcher in host double arithmetic
};

template <typename S, typename D, int KIND>
__device__ __forceinline__ D convert_sample(S v, int shift, int mask)
{
    if (KIND == X265HIP_FR_PLANECOPY_CP) return (D)((D)v << shift);                        // pixel.cpp:864-874
    if (KIND == X265HIP_FR_PLANECOPY_SP) return (D)(((int)v >> shift) & mask);             // :876-886
    if (KIND == X265HIP_FR_PLANECOPY_SP_SHL) return (D)(((int)v << shift) & mask);         // :888-898
    return (D)((int)v >> shift);                                                           // planecopy_pp_shr :900-910
}

// grid (ceil(w / 4096), h, njobs) x 256 threads: a thread owns 16 consecutive samples of a row - one or two 16-byte loads and stores when both row pointers are 16-byte
// aligned (whole planes of the PicYuv layout are: strides are multiples of 64), sample by sample otherwise (arbitrary offsets, the row's tail)
template <typename S, typename D, int KIND>
__global__ void __launch_bounds__(256) planecopy_kernel(FrameArgs a)
{
    const x265hip_job jb = a.jobs[blockIdx.z];
    const int y = blockIdx.y, x0 = (blockIdx.x * 256 + threadIdx.x) * 16;
    if (x0 >= a.w) return;
    const S* s = (const S*)a.p0.base + jb.off[0] + (intptr_t)y * a.p0.stride + x0;
    D* d = (D*)a.p1.base + jb.off[1] + (intptr_t)y * a.p1.stride + x0;
    const int n = min(16, a.w - x0);
    if (n == 16 && !(((uintptr_t)s | (uintptr_t)d) & 15))
    {
        S v[16]; D o[16];
        __builtin_memcpy(v, __builtin_assume_aligned(s, 16), sizeof(v));
#pragma unroll
        for (int i = 0; i < 16; i++) o[i] = convert_sample<S, D, KIND>(v[i], jb.arg[0], jb.arg[1]);
        __builtin_memcpy(__builtin_assume_aligned(d, 16), o, sizeof(o));
        return;
    }
    for (int i = 0; i < n; i++) d[i] = convert_sample<S, D, KIND>(s[i], jb.arg[0], jb.arg[1]);
}

// planeClipAndMax (pixel.cpp:996-1016): clamp in place, the plane's maximum and sum.  out[2 job] = max, out[2 job + 1] = sum (zeroed by
// the launcher).  A thread owns 16 samples of a row like the copies; a workgroup walks CLIP_ROWS rows, reduces through shuffles and LDS and issues ONE
// atomic pair: with one pair per wavefront and row the 200 000 atomics of 24 planes on 48 addresses were the whole launch (2.26 ms -> see profiles/r06_prims_frame_coeff.txt)
constexpr int CLIP_ROWS = 16;
template <typename Px>
__global__ void __launch_bounds__(256) plane_clip_max_kernel(FrameArgs a)
{
    __shared__ unsigned sMax[4], sSum[4];
    const x265hip_job jb = a.jobs[blockIdx.z];
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 16;
    const int lo = jb.arg[0], hi = jb.arg[1];
    unsigned mx = 0, sum = 0;
    if (x0 < a.w)
        for (int y = blockIdx.y * CLIP_ROWS; y < min(a.h, (int)(blockIdx.y + 1) * CLIP_ROWS); y++)
        {
            Px* p = (Px*)a.p0.base + jb.off[0] + (intptr_t)y * a.p0.stride + x0;
            const int n = min(16, a.w - x0);
            if (n == 16 && !((uintptr_t)p & 15))
            {
                Px v[16];
                __builtin_memcpy(v, __builtin_assume_aligned(p, 16), sizeof(v));
#pragma unroll
                for (int i = 0; i < 16; i++)
                {
                    const int c = min(max((int)v[i], lo), hi);
                    v[i] = (Px)c; mx = max(mx, (unsigned)c); sum += (unsigned)c;
                }
                __builtin_memcpy(__builtin_assume_aligned(p, 16), v, sizeof(v));
            }
            else
                for (int i = 0; i < n; i++)
                {
                    const int c = min(max((int)p[i], lo), hi);
                    p[i] = (Px)c; mx = max(mx, (unsigned)c); sum += (unsigned)c;
                }
        }
    // a thread's sum: 16 rows x 16 samples x 4095 < 2^21; a wavefront's < 2^27; the workgroup's < 2^29
    for (int o = 32; o; o >>= 1) { mx = max(mx, (unsigned)__shfl_xor((int)mx, o)); sum += (unsigned)__shfl_xor((int)sum, o); }
    if ((threadIdx.x & 63) == 0) { sMax[threadIdx.x >> 6] = mx; sSum[threadIdx.x >> 6] = sum; }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        mx = max(max(sMax[0], sMax[1]), max(sMax[2], sMax[3])); sum = sSum[0] + sSum[1] + sSum[2] + sSum[3];
        if (sum | mx)
        {
            unsigned long long* out = (unsigned long long*)a.out + 2 * blockIdx.z;
            atomicMax(&out[0], (unsigned long long)mx);
            atomicAdd(&out[1], (unsigned long long)sum);
        }
    }
}

// ssim_4x4x2_core (pixel.cpp:631-657): a thread per 4x4 block, two per job; out int32 [job][2][4] = {s1, s2, ss, s12}
template <typename Px>
__global__ void __launch_bounds__(256) ssim_core_kernel(FrameArgs a)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 2 * a.njobs) return;
    const x265hip_job jb = a.jobs[t >> 1];
    const int z = t & 1;
    const Px* p1 = (const Px*)a.p0.base + jb.off[0] + 4 * z;
    const Px* p2 = (const Px*)a.p1.base + jb.off[1] + 4 * z;
    uint32_t s1 = 0, s2 = 0, ss = 0, s12 = 0;
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++)
        {
            const uint32_t va = p1[x + y * a.p0.stride], vb = p2[x + y * a.p1.stride];
            s1 += va; s2 += vb; ss += va * va + vb * vb; s12 += va * vb;
        }
    int32_t* o = (int32_t*)a.out + (size_t)t * 4;
    o[0] = (int32_t)s1; o[1] = (int32_t)s2; o[2] = (int32_t)ss; o[3] = (int32_t)s12;
}

// ssim_end_1 (pixel.cpp:659-686): integer moments arithmetic in the 8-bit build, float in the high-bit-depth builds
template <bool HIGH>
__device__ float ssim_end_1(int s1, int s2, int ss, int s12, const FrameArgs& a)
{
    if (HIGH)
    {
        const float c1 = a.c1f, c2 = a.c2f;
        const float fs1 = (float)s1, fs2 = (float)s2, fss = (float)ss, fs12 = (float)s12;
        const float vars = __fsub_rn(__fsub_rn(__fmul_rn(fss, 64.0f), __fmul_rn(fs1, fs1)), __fmul_rn(fs2, fs2));
        const float covar = __fsub_rn(__fmul_rn(fs12, 64.0f), __fmul_rn(fs1, fs2));
        const float n0 = __fadd_rn(__fmul_rn(__fmul_rn(2.0f, fs1), fs2), c1), n1 = __fadd_rn(__fmul_rn(2.0f, covar), c2);
        const float d0 = __fadd_rn(__fadd_rn(__fmul_rn(fs1, fs1), __fmul_rn(fs2, fs2)), c1), d1 = __fadd_rn(vars, c2);
        return __fdiv_rn(__fmul_rn(n0, n1), __fmul_rn(d0, d1));
    }
    const int c1 = a.c1i, c2 = a.c2i;
    const int vars = ss * 64 - s1 * s1 - s2 * s2;
    const int covar = s12 * 64 - s1 * s2;
    return __fdiv_rn(__fmul_rn((float)(2 * s1 * s2 + c1), (float)(2 * covar + c2)), __fmul_rn((float)(s1 * s1 + s2 * s2 + c1), (float)(vars + c2)));
}

// ssim_end_4 (pixel.cpp:688-701): a thread per job; p0 = sum0, p1 = sum1 (int32 [5][4] at off[0] / off[1]), arg[0] = width 1..4
template <bool HIGH>
__global__ void __launch_bounds__(256) ssim_end_kernel(FrameArgs a)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= a.njobs) return;
    const x265hip_job jb = a.jobs[t];
    const int32_t* s0 = (const int32_t*)a.p0.base + jb.off[0];
    const int32_t* s1 = (const int32_t*)a.p1.base + jb.off[1];
    float ssim = 0.0f;
    for (int i = 0; i < jb.arg[0]; i++)
    {
        int m[4];
        for (int k = 0; k < 4; k++) m[k] = (int)((uint32_t)s0[i * 4 + k] + (uint32_t)s0[i * 4 + 4 + k] + (uint32_t)s1[i * 4 + k] + (uint32_t)s1[i * 4 + 4 + k]);
        ssim = __fadd_rn(ssim, ssim_end_1<HIGH>(m[0], m[1], m[2], m[3], a));
    }
    ((float*)a.out)[t] = ssim;
}

// cuTreeFix8Pack / Unpack (pixel.cpp:945-958): grid (ceil(w / 256), 1, njobs), w = count
__global__ void __launch_bounds__(256) fix8_kernel(FrameArgs a)
{
    const x265hip_job jb = a.jobs[blockIdx.z];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.w) return;
    if (a.kind == X265HIP_FR_FIX8_PACK)
        ((uint16_t*)a.p1.base + jb.off[1])[i] = (uint16_t)(int16_t)(int)__dmul_rn(((const double*)a.p0.base + jb.off[0])[i], 256.0);
    else
        ((double*)a.p1.base + jb.off[1])[i] = __ddiv_rn((double)(int16_t)((const uint16_t*)a.p0.base + jb.off[0])[i], 256.0);
}

struct LowresArgs { const void* src; void* dst[4]; intptr_t srcStride, dstStride; int w, h; };

// frameInitLowres (pixel.cpp:604-629): each sample = rounded average of two rounded vertical averages; a thread per lowres sample
template <typename Px>
__global__ void __launch_bounds__(256) frame_init_lowres_kernel(LowresArgs a)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= a.w) return;
    const Px* r0 = (const Px*)a.src + (intptr_t)2 * y * a.srcStride + 2 * x;
    const Px* r1 = r0 + a.srcStride;
    const Px* r2 = r1 + a.srcStride;
    auto avg = [](int p, int q) { return (p + q + 1) >> 1; };
    const int a01 = avg(r0[0], r1[0]), b01 = avg(r0[1], r1[1]), c01 = avg(r0[2], r1[2]);
    const int a12 = avg(r1[0], r2[0]), b12 = avg(r1[1], r2[1]), c12 = avg(r1[2], r2[2]);
    const intptr_t o = (intptr_t)y * a.dstStride + x;
    ((Px*)a.dst[0])[o] = (Px)avg(a01, b01);
    ((Px*)a.dst[1])[o] = (Px)avg(b01, c01);
    ((Px*)a.dst[2])[o] = (Px)avg(a12, b12);
    ((Px*)a.dst[3])[o] = (Px)avg(b12, c12);
}

struct PropagateArgs { int32_t* dst; const uint16_t* in; const int32_t* intra; const uint16_t* inter; const int32_t* invq; double fps; int len; };

// estimateCUPropagateCost (pixel.cpp:914-943)
__global__ void __launch_bounds__(256) propagate_cost_kernel(PropagateArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.len) return;
    const int intra = a.intra[i];
    const int inter = min(intra, (int)(a.inter[i] & 0x3fff));                               // LOWRES_COST_MASK, slicetype.h:41
    const double propagateIntra = (double)(int)((uint32_t)intra * (uint32_t)a.invq[i]);
    const double amount = __dadd_rn((double)a.in[i], __dmul_rn(propagateIntra, a.fps));
    const double r = __dadd_rn(__ddiv_rn(__dmul_rn(amount, (double)(intra - inter)), (double)intra), 0.5);
    a.dst[i] = (r >= -2147483649.0 && r < 2147483648.0) ? (int)r : (int)0x80000000;        // cvttsd2si's answer outside the int range / for NaN
}

// ------------------------------------------------------------------------------------------------ coefficient helpers (row a9)
__constant__ uint8_t kTransIdxLps[64] = {                 // ITU-T H.265 table 9-46
    0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
    24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63 };

struct EntropyBits { uint32_t v[128]; };

struct CoeffArgs
{
    int kind, depth, njobs;
    void* buf[5];
    const x265hip_coeff_job* jobs;
    uint32_t* result;
    EntropyBits bits;                                      // the host's per-state bit costs, by value (512 bytes of kernel arguments)
};

// scanPosLast (dct.cpp:757-791): a wavefront per call, lane = coefficient group.  buf0 scan (uint16), buf1 coeff (int16), outputs
// buf2 coeffSign / buf3 coeffFlag (uint16 [64]) / buf4 coeffNum (uint8 [64]); arg0 numSig, arg1 trSize; result = last scan position
__global__ void __launch_bounds__(256) scan_pos_last_kernel(CoeffArgs a)
{
    const int job = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (job >= a.njobs) return;
    const x265hip_coeff_job jb = a.jobs[job];
    const uint16_t* scan = (const uint16_t*)a.buf[0] + jb.off[0];
    const int16_t* coeff = (const int16_t*)a.buf[1] + jb.off[1];
    const int numSig = jb.arg[0], ncg = (jb.arg[1] * jb.arg[1]) >> 4;
    uint32_t nz = 0, neg = 0;
    if (lane < ncg)
        for (int i = 0; i < 16; i++)
        {
            const int c = coeff[scan[lane * 16 + i]];
            nz |= (uint32_t)(c != 0) << i; neg |= (uint32_t)(c < 0) << i;
        }
    const int cnt = __popc(nz);
    int incl = cnt;
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    const int before = incl - cnt;
    // the group the walk stops in: the one holding the numSig-th non-zero (numSig <= 0: the do-while still visits position 0)
    const bool stopHere = numSig <= 0 ? lane == 0 : (before < numSig && numSig <= incl);
    const unsigned long long stopMask = __ballot(stopHere);
    const int stopLane = stopMask ? __ffsll((long long)stopMask) - 1 : 64;     // 64: numSig exceeds the block's non-zeros (a caller error): every group counts
    int last = 15;
    if (stopHere)
    {
        last = 0;
        if (numSig > 0) { int k = numSig - before; uint32_t m = nz; while (--k) m &= m - 1; last = __ffs((int)m) - 1; }
    }
    uint32_t keep = lane < stopLane ? 0xffffu : (lane == stopLane ? (2u << last) - 1 : 0u);
    nz &= keep;
    uint32_t sign = 0, flag = 0; int n = 0;
    for (int i = 0; i <= last; i++)
    {
        const uint32_t b = (nz >> i) & 1;
        sign |= (b & (neg >> i)) << n;
        flag = (flag << 1) | b;
        n += (int)b;
    }
    if (lane > stopLane) flag = 0;
    ((uint16_t*)a.buf[2] + jb.off[2])[lane] = (uint16_t)sign;
    ((uint16_t*)a.buf[3] + jb.off[3])[lane] = (uint16_t)flag;
    ((uint8_t*)a.buf[4] + jb.off[4])[lane] = (uint8_t)n;
    if (lane == (stopLane < 64 ? stopLane : 0)) a.result[job] = stopLane < 64 ? (uint32_t)(stopLane * 16 + last) : (uint32_t)(ncg * 16 - 1);
}

// the serial helpers: a lane per call; the per-state costs and the CABAC transition table live in LDS
__global__ void __launch_bounds__(256) coeff_serial_kernel(CoeffArgs a)
{
    __shared__ uint32_t sBits[128];
    __shared__ uint8_t sNext[256];
    if (threadIdx.x < 128) sBits[threadIdx.x] = a.bits.v[threadIdx.x] & 0xFFFFFFu;
    {   // context byte = pStateIdx << 1 | valMps (contexts.h:116 sbacNext); transIdxMps = min(p + 1, 62)
        const int s = threadIdx.x >> 1, bin = threadIdx.x & 1, p = s >> 1, mps = s & 1;
        sNext[threadIdx.x] = (uint8_t)(bin == mps ? ((p < 62 ? p + 1 : p) << 1) | mps : ((int)kTransIdxLps[p] << 1) | (p == 0 ? 1 - mps : mps));
    }
    __syncthreads();
    const int job = blockIdx.x * 256 + threadIdx.x;
    if (job >= a.njobs) return;
    const x265hip_coeff_job jb = a.jobs[job];
    uint32_t res = 0;
    switch (a.kind)
    {
    case X265HIP_CF_FIND_POS_FIRST_LAST:
    {   // dct.cpp:794-835: buf0 scanTbl (uint16 [16]), buf1 the group's top-left coefficient; arg0 trSize
        const uint16_t* tbl = (const uint16_t*)a.buf[0] + jb.off[0];
        const int16_t* c = (const int16_t*)a.buf[1] + jb.off[1];
        const intptr_t tr = jb.arg[0];
        int v[16];
        for (int n = 0; n < 16; n++) v[n] = c[(tbl[n] >> 2) * tr + (tbl[n] & 3)];
        int last = 15, first = 0;
        while (last >= 0 && !v[last]) last--;
        while (first < 16 && !v[first]) first++;
        uint32_t sum = 0;
        for (int n = first; n <= last; n++) sum += (uint32_t)v[n];
        res = (sum << 31) | ((uint32_t)last << 8) | (uint32_t)first;
        break;
    }
    case X265HIP_CF_COST_COEFF_NXN:
    {   // dct.cpp:838-884: buf0 scan (uint16 [16]), buf1 the group's top-left coefficient, buf2 absCoeff (the pointer the caller hands over:
        // stepped back by the known-last-position slot and indexed from that slot on, i.e. the levels land at [0 ..)), buf3 tabSigCtx (uint8 [16]), buf4 contexts
        const uint16_t* scan = (const uint16_t*)a.buf[0] + jb.off[0];
        const int16_t* c = (const int16_t*)a.buf[1] + jb.off[1];
        uint16_t* absCoeff = (uint16_t*)a.buf[2] + jb.off[2];
        const uint8_t* tab = (const uint8_t*)a.buf[3] + jb.off[3];
        uint8_t* ctx = (uint8_t*)a.buf[4] + jb.off[4];
        const intptr_t tr = jb.arg[0];
        uint32_t mask = (uint32_t)jb.arg[1];
        const int offset = jb.arg[2], subPosBase = jb.arg[4];
        int pos = jb.arg[3];
        uint32_t numNonZero = pos < 15 ? 1 : 0, sum = 0;
        absCoeff -= numNonZero;
        do
        {
            const uint32_t blkPos = scan[pos], sig = mask & 1;
            mask >>= 1;
            if (pos != 0 || subPosBase == 0 || numNonZero)
            {
                const uint32_t ctxSig = (subPosBase + pos) ? (uint32_t)(tab[blkPos] + offset) : 0;
                const uint32_t st = ctx[ctxSig];
                sum += sBits[st ^ sig];
                ctx[ctxSig] = sNext[st * 2 + sig];
            }
            const int lv = c[(blkPos >> 2) * tr + (blkPos & 3)];
            absCoeff[numNonZero] = (uint16_t)(lv < 0 ? -lv : lv);
            numNonZero += sig;
        }
        while (--pos >= 0);
        res = sum & 0xFFFFFFu;
        break;
    }
    case X265HIP_CF_COST_COEFF_REMAIN:
    {   // dct.cpp:886-931: buf2 absCoeff; arg0 numNonZero, arg1 idx
        const uint16_t* absCoeff = (const uint16_t*)a.buf[2] + jb.off[2];
        uint32_t rice = 0, sum = 0;
        int baseLevel = 3, idx = jb.arg[1];
        do
        {
            if (idx >= 8) baseLevel = 1;                                   // C1FLAG_NUMBER
            int code = (int)absCoeff[idx] - baseLevel;
            if (code >= 0)
            {
                code = (int)((uint32_t)code >> rice) - 3;                  // COEF_REMAIN_BIN_REDUCTION
                if (code >= 0) code = 2 * (31 - __clz(code + 1));
                sum += 3 + 1 + rice + (uint32_t)code;
                if (absCoeff[idx] > (3u << rice)) rice = (rice + 1) - (rice >> 2);
            }
            baseLevel = 2;
        }
        while (++idx < jb.arg[0]);
        res = sum;
        break;
    }
    case X265HIP_CF_COST_C1C2:
    {   // dct.cpp:934-987: buf2 absCoeff, buf4 baseCtxMod; arg0 numC1Flag, arg1 ctxOffset
        const uint16_t* absCoeff = (const uint16_t*)a.buf[2] + jb.off[2];
        uint8_t* ctx = (uint8_t*)a.buf[4] + jb.off[4];
        uint32_t sum = 0, c1 = 1, firstC2Idx = 8, firstC2Flag = 2, c1Next = 0xFFFFFFFEu;
        int idx = 0;
        do
        {
            const uint32_t gt1 = absCoeff[idx] > 1, gt2 = absCoeff[idx] > 2;
            const uint32_t st = ctx[c1];
            ctx[c1] = sNext[st * 2 + gt1];
            sum += sBits[st ^ gt1];
            if (gt1) c1Next = 0;
            if (gt1 + firstC2Flag == 3) firstC2Flag = gt2;
            if (gt1 + firstC2Idx == 9) firstC2Idx = (uint32_t)idx;
            c1 = c1Next & 3;
            c1Next >>= 2;
        }
        while (++idx < jb.arg[0]);
        if (!c1)
        {
            ctx += jb.arg[1];
            const uint32_t st = ctx[0];
            ctx[0] = sNext[st * 2 + (firstC2Flag & 1)];
            sum += sBits[st ^ firstC2Flag];
        }
        res = (sum & 0x00FFFFFFu) + (c1 << 26) + (firstC2Idx << 28);
        break;
    }
    default:
    {   // the uncoded-cost pre-passes of one coefficient group (dct.cpp:988-1069): buf0 fenc's transform, buf1 the residual's transform (int16,
        // block origin), buf2 costUncoded (int64, block origin), buf3 {totalUncodedCost, totalRdCost} (int64 [2], added to), buf4 psyScale (int64);
        // arg0 blkPos, arg1 log2TrSize, arg2 the row stride when it is not 1 << log2TrSize (a densely staged group).  The reference converts through double: exact, every term is below 2^53
        const int16_t* fenc = (const int16_t*)a.buf[0] + jb.off[0];
        const int16_t* resi = (const int16_t*)a.buf[1] + jb.off[1];
        long long* cost = (long long*)a.buf[2] + jb.off[2];
        long long* tot = (long long*)a.buf[3] + jb.off[3];
        const bool square = a.kind != X265HIP_CF_RDOQ_PSY_2P, psy = a.kind == X265HIP_CF_RDOQ_PSY || a.kind == X265HIP_CF_RDOQ_PSY_2P;
        const long long psyScale = psy ? ((const long long*)a.buf[4] + jb.off[4])[0] : 0;
        const int log2 = jb.arg[1], transformShift = 15 - a.depth - log2, scaleBits = 15 - 2 * transformShift;
        const int psyShift = max(2 * transformShift + 1, 0);
        uint32_t blkPos = (uint32_t)jb.arg[0];
        long long tu = tot[0], trd = tot[1];
        const uint32_t rowStride = jb.arg[2] ? (uint32_t)jb.arg[2] : 1u << log2;
        for (int y = 0; y < 4; y++, blkPos += rowStride)
            for (int x = 0; x < 4; x++)
            {
                const long long cf = resi[blkPos + x];
                long long v = square ? (long long)(double)((cf * cf) << scaleBits) : cost[blkPos + x];
                if (psy) v -= (long long)(double)((psyScale * ((long long)fenc[blkPos + x] - cf)) >> psyShift);
                cost[blkPos + x] = v;
                tu += v; trd += v;
            }
        tot[0] = tu; tot[1] = trd;
        break;
    }
    }
    if (a.result) a.result[job] = res;
}

EntropyBits g_entropyBits;
bool g_haveEntropyBits = false;
std::mutex g_entropyMu;

} // namespace

bool entropy_bits_ready()
{
    std::lock_guard<std::mutex> lk(g_entropyMu);
    return g_haveEntropyBits;
}

} // namespace x265hip

using namespace x265hip;

extern "C" int x265hip_set_entropy_bits(const uint32_t* bits128)
{
    if (!bits128) { set_error("set_entropy_bits: NULL table"); return X265HIP_EINVAL; }
    std::lock_guard<std::mutex> lk(g_entropyMu);
    for (int i = 0; i < 128; i++) g_entropyBits.v[i] = bits128[i] & 0xFFFFFFu;       // x265_entropyStateBits keeps the transition in the top byte
    g_haveEntropyBits = true;
    return 0;
}

extern "C" int x265hip_frame_batch(int kind, int depth, int w, int h, const x265hip_plane planes[2], const x265hip_job* jobs, int njobs,
                                   void* out, void* stream)
{
    if (int rc = ensure_device()) return rc;
    if (njobs == 0) return 0;
    if (!planes || !jobs || njobs < 0) { set_error("frame_batch: NULL operand"); return X265HIP_EINVAL; }
    if (depth != 8 && depth != 10 && depth != 12) { set_error("frame_batch: depth %d", depth); return X265HIP_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    FrameArgs a = { kind, depth, w, h, planes[0], planes[1], jobs, njobs, out, 0.0f, 0.0f, 0, 0 };
    {   // pixel.cpp:666-674
        const int pixelMax = (1 << depth) - 1;
        const volatile double k1 = .01 * .01 * pixelMax * pixelMax * 64, k2 = .03 * .03 * pixelMax * pixelMax * 64 * 63;
        a.c1f = (float)k1; a.c2f = (float)k2;
        const volatile double r1 = k1 + .5, r2 = k2 + .5;
        a.c1i = (int)r1; a.c2i = (int)r2;
    }
    const bool hi = depth > 8;
    const bool plane_kind = kind >= X265HIP_FR_PLANECOPY_CP && kind <= X265HIP_FR_PLANE_CLIP_MAX;
    if (plane_kind && (w <= 0 || h <= 0 || h > 65535 || njobs > 65535)) { set_error("frame_batch: plane %dx%d x %d jobs", w, h, njobs); return X265HIP_EINVAL; }
    const dim3 pg((unsigned)((w + 4095) / 4096), (unsigned)(h > 0 ? h : 1), (unsigned)njobs);
    switch (kind)
    {
    case X265HIP_FR_PLANECOPY_CP:
        if (hi) hipLaunchKernelGGL((planecopy_kernel<uint8_t, uint16_t, X265HIP_FR_PLANECOPY_CP>), pg, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((planecopy_kernel<uint8_t, uint8_t, X265HIP_FR_PLANECOPY_CP>), pg, dim3(256), 0, s, a);
        break;
    case X265HIP_FR_PLANECOPY_SP:
        if (hi) hipLaunchKernelGGL((planecopy_kernel<uint16_t, uint16_t, X265HIP_FR_PLANECOPY_SP>), pg, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((planecopy_kernel<uint16_t, uint8_t, X265HIP_FR_PLANECOPY_SP>), pg, dim3(256), 0, s, a);
        break;
    case X265HIP_FR_PLANECOPY_SP_SHL:
        if (hi) hipLaunchKernelGGL((planecopy_kernel<uint16_t, uint16_t, X265HIP_FR_PLANECOPY_SP_SHL>), pg, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((planecopy_kernel<uint16_t, uint8_t, X265HIP_FR_PLANECOPY_SP_SHL>), pg, dim3(256), 0, s, a);
        break;
    case X265HIP_FR_PLANECOPY_PP_SHR:
        if (hi) hipLaunchKernelGGL((planecopy_kernel<uint16_t, uint16_t, X265HIP_FR_PLANECOPY_PP_SHR>), pg, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((planecopy_kernel<uint8_t, uint8_t, X265HIP_FR_PLANECOPY_PP_SHR>), pg, dim3(256), 0, s, a);
        break;
    case X265HIP_FR_PLANE_CLIP_MAX:
        if (!out) { set_error("frame_batch: planeClipAndMax needs out"); return X265HIP_EINVAL; }
        X265HIP_TRY(hipMemsetAsync(out, 0, (size_t)njobs * 16, s));
        {
            const dim3 cg(pg.x, (unsigned)((h + CLIP_ROWS - 1) / CLIP_ROWS), pg.z);
            if (hi) hipLaunchKernelGGL(plane_clip_max_kernel<uint16_t>, cg, dim3(256), 0, s, a);
            else hipLaunchKernelGGL(plane_clip_max_kernel<uint8_t>, cg, dim3(256), 0, s, a);
        }
        break;
    case X265HIP_FR_SSIM_CORE:
        if (!out) { set_error("frame_batch: ssim core needs out"); return X265HIP_EINVAL; }
        if (hi) hipLaunchKernelGGL(ssim_core_kernel<uint16_t>, dim3((2 * njobs + 255) / 256), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(ssim_core_kernel<uint8_t>, dim3((2 * njobs + 255) / 256), dim3(256), 0, s, a);
        break;
    case X265HIP_FR_SSIM_END4:
        if (!out) { set_error("frame_batch: ssim end needs out"); return X265HIP_EINVAL; }
        if (hi) hipLaunchKernelGGL(ssim_end_kernel<true>, dim3((njobs + 255) / 256), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(ssim_end_kernel<false>, dim3((njobs + 255) / 256), dim3(256), 0, s, a);
        break;
    case X265HIP_FR_FIX8_PACK:
    case X265HIP_FR_FIX8_UNPACK:
        if (w <= 0 || njobs > 65535) { set_error("frame_batch: fix8 count %d x %d jobs", w, njobs); return X265HIP_EINVAL; }
        hipLaunchKernelGGL(fix8_kernel, dim3((w + 255) / 256, 1, njobs), dim3(256), 0, s, a);
        break;
    default:
        set_error("frame_batch: unknown kind %d", kind);
        return X265HIP_EINVAL;
    }
    return check_hip(hipGetLastError(), "frame_batch launch");
}

extern "C" int x265hip_frame_init_lowres(int depth, const void* src, intptr_t src_stride, void* const dst[4], intptr_t dst_stride,
                                         int width, int height, void* stream)
{
    if (int rc = ensure_device()) return rc;
    if (!src || !dst || !dst[0] || !dst[1] || !dst[2] || !dst[3]) { set_error("frame_init_lowres: NULL plane"); return X265HIP_EINVAL; }
    if (width <= 0 || height <= 0 || height > 65535) { set_error("frame_init_lowres: size %dx%d", width, height); return X265HIP_EINVAL; }
    if (depth != 8 && depth != 10 && depth != 12) { set_error("frame_init_lowres: depth %d", depth); return X265HIP_EINVAL; }
    LowresArgs a = { src, { dst[0], dst[1], dst[2], dst[3] }, src_stride, dst_stride, width, height };
    const dim3 g((width + 255) / 256, height);
    if (depth == 8) hipLaunchKernelGGL(frame_init_lowres_kernel<uint8_t>, g, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(frame_init_lowres_kernel<uint16_t>, g, dim3(256), 0, (hipStream_t)stream, a);
    return check_hip(hipGetLastError(), "frame_init_lowres launch");
}

extern "C" int x265hip_propagate_cost(int32_t* dst, const uint16_t* propagate_in, const int32_t* intra_costs, const uint16_t* inter_costs,
                                      const int32_t* inv_qscales, double fps_factor, int len, void* stream)
{
    if (int rc = ensure_device()) return rc;
    if (len == 0) return 0;
    if (!dst || !propagate_in || !intra_costs || !inter_costs || !inv_qscales || len < 0) { set_error("propagate_cost: NULL operand"); return X265HIP_EINVAL; }
    PropagateArgs a = { dst, propagate_in, intra_costs, inter_costs, inv_qscales, fps_factor / 256, len };
    hipLaunchKernelGGL(propagate_cost_kernel, dim3((len + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    return check_hip(hipGetLastError(), "propagate_cost launch");
}

extern "C" int x265hip_coeff_batch(int kind, int depth, void* const bufs[5], const x265hip_coeff_job* jobs, int njobs, uint32_t* result, void* stream)
{
    if (int rc = ensure_device()) return rc;
    if (njobs == 0) return 0;
    if (!bufs || !jobs || njobs < 0) { set_error("coeff_batch: NULL operand"); return X265HIP_EINVAL; }
    if (kind < X265HIP_CF_SCAN_POS_LAST || kind > X265HIP_CF_RDOQ_PSY_2P) { set_error("coeff_batch: unknown kind %d", kind); return X265HIP_EINVAL; }
    if (depth != 8 && depth != 10 && depth != 12) { set_error("coeff_batch: depth %d", depth); return X265HIP_EINVAL; }
    const bool returns = kind <= X265HIP_CF_COST_C1C2;
    if (returns && !result) { set_error("coeff_batch: kind %d needs result", kind); return X265HIP_EINVAL; }
    CoeffArgs a;
    a.kind = kind; a.depth = depth; a.njobs = njobs; a.jobs = jobs; a.result = returns ? result : nullptr;
    for (int i = 0; i < 5; i++) a.buf[i] = bufs[i];
    if (kind == X265HIP_CF_COST_COEFF_NXN || kind == X265HIP_CF_COST_C1C2)
    {
        std::lock_guard<std::mutex> lk(g_entropyMu);
        if (!g_haveEntropyBits) { set_error("coeff_batch: the host's CABAC bit costs were not handed in (x265hip_set_entropy_bits)"); return X265HIP_EINVAL; }
        a.bits = g_entropyBits;
    }
    else
        a.bits = EntropyBits();
    hipStream_t s = (hipStream_t)stream;
    if (kind == X265HIP_CF_SCAN_POS_LAST) hipLaunchKernelGGL(scan_pos_last_kernel, dim3((njobs + 3) / 4), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(coeff_serial_kernel, dim3((njobs + 255) / 256), dim3(256), 0, s, a);
    return check_hip(hipGetLastError(), "coeff_batch launch");
}
