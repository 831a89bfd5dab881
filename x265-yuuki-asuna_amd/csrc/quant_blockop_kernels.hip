// quant_blockop_kernels.hip - batched quantisation and element-wise block primitives on gfx950.
//
// Reference semantics: source/common/dct.cpp quant_c :664-686, nquant_c :688-713, dequant_normal_c
// :612-634, dequant_scaling_c :636-662, denoiseDct_c :744-755, count_nonzero_c :714-726, copy_count
// :728-742; source/common/pixel.cpp blockcopy_* :759-812, pixel_sub_ps_c :814-826, pixel_add_ps_c
// :828-840, addAvg :842-862, pixelavg_pp :545-557, blockfill_s_c :393-399, cpy2Dto1D/1Dto2D :401-469,
// transpose :485-491, weight_sp_c/pp_c :493-543, scale1D_128to64 :559-583, scale2D_64to32 :585-602,
// sse<int16> :167-186, pixel_ssd_s_c :379-391, pixel_var :703-720.
//
// These are streaming integer ops (a few int-ops per byte): one workgroup per block, threads stride
// over the samples; the only cross-lane work is the wave-shuffle + LDS reduction of the counting /
// summing kinds.  All of them are HBM-bound by construction; the frame pipeline fuses them into the
// producing kernels, the table layer uses them one block at a time.
#include "common.h"

#include <cstdlib>
#include <type_traits>

namespace x265hip {

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

// block-wide sum of one value per thread (blockDim multiple of 64, <= 256); valid in thread 0
template <typename T> __device__ __forceinline__ T block_sum(T v, T* scratch)
{
    v = group_sum<64>(v);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) scratch[wave] = v;
    __syncthreads();
    T t = 0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += scratch[i];
    return t;
}

// ------------------------------------------------------------------------------ quant family
struct QArgs { x265hip_plane p[4]; const x265hip_job* jobs; uint32_t* result; };

template <int KIND>
__global__ void __launch_bounds__(256) quant_kernel(QArgs a)
{
    __shared__ uint32_t scratch[4];
    const x265hip_job jb = a.jobs[blockIdx.x];
    const int tid = threadIdx.x;
    uint32_t cnt = 0;
    if (KIND == X265HIP_Q_QUANT || KIND == X265HIP_Q_NQUANT)
    {
        const int16_t* coef = (const int16_t*)a.p[0].base + jb.off[0];
        const int32_t* qc = (const int32_t*)a.p[1].base + jb.off[1];
        int32_t* deltaU = KIND == X265HIP_Q_QUANT ? (int32_t*)a.p[2].base + jb.off[2] : nullptr;
        int16_t* q = (int16_t*)a.p[3].base + jb.off[3];
        const int qBits = jb.arg[0], add = jb.arg[1], n = jb.arg[2];
        auto one = [&](int c, int qcv, int& du, int& lv)
        {
            const int t = iabs(c) * qcv;
            int level = (t + add) >> qBits;
            du = (t - (level << qBits)) >> (qBits - 8);
            cnt += level != 0;
            if (c < 0) level = -level;
            level = clip3(-32768, 32767, level);
            lv = KIND == X265HIP_Q_QUANT ? level : iabs(level);
        };
        if ((n & 3) == 0)
        {
            // 4 coefficients per thread and step: packed int16 dwords in and out, 4 int32 scaling factors
            for (int i = tid * 4; i < n; i += 1024)
            {
                const uint32_t c01 = ld_u32(coef + i), c23 = ld_u32(coef + i + 2);
                const int c[4] = { (int16_t)(c01 & 0xffff), (int16_t)(c01 >> 16), (int16_t)(c23 & 0xffff), (int16_t)(c23 >> 16) };
                int qv[4], du[4], lv[4];
#pragma unroll
                for (int k = 0; k < 4; k++) qv[k] = (int)ld_u32(qc + i + k);
#pragma unroll
                for (int k = 0; k < 4; k++) one(c[k], qv[k], du[k], lv[k]);
                if (KIND == X265HIP_Q_QUANT)
#pragma unroll
                    for (int k = 0; k < 4; k++) *reinterpret_cast<u32_unaligned*>(deltaU + i + k) = (uint32_t)du[k];
                *reinterpret_cast<u32_unaligned*>(q + i) = ((uint32_t)lv[0] & 0xffffu) | ((uint32_t)lv[1] << 16);
                *reinterpret_cast<u32_unaligned*>(q + i + 2) = ((uint32_t)lv[2] & 0xffffu) | ((uint32_t)lv[3] << 16);
            }
        }
        else
            for (int i = tid; i < n; i += 256)
            {
                int du, lv;
                one(coef[i], qc[i], du, lv);
                if (KIND == X265HIP_Q_QUANT) deltaU[i] = du;
                q[i] = (int16_t)lv;
            }
    }
    else if (KIND == X265HIP_Q_DEQUANT_NORMAL)
    {
        const int16_t* q = (const int16_t*)a.p[0].base + jb.off[0];
        int16_t* coef = (int16_t*)a.p[3].base + jb.off[3];
        const int n = jb.arg[0], scale = jb.arg[1], shift = jb.arg[2];
        const int add = 1 << (shift - 1);
        auto deq = [&](int v) { return clip3(-32768, 32767, (v * scale + add) >> shift); };
        if ((n & 3) == 0)
            for (int i = tid * 4; i < n; i += 1024)
            {
                const uint32_t a01 = ld_u32(q + i), a23 = ld_u32(q + i + 2);
                const int r0 = deq((int16_t)(a01 & 0xffff)), r1 = deq((int16_t)(a01 >> 16)), r2 = deq((int16_t)(a23 & 0xffff)), r3 = deq((int16_t)(a23 >> 16));
                *reinterpret_cast<u32_unaligned*>(coef + i) = ((uint32_t)r0 & 0xffffu) | ((uint32_t)r1 << 16);
                *reinterpret_cast<u32_unaligned*>(coef + i + 2) = ((uint32_t)r2 & 0xffffu) | ((uint32_t)r3 << 16);
            }
        else
            for (int i = tid; i < n; i += 256) coef[i] = (int16_t)deq((int)q[i]);
    }
    else if (KIND == X265HIP_Q_DEQUANT_SCALING)
    {
        const int16_t* q = (const int16_t*)a.p[0].base + jb.off[0];
        const int32_t* dq = (const int32_t*)a.p[1].base + jb.off[1];
        int16_t* coef = (int16_t*)a.p[3].base + jb.off[3];
        const int n = jb.arg[0], per = jb.arg[1], shift = jb.arg[2] + 4;
        for (int i = tid; i < n; i += 256)
        {
            if (shift > per)
                coef[i] = (int16_t)clip3(-32768, 32767, ((int)q[i] * dq[i] + (1 << (shift - per - 1))) >> (shift - per));
            else
                coef[i] = (int16_t)clip3(-32768, 32767, clip3(-32768, 32767, (int)q[i] * dq[i]) << (per - shift));
        }
    }
    else if (KIND == X265HIP_Q_DENOISE)
    {
        int16_t* coef = (int16_t*)a.p[0].base + jb.off[0];
        uint32_t* resSum = (uint32_t*)a.p[1].base + jb.off[1];
        const uint16_t* offset = (const uint16_t*)a.p[2].base + jb.off[2];
        const int n = jb.arg[0];
        for (int i = tid; i < n; i += 256)
        {
            const int level = coef[i];
            const bool neg = level < 0;
            int mag = neg ? -level : level;
            resSum[i] += (uint32_t)mag;
            mag -= offset[i];
            coef[i] = (int16_t)(mag < 0 ? 0 : (neg ? -mag : mag));
        }
    }
    else if (KIND == X265HIP_Q_COUNT_NONZERO)
    {
        const int16_t* q = (const int16_t*)a.p[0].base + jb.off[0];
        for (int i = tid; i < jb.arg[0]; i += 256) cnt += q[i] != 0;
    }
    else if (KIND == X265HIP_Q_COPY_CNT)
    {
        const int16_t* r = (const int16_t*)a.p[0].base + jb.off[0];
        int16_t* coeff = (int16_t*)a.p[3].base + jb.off[3];
        const int n = jb.arg[0];
        for (int i = tid; i < n * n; i += 256)
        {
            const int16_t v = r[(long)(i / n) * a.p[0].stride + (i % n)];
            coeff[i] = v;
            cnt += v != 0;
        }
    }
    if (a.result)
    {
        const uint32_t tot = block_sum<uint32_t>(cnt, scratch);
        if (tid == 0) a.result[blockIdx.x] = tot;
    }
}

// ------------------------------------------------------------------------------ block ops
struct OpArgs { x265hip_plane p[3]; const x265hip_job* jobs; unsigned long long* result; int w, h, depth; };

template <typename Px, int OP>
__global__ void __launch_bounds__(256) blockop_kernel(OpArgs a)
{
    __shared__ unsigned long long scratch[4];
    const x265hip_job jb = a.jobs[blockIdx.x];
    const int tid = threadIdx.x, nth = blockDim.x;
    const int w = a.w, h = a.h, depth = a.depth;
    const int maxVal = (1 << depth) - 1;
    const long s0 = a.p[0].stride, s1 = a.p[1].stride, s2 = a.p[2].stride;
    unsigned long long red = 0, red2 = 0;

    if (OP == X265HIP_OP_SCALE1D_128TO64)
    {
        Px* d = (Px*)a.p[0].base + jb.off[0];
        const Px* s = (const Px*)a.p[1].base + jb.off[1];
        for (int x = tid; x < 64; x += nth)
        {
            d[x] = (Px)(((int)s[2 * x] + s[2 * x + 1] + 1) >> 1);
            d[64 + x] = (Px)(((int)s[128 + 2 * x] + s[128 + 2 * x + 1] + 1) >> 1);
        }
        return;
    }
    if (OP == X265HIP_OP_SCALE2D_64TO32)
    {
        Px* d = (Px*)a.p[0].base + jb.off[0];
        const Px* s = (const Px*)a.p[1].base + jb.off[1];
        for (int i = tid; i < 32 * 32; i += nth)
        {
            const int y = i >> 5, x = i & 31;
            const Px* p = s + (long)(2 * y) * s1 + 2 * x;
            d[i] = (Px)(((int)p[0] + p[1] + p[s1] + p[s1 + 1] + 2) >> 2);
        }
        return;
    }

    for (int i = tid; i < w * h; i += nth)
    {
        const int y = i / w, x = i - y * w;
        switch (OP)
        {
        case X265HIP_OP_COPY_PP: ((Px*)a.p[0].base + jb.off[0])[y * s0 + x] = ((const Px*)a.p[1].base + jb.off[1])[y * s1 + x]; break;
        case X265HIP_OP_COPY_PS: ((int16_t*)a.p[0].base + jb.off[0])[y * s0 + x] = (int16_t)((const Px*)a.p[1].base + jb.off[1])[y * s1 + x]; break;
        case X265HIP_OP_COPY_SP: ((Px*)a.p[0].base + jb.off[0])[y * s0 + x] = (Px)((const int16_t*)a.p[1].base + jb.off[1])[y * s1 + x]; break;
        case X265HIP_OP_COPY_SS: ((int16_t*)a.p[0].base + jb.off[0])[y * s0 + x] = ((const int16_t*)a.p[1].base + jb.off[1])[y * s1 + x]; break;
        case X265HIP_OP_SUB_PS:
            ((int16_t*)a.p[0].base + jb.off[0])[y * s0 + x] =
                (int16_t)((int)((const Px*)a.p[1].base + jb.off[1])[y * s1 + x] - (int)((const Px*)a.p[2].base + jb.off[2])[y * s2 + x]);
            break;
        case X265HIP_OP_ADD_PS:
            ((Px*)a.p[0].base + jb.off[0])[y * s0 + x] =
                (Px)clip3(0, maxVal, (int)((const Px*)a.p[1].base + jb.off[1])[y * s1 + x] + (int)((const int16_t*)a.p[2].base + jb.off[2])[y * s2 + x]);
            break;
        case X265HIP_OP_ADDAVG:
        {
            const int shift = 14 + 1 - depth, offset = (1 << (shift - 1)) + 2 * 8192;
            ((Px*)a.p[0].base + jb.off[0])[y * s0 + x] =
                (Px)clip3(0, maxVal, ((int)((const int16_t*)a.p[1].base + jb.off[1])[y * s1 + x] + (int)((const int16_t*)a.p[2].base + jb.off[2])[y * s2 + x] + offset) >> shift);
            break;
        }
        case X265HIP_OP_PIXELAVG:
            ((Px*)a.p[0].base + jb.off[0])[y * s0 + x] =
                (Px)(((int)((const Px*)a.p[1].base + jb.off[1])[y * s1 + x] + (int)((const Px*)a.p[2].base + jb.off[2])[y * s2 + x] + 1) >> 1);
            break;
        case X265HIP_OP_BLOCKFILL: ((int16_t*)a.p[0].base + jb.off[0])[y * s0 + x] = (int16_t)jb.arg[0]; break;
        case X265HIP_OP_CPY2DTO1D_SHL:
            ((int16_t*)a.p[0].base + jb.off[0])[i] = (int16_t)((int)((const int16_t*)a.p[1].base + jb.off[1])[y * s1 + x] << jb.arg[0]); break;
        case X265HIP_OP_CPY2DTO1D_SHR:
            ((int16_t*)a.p[0].base + jb.off[0])[i] =
                (int16_t)(((int)((const int16_t*)a.p[1].base + jb.off[1])[y * s1 + x] + (int16_t)(1 << (jb.arg[0] - 1))) >> jb.arg[0]);
            break;
        case X265HIP_OP_CPY1DTO2D_SHL:
            ((int16_t*)a.p[0].base + jb.off[0])[y * s0 + x] = (int16_t)((int)((const int16_t*)a.p[1].base + jb.off[1])[i] << jb.arg[0]); break;
        case X265HIP_OP_CPY1DTO2D_SHR:
            ((int16_t*)a.p[0].base + jb.off[0])[y * s0 + x] =
                (int16_t)(((int)((const int16_t*)a.p[1].base + jb.off[1])[i] + (int16_t)(1 << (jb.arg[0] - 1))) >> jb.arg[0]);
            break;
        case X265HIP_OP_TRANSPOSE:
            ((Px*)a.p[0].base + jb.off[0])[y * w + x] = ((const Px*)a.p[1].base + jb.off[1])[x * s1 + y]; break;
        case X265HIP_OP_WEIGHT_PP:
        {
            const int16_t v = (int16_t)((int)((const Px*)a.p[1].base + jb.off[1])[y * s1 + x] << (14 - depth));
            ((Px*)a.p[0].base + jb.off[0])[y * s0 + x] = (Px)clip3(0, maxVal, ((jb.arg[0] * (int)v + jb.arg[1]) >> jb.arg[2]) + jb.arg[3]);
            break;
        }
        case X265HIP_OP_WEIGHT_SP:
            ((Px*)a.p[0].base + jb.off[0])[y * s0 + x] =
                (Px)clip3(0, maxVal, ((jb.arg[0] * ((int)((const int16_t*)a.p[1].base + jb.off[1])[y * s1 + x] + 8192) + jb.arg[1]) >> jb.arg[2]) + jb.arg[3]);
            break;
        case X265HIP_OP_SSE_SS:
        {
            const int d = (int)((const int16_t*)a.p[0].base + jb.off[0])[y * s0 + x] - (int)((const int16_t*)a.p[1].base + jb.off[1])[y * s1 + x];
            red += (unsigned long long)(long long)(d * d);      // int product, then widened like the reference's sse_t +=
            break;
        }
        case X265HIP_OP_SSD_S:
        {
            const int v = ((const int16_t*)a.p[0].base + jb.off[0])[y * s0 + x];
            red += (unsigned long long)(long long)(v * v);
            break;
        }
        case X265HIP_OP_VAR:
        {
            const unsigned v = ((const Px*)a.p[0].base + jb.off[0])[y * s0 + x];
            red += (unsigned long long)v + ((unsigned long long)(v * v) << 32);   // low word sum, high word sum of squares
            break;
        }
        case X265HIP_OP_SSIM_DIST:                                                // ssimDist_c (pixel.cpp:958-981): two 64-bit sums
        {
            const int f = ((const Px*)a.p[0].base + jb.off[0])[y * s0 + x];
            const int d = f - (int)((const Px*)a.p[1].base + jb.off[1])[y * s1 + x];
            const unsigned t = (unsigned)f >> jb.arg[0];
            red += (unsigned long long)(long long)(d * d);
            red2 += (unsigned long long)(t * t);
            break;
        }
        case X265HIP_OP_NORM_FACT:                                                // normFact_c (pixel.cpp:983-995)
        {
            const unsigned t = (unsigned)((const Px*)a.p[0].base + jb.off[0])[y * s0 + x] >> jb.arg[0];
            red += (unsigned long long)(t * t);
            break;
        }
        default: break;
        }
    }
    if (OP == X265HIP_OP_SSIM_DIST)
    {
        __shared__ unsigned long long scratch3[4];
        const unsigned long long ss = block_sum<unsigned long long>(red, scratch);
        const unsigned long long ac = block_sum<unsigned long long>(red2, scratch3);
        if (tid == 0) { a.result[2 * blockIdx.x] = ss; a.result[2 * blockIdx.x + 1] = ac; }
    }
    if (OP == X265HIP_OP_SSE_SS || OP == X265HIP_OP_SSD_S || OP == X265HIP_OP_VAR || OP == X265HIP_OP_NORM_FACT)
    {
        // VAR packs two independent 32-bit accumulators; carries out of the low word must not leak
        // into the high word (the reference keeps two uint32_t), so reduce the halves separately.
        if (OP == X265HIP_OP_VAR)
        {
            __shared__ unsigned long long scratch2[4];
            const unsigned long long lo = block_sum<unsigned long long>(red & 0xffffffffull, scratch);
            const unsigned long long hi = block_sum<unsigned long long>(red >> 32, scratch2);
            if (tid == 0) a.result[blockIdx.x] = (lo & 0xffffffffull) | ((hi & 0xffffffffull) << 32);
        }
        else
        {
            const unsigned long long tot = block_sum<unsigned long long>(red, scratch);
            if (tid == 0) a.result[blockIdx.x] = tot;
        }
    }
}

// Fast path for the streaming ops when the width is a multiple of 4: a thread moves 4 samples per step with dword
// accesses (unaligned addresses are fine on gfx950), and small blocks share a 256-thread workgroup.
template <typename T> __device__ __forceinline__ void ld4(const T* p, int (&v)[4])
{
    const uint8_t* b = reinterpret_cast<const uint8_t*>(p);
    if (sizeof(T) == 1)
    {
        const uint32_t w = ld_u32(b);
        v[0] = w & 0xff; v[1] = (w >> 8) & 0xff; v[2] = (w >> 16) & 0xff; v[3] = w >> 24;
    }
    else
    {
        const uint32_t w0 = ld_u32(b), w1 = ld_u32(b + 4);
        const bool sgn = std::is_signed<T>::value;
        v[0] = sgn ? (int)(int16_t)w0 : (int)(w0 & 0xffff); v[1] = sgn ? (int)w0 >> 16 : (int)(w0 >> 16);
        v[2] = sgn ? (int)(int16_t)w1 : (int)(w1 & 0xffff); v[3] = sgn ? (int)w1 >> 16 : (int)(w1 >> 16);
    }
}
template <typename T> __device__ __forceinline__ void st4(T* p, const int (&v)[4])
{
    uint8_t* b = reinterpret_cast<uint8_t*>(p);
    if (sizeof(T) == 1)
        *reinterpret_cast<u32_unaligned*>(b) = (uint32_t)(v[0] & 0xff) | ((uint32_t)(v[1] & 0xff) << 8) | ((uint32_t)(v[2] & 0xff) << 16) | ((uint32_t)v[3] << 24);
    else
    {
        reinterpret_cast<u32_unaligned*>(b)[0] = ((uint32_t)v[0] & 0xffffu) | ((uint32_t)v[1] << 16);
        reinterpret_cast<u32_unaligned*>(b)[1] = ((uint32_t)v[2] & 0xffffu) | ((uint32_t)v[3] << 16);
    }
}

// An element-wise block operation on 4 consecutive samples at (x, y) of the job's block, in two halves so that a thread can have the
// loads of several quads in flight before its first store (source and destination may alias as far as the compiler knows)
template <typename Px, int OP>
__device__ __forceinline__ void blockop_load(const OpArgs& a, const x265hip_job& jb, int x, int y, int (&u)[4], int (&v)[4])
{
    const long s1 = a.p[1].stride, s2 = a.p[2].stride;
    switch (OP)
    {
    case X265HIP_OP_COPY_PP: case X265HIP_OP_COPY_PS: ld4((const Px*)a.p[1].base + jb.off[1] + y * s1 + x, u); break;
    case X265HIP_OP_COPY_SP: case X265HIP_OP_COPY_SS: ld4((const int16_t*)a.p[1].base + jb.off[1] + y * s1 + x, u); break;
    case X265HIP_OP_SUB_PS: case X265HIP_OP_PIXELAVG:
        ld4((const Px*)a.p[1].base + jb.off[1] + y * s1 + x, u); ld4((const Px*)a.p[2].base + jb.off[2] + y * s2 + x, v); break;
    case X265HIP_OP_ADD_PS:
        ld4((const Px*)a.p[1].base + jb.off[1] + y * s1 + x, u); ld4((const int16_t*)a.p[2].base + jb.off[2] + y * s2 + x, v); break;
    case X265HIP_OP_ADDAVG:
        ld4((const int16_t*)a.p[1].base + jb.off[1] + y * s1 + x, u); ld4((const int16_t*)a.p[2].base + jb.off[2] + y * s2 + x, v); break;
    default: break;
    }
}

template <typename Px, int OP>
__device__ __forceinline__ void blockop_store(const OpArgs& a, const x265hip_job& jb, int x, int y, int maxVal, const int (&u)[4], const int (&v)[4])
{
    const long s0 = a.p[0].stride;
    int r[4];
    switch (OP)
    {
    case X265HIP_OP_COPY_PP: case X265HIP_OP_COPY_SP: st4((Px*)a.p[0].base + jb.off[0] + y * s0 + x, u); break;
    case X265HIP_OP_COPY_PS: case X265HIP_OP_COPY_SS: st4((int16_t*)a.p[0].base + jb.off[0] + y * s0 + x, u); break;
    case X265HIP_OP_SUB_PS:
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = u[k] - v[k];
        st4((int16_t*)a.p[0].base + jb.off[0] + y * s0 + x, r); break;
    case X265HIP_OP_ADD_PS:
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = clip3(0, maxVal, u[k] + v[k]);
        st4((Px*)a.p[0].base + jb.off[0] + y * s0 + x, r); break;
    case X265HIP_OP_ADDAVG:
    {
        const int shift = 14 + 1 - a.depth, offset = (1 << (shift - 1)) + 2 * 8192;
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = clip3(0, maxVal, (u[k] + v[k] + offset) >> shift);
        st4((Px*)a.p[0].base + jb.off[0] + y * s0 + x, r); break;
    }
    case X265HIP_OP_PIXELAVG:
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = (u[k] + v[k] + 1) >> 1;
        st4((Px*)a.p[0].base + jb.off[0] + y * s0 + x, r); break;
    default: break;
    }
}

// QPT quads (of 4 samples) per thread and step: 4 for blocks whose width is a multiple of 16 (16 consecutive samples per thread, the
// four quads' loads in flight together), else 1
template <typename Px, int OP, int QPT>
__global__ void __launch_bounds__(256) blockop_quad_kernel(OpArgs a, int njobs, int threadsPerJob, int jobsPerWg)
{
    const int jw = threadIdx.x / threadsPerJob, t = threadIdx.x - jw * threadsPerJob;
    const int job = blockIdx.x * jobsPerWg + jw;
    if (jw >= jobsPerWg || job >= njobs) return;
    const x265hip_job jb = a.jobs[job];
    const int gpr = (a.w >> 2) / QPT, ng = gpr * a.h;          // groups of QPT quads per row / per block
    const int maxVal = (1 << a.depth) - 1;
    for (int g = t; g < ng; g += threadsPerJob)
    {
        const int y = g / gpr, x = (g - y * gpr) * 4 * QPT;
        int u[QPT][4], v[QPT][4];
#pragma unroll
        for (int k = 0; k < QPT; k++) blockop_load<Px, OP>(a, jb, x + 4 * k, y, u[k], v[k]);
#pragma unroll
        for (int k = 0; k < QPT; k++) blockop_store<Px, OP>(a, jb, x + 4 * k, y, maxVal, u[k], v[k]);
    }
}

template <typename Px, int OP> static void launch_quad(const OpArgs& a, int njobs, hipStream_t s)
{
    const bool wide = (a.w & 15) == 0;
    const int ng = (a.w >> 2) * a.h / (wide ? 4 : 1);
    int tpj = 1;
    while (tpj < ng && tpj < 256) tpj <<= 1;                   // threads per job: power of two, <= 256
    const int jpw = 256 / tpj;
    if (wide) hipLaunchKernelGGL((blockop_quad_kernel<Px, OP, 4>), dim3((njobs + jpw - 1) / jpw), dim3(256), 0, s, a, njobs, tpj, jpw);
    else hipLaunchKernelGGL((blockop_quad_kernel<Px, OP, 1>), dim3((njobs + jpw - 1) / jpw), dim3(256), 0, s, a, njobs, tpj, jpw);
}

template <typename Px> static int launch_op(int op, const OpArgs& a, int njobs, hipStream_t s)
{
    const int threads = (a.w * a.h <= 256 && op < X265HIP_OP_SCALE1D_128TO64) ? 64 : 256;
    static const bool forceGeneric = getenv("X265HIP_BLOCKOP_GENERIC") != nullptr;   // A/B switch, read once
    if ((a.w & 3) == 0 && !forceGeneric)
    {
#define QUAD(K) case K: launch_quad<Px, K>(a, njobs, s); X265HIP_TRY(hipGetLastError()); return 0;
        switch (op)
        {
            QUAD(X265HIP_OP_COPY_PP) QUAD(X265HIP_OP_COPY_PS) QUAD(X265HIP_OP_COPY_SP) QUAD(X265HIP_OP_COPY_SS) QUAD(X265HIP_OP_SUB_PS)
            QUAD(X265HIP_OP_ADD_PS) QUAD(X265HIP_OP_ADDAVG) QUAD(X265HIP_OP_PIXELAVG)
        default: break;
        }
#undef QUAD
    }
#define CASE(K) case K: hipLaunchKernelGGL((blockop_kernel<Px, K>), dim3(njobs), dim3(threads), 0, s, a); break;
    switch (op)
    {
        CASE(X265HIP_OP_COPY_PP) CASE(X265HIP_OP_COPY_PS) CASE(X265HIP_OP_COPY_SP) CASE(X265HIP_OP_COPY_SS) CASE(X265HIP_OP_SUB_PS)
        CASE(X265HIP_OP_ADD_PS) CASE(X265HIP_OP_ADDAVG) CASE(X265HIP_OP_PIXELAVG) CASE(X265HIP_OP_BLOCKFILL)
        CASE(X265HIP_OP_CPY2DTO1D_SHL) CASE(X265HIP_OP_CPY2DTO1D_SHR) CASE(X265HIP_OP_CPY1DTO2D_SHL) CASE(X265HIP_OP_CPY1DTO2D_SHR)
        CASE(X265HIP_OP_TRANSPOSE) CASE(X265HIP_OP_WEIGHT_PP) CASE(X265HIP_OP_WEIGHT_SP) CASE(X265HIP_OP_SCALE1D_128TO64)
        CASE(X265HIP_OP_SCALE2D_64TO32) CASE(X265HIP_OP_SSE_SS) CASE(X265HIP_OP_SSD_S) CASE(X265HIP_OP_VAR)
        CASE(X265HIP_OP_SSIM_DIST) CASE(X265HIP_OP_NORM_FACT)
    default: set_error("blockop_batch: unknown op %d", op); return X265HIP_EINVAL;
    }
#undef CASE
    X265HIP_TRY(hipGetLastError());
    return 0;
}

} // namespace x265hip

using namespace x265hip;

extern "C" int x265hip_quant_batch(int kind, const x265hip_plane planes[4], const x265hip_job* jobs, int njobs,
                                   uint32_t* result, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!planes || !jobs || njobs < 0) { set_error("quant_batch: NULL operand"); return X265HIP_EINVAL; }
    if (njobs == 0) return 0;
    QArgs a;
    for (int i = 0; i < 4; i++) a.p[i] = planes[i];
    a.jobs = jobs; a.result = result;
    hipStream_t s = (hipStream_t)stream;
#define CASE(K) case K: hipLaunchKernelGGL((quant_kernel<K>), dim3(njobs), dim3(256), 0, s, a); break;
    switch (kind)
    {
        CASE(X265HIP_Q_QUANT) CASE(X265HIP_Q_NQUANT) CASE(X265HIP_Q_DEQUANT_NORMAL) CASE(X265HIP_Q_DEQUANT_SCALING)
        CASE(X265HIP_Q_DENOISE) CASE(X265HIP_Q_COUNT_NONZERO) CASE(X265HIP_Q_COPY_CNT)
    default: set_error("quant_batch: unknown kind %d", kind); return X265HIP_EINVAL;
    }
#undef CASE
    X265HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int x265hip_blockop_batch(int op, int depth, int w, int h, const x265hip_plane planes[3],
                                     const x265hip_job* jobs, int njobs, uint64_t* result, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!planes || !jobs || njobs < 0) { set_error("blockop_batch: NULL operand"); return X265HIP_EINVAL; }
    if (njobs == 0) return 0;
    if (depth != 8 && depth != 10 && depth != 12) { set_error("blockop_batch: depth %d", depth); return X265HIP_EINVAL; }
    const bool fixedSize = op == X265HIP_OP_SCALE1D_128TO64 || op == X265HIP_OP_SCALE2D_64TO32;
    if (!fixedSize && (w < 1 || h < 1 || w > 64 * 64 || h > 4096)) { set_error("blockop_batch: block %dx%d unsupported", w, h); return X265HIP_EINVAL; }
    if ((op == X265HIP_OP_SSE_SS || op == X265HIP_OP_SSD_S || op == X265HIP_OP_VAR || op == X265HIP_OP_SSIM_DIST || op == X265HIP_OP_NORM_FACT) && !result) { set_error("blockop_batch: reduction needs result"); return X265HIP_EINVAL; }
    OpArgs a;
    for (int i = 0; i < 3; i++) a.p[i] = planes[i];
    a.jobs = jobs; a.result = (unsigned long long*)result; a.w = w; a.h = h; a.depth = depth;
    if (depth == 8) return launch_op<uint8_t>(op, a, njobs, (hipStream_t)stream);
    return launch_op<uint16_t>(op, a, njobs, (hipStream_t)stream);
}
