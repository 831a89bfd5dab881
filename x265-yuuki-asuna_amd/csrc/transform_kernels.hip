// transform_kernels.hip - batched HEVC integer transforms on gfx950.
//
// Reference semantics (source/common/dct.cpp): forward dct4/8/16/32_c :459-525 = two passes of
//   out[k][j] = (int16)((sum_i M[k][i] * in[j][i] + add) >> shift)   (partialButterflyN :83-240,:418-440 are
//   an exact refactoring of this product; the store TRUNCATES to int16 without clipping), shifts
//   log2N-1+depth-8 then log2N+6; inverse idct* :544-610 = two passes of
//   out[j][k] = clip16((sum_i M[i][k] * in[i][j] + add) >> shift), shifts 7 then 12-(depth-8);
//   DST-VII 4x4 :43-81,:442-457,:527-542; lowpass approximations lowpassdct.cpp:34-113.
// M is the standard's 32-point matrix (constants.cpp:270-344), generated here from its 31 distinct
// magnitudes by the cosine symmetries.
//
// Two implementations, bit-identical:
//   * VALU: one thread per output coefficient, operands in LDS (all sizes; default for 4x4 / 8x8);
//   * MFMA: 16x16 / 32x32 as int8 matrix products on the matrix cores.  int16 inputs are biased to
//     unsigned and split into two 8-bit limbs re-centred to signed (x = 256*hi + lo + 128 with
//     hi, lo in [-128,127]); M * x = 256 * (M*hi) + (M*lo) + 128 * rowsum(M), every partial sum fits
//     int32 exactly, so the rounding shift sees the same integer as the reference.
#include "common.h"
#include "mfma_dct.h"

#include <cstdlib>

namespace x265hip {

struct DctMat
{
    int8_t m[32][32];
    constexpr DctMat() : m{}
    {
        constexpr int basis[33] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67,
                                    64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };
        for (int k = 0; k < 32; k++)
            for (int n = 0; n < 32; n++)
            {
                const int q = (k * (2 * n + 1)) & 127;
                int v = 0;
                if (q <= 32) v = basis[q];
                else if (q <= 64) v = -basis[64 - q];
                else if (q <= 96) v = -basis[q - 64];
                else v = basis[128 - q];
                m[k][n] = (int8_t)v;
            }
    }
};
__constant__ DctMat kT = DctMat();
// column sums of the N-point matrix (the data-independent bias of the inverse passes' accumulators): sum_i M_N[i][col]
struct DctColSums
{
    int s16[16], s32[32];
    constexpr DctColSums() : s16{}, s32{}
    {
        constexpr DctMat t = DctMat();
        for (int c = 0; c < 32; c++) { int v = 0; for (int i = 0; i < 32; i++) v += t.m[i][c]; s32[c] = v; }
        for (int c = 0; c < 16; c++) { int v = 0; for (int i = 0; i < 16; i++) v += t.m[2 * i][c]; s16[c] = v; }
    }
};
__constant__ DctColSums kTColSum = DctColSums();
__constant__ int8_t kDst[4][4] = { { 29, 55, 74, 84 }, { 74, 74, 0, -74 }, { 84, -29, -74, 55 }, { 55, -84, 74, -29 } };

struct TrArgs
{
    const int16_t* src; long srcStride;
    int16_t* dst; long dstStride;
    const x265hip_job* jobs;
    int njobs, depth;
};

__device__ __forceinline__ int clip16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

// coefficient (row k, column n) of the N-point matrix; DST uses its own 4x4 matrix
template <int N, bool DST> __device__ __forceinline__ int coef(int k, int n) { return DST ? kDst[k][n] : kT.m[k * (32 / N)][n]; }

// KIND: 0 forward, 1 inverse.  JPB jobs per block (small transforms share a workgroup).
template <int N, int KIND, bool DST>
__global__ void __launch_bounds__(256) transform_valu_kernel(TrArgs a)
{
    constexpr int NN = N * N;
    constexpr int JPB = NN >= 256 ? 1 : 256 / NN;
    constexpr int LOG2N = N == 4 ? 2 : (N == 8 ? 3 : (N == 16 ? 4 : 5));
    __shared__ int16_t bufA[JPB * NN], bufB[JPB * NN];
    const int tid = threadIdx.x;
    const int jl = NN >= 256 ? 0 : tid / NN;                 // job slot inside the block
    const int e0 = NN >= 256 ? tid : tid - jl * NN;          // first element handled by this thread
    const long job = (long)blockIdx.x * JPB + jl;
    const bool live = job < a.njobs;
    const x265hip_job jb = a.jobs[live ? job : 0];
    const int16_t* src = a.src + jb.off[0];
    int16_t* dst = a.dst + jb.off[1];
    int16_t* A = bufA + jl * NN;
    int16_t* B = bufB + jl * NN;

    if (KIND == 0)
    {
        for (int e = e0; e < NN; e += 256)
        {
            const int y = e / N, x = e % N;
            A[e] = live ? src[(long)y * a.srcStride + x] : (int16_t)0;
        }
        __syncthreads();
        const int sh1 = (DST ? 1 : LOG2N - 1) + a.depth - 8, sh2 = DST ? 8 : LOG2N + 6;
        for (int e = e0; e < NN; e += 256)
        {
            const int k = e / N, j = e % N;
            int acc = 0;
#pragma unroll
            for (int i = 0; i < N; i++) acc += coef<N, DST>(k, i) * (int)A[j * N + i];
            B[k * N + j] = (int16_t)((acc + (1 << (sh1 - 1))) >> sh1);
        }
        __syncthreads();
        for (int e = e0; e < NN; e += 256)
        {
            const int k = e / N, j = e % N;
            int acc = 0;
#pragma unroll
            for (int i = 0; i < N; i++) acc += coef<N, DST>(k, i) * (int)B[j * N + i];
            if (live) dst[k * N + j] = (int16_t)((acc + (1 << (sh2 - 1))) >> sh2);
        }
    }
    else
    {
        for (int e = e0; e < NN; e += 256) A[e] = live ? src[e] : (int16_t)0;
        __syncthreads();
        const int sh2 = 12 - (a.depth - 8);
        for (int e = e0; e < NN; e += 256)
        {
            const int j = e / N, k = e % N;
            int acc = 0;
#pragma unroll
            for (int i = 0; i < N; i++) acc += coef<N, DST>(i, k) * (int)A[i * N + j];
            B[j * N + k] = (int16_t)clip16((acc + 64) >> 7);
        }
        __syncthreads();
        for (int e = e0; e < NN; e += 256)
        {
            const int j = e / N, k = e % N;
            int acc = 0;
#pragma unroll
            for (int i = 0; i < N; i++) acc += coef<N, DST>(i, k) * (int)B[i * N + j];
            if (live) dst[(long)j * a.dstStride + k] = (int16_t)clip16((acc + (1 << (sh2 - 1))) >> sh2);
        }
    }
}

// lowpass_dct (lowpassdct.cpp:34-113): 2x2 average -> half-size DCT into the top-left quadrant,
// zeros elsewhere, DC replaced by a scaled sum of the (int16-truncated) 2x2 sums.
template <int N>
__global__ void __launch_bounds__(256) lowpass_kernel(TrArgs a)
{
    constexpr int H = N / 2, HH = H * H;
    constexpr int LOG2H = H == 4 ? 2 : (H == 8 ? 3 : 4);
    __shared__ int16_t A[HH], B[HH];
    __shared__ int total;
    const int tid = threadIdx.x;
    const x265hip_job jb = a.jobs[blockIdx.x];
    const int16_t* src = a.src + jb.off[0];
    int16_t* dst = a.dst + jb.off[1];
    if (tid == 0) total = 0;
    __syncthreads();
    int part = 0;
    for (int e = tid; e < HH; e += 256)
    {
        const int i = e / H, j = e % H;
        const int16_t* p = src + (long)(2 * i) * a.srcStride + 2 * j;
        const int16_t s4 = (int16_t)((int)p[0] + p[1] + p[a.srcStride] + p[a.srcStride + 1]);
        A[e] = (int16_t)(s4 >> 2);
        part += s4;
    }
    atomicAdd(&total, part);
    for (int e = tid; e < N * N; e += 256) dst[e] = 0;
    __syncthreads();
    const int sh1 = LOG2H - 1 + a.depth - 8, sh2 = LOG2H + 6;
    for (int e = tid; e < HH; e += 256)
    {
        const int k = e / H, j = e % H;
        int acc = 0;
#pragma unroll
        for (int i = 0; i < H; i++) acc += coef<H, false>(k, i) * (int)A[j * H + i];
        B[k * H + j] = (int16_t)((acc + (1 << (sh1 - 1))) >> sh1);
    }
    __syncthreads();
    for (int e = tid; e < HH; e += 256)
    {
        const int k = e / H, j = e % H;
        int acc = 0;
#pragma unroll
        for (int i = 0; i < H; i++) acc += coef<H, false>(k, i) * (int)B[j * H + i];
        int16_t v = (int16_t)((acc + (1 << (sh2 - 1))) >> sh2);
        if (e == 0)
        {
            const int t32 = total;
            if (N == 8) v = (int16_t)((int)(int16_t)t32 << 1);       // int16_t running sum in the reference
            else if (N == 16) v = (int16_t)(t32 >> 1);
            else v = (int16_t)(t32 >> 3);
        }
        dst[k * N + j] = v;
    }
}

// ------------------------------------------------------------------------------------ MFMA path
// One wavefront per TU.  N = 32: one v_mfma_i32_32x32x32_i8 per limb; N = 16: v_mfma_i32_16x16x64_i8 would
// need K = 64, so the 16-point product is zero-padded to K = 32 inside the 32x32x32 shape?  No - the
// 16x16 case uses v_mfma_i32_16x16x64_i8 with the K range [16,64) fed zeros.
// P = Mop x X where Mop is the transform matrix (FWD: M[k][i]; INV: M^T, i.e. Mop[k][i] = M[i][k]) and
// X[i][col] comes from LDS as int16 via xsrc(i, col).  Returns the exact int32 products in C/D layout.
template <int N, bool INV, typename XF>
__device__ __forceinline__ typename Mfma<N>::Acc mfma_stage(int lane, XF xsrc)
{
    typedef Mfma<N> MF;
    const int kb = MF::kbase(lane), rn = MF::mn(lane);
    int aw[4], hw[4], lw[4];
#pragma unroll
    for (int q = 0; q < 4; q++)
    {
        int ab[4], hb[4], lb[4];
#pragma unroll
        for (int t = 0; t < 4; t++)
        {
            const int k = kb + 4 * q + t;
            if (k < N)
            {
                ab[t] = INV ? kT.m[k * (32 / N)][rn] : kT.m[rn * (32 / N)][k];
                const int u = (int)xsrc(k, rn) + 32768;           // bias to unsigned 16 bit
                hb[t] = (u >> 8) - 128;
                lb[t] = (u & 0xff) - 128;
            }
            else { ab[t] = 0; hb[t] = 0; lb[t] = 0; }
        }
        aw[q] = pack4(ab[0], ab[1], ab[2], ab[3]);
        hw[q] = pack4(hb[0], hb[1], hb[2], hb[3]);
        lw[q] = pack4(lb[0], lb[1], lb[2], lb[3]);
    }
    const v4i A = { aw[0], aw[1], aw[2], aw[3] };
    const v4i Bh = { hw[0], hw[1], hw[2], hw[3] };
    const v4i Bl = { lw[0], lw[1], lw[2], lw[3] };
    typename MF::Acc zero = {};
    typename MF::Acc ph = MF::run(A, Bh, zero);
    typename MF::Acc pl = MF::run(A, Bl, zero);
    typename MF::Acc out;
#pragma unroll
    for (int r = 0; r < MF::NACC; r++)
    {
        // + 128 * sum_i Mop[row][i]: only the DC basis row (forward) / is data independent; computed exactly
        const int rowIdx = MF::row(lane, r);
        int rs = 0;
        if (!INV) rs = rowIdx == 0 ? 64 * N : 0;                    // rows 1.. of the DCT matrix sum to zero
        out[r] = ph[r] * 256 + pl[r] + 128 * rs;
    }
    return out;
}

// column sums of M (needed for the inverse: Mop = M^T so "row sums" of Mop are column sums of M)
__device__ __forceinline__ int col_sum(int N, int k)
{
    int s = 0;
    for (int i = 0; i < N; i++) s += kT.m[i * (32 / N)][k];
    return s;
}

template <int N, int KIND>
__global__ void __launch_bounds__(64) transform_mfma_kernel(TrArgs a)
{
    typedef Mfma<N> MF;
    constexpr int NN = N * N;
    constexpr int LOG2N = N == 16 ? 4 : 5;
    __shared__ int16_t A[NN], B[NN];
    __shared__ int csum[32];
    const int lane = threadIdx.x;
    const x265hip_job jb = a.jobs[blockIdx.x];
    const int16_t* src = a.src + jb.off[0];
    int16_t* dst = a.dst + jb.off[1];
    if (KIND == 1 && lane < N) csum[lane] = col_sum(N, lane);

    if (KIND == 0)
    {
        for (int e = lane; e < NN; e += 64) A[e] = src[(long)(e / N) * a.srcStride + (e % N)];
        __syncthreads();
        const int sh1 = LOG2N - 1 + a.depth - 8, sh2 = LOG2N + 6;
        // stage 1: P[k][j] = sum_i M[k][i] * in[j][i]  -> X[i][col=j] = A[j*N + i]
        typename MF::Acc p = mfma_stage<N, false>(lane, [&](int i, int j) { return A[j * N + i]; });
#pragma unroll
        for (int r = 0; r < MF::NACC; r++)
            B[MF::row(lane, r) * N + MF::col(lane)] = (int16_t)((p[r] + (1 << (sh1 - 1))) >> sh1);
        __syncthreads();
        p = mfma_stage<N, false>(lane, [&](int i, int j) { return B[j * N + i]; });
#pragma unroll
        for (int r = 0; r < MF::NACC; r++)
            dst[MF::row(lane, r) * N + MF::col(lane)] = (int16_t)((p[r] + (1 << (sh2 - 1))) >> sh2);
    }
    else
    {
        for (int e = lane; e < NN; e += 64) A[e] = src[e];
        __syncthreads();
        const int sh2 = 12 - (a.depth - 8);
        // stage 1: P[k][j] = sum_i M[i][k] * in[i][j]; stored transposed: out[j][k]
        typename MF::Acc p = mfma_stage<N, true>(lane, [&](int i, int j) { return A[i * N + j]; });
#pragma unroll
        for (int r = 0; r < MF::NACC; r++)
        {
            const int k = MF::row(lane, r), j = MF::col(lane);
            B[j * N + k] = (int16_t)clip16((p[r] + 128 * csum[k] + 64) >> 7);
        }
        __syncthreads();
        p = mfma_stage<N, true>(lane, [&](int i, int j) { return B[i * N + j]; });
#pragma unroll
        for (int r = 0; r < MF::NACC; r++)
        {
            const int k = MF::row(lane, r), j = MF::col(lane);
            dst[(long)j * a.dstStride + k] = (int16_t)clip16((p[r] + 128 * csum[k] + (1 << (sh2 - 1))) >> sh2);
        }
    }
}

template <int N, int KIND, bool DST> static int launch_valu(const TrArgs& a, hipStream_t s)
{
    constexpr int JPB = N * N >= 256 ? 1 : 256 / (N * N);
    hipLaunchKernelGGL((transform_valu_kernel<N, KIND, DST>), dim3((a.njobs + JPB - 1) / JPB), dim3(256), 0, s, a);
    X265HIP_TRY(hipGetLastError());
    return 0;
}
// ------------------------------------------------------------------------------------ MFMA path, streaming form
// Same arithmetic as transform_mfma_kernel (kept above as the readable reference of the limb identities), organised
// for throughput: a wavefront loops over TUs with the transform-matrix fragment and the bias constants in
// registers; forward stage-1 operands are the TU's rows read straight from global memory (16 consecutive int16 per
// lane), the limb split is two v_perm_b32 + one v_xor per four samples (x = 256 * hi + lo with hi the signed high
// byte as it stands and lo - 128 the low byte with its top bit flipped), and the only LDS traffic is the 32 x 32
// transpose between the two passes (and in front of the inverse's first pass, whose contraction runs down columns).
template <int N, int KIND>
__global__ void __launch_bounds__(256) transform_mfma_stream_kernel(TrArgs a)
{
    typedef Mfma<N> MF;
    constexpr int NN = N * N;
    constexpr int LOG2N = N == 16 ? 4 : 5;
    constexpr bool INV = KIND == 1;
    __shared__ __attribute__((aligned(16))) int16_t lds[4][NN];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int16_t* T = lds[wave];
    const int kb = MF::kbase(lane), rn = MF::mn(lane);
    const bool kvalid = kb < N;                                    // 16x16x64: only the first K block is real

    // transform-matrix fragment (the A operand of both passes) and the data-independent bias of each accumulator row
    v4i Afrag = { 0, 0, 0, 0 };
    if (kvalid)
    {
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            int b[4];
#pragma unroll
            for (int t = 0; t < 4; t++)
            {
                const int k = kb + 4 * q + t;
                b[t] = INV ? kT.m[k * (32 / N)][rn] : kT.m[rn * (32 / N)][k];
            }
            Afrag[q] = pack4(b[0], b[1], b[2], b[3]);
        }
    }
    int bias[MF::NACC];
#pragma unroll
    for (int r = 0; r < MF::NACC; r++)
    {
        const int row = MF::row(lane, r);
        int rs = 0;
        if (INV) rs = N == 16 ? kTColSum.s16[row & 15] : kTColSum.s32[row];          // (a table: the sum itself was N loads per accumulator element and wavefront)
        else rs = row == 0 ? 64 * N : 0;
        bias[r] = 128 * rs;
    }
    const int shF1 = LOG2N - 1 + a.depth - 8, shF2 = LOG2N + 6, shI2 = 12 - (a.depth - 8);

    auto product = [&](const uint32_t (&d)[8], int (&out)[MF::NACC])
    {
        v4i hi = { 0, 0, 0, 0 }, lo = { 0, 0, 0, 0 };
        if (kvalid) split_limbs(d, hi, lo);
        typename MF::Acc zero = {};
        const typename MF::Acc ph = MF::run(Afrag, hi, zero);
        const typename MF::Acc pl = MF::run(Afrag, lo, zero);
#pragma unroll
        for (int r = 0; r < MF::NACC; r++) out[r] = ph[r] * 256 + pl[r] + bias[r];
    };
    // read T[row rn][kb .. kb + 15] (one transposed operand row) as 8 dwords
    auto lds_row = [&](uint32_t (&d)[8])
    {
        if (kvalid)
        {
            const u32x4* p = reinterpret_cast<const u32x4*>(T + rn * N + kb);
            const u32x4 v0 = p[0], v1 = p[1];
            d[0] = v0.x; d[1] = v0.y; d[2] = v0.z; d[3] = v0.w; d[4] = v1.x; d[5] = v1.y; d[6] = v1.z; d[7] = v1.w;
        }
    };

    const int nwaves = gridDim.x * 4;
    for (int job = blockIdx.x * 4 + wave; job < a.njobs; job += nwaves)
    {
        const x265hip_job jb = a.jobs[job];
        const int16_t* src = a.src + jb.off[0];
        int16_t* dst = a.dst + jb.off[1];
        uint32_t d[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        int p[MF::NACC];
        if (!INV)
        {
            // pass 1: P[k][j] = sum_i M[k][i] * in[j][i]: lane (column j = rn) supplies in[j][kb .. kb + 15]
            if (kvalid)
            {
                const u32x4_a2* g = reinterpret_cast<const u32x4_a2*>(src + (long)rn * a.srcStride + kb);
                const u32x4_a2 v0 = g[0], v1 = g[1];
                d[0] = v0.x; d[1] = v0.y; d[2] = v0.z; d[3] = v0.w; d[4] = v1.x; d[5] = v1.y; d[6] = v1.z; d[7] = v1.w;
            }
            product(d, p);
            // T[k][j] = (int16)((P + add) >> shift): the operand rows of pass 2 (contraction over j)
#pragma unroll
            for (int r = 0; r < MF::NACC; r++)
                T[MF::row(lane, r) * N + MF::col(lane)] = (int16_t)((p[r] + (1 << (shF1 - 1))) >> shF1);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): this wave's LDS writes have landed
            lds_row(d);
            product(d, p);
#pragma unroll
            for (int r = 0; r < MF::NACC; r++)
                dst[MF::row(lane, r) * N + MF::col(lane)] = (int16_t)((p[r] + (1 << (shF2 - 1))) >> shF2);
        }
        else
        {
            // pass 1 contracts over the ROWS of the coefficient block: lane (column j = rn) gathers in[kb .. kb + 15][j];
            // across the lanes of a row group every one of the 16 loads is a contiguous 2 * N byte run
            if (kvalid)
            {
#pragma unroll
                for (int q = 0; q < 8; q++)
                {
                    const uint32_t e0 = (uint16_t)src[(kb + 2 * q) * N + rn], e1 = (uint16_t)src[(kb + 2 * q + 1) * N + rn];
                    d[q] = e0 | (e1 << 16);
                }
            }
            product(d, p);                                          // P[k][j], lane: column j, rows k
            __builtin_amdgcn_wave_barrier();
            // out1[j][k] = clip16((P + 64) >> 7); pass 2 contracts over j for fixed k: T[k][j]
#pragma unroll
            for (int r = 0; r < MF::NACC; r++)
                T[MF::row(lane, r) * N + MF::col(lane)] = (int16_t)clip16((p[r] + 64) >> 7);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            lds_row(d);
            product(d, p);                                          // P2[k2][k] = sum_j M[j][k2] * out1[j][k]: lane column k, rows k2
            // a lane's accumulator rows come in runs of 4 consecutive k: one 8-byte store per run
#pragma unroll
            for (int g = 0; g < MF::NACC / 4; g++)
            {
                int v[4];
#pragma unroll
                for (int t = 0; t < 4; t++) v[t] = clip16((p[4 * g + t] + (1 << (shI2 - 1))) >> shI2);
                uint8_t* dp = reinterpret_cast<uint8_t*>(dst + (long)MF::col(lane) * a.dstStride + MF::row(lane, 4 * g));
                reinterpret_cast<u32_unaligned*>(dp)[0] = ((uint32_t)v[0] & 0xffffu) | ((uint32_t)v[1] << 16);
                reinterpret_cast<u32_unaligned*>(dp)[1] = ((uint32_t)v[2] & 0xffffu) | ((uint32_t)v[3] << 16);
            }
        }
        __builtin_amdgcn_wave_barrier();                            // T is rewritten by the next TU
    }
}

// ------------------------------------------------------------------------------------ MFMA path, 16 point, FOUR TUs per instruction (round 6)
// v_mfma_i32_16x16x64_i8 on a 16 x 16 block uses a quarter of its K range (the stream kernel above feeds zeros to K 16 .. 63: 48 of 64 lanes
// supply nothing, 16 lanes load the whole block).  A 32 x 32 x 32 product whose matrix operand is BLOCK-DIAGONAL, diag(M16, M16), is four
// independent 16 x 16 products: with the data operand B = [[X0, X1], [X2, X3]] the result is [[M X0, M X1], [M X2, M X3]].  So a wavefront takes
// FOUR TUs per step: lane (h = lane >> 5, c = lane & 31) supplies row c & 15 of TU 2 h + (c >> 4) - every lane loads 32 bytes - and the accumulator
// rows 0 .. 15 / 16 .. 31 of a lane belong to TUs (c >> 4) and 2 + (c >> 4).  Same limbs, biases and rounding as the kernels above (bit-exact by
// construction: the zero blocks of the matrix operand add nothing); half of each instruction still multiplies zeros, none of its K range is idle.
template <int KIND>
__global__ void __launch_bounds__(256) transform_mfma_quad16_kernel(TrArgs a)
{
    typedef Mfma<32> MF;
    constexpr int N = 16, NN = 256, LOG2N = 4;
    constexpr bool INV = KIND == 1;
    __shared__ __attribute__((aligned(16))) int16_t lds[4][4 * NN];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int16_t* T = lds[wave];
    const int h = lane >> 5, c = lane & 31, b = c >> 4, j16 = c & 15;
    const int tl = 2 * h + b;                                      // the TU whose operand row this lane supplies

    v4i Afrag = { 0, 0, 0, 0 };                                    // diag(M, M)[row c][k 16 h .. 16 h + 15]: zero off the diagonal blocks
    if (b == h)
    {
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            int v[4];
#pragma unroll
            for (int t = 0; t < 4; t++)
            {
                const int k = 4 * q + t;
                v[t] = INV ? kT.m[k * 2][j16] : kT.m[j16 * 2][k];
            }
            Afrag[q] = pack4(v[0], v[1], v[2], v[3]);
        }
    }
    int bias[MF::NACC];
#pragma unroll
    for (int r = 0; r < MF::NACC; r++)
    {
        const int row = MF::row(lane, r) & 15;
        int rs = 0;
        if (INV) rs = kTColSum.s16[row];
        else rs = row == 0 ? 64 * N : 0;
        bias[r] = 128 * rs;
    }
    const int shF1 = LOG2N - 1 + a.depth - 8, shF2 = LOG2N + 6, shI2 = 12 - (a.depth - 8);
    auto product = [&](const uint32_t (&d)[8], int (&out)[MF::NACC])
    {
        v4i hi, lo;
        split_limbs(d, hi, lo);
        typename MF::Acc zero = {};
        const typename MF::Acc ph = MF::run(Afrag, hi, zero);
        const typename MF::Acc pl = MF::run(Afrag, lo, zero);
#pragma unroll
        for (int r = 0; r < MF::NACC; r++) out[r] = ph[r] * 256 + pl[r] + bias[r];
    };
    auto lds_row = [&](uint32_t (&d)[8])                            // T of TU tl, row j16: the operand row of the second pass
    {
        const u32x4* p = reinterpret_cast<const u32x4*>(T + tl * NN + j16 * N);
        const u32x4 v0 = p[0], v1 = p[1];
        d[0] = v0.x; d[1] = v0.y; d[2] = v0.z; d[3] = v0.w; d[4] = v1.x; d[5] = v1.y; d[6] = v1.z; d[7] = v1.w;
    };
    const int nquads = (a.njobs + 3) >> 2;
    const int nwaves = gridDim.x * 4;
    for (int qd = blockIdx.x * 4 + wave; qd < nquads; qd += nwaves)
    {
        const int j0 = qd * 4;
        // the TU this lane reads, and the two TUs its accumulator rows 0 .. 15 / 16 .. 31 belong to (a missing TU of the last quad: zeros in, nothing out)
        const bool vl = j0 + tl < a.njobs, v0ok = j0 + b < a.njobs, v1ok = j0 + 2 + b < a.njobs;
        const int16_t* src = a.src + (vl ? a.jobs[j0 + tl].off[0] : 0);
        int16_t* dst0 = a.dst + (v0ok ? a.jobs[j0 + b].off[1] : 0);
        int16_t* dst1 = a.dst + (v1ok ? a.jobs[j0 + 2 + b].off[1] : 0);
        uint32_t d[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        int p[MF::NACC];
        if (!INV)
        {
            if (vl)
            {
                const u32x4_a2* g = reinterpret_cast<const u32x4_a2*>(src + (long)j16 * a.srcStride);
                const u32x4_a2 w0 = g[0], w1 = g[1];
                d[0] = w0.x; d[1] = w0.y; d[2] = w0.z; d[3] = w0.w; d[4] = w1.x; d[5] = w1.y; d[6] = w1.z; d[7] = w1.w;
            }
            product(d, p);
#pragma unroll
            for (int r = 0; r < MF::NACC; r++)
                T[(2 * (r >> 3) + b) * NN + (MF::row(lane, r) & 15) * N + j16] = (int16_t)((p[r] + (1 << (shF1 - 1))) >> shF1);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            lds_row(d);
            product(d, p);
#pragma unroll
            for (int r = 0; r < MF::NACC; r++)
            {
                const int v = (int16_t)((p[r] + (1 << (shF2 - 1))) >> shF2);
                if (r < 8) { if (v0ok) dst0[(MF::row(lane, r) & 15) * N + j16] = (int16_t)v; }
                else if (v1ok) dst1[(MF::row(lane, r) & 15) * N + j16] = (int16_t)v;
            }
        }
        else
        {
            // pass 1 contracts down the COLUMNS of the coefficient block.  The stream kernel gathers a lane's column with 16 two-byte global loads; here the TU comes in
            // row by row (every lane 32 contiguous bytes), goes through T as it is, and the lane takes its column from LDS - two global loads per lane instead of sixteen
            {
                u32x4 w0 = { 0, 0, 0, 0 }, w1 = { 0, 0, 0, 0 };
                if (vl)
                {
                    const u32x4_a2* g = reinterpret_cast<const u32x4_a2*>(src + j16 * N);
                    const u32x4_a2 g0 = g[0], g1 = g[1];
                    w0 = u32x4{ g0.x, g0.y, g0.z, g0.w }; w1 = u32x4{ g1.x, g1.y, g1.z, g1.w };
                }
                u32x4* tp = reinterpret_cast<u32x4*>(T + tl * NN + j16 * N);
                tp[0] = w0; tp[1] = w1;
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);
                const int16_t* col = T + tl * NN + j16;
#pragma unroll
                for (int q = 0; q < 8; q++)
                    d[q] = (uint32_t)(uint16_t)col[(2 * q) * N] | ((uint32_t)(uint16_t)col[(2 * q + 1) * N] << 16);
            }
            product(d, p);
            __builtin_amdgcn_wave_barrier();                        // (LDS operations of a wavefront execute in order: the column reads above are through)
#pragma unroll
            for (int r = 0; r < MF::NACC; r++)
                T[(2 * (r >> 3) + b) * NN + (MF::row(lane, r) & 15) * N + j16] = (int16_t)clip16((p[r] + 64) >> 7);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            lds_row(d);
            product(d, p);
            // A lane holds rows 4 h .. 4 h + 3 and 8 + 4 h .. of output row j16 of BOTH its TUs; lane ^ 32 holds the other rows.  One v_permlane32_swap per
            // dword hands the (2 + b) pieces of the lower half-wave to the upper one and the b pieces back: every lane then owns ONE whole output row
            // (32 contiguous bytes, two 16-byte stores) instead of four 8-byte pieces of two rows.
            uint32_t pk[8];
#pragma unroll
            for (int g = 0; g < 4; g++)
            {
                int v[4];
#pragma unroll
                for (int t = 0; t < 4; t++) v[t] = clip16((p[4 * g + t] + (1 << (shI2 - 1))) >> shI2);
                pk[2 * g] = ((uint32_t)v[0] & 0xffffu) | ((uint32_t)v[1] << 16);
                pk[2 * g + 1] = ((uint32_t)v[2] & 0xffffu) | ((uint32_t)v[3] << 16);
            }
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
            v2u sw[4];
#pragma unroll
            for (int i = 0; i < 4; i++) sw[i] = __builtin_amdgcn_permlane32_swap(pk[i], pk[4 + i], false, false);
            if (h ? v1ok : v0ok)
            {
                u32x4_a2* dp = reinterpret_cast<u32x4_a2*>((h ? dst1 : dst0) + (long)j16 * a.dstStride);
                dp[0] = u32x4_a2{ sw[0].x, sw[1].x, sw[0].y, sw[1].y };
                dp[1] = u32x4_a2{ sw[2].x, sw[3].x, sw[2].y, sw[3].y };
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int N, int KIND> static int launch_mfma(const TrArgs& a, hipStream_t s)
{
    static const bool simple = getenv("X265HIP_MFMA_SIMPLE") != nullptr;   // A/B switch (read once): one wavefront + one workgroup per TU
    static const bool quadOff = getenv("X265HIP_MFMA_QUAD16_OFF") != nullptr;   // A/B switch (read once): 16 point TUs one per 16x16x64 instruction, as before round 6
    if (simple)
        hipLaunchKernelGGL((transform_mfma_kernel<N, KIND>), dim3(a.njobs), dim3(64), 0, s, a);
    else if (N == 16 && !quadOff)
    {
        int wgs = ((a.njobs + 3) / 4 + 3) / 4;
        if (wgs > 256 * 8) wgs = 256 * 8;
        hipLaunchKernelGGL((transform_mfma_quad16_kernel<KIND>), dim3(wgs), dim3(256), 0, s, a);
    }
    else
    {
        int wgs = (a.njobs + 3) / 4;
        if (wgs > 256 * 8) wgs = 256 * 8;
        hipLaunchKernelGGL((transform_mfma_stream_kernel<N, KIND>), dim3(wgs), dim3(256), 0, s, a);
    }
    X265HIP_TRY(hipGetLastError());
    return 0;
}

} // namespace x265hip

using namespace x265hip;

extern "C" int x265hip_transform_batch(int kind, int depth, int n, x265hip_plane src, x265hip_plane dst,
                                       const x265hip_job* jobs, int njobs, int use_mfma, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!src.base || !dst.base || !jobs || njobs < 0) { set_error("transform_batch: NULL operand"); return X265HIP_EINVAL; }
    if (njobs == 0) return 0;
    if (depth != 8 && depth != 10 && depth != 12) { set_error("transform_batch: depth %d", depth); return X265HIP_EINVAL; }
    TrArgs a;
    a.src = (const int16_t*)src.base; a.srcStride = src.stride; a.dst = (int16_t*)dst.base; a.dstStride = dst.stride;
    a.jobs = jobs; a.njobs = njobs; a.depth = depth;
    hipStream_t s = (hipStream_t)stream;
    switch (kind)
    {
    case X265HIP_TR_DST4:  if (n != 4) break; return launch_valu<4, 0, true>(a, s);
    case X265HIP_TR_IDST4: if (n != 4) break; return launch_valu<4, 1, true>(a, s);
    case X265HIP_TR_DCT:
        if (n == 4) return launch_valu<4, 0, false>(a, s);
        if (n == 8) return launch_valu<8, 0, false>(a, s);
        if (n == 16) return use_mfma ? launch_mfma<16, 0>(a, s) : launch_valu<16, 0, false>(a, s);
        if (n == 32) return use_mfma ? launch_mfma<32, 0>(a, s) : launch_valu<32, 0, false>(a, s);
        break;
    case X265HIP_TR_IDCT:
        if (n == 4) return launch_valu<4, 1, false>(a, s);
        if (n == 8) return launch_valu<8, 1, false>(a, s);
        if (n == 16) return use_mfma ? launch_mfma<16, 1>(a, s) : launch_valu<16, 1, false>(a, s);
        if (n == 32) return use_mfma ? launch_mfma<32, 1>(a, s) : launch_valu<32, 1, false>(a, s);
        break;
    case X265HIP_TR_LOWPASS_DCT:
        if (n == 8) { hipLaunchKernelGGL((lowpass_kernel<8>), dim3(njobs), dim3(256), 0, s, a); X265HIP_TRY(hipGetLastError()); return 0; }
        if (n == 16) { hipLaunchKernelGGL((lowpass_kernel<16>), dim3(njobs), dim3(256), 0, s, a); X265HIP_TRY(hipGetLastError()); return 0; }
        if (n == 32) { hipLaunchKernelGGL((lowpass_kernel<32>), dim3(njobs), dim3(256), 0, s, a); X265HIP_TRY(hipGetLastError()); return 0; }
        break;
    default: break;
    }
    set_error("transform_batch: kind %d with n = %d is not a transform of the reference", kind, n);
    return X265HIP_EINVAL;
}
