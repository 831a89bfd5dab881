// mfma_dct.h - int8-limb MFMA building blocks of the HEVC 16 / 32 point transforms (shared by transform_kernels.hip and
// tu_kernels.hip).  See transform_kernels.hip for the arithmetic identities.
#pragma once
#include "common.h"

namespace x265hip {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// Operand fragment layouts (gfx950, 8-bit types): 32x32x32 - lane l supplies A[row = l & 31][k = 16*(l >> 5) + 0..15]
// and B[k = 16*(l >> 5) + 0..15][col = l & 31]; C/D: col = l & 31, row = (r & 3) + 8*(r >> 2) + 4*(l >> 5).
// 16x16x64 - A[row = l & 15][k = 16*(l >> 4) + 0..15], B[k = 16*(l >> 4) + 0..15][col = l & 15];
// C/D: col = l & 15, row = 4*(l >> 4) + r.
template <int N> struct Mfma;
template <> struct Mfma<32>
{
    typedef v16i Acc;
    static constexpr int NACC = 16;
    static __device__ __forceinline__ Acc run(v4i a, v4i b, Acc c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int lane, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
    static __device__ __forceinline__ int col(int lane) { return lane & 31; }
    static __device__ __forceinline__ int kbase(int lane) { return 16 * (lane >> 5); }
    static __device__ __forceinline__ int mn(int lane) { return lane & 31; }
};
template <> struct Mfma<16>
{
    typedef v4i Acc;
    static constexpr int NACC = 4;
    static __device__ __forceinline__ Acc run(v4i a, v4i b, Acc c) { return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }
    static __device__ __forceinline__ int col(int lane) { return lane & 15; }
    static __device__ __forceinline__ int kbase(int lane) { return 16 * (lane >> 4); }     // 0,16,32,48: only 0 is inside K = 16
    static __device__ __forceinline__ int mn(int lane) { return lane & 15; }
};

__device__ __forceinline__ int pack4(int b0, int b1, int b2, int b3)
{
    return (b0 & 0xff) | ((b1 & 0xff) << 8) | ((b2 & 0xff) << 16) | ((uint32_t)(b3 & 0xff) << 24);
}

typedef uint32_t __attribute__((ext_vector_type(4), aligned(2))) u32x4_a2;
typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;

// 16 consecutive int16 (8 dwords) -> high-byte and (low-byte - 128) fragments
__device__ __forceinline__ void split_limbs(const uint32_t (&d)[8], v4i& hi, v4i& lo)
{
#pragma unroll
    for (int q = 0; q < 4; q++)
    {
        hi[q] = (int)__builtin_amdgcn_perm(d[2 * q + 1], d[2 * q], 0x07050301u);
        lo[q] = (int)(__builtin_amdgcn_perm(d[2 * q + 1], d[2 * q], 0x06040200u) ^ 0x80808080u);
    }
}


// The transform-matrix operand of one pass and the data-independent bias of every accumulator element, for a lane.
// M is passed as a functor m(row, col) over the 32-point matrix.  FWD: Mop[k][i] = M_N[k][i]; INV: Mop[k][i] = M_N[i][k].
template <int N, bool INV>
struct DctOperand
{
    typedef Mfma<N> MF;
    v4i frag;
    int bias[MF::NACC];
    bool kvalid;
    template <typename MatF>
    __device__ __forceinline__ void init(int lane, MatF m)
    {
        init(lane, m, [&](int col) { int rs = 0; for (int i = 0; i < N; i++) rs += m(i * (32 / N), col); return rs; });
    }
    // colsum(c) = sum_i M_N[i][c]: callers that hold the sums in a table spare every wavefront N loads per accumulator element
    template <typename MatF, typename SumF>
    __device__ __forceinline__ void init(int lane, MatF m, SumF colsum)
    {
        const int kb = MF::kbase(lane), rn = MF::mn(lane);
        kvalid = kb < N;
        frag = v4i{ 0, 0, 0, 0 };
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
                    b[t] = INV ? m(k * (32 / N), rn) : m(rn * (32 / N), k);
                }
                frag[q] = pack4(b[0], b[1], b[2], b[3]);
            }
        }
#pragma unroll
        for (int r = 0; r < MF::NACC; r++)
        {
            const int row = MF::row(lane, r);
            int rs = 0;
            if (INV) rs = colsum(row);
            else rs = row == 0 ? 64 * N : 0;
            bias[r] = 128 * rs;
        }
    }
    // exact int32 products Mop x X for this lane's accumulator elements; d = 16 consecutive int16 of the operand row
    __device__ __forceinline__ void product(const uint32_t (&d)[8], int (&out)[MF::NACC]) const
    {
        v4i hi = { 0, 0, 0, 0 }, lo = { 0, 0, 0, 0 };
        if (kvalid) split_limbs(d, hi, lo);
        typename MF::Acc zero = {};
        const typename MF::Acc ph = MF::run(frag, hi, zero);
        const typename MF::Acc pl = MF::run(frag, lo, zero);
#pragma unroll
        for (int r = 0; r < MF::NACC; r++) out[r] = ph[r] * 256 + pl[r] + bias[r];
    }
};

} // namespace x265hip
