// tu_kernels.hip - fused inter prediction + residual coding round trip per square block, on gfx950.
//
// One launch performs, for every NxN block of every CTU (N = 8, 16, 32), the primitive sequence the
// reference issues from Predict::predInterLumaPixel (source/common/predict.cpp:245-265), Search's residual
// path (source/encoder/search.cpp:357-375), Quant::transformNxN non-RDOQ (source/common/quant.cpp:397-480)
// and Quant::invtransformNxN (:543-605):
//     luma_hvpp|hpp|vpp|copy_pp -> calcresidual -> dct -> quant (flat scaling) -> dequant_normal
//     -> DC-only blockfill shortcut | idct -> add_ps | copy_pp -> sse_pp
// with every intermediate held in LDS: per block HBM sees the source block, the (N+7)^2 reference patch,
// the quantised levels and the reconstructed block - nothing else (the primitive-by-primitive form moves
// ~9x that).  This is SURVEY section 8(f) item 2.  Arithmetic per primitive: ipfilter.cpp:79-369,
// dct.cpp:83-240,242-416,612-634,664-686, pixel.cpp:167-186,471-483,828-840.
#include "common.h"
#include "mfma_dct.h"
#include "intra_sample.h"

namespace x265hip {

struct DctMatrix
{
    int8_t m[32][32];
    constexpr DctMatrix() : m{}
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
static __constant__ DctMatrix kTu = DctMatrix();
// column sums of the 16 / 32 point matrices: the data-independent bias of the inverse passes' accumulators (DctOperand::init summed N matrix entries per
// accumulator element - 512 loads per lane at the start of every persistent wavefront of the 32 x 32 stage)
struct DctMatrixColSums
{
    int s16[16], s32[32];
    constexpr DctMatrixColSums() : s16{}, s32{}
    {
        constexpr DctMatrix t = DctMatrix();
        for (int c = 0; c < 32; c++) { int v = 0; for (int i = 0; i < 32; i++) v += t.m[i][c]; s32[c] = v; }
        for (int c = 0; c < 16; c++) { int v = 0; for (int i = 0; i < 16; i++) v += t.m[2 * i][c]; s16[c] = v; }
    }
};
static __constant__ DctMatrixColSums kTuColSum = DctMatrixColSums();
static __constant__ int8_t kTuDst[4][4] = { { 29, 55, 74, 84 }, { 74, 74, 0, -74 }, { 84, -29, -74, 55 }, { 55, -84, 74, -29 } };   // dct.cpp:43-81
static __constant__ int16_t kTuTaps[4][8] = {
    { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
    { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
static __constant__ int16_t kTuChromaTaps[8][4] = {                                      // constants.cpp:258-268
    { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 }, { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };
static __constant__ int kTuQuantScales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };   // scalinglist.cpp:129
static __constant__ int kTuInvQuantScales[6] = { 40, 45, 51, 57, 64, 72 };                  // scalinglist.cpp:130

// optional per-coefficient tables of the launch's block size (all DEVICE pointers, N * N entries, raster order; NULL = not used):
// scaling-list quantiser / dequantiser coefficients (ScalingList::m_quantCoef / m_dequantCoef [size][list][rem], quant.cpp:463,566) and
// the denoiser's offsets with its running residual sums (NoiseReduction, quant.cpp:444-451, dct.cpp:744-755)
struct TuTables
{
    const int32_t* qc; const int32_t* dqc; const uint16_t* nrOff; uint32_t* nrSum;
    int16_t* dctOut; int32_t* duOut;      // capture for a host-side RDOQ pass: transform coefficients / deltaU, laid out like the levels
    // the data-parallel half of Quant::rdoQuant (round 3): nquant's levels + count (quant.cpp:626, dct.cpp:688-713) and what the pre-pass
    // slots nonPsyRdoQuant / psyRdoQuant (= _1p + _2p) write per coefficient and add per 4x4 coefficient group (dct.cpp:986-1069)
    long long* rdoqCost; long long* rdoqCg; int16_t* rdoqLevels; uint32_t* rdoqNumSig; int16_t* fencDct; long long psyScale;
    __device__ __forceinline__ TuTables at(size_t elemOff) const
    {
        TuTables t = *this;
        if (t.dctOut) t.dctOut += elemOff;
        if (t.duOut) t.duOut += elemOff;
        if (t.rdoqCost) t.rdoqCost += elemOff;
        if (t.rdoqCg) t.rdoqCg += elemOff >> 3;
        if (t.rdoqLevels) t.rdoqLevels += elemOff;
        if (t.fencDct) t.fencDct += elemOff;
        return t;
    }
};

struct TuArgs
{
    const uint8_t* fenc; long fencStrideB;
    const uint8_t* fref; long frefStrideB;
    uint8_t* recon; long reconStrideB;
    int ctusW, depth, level;
    const int2* mv;                 // [ctu*85] {cost, qx | qy << 16}
    int qp, intraSlice;          // intraSlice: the X265HIP_TU_* flag bits (+ TU_FLAG_RASTER_ORDER, set by the launcher alone)
    int16_t* levels; uint32_t* numSig; unsigned long long* dist;
    TuTables tab;
};

struct TuArgs2 { TuArgs p[2]; };

__device__ __forceinline__ int tu_clip16(int v, int maxVal) { const int16_t s = (int16_t)v; return s < 0 ? 0 : (s > maxVal ? maxVal : s); }
__device__ __forceinline__ int tu_sat16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

// The transform-coding round trip of one N x N block whose prediction (pred) and source (fe) already sit in LDS:
// residual -> forward transform -> quant -> dequant -> (DC shortcut | inverse transform) -> reconstruction -> SSE.
// DST selects the 4x4 DST-VII of intra luma TUs (quant.cpp:426-431, no DC shortcut :583).  Every thread of the workgroup
// calls it; A / B are N*N int16 scratch, red / sNumSig reduction scratch (sNumSig must be 0 on entry).
template <int N, bool DST> __device__ __forceinline__ int tu_mat(int k, int i) { return DST ? (int)kTuDst[k][i] : (int)kTu.m[k * (32 / N)][i]; }

// The MFMA operands of the 16 / 32 point transforms (matrix fragments + biases of wave 0's lanes); empty below 16.  A
// persistent workgroup builds them once and reuses them for every block it processes.
template <int N, bool USE> struct TuOps { __device__ __forceinline__ void init(int) {} };
template <int N> struct TuOps<N, true>
{
    DctOperand<N, false> fw;
    DctOperand<N, true> iv;
    __device__ __forceinline__ void init(int lane)
    {
        auto matrix = [](int r, int c) { return (int)kTu.m[r][c]; };
        auto colsum = [](int c) { return N == 16 ? kTuColSum.s16[c] : kTuColSum.s32[c]; };
        fw.init(lane, matrix, colsum);
        iv.init(lane, matrix, colsum);
    }
};
template <int N, bool DST> using TuOpsFor = TuOps<N, (N >= 16 && !DST)>;

// ---- sign-bit hiding (Quant::signBitHidingHDQ, quant.cpp:247-395; on by default in x265: pps.bSignHideEnabled) -------------------
// Coefficient scans of the standard (6.5.3-6.5.5) over 4x4 coefficient groups: up-right diagonal everywhere, horizontal / vertical
// for 4x4 and 8x8 intra TUs whose direction asks for it (CUData::getTUEntropyCodingParameters, cudata.cpp:2067-2089).  The tables
// hold x | y << 3 of the k-th position of an n x n up-right diagonal scan.
__constant__ uint8_t kTuDiag2[4] = { 0, 8, 1, 9 };
__constant__ uint8_t kTuDiag4[16] = { 0, 8, 1, 16, 9, 2, 24, 17, 10, 3, 25, 18, 11, 26, 19, 27 };
__constant__ uint8_t kTuDiag8[64] = { 0, 8, 1, 16, 9, 2, 24, 17, 10, 3, 32, 25, 18, 11, 4, 40, 33, 26, 19, 12, 5, 48, 41, 34, 27, 20, 13, 6, 56, 49, 42, 35,
                                      28, 21, 14, 7, 57, 50, 43, 36, 29, 22, 15, 58, 51, 44, 37, 30, 23, 59, 52, 45, 38, 31, 60, 53, 46, 39, 61, 54, 47, 62, 55, 63 };
enum { TU_SCAN_DIAG = 0, TU_SCAN_HOR = 1, TU_SCAN_VER = 2, TU_FLAG_INTRA_SLICE = 1, TU_FLAG_SIGN_HIDE = 2, TU_FLAG_RASTER_ORDER = 1 << 30 };

// raster position inside the N x N block of scan position i (0..15) of coefficient group cg
template <int N> __device__ __forceinline__ int tu_scan_pos(int scanType, int cg, int i)
{
    constexpr int G = N / 4;
    int cx, cy, ix, iy;
    if (scanType == TU_SCAN_DIAG || N > 8)
    {
        const int c = G == 1 ? 0 : (G == 2 ? kTuDiag2[cg] : (G == 4 ? kTuDiag4[cg] : kTuDiag8[cg]));
        cx = c & 7; cy = c >> 3;
        const int q = kTuDiag4[i];
        ix = q & 7; iy = q >> 3;
    }
    else if (scanType == TU_SCAN_HOR) { cx = cg % G; cy = cg / G; ix = i & 3; iy = i >> 2; }
    else { cx = cg / G; cy = cg % G; ix = i >> 2; iy = i & 3; }
    return (cy * 4 + iy) * N + cx * 4 + ix;
}

// One lane per coefficient group (NN / 16 <= 64 groups, lanes of wavefront 0).  lev: the quantised levels (LDS, updated in place),
// aux: per coefficient (deltaU << 1) | (dct coefficient < 0) as the quantiser left it; returns the change of numSig of this lane's group.
template <int N> __device__ __forceinline__ int tu_sign_hide_group(int16_t* lev, const int16_t* aux, int16_t* lvOut, int scanType, int cg, int cgLast)
{
    int first = -1, last = -1, sum = 0;
    int pos[16];
#pragma unroll
    for (int i = 0; i < 16; i++)
    {
        pos[i] = tu_scan_pos<N>(scanType, cg, i);
        const int l = lev[pos[i]];
        if (l) { if (first < 0) first = i; last = i; sum += l; }
    }
    if (first < 0 || last - first < 4) return 0;
    const int signbit = lev[pos[first]] > 0 ? 0 : 1;
    if (signbit == (sum & 1)) return 0;
    int minCost = 0x7fffffff, minPos = -1, change = 0;
#pragma unroll
    for (int i = 15; i >= 0; i--)
    {
        if (cg == cgLast && i > last) continue;
        const int p = pos[i], l = lev[p], a = aux[p], du = a >> 1;
        int cost = 0x7fffffff, ch = 0;
        if (l)
        {
            if (du > 0) { cost = -du; ch = 1; }
            else if (!(i == first && (l == 1 || l == -1))) { cost = du; ch = -1; }
        }
        else if (i > first || (a & 1) == signbit) { cost = -du; ch = 1; }
        if (cost < minCost) { minCost = cost; minPos = p; change = ch; }
    }
    const int l = lev[minPos];
    if (l == 32767 || l == -32768) change = -1;
    int dn = 0;
    if (!l) dn = 1;
    else if (change == -1 && (l == 1 || l == -1)) dn = -1;
    const int nl = l + ((aux[minPos] & 1) ? -change : change);
    lev[minPos] = (int16_t)nl;
    lvOut[minPos] = (int16_t)nl;
    return dn;
}

// TAB: the launch carries scaling-list / denoiser tables (x265hip_tu_tables).  A separate instantiation: with the table branches compiled
// into the default kernels the 32x32 inter stage crossed a register-count step and lost half of its resident workgroups (89 -> 183 us).
template <typename Px, int N, bool DST, bool TAB = false>
__device__ __forceinline__ void tu_chain(const TuOpsFor<N, DST>& ops, const int16_t* pred, const int16_t* fe, int16_t* A, int16_t* B, unsigned long long* red, int& sNumSig,
                                         int depth, int qp, int flags, int16_t* lvOut, uint32_t* numSigOut, unsigned long long* distOut,
                                         Px* rec, long cst, int scanType = TU_SCAN_DIAG, const TuTables tab = TuTables{ nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0 },
                                         const size_t blockIndex = 0)
{
    constexpr int NN = N * N, LOG2N = N == 4 ? 2 : (N == 8 ? 3 : (N == 16 ? 4 : 5));
    const int tid = threadIdx.x, nth = blockDim.x;
    const int maxVal = (1 << depth) - 1;
    // ---- residual + forward transform (two passes, int16-truncating stores) -------------------------------
    for (int i = tid; i < NN; i += nth) A[i] = (int16_t)((int)fe[i] - (int)pred[i]);
    __syncthreads();
    const int sh1 = LOG2N - 1 + depth - 8, sh2 = LOG2N + 6;
    const int per = qp / 6, rem = qp - per * 6;
    const int transformShift = 15 - depth - LOG2N;
    const int qbits = 14 + per + transformShift;
    const int qadd = ((flags & TU_FLAG_INTRA_SLICE) ? 171 : 85) << (qbits - 9);
    const bool signHide = (flags & TU_FLAG_SIGN_HIDE) != 0;
    const int qbits8 = qbits - 8;
    const int qscale = kTuQuantScales[rem];
    // one transform coefficient -> (denoised coefficient, quantiser scale): primitives.denoiseDct before the quantiser (quant.cpp:444-451),
    // the scaling list's coefficient instead of the flat scale (quant.cpp:463)
    auto prepare = [&](const int e, int c, int& scale)
    {
        if (TAB && tab.nrOff)
        {
            const int sign = c >> 31;
            int level = (c + sign) ^ sign;
            atomicAdd(&tab.nrSum[e], (uint32_t)level);
            level -= (int)tab.nrOff[e];
            c = (int16_t)(level < 0 ? 0 : (level ^ sign) - sign);
        }
        scale = (TAB && tab.qc) ? tab.qc[e] : qscale;
        return c;
    };
    int16_t* lv = lvOut;
    int nz = 0;
    constexpr bool USE_MFMA = N >= 16 && !DST;           // the 16 / 32 point transforms are dense matrix products: matrix cores
    const int lane = tid & 63, wave = tid >> 6;
    // ---- RDOQ pre-passes (TAB launches that ask for them): Quant::rdoQuant (quant.cpp:609+) starts from primitives.nquant - the
    // quantiser with rounding 1/2, absolute levels (dct.cpp:688-713) - and needs, per coefficient, the cost of NOT coding it:
    // nonPsyRdoQuant (dct.cpp:986-1005) costUncoded = coef^2 << scaleBits; psyRdoQuant = _1p + _2p (:1006-1069) additionally
    // - ((psyScale * (fencDct - coef)) >> max(0, 2 * transformShift + 1)) with the SOURCE block's transform (m_fencDctCoeff, quant.cpp:436-441).
    // Every slot call adds 16 values of one 4x4 group to totalUncodedCost / totalRdCost: rdoqCg holds, per group (raster order), [0] the sum
    // of coef^2 << scaleBits (what nonPsyRdoQuant and psyRdoQuant_1p add) and [1] the sum of the finished costUncoded (what psyRdoQuant
    // and psyRdoQuant_2p add) - hosts without AVX-512 call _1p AND _2p on the same totals (quant.cpp:716-717), i.e. add [0] + [1].
    const bool rdoq = TAB && tab.rdoqCost != nullptr;
    const bool rdoqPsy = rdoq && tab.psyScale != 0;
    const int rdoqScaleBits = 15 - 2 * transformShift, rdoqPsyShift = 2 * transformShift + 1 > 0 ? 2 * transformShift + 1 : 0;
    const int nqAdd = 1 << (qbits - 1);
    __shared__ unsigned long long sCg[TAB ? NN / 8 : 1];
    __shared__ int sNq;
    if (rdoq)
    {
        for (int i = tid; i < NN / 8; i += nth) sCg[i] = 0;
        if (tid == 0) sNq = 0;
        if (nth > 64) __syncthreads();          // the emitting wavefront may not be the one that cleared a slot (uniform condition)
    }
    int nq = 0;
    // c = the coefficient the quantiser sees (after the denoiser), fd = the source block's coefficient at the same position
    auto rdoq_emit = [&](const int e, const int c, const int qs, const int fd)
    {
        long long cost = ((long long)c * c) << rdoqScaleBits;
        const int g = ((e >> LOG2N) >> 2) * (N >> 2) + ((e & (N - 1)) >> 2);
        atomicAdd(&sCg[2 * g], (unsigned long long)cost);
        if (rdoqPsy) cost -= (tab.psyScale * (long long)(fd - c)) >> rdoqPsyShift;
        tab.rdoqCost[e] = cost;
        atomicAdd(&sCg[2 * g + 1], (unsigned long long)cost);
        int level = (abs(c) * qs + nqAdd) >> qbits;
        nq += level != 0;
        if (c < 0) level = -level;
        level = abs(level < -32768 ? -32768 : (level > 32767 ? 32767 : level));
        if (tab.rdoqLevels) tab.rdoqLevels[e] = (int16_t)level;
    };
    // 16 consecutive int16 of an LDS row (forward operands) / 16 samples down a column (inverse operands)
    auto lds_row16 = [&](const int16_t* base, uint32_t (&d)[8])
    {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(base);
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = q[k];
    };
    auto lds_col16 = [&](const int16_t* base, uint32_t (&d)[8])
    {
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = (uint32_t)(uint16_t)base[(2 * k) * N] | ((uint32_t)(uint16_t)base[(2 * k + 1) * N] << 16);
    };
    if constexpr (USE_MFMA)
    {
        if (wave == 0)
        {
            typedef Mfma<N> MF;
            const DctOperand<N, false>& fw = ops.fw;
            const int kb = MF::kbase(lane), rn = MF::mn(lane);
            uint32_t d[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
            int p[MF::NACC];
            int fdReg[MF::NACC];
            if (rdoqPsy)
            {   // the source block's transform (quant.cpp:436-441: copy_ps + dct of fenc), same two passes; B is free until the residual's pass 1
                if (fw.kvalid) lds_row16(fe + rn * N + kb, d);
                fw.product(d, p);
#pragma unroll
                for (int r = 0; r < MF::NACC; r++) B[MF::row(lane, r) * N + MF::col(lane)] = (int16_t)((p[r] + (1 << (sh1 - 1))) >> sh1);
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);
                if (fw.kvalid) lds_row16(B + rn * N + kb, d);
                fw.product(d, p);
#pragma unroll
                for (int r = 0; r < MF::NACC; r++)
                {
                    fdReg[r] = (int16_t)((p[r] + (1 << (sh2 - 1))) >> sh2);
                    if (tab.fencDct) tab.fencDct[MF::row(lane, r) * N + MF::col(lane)] = (int16_t)fdReg[r];
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 8; k++) d[k] = 0;
            }
            // pass 1: P[k][j] = sum_i M[k][i] * resid[j][i]
            if (fw.kvalid) lds_row16(A + rn * N + kb, d);
            fw.product(d, p);
#pragma unroll
            for (int r = 0; r < MF::NACC; r++) B[MF::row(lane, r) * N + MF::col(lane)] = (int16_t)((p[r] + (1 << (sh1 - 1))) >> sh1);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            // pass 2 fused with quant (quant.cpp:462-469, dct.cpp:664-686)
            if (fw.kvalid) lds_row16(B + rn * N + kb, d);
            fw.product(d, p);
#pragma unroll
            for (int r = 0; r < MF::NACC; r++)
            {
                const int e = MF::row(lane, r) * N + MF::col(lane);
                int qs;
                const int c = prepare(e, (int16_t)((p[r] + (1 << (sh2 - 1))) >> sh2), qs);
                const int t = abs(c) * qs;
                int level = (t + qadd) >> qbits;
                // B's pass-1 values were consumed by the products above (one wavefront, LDS operations in order): it now keeps what
                // sign hiding needs per coefficient - deltaU (dct.cpp:679, within +-256) and the sign of the transform coefficient
                if (signHide) B[e] = (int16_t)((((t - (level << qbits)) >> qbits8) << 1) | (c < 0));
                if (TAB && tab.dctOut) tab.dctOut[e] = (int16_t)c;
                if (TAB && tab.duOut) tab.duOut[e] = (t - (level << qbits)) >> qbits8;
                if (rdoq) rdoq_emit(e, c, qs, rdoqPsy ? fdReg[r] : 0);
                nz += level != 0;
                if (c < 0) level = -level;
                level = tu_sat16(level);
                A[e] = (int16_t)level;            // A is free again: quantised levels
                lv[e] = (int16_t)level;
            }
        }
    }
    else
    {
    int auxReg = 0;
    int fdOne = 0;
    if (rdoqPsy)
    {   // the source block's transform - always the DCT, also where the residual takes the DST (quant.cpp:436-441) - NN <= 64 <= blockDim:
        // a thread owns at most one coefficient
        for (int e = tid; e < NN; e += nth)
        {
            const int k = e >> LOG2N, j = e & (N - 1);
            int acc = 0;
#pragma unroll
            for (int i = 0; i < N; i++) acc += tu_mat<N, false>(k, i) * (int)fe[j * N + i];
            B[k * N + j] = (int16_t)((acc + (1 << (sh1 - 1))) >> sh1);
        }
        __syncthreads();
        for (int e = tid; e < NN; e += nth)
        {
            const int k = e >> LOG2N, j = e & (N - 1);
            int acc = 0;
#pragma unroll
            for (int i = 0; i < N; i++) acc += tu_mat<N, false>(k, i) * (int)B[j * N + i];
            fdOne = (int16_t)((acc + (1 << (sh2 - 1))) >> sh2);
            if (tab.fencDct) tab.fencDct[e] = (int16_t)fdOne;
        }
        __syncthreads();
    }
    for (int e = tid; e < NN; e += nth)
    {
        const int k = e >> LOG2N, j = e & (N - 1);
        int acc = 0;
#pragma unroll
        for (int i = 0; i < N; i++) acc += tu_mat<N, DST>(k, i) * (int)A[j * N + i];
        B[k * N + j] = (int16_t)((acc + (1 << (sh1 - 1))) >> sh1);
    }
    __syncthreads();
    // ---- second pass fused with quant (quant.cpp:462-469, dct.cpp:664-686) ----------------------------------
    for (int e = tid; e < NN; e += nth)
    {
        const int k = e >> LOG2N, j = e & (N - 1);
        int acc = 0;
#pragma unroll
        for (int i = 0; i < N; i++) acc += tu_mat<N, DST>(k, i) * (int)B[j * N + i];
        int qs;
        const int c = prepare(e, (int16_t)((acc + (1 << (sh2 - 1))) >> sh2), qs);
        const int t = abs(c) * qs;
        int level = (t + qadd) >> qbits;
        auxReg = (((t - (level << qbits)) >> qbits8) << 1) | (c < 0);
        if (TAB && tab.dctOut) tab.dctOut[e] = (int16_t)c;
        if (TAB && tab.duOut) tab.duOut[e] = (t - (level << qbits)) >> qbits8;
        if (rdoq) rdoq_emit(e, c, qs, fdOne);
        nz += level != 0;
        if (c < 0) level = -level;
        level = tu_sat16(level);
        A[e] = (int16_t)level;            // A is free again: quantised levels
        lv[e] = (int16_t)level;
    }
    if (signHide)
    {   // NN <= 64 <= blockDim here: a thread owns at most one coefficient; B may be rewritten once every thread is through reading it
        __syncthreads();
        if (tid < NN) B[tid] = (int16_t)auxReg;
    }
    }
    nz = group_sum<64>(nz);
    if ((tid & 63) == 0 && nz) atomicAdd(&sNumSig, nz);
    if (rdoq)
    {
        nq = group_sum<64>(nq);
        if ((tid & 63) == 0 && nq) atomicAdd(&sNq, nq);
    }
    __syncthreads();
    if (rdoq)
    {
        if (tab.rdoqCg) for (int i = tid; i < NN / 8; i += nth) tab.rdoqCg[i] = (long long)sCg[i];
        if (tid == 0 && tab.rdoqNumSig) tab.rdoqNumSig[blockIndex] = (uint32_t)sNq;
    }
    if (signHide && sNumSig >= 2)
    {   // Quant::signBitHidingHDQ: one lane per 4x4 coefficient group, the groups are independent of each other
        if (tid < 64)
        {
            constexpr int NCG = NN / 16;
            bool any = false;
            if (tid < NCG)
            {
#pragma unroll
                for (int i = 0; i < 16; i++) any |= A[tu_scan_pos<N>(scanType, tid, i)] != 0;
            }
            const unsigned long long mask = __ballot(any);
            const int cgLast = 63 - __builtin_clzll(mask | 1ull);             // numSig >= 2: at least one group is populated
            int dn = 0;
            if (tid < NCG && any) dn = tu_sign_hide_group<N>(A, B, lv, scanType, tid, cgLast);
            dn = group_sum<64>(dn);
            if (tid == 0 && dn) sNumSig += dn;
        }
        __syncthreads();
    }
    const int numSig = sNumSig;
    if (tid == 0) *numSigOut = (uint32_t)numSig;

    // ---- inverse path --------------------------------------------------------------------------------------
    unsigned long long part = 0;
    if (numSig)
    {
        const int dqShift = 20 - 14 - transformShift, dqAdd = 1 << (dqShift - 1);
        const int dqScale = kTuInvQuantScales[rem] << per;
        // dequant_normal (flat lists) or dequant_scaling with the list's coefficient (dct.cpp:612-662, quant.cpp:562-572)
        auto dequant = [&](const int e, const int level)
        {
            if (!TAB || !tab.dqc) return tu_sat16((level * dqScale + dqAdd) >> dqShift);
            const int sh = dqShift + 4, prod = level * tab.dqc[e];
            if (sh > per) return tu_sat16((prod + (1 << (sh - per - 1))) >> (sh - per));
            return tu_sat16(tu_sat16(prod) << (per - sh));
        };
        if (numSig == 1 && A[0] != 0 && !DST)
        {
            // DC-only shortcut (quant.cpp:586-598)
            const int deq = dequant(0, (int)A[0]);
            const int shift2 = 12 - (depth - 8) - 3;
            const int dc = (int16_t)(((((deq + 1) >> 1) * 8) + (1 << (shift2 - 1))) >> shift2);
            for (int i = tid; i < NN; i += nth)
            {
                const int y = i >> LOG2N, x = i & (N - 1);
                const int v = clip3(0, maxVal, (int)pred[i] + dc);
                rec[y * cst + x] = (Px)v;
                const int d = (int)fe[i] - v;
                part += (unsigned)(d * d);
            }
        }
        else
        {
            for (int i = tid; i < NN; i += nth) B[i] = (int16_t)dequant(i, (int)A[i]);
            __syncthreads();
            const int shI = 12 - (depth - 8);
            if constexpr (USE_MFMA)
            {
                if (wave == 0)
                {
                    typedef Mfma<N> MF;
                    const DctOperand<N, true>& iv = ops.iv;
                    const int kb = MF::kbase(lane), rn = MF::mn(lane);
                    uint32_t d[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
                    int p[MF::NACC];
                    // pass 1: P[k][j] = sum_i M[i][k] * deq[i][j]  (contraction down the columns), stored as out1[j][k]
                    if (iv.kvalid) lds_col16(B + kb * N + rn, d);
                    iv.product(d, p);
#pragma unroll
                    for (int r = 0; r < MF::NACC; r++) A[MF::col(lane) * N + MF::row(lane, r)] = (int16_t)tu_sat16((p[r] + 64) >> 7);
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    // pass 2: r[j][k] = sum_i M[i][k] * out1[i][j]
                    if (iv.kvalid) lds_col16(A + kb * N + rn, d);
                    iv.product(d, p);
                    // a lane's accumulator rows come in runs of four consecutive k of output row j: one 4-byte (8-byte) store per run instead of four single-sample
                    // stores (16 store instructions per 32 x 32 block, every one of them 64 scattered bytes)
                    int vq[4];
#pragma unroll
                    for (int r = 0; r < MF::NACC; r++)
                    {
                        const int j = MF::col(lane), k = MF::row(lane, r);
                        const int res = tu_sat16((p[r] + (1 << (shI - 1))) >> shI);
                        const int v = clip3(0, maxVal, (int)pred[j * N + k] + res);
                        vq[r & 3] = v;
                        const int dd = (int)fe[j * N + k] - v;
                        part += (unsigned)(dd * dd);
                        if ((r & 3) == 3)
                        {
                            uint8_t* rp = reinterpret_cast<uint8_t*>(rec + j * cst + (k - 3));
                            if (sizeof(Px) == 1)
                                *reinterpret_cast<u32_unaligned*>(rp) = (uint32_t)vq[0] | ((uint32_t)vq[1] << 8) | ((uint32_t)vq[2] << 16) | ((uint32_t)vq[3] << 24);
                            else
                            {
                                reinterpret_cast<u32_unaligned*>(rp)[0] = (uint32_t)vq[0] | ((uint32_t)vq[1] << 16);
                                reinterpret_cast<u32_unaligned*>(rp)[1] = (uint32_t)vq[2] | ((uint32_t)vq[3] << 16);
                            }
                        }
                    }
                }
            }
            else
            {
            for (int e = tid; e < NN; e += nth)
            {
                const int j = e >> LOG2N, k = e & (N - 1);
                int acc = 0;
#pragma unroll
                for (int i = 0; i < N; i++) acc += tu_mat<N, DST>(i, k) * (int)B[i * N + j];
                A[j * N + k] = (int16_t)tu_sat16((acc + 64) >> 7);
            }
            __syncthreads();
            for (int e = tid; e < NN; e += nth)
            {
                const int j = e >> LOG2N, k = e & (N - 1);
                int acc = 0;
#pragma unroll
                for (int i = 0; i < N; i++) acc += tu_mat<N, DST>(i, k) * (int)A[i * N + j];
                const int r = tu_sat16((acc + (1 << (shI - 1))) >> shI);
                const int v = clip3(0, maxVal, (int)pred[e] + r);
                rec[j * cst + k] = (Px)v;
                const int d = (int)fe[e] - v;
                part += (unsigned)(d * d);
            }
            }
        }
    }
    else
    {
        for (int i = tid; i < NN; i += nth)
        {
            const int y = i >> LOG2N, x = i & (N - 1);
            const int v = pred[i];
            rec[y * cst + x] = (Px)v;
            const int d = (int)fe[i] - v;
            part += (unsigned)(d * d);
        }
    }
    part = group_sum<64>(part);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0)
    {
        unsigned long long t = 0;
        for (int i = 0; i < (nth >> 6); i++) t += red[i];
        *distOut = t;
    }
}

// CHROMA: one chroma plane of a 4:2:0 picture - N is the chroma block size (half the luma block), the mv counts 1/8 samples and
// the filters are the 4-tap chroma set (Predict::predInterChromaPixel, predict.cpp:304-351)
// threads per block of the uni-predictive inter stage.  16x16 and 32x32 blocks stay on ONE wavefront: four wavefronts per 32x32 block
// (real barriers, three of them idle through the matrix-core transforms) ran 241 us per 4K picture against 99 us (profiles/r03_tail_kernels.txt) -
// the stage is bound by the instructions it issues, not by the latency of one block's chain.
template <int N> struct InterReconThreads { static constexpr int value = N >= 16 ? 64 : 256; };
// Wavefronts per SIMD the single-wavefront 16 / 32 point kernels are compiled for.  Left to itself the compiler took 284 - 401 registers
// for the 32x32 kernels (a 64-thread workgroup may have 512): ONE resident wavefront per SIMD, every LDS and memory wait of a block's
// chain exposed - 103 us per 4K picture for luma.  Two wavefronts (256 registers, nothing spilled): 73 us.  Three (170 registers, 86
// spilled) lose again: 109 us; the 16x16 kernels sit at ~190 registers = two wavefronts by themselves, and forcing three or four
// (spills) made the chroma pair slower, 81 -> 93 / 85 us (profiles/r03_tail_kernels.txt).
template <int N> struct TuWavesPerSimd { static constexpr int value = N == 32 ? 2 : (N == 16 ? 2 : 1); };

template <typename Px, int N, bool CHROMA, bool TAB = false>
__global__ void __launch_bounds__(InterReconThreads<N>::value, TuWavesPerSimd<N>::value) inter_recon_kernel(TuArgs2 aa, int nblocks)
{
    const TuArgs& a = aa.p[blockIdx.y];          // grid.y = plane: Cb and Cr of a picture (or of a band of it) share one launch
    constexpr int NN = N * N, LOG2N = N == 4 ? 2 : (N == 8 ? 3 : (N == 16 ? 4 : 5));
    constexpr int TAPS = CHROMA ? 4 : 8, APRON = TAPS / 2 - 1, PW = N + TAPS - 1, PP = PW + 1;
    constexpr int NL = CHROMA ? 2 * N : N, CTU = CHROMA ? 32 : 64, MVSH = CHROMA ? 3 : 2, MVMASK = CHROMA ? 7 : 3;
    constexpr int BPP = sizeof(Px);
    __shared__ __attribute__((aligned(16))) int16_t patch[PW * PP];
    __shared__ int16_t immed[PW * N];
    __shared__ __attribute__((aligned(16))) int16_t pred[NN], fe[NN], A[NN], B[NN];
    __shared__ unsigned long long red[4];
    __shared__ int sNumSig;

    const int npu = (64 / NL) * (64 / NL);
    const int lbase = NL == 8 ? 0 : (NL == 16 ? 64 : 80);
    const int tid = threadIdx.x, nth = blockDim.x;
    const int maxVal = (1 << a.depth) - 1, headRoom = 14 - a.depth;
    auto tap = [](int f, int t) { return CHROMA ? (int)kTuChromaTaps[f][t] : (int)kTuTaps[f][t]; };
    // 16 / 32: a persistent single-wavefront workgroup walks blocks blockIdx.x, + gridDim.x, ... with the MFMA operands of the
    // transforms built once (its barriers are wave-local); 8: one block per workgroup
    TuOpsFor<N, false> ops;
    ops.init(tid & 63);
    // geometry of block `blk` given its packed quarter-sample motion vector
    struct Geo { int ctu, z, px, py, qx, qy, xf, yf; };
    auto geom = [&](const int blk, const int packed) -> Geo
    {
        Geo g;
        g.ctu = blk / npu; g.z = blk - g.ctu * npu;
        const int bxz = (g.z & 1) | ((g.z >> 1) & 2) | ((g.z >> 2) & 4), byz = ((g.z >> 1) & 1) | ((g.z >> 2) & 2) | ((g.z >> 3) & 4);
        g.px = (g.ctu % a.ctusW) * CTU + bxz * N; g.py = (g.ctu / a.ctusW) * CTU + byz * N;
        g.qx = (int16_t)(packed & 0xffff); g.qy = (int16_t)(packed >> 16);
        g.xf = g.qx & MVMASK; g.yf = g.qy & MVMASK;
        return g;
    };
    auto mv_of = [&](const int blk) -> int { const int ctu = blk / npu; return a.mv[(size_t)ctu * 85 + lbase + (blk - ctu * npu)].y; };
    // 16 / 32: dword loads (4 / 2 samples) of the source block and the reference patch into REGISTERS, all issued before the first one is
    // waited for; they go into LDS (widened to int16 pairs) at the top of the block's turn.  A patch row is read to the next dword boundary
    // (one sample beyond the 2 * apron + N needed: still inside the padded plane, and inside the row's LDS pitch).
    constexpr int NT = InterReconThreads<N>::value, SPD = 4 / BPP, FD = N / SPD, PWD = (PW + SPD - 1) / SPD;
    constexpr int FI = N >= 16 ? (N * FD) / NT : 1, PI = N >= 16 ? (PW * PWD + NT - 1) / NT : 1;
    static_assert(N < 16 || (PWD * SPD <= PP && (N * FD) % NT == 0), "patch pitch / source block size");
    auto load_regs = [&](const Geo& g, uint32_t (&fv)[FI], uint32_t (&pv)[PI])
    {
        const Px* f = reinterpret_cast<const Px*>(a.fenc + (long)g.py * a.fencStrideB) + g.px;
        const long fst = a.fencStrideB / BPP;
        const Px* r = reinterpret_cast<const Px*>(a.fref + (long)(g.py + (g.qy >> MVSH) - APRON) * a.frefStrideB) + (g.px + (g.qx >> MVSH) - APRON);
        const long rst = a.frefStrideB / BPP;
#pragma unroll
        for (int k = 0; k < FI; k++) { const int i = tid + k * NT, y = i / FD, c = i - y * FD; fv[k] = ld_u32(reinterpret_cast<const uint8_t*>(f + y * fst) + 4 * c); }
#pragma unroll
        for (int k = 0; k < PI; k++)
        {
            const int i = tid + k * NT, y = i / PWD, c = i - y * PWD;
            pv[k] = i < PW * PWD ? ld_u32(reinterpret_cast<const uint8_t*>(r + y * rst) + 4 * c) : 0u;
        }
    };
    auto store_regs = [&](const uint32_t (&fv)[FI], const uint32_t (&pv)[PI])
    {
        auto put = [&](int16_t* dst, const uint32_t w)
        {
            if (BPP == 1)
            {
                uint2 v;
                v.x = (w & 0xffu) | ((w & 0xff00u) << 8); v.y = ((w >> 16) & 0xffu) | ((w >> 8) & 0xff0000u);
                *reinterpret_cast<uint2*>(dst) = v;
            }
            else *reinterpret_cast<uint32_t*>(dst) = w;
        };
#pragma unroll
        for (int k = 0; k < FI; k++) { const int i = tid + k * NT; put(fe + i * SPD, fv[k]); }
#pragma unroll
        for (int k = 0; k < PI; k++) { const int i = tid + k * NT, y = i / PWD, c = i - y * PWD; if (i < PW * PWD) put(patch + y * PP + c * SPD, pv[k]); }
    };
    // everything behind the staging: prediction from the patch in LDS; `between` runs after the prediction (the next block's loads are issued
    // there: its motion vector has arrived by then and the loads have the whole transform chain to complete); then the transform chain
    auto do_rest = [&](const Geo& g, auto&& between)
    {
    const int ctu = g.ctu, z = g.z, px = g.px, py = g.py, xf = g.xf, yf = g.yf;
    // ---- predInterLumaPixel ----------------------------------------------------------------------------
    if (xf && yf)
    {
        const int shiftPS = 6 - headRoom, offPS = -(8192 << shiftPS);
        for (int i = tid; i < PW * N; i += nth)
        {
            const int y = i >> LOG2N, x = i & (N - 1);
            int s = 0;
#pragma unroll
            for (int t = 0; t < TAPS; t++) s += (int)patch[y * PP + x + t] * tap(xf, t);
            immed[i] = (int16_t)((s + offPS) >> shiftPS);
        }
        __syncthreads();
        const int shiftSP = 6 + headRoom, offSP = (1 << (shiftSP - 1)) + (8192 << 6);
        for (int i = tid; i < NN; i += nth)
        {
            const int y = i >> LOG2N, x = i & (N - 1);
            int s = 0;
#pragma unroll
            for (int t = 0; t < TAPS; t++) s += (int)immed[(y + t) * N + x] * tap(yf, t);
            pred[i] = (int16_t)tu_clip16((s + offSP) >> shiftSP, maxVal);
        }
    }
    else
    {
        for (int i = tid; i < NN; i += nth)
        {
            const int y = i >> LOG2N, x = i & (N - 1);
            int v;
            if (!(xf | yf)) v = patch[(y + APRON) * PP + x + APRON];
            else
            {
                int s = 0;
#pragma unroll
                for (int t = 0; t < TAPS; t++)
                    s += (int)(xf ? patch[(y + APRON) * PP + x + t] : patch[(y + t) * PP + x + APRON]) * tap(xf ? xf : yf, t);
                v = tu_clip16((s + 32) >> 6, maxVal);
            }
            pred[i] = (int16_t)v;
        }
    }
    __syncthreads();
    between();

    tu_chain<Px, N, false, TAB>(ops, pred, fe, A, B, red, sNumSig, a.depth, a.qp, a.intraSlice,
                           a.levels + ((size_t)ctu * npu + z) * NN, &a.numSig[(size_t)ctu * npu + z], &a.dist[(size_t)ctu * npu + z],
                           reinterpret_cast<Px*>(a.recon + (long)py * a.reconStrideB) + px, a.reconStrideB / BPP, TU_SCAN_DIAG,
                           TAB ? a.tab.at(((size_t)ctu * npu + z) * NN) : a.tab, (size_t)ctu * npu + z);
    __syncthreads();
    };
    if constexpr (N >= 16)
    {
        // A persistent wavefront walks blocks blockIdx.x, + gridDim.x, ...  Per block it used to pay two dependent trips to memory before any
        // work - the motion vector, then the source block / reference patch the vector points at - ~3 us of a 10 us (16x16) / 17 us (32x32)
        // block at one instruction per ~28 cycles (profiles/r03_inst_counters.txt).  Now the NEXT block's vector is requested at the top of a
        // block and its samples right after the prediction, into registers; they wait there until the block's turn.
        // (32x32: only the vector travels ahead - the 11 sample registers on top of its 252 made the compiler spill 261)
        constexpr bool AHEAD = N == 16;
        uint32_t fvC[FI], pvC[PI], fvN[AHEAD ? FI : 1], pvN[AHEAD ? PI : 1];
        // XCD-aware order (round 6): wavefront w sits on XCD w % 8 (observed placement, for speed only) and walks virtual indices w, w + gridDim.x, ...; the virtual index
        // -> block map hands every XCD a contiguous eighth of the picture's blocks, so the TUs of a CTU and the CTUs of a row - whose 32-byte source rows and straddling
        // reference rows share 64-byte lines - meet in ONE L2 instead of being fetched by up to four (inter_recon<32> fetched 2.3 - 4.6 x its planes, profiles/stage_traffic.json)
        int v = blockIdx.x;
        const bool xcdOrder = !(a.intraSlice & TU_FLAG_RASTER_ORDER);
        auto order = [&](const int i) { return xcdOrder ? xcd_swizzle(i, nblocks) : i; };
        if (v < nblocks)
        {
            int blk = order(v);
            Geo g = geom(blk, mv_of(blk));
            if constexpr (AHEAD) load_regs(g, fvC, pvC);
            for (; v < nblocks; v += gridDim.x)
            {
                const bool has = v + (int)gridDim.x < nblocks;
                const int nb = has ? order(v + (int)gridDim.x) : 0;
                int packedN = 0;
                if (has) packedN = mv_of(nb);
                if constexpr (!AHEAD) load_regs(g, fvC, pvC);
                store_regs(fvC, pvC);
                if (tid == 0) sNumSig = 0;
                __syncthreads();
                Geo gn = g;
                if constexpr (AHEAD)
                {
                    do_rest(g, [&]() { if (has) { gn = geom(nb, packedN); load_regs(gn, fvN, pvN); } });
#pragma unroll
                    for (int k = 0; k < FI; k++) fvC[k] = fvN[k];
#pragma unroll
                    for (int k = 0; k < PI; k++) pvC[k] = pvN[k];
                }
                else
                {
                    do_rest(g, []() {});
                    if (has) gn = geom(nb, packedN);
                }
                g = gn;
            }
        }
    }
    else
    {
        const int blk = blockIdx.x;
        const Geo g = geom(blk, mv_of(blk));
        {
            const Px* f = reinterpret_cast<const Px*>(a.fenc + (long)g.py * a.fencStrideB) + g.px;
            const long fst = a.fencStrideB / BPP;
            const Px* r = reinterpret_cast<const Px*>(a.fref + (long)(g.py + (g.qy >> MVSH) - APRON) * a.frefStrideB) + (g.px + (g.qx >> MVSH) - APRON);
            const long rst = a.frefStrideB / BPP;
            for (int i = tid; i < NN; i += nth) { const int y = i >> LOG2N, x = i & (N - 1); fe[i] = (int16_t)f[y * fst + x]; }
            for (int i = tid; i < PW * PW; i += nth) { const int y = i / PW, x = i - y * PW; patch[y * PP + x] = (int16_t)r[y * rst + x]; }
            if (tid == 0) sNumSig = 0;
        }
        __syncthreads();
        do_rest(g, []() {});
    }
}

// Bi-predictive flavour of the inter stage (B pictures; Predict::motionCompensation, predict.cpp:168-243 without weighted prediction):
// dir 1 / 2 = one list as above, dir 3 = predInterLumaShort of both lists (convert_p2s / luma_hps / luma_vps / luma_hps + luma_vss at
// 14-bit intermediate precision, :267-304) combined by addAvg.
struct TuBiArgs
{
    TuArgs t;                          // t.fref / t.mv = list 0
    const uint8_t* fref1; const int2* mv1;
    const uint8_t* dir;                // [ctu][npu]: 1, 2 or 3; NULL = all 3
    int wHave[2], wPresent[2], w[2], wOff[2], wDenom, wDenomUni[2];   // explicit weights of the two lists (wOff scaled to the bit depth; wDenom = list 0's)
};

// CHROMA: one chroma plane of a 4:2:0 picture (N = the chroma block size, 1/8-sample mvs, the 4-tap filters; predInterChromaShort,
// predict.cpp:355-409, for the short predictions)
template <typename Px, int N, bool CHROMA, bool TAB = false>
__global__ void __launch_bounds__(N >= 16 ? 64 : 256, TuWavesPerSimd<N>::value) inter_recon_bi_kernel(TuBiArgs b, int nblocks)
{
    const TuArgs& a = b.t;
    constexpr int NN = N * N, LOG2N = N == 4 ? 2 : (N == 8 ? 3 : (N == 16 ? 4 : 5));
    constexpr int TAPS = CHROMA ? 4 : 8, APRON = TAPS / 2 - 1, PW = N + TAPS - 1, PP = PW + 1;
    constexpr int NL = CHROMA ? 2 * N : N, CTU = CHROMA ? 32 : 64, MVSH = CHROMA ? 3 : 2, MVMASK = CHROMA ? 7 : 3;
    constexpr int BPP = sizeof(Px);
    auto tap = [](int f, int t) { return CHROMA ? (int)kTuChromaTaps[f][t] : (int)kTuTaps[f][t]; };
    __shared__ int16_t patch[PW * PP];
    __shared__ int16_t immed[PW * N];
    __shared__ int16_t pred[NN], ps0[NN], fe[NN], A[NN], B[NN];
    __shared__ unsigned long long red[4];
    __shared__ int sNumSig;
    const int npu = (64 / NL) * (64 / NL);
    const int lbase = NL == 8 ? 0 : (NL == 16 ? 64 : 80);
    const int tid = threadIdx.x, nth = blockDim.x;
    const int maxVal = (1 << a.depth) - 1, headRoom = 14 - a.depth;
    TuOpsFor<N, false> ops;
    ops.init(tid & 63);
    auto do_block = [&](const int blk)
    {
        const int ctu = blk / npu, z = blk - ctu * npu;
        const int bxz = (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4), byz = ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4);
        const int px = (ctu % a.ctusW) * CTU + bxz * N, py = (ctu / a.ctusW) * CTU + byz * N;
        const int d = b.dir ? b.dir[blk] : 3;
        {
            const Px* f = reinterpret_cast<const Px*>(a.fenc + (long)py * a.fencStrideB) + px;
            const long fst = a.fencStrideB / BPP;
            for (int i = tid; i < NN; i += nth) { const int y = i >> LOG2N, x = i & (N - 1); fe[i] = (int16_t)f[y * fst + x]; }
            if (tid == 0) sNumSig = 0;
        }
        for (int l = 0; l < 2; l++)
        {
            if (!(d & (1 << l))) continue;                              // uniform over the workgroup
            const int packed = (l ? b.mv1 : a.mv)[(size_t)ctu * 85 + lbase + z].y;
            const int qx = (int16_t)(packed & 0xffff), qy = (int16_t)(packed >> 16);
            const int xf = qx & MVMASK, yf = qy & MVMASK;
            __syncthreads();                                            // the patch of the other list is no longer read
            {
                const uint8_t* plane = l ? b.fref1 : a.fref;
                const Px* r = reinterpret_cast<const Px*>(plane + (long)(py + (qy >> MVSH) - APRON) * a.frefStrideB) + (px + (qx >> MVSH) - APRON);
                const long rst = a.frefStrideB / BPP;
                for (int i = tid; i < PW * PW; i += nth) { const int y = i / PW, x = i - y * PW; patch[y * PP + x] = (int16_t)r[y * rst + x]; }
            }
            __syncthreads();
            int16_t* out = (d == 3 && l == 0) ? ps0 : pred;            // list 0's short prediction waits in ps0
            // addWeightUni works on the short (14-bit) prediction like the bi-directional combination
            const bool shortOut = d == 3 || (b.wHave[l] && b.wPresent[l]);
            const int shiftPS = 6 - headRoom, offPS = -(8192 << shiftPS);
            if (xf && yf)
            {
                for (int i = tid; i < PW * N; i += nth)
                {
                    const int y = i >> LOG2N, x = i & (N - 1);
                    int s = 0;
#pragma unroll
                    for (int t = 0; t < TAPS; t++) s += (int)patch[y * PP + x + t] * tap(xf, t);
                    immed[i] = (int16_t)((s + offPS) >> shiftPS);
                }
                __syncthreads();
                const int shiftSP = 6 + headRoom, offSP = (1 << (shiftSP - 1)) + (8192 << 6);
                for (int i = tid; i < NN; i += nth)
                {
                    const int y = i >> LOG2N, x = i & (N - 1);
                    int s = 0;
#pragma unroll
                    for (int t = 0; t < TAPS; t++) s += (int)immed[(y + t) * N + x] * tap(yf, t);
                    out[i] = shortOut ? (int16_t)(s >> 6) : (int16_t)tu_clip16((s + offSP) >> shiftSP, maxVal);      // luma_vss : luma_vsp
                }
            }
            else
            {
                for (int i = tid; i < NN; i += nth)
                {
                    const int y = i >> LOG2N, x = i & (N - 1);
                    int v;
                    if (!(xf | yf))
                    {
                        const int c = patch[(y + APRON) * PP + x + APRON];
                        v = shortOut ? (int16_t)((c << headRoom) - 8192) : c;                                         // convert_p2s : copy_pp
                    }
                    else
                    {
                        int s = 0;
#pragma unroll
                        for (int t = 0; t < TAPS; t++)
                            s += (int)(xf ? patch[(y + APRON) * PP + x + t] : patch[(y + t) * PP + x + APRON]) * tap(xf ? xf : yf, t);
                        v = shortOut ? (int16_t)((s + offPS) >> shiftPS) : tu_clip16((s + 32) >> 6, maxVal);           // luma_hps / vps : hpp / vpp
                    }
                    out[i] = (int16_t)v;
                }
            }
        }
        __syncthreads();
        if (d == 3)
        {
            if (b.wHave[0] && b.wHave[1] && (b.wPresent[0] || b.wPresent[1]))
            {
                // addWeightBi (predict.cpp:411-456, weightBidir :52-55): list 0's denominator for both lists
                const int shift = b.wDenom + headRoom + 1, round = 1 << (shift - 1), offset = (b.wOff[0] + b.wOff[1]) * (1 << (shift - 1));
                for (int i = tid; i < NN; i += nth)
                    pred[i] = (int16_t)clip3(0, maxVal, (b.w[0] * ((int)ps0[i] + 8192) + b.w[1] * ((int)pred[i] + 8192) + round + offset) >> shift);
            }
            else
            {
                const int shiftAvg = 15 - a.depth, offAvg = (1 << (shiftAvg - 1)) + 2 * 8192;                          // addAvg
                for (int i = tid; i < NN; i += nth) pred[i] = (int16_t)clip3(0, maxVal, ((int)ps0[i] + (int)pred[i] + offAvg) >> shiftAvg);
            }
            __syncthreads();
        }
        else if (b.wHave[d == 2] && b.wPresent[d == 2])
        {
            // addWeightUni = weight_sp (pixel.cpp:493-515) on the short prediction of the one list used
            const int l = d == 2, shift = b.wDenomUni[l] + headRoom, round = shift ? 1 << (shift - 1) : 0;
            for (int i = tid; i < NN; i += nth)
                pred[i] = (int16_t)clip3(0, maxVal, ((b.w[l] * ((int)pred[i] + 8192) + round) >> shift) + b.wOff[l]);
            __syncthreads();
        }
        tu_chain<Px, N, false, TAB>(ops, pred, fe, A, B, red, sNumSig, a.depth, a.qp, a.intraSlice,
                               a.levels + ((size_t)ctu * npu + z) * NN, &a.numSig[(size_t)ctu * npu + z], &a.dist[(size_t)ctu * npu + z],
                               reinterpret_cast<Px*>(a.recon + (long)py * a.reconStrideB) + px, a.reconStrideB / BPP, TU_SCAN_DIAG,
                               TAB ? a.tab.at(((size_t)ctu * npu + z) * NN) : a.tab, (size_t)ctu * npu + z);
        __syncthreads();
    };
    if constexpr (N >= 16)
        for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) do_block(blk);
    else
        do_block(blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Intra TU candidate set (Search::codeIntraLumaQT's pixel work, search.cpp:335-373): one workgroup per (TU, mode) job -
// Predict::predIntraLumaAng (predict.cpp:579-588: filtered neighbours per g_intraFilterFlags & size, edge filter for
// sizes <= 16) computed sample by sample into LDS, then the same transform-coding round trip as the inter kernel (DST-VII
// for 4x4).  The candidate's reconstruction goes to its own block of the recon plane, so a host can evaluate many modes of
// many TUs in one launch and keep the winner.
struct IntraTuArgs
{
    const uint8_t* fenc; long fencStrideB;
    const uint8_t* nb;
    uint8_t* recon; long reconStrideB;
    const x265hip_job* jobs; int njobs;
    int depth, qp, intraSlice;
    int16_t* levels; uint32_t* numSig; unsigned long long* dist;
    TuTables tab;
    int chroma;             // predIntraChromaAng (predict.cpp:590-598): unfiltered neighbours, no edge smoothing
};

// DST: the 4x4 intra LUMA TU (chroma 4x4 takes the DCT)
template <typename Px, int N, bool DST, bool TAB = false>
__global__ void __launch_bounds__(N <= 8 ? 256 : 64, TuWavesPerSimd<N>::value) intra_recon_kernel(IntraTuArgs a)
{
    constexpr int NN = N * N, LOG2N = N == 4 ? 2 : (N == 8 ? 3 : (N == 16 ? 4 : 5));
    constexpr int BPP = sizeof(Px);
    __shared__ int16_t nbS[4 * N + 4];
    __shared__ int16_t pred[NN], fe[NN], A[NN], B[NN];
    __shared__ unsigned long long red[4];
    __shared__ int sNumSig;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int maxVal = (1 << a.depth) - 1;
    TuOpsFor<N, DST> ops;
    ops.init(tid & 63);
    // a persistent single-wavefront workgroup: the operands above are built once, the barriers below are wave-local
    auto do_job = [&](const int job)
    {
        const x265hip_job jb = a.jobs[job];
        const int mode = jb.arg[0];
        const bool filtered = !a.chroma && (kIsFilterFlags[mode] & N) != 0;
        const Px* nbp = reinterpret_cast<const Px*>(a.nb) + (filtered ? jb.off[2] : jb.off[1]);
        for (int i = tid; i < 4 * N + 1; i += nth) nbS[i] = (int16_t)nbp[i];
        {
            const Px* f = reinterpret_cast<const Px*>(a.fenc) + jb.off[0];
            const long fst = a.fencStrideB / BPP;
            for (int i = tid; i < NN; i += nth) { const int y = i >> LOG2N, x = i & (N - 1); fe[i] = (int16_t)f[y * fst + x]; }
        }
        if (tid == 0) sNumSig = 0;
        __syncthreads();
        // dcVal (intrapred.cpp:95-110): every lane sums a slice of the 2N neighbours, the wavefront adds them up
        int part = 0;
        for (int i = tid; i < 2 * N; i += nth) part += i < N ? nbS[1 + i] : nbS[2 * N + 1 + (i - N)];
        const int dc = (group_sum<64>(part) + N) / (2 * N);
        const int bFilter = !a.chroma && LOG2N <= 4;
        for (int i = tid; i < NN; i += nth)
        {
            const int y = i >> LOG2N, x = i & (N - 1);
            pred[i] = (int16_t)intra_sample(nbS, N, LOG2N, mode, bFilter, dc, maxVal, x, y);
        }
        __syncthreads();
        tu_chain<Px, N, DST, TAB>(ops, pred, fe, A, B, red, sNumSig, a.depth, a.qp, a.intraSlice,
                             a.levels + (size_t)job * NN, &a.numSig[job], &a.dist[job],
                             reinterpret_cast<Px*>(a.recon) + jb.off[3], a.reconStrideB / BPP,
                             // the scan sign hiding walks: mode-dependent for 4x4 TUs and 8x8 luma TUs (cudata.cpp:2083-2084)
                             (N == 4 || (!a.chroma && N == 8)) ? (mode >= 22 && mode <= 30 ? TU_SCAN_HOR : (mode >= 6 && mode <= 14 ? TU_SCAN_VER : TU_SCAN_DIAG))
                                                               : TU_SCAN_DIAG, TAB ? a.tab.at((size_t)job * NN) : a.tab, (size_t)job);
        __syncthreads();
    };
    // 4 / 8: one candidate per workgroup; 16 / 32: persistent, the MFMA operands above are reused
    if constexpr (N <= 8) do_job(blockIdx.x);
    else
        for (int job = blockIdx.x; job < a.njobs; job += gridDim.x) do_job(job);
}

} // namespace x265hip

using namespace x265hip;

static TuTables tu_tables_of(const x265hip_tu_tables* t)
{
    TuTables r = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0 };
    if (t)
    {
        r.qc = t->quant_coeff; r.dqc = t->dequant_coeff; r.nrOff = t->nr_offset; r.nrSum = t->nr_residual_sum; r.dctOut = t->dct_coeff_out; r.duOut = t->delta_u_out;
        r.rdoqCost = (long long*)t->rdoq_cost_uncoded; r.rdoqCg = (long long*)t->rdoq_cg_cost; r.rdoqLevels = t->rdoq_levels; r.rdoqNumSig = t->rdoq_num_sig;
        r.fencDct = t->fenc_dct_out; r.psyScale = t->psy_scale;
    }
    return r;
}
#define TABLES_OF(p) ((p)->tables)

// A/B switch (read once): X265HIP_TU_XCD_OFF = 1 / luma / chroma - the persistent TU kernels walk the blocks in raster order instead of the XCD-aware one
static bool tu_raster_order(const bool chroma)
{
    static const char* const e = getenv("X265HIP_TU_XCD_OFF");
    if (!e) return false;
    return e[0] == '1' || (chroma ? e[0] == 'c' : e[0] == 'l');
}

extern "C" int x265hip_inter_recon(const x265hip_recon_params* p, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !p->fenc || !p->fref || !p->recon || !p->mv || !p->levels || !p->num_sig || !p->dist)
    { set_error("inter_recon: NULL operand"); return X265HIP_EINVAL; }
    if ((p->width & 63) || (p->height & 63) || p->width <= 0 || p->height <= 0) { set_error("inter_recon: width/height must be multiples of 64"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("inter_recon: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->level < 0 || p->level > 2) { set_error("inter_recon: level %d (0..2 = 8x8, 16x16, 32x32)", p->level); return X265HIP_EINVAL; }
    if (p->qp < 0 || p->qp > 51 + 6 * (p->depth - 8)) { set_error("inter_recon: qp %d out of range", p->qp); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    TuArgs a;
    a.fenc = (const uint8_t*)p->fenc; a.fencStrideB = (long)p->fenc_stride * bpp;
    a.fref = (const uint8_t*)p->fref; a.frefStrideB = (long)p->fref_stride * bpp;
    a.recon = (uint8_t*)p->recon; a.reconStrideB = (long)p->recon_stride * bpp;
    a.ctusW = p->width / 64; a.depth = p->depth; a.level = p->level;
    a.mv = (const int2*)p->mv; a.qp = p->qp; a.intraSlice = (p->intra_slice & ~TU_FLAG_RASTER_ORDER) | (tu_raster_order(false) ? TU_FLAG_RASTER_ORDER : 0);
    a.levels = p->levels; a.numSig = p->num_sig; a.dist = (unsigned long long*)p->dist;
    a.tab = tu_tables_of(TABLES_OF(p));
    TuArgs2 aa = {};
    aa.p[0] = a;
    const int nctu = a.ctusW * (p->height / 64);
    const int npu = 64 >> (2 * p->level);
    hipStream_t s = (hipStream_t)stream;
    const int nblocks = nctu * npu;
    auto resident = [&](const void* fn, const int threads)
    {
        int dev = 0, cus = 256, per = 8;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, fn, threads, 0) != hipSuccess || per < 1) per = 8;
        const long r = (long)cus * per;
        return (int)(nblocks < r ? nblocks : r);
    };
#define GO_T(PX, TB) do { \
        if (p->level == 0) hipLaunchKernelGGL((inter_recon_kernel<PX, 8, false, TB>), dim3(nblocks), dim3(64), 0, s, aa, nblocks); \
        else if (p->level == 1) hipLaunchKernelGGL((inter_recon_kernel<PX, 16, false, TB>), dim3(resident((const void*)inter_recon_kernel<PX, 16, false, TB>, 64)), dim3(64), 0, s, aa, nblocks); \
        else hipLaunchKernelGGL((inter_recon_kernel<PX, 32, false, TB>), dim3(resident((const void*)inter_recon_kernel<PX, 32, false, TB>, 64)), dim3(64), 0, s, aa, nblocks); } while (0)
#define GO(PX) do { if (p->tables) GO_T(PX, true); else GO_T(PX, false); } while (0)
    if (p->depth == 8) GO(uint8_t); else GO(uint16_t);
#undef GO_T
#undef GO
    X265HIP_TRY(hipGetLastError());
    return 0;
}

static int inter_recon_bi_impl(const x265hip_recon_bi_params* q, void* stream, const bool chroma)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!q) { set_error("inter_recon_bi: NULL operand"); return X265HIP_EINVAL; }
    const x265hip_recon_params* p = &q->base;
    if (!p->fenc || !p->fref || !q->fref1 || !p->recon || !p->mv || !q->mv1 || !p->levels || !p->num_sig || !p->dist)
    { set_error("inter_recon_bi: NULL operand"); return X265HIP_EINVAL; }
    if ((p->width & 63) || (p->height & 63) || p->width <= 0 || p->height <= 0) { set_error("inter_recon_bi: width/height must be multiples of 64"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("inter_recon_bi: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->level < 0 || p->level > 2) { set_error("inter_recon_bi: level %d (0..2 = 8x8, 16x16, 32x32)", p->level); return X265HIP_EINVAL; }
    if (p->qp < 0 || p->qp > 51 + 6 * (p->depth - 8)) { set_error("inter_recon_bi: qp %d out of range", p->qp); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    TuBiArgs b;
    TuArgs& a = b.t;
    a.fenc = (const uint8_t*)p->fenc; a.fencStrideB = (long)p->fenc_stride * bpp;
    a.fref = (const uint8_t*)p->fref; a.frefStrideB = (long)p->fref_stride * bpp;
    a.recon = (uint8_t*)p->recon; a.reconStrideB = (long)p->recon_stride * bpp;
    a.ctusW = p->width / 64; a.depth = p->depth; a.level = p->level;
    a.mv = (const int2*)p->mv; a.qp = p->qp; a.intraSlice = p->intra_slice;
    a.levels = p->levels; a.numSig = p->num_sig; a.dist = (unsigned long long*)p->dist;
    a.tab = tu_tables_of(TABLES_OF(p));
    b.fref1 = (const uint8_t*)q->fref1; b.mv1 = (const int2*)q->mv1; b.dir = q->dir;
    {
        const x265hip_pred_weight* ws[2] = { q->weight0, q->weight1 };
        for (int l = 0; l < 2; l++)
        {
            b.wHave[l] = ws[l] != nullptr; b.wPresent[l] = 0; b.w[l] = 0; b.wOff[l] = 0; b.wDenomUni[l] = 0;
            if (!ws[l]) continue;
            if (ws[l]->log2_denom < 0 || ws[l]->log2_denom > 7 || ws[l]->weight < -128 || ws[l]->weight > 127 || ws[l]->offset < -128 || ws[l]->offset > 127)
            { set_error("inter_recon_bi: weight %d / offset %d / log2_denom %d of list %d out of range", ws[l]->weight, ws[l]->offset, ws[l]->log2_denom, l); return X265HIP_EINVAL; }
            b.wPresent[l] = ws[l]->present != 0; b.w[l] = ws[l]->weight; b.wOff[l] = ws[l]->offset * (1 << (p->depth - 8)); b.wDenomUni[l] = ws[l]->log2_denom;
        }
        b.wDenom = b.wDenomUni[0];
    }
    const int nblocks = a.ctusW * (p->height / 64) * (64 >> (2 * p->level));
    hipStream_t s = (hipStream_t)stream;
    auto resident = [&](const void* fn)
    {
        int dev = 0, cus = 256, per = 8;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, fn, 64, 0) != hipSuccess || per < 1) per = 8;
        const long r = (long)cus * per;
        return (int)(nblocks < r ? nblocks : r);
    };
#define GOB_T(PX, TB) do { \
        if (p->level == 0) hipLaunchKernelGGL((inter_recon_bi_kernel<PX, 8, false, TB>), dim3(nblocks), dim3(64), 0, s, b, nblocks); \
        else if (p->level == 1) hipLaunchKernelGGL((inter_recon_bi_kernel<PX, 16, false, TB>), dim3(resident((const void*)inter_recon_bi_kernel<PX, 16, false, TB>)), dim3(64), 0, s, b, nblocks); \
        else hipLaunchKernelGGL((inter_recon_bi_kernel<PX, 32, false, TB>), dim3(resident((const void*)inter_recon_bi_kernel<PX, 32, false, TB>)), dim3(64), 0, s, b, nblocks); } while (0)
#define GOBC_T(PX, TB) do { \
        if (p->level == 0) hipLaunchKernelGGL((inter_recon_bi_kernel<PX, 4, true, TB>), dim3(nblocks), dim3(64), 0, s, b, nblocks); \
        else if (p->level == 1) hipLaunchKernelGGL((inter_recon_bi_kernel<PX, 8, true, TB>), dim3(nblocks), dim3(64), 0, s, b, nblocks); \
        else hipLaunchKernelGGL((inter_recon_bi_kernel<PX, 16, true, TB>), dim3(resident((const void*)inter_recon_bi_kernel<PX, 16, true, TB>)), dim3(64), 0, s, b, nblocks); } while (0)
#define GOB(PX) do { if (chroma) { if (p->tables) GOBC_T(PX, true); else GOBC_T(PX, false); } else { if (p->tables) GOB_T(PX, true); else GOB_T(PX, false); } } while (0)
    if (p->depth == 8) GOB(uint8_t); else GOB(uint16_t);
#undef GOB_T
#undef GOBC_T
#undef GOB
    X265HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int x265hip_inter_recon_bi(const x265hip_recon_bi_params* q, void* stream) { return inter_recon_bi_impl(q, stream, false); }
/* one chroma plane of a 4:2:0 picture through the same stage (width / height = LUMA size, the luma stage's mv records, the plane's own
 * QP and weights) */
extern "C" int x265hip_inter_recon_chroma_bi(const x265hip_recon_bi_params* q, void* stream) { return inter_recon_bi_impl(q, stream, true); }

// One or both chroma planes of a picture: the same kernel, grid.y = plane.
static int inter_recon_chroma_planes(const x265hip_recon_params* const* pp, int nplanes, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    TuArgs2 aa = {};
    const x265hip_recon_params* p = pp[0];
    for (int i = 0; i < nplanes; i++)
    {
        const x265hip_recon_params* q = pp[i];
        if (!q || !q->fenc || !q->fref || !q->recon || !q->mv || !q->levels || !q->num_sig || !q->dist)
        { set_error("inter_recon_chroma: NULL operand"); return X265HIP_EINVAL; }
        if ((q->width & 63) || (q->height & 63) || q->width <= 0 || q->height <= 0) { set_error("inter_recon_chroma: width/height (luma) must be multiples of 64"); return X265HIP_EINVAL; }
        if (q->depth != 8 && q->depth != 10 && q->depth != 12) { set_error("inter_recon_chroma: depth %d", q->depth); return X265HIP_EINVAL; }
        if (q->level < 0 || q->level > 2) { set_error("inter_recon_chroma: level %d (0..2 = 8x8, 16x16, 32x32 luma blocks)", q->level); return X265HIP_EINVAL; }
        if (q->qp < 0 || q->qp > 51 + 6 * (q->depth - 8)) { set_error("inter_recon_chroma: qp %d out of range", q->qp); return X265HIP_EINVAL; }
        if (q->width != p->width || q->height != p->height || q->depth != p->depth || q->level != p->level || !q->tables != !p->tables)
        { set_error("inter_recon_chroma: the two planes differ in geometry / depth / level / use of tables"); return X265HIP_EINVAL; }
        const int bpp = q->depth == 8 ? 1 : 2;
        TuArgs& a = aa.p[i];
        a.fenc = (const uint8_t*)q->fenc; a.fencStrideB = (long)q->fenc_stride * bpp;
        a.fref = (const uint8_t*)q->fref; a.frefStrideB = (long)q->fref_stride * bpp;
        a.recon = (uint8_t*)q->recon; a.reconStrideB = (long)q->recon_stride * bpp;
        a.ctusW = q->width / 64; a.depth = q->depth; a.level = q->level;
        a.mv = (const int2*)q->mv; a.qp = q->qp; a.intraSlice = (q->intra_slice & ~TU_FLAG_RASTER_ORDER) | (tu_raster_order(true) ? TU_FLAG_RASTER_ORDER : 0);
        a.levels = q->levels; a.numSig = q->num_sig; a.dist = (unsigned long long*)q->dist;
        a.tab = tu_tables_of(TABLES_OF(q));
    }
    const int nctu = aa.p[0].ctusW * (p->height / 64);
    const int nblocks = nctu * (64 >> (2 * p->level));
    hipStream_t s = (hipStream_t)stream;
    auto resident = [&](const void* fn)
    {
        int dev = 0, cus = 256, per = 8;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, fn, 64, 0) != hipSuccess || per < 1) per = 8;
        const long r = (long)cus * per / nplanes;
        return (int)(nblocks < r ? nblocks : (r < 1 ? 1 : r));
    };
#define GOC_T(PX, TB) do { \
        if (p->level == 0) hipLaunchKernelGGL((inter_recon_kernel<PX, 4, true, TB>), dim3(nblocks, nplanes), dim3(64), 0, s, aa, nblocks); \
        else if (p->level == 1) hipLaunchKernelGGL((inter_recon_kernel<PX, 8, true, TB>), dim3(nblocks, nplanes), dim3(64), 0, s, aa, nblocks); \
        else hipLaunchKernelGGL((inter_recon_kernel<PX, 16, true, TB>), dim3(resident((const void*)inter_recon_kernel<PX, 16, true, TB>), nplanes), dim3(64), 0, s, aa, nblocks); } while (0)
#define GOC(PX) do { if (p->tables) GOC_T(PX, true); else GOC_T(PX, false); } while (0)
    if (p->depth == 8) GOC(uint8_t); else GOC(uint16_t);
#undef GOC_T
#undef GOC
    X265HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int x265hip_inter_recon_chroma(const x265hip_recon_params* p, void* stream)
{
    if (!p) { set_error("inter_recon_chroma: NULL operand"); return X265HIP_EINVAL; }
    return inter_recon_chroma_planes(&p, 1, stream);
}

/* Cb and Cr of one picture (same geometry, bit depth, block size; each with its own planes, QP and outputs) in ONE launch. */
extern "C" int x265hip_inter_recon_chroma_pair(const x265hip_recon_params* cb, const x265hip_recon_params* cr, void* stream)
{
    if (!cb || !cr) { set_error("inter_recon_chroma_pair: NULL operand"); return X265HIP_EINVAL; }
    const x265hip_recon_params* pp[2] = { cb, cr };
    return inter_recon_chroma_planes(pp, 2, stream);
}

extern "C" int x265hip_intra_recon_batch(const x265hip_intra_recon_params* p, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !p->fenc || !p->nb || !p->recon || !p->jobs || !p->levels || !p->num_sig || !p->dist) { set_error("intra_recon_batch: NULL operand"); return X265HIP_EINVAL; }
    if (p->njobs < 0) { set_error("intra_recon_batch: njobs %d", p->njobs); return X265HIP_EINVAL; }
    if (p->njobs == 0) return 0;
    if (p->n != 4 && p->n != 8 && p->n != 16 && p->n != 32) { set_error("intra_recon_batch: TU size %d", p->n); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("intra_recon_batch: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->qp < 0 || p->qp > 51 + 6 * (p->depth - 8)) { set_error("intra_recon_batch: qp %d out of range", p->qp); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    IntraTuArgs a;
    a.fenc = (const uint8_t*)p->fenc; a.fencStrideB = (long)p->fenc_stride * bpp;
    a.nb = (const uint8_t*)p->nb;
    a.recon = (uint8_t*)p->recon; a.reconStrideB = (long)p->recon_stride * bpp;
    a.jobs = p->jobs; a.njobs = p->njobs; a.depth = p->depth; a.qp = p->qp; a.intraSlice = p->intra_slice; a.chroma = p->chroma != 0;
    a.levels = p->levels; a.numSig = p->num_sig; a.dist = (unsigned long long*)p->dist;
    a.tab = tu_tables_of(TABLES_OF(p));
    hipStream_t s = (hipStream_t)stream;
    // 16 / 32: persistent single-wavefront workgroups, exactly one resident set (a second partial round would double the time);
    // 4 / 8: nothing to amortise, one workgroup per job
    auto resident = [&](const void* fn)
    {
        int dev = 0, cus = 256, per = 8;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, fn, 64, 0) != hipSuccess || per < 1) per = 8;
        const long r = (long)cus * per;
        return (int)(p->njobs < r ? p->njobs : r);
    };
#define GOI_T(PX, TB) do { \
        if (p->n == 4 && !p->chroma) hipLaunchKernelGGL((intra_recon_kernel<PX, 4, true, TB>), dim3(p->njobs), dim3(64), 0, s, a); \
        else if (p->n == 4) hipLaunchKernelGGL((intra_recon_kernel<PX, 4, false, TB>), dim3(p->njobs), dim3(64), 0, s, a); \
        else if (p->n == 8) hipLaunchKernelGGL((intra_recon_kernel<PX, 8, false, TB>), dim3(p->njobs), dim3(64), 0, s, a); \
        else if (p->n == 16) hipLaunchKernelGGL((intra_recon_kernel<PX, 16, false, TB>), dim3(resident((const void*)intra_recon_kernel<PX, 16, false, TB>)), dim3(64), 0, s, a); \
        else hipLaunchKernelGGL((intra_recon_kernel<PX, 32, false, TB>), dim3(resident((const void*)intra_recon_kernel<PX, 32, false, TB>)), dim3(64), 0, s, a); } while (0)
#define GOI(PX) do { if (p->tables) GOI_T(PX, true); else GOI_T(PX, false); } while (0)
    if (p->depth == 8) GOI(uint8_t); else GOI(uint16_t);
#undef GOI_T
#undef GOI
    X265HIP_TRY(hipGetLastError());
    return 0;
}
