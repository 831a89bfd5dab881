// me_cand_kernel.hip - the 8-bit exhaustive search, CANDIDATE-PER-LANE organisation (round 2; same outputs, bit for bit, as
// me_ctu_q_kernel in me_kernels.hip: SAD surfaces in either record format and / or the per-PU minima of cost << 32 | raster index).
//
// Reference semantics: pu[LUMA_NxN].sad / sad_x3 / sad_x4 (source/common/pixel.cpp:40-119) as issued by the full search of
// MotionEstimate::motionEstimate (source/encoder/motion.cpp:1397-1445), COPY2_IF_LT strict-less tie-break in raster order.
//
// Why another organisation.  me_ctu_q_kernel gives every LANE one 8x8 block and walks candidates row by row: the 16x16 / 32x32 / 64x64
// sums of every candidate row are cross-lane reductions (DPP + permlane swaps) and the minima need a key per row - 215 of its 620
// issue cycles per (wavefront, row) are that emission, only 405 the 16 v_qsad_pk_u16_u8 (profiles/r01_me_counters.txt).  Here a LANE
// owns one surface RECORD - the 4 horizontal displacements 4g .. 4g+3 at one vertical displacement m - and walks the 64 source rows of
// the CTU: one window row (17 LDS dwords, the lanes' windows slide by one dword) and one source row (16 SGPRs, a scalar load) feed 16
// v_qsad_pk_u16_u8, the 64 8x8 sums of the record accumulate in the lane's own registers, and every upper level is a handful of
// IN-LANE packed adds once per 16 source rows.  Per record: 1024 quad-SADs + ~800 other VALU operations instead of 1024 + ~3400, no
// cross-lane traffic, and the record leaves as 45 16-byte stores.  Minima: per lane and PU a 32-bit running key
// cost << 8 | step << 2 | column (the lane's steps visit records in raster order), widened to cost << 32 | raster index and merged
// across lanes / wavefronts once per CTU.
//
// LDS: the (64 + 2R)-row window with a row pitch congruent to NG modulo 32 dwords: lane l of a step reads
// dword (m * pitch + g + i), i.e. bank (record index + i) mod 32 - consecutive lanes, consecutive banks, whatever rows they sit on.
#include "common.h"

#include <cstdlib>

namespace x265hip {

struct MECandArgs
{
    const uint8_t* fenc;  long fencStrideB;
    const uint8_t* fref;  long frefStrideB;
    int ctusW, range;
    int pitchB, payloadDw, stepsPerWave, ntStores;
    uint8_t* surf;
    unsigned long long* best;
    const uint16_t* costX;
    const uint16_t* costY;
    const int16_t* centres;      // optional [ctu][2], see MEArgs
};

typedef unsigned long long mc_u64;
typedef unsigned short mc_v2u16 __attribute__((ext_vector_type(2)));
typedef unsigned int mc_v4u32 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mc_pkadd(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(mc_v2u16, a) + __builtin_bit_cast(mc_v2u16, b));
}
__device__ __forceinline__ uint32_t mc_min3(uint32_t a, uint32_t b, uint32_t c) { const uint32_t t = a < b ? a : b; return t < c ? t : c; }
// pins a value where it is computed: without it the compiler sinks the running-key updates of all four block-row pairs to the end of
// the step and keeps every accumulator alive (and spilled) until then
__device__ __forceinline__ void mc_pin(uint32_t& v) { asm volatile("" : "+v"(v)); }
// z-order index of block (x, y) among the blocks of its level (x -> even bits, y -> odd bits)
__host__ __device__ constexpr int mc_z(int x, int y)
{
    return (x & 1) | ((y & 1) << 1) | ((x & 2) << 1) | ((y & 2) << 2) | ((x & 4) << 2) | ((y & 4) << 3);
}
// one source row of the CTU (64 bytes) straight into 16 SGPRs: the row is the same for every lane, v_qsad_pk_u16_u8 takes it as a scalar
// operand, and the scalar cache keeps the 4 KiB source CTU - no LDS or vector-memory traffic for it.  The load is asynchronous;
// mc_swait is a full lgkmcnt(0) wait (also covers the LDS reads issued before it, which the consumer needs anyway).
typedef uint32_t mc_v16u __attribute__((ext_vector_type(16)));
__device__ __forceinline__ mc_v16u mc_sload16(const uint8_t* p)
{
    mc_v16u r;
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(r) : "s"(p) : "memory");
    return r;
}
__device__ __forceinline__ void mc_swait(mc_v16u& r) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r)); }
// ... placed after the instruction that produces `tie` (the row's last SAD): the prefetch has the whole row to arrive
__device__ __forceinline__ void mc_swait_after(mc_v16u& r, unsigned long long& tie) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r), "+v"(tie)); }
typedef const __attribute__((address_space(3))) uint32_t* mc_lds_cptr;
// the prefetch of the NEXT row: `tie` is an operand of the first SAD of the current row, so the load is issued before the row's SADs
// (an untied asm statement may be scheduled after them, which would expose the scalar-memory latency on every row)
__device__ __forceinline__ mc_v16u mc_sload16_before(const uint8_t* p, unsigned long long& tie)
{
    mc_v16u r;
    asm volatile("s_load_dwordx16 %0, %2, 0x0" : "=s"(r), "+v"(tie) : "s"(p) : "memory");
    return r;
}
// cost key of one packed u16 SAD: sad * 256 + base in ONE instruction (v_mad_u32_u16 reads either half of the register)
__device__ __forceinline__ uint32_t mc_key_lo(uint32_t x, uint32_t mul, uint32_t base)
{
    uint32_t d;
    asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(d) : "v"(x), "s"(mul), "v"(base));
    return d;
}
__device__ __forceinline__ uint32_t mc_key_hi(uint32_t x, uint32_t mul, uint32_t base)
{
    uint32_t d;
    asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(x), "s"(mul), "v"(base));
    return d;
}

// running key update with the 4 packed u16 SADs {lo: columns 0, 1; hi: columns 2, 3}
__device__ __forceinline__ uint32_t mc_upd16(uint32_t rk, uint32_t lo, uint32_t hi, const uint32_t (&base)[4])
{
    const uint32_t k0 = mc_key_lo(lo, 256u, base[0]), k1 = mc_key_hi(lo, 256u, base[1]);
    const uint32_t k2 = mc_key_lo(hi, 256u, base[2]), k3 = mc_key_hi(hi, 256u, base[3]);
    return mc_min3(mc_min3(rk, k0, k1), k2, k3);
}
__device__ __forceinline__ uint32_t mc_upd32(uint32_t rk, const uint32_t (&v)[4], const uint32_t (&base)[4])
{
    return mc_min3(mc_min3(rk, (v[0] << 8) + base[0], (v[1] << 8) + base[1]), (v[2] << 8) + base[2], (v[3] << 8) + base[3]);
}
// minimum over the wavefront, result in every lane: DPP rotations inside the 16-lane rows, then the two gfx950 row swaps
__device__ __forceinline__ uint32_t mc_wave_min(uint32_t v)
{
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    uint32_t o;
    o = (uint32_t)dpp<0x121>((int)v); v = o < v ? o : v;      // row_ror:1
    o = (uint32_t)dpp<0x122>((int)v); v = o < v ? o : v;      // row_ror:2
    o = (uint32_t)dpp<0x124>((int)v); v = o < v ? o : v;      // row_ror:4
    o = (uint32_t)dpp<0x128>((int)v); v = o < v ? o : v;      // row_ror:8
    v2u sw = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = sw.x < sw.y ? sw.x : sw.y;
    sw = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return sw.x < sw.y ? sw.x : sw.y;
}

// FMT: X265HIP_SURF_I32 / _PACKED / _PACKED_T / _PACKED_B (include/x265hip.h).  PACKED_B (round 3) is this kernel's native layout - the 64 records of
// a step form one contiguous block, a store instruction writes one aligned KiB; PACKED_T was it in round 2: a lane's 16-byte chunk c goes
// to row + (c * NG + g) * 16, so the lanes of a store instruction (consecutive g) write consecutive 16-byte pieces; with the
// record-contiguous formats the same instruction scatters 64 pieces 720 / 1360 bytes apart (measured: 3.2 / 6.0 ms instead of 1.5).
// VAR (A/B switches, X265HIP_ME_CAND_VARIANT): bit 0 - the source CTU sits in LDS and a row is read with 4 broadcast ds_read_b128
// instead of one scalar s_load_dwordx16; bit 1 - the odd dword pairs are shuffled together from the even ones (v_pk_mov_b32) instead
// of being loaded a second time.
template <bool SURF, bool BEST, int FMT, int VAR>
__global__ void __launch_bounds__(256, 2) me_ctu_c_kernel(MECandArgs a)
{
    constexpr bool SRC_LDS = (VAR & 1) != 0, ODD_SHUFFLE = (VAR & 2) != 0;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr bool PACKED = FMT != X265HIP_SURF_I32;
    constexpr int GB = PACKED ? 720 : 1360;
    const int R = a.range, NC = 2 * R + 1, NG = (NC + 3) >> 2, rows = 64 + 2 * R;
    const int ctu = blockIdx.x;
    const int cx = (ctu % a.ctusW) * 64, cy = (ctu / a.ctusW) * 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int pitchB = a.pitchB;
    uint8_t* win = smem;
    // BEST: raster index (m * NC + 4 g) of the record each lane worked on in each of its steps, [step][lane] per wavefront
    uint32_t* stepTab = reinterpret_cast<uint32_t*>(win + (size_t)rows * pitchB) + (size_t)wave * a.stepsPerWave * 64;
    // BEST: the two mv-cost tables (a global load at the top of a step would wait for the previous step's 45 stores: vmcnt is shared)
    uint16_t* costL = reinterpret_cast<uint16_t*>(win + (size_t)rows * pitchB + (BEST ? (size_t)nwaves * a.stepsPerWave * 256 : 0));
    uint32_t* srcL = reinterpret_cast<uint32_t*>(smem + (((size_t)rows * pitchB + (BEST ? (size_t)nwaves * a.stepsPerWave * 256 + 16 * NG : 0) + 15) & ~(size_t)15));   // [64][16] dwords (SRC_LDS)

    const uint8_t* fencCtu = a.fenc + (long)cy * a.fencStrideB + cx;          // uniform: rows are fetched with scalar loads
    if (BEST)
        for (int i = threadIdx.x; i < 8 * NG; i += blockDim.x)
        {
            const int t = i >= 4 * NG, c = i - t * 4 * NG;
            costL[i] = c < NC ? (t ? a.costY[c] : a.costX[c]) : (uint16_t)0;
        }
    if (SRC_LDS)
        for (int i = threadIdx.x; i < 1024; i += blockDim.x)
            srcL[i] = ld_u32(fencCtu + (long)(i >> 4) * a.fencStrideB + 4 * (i & 15));
    {
        const int ccx = a.centres ? a.centres[2 * ctu] : 0, ccy = a.centres ? a.centres[2 * ctu + 1] : 0;
        const uint8_t* g0 = a.fref + (long)(cy + ccy - R) * a.frefStrideB + (long)(cx + ccx - R);
        for (int r = wave; r < rows; r += nwaves)
        {
            const uint8_t* src = g0 + (long)r * a.frefStrideB;
            uint32_t* dst = reinterpret_cast<uint32_t*>(win + r * pitchB);
            for (int c = lane; c < a.payloadDw; c += 64) dst[c] = ld_u32(src + 4 * c);
        }
    }
    __syncthreads();

    uint32_t rk[85];
    if (BEST)
    {
#pragma unroll
        for (int i = 0; i < 85; i++) rk[i] = 0xffffffffu;
    }
    const int T = NC * NG, nsteps = (T + 63) >> 6;
    int sl = 0;                                                        // this wavefront's step counter (< 64)
    for (int S = wave; S < nsteps; S += nwaves, sl++)
    {
        const int rec0 = S * 64 + lane;
        const bool live = rec0 < T;
        const int rec = live ? rec0 : T - 1;
        const int m = rec / NG, g = rec - m * NG;
        uint32_t rowOff = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)win + (uint32_t)(m * pitchB + 4 * g);   // LDS byte address
        uint32_t base[4] = { 0, 0, 0, 0 };
        if (BEST)
        {
            stepTab[sl * 64 + lane] = (uint32_t)(m * NC + 4 * g);
            const uint32_t cyv = costL[4 * NG + m];
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const int col = 4 * g + k;
                const uint32_t tag = ((uint32_t)sl << 2) | (uint32_t)k;
                base[k] = (live && col < NC) ? (((uint32_t)costL[col] + cyv) << 8) | tag : 0x80000000u | tag;
            }
        }
        uint8_t* recp = nullptr;
        if (SURF && FMT != X265HIP_SURF_PACKED_B)
            recp = FMT == X265HIP_SURF_PACKED_T ? a.surf + (size_t)((long)ctu * NC + m) * NG * GB + (size_t)g * 16
                                                : a.surf + ((size_t)((long)ctu * NC + m) * NG + g) * GB;
        if (SURF && FMT == X265HIP_SURF_PACKED_B)          // the step's 64 records = one block: chunk c of lane l at block + (c * 64 + l) * 16
            recp = a.surf + ((size_t)ctu * nsteps + S) * (64 * GB) + (size_t)lane * 16;
        const int chunkStride = FMT == X265HIP_SURF_PACKED_B ? 1024 : NG * 16;
        // 16 bytes at byte offset `off` (a multiple of 16) of the lane's record
        auto put = [&](const int off, const mc_v4u32 v)
        {
            mc_v4u32* dst = reinterpret_cast<mc_v4u32*>((FMT == X265HIP_SURF_PACKED_T || FMT == X265HIP_SURF_PACKED_B) ? recp + (off >> 4) * chunkStride : recp + off);
            if (a.ntStores) __builtin_nontemporal_store(v, dst); else *dst = v;
        };

        uint32_t s32[2][4], s64[4];
#pragma unroll
        for (int bp = 0; bp < 4; bp++)
        {
            mc_u64 acc[16];
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = 0ull;
#pragma unroll
            for (int half = 0; half < 2; half++)
            {
                // rolled: one window row (two overlapping sets of 8 dword pairs, so that both 8-byte operands of a block column
                // arrive in consecutive registers) + one source row per iteration; the next source row is in flight during the SADs
                const uint8_t* frow = fencCtu + (long)((bp * 16 + half * 8) * a.fencStrideB);
                mc_v16u F;
                if (!SRC_LDS) F = mc_sload16(frow);
#pragma unroll 1
                for (int j = 0; j < 8; j++)
                {
                    const mc_lds_cptr dp = (mc_lds_cptr)(uintptr_t)rowOff;
                    mc_u64 W0[8], W1[8];
                    if (ODD_SHUFFLE)
                    {
                        uint32_t D[17];
#pragma unroll
                        for (int i = 0; i < 17; i++) D[i] = dp[i];
#pragma unroll
                        for (int bx = 0; bx < 8; bx++)
                        {
                            W0[bx] = ((mc_u64)D[2 * bx + 1] << 32) | D[2 * bx];
                            W1[bx] = ((mc_u64)D[2 * bx + 2] << 32) | D[2 * bx + 1];
                        }
                    }
                    else
                    {
                        uint32_t off1 = rowOff + 4;
                        asm("" : "+v"(off1));             // opaque: the odd pairs are LOADED (LDS has the headroom), not shuffled together on the VALU
                        const mc_lds_cptr dp1 = (mc_lds_cptr)(uintptr_t)off1;
#pragma unroll
                        for (int bx = 0; bx < 8; bx++)
                        {
                            W0[bx] = ((mc_u64)dp[2 * bx + 1] << 32) | dp[2 * bx];
                            W1[bx] = ((mc_u64)dp1[2 * bx + 1] << 32) | dp1[2 * bx];
                        }
                    }
                    if (SRC_LDS)
                    {
                        const mc_v4u32* fp = reinterpret_cast<const mc_v4u32*>(srcL + (bp * 16 + half * 8 + j) * 16);
                        uint32_t FL[16];
#pragma unroll
                        for (int i = 0; i < 4; i++) { const mc_v4u32 f = fp[i]; FL[4 * i] = f.x; FL[4 * i + 1] = f.y; FL[4 * i + 2] = f.z; FL[4 * i + 3] = f.w; }
#pragma unroll
                        for (int bx = 0; bx < 8; bx++)
                        {
                            mc_u64 v = acc[half * 8 + bx];
                            v = __builtin_amdgcn_qsad_pk_u16_u8(W0[bx], FL[2 * bx], v);
                            v = __builtin_amdgcn_qsad_pk_u16_u8(W1[bx], FL[2 * bx + 1], v);
                            acc[half * 8 + bx] = v;
                        }
                    }
                    else
                    {
                        mc_swait(F);
                        frow += a.fencStrideB;
                        mc_v16u Fn = mc_sload16_before(j < 7 ? frow : fencCtu, W0[0]);     // (the last prefetch of a half is a harmless re-read)
#pragma unroll
                        for (int bx = 0; bx < 8; bx++)
                        {
                            mc_u64 v = acc[half * 8 + bx];
                            v = __builtin_amdgcn_qsad_pk_u16_u8(W0[bx], F[2 * bx], v);
                            v = __builtin_amdgcn_qsad_pk_u16_u8(W1[bx], F[2 * bx + 1], v);
                            acc[half * 8 + bx] = v;
                        }
                        mc_swait_after(Fn, acc[half * 8 + 7]);
                        F = Fn;
                    }
                    rowOff += (uint32_t)pitchB;
                }
            }

            // ---- this pair of block rows is complete: 8x8 records, 16x16 sums, contributions to 32x32 / 64x64 ----------
            uint32_t q16lo[4], q16hi[4];
#pragma unroll
            for (int qx = 0; qx < 4; qx++)
            {
                const mc_u64 A = acc[2 * qx], B = acc[2 * qx + 1], C = acc[8 + 2 * qx], E = acc[8 + 2 * qx + 1];
                q16lo[qx] = mc_pkadd(mc_pkadd((uint32_t)A, (uint32_t)B), mc_pkadd((uint32_t)C, (uint32_t)E));     // <= 65280 per half: no carry
                q16hi[qx] = mc_pkadd(mc_pkadd((uint32_t)(A >> 32), (uint32_t)(B >> 32)), mc_pkadd((uint32_t)(C >> 32), (uint32_t)(E >> 32)));
            }
#pragma unroll
            for (int px = 0; px < 2; px++)
            {
                const uint32_t l0 = q16lo[2 * px], l1 = q16lo[2 * px + 1], h0 = q16hi[2 * px], h1 = q16hi[2 * px + 1];
                const uint32_t t[4] = { (l0 & 0xffffu) + (l1 & 0xffffu), (l0 >> 16) + (l1 >> 16), (h0 & 0xffffu) + (h1 & 0xffffu), (h0 >> 16) + (h1 >> 16) };
#pragma unroll
                for (int i = 0; i < 4; i++) s32[px][i] = (bp & 1) ? s32[px][i] + t[i] : t[i];
            }
            if (bp & 1)
            {
#pragma unroll
                for (int i = 0; i < 4; i++) s64[i] = bp == 1 ? s32[0][i] + s32[1][i] : s64[i] + s32[0][i] + s32[1][i];
            }
            if (SURF && live)
            {
#pragma unroll
                for (int half = 0; half < 2; half++)
#pragma unroll
                    for (int bx = 0; bx < 8; bx += 2)
                    {
                        const int z = mc_z(bx, 2 * bp + half);
                        const mc_u64 A0 = acc[half * 8 + bx], A1 = acc[half * 8 + bx + 1];
                        if (PACKED)
                            put(z * 8, mc_v4u32{ (uint32_t)A0, (uint32_t)(A0 >> 32), (uint32_t)A1, (uint32_t)(A1 >> 32) });
                        else
                        {
                            put(z * 16, mc_v4u32{ (uint32_t)A0 & 0xffffu, (uint32_t)A0 >> 16, (uint32_t)(A0 >> 32) & 0xffffu, (uint32_t)(A0 >> 48) });
                            put((z + 1) * 16, mc_v4u32{ (uint32_t)A1 & 0xffffu, (uint32_t)A1 >> 16, (uint32_t)(A1 >> 32) & 0xffffu, (uint32_t)(A1 >> 48) });
                        }
                    }
#pragma unroll
                for (int qx = 0; qx < 4; qx += 2)
                {
                    const int q = mc_z(qx, bp);
                    if (PACKED)
                        put(512 + q * 8, mc_v4u32{ q16lo[qx], q16hi[qx], q16lo[qx + 1], q16hi[qx + 1] });
                    else
                    {
                        put((64 + q) * 16, mc_v4u32{ q16lo[qx] & 0xffffu, q16lo[qx] >> 16, q16hi[qx] & 0xffffu, q16hi[qx] >> 16 });
                        put((65 + q) * 16, mc_v4u32{ q16lo[qx + 1] & 0xffffu, q16lo[qx + 1] >> 16, q16hi[qx + 1] & 0xffffu, q16hi[qx + 1] >> 16 });
                    }
                }
                if (bp & 1)
                {
#pragma unroll
                    for (int px = 0; px < 2; px++)
                    {
                        const int p = px | ((bp >> 1) << 1);
                        put(PACKED ? 640 + p * 16 : (80 + p) * 16, mc_v4u32{ s32[px][0], s32[px][1], s32[px][2], s32[px][3] });
                    }
                }
                if (bp == 3)
                    put(PACKED ? 704 : 84 * 16, mc_v4u32{ s64[0], s64[1], s64[2], s64[3] });
            }
            if (BEST)
            {
#pragma unroll
                for (int half = 0; half < 2; half++)
#pragma unroll
                    for (int bx = 0; bx < 8; bx++)
                    {
                        const int z = mc_z(bx, 2 * bp + half);
                        rk[z] = mc_upd16(rk[z], (uint32_t)acc[half * 8 + bx], (uint32_t)(acc[half * 8 + bx] >> 32), base);
                        mc_pin(rk[z]);
                    }
#pragma unroll
                for (int qx = 0; qx < 4; qx++)
                {
                    rk[64 + mc_z(qx, bp)] = mc_upd16(rk[64 + mc_z(qx, bp)], q16lo[qx], q16hi[qx], base);
                    mc_pin(rk[64 + mc_z(qx, bp)]);
                }
                if (bp & 1)
                {
#pragma unroll
                    for (int px = 0; px < 2; px++)
                    {
                        rk[80 + (px | ((bp >> 1) << 1))] = mc_upd32(rk[80 + (px | ((bp >> 1) << 1))], s32[px], base);
                        mc_pin(rk[80 + (px | ((bp >> 1) << 1))]);
                    }
                }
                if (bp == 3) { rk[84] = mc_upd32(rk[84], s64, base); mc_pin(rk[84]); }
            }
        }
    }

    if (BEST)
    {
        // merge the lanes' running keys: cost << 8 | step << 2 | column  ->  (cost, raster index), minimum over the wavefront, one
        // 64-bit atomicMin per (wavefront, PU).  8x8 / 16x16 costs fit 18 bits (65280 + 2 * 65535), so with a raster index below
        // 2^14 one 32-bit reduction orders both; the three largest levels (and wide windows) reduce cost first, then the index.
        unsigned long long* out = a.best + (size_t)ctu * 85;
        const bool narrow = NC * NC <= 16384;
#pragma unroll
        for (int pu = 0; pu < 85; pu++)
        {
            const uint32_t key = rk[pu];
            const uint32_t cost = key >> 8;
            uint32_t st = (key >> 2) & 63u;
            st = st < (uint32_t)a.stepsPerWave ? st : 0u;                         // a lane that never had a record carries the initial key
            const uint32_t raster = key == 0xffffffffu ? 0x3fffu : stepTab[st * 64 + lane] + (key & 3u);
            if (pu < 80 && narrow)
            {
                const uint32_t c18 = cost < 0x3ffffu ? cost : 0x3ffffu;          // pad columns / idle lanes: above every real cost
                const uint32_t v = mc_wave_min((c18 << 14) | (raster & 0x3fffu));
                if (lane == 0) atomicMin(&out[pu], ((unsigned long long)(v >> 14) << 32) | (v & 0x3fffu));
            }
            else
            {
                const uint32_t cmin = mc_wave_min(cost);
                const uint32_t rmin = mc_wave_min(cost == cmin ? raster : 0xffffffffu);
                if (lane == 0) atomicMin(&out[pu], ((unsigned long long)cmin << 32) | rmin);
            }
            __builtin_amdgcn_sched_barrier(0);          // one PU at a time: 85 hoisted table loads would cost 85 registers
        }
    }
}

// ------------------------------------------------------------------------------------------------ launcher (called from launch_me)
int launch_me_cand(const x265hip_me_params* p, hipStream_t s)
{
    MECandArgs a;
    a.fenc = (const uint8_t*)p->fenc;  a.fencStrideB = (long)p->fenc_stride;
    a.fref = (const uint8_t*)p->fref;  a.frefStrideB = (long)p->fref_stride;
    a.ctusW = p->width / 64; a.range = p->range;
    const int NC = 2 * p->range + 1, NG = (NC + 3) / 4, rows = 64 + 2 * p->range;
    a.payloadDw = NG + 16;
    int pitchDw = a.payloadDw;
    while ((pitchDw & 31) != (NG & 31)) pitchDw++;
    a.pitchB = pitchDw * 4;
    a.surf = (uint8_t*)p->surf; a.best = (unsigned long long*)p->best;
    a.costX = p->cost_x; a.costY = p->cost_y;
    a.centres = p->centres;
    const int nsteps = (NC * NG + 63) / 64;
    a.stepsPerWave = (nsteps + 3) / 4;
    static const char* const vs = getenv("X265HIP_ME_CAND_VARIANT");          // A/B switch, read once
    const int var = vs ? atoi(vs) & 3 : 3;            // default: source CTU in LDS, odd pairs shuffled (the fastest of the four, profiles/r02_me_cand_ab.txt)
    a.ntStores = vs ? (atoi(vs) >> 2) & 1 : 0;
    const size_t lds = (size_t)rows * a.pitchB + (p->best ? (size_t)4 * a.stepsPerWave * 256 + 16 * NG : 0) + ((var & 1) ? 4096 + 16 : 0);
    if (lds > 160 * 1024 || a.stepsPerWave > 64) return 1;          // caller falls back to the row-walking kernels
    const int nctu = a.ctusW * (p->height / 64);
    const bool anySurf = p->surf != nullptr, anyBest = p->best != nullptr;
    const int fmt = anySurf ? p->surf_format : X265HIP_SURF_I32;
#define MC_LAUNCH_V(SF, BS, FM, VR) do { \
        if (lds > 64 * 1024) X265HIP_TRY(hipFuncSetAttribute((const void*)me_ctu_c_kernel<SF, BS, FM, VR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((me_ctu_c_kernel<SF, BS, FM, VR>), dim3(nctu), dim3(256), lds, s, a); } while (0)
#define MC_LAUNCH(SF, BS, FM) do { \
        if (var == 0) MC_LAUNCH_V(SF, BS, FM, 0); else if (var == 1) MC_LAUNCH_V(SF, BS, FM, 1); \
        else if (var == 2) MC_LAUNCH_V(SF, BS, FM, 2); else MC_LAUNCH_V(SF, BS, FM, 3); } while (0)
#define MC_FMT(SF, BS) do { \
        if (fmt == X265HIP_SURF_PACKED_B) MC_LAUNCH(SF, BS, X265HIP_SURF_PACKED_B); \
        else if (fmt == X265HIP_SURF_PACKED_T) MC_LAUNCH(SF, BS, X265HIP_SURF_PACKED_T); \
        else if (fmt == X265HIP_SURF_PACKED) MC_LAUNCH(SF, BS, X265HIP_SURF_PACKED); \
        else MC_LAUNCH(SF, BS, X265HIP_SURF_I32); } while (0)
    if (anySurf && anyBest) MC_FMT(true, true);
    else if (anySurf) MC_FMT(true, false);
    else MC_LAUNCH(false, true, X265HIP_SURF_I32);
#undef MC_FMT
#undef MC_LAUNCH_V
#undef MC_LAUNCH
    X265HIP_TRY(hipGetLastError());
    return 0;
}

} // namespace x265hip
