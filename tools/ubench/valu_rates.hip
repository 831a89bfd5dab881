// (round 4) OPs 17 - 19 repeat the three SAD instructions WITHOUT the `| 1` on an operand: rounds 1 - 3 read "8.8 cycles per v_sad_u8 / u16, 25.3 per
// v_qsad_pk_u16_u8" off OPs 0 - 2, but every one of those iterations also issues the v_or_b32 (two for the 64-bit quad-SAD operand), i.e. the plain-VALU
// cost of 4.4 cycles on top.  The 10-bit search ran FASTER than the "floor" computed from 8.8 (2.67 ms against 2.89 ms, profiles/r04_bench_variants.txt),
// which is how the mistake was found.
// Micro-benchmark: issue rate of the packed-SAD VALU instructions on gfx950 (cycles per wave-instruction
// per SIMD), to size the motion-search kernels.  One workgroup of 256 threads per CU (1 wave / SIMD)
// and 4 waves / SIMD variants.  Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define N_ITERS 4096
#define UNROLL 32

template <int OP>
__global__ void k(uint32_t* out, uint32_t seed)
{
    uint32_t a[8], b = seed + threadIdx.x;
    unsigned long long q[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = seed * (i + 1) + threadIdx.x; q[i] = a[i]; }
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < N_ITERS; it++)
    {
#pragma unroll
        for (int u = 0; u < UNROLL / 8; u++)
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
                if (OP == 0) a[i] = __builtin_amdgcn_sad_u8(a[(i + 1) & 7] | 1, b, a[i]);
                if (OP == 1) a[i] = __builtin_amdgcn_sad_u16(a[(i + 1) & 7] | 1, b, a[i]);
                if (OP == 2) q[i] = __builtin_amdgcn_qsad_pk_u16_u8(q[(i + 1) & 7] | 1, b, q[i]);
                if (OP == 3) a[i] = a[i] + (a[(i + 1) & 7] ^ b);                         // v_xad / plain VALU baseline
                if (OP == 4) a[i] = __builtin_amdgcn_alignbit(a[(i + 1) & 7], a[i], b);
                if (OP == 5) a[i] = __builtin_amdgcn_sad_hi_u8(a[(i + 1) & 7] | 1, b, a[i]);
                if (OP == 6) a[i] = __builtin_amdgcn_msad_u8(a[(i + 1) & 7] | 1, b, a[i]);
                if (OP == 7) a[i] = __builtin_amdgcn_update_dpp(0, (int)a[(i + 1) & 7], 0xB1, 0xF, 0xF, true) + a[i];
                if (OP == 8) a[i] = __builtin_amdgcn_sdot4((int)a[(i + 1) & 7], (int)b, (int)a[i], false);
                if (OP == 9) { typedef short v2s __attribute__((ext_vector_type(2)));
                               a[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, a[(i + 1) & 7]), __builtin_bit_cast(v2s, b), (int)a[i], false); }
                if (OP == 10) a[i] = __mul24((int)a[(i + 1) & 7], (int)b) + a[i];      // v_mad_i32_i24
                if (OP == 11) a[i] = __builtin_amdgcn_alignbyte(a[(i + 1) & 7], a[i], b);
                if (OP == 12) a[i] = __builtin_amdgcn_perm(a[(i + 1) & 7], a[i], b);
                if (OP == 13) a[i] = a[(i + 1) & 7] * b + a[i];                                          // v_mad_u64_u32 / v_mul_lo_u32 + add
                if (OP == 14) { typedef short v2s __attribute__((ext_vector_type(2)));
                                v2s x = __builtin_bit_cast(v2s, a[(i + 1) & 7]), y = __builtin_bit_cast(v2s, b), z = __builtin_bit_cast(v2s, a[i]);
                                a[i] = __builtin_bit_cast(uint32_t, (v2s)(x * y + z)); }                  // v_pk_mad_i16
                if (OP == 15) a[i] = __builtin_amdgcn_udot4(a[(i + 1) & 7], b, a[i], false);
                if (OP == 16) a[i] = a[i] + a[(i + 1) & 7] + b;                                          // v_add3_u32
                if (OP == 17) a[i] = __builtin_amdgcn_sad_u8(a[(i + 1) & 7], b, a[i]);
                if (OP == 18) a[i] = __builtin_amdgcn_sad_u16(a[(i + 1) & 7], b, a[i]);
                if (OP == 19) q[i] = __builtin_amdgcn_qsad_pk_u16_u8(q[(i + 1) & 7], b, q[i]);
            }
    }
    long long t1 = __builtin_readcyclecounter();
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r += a[i] + (uint32_t)q[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (uint32_t)(t1 - t0);
}

template <int OP> void run(const char* name, int threads)
{
    uint32_t* d; hipMalloc(&d, 256 * 1024 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, d, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, d, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint32_t cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
    double instr_per_wave = (double)N_ITERS * UNROLL;
    int waves_per_simd = threads / 256;
    printf("%-22s threads/WG %4d: %.2f s_memtime-cycles per instr per wave; wall %.3f ms -> %.2f ns/instr/SIMD-slot\n",
           name, threads, cyc / instr_per_wave, ms, ms * 1e6 / (instr_per_wave * waves_per_simd));
    hipFree(d);
}

int main()
{
    for (int th : {256, 1024})
    {
        if (th == 256) {
            run<0>("v_sad_u8", 256); run<1>("v_sad_u16", 256); run<2>("v_qsad_pk_u16_u8", 256); run<3>("v_add+xor (2 ops)", 256);
            run<4>("v_alignbit", 256); run<5>("v_sad_hi_u8", 256); run<6>("v_msad_u8", 256); run<7>("dpp add", 256);
            run<8>("v_dot4_i32_i8", 256); run<9>("v_dot2_i32_i16", 256); run<10>("v_mad_i32_i24", 256); run<11>("v_alignbyte", 256);
            run<12>("v_perm_b32", 256); run<13>("mul_lo_u32 + add", 256); run<14>("v_pk_mad_i16", 256); run<15>("v_dot4_u32_u8", 256); run<16>("v_add3_u32", 256);
            run<17>("v_sad_u8 alone", 256); run<18>("v_sad_u16 alone", 256); run<19>("v_qsad_pk_u16_u8 alone", 256);
        } else {
            run<0>("v_sad_u8", 1024); run<1>("v_sad_u16", 1024); run<2>("v_qsad_pk_u16_u8", 1024); run<3>("v_add+xor (2 ops)", 1024);
            run<4>("v_alignbit", 1024); run<5>("v_sad_hi_u8", 1024); run<6>("v_msad_u8", 1024); run<7>("dpp add", 1024);
            run<8>("v_dot4_i32_i8", 1024); run<9>("v_dot2_i32_i16", 1024); run<10>("v_mad_i32_i24", 1024); run<11>("v_alignbyte", 1024);
            run<12>("v_perm_b32", 1024); run<13>("mul_lo_u32 + add", 1024); run<14>("v_pk_mad_i16", 1024); run<15>("v_dot4_u32_u8", 1024); run<16>("v_add3_u32", 1024);
            run<17>("v_sad_u8 alone", 1024); run<18>("v_sad_u16 alone", 1024); run<19>("v_qsad_pk_u16_u8 alone", 1024);
        }
    }
    return 0;
}
