// interp_kernels.hip - batched HEVC sub-pel interpolation on gfx950.
//
// Reference semantics: source/common/ipfilter.cpp - interp_horiz_pp_c :79-118, interp_horiz_ps_c
// :120-162 (isRowExt adds N-1 rows above/below), interp_vert_pp_c :164-203, _ps :205-239, _sp
// :241-282, _ss :284-317, interp_hv_pp_c :362-369, filterPixelToShort_c :40-57; taps
// constants.cpp:250-268.  Every variant forms (int16_t)((sum + offset) >> shift) BEFORE clipping.
//
// Mapping: one workgroup per block ("job").  The source tile including its (N-1)-sample apron is
// staged once in LDS with row-contiguous loads, each output sample then reads its N taps from LDS;
// the hv variant keeps the horizontally filtered 14-bit intermediate in LDS as well, so the source is
// read from HBM/L2 once and nothing but the final block is written.
#include "common.h"

#include <type_traits>
#include <cstdlib>

namespace x265hip {

__constant__ int16_t kLumaTaps[4][8] = {
    { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
    { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
__constant__ int16_t kChromaTaps[8][4] = {
    { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
    { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };

struct IPArgs
{
    const uint8_t* src; long srcStride;      // element strides
    uint8_t* dst; long dstStride;
    const x265hip_job* jobs;
    int w, h, depth;
};

constexpr int IF_PREC = 14, IF_FPREC = 6, IF_OFFS = 1 << (IF_PREC - 1);
constexpr int TILE_PITCH = 64 + 8;            // int16 elements per LDS tile row (64 + 7 apron, padded)

template <int N> __device__ __forceinline__ int tap(int idx, int t) { return N == 8 ? kLumaTaps[idx][t] : kChromaTaps[idx][t]; }

__device__ __forceinline__ int clip_val16(int v, int maxVal)
{
    const int16_t s = (int16_t)v;                 // the reference narrows to int16_t first
    return s < 0 ? 0 : (s > maxVal ? maxVal : s);
}

// SRC_SHORT: source samples are int16 (vsp / vss), else pixels.  DST_SHORT likewise.
template <typename Px, int KIND, int N>
__global__ void __launch_bounds__(256) interp_kernel(IPArgs a)
{
    constexpr bool SRC_SHORT = KIND == X265HIP_IP_VSP || KIND == X265HIP_IP_VSS;
    constexpr bool DST_SHORT = KIND == X265HIP_IP_HPS || KIND == X265HIP_IP_VPS || KIND == X265HIP_IP_VSS || KIND == X265HIP_IP_P2S;
    constexpr bool HORIZ = KIND == X265HIP_IP_HPP || KIND == X265HIP_IP_HPS || KIND == X265HIP_IP_HVPP;
    constexpr bool VERT = KIND == X265HIP_IP_VPP || KIND == X265HIP_IP_VPS || KIND == X265HIP_IP_VSP || KIND == X265HIP_IP_VSS || KIND == X265HIP_IP_HVPP;
    typedef typename std::conditional<SRC_SHORT, int16_t, Px>::type S;
    typedef typename std::conditional<DST_SHORT, int16_t, Px>::type Dt;
    __shared__ int16_t tile[(64 + 8) * TILE_PITCH];
    __shared__ int16_t immed[(64 + 8) * 64];

    const x265hip_job jb = a.jobs[blockIdx.x];
    const S* src = reinterpret_cast<const S*>(a.src) + jb.off[0];
    Dt* dst = reinterpret_cast<Dt*>(a.dst) + jb.off[1];
    const int w = a.w, h = a.h, depth = a.depth;
    const int maxVal = (1 << depth) - 1;
    const int headRoom = IF_PREC - depth;
    const int tid = threadIdx.x, nth = blockDim.x;

    if (KIND == X265HIP_IP_P2S)
    {
        if ((w & 3) == 0)
        {
            // 4 samples per thread and step: one packed source dword (two for 16-bit pixels), two packed int16 destination dwords
            const int qpr = w >> 2;
            for (int q = tid; q < qpr * h; q += nth)
            {
                const int y = q / qpr, x = (q - y * qpr) * 4;
                const uint8_t* sp = reinterpret_cast<const uint8_t*>(src + (long)y * a.srcStride + x);
                int c[4];
                if (sizeof(S) == 1)
                {
                    const uint32_t v = ld_u32(sp);
                    c[0] = v & 0xff; c[1] = (v >> 8) & 0xff; c[2] = (v >> 16) & 0xff; c[3] = v >> 24;
                }
                else
                {
                    const uint32_t v0 = ld_u32(sp), v1 = ld_u32(sp + 4);
                    c[0] = v0 & 0xffff; c[1] = v0 >> 16; c[2] = v1 & 0xffff; c[3] = v1 >> 16;
                }
                uint32_t r[4];
#pragma unroll
                for (int k = 0; k < 4; k++) r[k] = (uint32_t)(uint16_t)(int16_t)((int16_t)(c[k] << headRoom) - (int16_t)IF_OFFS);
                uint8_t* dp = reinterpret_cast<uint8_t*>(dst + (long)y * a.dstStride + x);
                *reinterpret_cast<u32_unaligned*>(dp) = r[0] | (r[1] << 16);
                *reinterpret_cast<u32_unaligned*>(dp + 4) = r[2] | (r[3] << 16);
            }
            return;
        }
        for (int i = tid; i < w * h; i += nth)
        {
            const int y = i / w, x = i - y * w;
            const int16_t v = (int16_t)((int)src[(long)y * a.srcStride + x] << headRoom);
            dst[(long)y * a.dstStride + x] = (Dt)(int16_t)(v - (int16_t)IF_OFFS);
        }
        return;
    }

    const bool rowExt = (KIND == X265HIP_IP_HPS && jb.arg[1] != 0) || KIND == X265HIP_IP_HVPP;
    const int apronX = HORIZ ? N / 2 - 1 : 0;
    const int apronY = (VERT || rowExt) ? N / 2 - 1 : 0;
    const int tw = w + (HORIZ ? N - 1 : 0);
    const int th = h + ((VERT || rowExt) ? N - 1 : 0);
    const S* org = src - (long)apronY * a.srcStride - apronX;
    for (int i = tid; i < tw * th; i += nth)
    {
        const int y = i / tw, x = i - y * tw;
        tile[y * TILE_PITCH + x] = (int16_t)org[(long)y * a.srcStride + x];
    }
    __syncthreads();

    const int idx0 = jb.arg[0];
    if (KIND == X265HIP_IP_HPP || KIND == X265HIP_IP_HPS)
    {
        const int oh = th;                                   // h, or h + N - 1 with row extension
        const int shiftPS = IF_FPREC - headRoom;
        const int offPS = -(IF_OFFS << shiftPS);
        for (int i = tid; i < w * oh; i += nth)
        {
            const int y = i / w, x = i - y * w;
            int sum = 0;
#pragma unroll
            for (int t = 0; t < N; t++) sum += (int)(uint16_t)tile[y * TILE_PITCH + x + t] * tap<N>(idx0, t);
            if (KIND == X265HIP_IP_HPP)
                dst[(long)y * a.dstStride + x] = (Dt)clip_val16((sum + (1 << (IF_FPREC - 1))) >> IF_FPREC, maxVal);
            else
                dst[(long)y * a.dstStride + x] = (Dt)(int16_t)((sum + offPS) >> shiftPS);
        }
        return;
    }
    if (KIND == X265HIP_IP_HVPP)
    {
        // horizontal ps pass over h + N - 1 rows into the intermediate (pitch w)
        const int shiftPS = IF_FPREC - headRoom;
        const int offPS = -(IF_OFFS << shiftPS);
        for (int i = tid; i < w * th; i += nth)
        {
            const int y = i / w, x = i - y * w;
            int sum = 0;
#pragma unroll
            for (int t = 0; t < N; t++) sum += (int)(uint16_t)tile[y * TILE_PITCH + x + t] * tap<N>(idx0, t);
            immed[y * 64 + x] = (int16_t)((sum + offPS) >> shiftPS);
        }
        __syncthreads();
        const int idxY = jb.arg[1];
        const int shift = IF_FPREC + headRoom;
        const int offset = (1 << (shift - 1)) + (IF_OFFS << IF_FPREC);
        for (int i = tid; i < w * h; i += nth)
        {
            const int y = i / w, x = i - y * w;
            int sum = 0;
#pragma unroll
            for (int t = 0; t < N; t++) sum += (int)immed[(y + t) * 64 + x] * tap<N>(idxY, t);
            dst[(long)y * a.dstStride + x] = (Dt)clip_val16((sum + offset) >> shift, maxVal);
        }
        return;
    }
    // vertical kinds
    for (int i = tid; i < w * h; i += nth)
    {
        const int y = i / w, x = i - y * w;
        int sum = 0;
#pragma unroll
        for (int t = 0; t < N; t++)
        {
            const int16_t s = tile[(y + t) * TILE_PITCH + x];
            sum += (SRC_SHORT ? (int)s : (int)(uint16_t)s) * tap<N>(idx0, t);
        }
        if (KIND == X265HIP_IP_VPP)
            dst[(long)y * a.dstStride + x] = (Dt)clip_val16((sum + (1 << (IF_FPREC - 1))) >> IF_FPREC, maxVal);
        else if (KIND == X265HIP_IP_VPS)
        {
            const int shift = IF_FPREC - headRoom;
            dst[(long)y * a.dstStride + x] = (Dt)(int16_t)((sum - (IF_OFFS << shift)) >> shift);
        }
        else if (KIND == X265HIP_IP_VSP)
        {
            const int shift = IF_FPREC + headRoom;
            dst[(long)y * a.dstStride + x] = (Dt)clip_val16((sum + (1 << (shift - 1)) + (IF_OFFS << IF_FPREC)) >> shift, maxVal);
        }
        else
            dst[(long)y * a.dstStride + x] = (Dt)(int16_t)(sum >> IF_FPREC);
    }
}

// ------------------------------------------------------------------------------------------------
// Fast path (block width a multiple of 4, every kind but P2S): no LDS, no barriers.  A thread owns a strip of
// 4 pixels x STRIP_ROWS rows of one job; a 256-thread workgroup carries as many jobs as fit, so small blocks
// (8x8 chroma: 2 strips) do not pay one workgroup launch each.  Arithmetic is packed:
//   horizontal, 8-bit pixels : v_dot4_i32_i8 on (pixel - 128) bytes - the bias 128 * sum(taps) = 8192 is the
//                              accumulator's start value; the 4 shifted windows of a row come from 3 dwords by
//                              v_alignbyte
//   horizontal, 16-bit pixels: v_dot2_i32_i16 on pixel pairs (odd windows by v_alignbyte 2)
//   vertical (all sources)   : v_dot2_i32_i16 on (row r, row r+1) pairs built with one v_perm_b32 per pixel from
//                              two row dwords (bytes zero-extended, halfwords as they are, or the 14-bit
//                              horizontal intermediates of the hv kind, which never leave registers)
// Both give exact int32 sums, so the rounding code below is the same as the generic kernel's.
__constant__ uint32_t kLumaDot4[4][2] = {          // taps as 4 signed bytes: {c0..c3}, {c4..c7}
    { 0x40000000u, 0x00000000u }, { 0x3af604ffu, 0x0001fb11u }, { 0x28f504ffu, 0xff04f528u }, { 0x11fb0100u, 0xff04f63au } };
__constant__ uint32_t kChromaDot4[8] = {
    0x00004000u, 0xfe0a3afeu, 0xfe1036fcu, 0xfc1c2efau, 0xfc2424fcu, 0xfa2e1cfcu, 0xfc3610feu, 0xfe3a0afeu };

template <int N> __device__ __forceinline__ uint32_t tap_pair(int idx, int j)          // (c[2j], c[2j+1]) as int16 x 2
{
    const int a = tap<N>(idx, 2 * j), b = tap<N>(idx, 2 * j + 1);
    return ((uint32_t)a & 0xffffu) | ((uint32_t)b << 16);
}

typedef short v2i16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(v2i16, a), __builtin_bit_cast(v2i16, b), c, false);
}

constexpr int STRIP_ROWS = 8;

template <typename Px, int KIND, int N>
__global__ void __launch_bounds__(256) interp_strip_kernel(IPArgs a, int njobs, int stripsPerJob, int jobsPerWg)
{
    constexpr bool SRC_SHORT = KIND == X265HIP_IP_VSP || KIND == X265HIP_IP_VSS;
    constexpr bool DST_SHORT = KIND == X265HIP_IP_HPS || KIND == X265HIP_IP_VPS || KIND == X265HIP_IP_VSS;
    constexpr bool HORIZ = KIND == X265HIP_IP_HPP || KIND == X265HIP_IP_HPS || KIND == X265HIP_IP_HVPP;
    constexpr bool VERT = !HORIZ || KIND == X265HIP_IP_HVPP;
    typedef typename std::conditional<SRC_SHORT, int16_t, Px>::type S;
    typedef typename std::conditional<DST_SHORT, int16_t, Px>::type Dt;
    constexpr int SB = sizeof(S);
    constexpr int HALF = N / 2 - 1;
    constexpr int NSRC = STRIP_ROWS + (VERT ? N - 1 : 0);           // source rows a strip touches

    const int jw = threadIdx.x / stripsPerJob;
    const int strip = threadIdx.x - jw * stripsPerJob;
    const int job = blockIdx.x * jobsPerWg + jw;
    if (jw >= jobsPerWg || job >= njobs) return;
    const x265hip_job jb = a.jobs[job];
    const int w = a.w, depth = a.depth;
    const bool rowExt = KIND == X265HIP_IP_HPS && jb.arg[1] != 0;
    const int hEff = a.h + (rowExt ? N - 1 : 0);                     // rows this job writes
    const int spr = w >> 2;
    const int sy = (strip / spr) * STRIP_ROWS, sx = (strip - (strip / spr) * spr) * 4;
    const int nr = hEff - sy < STRIP_ROWS ? hEff - sy : STRIP_ROWS;
    const int maxVal = (1 << depth) - 1, headRoom = IF_PREC - depth;
    const int idx0 = jb.arg[0], idx1 = jb.arg[1];

    const uint8_t* src = reinterpret_cast<const uint8_t*>(reinterpret_cast<const S*>(a.src) + jb.off[0]);
    const long ssB = a.srcStride * SB;
    // first source row / column this strip reads
    const long row0 = sy - ((VERT || rowExt) ? HALF : 0);
    const uint8_t* sp = src + row0 * ssB + (long)(sx - (HORIZ ? HALF : 0)) * SB;
    const int lastRow = (VERT ? nr + N - 2 : nr - 1);               // rows past it are clamped (never stored)

    // ---- stage 1: per source row, the 4 samples of the strip (vertical kinds) or the 4 horizontal sums ----
    int hsum[HORIZ ? NSRC : 1][4];
    uint32_t raw[(!HORIZ) ? NSRC : 1][2];
    if (HORIZ)
    {
        uint32_t c03 = 0, c47 = 0, cp[4] = { 0, 0, 0, 0 };
        if (sizeof(Px) == 1) { c03 = N == 8 ? kLumaDot4[idx0][0] : kChromaDot4[idx0]; c47 = N == 8 ? kLumaDot4[idx0][1] : 0; }
        else
        {
#pragma unroll
            for (int j = 0; j < N / 2; j++) cp[j] = tap_pair<N>(idx0, j);
        }
#pragma unroll
        for (int r = 0; r < NSRC; r++)
        {
            const uint8_t* rp = sp + (long)(r < lastRow ? r : lastRow) * ssB;
            if (sizeof(Px) == 1)
            {
                uint32_t w0 = ld_u32(rp) ^ 0x80808080u, w1 = ld_u32(rp + 4) ^ 0x80808080u, w2 = N == 8 ? ld_u32(rp + 8) ^ 0x80808080u : 0;
#pragma unroll
                for (int x = 0; x < 4; x++)
                {
                    const uint32_t lo = x ? __builtin_amdgcn_alignbyte(w1, w0, x) : w0;
                    int sacc = __builtin_amdgcn_sdot4((int)lo, (int)c03, 8192, false);
                    if (N == 8)
                    {
                        const uint32_t hi = x ? __builtin_amdgcn_alignbyte(w2, w1, x) : w1;
                        sacc = __builtin_amdgcn_sdot4((int)hi, (int)c47, sacc, false);
                    }
                    hsum[r][x] = sacc;
                }
            }
            else
            {
                // N + 3 samples = (N + 4) / 2 dwords: even windows are the dwords themselves, odd ones are shifted by one sample
                constexpr int ND = (N + 4) / 2;
                uint32_t d[ND];
#pragma unroll
                for (int k = 0; k < ND; k++) d[k] = ld_u32(rp + 4 * k);
#pragma unroll
                for (int x = 0; x < 4; x++)
                {
                    int sacc = 0;
#pragma unroll
                    for (int j = 0; j < N / 2; j++)
                    {
                        const int k = (x >> 1) + j;
                        const uint32_t pr = (x & 1) ? __builtin_amdgcn_alignbyte(d[k + 1], d[k], 2) : d[k];
                        sacc = dot2(pr, cp[j], sacc);
                    }
                    hsum[r][x] = sacc;
                }
            }
        }
    }
    else
    {
#pragma unroll
        for (int r = 0; r < NSRC; r++)
        {
            const uint8_t* rp = sp + (long)(r < lastRow ? r : lastRow) * ssB;
            raw[r][0] = ld_u32(rp);
            raw[r][1] = SB == 2 ? ld_u32(rp + 4) : 0;
        }
    }

    // ---- stage 2: finish the row (horizontal kinds) or run the vertical taps over row pairs ----
    Dt* dst = reinterpret_cast<Dt*>(a.dst) + jb.off[1] + (long)sy * a.dstStride + sx;
    auto store4 = [&](const int y, const int (&v)[4])
    {
        uint8_t* dp = reinterpret_cast<uint8_t*>(dst + (long)y * a.dstStride);
        if (sizeof(Dt) == 1)
            *reinterpret_cast<u32_unaligned*>(dp) = (uint32_t)(v[0] & 0xff) | ((uint32_t)(v[1] & 0xff) << 8) | ((uint32_t)(v[2] & 0xff) << 16) | ((uint32_t)v[3] << 24);
        else
        {
            *reinterpret_cast<u32_unaligned*>(dp) = ((uint32_t)v[0] & 0xffffu) | ((uint32_t)v[1] << 16);
            *reinterpret_cast<u32_unaligned*>(dp + 4) = ((uint32_t)v[2] & 0xffffu) | ((uint32_t)v[3] << 16);
        }
    };
    if (KIND == X265HIP_IP_HPP || KIND == X265HIP_IP_HPS)
    {
        const int shiftPS = IF_FPREC - headRoom, offPS = -(IF_OFFS << shiftPS);
#pragma unroll
        for (int y = 0; y < STRIP_ROWS; y++)
        {
            if (y >= nr) break;
            int v[4];
#pragma unroll
            for (int x = 0; x < 4; x++)
                v[x] = KIND == X265HIP_IP_HPP ? clip_val16((hsum[y][x] + (1 << (IF_FPREC - 1))) >> IF_FPREC, maxVal)
                                              : (int)(int16_t)((hsum[y][x] + offPS) >> shiftPS);
            store4(y, v);
        }
        return;
    }
    // vertical taps
    const int idxV = KIND == X265HIP_IP_HVPP ? idx1 : idx0;
    uint32_t cv[N / 2];
#pragma unroll
    for (int j = 0; j < N / 2; j++) cv[j] = tap_pair<N>(idxV, j);
    uint32_t pairs[NSRC - 1][4];                                     // (row r, row r + 1) at each of the 4 columns
    if (KIND == X265HIP_IP_HVPP)
    {
        const int shiftPS = IF_FPREC - headRoom, offPS = -(IF_OFFS << shiftPS);
        int im[NSRC][4];
#pragma unroll
        for (int r = 0; r < NSRC; r++)
#pragma unroll
            for (int x = 0; x < 4; x++) im[r][x] = (hsum[r][x] + offPS) >> shiftPS;
#pragma unroll
        for (int r = 0; r < NSRC - 1; r++)
#pragma unroll
            for (int x = 0; x < 4; x++) pairs[r][x] = __builtin_amdgcn_perm((uint32_t)im[r + 1][x], (uint32_t)im[r][x], 0x05040100u);
    }
    else
    {
#pragma unroll
        for (int r = 0; r < NSRC - 1; r++)
#pragma unroll
            for (int x = 0; x < 4; x++)
            {
                if (SB == 1)     // byte x of both rows, zero-extended to halfwords
                    pairs[r][x] = __builtin_amdgcn_perm(raw[r + 1][0], raw[r][0], 0x0c000c00u | (uint32_t)x | ((uint32_t)(4 + x) << 16));
                else             // halfword x of both rows
                    pairs[r][x] = __builtin_amdgcn_perm(raw[r + 1][x >> 1], raw[r][x >> 1], (x & 1) ? 0x07060302u : 0x05040100u);
            }
    }
#pragma unroll
    for (int y = 0; y < STRIP_ROWS; y++)
    {
        if (y >= nr) break;
        int v[4];
#pragma unroll
        for (int x = 0; x < 4; x++)
        {
            int sum = 0;
#pragma unroll
            for (int j = 0; j < N / 2; j++) sum = dot2(pairs[y + 2 * j][x], cv[j], sum);
            if (KIND == X265HIP_IP_VPP)
                v[x] = clip_val16((sum + (1 << (IF_FPREC - 1))) >> IF_FPREC, maxVal);
            else if (KIND == X265HIP_IP_VPS)
            {
                const int shift = IF_FPREC - headRoom;
                v[x] = (int)(int16_t)((sum - (IF_OFFS << shift)) >> shift);
            }
            else if (KIND == X265HIP_IP_VSS)
                v[x] = (int)(int16_t)(sum >> IF_FPREC);
            else                                                      // VSP and the second stage of HVPP
            {
                const int shift = IF_FPREC + headRoom;
                v[x] = clip_val16((sum + (1 << (shift - 1)) + (IF_OFFS << IF_FPREC)) >> shift, maxVal);
            }
        }
        store4(y, v);
    }
}

template <typename Px, int KIND, int N>
static void launch_strip(const IPArgs& a, int njobs, hipStream_t s)
{
    const int hEffMax = a.h + (KIND == X265HIP_IP_HPS ? N - 1 : 0);
    const int spj = (a.w >> 2) * ((hEffMax + STRIP_ROWS - 1) / STRIP_ROWS);          // <= 16 * 9 = 144
    const int jpw = 256 / spj > 0 ? 256 / spj : 1;
    const int threads = ((spj * jpw + 63) / 64) * 64;
    hipLaunchKernelGGL((interp_strip_kernel<Px, KIND, N>), dim3((njobs + jpw - 1) / jpw), dim3(threads), 0, s, a, njobs, spj, jpw);
}

template <typename Px, int N>
static int launch_ip(int kind, const IPArgs& a, int njobs, hipStream_t s)
{
    const int threads = a.w * a.h <= 256 ? 64 : 256;
    static const bool forceGeneric = getenv("X265HIP_INTERP_GENERIC") != nullptr;    // A/B switch, read once: this launcher sits under every table stub
    const bool strip = (a.w & 3) == 0 && kind != X265HIP_IP_P2S && !forceGeneric;
#define CASE(K) case K: if (strip && K != X265HIP_IP_P2S) launch_strip<Px, K == X265HIP_IP_P2S ? X265HIP_IP_HPP : K, N>(a, njobs, s); \
                        else hipLaunchKernelGGL((interp_kernel<Px, K, N>), dim3(njobs), dim3(threads), 0, s, a); break;
    switch (kind)
    {
        CASE(X265HIP_IP_HPP) CASE(X265HIP_IP_HPS) CASE(X265HIP_IP_VPP) CASE(X265HIP_IP_VPS)
        CASE(X265HIP_IP_VSP) CASE(X265HIP_IP_VSS) CASE(X265HIP_IP_HVPP) CASE(X265HIP_IP_P2S)
    default: set_error("interp_batch: unknown kind %d", kind); return X265HIP_EINVAL;
    }
#undef CASE
    X265HIP_TRY(hipGetLastError());
    return 0;
}

} // namespace x265hip

using namespace x265hip;

extern "C" int x265hip_interp_batch(int kind, int depth, int taps, int w, int h, x265hip_plane src, x265hip_plane dst,
                                    const x265hip_job* jobs, int njobs, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!src.base || !dst.base || !jobs || njobs < 0) { set_error("interp_batch: NULL operand"); return X265HIP_EINVAL; }
    if (njobs == 0) return 0;
    if (taps != 8 && taps != 4) { set_error("interp_batch: taps must be 8 (luma) or 4 (chroma)"); return X265HIP_EINVAL; }
    if (w < 2 || h < 2 || w > 64 || h > 64) { set_error("interp_batch: block %dx%d unsupported", w, h); return X265HIP_EINVAL; }
    if (depth != 8 && depth != 10 && depth != 12) { set_error("interp_batch: depth %d", depth); return X265HIP_EINVAL; }
    IPArgs a;
    a.src = (const uint8_t*)src.base; a.srcStride = src.stride; a.dst = (uint8_t*)dst.base; a.dstStride = dst.stride;
    a.jobs = jobs; a.w = w; a.h = h; a.depth = depth;
    hipStream_t s = (hipStream_t)stream;
    if (depth == 8) return taps == 8 ? launch_ip<uint8_t, 8>(kind, a, njobs, s) : launch_ip<uint8_t, 4>(kind, a, njobs, s);
    return taps == 8 ? launch_ip<uint16_t, 8>(kind, a, njobs, s) : launch_ip<uint16_t, 4>(kind, a, njobs, s);
}
