// pixel_kernels.hip - batched pixel-compare primitives on gfx950: SAD, SATD, SA8D, SSE, psyCost.
//
// Reference semantics (source/common/pixel.cpp): sad :40-55, sse :167-186, satd_4x4/8x4 :210-297 (size
// map :1131-1155), sa8d :299-377 (8x8 rounded (s+2)>>2; 16x16 rounds ONCE over four 8x8 sums; larger =
// sum of 16x16 units), psyCost_pp :726-757.
//
// Mapping: one candidate ("job") per group of G lanes, G in {4,8,16,32,64} picked from the block
// size so small PUs pack several candidates into a wavefront and large PUs own a whole one.
//   SAD/SSE : lane = 4-pixel chunk (one dword for u8), v_sad_u8 / v_sad_u16, xor-shuffle tree.
//   SATD    : lane = one 4x4 tile, Hadamard entirely in registers, shuffle tree for the sum.
//   SA8D    : 8 lanes = one 8x8 tile (lane = row): horizontal 8-point Hadamard in registers,
//             vertical 8-point Hadamard as a 3-stage butterfly ACROSS lanes (wave shuffles).
#include "common.h"

namespace x265hip {

struct CmpArgs
{
    const uint8_t* a; long aStride;            // element strides
    const uint8_t* b; long bStride;
    const long long* aOff; long long aStep;    // element offsets
    const long long* bOff; long long bStep;
    int w, h, njobs;
    unsigned long long* out;
};

template <typename Px> __device__ __forceinline__ void load4(const Px* p, int v[4]);
template <> __device__ __forceinline__ void load4<uint8_t>(const uint8_t* p, int v[4])
{
    uint32_t d = ld_u32(p);
    v[0] = d & 0xff; v[1] = (d >> 8) & 0xff; v[2] = (d >> 16) & 0xff; v[3] = d >> 24;
}
template <> __device__ __forceinline__ void load4<uint16_t>(const uint16_t* p, int v[4])
{
    uint32_t d0 = *reinterpret_cast<const u32_align2*>(p), d1 = *reinterpret_cast<const u32_align2*>(p + 2);
    v[0] = d0 & 0xffff; v[1] = d0 >> 16; v[2] = d1 & 0xffff; v[3] = d1 >> 16;
}

// ------------------------------------------------------------------------------ SAD / SSE
template <typename Px, int G, bool SSE>
__global__ void __launch_bounds__(256) sad_sse_kernel(CmpArgs c)
{
    const int lane = threadIdx.x & 63;
    const int sub = lane & (G - 1);
    const long job = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (64 / G) + lane / G;
    const bool live = job < c.njobs;
    const int cpr = c.w >> 2;                           // 4-pixel chunks per row
    const int rpi = G / cpr > 0 ? G / cpr : 1;          // rows covered per iteration
    const int lr = sub / cpr, lc = sub - lr * cpr;
    unsigned long long acc = 0;
    if (live && lr < rpi)
    {
        const Px* pa = reinterpret_cast<const Px*>(c.a) + (c.aOff ? c.aOff[job] : job * c.aStep);
        const Px* pb = reinterpret_cast<const Px*>(c.b) + (c.bOff ? c.bOff[job] : job * c.bStep);
        uint32_t acc32 = 0;
        for (int r = lr; r < c.h; r += rpi)
        {
            const Px* ra = pa + (long)r * c.aStride + lc * 4;
            const Px* rb = pb + (long)r * c.bStride + lc * 4;
            if (!SSE)
            {
                if (sizeof(Px) == 1) acc32 = sad4_u8(ld_u32(ra), ld_u32(rb), acc32);
                else
                {
                    acc32 = sad2_u16(*reinterpret_cast<const u32_align2*>(ra), *reinterpret_cast<const u32_align2*>(rb), acc32);
                    acc32 = sad2_u16(*reinterpret_cast<const u32_align2*>(ra + 2), *reinterpret_cast<const u32_align2*>(rb + 2), acc32);
                }
            }
            else
            {
                int va[4], vb[4];
                load4<Px>(ra, va); load4<Px>(rb, vb);
#pragma unroll
                for (int i = 0; i < 4; i++) { int d = va[i] - vb[i]; acc += (unsigned)(d * d); }
            }
        }
        if (!SSE) acc = acc32;
    }
    acc = group_sum<G>(acc);
    if (live && sub == 0) c.out[job] = acc;
}

// ------------------------------------------------------------------------------ SATD
__device__ __forceinline__ int hadamard4x4_abs(const int d[4][4])
{
    int t[4][4];
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        int s0 = d[y][0] + d[y][1], s1 = d[y][0] - d[y][1], s2 = d[y][2] + d[y][3], s3 = d[y][2] - d[y][3];
        t[y][0] = s0 + s2; t[y][1] = s1 + s3; t[y][2] = s0 - s2; t[y][3] = s1 - s3;
    }
    int acc = 0;
#pragma unroll
    for (int x = 0; x < 4; x++)
    {
        int s0 = t[0][x] + t[1][x], s1 = t[0][x] - t[1][x], s2 = t[2][x] + t[3][x], s3 = t[2][x] - t[3][x];
        acc += abs(s0 + s2) + abs(s1 + s3) + abs(s0 - s2) + abs(s1 - s3);
    }
    return acc;
}

// PSY4 = psyCost for the 4x4 CU: |(satd(src,0) - sad(src,0)>>2) - (satd(rec,0) - sad(rec,0)>>2)|
template <typename Px, int G, bool PSY4>
__global__ void __launch_bounds__(256) satd_kernel(CmpArgs c)
{
    const int lane = threadIdx.x & 63;
    const int sub = lane & (G - 1);
    const long job = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (64 / G) + lane / G;
    const bool live = job < c.njobs;
    const int tpr = c.w >> 2, ntiles = tpr * (c.h >> 2);
    int acc = 0;
    if (live)
    {
        const Px* pa = reinterpret_cast<const Px*>(c.a) + (c.aOff ? c.aOff[job] : job * c.aStep);
        const Px* pb = reinterpret_cast<const Px*>(c.b) + (c.bOff ? c.bOff[job] : job * c.bStep);
        for (int t = sub; t < ntiles; t += G)
        {
            const int ty = t / tpr, tx = t - ty * tpr;
            const Px* ta = pa + (long)(ty * 4) * c.aStride + tx * 4;
            const Px* tb = pb + (long)(ty * 4) * c.bStride + tx * 4;
            int d[4][4];
            if (!PSY4)
            {
#pragma unroll
                for (int y = 0; y < 4; y++)
                {
                    int va[4], vb[4];
                    load4<Px>(ta + (long)y * c.aStride, va); load4<Px>(tb + (long)y * c.bStride, vb);
#pragma unroll
                    for (int x = 0; x < 4; x++) d[y][x] = va[x] - vb[x];
                }
                acc += hadamard4x4_abs(d) >> 1;
            }
            else
            {
                int e[2];
#pragma unroll
                for (int s = 0; s < 2; s++)
                {
                    int sum = 0;
#pragma unroll
                    for (int y = 0; y < 4; y++)
                    {
                        int v[4];
                        if (s == 0) load4<Px>(ta + (long)y * c.aStride, v); else load4<Px>(tb + (long)y * c.bStride, v);
#pragma unroll
                        for (int x = 0; x < 4; x++) { d[y][x] = v[x]; sum += v[x]; }
                    }
                    e[s] = (hadamard4x4_abs(d) >> 1) - (sum >> 2);
                }
                acc += abs(e[0] - e[1]);
            }
        }
    }
    acc = group_sum<G>(acc);
    if (live && sub == 0) c.out[job] = (unsigned long long)(unsigned)acc;
}

// ------------------------------------------------------------------------------ SA8D / psyCost
// raw 8x8 Hadamard abs-sum of (rowvals) distributed one row per lane over 8 consecutive lanes;
// result valid in all 8 lanes.
__device__ __forceinline__ int hadamard8x8_rows(int v[8], int lane)
{
    // horizontal: 3 butterfly stages inside the lane
#pragma unroll
    for (int step = 1; step < 8; step <<= 1)
#pragma unroll
        for (int i = 0; i < 8; i += step << 1)
#pragma unroll
            for (int j = i; j < i + step; j++)
            {
                int p = v[j], q = v[j + step];
                v[j] = p + q; v[j + step] = p - q;
            }
    // vertical: the same butterfly across the 8 lanes that hold the 8 rows
#pragma unroll
    for (int s = 1; s < 8; s <<= 1)
    {
        const bool hi = (lane & s) != 0;
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            int o = __shfl_xor(v[k], s, 64);
            v[k] = hi ? o - v[k] : v[k] + o;
        }
    }
    int acc = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) acc += abs(v[k]);
    return group_sum<8>(acc);
}

template <typename Px> __device__ __forceinline__ void load8(const Px* p, int v[8])
{
    load4<Px>(p, v); load4<Px>(p + 4, v + 4);
}

// G lanes per job (8,16,32,64); UNIT16: round once per 16x16 unit, else once per 8x8.
// Tiles are enumerated so that the 4 tiles of a 16x16 unit are consecutive.
template <typename Px, int G, bool UNIT16, bool PSY>
__global__ void __launch_bounds__(256) sa8d_kernel(CmpArgs c)
{
    const int lane = threadIdx.x & 63;
    const int sub = lane & (G - 1);
    const long job = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (64 / G) + lane / G;
    const bool live = job < c.njobs;
    const long jj = live ? job : 0;
    const Px* pa = reinterpret_cast<const Px*>(c.a) + (c.aOff ? c.aOff[jj] : jj * c.aStep);
    const Px* pb = reinterpret_cast<const Px*>(c.b) + (c.bOff ? c.bOff[jj] : jj * c.bStep);
    const int row = sub & 7;
    const int ntiles = (c.w >> 3) * (c.h >> 3);
    constexpr int TPI = G / 8;                            // tiles in flight per job per iteration
    int total = 0;
    for (int t0 = 0; t0 < ntiles; t0 += TPI)              // uniform trip count across the wave
    {
        const int t = t0 + (sub >> 3);
        int tx, ty;
        if (UNIT16)
        {
            const int upr = c.w >> 4, u = t >> 2, q = t & 3;
            const int uy = u / upr, ux = u - uy * upr;
            tx = ux * 2 + (q & 1); ty = uy * 2 + (q >> 1);
        }
        else
        {
            const int tpr = c.w >> 3;
            ty = t / tpr; tx = t - ty * tpr;
        }
        const bool tl = live && t < ntiles;
        int va[8], vb[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { va[i] = 0; vb[i] = 0; }
        if (tl)
        {
            load8<Px>(pa + (long)(ty * 8 + row) * c.aStride + tx * 8, va);
            load8<Px>(pb + (long)(ty * 8 + row) * c.bStride + tx * 8, vb);
        }
        if (!PSY)
        {
            int d[8];
#pragma unroll
            for (int i = 0; i < 8; i++) d[i] = va[i] - vb[i];
            int raw = hadamard8x8_rows(d, lane);
            if (UNIT16)
            {
                raw += __shfl_xor(raw, 8, 64);
                raw += __shfl_xor(raw, 16, 64);
                // one value per 32 lanes; count it once per job-iteration (sub 0 and, for G=64, sub 32)
                if ((sub & 31) == 0 && tl) total += (raw + 2) >> 2;
            }
            else if (row == 0 && tl)
                total += (raw + 2) >> 2;
        }
        else
        {
            int sa = 0, sb = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) { sa += va[i]; sb += vb[i]; }
            sa = group_sum<8>(sa); sb = group_sum<8>(sb);
            const int ea = ((hadamard8x8_rows(va, lane) + 2) >> 2) - (sa >> 2);
            const int eb = ((hadamard8x8_rows(vb, lane) + 2) >> 2) - (sb >> 2);
            if (row == 0 && tl) total += abs(ea - eb);
        }
    }
    total = group_sum<G>(total);
    if (live && sub == 0) c.out[job] = (unsigned long long)(unsigned)total;
}

template <typename K> static int launch(K kernel, const CmpArgs& c, int G, hipStream_t s)
{
    const int jobsPerBlock = 4 * (64 / G);
    const int blocks = (c.njobs + jobsPerBlock - 1) / jobsPerBlock;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, s, c);
    X265HIP_TRY(hipGetLastError());
    return 0;
}

template <typename Px>
static int dispatch_cmp(int kind, const CmpArgs& c, hipStream_t s)
{
    const int w = c.w, h = c.h;
    if (kind == X265HIP_CMP_SAD || kind == X265HIP_CMP_SSE_PP)
    {
        const int nchunk = (w >> 2) * h;
        const bool sse = kind == X265HIP_CMP_SSE_PP;
#define GO(G) return sse ? launch(sad_sse_kernel<Px, G, true>, c, G, s) : launch(sad_sse_kernel<Px, G, false>, c, G, s)
        if (nchunk <= 8) GO(4);
        if (nchunk <= 32) GO(16);
        GO(64);
#undef GO
    }
    if (kind == X265HIP_CMP_SATD || ((kind == X265HIP_CMP_SA8D) && (w < 8 || h < 8)) || (kind == X265HIP_CMP_PSY_COST && w == 4))
    {
        const int nt = (w >> 2) * (h >> 2);
        if (kind == X265HIP_CMP_PSY_COST) return launch(satd_kernel<Px, 4, true>, c, 4, s);
        if (nt <= 4) return launch(satd_kernel<Px, 4, false>, c, 4, s);
        if (nt <= 16) return launch(satd_kernel<Px, 16, false>, c, 16, s);
        return launch(satd_kernel<Px, 64, false>, c, 64, s);
    }
    if (kind == X265HIP_CMP_SA8D)
    {
        const bool u16 = !(w & 15) && !(h & 15);
        const int nt = (w >> 3) * (h >> 3);
        if (u16) return nt <= 4 ? launch(sa8d_kernel<Px, 32, true, false>, c, 32, s) : launch(sa8d_kernel<Px, 64, true, false>, c, 64, s);
        if (nt == 1) return launch(sa8d_kernel<Px, 8, false, false>, c, 8, s);
        if (nt == 2) return launch(sa8d_kernel<Px, 16, false, false>, c, 16, s);
        return launch(sa8d_kernel<Px, 64, false, false>, c, 64, s);
    }
    if (kind == X265HIP_CMP_PSY_COST)
    {
        const int nt = (w >> 3) * (h >> 3);
        if (nt == 1) return launch(sa8d_kernel<Px, 8, false, true>, c, 8, s);
        if (nt <= 4) return launch(sa8d_kernel<Px, 32, false, true>, c, 32, s);
        return launch(sa8d_kernel<Px, 64, false, true>, c, 64, s);
    }
    set_error("pixelcmp_batch: unknown kind %d", kind);
    return X265HIP_EINVAL;
}

} // namespace x265hip

using namespace x265hip;

extern "C" int x265hip_pixelcmp_batch(int kind, int depth, int w, int h,
                                      const void* a, intptr_t a_stride, const int64_t* a_off, int64_t a_step,
                                      const void* b, intptr_t b_stride, const int64_t* b_off, int64_t b_step,
                                      int njobs, uint64_t* out, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!a || !b || !out || njobs < 0) { set_error("pixelcmp_batch: NULL operand"); return X265HIP_EINVAL; }
    if (njobs == 0) return 0;
    if (w < 4 || h < 4 || w > 64 || h > 64 || (w & 3) || (h & 3)) { set_error("pixelcmp_batch: block %dx%d unsupported", w, h); return X265HIP_EINVAL; }
    if (depth != 8 && depth != 10 && depth != 12) { set_error("pixelcmp_batch: depth %d", depth); return X265HIP_EINVAL; }
    if ((kind == X265HIP_CMP_SA8D || kind == X265HIP_CMP_PSY_COST) && w >= 8 && h >= 8 && ((w & 7) || (h & 7)))
    { set_error("pixelcmp_batch: sa8d/psy need multiples of 8 (got %dx%d)", w, h); return X265HIP_EINVAL; }
    if (kind == X265HIP_CMP_PSY_COST && w != h) { set_error("pixelcmp_batch: psy_cost is square-only"); return X265HIP_EINVAL; }
    CmpArgs c;
    c.a = (const uint8_t*)a; c.aStride = a_stride; c.b = (const uint8_t*)b; c.bStride = b_stride;
    c.aOff = (const long long*)a_off; c.aStep = a_step; c.bOff = (const long long*)b_off; c.bStep = b_step;
    c.w = w; c.h = h; c.njobs = njobs; c.out = (unsigned long long*)out;
    if (depth == 8) return dispatch_cmp<uint8_t>(kind, c, (hipStream_t)stream);
    return dispatch_cmp<uint16_t>(kind, c, (hipStream_t)stream);
}
