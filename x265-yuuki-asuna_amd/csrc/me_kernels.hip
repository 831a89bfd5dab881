// me_kernels.hip - CTU-tiled exhaustive integer motion search on gfx950.
//
// Reference semantics: pu[LUMA_NxN].sad / sad_x3 / sad_x4 (source/common/pixel.cpp:40-119) as
// issued by the full search of MotionEstimate::motionEstimate (source/encoder/motion.cpp:1397-1445:
// every mv of the [-range,range]^2 window in raster order, COPY2_IF_LT strict-less tie-break,
// cost = sad + mvcost).  SAD is additive over sub-blocks, so the 16x16/32x32/64x64 values are the
// exact sums of the 8x8 ones.
//
// Mapping (one workgroup per 64x64 CTU):
//   * the (64+2R)^2 reference window is staged once in LDS (dword-aligned copy of the rows, so HBM is
//     read with coalesced aligned dwords and each reference pixel leaves HBM/L2 once per CTU);
//   * a wavefront owns one mv column (fixed mvx): lane l holds the l-th 8x8 block (z-order) of the
//     CTU in registers and walks DOWN the window row by row.  A window row contributes to the 8
//     vertical displacements it overlaps, so each lane keeps a ring of 8 running SADs: 3 LDS dwords
//     (12 B) + 2 v_alignbit feed 16 v_sad_u8 - register reuse keeps LDS at ~1/5 of its bandwidth;
//   * every row step completes one mv: 8x8 SADs are summed to 16x16 (DPP quad_perm), 32x32 (DPP
//     row_ror) and 64x64 (v_readlane) without touching LDS, written as [ctu][mvy][mvx][pu] so a
//     wavefront stores 256 contiguous bytes, and/or folded into a per-PU running minimum of
//     (cost << 32 | raster index) that is merged across wavefronts with one 64-bit atomicMin.
#include "common.h"

#include <type_traits>

namespace x265hip {

struct MEArgs
{
    const uint8_t* fenc;  long fencStrideB;     // byte strides
    const uint8_t* fref;  long frefStrideB;
    int ctusW;
    int range;            // R
    int rowBytes;         // LDS row pitch (multiple of 4, odd number of dwords)
    int32_t*  surf[4];
    unsigned long long* best[4];
    const uint16_t* costX;
    const uint16_t* costY;
};

// lane -> 8x8 block coordinates inside the CTU, z-order (quad = one 16x16, 16 lanes = one 32x32)
__device__ __forceinline__ void zorder_xy(int lane, int& bx, int& by)
{
    bx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4);
    by = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
}

template <typename Px> struct MECfg;
template <> struct MECfg<uint8_t>  { static constexpr int DWPR = 2; static constexpr int NLD = 3; };   // dwords per 8-px row; dwords loaded
template <> struct MECfg<uint16_t> { static constexpr int DWPR = 4; static constexpr int NLD = 5; };

template <typename Px, bool SURF, bool BEST>
__global__ void __launch_bounds__(1024) me_ctu_kernel(MEArgs a)
{
    constexpr int BPP  = PxInfo<Px>::BPP;
    constexpr int DWPR = MECfg<Px>::DWPR;
    constexpr int NLD  = MECfg<Px>::NLD;
    extern __shared__ __attribute__((aligned(16))) uint8_t win[];

    const int R = a.range;
    const int NC = 2 * R + 1;                 // mv columns == mv rows
    const int rows = 64 + 2 * R;
    const int ctu = blockIdx.x;
    const int cx = (ctu % a.ctusW) * 64, cy = (ctu / a.ctusW) * 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;

    // ---- stage the search window: aligned dword copy of each row -------------------------------
    const uint8_t* g0 = a.fref + (long)(cy - R) * a.frefStrideB + (long)(cx - R) * BPP;
    const int adj = (int)((uintptr_t)g0 & 3);           // identical for every row (stride % 4 == 0)
    const uint8_t* g0a = g0 - adj;
    const int rowDw = a.rowBytes >> 2;
    for (int r = wave; r < rows; r += nwaves)
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(g0a + (long)r * a.frefStrideB);
        uint32_t* dst = reinterpret_cast<uint32_t*>(win + r * a.rowBytes);
        for (int c = lane; c < rowDw; c += 64)
            dst[c] = src[c];
    }

    // ---- this lane's 8x8 source block, kept in registers ---------------------------------------
    int bx, by;
    zorder_xy(lane, bx, by);
    uint32_t F[8][DWPR];
    {
        const uint8_t* fe = a.fenc + (long)(cy + by * 8) * a.fencStrideB + (long)(cx + bx * 8) * BPP;
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int k = 0; k < DWPR; k++)
                F[j][k] = ld_u32(fe + (long)j * a.fencStrideB + 4 * k);
    }
    __syncthreads();

    unsigned long long bk8 = ~0ull, bk16 = ~0ull, bk32 = ~0ull, bk64 = ~0ull;

    const int T = 2 * R + 8;                  // window rows a lane walks through
    for (int mvxi = wave; mvxi < NC; mvxi += nwaves)
    {
        const int xb = adj + (bx * 8 + mvxi) * BPP;                 // byte column inside the LDS row
        const int sh = __builtin_amdgcn_readfirstlane((xb & 3) * 8); // wave-uniform (bx*8*BPP % 4 == 0)
        const uint8_t* rp = win + (by * 8) * a.rowBytes + (xb & ~3);
        const uint32_t cxv = BEST ? a.costX[mvxi] : 0;

        uint32_t acc[8];
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = 0;

        // 8 window rows per call; FIRST = the warm-up rows where displacement index m = t - j < 0
        auto rows8 = [&](auto firstTag, const int t0)
        {
            constexpr bool FIRST = decltype(firstTag)::value;
#pragma unroll
            for (int p = 0; p < 8; p++)
            {
                const int t = t0 + p;
                if (FIRST || t < T)     // wave-uniform
                {
                    const uint32_t* lp = reinterpret_cast<const uint32_t*>(rp + t * a.rowBytes);
                    uint32_t d[NLD];
#pragma unroll
                    for (int k = 0; k < NLD; k++) d[k] = lp[k];
                    uint32_t rr[DWPR];
#pragma unroll
                    for (int k = 0; k < DWPR; k++) rr[k] = __builtin_amdgcn_alignbit(d[k + 1], d[k], sh);

                    // window row t meets source row j at vertical displacement index m = t - j
#pragma unroll
                    for (int j = 0; j < 8; j++)
                    {
                        if (FIRST && j > p) continue;
                        uint32_t v = acc[(p - j) & 7];
#pragma unroll
                        for (int k = 0; k < DWPR; k++) v = sad_dw<Px>(F[j][k], rr[k], v);
                        acc[(p - j) & 7] = v;
                    }

                    if (!FIRST || p == 7)
                    {
                        const int m = t - 7;                      // completed displacement row
                        const int slot = (p + 1) & 7;
                        const int s8 = (int)acc[slot];
                        acc[slot] = 0;
                        const int s16 = quad_sum(s8);
                        const int s32 = row_sum_of_quads(s16);
                        const int s64 = wave_sum_of_rows(s32);
                        const size_t o = ((size_t)ctu * NC + m) * NC + mvxi;
                        if (SURF)
                        {
                            if (a.surf[0]) a.surf[0][o * 64 + lane] = s8;
                            if (a.surf[1] && (lane & 3) == 0) a.surf[1][o * 16 + (lane >> 2)] = s16;
                            if (a.surf[2] && (lane & 15) == 0) a.surf[2][o * 4 + (lane >> 4)] = s32;
                            if (a.surf[3] && lane == 0) a.surf[3][o] = s64;
                        }
                        if (BEST)
                        {
                            const uint32_t mvc = cxv + a.costY[m];
                            const uint32_t idx = (uint32_t)(m * NC + mvxi);
                            const unsigned long long k8 = ((unsigned long long)((uint32_t)s8 + mvc) << 32) | idx;
                            const unsigned long long k16 = ((unsigned long long)((uint32_t)s16 + mvc) << 32) | idx;
                            const unsigned long long k32 = ((unsigned long long)((uint32_t)s32 + mvc) << 32) | idx;
                            const unsigned long long k64 = ((unsigned long long)((uint32_t)s64 + mvc) << 32) | idx;
                            bk8 = k8 < bk8 ? k8 : bk8;
                            bk16 = k16 < bk16 ? k16 : bk16;
                            bk32 = k32 < bk32 ? k32 : bk32;
                            bk64 = k64 < bk64 ? k64 : bk64;
                        }
                    }
                }
            }
        };
        rows8(std::true_type{}, 0);                      // T >= 10 always, the first 8 rows exist
        for (int t0 = 8; t0 < T; t0 += 8)
            rows8(std::false_type{}, t0);
    }

    if (BEST)
    {
        if (a.best[0]) atomicMin(&a.best[0][(size_t)ctu * 64 + lane], bk8);
        if (a.best[1] && (lane & 3) == 0) atomicMin(&a.best[1][(size_t)ctu * 16 + (lane >> 2)], bk16);
        if (a.best[2] && (lane & 15) == 0) atomicMin(&a.best[2][(size_t)ctu * 4 + (lane >> 4)], bk32);
        if (a.best[3] && lane == 0) atomicMin(&a.best[3][ctu], bk64);
    }
}

// number of wavefronts per workgroup: prefer an exact divisor of the column count (no idle tail)
static int pick_waves(int ncols)
{
    int bestW = 16, bestWaste = 1 << 30;
    for (int w = 16; w >= 4; w--)
    {
        int waste = ((ncols + w - 1) / w) * w - ncols;
        if (waste * bestW < bestWaste * w) { bestWaste = waste; bestW = w; }   // compare waste fraction
        if (waste == 0) { bestW = w; break; }
    }
    return bestW;
}

template <typename Px>
static int launch_me(const x265hip_me_params* p, hipStream_t s)
{
    constexpr int BPP = PxInfo<Px>::BPP;
    MEArgs a;
    a.fenc = (const uint8_t*)p->fenc;  a.fencStrideB = (long)p->fenc_stride * BPP;
    a.fref = (const uint8_t*)p->fref;  a.frefStrideB = (long)p->fref_stride * BPP;
    a.ctusW = p->width / 64;
    a.range = p->range;
    int rowBytes = (3 + (56 + 2 * p->range) * BPP + 4 * MECfg<Px>::NLD + 3) & ~3;
    if (((rowBytes >> 2) & 1) == 0) rowBytes += 4;            // odd dword pitch spreads rows over LDS banks
    a.rowBytes = rowBytes;
    bool anySurf = false, anyBest = false;
    for (int l = 0; l < 4; l++)
    {
        a.surf[l] = p->surf[l]; a.best[l] = (unsigned long long*)p->best[l];
        anySurf |= p->surf[l] != nullptr; anyBest |= p->best[l] != nullptr;
    }
    a.costX = p->cost_x; a.costY = p->cost_y;
    const int nctu = a.ctusW * (p->height / 64);
    const size_t lds = (size_t)rowBytes * (64 + 2 * p->range);
    if (lds > 160 * 1024) { set_error("me_fullsearch: range %d needs %zu B of LDS (> 160 KiB)", p->range, lds); return X265HIP_EINVAL; }
    const int nw = pick_waves(2 * p->range + 1);
    dim3 grid(nctu), block(nw * 64);
#define LAUNCH(SF, BS) do { \
        if (lds > 64 * 1024) X265HIP_TRY(hipFuncSetAttribute((const void*)me_ctu_kernel<Px, SF, BS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((me_ctu_kernel<Px, SF, BS>), grid, block, lds, s, a); } while (0)
    if (anySurf && anyBest) LAUNCH(true, true);
    else if (anySurf) LAUNCH(true, false);
    else LAUNCH(false, true);
#undef LAUNCH
    X265HIP_TRY(hipGetLastError());
    return 0;
}

__global__ void fill_u64_kernel(unsigned long long* p, size_t n, unsigned long long v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) p[i] = v;
}

} // namespace x265hip

using namespace x265hip;

extern "C" int x265hip_me_fullsearch(const x265hip_me_params* p, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !p->fenc || !p->fref) { set_error("me_fullsearch: NULL plane"); return X265HIP_EINVAL; }
    if (p->width <= 0 || p->height <= 0 || (p->width & 63) || (p->height & 63))
    { set_error("me_fullsearch: width/height must be positive multiples of 64 (got %dx%d)", p->width, p->height); return X265HIP_EINVAL; }
    if (p->range < 1 || p->range > 256) { set_error("me_fullsearch: range %d out of [1,256]", p->range); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("me_fullsearch: depth %d", p->depth); return X265HIP_EINVAL; }
    const int bpp = p->depth == 8 ? 1 : 2;
    if (((p->fenc_stride * bpp) & 3) || ((p->fref_stride * bpp) & 3) || ((uintptr_t)p->fenc & 3))
    { set_error("me_fullsearch: plane strides must be multiples of 4 bytes and fenc 4-byte aligned"); return X265HIP_EINVAL; }
    bool anyOut = false, anyBest = false;
    for (int l = 0; l < 4; l++) { anyOut |= p->surf[l] || p->best[l]; anyBest |= p->best[l] != nullptr; }
    if (!anyOut) { set_error("me_fullsearch: no output requested"); return X265HIP_EINVAL; }
    if (anyBest && (!p->cost_x || !p->cost_y)) { set_error("me_fullsearch: best[] needs cost_x / cost_y"); return X265HIP_EINVAL; }
    if (p->depth == 8) return launch_me<uint8_t>(p, (hipStream_t)stream);
    return launch_me<uint16_t>(p, (hipStream_t)stream);
}

extern "C" int x265hip_me_best_reset(uint64_t* best, size_t count, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!best) { set_error("me_best_reset: NULL"); return X265HIP_EINVAL; }
    if (!count) return 0;
    int blocks = (int)((count + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fill_u64_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (unsigned long long*)best, count, ~0ull);
    X265HIP_TRY(hipGetLastError());
    return 0;
}
