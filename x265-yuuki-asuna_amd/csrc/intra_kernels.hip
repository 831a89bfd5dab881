// intra_kernels.hip - batched HEVC intra prediction on gfx950.
//
// Reference semantics (source/common/intrapred.cpp): intraFilter :31-51, intra_pred_dc_c + dcPredFilter
// :53-85, planar_pred_c :87-100, intra_pred_ang_c :102-204 (modes 2..17 are predicted from the swapped
// neighbour arms and transposed), all_angs_pred_c :206-234 (modes 2..34 packed at dest + (mode-2)*N*N,
// horizontal modes left transposed, filtered neighbours per g_intraFilterFlags, constants.cpp:561).
// Neighbour buffer: [0] top-left, [1..2N] above + above-right, [2N+1..4N] left + below-left.
//
// Mapping: one workgroup per (TU, mode) candidate; the 4N+1 neighbours (and, for negative angles, the
// projected reference line) live in LDS, one thread per predicted sample.
#include "common.h"

#include <cstdlib>

namespace x265hip {

__constant__ int8_t kAngle[17] = { -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
__constant__ int16_t kInvAngle[8] = { 4096, 1638, 910, 630, 482, 390, 315, 256 };
__constant__ uint8_t kIntraFilterFlags[35] = {
    0x38, 0x00,
    0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30,
    0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30,
    0x38 };

struct IntraArgs
{
    const uint8_t* src; long srcStride;
    uint8_t* dst; long dstStride;
    const x265hip_job* jobs;
    int n, log2n, depth;
};

// Predict one N x N block into dst (row stride ds).  nb = neighbour samples in LDS (int), line = LDS
// scratch of >= 3N+2 ints.  keepTransposed: leave horizontal modes un-flipped (all-angs packing).
template <typename Px>
__device__ void predict_block(Px* dst, long ds, const int* nb0, int* swapped, int* line, int mode, int bFilter,
                              int n, int log2n, int depth, bool keepTransposed)
{
    const int tid = threadIdx.x, nth = blockDim.x;
    const int n2 = 2 * n;
    const int maxVal = (1 << depth) - 1;
    if (mode == 0)          // planar
    {
        const int tr = nb0[1 + n], bl = nb0[n2 + 1 + n];
        for (int i = tid; i < n * n; i += nth)
        {
            const int y = i >> log2n, x = i & (n - 1);
            dst[y * ds + x] = (Px)(((n - 1 - x) * nb0[n2 + 1 + y] + (n - 1 - y) * nb0[1 + x] + (x + 1) * tr + (y + 1) * bl + n) >> (log2n + 1));
        }
        return;
    }
    if (mode == 1)          // DC
    {
        int sum = n;
        for (int i = 0; i < n; i++) sum += nb0[1 + i] + nb0[n2 + 1 + i];      // every thread: 2N LDS broadcasts
        const int dc = sum / n2;
        for (int i = tid; i < n * n; i += nth)
        {
            const int y = i >> log2n, x = i & (n - 1);
            int v = dc;
            if (bFilter)
            {
                if (x == 0 && y == 0) v = (nb0[1] + nb0[n2 + 1] + 2 * dc + 2) >> 2;
                else if (y == 0) v = (nb0[1 + x] + 3 * dc + 2) >> 2;
                else if (x == 0) v = (nb0[n2 + 1 + y] + 3 * dc + 2) >> 2;
            }
            dst[y * ds + x] = (Px)v;
        }
        return;
    }
    // angular
    const bool hor = mode < 18;
    const int* nb = nb0;
    if (hor)
    {
        for (int i = tid; i < n2; i += nth)
        {
            swapped[1 + i] = nb0[n2 + 1 + i];
            swapped[n2 + 1 + i] = nb0[1 + i];
        }
        if (tid == 0) swapped[0] = nb0[0];
        __syncthreads();
        nb = swapped;
    }
    const int aoff = hor ? 10 - mode : mode - 26;
    const int angle = kAngle[8 + aoff];
    // a sample predicted at (x, y) of the un-flipped block lands at (y, x) for horizontal modes
    const bool flip = hor && !keepTransposed;
    if (angle == 0)
    {
        const int tl = nb[0], top = nb[1];
        for (int i = tid; i < n * n; i += nth)
        {
            const int y = i >> log2n, x = i & (n - 1);
            int v = nb[1 + x];
            if (bFilter && x == 0)
            {
                const int16_t t = (int16_t)(top + ((nb[n2 + 1 + y] - tl) >> 1));
                v = t < 0 ? 0 : (t > maxVal ? maxVal : t);
            }
            if (flip) dst[x * ds + y] = (Px)v; else dst[y * ds + x] = (Px)v;
        }
        return;
    }
    const int* ref;
    if (angle < 0)
    {
        const int nproj = -((n * angle) >> 5) - 1;
        int* base = line + nproj + 1;                        // base[-1] = top-left, base[0..n-1] = main arm
        const int inv = kInvAngle[-aoff - 1];
        for (int i = tid; i < nproj; i += nth)
            base[-2 - i] = nb[n2 + ((128 + (i + 1) * inv) >> 8)];
        for (int i = tid; i <= n; i += nth)
            base[-1 + i] = nb[i];
        __syncthreads();
        ref = base;
    }
    else
        ref = nb + 1;
    for (int i = tid; i < n * n; i += nth)
    {
        const int y = i >> log2n, x = i & (n - 1);
        const int pos = (y + 1) * angle;
        const int off = pos >> 5, frac = pos & 31;
        const int v = frac ? ((32 - frac) * ref[off + x] + frac * ref[off + x + 1] + 16) >> 5 : ref[off + x];
        if (flip) dst[x * ds + y] = (Px)v; else dst[y * ds + x] = (Px)v;
    }
}

template <typename Px, int KIND>
__global__ void __launch_bounds__(256) intra_kernel(IntraArgs a)
{
    __shared__ int nbA[4 * 32 + 1 + 3], nbB[4 * 32 + 1 + 3], swapped[4 * 32 + 1 + 3], line[3 * 32 + 8];
    const x265hip_job jb = a.jobs[blockIdx.x];
    const int tid = threadIdx.x, nth = blockDim.x;
    const int n = a.n, n2 = 2 * n, cnt = 4 * n + 1;
    const Px* s = reinterpret_cast<const Px*>(a.src) + jb.off[0];
    Px* d = reinterpret_cast<Px*>(a.dst) + jb.off[1];
    for (int i = tid; i < cnt; i += nth) nbA[i] = s[i];
    if (KIND == X265HIP_INTRA_ALLANGS)
    {
        const Px* f = reinterpret_cast<const Px*>(a.src) + jb.off[2];
        for (int i = tid; i < cnt; i += nth) nbB[i] = f[i];
    }
    __syncthreads();
    if (KIND == X265HIP_INTRA_FILTER)
    {
        for (int i = tid; i < cnt; i += nth)
        {
            int v;
            if (i == 0) v = (2 * nbA[0] + nbA[1] + nbA[n2 + 1] + 2) >> 2;
            else if (i == n2 || i == 2 * n2) v = nbA[i];
            else if (i == n2 + 1) v = (2 * nbA[n2 + 1] + nbA[0] + nbA[n2 + 2] + 2) >> 2;
            else v = (2 * nbA[i] + nbA[i - 1] + nbA[i + 1] + 2) >> 2;
            d[i] = (Px)v;
        }
        return;
    }
    if (KIND == X265HIP_INTRA_PRED)
    {
        predict_block<Px>(d, a.dstStride, nbA, swapped, line, jb.arg[0], jb.arg[1], n, a.log2n, a.depth, false);
        return;
    }
    for (int mode = 2; mode <= 34; mode++)
    {
        const int* nb = (kIntraFilterFlags[mode] & n) ? nbB : nbA;
        predict_block<Px>(d + (long)(mode - 2) * n * n, n, nb, swapped, line, mode, jb.arg[0], n, a.log2n, a.depth, true);
        __syncthreads();           // swapped / line are reused by the next mode
    }
}

// ------------------------------------------------------------------------------------------------
// Fast path for PRED and ALLANGS: a thread owns 4 horizontally adjacent output samples (one dword / two dword
// store), a 256-thread workgroup carries 256 / (N*N/4) candidates (64 for 4x4 ... 1 for 32x32).  Per candidate the
// neighbours are staged in LDS and the angular reference line (main arm, corner and - for negative angles - the
// samples projected from the side arm, intrapred.cpp:150-160) is laid out once, so a predicted sample is two LDS
// reads and one interpolation.  Horizontal modes read the swapped arms and write transposed, as index arithmetic.
struct IntraMode
{
    int hor, angle, inv, mainBase, sideBase;
};
__device__ __forceinline__ IntraMode intra_mode(int mode, int n2)
{
    IntraMode m;
    m.hor = mode < 18;
    const int aoff = m.hor ? 10 - mode : mode - 26;
    m.angle = mode >= 2 ? kAngle[8 + aoff] : 0;
    m.inv = (mode >= 2 && m.angle < 0) ? kInvAngle[-aoff - 1] : 0;
    m.mainBase = m.hor ? n2 : 0;
    m.sideBase = m.hor ? 0 : n2;
    return m;
}

template <typename Px, int KIND>
__global__ void __launch_bounds__(256) intra_quad_kernel(IntraArgs a, int njobs, int log2tpj, int quadsPerThread)
{
    constexpr bool ALL = KIND == X265HIP_INTRA_ALLANGS;
    __shared__ Px nbs[64 * 20];                 // jobsPerWg * (4n + 4) samples: 64 * 20 (n = 4) ... 1 * 132 (n = 32)
    __shared__ Px nbf[ALL ? 64 * 20 : 1];       // the filtered neighbours of ALLANGS
    __shared__ Px lines[64 * 16];               // jobsPerWg * (3n + 4): 64 * 16 ... 1 * 100
    __shared__ int dcSum[64];
    const int tid = threadIdx.x;
    const int n = a.n, log2n = a.log2n, n2 = 2 * n, cnt = 4 * n + 1, pitch = 4 * n + 4, lpitch = 3 * n + 4;
    const int tpj = 1 << log2tpj, jpw = 256 >> log2tpj;
    const int jw = tid >> log2tpj, q = tid & (tpj - 1);
    const int job = blockIdx.x * jpw + jw;
    const bool live = job < njobs;
    x265hip_job jb;
    if (live) jb = a.jobs[job];
    if (tid < 64) dcSum[tid] = 0;
    __syncthreads();
    Px* nb0 = nbs + jw * pitch;
    Px* nbF = ALL ? nbf + jw * pitch : nb0;
    Px* line = lines + jw * lpitch + n;          // line[k], k in [-n, 2n]
    if (live)
    {
        const Px* sp = reinterpret_cast<const Px*>(a.src) + jb.off[0];
        int part = 0;
        for (int k = q; k < cnt; k += tpj)
        {
            const Px v = sp[k];
            nb0[k] = v;
            if (ALL) nbF[k] = (reinterpret_cast<const Px*>(a.src) + jb.off[2])[k];
            else if ((k >= 1 && k <= n) || (k >= n2 + 1 && k <= n2 + n)) part += (int)v;
        }
        if (!ALL && jb.arg[0] == 1) atomicAdd(&dcSum[jw], part);
    }
    __syncthreads();
    const int qpr = n >> 2;
    // a thread predicts `quadsPerThread` row-quads (4 samples each): quad index q, q + tpj, ... (fewer, fatter workgroups for the big blocks)
    int y = q >> (log2n - 2), x0 = (q & (qpr - 1)) * 4;
    auto set_quad = [&](int r) { const int qq = q + r * tpj; y = qq >> (log2n - 2); x0 = (qq & (qpr - 1)) * 4; };
    const int maxVal = (1 << a.depth) - 1;
    Px* d = live ? reinterpret_cast<Px*>(a.dst) + jb.off[1] : nullptr;
    auto put4 = [&](Px* p, const int (&v)[4])
    {
        if (sizeof(Px) == 1)
            *reinterpret_cast<u32_unaligned*>(p) = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
        else
        {
            reinterpret_cast<u32_unaligned*>(p)[0] = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
            reinterpret_cast<u32_unaligned*>(p)[1] = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
        }
    };
    // lay out the reference line of `mode` from neighbour set nb (all threads of the candidate), then predict 4 samples
    auto build_line = [&](const Px* nb, const IntraMode& m)
    {
        const int lowest = (n * m.angle) >> 5;                    // most negative index read (0 for angle >= 0)
        for (int e = q; e < 3 * n + 2; e += tpj)
        {
            const int k = e - n;
            Px v = 0;
            if (k >= 0) v = k < n2 ? nb[m.mainBase + 1 + k] : nb[m.mainBase + n2];      // line[2n] is only ever weighted by 0
            else if (k == -1) v = nb[0];
            else if (k >= lowest) v = nb[m.sideBase + ((128 + (-1 - k) * m.inv) >> 8)];
            line[k] = v;
        }
    };
    auto angular4 = [&](const Px* nb, const IntraMode& m, const int bFilter, const bool flip, int (&v)[4])
    {
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const int r = flip ? x0 + k : y, c = flip ? y : x0 + k;
            if (m.angle == 0)
            {
                int t = nb[m.mainBase + 1 + c];
                if (bFilter && c == 0)
                {
                    const int16_t f = (int16_t)(nb[m.mainBase + 1] + (((int)nb[m.sideBase + 1 + r] - (int)nb[0]) >> 1));
                    t = f < 0 ? 0 : (f > maxVal ? maxVal : f);
                }
                v[k] = t;
            }
            else
            {
                const int pos = (r + 1) * m.angle, off = pos >> 5, frac = pos & 31;
                v[k] = ((32 - frac) * (int)line[off + c] + frac * (int)line[off + c + 1] + 16) >> 5;
            }
        }
    };
    if (!ALL)
    {
        const int mode = live ? jb.arg[0] : 0, bFilter = live ? jb.arg[1] : 0;
        const IntraMode m = intra_mode(mode, n2);
        if (live && mode >= 2 && m.angle != 0) build_line(nb0, m);
        __syncthreads();
        if (!live) return;
        for (int rq = 0; rq < quadsPerThread; rq++)
        {
        set_quad(rq);
        int v[4];
        if (mode == 0)
        {
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const int x = x0 + k;
                v[k] = ((n - 1 - x) * nb0[n2 + 1 + y] + (n - 1 - y) * nb0[1 + x] + (x + 1) * nb0[1 + n] + (y + 1) * nb0[n2 + 1 + n] + n) >> (log2n + 1);
            }
        }
        else if (mode == 1)
        {
            const int dc = (dcSum[jw] + n) / n2;
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const int x = x0 + k;
                int t = dc;
                if (bFilter)
                {
                    if (x == 0 && y == 0) t = (nb0[1] + nb0[n2 + 1] + 2 * dc + 2) >> 2;
                    else if (y == 0) t = (nb0[1 + x] + 3 * dc + 2) >> 2;
                    else if (x == 0) t = (nb0[n2 + 1 + y] + 3 * dc + 2) >> 2;
                }
                v[k] = t;
            }
        }
        else
            angular4(nb0, m, bFilter, m.hor != 0, v);
        put4(d + (long)y * a.dstStride + x0, v);
        }
        return;
    }
    for (int mode = 2; mode <= 34; mode++)
    {
        const Px* nb = (kIntraFilterFlags[mode] & n) ? nbF : nb0;
        const IntraMode m = intra_mode(mode, n2);
        if (live && m.angle != 0) build_line(nb, m);
        __syncthreads();
        if (live)
            for (int rq = 0; rq < quadsPerThread; rq++)
            {
                set_quad(rq);
                int v[4];
                angular4(nb, m, jb.arg[0], false, v);              // all-angs keeps the horizontal modes transposed
                put4(d + (long)(mode - 2) * n * n + y * n + x0, v);
            }
        __syncthreads();                                            // the line is rebuilt for the next mode
    }
}

} // namespace x265hip

using namespace x265hip;

extern "C" int x265hip_intra_batch(int kind, int depth, int n, x265hip_plane src, x265hip_plane dst,
                                   const x265hip_job* jobs, int njobs, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!src.base || !dst.base || !jobs || njobs < 0) { set_error("intra_batch: NULL operand"); return X265HIP_EINVAL; }
    if (njobs == 0) return 0;
    if (n != 4 && n != 8 && n != 16 && n != 32) { set_error("intra_batch: TU size %d", n); return X265HIP_EINVAL; }
    if (depth != 8 && depth != 10 && depth != 12) { set_error("intra_batch: depth %d", depth); return X265HIP_EINVAL; }
    IntraArgs a;
    a.src = (const uint8_t*)src.base; a.srcStride = src.stride; a.dst = (uint8_t*)dst.base; a.dstStride = dst.stride;
    a.jobs = jobs; a.n = n; a.log2n = n == 4 ? 2 : (n == 8 ? 3 : (n == 16 ? 4 : 5)); a.depth = depth;
    const int threads = n * n <= 64 ? 64 : 256;
    hipStream_t s = (hipStream_t)stream;
    // fast path geometry: a thread per row-quad for 4x4 / 8x8, four row-quads per thread for 16x16 / 32x32 (the LDS arrays hold the
    // neighbours of up to 64 / 16 / 16 / 4 candidates)
    const int qpt = n >= 16 ? 4 : 1;
    const int log2tpj = 2 * a.log2n - 2 - (qpt == 4 ? 2 : 0), jpw = 256 >> log2tpj, wgs = (njobs + jpw - 1) / jpw;
    static const bool generic = getenv("X265HIP_INTRA_GENERIC") != nullptr;          // A/B switch (read once): one workgroup per candidate
#define GO(PX) do { switch (kind) { \
        case X265HIP_INTRA_PRED:    if (generic) hipLaunchKernelGGL((intra_kernel<PX, X265HIP_INTRA_PRED>), dim3(njobs), dim3(threads), 0, s, a); \
                                    else hipLaunchKernelGGL((intra_quad_kernel<PX, X265HIP_INTRA_PRED>), dim3(wgs), dim3(256), 0, s, a, njobs, log2tpj, qpt); break; \
        case X265HIP_INTRA_FILTER:  hipLaunchKernelGGL((intra_kernel<PX, X265HIP_INTRA_FILTER>), dim3(njobs), dim3(threads), 0, s, a); break; \
        case X265HIP_INTRA_ALLANGS: if (generic) hipLaunchKernelGGL((intra_kernel<PX, X265HIP_INTRA_ALLANGS>), dim3(njobs), dim3(threads), 0, s, a); \
                                    else hipLaunchKernelGGL((intra_quad_kernel<PX, X265HIP_INTRA_ALLANGS>), dim3(wgs), dim3(256), 0, s, a, njobs, log2tpj, qpt); break; \
        default: set_error("intra_batch: unknown kind %d", kind); return X265HIP_EINVAL; } } while (0)
    if (depth == 8) GO(uint8_t); else GO(uint16_t);
#undef GO
    X265HIP_TRY(hipGetLastError());
    return 0;
}
