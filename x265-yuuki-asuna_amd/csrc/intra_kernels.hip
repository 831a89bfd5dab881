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
#define GO(PX) do { switch (kind) { \
        case X265HIP_INTRA_PRED:    hipLaunchKernelGGL((intra_kernel<PX, X265HIP_INTRA_PRED>), dim3(njobs), dim3(threads), 0, s, a); break; \
        case X265HIP_INTRA_FILTER:  hipLaunchKernelGGL((intra_kernel<PX, X265HIP_INTRA_FILTER>), dim3(njobs), dim3(threads), 0, s, a); break; \
        case X265HIP_INTRA_ALLANGS: hipLaunchKernelGGL((intra_kernel<PX, X265HIP_INTRA_ALLANGS>), dim3(njobs), dim3(threads), 0, s, a); break; \
        default: set_error("intra_batch: unknown kind %d", kind); return X265HIP_EINVAL; } } while (0)
    if (depth == 8) GO(uint8_t); else GO(uint16_t);
#undef GO
    X265HIP_TRY(hipGetLastError());
    return 0;
}
