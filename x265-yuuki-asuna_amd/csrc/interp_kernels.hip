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

template <typename Px, int N>
static int launch_ip(int kind, const IPArgs& a, int njobs, hipStream_t s)
{
    const int threads = a.w * a.h <= 256 ? 64 : 256;
#define CASE(K) case K: hipLaunchKernelGGL((interp_kernel<Px, K, N>), dim3(njobs), dim3(threads), 0, s, a); break;
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
