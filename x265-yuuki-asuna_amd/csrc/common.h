// common.h - shared device/host helpers for the gfx950 (CDNA4) block-primitive kernels.
// gfx950 only: 64-lane wavefronts are assumed everywhere.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <mutex>

#include "x265hip.h"

#define WAVE 64

namespace x265hip {

// ---------------------------------------------------------------- pixel traits
template <int DEPTH> struct PixT        { typedef uint16_t type; };
template <>          struct PixT<8>     { typedef uint8_t  type; };

template <typename Px> struct PxInfo;
template <> struct PxInfo<uint8_t>  { static constexpr int BPP = 1; static constexpr int PER_DW = 4; };
template <> struct PxInfo<uint16_t> { static constexpr int BPP = 2; static constexpr int PER_DW = 2; };

// ---------------------------------------------------------------- host-side error plumbing
void set_error(const char* fmt, ...);
int  check_hip(hipError_t e, const char* what);   // 0 or X265HIP_ENODEV with last-error text
int  ensure_device();                             // lazy x265hip_init(-1): validates the calling thread's current device
void apply_wait_policy(int device);               // runtime.hip: blocking host waits on `device` (once per device; see x265hip_set_wait_policy)
void* stream_scratch(hipStream_t s, int slot, size_t bytes);   // runtime.hip: grow-only scratch per (device, stream, slot), NULL on failure
std::unique_lock<std::mutex> stream_sequence_lock(hipStream_t s);   // runtime.hip: hold it while enqueuing a clear-then-launch sequence that uses stream_scratch

// csrc/phase_kernels.hip: the launch behind x265hip_phase_planes with the distance between phase planes as a parameter (bands)
int  phase_planes_launch(int depth, int chroma, const void* src, void* dst, intptr_t stride, int rows, size_t plane_bytes, hipStream_t s);

#define X265HIP_TRY(expr) do { int _rc = ::x265hip::check_hip((expr), #expr); if (_rc) return _rc; } while (0)

#ifdef __HIPCC__
// ---------------------------------------------------------------- device helpers
// Unaligned loads: the reference passes arbitrary pixel addresses (motion vectors move the
// pointer byte by byte).  gfx950 global memory handles unaligned dword accesses in hardware;
// the packed/aligned(1) types make that legal for the compiler.
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
typedef uint32_t __attribute__((aligned(2))) u32_align2;

__device__ __forceinline__ uint32_t ld_u32(const void* p) { return *reinterpret_cast<const u32_unaligned*>(p); }
__device__ __forceinline__ uint64_t ld_u64(const void* p) { return *reinterpret_cast<const u64_unaligned*>(p); }

// 4 packed u8 |a-b| sum + acc   (v_sad_u8)
__device__ __forceinline__ uint32_t sad4_u8(uint32_t a, uint32_t b, uint32_t acc) { return __builtin_amdgcn_sad_u8(a, b, acc); }
// 2 packed u16 |a-b| sum + acc  (v_sad_u16)
__device__ __forceinline__ uint32_t sad2_u16(uint32_t a, uint32_t b, uint32_t acc) { return __builtin_amdgcn_sad_u16(a, b, acc); }

template <typename Px> __device__ __forceinline__ uint32_t sad_dw(uint32_t a, uint32_t b, uint32_t acc);
template <> __device__ __forceinline__ uint32_t sad_dw<uint8_t>(uint32_t a, uint32_t b, uint32_t acc) { return sad4_u8(a, b, acc); }
template <> __device__ __forceinline__ uint32_t sad_dw<uint16_t>(uint32_t a, uint32_t b, uint32_t acc) { return sad2_u16(a, b, acc); }

// DPP cross-lane moves (no LDS traffic).  ctrl encodings: quad_perm = p0|p1<<2|p2<<4|p3<<6,
// row_shr:n = 0x110+n, row_ror:n = 0x120+n.
template <int CTRL> __device__ __forceinline__ int dpp(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
// sum over the 4 lanes of each quad, result in all 4 lanes
__device__ __forceinline__ int quad_sum(int v)
{
    v += dpp<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp<0x4E>(v);   // quad_perm [2,3,0,1]
    return v;
}
// given quad-uniform values: sum over the 4 quads of each 16-lane row, result in all 16 lanes
__device__ __forceinline__ int row_sum_of_quads(int v)
{
    v += dpp<0x124>(v);  // row_ror:4
    v += dpp<0x128>(v);  // row_ror:8
    return v;
}
// full 16-lane row sum of arbitrary values
__device__ __forceinline__ int row_sum(int v)
{
    v += dpp<0x121>(v);  // row_ror:1
    v += dpp<0x122>(v);  // row_ror:2
    return row_sum_of_quads(v);
}
// sum of one value per 16-lane row across the 4 rows of the wave (input row-uniform) -> uniform
__device__ __forceinline__ int wave_sum_of_rows(int v)
{
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)
         + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
// xor-butterfly reduction over groups of G lanes (G power of two <= 64), result in every lane
template <int G, typename T> __device__ __forceinline__ T group_sum(T v)
{
#pragma unroll
    for (int m = 1; m < G; m <<= 1)
        v += __shfl_xor(v, m, 64);
    return v;
}

// Workgroups are dealt to the 8 XCDs round-robin (workgroup i runs on XCD i % 8) and every XCD has its own L2: hand each XCD a
// contiguous eighth of the index range, so that neighbouring CTUs (shared reference rows and search-window overlap) meet in one L2.
__device__ __forceinline__ int xcd_swizzle(int i, int n)
{
    const int per = n >> 3;
    return i < (per << 3) ? (i & 7) * per + (i >> 3) : i;
}

// the same for a 2-D grid whose rows share input with the rows above / below them (halo rows): every XCD gets a contiguous band of grid rows
__device__ __forceinline__ void xcd_swizzle_2d(int& bx, int& by)
{
    const int t = xcd_swizzle((int)(blockIdx.x + blockIdx.y * gridDim.x), (int)(gridDim.x * gridDim.y));
    bx = t % (int)gridDim.x; by = t / (int)gridDim.x;
}

__device__ __forceinline__ int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
#endif  // __HIPCC__

} // namespace x265hip
