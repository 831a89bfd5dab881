// Micro-benchmark: what does rocprofv3's FETCH_SIZE report on gfx950 for a KNOWN number of bytes, per access pattern?  (round-5 verdict, weak 5 / next 4c:
// the guide calibrates the x2 correction only for wide coalesced streaming reads; the tail kernels of the step read narrow rows, small tiles at arbitrary
// positions and overlapping halos.)  Every pattern is its own kernel over a 1 GiB buffer (4 x the Infinity Cache, first touch per launch), and the program
// prints, per kernel, the bytes the lanes REQUEST and the bytes of the distinct 64-byte and 128-byte lines they touch; run it under
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o f -- tools/ubench/fetch_calibration
// and tools/fetch_calibration.py divides.  Build: hipcc --offload-arch=gfx950 -O3 fetch_calibration.hip -o fetch_calibration
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)

// (a) the guide's reference point: 16 bytes per lane, fully coalesced, streaming
__global__ void cal_stream_b128(const v4u* p, size_t n, unsigned* sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i < n; i += stride) { const v4u v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
// (b) 4 bytes per lane, coalesced (a wavefront = 256 contiguous bytes)
__global__ void cal_stream_b32(const unsigned* p, size_t n, unsigned* sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i < n; i += stride) acc += p[i];
    if (acc == 0x12345678u) *sink = acc;
}
// (c) 1 byte per lane, coalesced (a wavefront = 64 contiguous bytes: the plane copies' 8-bit rows)
__global__ void cal_stream_b8(const uint8_t* p, size_t n, unsigned* sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i < n; i += stride) acc += p[i];
    if (acc == 0x12345678u) *sink = acc;
}
// (d) rows of a 32 x 32 byte tile per wavefront: lane l reads 4 bytes of row (l / 8) + 8 k, column 4 (l % 8) - the reconstruction / SAO tile walk: 32 contiguous
// bytes per row, rows `pitch` apart, tiles side by side so that every byte of the plane is read exactly once
__global__ void cal_tile_rows32(const uint8_t* p, int pitch, int tilesX, int tilesY, unsigned* sink)
{
    const int lane = threadIdx.x & 63, wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= tilesX * tilesY) return;
    const int tx = wave % tilesX, ty = wave / tilesX;
    const uint8_t* t = p + (size_t)ty * 32 * pitch + tx * 32;
    unsigned acc = 0;
    for (int k = 0; k < 4; k++) acc += *(const unsigned*)(t + (size_t)((lane >> 3) + 8 * k) * pitch + 4 * (lane & 7));
    if (acc == 0x12345678u) *sink = acc;
}
// (e) 4 x 4 byte tiles at ARBITRARY positions (the sub-sample refinement's candidates): a lane reads the 4 rows of one tile, 4 unaligned bytes each
__global__ void cal_scatter_4x4(const uint8_t* p, int pitch, const unsigned* pos, int n, unsigned* sink)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* t = p + pos[i];
    unsigned acc = 0;
    for (int r = 0; r < 4; r++) { unsigned v; __builtin_memcpy(&v, t + (size_t)r * pitch, 4); acc += v; }
    if (acc == 0x12345678u) *sink = acc;
}
// (f) 48 x 48 byte windows around 32 x 32 tiles (a 8-sample halo each side, the interpolation / deblocking reach): neighbouring workgroups re-read the halo
__global__ void cal_halo_48(const uint8_t* p, int pitch, int tilesX, int tilesY, unsigned* sink)
{
    const int tx = blockIdx.x % tilesX, ty = blockIdx.x / tilesX;
    if (ty >= tilesY) return;
    const uint8_t* t = p + (size_t)(ty * 32 + 8) * pitch + tx * 32 + 8 - 8 * pitch - 8;
    unsigned acc = 0;
    for (int i = threadIdx.x; i < 48 * 12; i += blockDim.x) acc += *(const unsigned*)(t + (size_t)(i / 12) * pitch + 4 * (i % 12));
    if (acc == 0x12345678u) *sink = acc;
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    uint8_t* buf; unsigned* sink;
    CHECK(hipMalloc(&buf, bytes + 4096));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 1, bytes + 4096));
    const int pitch = 16384, rows = (int)(bytes / pitch);            // the buffer as a 16384 x 65536 byte plane
    const int tilesX = pitch / 32, tilesY = rows / 32;
    // scattered 4x4 tiles: 4 M tiles at pseudo-random positions (LCG), the distinct lines they touch counted on the host
    const int nt = 4 << 20;
    std::vector<unsigned> pos(nt);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    std::vector<uint8_t> l64(bytes >> 6, 0), l128(bytes >> 7, 0);
    for (int i = 0; i < nt; i++)
    {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const unsigned x = (unsigned)((s >> 33) % (pitch - 8)), y = (unsigned)((s >> 13) % (unsigned)(rows - 8));
        pos[i] = y * (unsigned)pitch + x;
        for (int r = 0; r < 4; r++)
            for (unsigned b = pos[i] + r * pitch; b <= pos[i] + r * pitch + 3; b += 3) { l64[b >> 6] = 1; l128[b >> 7] = 1; }
    }
    size_t n64 = 0, n128 = 0;
    for (uint8_t v : l64) n64 += v;
    for (uint8_t v : l128) n128 += v;
    unsigned* dpos;
    CHECK(hipMalloc(&dpos, (size_t)nt * 4));
    CHECK(hipMemcpy(dpos, pos.data(), (size_t)nt * 4, hipMemcpyHostToDevice));
    const size_t haloReq = (size_t)tilesX * (tilesY - 1) * 48 * 48;
    printf("# kernel requested_bytes distinct_64B_line_bytes distinct_128B_line_bytes\n");
    printf("cal_stream_b128 %zu %zu %zu\n", bytes, bytes, bytes);
    printf("cal_stream_b32 %zu %zu %zu\n", bytes, bytes, bytes);
    printf("cal_stream_b8 %zu %zu %zu\n", bytes / 4, bytes / 4, bytes / 4);
    printf("cal_tile_rows32 %zu %zu %zu\n", bytes, bytes, bytes);
    printf("cal_scatter_4x4 %zu %zu %zu\n", (size_t)nt * 16, n64 * 64, n128 * 128);
    printf("cal_halo_48 %zu %zu %zu\n", haloReq, bytes, bytes);
    for (int rep = 0; rep < 3; rep++)
    {
        hipLaunchKernelGGL(cal_stream_b128, dim3(4096), dim3(256), 0, 0, (const v4u*)buf, bytes / 16, sink);
        hipLaunchKernelGGL(cal_stream_b32, dim3(4096), dim3(256), 0, 0, (const unsigned*)buf, bytes / 4, sink);
        hipLaunchKernelGGL(cal_stream_b8, dim3(4096), dim3(256), 0, 0, buf, bytes / 4, sink);
        hipLaunchKernelGGL(cal_tile_rows32, dim3((tilesX * tilesY + 3) / 4), dim3(256), 0, 0, buf, pitch, tilesX, tilesY, sink);
        hipLaunchKernelGGL(cal_scatter_4x4, dim3((nt + 255) / 256), dim3(256), 0, 0, buf, pitch, dpos, nt, sink);
        hipLaunchKernelGGL(cal_halo_48, dim3(tilesX * (tilesY - 1)), dim3(256), 0, 0, buf, pitch, tilesX, tilesY - 1, sink);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipGetLastError());
    return 0;
}
