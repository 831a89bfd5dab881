// Micro-benchmark: HBM write bandwidth on gfx950 for (a) a plain streaming 16-B/lane store and (b) the motion-search
// surface pattern (per wavefront and step: 1 KiB contiguous + 256 B contiguous + 20 scattered dwords, 510 workgroups).
// Build: hipcc --offload-arch=gfx950 -O3 store_bw.hip -o store_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void stream_store(v4i* p, size_t n, int seed)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    v4i v = { seed, seed + 1, (int)threadIdx.x, (int)blockIdx.x };
    for (; i < n; i += stride) p[i] = v;
}

// same addressing as me_ctu_q_kernel<SURF>: [ctu][m][g][85][4] int32
__global__ void surf_store(int* surf, int NC, int NG, int seed)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int ctu = blockIdx.x;
    const bool is64 = lane >= 52 && lane < 56;
    const bool uMask = (lane & 15) < 4 || is64;
    const int uOffDw = is64 ? 84 * 4 + (lane & 3) : (80 + (lane >> 4)) * 4 + (lane & 3);
    for (int g = wave; g < NG; g += nwaves)
        for (int m = 0; m < NC; m++)
        {
            const unsigned long long gofs = (unsigned long long)((((long)ctu * NC + m) * NG + g) * 340) * 4;
            char* grp = reinterpret_cast<char*>(surf) + gofs;
            v4i v = { seed + m, seed + g, lane, ctu };
            *reinterpret_cast<v4i*>(grp + (uint32_t)(lane * 16)) = v;
            *reinterpret_cast<int*>(grp + (uint32_t)(1024 + lane * 4)) = seed + m;
            if (uMask) *reinterpret_cast<int*>(grp + (uint32_t)(uOffDw * 4)) = seed;
        }
}

int main()
{
    const int nctu = 510, NC = 115, NG = 29;
    const size_t bytes = (size_t)nctu * NC * NG * 340 * 4;
    int* d; hipMalloc(&d, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int blocks : {1024, 2048, 8192})
    {
        hipLaunchKernelGGL(stream_store, dim3(blocks), dim3(256), 0, 0, (v4i*)d, bytes / 16, 1);
        hipEventRecord(e0);
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(stream_store, dim3(blocks), dim3(256), 0, 0, (v4i*)d, bytes / 16, i);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("stream_store  %5d blocks x 256: %.3f ms per %.3f GB -> %.0f GB/s\n", blocks, ms / 5, bytes / 1e9, bytes / (ms / 5 * 1e-3) / 1e9);
    }
    for (int threads : {512, 1024})
    {
        hipLaunchKernelGGL(surf_store, dim3(nctu), dim3(threads), 0, 0, d, NC, NG, 1);
        hipEventRecord(e0);
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(surf_store, dim3(nctu), dim3(threads), 0, 0, d, NC, NG, i);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("surf_store    %5d WGs x %4d: %.3f ms per %.3f GB -> %.0f GB/s\n", nctu, threads, ms / 5, bytes / 1e9, bytes / (ms / 5 * 1e-3) / 1e9);
    }
    return 0;
}
