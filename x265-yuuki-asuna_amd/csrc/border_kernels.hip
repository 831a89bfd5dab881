// border_kernels.hip - picture border extension on gfx950.
//
// Reference semantics: extendPicBorder (source/common/pixel.cpp:1027-1041) = primitives.extendRowBorder
// (extendCURowColBorder, source/common/ipfilter.cpp:59-77: every row's first / last pixel replicated marginX
// times to the left / right) followed by replicating the extended top and bottom rows marginY times.  The net
// effect on every margin sample is pic[clamp(y)][clamp(x)], which is what each thread writes - one pass, no
// ordering between rows and columns needed.  Keeps a reconstructed picture usable as a motion-search
// reference without leaving HBM (framefilter.cpp:346-436 does this on the host after every CTU row).
#include "common.h"

namespace x265hip {

template <typename Px>
__global__ void __launch_bounds__(256) extend_border_kernel(Px* pic, long stride, int width, int height, int marginX, int marginY)
{
    const int pw = width + 2 * marginX, ph = height + 2 * marginY;
    // threads walk the padded plane row-major and skip the interior
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)pw * ph; i += (long)gridDim.x * blockDim.x)
    {
        const int y = (int)(i / pw) - marginY, x = (int)(i % pw) - marginX;
        if (x >= 0 && x < width && y >= 0 && y < height)
            continue;
        const int sx = x < 0 ? 0 : (x >= width ? width - 1 : x), sy = y < 0 ? 0 : (y >= height ? height - 1 : y);
        pic[(long)y * stride + x] = pic[(long)sy * stride + sx];
    }
}

} // namespace x265hip

using namespace x265hip;

extern "C" int x265hip_extend_border(void* pic, intptr_t stride, int width, int height, int margin_x, int margin_y, int depth, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!pic || width <= 0 || height <= 0 || margin_x < 0 || margin_y < 0) { set_error("extend_border: bad argument"); return X265HIP_EINVAL; }
    if (depth != 8 && depth != 10 && depth != 12) { set_error("extend_border: depth %d", depth); return X265HIP_EINVAL; }
    const long total = (long)(width + 2 * margin_x) * (height + 2 * margin_y);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (depth == 8) hipLaunchKernelGGL(extend_border_kernel<uint8_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (uint8_t*)pic, (long)stride, width, height, margin_x, margin_y);
    else hipLaunchKernelGGL(extend_border_kernel<uint16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (uint16_t*)pic, (long)stride, width, height, margin_x, margin_y);
    X265HIP_TRY(hipGetLastError());
    return 0;
}
