// sea_kernels.hip - the block-sum ("integral") planes X265_SEA filters its candidates with.
//
// Reference semantics: FrameFilter::processPostRow's integral section (encoder/framefilter.cpp:716-823) runs the
// integral_init{4,8,12,16,24,32}h / v primitives (:39-140) over every row of the padded reconstruction, leaving in plane k the
// sum of the bw x bh block of samples whose top-left corner is (x, y), for the twelve (bw, bh) of framedata.h:171.  The
// reference gets there through running column sums because a CPU walks rows; here the two separable passes are sliding windows
// per thread (16 outputs each), the row-sum pass shared by the planes of equal bw.
#include "common.h"

namespace x265hip {

static const int kSeaDims[12][2] = { { 32, 32 }, { 32, 24 }, { 32, 8 }, { 24, 32 }, { 16, 16 }, { 16, 12 }, { 16, 4 }, { 12, 16 },
                                     { 8, 32 }, { 8, 8 }, { 4, 16 }, { 4, 4 } };
constexpr int kSeaSeg = 16;

// tmp(x, y) = sum of ref(x .. x + bw - 1, y) for x0 <= x <= x1, y0 <= y <= y1; one thread per 16 consecutive x
template <typename Px>
__global__ void __launch_bounds__(256) sea_hsum_kernel(const Px* ref, long stride, uint32_t* tmp, int x0, int x1, int y0, int y1, int bw)
{
    const int nseg = (x1 - x0 + kSeaSeg) / kSeaSeg;
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= (long)nseg * (y1 - y0 + 1)) return;
    const int row = (int)(i / nseg), seg = (int)(i - (long)row * nseg);
    const int y = y0 + row, xs = x0 + seg * kSeaSeg, xe = min(xs + kSeaSeg - 1, x1);
    const Px* r = ref + (long)y * stride;
    uint32_t* o = tmp + (long)y * stride;
    uint32_t v = 0;
    for (int k = 0; k < bw; k++) v += r[xs + k];
    for (int x = xs; ; x++)
    {
        o[x] = v;
        if (x == xe) break;
        v += (uint32_t)r[x + bw] - (uint32_t)r[x];
    }
}

// out(x, y) = sum of tmp(x, y .. y + bh - 1) for x0 <= x <= x1, y0 <= y <= y1; one thread per column and 16 consecutive rows
__global__ void __launch_bounds__(256) sea_vsum_kernel(const uint32_t* tmp, long stride, uint32_t* out, int x0, int x1, int y0, int y1, int bh)
{
    const int nx = x1 - x0 + 1, nseg = (y1 - y0 + kSeaSeg) / kSeaSeg;
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= (long)nx * nseg) return;
    const int seg = (int)(i / nx), x = x0 + (int)(i - (long)seg * nx);
    const int ys = y0 + seg * kSeaSeg, ye = min(ys + kSeaSeg - 1, y1);
    uint32_t v = 0;
    for (int k = 0; k < bh; k++) v += tmp[(long)(ys + k) * stride + x];
    for (int y = ys; ; y++)
    {
        out[(long)y * stride + x] = v;
        if (y == ye) break;
        v += tmp[(long)(y + bh) * stride + x] - tmp[(long)y * stride + x];
    }
}

} // namespace x265hip

using namespace x265hip;

extern "C" int x265hip_sea_integral(const x265hip_sea_integral_params* p, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!p || !p->ref) { set_error("sea_integral: NULL operand"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("sea_integral: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->width <= 0 || p->height <= 0 || p->margin_x < 0 || p->margin_y < 0 || p->stride < p->width + 2 * p->margin_x)
    { set_error("sea_integral: bad geometry"); return X265HIP_EINVAL; }
    if (p->width + 2 * p->margin_x < 32 || p->height + 2 * p->margin_y < 32) { set_error("sea_integral: padded picture smaller than 32x32"); return X265HIP_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    const long stride = (long)p->stride;
    const int W = p->width, H = p->height, mx = p->margin_x, my = p->margin_y;
    uint32_t* tmpBase = nullptr;
    X265HIP_TRY(hipMallocAsync((void**)&tmpBase, sizeof(uint32_t) * (size_t)stride * (H + 2 * my), s));
    uint32_t* tmp = tmpBase + (long)my * stride + mx;
    int lastBw = 0;
    for (int k = 0; k < 12; k++)
    {
        if (!p->planes[k]) continue;
        const int bw = kSeaDims[k][0], bh = kSeaDims[k][1];
        const int x0 = -mx, x1 = W + mx - bw, y0 = -my;
        if (bw != lastBw)
        {
            const long n = (long)((x1 - x0 + kSeaSeg) / kSeaSeg) * (H + 2 * my);
            if (p->depth == 8)
                hipLaunchKernelGGL(sea_hsum_kernel<uint8_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint8_t*)p->ref, stride, tmp, x0, x1, y0, H + my - 1, bw);
            else
                hipLaunchKernelGGL(sea_hsum_kernel<uint16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint16_t*)p->ref, stride, tmp, x0, x1, y0, H + my - 1, bw);
            lastBw = bw;
        }
        const int y1 = H + my - bh;
        const long n = (long)(x1 - x0 + 1) * ((y1 - y0 + kSeaSeg) / kSeaSeg);
        hipLaunchKernelGGL(sea_vsum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint32_t*)tmp, stride, p->planes[k], x0, x1, y0, y1, bh);
    }
    X265HIP_TRY(hipGetLastError());
    X265HIP_TRY(hipFreeAsync(tmpBase, s));
    return 0;
}
