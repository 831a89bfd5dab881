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

struct BorderArgs
{
    void* pic[4];                     // up to four planes of one geometry (blockIdx.y selects)
    long stride;
    int width, height, marginX, marginY;   // marginY: rows above
    int marginBottom;                      // rows below (the whole-picture entry passes marginY again)
};

// Only the margin is walked: first the bands above and below the picture (full padded width), then the left / right bands of the
// picture rows.
template <typename Px>
__global__ void __launch_bounds__(256) extend_border_kernel(BorderArgs a)
{
    Px* pic = reinterpret_cast<Px*>(a.pic[blockIdx.y]);
    const int width = a.width, height = a.height, marginX = a.marginX, marginY = a.marginY;
    const int pw = width + 2 * marginX, sideW = 2 * marginX;
    const long nBands = (long)pw * (marginY + a.marginBottom), nSides = (long)height * sideW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nBands + nSides; i += (long)gridDim.x * blockDim.x)
    {
        int x, y;
        if (i < nBands)
        {
            const int r = (int)(i / pw);
            x = (int)(i - (long)r * pw) - marginX;
            y = r < marginY ? r - marginY : height + (r - marginY);
        }
        else
        {
            const long j = i - nBands;
            y = (int)(j / sideW);
            const int k = (int)(j - (long)y * sideW);
            x = k < marginX ? k - marginX : width + (k - marginX);
        }
        const int sx = x < 0 ? 0 : (x >= width ? width - 1 : x), sy = y < 0 ? 0 : (y >= height ? height - 1 : y);
        pic[(long)y * a.stride + x] = pic[(long)sy * a.stride + sx];
    }
}

// nplanes planes of one geometry in one launch (the four lookahead planes)
// the planes of one picture, each with a geometry of its own (blockIdx.y = plane)
struct BorderArgs3 { void* pic[3]; long stride[3]; int width[3], height[3], marginX[3], marginY[3], marginBottom[3]; };

template <typename Px>
__global__ void __launch_bounds__(256) extend_border_planes_kernel(BorderArgs3 a)
{
    // a thread writes FOUR consecutive margin samples (one 4- / 8-byte store; the first version moved one sample per thread): the items are quads of the
    // top / bottom margin rows (whole padded rows: copies of the first / last picture row, whose own side margins are the edge samples) and quads of the
    // left / right margins of the picture's rows (the row's edge sample four times).  Widths and margins of x265's planes are multiples of 4.
    const int p = blockIdx.y;
    Px* pic = reinterpret_cast<Px*>(a.pic[p]);
    const int width = a.width[p], height = a.height[p], marginX = a.marginX[p], marginY = a.marginY[p];
    const long stride = a.stride[p];
    const int pw = width + 2 * marginX, sideW = 2 * marginX;
    const bool quads = !((width | marginX) & 3);
    const int Q = quads ? 4 : 1;
    const long nBands = (long)(pw / Q) * (marginY + a.marginBottom[p]), nSides = (long)height * (sideW / Q);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nBands + nSides; i += (long)gridDim.x * blockDim.x)
    {
        int x, y;
        if (i < nBands)
        {
            const int r = (int)(i / (pw / Q));
            x = (int)(i - (long)r * (pw / Q)) * Q - marginX;
            y = r < marginY ? r - marginY : height + (r - marginY);
        }
        else
        {
            const long j = i - nBands;
            y = (int)(j / (sideW / Q));
            const int k = (int)(j - (long)y * (sideW / Q)) * Q;
            x = k < marginX ? k - marginX : width + (k - marginX);
        }
        const int sy = y < 0 ? 0 : (y >= height ? height - 1 : y);
        Px v[4];
        const Px* srow = pic + (long)sy * stride;
        if (quads && x >= 0 && x + 4 <= width)
        {   // a quad never straddles the picture edge (x, width, marginX are multiples of 4): four samples of the first / last row
            const uint8_t* b = reinterpret_cast<const uint8_t*>(srow + x);
            if (sizeof(Px) == 1) { const uint32_t w = ld_u32(b); v[0] = (Px)(w & 0xff); v[1] = (Px)((w >> 8) & 0xff); v[2] = (Px)((w >> 16) & 0xff); v[3] = (Px)(w >> 24); }
            else { const uint32_t w0 = ld_u32(b), w1 = ld_u32(b + 4); v[0] = (Px)(w0 & 0xffff); v[1] = (Px)(w0 >> 16); v[2] = (Px)(w1 & 0xffff); v[3] = (Px)(w1 >> 16); }
        }
        else
        {
            const Px e = srow[x < 0 ? 0 : (x >= width ? width - 1 : x)];
            v[0] = v[1] = v[2] = v[3] = e;
        }
        Px* d = pic + (long)y * stride + x;
        if (!quads) d[0] = v[0];
        else if (sizeof(Px) == 1) *reinterpret_cast<u32_unaligned*>(d) = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
        else
        {
            reinterpret_cast<u32_unaligned*>(d)[0] = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
            reinterpret_cast<u32_unaligned*>(d)[1] = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
        }
    }
}

int extend_borders_tb(void* const* pics, int nplanes, intptr_t stride, int width, int height, int margin_x, int margin_y, int depth, hipStream_t s,
                      int margin_bottom)
{
    BorderArgs a;
    for (int i = 0; i < 4; i++) a.pic[i] = i < nplanes ? pics[i] : nullptr;
    a.stride = (long)stride; a.width = width; a.height = height; a.marginX = margin_x; a.marginY = margin_y;
    a.marginBottom = margin_bottom < 0 ? margin_y : margin_bottom;
    const long total = (long)(width + 2 * margin_x) * (margin_y + a.marginBottom) + (long)height * 2 * margin_x;
    if (total <= 0) return 0;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (depth == 8) hipLaunchKernelGGL(extend_border_kernel<uint8_t>, dim3(blocks, nplanes), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(extend_border_kernel<uint16_t>, dim3(blocks, nplanes), dim3(256), 0, s, a);
    X265HIP_TRY(hipGetLastError());
    return 0;
}

int extend_borders(void* const* pics, int nplanes, intptr_t stride, int width, int height, int margin_x, int margin_y, int depth, hipStream_t s)
{
    return extend_borders_tb(pics, nplanes, stride, width, height, margin_x, margin_y, depth, s, -1);
}

} // namespace x265hip

using namespace x265hip;

extern "C" int x265hip_extend_border(void* pic, intptr_t stride, int width, int height, int margin_x, int margin_y, int depth, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!pic || width <= 0 || height <= 0 || margin_x < 0 || margin_y < 0) { set_error("extend_border: bad argument"); return X265HIP_EINVAL; }
    if (depth != 8 && depth != 10 && depth != 12) { set_error("extend_border: depth %d", depth); return X265HIP_EINVAL; }
    void* pics[1] = { pic };
    return extend_borders(pics, 1, stride, width, height, margin_x, margin_y, depth, (hipStream_t)stream);
}

extern "C" int x265hip_extend_border_planes(const x265hip_border_plane* planes, int nplanes, int depth, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!planes || nplanes < 1 || nplanes > 3) { set_error("extend_border_planes: %d planes", nplanes); return X265HIP_EINVAL; }
    if (depth != 8 && depth != 10 && depth != 12) { set_error("extend_border_planes: depth %d", depth); return X265HIP_EINVAL; }
    BorderArgs3 a = {};
    long most = 0;
    for (int i = 0; i < nplanes; i++)
    {
        const x265hip_border_plane& q = planes[i];
        if (!q.pic || q.width <= 0 || q.height <= 0 || q.margin_x < 0 || q.margin_top < 0 || q.margin_bottom < 0) { set_error("extend_border_planes: bad plane %d", i); return X265HIP_EINVAL; }
        a.pic[i] = q.pic; a.stride[i] = (long)q.stride; a.width[i] = q.width; a.height[i] = q.height; a.marginX[i] = q.margin_x; a.marginY[i] = q.margin_top; a.marginBottom[i] = q.margin_bottom;
        const int quad = ((q.width | q.margin_x) & 3) ? 1 : 4;
        const long total = ((long)(q.width + 2 * q.margin_x) * (q.margin_top + q.margin_bottom) + (long)q.height * 2 * q.margin_x) / quad;
        most = total > most ? total : most;
    }
    if (most <= 0) return 0;
    int blocks = (int)((most + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (depth == 8) hipLaunchKernelGGL(extend_border_planes_kernel<uint8_t>, dim3(blocks, nplanes), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(extend_border_planes_kernel<uint16_t>, dim3(blocks, nplanes), dim3(256), 0, (hipStream_t)stream, a);
    X265HIP_TRY(hipGetLastError());
    return 0;
}

/* A band of rows of a picture: (0,0) = the first sample of the band's first row; the left / right margins of its `height` rows are
 * filled, plus margin_top rows above (the picture's first band) and margin_bottom rows below (its last band) - FrameFilter's
 * row-wise form of the same extension (framefilter.cpp:346-436: extendRowBorder per CTU row, the top / bottom rows replicated for the
 * first / last row). */
extern "C" int x265hip_extend_border_rows(void* band, intptr_t stride, int width, int height, int margin_x, int margin_top, int margin_bottom,
                                          int depth, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (!band || width <= 0 || height <= 0 || margin_x < 0 || margin_top < 0 || margin_bottom < 0) { set_error("extend_border_rows: bad argument"); return X265HIP_EINVAL; }
    if (depth != 8 && depth != 10 && depth != 12) { set_error("extend_border_rows: depth %d", depth); return X265HIP_EINVAL; }
    void* pics[1] = { band };
    return extend_borders_tb(pics, 1, stride, width, height, margin_x, margin_top, depth, (hipStream_t)stream, margin_bottom);
}
