/* oracle/ref_lookahead.cpp - TEST INFRASTRUCTURE, never part of the product path.
 *
 * C-ABI window onto the REAL reference lookahead preparation: builds x265::PicYuv / x265::Lowres / x265::LookaheadTLD
 * objects over a caller-supplied padded luma plane and runs Lowres::init() (common/lowres.cpp:245-306) and
 * LookaheadTLD::lowresIntraEstimate() (encoder/slicetype.cpp:696-772), then copies the half-resolution planes and the
 * per-block results out.  The tests use it to pin oracle/x265_oracle_pipeline3.c against the reference classes.
 */
#include "common.h"
#include "primitives.h"
#include "picyuv.h"
#include "lowres.h"
#include "slicetype.h"
#include "x265.h"

#include <cstring>

using namespace X265_NS;

extern "C" void x265ref_encoder_table_reset_c(void);

extern "C" {

/* plane: the ALLOCATION START of a padded luma plane with the reference's PicYuv geometry for `width` x `height`
 * (maxCUSize 64: margins 96 / 80, stride = ceil(width / 64) * 64 + 192); it is copied wholesale.
 * Outputs: geometry[] = { lowres width, lines, lumaStride, blocks per row, blocks per column };
 * planes[i]: room for lumaStride * (lines + 2 * 80) pixels each, receives the whole padded plane i;
 * intraCost / intraMode / lowresCosts: one entry per 8x8 lowres block.  Returns 0 on success. */
int x265ref_lowres_intra(const void* plane, int width, int height, int* geometry,
                         void* plane0, void* plane1, void* plane2, void* plane3,
                         int32_t* intraCost, uint8_t* intraMode, uint16_t* lowresCosts)
{
    static bool tableReady = false;
    if (!tableReady) { x265ref_encoder_table_reset_c(); tableReady = true; }
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width;
    param->sourceHeight = height;
    param->internalCsp = X265_CSP_I400;
    param->maxCUSize = 64;
    param->rc.aqMode = 0;
    param->rc.hevcAq = 0;
    param->bAQMotion = 0;
    param->bEnableHME = 0;
    PicYuv pic;
    pic.m_param = param;
    if (!pic.create(param, true)) return -1;
    const int h64 = (height + 63) / 64 * 64;
    memcpy(pic.m_picOrg[0] - pic.m_lumaMarginY * pic.m_stride - pic.m_lumaMarginX, plane,
           sizeof(pixel) * pic.m_stride * (h64 + 2 * pic.m_lumaMarginY));
    Lowres lr;
    memset((void*)&lr, 0, sizeof(lr));
    const uint32_t qgSize = 32;
    if (!lr.create(param, &pic, qgSize)) return -2;
    lr.init(&pic, 0);
    LookaheadTLD tld;
    tld.init(lr.maxBlocksInRow, lr.maxBlocksInCol, lr.maxBlocksInRow * lr.maxBlocksInCol);
    tld.lowresIntraEstimate(lr, qgSize);

    geometry[0] = lr.width; geometry[1] = lr.lines; geometry[2] = (int)lr.lumaStride;
    geometry[3] = lr.maxBlocksInRow; geometry[4] = lr.maxBlocksInCol;
    const size_t planesize = (size_t)lr.lumaStride * (lr.lines + 2 * pic.m_lumaMarginY);
    void* outs[4] = { plane0, plane1, plane2, plane3 };
    for (int i = 0; i < 4; i++)
        memcpy(outs[i], lr.lowresPlane[i] - pic.m_lumaMarginY * lr.lumaStride - pic.m_lumaMarginX, planesize * sizeof(pixel));
    const int ncu = lr.maxBlocksInRow * lr.maxBlocksInCol;
    memcpy(intraCost, lr.intraCost, ncu * sizeof(int32_t));
    memcpy(intraMode, lr.intraMode, ncu * sizeof(uint8_t));
    memcpy(lowresCosts, lr.lowresCosts[0][0], ncu * sizeof(uint16_t));
    lr.destroy();
    pic.destroy();
    x265_param_free(param);
    return 0;
}

} // extern "C"
