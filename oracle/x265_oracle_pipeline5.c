/* oracle/x265_oracle_pipeline5.c
 *
 * TEST INFRASTRUCTURE - NOT PRODUCT CODE (same rules as x265_oracle.c).
 *
 * Stage: the fractional-phase planes of a reference plane (x265hip_phase_planes).  Restated ON TOP OF THE ORACLE'S PRIMITIVE TABLE -
 * whose interpolation entries are pinned against the real reference - exactly the way the reference produces the samples:
 * MotionEstimate::subpelCompare (motion.cpp:1571-1664) and Predict::predInterLumaPixel / predInterChromaPixel (predict.cpp:261-351)
 *   luma   xFrac only -> pu[].luma_hpp, yFrac only -> pu[].luma_vpp, both -> pu[].luma_hvpp
 *   chroma xFrac only -> filter_hpp,    yFrac only -> filter_vpp,    both -> filter_hps(isRowExt = 1) + filter_vsp on row halfFilterSize - 1
 * applied block by block (8x8 blocks) over the interior of the plane: blocks whose taps stay inside the buffer.  The 8-sample border
 * of every output plane is left zero (no valid block lies there; the product's planes are undefined there). */
#ifndef X265HIP_DEPTH
#error "compile with -DX265HIP_DEPTH=8|10|12"
#endif
#include "x265hip_table.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef x265hip_pixel pixel;
#define CAT_(a, b)   a##b
#define CAT(a, b)    CAT_(a, b)
#define EXPORT(name) CAT(CAT(name, _d), X265HIP_DEPTH)

void EXPORT(x265oracle_setup_primitives)(x265hip_EncoderPrimitives* p);
void EXPORT(x265oracle_prims_once)(x265hip_EncoderPrimitives* p, int* state);

/* src: stride x rows samples; dst: 15 (luma) or 63 (chroma) planes of the same geometry, zero-filled by the caller or not - the
 * interior is written, the border zeroed here */
void EXPORT(x265oracle_phase_planes)(const pixel* src, intptr_t stride, int rows, int chroma, pixel* dst)
{
    static x265hip_EncoderPrimitives prim;
    static int ready;
    EXPORT(x265oracle_prims_once)(&prim, &ready);
    const int nph = chroma ? 63 : 15, mask = chroma ? 7 : 3, sh = chroma ? 3 : 2;
    const size_t plane = (size_t)stride * rows;
    memset(dst, 0, plane * nph * sizeof(pixel));
    /* chroma blocks are addressed by the LUMA partition (primitives.h:77-79): LUMA_16x16 is the 8x8 chroma block of 4:2:0 */
    /* chroma[1] = X265_CSP_I420 (x265.h) */
    const int part = chroma ? X265HIP_LUMA_16x16 : X265HIP_LUMA_8x8;
#pragma omp parallel for schedule(dynamic)
    for (int job = 0; job < nph * (rows / 8 - 2); job++)
    {
        const int ph = job / (rows / 8 - 2) + 1, by = job % (rows / 8 - 2) + 1;
        const int xf = ph & mask, yf = ph >> sh;
        pixel* out = dst + (size_t)(ph - 1) * plane;
        int16_t immed[8 * (8 + 3)];
        for (int bx = 1; bx < stride / 8 - 1; bx++)
        {
            const pixel* s = src + (size_t)by * 8 * stride + bx * 8;
            pixel* d = out + (size_t)by * 8 * stride + bx * 8;
            if (!chroma)
            {
                if (!yf) prim.pu[part].luma_hpp(s, stride, d, stride, xf);
                else if (!xf) prim.pu[part].luma_vpp(s, stride, d, stride, yf);
                else prim.pu[part].luma_hvpp(s, stride, d, stride, xf, yf);
            }
            else
            {
                if (!yf) prim.chroma[1].pu[part].filter_hpp(s, stride, d, stride, xf);
                else if (!xf) prim.chroma[1].pu[part].filter_vpp(s, stride, d, stride, yf);
                else
                {
                    prim.chroma[1].pu[part].filter_hps(s, stride, immed, 8, xf, 1);
                    prim.chroma[1].pu[part].filter_vsp(immed + (4 / 2 - 1) * 8, 8, d, stride, yf);
                }
            }
        }
    }
}
