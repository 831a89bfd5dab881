/* oracle/ref_encode.cpp - TEST INFRASTRUCTURE, never part of the product path.
 *
 * Drives the REAL reference encoder (compiled from /root/reference by oracle/Makefile into
 * oracle/_ref/libx265ref<depth>.so) through its public C API, optionally with the global primitive
 * table pre-filled by a drop-in filler (x265hip_setup_primitives).  This is tier T3 of SURVEY.md
 * section 7: whole-bitstream equality between the reference's own C primitives and the MI355X path,
 * and the integration recipe INTEGRATION.md documents:
 *
 *     fill the table the way x265_setup_primitives() would (primitives.cpp:248-282), let the filler
 *     overwrite its slots, THEN call x265_encoder_open(): a pre-filled table is kept (primitives.cpp:250).
 */
#include "common.h"
#include "primitives.h"
#include "x265.h"

#include <chrono>
#include <cstring>
#include <vector>

using namespace X265_NS;

extern "C" void x265ref_encoder_table_reset_c(void);

typedef int (*table_filler_t)(void* table, size_t bytes, int depth);

extern "C" {

/* yuv: nframes of planar 4:2:0 (Y, U, V), samples of 1 byte (8-bit build) or 2 bytes.
 * opts: nopts pairs of (name, value) passed to x265_param_parse; value may be NULL.
 * Returns the number of bitstream bytes written to out (<= cap), negative on error. */
long x265ref_encode(const void* yuv, int width, int height, int nframes, const char* preset,
                    const char* const* opts, int nopts, table_filler_t filler,
                    unsigned char* out, long cap, double* seconds, int* slotsFilled)
{
    x265_param* param = x265_param_alloc();
    if (!param) return -1;
    if (x265_param_default_preset(param, preset, NULL) < 0) return -2;
    param->sourceWidth = width;
    param->sourceHeight = height;
    param->fpsNum = 30; param->fpsDenom = 1;
    param->internalCsp = X265_CSP_I420;
    param->bRepeatHeaders = 1;
    param->bEmitInfoSEI = 0;              /* --no-info: cpuid / thread counts are serialised otherwise (param.cpp:2099) */
    param->logLevel = X265_LOG_NONE;
    for (int i = 0; i < nopts; i++)
        if (x265_param_parse(param, opts[2 * i], opts[2 * i + 1]) < 0) return -3;

    /* the drop-in hook */
    x265ref_encoder_table_reset_c();
    int filled = 0;
    if (filler)
    {
        filled = filler(&primitives, sizeof(primitives), X265_DEPTH);
        if (filled < 0) return -4;
    }
    if (slotsFilled) *slotsFilled = filled;

    x265_encoder* enc = x265_encoder_open(param);
    if (!enc) return -5;
    x265_picture pic;
    x265_picture_init(param, &pic);
    const size_t es = X265_DEPTH > 8 ? 2 : 1;
    const size_t ysz = (size_t)width * height * es, csz = (size_t)(width / 2) * (height / 2) * es;
    long total = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; ; f++)
    {
        x265_nal* nals = NULL; uint32_t nnal = 0;
        int ret;
        if (f < nframes)
        {
            const unsigned char* base = (const unsigned char*)yuv + (size_t)f * (ysz + 2 * csz);
            pic.planes[0] = (void*)base; pic.planes[1] = (void*)(base + ysz); pic.planes[2] = (void*)(base + ysz + csz);
            pic.stride[0] = (int)(width * es); pic.stride[1] = pic.stride[2] = (int)((width / 2) * es);
            pic.bitDepth = X265_DEPTH;
            pic.pts = f;
            ret = x265_encoder_encode(enc, &nals, &nnal, &pic, NULL);
        }
        else
            ret = x265_encoder_encode(enc, &nals, &nnal, NULL, NULL);
        if (ret < 0) { total = -6; break; }
        for (uint32_t i = 0; i < nnal; i++)
        {
            if (total + (long)nals[i].sizeBytes > cap) { total = -7; break; }
            memcpy(out + total, nals[i].payload, nals[i].sizeBytes);
            total += nals[i].sizeBytes;
        }
        if (total < 0) break;
        if (f >= nframes && ret <= 0) break;
    }
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    x265_encoder_close(enc);
    x265_param_free(param);
    x265_cleanup();
    return total;
}

} // extern "C"
