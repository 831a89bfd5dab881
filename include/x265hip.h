/* x265hip.h - C ABI of the MI355X-native block-primitive path (libx265hip.so).
 *
 * Drop-in boundary = the reference's EncoderPrimitives function-pointer table
 * (reference: source/common/primitives.h:237-429; filled by setupCPrimitives /
 * setupAssemblyPrimitives, primitives.h:468-471; entry x265_setup_primitives,
 * primitives.cpp:248).  Two layers sit behind it:
 *
 *   1. TABLE LAYER  - x265hip_setup_primitives() overwrites slots of a caller-owned table with
 *      synchronous host-pointer stubs of exactly the reference's typedef'd signatures
 *      (primitives.h:133-234).  Call it on `x265::primitives` BEFORE x265_encoder_open():
 *      x265_setup_primitives keeps a pre-filled table (primitives.cpp:250).  Every stub stages its
 *      operands to the GPU, runs the same HIP kernel the batch layer uses (batch of one) and copies
 *      the result back - bit-exact, re-entrant (per-thread stream + staging), slow per call.
 *
 *   2. BATCH LAYER  - device-pointer entry points that evaluate MANY blocks per launch (one
 *      wavefront or sub-wavefront group per PU/TU candidate; CTU search windows staged in LDS).
 *      This is the performance path bench.py measures; all pointers are DEVICE pointers, strides
 *      are in ELEMENTS (pixels / int16 / int32), `stream` is a hipStream_t (NULL = default).
 *
 * `depth` is the encoder bit depth: 8 -> pixel = uint8_t, 10/12 -> pixel = uint16_t
 * (reference common.h:126-148).  All arithmetic is integer and bit-exact vs the reference C
 * primitives.  Functions return 0 on success, a negative X265HIP_E* code otherwise; there is NO
 * CPU fallback anywhere in this library - without a usable HIP device every entry fails loudly.
 */
#ifndef X265HIP_H
#define X265HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define X265HIP_ME_PUS_PER_CTU 85  /* 64 + 16 + 4 + 1 */
#define X265HIP_OK            0
#define X265HIP_ENODEV       -1   /* no HIP device / runtime error (see x265hip_last_error) */
#define X265HIP_EINVAL       -2   /* bad argument (size, depth, alignment, NULL) */
#define X265HIP_EUNSUPPORTED -3   /* valid in the reference, not implemented on the GPU path */

const char* x265hip_version(void);
const char* x265hip_last_error(void);          /* thread-local text of the last failure */
int         x265hip_device_count(void);
int         x265hip_init(int device);          /* select device, create per-process state */

/* ------------------------------------------------------------------ 1. table layer */
/* Overwrite the GPU-backed slots of an EncoderPrimitives-layout table (18240 bytes, see
 * include/x265hip_table.h) for the given bit depth.  Slots the GPU path does not implement are
 * left untouched (the caller's C/asm entries stay).  `table_bytes` must equal the table size.
 * Returns the number of slots written, or a negative error. */
int x265hip_setup_primitives(void* table, size_t table_bytes, int depth);
/* Number of table calls served by the GPU since init (to prove stubs really ran). */
uint64_t x265hip_table_calls(void);

/* ------------------------------------------------------------------ 2. batch layer */
/* pixel-compare family (reference pixel.cpp:40-55 sad, :210-297 satd, :299-377 sa8d, :167-186 sse,
 * :726-757 psyCost): out[i] = f(a + a_off[i], a_stride, b + b_off[i], b_stride) for a WxH block.
 * a_off / b_off are device arrays of element offsets (NULL = i * a_step / i * b_step). */
enum x265hip_cmp_kind
{
    X265HIP_CMP_SAD = 0, X265HIP_CMP_SATD = 1, X265HIP_CMP_SA8D = 2, X265HIP_CMP_SSE_PP = 3,
    X265HIP_CMP_PSY_COST = 4
};
int x265hip_pixelcmp_batch(int kind, int depth, int w, int h,
                           const void* a, intptr_t a_stride, const int64_t* a_off, int64_t a_step,
                           const void* b, intptr_t b_stride, const int64_t* b_off, int64_t b_step,
                           int njobs, uint64_t* out /* one u64 per job (int results zero-extended) */,
                           void* stream);

/* CTU-tiled exhaustive integer motion search (the batched form of pu[].sad / sad_x3 / sad_x4 as
 * issued by the reference's full search, motion.cpp:1395-1430: every mv in [-range, range]^2,
 * raster order, strict '<' tie-break).  The frame is processed in 64x64 CTUs; each CTU's
 * (64+2*range)^2 reference window is staged in LDS once and SAD is evaluated for all 8x8 blocks,
 * then summed hierarchically into the 16x16 / 32x32 / 64x64 PUs (SAD is additive, so every level
 * equals pu[LUMA_NxN].sad on the same pixels).
 *
 *   fenc, fref : luma planes, (0,0) pixel pointers; fref must have >= range + 8 valid pixels of
 *                margin on every side (reference picyuv.cpp:87-114 guarantees 96 / 80).
 *   width, height : multiples of 64 (the reference allocates whole CTUs).
 *   surf       : optional SAD surfaces, int32 [ctu][mvy][mvx][85]: one 85-int record per motion
 *                vector holding every PU of the CTU - [0,64) the 8x8 PUs, [64,80) the 16x16, [80,84)
 *                the 32x32, [84] the 64x64, each group in z-order inside the CTU.
 *   best       : optional per-PU minimum of (sad + cost_x[mvx] + cost_y[mvy]), uint64 [ctu][85] (same
 *                PU order) = cost << 32 | (mvy_index * (2*range+1) + mvx_index); must be pre-set to
 *                all-ones by the caller (x265hip_me_best_reset).  Ties resolve to the smallest raster
 *                index, i.e. the reference's scan order with its strict '<'.
 *   cost_x/y   : uint16 [2*range+1] mv bit-cost tables built on the host (bitcost.cpp:51-55).
 */
typedef struct x265hip_me_params
{
    int depth;
    int width, height;
    int range;
    const void* fenc;  intptr_t fenc_stride;
    const void* fref;  intptr_t fref_stride;
    int32_t*  surf;
    uint64_t* best;
    const uint16_t* cost_x;
    const uint16_t* cost_y;
} x265hip_me_params;
int x265hip_me_fullsearch(const x265hip_me_params* p, void* stream);
int x265hip_me_best_reset(uint64_t* best, size_t count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* X265HIP_H */
