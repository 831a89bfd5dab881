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
#define X265HIP_EBUSY        -4   /* every entry of a bounded pool is in use; retry later or create the object with a larger pool */

const char* x265hip_version(void);
/* Before a host destroys a stream it has passed to this library: the per-stream scratch buffers (x265hip_me_search, split x265hip_lowres_cost) return to a pool the next
 * stream adopts from, the stream's enqueue lock is dropped.  Stream idle, no HIP graph captured on it still in use.  Returns the number of buffers pooled (>= 0),
 * X265HIP_EBUSY while another thread is enqueuing on the stream.  Optional: a process with a fixed set of streams never needs it. */
int x265hip_stream_release(void* stream);
const char* x265hip_last_error(void);          /* thread-local text of the last failure */
int         x265hip_device_count(void);
int         x265hip_init(int device);          /* device >= 0: validate (gfx950) and hipSetDevice() it for the calling thread; -1: validate the thread's current device and keep it */
/* How host threads wait for the device (round 6).  The runtime's default SPINS: a thread waiting 200 ms for a kernel burns 200 ms of a core (tools/ubench/wait_cpu.hip);
 * under hipDeviceScheduleBlockingSync it burns 1.5 ms, and a short launch + wait costs the same 31 us.  The consumer services wait in worker threads and inside the lookahead's /
 * AQ's / weightAnalyse's callers: with X265HIP_WAIT_BLOCK the real encode needs 8 % fewer CPU seconds at the same fps (- 0 .. 3 %: DESIGN.md 5.1).  The flag is a per-device,
 * process-wide runtime setting, so the library leaves it alone unless the host asks: this call (applies to the calling thread's current device and to every device the library
 * is initialised on afterwards) or X265HIP_WAIT=block in the environment. */
#define X265HIP_WAIT_BLOCK 0
#define X265HIP_WAIT_SPIN  1
int         x265hip_set_wait_policy(int policy);

/* ------------------------------------------------------------------ 1. table layer */
/* Overwrite the GPU-backed slots of an EncoderPrimitives-layout table (18240 bytes, see
 * include/x265hip_table.h) for the given bit depth: every slot the reference's C filler sets (primitives.cpp:250-282) is GPU-backed
 * from round 6 on - the frame-level helpers and the RDOQ helpers (rows a16 / a9) included.  Two slots, costCoeffNxN and costC1C2Flag, price
 * context-coded bins with the host's own per-state table: they are written only when x265hip_set_entropy_bits() was called before;
 * otherwise the caller's C/asm entries stay.  `table_bytes` must equal the table size.
 * Returns the number of slots written, or a negative error. */
int x265hip_setup_primitives(void* table, size_t table_bytes, int depth);
/* Number of table calls served by the GPU since init (to prove stubs really ran). */
uint64_t x265hip_table_calls(void);
/* Error policy of the table layer (the reference's slot signatures cannot report an error).  X265HIP_ON_ERROR_ABORT (default): a HIP
 * failure inside a stub aborts loudly.  X265HIP_ON_ERROR_RESTORE_HOST: the first failure is reported once on stderr, that call and every
 * later call of every GPU-backed slot are answered by the function the HOST had in the slot before x265hip_setup_primitives (its own
 * C / asm primitive, never code of this library); x265hip_table_failures() counts the failed calls. */
#define X265HIP_ON_ERROR_ABORT        0
#define X265HIP_ON_ERROR_RESTORE_HOST 1
int         x265hip_set_error_policy(int policy);
uint64_t    x265hip_table_failures(void);
void        x265hip_table_inject_failure(long n);   /* test hook: the n-th stub call from now fails */
/* the table layer keeps a stream + pinned / device staging per calling host thread, released when that thread exits */
void        x265hip_table_stage_counts(uint64_t* created, uint64_t* released);

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
 *   fenc, fref : luma planes, (0,0) pixel pointers; fref must have >= range + 12 valid pixels of
 *                margin on every side (reference picyuv.cpp:87-114 guarantees 96 / 80).
 *   width, height : multiples of 64 (the reference allocates whole CTUs).
 *   surf       : optional SAD surfaces, int32 [ctu][mvy][(2*range+1+3)/4][85][4]: for every motion-vector
 *                row, columns are stored in groups of 4 (the last group padded, pad values unspecified); a
 *                group holds, for each of the 85 PUs of the CTU, the 4 SADs of its 4 columns - exactly the
 *                res[4] a sad_x4 call on those 4 horizontal displacements returns (motion.cpp:1415-1430).
 *                PU order: [0,64) 8x8, [64,80) 16x16, [80,84) 32x32, [84] 64x64, each level in z-order.
 *   surf_format: X265HIP_SURF_I32 (default, any depth): the layout above, 1360 bytes per group.
 *                X265HIP_SURF_PACKED (8-bit only): 720 bytes per (ctu, mvy, group) -
 *                uint16 [64][4] 8x8 SADs (<= 16320), uint16 [16][4] 16x16 SADs (<= 65280), int32 [4][4] 32x32,
 *                int32 [1][4] 64x64; same values, 47 % fewer bytes through HBM (the search is write-bound).
 *                X265HIP_SURF_PACKED_T (8-bit only): the packed record cut into its 45 16-byte chunks and stored CHUNK-MAJOR inside
 *                a motion-vector row: chunk c of group g at row + (c * groups + g) * 16, row = ((ctu * (2*range+1) + mvy) * groups) * 720.
 *                Same bytes per row; the 4 x 4 = 16 horizontal displacements of 4 neighbouring groups of one PU pair share a
 *                cache line (a search that walks in x stays in it), and the kernel that owns one record per lane
 *                (csrc/me_cand_kernel.hip) stores 16 contiguous bytes per lane.
 *                X265HIP_SURF_PACKED_B (8-bit only, round 3): the same 720-byte packed records in BLOCKS of 64: the records of a CTU in raster order
 *                r = mvy_index * groups + group; block b = r / 64 holds records 64 b .. 64 b + 63 chunk-major - 16-byte chunk c (0 .. 44) of record r
 *                at ctu_base + b * 46080 + (c * 64 + r % 64) * 16, ctu_base = ctu * x265hip_surf_ctu_bytes(format, range) (the last block of a
 *                CTU is allocated whole, its unused slots are never written).  One wavefront of the record-per-lane kernel writes one block:
 *                every store instruction is one aligned KiB and a step's 45 stores one contiguous 45 KiB - the 4.9 GB of records of a 4K
 *                picture leave 9 % faster than with PACKED_T, whose stores scatter 464-byte runs over a 21 KB row (profiles/r03_me_block_major.txt).
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
    int surf_format;                /* X265HIP_SURF_* */
    /* optional (round 4), DEVICE int16 [ctu][2]: the window of CTU c covers displacements centres[2c] +- range x centres[2c+1] +- range
     * instead of (0, 0) +- range (record / raster index i still means centre + (i - range)); the caller keeps every window inside the
     * margins: |centre| + range + 12 <= margin.  NULL = windows centred on (0, 0) */
    const int16_t* centres;
} x265hip_me_params;
enum { X265HIP_SURF_I32 = 0, X265HIP_SURF_PACKED = 1, X265HIP_SURF_PACKED_T = 2, X265HIP_SURF_PACKED_B = 3 };
#define X265HIP_SURF_GROUP_BYTES_I32    1360
#define X265HIP_SURF_GROUP_BYTES_PACKED 720
#define X265HIP_SURF_BLOCK_BYTES_PACKED (64 * X265HIP_SURF_GROUP_BYTES_PACKED)      /* X265HIP_SURF_PACKED_B: 64 records */
/* bytes of one CTU's surfaces in a format (a surface buffer is ctus * this) */
static inline size_t x265hip_surf_ctu_bytes(int surf_format, int range)
{
    const size_t nc = (size_t)(2 * range + 1), recs = nc * ((nc + 3) >> 2);
    if (surf_format == X265HIP_SURF_I32) return recs * X265HIP_SURF_GROUP_BYTES_I32;
    if (surf_format == X265HIP_SURF_PACKED_B) return ((recs + 63) >> 6) * X265HIP_SURF_BLOCK_BYTES_PACKED;
    return recs * X265HIP_SURF_GROUP_BYTES_PACKED;
}
int x265hip_me_fullsearch(const x265hip_me_params* p, void* stream);
int x265hip_me_best_reset(uint64_t* best, size_t count, void* stream);
/* Name of the kernel a minima-only launch (surf == NULL) of x265hip_me_fullsearch runs at this depth and range - what a profile of the caller lists as its
 * dominant kernel (bench.py prints it as roofline.kernel); derived from the same switches and the same LDS-pitch function as the launch itself. */
const char* x265hip_me_minima_kernel_name(int depth, int range);
/* TEST-ONLY: the X265HIP_ME_* A/B switches are read once per process; parity tests and soaks that flip one between launches call this to have them read again
 * (never while launches are in flight on other threads) */
void x265hip_me_env_refresh(void);

/* Sub-pel refinement of every PU's integer motion vector (the caller loop of SURVEY section 8(f) item 1,
 * reference MotionEstimate::motionEstimate, motion.cpp:1448-1561 + subpelCompare :1571-1664, luma):
 * square1 half-pel then quarter-pel iterations per the SubpelWorkload row `subme` (motion.cpp:48-58),
 * candidates measured with luma_hpp / luma_vpp / luma_hvpp + sad / satd, strict '<' updates.
 *   best_in : uint64 [ctu][85] from x265hip_me_fullsearch (cost << 32 | raster mv index)
 *   cost_q  : uint16 cost of a quarter-pel mv component, indexed by q + qoff (q in [-4*range-8, 4*range+8])
 *   out     : per PU { int32 cost; int16 qmvx; int16 qmvy }, [ctu][85]
 * fref needs >= range + 12 valid pixels of margin (8-tap apron around the +-1.5 pixel drift). */
typedef struct x265hip_subpel_params
{
    int depth;
    int width, height;
    int range;
    int subme;                      /* 0..7, row of the reference's workload[] table */
    const void* fenc;  intptr_t fenc_stride;
    const void* fref;  intptr_t fref_stride;
    const uint64_t* best_in;
    const uint16_t* cost_q;  int qoff;
    void* out;
    /* optional: the 15 luma phase planes of the reference picture (x265hip_phase_planes) - phase_planes = sample (0,0) of phase 1,
     * plane p at + (p - 1) * phase_plane_samples.  The candidates are then read from the planes instead of being interpolated per
     * candidate (same samples, same result); NULL = interpolate */
    const void* phase_planes;  intptr_t phase_plane_samples;
} x265hip_subpel_params;
int x265hip_subpel_refine(const x265hip_subpel_params* p, void* stream);

/* Fused inter prediction + residual coding round trip of every NxN block (N = 8 << level, level 0..2), the
 * caller sequence of SURVEY section 8(f) item 2: Predict::predInterLumaPixel (predict.cpp:245-265),
 * calcresidual, Quant::transformNxN without RDOQ (quant.cpp:397-480, flat scaling lists; sign-bit hiding optional),
 * Quant::invtransformNxN incl. the DC-only shortcut (quant.cpp:543-605), add_ps / copy_pp, sse_pp.
 *   mv      : int32 [ctu*85][2] from x265hip_subpel_refine ({cost, qmvx | qmvy << 16}); block z of the level uses
 *             entry base(level) + z (base = 0, 64, 80)
 *   qp      : scaled luma QP (per = qp / 6, rem = qp % 6)
 *   intra_slice : flag bits - X265HIP_TU_INTRA_SLICE selects the I-slice rounding offset (171 instead of 85, quant.cpp:466),
 *             X265HIP_TU_SIGN_HIDE runs Quant::signBitHidingHDQ after the quantiser (quant.cpp:247-395, 471-476: the x265 default,
 *             pps.bSignHideEnabled; up-right diagonal scan for inter TUs, the mode-dependent scan for 4x4 / luma 8x8 intra TUs).
 *             Scaling lists and the denoiser come in through `tables`; RDOQ (the host's row a9) is not part of the fused stages.
 *   recon   : reconstructed luma plane, same geometry as fenc (margins are not written)
 *   levels  : int16 [ctu][blocks][N*N] quantised coefficients   num_sig : uint32 [ctu][blocks]
 *   dist    : uint64 [ctu][blocks] sse_pp(fenc, recon) */
#define X265HIP_TU_INTRA_SLICE 1
#define X265HIP_TU_SIGN_HIDE   2
/* Optional per-coefficient tables of a TU stage launch (DEVICE pointers, n * n entries of the launch's transform size in raster order;
 * any of them NULL = not used): the scaling list's quantiser / dequantiser coefficients the host selected for (size, list type, qp % 6)
 * - ScalingList::m_quantCoef / m_dequantCoef, used by quant.cpp:463 and dequant_scaling (dct.cpp:612-662, quant.cpp:562-567) - and the
 * denoiser's offsets with its running residual sums for the TU category (primitives.denoiseDct before the quantiser, quant.cpp:444-451,
 * dct.cpp:744-755: nr_residual_sum[i] += |coef|, atomically).  A NULL `tables` pointer = flat lists, no denoising (the x265 defaults). */
typedef struct x265hip_tu_tables
{
    const int32_t* quant_coeff;
    const int32_t* dequant_coeff;
    const uint16_t* nr_offset;
    uint32_t* nr_residual_sum;
    /* capture for a host-side RDOQ pass (Quant::rdoQuant, quant.cpp:627+, stays on the host - SURVEY row a9 - but can run on
     * device-produced transforms): the coefficients transformNxN hands to the quantiser (m_resiDctCoeff, after the denoiser when it is
     * on) and the quantiser's deltaU (dct.cpp:679), both laid out like `levels`; NULL = not wanted */
    int16_t* dct_coeff_out;
    int32_t* delta_u_out;
    /* The data-parallel half of Quant::rdoQuant (quant.cpp:609+; every preset from `slow` up runs rdoqLevel 2, param.cpp:489-553), produced
     * next to the transform instead of per coefficient group on the host (round 3; all optional, NULL = not wanted):
     *   rdoq_levels / rdoq_num_sig : primitives.nquant (dct.cpp:688-713: rounding 1/2, ABSOLUTE levels) - what rdoQuant starts from
     *                                (quant.cpp:626); laid out like `levels` / `num_sig`
     *   rdoq_cost_uncoded          : int64 per coefficient, laid out like `levels`: what the pre-pass slots cu[].nonPsyRdoQuant (psy_scale 0)
     *                                or cu[].psyRdoQuant = psyRdoQuant_1p + psyRdoQuant_2p (dct.cpp:986-1069) write into costUncoded[] for
     *                                every 4x4 coefficient group of the block
     *   rdoq_cg_cost               : int64 [blocks][n * n / 16][2]: what a slot call adds to *totalUncodedCost and *totalRdCost for coefficient
     *                                group g = (blkPos / (4 n)) * (n / 4) + (blkPos % n) / 4 (raster order of the groups): [0] = sum of
     *                                coef^2 << scaleBits (nonPsyRdoQuant, psyRdoQuant_1p), [1] = sum of the finished costUncoded (psyRdoQuant,
     *                                psyRdoQuant_2p; = [0] without psy).  A host without AVX-512 calls _1p AND _2p on the same totals
     *                                (quant.cpp:716-717, 803-808), i.e. adds [0] + [1]
     *   psy_scale                  : Quant::m_psyRdoqScale * lambda (quant.cpp:634); != 0 selects the psy pre-pass, which needs the SOURCE
     *                                block's transform (m_fencDctCoeff, quant.cpp:436-441: copy_ps + dct of fenc - always the DCT)
     *   fenc_dct_out               : that transform, laid out like `levels`
     * The CABAC-coupled rest of rdoQuant (bit costs from the entropy coder's context state, the serial level decisions) stays host work. */
    int64_t* rdoq_cost_uncoded;
    int64_t* rdoq_cg_cost;
    int16_t* rdoq_levels;
    uint32_t* rdoq_num_sig;
    int16_t* fenc_dct_out;
    int64_t psy_scale;
} x265hip_tu_tables;
typedef struct x265hip_recon_params
{
    int depth;
    int width, height;
    int level;
    int qp, intra_slice;
    const void* fenc;  intptr_t fenc_stride;
    const void* fref;  intptr_t fref_stride;
    void* recon;       intptr_t recon_stride;
    const void* mv;
    int16_t* levels; uint32_t* num_sig; uint64_t* dist;
    const x265hip_tu_tables* tables;            /* HOST pointer to the table record, or NULL */
} x265hip_recon_params;
int x265hip_inter_recon(const x265hip_recon_params* p, void* stream);
/* One chroma plane of the same stage for 4:2:0 pictures (Predict::predInterChromaPixel, predict.cpp:304-351, + the same residual
 * round trip on (n/2) x (n/2) blocks; DCT also for 4x4): fenc / fref / recon = sample (0,0) of the chroma planes with their
 * strides, width / height = LUMA size, mv = the luma stage's records, qp = the plane's quantiser QP (chroma QP mapping and PPS /
 * slice offsets applied by the caller, + QP_BD_OFFSET); levels hold (n/2)^2 entries per block. */
int x265hip_inter_recon_chroma(const x265hip_recon_params* p, void* stream);
/* Cb and Cr of one picture in ONE launch: same geometry, bit depth, block size and use of `tables`; each record carries its own planes,
 * QP and outputs.  Results are those of two x265hip_inter_recon_chroma calls. */
int x265hip_inter_recon_chroma_pair(const x265hip_recon_params* cb, const x265hip_recon_params* cr, void* stream);
/* Bi-predictive flavour (B pictures - and P pictures with explicit weights -, luma): Predict::motionCompensation (predict.cpp:77-243).
 * base = the uni-directional parameters with fref / mv = list 0 (both references share fref_stride); dir = uint8 [ctu][blocks]: 1 =
 * list 0 only, 2 = list 1 only, 3 = predInterLumaShort of both lists combined by addAvg; NULL = all 3.
 * weight0 / weight1 (HOST pointers, read at the call): the luma WeightParam of the list-0 / list-1 reference, NULL = the list has no
 * table (pps.bUseWeightPred / bUseWeightedBiPred off).  A block predicted from one list whose table is `present` takes
 * predInterLumaShort + addWeightUni (weight_sp, predict.cpp:525-545); a block predicted from both takes addWeightBi (:411-456) when
 * both tables exist and at least one is present - with list 0's denominator for both, as the reference does - and addAvg otherwise. */
typedef struct x265hip_pred_weight
{
    int present;          /* WeightParam::wtPresent of the reference picture's LUMA entry - also for a chroma plane: the reference decides
                           * weighted prediction for all three planes on the luma entry (pwp->wtPresent, predict.cpp:94, :187, :196) and
                           * then takes weight / offset / denominator from the plane's own entry */
    int weight;           /* inputWeight */
    int offset;           /* inputOffset (8-bit domain; scaled by 1 << (depth - 8) inside) */
    int log2_denom;       /* log2WeightDenom */
} x265hip_pred_weight;
typedef struct x265hip_recon_bi_params
{
    x265hip_recon_params base;
    const void* fref1;
    const int32_t* mv1;
    const uint8_t* dir;
    const x265hip_pred_weight* weight0;
    const x265hip_pred_weight* weight1;
} x265hip_recon_bi_params;
int x265hip_inter_recon_bi(const x265hip_recon_bi_params* p, void* stream);
/* One chroma plane of a 4:2:0 picture through the same stage - predInterChromaPixel / predInterChromaShort (predict.cpp:304-409), the
 * plane's own weights (WeightParam of Cb or Cr) - with the conventions of x265hip_inter_recon_chroma: planes and strides of the chroma
 * plane, width / height = LUMA size, mv / mv1 = the luma stage's records, (n/2)^2 levels per block. */
int x265hip_inter_recon_chroma_bi(const x265hip_recon_bi_params* p, void* stream);

/* Picture border extension (reference extendPicBorder, pixel.cpp:1027-1041 = extendRowBorder slot,
 * ipfilter.cpp:59-77, + top/bottom row replication): `pic` points at pixel (0,0) of a plane that has
 * margin_x columns / margin_y rows of padding on every side. */
int x265hip_extend_border(void* pic, intptr_t stride, int width, int height, int margin_x, int margin_y, int depth, void* stream);
/* round 6: the planes of ONE picture (Y, Cb, Cr - each with its own geometry) in one launch: the three back-to-back launches of a 4:2:0 picture were 15 us of
 * the step for 2 MB of copies (PicYuv's planes after FrameFilter finished the picture, framefilter.cpp:346-436 / picyuv.cpp). */
typedef struct x265hip_border_plane { void* pic; intptr_t stride; int width, height, margin_x, margin_top, margin_bottom; } x265hip_border_plane;   /* a whole picture: margin_top = margin_bottom = its margin; a band of rows: as x265hip_extend_border_rows (either may be 0) */
int x265hip_extend_border_planes(const x265hip_border_plane* planes, int nplanes, int depth, void* stream);
/* the row-wise form FrameFilter uses (framefilter.cpp:346-436): `band` = first sample of a band of `height` rows; left / right margins of
 * those rows, margin_top rows above (first band of a picture) and margin_bottom rows below (last band), either may be 0 */
int x265hip_extend_border_rows(void* band, intptr_t stride, int width, int height, int margin_x, int margin_top, int margin_bottom,
                               int depth, void* stream);

/* ---- motion search drivers: MotionEstimate::motionEstimate for a list of PUs ----
 * One job = one prediction unit of one reference picture: position, size (any of the reference's 24 inter partitions,
 * primitives.h:41-55 minus 4x4), quarter-pel predictor.  For every job the kernels reproduce motionEstimate() with the optional
 * extra candidates mvc[] (encoder/motion.cpp:739-1561; UMH also sizes its range from them, :982-1040): predictor / zero start, the integer pattern `method` (X265_DIA_SEARCH,
 * X265_HEX_SEARCH, X265_UMH_SEARCH, X265_STAR_SEARCH, X265_SEA, X265_FULL_SEARCH of x265.h:492-497; SEA needs `integral` and
 * refuses, with out_cost = -1, the four PU sizes whose DC terms the reference reads from outside the PU: 8x4, 4x8, 32x8, 8x32), the predictor-vs-search
 * choice and the sub-pel refinement level `subme`; out_qmv / out_cost are its outQMv and return value.
 *   fenc, fref : pixel (0,0) of the padded source / reference luma planes; fref needs mvmax + 8 valid pixels of margin
 *   cost_q     : uint16 bit cost of a quarter-pel mv DIFFERENCE component, cost_q[qoff + d] (BitCost::s_costs, built on the
 *                host, bitcost.cpp:40-58); d ranges over mv - predictor and, for one STAR raster candidate, 8 x mv - predictor
 *   mvmin/mvmax: integer-pel search bounds applied to every job */
enum { X265HIP_ME_DIA = 0, X265HIP_ME_HEX = 1, X265HIP_ME_UMH = 2, X265HIP_ME_STAR = 3, X265HIP_ME_SEA = 4, X265HIP_ME_FULL = 5 };
typedef struct x265hip_me_search_job
{
    int32_t px, py, w, h;
    int32_t qmvpx, qmvpy;
    int32_t out_qmvx, out_qmvy, out_cost;
} x265hip_me_search_job;
typedef struct x265hip_me_search_params
{
    int depth;
    const void* fenc;  intptr_t fenc_stride;
    const void* fref;  intptr_t fref_stride;
    int method, subme, merange;
    const uint16_t* cost_q;  int qoff;
    int mvmin_x, mvmin_y, mvmax_x, mvmax_y;
    x265hip_me_search_job* jobs;  int njobs;      /* DEVICE array, results written in place */
    const int32_t* mvc;                           /* optional DEVICE int32 [njobs][12][2]: motionEstimate's extra quarter-pel */
    const int32_t* num_mvc;                       /* candidates mvc[] (at most 12, search.cpp:2094) and their count per job */
    /* X265_SEA only: the reference picture's twelve block-sum planes (x265hip_sea_integral), DEVICE uint32 pointers to the entry of
     * sample (0,0), stride = fref_stride; only the planes the job sizes select are read (motion.cpp:1315-1347) */
    const uint32_t* integral[12];
} x265hip_me_search_params;
int x265hip_me_search(const x265hip_me_search_params* p, void* stream);

/* x265hip_sea_integral = the integral section of FrameFilter::processPostRow (encoder/framefilter.cpp:716-823) with the
 *   integral_init*h / *v primitives (:39-140) for one reference picture: planes[k](x, y) = sum of the bw x bh block of samples whose
 *   top-left corner is (x, y), (bw, bh) = 32x32, 32x24, 32x8, 24x32, 16x16, 16x12, 16x4, 12x16, 8x32, 8x8, 4x16, 4x4 (framedata.h:171),
 *   for -margin_x <= x <= width + margin_x - bw and -margin_y <= y <= height + margin_y - bh (the reference leaves row -margin_y
 *   and the last rows unset; a search window must stay inside either way).  ref = sample (0,0) of the padded plane; planes[k] =
 *   DEVICE pointer to the entry of sample (0,0) in a uint32 plane of the same stride and margins, NULL = not wanted. */
typedef struct x265hip_sea_integral_params
{
    int depth;
    const void* ref; intptr_t stride;
    int width, height, margin_x, margin_y;
    uint32_t* planes[12];
} x265hip_sea_integral_params;
int x265hip_sea_integral(const x265hip_sea_integral_params* p, void* stream);

/* Lookahead frame cost estimate - CostEstimateGroup::estimateFrameCost + estimateCUCost (no HME, no weighted reference;
 * encoder/slicetype.cpp:3115-3213,3216-3388) for P pictures (one list, intra competes) and B pictures (two lists, skip shortcut
 * :3311-3315, the two bi-directional candidates :3322-3343, score scaling :3203-3204): every 8x8 block of the half-resolution picture, in the
 * reference's reverse raster order, tries the final mvs of its right / lower neighbours as predictors (SATD at lowresMC,
 * :3284-3301), runs MotionEstimate::motionEstimate in its lowres flavour (HEX, merange 16, subpelRefine 1, quarter-pel positions
 * from the four half-pel phase planes: motion.cpp:775,1471-1503, lowres.h:66-121), adds lowresPenalty and lets the intra cost win
 * (:3346-3356).  The neighbour dependency makes a picture a wavefront of rows, each two blocks behind the row below it; one
 * workgroup walks it (a DPP quad per row); independent pictures (frame-parallel encoding has them) fill the chip.
 *   cur: pixel (0,0) of the current picture's plane 0; ref[0..3]: pixel (0,0) of the reference's planes (lowres_init's outputs);
 *   intra_cost: x265hip_lowres_intra's output for the current picture; inv_qscale: optional (fenc->invQscaleFactor) or NULL.
 * Outputs: mvs int32 [n][2] quarter-pel (lowresMvs), mv_costs int32 [n] (lowresMvCosts), lowres_costs uint16 [n]
 * (cost | listused << 14), row_satds int32 [height_in_cu], frame int64 [4] = { costEst, costEstAq, intraMbs, score }.
 * All pairs of one call are P or all are B.
 * A call takes `npairs` independent (current, reference) pairs of one geometry - `pairs` is a HOST array, copied to the device in
 * stream order - and runs one workgroup per pair.
 * Limit: width_in_cu <= 2 * min(256, height_in_cu rounded up to 16), i.e. any picture up to 8192 x 8192. */
typedef struct x265hip_lowres_cost_pair
{
    const void* cur;
    const void* ref[4];                             /* list 0 */
    const void* ref1[4];                            /* list 1: a B picture (b < p1); all NULL for a P picture */
    const int32_t* intra_cost;
    const int32_t* inv_qscale;
    int32_t* mvs;   int32_t* mv_costs;              /* list 0 */
    int32_t* mvs1;  int32_t* mv_costs1;             /* list 1 */
    int32_t do_search[2];                           /* estimateFrameCost's bDoSearch: 0 keeps the list's given mvs / mv_costs */
    uint16_t* lowres_costs;  int32_t* row_satds;
    int64_t* frame;                                 /* [4] = costEst (unscaled sum), costEstAq, intraMbs, score */
    const void* ref_bi[4];                          /* --weightp on a B picture: ref[] = the WEIGHTED list-0 planes (search, predictor
                                                       candidates, skip cost: slicetype.cpp:3222,3267), ref_bi[] = the unweighted ones,
                                                       which the two bi-directional candidates keep (:3328); all NULL = ref[] */
} x265hip_lowres_cost_pair;
typedef struct x265hip_lowres_cost_params
{
    int depth;
    intptr_t stride;
    int width_in_cu, height_in_cu;
    const uint16_t* cost_q;  int qoff;
    int bframe_bias;                                /* param->bFrameBias: B score = costEst * 100 / (130 + bias) */
    const x265hip_lowres_cost_pair* pairs;  int npairs;
    int pairs_on_device;                            /* 0: `pairs` is host memory (copied in stream order, may block the caller);
                                                       1 / 2: `pairs` already is a device array of P (1) / B (2) pictures (no allocation, no copy, no validation);
                                                       | 4 (round 6): none of those pairs needs a search (every do_search is 0) - the dependency-free launch may be used
                                                       | 8, | 16 (round 6): every pair searches list 0 / list 1 (do_search[0] / [1]) - a split B estimate may walk the two
                                                       lists side by side; without them a device table of B pictures takes the one-walk form */
} x265hip_lowres_cost_params;
/* One workgroup walks a picture (a wavefront of dependent block rows); a call of up to four pictures of 32 or more block rows - the
 * latency case: a host thread waits for one estimate - gives every picture several workgroups, one per band of block rows, the
 * boundary mvs handed upward through L2 (same integers; 4K: 8.9 -> 5.2 ms per estimate).  Uses a few KB of per-stream scratch. */
int x265hip_lowres_cost(const x265hip_lowres_cost_params* p, void* stream);
/* Round 6: a call none of whose pairs needs a search (do_search 0 / 0: both lists were searched by earlier estimates - a third of the triples of the slice-type decision) has no
 * dependency between blocks and runs as ONE flat launch (a wavefront per block row) instead of the W + 2 H lock-steps.  Diagnostic: launches so far as
 * { flat, one-workgroup walk, split walk } (process-wide).
 * A split B estimate that searches (round 6, second half): its lists are independent of each other, so each searched list is walked by its own bands (one HEX search per
 * lock-step instead of two searches + the bi-directional candidates) and the same flat launch finishes the estimate.  X265HIP_LOWRES_COST_SO_OFF=1 = the one-walk form. */
void x265hip_lowres_cost_launch_counts(uint64_t out[3]);
/* X265HIP_LOWRES_COST_SPLIT=<bands> / X265HIP_LOWRES_COST_SO_OFF=1 are read when the library is loaded; this reads them again (tests, A/B tools). */
void x265hip_lowres_cost_env_refresh(void);

/* The same estimate behind host pointers, shaped like the loop it replaces (csrc/lookahead_host.hip): ONE call = the estimateCUCost
 * loop of CostEstimateGroup::estimateFrameCost (slicetype.cpp:3178-3196 over :3216-3388) for one (p0, b, p1) triple whose Lowres
 * planes and per-block arrays live in HOST memory; synchronous, re-entrant (per-thread stream + device scratch).
 *   cur, ref[], ref1[], ref_bi[] : sample (0,0) of host planes that start margin_y rows / margin_x samples earlier and hold
 *                 lines + 2 * margin_y rows of `stride` samples (Lowres::create, lowres.cpp:50-72); ref1 all NULL = P picture;
 *                 ref_bi: the unweighted list-0 planes when ref[] is weighted on a B picture (see the pair structure above), else all NULL
 *   cost_q      : the CENTRE of the mv-difference cost table (BitCost::m_cost, bitcost.h:45), valid for [-cost_q_half, cost_q_half]
 *   mvs / mv_costs : lowresMvs[l][dist] (int32 [n][2], 8-byte aligned) / lowresMvCosts[l][dist]: outputs for do_search[l] != 0,
 *                 inputs otherwise (a list searched earlier is reused, slicetype.cpp:3126-3127, 3256-3260)
 *   lowres_costs uint16 [n], row_satds int32 [height_in_cu], frame int64 [4] = { costEst sum, costEstAq sum, intraMbs, score } */
typedef struct x265hip_lowres_cost_host_params
{
    int depth;
    intptr_t stride;
    int width_in_cu, height_in_cu;
    int lines, margin_x, margin_y;
    const void* cur;
    const void* ref[4];
    const void* ref1[4];
    const void* ref_bi[4];
    const int32_t* intra_cost;
    const int32_t* inv_qscale;                      /* optional */
    const uint16_t* cost_q;  int cost_q_half;
    int bframe_bias;
    int do_search[2];
    int32_t* mvs[2];  int32_t* mv_costs[2];
    uint16_t* lowres_costs;  int32_t* row_satds;  int64_t* frame;
    /* optional content keys (0 = upload on every call): a non-zero key vouches that the plane set at these host addresses does not
     * change while the key stays the same, so the library keeps ONE device copy across the many triples that involve the picture.
     * The key must name the CONTENT for as long as the process lives: a frame number alone repeats in the next encode of the same
     * process (and the allocator hands out the same addresses again) - combine it with an encoder-instance number, e.g.
     * instance << 32 | (Lowres::frameNum + 1), or call x265hip_lowres_planes_forget() when an encoder closes.
     * cur / ref / ref1 / ref_bi planes respectively; weighted scratch planes must pass 0.
     * Round 6: plane_key_cur != 0 also vouches for the picture's vector arrays: while the key stays the same, mvs[l] / mv_costs[l] at these addresses are written only by
     * this function's own searches, or before the first call that reuses them (lowresMvs / lowresMvCosts: searched once per lifetime of a Lowres, reset by Lowres::init,
     * lowres.cpp:283-284).  The library then keeps the device copy a search left behind (or uploads a reused list once) and the estimates that reuse the list -
     * three quarters of the slice-type decision's - upload no vectors at all; whatever a call does transfer goes as ONE upload and ONE download through pinned staging
     * (4K: 1.8 -> 0.3 - 0.5 ms per reusing estimate).  X265HIP_LA_RESIDENT_OFF=1 (read when the library loads) uploads reused lists every time, as before. */
    uint64_t plane_key_cur, plane_key_ref, plane_key_ref1, plane_key_ref_bi;
} x265hip_lowres_cost_host_params;
int x265hip_lowres_cost_host(const x265hip_lowres_cost_host_params* p);
/* the intra half the same way: LookaheadTLD::lowresIntraEstimate's per-block work (slicetype.cpp:696-772) for one picture whose lowres
 * plane 0 is in host memory (geometry and plane_key as above); intra_cost int32 [n], intra_mode uint8 [n], lowres_costs uint16 [n] are
 * HOST outputs; the AQ weighting and the row / frame sums of :779-803 stay with the caller. */
typedef struct x265hip_lowres_intra_host_params
{
    int depth;
    intptr_t stride;
    int width_in_cu, height_in_cu;
    int lines, margin_x, margin_y;
    const void* plane;
    int intra_penalty;
    int32_t* intra_cost; uint8_t* intra_mode; uint16_t* lowres_costs;
    uint64_t plane_key;
} x265hip_lowres_intra_host_params;
int x265hip_lowres_intra_host(const x265hip_lowres_intra_host_params* p);
/* drops every device copy the two host entries above keep under plane keys (call between encodes, with no estimate in flight) */
void x265hip_lowres_planes_forget(void);
/* LookaheadTLD::calcAdaptiveQuantFrame (encoder/slicetype.cpp:444-694, called once per source picture from PreLookaheadGroup::processTasks,
 * :1395) behind host pointers, for the AQ modes x265hip_aq_offsets covers (1 - 3, strength > 0; 4:2:0 or 4:0:0): ONE call uploads the
 * source picture's planes (the blocks' footprint: width / height rounded up to qg_size, which lies inside PicYuv's padding as it does
 * for the reference's own loop), runs x265hip_aq_energy, downloads the block energies and the six totals, and finishes on the calling
 * thread with x265hip_aq_offsets - outputs straight into the caller's Lowres arrays:
 *   qp_aq_offset, qp_cutree_offset : HOST double [blocks]  (Lowres::qpAqOffset / qpCuTreeOffset, both the same values, :617-618)
 *   inv_qscale                     : HOST int32  [blocks]  (Lowres::invQscaleFactor = x265_exp2fix8, :619)
 *   inv_qscale_8x8                 : HOST int32  [width_in_cu * height_in_cu] or NULL; qg_size 8 only: the 2x2 averages of :626-640
 *   energy                         : HOST uint32 [blocks] or NULL (what acEnergyCu returned per block; Lowres::blockVariance under --fades)
 *   wp_sum / wp_ssd                : HOST uint64 [3]: Lowres::wp_sum, and wp_ssd - raw totals, or with normalise_wp != 0 the
 *                                    ssd - (sum^2 + n / 2) / n of :662-675 (n = the plane's ((dim + 8) >> 4) << 4 area) that
 *                                    --weightp / --weightb ask for.
 * blocks = ceil(width / qg_size) * ceil(height / qg_size), row-major.  Stateless and re-entrant like x265hip_lowres_cost_host (a stream
 * and grow-only scratch per calling thread).  quantOffsets, --hdr10-opt, --hevc-aq, the edge mode and the 2-pass cuTree reuse are the
 * caller's to keep on its own loop. */
typedef struct x265hip_aq_frame_host_params
{
    int depth;
    const void* y; const void* cb; const void* cr;          /* HOST: sample (0,0) of the picture's planes (PicYuv::m_picOrg[]); cb = cr = NULL: 4:0:0 */
    intptr_t stride, stride_c;                              /* in samples */
    int width, height, qg_size, aq_mode;
    double aq_strength;
    int width_in_cu, height_in_cu;                          /* the lowres 8x8 grid (inv_qscale_8x8 only) */
    int normalise_wp;
    double* qp_aq_offset; double* qp_cutree_offset; int32_t* inv_qscale; int32_t* inv_qscale_8x8; uint32_t* energy;
    uint64_t* wp_sum; uint64_t* wp_ssd;
} x265hip_aq_frame_host_params;
int x265hip_aq_frame_host(const x265hip_aq_frame_host_params* p);
/* weightAnalyse (encoder/weightPrediction.cpp:222-497; once per P / B slice from FrameEncoder::compressFrame when --weightp / --weightb
 * are on) behind host pointers, 4:2:0: per list the float guess from the pictures' wp_ssd / wp_sum, and - unless the early exits take the
 * plane - a motion-compensated copy of the reference plane from the lookahead's lowres vectors (mcLuma :59-92 through Lowres::lowresMC,
 * mcChroma :96-166 with the 4-tap filters), then weightCost (:172-217: weight_pp + the 8x8 SATD sum, luma blocks capped by the intra
 * cost) for the unweighted plane and EVERY (scale, offset) pair the reference's scan could visit (<= 9 x 5) in one launch; the scan's
 * own order and early break, sliceHeaderCost, the denominator reduction and the 0.998 acceptance are replayed on the calling thread from
 * the downloaded scores.  Same weights as the reference's loop, which spends 2 x (weight_pp + SATD) of a whole plane per pair visited.
 *   lowres / ref[].lowres[4] : HOST, sample (0,0) of lowres planes (Lowres::lowresPlane) of one geometry: lowres_stride, lowres_width x
 *                              lowres_lines (multiples of 8), margins as allocated (the compensated blocks reach 8 + 1 samples outside)
 *   cb / cr, ref[].cb / cr   : HOST, sample (0,0) of the SOURCE pictures' chroma planes (PicYuv::m_picOrg[1 / 2]); the references' with
 *                              their borders extended (weightPrediction.cpp:333-343), margin_xc / margin_yc >= 16 samples of it are read
 *   ref[].mvs                : HOST int32 [lowres blocks][2] (Lowres::lowresMvs[list][distance]) when the lookahead searched that
 *                              distance (distance <= bframes + 1 and mvs[0].x != 0x7FFF), else NULL: no compensation
 *   intra_cost               : HOST int32 [lowres blocks] (Lowres::intraCost)
 *   wp_ssd / wp_sum          : the pictures' Lowres::wp_ssd / wp_sum (x265hip_aq_frame_host fills them)
 *   plane_key / ref[].plane_key : as x265hip_lowres_cost_host's keys (0 = upload on every call): the lowres planes of a picture keep one
 *                              device copy across calls, shared with the frame cost estimates
 *   weights                  : HOST int32 [2][3][4] out = { wtPresent, inputWeight, log2WeightDenom, inputOffset } of reference 0 per
 *                              (list, plane); denoms: HOST int32 [2][2] out = lumaDenom, chromaDenom after each list (:468-474: what the
 *                              list's other references are reset to) */
typedef struct x265hip_weight_analyse_ref
{
    const void* lowres[4];
    const void* cb; const void* cr;
    const int32_t* mvs;
    uint64_t wp_ssd[3], wp_sum[3];
    uint64_t plane_key;
} x265hip_weight_analyse_ref;
typedef struct x265hip_weight_analyse_host_params
{
    int depth;
    const void* lowres; intptr_t lowres_stride;
    int lowres_width, lowres_lines, lowres_margin_x, lowres_margin_y;
    const void* cb; const void* cr; intptr_t stride_c;
    int margin_xc, margin_yc;
    int pic_width, pic_height;
    const int32_t* intra_cost;
    uint64_t wp_ssd[3], wp_sum[3];
    uint64_t plane_key;
    int nlists;
    x265hip_weight_analyse_ref ref[2];
    int32_t* weights; int32_t* denoms;
} x265hip_weight_analyse_host_params;
int x265hip_weight_analyse_host(const x265hip_weight_analyse_host_params* p);

/* ---- in-loop deblocking of a device-resident luma reconstruction (deblocking half) ----
 * x265hip_deblock_bs_inter = Deblock::getBoundaryStrength (deblock.cpp:191-215) for a P picture with one reference cut into
 *   square inter blocks of 8 << level samples: inputs are the sub-pel stage's mv array (int32 [ctu*85][2], z-order) and the
 *   reconstruction stage's num_sig (uint32 [ctu][blocks]); outputs bs_ver uint8 [height/4][width/8] (4-row unit u of the
 *   vertical edge at x = 8 * ex) and bs_hor uint8 [height/8][width/4].  Picture borders get Bs 0.
 * x265hip_deblock_luma = Deblock::edgeFilterLuma (deblock.cpp:317-415) over the whole picture, vertical edges then horizontal
 *   edges, in place: rec = pixel (0,0).  Any Bs maps may be supplied (Bs 2 = intra edges).  qp_map (int8 [height/8][width/8])
 *   overrides the uniform qp; the offsets are the PPS's deblockingFilter{Beta,Tc}OffsetDiv2. */
typedef struct x265hip_deblock_bs_params
{
    int width, height, level;
    const int32_t* mv; const uint32_t* num_sig;
    uint8_t* bs_ver; uint8_t* bs_hor;
    const uint8_t* intra;          /* optional uint8 [ctu][blocks]: non-zero = intra CU, its edges get Bs 2 (deblock.cpp:198-199) */
    /* Several reference pictures / B pictures (deblock.cpp:217-247); all zero / NULL = one list-0 picture for every block.
     * ref0 / ref1: optional int8 [ctu][blocks], the reference PICTURE id of the block's list-0 / list-1 prediction (equal ids =
     * the same picture, whichever list it came from; -1 = list unused, its mv reads as zero).  mv1: list-1 records laid out like
     * mv.  slice_b != 0 selects the B-picture comparison of (ref0, ref1) x (mv, mv1). */
    int slice_b;
    const int32_t* mv1; const int8_t* ref0; const int8_t* ref1;
} x265hip_deblock_bs_params;
int x265hip_deblock_bs_inter(const x265hip_deblock_bs_params* p, void* stream);
typedef struct x265hip_deblock_params
{
    int depth;
    void* rec; intptr_t stride;
    int width, height;
    const uint8_t* bs_ver; const uint8_t* bs_hor;
    int qp; const int8_t* qp_map;
    int beta_offset_div2, tc_offset_div2;
} x265hip_deblock_params;
int x265hip_deblock_luma(const x265hip_deblock_params* p, void* stream);
/* x265hip_deblock_chroma = Deblock::edgeFilterChroma (deblock.cpp:417-497) + pelFilterChroma (loopfilter.cpp:160-180) for the Cb / Cr
 *   planes of a 4:2:0 picture, in place: only Bs 2 units on the 8-sample chroma grid (luma multiples of 16) are filtered; tc from the
 *   mean QP + the plane's PPS offset through the chroma QP mapping (constants.cpp:346-350).  cb / cr = sample (0,0), stride in
 *   samples; width / height = LUMA size (multiples of 16); bs maps / qp_map as above; qp = the CUs' m_qp (0..51). */
typedef struct x265hip_deblock_chroma_params
{
    int depth;
    void* cb; void* cr; intptr_t stride;
    int width, height;
    const uint8_t* bs_ver; const uint8_t* bs_hor;
    int qp; const int8_t* qp_map;
    int cb_qp_offset, cr_qp_offset, tc_offset_div2;
} x265hip_deblock_chroma_params;
int x265hip_deblock_chroma(const x265hip_deblock_chroma_params* p, void* stream);

/* ---- lookahead picture preparation and intra cost estimate (the intra half) ----
 * x265hip_lowres_init = Lowres::init's pixel work (lowres.cpp:294-306): frameInitLowres (pixel.cpp:604-629) into the four
 *   half-resolution planes (full-pel, H, V, HV phase) followed by extendPicBorder of each.  `src` = pixel (0,0) of the padded
 *   full-resolution luma plane (it is read up to 2 * lines + 1 rows / 2 * width + 1 columns, i.e. into its margin when the
 *   lowres size was rounded up to whole 8x8 blocks); plane[i] = pixel (0,0) of lowres plane i, all with `stride` and
 *   margin_x / margin_y pixels of padding.  width / lines: lowres size, multiples of 8.
 * x265hip_lowres_intra = LookaheadTLD::lowresIntraEstimate's per-block work (slicetype.cpp:696-772) on lowres plane 0:
 *   intra_cost int32 / intra_mode uint8 / lowres_costs uint16 per 8x8 block in raster order; intra_penalty =
 *   5 * (int)x265_lambda_tab[X265_LOOKAHEAD_QP] is computed by the host (constants.cpp double table). */
typedef struct x265hip_lowres_init_params
{
    int depth;
    const void* src; intptr_t src_stride;
    void* plane[4]; intptr_t stride;
    int width, lines;
    int margin_x, margin_y;
} x265hip_lowres_init_params;
int x265hip_lowres_init(const x265hip_lowres_init_params* p, void* stream);
typedef struct x265hip_lowres_intra_params
{
    int depth;
    const void* plane; intptr_t stride;
    int width_in_cu, height_in_cu;
    int intra_penalty;
    int32_t* intra_cost; uint8_t* intra_mode; uint16_t* lowres_costs;
} x265hip_lowres_intra_params;
int x265hip_lowres_intra(const x265hip_lowres_intra_params* p, void* stream);

/* Weighted-reference analysis of the lookahead - the pixel work of LookaheadTLD::weightsAnalyse (encoder/slicetype.cpp:860-957).
 * x265hip_lowres_weight_cost = LookaheadTLD::weightCostLuma (:807-841) for up to four candidate weights in one launch: the
 *   reference's lowres plane 0 weighted like primitives.weight_pp (pixel.cpp:518-543, 14-bit intermediate: round << (14 - depth),
 *   shift denom + (14 - depth), offset << (depth - 8)) - without materialising the weighted plane - then the 8x8 SATD against the
 *   current picture's plane, each block capped by its intra cost.  cand[i] = { present, scale, log2 denom, offset }; present = 0
 *   scores the unweighted reference (what weightsAnalyse measures first).  cost: DEVICE uint32 [ncand], overwritten.
 * x265hip_lowres_weight_apply = the weighting of the reference's four lowres buffers once a weight is accepted (:943-952):
 *   src[i] / dst[i] = ALLOCATION START of plane i, rows x stride samples each.
 * The float guess between the calls (variance ratio, means, the 0.998 acceptance test) is host logic: stages.py WeightAnalysis. */
typedef struct x265hip_lowres_weight_cost_params
{
    int depth;
    const void* fenc; const void* ref; intptr_t stride;      /* sample (0,0) of the two planes 0 */
    int width, lines;                                        /* the lowres picture (blocks may reach 7 samples into the padding) */
    const int32_t* intra_cost;                               /* DEVICE int32 [ceil(lines/8) * ceil(width/8)] */
    int ncand; int cand[4][4];
    uint32_t* cost;
} x265hip_lowres_weight_cost_params;
int x265hip_lowres_weight_cost(const x265hip_lowres_weight_cost_params* p, void* stream);
/* x265hip_aq_energy = the pixel work of LookaheadTLD::calcAdaptiveQuantFrame (encoder/slicetype.cpp:439-694): acEnergyCu (:256-275)
 *   for every qg_size x qg_size block of the source picture - cu[].var of the luma block plus, for 4:2:0, of the two half-size chroma
 *   blocks; energy = ssd - (sum^2 >> shift) per plane, added up - and the picture totals every acEnergyCu call accumulates into
 *   Lowres::wp_sum / wp_ssd.  The double-precision QP offsets of the AQ modes, x265_exp2fix8 and the final wp_ssd normalisation
 *   (:508-632, :662-675) are host logic: stages.AdaptiveQuant.
 *   y / cb / cr: sample (0,0) of padded planes (cb = cr = NULL: 4:0:0); blocks run over [0, width) x [0, height) in steps of qg_size
 *   (16 or 8) and may reach into the padding.  energy: DEVICE uint32 [blocks], row-major; wp: DEVICE uint64 [6] =
 *   { sum Y, Cb, Cr, ssd Y, Cb, Cr } raw totals, overwritten. */
typedef struct x265hip_aq_energy_params
{
    int depth;
    const void* y; const void* cb; const void* cr;
    intptr_t stride, stride_c;
    int width, height, qg_size;
    uint32_t* energy; uint64_t* wp;
} x265hip_aq_energy_params;
int x265hip_aq_energy(const x265hip_aq_energy_params* p, void* stream);
/* x265hip_aq_offsets - HOST-side companion (no device work, host pointers): the double-precision part of calcAdaptiveQuantFrame
 *   (slicetype.cpp:508-632) for AQ modes 1-3 from the block energies - pow(energy * c + 1, 0.1) with the frame-average correction
 *   (modes 2, 3; the bias term of mode 3), log2 (mode 1), the reference's float constants and summation order, through the C
 *   library's pow / log2 as the reference does - and invQscaleFactor = x265_exp2fix8(offset) (common.cpp:96-103).  aq_mode 0 or
 *   aq_strength 0: offsets 0, factors 256. */
typedef struct x265hip_aq_offsets_params
{
    int depth, qg_size, aq_mode;
    double aq_strength;
    int nblocks;
    const uint32_t* energy;          /* HOST uint32 [nblocks] */
    double* qp_aq_offset;            /* HOST double [nblocks] */
    int32_t* inv_qscale;             /* HOST int32 [nblocks] */
} x265hip_aq_offsets_params;
int x265hip_aq_offsets(const x265hip_aq_offsets_params* p);
/* --hevc-aq (rc.hevcAq): what calcAdaptiveQuantFrame runs instead of the AQ modes - LookaheadTLD::xPreanalyze / xPreanalyzeQp
 *   (encoder/slicetype.cpp:293-441, 507-511).  x265hip_aq_hevc_quadrants = its pixel work for ONE layer: per partition of part x part
 *   samples (part = maxCUSize >> d for the layers aqLayerDepth enables, lowres.h:123-142; clipped at the right / bottom edge) the sum
 *   and the sum of squares of the four quadrants, split at half the clipped size.  sums: DEVICE uint64 [partitions][4][2], partitions
 *   row-major, ceil(width / part) per row.  x265hip_aq_hevc_offsets - HOST-side companion (host pointers, no device work): activity =
 *   1 + the smallest quadrant variance (every quadrant divided by (cw / 2) * (ch / 2) as the reference does), the layer's average
 *   activity, dQpOffset = log2((s * act + avg) / (act + s * avg)) * 6 with s = 2^(qp_adaptation_range / 6), and - for the deepest
 *   enabled layer, whose partitions invQscaleFactor is indexed by - x265_exp2fix8(dQpOffset).  The wp_sum / wp_ssd statistics the
 *   reference gathers in the same loop are those of x265hip_aq_energy. */
typedef struct x265hip_aq_hevc_params
{
    int depth;
    const void* y; intptr_t stride;         /* sample (0,0) of the source luma plane */
    int width, height, part;
    uint64_t* sums;
} x265hip_aq_hevc_params;
int x265hip_aq_hevc_quadrants(const x265hip_aq_hevc_params* p, void* stream);
typedef struct x265hip_aq_hevc_offsets_params
{
    int width, height, part;
    double qp_adaptation_range;             /* rc.qpAdaptationRange, 1.0 .. 6.0 */
    const uint64_t* sums;                   /* HOST uint64 [partitions][4][2] */
    double* activity;                       /* HOST double [partitions]: dActivity */
    double* qp_offset;                      /* HOST double [partitions]: dQpOffset (= dCuTreeOffset) */
    double* avg_activity;                   /* HOST double [1] or NULL: dAvgActivity */
    int32_t* inv_qscale;                    /* HOST int32 [partitions] or NULL */
} x265hip_aq_hevc_offsets_params;
int x265hip_aq_hevc_offsets(const x265hip_aq_hevc_offsets_params* p);
/* x265hip_cutree_propagate = one cuTree propagation step, Lookahead::estimateCUPropagate (encoder/slicetype.cpp:2641-2753) with
 *   primitives.propagateCost (pixel.cpp:914-940): every 8x8 lowres block of picture b passes on
 *   (propagate_in + intra_cost * inv_qscale * fps_factor / 256) * (intra - min(intra, inter)) / intra   (double arithmetic, exactly as
 *   the reference's C: no fused multiply-add) to the blocks its mvs point at in the list-0 / list-1 reference - split bilinearly over
 *   four blocks, targets outside the picture dropped, weighted by bipred_weight / 64 when both lists are used - and the references'
 *   propagateCost accumulate with saturation at 65535 (saturating sums of non-negative terms do not depend on the order: the
 *   kernel adds into 64-bit counters and clamps once).  Amounts are assumed below 2^21, as in any real encode: beyond that the
 *   reference's own int32 products `listamount * weight` overflow (:2704-2716) and its result depends on the block order.
 *   All arrays DEVICE, one entry per 8x8 block, row-major: propagate_in uint16 (NULL: a non-referenced picture, zeros), intra_cost
 *   int32, lowres_costs uint16 (cost | lists used << 14), inv_qscale int32, mvs0 / mvs1 int32 [n][2] (x265hip_lowres_cost's outputs;
 *   mvs1 / ref_cost1 NULL for a P picture), ref_cost0 / ref_cost1 uint16, updated in place. */
typedef struct x265hip_cutree_propagate_params
{
    int width_in_cu, height_in_cu;
    const uint16_t* propagate_in; const int32_t* intra_cost; const uint16_t* lowres_costs; const int32_t* inv_qscale;
    const int32_t* mvs0; const int32_t* mvs1;
    double fps_factor;               /* CLIP_DURATION(fpsDenom / fpsNum) / CLIP_DURATION(averageDuration), slicetype.cpp:2654 */
    int bipred_weight;               /* 32, or 64 - (distScaleFactor >> 2) with --weightb (:2644-2646) */
    uint16_t* ref_cost0; uint16_t* ref_cost1;
} x265hip_cutree_propagate_params;
int x265hip_cutree_propagate(const x265hip_cutree_propagate_params* p, void* stream);
/* x265hip_cutree_finish - HOST-side companion of the propagation step (no device work, host pointers): Lookahead::cuTreeFinish
 *   (slicetype.cpp:2889-2937; quantisation groups of 16 or more, hevcAq off) - the propagated cost of every 8x8 lowres block becomes
 *   qp_cutree_offset = qp_aq_offset - strength * (log2(intra + propagate) - log2(intra) + weight_delta), intra = (intra_cost *
 *   inv_qscale + 128) >> 8, propagate = (propagate_cost * fps_factor_q8 + 128) >> 8; blocks whose scaled intra cost is 0 keep their
 *   value.  fps_factor_q8 = (int)(CLIP_DURATION(averageDuration) / CLIP_DURATION(fpsDenom / fpsNum) * 256); strength = 5.0 * (1.0 -
 *   qCompress) (:989); weight_delta = 1 - weightedCostDelta[ref0Distance - 1] when that is positive, else 0. */
typedef struct x265hip_cutree_finish_params
{
    int nblocks;
    const int32_t* intra_cost; const int32_t* inv_qscale; const uint16_t* propagate_cost; const double* qp_aq_offset;   /* HOST */
    int fps_factor_q8;
    double weight_delta, strength;
    double* qp_cutree_offset;                                                                                        /* HOST, in place */
} x265hip_cutree_finish_params;
int x265hip_cutree_finish(const x265hip_cutree_finish_params* p);
/* x265hip_frame_cost_recalculate - HOST-side: Lookahead::frameCostRecalculate (slicetype.cpp:2941-3011; P pictures, quantisation groups
 *   of 16 or more, hevcAq off) - the frame cost after cuTree changed the quantisers: every block's lowres cost (low 14 bits of
 *   lowres_costs) scaled by x265_exp2fix8(qp_cutree_offset), summed per row into row_satds and over the interior blocks into *score. */
typedef struct x265hip_frame_cost_recalculate_params
{
    int width_in_cu, height_in_cu;
    const uint16_t* lowres_costs; const double* qp_cutree_offset;      /* HOST */
    int32_t* row_satds; int64_t* score;                                /* HOST outputs */
} x265hip_frame_cost_recalculate_params;
int x265hip_frame_cost_recalculate(const x265hip_frame_cost_recalculate_params* p);
/* The --qg-size 8 branches of the two functions above (slicetype.cpp:2903-2921, 2990-3002): qp_aq_offset / qp_cutree_offset are the
 * FULL-RESOLUTION 8x8 grids (2 * width_in_cu per row, two by two per lowres block), intra_cost / propagate_cost / lowres_costs stay on
 * the lowres grid, inv_qscale is invQscaleFactor8x8 (the average calcAdaptiveQuantFrame leaves per lowres block); intra and propagated
 * costs enter at a quarter; the recalculation averages the block's four offsets. */
int x265hip_cutree_finish_qg8(const x265hip_cutree_finish_params* p, int width_in_cu, int height_in_cu);
/* cuTreeFinish with --hevc-aq (HOST-side): Lookahead::computeCUTreeQpOffset (slicetype.cpp:2749-2887), quantisation groups of 16 or more,
 * one layer per call - dCuTreeOffset of every partition = its dQpOffset - strength * mean over the 16x16 blocks it covers (clipped at the
 * picture edge) of log2(intra + propagate) - log2(intra) + weight_delta, with intra / propagate scaled as in x265hip_cutree_finish.
 * A block whose scaled intra cost is 0 contributes +inf or NaN, as it does upstream.  The frame cost recalculation of such a picture is
 * x265hip_frame_cost_recalculate on the deepest layer's offsets (16 x 16 partitions = the lowres grid). */
typedef struct x265hip_cutree_finish_hevc_params
{
    int width, height;                    /* full-resolution picture size */
    int part;                             /* the layer's partition size: 64, 32 or 16 */
    int blocks_in_row;                    /* lowres blocks per row (Lowres::maxBlocksInRow) */
    const int32_t* intra_cost; const int32_t* inv_qscale; const uint16_t* propagate_cost;      /* HOST, lowres grid */
    int fps_factor_q8;
    double weight_delta, strength;
    const double* qp_offset;              /* HOST double [partitions]: the layer's dQpOffset */
    double* cutree_offset;                /* HOST double [partitions]: the layer's dCuTreeOffset (out) */
} x265hip_cutree_finish_hevc_params;
int x265hip_cutree_finish_hevc_aq(const x265hip_cutree_finish_hevc_params* p);
int x265hip_frame_cost_recalculate_qg8(const x265hip_frame_cost_recalculate_params* p);
typedef struct x265hip_lowres_weight_apply_params
{
    int depth;
    const void* src[4]; void* dst[4];
    intptr_t stride; int rows;
    int scale, denom, offset;
} x265hip_lowres_weight_apply_params;
int x265hip_lowres_weight_apply(const x265hip_lowres_weight_apply_params* p, void* stream);

/* Sample adaptive offset of a deblocked plane (luma, or a chroma plane through ctu_width / ctu_height / plane_offset) - the two
 * pixel passes of encoder/sao.cpp; the rate-distortion choice of
 * the parameters between them (rdoSaoUnitCu, sao.cpp:1225-1605) stays with the host.
 * x265hip_sao_stats: SAO::calcSaoStatsCTU (sao.cpp:735-917) for every 64x64 CTU (bSaoNonDeblocked = 0, bLimitSAO = 0, one slice):
 *   count / offset_org int32 [numCtu][5][32] = samples and sum of (source - deblocked) per type (EO_0, EO_1, EO_2, EO_3, BO) and
 *   class (edge classes 0..4, bands 0..31), over the reference's sub-rectangles of each CTU.
 * x265hip_sao_apply: SAO::generateLumaOffsets + applyPixelOffsets (sao.cpp:572-630, 274-570) for every CTU, out of place:
 *   ctu_params int32 [numCtu][7] (DEVICE) = { typeIdx (-1 off, 0..3 EO, 4 BO), bandPos, offset[4], mergeLeft (informational: a
 *   merged CTU carries the values it inherited) }.  width / height need not be multiples of 64; planes are padded pictures. */
typedef struct x265hip_sao_stats_params
{
    int depth;
    const void* fenc;  intptr_t fenc_stride;
    const void* rec;   intptr_t rec_stride;
    int width, height;
    int32_t* count;  int32_t* offset_org;
    int ctu_width, ctu_height;     /* the CTU's footprint in this plane; 0 = 64 (luma).  4:2:0 chroma: 32 x 32 with the plane's own width / height */
    int plane_offset;              /* the reference's plane_offset (sao.cpp:782): 0 luma, 2 chroma */
} x265hip_sao_stats_params;
int x265hip_sao_stats(const x265hip_sao_stats_params* p, void* stream);
typedef struct x265hip_sao_apply_params
{
    int depth;
    const void* src;  intptr_t src_stride;
    void* dst;        intptr_t dst_stride;
    int width, height;
    const int32_t* ctu_params;
    int ctu_width, ctu_height;     /* as in x265hip_sao_stats_params */
} x265hip_sao_apply_params;
int x265hip_sao_apply(const x265hip_sao_apply_params* p, void* stream);
/* x265hip_sao_decide: the parameters between the two passes for a pipeline that never leaves the device - for every CTU
 * SAO::saoStatsInitialOffset (sao.cpp:1378-1433, exact) and a DISTORTION-ONLY choice of the type (smallest sum of estSaoDist,
 * sao.cpp:56-59, over EO_0..EO_3 and the best four-band window of BO; off when nothing gains).  A cheap stand-in kept for A/B runs;
 * the reference's own decision is x265hip_sao_rdo below.
 *   count / offset_org : DEVICE int32 [nctu][5][32] from x265hip_sao_stats;  init_offset : optional DEVICE int32 [nctu][5][32]
 *   (SAO::m_offset);  ctu_params : DEVICE int32 [nctu][7] in x265hip_sao_apply's format. */
int x265hip_sao_decide(int depth, const int32_t* count, const int32_t* offset_org, int nctu, int32_t* init_offset, int32_t* ctu_params, void* stream);
/* x265hip_sao_rdo (round 3): the REAL rate-distortion decision of the parameters on the device - SAO::rdoSaoUnitCu (sao.cpp:1225-1376) with
 * saoStatsInitialOffset, estIterOffset, saoLumaComponentParamDist / saoChromaComponentParamDist (:1378-1760) and the entropy coder's bit
 * counts for the SAO syntax (entropy.cpp:1221-1292, :2198-2214), for bLimitSAO = 0 / bSaoNonDeblocked = 0 (the x265 defaults).  Every CTU
 * row carries its own context state from the slice's initial state (sao.cpp:245-247), the merge-up candidate reads the row above: two
 * launches, a neighbour-independent one (a workgroup per CTU) and the serial one (a lane per CTU row, rows staggered by a column).
 *   count / offset_org : DEVICE int32 [nctu][5][32] per plane from x265hip_sao_stats (planes = 1: luma only, 4:0:0; 3: Y, Cb, Cr of 4:2:0)
 *   lambda             : floor(256 * x265_lambda2_tab[qp]) for luma and for chroma at the Cb QP (sao.cpp:1229-1238), the HOST's tables;
 *   lambda_ctu         : optional DEVICE int64 [nctu][2] when cu->m_qp[0] varies over the picture (replaces lambda[])
 *   ctx_merge/ctx_type : the slice's initial context states of sao_merge_left/up_flag and sao_type_idx (sbacInit of the slice QP,
 *                        entropy.cpp:1297-1308, 196-208);  frac_bits : Entropy::m_fracBits & 32767 of that state (0 after resetEntropy)
 *   entropy_bits       : HOST pointer to the host's 128 per-state bit costs (g_entropyBits, entropy.cpp:2611) - host-built tables are
 *                        handed in, never recomputed here
 *   sao_flag           : saoParam->bSaoFlag[luma, chroma] (sao.cpp:257-270)
 *   scratch            : DEVICE, x265hip_sao_rdo_scratch_bytes(ctus_w, ctus_h)
 *   ctu_params         : DEVICE int32 [nctu][7] per plane = { typeIdx (-1 off, 0..3 EO, 4 BO), bandPos, offset[4], mergeMode (0 none, 1 left,
 *                        2 up: merged CTUs carry the source's values) } - x265hip_sao_apply's format;  num_no_sao : optional DEVICE int32 [2] */
typedef struct x265hip_sao_rdo_params
{
    int depth;
    int planes;
    int ctus_w, ctus_h;
    const int32_t* count[3];
    const int32_t* offset_org[3];
    int64_t lambda[2];
    const int64_t* lambda_ctu;
    int ctx_merge, ctx_type;
    uint32_t frac_bits;
    const uint32_t* entropy_bits;
    int sao_flag[2];
    void* scratch;
    int32_t* ctu_params[3];
    int32_t* num_no_sao;
} x265hip_sao_rdo_params;
/* the application of 1..3 planes as one launch (the step after x265hip_sao_rdo) */
int x265hip_sao_apply_planes(int nplanes, const x265hip_sao_apply_params* apply, void* stream);
size_t x265hip_sao_rdo_scratch_bytes(int ctus_w, int ctus_h);
int x265hip_sao_rdo(const x265hip_sao_rdo_params* p, void* stream);
/* 1..3 planes of one picture (Y, Cb, Cr) through statistics -> parameters -> application with ONE launch per step (the steps are
 * latency-bound rounds of workgroups: three planes in one launch cost one latency, not three).  stats[i] / apply[i] as the single-plane
 * entries take them; apply[i].ctu_params (DEVICE, written) receives plane i's parameters; apply = NULL: statistics only. */
int x265hip_sao_planes(int nplanes, const x265hip_sao_stats_params* stats, const x265hip_sao_apply_params* apply, void* stream);

/* ------------------------------------------------------------------ generic job-list entry points
 * Every remaining family evaluates `njobs` independent blocks of one size per launch.  An operand
 * is a device plane (base pointer + element stride); a job carries up to four element offsets into
 * the operand planes plus four integer arguments.  `jobs` is a DEVICE array. */
typedef struct x265hip_plane { void* base; intptr_t stride; } x265hip_plane;
typedef struct x265hip_job { int64_t off[4]; int32_t arg[4]; } x265hip_job;

/* Intra TU candidate set - the pixel work of Search::codeIntraLumaQT for a list of (TU, mode) candidates
 * (encoder/search.cpp:335-373): Predict::predIntraLumaAng (predict.cpp:579-588), calcresidual, Quant::transformNxN (non-RDOQ,
 * flat scaling; DST-VII for the 4x4 luma TU), Quant::invtransformNxN, add_ps / copy_pp, sse_pp.  One job per candidate:
 *   off[0] source block (elements into fenc), off[1] unfiltered / off[2] filtered neighbour arrays ([0] corner, [1..2n] above,
 *   [2n+1..4n] left; elements into nb), off[3] the candidate's reconstruction block (elements into recon); arg[0] = mode 0..34.
 * Outputs per job: levels int16 [n*n], num_sig, dist (SSE source vs reconstruction).  Bit costs stay with the host. */
typedef struct x265hip_intra_recon_params
{
    int depth, n;
    const void* fenc;  intptr_t fenc_stride;
    const void* nb;
    void* recon;       intptr_t recon_stride;
    int qp, intra_slice;           /* intra_slice: X265HIP_TU_* flag bits, as in x265hip_recon_params */
    const x265hip_job* jobs;  int njobs;
    int16_t* levels; uint32_t* num_sig; uint64_t* dist;
    /* non-zero: the 4:2:0 chroma flavour (Search::codeIntraChromaQt's pixel work, search.cpp:899-930) - Predict::predIntraChromaAng
     * (predict.cpp:590-598) predicts from the UNFILTERED neighbours (off[2] is not read) with bFilter = 0, the 4x4 TU takes the DCT
     * (useDST needs TEXT_LUMA, quant.cpp:426,583); fenc / nb / recon are the chroma plane's, n = 4..32, qp = the chroma QP the
     * host mapped (Quant::setChromaQP) + QP_BD_OFFSET */
    int chroma;
    const x265hip_tu_tables* tables;            /* HOST pointer to the table record, or NULL */
} x265hip_intra_recon_params;
int x265hip_intra_recon_batch(const x265hip_intra_recon_params* p, void* stream);

/* Sub-pel interpolation (reference ipfilter.cpp:79-369; taps constants.cpp:250-268).
 * taps = 8 (luma, coeffIdx 0..3) or 4 (chroma, 0..7).  src = plane 0 (off[0]), dst = plane 1 (off[1]).
 *   HPP/VPP: pixel->pixel  arg[0]=coeffIdx            HPS: pixel->int16 arg[0]=coeffIdx arg[1]=isRowExt
 *   VPS: pixel->int16       VSP: int16->pixel          VSS: int16->int16
 *   HVPP: pixel->pixel arg[0]=idxX arg[1]=idxY (hps with row extension + vsp, ipfilter.cpp:362-369)
 *   P2S: pixel->int16 (ipfilter.cpp:40-57) */
enum x265hip_interp_kind
{
    X265HIP_IP_HPP = 0, X265HIP_IP_HPS, X265HIP_IP_VPP, X265HIP_IP_VPS, X265HIP_IP_VSP, X265HIP_IP_VSS,
    X265HIP_IP_HVPP, X265HIP_IP_P2S
};
int x265hip_interp_batch(int kind, int depth, int taps, int w, int h, x265hip_plane src, x265hip_plane dst,
                         const x265hip_job* jobs, int njobs, void* stream);

/* Transforms (reference dct.cpp:43-610, lowpassdct.cpp:34-113).  n = 4, 8, 16, 32.
 *   DCT / DST4 / LOWPASS: src plane 0 = int16 residual with stride (off[0]); dst plane 1 = n*n contiguous
 *   int16 per job at off[1].   IDCT / IDST4: src plane 0 = n*n contiguous (off[0]); dst plane 1 strided (off[1]).
 * use_mfma != 0 selects the int8-limb MFMA kernels for n = 16 / 32 (bit-identical results). */
enum x265hip_tr_kind { X265HIP_TR_DCT = 0, X265HIP_TR_IDCT, X265HIP_TR_DST4, X265HIP_TR_IDST4, X265HIP_TR_LOWPASS_DCT };
int x265hip_transform_batch(int kind, int depth, int n, x265hip_plane src, x265hip_plane dst,
                            const x265hip_job* jobs, int njobs, int use_mfma, void* stream);

/* Quantisation family (reference dct.cpp:612-755).  All operands are contiguous per job; planes give the
 * base pointers (stride unused).  arg[] meanings follow the reference signatures:
 *   QUANT   : p0 coef(int16) p1 quantCoeff(int32) p2 deltaU(int32 out) p3 qCoef(int16 out); arg = {qBits, add, numCoeff}
 *   NQUANT  : p0 coef p1 quantCoeff p3 qCoef(out); arg = {qBits, add, numCoeff}
 *   DEQUANT_NORMAL : p0 quantCoef(int16) p3 coef(int16 out); arg = {num, scale, shift}
 *   DEQUANT_SCALING: p0 quantCoef p1 deQuantCoef(int32) p3 coef(out); arg = {num, per, shift}
 *   DENOISE : p0 dctCoef(int16 in/out) p1 resSum(uint32 in/out) p2 offset(uint16); arg = {numCoeff}
 *   COUNT_NONZERO : p0 coef; arg = {numCoeff}       COPY_CNT: p0 residual (strided, stride = plane stride) p3 coeff out; arg = {n}
 * result[i] (uint32, may be NULL for void kinds) = numSig / count. */
enum x265hip_quant_kind
{
    X265HIP_Q_QUANT = 0, X265HIP_Q_NQUANT, X265HIP_Q_DEQUANT_NORMAL, X265HIP_Q_DEQUANT_SCALING, X265HIP_Q_DENOISE,
    X265HIP_Q_COUNT_NONZERO, X265HIP_Q_COPY_CNT
};
int x265hip_quant_batch(int kind, const x265hip_plane planes[4], const x265hip_job* jobs, int njobs,
                        uint32_t* result, void* stream);

/* Intra prediction (reference intrapred.cpp:31-234).  n = 4, 8, 16, 32.
 *   PRED   : plane 0 = neighbour buffers (off[0] -> srcPix[0], layout [0]=top-left, [1..2n]=above,
 *            [2n+1..4n]=left), plane 1 = dst (off[1], strided); arg = {dirMode 0..34, bFilter}
 *   FILTER : plane 0 = samples (off[0]), plane 1 = filtered out (off[1])
 *   ALLANGS: plane 0 = refPix (off[0]) and filtPix (off[2]), plane 1 = dest 33*n*n (off[1]); arg = {bLuma} */
enum x265hip_intra_kind { X265HIP_INTRA_PRED = 0, X265HIP_INTRA_FILTER, X265HIP_INTRA_ALLANGS };
int x265hip_intra_batch(int kind, int depth, int n, x265hip_plane src, x265hip_plane dst,
                        const x265hip_job* jobs, int njobs, void* stream);

/* Element-wise block operations (reference pixel.cpp:393-602,759-862).  Operand planes / offsets:
 *   COPY_PP/PS/SP/SS: dst=p0 src=p1           SUB_PS: dst(int16)=p0 a=p1 b=p2      ADD_PS: dst(pixel)=p0 a(pixel)=p1 r(int16)=p2
 *   ADDAVG: dst(pixel)=p0 a(int16)=p1 b(int16)=p2   PIXELAVG: dst=p0 a=p1 b=p2      BLOCKFILL: dst(int16)=p0 arg[0]=val
 *   CPY2DTO1D_SHL/SHR: dst contiguous=p0 src strided=p1 arg[0]=shift   CPY1DTO2D_SHL/SHR: dst strided=p0 src contiguous=p1
 *   TRANSPOSE: dst contiguous=p0 src=p1       WEIGHT_PP: dst=p0 src=p1 arg={w0, round, shift, offset}
 *   WEIGHT_SP: dst(pixel)=p0 src(int16)=p1 arg={w0, round, shift, offset}
 *   SCALE1D_128TO64: dst=p0 src=p1 (w=h=0)    SCALE2D_64TO32: dst=p0 src=p1
 *   SSE_SS (int16,int16), SSD_S (int16), VAR (pixel): reductions, result in `result` (uint64 per job)
 *   SSIM_DIST: cu[].ssimDist (pixel.cpp:958-981): fenc=p0 recon=p1 arg[0]=shift; result[2 job] = sum (fenc - recon)^2, result[2 job + 1] =
 *              sum (fenc >> shift)^2      NORM_FACT: cu[].normFact (:983-995): src=p0 arg[0]=shift; result[job] = sum (src >> shift)^2 */
enum x265hip_blockop_kind
{
    X265HIP_OP_COPY_PP = 0, X265HIP_OP_COPY_PS, X265HIP_OP_COPY_SP, X265HIP_OP_COPY_SS, X265HIP_OP_SUB_PS, X265HIP_OP_ADD_PS,
    X265HIP_OP_ADDAVG, X265HIP_OP_PIXELAVG, X265HIP_OP_BLOCKFILL, X265HIP_OP_CPY2DTO1D_SHL, X265HIP_OP_CPY2DTO1D_SHR,
    X265HIP_OP_CPY1DTO2D_SHL, X265HIP_OP_CPY1DTO2D_SHR, X265HIP_OP_TRANSPOSE, X265HIP_OP_WEIGHT_PP, X265HIP_OP_WEIGHT_SP,
    X265HIP_OP_SCALE1D_128TO64, X265HIP_OP_SCALE2D_64TO32, X265HIP_OP_SSE_SS, X265HIP_OP_SSD_S, X265HIP_OP_VAR,
    X265HIP_OP_SSIM_DIST, X265HIP_OP_NORM_FACT
};
int x265hip_blockop_batch(int op, int depth, int w, int h, const x265hip_plane planes[3],
                          const x265hip_job* jobs, int njobs, uint64_t* result, void* stream);

/* Loop-filter family: SAO apply / statistics, deblocking edge filters, sign, SEA integrals, ADS.
 * (reference loopfilter.cpp:39-180, encoder/sao.cpp:1762-1925, encoder/framefilter.cpp:39-140,
 * pixel.cpp:121-165).  Operand conventions are documented with each kind in csrc/loopfilter_kernels.hip;
 * the table layer is their main client in this round. */
enum x265hip_lf_kind
{
    X265HIP_LF_SIGN = 0, X265HIP_LF_SAO_E0, X265HIP_LF_SAO_E1, X265HIP_LF_SAO_E1_2ROWS, X265HIP_LF_SAO_E2, X265HIP_LF_SAO_E3,
    X265HIP_LF_SAO_B0, X265HIP_LF_STATS_BO, X265HIP_LF_STATS_E0, X265HIP_LF_STATS_E1, X265HIP_LF_STATS_E2, X265HIP_LF_STATS_E3,
    X265HIP_LF_DEBLOCK_LUMA_STRONG, X265HIP_LF_DEBLOCK_CHROMA, X265HIP_LF_INTEGRAL_H, X265HIP_LF_INTEGRAL_V, X265HIP_LF_ADS
};
int x265hip_loopfilter_batch(int kind, int depth, const x265hip_plane planes[4], const x265hip_job* jobs, int njobs,
                             uint32_t* result, void* stream);

/* Frame-level helpers (csrc/frame_coeff_kernels.hip).  src = plane 0 (off[0]), dst = plane 1 (off[1]), offsets and
 * strides in elements of the operand's own type; w x h is the size every job of the batch shares.
 *   PLANECOPY_CP     : planecopy_cp (pixel.cpp:864-874), uint8 -> pixel;   arg = {shift}
 *   PLANECOPY_SP     : planecopy_sp (:876-886), uint16 -> pixel;           arg = {shift, mask}      (src >> shift) & mask
 *   PLANECOPY_SP_SHL : planecopy_sp_shl (:888-898), uint16 -> pixel;       arg = {shift, mask}      (src << shift) & mask
 *   PLANECOPY_PP_SHR : planecopy_pp_shr (:900-910), pixel -> pixel;        arg = {shift}
 *   PLANE_CLIP_MAX   : planeClipAndMax (:996-1016), plane 0 clamped IN PLACE to arg = {minPix, maxPix}; out uint64 [job][2] = {maximum, sum}
 *   SSIM_CORE        : ssim_4x4x2_core (:631-657), pix1 = plane 0, pix2 = plane 1 (w, h unused); out int32 [job][2][4]
 *   SSIM_END4        : ssim_end_4 (:659-701), sum0 = plane 0, sum1 = plane 1 (int32 [5][4] each); arg = {width 1..4}; out float [job]
 *                      (integer moment arithmetic at depth 8, float above: the reference's HIGH_BIT_DEPTH split)
 *   FIX8_PACK        : cuTreeFix8Pack (:945-950), double -> uint16, w = count (h unused)
 *   FIX8_UNPACK      : cuTreeFix8Unpack (:952-958), uint16 -> double, w = count */
enum x265hip_frame_kind
{
    X265HIP_FR_PLANECOPY_CP = 0, X265HIP_FR_PLANECOPY_SP, X265HIP_FR_PLANECOPY_SP_SHL, X265HIP_FR_PLANECOPY_PP_SHR, X265HIP_FR_PLANE_CLIP_MAX,
    X265HIP_FR_SSIM_CORE, X265HIP_FR_SSIM_END4, X265HIP_FR_FIX8_PACK, X265HIP_FR_FIX8_UNPACK
};
int x265hip_frame_batch(int kind, int depth, int w, int h, const x265hip_plane planes[2], const x265hip_job* jobs, int njobs,
                        void* out, void* stream);
/* frameInitLowres / frameInitLowerRes (pixel.cpp:604-629) of any size: `src` is read up to row 2 * height and column 2 * width; DEVICE
 * pointers.  (x265hip_lowres_init is the whole-picture form with the border extension fused.) */
int x265hip_frame_init_lowres(int depth, const void* src, intptr_t src_stride, void* const dst[4], intptr_t dst_stride,
                              int width, int height, void* stream);
/* propagateCost = estimateCUPropagateCost (pixel.cpp:914-943) over `len` lowres blocks, DEVICE pointers; fps_factor as the slot takes it
 * (divided by 256 inside).  (x265hip_cutree_propagate is the fused form with the scatter to the references.) */
int x265hip_propagate_cost(int32_t* dst, const uint16_t* propagate_in, const int32_t* intra_costs, const uint16_t* inter_costs,
                           const int32_t* inv_qscales, double fps_factor, int len, void* stream);

/* RDOQ helpers of Quant::rdoQuant (reference dct.cpp:757-1069; call sites quant.cpp:683-830, 1301, entropy.cpp:1856-2112),
 * batch forms: one call of the slot per job, thousands of coefficient groups per launch.  The CABAC state makes every call serial inside
 * (a lane walks it); each job works on its OWN copy of the context bytes.  bufs[0..4] are DEVICE base pointers, off[k] the element offset
 * of the job's operand in bufs[k]:
 *   SCAN_POS_LAST        : scanPosLast.   bufs0 scan (uint16), bufs1 coeff (int16), out bufs2 coeffSign / bufs3 coeffFlag (uint16 [64]),
 *                          bufs4 coeffNum (uint8 [64]); arg = {numSig, trSize}; result = the last scan position
 *   FIND_POS_FIRST_LAST  : findPosFirstLast.  bufs0 scanTbl (uint16 [16]), bufs1 the group's first coefficient; arg = {trSize}; result = packed
 *   COST_COEFF_NXN       : costCoeffNxN.  bufs0 scan (uint16 [16]), bufs1 the group's first coefficient, bufs2 absCoeff (uint16 out, the pointer the
 *                          caller passes: the levels of the non-zeros in coding order from [0]), bufs3 tabSigCtx (uint8 [16]), bufs4 baseCtx (uint8, updated);
 *                          arg = {trSize, scanFlagMask, offset, scanPosSigOff, subPosBase}; result = bits
 *   COST_COEFF_REMAIN    : costCoeffRemain.  bufs2 absCoeff; arg = {numNonZero, idx}; result = bits
 *   COST_C1C2            : costC1C2Flag.  bufs2 absCoeff, bufs4 baseCtxMod (updated); arg = {numC1Flag, ctxOffset}; result = packed
 *   RDOQ_NONPSY / _PSY / _PSY_1P / _PSY_2P : cu[].nonPsyRdoQuant / psyRdoQuant / psyRdoQuant_1p / _2p of ONE coefficient group.  bufs0 fenc's
 *                          transform, bufs1 the residual's transform (int16, block origin), bufs2 costUncoded (int64, block origin), bufs3
 *                          {totalUncodedCost, totalRdCost} (int64 [2], added to), bufs4 psyScale (int64); arg = {blkPos, log2TrSize, row stride or 0 = 1 << log2TrSize}; no result
 * costCoeffNxN / costC1C2Flag price bins with the HOST's per-state table: hand g_entropyBits (entropy.cpp:2611; x265_entropyStateBits is
 * accepted as well, its top byte is ignored) to x265hip_set_entropy_bits once - the table is the encoder's data, not this library's. */
enum x265hip_coeff_kind
{
    X265HIP_CF_SCAN_POS_LAST = 0, X265HIP_CF_FIND_POS_FIRST_LAST, X265HIP_CF_COST_COEFF_NXN, X265HIP_CF_COST_COEFF_REMAIN, X265HIP_CF_COST_C1C2,
    X265HIP_CF_RDOQ_NONPSY, X265HIP_CF_RDOQ_PSY, X265HIP_CF_RDOQ_PSY_1P, X265HIP_CF_RDOQ_PSY_2P
};
typedef struct x265hip_coeff_job { int64_t off[5]; int32_t arg[5]; int32_t reserved; } x265hip_coeff_job;
int x265hip_set_entropy_bits(const uint32_t* bits128);
int x265hip_coeff_batch(int kind, int depth, void* const bufs[5], const x265hip_coeff_job* jobs, int njobs, uint32_t* result, void* stream);


/* ---------------------------------------------------------------------------------------------------------------
 * Frame-granular CONSUMER of the exhaustive search (csrc/me_cache.hip): host planes in, SAD surfaces in pinned host memory
 * out, one launch per (source picture, reference picture); the encoder's sad / sad_x3 / sad_x4 stubs look the results up
 * (SURVEY.md section 7 step 6).  Serves the call sites motion.cpp:228-328, :397-604, :1069-1083, :1397-1445 without touching
 * the decision code; a lookup that cannot be served (row not there yet, displacement outside the window, PU not a union of
 * 8x8 blocks) falls back to the host's own primitive - identical values either way.
 *   width, height : whole CTUs (multiples of 64);  stride / margin_x / margin_y : the PicYuv geometry (picyuv.cpp:87-114),
 *                   buffers handed to submit are the WHOLE allocated planes: stride * (height + 2 * margin_y) samples
 *   range         : surfaces cover displacements [-range, range]^2 around each CTU's own position
 *   slots         : number of (source, reference) pairs resident in host memory at once */
typedef struct x265hip_me_cache x265hip_me_cache;
typedef struct x265hip_me_cache_params
{
    int depth;
    int width, height;
    intptr_t stride;
    int margin_x, margin_y;
    int range;
    int surf_format;                /* X265HIP_SURF_PACKED / X265HIP_SURF_PACKED_T (8-bit) or X265HIP_SURF_I32 */
    int slots;
} x265hip_me_cache_params;
typedef struct x265hip_me_cache_stats_t
{
    uint64_t fills, failed, batches;
    uint64_t us_upload, us_kernel, us_download;     /* summed over the batches, worker-thread wall time: us_upload = the source pictures' uploads, us_kernel = uploads + searches */
    uint64_t bytes_downloaded, surface_bytes;
} x265hip_me_cache_stats_t;
int  x265hip_me_cache_create(x265hip_me_cache** out, const x265hip_me_cache_params* p);
void x265hip_me_cache_destroy(x265hip_me_cache* c);
/* copies both planes, queues upload + search + row-streamed download on the cache's worker thread, returns the slot's new
 * GENERATION (> 0) at once, or a negative error.  fenc_key names the source picture: it is copied once per key, and every queued pair
 * keeps ITS source until the worker has searched it - pairs of several source pictures may be queued back to back without waiting
 * (frame threads); X265HIP_EBUSY only when 64 source pictures are waiting for the worker */
int  x265hip_me_cache_submit(x265hip_me_cache* c, int slot, const void* fenc_buf, uint64_t fenc_key, const void* ref_buf);
/* the references of one source picture as ONE batch: searched back to back, surfaces downloaded row-interleaved across the pairs
 * (row 0 of every pair first - the order a wavefront-parallel encoder needs them); generations[i] = slot i's new generation */
int  x265hip_me_cache_submit_batch(x265hip_me_cache* c, int n, const int* slots, const void* fenc_buf, uint64_t fenc_key,
                                   const void* const* ref_bufs, int* generations);
const void* x265hip_me_cache_surface(x265hip_me_cache* c, int slot);
/* int [height / 64]: CTU row r of the slot is complete when ready[r] == the generation submit returned */
const volatile int* x265hip_me_cache_ready(x265hip_me_cache* c, int slot);
int  x265hip_me_cache_stats(x265hip_me_cache* c, x265hip_me_cache_stats_t* st);

/* ---------------------------------------------------------------------------------------------------------------
 * ROW-GRANULAR consumer of the exhaustive search (csrc/me_stream.hip) for hosts that encode several pictures at once - the reference's
 * frame threads.  Picture k + 1 starts while picture k is still being reconstructed; the reference publishes every finished CTU row of a
 * reconstructed picture through Frame::m_reconRowFlag (encoder/framefilter.cpp:664) and consumers wait row by row
 * (encoder/frameencoder.cpp:852-868).  The host hands rows over where it raises that flag (x265hip_me_stream_picture_rows), opens a
 * (source, reference) pair when the first search of the pair is about to run (x265hip_me_stream_pair_open), and the service searches
 * every CTU row of every open pair as soon as the reference rows its window reaches - row + (63 + range) / 64 - are on the device,
 * raising ready[row] as the band's surfaces land in pinned host memory.  The host's own consumers wait for row + 3 (m_refLagRows with
 * the sub-pel taps, frameencoder.cpp:161-164), so the device is two reference rows ahead of the first lookup of a row.
 *   surf_format : record-contiguous formats only - X265HIP_SURF_PACKED (8-bit) or X265HIP_SURF_I32
 *   min_level   : 0 = whole records; 2 (X265HIP_STREAM_PLANES only) = the 32x32 / 64x64 rasters only - in the real encoder the 16x16-level lookups are
 *                 fps-neutral (a lookup costs what the host's own 16x16 SAD costs, profiles/r04_encoder_legs.txt) and 62 % of the bytes; 1 = the 16x16 / 32x32 / 64x64 levels only: the LAST X265HIP_SURF_TAIL_BYTES_* bytes of every record
 *                 (packed: uint16 [16][4] + int32 [5][4] = 208 bytes; int32: [21][4] = 336 bytes), same group order
 *   slots       : (source, reference) pairs resident in host memory at once;  pictures : pictures resident on the device at once
 *   band_rows   : most CTU rows searched by one launch (0 = 8)
 * Keys name picture CONTENT (e.g. POC * 2 + is-reconstruction); a key may be reused once no open pair refers to it.
 * Readers: a row of a slot is valid while ready[row] == the generation pair_open returned - check it before AND after reading (a
 * reopened slot has its flags cleared before any row is rewritten).  Anything else is the host primitive's to answer. */
#define X265HIP_SURF_TAIL_BYTES_PACKED 208
#define X265HIP_SURF_TAIL_BYTES_I32    336
typedef struct x265hip_me_stream x265hip_me_stream;
typedef struct x265hip_me_stream_params
{
    int depth;
    int width, height;
    intptr_t stride;
    int margin_x, margin_y;
    int range;
    int surf_format;
    int min_level;
    int slots;
    int pictures;
    int band_rows;
    int layout;                              /* X265HIP_STREAM_RECORDS (0) or X265HIP_STREAM_PLANES */
    int centre_range;                        /* 0: windows centred on (0, 0); R' >= range: on each CTU's own displacement (see below) */
    int device_plus_1;                       /* 0: the calling thread's current device; d + 1: device d - one instance per GPU, e.g. one per frame-encoder
                                              * pool of a host that spreads its FrameEncoders over the GPUs of a node (encoder/encoder.cpp:304-321) */
} x265hip_me_stream_params;
/* layout X265HIP_STREAM_PLANES (round 4) - what a host search wants to read: PU-MAJOR.  Per CTU (x265hip_me_stream_ctu_bytes apart),
 * for every square PU from the first served level on (min_level 0: 64 8x8, then 16 16x16, 4 32x32, 1 64x64, z-order per level) one
 * raster of (2 range + 1) rows x pitch = 4 * ((2 range + 4) / 4) entries: entry [dy + range][dx + range] = the PU's SAD at
 * displacement centre + (dx, dy).  8x8 and 16x16 rasters are uint16 and SATURATE: 65535 = "not representable" (only possible above
 * 8 bits) - the host computes that one itself; 32x32 and 64x64 rasters are uint32.  One search walks one PU over neighbouring
 * displacements, so its probes share a few cache lines of a 2 - 10 KB raster instead of missing once per probe.
 * centre_range = R' > 0: before a band's surfaces are computed, a minima-only exhaustive search of +-R' (SAD alone) finds where each
 * CTU's 64x64 block went; the CTU's window is centred there (clamped to |c| <= margin - range - 12 so it stays inside the picture
 * margins).  x265hip_me_stream_centres(slot)[2 ctu], [2 ctu + 1] = that displacement, valid with the row's ready flag. */
enum { X265HIP_STREAM_RECORDS = 0, X265HIP_STREAM_PLANES = 1 };
static inline size_t x265hip_stream_planes_ctu_bytes(int range, int min_level)
{
    const size_t nc = (size_t)(2 * range + 1), pitch = 4 * ((nc + 3) >> 2);
    return nc * pitch * ((min_level ? 0 : 64 * 2) + (min_level > 1 ? 0 : 16 * 2) + 5 * 4);
}
/* byte offset of square PU (level 0..3, z-order index z) inside a CTU's planes, and its entry size (2 or 4) */
static inline size_t x265hip_stream_planes_pu_offset(int range, int min_level, int level, int z, int* entry_bytes)
{
    const size_t nc = (size_t)(2 * range + 1), pitch = 4 * ((nc + 3) >> 2), ps = nc * pitch * 2, pw = nc * pitch * 4;
    const size_t n0 = min_level ? 0 : 64, n1 = min_level > 1 ? 0 : 16;     /* a level below min_level has no rasters: do not ask for it */
    if (entry_bytes) *entry_bytes = level < 2 ? 2 : 4;
    if (level == 0) return (size_t)z * ps;
    if (level == 1) return (n0 + (size_t)z) * ps;
    return (n0 + n1) * ps + (size_t)(level == 2 ? z : 4) * pw;
}
typedef struct x265hip_me_stream_stats_t
{
    uint64_t pairs_opened, pairs_completed, bands, rows_searched, rows_uploaded, failed, stale_pairs;
    uint64_t us_busy;                        /* worker-thread wall time inside uploads / launches / downloads */
    uint64_t bytes_downloaded, bytes_uploaded, surface_bytes;
    uint64_t rows_weighted, weighted_pairs;  /* CTU rows weighted on the device; pairs opened on a weighted reference */
} x265hip_me_stream_stats_t;
int  x265hip_me_stream_create(x265hip_me_stream** out, const x265hip_me_stream_params* p);
void x265hip_me_stream_destroy(x265hip_me_stream* s);
/* CTU rows [ctu_row0, ctu_row0 + ctu_rows) of picture `key` are final in `buf` = the whole allocated plane (the top margin travels with
 * row 0, the bottom margin with the last row); the rows are copied before the call returns.  X265HIP_EBUSY: no picture entry free. */
int  x265hip_me_stream_picture_rows(x265hip_me_stream* s, uint64_t key, const void* buf, int ctu_row0, int ctu_rows);
/* returns the slot's new GENERATION (> 0) or a negative error; the pictures' rows may arrive before or after */
int  x265hip_me_stream_pair_open(x265hip_me_stream* s, int slot, uint64_t fenc_key, uint64_t ref_key);
/* The same on a WEIGHTED reference (x265's default --weightp: MotionReference::applyWeight materialises primitives.weight_pp of every
 * finished reconstructed row into a second plane and the search reads that plane, encoder/reference.cpp:119-178,
 * encoder/frameencoder.cpp:865-866).  w = the arguments of weight_pp (common/pixel.cpp:518-543):
 *     dst = clip(((w0 * (src << (14 - depth)) + round) >> shift) + offset)      i.e. round and shift INCLUDE the 14 - depth correction
 * The service weights the rows of picture ref_key on the device as they arrive (margins included: a replicated border sample weights
 * to the replicated weighted sample, so the result is the host's plane sample for sample) and searches that plane; pairs with the
 * same (ref_key, w) share it.  w = NULL is x265hip_me_stream_pair_open. */
typedef struct x265hip_weight { int w0, round, shift, offset; } x265hip_weight;
int  x265hip_me_stream_pair_open_weighted(x265hip_me_stream* s, int slot, uint64_t fenc_key, uint64_t ref_key, const x265hip_weight* w);
const void* x265hip_me_stream_surface(x265hip_me_stream* s, int slot);
const volatile int* x265hip_me_stream_ready(x265hip_me_stream* s, int slot);      /* int [height / 64] */
int  x265hip_me_stream_record_bytes(x265hip_me_stream* s);           /* X265HIP_STREAM_RECORDS; 0 in the planes layout */
size_t x265hip_me_stream_ctu_bytes(x265hip_me_stream* s);            /* bytes of one CTU's surfaces in the slot buffers, either layout */
const int16_t* x265hip_me_stream_centres(x265hip_me_stream* s, int slot);
int  x265hip_me_stream_stats(x265hip_me_stream* s, x265hip_me_stream_stats_t* st);

/* Address arithmetic of a surface record (every format), usable from any host language: the SAD of square PU `z` (z-order
 * index inside its level; level 0..3 = 8x8, 16x16, 32x32, 64x64) of CTU `ctu` at displacement (dx, dy), |dx|, |dy| <= range. */
static inline int32_t x265hip_surf_lookup(const void* surf, int surf_format, int range, int ctu, int level, int z, int dx, int dy)
{
    const int nc = 2 * range + 1, ng = (nc + 3) >> 2, col = dx + range;
    if (surf_format == X265HIP_SURF_I32)
    {
        static const int base[4] = { 0, 64, 80, 84 };
        const unsigned char* g = (const unsigned char*)surf + (((size_t)ctu * nc + (size_t)(dy + range)) * ng + (size_t)(col >> 2)) * X265HIP_SURF_GROUP_BYTES_I32;
        return ((const int32_t*)g)[(base[level] + z) * 4 + (col & 3)];
    }
    /* byte offset of the value inside the 720-byte packed record */
    static const int pbase[4] = { 0, 512, 640, 704 };
    const size_t o = (size_t)pbase[level] + (size_t)(z * 4 + (col & 3)) * (level < 2 ? 2 : 4);
    if (surf_format == X265HIP_SURF_PACKED_B)
    {
        const size_t r = (size_t)(dy + range) * ng + (size_t)(col >> 2);
        const unsigned char* vb = (const unsigned char*)surf + (size_t)ctu * x265hip_surf_ctu_bytes(surf_format, range) + (r >> 6) * X265HIP_SURF_BLOCK_BYTES_PACKED
                                  + ((o >> 4) * 64 + (r & 63)) * 16 + (o & 15);
        return level < 2 ? *(const uint16_t*)vb : *(const int32_t*)vb;
    }
    const unsigned char* row = (const unsigned char*)surf + ((size_t)ctu * nc + (size_t)(dy + range)) * ng * X265HIP_SURF_GROUP_BYTES_PACKED;
    const unsigned char* v = surf_format == X265HIP_SURF_PACKED_T ? row + ((o >> 4) * ng + (size_t)(col >> 2)) * 16 + (o & 15)
                                                                   : row + (size_t)(col >> 2) * X265HIP_SURF_GROUP_BYTES_PACKED + o;
    return level < 2 ? *(const uint16_t*)v : *(const int32_t*)v;
}


/* ---------------------------------------------------------------------------------------------------------------
 * Fractional-phase planes of a reference picture (csrc/phase_kernels.hip, csrc/phase_cache.hip): every sub-sample position an
 * encoder can ask of a reference picture, interpolated ONCE per picture instead of once per candidate block.
 *   luma   : 15 planes, phase = yFrac * 4 + xFrac (quarter samples), plane of phase p at dst + (p - 1) * plane_bytes; sample (x, y) of
 *            a plane = what luma_hpp (yFrac 0) / luma_vpp (xFrac 0) / luma_hvpp (both) write for source position (x, y)
 *            (ipfilter.cpp:79-118, :164-203, :362-369 - the calls of MotionEstimate::subpelCompare, motion.cpp:1593-1598, and of
 *            Predict::predInterLumaPixel, predict.cpp:261-281)
 *   chroma : 63 planes per chroma plane, phase = yFrac * 8 + xFrac (eighth samples, 4:2:0): filter_hpp / filter_vpp /
 *            filter_hps(isRowExt) + filter_vsp (motion.cpp:1628-1657, predict.cpp:304-351)
 * The planes have the geometry of the source plane (same pitch, same rows), so a block's address in a phase plane is its address in
 * the source plane plus a constant; the interpolated block needs no copy, the comparison primitive reads it in place.
 * Batch-layer entry (device pointers):
 *   src  : the padded plane, `rows` rows of `stride` samples (no guard memory needed: the first 4 and the last 8 rows of every phase
 *          plane are not produced, and samples within 4 of a row end depend on the neighbouring rows - no valid block lies there)
 *   dst  : (chroma ? 63 : 15) planes of stride * rows samples */
typedef struct x265hip_phase_planes_params
{
    int depth;
    int chroma;                     /* 0: 8-tap luma set, quarter phases; 1: 4-tap chroma set, eighth phases */
    const void* src;
    void* dst;
    intptr_t stride;                /* samples; stride * bytes-per-sample must be a multiple of 4 */
    int rows;                       /* multiple of 4 */
} x265hip_phase_planes_params;
int x265hip_phase_planes(const x265hip_phase_planes_params* p, void* stream);

/* Picture-granular CONSUMER: host planes in (the reference's PicYuv buffers, margins included), phase planes in pinned host memory
 * out; a worker thread uploads, interpolates and downloads while the encoder goes on.  `ready` is int[2]: luma planes are complete when
 * ready[0] == the generation submit returned, both chroma planes when ready[1] == it.  Until then (and for anything else) the host
 * interpolates the block itself - same samples either way. */
typedef struct x265hip_phase_cache x265hip_phase_cache;
typedef struct x265hip_phase_cache_params
{
    int depth;
    intptr_t stride;   int rows;       /* luma buffer: pitch in samples, allocated rows (height + 2 * margin) */
    intptr_t stride_c; int rows_c;     /* each chroma buffer; rows_c = 0: luma only */
    int slots;                         /* reference pictures resident in host memory at once */
} x265hip_phase_cache_params;
typedef struct x265hip_phase_cache_stats_t
{
    uint64_t fills, failed;
    uint64_t us_upload_kernel, us_download;
    uint64_t bytes_downloaded, bytes_per_picture;
} x265hip_phase_cache_stats_t;
int  x265hip_phase_cache_create(x265hip_phase_cache** out, const x265hip_phase_cache_params* p);
void x265hip_phase_cache_destroy(x265hip_phase_cache* c);
/* copies the three buffers (cb / cr may be NULL when rows_c = 0), queues the work, returns the slot's new GENERATION (> 0) or < 0 */
int  x265hip_phase_cache_submit(x265hip_phase_cache* c, int slot, const void* luma_buf, const void* cb_buf, const void* cr_buf);
/* plane = 0: the 15 luma planes, 1 / 2: the 63 Cb / Cr planes (pinned host memory, fixed for the life of the cache) */
const void* x265hip_phase_cache_planes(x265hip_phase_cache* c, int slot, int plane);
const volatile int* x265hip_phase_cache_ready(x265hip_phase_cache* c, int slot);
int  x265hip_phase_cache_stats(x265hip_phase_cache* c, x265hip_phase_cache_stats_t* st);


/* ROW-GRANULAR flavour (csrc/phase_stream.hip) for hosts that encode several pictures at once - the reference's frame threads.
 * PICTURES are named by a key; the producer hands their CTU rows over as they become final (where the reference raises
 * Frame::m_reconRowFlag, encoder/framefilter.cpp:664: x265hip_phase_stream_picture_rows).  A slot holds a VIEW of one picture - its
 * fractional-phase planes, growing line by line behind the rows - opened by the consumer the first time a search refers to it
 * (x265hip_phase_stream_view_open); a WEIGHTED reference (x265's default --weightp: the search reads primitives.weight_pp of the
 * reconstruction, encoder/reference.cpp:119-178) is a view of its own, weighted on the device before the interpolation.
 *   progress : uint64 [2] per slot, [0] luma planes, [1] both chroma plane sets: generation << 32 | lines finished, counted from the top
 *              of the buffer (lines [4, finished) of every phase plane are valid).  A reader checks it before AND after reading a block.
 *   4:2:0 only when rows_c > 0: a CTU row is 64 luma / 32 chroma lines; rows = ctu_rows * 64 + 2 * margin_y, rows_c likewise with 32. */
typedef struct x265hip_phase_stream x265hip_phase_stream;
typedef struct x265hip_phase_stream_params
{
    int depth;
    intptr_t stride;   int rows;   int margin_y;        /* luma buffer: pitch in samples, allocated rows, rows above sample (0,0) */
    intptr_t stride_c; int rows_c; int margin_y_c;      /* each chroma buffer; rows_c = 0: luma only */
    int ctu_rows;
    int slots;                                          /* views resident (device + pinned host memory) at once */
    int pictures;                                       /* source pictures resident at once (0 = slots) */
    int device_plus_1;                                  /* 0: the calling thread's current device; d + 1: device d (see x265hip_me_stream_params) */
} x265hip_phase_stream_params;
typedef struct x265hip_phase_stream_stats_t
{
    uint64_t opened, completed, bands, failed;
    uint64_t us_busy;
    uint64_t bytes_downloaded, bytes_uploaded, bytes_per_picture;
    uint64_t weighted_views, lines_weighted;
} x265hip_phase_stream_stats_t;
int  x265hip_phase_stream_create(x265hip_phase_stream** out, const x265hip_phase_stream_params* p);
void x265hip_phase_stream_destroy(x265hip_phase_stream* s);
/* CTU rows [ctu_row0, ctu_row0 + ctu_rows) of picture `key` are final in the three whole-plane buffers (cb / cr NULL when rows_c = 0);
 * copied before the call returns.  X265HIP_EBUSY: every picture entry still feeds a view. */
int  x265hip_phase_stream_picture_rows(x265hip_phase_stream* s, uint64_t key, const void* luma_buf, const void* cb_buf, const void* cr_buf,
                                       int ctu_row0, int ctu_rows);
/* `slot` becomes the view of picture `key`; w = NULL: as reconstructed, else plane c (0 Y, 1 Cb, 2 Cr) is weighted with w[c] (the
 * arguments of primitives.weight_pp, see x265hip_weight) where bit c of planes_weighted is set.  -> the slot's new GENERATION (> 0) */
int  x265hip_phase_stream_view_open(x265hip_phase_stream* s, int slot, uint64_t key, const x265hip_weight* w, unsigned planes_weighted);
/* round-3 entries, kept: the producer opens a slot for an (anonymous) picture and feeds the slot */
int  x265hip_phase_stream_open(x265hip_phase_stream* s, int slot);                       /* -> the slot's new GENERATION (> 0) */
int  x265hip_phase_stream_rows(x265hip_phase_stream* s, int slot, int generation, const void* luma_buf, const void* cb_buf, const void* cr_buf,
                               int ctu_row0, int ctu_rows);
const void* x265hip_phase_stream_planes(x265hip_phase_stream* s, int slot, int plane);   /* 0: 15 luma planes, 1 / 2: 63 Cb / Cr planes */
const volatile uint64_t* x265hip_phase_stream_progress(x265hip_phase_stream* s, int slot);
int  x265hip_phase_stream_stats(x265hip_phase_stream* s, x265hip_phase_stream_stats_t* st);


/* ---------------------------------------------------------------------------------------------------------------
 * SUB-SAMPLE COST TABLES (round 6, csrc/cost_stream.hip): the sub-sample half of MotionEstimate::motionEstimate served as VALUES.
 * After its integer search the reference refines every PU's vector with SATD comparisons at half- and quarter-sample positions
 * (motion.cpp:1456-1561: the SubpelWorkload rows of :48-58; each comparison = MotionEstimate::subpelCompare, :1571-1664 - luma_hpp /
 * luma_vpp / luma_hvpp into a scratch block, pu[].satd against the source block and, from --subme 3 on (bChromaSATD, :212), the 4-tap
 * chroma filters + chroma[].pu[].satd of Cb and Cr).  Every one of those costs is a pure function of (source block, reference plane,
 * quarter-sample vector), and the positions a refinement can visit from an integer vector v form a small fixed set: square1 steps of
 * the workload row - 49 positions for --subme 3, 85 for --subme 4 (x265hip_cost_positions).  So, per (source picture, reference view):
 *   1. where did each CTU go?  the minima-only exhaustive search of +-centre_range on SAD alone (x265hip_me_fullsearch) -> the centre;
 *   2. SAD rasters of +-window around the centre for the 85 square PUs (x265hip_me_fullsearch, surfaces);
 *   3. x265hip_cost_candidates: for EVERY PU shape of the list (x265hip_cost_pu_rect: the squares, the 2NxN / Nx2N halves, the AMP
 *      parts that are unions of 8x8 blocks - summed from the square rasters) the `candidates` displacements of smallest SAD;
 *   4. x265hip_cost_tables: around each candidate the SATD (luma, or luma + Cb + Cr) of every position of the set, read from the
 *      view's fractional-phase planes (x265hip_phase_planes: the same samples the three luma / four chroma filter calls produce).
 * What travels to the host is one RECORD per (CTU, PU, candidate): { int16 mvx, mvy (integer displacement; mvx = -32768: no record);
 * uint32 base; uint16 delta[positions] }, cost(position i) = base + delta[i], delta 65535 = not representable (the host computes
 * that one itself; sad_costs = 1 appends { uint32 base; uint16 delta[positions] } of the SAD-typed comparisons); x265hip_cost_record_bytes() apart, PU-major,
 * candidates of a PU adjacent; x265hip_cost_ctu_bytes() per CTU.
 * A host stub of subpelCompare answers (cmp == satd, qmv - 4 * (mvx, mvy) inside the position set) from the record and everything
 * else with the host's own primitives - the same integers either way, so the bitstream cannot change.
 * The ROW-GRANULAR service follows the reference's frame threads like x265hip_me_stream / x265hip_phase_stream: pictures (source AND
 * reconstructed, three planes) arrive by key, reconstructed ones CTU row by CTU row where the reference raises m_reconRowFlag
 * (framefilter.cpp:664); a pair (source, reference[, weights]) is opened by the first search that refers to it; reference VIEWS
 * (weighted planes + their 15 + 2 x 63 phase planes) live on the DEVICE only and are shared by the pairs on the same (picture,
 * weights); CTU row r of a pair is computed when its source row and reference rows <= r + 2 are there (the host's own consumers
 * wait for the same rows, frameencoder.cpp:161-164), and ready[r] carries the pair's generation once the row's records have landed
 * in pinned host memory - check it before AND after reading. */
int x265hip_cost_pu_count(int shapes);                              /* shapes 0: the 85 squares; 1: + 2NxN / Nx2N (169); 2: + AMP of the 32 / 64 CUs (209) */
int x265hip_cost_pu_rect(int shapes, int pu, int rect[4]);         /* x, y, width, height in luma samples inside the CTU; PUs [0, 85) in the surfaces' order */
int x265hip_cost_positions(int subme, int8_t* xy, int max_positions);   /* -> count; xy[2 i], xy[2 i + 1] = quarter-sample offset of position i, raster order (y, then x) */
int x265hip_cost_record_bytes(int subme, int sad_costs);
size_t x265hip_cost_ctu_bytes(int subme, int shapes, int candidates, int sad_costs);
typedef struct x265hip_cost_candidates_params
{
    int nctu;
    int window;                         /* the surfaces hold (2 window + 1)^2 displacements per CTU, X265HIP_SURF_I32 records */
    const int32_t* surf;                /* DEVICE */
    const int16_t* centres;             /* DEVICE int16 [ctu][2] the surfaces were centred on, or NULL = (0, 0) */
    int shapes, candidates;             /* candidates: 1 or 2 displacements of smallest cost per PU (ties: raster order, like the search) */
    int16_t* cand;                      /* DEVICE out, int16 [ctu][pu][candidate][2] = absolute integer displacement */
    /* optional, DEVICE uint16 [2 window + 1]: vector cost of a displacement component relative to the window's centre; the ranking key is
     * sad + mv_cost[col] + mv_cost[row] - what the host's search minimises (SAD + mvcost(mv - mvp), motion.cpp:246-328) with the CTU's own
     * displacement standing in for the predictor.  NULL = SAD alone */
    const uint16_t* mv_cost;
} x265hip_cost_candidates_params;
int x265hip_cost_candidates(const x265hip_cost_candidates_params* p, void* stream);
typedef struct x265hip_cost_tables_params
{
    int depth;
    int width;                          /* luma samples, whole CTUs */
    intptr_t stride;   int margin_x, margin_y;          /* PicYuv layout of every luma plane below */
    intptr_t stride_c; int margin_y_c;                  /* 4:2:0 chroma planes (same margin_x); ignored when chroma = 0 */
    int ctu_row0, ctu_rows;             /* the band */
    const void* fenc[3];                /* DEVICE, allocation starts of the source picture's planes */
    const void* ref[3];                 /* DEVICE, allocation starts of the reference's (weighted) planes */
    const void* phases[3];              /* DEVICE, the 15 luma / 63 Cb / 63 Cr phase planes of ref (x265hip_phase_planes), plane_bytes[_c] apart */
    size_t plane_bytes, plane_bytes_c;
    int shapes, candidates, subme, chroma;
    const int16_t* cand;                /* DEVICE, the band's first CTU first */
    void* tables;                       /* DEVICE out, the band's first CTU first */
    int sad_costs;                      /* 1: every record carries a second { uint32 base; uint16 delta[positions] } behind the first (4-byte aligned): the costs of the
                                         * SAD-typed comparisons - subpelCompare(ref, mv, sad) of the search's predictor candidates, motion.cpp:773-812 - luma SAD + (chroma = 1)
                                         * the SATD of Cb and Cr, which subpelCompare adds whatever the luma comparison is (:1601-1661) */
} x265hip_cost_tables_params;
int x265hip_cost_tables(const x265hip_cost_tables_params* p, void* stream);

typedef struct x265hip_cost_stream x265hip_cost_stream;
typedef struct x265hip_cost_stream_params
{
    int depth;
    int width, height;                  /* luma samples, whole CTUs */
    intptr_t stride;   int margin_x, margin_y;
    intptr_t stride_c; int margin_y_c;  /* 4:2:0 chroma planes; stride_c = 0: luma only (chroma must be 0 then) */
    int centre_range;                   /* step 1 (the host's merange; 0 = windows around (0, 0)) */
    int window;                         /* step 2 */
    int candidates, shapes, subme, chroma;
    int sad_costs;                      /* see x265hip_cost_tables_params */
    int slots;                          /* pairs resident in pinned host memory at once */
    int pictures;                       /* pictures (source + reconstructed) resident on the device at once */
    int views;                          /* reference views resident on the device at once (each 15 luma + 126 chroma planes) */
    int band_rows;                      /* most CTU rows per launch chain (0 = 8) */
    int device_plus_1;                  /* 0: the calling thread's current device; d + 1: device d */
} x265hip_cost_stream_params;
typedef struct x265hip_cost_stream_stats_t
{
    uint64_t pairs_opened, pairs_completed, bands, rows_served, rows_uploaded, failed, stale_pairs;
    uint64_t views_opened, views_shared, lines_weighted;
    uint64_t us_busy, bytes_downloaded, bytes_uploaded, table_bytes;
} x265hip_cost_stream_stats_t;
int  x265hip_cost_stream_create(x265hip_cost_stream** out, const x265hip_cost_stream_params* p);
void x265hip_cost_stream_destroy(x265hip_cost_stream* s);
/* CTU rows [ctu_row0, ctu_row0 + ctu_rows) of picture `key` are final in the three buffers (whole allocated planes; cb / cr NULL when
 * stride_c = 0); copied before the call returns.  X265HIP_EBUSY: no picture entry free. */
int  x265hip_cost_stream_picture_rows(x265hip_cost_stream* s, uint64_t key, const void* luma_buf, const void* cb_buf, const void* cr_buf, int ctu_row0, int ctu_rows);
/* -> the slot's new GENERATION (> 0).  w = NULL: the reference as reconstructed; otherwise plane c is weighted with w[c] (weight_pp
 * arguments, round / shift including the 14 - depth correction) when bit c of planes_weighted is set.
 * mv_cost (HOST uint16 [2 window + 1], copied; NULL = rank by SAD alone): the host's vector cost of an integer displacement component
 * i - window relative to its predictor (BitCost::mvcost's table at the slice's QP, bitcost.h:45: m_cost[4 (i - window)]) - the candidates
 * are then the minima of SAD + cost, what the host's own search minimises, with each CTU's displacement standing in for the predictor */
int  x265hip_cost_stream_pair_open(x265hip_cost_stream* s, int slot, uint64_t fenc_key, uint64_t ref_key, const x265hip_weight* w, unsigned planes_weighted,
                                   const uint16_t* mv_cost);
const void* x265hip_cost_stream_tables(x265hip_cost_stream* s, int slot);             /* pinned host memory, CTU-major */
const volatile int* x265hip_cost_stream_ready(x265hip_cost_stream* s, int slot);      /* int [height / 64] */
int  x265hip_cost_stream_stats(x265hip_cost_stream* s, x265hip_cost_stream_stats_t* st);


/* ---------------------------------------------------------------------------------------------------------------
 * Multi-GPU seam (csrc/recon_publish.hip): hand a finished band of reconstructed CTU rows - Y, Cb, Cr with their margins - from the
 * GPU that produced it to the GPU(s) whose in-flight pictures reference it, where the reference raises m_reconRowFlag
 * (encoder/framefilter.cpp:664; consumers wait in encoder/frameencoder.cpp:852-868).  One process per GPU; `comm` is the host's
 * ncclComm_t (RCCL, resolved at run time).  Every rank of the communicator (peer < 0: broadcast from `root`) or the two ranks
 * involved (peer >= 0: ncclSend on `root`, ncclRecv on `peer`) make the same call with their own plane pointers; the three planes
 * travel in one RCCL group on `stream` - give it a copy stream so the next band's kernels overlap.
 *   plane[]  : ALLOCATION STARTS of the padded planes (PicYuv layout); plane[1] / plane[2] NULL = luma only
 *   height   : picture height in luma samples (whole CTUs); the first band also carries the top margin rows, the last band the bottom ones */
typedef struct x265hip_recon_publish_params
{
    void* comm;                     /* ncclComm_t */
    int rank, root, peer;           /* this process' rank; the producer; the one consumer, or -1 = everyone (broadcast) */
    int depth;
    void* plane[3];
    intptr_t stride, stride_c;
    int margin_y, margin_y_c;
    int height;
    int ctu_row0, ctu_rows;
} x265hip_recon_publish_params;
int x265hip_recon_publish_rows(const x265hip_recon_publish_params* p, void* stream);
/* the producer's band to SEVERAL consumers - a picture that is the reference of several in-flight pictures (preset slow: 4 references and
 * B pictures) - as ONE group of point-to-point sends (xGMI is point to point: k consumers = k links, no ring); a consumer calls it with
 * rank != root and receives from root (peers ignored) */
int x265hip_recon_publish_rows_to(const x265hip_recon_publish_params* p, int npeers, const int* peers, void* stream);
/* Communicator plumbing for a host without an RCCL binding of its own (one process per GPU): rank 0 makes the 128-byte id
 * (ncclGetUniqueId), the host ships it to the other ranks by whatever it has, every rank joins on its CURRENT device (ncclCommInitRank).
 * RCCL is resolved at run time (dlopen): a single-GPU host never loads it. */
int x265hip_comm_unique_id(void* id128);
int x265hip_comm_init(void** comm, int nranks, const void* id128, int rank);
int x265hip_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* X265HIP_H */
