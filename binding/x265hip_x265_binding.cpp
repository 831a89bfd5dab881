/* binding/x265hip_x265_binding.cpp - the REFERENCE-SIDE BINDING of libx265hip's consumer services (INTEGRATION.md sections 3c - 3g, 4): what a
 * maintainer of x265 3.5 adds to the encoder so that MotionEstimate::motionEstimate / subpelCompare, CostEstimateGroup::estimateFrameCost,
 * LookaheadTLD::lowresIntraEstimate / calcAdaptiveQuantFrame, weightAnalyse and FrameFilter::processPostRow consume the device's results.  It is written
 * against the reference's PUBLIC members only and compiled with the reference's own headers; it holds no reference source text.
 *
 * TWO WAYS TO ATTACH IT
 *   (a) source patch: seven `#if ENABLE_HIP_PRIMITIVES` hooks - motion.cpp:739 / :1571, slicetype.cpp:3115 / :696 / :444,
 *       framefilter.cpp:657, weightPrediction.cpp:222 - rename the reference's bodies and export them as the extern "C" x265ref_orig_* trampolines declared
 *       below; this file then supplies the public symbols.  Build it with the encoder (X265HIP_BINDING_TEST_HOOKS unset = 0).
 *   (b) this repository's TEST RIG (oracle/Makefile): no reference source is touched - objcopy renames / weakens the same seven symbols in the reference's
 *       compiled objects (--redefine-sym / --weaken-symbol + --add-symbol x265ref_orig_*), and this file is compiled with -DX265HIP_BINDING_TEST_HOOKS=1
 *       into oracle/_ref/libx265ref<depth>_seam.so.  tests/test_seam_cpu.py, tests/test_gpu_seam.py and bench.py's encoder legs run on that library.
 * What X265HIP_BINDING_TEST_HOOKS adds is measurement and fixture code only (stage timers for tools/encoder_profile.py, the integer-vector predictor probe,
 * the X265REF_AQ_DUMP / X265REF_WA_DUMP fixture writers, the *_profiled table fillers); the binding proper - lookup stubs, the thread-local search context, the
 * size and hit-rate gates, slot bookkeeping, the providers' plumbing and the in-flight verification switch a maintainer wants while bringing it up - is the rest.
 *
 * Mechanism:
 *   * the reference's MotionEstimate::motionEstimate answers to x265ref_orig_motionEstimate (hook (a) or objcopy (b)) and this file supplies
 *     MotionEstimate::motionEstimate: a wrapper that
 *     identifies (source picture, reference picture, CTU, PU) for the calling worker thread, makes sure the pair's SAD surfaces
 *     have been requested from the provider (ONE exhaustive-search launch per pair, x265hip_me_cache_submit), publishes a
 *     thread-local lookup context and then runs the reference's own, untouched search (all --me methods, all its quirks).
 *   * the table filler x265ref_seam_fill_table() replaces pu[].sad / sad_x3 / sad_x4 (primitives.h:247-249) of the partitions
 *     that are unions of 8x8 blocks with LOOKUP stubs: inside a wrapped search they translate the reference pointer into a
 *     displacement and read the SAD from the surfaces (summing the square sub-blocks of rectangular / asymmetric partitions);
 *     anything else - other callers, a CTU row whose surfaces have not arrived, displacements outside the window - goes to the
 *     host's original primitive.  Both routes return the same integers, so the bitstream cannot change; with
 *     X265REF_SEAM_VERIFY=1 every lookup is checked against the original primitive on the spot.
 * Two provider flavours.  PICTURE-GRANULAR (x265hip_me_cache / x265hip_phase_cache): a reference picture must be complete when the
 * first PU of the next picture searches it, i.e. --frame-threads 1; with more frame threads these seams step aside.
 * ROW-GRANULAR (x265hip_me_stream / x265hip_phase_stream, round 3): this file also takes over FrameFilter::processPostRow - the
 * function that raises Frame::m_reconRowFlag[row] (framefilter.cpp:664) - and hands every finished CTU row of a reconstructed picture
 * to the providers right after the reference's own body; pairs are opened by the first search of a (picture, reference) and searched
 * on the device row by row behind the producer, so the seams serve under the reference's own frame threads (-F 5 on 16 cores).
 * WEIGHTED references (x265's default --weightp / --weightb: MotionReference::applyWeight materialises primitives.weight_pp of every
 * finished reconstructed row, reference.cpp:119-178, frameencoder.cpp:865-866) are served by the row-granular providers too (round 4):
 * the weighted plane is the reconstructed plane weighted sample by sample, margins included, so the provider gets the weight_pp
 * arguments with the pair / view and weights the rows it already holds on the device; the picture-granular providers still step aside.
 * WHOLE FUNCTIONS (end of round 4): LookaheadTLD::calcAdaptiveQuantFrame and the frame encoder's weightAnalyse are defined here too (the reference's
 * bodies answer to x265ref_orig_*), each one provider call (x265hip_aq_frame_host / x265hip_weight_analyse_host or the oracle's restatements) whose
 * results - with verify - are compared with what the reference's own function then computes for the same picture / slice; X265REF_AQ_DUMP /
 * X265REF_WA_DUMP write the inputs and the reference's answers for tools/gen_weight_golden.py.
 * HOST-ONLY CONTROL: x265ref_split_fill_table = the C table with sad_x3 / sad_x4 answered by N single SADs (what the lookup stubs do on a miss): g++
 * vectorises the reference's single-reference SAD loop and not the multi-reference ones, so this alone is faster than the C table, and encoder legs
 * must be compared with it to see what the services contribute.
 *
 * The provider is a table of C function pointers with the signatures of x265hip_me_cache_submit / _surface / _ready
 * (include/x265hip.h), so the GPU library plugs in directly; the CPU-only tests plug in the oracle's exhaustive search instead. */
#include "common.h"
#include "primitives.h"
#include "constants.h"
#include "picyuv.h"
#include "frame.h"
#include "framedata.h"
#include "slice.h"
#include "search.h"
#include "motion.h"
#include "reference.h"
#include "lowres.h"
#include "slicetype.h"
#include "framefilter.h"

#include <atomic>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>
#include <time.h>
#include <x86intrin.h>

using namespace X265_NS;

#ifndef X265HIP_BINDING_TEST_HOOKS
#define X265HIP_BINDING_TEST_HOOKS 0          /* 1: the test rig's measurement / fixture code is compiled in (see the header) */
#endif
#if X265HIP_BINDING_TEST_HOOKS
#define SEAM_PROF_ON (gp.on)                   /* stage timers of tools/encoder_profile.py --seams */
#define SEAM_PROBE_ON (gpp.on)                 /* integer-vector predictor probe (x265ref_predict_probe) */
#else
#define SEAM_PROF_ON false
#define SEAM_PROBE_ON false
#endif

extern "C" int x265ref_orig_motionEstimate(MotionEstimate* self, ReferencePlanes* ref, const MV* mvmin, const MV* mvmax, const MV* qmvp,
                                           int numCandidates, const MV* mvc, int merange, MV* outQMv, uint32_t maxSlices, pixel* srcReferencePlane);

extern "C" void x265ref_orig_processPostRow(FrameFilter* self, int row);
extern "C" int x265ref_orig_subpelCompare(MotionEstimate* self, ReferencePlanes* ref, const MV* qmv, pixelcmp_t cmp);
extern "C" int64_t x265ref_orig_estimateFrameCost(CostEstimateGroup* self, LookaheadTLD* tld, int p0, int p1, int b, bool bIntraPenalty);
extern "C" void x265ref_orig_lowresIntraEstimate(LookaheadTLD* self, Lowres* fenc, uint32_t qgSize);
extern "C" void x265ref_orig_calcAdaptiveQuantFrame(LookaheadTLD* self, Frame* curFrame, x265_param* param);
extern "C" void x265ref_orig_weightAnalyse(Slice* slice, Frame* frame, x265_param* param);          /* references are pointers at the ABI level */
#if X265HIP_BINDING_TEST_HOOKS
extern "C" int x265ref_profile_fill_table(void* table, size_t bytes, int depth);          /* oracle/ref_profile.cpp: cycle-counting thunks */
#endif


namespace {

/* ---- the lookahead seam: CostEstimateGroup::estimateFrameCost's estimateCUCost loop as ONE call of the provider ------------------
 * x265hip_lowres_cost_host_params (include/x265hip.h), mirrored field by field */
struct LaHostParams
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
    const int32_t* inv_qscale;
    const uint16_t* cost_q;  int cost_q_half;
    int bframe_bias;
    int do_search[2];
    int32_t* mvs[2];  int32_t* mv_costs[2];
    uint16_t* lowres_costs;  int32_t* row_satds;  int64_t* frame;
    uint64_t plane_key_cur, plane_key_ref, plane_key_ref1, plane_key_ref_bi;
};
typedef int (*la_host_fn)(const LaHostParams*);
/* the oracle's CPU restatement (x265oracle_lowres_cost_wp_d<depth>): the checker-only provider of the GPU-less tests */
typedef int (*la_oracle_fn)(const pixel* cur, const pixel* const* refs0, const pixel* const* refs1, intptr_t stride, int widthInCU, int heightInCU,
                            const uint16_t* cost, int qoff, const int32_t* intraCost, const int32_t* invQscale, const int* doSearch, int bFrameBias,
                            int32_t* mvs0, int32_t* mvCosts0, int32_t* mvs1, int32_t* mvCosts1, uint16_t* lowresCosts, int32_t* rowSatds,
                            int64_t* frame, const pixel* const* refs0Bi);
struct LaIntraHostParams          /* x265hip_lowres_intra_host_params */
{
    int depth;
    intptr_t stride;
    int width_in_cu, height_in_cu;
    int lines, margin_x, margin_y;
    const void* plane;
    int intra_penalty;
    int32_t* intra_cost; uint8_t* intra_mode; uint16_t* lowres_costs;
    uint64_t plane_key;
};
typedef int (*la_intra_host_fn)(const LaIntraHostParams*);
typedef void (*la_intra_oracle_fn)(const pixel* plane, intptr_t stride, int widthInCU, int heightInCU, int intraPenalty,
                                   int32_t* intraCost, uint8_t* intraMode, uint16_t* lowresCosts, int nthreads);
struct LookaheadSeam
{
    bool enabled = false;
    la_host_fn host = NULL;
    la_oracle_fn oracle = NULL;
    la_intra_host_fn intraHost = NULL;
    la_intra_oracle_fn intraOracle = NULL;
    std::atomic<uint64_t> served{0}, passed{0}, failed{0}, intraServed{0}, mismatches{0};
    uint64_t instance = 0;          /* bumped by every configure: a frame number names a picture's content only within one encode */
    /* A frame cost estimate is one latency-bound walk on the device (1.3 ms at 4K against 49 ms of one host thread; 2.9 ms against a few ms
     * at 1080p, where the seam LOSES 10 %, profiles/r03_encoder_other_configs.txt): pictures with fewer 8x8 lowres blocks than this go to
     * the reference's own loop.  16384 lies between 1080p (8160) and 4K (32400); x265ref_lookahead_seam_min_blocks overrides (tests: 0). */
    int minBlocks = 16384;
    std::atomic<uint64_t> gated{0};
} gla;

/* x265hip_aq_frame_host_params (include/x265hip.h), mirrored field by field */
struct AqFrameHostParams
{
    int depth;
    const void* y; const void* cb; const void* cr;
    intptr_t stride, stride_c;
    int width, height, qg_size, aq_mode;
    double aq_strength;
    int width_in_cu, height_in_cu;
    int normalise_wp;
    double* qp_aq_offset; double* qp_cutree_offset; int32_t* inv_qscale; int32_t* inv_qscale_8x8; uint32_t* energy;
    uint64_t* wp_sum; uint64_t* wp_ssd;
};
typedef int (*aq_host_fn)(const AqFrameHostParams*);
/* x265oracle_aq_frame_d<depth> (oracle/x265_oracle_pipeline3.c) */
typedef void (*aq_oracle_fn)(const pixel* y, const pixel* cb, const pixel* cr, intptr_t stride, intptr_t strideC, int width, int height, int qgSize, int aqMode,
                             double aqStrength, int weightp, uint32_t* energy, double* qpAqOffset, int32_t* invQscale, uint64_t* wpSum, uint64_t* wpSsd);
struct AqSeam
{
    bool enabled = false, verify = false;
    aq_host_fn host = NULL;
    aq_oracle_fn oracle = NULL;
    int minBlocks = 0;
    std::atomic<uint64_t> served{0}, passed{0}, failed{0}, mismatches{0}, gated{0};
} gaq;

/* x265hip_weight_analyse_ref / x265hip_weight_analyse_host_params (include/x265hip.h), mirrored field by field */
struct WaRef { const void* lowres[4]; const void* cb; const void* cr; const int32_t* mvs; uint64_t wp_ssd[3], wp_sum[3]; uint64_t plane_key; };
struct WaHostParams
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
    WaRef ref[2];
    int32_t* weights; int32_t* denoms;
};
typedef int (*wa_host_fn)(const WaHostParams*);
/* x265oracle_wa_list / x265oracle_weight_analyse_d<depth> (oracle/x265_oracle_pipeline7.c) */
struct WaOracleList { const pixel* lowres[4]; const pixel* cb; const pixel* cr; const int32_t* mvs; uint64_t wp_ssd[3], wp_sum[3]; };
typedef void (*wa_oracle_fn)(const pixel* fencLowres, intptr_t lowresStride, int lowresWidth, int lowresLines, const pixel* fencCb, const pixel* fencCr, intptr_t strideC,
                             int picWidth, int picHeight, const int32_t* intraCost, const uint64_t* fencSsd, const uint64_t* fencSum, int nlists, const WaOracleList* lists,
                             pixel* scratch, size_t scratchHalf, int32_t* out, int32_t* denoms);
struct WaSeam
{
    bool enabled = false, verify = false;
    wa_host_fn host = NULL;
    wa_oracle_fn oracle = NULL;
    int minBlocks = 0;
    std::atomic<uint64_t> served{0}, passed{0}, failed{0}, mismatches{0}, gated{0}, weighted{0};
} gwa;

/* measurement aid (tools/encoder_profile.py --seams): cycles inside the wrapped stages, next to ref_profile.cpp's per-family thunks */
struct SeamProf
{
    bool on = false;
    std::atomic<uint64_t> meCyc{0}, meCalls{0}, subCyc{0}, subCalls{0}, laCyc{0}, laCalls{0}, rowCyc{0}, rows{0}, lookCyc{0}, lookups{0}, ctxCyc{0};
} gp;
struct ProfScope
{
    std::atomic<uint64_t>& cyc; std::atomic<uint64_t>& n; uint64_t t0;
    ProfScope(std::atomic<uint64_t>& c, std::atomic<uint64_t>& k) : cyc(c), n(k), t0(SEAM_PROF_ON ? __rdtsc() : 0) {}
    ~ProfScope() { if (SEAM_PROF_ON) { cyc.fetch_add(__rdtsc() - t0, std::memory_order_relaxed); n.fetch_add(1, std::memory_order_relaxed); } }
};

struct MeCostProbe : public MotionEstimate { const uint16_t* costCentre() const { return m_cost; } };

enum { SURF_I32 = 0, SURF_PACKED = 1, SURF_PACKED_T = 2, GROUP_I32 = 1360, GROUP_PACKED = 720, MAX_PARTS = 6, MAX_SLOTS = 64 };

struct Provider
{
    void* ctx;
    int (*submit)(void* ctx, int slot, const void* fenc_buf, uint64_t fenc_key, const void* ref_buf);
    int (*submit_batch)(void* ctx, int n, const int* slots, const void* fenc_buf, uint64_t fenc_key, const void* const* ref_bufs, int* generations);
    const void* (*surface)(void* ctx, int slot);
    const volatile int* (*ready)(void* ctx, int slot);
    /* row-granular flavour (x265hip_me_stream_picture_rows / _pair_open signatures); streamed = both are set */
    int (*picture_rows)(void* ctx, uint64_t key, const void* buf, int ctu_row0, int ctu_rows);
    int (*pair_open)(void* ctx, int slot, uint64_t fenc_key, uint64_t ref_key);
    int (*pair_open_w)(void* ctx, int slot, uint64_t fenc_key, uint64_t ref_key, const void* weight);      /* x265hip_me_stream_pair_open_weighted; NULL: weighted references pass */
    const int16_t* (*centres)(void* ctx, int slot);      /* x265hip_me_stream_centres; NULL or a NULL result: windows centred on (0, 0) */
    bool streamed;
    int layout;                   /* 0: records; 1: PU-major planes (X265HIP_STREAM_PLANES) */
    int min_level;                /* 1: records hold the 16x16 / 32x32 / 64x64 levels only (X265HIP_SURF_TAIL_BYTES_*) */
    int range, surf_format, slots;
    int width, height;            /* whole CTUs */
    intptr_t stride;
    int margin_x, margin_y;
    int min_pu;                   /* serve partitions whose smaller side is >= min_pu (8, 16, 32 or 64) */
};

/* the arguments of primitives.weight_pp as reference.cpp:154 passes them = x265hip_weight; present = 0: the plane as reconstructed */
struct Wt { int w0, round, shift, offset; int present; };
inline bool same_wt(const Wt& a, const Wt& b) { return a.present == b.present && (!a.present || (a.w0 == b.w0 && a.round == b.round && a.shift == b.shift && a.offset == b.offset)); }
inline Wt plane_weight(const ReferencePlanes* ref, int c)
{
    Wt w = { 0, 0, 0, 0, 0 };
    if (ref->isWeighted && ref->reconPic && ref->fpelPlane[c] != ref->reconPic->m_picOrg[c])
    {
        const int correction = IF_INTERNAL_PREC - X265_DEPTH;
        w.w0 = ref->w[c].weight; w.round = ref->w[c].round << correction; w.shift = ref->w[c].shift + correction; w.offset = ref->w[c].offset; w.present = 1;
    }
    return w;
}
struct Pair { int fencPoc; const PicYuv* rec; int recPoc; int slot; int gen; bool used; int encodeOrder; Wt wt; };
struct FencStaged { int poc; int encodeOrder; bool used; };

struct Seam
{
    bool enabled = false, verify = false, wait = false;
    bool batchMisses = false;     /* X265REF_SEAM_BATCH_MISS=1 (A/B): a sad_x3 / sad_x4 without a hit goes to the table's own batched primitive */
    Provider p;
    int nc, ng, groupBytes, ctusW, pitch;
    uint64_t strideMagic;               /* ceil(2^40 / stride): row = (t * magic) >> 40 for every offset a lookup can see */
    size_t ctuBytes;
    std::mutex mu;
    Pair pairs[MAX_SLOTS];
    FencStaged fencs[32];               /* streamed: source pictures already handed to the provider */
    uint64_t instance = 0;              /* bumped by every configure: a POC names a picture's content only within one encode */
    std::atomic<int> epoch{0};          /* bumped on every (re)assignment of a slot: invalidates the thread-local pair caches */
    pixelcmp_t sad[NUM_PU_SIZES];
    pixelcmp_x3_t sad_x3[NUM_PU_SIZES];
    pixelcmp_x4_t sad_x4[NUM_PU_SIZES];
    std::atomic<uint64_t> hits{0}, outside{0}, notReady{0}, meCalls{0}, meServed{0}, submits{0}, mismatches{0}, noSlot{0}, foreign{0};
    std::atomic<uint64_t> rowsPublished{0}, rowsRefused{0}, torn{0}, weightedPairs{0}, weightedHits{0}, weightedCalls{0}, saturated{0};
} g;
enum { MAX_SLOTS_STREAMED = 64 };
/* The search seams move whole surfaces / phase planes per (picture, reference) to serve a search that, on small pictures and fast presets,
 * asks for little: at 1080p --preset medium they cost 4 - 5 % (profiles/r04_encoder_legs.txt) where 4K gains 42 - 90 %.  Pictures of fewer
 * CTUs than this are left alone by BOTH search seams (the lookahead seam has its own gate, gla.minBlocks); 1000 lies between 1080p (510)
 * and 4K (2040).  x265ref_seam_min_ctus overrides (tests on small pictures: 0). */
int g_minCtus = 1000;
bool g_gated = false;
/* The HIT-RATE gate of the SAD seam (round 5; round-4 verdict, next 4): on content whose searches leave the windows - a fade: the unweighted references' star
 * searches wander over the whole +-57 area, 0.11 - 0.17 of the lookups hit - the seam costs more than it gives (one box, interleaved: 4.46 fps without it, 4.10 - 4.20
 * with it, profiles/r05_seam_matrix.txt).  Every g_gateWindow lookups the share served in that window is looked at; below g_gatePct % no NEW pairs are opened
 * (open ones keep serving) until a probe - every 8th source picture opens its pairs regardless - shows the windows being hit again.  0 = off. */
uint64_t g_gateWindow = 2000000;
int g_gatePct = 50;
struct { uint64_t h0 = 0, o0 = 0; bool closed = false; std::atomic<uint64_t> skipped{0}, closings{0}; } g_gate;
inline uint64_t pic_key(uint64_t instance, int poc, int isRecon) { return (instance << 40) | ((uint64_t)(uint32_t)poc << 1) | (uint64_t)isRecon; }

struct Part { uint32_t off; uint32_t wide; };      /* records: byte offset of entry [z][0] inside a group record; planes: of the PU's raster inside the CTU; wide = 32-bit entries */

struct Ctx
{
    bool valid;
    int part;                 /* LumaPU enum of the PU being searched */
    const pixel* fenc;
    const pixel* fref0;       /* reference pointer of the window's centre displacement ((0,0) until the row's centre is known) */
    const int16_t* centre;    /* this CTU's entry of the slot's centres, or NULL */
    bool centred;
    intptr_t stride;
    ptrdiff_t bias;           /* range * stride + range */
    const uint8_t* ctuBase;   /* surfaces of this CTU */
    const volatile int* ready;
    int ctuRow, gen, nparts;
    bool weighted;
    Part parts[MAX_PARTS];
    uint64_t hits, outside, notReady, saturated, cyc;
};
thread_local Ctx t_ctx;
/* MEASUREMENT MODE (x265ref_predict_probe, off by default): how well would a device-side table of sub-sample costs "around the integer vector the
 * device predicts" be addressed?  At the first sub-sample comparison of a search - the integer vector the host's search ended on - the rank of that vector among the
 * window's SADs alone (what the device knows: no motion-vector cost, no predictors) is taken from the rasters the SAD seam already holds. */
struct { bool on = false; std::atomic<uint64_t> total{0}, noCtx{0}, outside{0}, top1{0}, top2{0}, top4{0}, top8{0}; } gpp;
struct TlsPair { const PicYuv* rec; int recPoc; int slot; int gen; Wt wt; };
thread_local struct { int fencPoc; int epoch; int n; TlsPair e[8]; } t_pairs = { -0x7fffffff, -1, 0, {} };

/* width / height per LumaPU enum (primitives.h:41-55); the reference never sets MotionEstimate::blockheight (motion.cpp:178,217) */
const int PU_DIMS[NUM_PU_SIZES][2] = { {4,4},{8,8},{16,16},{32,32},{64,64},{8,4},{4,8},{16,8},{8,16},{32,16},{16,32},{64,32},{32,64},
                                       {16,12},{12,16},{16,4},{4,16},{32,24},{24,32},{32,8},{8,32},{64,48},{48,64},{64,16},{16,64} };

/* decompose the PU rectangle (CTU-relative, multiples of 8) into the squares the surfaces hold, quadtree order */
bool decompose(int px, int py, int w, int h, int bx, int by, int size, Ctx& c)
{
    const int x0 = bx > px ? bx : px, x1 = (bx + size < px + w) ? bx + size : px + w;
    const int y0 = by > py ? by : py, y1 = (by + size < py + h) ? by + size : py + h;
    if (x0 >= x1 || y0 >= y1) return true;                                 /* no overlap */
    if (x1 - x0 == size && y1 - y0 == size)                                /* fully inside */
    {
        if (c.nparts == MAX_PARTS) return false;
        const int level = size == 8 ? 0 : size == 16 ? 1 : size == 32 ? 2 : 3;
        if (level < g.p.min_level) return false;                           /* that level was not downloaded (min_level 1: no 8x8 rasters; 2: no 16x16 ones either) */
        const int ux = bx / size, uy = by / size;
        int z = 0;
        for (int b = 0; b < 3; b++) z |= ((ux >> b) & 1) << (2 * b) | ((uy >> b) & 1) << (2 * b + 1);
        Part& q = c.parts[c.nparts++];
        if (g.p.layout)
        {
            /* x265hip_stream_planes_pu_offset: uint16 rasters of the 8x8 (min_level 0) and 16x16 PUs, then uint32 rasters of 32x32 and 64x64 */
            const size_t ps = (size_t)g.nc * g.pitch * 2, pw = (size_t)g.nc * g.pitch * 4, n0 = g.p.min_level ? 0 : 64, n1 = g.p.min_level > 1 ? 0 : 16;
            q.wide = level >= 2;
            q.off = (uint32_t)(level == 0 ? z * ps : level == 1 ? (n0 + z) * ps : (n0 + n1) * ps + (level == 2 ? z : 4) * pw);
        }
        else if (g.p.surf_format != SURF_I32)
        {
            static const int base[4] = { 0, 512, 640, 704 };
            q.wide = level >= 2;
            const int o = base[level] + z * (q.wide ? 16 : 8) - (g.p.min_level ? 512 : 0);      /* byte offset inside the 720-byte packed record (208-byte tail) */
            /* chunk-major rows (X265HIP_SURF_PACKED_T): 16-byte chunk c of group g sits at row + (c * groups + g) * 16 */
            q.off = (uint32_t)(g.p.surf_format == SURF_PACKED_T ? (o >> 4) * g.ng * 16 + (o & 15) : o);
        }
        else
        {
            static const int base[4] = { 0, 64, 80, 84 };
            q.wide = 1;
            q.off = (uint32_t)((base[level] + z) * 16 - (g.p.min_level ? 1024 : 0));
        }
        return true;
    }
    if (size == 8) return false;                                           /* a partial 8x8 block: not derivable */
    const int hs = size >> 1;
    return decompose(px, py, w, h, bx, by, hs, c) && decompose(px, py, w, h, bx + hs, by, hs, c) &&
           decompose(px, py, w, h, bx, by + hs, hs, c) && decompose(px, py, w, h, bx + hs, by + hs, hs, c);
}

inline bool lookup_raw(Ctx& c, const pixel* fref, int& out)
{
    if (c.ready[c.ctuRow] != c.gen)
    {
        /* default: never wait, the host primitive answers instead.  Test mode (wait): give the transfer up to 2 s, so that small
         * pictures - encoded faster than their surfaces travel - still exercise the lookups */
        bool arrived = false;
        if (g.wait)
            for (int spin = 0; spin < 20000 && !arrived; spin++)
            {
                struct timespec ts = { 0, 100000 };
                nanosleep(&ts, NULL);
                arrived = c.ready[c.ctuRow] == c.gen;
            }
        if (!arrived) { c.notReady++; return false; }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (!c.centred)
    {
        /* the window of this CTU is centred on the displacement the provider found for it; its centre arrived with the row's flag */
        if (c.centre) c.fref0 += (ptrdiff_t)c.centre[1] * c.stride + c.centre[0];
        c.centred = true;
    }
    const ptrdiff_t t = (fref - c.fref0) + c.bias;
    if (t < 0 || t >= ((ptrdiff_t)1 << 27)) { c.outside++; return false; }
    const uint64_t row = ((uint64_t)t * g.strideMagic) >> 40, col = (uint64_t)t - row * (uint64_t)c.stride;
    const uint64_t span = 2 * (uint64_t)g.p.range;
    if (row > span || col > span) { c.outside++; return false; }
    int sum = 0;
    if (g.p.layout)
    {
        const size_t idx = (size_t)row * g.pitch + col;
        for (int i = 0; i < c.nparts; i++)
        {
            if (c.parts[i].wide) sum += (int)((const uint32_t*)(c.ctuBase + c.parts[i].off))[idx];
            else
            {
                const unsigned v = ((const uint16_t*)(c.ctuBase + c.parts[i].off))[idx];
                if (v == 65535u) { c.saturated++; return false; }              /* not representable in 16 bits (above 8 bits only): the host's to compute */
                sum += (int)v;
            }
        }
    }
    else
    {
        const uint8_t* rec = g.p.surf_format == SURF_PACKED_T ? c.ctuBase + row * g.ng * g.groupBytes + (col >> 2) * 16
                                                               : c.ctuBase + (row * g.ng + (col >> 2)) * g.groupBytes;
        const int k = (int)(col & 3);
        for (int i = 0; i < c.nparts; i++)
            sum += c.parts[i].wide ? ((const int32_t*)(rec + c.parts[i].off))[k] : ((const uint16_t*)(rec + c.parts[i].off))[k];
    }
    /* the row must STILL be this generation's after the read: a reopened slot has its flags cleared before any row is rewritten */
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (c.ready[c.ctuRow] != c.gen) { g.torn.fetch_add(1, std::memory_order_relaxed); return false; }
    c.hits++;
    out = sum;
    return true;
}
inline bool lookup(Ctx& c, const pixel* fref, int& out)
{
    if (!SEAM_PROF_ON) return lookup_raw(c, fref, out);
    const uint64_t t0 = __rdtsc();
    const bool ok = lookup_raw(c, fref, out);
    c.cyc += __rdtsc() - t0;
    return ok;
}

void verify_fail(int part, int got, int want)
{
    g.mismatches++;
    fprintf(stderr, "ref_seam: VERIFY MISMATCH partition %d: surface %d, primitive %d\n", part, got, want);
    abort();
}

template <int P> int sad_seam(const pixel* fenc, intptr_t fstride, const pixel* fref, intptr_t rstride)
{
    Ctx& c = t_ctx;
    int v;
    if (c.valid && c.part == P && fenc == c.fenc && rstride == c.stride && lookup(c, fref, v))
    {
        if (g.verify) { const int w = g.sad[P](fenc, fstride, fref, rstride); if (w != v) verify_fail(P, v, w); }
        return v;
    }
    return g.sad[P](fenc, fstride, fref, rstride);
}

template <int P, int N> inline void sad_xn_seam(const pixel* fenc, const pixel* const* r, intptr_t rstride, int32_t* res)
{
    Ctx& c = t_ctx;
    int v[N];
    unsigned hit = 0;
    for (int i = 0; i < N; i++)
        if (lookup(c, r[i], v[i])) hit |= 1u << i;
    if (!hit && g.batchMisses)
    {
        /* A/B only (off by default): none of the candidates lies in the window - the host's own batched primitive, one call, as if the seam were not
         * there.  It LOSES a third of the fps on the fade (2.23 against 3.40, profiles/r04_encoder_legs.txt): g++ -O3 vectorises the reference's
         * single-reference SAD loop and not its three- / four-reference loops, so N calls of `sad` beat one call of sad_x4 - which also means that part of
         * what the seams gain over the C table is this host path and not the services (the csplit control table measures it) */
        if (N == 3) g.sad_x3[P](fenc, r[0], r[1], r[2], rstride, res);
        else g.sad_x4[P](fenc, r[0], r[1], r[2], r[3], rstride, res);
        return;
    }
    for (int i = 0; i < N; i++)
    {
        if (hit >> i & 1)
        {
            if (g.verify) { const int w = g.sad[P](fenc, FENC_STRIDE, r[i], rstride); if (w != v[i]) verify_fail(P, v[i], w); }
            res[i] = v[i];
        }
        else
            res[i] = g.sad[P](fenc, FENC_STRIDE, r[i], rstride);      /* sad_x3 / sad_x4 are N independent SADs (pixel.cpp:74-119) */
    }
}
template <int P> void sad_x3_seam(const pixel* fenc, const pixel* r0, const pixel* r1, const pixel* r2, intptr_t rstride, int32_t* res)
{
    Ctx& c = t_ctx;
    if (c.valid && c.part == P && fenc == c.fenc && rstride == c.stride) { const pixel* r[3] = { r0, r1, r2 }; sad_xn_seam<P, 3>(fenc, r, rstride, res); }
    else g.sad_x3[P](fenc, r0, r1, r2, rstride, res);
}
template <int P> void sad_x4_seam(const pixel* fenc, const pixel* r0, const pixel* r1, const pixel* r2, const pixel* r3, intptr_t rstride, int32_t* res)
{
    Ctx& c = t_ctx;
    if (c.valid && c.part == P && fenc == c.fenc && rstride == c.stride) { const pixel* r[4] = { r0, r1, r2, r3 }; sad_xn_seam<P, 4>(fenc, r, rstride, res); }
    else g.sad_x4[P](fenc, r0, r1, r2, r3, rstride, res);
}

template <int P> struct Install
{
    static void run(EncoderPrimitives& t, int& n)
    {
        const int w = PU_DIMS[P][0], h = PU_DIMS[P][1];
        g.sad[P] = t.pu[P].sad; g.sad_x3[P] = t.pu[P].sad_x3; g.sad_x4[P] = t.pu[P].sad_x4;
        if (!(w & 7) && !(h & 7) && (w < h ? w : h) >= g.p.min_pu && t.pu[P].sad)
        {
            t.pu[P].sad = sad_seam<P>; t.pu[P].sad_x3 = sad_x3_seam<P>; t.pu[P].sad_x4 = sad_x4_seam<P>;
            n += 3;
        }
        Install<P + 1>::run(t, n);
    }
};
template <> struct Install<NUM_PU_SIZES> { static void run(EncoderPrimitives&, int&) {} };

/* the host-only control table (x265ref_split_fill_table below) */
pixelcmp_t g_splitSad[NUM_PU_SIZES];
template <int P> void split_x3(const pixel* fenc, const pixel* r0, const pixel* r1, const pixel* r2, intptr_t rstride, int32_t* res)
{
    res[0] = g_splitSad[P](fenc, FENC_STRIDE, r0, rstride); res[1] = g_splitSad[P](fenc, FENC_STRIDE, r1, rstride); res[2] = g_splitSad[P](fenc, FENC_STRIDE, r2, rstride);
}
template <int P> void split_x4(const pixel* fenc, const pixel* r0, const pixel* r1, const pixel* r2, const pixel* r3, intptr_t rstride, int32_t* res)
{
    res[0] = g_splitSad[P](fenc, FENC_STRIDE, r0, rstride); res[1] = g_splitSad[P](fenc, FENC_STRIDE, r1, rstride);
    res[2] = g_splitSad[P](fenc, FENC_STRIDE, r2, rstride); res[3] = g_splitSad[P](fenc, FENC_STRIDE, r3, rstride);
}
template <int P> struct InstallSplit
{
    static void run(EncoderPrimitives& t, int& n)
    {
        g_splitSad[P] = t.pu[P].sad;
        if (t.pu[P].sad) { t.pu[P].sad_x3 = split_x3<P>; t.pu[P].sad_x4 = split_x4<P>; n += 2; }
        InstallSplit<P + 1>::run(t, n);
    }
};
template <> struct InstallSplit<NUM_PU_SIZES> { static void run(EncoderPrimitives&, int&) {} };

/* slot of (source picture poc, reference picture); -1 when none can be had.  The first query for a new source picture requests the
 * surfaces of ALL its unweighted references as one batch (x265hip_me_cache_submit_batch: searched back to back, downloaded
 * row-interleaved), so the top CTU rows of every reference arrive first. */
int pair_slot(int fencPoc, const PicYuv* fencPic, const Slice* slice, const PicYuv* rec, int recPoc, const Wt& wt, int& gen, int encodeOrder, int frameThreads)
{
    const int epoch = g.epoch.load(std::memory_order_acquire);
    if (t_pairs.fencPoc != fencPoc || t_pairs.epoch != epoch) { t_pairs.fencPoc = fencPoc; t_pairs.epoch = epoch; t_pairs.n = 0; }
    for (int i = 0; i < t_pairs.n; i++)
        if (t_pairs.e[i].rec == rec && t_pairs.e[i].recPoc == recPoc && same_wt(t_pairs.e[i].wt, wt)) { gen = t_pairs.e[i].gen; return t_pairs.e[i].slot; }
    int slot = -1;
    {
        std::lock_guard<std::mutex> lk(g.mu);
        bool known = false;
        for (int i = 0; i < g.p.slots; i++)
        {
            Pair& q = g.pairs[i];
            if (q.used && q.fencPoc == fencPoc) known = true;
            if (q.used && q.fencPoc == fencPoc && q.rec == rec && q.recPoc == recPoc && same_wt(q.wt, wt)) { slot = i; gen = q.gen; }
        }
        if (slot < 0 && g_gateWindow)
        {
            const uint64_t h = g.hits.load(std::memory_order_relaxed), o = g.outside.load(std::memory_order_relaxed);
            if ((h - g_gate.h0) + (o - g_gate.o0) >= g_gateWindow)
            {
                const bool low = (h - g_gate.h0) * 100 < ((h - g_gate.h0) + (o - g_gate.o0)) * (uint64_t)g_gatePct;
                if (low && !g_gate.closed) g_gate.closings++;
                g_gate.closed = low;
                g_gate.h0 = h; g_gate.o0 = o;
            }
            if (g_gate.closed && (encodeOrder & 7) != 0)
            {
                /* the host's own search runs (the lock guard releases); remembered per thread, so that the other searches of this picture on this reference do not come
                 * back to the mutex for the same answer */
                g_gate.skipped++;
                if (t_pairs.n < 8) { TlsPair& e = t_pairs.e[t_pairs.n++]; e.rec = rec; e.recPoc = recPoc; e.slot = -1; e.gen = 0; e.wt = wt; }
                return -1;
            }
        }
        if (slot < 0)
        {
            /* candidates: a new source picture -> every reference of the slice; otherwise just the one asked for */
            const PicYuv* want[2 * (MAX_NUM_REF + 1) + 1]; int wantPoc[2 * (MAX_NUM_REF + 1) + 1]; Wt wantWt[2 * (MAX_NUM_REF + 1) + 1]; int nwant = 0;
            want[nwant] = rec; wantPoc[nwant] = recPoc; wantWt[nwant++] = wt;
            if (!known)
                for (int l = 0; l < 2; l++)
                    for (int r = 0; r < slice->m_numRefIdx[l]; r++)
                    {
                        const MotionReference& mr = slice->m_mref[l][r];
                        if (!mr.reconPic || mr.reconPic->m_stride != g.p.stride) continue;
                        const Wt mw = plane_weight(&mr, 0);
                        if (mw.present ? !(g.p.streamed && g.p.pair_open_w) : mr.isWeighted) continue;       /* isWeighted without a weighted luma plane does not occur (reference.cpp:73) */
                        bool dup = false;
                        for (int k = 0; k < nwant; k++) dup |= want[k] == mr.reconPic && wantPoc[k] == slice->m_refPOCList[l][r] && same_wt(wantWt[k], mw);
                        if (!dup) { want[nwant] = mr.reconPic; wantPoc[nwant] = slice->m_refPOCList[l][r]; wantWt[nwant++] = mw; }
                    }
            int slots[2 * (MAX_NUM_REF + 1) + 1], gens[2 * (MAX_NUM_REF + 1) + 1]; const void* bufs[2 * (MAX_NUM_REF + 1) + 1]; int n = 0;
            for (int k = 0; k < nwant; k++)
            {
                int freeSlot = -1;
                for (int i = 0; i < g.p.slots && freeSlot < 0; i++)
                {
                    bool taken = false;
                    for (int j = 0; j < n; j++) taken |= slots[j] == i;
                    /* picture-granular: an earlier picture's surfaces - its encode is over (-F 1).  Row-granular: frame encoders take the
                     * pictures round robin in encode order and finish one before they start the next (encoder.cpp:1988-1989, :2394), so while
                     * picture e is being encoded every picture of encode order <= e - frameThreads is done */
                    const bool dead = !g.pairs[i].used || (g.p.streamed ? g.pairs[i].encodeOrder <= encodeOrder - frameThreads : g.pairs[i].fencPoc != fencPoc);
                    if (!taken && dead) freeSlot = i;
                }
                if (freeSlot < 0) break;
                slots[n] = freeSlot; bufs[n] = want[k]->m_picBuf[0]; n++;
            }
            int rc = -1;
            if (n > 0 && g.p.streamed)
            {
                /* the source picture travels once (whole: it is complete before its encode starts), then every pair is opened; the
                 * reference pictures' rows arrive from the producer hook (FrameFilter::processPostRow below), before or after */
                const uint64_t fkey = pic_key(g.instance, fencPoc, 0);
                bool staged = false;
                int freeF = -1;
                for (int i = 0; i < 32; i++)
                {
                    if (g.fencs[i].used && g.fencs[i].poc == fencPoc) staged = true;
                    if (!g.fencs[i].used || g.fencs[i].encodeOrder <= encodeOrder - frameThreads) freeF = i;
                }
                rc = 0;
                if (!staged)
                {
                    rc = freeF < 0 ? -1 : g.p.picture_rows(g.p.ctx, fkey, fencPic->m_picBuf[0], 0, g.p.height / 64);
                    if (!rc) { g.fencs[freeF].used = true; g.fencs[freeF].poc = fencPoc; g.fencs[freeF].encodeOrder = encodeOrder; }
                }
                for (int k = 0; k < n && rc == 0; k++)
                {
                    gens[k] = wantWt[k].present ? g.p.pair_open_w(g.p.ctx, slots[k], fkey, pic_key(g.instance, wantPoc[k], 1), &wantWt[k])
                                                : g.p.pair_open(g.p.ctx, slots[k], fkey, pic_key(g.instance, wantPoc[k], 1));
                    if (gens[k] <= 0) rc = -1;
                }
            }
            else if (n > 0 && g.p.submit_batch)
                rc = g.p.submit_batch(g.p.ctx, n, slots, fencPic->m_picBuf[0], (uint64_t)(uint32_t)fencPoc, bufs, gens);
            else if (n > 0)
            {
                rc = 0;
                for (int k = 0; k < n && rc == 0; k++)
                {
                    gens[k] = g.p.submit(g.p.ctx, slots[k], fencPic->m_picBuf[0], (uint64_t)(uint32_t)fencPoc, bufs[k]);
                    if (gens[k] <= 0) rc = -1;
                }
            }
            if (rc == 0)
            {
                for (int k = 0; k < n; k++)
                {
                    Pair& q = g.pairs[slots[k]];
                    q.used = true; q.fencPoc = fencPoc; q.rec = want[k]; q.recPoc = wantPoc[k]; q.slot = slots[k]; q.gen = gens[k];
                    q.encodeOrder = encodeOrder; q.wt = wantWt[k];
                    g.submits++;
                    if (wantWt[k].present) g.weightedPairs++;
                }
                slot = slots[0]; gen = gens[0];            /* want[0] is the pair that was asked for */
                g.epoch.fetch_add(1, std::memory_order_release);
                t_pairs.epoch = g.epoch.load(); t_pairs.n = 0;
            }
        }
    }
    if (slot < 0) { g.noSlot++; return -1; }
    if (t_pairs.n < 8) { TlsPair& e = t_pairs.e[t_pairs.n++]; e.rec = rec; e.recPoc = recPoc; e.slot = slot; e.gen = gen; e.wt = wt; }
    return slot;
}

/* ---- the sub-sample seam: MotionEstimate::subpelCompare compares the source block with a block of a precomputed PHASE PLANE of
 * the reference picture (x265hip_phase_cache: every fractional phase interpolated once per picture) instead of interpolating the
 * block per candidate.  Provider = C function pointers with the signatures of x265hip_phase_cache_submit / _planes / _ready. */
struct PhaseProvider
{
    void* ctx;
    int (*submit)(void* ctx, int slot, const void* luma, const void* cb, const void* cr);
    const void* (*planes)(void* ctx, int slot, int plane);
    const volatile int* (*ready)(void* ctx, int slot);
    /* row-granular flavour (x265hip_phase_stream_view_open / _picture_rows / _progress signatures); streamed = all three are set */
    int (*open)(void* ctx, int slot, uint64_t key, const void* weights, unsigned planes_weighted);
    int (*rows_fn)(void* ctx, uint64_t key, const void* luma, const void* cb, const void* cr, int ctu_row0, int ctu_rows);
    const volatile uint64_t* (*progress)(void* ctx, int slot);
    bool streamed;
    int ctuRows;
    int slots;
    intptr_t stride;  int rows;
    intptr_t strideC; int rowsC;
};
struct PhaseEntry { const PicYuv* rec; int poc; int gen; bool used; uint64_t lastUse; Wt wt[3]; };
inline bool same_wt3(const Wt* a, const Wt* b) { return same_wt(a[0], b[0]) && same_wt(a[1], b[1]) && same_wt(a[2], b[2]); }
struct SubSeam
{
    bool enabled = false, verify = false, wait = false;
    PhaseProvider p;
    std::mutex mu;
    PhaseEntry e[MAX_SLOTS];
    uint64_t tick = 0;
    uint64_t instance = 0;               /* bumped by every configure: a POC names a picture's content only within one encode */
    std::atomic<int> epoch{0};
    std::atomic<uint64_t> served{0}, notReady{0}, noContext{0}, submits{0}, mismatches{0}, noSlot{0}, rowsPublished{0}, torn{0}, weightedViews{0}, weightedServed{0};
} gs;

struct SubCtx
{
    bool valid;
    const ReferencePlanes* ref;
    const PicYuv* rec;
    const pixel* luma; const pixel* cb; const pixel* cr;      /* phase 1 of each plane set, buffer coordinates */
    const pixel* base[3];                                     /* allocation start of the plane the reference reads: PicYuv::m_picBuf or MotionReference::weightBuffer */
    bool weighted;
    size_t planeL, planeC;                                    /* samples per plane */
    const volatile int* ready;
    const volatile uint64_t* progress;                        /* streamed: generation << 32 | lines finished, [0] luma [1] chroma */
    int gen;
    bool arrived[2];
    uint64_t nServed, nWeighted, nNotReady, nTorn;           /* per search, added to the shared counters once at its end (16 threads x 13 M compares) */
};
thread_local SubCtx t_sub;
thread_local struct { int epoch; int n; struct { const PicYuv* rec; int poc; int slot; int gen; Wt wt[3]; } e[8]; } t_phase = { -1, 0, {} };

/* slot that holds (or will hold) the phase planes of reconstructed picture (rec, poc) seen through the weights wt[3]; -1 when none can
 * be had.  A view is submitted / opened the first time a search refers to it.  Picture-granular: the least recently used slot whose
 * picture the current slice does not reference is recycled (-F 1: earlier pictures' searches are over).  Row-granular: the least
 * recently used slot (more slots than views in use at once; a reader of a recycled slot sees the generation change and passes). */
int phase_slot(const PicYuv* rec, int poc, const Slice* slice, const Wt* wt, int& gen)
{
    const int epoch = gs.epoch.load(std::memory_order_acquire);
    if (t_phase.epoch != epoch) { t_phase.epoch = epoch; t_phase.n = 0; }
    for (int i = 0; i < t_phase.n; i++)
        if (t_phase.e[i].rec == rec && t_phase.e[i].poc == poc && same_wt3(t_phase.e[i].wt, wt)) { gen = t_phase.e[i].gen; return t_phase.e[i].slot; }
    int slot = -1;
    {
        std::lock_guard<std::mutex> lk(gs.mu);
        gs.tick++;
        for (int i = 0; i < gs.p.slots; i++)
            if (gs.e[i].used && gs.e[i].rec == rec && gs.e[i].poc == poc && same_wt3(gs.e[i].wt, wt)) { slot = i; gen = gs.e[i].gen; gs.e[i].lastUse = gs.tick; }
        if (slot < 0)
        {
            int victim = -1;
            for (int i = 0; i < gs.p.slots; i++)
            {
                if (!gs.e[i].used) { victim = i; break; }
                bool live = false;
                if (!gs.p.streamed)
                    for (int l = 0; l < 2 && !live; l++)
                        for (int r = 0; r < slice->m_numRefIdx[l] && !live; r++)
                            live = slice->m_mref[l][r].reconPic == gs.e[i].rec && slice->m_refPOCList[l][r] == gs.e[i].poc;
                if (!live && (victim < 0 || gs.e[i].lastUse < gs.e[victim].lastUse)) victim = i;
            }
            if (victim >= 0)
            {
                int gnew;
                if (gs.p.streamed)
                {
                    const unsigned mask = (wt[0].present ? 1u : 0u) | (wt[1].present ? 2u : 0u) | (wt[2].present ? 4u : 0u);
                    struct { int w0, round, shift, offset; } w3[3];
                    for (int c = 0; c < 3; c++) { w3[c].w0 = wt[c].w0; w3[c].round = wt[c].round; w3[c].shift = wt[c].shift; w3[c].offset = wt[c].offset; }
                    gnew = gs.p.open(gs.p.ctx, victim, pic_key(gs.instance, poc, 1), mask ? w3 : NULL, mask);
                    if (gnew > 0 && mask) gs.weightedViews++;
                }
                else
                    gnew = gs.p.submit(gs.p.ctx, victim, rec->m_picBuf[0], rec->m_picBuf[1], rec->m_picBuf[2]);
                if (gnew > 0)
                {
                    PhaseEntry& q = gs.e[victim];
                    q.used = true; q.rec = rec; q.poc = poc; q.gen = gnew; q.lastUse = gs.tick;
                    q.wt[0] = wt[0]; q.wt[1] = wt[1]; q.wt[2] = wt[2];
                    slot = victim; gen = gnew;
                    gs.submits++;
                    gs.epoch.fetch_add(1, std::memory_order_release);
                    t_phase.epoch = gs.epoch.load(); t_phase.n = 0;
                }
            }
        }
    }
    if (slot < 0) { gs.noSlot++; return -1; }
    if (t_phase.n < 8) { auto& e = t_phase.e[t_phase.n++]; e.rec = rec; e.poc = poc; e.slot = slot; e.gen = gen; e.wt[0] = wt[0]; e.wt[1] = wt[1]; e.wt[2] = wt[2]; }
    return slot;
}

/* POC of the reference picture `ref` points at: ref is an element of slice->m_mref[list][idx] (slice.h:337) */
int ref_poc(const Slice* slice, const ReferencePlanes* ref)
{
    const ptrdiff_t idx = static_cast<const MotionReference*>(ref) - &slice->m_mref[0][0];
    if (idx < 0 || idx >= 2 * (MAX_NUM_REF + 1)) return -0x7fffffff;
    return slice->m_refPOCList[idx / (MAX_NUM_REF + 1)][idx % (MAX_NUM_REF + 1)];
}

void sub_context(const Search* s, ReferencePlanes* ref)
{
    SubCtx& c = t_sub;
    c.valid = false;
    const PicYuv* rec = ref->reconPic;
    if (!rec || ref->isLowres || (s->m_param->frameNumThreads != 1 && !gs.p.streamed) || (ref->isWeighted && !gs.p.streamed) || rec->m_picCsp != X265_CSP_I420 ||
        rec->m_stride != gs.p.stride || rec->m_strideC != gs.p.strideC) { gs.noContext.fetch_add(1, std::memory_order_relaxed); return; }
    /* the planes the reference reads: the reconstruction, or - plane by plane - MotionReference::weightBuffer at the same pad (reference.cpp:103) */
    Wt wt[3];
    for (int k = 0; k < 3; k++)
    {
        wt[k] = plane_weight(ref, k);
        c.base[k] = wt[k].present ? static_cast<const MotionReference*>(ref)->weightBuffer[k] : rec->m_picBuf[k];
        if (!c.base[k] || ref->fpelPlane[k] - c.base[k] != rec->m_picOrg[k] - rec->m_picBuf[k]) { gs.noContext.fetch_add(1, std::memory_order_relaxed); return; }
    }
    const int maxH = (int)((rec->m_picHeight + s->m_param->maxCUSize - 1) / s->m_param->maxCUSize * s->m_param->maxCUSize);
    if (maxH + 2 * (int)rec->m_lumaMarginY != gs.p.rows || (maxH >> 1) + 2 * (int)rec->m_chromaMarginY != gs.p.rowsC)
    { gs.noContext.fetch_add(1, std::memory_order_relaxed); return; }
    const int poc = ref_poc(s->m_slice, ref);
    if (poc == -0x7fffffff) { gs.noContext.fetch_add(1, std::memory_order_relaxed); return; }
    int gen = 0;
    const int slot = phase_slot(rec, poc, s->m_slice, wt, gen);
    if (slot < 0) return;
    c.ref = ref; c.rec = rec;
    c.weighted = wt[0].present || wt[1].present || wt[2].present;
    c.luma = (const pixel*)gs.p.planes(gs.p.ctx, slot, 0);
    c.cb = (const pixel*)gs.p.planes(gs.p.ctx, slot, 1);
    c.cr = (const pixel*)gs.p.planes(gs.p.ctx, slot, 2);
    c.planeL = (size_t)gs.p.stride * gs.p.rows; c.planeC = (size_t)gs.p.strideC * gs.p.rowsC;
    c.ready = gs.p.streamed ? NULL : gs.p.ready(gs.p.ctx, slot);
    c.progress = gs.p.streamed ? gs.p.progress(gs.p.ctx, slot) : NULL;
    c.gen = gen;
    c.arrived[0] = c.arrived[1] = false;
    c.nServed = c.nWeighted = c.nNotReady = c.nTorn = 0;
    c.valid = c.luma && c.cb && c.cr && (c.ready || c.progress);
}

/* streamed: are buffer lines [line0, line1) of plane kind `which` (0 luma, 1 chroma) finished for this context's generation? */
inline bool sub_lines(SubCtx& c, int which, long line0, long line1)
{
    if (line0 < 4) return false;
    uint64_t v = c.progress[which];
    if (gs.wait)
        for (int spin = 0; spin < 20000 && ((int)(v >> 32) != c.gen || (long)(uint32_t)v < line1); spin++)
        {
            if ((int)(v >> 32) > c.gen) break;                  /* the slot went to another picture: never */
            struct timespec ts = { 0, 100000 };
            nanosleep(&ts, NULL);
            v = c.progress[which];
        }
    if ((int)(v >> 32) != c.gen || (long)(uint32_t)v < line1) return false;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return true;
}

inline bool sub_arrived(SubCtx& c, int which)
{
    if (c.arrived[which]) return true;
    bool ok = c.ready[which] == c.gen;
    if (!ok && gs.wait)          /* test mode: small pictures are encoded faster than their planes travel */
        for (int spin = 0; spin < 20000 && !ok; spin++)
        {
            struct timespec ts = { 0, 100000 };
            nanosleep(&ts, NULL);
            ok = c.ready[which] == c.gen;
        }
    if (ok) { __atomic_thread_fence(__ATOMIC_ACQUIRE); c.arrived[which] = true; }
    return ok;
}


/* ---- the COST-TABLE seam (round 6): MotionEstimate::subpelCompare's SATD comparisons answered as VALUES from x265hip_cost_stream's records
 * (include/x265hip.h, "SUB-SAMPLE COST TABLES").  Per (source picture, reference[, weights]) pair the service holds, for every PU made of 8x8 blocks,
 * records { int16 mvx, mvy; uint32 base; uint16 delta[positions] } around the 1 - 2 integer displacements of smallest SAD: the refinement of
 * motion.cpp:1456-1561 starts from the vector the integer search ended on and measures quarter-sample vectors of a fixed position set around it, so
 * when that vector is a record's (mvx, mvy) every comparison of the refinement is one table read.  Everything else - another start vector, a
 * position outside the set (a refinement that started from a fractional predictor), SAD comparisons (the predictor candidates of :773-812), a row
 * whose records have not landed, a saturated delta - is the next seam's or the reference's own to compute: the same integer.
 * Provider = C function pointers with the signatures of x265hip_cost_stream_picture_rows / _pair_open / _tables / _ready. */
struct CostProvider
{
    void* ctx;
    int (*picture_rows)(void* ctx, uint64_t key, const void* luma, const void* cb, const void* cr, int ctu_row0, int ctu_rows);
    int (*pair_open)(void* ctx, int slot, uint64_t fenc_key, uint64_t ref_key, const void* weights, unsigned planes_weighted, const uint16_t* mv_cost);
    const void* (*tables)(void* ctx, int slot);
    const volatile int* (*ready)(void* ctx, int slot);
    int slots;
    intptr_t stride, strideC;
    int width, height, marginX, marginY;
    int K, subme, chroma, recBytes, npos, npu, window;
    int rec2Off;                        /* > 0: every record carries the costs of the SAD-typed comparisons at this byte offset ({ uint32 base; uint16 delta[] }) */
    unsigned coverMask;                 /* bit s: the refinement of --subme s stays inside the service's position set when it starts on a record's vector */
    size_t ctuBytes;
};
struct CostPair { bool used; int fencPoc; const PicYuv* rec; int recPoc; Wt wt[3]; int gen; int encodeOrder; };
struct CostSeam
{
    bool enabled = false, verify = false, wait = false, useMvCost = true;
    CostProvider p;
    int16_t posMap[13 * 13];            /* (dy + 6) * 13 + dx + 6 -> index into delta[], -1 = not a position of the set */
    int16_t puIndex[8][8][8][8];        /* [w / 8 - 1][h / 8 - 1][y / 8][x / 8] -> PU of the service's list, -1 = not listed */
    std::mutex mu;
    CostPair pairs[MAX_SLOTS];
    FencStaged fencs[32];
    uint64_t instance = 0;
    std::atomic<int> epoch{0};
    std::atomic<uint64_t> servedSad{0};
    std::atomic<uint64_t> served{0}, otherVector{0}, notReady{0}, saturated{0}, torn{0}, noContext{0}, contexts{0}, pairsOpened{0}, noSlot{0}, rowsPublished{0}, rowsRefused{0},
                          mismatches{0}, weightedPairs{0};
} gc;
struct CostCtx
{
    bool valid;
    const ReferencePlanes* ref;
    const uint8_t* recs;                /* the PU's K records */
    const volatile int* ready;
    int row, gen;
    uint64_t nServed, nServedSad, nOther, nNotReady, nSaturated, nTorn;      /* per search, added to the shared counters once at its end */
};
thread_local CostCtx t_cost;
thread_local struct { int fencPoc; int epoch; int n; struct { const PicYuv* rec; int recPoc; Wt wt[3]; int slot; int gen; } e[8]; } t_cpairs = { -0x7fffffff, -1, 0, {} };

/* slot of the pair (source picture fencPoc, reference picture, weights); -1 when none can be had.  The first query of a source picture hands its three
 * planes to the provider (it is complete before its encode starts); the reference's rows arrive from the producer hook, before or after.  Slots are
 * recycled like the SAD seam's: frame encoders take pictures round robin in encode order (encoder.cpp:1988-1989, :2394), so while picture e is being
 * encoded every picture of encode order <= e - frameThreads is done. */
int cost_slot(int fencPoc, const PicYuv* fencPic, const PicYuv* rec, int recPoc, const Wt* wt, int& gen, int encodeOrder, int frameThreads, const uint16_t* mvCostQpel)
{
    const int epoch = gc.epoch.load(std::memory_order_acquire);
    if (t_cpairs.fencPoc != fencPoc || t_cpairs.epoch != epoch) { t_cpairs.fencPoc = fencPoc; t_cpairs.epoch = epoch; t_cpairs.n = 0; }
    for (int i = 0; i < t_cpairs.n; i++)
        if (t_cpairs.e[i].rec == rec && t_cpairs.e[i].recPoc == recPoc && same_wt3(t_cpairs.e[i].wt, wt)) { gen = t_cpairs.e[i].gen; return t_cpairs.e[i].slot; }
    int slot = -1;
    {
        std::lock_guard<std::mutex> lk(gc.mu);
        for (int i = 0; i < gc.p.slots; i++)
        {
            CostPair& q = gc.pairs[i];
            if (q.used && q.fencPoc == fencPoc && q.rec == rec && q.recPoc == recPoc && same_wt3(q.wt, wt)) { slot = i; gen = q.gen; }
        }
        if (slot < 0)
        {
            int freeSlot = -1;
            for (int i = 0; i < gc.p.slots && freeSlot < 0; i++)
                if (!gc.pairs[i].used || gc.pairs[i].encodeOrder <= encodeOrder - frameThreads) freeSlot = i;
            const uint64_t fkey = pic_key(gc.instance, fencPoc, 0);
            bool staged = false;
            int freeF = -1;
            for (int i = 0; i < 32; i++)
            {
                if (gc.fencs[i].used && gc.fencs[i].poc == fencPoc) staged = true;
                if (!gc.fencs[i].used || gc.fencs[i].encodeOrder <= encodeOrder - frameThreads) freeF = i;
            }
            int rc = freeSlot < 0 ? -1 : 0;
            if (!rc && !staged)
            {
                rc = freeF < 0 ? -1 : gc.p.picture_rows(gc.p.ctx, fkey, fencPic->m_picBuf[0], fencPic->m_picBuf[1], fencPic->m_picBuf[2], 0, gc.p.height / 64);
                if (!rc) { gc.fencs[freeF].used = true; gc.fencs[freeF].poc = fencPoc; gc.fencs[freeF].encodeOrder = encodeOrder; }
            }
            if (!rc)
            {
                const unsigned mask = (wt[0].present ? 1u : 0u) | (wt[1].present ? 2u : 0u) | (wt[2].present ? 4u : 0u);
                struct { int w0, round, shift, offset; } w3[3];
                for (int c = 0; c < 3; c++) { w3[c].w0 = wt[c].w0; w3[c].round = wt[c].round; w3[c].shift = wt[c].shift; w3[c].offset = wt[c].offset; }
                /* what the search minimises is SAD + mvcost(mv - mvp) (motion.cpp:246-328); the service ranks its candidates by the same sum with each CTU's own displacement
                 * standing in for the predictor: the cost of an integer displacement component d = BitCost's table at 4 d quarter samples (bitcost.h:45), at the QP of
                 * the first search of the pair */
                uint16_t mvCost[2 * 32 + 1];
                for (int i = 0; i <= 2 * gc.p.window; i++) mvCost[i] = mvCostQpel ? mvCostQpel[4 * (i - gc.p.window)] : 0;
                const int gnew = gc.p.pair_open(gc.p.ctx, freeSlot, fkey, pic_key(gc.instance, recPoc, 1), mask ? w3 : NULL, mask, mvCostQpel ? mvCost : NULL);
                if (gnew > 0)
                {
                    CostPair& q = gc.pairs[freeSlot];
                    q.used = true; q.fencPoc = fencPoc; q.rec = rec; q.recPoc = recPoc; q.wt[0] = wt[0]; q.wt[1] = wt[1]; q.wt[2] = wt[2]; q.gen = gnew; q.encodeOrder = encodeOrder;
                    slot = freeSlot; gen = gnew;
                    gc.pairsOpened++;
                    if (mask) gc.weightedPairs++;
                    gc.epoch.fetch_add(1, std::memory_order_release);
                    t_cpairs.epoch = gc.epoch.load(); t_cpairs.n = 0;
                }
            }
        }
    }
    if (slot < 0) gc.noSlot++;
    /* remembered per thread either way: the other searches of this picture on this reference do not come back to the mutex for the same answer */
    if (t_cpairs.n < 8) { auto& e = t_cpairs.e[t_cpairs.n++]; e.rec = rec; e.recPoc = recPoc; e.wt[0] = wt[0]; e.wt[1] = wt[1]; e.wt[2] = wt[2]; e.slot = slot; e.gen = gen; }
    return slot;
}

/* the cost-table context of one motionEstimate call: which PU of which CTU on which pair */
void cost_context(const Search* s, const uint16_t* mvCostQpel, ReferencePlanes* ref, int ctuAddr, int absPartIdx, int partEnum, int blockwidth, bool chromaSatd, int subme)
{
    CostCtx& c = t_cost;
    const PicYuv* rec = ref->reconPic;
    const Frame* frame = s->m_frame;
    const PicYuv* src = frame ? frame->m_fencPic : NULL;
    if (!rec || !src || rec->m_picCsp != X265_CSP_I420 || rec->m_stride != gc.p.stride || rec->m_strideC != gc.p.strideC || src->m_stride != gc.p.stride ||
        (int)rec->m_lumaMarginX != gc.p.marginX || (int)rec->m_lumaMarginY != gc.p.marginY || s->m_param->maxCUSize != 64 ||
        (int)chromaSatd != gc.p.chroma || subme < 0 || subme > 7 || !((gc.p.coverMask >> subme) & 1) || partEnum < 0 || partEnum >= NUM_PU_SIZES || PU_DIMS[partEnum][0] != blockwidth)
    { gc.noContext.fetch_add(1, std::memory_order_relaxed); return; }
    const int w = blockwidth, h = PU_DIMS[partEnum][1], px = g_zscanToPelX[absPartIdx], py = g_zscanToPelY[absPartIdx];
    if ((w | h | px | py) & 7) { gc.noContext.fetch_add(1, std::memory_order_relaxed); return; }
    const int pu = gc.puIndex[w / 8 - 1][h / 8 - 1][py / 8][px / 8];
    if (pu < 0) { gc.noContext.fetch_add(1, std::memory_order_relaxed); return; }
    /* the planes the reference reads: the reconstruction, or - plane by plane - MotionReference::weightBuffer at the same pad (reference.cpp:103) */
    Wt wt[3];
    for (int k = 0; k < 3; k++)
    {
        wt[k] = plane_weight(ref, k);
        const pixel* base = wt[k].present ? static_cast<const MotionReference*>(ref)->weightBuffer[k] : rec->m_picBuf[k];
        if (!base || ref->fpelPlane[k] - base != rec->m_picOrg[k] - rec->m_picBuf[k]) { gc.noContext.fetch_add(1, std::memory_order_relaxed); return; }
    }
    const int poc = ref_poc(s->m_slice, ref);
    if (poc == -0x7fffffff) { gc.noContext.fetch_add(1, std::memory_order_relaxed); return; }
    int gen = 0;
    const int slot = cost_slot(frame->m_poc, src, rec, poc, wt, gen, frame->m_encodeOrder, s->m_param->frameNumThreads, mvCostQpel);
    if (slot < 0) return;
    const uint8_t* tab = (const uint8_t*)gc.p.tables(gc.p.ctx, slot);
    c.ready = gc.p.ready(gc.p.ctx, slot);
    if (!tab || !c.ready) return;
    c.ref = ref;
    c.recs = tab + (size_t)ctuAddr * gc.p.ctuBytes + (size_t)pu * gc.p.K * gc.p.recBytes;
    c.row = ctuAddr / (gc.p.width / 64);
    c.gen = gen;
    c.nServed = c.nServedSad = c.nOther = c.nNotReady = c.nSaturated = c.nTorn = 0;
    c.valid = true;
}

/* one comparison from the records (sadTyped: the SAD-typed costs behind the SATD ones); false = not this seam's */
inline bool cost_lookup(CostCtx& c, const MV& qmv, int& out, bool sadTyped)
{
    if (c.ready[c.row] != c.gen)
    {
        bool arrived = false;
        if (gc.wait)          /* test mode: small pictures are encoded faster than their records travel */
            for (int spin = 0; spin < 20000 && !arrived; spin++)
            {
                struct timespec ts = { 0, 100000 };
                nanosleep(&ts, NULL);
                arrived = c.ready[c.row] == c.gen;
            }
        if (!arrived)
        {
            /* a row that does not come within 2 s will not come for the next lookup either: after a few timeouts the test mode stops waiting (and says so) */
            static std::atomic<int> timeouts{0};
            if (gc.wait && timeouts.fetch_add(1) < 8)
                fprintf(stderr, "ref_seam: cost records of CTU row %d (generation %d, flag %d) did not arrive within 2 s\n", c.row, c.gen, (int)c.ready[c.row]);
            if (gc.wait && timeouts.load() >= 8) gc.wait = false;
            c.nNotReady++;
            return false;
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    for (int k = 0; k < gc.p.K; k++)
    {
        const uint8_t* rec = c.recs + (size_t)k * gc.p.recBytes;
        const int mvx = ((const int16_t*)rec)[0], mvy = ((const int16_t*)rec)[1];
        if (mvx == -32768) continue;
        const int dx = qmv.x - 4 * mvx + 6, dy = qmv.y - 4 * mvy + 6;
        if ((unsigned)dx > 12u || (unsigned)dy > 12u) continue;
        const int idx = gc.posMap[dy * 13 + dx];
        if (idx < 0) continue;
        const uint8_t* part = sadTyped ? rec + gc.p.rec2Off : rec + 4;
        const unsigned delta = ((const uint16_t*)(part + 4))[idx];
        if (delta == 65535u) { c.nSaturated++; return false; }
        const int cost = (int)(*(const uint32_t*)part + delta);
        /* the row must STILL be this generation's after the read: a reopened slot has its flags cleared before any record is rewritten */
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        if (c.ready[c.row] != c.gen) { c.nTorn++; return false; }
        if (sadTyped) c.nServedSad++; else c.nServed++;
        out = cost;
        return true;
    }
    c.nOther++;
    return false;
}

} // namespace

/* the seam: same signature, same symbol as the reference's function (whose compiled body now answers to x265ref_orig_motionEstimate) */
int MotionEstimate::motionEstimate(ReferencePlanes* ref, const MV& mvmin, const MV& mvmax, const MV& qmvp, int numCandidates, const MV* mvc,
                                   int merange, MV& outQMv, uint32_t maxSlices, pixel* srcReferencePlane)
{
    ProfScope prof(gp.meCyc, gp.meCalls);
    Ctx& c = t_ctx;
    c.valid = false;
    const uint64_t tctx = SEAM_PROF_ON ? __rdtsc() : 0;
    if (g.enabled)
    {
        g.meCalls.fetch_add(1, std::memory_order_relaxed);
        const Wt wt = (ref->isLowres || !ref->reconPic) ? Wt{ 0, 0, 0, 0, 0 } : plane_weight(ref, 0);
        /* a weighted reference (fpelPlane[0] = MotionReference::weightBuffer[0] + the PicYuv pad, reference.cpp:103) needs the row-granular provider */
        const bool wtOk = wt.present ? (g.p.streamed && g.p.pair_open_w != NULL) : !ref->isWeighted;
        if (g.p.min_pu <= 64 && ctuAddr >= 0 && !srcReferencePlane && !ref->isLowres && wtOk && ref->reconPic && partEnum >= 0 && partEnum < NUM_PU_SIZES &&
            PU_DIMS[partEnum][0] == blockwidth && !(blockwidth & 7) && !(PU_DIMS[partEnum][1] & 7))
        {
            /* a MotionEstimate with ctuAddr >= 0 is the m_me member of a Search (search.h:257; the lookahead's instances use the other
             * setSourcePU overload, ctuAddr -1) */
            const Search* s = reinterpret_cast<const Search*>(reinterpret_cast<const char*>(this) - offsetof(Search, m_me));
            const Frame* frame = s->m_frame;
            const PicYuv* rec = ref->reconPic;
            const PicYuv* src = frame ? frame->m_fencPic : NULL;
            const pixel* lumaOrg = wt.present ? static_cast<const MotionReference*>(ref)->weightBuffer[0] + (rec->m_picOrg[0] - rec->m_picBuf[0]) : rec->m_picOrg[0];
            if (src && (s->m_param->frameNumThreads == 1 || g.p.streamed) && ref->fpelPlane[0] == lumaOrg &&
                rec->m_stride == g.p.stride && src->m_stride == g.p.stride && (int)rec->m_lumaMarginX == g.p.margin_x && (int)rec->m_lumaMarginY == g.p.margin_y &&
                (int)src->m_lumaMarginX == g.p.margin_x && (int)src->m_lumaMarginY == g.p.margin_y && s->m_param->maxCUSize == 64)
            {
                /* which reference picture: ref points into slice->m_mref[list][idx] (slice.h:337) */
                const Slice* slice = s->m_slice;
                const ptrdiff_t idx = static_cast<const MotionReference*>(ref) - &slice->m_mref[0][0];
                int recPoc = -0x7fffffff;
                if (idx >= 0 && idx < 2 * (MAX_NUM_REF + 1))
                    recPoc = slice->m_refPOCList[idx / (MAX_NUM_REF + 1)][idx % (MAX_NUM_REF + 1)];
                int gen = 0;
                const int slot = recPoc == -0x7fffffff ? -1 : pair_slot(frame->m_poc, src, slice, rec, recPoc, wt, gen, frame->m_encodeOrder, s->m_param->frameNumThreads);
                if (wt.present) g.weightedCalls.fetch_add(1, std::memory_order_relaxed);
                if (slot >= 0)
                {
                    c.nparts = 0;
                    const int px = g_zscanToPelX[absPartIdx], py = g_zscanToPelY[absPartIdx];
                    if (decompose(px, py, blockwidth, PU_DIMS[partEnum][1], 0, 0, 64, c) && c.nparts)
                    {
                        c.part = partEnum;
                        c.fenc = fencPUYuv.m_buf[0];
                        const intptr_t off = rec->getLumaAddr(ctuAddr, absPartIdx) - rec->getLumaAddr(0);
                        c.fref0 = ref->fpelPlane[0] + off;
                        const int16_t* cen = g.p.centres ? g.p.centres(g.p.ctx, slot) : NULL;
                        c.centre = cen ? cen + 2 * (size_t)ctuAddr : NULL;
                        c.centred = false;
                        c.stride = ref->lumaStride;
                        c.bias = (ptrdiff_t)g.p.range * c.stride + g.p.range;
                        c.ctuBase = (const uint8_t*)g.p.surface(g.p.ctx, slot) + (size_t)ctuAddr * g.ctuBytes;
                        c.ready = g.p.ready(g.p.ctx, slot);
                        c.ctuRow = ctuAddr / g.ctusW;
                        c.gen = gen;
                        c.weighted = wt.present != 0;
                        c.hits = c.outside = c.notReady = c.saturated = c.cyc = 0;
                        c.valid = true;
                    }
                }
            }
            else
                g.foreign.fetch_add(1, std::memory_order_relaxed);
        }
    }
    t_sub.valid = false;
    if (gs.enabled && ctuAddr >= 0 && !srcReferencePlane && !ref->isLowres)
        sub_context(reinterpret_cast<const Search*>(reinterpret_cast<const char*>(this) - offsetof(Search, m_me)), ref);
    t_cost.valid = false;
    if (gc.enabled && ctuAddr >= 0 && !srcReferencePlane && !ref->isLowres)
    {
        gc.contexts.fetch_add(1, std::memory_order_relaxed);
        cost_context(reinterpret_cast<const Search*>(reinterpret_cast<const char*>(this) - offsetof(Search, m_me)), gc.useMvCost ? m_cost : NULL, ref, ctuAddr, absPartIdx, partEnum, blockwidth, bChromaSATD, subpelRefine);
    }
    if (SEAM_PROF_ON) gp.ctxCyc.fetch_add(__rdtsc() - tctx, std::memory_order_relaxed);
    const int cost = x265ref_orig_motionEstimate(this, ref, &mvmin, &mvmax, &qmvp, numCandidates, mvc, merange, &outQMv, maxSlices, srcReferencePlane);
    if (t_sub.valid)
    {
        SubCtx& sc = t_sub;
        if (sc.nServed) gs.served.fetch_add(sc.nServed, std::memory_order_relaxed);
        if (sc.nWeighted) gs.weightedServed.fetch_add(sc.nWeighted, std::memory_order_relaxed);
        if (sc.nNotReady) gs.notReady.fetch_add(sc.nNotReady, std::memory_order_relaxed);
        if (sc.nTorn) gs.torn.fetch_add(sc.nTorn, std::memory_order_relaxed);
    }
    t_sub.valid = false;
    if (t_cost.valid)
    {
        CostCtx& cc = t_cost;
        if (cc.nServed) gc.served.fetch_add(cc.nServed, std::memory_order_relaxed);
        if (cc.nServedSad) gc.servedSad.fetch_add(cc.nServedSad, std::memory_order_relaxed);
        if (cc.nOther) gc.otherVector.fetch_add(cc.nOther, std::memory_order_relaxed);
        if (cc.nNotReady) gc.notReady.fetch_add(cc.nNotReady, std::memory_order_relaxed);
        if (cc.nSaturated) gc.saturated.fetch_add(cc.nSaturated, std::memory_order_relaxed);
        if (cc.nTorn) gc.torn.fetch_add(cc.nTorn, std::memory_order_relaxed);
    }
    t_cost.valid = false;
    if (c.valid)
    {
        c.valid = false;
        g.meServed.fetch_add(1, std::memory_order_relaxed);
        g.hits.fetch_add(c.hits, std::memory_order_relaxed);
        if (c.weighted) g.weightedHits.fetch_add(c.hits, std::memory_order_relaxed);
        g.outside.fetch_add(c.outside, std::memory_order_relaxed);
        g.notReady.fetch_add(c.notReady, std::memory_order_relaxed);
        if (c.saturated) g.saturated.fetch_add(c.saturated, std::memory_order_relaxed);
        if (SEAM_PROF_ON) { gp.lookCyc.fetch_add(c.cyc, std::memory_order_relaxed); gp.lookups.fetch_add(c.hits + c.outside + c.notReady + c.saturated, std::memory_order_relaxed); }
    }
    return cost;
}

/* The sub-sample seam: same signature, same symbol as the reference's function (motion.cpp:1571-1664; the compiled original is weak and
 * also answers to x265ref_orig_subpelCompare).  Inside a wrapped search on an unweighted 4:2:0 reference whose phase planes have
 * arrived, the block the reference would interpolate into subpelbuf (luma_hpp / luma_vpp / luma_hvpp; chroma filter_hpp / filter_vpp /
 * filter_hps + filter_vsp) is read in place from the plane of that phase - same samples, so the same cost; everything else goes to
 * the original. */
int MotionEstimate::subpelCompare(ReferencePlanes* ref, const MV& qmv, pixelcmp_t cmp)
{
    ProfScope prof(gp.subCyc, gp.subCalls);
    if (SEAM_PROBE_ON && !((qmv.x | qmv.y) & 3) && cmp == primitives.pu[partEnum].satd)
    {
        /* the refinement starts here: qmv is the integer vector the search ended on (motion.cpp:1515-1519) */
        Ctx& k = t_ctx;
        gpp.total++;
        if (!k.valid || !k.centred || !g.p.layout || k.part != partEnum || k.ready[k.ctuRow] != k.gen) gpp.noCtx++;
        else
        {
            const pixel* at = ref->fpelPlane[0] + blockOffset + (qmv.x >> 2) + (qmv.y >> 2) * k.stride;
            const ptrdiff_t t = (at - k.fref0) + k.bias;
            const uint64_t span = 2 * (uint64_t)g.p.range;
            const uint64_t row = t < 0 ? ~0ull : (uint64_t)t / (uint64_t)k.stride, col = t < 0 ? ~0ull : (uint64_t)t - row * (uint64_t)k.stride;
            if (t < 0 || row > span || col > span) gpp.outside++;
            else
            {
                auto sad_at = [&](size_t idx)
                {
                    unsigned s = 0;
                    for (int i = 0; i < k.nparts; i++)
                        s += k.parts[i].wide ? ((const uint32_t*)(k.ctuBase + k.parts[i].off))[idx] : ((const uint16_t*)(k.ctuBase + k.parts[i].off))[idx];
                    return s;
                };
                const size_t here = (size_t)row * g.pitch + col;
                const unsigned mine = sad_at(here);
                unsigned better = 0;                                    /* candidates the device would rank before this one: smaller SAD, or equal and earlier in raster order */
                for (uint64_t y = 0; y <= span && better < 8; y++)
                    for (uint64_t x = 0; x <= span; x++)
                    {
                        const size_t idx = (size_t)y * g.pitch + x;
                        const unsigned v = sad_at(idx);
                        better += v < mine || (v == mine && idx < here);
                    }
                if (better < 1) gpp.top1++;
                if (better < 2) gpp.top2++;
                if (better < 4) gpp.top4++;
                if (better < 8) gpp.top8++;
            }
        }
    }
    if (t_cost.valid && t_cost.ref == ref && sad != satd && (cmp == satd || (cmp == sad && gc.p.rec2Off > 0)))
    {
        int cost;
        if (cost_lookup(t_cost, qmv, cost, cmp == sad))
        {
            if (gc.verify)
            {
                const int want = x265ref_orig_subpelCompare(this, ref, &qmv, cmp);
                if (want != cost)
                {
                    gc.mismatches++;
                    fprintf(stderr, "ref_seam: COST TABLE VERIFY MISMATCH partition %d mv (%d,%d): record %d, reference %d\n", partEnum, qmv.x, qmv.y, cost, want);
                    abort();
                }
            }
            return cost;
        }
    }
    SubCtx& c = t_sub;
    if (!c.valid || c.ref != ref || (!c.progress && (!sub_arrived(c, 0) || (bChromaSATD && !sub_arrived(c, 1)))))
    {
        if (c.valid && c.ref == ref) c.nNotReady++;
        return x265ref_orig_subpelCompare(this, ref, &qmv, cmp);
    }
    const intptr_t stride = ref->lumaStride;
    const ptrdiff_t pos = blockOffset + (qmv.x >> 2) + (qmv.y >> 2) * stride;            /* relative to fpelPlane[0] */
    const int lph = (qmv.y & 3) * 4 + (qmv.x & 3);
    const ptrdiff_t lbuf = (ref->fpelPlane[0] - c.base[0]) + pos;                        /* relative to the buffer start */
    const intptr_t strideC = c.rec->m_strideC;
    const ptrdiff_t cpos = (qmv.x >> 3) + (qmv.y >> 3) * strideC;                        /* 4:2:0: the quarter-sample luma vector is an eighth-sample chroma vector */
    const int cph = (qmv.y & 7) * 8 + (qmv.x & 7);
    const pixel* cb = bChromaSATD ? ref->getCbAddr(ctuAddr, absPartIdx) + cpos : NULL;
    const pixel* cr = bChromaSATD ? ref->getCrAddr(ctuAddr, absPartIdx) + cpos : NULL;
    if (c.progress)
    {
        /* row-granular planes: the block's lines must be finished (the picture may still be growing under the producer) */
        const int bh = PU_DIMS[partEnum][1];
        bool ok = !lph || sub_lines(c, 0, lbuf / stride, lbuf / stride + bh);
        if (ok && bChromaSATD && cph)
        {
            const long l0 = (long)((cb - c.base[1]) / strideC);
            ok = sub_lines(c, 1, l0, l0 + (bh >> 1));
        }
        if (!ok)
        {
            c.nNotReady++;
            return x265ref_orig_subpelCompare(this, ref, &qmv, cmp);
        }
    }
    const pixel* lp = lph ? c.luma + (size_t)(lph - 1) * c.planeL + lbuf : ref->fpelPlane[0] + pos;
    int cost = cmp(fencPUYuv.m_buf[0], FENC_STRIDE, lp, stride);
    if (bChromaSATD)
    {
        if (cph)
        {
            cb = c.cb + (size_t)(cph - 1) * c.planeC + (cb - c.base[1]);
            cr = c.cr + (size_t)(cph - 1) * c.planeC + (cr - c.base[2]);
        }
        cost += chromaSatd(fencPUYuv.m_buf[1], fencPUYuv.m_csize, cb, strideC);
        cost += chromaSatd(fencPUYuv.m_buf[2], fencPUYuv.m_csize, cr, strideC);
    }
    if (c.progress)
    {
        /* the slot must STILL hold this picture after the read (open() clears the progress before a slot's planes are rewritten) */
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        if ((int)(c.progress[0] >> 32) != c.gen || (bChromaSATD && (int)(c.progress[1] >> 32) != c.gen))
        {
            c.nTorn++;
            return x265ref_orig_subpelCompare(this, ref, &qmv, cmp);
        }
    }
    c.nServed++;
    if (c.weighted) c.nWeighted++;
    if (gs.verify)
    {
        const int want = x265ref_orig_subpelCompare(this, ref, &qmv, cmp);
        if (want != cost)
        {
            gs.mismatches++;
            fprintf(stderr, "ref_seam: SUBPEL VERIFY MISMATCH partition %d mv (%d,%d): planes %d, reference %d\n", partEnum, qmv.x, qmv.y, cost, want);
            abort();
        }
    }
    return cost;
}

/* The PRODUCER hook of the row-granular providers: same signature, same symbol as the reference's function (framefilter.cpp:657-719;
 * oracle/Makefile makes the compiled original weak and reachable as x265ref_orig_processPostRow).  The reference's body raises
 * m_frame->m_reconRowFlag[row] (:664) - the row is final, borders extended - and from then on any frame encoder may search it; right
 * after it the row is handed to the providers: copied into their pinned staging inside the calls, so nothing here outlives the row. */
void FrameFilter::processPostRow(int row)
{
    x265ref_orig_processPostRow(this, row);
    ProfScope prof(gp.rowCyc, gp.rows);
    const bool sad = g.enabled && g.p.streamed && g.p.min_pu <= 64, sub = gs.enabled && gs.p.streamed;        /* min_pu > 64: no SAD stub is installed */
    if (gc.enabled)
    {
        const Frame* f = m_frame;
        const PicYuv* r = f ? f->m_reconPic : NULL;
        if (r && IS_REFERENCED(f) && m_param->maxCUSize == 64 && r->m_picCsp == X265_CSP_I420 && r->m_stride == gc.p.stride && r->m_strideC == gc.p.strideC &&
            (int)r->m_lumaMarginX == gc.p.marginX && (int)r->m_lumaMarginY == gc.p.marginY && m_numRows == gc.p.height / 64 && row < m_numRows)
        {
            if (gc.p.picture_rows(gc.p.ctx, pic_key(gc.instance, f->m_poc, 1), r->m_picBuf[0], r->m_picBuf[1], r->m_picBuf[2], row, 1) == 0)
                gc.rowsPublished.fetch_add(1, std::memory_order_relaxed);
            else
                gc.rowsRefused.fetch_add(1, std::memory_order_relaxed);
        }
    }
    if (!sad && !sub) return;
    const Frame* frame = m_frame;
    const PicYuv* rec = frame ? frame->m_reconPic : NULL;
    if (!rec || !IS_REFERENCED(frame) || m_param->maxCUSize != 64) return;            /* nobody will search an unreferenced B picture */
    if (sad && rec->m_stride == g.p.stride && (int)rec->m_lumaMarginX == g.p.margin_x && (int)rec->m_lumaMarginY == g.p.margin_y &&
        m_numRows == g.p.height / 64 && row < m_numRows)
    {
        if (g.p.picture_rows(g.p.ctx, pic_key(g.instance, frame->m_poc, 1), rec->m_picBuf[0], row, 1) == 0)
            g.rowsPublished.fetch_add(1, std::memory_order_relaxed);
        else
            g.rowsRefused.fetch_add(1, std::memory_order_relaxed);
    }
    if (sub && rec->m_picCsp == X265_CSP_I420 && rec->m_stride == gs.p.stride && rec->m_strideC == gs.p.strideC && m_numRows == gs.p.ctuRows &&
        m_numRows * 64 + 2 * (int)rec->m_lumaMarginY == gs.p.rows && m_numRows * 32 + 2 * (int)rec->m_chromaMarginY == gs.p.rowsC)
    {
        if (gs.p.rows_fn(gs.p.ctx, pic_key(gs.instance, frame->m_poc, 1), rec->m_picBuf[0], rec->m_picBuf[1], rec->m_picBuf[2], row, 1) == 0)
            gs.rowsPublished.fetch_add(1, std::memory_order_relaxed);
    }
}

/* The lookahead seam: same signature, same symbol as the reference's function.  oracle/Makefile makes the compiled original weak and
 * reachable as x265ref_orig_estimateFrameCost; this definition wins at link time for every caller (singleCost, the batch tasks).
 * Only the compute path is taken over - the bookkeeping around the block loop is restated from slicetype.cpp:3121-3146 / 3199-3211,
 * the loop itself (:3178-3196 over estimateCUCost) becomes one provider call.  Cached triples, HME and cooperative slices go to
 * the original: run with --lookahead-slices 1, where the reference walks the picture in one piece too (sliced walks reset the
 * row predictors at every slice boundary, a different - equally valid - estimate). */
int64_t CostEstimateGroup::estimateFrameCost(LookaheadTLD& tld, int p0, int p1, int b, bool bIntraPenalty)
{
    ProfScope prof(gp.laCyc, gp.laCalls);
    Lowres* fenc = m_frames[b];
    x265_param* param = m_lookahead.m_param;
    const bool cached = fenc->costEst[b - p0][p1 - b] >= 0 && fenc->rowSatds[b - p0][p1 - b][0] != -1;
    const bool small = m_lookahead.m_8x8Width * m_lookahead.m_8x8Height < gla.minBlocks;
    if (!gla.enabled || cached || small || param->bEnableHME || (!m_batchMode && m_lookahead.m_numCoopSlices > 1) || param->rc.qgSize == 8)
    {
        if (gla.enabled && !cached) (small ? gla.gated : gla.passed).fetch_add(1, std::memory_order_relaxed);
        return x265ref_orig_estimateFrameCost(this, &tld, p0, p1, b, bIntraPenalty);
    }
    bool bDoSearch[2];
    bDoSearch[0] = fenc->lowresMvs[0][b - p0][0].x == 0x7FFF;
    bDoSearch[1] = p1 > b && fenc->lowresMvs[1][p1 - b][0].x == 0x7FFF;
    fenc->weightedRef[b - p0].isWeighted = false;
    if (param->bEnableWeightedPred && bDoSearch[0])
        tld.weightsAnalyse(*m_frames[b], *m_frames[p0]);
    fenc->costEst[b - p0][p1 - b] = 0;
    fenc->costEstAq[b - p0][p1 - b] = 0;

    Lowres* fref0 = m_frames[p0];
    Lowres* fref1 = m_frames[p1];
    const bool bidir = b < p1;
    const ReferencePlanes* wfref0 = fenc->weightedRef[b - p0].isWeighted ? &fenc->weightedRef[b - p0] : fref0;
    const ptrdiff_t pad = fenc->lowresPlane[0] - fenc->buffer[0];
    int64_t frame[4] = { 0, 0, 0, 0 };
    const int half = 4 * (fenc->width > fenc->lines ? fenc->width : fenc->lines) + 4 * 256;      /* mv differences stay far inside */
    int rc;
    if (gla.host)
    {
        LaHostParams q;
        memset(&q, 0, sizeof(q));
        q.depth = X265_DEPTH; q.stride = fenc->lumaStride;
        q.width_in_cu = m_lookahead.m_8x8Width; q.height_in_cu = m_lookahead.m_8x8Height;
        q.lines = fenc->lines; q.margin_y = (int)(pad / fenc->lumaStride); q.margin_x = (int)(pad % fenc->lumaStride);
        q.cur = fenc->lowresPlane[0];
        for (int i = 0; i < 4; i++)
        {
            q.ref[i] = wfref0->lowresPlane[i];
            q.ref1[i] = bidir ? fref1->lowresPlane[i] : NULL;
            q.ref_bi[i] = (bidir && wfref0 != fref0) ? fref0->lowresPlane[i] : NULL;
        }
        q.intra_cost = fenc->intraCost; q.inv_qscale = fenc->invQscaleFactor;
        q.cost_q = static_cast<const MeCostProbe&>(tld.me).costCentre(); q.cost_q_half = half;
        q.bframe_bias = param->bFrameBias;
        q.do_search[0] = bDoSearch[0]; q.do_search[1] = bDoSearch[1];
        q.mvs[0] = &fenc->lowresMvs[0][b - p0][0].x; q.mv_costs[0] = fenc->lowresMvCosts[0][b - p0];
        q.mvs[1] = bidir ? &fenc->lowresMvs[1][p1 - b][0].x : NULL; q.mv_costs[1] = bidir ? fenc->lowresMvCosts[1][p1 - b] : NULL;
        q.lowres_costs = fenc->lowresCosts[b - p0][p1 - b]; q.row_satds = fenc->rowSatds[b - p0][p1 - b]; q.frame = frame;
        /* a Lowres' planes are written once per picture (Lowres::init): its frame number names their content; weighted planes live in
         * the thread's scratch (tld.wbuffer) and are uploaded every time */
        const uint64_t inst = gla.instance << 32;
        q.plane_key_cur = inst | ((uint64_t)(uint32_t)fenc->frameNum + 1);
        q.plane_key_ref = wfref0 == fref0 ? inst | ((uint64_t)(uint32_t)fref0->frameNum + 1) : 0;
        q.plane_key_ref1 = bidir ? inst | ((uint64_t)(uint32_t)fref1->frameNum + 1) : 0;
        q.plane_key_ref_bi = (bidir && wfref0 != fref0) ? inst | ((uint64_t)(uint32_t)fref0->frameNum + 1) : 0;
        /* both providers configured = verify mode (tests): the oracle scores the same triple from the same inputs into scratch arrays */
        std::vector<int32_t> vm0, vc0, vm1, vc1, vrs; std::vector<uint16_t> vlc; int64_t vframe[4] = { 0, 0, 0, 0 };
        const int nblk = q.width_in_cu * q.height_in_cu;
        if (gla.oracle)
        {
            vm0.assign(q.mvs[0], q.mvs[0] + 2 * nblk); vc0.assign(q.mv_costs[0], q.mv_costs[0] + nblk);
            if (bidir) { vm1.assign(q.mvs[1], q.mvs[1] + 2 * nblk); vc1.assign(q.mv_costs[1], q.mv_costs[1] + nblk); }
            vlc.assign(q.lowres_costs, q.lowres_costs + nblk); vrs.assign(q.row_satds, q.row_satds + q.height_in_cu);
        }
        rc = gla.host(&q);
#if X265HIP_BINDING_TEST_HOOKS
        {   /* diagnostic (tools/r6_lookahead_bound_ab.sh): X265REF_LA_DELAY_US adds latency to every served estimate - does the encode's fps follow it? */
            static const int delayUs = getenv("X265REF_LA_DELAY_US") ? atoi(getenv("X265REF_LA_DELAY_US")) : 0;
            if (delayUs > 0) { struct timespec ts = { 0, (long)delayUs * 1000L }; nanosleep(&ts, NULL); }
        }
#endif
        if (gla.oracle && !rc)
        {
            const pixel* r0[4]; const pixel* r1[4]; const pixel* rb[4];
            for (int i = 0; i < 4; i++) { r0[i] = wfref0->lowresPlane[i]; r1[i] = fref1->lowresPlane[i]; rb[i] = fref0->lowresPlane[i]; }
            const int ds[2] = { bDoSearch[0], bDoSearch[1] };
            gla.oracle(fenc->lowresPlane[0], r0, bidir ? r1 : NULL, fenc->lumaStride, q.width_in_cu, q.height_in_cu,
                       static_cast<const MeCostProbe&>(tld.me).costCentre(), 0, fenc->intraCost, fenc->invQscaleFactor, ds, param->bFrameBias,
                       vm0.data(), vc0.data(), bidir ? vm1.data() : NULL, bidir ? vc1.data() : NULL, vlc.data(), vrs.data(), vframe,
                       (bidir && wfref0 != fref0) ? rb : NULL);
            int bad = 0;
            auto cmp = [&](const char* what, const void* a, const void* bb, size_t bytes)
            { if (memcmp(a, bb, bytes)) { bad++; fprintf(stderr, "ref_seam: LOOKAHEAD VERIFY MISMATCH (%d, %d, %d) %s\n", p0, b, p1, what); } };
            cmp("mvs[0]", q.mvs[0], vm0.data(), (size_t)2 * nblk * 4); cmp("mv_costs[0]", q.mv_costs[0], vc0.data(), (size_t)nblk * 4);
            if (bidir) { cmp("mvs[1]", q.mvs[1], vm1.data(), (size_t)2 * nblk * 4); cmp("mv_costs[1]", q.mv_costs[1], vc1.data(), (size_t)nblk * 4); }
            cmp("lowres_costs", q.lowres_costs, vlc.data(), (size_t)nblk * 2); cmp("row_satds", q.row_satds, vrs.data(), (size_t)q.height_in_cu * 4);
            cmp("frame totals", frame, vframe, sizeof(vframe));
            if (bad)
            {
                fprintf(stderr, "ref_seam: do_search %d/%d weighted %d bidir %d keys %llu %llu %llu frame gpu %lld %lld %lld %lld oracle %lld %lld %lld %lld\n",
                        bDoSearch[0], bDoSearch[1], wfref0 != fref0, bidir, (unsigned long long)q.plane_key_cur, (unsigned long long)q.plane_key_ref,
                        (unsigned long long)q.plane_key_ref1, (long long)frame[0], (long long)frame[1], (long long)frame[2], (long long)frame[3],
                        (long long)vframe[0], (long long)vframe[1], (long long)vframe[2], (long long)vframe[3]);
                gla.mismatches.fetch_add(1, std::memory_order_relaxed);
            }
        }
    }
    else
    {
        const pixel* r0[4]; const pixel* r1[4]; const pixel* rb[4];
        for (int i = 0; i < 4; i++) { r0[i] = wfref0->lowresPlane[i]; r1[i] = fref1->lowresPlane[i]; rb[i] = fref0->lowresPlane[i]; }
        const int ds[2] = { bDoSearch[0], bDoSearch[1] };
        rc = gla.oracle(fenc->lowresPlane[0], r0, bidir ? r1 : NULL, fenc->lumaStride, m_lookahead.m_8x8Width, m_lookahead.m_8x8Height,
                        static_cast<const MeCostProbe&>(tld.me).costCentre(), 0, fenc->intraCost, fenc->invQscaleFactor, ds, param->bFrameBias,
                        &fenc->lowresMvs[0][b - p0][0].x, fenc->lowresMvCosts[0][b - p0],
                        bidir ? &fenc->lowresMvs[1][p1 - b][0].x : NULL, bidir ? fenc->lowresMvCosts[1][p1 - b] : NULL,
                        fenc->lowresCosts[b - p0][p1 - b], fenc->rowSatds[b - p0][p1 - b], frame, (bidir && wfref0 != fref0) ? rb : NULL);
    }
    if (rc)
    {
        /* provider failure: loud, and the reference's own loop takes over for this triple (its state is as the original expects it) */
        gla.failed.fetch_add(1, std::memory_order_relaxed);
        fprintf(stderr, "ref_seam: lookahead provider failed (%d) for (%d, %d, %d); the reference's loop runs instead\n", rc, p0, b, p1);
        fenc->costEst[b - p0][p1 - b] = -1;
        return x265ref_orig_estimateFrameCost(this, &tld, p0, p1, b, bIntraPenalty);
    }
    gla.served.fetch_add(1, std::memory_order_relaxed);
    fenc->costEstAq[b - p0][p1 - b] = frame[1];
    if (p1 == b) fenc->intraMbs[b - p0] += (int)frame[2];
    int64_t score = frame[0];
    if (b != p1)
        score = score * 100 / (130 + param->bFrameBias);
    fenc->costEst[b - p0][p1 - b] = score;
    if (bIntraPenalty)
        score += score * fenc->intraMbs[b - p0] / (tld.ncu * 8);
    return score;
}

#if X265HIP_BINDING_TEST_HOOKS
/* fixture generation (tools/gen_weight_golden.py): X265REF_WA_DUMP=<dir> + verify writes, per served slice, everything x265hip_weight_analyse_host is
 * handed and what the REFERENCE's own weightAnalyse answered, as (name, element size, count, bytes) records */
static void wa_dump_arr(FILE* f, const char* name, int elem, size_t count, const void* data)
{
    const uint32_t n = (uint32_t)strlen(name), e = (uint32_t)elem;
    const uint64_t c = count;
    fwrite(&n, 4, 1, f); fwrite(name, 1, n, f); fwrite(&e, 4, 1, f); fwrite(&c, 8, 1, f); fwrite(data, (size_t)elem, count, f);
}
#endif


/* LookaheadTLD::calcAdaptiveQuantFrame (slicetype.cpp:444-694; once per source picture from PreLookaheadGroup::processTasks, :1395): the
 * acEnergyCu loop over every quantisation group of the picture - the pixel work - and the double-precision offsets as ONE provider call
 * (x265hip_aq_frame_host) that writes straight into the Lowres arrays.  The AQ modes 1 - 3 of 4:2:0 / 4:0:0 pictures; everything the
 * service does not restate (quantOffsets, --hdr10-opt, --hevc-aq, the edge mode, the 2-pass cuTree reuse, --dynamic-refine / --fades'
 * blockVariance pass) keeps the reference's own body, as do pictures below the size gate. */
void LookaheadTLD::calcAdaptiveQuantFrame(Frame* curFrame, x265_param* param)
{
    const PicYuv* pic = curFrame->m_fencPic;
    Lowres& lr = curFrame->m_lowres;
    const int q = param->rc.qgSize == 8 ? 8 : 16;
    const int bw = (pic->m_picWidth + q - 1) / q, bh = (pic->m_picHeight + q - 1) / q;
    const int blockCount = q == 8 ? (int)(lr.maxBlocksInRowFullRes * lr.maxBlocksInColFullRes) : widthInCU * heightInCU;
    const bool mono = param->internalCsp == X265_CSP_I400 || pic->m_picCsp == X265_CSP_I400;
    const bool covered = gaq.enabled && param->rc.aqMode >= X265_AQ_VARIANCE && param->rc.aqMode <= X265_AQ_AUTO_VARIANCE_BIASED && param->rc.aqStrength > 0 &&
                         !param->rc.hevcAq && !(param->rc.bStatRead && param->rc.cuTree && IS_REFERENCED(curFrame)) && !param->bHDR10Opt && !curFrame->m_quantOffsets &&
                         !param->bDynamicRefine && !param->bEnableFades && (mono || (param->internalCsp == X265_CSP_I420 && pic->m_picCsp == X265_CSP_I420)) &&
                         bw * bh == blockCount && (q == 16 || bw == (int)lr.maxBlocksInRowFullRes);
    if (!covered || blockCount < gaq.minBlocks)
    {
        if (gaq.enabled) (covered ? gaq.gated : gaq.passed).fetch_add(1, std::memory_order_relaxed);
        x265ref_orig_calcAdaptiveQuantFrame(this, curFrame, param);
        return;
    }
    const int weightp = param->bEnableWeightedPred || param->bEnableWeightedBiPred;
    int rc = 0;
    if (gaq.host)
    {
        AqFrameHostParams a;
        memset(&a, 0, sizeof(a));
        a.depth = X265_DEPTH; a.y = pic->m_picOrg[0]; a.stride = pic->m_stride;
        if (!mono) { a.cb = pic->m_picOrg[1]; a.cr = pic->m_picOrg[2]; a.stride_c = pic->m_strideC; }
        a.width = pic->m_picWidth; a.height = pic->m_picHeight; a.qg_size = q; a.aq_mode = param->rc.aqMode; a.aq_strength = param->rc.aqStrength;
        a.width_in_cu = widthInCU; a.height_in_cu = heightInCU; a.normalise_wp = weightp;
        a.qp_aq_offset = lr.qpAqOffset; a.qp_cutree_offset = lr.qpCuTreeOffset; a.inv_qscale = lr.invQscaleFactor;
        a.inv_qscale_8x8 = q == 8 ? lr.invQscaleFactor8x8 : NULL;
        a.wp_sum = lr.wp_sum; a.wp_ssd = lr.wp_ssd;
        rc = gaq.host(&a);
    }
    else
    {
        std::vector<uint32_t> energy(blockCount);
        gaq.oracle(pic->m_picOrg[0], mono ? NULL : pic->m_picOrg[1], mono ? NULL : pic->m_picOrg[2], pic->m_stride, pic->m_strideC, pic->m_picWidth, pic->m_picHeight,
                   q, param->rc.aqMode, param->rc.aqStrength, weightp, energy.data(), lr.qpAqOffset, lr.invQscaleFactor, lr.wp_sum, lr.wp_ssd);
        memcpy(lr.qpCuTreeOffset, lr.qpAqOffset, (size_t)blockCount * sizeof(double));
        if (q == 8)
            for (int cy = 0; cy < heightInCU; cy++)
                for (int cx = 0; cx < widthInCU; cx++)
                {
                    const int* f = lr.invQscaleFactor + cx * 2 + cy * widthInCU * 4;
                    lr.invQscaleFactor8x8[cx + cy * widthInCU] = (f[0] + f[1] + f[bw] + f[bw + 1]) / 4;
                }
    }
    if (rc)
    {
        gaq.failed.fetch_add(1, std::memory_order_relaxed);
        fprintf(stderr, "ref_seam: adaptive-quantisation provider failed (%d); the reference's loop runs instead\n", rc);
        x265ref_orig_calcAdaptiveQuantFrame(this, curFrame, param);
        return;
    }
    gaq.served.fetch_add(1, std::memory_order_relaxed);
    if (gaq.verify)
    {
        const std::vector<double> aq(lr.qpAqOffset, lr.qpAqOffset + blockCount), ct(lr.qpCuTreeOffset, lr.qpCuTreeOffset + blockCount);
        const std::vector<int> inv(lr.invQscaleFactor, lr.invQscaleFactor + blockCount);
        std::vector<int> inv8;
        if (q == 8) inv8.assign(lr.invQscaleFactor8x8, lr.invQscaleFactor8x8 + widthInCU * heightInCU);
        uint64_t sum[3], ssd[3];
        memcpy(sum, lr.wp_sum, sizeof(sum)); memcpy(ssd, lr.wp_ssd, sizeof(ssd));
        x265ref_orig_calcAdaptiveQuantFrame(this, curFrame, param);
        if (memcmp(aq.data(), lr.qpAqOffset, (size_t)blockCount * sizeof(double)) || memcmp(ct.data(), lr.qpCuTreeOffset, (size_t)blockCount * sizeof(double)) ||
            memcmp(inv.data(), lr.invQscaleFactor, (size_t)blockCount * sizeof(int)) || memcmp(sum, lr.wp_sum, sizeof(sum)) || memcmp(ssd, lr.wp_ssd, sizeof(ssd)) ||
            (q == 8 && memcmp(inv8.data(), lr.invQscaleFactor8x8, inv8.size() * sizeof(int))))
        {
            fprintf(stderr, "ref_seam: ADAPTIVE QUANTISATION VERIFY MISMATCH poc %d\n", curFrame->m_poc);
            gaq.mismatches.fetch_add(1, std::memory_order_relaxed);
        }
#if X265HIP_BINDING_TEST_HOOKS
        if (const char* dir = getenv("X265REF_AQ_DUMP"))          /* fixture generation (tools/gen_weight_golden.py): the picture and what the REFERENCE's function left in Lowres */
        {
            static std::atomic<int> serial{0};
            char path[1024];
            snprintf(path, sizeof(path), "%s/aq_%03d_poc%d.bin", dir, serial.fetch_add(1), curFrame->m_poc);
            if (FILE* f = fopen(path, "wb"))
            {
                const int chh = pic->m_picHeight >> 1;
                const size_t yplane = (size_t)pic->m_stride * (pic->m_picHeight + 2 * pic->m_lumaMarginY), yorg = (size_t)pic->m_lumaMarginY * pic->m_stride + pic->m_lumaMarginX;
                const size_t cplane = (size_t)pic->m_strideC * (chh + 2 * pic->m_chromaMarginY), corg = (size_t)pic->m_chromaMarginY * pic->m_strideC + pic->m_chromaMarginX;
                const int32_t geo[12] = { X265_DEPTH, (int32_t)pic->m_stride, (int32_t)pic->m_lumaMarginX, (int32_t)pic->m_lumaMarginY, (int32_t)pic->m_strideC, (int32_t)pic->m_chromaMarginX,
                                          (int32_t)pic->m_chromaMarginY, pic->m_picWidth, pic->m_picHeight, q, param->rc.aqMode, weightp };
                const double strength = param->rc.aqStrength;
                wa_dump_arr(f, "geo", 4, 12, geo);
                wa_dump_arr(f, "strength", 8, 1, &strength);
                const int32_t grid[2] = { widthInCU, heightInCU };
                wa_dump_arr(f, "grid", 4, 2, grid);
                wa_dump_arr(f, "y", sizeof(pixel), yplane, pic->m_picOrg[0] - yorg);
                if (!mono) { wa_dump_arr(f, "cb", sizeof(pixel), cplane, pic->m_picOrg[1] - corg); wa_dump_arr(f, "cr", sizeof(pixel), cplane, pic->m_picOrg[2] - corg); }
                wa_dump_arr(f, "qp_aq_offset", 8, blockCount, lr.qpAqOffset);
                wa_dump_arr(f, "qp_cutree_offset", 8, blockCount, lr.qpCuTreeOffset);
                wa_dump_arr(f, "inv_qscale", 4, blockCount, lr.invQscaleFactor);
                if (q == 8) wa_dump_arr(f, "inv_qscale_8x8", 4, (size_t)widthInCU * heightInCU, lr.invQscaleFactor8x8);
                wa_dump_arr(f, "wp_sum", 8, 3, lr.wp_sum); wa_dump_arr(f, "wp_ssd", 8, 3, lr.wp_ssd);
                fclose(f);
            }
        }
#endif

    }
}

/* weightAnalyse (weightPrediction.cpp:222-520; FrameEncoder::compressFrame calls it for every P / B slice with --weightp / --weightb on): the
 * compensated reference planes, weightCost of every (scale, offset) pair and the decision logic as ONE provider call
 * (x265hip_weight_analyse_host) that returns reference 0's weights per (list, plane) and the denominators the other references are reset to;
 * this wrapper does what stays host-side in the reference - the lazy border extension of the references' source chroma (:333-343) before,
 * the table assembly (:468-479) after.  4:2:0 pictures whose lowres planes are a multiple of 8 wide and high (otherwise the reference's loop
 * reads rows weight_pp never wrote, :180-184 against :196). */
namespace X265_NS {
void weightAnalyse(Slice& slice, Frame& frame, x265_param& param)
{
    PicYuv* pic = frame.m_fencPic;
    Lowres& fenc = frame.m_lowres;
    const int numPredDir = slice.isInterP() ? 1 : 2;
    const bool covered = gwa.enabled && param.internalCsp == X265_CSP_I420 && pic->m_picCsp == X265_CSP_I420 && !(fenc.width & 7) && !(fenc.lines & 7) &&
                         pic->m_picWidth >= 16 && pic->m_picHeight >= 16;
    if (!covered || (fenc.width >> 3) * (fenc.lines >> 3) < gwa.minBlocks)
    {
        if (gwa.enabled) (covered ? gwa.gated : gwa.passed).fetch_add(1, std::memory_order_relaxed);
        x265ref_orig_weightAnalyse(&slice, &frame, &param);
        return;
    }
    const int hshift = 1, vshift = 1;
    const int32_t* mvs[2] = { NULL, NULL };
    for (int list = 0; list < numPredDir; list++)
    {
        Frame* refFrame = slice.m_refFrameList[list][0];
        const int diffPoc = abs(slice.m_poc - refFrame->m_poc);
        if (diffPoc <= param.bframes + 1 && fenc.lowresMvs[list][diffPoc][0].x != 0x7FFF)
        {
            mvs[list] = (const int32_t*)fenc.lowresMvs[list][diffPoc];
            if (!refFrame->m_bChromaExtended)                              /* :333-343 */
            {
                refFrame->m_bChromaExtended = true;
                PicYuv* refPic = refFrame->m_fencPic;
                const int width = refPic->m_picWidth >> hshift, height = refPic->m_picHeight >> vshift;
                extendPicBorder(refPic->m_picOrg[1], refPic->m_strideC, width, height, refPic->m_chromaMarginX, refPic->m_chromaMarginY);
                extendPicBorder(refPic->m_picOrg[2], refPic->m_strideC, width, height, refPic->m_chromaMarginX, refPic->m_chromaMarginY);
            }
        }
    }
    int32_t out[2][3][4], denoms[2][2];
    int rc = 0;
    const ptrdiff_t pad = fenc.lowresPlane[0] - fenc.buffer[0];
    if (gwa.host)
    {
        WaHostParams a;
        memset(&a, 0, sizeof(a));
        a.depth = X265_DEPTH; a.lowres = fenc.lowresPlane[0]; a.lowres_stride = fenc.lumaStride; a.lowres_width = fenc.width; a.lowres_lines = fenc.lines;
        a.lowres_margin_y = (int)(pad / fenc.lumaStride); a.lowres_margin_x = (int)(pad % fenc.lumaStride);
        a.cb = pic->m_picOrg[1]; a.cr = pic->m_picOrg[2]; a.stride_c = pic->m_strideC; a.margin_xc = pic->m_chromaMarginX; a.margin_yc = pic->m_chromaMarginY;
        a.pic_width = pic->m_picWidth; a.pic_height = pic->m_picHeight; a.intra_cost = fenc.intraCost;
        memcpy(a.wp_ssd, fenc.wp_ssd, sizeof(a.wp_ssd)); memcpy(a.wp_sum, fenc.wp_sum, sizeof(a.wp_sum));
        const bool keyed = gla.enabled && gla.host;                        /* the lookahead seam's device copies of the same planes, same keys */
        a.plane_key = keyed ? (gla.instance << 32) | ((uint64_t)(uint32_t)fenc.frameNum + 1) : 0;
        a.nlists = numPredDir;
        for (int list = 0; list < numPredDir; list++)
        {
            Frame* refFrame = slice.m_refFrameList[list][0];
            Lowres& r = refFrame->m_lowres;
            for (int k = 0; k < 4; k++) a.ref[list].lowres[k] = r.lowresPlane[k];
            a.ref[list].cb = refFrame->m_fencPic->m_picOrg[1]; a.ref[list].cr = refFrame->m_fencPic->m_picOrg[2];
            a.ref[list].mvs = mvs[list];
            memcpy(a.ref[list].wp_ssd, r.wp_ssd, sizeof(r.wp_ssd)); memcpy(a.ref[list].wp_sum, r.wp_sum, sizeof(r.wp_sum));
            a.ref[list].plane_key = keyed ? (gla.instance << 32) | ((uint64_t)(uint32_t)r.frameNum + 1) : 0;
        }
        a.weights = &out[0][0][0]; a.denoms = &denoms[0][0];
        rc = gwa.host(&a);
    }
    else
    {
        WaOracleList lists[2];
        memset(lists, 0, sizeof(lists));
        for (int list = 0; list < numPredDir; list++)
        {
            Frame* refFrame = slice.m_refFrameList[list][0];
            Lowres& r = refFrame->m_lowres;
            for (int k = 0; k < 4; k++) lists[list].lowres[k] = r.lowresPlane[k];
            lists[list].cb = refFrame->m_fencPic->m_picOrg[1]; lists[list].cr = refFrame->m_fencPic->m_picOrg[2];
            lists[list].mvs = mvs[list];
            memcpy(lists[list].wp_ssd, r.wp_ssd, sizeof(r.wp_ssd)); memcpy(lists[list].wp_sum, r.wp_sum, sizeof(r.wp_sum));
        }
        const size_t half = (size_t)pic->m_stride * pic->m_picHeight;
        std::vector<pixel> scratch(2 * half + 64);
        gwa.oracle(fenc.lowresPlane[0], fenc.lumaStride, fenc.width, fenc.lines, pic->m_picOrg[1], pic->m_picOrg[2], pic->m_strideC, pic->m_picWidth, pic->m_picHeight,
                   fenc.intraCost, fenc.wp_ssd, fenc.wp_sum, numPredDir, lists, scratch.data(), half, &out[0][0][0], &denoms[0][0]);
    }
    if (rc)
    {
        gwa.failed.fetch_add(1, std::memory_order_relaxed);
        fprintf(stderr, "ref_seam: weightAnalyse provider failed (%d); the reference's loop runs instead\n", rc);
        x265ref_orig_weightAnalyse(&slice, &frame, &param);
        return;
    }
    gwa.served.fetch_add(1, std::memory_order_relaxed);
    WeightParam wp[2][MAX_NUM_REF][3];
    memset(wp, 0, sizeof(wp));
    bool any = false;
    for (int list = 0; list < numPredDir; list++)
    {
        for (int plane = 0; plane < 3; plane++)
        {
            SET_WEIGHT(wp[list][0][plane], out[list][plane][0] != 0, out[list][plane][1], (uint32_t)out[list][plane][2], out[list][plane][3]);
            any |= out[list][plane][0] != 0;
        }
        for (int ref = 1; ref < slice.m_numRefIdx[list]; ref++)          /* :468-474 */
        {
            SET_WEIGHT(wp[list][ref][0], false, 1 << denoms[list][0], (uint32_t)denoms[list][0], 0);
            SET_WEIGHT(wp[list][ref][1], false, 1 << denoms[list][1], (uint32_t)denoms[list][1], 0);
            SET_WEIGHT(wp[list][ref][2], false, 1 << denoms[list][1], (uint32_t)denoms[list][1], 0);
        }
    }
    if (any) gwa.weighted.fetch_add(1, std::memory_order_relaxed);
    memcpy(slice.m_weightPredTable, wp, sizeof(wp));
    if (gwa.verify)
    {
        x265ref_orig_weightAnalyse(&slice, &frame, &param);
        bool same = true;
        for (int list = 0; list < numPredDir; list++)                    /* the reference leaves the entries past numRefIdx uninitialised */
            for (int ref = 0; ref < slice.m_numRefIdx[list]; ref++)
                for (int plane = 0; plane < 3; plane++)
                {
                    const WeightParam& x = wp[list][ref][plane]; const WeightParam& y = slice.m_weightPredTable[list][ref][plane];
                    same &= x.log2WeightDenom == y.log2WeightDenom && x.inputWeight == y.inputWeight && x.inputOffset == y.inputOffset && !x.wtPresent == !y.wtPresent;
                }
#if X265HIP_BINDING_TEST_HOOKS
        if (const char* dir = getenv("X265REF_WA_DUMP"))
        {
            static std::atomic<int> serial{0};
            char path[1024];
            snprintf(path, sizeof(path), "%s/wa_%03d_poc%d.bin", dir, serial.fetch_add(1), slice.m_poc);
            if (FILE* f = fopen(path, "wb"))
            {
                const int marginY = (int)(pad / fenc.lumaStride), marginX = (int)(pad % fenc.lumaStride);
                const size_t lplane = (size_t)fenc.lumaStride * (fenc.lines + 2 * marginY);
                const int chh = pic->m_picHeight >> 1;
                const size_t cplane = (size_t)pic->m_strideC * (chh + 2 * pic->m_chromaMarginY);
                const size_t corg = (size_t)pic->m_chromaMarginY * pic->m_strideC + pic->m_chromaMarginX;
                const int32_t geo[12] = { X265_DEPTH, (int32_t)fenc.lumaStride, fenc.width, fenc.lines, marginX, marginY, (int32_t)pic->m_strideC, (int32_t)pic->m_chromaMarginX,
                                          (int32_t)pic->m_chromaMarginY, pic->m_picWidth, pic->m_picHeight, numPredDir };
                wa_dump_arr(f, "geo", 4, 12, geo);
                wa_dump_arr(f, "cur_lowres", sizeof(pixel), lplane, fenc.buffer[0]);
                wa_dump_arr(f, "cur_cb", sizeof(pixel), cplane, pic->m_picOrg[1] - corg);
                wa_dump_arr(f, "cur_cr", sizeof(pixel), cplane, pic->m_picOrg[2] - corg);
                wa_dump_arr(f, "intra_cost", 4, (size_t)(fenc.width >> 3) * (fenc.lines >> 3), fenc.intraCost);
                wa_dump_arr(f, "cur_wp_ssd", 8, 3, fenc.wp_ssd); wa_dump_arr(f, "cur_wp_sum", 8, 3, fenc.wp_sum);
                for (int list = 0; list < numPredDir; list++)
                {
                    Frame* refFrame = slice.m_refFrameList[list][0];
                    Lowres& r = refFrame->m_lowres;
                    char nm[64];
                    for (int k = 0; k < 4; k++) { snprintf(nm, sizeof(nm), "ref%d_lowres%d", list, k); wa_dump_arr(f, nm, sizeof(pixel), lplane, r.buffer[k]); }
                    snprintf(nm, sizeof(nm), "ref%d_cb", list); wa_dump_arr(f, nm, sizeof(pixel), cplane, refFrame->m_fencPic->m_picOrg[1] - corg);
                    snprintf(nm, sizeof(nm), "ref%d_cr", list); wa_dump_arr(f, nm, sizeof(pixel), cplane, refFrame->m_fencPic->m_picOrg[2] - corg);
                    snprintf(nm, sizeof(nm), "ref%d_mvs", list); wa_dump_arr(f, nm, 4, mvs[list] ? (size_t)2 * (fenc.width >> 3) * (fenc.lines >> 3) : 0, mvs[list]);
                    snprintf(nm, sizeof(nm), "ref%d_wp_ssd", list); wa_dump_arr(f, nm, 8, 3, r.wp_ssd);
                    snprintf(nm, sizeof(nm), "ref%d_wp_sum", list); wa_dump_arr(f, nm, 8, 3, r.wp_sum);
                    int32_t want[3][4];                                   /* what the REFERENCE's function left in the slice for reference 0 of this list */
                    for (int plane = 0; plane < 3; plane++)
                    {
                        const WeightParam& y = slice.m_weightPredTable[list][0][plane];
                        want[plane][0] = y.wtPresent != 0; want[plane][1] = y.inputWeight; want[plane][2] = (int32_t)y.log2WeightDenom; want[plane][3] = y.inputOffset;
                    }
                    snprintf(nm, sizeof(nm), "ref%d_expected", list); wa_dump_arr(f, nm, 4, 12, want);
                }
                fclose(f);
            }
        }
#endif

        if (!same)
        {
            fprintf(stderr, "ref_seam: WEIGHT ANALYSE VERIFY MISMATCH poc %d: served luma (%d %d %u %d), reference (%d %d %u %d)\n", slice.m_poc,
                    wp[0][0][0].wtPresent, wp[0][0][0].inputWeight, wp[0][0][0].log2WeightDenom, wp[0][0][0].inputOffset, slice.m_weightPredTable[0][0][0].wtPresent,
                    slice.m_weightPredTable[0][0][0].inputWeight, slice.m_weightPredTable[0][0][0].log2WeightDenom, slice.m_weightPredTable[0][0][0].inputOffset);
            gwa.mismatches.fetch_add(1, std::memory_order_relaxed);
        }
    }
}
}

/* The intra half of the lookahead: the per-block work of LookaheadTLD::lowresIntraEstimate (slicetype.cpp:716-777: DC, planar and the
 * angular scan of every 8x8 block) as one provider call; the AQ weighting and the sums are the reference's own lines :779-803. */
void LookaheadTLD::lowresIntraEstimate(Lowres& fenc, uint32_t qgSize)
{
    if (!gla.enabled || (!gla.intraHost && !gla.intraOracle) || widthInCU * heightInCU < gla.minBlocks) { x265ref_orig_lowresIntraEstimate(this, &fenc, qgSize); return; }
    const int intraPenalty = 5 * (int)x265_lambda_tab[X265_LOOKAHEAD_QP];
    int rc = 0;
    if (gla.intraHost)
    {
        const ptrdiff_t pad = fenc.lowresPlane[0] - fenc.buffer[0];
        LaIntraHostParams q;
        memset(&q, 0, sizeof(q));
        q.depth = X265_DEPTH; q.stride = fenc.lumaStride; q.width_in_cu = widthInCU; q.height_in_cu = heightInCU;
        q.lines = fenc.lines; q.margin_y = (int)(pad / fenc.lumaStride); q.margin_x = (int)(pad % fenc.lumaStride);
        q.plane = fenc.lowresPlane[0]; q.intra_penalty = intraPenalty;
        q.intra_cost = fenc.intraCost; q.intra_mode = fenc.intraMode; q.lowres_costs = fenc.lowresCosts[0][0];
        q.plane_key = (gla.instance << 32) | ((uint64_t)(uint32_t)fenc.frameNum + 1);
        rc = gla.intraHost(&q);
        if (gla.intraOracle && !rc)
        {
            const int nblk = widthInCU * heightInCU;
            std::vector<int32_t> vc(nblk); std::vector<uint8_t> vmode(nblk); std::vector<uint16_t> vlc(fenc.lowresCosts[0][0], fenc.lowresCosts[0][0] + nblk);
            gla.intraOracle(fenc.lowresPlane[0], fenc.lumaStride, widthInCU, heightInCU, intraPenalty, vc.data(), vmode.data(), vlc.data(), 1);
            if (memcmp(vc.data(), fenc.intraCost, (size_t)nblk * 4) || memcmp(vmode.data(), fenc.intraMode, nblk))
            {
                fprintf(stderr, "ref_seam: LOOKAHEAD INTRA VERIFY MISMATCH frame %d\n", fenc.frameNum);
                gla.mismatches.fetch_add(1, std::memory_order_relaxed);
            }
        }
    }
    else
        gla.intraOracle(fenc.lowresPlane[0], fenc.lumaStride, widthInCU, heightInCU, intraPenalty, fenc.intraCost, fenc.intraMode, fenc.lowresCosts[0][0], 1);
    if (rc)
    {
        gla.failed.fetch_add(1, std::memory_order_relaxed);
        fprintf(stderr, "ref_seam: lookahead intra provider failed (%d); the reference's loop runs instead\n", rc);
        x265ref_orig_lowresIntraEstimate(this, &fenc, qgSize);
        return;
    }
    gla.intraServed.fetch_add(1, std::memory_order_relaxed);
    int costEst = 0, costEstAq = 0;
    for (int cuY = 0; cuY < heightInCU; cuY++)
    {
        fenc.rowSatds[0][0][cuY] = 0;
        for (int cuX = 0; cuX < widthInCU; cuX++)
        {
            const int cuXY = cuX + cuY * widthInCU;
            const int icost = fenc.intraCost[cuXY];
            const bool bFrameScoreCU = (cuX > 0 && cuX < widthInCU - 1 && cuY > 0 && cuY < heightInCU - 1) || widthInCU <= 2 || heightInCU <= 2;
            const int* scale = qgSize == 8 ? fenc.invQscaleFactor8x8 : fenc.invQscaleFactor;
            const int icostAq = (bFrameScoreCU && fenc.invQscaleFactor) ? ((icost * scale[cuXY] + 128) >> 8) : icost;
            if (bFrameScoreCU) { costEst += icost; costEstAq += icostAq; }
            fenc.rowSatds[0][0][cuY] += icostAq;
        }
    }
    fenc.costEst[0][0] = costEst;
    fenc.costEstAq[0][0] = costEstAq;
}

extern "C" {

/* provider: see the header comment; geometry = the PicYuv layout of the encode about to start.  Call before x265ref_encode. */
int x265ref_seam_configure(void* ctx, void* submit, void* submit_batch, void* surface, void* ready, int range, int surf_format, int slots,
                           int width, int height, intptr_t stride, int margin_x, int margin_y, int min_pu, int verify)
{
    if (slots < 1 || slots > MAX_SLOTS || range < 1 || (width & 63) || (height & 63)) return -1;
    if (surf_format != SURF_I32 && X265_DEPTH != 8) return -2;
    g_gated = (width / 64) * (height / 64) < g_minCtus;
    if (surf_format == SURF_PACKED_T && 45 * ((2 * range + 4) / 4) * 16 > 65535) return -3;          /* Part::off is 16 bits */
    g.p.ctx = ctx;
    g.p.submit = (int (*)(void*, int, const void*, uint64_t, const void*))submit;
    g.p.submit_batch = (int (*)(void*, int, const int*, const void*, uint64_t, const void* const*, int*))submit_batch;      /* may be NULL */
    g.p.surface = (const void* (*)(void*, int))surface;
    g.p.ready = (const volatile int* (*)(void*, int))ready;
    g.p.picture_rows = NULL; g.p.pair_open = NULL; g.p.pair_open_w = NULL; g.p.centres = NULL; g.p.streamed = false; g.p.min_level = 0; g.p.layout = 0;
    g.saturated = 0;
    g.weightedPairs = 0; g.weightedHits = 0; g.weightedCalls = 0;
    memset(g.fencs, 0, sizeof(g.fencs));
    g.instance++;
    g.rowsPublished = 0; g.rowsRefused = 0; g.torn = 0;
    g.p.range = range; g.p.surf_format = surf_format; g.p.slots = slots;
    g.p.width = width; g.p.height = height; g.p.stride = stride; g.p.margin_x = margin_x; g.p.margin_y = margin_y;
    g.p.min_pu = min_pu < 8 ? 8 : min_pu;
    g.batchMisses = getenv("X265REF_SEAM_BATCH_MISS") != NULL;
    g.nc = 2 * range + 1; g.ng = (g.nc + 3) >> 2; g.pitch = 4 * g.ng;
    g.strideMagic = (((uint64_t)1 << 40) + (uint64_t)stride - 1) / (uint64_t)stride;
    g.groupBytes = surf_format == SURF_I32 ? GROUP_I32 : GROUP_PACKED;
    g.ctusW = width / 64;
    g.ctuBytes = (size_t)g.nc * g.ng * g.groupBytes;
    memset(g.pairs, 0, sizeof(g.pairs));
    g.verify = (verify & 1) != 0 || (getenv("X265REF_SEAM_VERIFY") && atoi(getenv("X265REF_SEAM_VERIFY")));
    g.wait = (verify & 2) != 0;
    g.hits = g.outside = g.notReady = g.meCalls = g.meServed = g.submits = g.mismatches = g.noSlot = g.foreign = 0;
    g_gate.h0 = g_gate.o0 = 0; g_gate.closed = false; g_gate.skipped = 0; g_gate.closings = 0;
    g.epoch.fetch_add(1);
    g.enabled = !g_gated;
    return 0;
}

/* row-granular provider (x265hip_me_stream_picture_rows / _pair_open / _surface / _ready signatures): serves under any --frame-threads.
 * record_bytes = x265hip_me_stream_record_bytes (the whole record, or its 16x16-and-up tail with min_level 1). */
int x265ref_seam_configure_streamed(void* ctx, void* picture_rows, void* pair_open, void* pair_open_weighted, void* surface, void* ready, void* centres, int layout,
                                    int range, int surf_format, int min_level, int slots, int width, int height, intptr_t stride, int margin_x, int margin_y,
                                    int min_pu, int verify)
{
    if (!picture_rows || !pair_open || surf_format == SURF_PACKED_T || min_level < 0 || min_level > (layout ? 2 : 1) || layout < 0 || layout > 1) return -4;
    const int rc = x265ref_seam_configure(ctx, NULL, NULL, surface, ready, range, surf_format, slots, width, height, stride, margin_x, margin_y,
                                          min_level > 1 && min_pu < 32 ? 32 : (min_level && min_pu < 16 ? 16 : min_pu), verify);
    if (rc) return rc;
    g.p.picture_rows = (int (*)(void*, uint64_t, const void*, int, int))picture_rows;
    g.p.pair_open = (int (*)(void*, int, uint64_t, uint64_t))pair_open;
    g.p.pair_open_w = (int (*)(void*, int, uint64_t, uint64_t, const void*))pair_open_weighted;      /* NULL: weighted references pass */
    g.p.centres = (const int16_t* (*)(void*, int))centres;      /* NULL: windows centred on (0, 0) */
    g.p.layout = layout;
    g.p.min_level = min_level;
    g.p.streamed = true;
    if (layout)
        g.ctuBytes = (size_t)g.nc * g.pitch * ((min_level ? 0 : 64 * 2) + (min_level > 1 ? 0 : 16 * 2) + 5 * 4);          /* x265hip_stream_planes_ctu_bytes */
    else if (min_level)
    {
        g.groupBytes = surf_format == SURF_I32 ? 336 : 208;          /* X265HIP_SURF_TAIL_BYTES_I32 / _PACKED */
        g.ctuBytes = (size_t)g.nc * g.ng * g.groupBytes;
    }
    return 0;
}

void x265ref_seam_disable(void) { g.enabled = false; gla.enabled = false; gs.enabled = false; gaq.enabled = false; gwa.enabled = false; gc.enabled = false; }

/* cost-table seam: provider = x265hip_cost_stream_picture_rows / _pair_open / _tables / _ready signatures (NULL pair_open = off); geometry = the PicYuv
 * buffers of the encode about to start; pu_rects = int [npu][4] (x, y, w, h: x265hip_cost_pu_rect), positions = int8 [npos][2] (x265hip_cost_positions),
 * record_bytes / ctu_bytes as the service lays the records out, cover_mask bit s = the position set contains every position a --subme s refinement can reach from a record's
 * vector, sad_costs_offset > 0 = the records also carry the SAD-typed costs at that byte offset (the service may hold a LARGER set than the encode's own --subme needs: refinements that start from a fractional predictor then stay inside it more often).  flags: 1 = verify every served value against the reference's own function, 2 = wait
 * for records, 4 = ignore the size gate, 8 = rank the candidates by SAD alone (no vector-cost table travels with the pairs).  window = the service's (the table covers
 * displacements of +-window around each CTU's centre).  Needs the SAD seam's configure first when the size gate is to apply (it decides g_gated). */
int x265ref_cost_seam_configure(void* ctx, void* picture_rows, void* pair_open, void* tables, void* ready, int slots, int width, int height, intptr_t stride, intptr_t stride_c,
                                int margin_x, int margin_y, int candidates, int subme, int chroma, const int* pu_rects, int npu, const int8_t* positions, int npos,
                                int record_bytes, size_t ctu_bytes, int window, unsigned cover_mask, int sad_costs_offset, int flags)
{
    gc.enabled = false;
    if (!pair_open) return 0;
    if (slots < 1 || slots > MAX_SLOTS || !picture_rows || !tables || !ready || !pu_rects || !positions || npu < 1 || npos < 1 || npos > 169 || candidates < 1 || candidates > 2 ||
        (width & 63) || (height & 63) || record_bytes < 8 + 2 * npos || sad_costs_offset < 0 || (sad_costs_offset && (sad_costs_offset < 8 + 2 * npos || record_bytes < sad_costs_offset + 4 + 2 * npos)) || ctu_bytes < (size_t)record_bytes * npu * candidates || window < 0 || window > 32) return -1;
    gc.p.ctx = ctx;
    gc.p.picture_rows = (int (*)(void*, uint64_t, const void*, const void*, const void*, int, int))picture_rows;
    gc.p.pair_open = (int (*)(void*, int, uint64_t, uint64_t, const void*, unsigned, const uint16_t*))pair_open;
    gc.p.window = window; gc.p.coverMask = cover_mask; gc.p.rec2Off = sad_costs_offset;
    gc.p.tables = (const void* (*)(void*, int))tables;
    gc.p.ready = (const volatile int* (*)(void*, int))ready;
    gc.p.slots = slots; gc.p.width = width; gc.p.height = height; gc.p.stride = stride; gc.p.strideC = stride_c; gc.p.marginX = margin_x; gc.p.marginY = margin_y;
    gc.p.K = candidates; gc.p.subme = subme; gc.p.chroma = chroma; gc.p.recBytes = record_bytes; gc.p.npos = npos; gc.p.npu = npu; gc.p.ctuBytes = ctu_bytes;
    for (int i = 0; i < 13 * 13; i++) gc.posMap[i] = -1;
    for (int i = 0; i < npos; i++)
    {
        const int dx = positions[2 * i] + 6, dy = positions[2 * i + 1] + 6;
        if ((unsigned)dx > 12u || (unsigned)dy > 12u) return -1;
        gc.posMap[dy * 13 + dx] = (int16_t)i;
    }
    memset(gc.puIndex, -1, sizeof(gc.puIndex));
    for (int i = 0; i < npu; i++)
    {
        const int x = pu_rects[4 * i], y = pu_rects[4 * i + 1], w = pu_rects[4 * i + 2], h = pu_rects[4 * i + 3];
        if (((x | y | w | h) & 7) || w < 8 || h < 8 || x + w > 64 || y + h > 64) return -1;
        gc.puIndex[w / 8 - 1][h / 8 - 1][y / 8][x / 8] = (int16_t)i;
    }
    gc.instance++;
    memset(gc.pairs, 0, sizeof(gc.pairs)); memset(gc.fencs, 0, sizeof(gc.fencs));
    gc.verify = (flags & 1) != 0; gc.wait = (flags & 2) != 0; gc.useMvCost = (flags & 8) == 0;
    gc.served = 0; gc.servedSad = 0; gc.otherVector = 0; gc.notReady = 0; gc.saturated = 0; gc.torn = 0; gc.noContext = 0; gc.contexts = 0; gc.pairsOpened = 0; gc.noSlot = 0;
    gc.rowsPublished = 0; gc.rowsRefused = 0; gc.mismatches = 0; gc.weightedPairs = 0;
    gc.epoch.fetch_add(1);
    gc.enabled = (flags & 4) ? true : !g_gated;
    return 0;
}

/* out[14]: SATD comparisons served from records, passed on because the search did not end on a record's vector (or left the position set), because the row's records had
 * not landed, because a delta was saturated, because the slot was reopened under the read; motionEstimate calls seen, of those without a usable context; pairs opened,
 * pair requests without a free slot, rows of reconstructed pictures handed to the provider / refused by it, verify mismatches, pairs on weighted references */
void x265ref_cost_seam_stats(uint64_t* out)
{
    out[0] = gc.served; out[1] = gc.otherVector; out[2] = gc.notReady; out[3] = gc.saturated; out[4] = gc.torn; out[5] = gc.contexts; out[6] = gc.noContext;
    out[7] = gc.pairsOpened; out[8] = gc.noSlot; out[9] = gc.rowsPublished; out[10] = gc.rowsRefused; out[11] = gc.mismatches; out[12] = gc.weightedPairs;
    out[13] = gc.servedSad;          /* SAD-typed comparisons (the predictor candidates') served from records */
}

/* sub-sample seam: provider = x265hip_phase_cache_submit / _planes / _ready signatures (NULL submit = off); geometry = the PicYuv
 * buffers of the encode about to start.  flags: 1 = verify every served call against the reference's own function, 2 = wait for planes */
int x265ref_subpel_seam_configure(void* ctx, void* submit, void* planes, void* ready, int slots, intptr_t stride, int rows, intptr_t stride_c, int rows_c, int flags)
{
    gs.enabled = false;
    if (!submit) return 0;
    if (slots < 1 || slots > MAX_SLOTS || !planes || !ready) return -1;
    gs.p.ctx = ctx;
    gs.p.submit = (int (*)(void*, int, const void*, const void*, const void*))submit;
    gs.p.planes = (const void* (*)(void*, int, int))planes;
    gs.p.ready = (const volatile int* (*)(void*, int))ready;
    gs.p.open = NULL; gs.p.rows_fn = NULL; gs.p.progress = NULL; gs.p.streamed = false; gs.p.ctuRows = 0;
    gs.p.slots = slots; gs.p.stride = stride; gs.p.rows = rows; gs.p.strideC = stride_c; gs.p.rowsC = rows_c;
    memset(gs.e, 0, sizeof(gs.e));
    gs.verify = (flags & 1) != 0; gs.wait = (flags & 2) != 0;
    gs.served = gs.notReady = gs.noContext = gs.submits = gs.mismatches = gs.noSlot = 0; gs.rowsPublished = 0; gs.torn = 0;
    gs.weightedViews = 0; gs.weightedServed = 0;
    gs.epoch.fetch_add(1);
    gs.enabled = true;
    return 0;
}

/* row-granular provider (x265hip_phase_stream_view_open / _picture_rows / _planes / _progress signatures): the producer hook feeds it
 * pictures, searches open views (weighted or not); serves under any --frame-threads */
int x265ref_subpel_seam_configure_streamed(void* ctx, void* open, void* rows_fn, void* planes, void* progress, int slots, intptr_t stride, int rows,
                                           intptr_t stride_c, int rows_c, int ctu_rows, int flags)
{
    gs.enabled = false;
    if (!open) return 0;
    if (slots < 1 || slots > MAX_SLOTS || !rows_fn || !planes || !progress || rows_c <= 0) return -1;
    gs.p.ctx = ctx;
    gs.p.submit = NULL; gs.p.ready = NULL;
    gs.p.open = (int (*)(void*, int, uint64_t, const void*, unsigned))open;
    gs.p.rows_fn = (int (*)(void*, uint64_t, const void*, const void*, const void*, int, int))rows_fn;
    gs.instance++;
    gs.p.planes = (const void* (*)(void*, int, int))planes;
    gs.p.progress = (const volatile uint64_t* (*)(void*, int))progress;
    gs.p.streamed = true; gs.p.ctuRows = ctu_rows;
    gs.p.slots = slots; gs.p.stride = stride; gs.p.rows = rows; gs.p.strideC = stride_c; gs.p.rowsC = rows_c;
    memset(gs.e, 0, sizeof(gs.e));
    gs.verify = (flags & 1) != 0; gs.wait = (flags & 2) != 0;
    gs.served = gs.notReady = gs.noContext = gs.submits = gs.mismatches = gs.noSlot = 0; gs.rowsPublished = 0; gs.torn = 0;
    gs.weightedViews = 0; gs.weightedServed = 0;
    gs.epoch.fetch_add(1);
    gs.enabled = !g_gated;          /* the size gate of the SAD seam's configure (made first) holds for this seam too */
    return 0;
}

/* pictures of fewer CTUs than this keep the reference's own search untouched (default 1000: the search seams serve from 4K up); call it
 * BEFORE the configure calls.  Returns whether the last configure was gated. */
int x265ref_seam_min_ctus(int min_ctus) { if (min_ctus >= 0) g_minCtus = min_ctus; return g_gated ? 1 : 0; }

#if X265HIP_BINDING_TEST_HOOKS
/* measurement mode of the integer-vector predictor (see gpp): on = 1 / 0 (< 0: leave); out[7] (may be NULL): refinements seen, without a usable SAD context, ended outside the
 * window, and - of those inside - ended on the window's SAD minimum / among its 2 / 4 / 8 smallest SADs */
void x265ref_predict_probe(int on, uint64_t* out)
{
    if (on >= 0) { gpp.on = on != 0; gpp.total = 0; gpp.noCtx = 0; gpp.outside = 0; gpp.top1 = 0; gpp.top2 = 0; gpp.top4 = 0; gpp.top8 = 0; }
    if (out) { out[0] = gpp.total; out[1] = gpp.noCtx; out[2] = gpp.outside; out[3] = gpp.top1; out[4] = gpp.top2; out[5] = gpp.top4; out[6] = gpp.top8; }
}
#endif


/* the hit-rate gate: window = lookups per decision (0 = gate off, < 0 = leave), pct = the share of served lookups below which no new pairs are opened.
 * out[3] (may be NULL): searches that went to the host because the gate was closed, times the gate closed, whether it is closed now, window */
void x265ref_seam_hit_rate_gate(long long window, int pct, uint64_t* out)
{
    if (window >= 0) { g_gateWindow = (uint64_t)window; g_gate.h0 = g_gate.o0 = 0; g_gate.closed = false; g_gate.skipped = 0; g_gate.closings = 0; }
    if (pct >= 0) g_gatePct = pct;
    if (out) { out[0] = g_gate.skipped; out[1] = g_gate.closings; out[2] = g_gate.closed ? 1 : 0; out[3] = g_gateWindow; }
}

/* out[4]: rows of reconstructed pictures handed to the SAD provider / refused by it, rows handed to the phase provider, lookups dropped
 * because their slot was reopened under the read (SAD + sub-sample) */
void x265ref_seam_stream_stats(uint64_t* out)
{
    out[0] = g.rowsPublished; out[1] = g.rowsRefused; out[2] = gs.rowsPublished; out[3] = g.torn + gs.torn;
}

/* out[6]: pairs opened on a weighted reference, motionEstimate calls on weighted references that got a lookup context, lookups served on
 * weighted references, phase-plane views opened on weighted references, subpelCompare calls served from weighted views; lookups that
 * met a saturated 16-bit entry of the planes layout (answered by the host) */
void x265ref_seam_weighted_stats(uint64_t* out)
{
    out[0] = g.weightedPairs; out[1] = g.weightedCalls; out[2] = g.weightedHits; out[3] = gs.weightedViews; out[4] = gs.weightedServed;
    out[5] = g.saturated;
}

/* out[6]: subpelCompare calls served from phase planes, passed on because the planes had not arrived, searches without a usable
 * context (weighted / foreign geometry), pictures submitted, verify mismatches, searches without a free slot */
void x265ref_subpel_seam_stats(uint64_t* out)
{
    out[0] = gs.served; out[1] = gs.notReady; out[2] = gs.noContext; out[3] = gs.submits; out[4] = gs.mismatches; out[5] = gs.noSlot;
}

/* lookahead seam: host_fn = x265hip_lowres_cost_host (the product) or NULL; oracle_fn = x265oracle_lowres_cost_wp_d<depth> (CPU checker,
 * tests only) or NULL.  Both NULL switches the seam off. */
int x265ref_lookahead_seam_configure(void* host_fn, void* oracle_fn, void* intra_host_fn, void* intra_oracle_fn)
{
    gla.host = (la_host_fn)host_fn;
    gla.oracle = (la_oracle_fn)oracle_fn;
    gla.intraHost = (la_intra_host_fn)intra_host_fn;
    gla.intraOracle = (la_intra_oracle_fn)intra_oracle_fn;
    gla.served = 0; gla.passed = 0; gla.failed = 0; gla.intraServed = 0; gla.mismatches = 0; gla.gated = 0;
    gla.instance++;
    gla.enabled = host_fn || oracle_fn;
    return 0;
}

/* out[4]: frame cost estimates served by the provider, passed to the reference's loop (HME / cooperative slices / qg 8), failed,
 * intra estimates served */
void x265ref_lookahead_seam_stats(uint64_t* out) { out[0] = gla.served; out[1] = gla.passed; out[2] = gla.failed; out[3] = gla.intraServed; }
/* pictures with fewer lowres 8x8 blocks than this keep the reference's own loop (default 16384: the seam serves from 4K up); returns the
 * number of estimates the gate has sent there since the last configure */
uint64_t x265ref_lookahead_seam_min_blocks(int min_blocks) { if (min_blocks >= 0) gla.minBlocks = min_blocks; return gla.gated; }
uint64_t x265ref_lookahead_seam_mismatches(void) { return gla.mismatches; }

/* weightAnalyse seam: host_fn = x265hip_weight_analyse_host (the product) or NULL; oracle_fn = x265oracle_weight_analyse_d<depth> (CPU checker)
 * or NULL; both NULL: off.  verify: the reference's own weightAnalyse runs after every served slice and the weight tables are compared.
 * min_blocks: pictures with fewer lowres 8x8 blocks keep the reference's loop. */
int x265ref_weight_seam_configure(void* host_fn, void* oracle_fn, int verify, int min_blocks)
{
    gwa.host = (wa_host_fn)host_fn;
    gwa.oracle = (wa_oracle_fn)oracle_fn;
    gwa.verify = verify != 0;
    gwa.minBlocks = min_blocks < 0 ? 0 : min_blocks;
    gwa.served = 0; gwa.passed = 0; gwa.failed = 0; gwa.mismatches = 0; gwa.gated = 0; gwa.weighted = 0;
    gwa.enabled = host_fn || oracle_fn;
    return 0;
}
/* out[6]: slices served, passed to the reference's loop (not 4:2:0 / lowres size not a multiple of 8), failed, verify mismatches, gated by size,
 * served slices that came back with at least one weight present */
void x265ref_weight_seam_stats(uint64_t* out)
{
    out[0] = gwa.served; out[1] = gwa.passed; out[2] = gwa.failed; out[3] = gwa.mismatches; out[4] = gwa.gated; out[5] = gwa.weighted;
}

/* AQ seam: host_fn = x265hip_aq_frame_host (the product) or NULL; oracle_fn = x265oracle_aq_frame_d<depth> (CPU checker, GPU-less tests) or
 * NULL; both NULL: off.  verify: the reference's own calcAdaptiveQuantFrame runs after every served picture and its arrays are compared
 * with the served ones bit for bit.  min_blocks: pictures with fewer quantisation groups keep the reference's loop. */
int x265ref_aq_seam_configure(void* host_fn, void* oracle_fn, int verify, int min_blocks)
{
    gaq.host = (aq_host_fn)host_fn;
    gaq.oracle = (aq_oracle_fn)oracle_fn;
    gaq.verify = verify != 0;
    gaq.minBlocks = min_blocks < 0 ? 0 : min_blocks;
    gaq.served = 0; gaq.passed = 0; gaq.failed = 0; gaq.mismatches = 0; gaq.gated = 0;
    gaq.enabled = host_fn || oracle_fn;
    return 0;
}
/* out[5]: pictures served, passed to the reference's loop (a mode / option the service does not cover), failed, verify mismatches, gated by size */
void x265ref_aq_seam_stats(uint64_t* out) { out[0] = gaq.served; out[1] = gaq.passed; out[2] = gaq.failed; out[3] = gaq.mismatches; out[4] = gaq.gated; }

/* HOST-ONLY control (no provider, no GPU): sad_x3 / sad_x4 of every partition answered by N calls of the table's own `sad` - what the lookup stubs do for
 * candidates they cannot serve.  g++ -O3 turns the reference's single-reference SAD loop (pixel.cpp:60-72) into psadbw but leaves the three- / four-reference
 * loops (pixel.cpp:74-119) scalar, so this table alone is faster than the C table; an encode with it next to the C table and the seams separates what the
 * services contribute from what the stubs' host path contributes (tools/encoder_bench.py --tables c,csplit,seam). */
int x265ref_split_fill_table(void* table, size_t bytes, int depth)
{
    if (!table || bytes != sizeof(EncoderPrimitives) || depth != X265_DEPTH) return -1;
    int n = 0;
    InstallSplit<0>::run(*(EncoderPrimitives*)table, n);
    return n;
}

#if X265HIP_BINDING_TEST_HOOKS
/* the control under tools/encoder_profile.py: counting thunks first, the split on top (its single SADs are then counted in the `sad` family) */
int x265ref_split_fill_table_profiled(void* table, size_t bytes, int depth)
{
    const int a = x265ref_profile_fill_table(table, bytes, depth);
    if (a < 0) return a;
    const int b = x265ref_split_fill_table(table, bytes, depth);
    return b < 0 ? b : a + b;
}
#endif


/* table filler with the x265hip_setup_primitives signature: installs the lookup stubs over the host's own sad family */
int x265ref_seam_fill_table(void* table, size_t bytes, int depth)
{
    if (!table || bytes != sizeof(EncoderPrimitives) || depth != X265_DEPTH) return -1;
    int n = 0;
    if (g_gated)                    /* a picture below the size gate: no lookup stub is installed, the encode is the reference's own (or the host-only control's) */
    {
        if (getenv("X265REF_SEAM_SPLIT_REST")) InstallSplit<0>::run(*static_cast<EncoderPrimitives*>(table), n);
        return n;
    }
    if (!g.enabled) return -1;
    /* X265REF_SEAM_SPLIT_REST=1: everything the services do NOT answer - partitions below min_pu, searches without a context - goes through the host-only control's
     * split sad_x3 / sad_x4 as well, so that an encode with the seams and an encode with the control table differ by the services alone */
    if (getenv("X265REF_SEAM_SPLIT_REST")) InstallSplit<0>::run(*static_cast<EncoderPrimitives*>(table), n);
    Install<0>::run(*static_cast<EncoderPrimitives*>(table), n);
    return n;
}

#if X265HIP_BINDING_TEST_HOOKS
/* the seam's lookup stubs over ref_profile.cpp's cycle-counting thunks (the stubs' fall-backs then count as host sad time), and the
 * stage timers of this file switched on: where the encode's time goes WITH the seams in place (tools/encoder_profile.py --seams) */
int x265ref_seam_fill_table_profiled(void* table, size_t bytes, int depth)
{
    const int a = x265ref_profile_fill_table(table, bytes, depth);
    if (a < 0) return a;
    gp.meCyc = gp.meCalls = gp.subCyc = gp.subCalls = gp.laCyc = gp.laCalls = gp.rowCyc = gp.rows = 0; gp.lookCyc = gp.lookups = gp.ctxCyc = 0;
    gp.on = true;
    const int b = g.enabled ? x265ref_seam_fill_table(table, bytes, depth) : 0;
    return b < 0 ? b : a + b;
}
/* out[12]: cycles / calls of MotionEstimate::motionEstimate (whole, wrapper included), subpelCompare, CostEstimateGroup::estimateFrameCost,
 * the row hand-over inside FrameFilter::processPostRow, the SAD lookups themselves (served or not), the context set-up of the wrapper */
void x265ref_seam_profile_report(uint64_t* out)
{
    out[0] = gp.meCyc; out[1] = gp.meCalls; out[2] = gp.subCyc; out[3] = gp.subCalls; out[4] = gp.laCyc; out[5] = gp.laCalls; out[6] = gp.rowCyc; out[7] = gp.rows;
    out[8] = gp.lookCyc; out[9] = gp.lookups; out[10] = gp.ctxCyc; out[11] = gp.meCalls;
    gp.on = false;
}
#endif


/* out[10]: lookups served, outside the window, row not ready, motionEstimate calls, calls with a lookup context, pair submits,
 * verify mismatches, calls without a free slot, calls on foreign geometry / frame threads, verify flag */
void x265ref_seam_stats(uint64_t* out)
{
    out[0] = g.hits; out[1] = g.outside; out[2] = g.notReady; out[3] = g.meCalls; out[4] = g.meServed; out[5] = g.submits;
    out[6] = g.mismatches; out[7] = g.noSlot; out[8] = g.foreign; out[9] = g.verify;
}

} // extern "C"
