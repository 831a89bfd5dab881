// table_stubs.hip - TABLE LAYER: fills an EncoderPrimitives-layout table with host-pointer stubs
// that have exactly the reference's slot signatures (source/common/primitives.h:133-234) and run the
// batch-layer HIP kernels on a batch of one.  Compiled once per bit depth (-DX265HIP_DEPTH=8|10|12):
// the assignments below are type-checked against the generated mirror include/x265hip_table.h.
//
// Install order for a real encoder (reference primitives.cpp:248-282): let the host fill its own
// C table first (setupCPrimitives + setupAliasPrimitives), then call x265hip_setup_primitives() on
// it BEFORE x265_encoder_open(); slots we do not implement keep the host's entries.
//
// Every stub: pack the operands the reference function would touch (including filter aprons and
// carried sign buffers) into the calling thread's pinned staging buffer, one H2D copy, the batch
// kernel on a single job, one D2H copy, scatter the outputs back with the caller's strides.
#ifndef X265HIP_DEPTH
#error "compile with -DX265HIP_DEPTH=8|10|12"
#endif
#include "x265hip_table.h"
#include "host_stage.h"

namespace x265hip {
namespace {

typedef x265hip_pixel pixel;
typedef x265hip_sse_t sse_t;
constexpr int D = X265HIP_DEPTH;
constexpr int ES = sizeof(pixel);

// pack a single job record and return a reference that resolves to its DEVICE pointer only when it is
// handed to the launch: a later st.alloc() may regrow (and move) the staging buffers, so no device
// address may be taken before the last alloc of a call.
struct JobRef
{
    const ThreadStage* st; size_t off;
    operator const x265hip_job*() const { return st->dptr<const x265hip_job>(off); }
};
static JobRef put_job(ThreadStage& st, const x265hip_job& jb)
{
    const JobRef r = { &st, st.in1d(&jb, sizeof(jb)) };
    return r;
}
static x265hip_plane plane(ThreadStage& st, size_t off, intptr_t stride) { x265hip_plane p = { st.dptr<void>(off), stride }; return p; }

// ---------------------------------------------------------------- pixel-compare family
static uint64_t cmp_one(int kind, int w, int h, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t oa = st.in2d(a, sa, w, h, ES);
    const size_t ob = st.in2d(b, sb, w, h, ES);
    const size_t oo = st.alloc(8);
    st.upload();
    st.require(x265hip_pixelcmp_batch(kind, D, w, h, st.dptr<void>(oa), w, nullptr, 0, st.dptr<void>(ob), w, nullptr, 0,
                                      1, st.dptr<uint64_t>(oo), st.stream), "pixelcmp");
    st.download(oo, 8);
    return *st.hptr<uint64_t>(oo);
}
template <int KIND, int W, int H> static int cmp_stub(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{ return (int)cmp_one(KIND, W, H, a, sa, b, sb); }
template <int W, int H> static sse_t sse_stub(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{ return (sse_t)cmp_one(X265HIP_CMP_SSE_PP, W, H, a, sa, b, sb); }

// sad_x3 / sad_x4 (pixel.cpp:74-119): one fenc block at the fixed FENC_STRIDE 64, N refs sharing a stride
template <int W, int H, int N> static void sad_xn(const pixel* fenc, const pixel* const* refs, intptr_t rs, int32_t* res)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t oa = st.in2d(fenc, 64, W, H, ES);
    size_t ob = 0;
    for (int i = 0; i < N; i++)
    {
        const size_t o = st.in2d(refs[i], rs, W, H, ES);
        if (!i) ob = o;
    }
    const size_t pitch = (((size_t)W * H * ES) + 63) & ~(size_t)63;   // in2d starts every block on a 64-byte boundary
    const size_t oo = st.alloc(8 * N);
    st.upload();
    st.require(x265hip_pixelcmp_batch(X265HIP_CMP_SAD, D, W, H, st.dptr<void>(oa), W, nullptr, 0,
                                      st.dptr<void>(ob), W, nullptr, (int64_t)(pitch / ES),
                                      N, st.dptr<uint64_t>(oo), st.stream), "sad_xN");
    st.download(oo, 8 * N);
    for (int i = 0; i < N; i++) res[i] = (int32_t)st.hptr<uint64_t>(oo)[i];
}
template <int W, int H> static void sad_x3_stub(const pixel* f, const pixel* r0, const pixel* r1, const pixel* r2, intptr_t rs, int32_t* res)
{ const pixel* r[3] = { r0, r1, r2 }; sad_xn<W, H, 3>(f, r, rs, res); }
template <int W, int H> static void sad_x4_stub(const pixel* f, const pixel* r0, const pixel* r1, const pixel* r2, const pixel* r3, intptr_t rs, int32_t* res)
{ const pixel* r[4] = { r0, r1, r2, r3 }; sad_xn<W, H, 4>(f, r, rs, res); }

// ---------------------------------------------------------------- interpolation family
template <int KIND, int N, int W, int H, typename S, typename Dt>
static void ip_core(const S* src, intptr_t ss, Dt* dst, intptr_t ds, int a0, int a1)
{
    constexpr bool horiz = KIND == X265HIP_IP_HPP || KIND == X265HIP_IP_HPS || KIND == X265HIP_IP_HVPP;
    constexpr bool vert = KIND == X265HIP_IP_VPP || KIND == X265HIP_IP_VPS || KIND == X265HIP_IP_VSP || KIND == X265HIP_IP_VSS || KIND == X265HIP_IP_HVPP;
    const bool ext = (KIND == X265HIP_IP_HPS && a1) || vert;
    const int ax = (horiz && KIND != X265HIP_IP_P2S) ? N / 2 - 1 : 0, ay = ext ? N / 2 - 1 : 0;
    const int tw = W + (ax ? N - 1 : 0), th = H + (ay ? N - 1 : 0);
    const int oh = (KIND == X265HIP_IP_HPS && a1) ? H + N - 1 : H;
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t oa = st.in2d(src - (intptr_t)ay * ss - ax, ss, tw, th, sizeof(S));
    x265hip_job jb = {};
    jb.off[0] = (int64_t)ay * tw + ax; jb.arg[0] = a0; jb.arg[1] = a1;
    const JobRef dj = put_job(st, jb);
    const size_t od = st.alloc((size_t)W * oh * sizeof(Dt));
    st.upload();
    st.require(x265hip_interp_batch(KIND, D, N, W, H, plane(st, oa, tw), plane(st, od, W), dj, 1, st.stream), "interp");
    st.download(od, (size_t)W * oh * sizeof(Dt));
    st.out2d(od, dst, ds, W, oh, sizeof(Dt));
}
template <int N, int W, int H> static void hpp_stub(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int idx) { ip_core<X265HIP_IP_HPP, N, W, H>(s, ss, d, ds, idx, 0); }
template <int N, int W, int H> static void hps_stub(const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int idx, int ext) { ip_core<X265HIP_IP_HPS, N, W, H>(s, ss, d, ds, idx, ext); }
template <int N, int W, int H> static void vpp_stub(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int idx) { ip_core<X265HIP_IP_VPP, N, W, H>(s, ss, d, ds, idx, 0); }
template <int N, int W, int H> static void vps_stub(const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int idx) { ip_core<X265HIP_IP_VPS, N, W, H>(s, ss, d, ds, idx, 0); }
template <int N, int W, int H> static void vsp_stub(const int16_t* s, intptr_t ss, pixel* d, intptr_t ds, int idx) { ip_core<X265HIP_IP_VSP, N, W, H>(s, ss, d, ds, idx, 0); }
template <int N, int W, int H> static void vss_stub(const int16_t* s, intptr_t ss, int16_t* d, intptr_t ds, int idx) { ip_core<X265HIP_IP_VSS, N, W, H>(s, ss, d, ds, idx, 0); }
template <int N, int W, int H> static void hvpp_stub(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int ix, int iy) { ip_core<X265HIP_IP_HVPP, N, W, H>(s, ss, d, ds, ix, iy); }
template <int W, int H> static void p2s_stub(const pixel* s, intptr_t ss, int16_t* d, intptr_t ds) { ip_core<X265HIP_IP_P2S, 8, W, H>(s, ss, d, ds, 0, 0); }

// ---------------------------------------------------------------- transforms
template <int KIND, int N> static void fwd_tr_stub(const int16_t* src, int16_t* dst, intptr_t srcStride)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t oa = st.in2d(src, srcStride, N, N, 2);
    x265hip_job jb = {};
    const JobRef dj = put_job(st, jb);
    const size_t od = st.alloc(N * N * 2);
    st.upload();
    st.require(x265hip_transform_batch(KIND, D, N, plane(st, oa, N), plane(st, od, N), dj, 1, N >= 16, st.stream), "transform");
    st.download(od, N * N * 2);
    memcpy(dst, st.hptr<int16_t>(od), N * N * 2);
}
template <int KIND, int N> static void inv_tr_stub(const int16_t* src, int16_t* dst, intptr_t dstStride)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t oa = st.in1d(src, N * N * 2);
    x265hip_job jb = {};
    const JobRef dj = put_job(st, jb);
    const size_t od = st.alloc(N * N * 2);
    st.upload();
    st.require(x265hip_transform_batch(KIND, D, N, plane(st, oa, N), plane(st, od, N), dj, 1, N >= 16, st.stream), "inverse transform");
    st.download(od, N * N * 2);
    st.out2d(od, dst, dstStride, N, N, 2);
}

// ---------------------------------------------------------------- quantisation
static uint32_t quant_stub(const int16_t* coef, const int32_t* quantCoeff, int32_t* deltaU, int16_t* qCoef, int qBits, int add, int numCoeff)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o0 = st.in1d(coef, numCoeff * 2), o1 = st.in1d(quantCoeff, numCoeff * 4);
    x265hip_job jb = {}; jb.arg[0] = qBits; jb.arg[1] = add; jb.arg[2] = numCoeff;
    const JobRef dj = put_job(st, jb);
    const size_t o2 = st.alloc(numCoeff * 4), o3 = st.alloc(numCoeff * 2), orr = st.alloc(4);
    st.upload();
    const x265hip_plane pl[4] = { plane(st, o0, 0), plane(st, o1, 0), plane(st, o2, 0), plane(st, o3, 0) };
    st.require(x265hip_quant_batch(X265HIP_Q_QUANT, pl, dj, 1, st.dptr<uint32_t>(orr), st.stream), "quant");
    st.download(o2, orr + 4 - o2);
    memcpy(deltaU, st.hptr<void>(o2), numCoeff * 4);
    memcpy(qCoef, st.hptr<void>(o3), numCoeff * 2);
    return *st.hptr<uint32_t>(orr);
}
static uint32_t nquant_stub(const int16_t* coef, const int32_t* quantCoeff, int16_t* qCoef, int qBits, int add, int numCoeff)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o0 = st.in1d(coef, numCoeff * 2), o1 = st.in1d(quantCoeff, numCoeff * 4);
    x265hip_job jb = {}; jb.arg[0] = qBits; jb.arg[1] = add; jb.arg[2] = numCoeff;
    const JobRef dj = put_job(st, jb);
    const size_t o3 = st.alloc(numCoeff * 2), orr = st.alloc(4);
    st.upload();
    const x265hip_plane pl[4] = { plane(st, o0, 0), plane(st, o1, 0), plane(st, o3, 0), plane(st, o3, 0) };
    st.require(x265hip_quant_batch(X265HIP_Q_NQUANT, pl, dj, 1, st.dptr<uint32_t>(orr), st.stream), "nquant");
    st.download(o3, orr + 4 - o3);
    memcpy(qCoef, st.hptr<void>(o3), numCoeff * 2);
    return *st.hptr<uint32_t>(orr);
}
static void dequant_normal_stub(const int16_t* quantCoef, int16_t* coef, int num, int scale, int shift)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o0 = st.in1d(quantCoef, num * 2);
    x265hip_job jb = {}; jb.arg[0] = num; jb.arg[1] = scale; jb.arg[2] = shift;
    const JobRef dj = put_job(st, jb);
    const size_t o3 = st.alloc(num * 2);
    st.upload();
    const x265hip_plane pl[4] = { plane(st, o0, 0), plane(st, o0, 0), plane(st, o3, 0), plane(st, o3, 0) };
    st.require(x265hip_quant_batch(X265HIP_Q_DEQUANT_NORMAL, pl, dj, 1, nullptr, st.stream), "dequant_normal");
    st.download(o3, num * 2);
    memcpy(coef, st.hptr<void>(o3), num * 2);
}
static void dequant_scaling_stub(const int16_t* quantCoef, const int32_t* deQuantCoef, int16_t* coef, int num, int per, int shift)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o0 = st.in1d(quantCoef, num * 2), o1 = st.in1d(deQuantCoef, num * 4);
    x265hip_job jb = {}; jb.arg[0] = num; jb.arg[1] = per; jb.arg[2] = shift;
    const JobRef dj = put_job(st, jb);
    const size_t o3 = st.alloc(num * 2);
    st.upload();
    const x265hip_plane pl[4] = { plane(st, o0, 0), plane(st, o1, 0), plane(st, o3, 0), plane(st, o3, 0) };
    st.require(x265hip_quant_batch(X265HIP_Q_DEQUANT_SCALING, pl, dj, 1, nullptr, st.stream), "dequant_scaling");
    st.download(o3, num * 2);
    memcpy(coef, st.hptr<void>(o3), num * 2);
}
static void denoise_stub(int16_t* dctCoef, uint32_t* resSum, const uint16_t* offset, int numCoeff)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o0 = st.in1d(dctCoef, numCoeff * 2), o1 = st.in1d(resSum, numCoeff * 4), o2 = st.in1d(offset, numCoeff * 2);
    x265hip_job jb = {}; jb.arg[0] = numCoeff;
    const JobRef dj = put_job(st, jb);
    st.upload();
    const x265hip_plane pl[4] = { plane(st, o0, 0), plane(st, o1, 0), plane(st, o2, 0), plane(st, o2, 0) };
    st.require(x265hip_quant_batch(X265HIP_Q_DENOISE, pl, dj, 1, nullptr, st.stream), "denoiseDct");
    st.download(o0, o1 + (size_t)numCoeff * 4 - o0);
    memcpy(dctCoef, st.hptr<void>(o0), numCoeff * 2);
    memcpy(resSum, st.hptr<void>(o1), numCoeff * 4);
}
template <int N> static int count_nonzero_stub(const int16_t* q)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o0 = st.in1d(q, N * N * 2);
    x265hip_job jb = {}; jb.arg[0] = N * N;
    const JobRef dj = put_job(st, jb);
    const size_t orr = st.alloc(4);
    st.upload();
    const x265hip_plane pl[4] = { plane(st, o0, 0), plane(st, o0, 0), plane(st, o0, 0), plane(st, o0, 0) };
    st.require(x265hip_quant_batch(X265HIP_Q_COUNT_NONZERO, pl, dj, 1, st.dptr<uint32_t>(orr), st.stream), "count_nonzero");
    st.download(orr, 4);
    return (int)*st.hptr<uint32_t>(orr);
}
template <int N> static uint32_t copy_cnt_stub(int16_t* coeff, const int16_t* residual, intptr_t resiStride)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o0 = st.in2d(residual, resiStride, N, N, 2);
    x265hip_job jb = {}; jb.arg[0] = N;
    const JobRef dj = put_job(st, jb);
    const size_t o3 = st.alloc(N * N * 2), orr = st.alloc(4);
    st.upload();
    const x265hip_plane pl[4] = { plane(st, o0, N), plane(st, o0, 0), plane(st, o0, 0), plane(st, o3, 0) };
    st.require(x265hip_quant_batch(X265HIP_Q_COPY_CNT, pl, dj, 1, st.dptr<uint32_t>(orr), st.stream), "copy_cnt");
    st.download(o3, orr + 4 - o3);
    memcpy(coeff, st.hptr<void>(o3), N * N * 2);
    return *st.hptr<uint32_t>(orr);
}

// ---------------------------------------------------------------- intra prediction
// SLOT: 0 = planar slot, 1 = DC slot (both IGNORE the dirMode argument, intrapred.cpp:70,88 - callers do pass
// other values there), 2 = angular slots (mode taken from the argument)
template <int N, int SLOT> static void intra_pred_stub(pixel* dst, intptr_t dstStride, const pixel* srcPix, int dirMode, int bFilter)
{
    if (SLOT < 2) dirMode = SLOT;
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o0 = st.in1d(srcPix, (4 * N + 1) * ES);
    x265hip_job jb = {}; jb.arg[0] = dirMode; jb.arg[1] = bFilter;
    const JobRef dj = put_job(st, jb);
    const size_t od = st.alloc(N * N * ES);
    st.upload();
    st.require(x265hip_intra_batch(X265HIP_INTRA_PRED, D, N, plane(st, o0, 0), plane(st, od, N), dj, 1, st.stream), "intra_pred");
    st.download(od, N * N * ES);
    st.out2d(od, dst, dstStride, N, N, ES);
}
template <int N> static void intra_filter_stub(const pixel* references, pixel* filtered)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o0 = st.in1d(references, (4 * N + 1) * ES);
    x265hip_job jb = {};
    const JobRef dj = put_job(st, jb);
    const size_t od = st.alloc((4 * N + 1) * ES);
    st.upload();
    st.require(x265hip_intra_batch(X265HIP_INTRA_FILTER, D, N, plane(st, o0, 0), plane(st, od, 0), dj, 1, st.stream), "intra_filter");
    st.download(od, (4 * N + 1) * ES);
    memcpy(filtered, st.hptr<void>(od), (4 * N + 1) * ES);
}
template <int N> static void intra_allangs_stub(pixel* dest, pixel* refPix, pixel* filtPix, int bLuma)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o0 = st.in1d(refPix, (4 * N + 1) * ES), o1 = st.in1d(filtPix, (4 * N + 1) * ES);
    x265hip_job jb = {}; jb.off[2] = (int64_t)((o1 - o0) / ES); jb.arg[0] = bLuma;
    const JobRef dj = put_job(st, jb);
    const size_t od = st.alloc(33 * N * N * ES);
    st.upload();
    st.require(x265hip_intra_batch(X265HIP_INTRA_ALLANGS, D, N, plane(st, o0, 0), plane(st, od, N), dj, 1, st.stream), "intra_allangs");
    st.download(od, 33 * N * N * ES);
    memcpy(dest, st.hptr<void>(od), 33 * N * N * ES);
}

// ---------------------------------------------------------------- element-wise block ops
// generic: up to two strided inputs, one output (strided or contiguous)
template <typename T0, typename T1, typename T2>
static void op_core(int op, int w, int h, T0* dst, intptr_t ds, bool dstContig, const T1* s1, intptr_t ss1, bool s1Contig,
                    const T2* s2, intptr_t ss2, const int args[4])
{
    ThreadStage& st = thread_stage();
    st.begin();
    size_t o1 = 0, o2 = 0;
    if (s1) o1 = s1Contig ? st.in1d(s1, (size_t)w * h * sizeof(T1)) : st.in2d(s1, ss1, w, h, sizeof(T1));
    if (s2) o2 = st.in2d(s2, ss2, w, h, sizeof(T2));
    x265hip_job jb = {};
    for (int i = 0; i < 4; i++) jb.arg[i] = args ? args[i] : 0;
    const JobRef dj = put_job(st, jb);
    const size_t od = st.alloc((size_t)w * h * sizeof(T0));
    st.upload();
    const x265hip_plane pl[3] = { plane(st, od, w), plane(st, o1, w), plane(st, o2, w) };
    st.require(x265hip_blockop_batch(op, D, w, h, pl, dj, 1, nullptr, st.stream), "blockop");
    st.download(od, (size_t)w * h * sizeof(T0));
    if (dstContig) memcpy(dst, st.hptr<void>(od), (size_t)w * h * sizeof(T0));
    else st.out2d(od, dst, ds, w, h, sizeof(T0));
}
template <int W, int H> static void copy_pp_stub(pixel* d, intptr_t ds, const pixel* s, intptr_t ss) { op_core<pixel, pixel, pixel>(X265HIP_OP_COPY_PP, W, H, d, ds, false, s, ss, false, nullptr, 0, nullptr); }
template <int W, int H> static void copy_ps_stub(int16_t* d, intptr_t ds, const pixel* s, intptr_t ss) { op_core<int16_t, pixel, pixel>(X265HIP_OP_COPY_PS, W, H, d, ds, false, s, ss, false, nullptr, 0, nullptr); }
template <int W, int H> static void copy_sp_stub(pixel* d, intptr_t ds, const int16_t* s, intptr_t ss) { op_core<pixel, int16_t, pixel>(X265HIP_OP_COPY_SP, W, H, d, ds, false, s, ss, false, nullptr, 0, nullptr); }
template <int W, int H> static void copy_ss_stub(int16_t* d, intptr_t ds, const int16_t* s, intptr_t ss) { op_core<int16_t, int16_t, pixel>(X265HIP_OP_COPY_SS, W, H, d, ds, false, s, ss, false, nullptr, 0, nullptr); }
template <int W, int H> static void sub_ps_stub(int16_t* d, intptr_t ds, const pixel* a, const pixel* b, intptr_t sa, intptr_t sb) { op_core<int16_t, pixel, pixel>(X265HIP_OP_SUB_PS, W, H, d, ds, false, a, sa, false, b, sb, nullptr); }
template <int W, int H> static void add_ps_stub(pixel* d, intptr_t ds, const pixel* a, const int16_t* r, intptr_t sa, intptr_t sr) { op_core<pixel, pixel, int16_t>(X265HIP_OP_ADD_PS, W, H, d, ds, false, a, sa, false, r, sr, nullptr); }
template <int W, int H> static void addavg_stub(const int16_t* a, const int16_t* b, pixel* d, intptr_t sa, intptr_t sb, intptr_t ds) { op_core<pixel, int16_t, int16_t>(X265HIP_OP_ADDAVG, W, H, d, ds, false, a, sa, false, b, sb, nullptr); }
template <int W, int H> static void pixelavg_stub(pixel* d, intptr_t ds, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb, int) { op_core<pixel, pixel, pixel>(X265HIP_OP_PIXELAVG, W, H, d, ds, false, a, sa, false, b, sb, nullptr); }
template <int N> static void calcresidual_stub(const pixel* fenc, const pixel* pred, int16_t* resi, intptr_t stride) { op_core<int16_t, pixel, pixel>(X265HIP_OP_SUB_PS, N, N, resi, stride, false, fenc, stride, false, pred, stride, nullptr); }
template <int N> static void blockfill_stub(int16_t* d, intptr_t ds, int16_t v) { const int a[4] = { v, 0, 0, 0 }; op_core<int16_t, pixel, pixel>(X265HIP_OP_BLOCKFILL, N, N, d, ds, false, nullptr, 0, false, nullptr, 0, a); }
template <int N> static void cpy2d1d_shl_stub(int16_t* d, const int16_t* s, intptr_t ss, int sh) { const int a[4] = { sh, 0, 0, 0 }; op_core<int16_t, int16_t, pixel>(X265HIP_OP_CPY2DTO1D_SHL, N, N, d, 0, true, s, ss, false, nullptr, 0, a); }
template <int N> static void cpy2d1d_shr_stub(int16_t* d, const int16_t* s, intptr_t ss, int sh) { const int a[4] = { sh, 0, 0, 0 }; op_core<int16_t, int16_t, pixel>(X265HIP_OP_CPY2DTO1D_SHR, N, N, d, 0, true, s, ss, false, nullptr, 0, a); }
template <int N> static void cpy1d2d_shl_stub(int16_t* d, const int16_t* s, intptr_t ds, int sh) { const int a[4] = { sh, 0, 0, 0 }; op_core<int16_t, int16_t, pixel>(X265HIP_OP_CPY1DTO2D_SHL, N, N, d, ds, false, s, 0, true, nullptr, 0, a); }
template <int N> static void cpy1d2d_shr_stub(int16_t* d, const int16_t* s, intptr_t ds, int sh) { const int a[4] = { sh, 0, 0, 0 }; op_core<int16_t, int16_t, pixel>(X265HIP_OP_CPY1DTO2D_SHR, N, N, d, ds, false, s, 0, true, nullptr, 0, a); }
template <int N> static void transpose_stub(pixel* d, const pixel* s, intptr_t ss) { op_core<pixel, pixel, pixel>(X265HIP_OP_TRANSPOSE, N, N, d, 0, true, s, ss, false, nullptr, 0, nullptr); }
static void weight_pp_stub(const pixel* src, pixel* dst, intptr_t stride, int width, int height, int w0, int round, int shift, int offset)
{ const int a[4] = { w0, round, shift, offset }; op_core<pixel, pixel, pixel>(X265HIP_OP_WEIGHT_PP, width, height, dst, stride, false, src, stride, false, nullptr, 0, a); }
static void weight_sp_stub(const int16_t* src, pixel* dst, intptr_t srcStride, intptr_t dstStride, int width, int height, int w0, int round, int shift, int offset)
{ const int a[4] = { w0, round, shift, offset }; op_core<pixel, int16_t, pixel>(X265HIP_OP_WEIGHT_SP, width, height, dst, dstStride, false, src, srcStride, false, nullptr, 0, a); }

static void scale1d_stub(pixel* dst, const pixel* src)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o1 = st.in1d(src, 256 * ES);
    x265hip_job jb = {};
    const JobRef dj = put_job(st, jb);
    const size_t od = st.alloc(128 * ES);
    st.upload();
    const x265hip_plane pl[3] = { plane(st, od, 0), plane(st, o1, 0), plane(st, o1, 0) };
    st.require(x265hip_blockop_batch(X265HIP_OP_SCALE1D_128TO64, D, 0, 0, pl, dj, 1, nullptr, st.stream), "scale1D");
    st.download(od, 128 * ES);
    memcpy(dst, st.hptr<void>(od), 128 * ES);
}
static void scale2d_stub(pixel* dst, const pixel* src, intptr_t stride)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o1 = st.in2d(src, stride, 64, 64, ES);
    x265hip_job jb = {};
    const JobRef dj = put_job(st, jb);
    const size_t od = st.alloc(32 * 32 * ES);
    st.upload();
    const x265hip_plane pl[3] = { plane(st, od, 0), plane(st, o1, 64), plane(st, o1, 64) };
    st.require(x265hip_blockop_batch(X265HIP_OP_SCALE2D_64TO32, D, 0, 0, pl, dj, 1, nullptr, st.stream), "scale2D");
    st.download(od, 32 * 32 * ES);
    memcpy(dst, st.hptr<void>(od), 32 * 32 * ES);
}
// reductions: sse_ss, ssd_s, var
template <typename T> static uint64_t reduce_core(int op, int n, const T* a, intptr_t sa, const T* b, intptr_t sb)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o0 = st.in2d(a, sa, n, n, sizeof(T));
    const size_t o1 = b ? st.in2d(b, sb, n, n, sizeof(T)) : o0;
    x265hip_job jb = {};
    const JobRef dj = put_job(st, jb);
    const size_t orr = st.alloc(8);
    st.upload();
    const x265hip_plane pl[3] = { plane(st, o0, n), plane(st, o1, n), plane(st, o1, n) };
    st.require(x265hip_blockop_batch(op, D, n, n, pl, dj, 1, st.dptr<uint64_t>(orr), st.stream), "block reduction");
    st.download(orr, 8);
    return *st.hptr<uint64_t>(orr);
}
template <int N> static sse_t sse_ss_stub(const int16_t* a, intptr_t sa, const int16_t* b, intptr_t sb) { return (sse_t)reduce_core<int16_t>(X265HIP_OP_SSE_SS, N, a, sa, b, sb); }
template <int N> static sse_t ssd_s_stub(const int16_t* a, intptr_t sa) { return (sse_t)reduce_core<int16_t>(X265HIP_OP_SSD_S, N, a, sa, nullptr, 0); }
template <int N> static uint64_t var_stub(const pixel* p, intptr_t s) { return reduce_core<pixel>(X265HIP_OP_VAR, N, p, s, nullptr, 0); }
// ssimDist / normFact (pixel.cpp:958-995): 64-bit sums handed back through pointers
template <int N> static void ssim_dist_stub(const pixel* fenc, uint32_t fstride, const pixel* recon, intptr_t rstride, uint64_t* ssBlock, int shift, uint64_t* ac_k)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o0 = st.in2d(fenc, (intptr_t)fstride, N, N, sizeof(pixel));
    const size_t o1 = st.in2d(recon, rstride, N, N, sizeof(pixel));
    x265hip_job jb = {};
    jb.arg[0] = shift;
    const JobRef dj = put_job(st, jb);
    const size_t orr = st.alloc(16);
    st.upload();
    const x265hip_plane pl[3] = { plane(st, o0, N), plane(st, o1, N), plane(st, o1, N) };
    st.require(x265hip_blockop_batch(X265HIP_OP_SSIM_DIST, D, N, N, pl, dj, 1, st.dptr<uint64_t>(orr), st.stream), "ssimDist");
    st.download(orr, 16);
    *ssBlock = st.hptr<uint64_t>(orr)[0];
    *ac_k = st.hptr<uint64_t>(orr)[1];
}
static void norm_fact_stub(const pixel* src, uint32_t blockSize, int shift, uint64_t* z_k)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const int n = (int)blockSize;
    const size_t o0 = st.in2d(src, n, n, n, sizeof(pixel));
    x265hip_job jb = {};
    jb.arg[0] = shift;
    const JobRef dj = put_job(st, jb);
    const size_t orr = st.alloc(8);
    st.upload();
    const x265hip_plane pl[3] = { plane(st, o0, n), plane(st, o0, n), plane(st, o0, n) };
    st.require(x265hip_blockop_batch(X265HIP_OP_NORM_FACT, D, n, n, pl, dj, 1, st.dptr<uint64_t>(orr), st.stream), "normFact");
    st.download(orr, 8);
    *z_k = *st.hptr<uint64_t>(orr);
}

// ---------------------------------------------------------------- loop filter family
static constexpr size_t NO_RESULT = ~(size_t)0;
static void lf_run(ThreadStage& st, int kind, const size_t o[4], const intptr_t strides[4], const x265hip_job& jb, size_t ores)
{
    const JobRef dj = put_job(st, jb);     // last alloc of the call: device addresses are taken below
    st.upload();
    uint32_t* dres = ores == NO_RESULT ? nullptr : st.dptr<uint32_t>(ores);
    const x265hip_plane pl[4] = { plane(st, o[0], strides[0]), plane(st, o[1], strides[1]), plane(st, o[2], strides[2]), plane(st, o[3], strides[3]) };
    st.require(x265hip_loopfilter_batch(kind, D, pl, dj, 1, dres, st.stream), "loopfilter");
}
static void sign_stub(int8_t* dst, const pixel* src1, const pixel* src2, const int endX)
{
    if (endX <= 0) return;
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o1 = st.in1d(src1, endX * ES), o2 = st.in1d(src2, endX * ES), od = st.in1d(dst, endX);
    const size_t o[4] = { od, o1, o2, o2 }; const intptr_t s[4] = { 0, 0, 0, 0 };
    x265hip_job jb = {}; jb.arg[0] = endX;
    lf_run(st, X265HIP_LF_SIGN, o, s, jb, NO_RESULT);
    st.download(od, endX);
    memcpy(dst, st.hptr<void>(od), endX);
}
// stage `rows` rows of rec with `left`/`right` extra columns; returns offset and sets *org to the element offset of rec[0]
static size_t stage_rec(ThreadStage& st, const pixel* rec, intptr_t stride, int width, int rows, int left, int right, int rowsAbove, int* tw, int* org)
{
    *tw = width + left + right;
    *org = rowsAbove * *tw + left;
    return st.in2d(rec - (intptr_t)rowsAbove * stride - left, stride, *tw, rows + rowsAbove, ES);
}
static void unstage_rec(ThreadStage& st, size_t off, int tw, int org, pixel* rec, intptr_t stride, int width, int rows)
{
    st.download(off, (size_t)(org + (rows - 1) * tw + width) * ES);
    const pixel* s = st.hptr<pixel>(off) + org;
    for (int y = 0; y < rows; y++) memcpy(rec + (intptr_t)y * stride, s + (size_t)y * tw, (size_t)width * ES);
}
static void sao_e0_stub(pixel* rec, int8_t* offsetEo, int width, int8_t* signLeft, intptr_t stride)
{
    ThreadStage& st = thread_stage();
    st.begin();
    int tw, org;
    const size_t orec = stage_rec(st, rec, stride, width, 2, 1, 1, 0, &tw, &org);
    const size_t oo = st.in1d(offsetEo, 5), osl = st.in1d(signLeft, 2);
    const size_t o[4] = { orec, oo, osl, osl }; const intptr_t s[4] = { tw, 0, 0, 0 };
    x265hip_job jb = {}; jb.off[0] = org; jb.arg[0] = width;
    lf_run(st, X265HIP_LF_SAO_E0, o, s, jb, NO_RESULT);
    unstage_rec(st, orec, tw, org, rec, stride, width, 2);
}
template <int ROWS> static void sao_e1_stub(pixel* rec, int8_t* upBuff1, int8_t* offsetEo, intptr_t stride, int width)
{
    ThreadStage& st = thread_stage();
    st.begin();
    int tw, org;
    const size_t orec = stage_rec(st, rec, stride, width, ROWS + 1, 0, 0, 0, &tw, &org);
    const size_t oo = st.in1d(offsetEo, 5), oup = st.in1d(upBuff1, width);
    const size_t o[4] = { orec, oo, oup, oup }; const intptr_t s[4] = { tw, 0, 0, 0 };
    x265hip_job jb = {}; jb.off[0] = org; jb.arg[0] = width;
    lf_run(st, ROWS == 1 ? X265HIP_LF_SAO_E1 : X265HIP_LF_SAO_E1_2ROWS, o, s, jb, NO_RESULT);
    st.download(orec, oup + width - orec);
    const pixel* r = st.hptr<pixel>(orec);
    for (int y = 0; y < ROWS; y++) memcpy(rec + (intptr_t)y * stride, r + (size_t)y * tw, (size_t)width * ES);
    memcpy(upBuff1, st.hptr<void>(oup), width);
}
static void sao_e2_stub(pixel* rec, int8_t* bufft, int8_t* buff1, int8_t* offsetEo, int width, intptr_t stride)
{
    ThreadStage& st = thread_stage();
    st.begin();
    int tw, org;
    const size_t orec = stage_rec(st, rec, stride, width, 2, 0, 1, 0, &tw, &org);
    const size_t oo = st.in1d(offsetEo, 5), ob1 = st.in1d(buff1, width), obt = st.in1d(bufft, width + 1);
    const size_t o[4] = { orec, oo, ob1, obt }; const intptr_t s[4] = { tw, 0, 0, 0 };
    x265hip_job jb = {}; jb.off[0] = org; jb.arg[0] = width;
    lf_run(st, X265HIP_LF_SAO_E2, o, s, jb, NO_RESULT);
    st.download(orec, obt + width + 1 - orec);
    memcpy(rec, st.hptr<pixel>(orec), (size_t)width * ES);
    memcpy(bufft + 1, st.hptr<int8_t>(obt) + 1, width);
}
static void sao_e3_stub(pixel* rec, int8_t* upBuff1, int8_t* offsetEo, intptr_t stride, int startX, int endX)
{
    if (endX - startX < 2) return;
    ThreadStage& st = thread_stage();
    st.begin();
    // rows 0 and 1, columns [startX, endX]; the sign buffer is touched on [startX, endX-1]
    const int width = endX - startX;
    int tw, org;
    const size_t orec = stage_rec(st, rec + startX, stride, width, 2, 0, 0, 0, &tw, &org);
    const size_t oo = st.in1d(offsetEo, 5), oup = st.in1d(upBuff1 + startX, width);
    const size_t o[4] = { orec, oo, oup, oup }; const intptr_t s[4] = { tw, 0, 0, 0 };
    x265hip_job jb = {}; jb.off[0] = org; jb.arg[0] = 0; jb.arg[1] = width;
    lf_run(st, X265HIP_LF_SAO_E3, o, s, jb, NO_RESULT);
    st.download(orec, oup + width - orec);
    memcpy(rec + startX + 1, st.hptr<pixel>(orec) + 1, (size_t)(width - 1) * ES);
    memcpy(upBuff1 + startX, st.hptr<int8_t>(oup), width - 1);
}
static void sao_b0_stub(pixel* rec, const int8_t* offsetBo, int ctuWidth, int ctuHeight, intptr_t stride)
{
    ThreadStage& st = thread_stage();
    st.begin();
    int tw, org;
    const size_t orec = stage_rec(st, rec, stride, ctuWidth, ctuHeight, 0, 0, 0, &tw, &org);
    const size_t oo = st.in1d(offsetBo, 32);
    const size_t o[4] = { orec, oo, oo, oo }; const intptr_t s[4] = { tw, 0, 0, 0 };
    x265hip_job jb = {}; jb.off[0] = org; jb.arg[0] = ctuWidth; jb.arg[1] = ctuHeight;
    lf_run(st, X265HIP_LF_SAO_B0, o, s, jb, NO_RESULT);
    unstage_rec(st, orec, tw, org, rec, stride, ctuWidth, ctuHeight);
}
// SAO statistics: stage the rec neighbourhood each class reads (one column left/right, one row below, one row above
// is never read: row 0 uses the carried buffer), the 64-stride diff block, the carried sign buffers and the stats/count arrays.
static void sao_stats(int kind, const int16_t* diff, const pixel* rec, intptr_t stride, int8_t* upBuff1, int8_t* upBufft,
                      int endX, int endY, int32_t* stats, int32_t* count)
{
    if (endX <= 0 || endY <= 0) return;
    ThreadStage& st = thread_stage();
    st.begin();
    const int bins = kind == X265HIP_LF_STATS_BO ? 32 : 5;
    int tw, org;
    const size_t orec = stage_rec(st, rec, stride, endX, endY + 1, 1, 1, 0, &tw, &org);
    const size_t odiff = st.in2d(diff, 64, endX, endY, 2);
    // carried buffers: E3 writes index -1, E2 writes index endX -> stage [-1, endX]
    const size_t oup = upBuff1 ? st.in1d(upBuff1 - 1, endX + 2) : orec;
    const size_t out_ = upBufft ? st.in1d(upBufft - 1, endX + 2) : oup;
    int32_t packed[64] = { 0 };
    memcpy(packed, stats, bins * 4); memcpy(packed + 32, count, bins * 4);
    const size_t ores = st.in1d(packed, sizeof(packed));
    const size_t o[4] = { odiff, orec, oup, out_ }; const intptr_t s[4] = { endX, tw, 0, 0 };   // diff packed at pitch endX
    x265hip_job jb = {}; jb.off[1] = org; jb.off[2] = 1; jb.off[3] = 1; jb.arg[0] = endX; jb.arg[1] = endY;
    lf_run(st, kind, o, s, jb, ores);
    st.download(oup, ores + sizeof(packed) - oup);
    const int32_t* r = st.hptr<int32_t>(ores);
    memcpy(stats, r, bins * 4); memcpy(count, r + 32, bins * 4);
    if (kind == X265HIP_LF_STATS_E1) memcpy(upBuff1, st.hptr<int8_t>(oup) + 1, endX);
    if (kind == X265HIP_LF_STATS_E3) memcpy(upBuff1 - 1, st.hptr<int8_t>(oup), endX + 1);
    if (kind == X265HIP_LF_STATS_E2)
    {
        memcpy(upBufft, st.hptr<int8_t>(out_) + 1, endX + 1);
        if (endY >= 2) memcpy(upBuff1, st.hptr<int8_t>(oup) + 1, endX + 1);
    }
}
static void stats_bo_stub(const int16_t* diff, const pixel* rec, intptr_t stride, int endX, int endY, int32_t* stats, int32_t* count)
{ sao_stats(X265HIP_LF_STATS_BO, diff, rec, stride, nullptr, nullptr, endX, endY, stats, count); }
static void stats_e0_stub(const int16_t* diff, const pixel* rec, intptr_t stride, int endX, int endY, int32_t* stats, int32_t* count)
{ sao_stats(X265HIP_LF_STATS_E0, diff, rec, stride, nullptr, nullptr, endX, endY, stats, count); }
static void stats_e1_stub(const int16_t* diff, const pixel* rec, intptr_t stride, int8_t* up, int endX, int endY, int32_t* stats, int32_t* count)
{ sao_stats(X265HIP_LF_STATS_E1, diff, rec, stride, up, nullptr, endX, endY, stats, count); }
static void stats_e2_stub(const int16_t* diff, const pixel* rec, intptr_t stride, int8_t* up, int8_t* upt, int endX, int endY, int32_t* stats, int32_t* count)
{ sao_stats(X265HIP_LF_STATS_E2, diff, rec, stride, up, upt, endX, endY, stats, count); }
static void stats_e3_stub(const int16_t* diff, const pixel* rec, intptr_t stride, int8_t* up, int endX, int endY, int32_t* stats, int32_t* count)
{ sao_stats(X265HIP_LF_STATS_E3, diff, rec, stride, up, nullptr, endX, endY, stats, count); }

// deblocking: 4 lines x 8 (luma) / 4 (chroma) samples across the edge
template <bool CHROMA> static void deblock_core(pixel* src, intptr_t srcStep, intptr_t offset, int32_t a2, int32_t a3, int32_t maskQ)
{
    // gather the 4 x 8 neighbourhood into a dense [line][tap] block (tap stride 1, line stride 8)
    ThreadStage& st = thread_stage();
    st.begin();
    pixel blk[4 * 8];
    for (int l = 0; l < 4; l++)
        for (int t = -4; t < 4; t++)
            blk[l * 8 + t + 4] = (CHROMA && (t < -2 || t > 1)) ? (pixel)0 : src[l * srcStep + t * offset];
    const size_t ob = st.in1d(blk, sizeof(blk));
    const size_t o[4] = { ob, ob, ob, ob }; const intptr_t s[4] = { 0, 0, 0, 0 };
    x265hip_job jb = {}; jb.off[0] = 4; jb.arg[0] = 8; jb.arg[1] = 1; jb.arg[2] = a2; jb.arg[3] = a3;
    if (CHROMA) { jb.off[2] = a3; jb.off[3] = maskQ; }
    lf_run(st, CHROMA ? X265HIP_LF_DEBLOCK_CHROMA : X265HIP_LF_DEBLOCK_LUMA_STRONG, o, s, jb, NO_RESULT);
    st.download(ob, sizeof(blk));
    const pixel* r = st.hptr<pixel>(ob);
    const int lo = CHROMA ? -1 : -3, hi = CHROMA ? 0 : 2;
    for (int l = 0; l < 4; l++)
        for (int t = lo; t <= hi; t++)
            src[l * srcStep + t * offset] = r[l * 8 + t + 4];
}
static void deblock_luma_stub(pixel* src, intptr_t srcStep, intptr_t offset, int32_t tcP, int32_t tcQ) { deblock_core<false>(src, srcStep, offset, tcP, tcQ, 0); }
static void deblock_chroma_stub(pixel* src, intptr_t srcStep, intptr_t offset, int32_t tc, int32_t maskP, int32_t maskQ) { deblock_core<true>(src, srcStep, offset, tc, maskP, maskQ); }

template <int N> static void integral_h_stub(uint32_t* sum, pixel* pix, intptr_t stride)
{
    if (stride <= N) return;
    ThreadStage& st = thread_stage();
    st.begin();
    const int cnt = (int)stride - N;
    const size_t oabove = st.in1d(sum - stride, (size_t)cnt * 4);
    const size_t opix = st.in1d(pix, (size_t)stride * ES);
    // sum row sits `cnt` elements after the staged row above: tell the kernel stride = cnt for the sum plane
    const size_t osum = st.alloc((size_t)cnt * 4);
    (void)osum;
    // kernel: sum[x] = v + sum[x - strideArg]; use strideArg = distance between the two staged rows
    const intptr_t dist = (intptr_t)((osum - oabove) / 4);
    const size_t o[4] = { osum, opix, opix, opix }; const intptr_t s[4] = { 0, 0, 0, 0 };
    x265hip_job jb = {}; jb.arg[0] = (int)dist; jb.arg[1] = N; jb.arg[2] = cnt;
    lf_run(st, X265HIP_LF_INTEGRAL_H, o, s, jb, NO_RESULT);
    st.download(osum, (size_t)cnt * 4);
    memcpy(sum, st.hptr<void>(osum), (size_t)cnt * 4);
}
template <int N> static void integral_v_stub(uint32_t* sum, intptr_t stride)
{
    if (stride <= 0) return;
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o0 = st.in1d(sum, (size_t)stride * 4);
    const size_t oN = st.in1d(sum + (intptr_t)N * stride, (size_t)stride * 4);
    const intptr_t dist = (intptr_t)((oN - o0) / 4);           // kernel reads sum[x + N * strideArg]
    const size_t o[4] = { o0, o0, o0, o0 }; const intptr_t s[4] = { 0, 0, 0, 0 };
    x265hip_job jb = {}; jb.arg[0] = (int)dist; jb.arg[1] = 1; jb.arg[2] = (int)stride;
    lf_run(st, X265HIP_LF_INTEGRAL_V, o, s, jb, NO_RESULT);
    st.download(o0, (size_t)stride * 4);
    memcpy(sum, st.hptr<void>(o0), (size_t)stride * 4);
}
template <int LX, int NSUM> static int ads_stub(int* encDC, uint32_t* sums, int delta, uint16_t* costMvX, int16_t* mvs, int width, int thresh)
{
    if (width <= 0) return 0;
    ThreadStage& st = thread_stage();
    st.begin();
    const int span = width + (NSUM == 4 ? delta + (LX >> 1) : (NSUM == 2 ? delta : 0));
    const size_t os = st.in1d(sums, (size_t)span * 4), oc = st.in1d(costMvX, (size_t)width * 2), oe = st.in1d(encDC, 16);
    x265hip_job jb = {}; jb.arg[0] = delta; jb.arg[1] = width; jb.arg[2] = thresh; jb.arg[3] = LX | (NSUM << 16);
    const JobRef dj = put_job(st, jb);
    const size_t om = st.alloc((size_t)width * 2), orr = st.alloc(4);
    st.upload();
    const x265hip_plane pl[4] = { plane(st, os, 0), plane(st, oc, 0), plane(st, om, 0), plane(st, oe, 0) };
    st.require(x265hip_loopfilter_batch(X265HIP_LF_ADS, D, pl, dj, 1, st.dptr<uint32_t>(orr), st.stream), "ads");
    st.download(om, orr + 4 - om);
    const int n = (int)*st.hptr<uint32_t>(orr);
    memcpy(mvs, st.hptr<void>(om), (size_t)n * 2);
    return n;
}

// ---------------------------------------------------------------- frame-level helpers (pixel.cpp:604-701, 864-1016, ipfilter.cpp:59-77)
template <int KIND, typename S> static void planecopy_run(const S* src, intptr_t srcStride, pixel* dst, intptr_t dstStride, int width, int height, int shift, int mask)
{
    if (width <= 0 || height <= 0) return;
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o0 = st.in2d(src, srcStride, width, height, sizeof(S));
    x265hip_job jb = {}; jb.arg[0] = shift; jb.arg[1] = mask;
    const JobRef dj = put_job(st, jb);
    const size_t o1 = st.alloc((size_t)width * height * ES);
    st.upload();
    const x265hip_plane pl[2] = { plane(st, o0, width), plane(st, o1, width) };
    st.require(x265hip_frame_batch(KIND, D, width, height, pl, dj, 1, nullptr, st.stream), "planecopy");
    st.download(o1, (size_t)width * height * ES);
    st.out2d(o1, dst, dstStride, width, height, ES);
}
static void planecopy_cp_stub(const uint8_t* src, intptr_t srcStride, pixel* dst, intptr_t dstStride, int width, int height, int shift)
{ planecopy_run<X265HIP_FR_PLANECOPY_CP>(src, srcStride, dst, dstStride, width, height, shift, 0); }
static void planecopy_sp_stub(const uint16_t* src, intptr_t srcStride, pixel* dst, intptr_t dstStride, int width, int height, int shift, uint16_t mask)
{ planecopy_run<X265HIP_FR_PLANECOPY_SP>(src, srcStride, dst, dstStride, width, height, shift, mask); }
static void planecopy_sp_shl_stub(const uint16_t* src, intptr_t srcStride, pixel* dst, intptr_t dstStride, int width, int height, int shift, uint16_t mask)
{ planecopy_run<X265HIP_FR_PLANECOPY_SP_SHL>(src, srcStride, dst, dstStride, width, height, shift, mask); }
static void planecopy_pp_shr_stub(const pixel* src, intptr_t srcStride, pixel* dst, intptr_t dstStride, int width, int height, int shift)
{ planecopy_run<X265HIP_FR_PLANECOPY_PP_SHR>(src, srcStride, dst, dstStride, width, height, shift, 0); }
#if X265HIP_DEPTH > 8
// the reference has this slot in its high-bit-depth builds only (pixel.cpp:996, 1345-1347)
static pixel plane_clip_max_stub(pixel* src, intptr_t stride, int width, int height, uint64_t* outsum, const pixel minPix, const pixel maxPix)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o0 = st.in2d(src, stride, width, height, ES);
    x265hip_job jb = {}; jb.arg[0] = minPix; jb.arg[1] = maxPix;
    const JobRef dj = put_job(st, jb);
    const size_t oo = st.alloc(16);
    st.upload();
    const x265hip_plane pl[2] = { plane(st, o0, width), plane(st, o0, width) };
    st.require(x265hip_frame_batch(X265HIP_FR_PLANE_CLIP_MAX, D, width, height, pl, dj, 1, st.dptr<void>(oo), st.stream), "planeClipAndMax");
    st.download(o0, oo + 16 - o0);
    st.out2d(o0, src, stride, width, height, ES);
    *outsum = st.hptr<uint64_t>(oo)[1];
    return (pixel)st.hptr<uint64_t>(oo)[0];
}
#endif
static void ssim_core_stub(const pixel* pix1, intptr_t stride1, const pixel* pix2, intptr_t stride2, int* sums)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o0 = st.in2d(pix1, stride1, 8, 4, ES), o1 = st.in2d(pix2, stride2, 8, 4, ES);
    x265hip_job jb = {};
    const JobRef dj = put_job(st, jb);
    const size_t oo = st.alloc(32);
    st.upload();
    const x265hip_plane pl[2] = { plane(st, o0, 8), plane(st, o1, 8) };
    st.require(x265hip_frame_batch(X265HIP_FR_SSIM_CORE, D, 8, 4, pl, dj, 1, st.dptr<void>(oo), st.stream), "ssim_4x4x2_core");
    st.download(oo, 32);
    memcpy(sums, st.hptr<void>(oo), 32);
}
static float ssim_end4_stub(int* sum0, int* sum1, int width)
{
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o0 = st.in1d(sum0, 80), o1 = st.in1d(sum1, 80);
    x265hip_job jb = {}; jb.arg[0] = width;
    const JobRef dj = put_job(st, jb);
    const size_t oo = st.alloc(4);
    st.upload();
    const x265hip_plane pl[2] = { plane(st, o0, 0), plane(st, o1, 0) };
    st.require(x265hip_frame_batch(X265HIP_FR_SSIM_END4, D, 0, 0, pl, dj, 1, st.dptr<void>(oo), st.stream), "ssim_end_4");
    st.download(oo, 4);
    return *st.hptr<float>(oo);
}
static void fix8_pack_stub(uint16_t* dst, double* src, int count)
{
    if (count <= 0) return;
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o0 = st.in1d(src, (size_t)count * 8);
    x265hip_job jb = {};
    const JobRef dj = put_job(st, jb);
    const size_t o1 = st.alloc((size_t)count * 2);
    st.upload();
    const x265hip_plane pl[2] = { plane(st, o0, 0), plane(st, o1, 0) };
    st.require(x265hip_frame_batch(X265HIP_FR_FIX8_PACK, D, count, 1, pl, dj, 1, nullptr, st.stream), "fix8Pack");
    st.download(o1, (size_t)count * 2);
    memcpy(dst, st.hptr<void>(o1), (size_t)count * 2);
}
static void fix8_unpack_stub(double* dst, uint16_t* src, int count)
{
    if (count <= 0) return;
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o0 = st.in1d(src, (size_t)count * 2);
    x265hip_job jb = {};
    const JobRef dj = put_job(st, jb);
    const size_t o1 = st.alloc((size_t)count * 8);
    st.upload();
    const x265hip_plane pl[2] = { plane(st, o0, 0), plane(st, o1, 0) };
    st.require(x265hip_frame_batch(X265HIP_FR_FIX8_UNPACK, D, count, 1, pl, dj, 1, nullptr, st.stream), "fix8Unpack");
    st.download(o1, (size_t)count * 8);
    memcpy(dst, st.hptr<void>(o1), (size_t)count * 8);
}
// frameInitLowres reads rows 0 .. 2 * height and columns 0 .. 2 * width of the source (pixel.cpp:604-629)
static void lowres_stub(const pixel* src0, pixel* dst0, pixel* dsth, pixel* dstv, pixel* dstc, intptr_t srcStride, intptr_t dstStride, int width, int height)
{
    if (width <= 0 || height <= 0) return;
    ThreadStage& st = thread_stage();
    st.begin();
    const int sw = 2 * width + 1, sh = 2 * height + 1;
    const size_t o0 = st.in2d(src0, srcStride, sw, sh, ES);
    const size_t plane_bytes = ((size_t)width * height * ES + 63) & ~(size_t)63;
    const size_t o1 = st.alloc(4 * plane_bytes);
    st.upload();
    void* const d[4] = { st.dptr<uint8_t>(o1), st.dptr<uint8_t>(o1) + plane_bytes, st.dptr<uint8_t>(o1) + 2 * plane_bytes, st.dptr<uint8_t>(o1) + 3 * plane_bytes };
    st.require(x265hip_frame_init_lowres(D, st.dptr<void>(o0), sw, d, width, width, height, st.stream), "frameInitLowres");
    st.download(o1, 4 * plane_bytes);
    pixel* const out[4] = { dst0, dsth, dstv, dstc };
    for (int i = 0; i < 4; i++) st.out2d(o1 + i * plane_bytes, out[i], dstStride, width, height, ES);
}
static void propagate_cost_stub(int* dst, const uint16_t* propagateIn, const int32_t* intraCosts, const uint16_t* interCosts, const int32_t* invQscales,
                                const double* fpsFactor, int len)
{
    if (len <= 0) return;
    ThreadStage& st = thread_stage();
    st.begin();
    const size_t o0 = st.in1d(propagateIn, (size_t)len * 2), o1 = st.in1d(intraCosts, (size_t)len * 4), o2 = st.in1d(interCosts, (size_t)len * 2),
                 o3 = st.in1d(invQscales, (size_t)len * 4);
    const size_t oo = st.alloc((size_t)len * 4);
    st.upload();
    st.require(x265hip_propagate_cost(st.dptr<int32_t>(oo), st.dptr<uint16_t>(o0), st.dptr<int32_t>(o1), st.dptr<uint16_t>(o2), st.dptr<int32_t>(o3),
                                      *fpsFactor, len, st.stream), "propagateCost");
    st.download(oo, (size_t)len * 4);
    memcpy(dst, st.hptr<void>(oo), (size_t)len * 4);
}
// extendCURowBorder (ipfilter.cpp:59-77): the first / last sample of every row replicated into marginX columns either side
static void extend_row_border_stub(pixel* txt, intptr_t stride, int width, int height, int marginX)
{
    if (width <= 0 || height <= 0 || marginX <= 0) return;
    ThreadStage& st = thread_stage();
    st.begin();
    const int fw = width + 2 * marginX;
    const size_t o0 = st.in2d(txt - marginX, stride, fw, height, ES);
    st.upload();
    st.require(x265hip_extend_border_rows(st.dptr<pixel>(o0) + marginX, fw, width, height, marginX, 0, 0, D, st.stream), "extendRowBorder");
    st.download(o0, (size_t)fw * height * ES);
    st.out2d(o0, txt - marginX, stride, fw, height, ES);
}

// ---------------------------------------------------------------- RDOQ helpers (dct.cpp:757-1069): a batch of one call
struct CoeffCall
{
    ThreadStage& st;
    size_t o[5] = { 0, 0, 0, 0, 0 };
    x265hip_coeff_job jb = {};
    CoeffCall() : st(thread_stage()) { st.begin(); }
    // inputs first (in1d / in2d), then run(): packs the job, allocates the result word, uploads, launches
    size_t orr = 0;
    void run(int kind, const char* what)
    {
        const size_t oj = st.in1d(&jb, sizeof(jb));
        orr = st.alloc(4);
        st.upload();
        void* const bufs[5] = { st.dptr<void>(o[0]), st.dptr<void>(o[1]), st.dptr<void>(o[2]), st.dptr<void>(o[3]), st.dptr<void>(o[4]) };
        st.require(x265hip_coeff_batch(kind, D, bufs, st.dptr<const x265hip_coeff_job>(oj), 1, st.dptr<uint32_t>(orr), st.stream), what);
    }
    uint32_t result() const { return *st.hptr<uint32_t>(orr); }
};
static int scan_pos_last_stub(const uint16_t* scan, const int16_t* coeff, uint16_t* coeffSign, uint16_t* coeffFlag, uint8_t* coeffNum, int numSig,
                              const uint16_t* /* scanCG4x4 */, const int trSize)
{
    CoeffCall c;
    const size_t n = (size_t)trSize * trSize;
    c.o[0] = c.st.in1d(scan, n * 2); c.o[1] = c.st.in1d(coeff, n * 2);
    // the three outputs live in the input half of the staging buffer (they are small): in1d of the caller's arrays keeps the layout simple
    c.o[2] = c.st.in1d(coeffSign, 128); c.o[3] = c.st.in1d(coeffFlag, 128); c.o[4] = c.st.in1d(coeffNum, 64);
    c.jb.arg[0] = numSig; c.jb.arg[1] = trSize;
    c.run(X265HIP_CF_SCAN_POS_LAST, "scanPosLast");
    c.st.download(c.o[2], c.orr + 4 - c.o[2]);
    memcpy(coeffSign, c.st.hptr<void>(c.o[2]), 128); memcpy(coeffFlag, c.st.hptr<void>(c.o[3]), 128); memcpy(coeffNum, c.st.hptr<void>(c.o[4]), 64);
    return (int)c.result();
}
static uint32_t find_pos_first_last_stub(const int16_t* dstCoeff, const intptr_t trSize, const uint16_t scanTbl[16])
{
    CoeffCall c;
    c.o[0] = c.st.in1d(scanTbl, 32); c.o[1] = c.st.in2d(dstCoeff, trSize, 4, 4, 2);
    c.jb.arg[0] = 4;
    c.run(X265HIP_CF_FIND_POS_FIRST_LAST, "findPosFirstLast");
    c.st.download(c.orr, 4);
    return c.result();
}
static uint32_t cost_coeff_nxn_stub(const uint16_t* scan, const int16_t* coeff, intptr_t trSize, uint16_t* absCoeff, const uint8_t* tabSigCtx,
                                    uint32_t scanFlagMask, uint8_t* baseCtx, int offset, int scanPosSigOff, int subPosBase)
{
    CoeffCall c;
    int nctx = 1;                                         // context 0 (the DC position) .. the largest significance context of the group
    for (int i = 0; i < 16; i++) nctx = nctx > tabSigCtx[i] + offset + 1 ? nctx : tabSigCtx[i] + offset + 1;
    // the levels go to absCoeff[0 .. number of non-zeros): the function steps its pointer back by the "last position already known" slot and
    // indexes from that slot on, i.e. it writes from the pointer it was given; at most one entry per visited position
    const int nabs = scanPosSigOff + 1;
    c.o[0] = c.st.in1d(scan, 32); c.o[1] = c.st.in2d(coeff, trSize, 4, 4, 2);
    c.o[2] = c.st.in1d(absCoeff, (size_t)nabs * 2);
    c.o[3] = c.st.in1d(tabSigCtx, 16); c.o[4] = c.st.in1d(baseCtx, (size_t)nctx);
    c.jb.arg[0] = 4; c.jb.arg[1] = (int32_t)scanFlagMask; c.jb.arg[2] = offset; c.jb.arg[3] = scanPosSigOff; c.jb.arg[4] = subPosBase;
    c.run(X265HIP_CF_COST_COEFF_NXN, "costCoeffNxN");
    c.st.download(c.o[2], c.orr + 4 - c.o[2]);
    memcpy(absCoeff, c.st.hptr<void>(c.o[2]), (size_t)nabs * 2);
    memcpy(baseCtx, c.st.hptr<void>(c.o[4]), (size_t)nctx);
    return c.result();
}
static uint32_t cost_coeff_remain_stub(uint16_t* absCoeff, int numNonZero, int idx)
{
    CoeffCall c;
    const int n = numNonZero > idx + 1 ? numNonZero : idx + 1;
    c.o[2] = c.st.in1d(absCoeff, (size_t)n * 2);
    c.jb.arg[0] = numNonZero; c.jb.arg[1] = idx;
    c.run(X265HIP_CF_COST_COEFF_REMAIN, "costCoeffRemain");
    c.st.download(c.orr, 4);
    return c.result();
}
static uint32_t cost_c1c2_stub(uint16_t* absCoeff, intptr_t numC1Flag, uint8_t* baseCtxMod, intptr_t ctxOffset)
{
    CoeffCall c;
    const intptr_t lo = ctxOffset < 0 ? ctxOffset : 0, hi = ctxOffset + 1 > 4 ? ctxOffset + 1 : 4;
    c.o[2] = c.st.in1d(absCoeff, (size_t)(numC1Flag > 1 ? numC1Flag : 1) * 2);
    c.o[4] = c.st.in1d(baseCtxMod + lo, (size_t)(hi - lo));
    c.jb.off[4] = -lo;
    c.jb.arg[0] = (int32_t)numC1Flag; c.jb.arg[1] = (int32_t)ctxOffset;
    c.run(X265HIP_CF_COST_C1C2, "costC1C2Flag");
    c.st.download(c.o[4], c.orr + 4 - c.o[4]);
    // only the greater-than-1 contexts 0..3 and the one greater-than-2 context can have moved
    memcpy(baseCtxMod, c.st.hptr<uint8_t>(c.o[4]) - lo, 4);
    baseCtxMod[ctxOffset] = c.st.hptr<uint8_t>(c.o[4])[ctxOffset - lo];
    return c.result();
}
// one coefficient group of the uncoded-cost pre-passes: the 4x4 group is staged densely (row stride 4 through arg[2])
template <int KIND, int LOG2> static void rdoq_run(int16_t* resi, int16_t* fenc, int64_t* costUncoded, int64_t* totalUncoded, int64_t* totalRd, int64_t* psyScale, uint32_t blkPos)
{
    CoeffCall c;
    const intptr_t tr = (intptr_t)1 << LOG2;
    const int64_t tot[2] = { *totalUncoded, *totalRd }, psy = psyScale ? *psyScale : 0;
    c.o[0] = c.st.in2d((fenc ? fenc : resi) + blkPos, tr, 4, 4, 2); c.o[1] = c.st.in2d(resi + blkPos, tr, 4, 4, 2);
    c.o[2] = c.st.in2d(costUncoded + blkPos, tr, 4, 4, 8); c.o[3] = c.st.in1d(tot, 16); c.o[4] = c.st.in1d(&psy, 8);
    c.jb.arg[0] = 0; c.jb.arg[1] = LOG2; c.jb.arg[2] = 4;
    c.run(KIND, "rdoQuant pre-pass");
    c.st.download(c.o[2], c.o[3] + 16 - c.o[2]);
    c.st.out2d(c.o[2], costUncoded + blkPos, tr, 4, 4, 8);
    *totalUncoded = c.st.hptr<int64_t>(c.o[3])[0]; *totalRd = c.st.hptr<int64_t>(c.o[3])[1];
}
template <int LOG2> static void nonpsy_rdoq_stub(int16_t* resi, int64_t* cost, int64_t* tu, int64_t* trd, uint32_t blkPos)
{ rdoq_run<X265HIP_CF_RDOQ_NONPSY, LOG2>(resi, nullptr, cost, tu, trd, nullptr, blkPos); }
template <int LOG2> static void psy_rdoq_stub(int16_t* resi, int16_t* fenc, int64_t* cost, int64_t* tu, int64_t* trd, int64_t* psyScale, uint32_t blkPos)
{ rdoq_run<X265HIP_CF_RDOQ_PSY, LOG2>(resi, fenc, cost, tu, trd, psyScale, blkPos); }
template <int LOG2> static void psy1_rdoq_stub(int16_t* resi, int64_t* cost, int64_t* tu, int64_t* trd, uint32_t blkPos)
{ rdoq_run<X265HIP_CF_RDOQ_PSY_1P, LOG2>(resi, nullptr, cost, tu, trd, nullptr, blkPos); }
template <int LOG2> static void psy2_rdoq_stub(int16_t* resi, int16_t* fenc, int64_t* cost, int64_t* tu, int64_t* trd, int64_t* psyScale, uint32_t blkPos)
{ rdoq_run<X265HIP_CF_RDOQ_PSY_2P, LOG2>(resi, fenc, cost, tu, trd, psyScale, blkPos); }

#define PU_LIST(X) X(4,4) X(8,8) X(16,16) X(32,32) X(64,64) X(8,4) X(4,8) X(16,8) X(8,16) X(32,16) X(16,32) \
    X(64,32) X(32,64) X(16,12) X(12,16) X(16,4) X(4,16) X(32,24) X(24,32) X(32,8) X(8,32) X(64,48) X(48,64) X(64,16) X(16,64)

// ADS variant per PU (pixel.cpp:1105-1129)
template <int W, int H> struct AdsN { static constexpr int v = 4; };
#define ADSN(W, H, N) template <> struct AdsN<W, H> { static constexpr int v = N; };
ADSN(4,4,1) ADSN(8,8,1) ADSN(8,4,2) ADSN(4,8,2) ADSN(16,8,2) ADSN(8,16,2) ADSN(16,12,1) ADSN(12,16,1) ADSN(16,4,1) ADSN(4,16,1)
ADSN(32,16,2) ADSN(16,32,2) ADSN(64,32,2) ADSN(32,64,2)

// Every stub goes into the table behind a guard that remembers what the host had in THAT SLOT: under X265HIP_ON_ERROR_RESTORE_HOST a
// failed stub (and, from then on, every stub) answers with the host's own function.  The guard is keyed by (stub, slot tag), not by
// the stub alone: one stub serves many slots (the [0]/[1] alignment pairs, dct / standard_dct, the 33 angular intra slots, the
// chroma tables' aliases of luma sizes) and an asm-enabled host has DIFFERENT functions there (pelFilterLumaStrong_V / _H,
// saoCuOrgE2 / saoCuOrgE2_32, normFact8 / 16 / 32 / 64 ...) - each slot must be answered by its own (round-2 advisor finding).
// A slot the host left NULL has nothing to restore to: its guard aborts like the default policy.  The saved pointers are
// per-process statics like the reference's table itself (primitives.h:432): filling a second table re-points them at that table's host.
template <auto Stub, int Tag> struct Guard;
template <typename R, typename... A, R (*Stub)(A...), int Tag> struct Guard<Stub, Tag>
{
    static inline R (*host)(A...) = nullptr;
    static R call(A... a)
    {
        if (g_tableFailed.load(std::memory_order_relaxed) && host) return host(a...);
        try { return Stub(a...); }
        catch (const StubFailure&)
        {
            if (!host) { fprintf(stderr, "libx265hip: fatal: the failed slot has no host function to restore\n"); abort(); }
            return host(a...);
        }
    }
};
template <auto Stub, int Tag, typename F> static inline void set_slot(F& slot, int& n)
{
    typedef Guard<Stub, Tag> G;
    if (slot != &G::call) G::host = slot;
    slot = &G::call;
    n++;
}
// consecutive slots served by one stub (intra_pred[2..34]): a guard per slot all the same
template <auto Stub, int Tag, typename F, size_t... M> static inline void set_slots(F* slots, int& n, std::index_sequence<M...>)
{
    (set_slot<Stub, 1000000 + Tag * 64 + (int)M>(slots[M], n), ...);
}

} // namespace

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

int CAT(setup_primitives_d, X265HIP_DEPTH)(x265hip_EncoderPrimitives* p)
{
    int n = 0;
#define SET(slot, fn) set_slot<&fn, __COUNTER__>((slot), n)
#define SET2(slot, fn) do { SET((slot)[0], fn); SET((slot)[1], fn); } while (0)

#define SET_PU(W, H) { auto& u = p->pu[X265HIP_LUMA_##W##x##H]; \
    SET(u.sad, (cmp_stub<X265HIP_CMP_SAD, W, H>)); SET(u.sad_x3, (sad_x3_stub<W, H>)); SET(u.sad_x4, (sad_x4_stub<W, H>)); \
    SET(u.ads, (ads_stub<W, AdsN<W, H>::v>)); SET(u.satd, (cmp_stub<X265HIP_CMP_SATD, W, H>)); \
    SET(u.luma_hpp, (hpp_stub<8, W, H>)); SET(u.luma_hps, (hps_stub<8, W, H>)); SET(u.luma_vpp, (vpp_stub<8, W, H>)); \
    SET(u.luma_vps, (vps_stub<8, W, H>)); SET(u.luma_vsp, (vsp_stub<8, W, H>)); SET(u.luma_vss, (vss_stub<8, W, H>)); \
    SET(u.luma_hvpp, (hvpp_stub<8, W, H>)); SET2(u.pixelavg_pp, (pixelavg_stub<W, H>)); SET2(u.addAvg, (addavg_stub<W, H>)); \
    SET(u.copy_pp, (copy_pp_stub<W, H>)); SET2(u.convert_p2s, (p2s_stub<W, H>)); }
    PU_LIST(SET_PU)

#define SET_CU(I, N) { auto& c = p->cu[I]; \
    SET2(c.calcresidual, (calcresidual_stub<N>)); SET(c.sub_ps, (sub_ps_stub<N, N>)); SET2(c.add_ps, (add_ps_stub<N, N>)); \
    SET2(c.blockfill_s, (blockfill_stub<N>)); SET(c.cpy2Dto1D_shl, (cpy2d1d_shl_stub<N>)); SET(c.cpy2Dto1D_shr, (cpy2d1d_shr_stub<N>)); \
    SET2(c.cpy1Dto2D_shl, (cpy1d2d_shl_stub<N>)); SET(c.cpy1Dto2D_shr, (cpy1d2d_shr_stub<N>)); \
    SET(c.copy_sp, (copy_sp_stub<N, N>)); SET(c.copy_ps, (copy_ps_stub<N, N>)); SET(c.copy_ss, (copy_ss_stub<N, N>)); SET(c.copy_pp, (copy_pp_stub<N, N>)); \
    SET(c.var, (var_stub<N>)); SET(c.sse_pp, (sse_stub<N, N>)); SET(c.sse_ss, (sse_ss_stub<N>)); SET(c.psy_cost_pp, (cmp_stub<X265HIP_CMP_PSY_COST, N, N>)); \
    SET2(c.ssd_s, (ssd_s_stub<N>)); SET(c.sa8d, (cmp_stub<X265HIP_CMP_SA8D, N, N>)); SET(c.transpose, (transpose_stub<N>)); \
    SET(c.ssimDist, (ssim_dist_stub<N>)); }
    SET_CU(0, 4) SET_CU(1, 8) SET_CU(2, 16) SET_CU(3, 32) SET_CU(4, 64)
    SET(p->cu[1].normFact, norm_fact_stub); SET(p->cu[2].normFact, norm_fact_stub); SET(p->cu[3].normFact, norm_fact_stub);
    SET(p->cu[4].normFact, norm_fact_stub);          // the reference leaves cu[BLOCK_4x4].normFact NULL (pixel.cpp:1354-1357)

#define SET_TU(I, N) { auto& c = p->cu[I]; \
    SET(c.dct, (fwd_tr_stub<X265HIP_TR_DCT, N>)); SET(c.standard_dct, (fwd_tr_stub<X265HIP_TR_DCT, N>)); SET(c.idct, (inv_tr_stub<X265HIP_TR_IDCT, N>)); \
    SET(c.copy_cnt, (copy_cnt_stub<N>)); SET(c.count_nonzero, (count_nonzero_stub<N>)); \
    SET(c.intra_filter, (intra_filter_stub<N>)); SET(c.intra_pred_allangs, (intra_allangs_stub<N>)); \
    SET(c.intra_pred[0], (intra_pred_stub<N, 0>)); SET(c.intra_pred[1], (intra_pred_stub<N, 1>)); \
    set_slots<&intra_pred_stub<N, 2>, __COUNTER__>(&c.intra_pred[2], n, std::make_index_sequence<33>()); }
    SET_TU(0, 4) SET_TU(1, 8) SET_TU(2, 16) SET_TU(3, 32)
    SET(p->cu[1].lowpass_dct, (fwd_tr_stub<X265HIP_TR_LOWPASS_DCT, 8>));
    SET(p->cu[2].lowpass_dct, (fwd_tr_stub<X265HIP_TR_LOWPASS_DCT, 16>));
    SET(p->cu[3].lowpass_dct, (fwd_tr_stub<X265HIP_TR_LOWPASS_DCT, 32>));

    SET(p->dst4x4, (fwd_tr_stub<X265HIP_TR_DST4, 4>)); SET(p->idst4x4, (inv_tr_stub<X265HIP_TR_IDST4, 4>));
    SET(p->quant, quant_stub); SET(p->nquant, nquant_stub); SET(p->dequant_scaling, dequant_scaling_stub);
    SET(p->dequant_normal, dequant_normal_stub); SET(p->denoiseDct, denoise_stub);
    SET2(p->scale1D_128to64, scale1d_stub); SET(p->scale2D_64to32, scale2d_stub);
    SET(p->sign, sign_stub); SET(p->saoCuOrgE0, sao_e0_stub); SET(p->saoCuOrgE1, (sao_e1_stub<1>)); SET(p->saoCuOrgE1_2Rows, (sao_e1_stub<2>));
    SET2(p->saoCuOrgE2, sao_e2_stub); SET2(p->saoCuOrgE3, sao_e3_stub); SET(p->saoCuOrgB0, sao_b0_stub);
    SET(p->saoCuStatsBO, stats_bo_stub); SET(p->saoCuStatsE0, stats_e0_stub); SET(p->saoCuStatsE1, stats_e1_stub);
    SET(p->saoCuStatsE2, stats_e2_stub); SET(p->saoCuStatsE3, stats_e3_stub);
    SET(p->weight_sp, weight_sp_stub); SET(p->weight_pp, weight_pp_stub);
    SET2(p->pelFilterLumaStrong, deblock_luma_stub); SET2(p->pelFilterChroma, deblock_chroma_stub);
#define SET_INTEG(I, N) SET(p->integral_initv[I], (integral_v_stub<N>)); SET(p->integral_inith[I], (integral_h_stub<N>));
    SET_INTEG(0, 4) SET_INTEG(1, 8) SET_INTEG(2, 12) SET_INTEG(3, 16) SET_INTEG(4, 24) SET_INTEG(5, 32)

    // ---- rows a16 / a9: the frame-level helpers and the RDOQ helpers (csrc/frame_coeff_kernels.hip) ----
    SET(p->planecopy_cp, planecopy_cp_stub); SET(p->planecopy_sp, planecopy_sp_stub); SET(p->planecopy_sp_shl, planecopy_sp_shl_stub);
    SET(p->planecopy_pp_shr, planecopy_pp_shr_stub);
#if X265HIP_DEPTH > 8
    SET(p->planeClipAndMax, plane_clip_max_stub);
#endif
    SET(p->ssim_4x4x2_core, ssim_core_stub); SET(p->ssim_end_4, ssim_end4_stub);
    SET(p->fix8Pack, fix8_pack_stub); SET(p->fix8Unpack, fix8_unpack_stub);
    SET(p->frameInitLowres, lowres_stub); SET(p->frameInitLowerRes, lowres_stub);
    SET(p->propagateCost, propagate_cost_stub); SET(p->extendRowBorder, extend_row_border_stub);
    SET(p->scanPosLast, scan_pos_last_stub); SET(p->findPosFirstLast, find_pos_first_last_stub); SET(p->costCoeffRemain, cost_coeff_remain_stub);
    // the two estimators that price context-coded bins need the HOST's per-state table: without x265hip_set_entropy_bits they stay the host's
    if (entropy_bits_ready()) { SET(p->costCoeffNxN, cost_coeff_nxn_stub); SET(p->costC1C2Flag, cost_c1c2_stub); }
#define SET_RDOQ(I, L2) { auto& c = p->cu[I]; SET(c.nonPsyRdoQuant, (nonpsy_rdoq_stub<L2>)); SET(c.psyRdoQuant, (psy_rdoq_stub<L2>)); \
    SET(c.psyRdoQuant_1p, (psy1_rdoq_stub<L2>)); SET(c.psyRdoQuant_2p, (psy2_rdoq_stub<L2>)); }
    SET_RDOQ(0, 2) SET_RDOQ(1, 3) SET_RDOQ(2, 4) SET_RDOQ(3, 5)

    // ---- chroma tables (indexed by the LUMA enum; primitives.h:77-79,393-428) ----
    // satd only where the reference has a function (multiple of 4x4: pixel.cpp:1200-1226,1279-1305); 4:2:0 2x2 has
    // no interpolation / p2s (ipfilter.cpp:416-521)
#define OKSZ(CW, CH) (((CW) % 4 == 0) && ((CH) % 4 == 0))
#define SAFE(V, CW, CH) (OKSZ(CW, CH) ? (V) : 4)
#define SET_CPU(CSP, W, H, CW, CH) { auto& u = p->chroma[CSP].pu[X265HIP_LUMA_##W##x##H]; \
    if (OKSZ(CW, CH)) SET(u.satd, (cmp_stub<X265HIP_CMP_SATD, SAFE(CW, CW, CH), SAFE(CH, CW, CH)>)); \
    if (!((CW) == 2 && (CH) == 2)) { \
        SET(u.filter_vpp, (vpp_stub<4, CW, CH>)); SET(u.filter_vps, (vps_stub<4, CW, CH>)); SET(u.filter_vsp, (vsp_stub<4, CW, CH>)); \
        SET(u.filter_vss, (vss_stub<4, CW, CH>)); SET(u.filter_hpp, (hpp_stub<4, CW, CH>)); SET(u.filter_hps, (hps_stub<4, CW, CH>)); \
        SET2(u.p2s, (p2s_stub<CW, CH>)); } \
    SET2(u.addAvg, (addavg_stub<CW, CH>)); SET(u.copy_pp, (copy_pp_stub<CW, CH>)); }
#define SET_C420(W, H) SET_CPU(1, W, H, W / 2, H / 2)
#define SET_C422(W, H) SET_CPU(2, W, H, W / 2, H)
#define SET_C444(W, H) SET_CPU(3, W, H, W, H)
    PU_LIST(SET_C420) PU_LIST(SET_C422) PU_LIST(SET_C444)

#define SET_CCU(CSP, I, CW, CH) { auto& c = p->chroma[CSP].cu[I]; \
    SET(c.sub_ps, (sub_ps_stub<CW, CH>)); SET2(c.add_ps, (add_ps_stub<CW, CH>)); SET(c.copy_ps, (copy_ps_stub<CW, CH>)); \
    SET(c.copy_sp, (copy_sp_stub<CW, CH>)); SET(c.copy_ss, (copy_ss_stub<CW, CH>)); SET(c.copy_pp, (copy_pp_stub<CW, CH>)); }
    SET_CCU(1, 0, 2, 2) SET_CCU(1, 1, 4, 4) SET_CCU(1, 2, 8, 8) SET_CCU(1, 3, 16, 16) SET_CCU(1, 4, 32, 32)
    SET_CCU(2, 0, 2, 4) SET_CCU(2, 1, 4, 8) SET_CCU(2, 2, 8, 16) SET_CCU(2, 3, 16, 32) SET_CCU(2, 4, 32, 64)
    SET_CCU(3, 0, 4, 4) SET_CCU(3, 1, 8, 8) SET_CCU(3, 2, 16, 16) SET_CCU(3, 3, 32, 32) SET_CCU(3, 4, 64, 64)
    // chroma CU costs (pixel.cpp:1243-1246,1322-1325; primitives.cpp:184-208)
    SET(p->chroma[1].cu[1].sa8d, (cmp_stub<X265HIP_CMP_SATD, 4, 4>)); SET(p->chroma[1].cu[2].sa8d, (cmp_stub<X265HIP_CMP_SA8D, 8, 8>));
    SET(p->chroma[1].cu[3].sa8d, (cmp_stub<X265HIP_CMP_SA8D, 16, 16>)); SET(p->chroma[1].cu[4].sa8d, (cmp_stub<X265HIP_CMP_SA8D, 32, 32>));
    SET(p->chroma[2].cu[1].sa8d, (cmp_stub<X265HIP_CMP_SATD, 4, 8>)); SET(p->chroma[2].cu[2].sa8d, (cmp_stub<X265HIP_CMP_SA8D, 8, 16>));
    SET(p->chroma[2].cu[3].sa8d, (cmp_stub<X265HIP_CMP_SA8D, 16, 32>)); SET(p->chroma[2].cu[4].sa8d, (cmp_stub<X265HIP_CMP_SA8D, 32, 64>));
    SET(p->chroma[3].cu[0].sa8d, (cmp_stub<X265HIP_CMP_SATD, 4, 4>)); SET(p->chroma[3].cu[1].sa8d, (cmp_stub<X265HIP_CMP_SA8D, 8, 8>));
    SET(p->chroma[3].cu[2].sa8d, (cmp_stub<X265HIP_CMP_SA8D, 16, 16>)); SET(p->chroma[3].cu[3].sa8d, (cmp_stub<X265HIP_CMP_SA8D, 32, 32>));
    SET(p->chroma[3].cu[4].sa8d, (cmp_stub<X265HIP_CMP_SA8D, 64, 64>));
    SET(p->chroma[1].cu[1].sse_pp, (sse_stub<4, 4>)); SET(p->chroma[1].cu[2].sse_pp, (sse_stub<8, 8>));
    SET(p->chroma[1].cu[3].sse_pp, (sse_stub<16, 16>)); SET(p->chroma[1].cu[4].sse_pp, (sse_stub<32, 32>));
    SET(p->chroma[2].cu[1].sse_pp, (sse_stub<4, 8>)); SET(p->chroma[2].cu[2].sse_pp, (sse_stub<8, 16>));
    SET(p->chroma[2].cu[3].sse_pp, (sse_stub<16, 32>)); SET(p->chroma[2].cu[4].sse_pp, (sse_stub<32, 64>));
    SET(p->chroma[3].cu[0].sse_pp, (sse_stub<4, 4>)); SET(p->chroma[3].cu[1].sse_pp, (sse_stub<8, 8>));
    SET(p->chroma[3].cu[2].sse_pp, (sse_stub<16, 16>)); SET(p->chroma[3].cu[3].sse_pp, (sse_stub<32, 32>));
    SET(p->chroma[3].cu[4].sse_pp, (sse_stub<64, 64>));
    return n;
}

} // namespace x265hip
