// table_stubs.hip - TABLE LAYER: fills an EncoderPrimitives-layout table with host-pointer stubs
// that have exactly the reference's slot signatures (source/common/primitives.h:133-234) and run the
// batch-layer HIP kernels on a batch of one.  Compiled once per bit depth (-DX265HIP_DEPTH=8|10|12):
// the assignments below are type-checked against the generated mirror include/x265hip_table.h.
//
// Install order for a real encoder (reference primitives.cpp:248-282): let the host fill its own
// C table first (setupCPrimitives + setupAliasPrimitives), then call x265hip_setup_primitives() on
// it BEFORE x265_encoder_open(); slots we do not implement keep the host's entries.
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
{
    return (int)cmp_one(KIND, W, H, a, sa, b, sb);
}
template <int W, int H> static sse_t sse_stub(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    return (sse_t)cmp_one(X265HIP_CMP_SSE_PP, W, H, a, sa, b, sb);
}

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
    // in2d pads every block start to 64 bytes: the job step is the padded block pitch
    const size_t pitch = (((size_t)W * H * ES) + 63) & ~(size_t)63;
    const size_t oo = st.alloc(8 * N);
    st.upload();
    st.require(x265hip_pixelcmp_batch(X265HIP_CMP_SAD, D, W, H, st.dptr<void>(oa), W, nullptr, 0,
                                      st.dptr<void>(ob), W, nullptr, (int64_t)(pitch / ES),
                                      N, st.dptr<uint64_t>(oo), st.stream), "sad_xN");
    st.download(oo, 8 * N);
    for (int i = 0; i < N; i++) res[i] = (int32_t)st.hptr<uint64_t>(oo)[i];
}
template <int W, int H> static void sad_x3_stub(const pixel* f, const pixel* r0, const pixel* r1, const pixel* r2, intptr_t rs, int32_t* res)
{
    const pixel* r[3] = { r0, r1, r2 };
    sad_xn<W, H, 3>(f, r, rs, res);
}
template <int W, int H> static void sad_x4_stub(const pixel* f, const pixel* r0, const pixel* r1, const pixel* r2, const pixel* r3, intptr_t rs, int32_t* res)
{
    const pixel* r[4] = { r0, r1, r2, r3 };
    sad_xn<W, H, 4>(f, r, rs, res);
}

#define PU_LIST(X) X(4,4) X(8,8) X(16,16) X(32,32) X(64,64) X(8,4) X(4,8) X(16,8) X(8,16) X(32,16) X(16,32) \
    X(64,32) X(32,64) X(16,12) X(12,16) X(16,4) X(4,16) X(32,24) X(24,32) X(32,8) X(8,32) X(64,48) X(48,64) X(64,16) X(16,64)

} // namespace

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

int CAT(setup_primitives_d, X265HIP_DEPTH)(x265hip_EncoderPrimitives* p)
{
    int n = 0;
#define SET(slot, fn) do { (slot) = (fn); n++; } while (0)

#define SET_PU(W, H) { auto& u = p->pu[X265HIP_LUMA_##W##x##H]; \
    SET(u.sad, (cmp_stub<X265HIP_CMP_SAD, W, H>)); SET(u.sad_x3, (sad_x3_stub<W, H>)); SET(u.sad_x4, (sad_x4_stub<W, H>)); \
    SET(u.satd, (cmp_stub<X265HIP_CMP_SATD, W, H>)); }
    PU_LIST(SET_PU)

#define SET_CU(I, N) { auto& c = p->cu[I]; \
    SET(c.sa8d, (cmp_stub<X265HIP_CMP_SA8D, N, N>)); SET(c.sse_pp, (sse_stub<N, N>)); SET(c.psy_cost_pp, (cmp_stub<X265HIP_CMP_PSY_COST, N, N>)); }
    SET_CU(0, 4) SET_CU(1, 8) SET_CU(2, 16) SET_CU(3, 32) SET_CU(4, 64)

    // chroma satd: same kernels on the chroma block size, only where the reference has a function
    // (NULL when the chroma PU is not a multiple of 4x4: primitives.h:398, pixel.cpp:1200-1226,1279-1305)
#define SET_CSATD(CSP, W, H, CW, CH) if (((CW) % 4 == 0) && ((CH) % 4 == 0)) SET(p->chroma[CSP].pu[X265HIP_LUMA_##W##x##H].satd, (cmp_stub<X265HIP_CMP_SATD, ((CW) % 4 || (CH) % 4) ? 4 : (CW), ((CW) % 4 || (CH) % 4) ? 4 : (CH)>));
#define SET_C420(W, H) SET_CSATD(1, W, H, W / 2, H / 2)
#define SET_C422(W, H) SET_CSATD(2, W, H, W / 2, H)
#define SET_C444(W, H) SET_CSATD(3, W, H, W, H)
    PU_LIST(SET_C420) PU_LIST(SET_C422) PU_LIST(SET_C444)

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
