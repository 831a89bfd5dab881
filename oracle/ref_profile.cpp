/* oracle/ref_profile.cpp - TEST INFRASTRUCTURE (measurement aid), never part of the product path.
 *
 * A table filler (x265hip_setup_primitives signature) that wraps the host's own primitives of the families listed below with
 * cycle-counting thunks, so that one encode with the REAL reference tells where its worker threads spend their time, per family
 * of EncoderPrimitives slots (primitives.h:237-429).  Used to decide which loop to take off the CPU next; the
 * thunks call the original function with the original arguments, the bitstream does not change. */
#include "common.h"
#include "primitives.h"

#include <mutex>
#include <vector>
#include <x86intrin.h>

using namespace X265_NS;

namespace {

enum Fam { F_SAD, F_SADX, F_SATD, F_SA8D, F_LUMA_PP, F_LUMA_PS, F_CHROMA_PP, F_CHROMA_PS, F_DCT, F_IDCT, F_QUANT, F_SSE, F_INTRA, F_COPY, F_PSY, F_OTHER, NFAM };
const char* const kFamName[NFAM] = { "sad", "sad_x3/x4", "satd (luma+chroma)", "sa8d", "luma interp pp (hpp/vpp/hvpp)", "luma interp ps/sp/ss",
                                     "chroma interp pp", "chroma interp ps/sp/ss", "dct", "idct", "quant/nquant/dequant", "sse/ssd", "intra_pred (+allangs, filter)",
                                     "copies / residual / add / addAvg / p2s", "psy_cost", "other wrapped" };

struct Acc { uint64_t cyc[NFAM], cnt[NFAM]; };
std::mutex g_lock;
std::vector<Acc*> g_all;
Acc* mine()
{
    static thread_local Acc* a = nullptr;
    if (!a)
    {
        a = new Acc();
        std::lock_guard<std::mutex> l(g_lock);
        g_all.push_back(a);
    }
    return a;
}

template <int FAM, int ID, typename Sig> struct Thunk;
template <int FAM, int ID, typename R, typename... A> struct Thunk<FAM, ID, R (*)(A...)>
{
    static R (*orig)(A...);
    static R call(A... a)
    {
        const uint64_t t = __rdtsc();
        R r = orig(a...);
        Acc* m = mine();
        m->cyc[FAM] += __rdtsc() - t; m->cnt[FAM]++;
        return r;
    }
};
template <int FAM, int ID, typename R, typename... A> R (*Thunk<FAM, ID, R (*)(A...)>::orig)(A...) = nullptr;
template <int FAM, int ID, typename... A> struct Thunk<FAM, ID, void (*)(A...)>
{
    static void (*orig)(A...);
    static void call(A... a)
    {
        const uint64_t t = __rdtsc();
        orig(a...);
        Acc* m = mine();
        m->cyc[FAM] += __rdtsc() - t; m->cnt[FAM]++;
    }
};
template <int FAM, int ID, typename... A> void (*Thunk<FAM, ID, void (*)(A...)>::orig)(A...) = nullptr;

int g_wrapped;
template <int FAM, int ID, typename Sig> void wrap(Sig& slot)
{
    if (!slot) return;
    typedef Thunk<FAM, ID, Sig> T;
    if (slot == &T::call) return;
    T::orig = slot;
    slot = &T::call;
    g_wrapped++;
}
#define W(fam, slot) wrap<fam, __COUNTER__ * 64 + I>(slot)

template <int I> struct PerPU
{
    static void run(EncoderPrimitives& p)
    {
        W(F_SAD, p.pu[I].sad); W(F_SADX, p.pu[I].sad_x3); W(F_SADX, p.pu[I].sad_x4); W(F_SATD, p.pu[I].satd);
        W(F_LUMA_PP, p.pu[I].luma_hpp); W(F_LUMA_PP, p.pu[I].luma_vpp); W(F_LUMA_PP, p.pu[I].luma_hvpp);
        W(F_LUMA_PS, p.pu[I].luma_hps); W(F_LUMA_PS, p.pu[I].luma_vps); W(F_LUMA_PS, p.pu[I].luma_vsp); W(F_LUMA_PS, p.pu[I].luma_vss);
        W(F_COPY, p.pu[I].copy_pp); W(F_COPY, p.pu[I].addAvg[0]); W(F_COPY, p.pu[I].addAvg[1]); W(F_COPY, p.pu[I].convert_p2s[0]); W(F_COPY, p.pu[I].convert_p2s[1]);
        W(F_COPY, p.pu[I].pixelavg_pp[0]); W(F_COPY, p.pu[I].pixelavg_pp[1]);
        auto& c = p.chroma[X265_CSP_I420].pu[I];
        W(F_SATD, c.satd);
        W(F_CHROMA_PP, c.filter_hpp); W(F_CHROMA_PP, c.filter_vpp);
        W(F_CHROMA_PS, c.filter_hps); W(F_CHROMA_PS, c.filter_vps); W(F_CHROMA_PS, c.filter_vsp); W(F_CHROMA_PS, c.filter_vss);
        W(F_COPY, c.copy_pp); W(F_COPY, c.addAvg[0]); W(F_COPY, c.addAvg[1]); W(F_COPY, c.p2s[0]); W(F_COPY, c.p2s[1]);
        PerPU<I + 1>::run(p);
    }
};
template <> struct PerPU<NUM_PU_SIZES> { static void run(EncoderPrimitives&) {} };

template <int I> struct PerCU
{
    static void run(EncoderPrimitives& p)
    {
        auto& c = p.cu[I];
        W(F_SA8D, c.sa8d); W(F_DCT, c.dct); W(F_DCT, c.standard_dct); W(F_DCT, c.lowpass_dct); W(F_IDCT, c.idct);
        W(F_SSE, c.sse_pp); W(F_SSE, c.sse_ss); W(F_SSE, c.ssd_s[0]); W(F_SSE, c.ssd_s[1]); W(F_SSE, c.var);
        W(F_PSY, c.psy_cost_pp);
        W(F_COPY, c.calcresidual[0]); W(F_COPY, c.calcresidual[1]); W(F_COPY, c.sub_ps); W(F_COPY, c.add_ps[0]); W(F_COPY, c.add_ps[1]);
        W(F_COPY, c.copy_ss); W(F_COPY, c.copy_sp); W(F_COPY, c.copy_ps); W(F_COPY, c.blockfill_s[0]); W(F_COPY, c.blockfill_s[1]);
        W(F_COPY, c.cpy2Dto1D_shl); W(F_COPY, c.cpy2Dto1D_shr); W(F_COPY, c.cpy1Dto2D_shl[0]); W(F_COPY, c.cpy1Dto2D_shl[1]); W(F_COPY, c.cpy1Dto2D_shr);
        W(F_QUANT, c.copy_cnt); W(F_QUANT, c.count_nonzero);
        W(F_INTRA, c.intra_pred_allangs); W(F_INTRA, c.intra_filter); W(F_COPY, c.transpose);
        W(F_OTHER, c.nonPsyRdoQuant); W(F_OTHER, c.psyRdoQuant);
        auto& k = p.chroma[X265_CSP_I420].cu[I];
        W(F_SA8D, k.sa8d); W(F_SSE, k.sse_pp); W(F_COPY, k.sub_ps); W(F_COPY, k.add_ps[0]); W(F_COPY, k.add_ps[1]);
        W(F_COPY, k.copy_ss); W(F_COPY, k.copy_sp); W(F_COPY, k.copy_ps);
        PerCU<I + 1>::run(p);
    }
};
template <> struct PerCU<NUM_CU_SIZES> { static void run(EncoderPrimitives&) {} };

template <int CU, int M> struct PerMode
{
    static void run(EncoderPrimitives& p)
    {
        constexpr int I = CU * 64 + M;       /* the W macro's id uses I */
        wrap<F_INTRA, 1000000 + I>(p.cu[CU].intra_pred[M]);
        PerMode<CU, M + 1>::run(p);
    }
};
template <int CU> struct PerMode<CU, NUM_INTRA_MODE> { static void run(EncoderPrimitives&) {} };

} // namespace

extern "C" {

int x265ref_profile_fill_table(void* table, size_t bytes, int depth)
{
    if (!table || bytes != sizeof(EncoderPrimitives) || depth != X265_DEPTH) return -1;
    EncoderPrimitives& p = *static_cast<EncoderPrimitives*>(table);
    g_wrapped = 0;
    PerPU<0>::run(p);
    PerCU<0>::run(p);
    PerMode<0, 0>::run(p); PerMode<1, 0>::run(p); PerMode<2, 0>::run(p); PerMode<3, 0>::run(p);
    constexpr int I = 0;
    W(F_QUANT, p.quant); W(F_QUANT, p.nquant); W(F_QUANT, p.dequant_normal); W(F_QUANT, p.dequant_scaling); W(F_QUANT, p.denoiseDct);
    W(F_DCT, p.dst4x4); W(F_IDCT, p.idst4x4);
    return g_wrapped;
}

/* cycles[NFAM], calls[NFAM], names[NFAM]; returns the number of families */
int x265ref_profile_report(uint64_t* cycles, uint64_t* calls, const char** names)
{
    std::lock_guard<std::mutex> l(g_lock);
    for (int f = 0; f < NFAM; f++)
    {
        cycles[f] = calls[f] = 0;
        names[f] = kFamName[f];
        for (Acc* a : g_all) { cycles[f] += a->cyc[f]; calls[f] += a->cnt[f]; }
    }
    return NFAM;
}

void x265ref_profile_reset(void)
{
    std::lock_guard<std::mutex> l(g_lock);
    for (Acc* a : g_all) *a = Acc();
}

uint64_t x265ref_profile_tsc(void) { return __rdtsc(); }

} // extern "C"
