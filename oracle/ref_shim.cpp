/* oracle/ref_shim.cpp - TEST INFRASTRUCTURE, never part of the product path.
 *
 * Thin C-ABI window onto the *real* reference primitives.  It is compiled by
 * oracle/Makefile together with the reference's own sources (taken where they lie
 * under /root/reference/source, never copied) into oracle/_ref/libx265ref<depth>.so.
 *
 *   x265ref_table()  -> a filled EncoderPrimitives exactly like the reference TestBench's
 *                       `cprim` (source/test/testbench.cpp:155-158: setupCPrimitives then
 *                       setupAliasPrimitives).
 *   x265ref_encoder_table() -> the live global `x265::primitives` the encoder uses
 *                       (primitives.h:432), so a harness can pre-fill it before
 *                       x265_encoder_open (primitives.cpp:250 keeps a pre-filled table).
 *
 * It also proves, at compile time, that include/x265hip_table.h (our generated mirror)
 * has the same offset for every one of the 2280 slots as the reference header.
 */
#include "common.h"
#include "primitives.h"

#include <cstddef>
#include <cstring>

using namespace X265_NS;

/* slot-by-slot layout proof: X265HIP_CHECK_SLOT(path, index) */
#define X265HIP_CHECK_SLOT(path, idx) \
    static_assert(offsetof(EncoderPrimitives, path) == (size_t)(idx) * sizeof(void*), "mirror layout mismatch: " #path);
#include "_gen_offsets.inc"
#undef X265HIP_CHECK_SLOT
static_assert(sizeof(EncoderPrimitives) == 2280 * sizeof(void*), "EncoderPrimitives size");

static EncoderPrimitives g_cprim;
static bool g_cprimReady = false;

extern "C" {

void* x265ref_table(void)
{
    if (!g_cprimReady)
    {
        memset(&g_cprim, 0, sizeof(g_cprim));
        setupCPrimitives(g_cprim);
        setupAliasPrimitives(g_cprim);
        /* lowpassdct.cpp:118-122 keeps POINTERS to the standard_dct slots of the last table that
         * was set up; fill them (as enableLowpassDCTPrimitives does, primitives.cpp:75-81) so the
         * lowpass slots are callable */
        for (int i = 0; i < NUM_TR_SIZE; i++)
            g_cprim.cu[i].standard_dct = g_cprim.cu[i].dct;
        g_cprimReady = true;
    }
    return &g_cprim;
}

void* x265ref_encoder_table(void) { return &primitives; }

size_t x265ref_table_bytes(void) { return sizeof(EncoderPrimitives); }

int x265ref_depth(void) { return X265_DEPTH; }

/* Rebuild the encoder's global table the way x265_setup_primitives() does for a
 * C-only build (primitives.cpp:248-282): C prims, all-angs slots nulled, aliases. */
void x265ref_encoder_table_reset_c(void)
{
    memset(&primitives, 0, sizeof(primitives));
    setupCPrimitives(primitives);
    for (int i = 0; i < NUM_TR_SIZE; i++)
        primitives.cu[i].intra_pred_allangs = NULL;
    setupAliasPrimitives(primitives);
    for (int i = 0; i < NUM_TR_SIZE; i++)
        primitives.cu[i].standard_dct = primitives.cu[i].dct;
    g_cprimReady = false;   /* lowpass statics now point at `primitives`; rebuild g_cprim on next use */
}

/* A stronger host baseline that CAN be built here (round-4 verdict, optional 9): the reference's own SSE3 / SSSE3 / SSE4.1 intrinsic transforms
 * (common/vec/*.cpp - what x265_setup_primitives installs under ENABLE_ASSEMBLY before the NASM kernels, primitives.cpp:261-264) over the C table.
 * Only in the flavours built with them (oracle/Makefile refv3: -DX265REF_WITH_VEC); a table filler like x265hip_setup_primitives. */
int x265ref_sse_fill_table(void* table, size_t bytes, int depth)
{
#if X265REF_WITH_VEC
    if (!table || bytes != sizeof(EncoderPrimitives) || depth != X265_DEPTH) return -1;
    EncoderPrimitives& t = *(EncoderPrimitives*)table;
    EncoderPrimitives before = t;
    setupInstrinsicPrimitives(t, X265_CPU_SSE2 | X265_CPU_SSE3 | X265_CPU_SSSE3 | X265_CPU_SSE4);
    for (int i = 0; i < NUM_TR_SIZE; i++)
        t.cu[i].standard_dct = t.cu[i].dct;
    int n = 0;
    const void* const* a = (const void* const*)&before; const void* const* b = (const void* const*)&t;
    for (size_t i = 0; i < sizeof(EncoderPrimitives) / sizeof(void*); i++) n += a[i] != b[i];
    return n;
#else
    (void)table; (void)bytes; (void)depth;
    return -1;
#endif
}

} // extern "C"
