// table_api.hip - entry point of the TABLE LAYER and its per-thread staging state.
#include "host_stage.h"

struct x265hip_EncoderPrimitives;

namespace x265hip {

std::atomic<uint64_t> g_tableCalls{0};
std::atomic<uint64_t> g_stagesCreated{0}, g_stagesReleased{0};
std::atomic<int> g_errorPolicy{X265HIP_ON_ERROR_ABORT};
std::atomic<bool> g_tableFailed{false};
std::atomic<uint64_t> g_tableFailures{0};
std::atomic<long> g_injectFailureAt{0};

[[noreturn]] void stub_failure(const char* what, const char* detail)
{
    if (g_errorPolicy.load() == X265HIP_ON_ERROR_RESTORE_HOST)
    {
        if (!g_tableFailed.exchange(true))
            fprintf(stderr, "libx265hip: primitive stub failed (%s: %s); every GPU-backed slot is handed back to the host's own primitives\n", what, detail);
        g_tableFailures.fetch_add(1, std::memory_order_relaxed);
        throw StubFailure();
    }
    fprintf(stderr, "libx265hip: fatal in primitive stub: %s: %s (no CPU fallback)\n", what, detail);
    abort();
}

ThreadStage& thread_stage()
{
    static thread_local ThreadStage st;    // stream + pinned/device staging live as long as the thread
    return st;
}

int setup_primitives_d8(x265hip_EncoderPrimitives* p);
int setup_primitives_d10(x265hip_EncoderPrimitives* p);
int setup_primitives_d12(x265hip_EncoderPrimitives* p);

} // namespace x265hip

using namespace x265hip;

extern "C" int x265hip_setup_primitives(void* table, size_t table_bytes, int depth)
{
    if (!table) { set_error("setup_primitives: NULL table"); return X265HIP_EINVAL; }
    if (table_bytes != 2280 * sizeof(void*))
    {
        set_error("setup_primitives: table is %zu bytes, the x265 3.5 EncoderPrimitives layout is %zu", table_bytes, 2280 * sizeof(void*));
        return X265HIP_EINVAL;
    }
    int rc = ensure_device();          // refuse to install stubs that could only abort later
    if (rc) return rc;
    x265hip_EncoderPrimitives* p = (x265hip_EncoderPrimitives*)table;
    switch (depth)
    {
    case 8:  return setup_primitives_d8(p);
    case 10: return setup_primitives_d10(p);
    case 12: return setup_primitives_d12(p);
    default: set_error("setup_primitives: depth %d (8, 10 or 12)", depth); return X265HIP_EINVAL;
    }
}

extern "C" uint64_t x265hip_table_calls(void) { return g_tableCalls.load(); }
extern "C" int x265hip_set_error_policy(int policy)
{
    if (policy != X265HIP_ON_ERROR_ABORT && policy != X265HIP_ON_ERROR_RESTORE_HOST) { set_error("set_error_policy: %d", policy); return X265HIP_EINVAL; }
    g_errorPolicy.store(policy);
    return 0;
}
extern "C" uint64_t x265hip_table_failures(void) { return g_tableFailures.load(); }
/* test hook: the n-th stub call from now behaves as if HIP had failed (n <= 0 switches it off); also clears the failed state */
extern "C" void x265hip_table_inject_failure(long n) { g_tableFailed.store(false); g_injectFailureAt.store(n > 0 ? n : 0); }
/* per-thread staging (stream + pinned + device buffer) of the table layer: how many host threads created one / gave it back at thread exit */
extern "C" void x265hip_table_stage_counts(uint64_t* created, uint64_t* released)
{
    if (created) *created = g_stagesCreated.load();
    if (released) *released = g_stagesReleased.load();
}
