// table_api.hip - entry point of the TABLE LAYER and its per-thread staging state.
#include "host_stage.h"

struct x265hip_EncoderPrimitives;

namespace x265hip {

std::atomic<uint64_t> g_tableCalls{0};
std::atomic<uint64_t> g_stagesCreated{0}, g_stagesReleased{0};

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
/* per-thread staging (stream + pinned + device buffer) of the table layer: how many host threads created one / gave it back at thread exit */
extern "C" void x265hip_table_stage_counts(uint64_t* created, uint64_t* released)
{
    if (created) *created = g_stagesCreated.load();
    if (released) *released = g_stagesReleased.load();
}
