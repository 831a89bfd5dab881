// recon_publish.hip - the multi-GPU seam behind the C ABI: hand a finished band of reconstructed CTU rows (Y, Cb, Cr incl. their margins)
// from the GPU that produced it to the GPU(s) whose in-flight pictures search it - where the reference raises m_reconRowFlag
// (encoder/framefilter.cpp:664) and its consumers wait (encoder/frameencoder.cpp:852-868).  One process per GPU; the host owns an
// RCCL communicator (ncclCommInitRank over its own bootstrap) and passes it in, so this file only issues the data movement:
//   x265hip_recon_publish_rows  : root -> everyone (ncclBroadcast, the one-to-many hand-off of SURVEY.md section 8e) or root -> one
//                                 peer (ncclSend / ncclRecv: xGMI is point to point, a picture with one consumer needs one link)
// in ONE group call for the three planes, on the stream the caller names - a copy stream, so the next band's kernels overlap.
// RCCL is resolved at run time (dlopen of librccl.so): libx265hip.so keeps linking against the HIP runtime only, and a single-GPU host
// never loads it.  Python hosts use torch.distributed (the same RCCL) through pipeline.FrameParallelRing instead.
#include "common.h"

#include <cstring>
#include <dlfcn.h>
#include <mutex>

using namespace x265hip;

namespace {

typedef int (*nccl_bcast_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_p2p_t)(void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_void_t)(void);
typedef const char* (*nccl_err_t)(int);
typedef int (*nccl_uid_t)(void*);                       // ncclGetUniqueId(ncclUniqueId*): 128 bytes
struct NcclUid { char b[128]; };
typedef int (*nccl_init_t)(void**, int, NcclUid, int);  // ncclCommInitRank(ncclComm_t*, nranks, ncclUniqueId BY VALUE, rank)
typedef int (*nccl_destroy_t)(void*);

struct Rccl
{
    void* lib = nullptr;
    nccl_bcast_t bcast = nullptr;
    nccl_p2p_t send = nullptr, recv = nullptr;
    nccl_void_t groupStart = nullptr, groupEnd = nullptr;
    nccl_err_t errStr = nullptr;
    nccl_uid_t uid = nullptr;
    nccl_init_t init = nullptr;
    nccl_destroy_t destroy = nullptr;
} g_rccl;
std::once_flag g_rcclOnce;

void load_rccl()
{
    // The SONAME first: a process that already holds an RCCL (a torch host: torch/lib/librccl.so, SONAME librccl.so.1) gets THAT copy back -
    // one RCCL instance with several communicators, the arrangement RCCL is used in everywhere - instead of a second copy from the
    // library path next to it.  RTLD_LOCAL: every entry is taken with dlsym from the handle, nothing is offered for interposition.
    const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so" };
    for (const char* n : names)
        if ((g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)))
            break;
    if (!g_rccl.lib) return;
    g_rccl.bcast = (nccl_bcast_t)dlsym(g_rccl.lib, "ncclBroadcast");
    g_rccl.send = (nccl_p2p_t)dlsym(g_rccl.lib, "ncclSend");
    g_rccl.recv = (nccl_p2p_t)dlsym(g_rccl.lib, "ncclRecv");
    g_rccl.groupStart = (nccl_void_t)dlsym(g_rccl.lib, "ncclGroupStart");
    g_rccl.groupEnd = (nccl_void_t)dlsym(g_rccl.lib, "ncclGroupEnd");
    g_rccl.errStr = (nccl_err_t)dlsym(g_rccl.lib, "ncclGetErrorString");
    g_rccl.uid = (nccl_uid_t)dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.init = (nccl_init_t)dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.destroy = (nccl_destroy_t)dlsym(g_rccl.lib, "ncclCommDestroy");
}

int need_rccl(const char* who)
{
    std::call_once(g_rcclOnce, load_rccl);
    if (g_rccl.lib && g_rccl.bcast && g_rccl.send && g_rccl.recv && g_rccl.groupStart && g_rccl.groupEnd && g_rccl.uid && g_rccl.init && g_rccl.destroy) return 0;
    const char* de = dlerror();          // ONE call: dlerror() clears the state it reports
    set_error("%s: librccl.so (ncclBroadcast / ncclSend / ncclRecv / ncclCommInitRank) is not available: %s", who, de ? de : "symbols missing");
    return X265HIP_ENODEV;
}

enum { NCCL_UINT8 = 1 };      // ncclUint8 / ncclChar family: rccl.h ncclDataType_t { ncclInt8 = 0, ncclUint8 = 1, ... }

} // namespace

// mode: 0 = broadcast over the whole communicator, 1 = point to point (the producer sends to peers[0 .. npeers), anyone else receives from root)
static int publish(const x265hip_recon_publish_params* p, int mode, int npeers, const int* peers, void* stream)
{
    if (!p || !p->comm || !p->plane[0]) { set_error("recon_publish_rows: NULL communicator / plane"); return X265HIP_EINVAL; }
    if (p->depth != 8 && p->depth != 10 && p->depth != 12) { set_error("recon_publish_rows: depth %d", p->depth); return X265HIP_EINVAL; }
    if (p->ctu_rows <= 0 || p->ctu_row0 < 0 || (p->ctu_row0 + p->ctu_rows) * 64 > p->height || (p->height & 63) || p->height <= 0)
    { set_error("recon_publish_rows: rows [%d, %d) of a picture of %d rows", p->ctu_row0, p->ctu_row0 + p->ctu_rows, p->height / 64); return X265HIP_EINVAL; }
    if (p->rank < 0 || p->root < 0 || p->peer < -1) { set_error("recon_publish_rows: rank %d root %d peer %d", p->rank, p->root, p->peer); return X265HIP_EINVAL; }
    if (p->stride <= 0 || p->margin_y < 0 || (p->plane[1] && (p->stride_c <= 0 || p->margin_y_c < 0 || !p->plane[2])))
    { set_error("recon_publish_rows: plane geometry"); return X265HIP_EINVAL; }
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = need_rccl("recon_publish_rows"))) return rc;
    const int bpp = p->depth == 8 ? 1 : 2;
    const bool first = p->ctu_row0 == 0, last = (p->ctu_row0 + p->ctu_rows) * 64 == p->height;
    // the band's rows as one contiguous slice of each padded plane: whole rows (side margins included), plus the top margin with the
    // first band and the bottom margin with the last one.  plane[] = allocation starts.
    struct Slice { uint8_t* ptr; size_t bytes; } sl[3];
    int ns = 0;
    {
        const long y0 = p->margin_y + (long)p->ctu_row0 * 64 - (first ? p->margin_y : 0), y1 = p->margin_y + (long)(p->ctu_row0 + p->ctu_rows) * 64 + (last ? p->margin_y : 0);
        sl[ns++] = { (uint8_t*)p->plane[0] + (size_t)y0 * p->stride * bpp, (size_t)(y1 - y0) * p->stride * bpp };
        if (p->plane[1])
        {
            const long c0 = p->margin_y_c + (long)p->ctu_row0 * 32 - (first ? p->margin_y_c : 0), c1 = p->margin_y_c + (long)(p->ctu_row0 + p->ctu_rows) * 32 + (last ? p->margin_y_c : 0);
            for (int i = 1; i < 3; i++)
                sl[ns++] = { (uint8_t*)p->plane[i] + (size_t)c0 * p->stride_c * bpp, (size_t)(c1 - c0) * p->stride_c * bpp };
        }
    }
    hipStream_t s = (hipStream_t)stream;
    int e = g_rccl.groupStart();
    if (e) { set_error("recon_publish_rows: ncclGroupStart failed: %d (%s)", e, g_rccl.errStr ? g_rccl.errStr(e) : "?"); return X265HIP_ENODEV; }
    for (int i = 0; i < ns && !e; i++)
    {
        if (mode == 0) e = g_rccl.bcast(sl[i].ptr, sl[i].ptr, sl[i].bytes, NCCL_UINT8, p->root, p->comm, s);            // one-to-many over the whole communicator
        else if (p->rank == p->root)                                                                                          // producer -> each of its consumers: xGMI is point to point,
            for (int k = 0; k < npeers && !e; k++) e = g_rccl.send(sl[i].ptr, sl[i].bytes, NCCL_UINT8, peers[k], p->comm, s); //   a picture with k consumers needs k links, all in one group
        else e = g_rccl.recv(sl[i].ptr, sl[i].bytes, NCCL_UINT8, p->root, p->comm, s);                                       // a consumer's side
    }
    const int e2 = g_rccl.groupEnd();
    if (e || e2)
    {
        set_error("recon_publish_rows: RCCL error %d (%s)", e ? e : e2, g_rccl.errStr ? g_rccl.errStr(e ? e : e2) : "?");
        return X265HIP_ENODEV;
    }
    return 0;
}

extern "C" int x265hip_recon_publish_rows(const x265hip_recon_publish_params* p, void* stream)
{
    if (p && p->peer >= 0) { const int peer = p->peer; return publish(p, 1, 1, &peer, stream); }
    return publish(p, 0, 0, nullptr, stream);
}

/* the producer's band to SEVERAL consumers (a picture that is the reference of several in-flight pictures: preset slow has 4 references
 * and B pictures) as one group of point-to-point sends; a consumer calls it with rank != root and receives from root (peers ignored) */
extern "C" int x265hip_recon_publish_rows_to(const x265hip_recon_publish_params* p, int npeers, const int* peers, void* stream)
{
    // only the producer's peer list means anything; a consumer may pass (0, NULL) as documented and is routed to the receive branch, never
    // to the broadcast one (round-3 advisor)
    const bool producer = p && p->rank == p->root;
    if (producer)
    {
        if (npeers < 1 || npeers > 16 || !peers) { set_error("recon_publish_rows_to: %d peers", npeers); return X265HIP_EINVAL; }
        for (int k = 0; k < npeers; k++) if (peers[k] < 0) { set_error("recon_publish_rows_to: peer %d", peers[k]); return X265HIP_EINVAL; }
    }
    return publish(p, 1, producer ? npeers : 0, producer ? peers : nullptr, stream);
}

/* Communicator plumbing for a host that has no RCCL binding of its own (one process per GPU): rank 0 makes the 128-byte id, the host
 * ships it to the other ranks by whatever it has (a file, a socket, torch.distributed), every rank joins on its CURRENT device. */
extern "C" int x265hip_comm_unique_id(void* id128)
{
    if (!id128) { set_error("comm_unique_id: NULL"); return X265HIP_EINVAL; }
    int rc = need_rccl("comm_unique_id");
    if (rc) return rc;
    const int e = g_rccl.uid(id128);
    if (e) { set_error("comm_unique_id: RCCL error %d (%s)", e, g_rccl.errStr ? g_rccl.errStr(e) : "?"); return X265HIP_ENODEV; }
    return 0;
}

extern "C" int x265hip_comm_init(void** comm, int nranks, const void* id128, int rank)
{
    if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) { set_error("comm_init: bad argument"); return X265HIP_EINVAL; }
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = need_rccl("comm_init"))) return rc;
    NcclUid id;
    memcpy(id.b, id128, sizeof(id.b));
    const int e = g_rccl.init(comm, nranks, id, rank);
    if (e) { set_error("comm_init: RCCL error %d (%s)", e, g_rccl.errStr ? g_rccl.errStr(e) : "?"); return X265HIP_ENODEV; }
    return 0;
}

extern "C" int x265hip_comm_destroy(void* comm)
{
    if (!comm) return 0;
    int rc = need_rccl("comm_destroy");
    if (rc) return rc;
    return g_rccl.destroy(comm) ? X265HIP_ENODEV : 0;
}
