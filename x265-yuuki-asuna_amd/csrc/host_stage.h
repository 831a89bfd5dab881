// host_stage.h - per-thread staging used by the TABLE LAYER stubs (host pointers in, host
// pointers out).  The reference calls primitives concurrently from every pool worker without locks
// (SURVEY.md section 8b "Threading"), so each host thread owns a stream, a pinned host buffer and a
// device buffer with the same layout: operands are packed into the pinned buffer, shipped with one
// H2D copy, the batch-layer kernel runs on them (batch of one), results come back with one D2H copy.
// No error can be returned through the reference's signatures: any HIP failure aborts loudly - or, if the host opted in, hands the
// slots back to the host's own functions (see the error policy below).
#pragma once

#include "common.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace x265hip {

extern std::atomic<uint64_t> g_tableCalls;
extern std::atomic<uint64_t> g_stagesCreated, g_stagesReleased;

// Error policy of the table layer.  The reference's slot signatures cannot return an error, so by default any HIP failure inside a stub
// aborts loudly.  A host may opt in to X265HIP_ON_ERROR_RESTORE_HOST (x265hip_set_error_policy): the first failure is reported loudly
// once, the failing call and every later call of every GPU-backed slot are answered by the function the HOST had in that slot before
// x265hip_setup_primitives overwrote it (its own C / asm primitive - never code of this library), and x265hip_table_failures() counts.
extern std::atomic<int> g_errorPolicy;          // X265HIP_ON_ERROR_*
extern std::atomic<bool> g_tableFailed;
extern std::atomic<uint64_t> g_tableFailures;
extern std::atomic<long> g_injectFailureAt;     // test hook: the n-th stub call from now fails (0 = off)
struct StubFailure { };
[[noreturn]] void stub_failure(const char* what, const char* detail);

struct ThreadStage
{
    hipStream_t stream = nullptr;
    uint8_t* host = nullptr;      // pinned
    uint8_t* dev = nullptr;
    size_t cap = 0;
    size_t used = 0;              // bytes packed so far (inputs first, then outputs)
    size_t inBytes = 0;

    // a pool worker that exits (the reference tears its thread pools down in x265_encoder_close) gives its stream and staging back;
    // errors are ignored: at process exit the runtime may already be shutting down
    ~ThreadStage()
    {
        if (!stream) return;
        (void)hipStreamSynchronize(stream);
        if (host) (void)hipHostFree(host);
        if (dev) (void)hipFree(dev);
        (void)hipStreamDestroy(stream);
        stream = nullptr; host = nullptr; dev = nullptr; cap = 0;
        g_stagesReleased.fetch_add(1, std::memory_order_relaxed);
    }
    ThreadStage() = default;
    ThreadStage(const ThreadStage&) = delete;
    ThreadStage& operator=(const ThreadStage&) = delete;

    static void die(const char* what, hipError_t e) { stub_failure(what, hipGetErrorString(e)); }
    void ensure(size_t need)
    {
        if (!stream)
        {
            if (ensure_device()) stub_failure("device", x265hip_last_error());
            hipError_t e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
            if (e != hipSuccess) die("hipStreamCreate", e);
            g_stagesCreated.fetch_add(1, std::memory_order_relaxed);
        }
        if (need <= cap) return;
        size_t ncap = cap ? cap : (size_t)1 << 20;
        while (ncap < need) ncap <<= 1;
        uint8_t* nh = nullptr; uint8_t* nd = nullptr;
        hipError_t e = hipHostMalloc((void**)&nh, ncap, hipHostMallocDefault);
        if (e != hipSuccess) die("hipHostMalloc", e);
        e = hipMalloc((void**)&nd, ncap);
        if (e != hipSuccess) die("hipMalloc", e);
        if (used) memcpy(nh, host, used);
        if (host) (void)hipHostFree(host);
        if (dev) (void)hipFree(dev);
        host = nh; dev = nd; cap = ncap;
    }
    void begin()
    {
        used = 0; inBytes = 0;
        g_tableCalls.fetch_add(1, std::memory_order_relaxed);
        if (g_injectFailureAt.load(std::memory_order_relaxed) > 0 && g_injectFailureAt.fetch_sub(1) == 1)
            stub_failure("injected", "test hook x265hip_table_inject_failure");
        ensure(1 << 16);
    }
    size_t alloc(size_t bytes)
    {
        size_t off = (used + 63) & ~(size_t)63;
        ensure(off + bytes + 64);
        used = off + bytes;
        return off;
    }
    // pack a strided 2-D block (w x h elements of `es` bytes) contiguously; returns its offset
    size_t in2d(const void* p, intptr_t strideElems, int w, int h, int es)
    {
        size_t off = alloc((size_t)w * h * es);
        const uint8_t* s = (const uint8_t*)p;
        uint8_t* d = host + off;
        for (int y = 0; y < h; y++)
            memcpy(d + (size_t)y * w * es, s + (intptr_t)y * strideElems * es, (size_t)w * es);
        inBytes = used;
        return off;
    }
    size_t in1d(const void* p, size_t bytes)
    {
        size_t off = alloc(bytes);
        memcpy(host + off, p, bytes);
        inBytes = used;
        return off;
    }
    void upload()
    {
        if (!inBytes) return;
        hipError_t e = hipMemcpyAsync(dev, host, inBytes, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) die("hipMemcpyAsync H2D", e);
    }
    // bring [off, off+bytes) back and wait
    void download(size_t off, size_t bytes)
    {
        hipError_t e = hipMemcpyAsync(host + off, dev + off, bytes, hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) die("hipMemcpyAsync D2H", e);
        e = hipStreamSynchronize(stream);
        if (e != hipSuccess) die("hipStreamSynchronize", e);
    }
    void out2d(size_t off, void* p, intptr_t strideElems, int w, int h, int es) const
    {
        uint8_t* d = (uint8_t*)p;
        const uint8_t* s = host + off;
        for (int y = 0; y < h; y++)
            memcpy(d + (intptr_t)y * strideElems * es, s + (size_t)y * w * es, (size_t)w * es);
    }
    template <typename T> T* dptr(size_t off) const { return reinterpret_cast<T*>(dev + off); }
    template <typename T> T* hptr(size_t off) const { return reinterpret_cast<T*>(host + off); }
    void require(int rc, const char* what)
    {
        if (rc) stub_failure(what, x265hip_last_error());
    }
};

ThreadStage& thread_stage();
bool entropy_bits_ready();          // frame_coeff_kernels.hip: did the host hand its CABAC bit costs in (x265hip_set_entropy_bits)?

} // namespace x265hip
