// Host-side plumbing shared by the entry points: error reporting, HIP call
// checking, RAII device buffers, kernel timing.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/lzma_amd.h"

namespace lzh {

struct ThreadState {
    char err[512];
    float ms_total, ms_finder, ms_coder;
    int coder_launches, coop_launches;
};
ThreadState &tls();
void release_decode_workspace();  // decode.hip: what the calling thread's decode calls keep
int noalloc_prepare();  // decode.hip: address of the no-allocation entry point's device array
int fail(int code, const char *fmt, ...);

#define LZ_HIP(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return lzh::fail(LZMA_AMD_ERR_HIP, "%s failed: %s (%s:%d)", #expr,            \
                             hipGetErrorString(e_), __FILE__, __LINE__);                  \
    } while (0)

// Device allocation released at scope exit.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    hipError_t alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n; else p = nullptr;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Device-side copies of a host call's buffers, kept by the calling thread from one call to the
// next (grow-only; lzma_amd_release_workspace frees them): the host entry points allocate
// nothing once they have seen their largest batch.
struct HostStage {
    static constexpr int kBufs = 6;
    DevBuf buf[kBufs];
    hipStream_t stream = nullptr;
    int device = -1;
    int prepare() {
        int dev = 0;
        LZ_HIP(hipGetDevice(&dev));
        if (dev != device) {
            release();
            device = dev;
            LZ_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        }
        return LZMA_AMD_OK;
    }
    int reserve(int i, size_t bytes) {
        if (buf[i].bytes < bytes || !buf[i].p) LZ_HIP(buf[i].alloc(bytes + bytes / 8));
        return LZMA_AMD_OK;
    }
    void release() {
        for (auto &b : buf) b.release();
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
        device = -1;
    }
};
HostStage &host_stage();  // the calling thread's

// Pair of events around a region of one stream.
struct Timer {
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t s = nullptr;
    bool on = false;
    void start(hipStream_t st) {
        s = st;
        if (!a) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); }
        on = hipEventRecord(a, s) == hipSuccess;
    }
    void stop() { if (on) (void)hipEventRecord(b, s); }
    float ms() {  // call after the stream was synchronized
        float t = 0.f;
        if (on && hipEventElapsedTime(&t, a, b) != hipSuccess) t = 0.f;
        return t;
    }
    ~Timer() { if (a) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } }
};

}  // namespace lzh
