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
