// Probe: which kind of kernel stalls when launched on a second stream while a
// long-running kernel of small LDS-heavy workgroups is resident?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int VG, bool BUSY>
__global__ __launch_bounds__(64) void resident(unsigned long long cycles, unsigned *sink) {
    extern __shared__ unsigned lds[];
    lds[threadIdx.x] = threadIdx.x;
    if (VG >= 150) asm volatile("v_mov_b32 v149, 0" ::: "v149");
    const unsigned long long t0 = wall_clock64();
    unsigned acc = 0;
    while (wall_clock64() - t0 < cycles) {  // wall clock: 100 MHz
        acc += lds[(threadIdx.x + acc) & 63];
        if (!BUSY) __builtin_amdgcn_s_sleep(8);
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

template <int BLOCK, bool LDS, bool SCRATCH>
__global__ __launch_bounds__(BLOCK) void victim(const unsigned *in, unsigned *out, unsigned n, unsigned sel) {
    __shared__ unsigned sh[LDS ? 5120 : 1];
    unsigned priv[96];
    unsigned i = blockIdx.x * BLOCK + threadIdx.x;
    unsigned v = i < n ? in[i] : 0;
    if (LDS) {
        sh[threadIdx.x] = v;
        __syncthreads();
        v += sh[(threadIdx.x + 1) % BLOCK];
    }
    if (SCRATCH) {
#pragma unroll 1
        for (unsigned k = 0; k < 96; k++) priv[(k * sel) % 96] = v + k;  // dynamic index: forces scratch
        v = priv[(v + sel) % 96];
    }
    if (i < n) out[i] = v;
}

template <int BLOCK, bool LDS, bool SCRATCH>
float run_victim(hipStream_t s, const unsigned *in, unsigned *out, unsigned n, const char *name) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, s));
    hipLaunchKernelGGL((victim<BLOCK, LDS, SCRATCH>), dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, in, out, n, 3u);
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("  %-40s %9.2f ms\n", name, ms);
    fflush(stdout);
    return ms;
}

int main(int argc, char **argv) {
    const unsigned n = 64u << 20;
    const int res_wgs = argc > 1 ? atoi(argv[1]) : 256;
    const int res_lds = argc > 2 ? atoi(argv[2]) : 22112;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;  // 1: 150 VGPRs, 2: busy, 3: both
    unsigned *in, *out, *sink;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(in, 1, n * 4));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int phase = 1; phase < 2; phase++) {
        printf(phase ? "with %d resident workgroups (64 threads, %d B LDS, mode %d) on another stream:\n" : "alone:\n", res_wgs, res_lds, mode);
        if (phase) {
            const unsigned long long ticks = 300000000ull;  // 3 s of the 100 MHz wall clock
            if (mode == 0) hipLaunchKernelGGL((resident<0, false>), dim3(res_wgs), dim3(64), res_lds, s2, ticks, sink);
            if (mode == 1) hipLaunchKernelGGL((resident<150, false>), dim3(res_wgs), dim3(64), res_lds, s2, ticks, sink);
            if (mode == 2) hipLaunchKernelGGL((resident<0, true>), dim3(res_wgs), dim3(64), res_lds, s2, ticks, sink);
            if (mode == 3) hipLaunchKernelGGL((resident<150, true>), dim3(res_wgs), dim3(64), res_lds, s2, ticks, sink);
            CK(hipGetLastError());
        }
        run_victim<256, false, false>(s1, in, out, n, "256 threads");
        run_victim<256, true, false>(s1, in, out, n, "256 threads + LDS");
        run_victim<256, false, true>(s1, in, out, n, "256 threads + scratch");
        run_victim<1024, false, false>(s1, in, out, n, "1024 threads");
        run_victim<1024, true, false>(s1, in, out, n, "1024 threads + LDS");
        run_victim<1024, true, true>(s1, in, out, n, "1024 threads + LDS + scratch");
        run_victim<256, false, true>(s1, in, out, n, "256 threads + scratch (again)");
    }
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
