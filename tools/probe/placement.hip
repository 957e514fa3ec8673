// Probe: where do single-wave workgroups land?  Prints, for a grid of 64-thread
// workgroups that are all resident at once, how many share a CU and a SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int VG>
__global__ __launch_bounds__(64) void where(unsigned *ids, unsigned long long ticks) {
    extern __shared__ unsigned lds[];
    lds[threadIdx.x] = threadIdx.x;
    if (VG >= 150) asm volatile("v_mov_b32 v149, 0" ::: "v149");
    else if (VG >= 100) asm volatile("v_mov_b32 v99, 0" ::: "v99");
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0) {
        ids[2 * blockIdx.x] = hw;
        ids[2 * blockIdx.x + 1] = xcc;
    }
}

int main(int argc, char **argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 1792;
    const int lds = argc > 2 ? atoi(argv[2]) : 22112;
    const int vg = argc > 3 ? atoi(argv[3]) : 150;
    unsigned *d;
    CK(hipMalloc(&d, grid * 8));
    if (vg >= 150) hipLaunchKernelGGL(where<150>, dim3(grid), dim3(64), lds, 0, d, 20000000ull);
    else if (vg >= 100) hipLaunchKernelGGL(where<100>, dim3(grid), dim3(64), lds, 0, d, 20000000ull);
    else hipLaunchKernelGGL(where<0>, dim3(grid), dim3(64), lds, 0, d, 20000000ull);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(2 * grid);
    CK(hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost));
    std::map<unsigned, int> per_cu, per_simd;
    for (int i = 0; i < grid; i++) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const unsigned cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        per_cu[cukey]++;
        per_simd[(cukey << 2) | simd]++;
    }
    std::map<int, int> hist_cu, hist_simd;
    for (auto &kv : per_cu) hist_cu[kv.second]++;
    for (auto &kv : per_simd) hist_simd[kv.second]++;
    printf("grid %d lds %d vgprs>=%d: %zu CUs used, %zu SIMDs used\n", grid, lds, vg, per_cu.size(), per_simd.size());
    printf("  workgroups per CU   :");
    for (auto &kv : hist_cu) printf("  %d WG x %d CUs", kv.first, kv.second);
    printf("\n  workgroups per SIMD :");
    for (auto &kv : hist_simd) printf("  %d WG x %d SIMDs", kv.first, kv.second);
    printf("\n  first 12 raw (hw_id simd cu se xcc):");
    for (int i = 0; i < 12 && i < grid; i++) printf(" [%u %u %u %u]", (h[2*i] >> 4) & 3, (h[2*i] >> 8) & 15, (h[2*i] >> 13) & 7, h[2*i+1] & 15);
    printf("\n");
    return 0;
}
