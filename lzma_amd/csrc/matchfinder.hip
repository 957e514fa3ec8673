// Exhaustive BT4 match finder for gfx950, position-parallel.
//
// Replaces lzBinTree.getMatches/skip of the reference (lz_bin_tree.go:115-337) run
// at every position of a stream.  Pipeline per batch of positions:
//   1. key kernels: hash-2 / hash-3 / hash-4 slots (lz_bin_tree.go:139-143) and the
//      first 8 bytes of every position;
//   2. stable radix sorts of (key, position) -- rocPRIM device radix sort, the one
//      library primitive used -- giving each bucket's positions in ascending order;
//   3. scatter kernels: rank of every position in its bucket, previous position with
//      the same hash-2 / hash-3 slot;
//   4. the scan kernel (lzma_mf_core.hpp): one wave per position, 64 candidates per
//      step, record breakers by ballot.  Run twice (count, then fill at the offsets
//      of an exclusive scan) so that the output is ordered by position.
//
// Roofline class: HBM (N input bytes read, 4N count bytes + 8 bytes per pair
// written); the scan is latency- and issue-bound on gathers through L2.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "mf_tables.hpp"

namespace {

using lz::kNone;

enum { KEY2 = 2, KEY3 = 3, KEY4 = 4, KEY8 = 8 };

template <int KIND, class K>
__global__ __launch_bounds__(256) void mf_key_kernel(const uint8_t *__restrict__ data, uint32_t n,
                                                     uint32_t mask, K *__restrict__ keys) {
    __shared__ uint32_t crc[256];
    crc[threadIdx.x] = lz::crc32_entry(threadIdx.x);
    __syncthreads();
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t w = lz::mf_load8(data, n, i);
        if (KIND == KEY8) {
            keys[i] = (K)w;
        } else {
            const uint32_t b0 = w & 0xFF, b1 = (w >> 8) & 0xFF, b2 = (w >> 16) & 0xFF,
                           b3 = (w >> 24) & 0xFF;
            uint32_t t = crc[b0] ^ b1;  // lz_bin_tree.go:139-143
            if (KIND == KEY2) {
                keys[i] = (K)(t & (lz::kHash2Size - 1));
            } else {
                t ^= b2 << 8;
                if (KIND == KEY3)
                    keys[i] = (K)(t & (lz::kHash3Size - 1));
                else
                    keys[i] = (K)((t ^ (crc[b3] << 5)) & mask);
            }
        }
    }
}

// rank[ord[j]] = j
__global__ __launch_bounds__(256) void mf_rank_kernel(const uint32_t *__restrict__ ord, uint32_t n,
                                                      uint32_t *__restrict__ rank) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) rank[ord[j]] = j;
}

// prev[ord[j]] = ord[j-1] if it has the same key, else kNone
__global__ __launch_bounds__(256) void mf_prev_kernel(const uint32_t *__restrict__ ord,
                                                      const uint32_t *__restrict__ skey, uint32_t n,
                                                      uint32_t *__restrict__ prev) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride)
        prev[ord[j]] = (j > 0 && skey[j - 1] == skey[j]) ? ord[j - 1] : kNone;
}

unsigned grid_for(uint32_t n, unsigned block, unsigned max_blocks = 256 * 32) {
    unsigned g = (n + block - 1) / block;
    if (g < 1) g = 1;
    return g < max_blocks ? g : max_blocks;
}

template <class K>
int sort_by_key(K *keys_in, K *keys_out, uint32_t *vals_out, uint32_t n, unsigned end_bit,
                hipStream_t st) {
    rocprim::counting_iterator<uint32_t> iota(0);
    size_t bytes = 0;
    LZ_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, iota, vals_out, n, 0, end_bit,
                                     st));
    lzh::DevBuf temp;
    LZ_HIP(temp.alloc(bytes));
    LZ_HIP(rocprim::radix_sort_pairs(temp.p, bytes, keys_in, keys_out, iota, vals_out, n, 0, end_bit,
                                     st));
    LZ_HIP(hipStreamSynchronize(st));  // temp is released at scope exit
    return LZMA_AMD_OK;
}

unsigned bits_of(uint32_t mask) {
    unsigned b = 0;
    while (b < 32 && (mask >> b) != 0) b++;
    return b ? b : 1;
}

// Stand-alone finder: one wave per position.  FILL == false: counts only.
template <bool FILL>
__global__ __launch_bounds__(256) void mf_scan_kernel(lz::MfTables t, uint32_t n,
                                                      uint32_t *__restrict__ counts,
                                                      const uint64_t *__restrict__ offsets,
                                                      uint32_t *__restrict__ pairs, uint64_t pairs_cap) {
    __shared__ uint32_t s_out[4][2 * (lz::kMatchMax + 1) + 2];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nwaves = gridDim.x * 4;
    uint32_t *out = s_out[wave];
    for (uint32_t P = blockIdx.x * 4 + wave; P < n; P += nwaves) {
        const uint32_t c = lz::mf_find(t, P, 0u, n, out, lane);
        if (!FILL) {
            if (lane == 0) counts[P] = c;
        } else {
            const uint64_t off = offsets[P];
            if (off + c <= pairs_cap)
                for (uint32_t k = lane; k < 2 * c; k += 64) pairs[2 * off + k] = out[k];
        }
    }
}

}  // namespace

namespace lzh {

int MfWorkspace::build(const uint8_t *d_data, uint32_t n, uint32_t dict_size, uint32_t fb,
                       hipStream_t st) {
    const uint32_t mask = lz::bt4_hash_mask(dict_size);
    const size_t w = (size_t)(n ? n : 1) * sizeof(uint32_t);
    LZ_HIP(ord4.alloc(w));
    LZ_HIP(skey4.alloc(w));
    LZ_HIP(rank4.alloc(w));
    LZ_HIP(ord8.alloc(w));
    LZ_HIP(rank8.alloc(w));
    LZ_HIP(prev2.alloc(w));
    LZ_HIP(prev3.alloc(w));
    view.data = d_data;
    view.n = n;
    view.ord4 = ord4.as<uint32_t>();
    view.skey4 = skey4.as<uint32_t>();
    view.rank4 = rank4.as<uint32_t>();
    view.ord8 = ord8.as<uint32_t>();
    view.rank8 = rank8.as<uint32_t>();
    view.prev2 = prev2.as<uint32_t>();
    view.prev3 = prev3.as<uint32_t>();
    view.dict = dict_size;
    view.fb = fb;
    view.use8 = fb >= 8;
    if (n == 0) return LZMA_AMD_OK;

    DevBuf kin, kout, vout;  // temporaries sized for the 64-bit keys
    LZ_HIP(kin.alloc((size_t)n * 8));
    LZ_HIP(kout.alloc((size_t)n * 8));
    LZ_HIP(vout.alloc(w));
    const unsigned g = grid_for(n, 256);
    int rc;
    // hash-4 buckets
    hipLaunchKernelGGL((mf_key_kernel<KEY4, uint32_t>), dim3(g), dim3(256), 0, st, d_data, n, mask,
                       kin.as<uint32_t>());
    if ((rc = sort_by_key(kin.as<uint32_t>(), skey4.as<uint32_t>(), ord4.as<uint32_t>(), n,
                          bits_of(mask), st)))
        return rc;
    hipLaunchKernelGGL(mf_rank_kernel, dim3(g), dim3(256), 0, st, ord4.as<uint32_t>(), n,
                       rank4.as<uint32_t>());
    // 8-gram groups
    if (view.use8) {
        hipLaunchKernelGGL((mf_key_kernel<KEY8, uint64_t>), dim3(g), dim3(256), 0, st, d_data, n, 0u,
                           kin.as<uint64_t>());
        if ((rc = sort_by_key(kin.as<uint64_t>(), kout.as<uint64_t>(), ord8.as<uint32_t>(), n, 64, st)))
            return rc;
        hipLaunchKernelGGL(mf_rank_kernel, dim3(g), dim3(256), 0, st, ord8.as<uint32_t>(), n,
                           rank8.as<uint32_t>());
    }
    // hash-3 and hash-2 heads
    hipLaunchKernelGGL((mf_key_kernel<KEY3, uint32_t>), dim3(g), dim3(256), 0, st, d_data, n, 0u,
                       kin.as<uint32_t>());
    if ((rc = sort_by_key(kin.as<uint32_t>(), kout.as<uint32_t>(), vout.as<uint32_t>(), n, 16, st)))
        return rc;
    hipLaunchKernelGGL(mf_prev_kernel, dim3(g), dim3(256), 0, st, vout.as<uint32_t>(),
                       kout.as<uint32_t>(), n, prev3.as<uint32_t>());
    hipLaunchKernelGGL((mf_key_kernel<KEY2, uint32_t>), dim3(g), dim3(256), 0, st, d_data, n, 0u,
                       kin.as<uint32_t>());
    if ((rc = sort_by_key(kin.as<uint32_t>(), kout.as<uint32_t>(), vout.as<uint32_t>(), n, 10, st)))
        return rc;
    hipLaunchKernelGGL(mf_prev_kernel, dim3(g), dim3(256), 0, st, vout.as<uint32_t>(),
                       kout.as<uint32_t>(), n, prev2.as<uint32_t>());
    LZ_HIP(hipGetLastError());
    LZ_HIP(hipStreamSynchronize(st));  // temporaries go out of scope
    return LZMA_AMD_OK;
}

}  // namespace lzh

extern "C" int lzma_amd_bt4_matches_dev(const uint8_t *d_data, size_t n, uint32_t dict_size,
                                        uint32_t fb, uint32_t *d_counts, uint64_t *d_offsets,
                                        uint32_t *d_pairs, size_t pairs_cap, uint64_t *d_total_pairs,
                                        void *stream) {
    hipStream_t st = (hipStream_t)stream;
    lzh::tls().ms_total = lzh::tls().ms_finder = lzh::tls().ms_coder = 0.f;
    if (dict_size < 1 || fb < 5 || fb > lz::kMatchMax)
        return lzh::fail(LZMA_AMD_ERR_ARG, "bt4_matches: dict_size/fb out of range");
    if (n >= (1ull << 31))
        return lzh::fail(LZMA_AMD_ERR_ARG, "bt4_matches: at most 2^31-1 bytes per call");
    if (!d_counts || !d_total_pairs || (n && !d_data))
        return lzh::fail(LZMA_AMD_ERR_ARG, "bt4_matches: null pointer argument");
    if (n == 0) {
        LZ_HIP(hipMemsetAsync(d_total_pairs, 0, 8, st));
        LZ_HIP(hipStreamSynchronize(st));
        return LZMA_AMD_OK;
    }
    const uint32_t N = (uint32_t)n;
    lzh::Timer t_all, t_scan;
    t_all.start(st);
    lzh::MfWorkspace ws;
    int rc = ws.build(d_data, N, dict_size, fb, st);
    if (rc) return rc;
    lzh::DevBuf offs;  // n+1 exclusive offsets
    uint64_t *d_off = d_offsets;
    LZ_HIP(offs.alloc((n + 1) * 8));
    const unsigned g = grid_for(N, 4, 256 * 64);
    t_scan.start(st);
    hipLaunchKernelGGL(mf_scan_kernel<false>, dim3(g), dim3(256), 0, st, ws.view, N, d_counts,
                       (const uint64_t *)nullptr, (uint32_t *)nullptr, (uint64_t)0);
    {
        size_t bytes = 0;
        auto in = rocprim::make_transform_iterator(d_counts, [] __device__(uint32_t c) { return (uint64_t)c; });
        LZ_HIP(rocprim::exclusive_scan(nullptr, bytes, in, offs.as<uint64_t>(), (uint64_t)0, n,
                                       rocprim::plus<uint64_t>(), st));
        lzh::DevBuf temp;
        LZ_HIP(temp.alloc(bytes));
        LZ_HIP(rocprim::exclusive_scan(temp.p, bytes, in, offs.as<uint64_t>(), (uint64_t)0, n,
                                       rocprim::plus<uint64_t>(), st));
        LZ_HIP(hipStreamSynchronize(st));
    }
    // total = offs[n-1] + counts[n-1]
    uint64_t last_off = 0;
    uint32_t last_cnt = 0;
    LZ_HIP(hipMemcpyAsync(&last_off, offs.as<uint64_t>() + (n - 1), 8, hipMemcpyDeviceToHost, st));
    LZ_HIP(hipMemcpyAsync(&last_cnt, d_counts + (n - 1), 4, hipMemcpyDeviceToHost, st));
    LZ_HIP(hipStreamSynchronize(st));
    const uint64_t total = last_off + last_cnt;
    LZ_HIP(hipMemcpyAsync(d_total_pairs, &total, 8, hipMemcpyHostToDevice, st));
    if (d_off) LZ_HIP(hipMemcpyAsync(d_off, offs.p, n * 8, hipMemcpyDeviceToDevice, st));
    if (d_pairs && total <= pairs_cap) {
        hipLaunchKernelGGL(mf_scan_kernel<true>, dim3(g), dim3(256), 0, st, ws.view, N, d_counts,
                           offs.as<uint64_t>(), d_pairs, (uint64_t)pairs_cap);
    }
    t_scan.stop();
    t_all.stop();
    LZ_HIP(hipGetLastError());
    LZ_HIP(hipStreamSynchronize(st));
    lzh::tls().ms_total = t_all.ms();
    lzh::tls().ms_finder = t_scan.ms();
    if (total > pairs_cap)
        return lzh::fail(LZMA_AMD_ERR_OUT_CAP, "bt4_matches: %llu pairs exceed capacity %zu",
                         (unsigned long long)total, pairs_cap);
    return LZMA_AMD_OK;
}

extern "C" int lzma_amd_bt4_matches(const uint8_t *data, size_t n, uint32_t dict_size, uint32_t fb,
                                    uint32_t *counts, uint64_t *offsets, uint32_t *pairs,
                                    size_t pairs_cap, size_t *total_pairs) {
    if (!counts || !total_pairs || (n && !data))
        return lzh::fail(LZMA_AMD_ERR_ARG, "bt4_matches: null pointer argument");
    *total_pairs = 0;
    lzh::DevBuf d_data, d_counts, d_off, d_pairs, d_total;
    LZ_HIP(d_data.alloc(n));
    LZ_HIP(d_counts.alloc(n * 4));
    LZ_HIP(d_off.alloc(n * 8));
    LZ_HIP(d_pairs.alloc(pairs_cap * 8));
    LZ_HIP(d_total.alloc(8));
    if (n) LZ_HIP(hipMemcpy(d_data.p, data, n, hipMemcpyHostToDevice));
    int rc = lzma_amd_bt4_matches_dev(d_data.as<uint8_t>(), n, dict_size, fb, d_counts.as<uint32_t>(),
                                      d_off.as<uint64_t>(), pairs ? d_pairs.as<uint32_t>() : nullptr,
                                      pairs_cap, d_total.as<uint64_t>(), nullptr);
    if (rc != LZMA_AMD_OK && rc != LZMA_AMD_ERR_OUT_CAP) return rc;
    uint64_t total = 0;
    LZ_HIP(hipMemcpy(&total, d_total.p, 8, hipMemcpyDeviceToHost));
    *total_pairs = (size_t)total;
    if (n) LZ_HIP(hipMemcpy(counts, d_counts.p, n * 4, hipMemcpyDeviceToHost));
    if (offsets && n) LZ_HIP(hipMemcpy(offsets, d_off.p, n * 8, hipMemcpyDeviceToHost));
    if (rc == LZMA_AMD_OK && pairs && total)
        LZ_HIP(hipMemcpy(pairs, d_pairs.p, total * 8, hipMemcpyDeviceToHost));
    return rc;
}
