// Exhaustive BT4 match finder for gfx950, position-parallel.
//
// Replaces lzBinTree.getMatches/skip of the reference (lz_bin_tree.go:115-337) run
// at every position of a stream.  Pipeline per batch of positions:
//   1. key kernels: hash-2 / hash-3 / hash-4 slots (lz_bin_tree.go:139-143) and the first
//      6 and 8 bytes of every position;
//   2. stable radix sorts of (key, position) -- rocPRIM's device radix sort when the finder
//      has the device to itself, its own (rs_* kernels) beside resident encode launches --
//      giving each group's positions in ascending order.  The
//      16-gram table needs no sort over 16 bytes: the positions in 8-gram order, each
//      moved back by 8, are already ordered by their bytes 8..15, and one stable sort by
//      the 8-gram group of bytes 0..7 (a 31-bit key) finishes the order; the 32-, 64- and
//      128-gram tables built for searches of 256 bytes and more come the same way, each
//      from the one before it;
//   3. table kernels: 12-byte entries (position, 8 bytes of its text; a flag on the oldest
//      entry of each group) in sorted order for the hash-4 buckets and the 6-, 8- and
//      16-gram (... 128-gram) groups, and per position its rank in each table and the
//      previous position with the same hash-2 / hash-3 slot;
//   4. the scan kernel (lzma_mf_core.hpp): four positions per wave, one per 16-lane row,
//      16 candidates of every level per step, record breakers by a prefix maximum over the
//      row.  The loads of a position (its record, then the newest entries of its groups)
//      are issued one and two positions ahead of the position being worked on.  Each row
//      walks a run of consecutive positions and appends their pairs to a private page of a
//      global arena (one atomic per 4096 pairs); a word per position records where its
//      pairs are.
//   5. (stand-alone entry point only) an exclusive scan of the counts and a gather
//      put the pairs in position order.
//
// Roofline class: HBM (N input bytes read, 4N count bytes + 8 bytes per pair
// written).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "mf_tables.hpp"

namespace {

using lz::kMfFirst;
using lz::kNone;
using lz::MfEnt;
using lz::MfInfo;

enum { KEY2 = 2, KEY3 = 3, KEY4 = 4, KEY6 = 6, KEY8 = 8 };

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
        } else if (KIND == KEY6) {
            keys[i] = (K)(w & 0xFFFFFFFFFFFFull);
        } else {
            const uint32_t b0 = w & 0xFF, b1 = (w >> 8) & 0xFF, b2 = (w >> 16) & 0xFF,
                           b3 = (w >> 24) & 0xFF;
            uint32_t t = crc[b0] ^ b1;  // lz_bin_tree.go:139-143
            if (KIND == KEY2) {
                keys[i] = (K)((t & (lz::kHash2Size - 1)) | (b0 << 24));  // first byte rides along, unsorted
            } else {
                t ^= b2 << 8;
                if (KIND == KEY3)
                    keys[i] = (K)((t & (lz::kHash3Size - 1)) | (b0 << 24));
                else
                    keys[i] = (K)((t ^ (crc[b3] << 5)) & mask);
            }
        }
    }
}

// head[j] = j if entry j starts a new key group, else 0 (max-scanned into group starts)
template <class K>
__global__ __launch_bounds__(256) void mf_head_kernel(const K *__restrict__ skey, uint32_t n,
                                                      uint32_t *__restrict__ head) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride)
        head[j] = (j > 0 && skey[j - 1] != skey[j]) ? j : 0;
}

// gid[ord[j]] = start of the group entry j belongs to: a number that names the 8-gram of a position
__global__ __launch_bounds__(256) void mf_gid_kernel(const uint32_t *__restrict__ ord,
                                                     const uint32_t *__restrict__ gstart, uint32_t n,
                                                     uint32_t *__restrict__ gid) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) gid[ord[j]] = gstart[j];
}

// Input of the sort that doubles the group length from k to 2k bytes (k = 8: the 16-gram
// level from the 8-gram order, and so on).  ordk holds the positions ordered by (bytes 0..k-1,
// position); position p - k has p's bytes 0..k-1 as its bytes k..2k-1, so the sequence p - k is
// ordered by (bytes k..2k-1, position), and a stable sort of it by the k-gram group of bytes
// 0..k-1 orders it by the first 2k bytes.  The k positions at the end of the text, which have
// no position k ahead, take the places of p = 0..k-1 with keys of their own: no search looks
// at them (a search at level 2k needs 2k bytes after its position and only meets older ones).
__global__ __launch_bounds__(256) void mf_shift_kernel(const uint32_t *__restrict__ ordk,
                                                       const uint32_t *__restrict__ gidk, uint32_t n, uint32_t k,
                                                       uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const uint32_t p = ordk[j];
        if (p >= k) {
            keys[j] = gidk[p - k];
            vals[j] = p - k;
        } else {
            keys[j] = n + p;
            vals[j] = n - k + p;
        }
    }
}

// Keys that tell the 2k-gram groups apart in the sorted order: (k-gram group of the position,
// k-gram group of the position + k), the latter replaced by a number of its own for the last
// k positions of the text.
__global__ __launch_bounds__(256) void mf_keydbl_kernel(const uint32_t *__restrict__ ord,
                                                        const uint32_t *__restrict__ gidk, uint32_t n, uint32_t k,
                                                        uint64_t *__restrict__ keys) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const uint32_t i = ord[j];
        const uint32_t lo = i + k < n ? gidk[i + k] : (0x80000000u | i);
        keys[j] = ((uint64_t)gidk[i] << 32) | lo;
    }
}

// ent[j] = {ord[j] (flagged if it is the oldest of its group), 8 bytes at data[ord[j] + sig_off]};
// rank of ord[j] = j, flagged the same way
template <class K>
__global__ __launch_bounds__(256) void mf_ent_kernel(const uint8_t *__restrict__ data, uint32_t n,
                                                     const uint32_t *__restrict__ ord,
                                                     const K *__restrict__ skey, uint32_t sig_off,
                                                     MfEnt *__restrict__ ent, uint32_t *__restrict__ info,
                                                     uint32_t info_words, uint32_t level, uint32_t wide) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const uint32_t p = ord[j];
        const K key = skey[j];
        // older members of the group, as far as the rank word tells them apart (the keys are
        // sorted: an equal key k places down means k equal keys in between)
        uint32_t older = 0;
        if (j >= 1 && skey[j - 1] == key) older = (j >= 3 && skey[j - 3] == key) ? ((j >= 5 && skey[j - 5] == key) ? 5u : 3u) : 1u;
        const uint32_t first = older == 0 ? kMfFirst : 0u;
        const uint64_t sig = lz::mf_load8(data, n, p + sig_off);
        MfEnt e;
        e.pos = p | first;
        e.sig_lo = (uint32_t)sig, e.sig_hi = (uint32_t)(sig >> 32);
        ent[j] = e;
        info[(size_t)p * info_words + level] = lz::mf_rank_word(j, older, wide != 0);
    }
}

// prev of ord[j] = ord[j-1] if it is in the same hash slot AND starts with the same byte, else
// kNone.  The reference looks at the slot's most recent position only and reports it if the
// first bytes agree (lz_bin_tree.go:154-172), so a head whose first byte differs is the same
// as no head; deciding that here (the first byte rides in bits 24..31 of the key, outside the
// sorted bits) saves the scan two random one-byte reads per position.
__global__ __launch_bounds__(256) void mf_prev_kernel(const uint32_t *__restrict__ ord,
                                                      const uint32_t *__restrict__ skey, uint32_t n,
                                                      uint32_t *__restrict__ info, uint32_t info_words,
                                                      uint32_t word) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const uint32_t v = (j > 0 && skey[j - 1] == skey[j]) ? ord[j - 1] : kNone;
        info[(size_t)ord[j] * info_words + word] = v;
    }
}

unsigned grid_for(uint64_t n, unsigned block, unsigned max_blocks = 256 * 32) {
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    return (unsigned)(g < max_blocks ? g : max_blocks);
}

// Sort configurations.  The library's default for large inputs, the single-pass ("onesweep")
// radix sort, is the fastest sort here when the finder has the device to itself
// (MfScratch::alone), but it makes next to no progress while an encode kernel of another
// stream is resident, even a single launch (its workgroups spin on the progress of other
// workgroups), which would serialize the finder/encoder pipeline.  Its merge-sort path runs
// next to the encoder unhindered but is three times slower.  Batches sorted beside encode
// launches therefore use the finder's own radix sort further down (the merge path stays
// selectable for comparison).
using MergeCfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                            rocprim::default_config, ~(size_t)0>;
using AloneCfg = rocprim::default_config;

template <class Cfg, class K>
size_t sort_temp_bytes(uint32_t n, unsigned end_bit) {
    rocprim::counting_iterator<uint32_t> iota(0);
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs<Cfg>(nullptr, bytes, (K *)nullptr, (K *)nullptr, iota, (uint32_t *)nullptr, n,
                                         0, end_bit, (hipStream_t) nullptr);
    return bytes;
}

size_t scan_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, n,
                                  rocprim::maximum<uint32_t>(), (hipStream_t) nullptr);
    return bytes;
}

template <class Cfg, class K, class V>
int sort_pairs_cfg(lzh::MfScratch &s, K *keys_in, K *keys_out, V vals_in, uint32_t *vals_out, uint32_t n,
                   unsigned end_bit, hipStream_t st) {
    size_t bytes = 0;
    LZ_HIP(rocprim::radix_sort_pairs<Cfg>(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit, st));
    if (bytes > s.temp_bytes) return lzh::fail(LZMA_AMD_ERR_NOMEM, "match finder: sort scratch underestimated");
    bytes = s.temp_bytes;
    LZ_HIP(rocprim::radix_sort_pairs<Cfg>(s.temp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit, st));
    return LZMA_AMD_OK;
}

// ---- the finder's own radix sort, for batches sorted while encode launches are resident.
// Least significant digit first, 8 bits per pass, three kernels per pass and no dependence
// between workgroups inside a kernel (nothing spins on another workgroup's progress, which is
// what the library's single-pass sort does and what stalls it beside long-running waves):
//   1. every workgroup counts the digits of its tile of kRsTile consecutive items;
//   2. an exclusive scan over the counts in (digit, tile) order gives every tile the place
//      where its items of each digit start;
//   3. every workgroup sorts its tile by the digit inside LDS -- each wave takes a quarter of
//      the tile 64 items at a time, the lanes holding the same digit find each other with
//      eight ballots and take consecutive ranks behind the wave's running count for that
//      digit; the waves' counts are then stacked in wave order -- and writes the sorted tile
//      out, so that the items of one digit go to consecutive addresses in one piece.
// Stable, so the passes compose.
constexpr uint32_t kRsTile = 2048, kRsThreads = 256, kRsPerThread = kRsTile / kRsThreads;

template <class K>
__global__ __launch_bounds__(kRsThreads) void rs_count_kernel(const K *__restrict__ keys, uint32_t n, uint32_t shift,
                                                              uint32_t *__restrict__ counts, uint32_t ntiles) {
    __shared__ uint32_t h[256];
    const uint32_t tid = threadIdx.x, tile = blockIdx.x;
    h[tid] = 0;
    __syncthreads();
    const uint32_t base = tile * kRsTile;
#pragma unroll
    for (uint32_t s = 0; s < kRsPerThread; s++) {
        const uint32_t i = base + s * kRsThreads + tid;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    counts[(size_t)tid * ntiles + tile] = h[tid];
}

template <class K>
__global__ __launch_bounds__(kRsThreads) void rs_scatter_kernel(const K *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                                K *__restrict__ kout, uint32_t *__restrict__ vout,
                                                                uint32_t n, uint32_t shift,
                                                                const uint32_t *__restrict__ starts, uint32_t ntiles) {
    constexpr uint32_t kWaves = kRsThreads / 64, kWaveItems = kRsTile / kWaves;
    __shared__ K sk[kRsTile];
    __shared__ uint32_t sv[kRsTile];
    __shared__ uint32_t wcount[kWaves][256];  // per wave: items of each digit, then where they start in the tile
    __shared__ uint32_t dstart[256], gstart[256], wsum[kWaves];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6, tile = blockIdx.x;
    const uint32_t base = tile * kRsTile;
    const uint32_t cnt = n - base < kRsTile ? n - base : kRsTile;
#pragma unroll
    for (uint32_t i = 0; i < kWaves; i++) wcount[i][tid] = 0;
    gstart[tid] = starts[(size_t)tid * ntiles + tile];
    // the wave's items, 64 consecutive ones per step: all requested at once
    K k[kRsPerThread];
    uint32_t v[kRsPerThread], r[kRsPerThread];
#pragma unroll
    for (uint32_t s = 0; s < kRsPerThread; s++) {
        const uint32_t j = w * kWaveItems + s * 64 + lane;  // index in the tile
        k[s] = (K)0, v[s] = 0;
        if (j < cnt) {
            k[s] = kin[base + j];
            v[s] = vin ? vin[base + j] : base + j;
        }
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (uint32_t s = 0; s < kRsPerThread; s++) {
        const bool valid = w * kWaveItems + s * 64 + lane < cnt;
        const uint32_t d = (uint32_t)(k[s] >> shift) & 255u;
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (uint32_t b = 0; b < 8; b++) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? m : ~m;
        }
        r[s] = wcount[w][d] + (uint32_t)__builtin_popcountll(same & below);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // the lanes' reads come before the update
        if (valid && (same >> lane) == 1ull) wcount[w][d] += (uint32_t)__builtin_popcountll(same);  // the group's last lane
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
    __syncthreads();
    // thread d: the waves' counts of digit d stacked in wave order, then the digits in order
    uint32_t total = 0;
#pragma unroll
    for (uint32_t i = 0; i < kWaves; i++) {
        const uint32_t c = wcount[i][tid];
        wcount[i][tid] = total;
        total += c;
    }
    uint32_t inc = total;  // inclusive scan over the wave's 64 digits
#pragma unroll
    for (uint32_t o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)inc, (int)o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t before = 0;
#pragma unroll
    for (uint32_t i = 0; i < kWaves; i++)
        if (i < w) before += wsum[i];
    dstart[tid] = before + inc - total;
    __syncthreads();
#pragma unroll
    for (uint32_t s = 0; s < kRsPerThread; s++) {
        if (w * kWaveItems + s * 64 + lane < cnt) {
            const uint32_t d = (uint32_t)(k[s] >> shift) & 255u;
            const uint32_t at = dstart[d] + wcount[w][d] + r[s];
            sk[at] = k[s];
            sv[at] = v[s];
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < cnt; i += kRsThreads) {
        const K key = sk[i];
        const uint32_t d = (uint32_t)(key >> shift) & 255u;
        const uint32_t at = gstart[d] + (i - dstart[d]);
        kout[at] = key;
        vout[at] = sv[i];
    }
}

size_t rs_scan_temp_bytes(uint32_t ncounts) {
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, ncounts,
                                  rocprim::plus<uint32_t>(), (hipStream_t) nullptr);
    return bytes;
}
// bytes of temporaries own_sort_pairs needs for n items
size_t rs_temp_bytes(uint32_t n) {
    const uint32_t ntiles = (n + kRsTile - 1) / kRsTile;
    const size_t a = ((size_t)n * 4 + 255) & ~(size_t)255, b = ((size_t)ntiles * 256 * 4 + 255) & ~(size_t)255;
    return a + b + rs_scan_temp_bytes(ntiles * 256) + 256;
}

// keys_in is used as the second key buffer and does not survive; vals_in == nullptr stands
// for the values 0..n-1.
template <class K>
int own_sort_pairs(lzh::MfScratch &s, K *keys_in, K *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                   uint32_t n, unsigned end_bit, hipStream_t st) {
    if (rs_temp_bytes(n) > s.temp_bytes) return lzh::fail(LZMA_AMD_ERR_NOMEM, "match finder: sort scratch underestimated");
    const uint32_t ntiles = (n + kRsTile - 1) / kRsTile, ncounts = ntiles * 256;
    uint8_t *q = (uint8_t *)s.temp;
    uint32_t *vtmp = (uint32_t *)q;
    q += ((size_t)n * 4 + 255) & ~(size_t)255;
    uint32_t *counts = (uint32_t *)q;
    q += ((size_t)ncounts * 4 + 255) & ~(size_t)255;
    size_t scan_bytes = s.temp_bytes - (size_t)(q - (uint8_t *)s.temp);
    const unsigned passes = (end_bit + 7) / 8;
    K *kc = keys_in, *kn = keys_out;
    const uint32_t *vc = vals_in;
    for (unsigned p = 0; p < passes; p++) {
        // the last pass writes vals_out, the one before it the temporary, and so on back
        uint32_t *vn = ((passes - 1 - p) & 1u) ? vtmp : vals_out;
        hipLaunchKernelGGL(rs_count_kernel<K>, dim3(ntiles), dim3(kRsThreads), 0, st, (const K *)kc, n, 8 * p, counts, ntiles);
        size_t bytes = scan_bytes;
        LZ_HIP(rocprim::exclusive_scan(q, bytes, counts, counts, 0u, ncounts, rocprim::plus<uint32_t>(), st));
        hipLaunchKernelGGL(rs_scatter_kernel<K>, dim3(ntiles), dim3(kRsThreads), 0, st, (const K *)kc, vc, kn, vn, n, 8 * p,
                           (const uint32_t *)counts, ntiles);
        std::swap(kc, kn);
        vc = vn;
    }
    // an even number of passes leaves the keys in keys_in
    if (kc != keys_out) LZ_HIP(hipMemcpyAsync(keys_out, kc, (size_t)n * sizeof(K), hipMemcpyDeviceToDevice, st));
    LZ_HIP(hipGetLastError());
    return LZMA_AMD_OK;
}

// LZMA_AMD_SORT=merge: the library's merge sort instead of the finder's own radix sort beside
// resident encode launches (for comparison)
bool use_own_sort() {
    static const bool own = [] {
        const char *e = getenv("LZMA_AMD_SORT");
        return !(e && strcmp(e, "merge") == 0);
    }();
    return own;
}
// LZMA_AMD_SORT=own: the finder's own sort also when it has the device to itself
bool own_sort_always() {
    static const bool v = [] {
        const char *e = getenv("LZMA_AMD_SORT");
        return e && strcmp(e, "own") == 0;
    }();
    return v;
}

// values out = the positions 0..n-1 ordered by key
template <class K>
int sort_by_key(lzh::MfScratch &s, K *keys_in, K *keys_out, uint32_t *vals_out, uint32_t n,
                unsigned end_bit, hipStream_t st) {
    rocprim::counting_iterator<uint32_t> iota(0);
    if (s.alone && !own_sort_always()) return sort_pairs_cfg<AloneCfg>(s, keys_in, keys_out, iota, vals_out, n, end_bit, st);
    if (use_own_sort()) return own_sort_pairs<K>(s, keys_in, keys_out, nullptr, vals_out, n, end_bit, st);
    return sort_pairs_cfg<MergeCfg>(s, keys_in, keys_out, iota, vals_out, n, end_bit, st);
}
// values out = vals_in ordered by key (stable)
int sort_vals_by_key(lzh::MfScratch &s, uint32_t *keys_in, uint32_t *keys_out, uint32_t *vals_in,
                     uint32_t *vals_out, uint32_t n, unsigned end_bit, hipStream_t st) {
    if (s.alone && !own_sort_always()) return sort_pairs_cfg<AloneCfg>(s, keys_in, keys_out, vals_in, vals_out, n, end_bit, st);
    if (use_own_sort()) return own_sort_pairs<uint32_t>(s, keys_in, keys_out, vals_in, vals_out, n, end_bit, st);
    return sort_pairs_cfg<MergeCfg>(s, keys_in, keys_out, vals_in, vals_out, n, end_bit, st);
}

int max_scan(lzh::MfScratch &s, uint32_t *io, uint32_t n, hipStream_t st) {
    size_t bytes = 0;
    LZ_HIP(rocprim::inclusive_scan(nullptr, bytes, io, io, n, rocprim::maximum<uint32_t>(), st));
    if (bytes > s.temp_bytes) return lzh::fail(LZMA_AMD_ERR_NOMEM, "match finder: scan scratch underestimated");
    bytes = s.temp_bytes;
    LZ_HIP(rocprim::inclusive_scan(s.temp, bytes, io, io, n, rocprim::maximum<uint32_t>(), st));
    return LZMA_AMD_OK;
}

unsigned bits_of(uint64_t mask) {
    unsigned b = 0;
    while (b < 64 && (mask >> b) != 0) b++;
    return b ? b : 1;
}

constexpr uint32_t kRun = 256;     // consecutive positions handled by one wave before it moves on
constexpr uint32_t kRowRun = 64;   // ... of which each 16-lane row takes a quarter

// ctrl[0] = next free arena pair index, ctrl[1] = overflow flag, ctrl[2] = pairs emitted
//
// Each row of a wave walks its own run of consecutive positions and appends their pairs to
// a private page of the arena (one atomic per page); the result word of a position says
// where its pairs are.  The walk is software-pipelined: while position k is worked on, the
// newest table entries of position k + 1 and the record and text of position k + 2 are on
// their way, so a position's two dependent memory round trips overlap the work on the two
// before it.
#ifndef MF_SCAN_WAVES
#define MF_SCAN_WAVES 4  // 112 registers, no spills: faster than more waves that spill
#endif
// NL: kMfLevels, or kMfMaxLevels when the deep levels were built (more registers: fewer waves);
// TINY: a text of fewer than 32 bytes (see mf_find4)
template <uint32_t NL, bool TINY>
__global__ __launch_bounds__(64, NL == lz::kMfLevels ? MF_SCAN_WAVES : 2) void mf_scan_kernel(lz::MfTables t, const uint64_t *__restrict__ job_off,
                                                      uint64_t base, uint64_t j0, uint32_t njobs,
                                                      uint64_t *__restrict__ meta,
                                                      uint32_t *__restrict__ arena, uint64_t arena_pairs,
                                                      unsigned long long *ctrl) {
    const uint32_t lane = threadIdx.x, idx = lane & 15, row = lane >> 4, row_base = lane & 48;
    // one wave per workgroup: next to resident encode waves a SIMD often has room for one more
    // wave but not all four SIMDs of a CU at once, which a larger workgroup would need
    const uint32_t wave = blockIdx.x, nwaves = gridDim.x;
    const uint32_t n = t.n;
    const uint32_t nruns = (n + kRun - 1) / kRun;
    uint64_t cursor = 0, page_end = 0;  // the row's arena page
    uint64_t emitted = 0;
    bool overflow = false;
#if defined(MF_PROFILE)
    lz::MfProf pf = {};
    pf.last = __builtin_readcyclecounter();
#endif
    for (uint32_t run = wave; run < nruns; run += nwaves) {
        const uint32_t p_begin = run * kRun + row * kRowRun;
        uint32_t p_end = p_begin + kRowRun;
        if (p_end > n) p_end = n;
        const bool row_on = p_begin < n;
        // job containing p_begin: last j with job_off[j] - base <= p_begin
        uint32_t job = 0, js = 0, je = 0;
        if (row_on) {
            uint32_t lo = 0, hi = njobs;  // invariant: off[lo] <= p_begin < off[hi]
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (job_off[j0 + mid] - base <= p_begin) lo = mid; else hi = mid;
            }
            job = lo;
            js = (uint32_t)(job_off[j0 + job] - base), je = (uint32_t)(job_off[j0 + job + 1] - base);
        }
        // pipeline prologue: positions 0 and 1 of the run
        lz::MfPos<NL> a_cur, a_next;
        lz::MfFront<NL> f_cur;
        lz::mf_load_pos(t, p_begin, row_on && p_begin < p_end, a_cur);
        lz::mf_load_front(t, a_cur, row_on && p_begin < p_end, idx, f_cur);
        lz::mf_load_pos(t, p_begin + 1, row_on && p_begin + 1 < p_end, a_next);
        for (uint32_t k = 0; k < kRowRun; k++) {
            const uint32_t P = p_begin + k;
            const bool act = row_on && P < p_end;
            if (!__any(act)) break;
            MF_LAP(4)
            // entries of position k + 1 (its record has had a whole step to arrive), then the
            // record of position k + 2
            lz::MfFront<NL> f_next;
            lz::mf_load_front(t, a_next, row_on && P + 1 < p_end, idx, f_next);
            lz::MfPos<NL> a_next2;
            lz::mf_load_pos(t, P + 2, row_on && P + 2 < p_end && k + 2 < kRowRun, a_next2);
            if (act)
                while (P >= je) {  // empty jobs are skipped
                    job++;
                    js = je;
                    je = (uint32_t)(job_off[j0 + job + 1] - base);
                }
            // a fresh page if this one cannot take a full list
            const bool need = act && !overflow && page_end - cursor < lz::kMatchMax + 1;
            if (__any(need)) {
                unsigned long long b = 0;
                if (need && idx == 0) b = atomicAdd(&ctrl[0], (unsigned long long)lzh::kMfPagePairs);
                const uint32_t b_lo = lz::row_first((uint32_t)b, row_base), b_hi = lz::row_first((uint32_t)(b >> 32), row_base);
                b = ((unsigned long long)b_hi << 32) | b_lo;
                if (need) {
                    if (b + lzh::kMfPagePairs > arena_pairs) {
                        overflow = true;
                    } else {
                        cursor = b;
                        page_end = b + lzh::kMfPagePairs;
                    }
                }
            }
            MF_LAP(0)
            lz::MfRowOut out;
            out.packed = t.packed != 0;
            out.dst = arena + (out.packed ? cursor : 2 * cursor);
            out.on = act && !overflow;
            const uint32_t c = lz::mf_find4<NL, TINY>(t, P, a_cur, f_cur, js, je, act, out, lane MF_PROF_PASS);
            if (act) {
                if (idx == 0) meta[P] = lz::mf_meta(cursor, c);
                if (!overflow) cursor += c;
                emitted += c;
            }
            a_cur = a_next;
            f_cur = f_next;
            a_next = a_next2;
        }
    }
    if (idx == 0) {
        if (overflow) atomicAdd(&ctrl[1], 1ull);
        atomicAdd(&ctrl[2], (unsigned long long)emitted);
    }
#if defined(MF_PROFILE)
    if (wave == 7 && lane == 0)
        printf("MFPROF wave 7: issue-loads+page %llu | find setup %llu | first steps %llu | deep steps %llu | loop+meta %llu\n",
               (unsigned long long)pf.t[0], (unsigned long long)pf.t[1], (unsigned long long)pf.t[2],
               (unsigned long long)pf.t[3], (unsigned long long)pf.t[4]);
#endif
}

// Stand-alone entry point: pairs of position P to out[offsets[P]...] in position order.
__global__ __launch_bounds__(256) void mf_gather_kernel(const uint64_t *__restrict__ meta,
                                                        const uint32_t *__restrict__ arena, uint32_t n,
                                                        const uint64_t *__restrict__ offsets,
                                                        uint32_t *__restrict__ pairs, uint64_t pairs_cap,
                                                        uint32_t packed) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t P = blockIdx.x * blockDim.x + threadIdx.x; P < n; P += stride) {
        const uint64_t m = meta[P], src = lz::mf_meta_off(m), dst = offsets[P];
        const uint32_t c = lz::mf_meta_count(m);
        if (dst + c > pairs_cap) continue;
        if (packed) {
            for (uint32_t k = 0; k < c; k++) {
                const uint32_t w = arena[src + k];
                pairs[2 * (dst + k)] = lz::mf_pack_len(w);
                pairs[2 * (dst + k) + 1] = lz::mf_pack_dist(w);
            }
        } else {
            for (uint32_t k = 0; k < 2 * c; k++) pairs[2 * dst + k] = arena[2 * src + k];
        }
    }
}

__global__ __launch_bounds__(256) void mf_counts_kernel(const uint64_t *__restrict__ meta, uint32_t n,
                                                        uint32_t *__restrict__ counts) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t P = blockIdx.x * blockDim.x + threadIdx.x; P < n; P += stride)
        counts[P] = lz::mf_meta_count(meta[P]);
}

}  // namespace

namespace lzh {

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

template <class Cfg>
static size_t lib_temp_bytes_cfg(uint32_t n) {
    size_t t = 0;
    const unsigned bits32[] = {10, 16, 23, 32};
    for (unsigned b : bits32) t = std::max(t, sort_temp_bytes<Cfg, uint32_t>(n, b));
    t = std::max(t, sort_temp_bytes<Cfg, uint64_t>(n, 64));
    {
        size_t b = 0;
        (void)rocprim::radix_sort_pairs<Cfg>(nullptr, b, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                             (uint32_t *)nullptr, (uint32_t *)nullptr, n, 0, 32,
                                             (hipStream_t) nullptr);
        t = std::max(t, b);
    }
    {
        size_t b = 0;
        (void)rocprim::radix_sort_pairs<Cfg>(nullptr, b, (uint64_t *)nullptr, (uint64_t *)nullptr,
                                             (uint32_t *)nullptr, (uint32_t *)nullptr, n, 0, 64,
                                             (hipStream_t) nullptr);
        t = std::max(t, b);
    }
    return t;
}
static size_t lib_temp_bytes(uint32_t n) {  // largest request of any sort or scan of n items
    size_t t = std::max(lib_temp_bytes_cfg<MergeCfg>(n), lib_temp_bytes_cfg<AloneCfg>(n));
    t = std::max(t, scan_temp_bytes(n));
    t = std::max(t, rs_temp_bytes(n));
    return t + 256;
}

size_t MfScratch::tmp_bytes(uint32_t n) {
    const size_t n1 = n ? n : 1;
    return 2 * align256(n1 * 8) + 3 * align256(n1 * 4) + align256(lib_temp_bytes((uint32_t)n1));
}

int MfScratch::reserve(uint32_t n, uint32_t fb) {
    const uint32_t levels = lz::mf_levels_for(fb, n);
    if (n <= cap && cap != 0 && levels <= cap_levels) return LZMA_AMD_OK;
    const size_t n1 = std::max<size_t>(n ? n : 1, cap);
    const uint32_t lv = std::max(levels, cap_levels);
    for (uint32_t l = 0; l < lv; l++)
        if (ent[l].bytes < n1 * sizeof(MfEnt)) LZ_HIP(ent[l].alloc(n1 * sizeof(MfEnt)));
    // records sized for the deep form once any build needed it
    const size_t info_bytes = n1 * lz::mf_info_words(lv) * 4;
    if (info.bytes < info_bytes) LZ_HIP(info.alloc(info_bytes));
    cap = (uint32_t)n1;
    cap_levels = lv;
    return LZMA_AMD_OK;
}

int MfScratch::build(const uint8_t *d_data, uint32_t n, uint32_t dict_size, uint32_t fb, hipStream_t st,
                     void *tmp, size_t tmp_cap) {
    int rc = reserve(n, fb);
    if (rc) return rc;
    {   // lay out the temporaries
        const size_t n1 = n ? n : 1, need = tmp_bytes(n);
        if (!tmp || tmp_cap < need) {
            if (own_tmp.bytes < need) LZ_HIP(own_tmp.alloc(need));
            tmp = own_tmp.p;
        }
        uint8_t *q = (uint8_t *)tmp;
        kin = q, q += align256(n1 * 8);   // sized for the 64-bit keys
        kout = q, q += align256(n1 * 8);
        vout = q, q += align256(n1 * 4);
        grp = q, q += align256(n1 * 4);
        gid = q, q += align256(n1 * 4);
        temp = q;
        temp_bytes = need - (size_t)(q - (uint8_t *)tmp);
    }
    const uint32_t mask = lz::bt4_hash_mask(dict_size);
    view.data = d_data;
    view.n = n;
    for (uint32_t l = 0; l < lz::kMfMaxLevels; l++) view.ent[l] = ent[l].as<MfEnt>();
    view.info = info.p;
    view.dict = dict_size;
    view.fb = fb;
    view.levels = lz::mf_levels_for(fb, n);
    view.packed = lz::mf_can_pack(dict_size) ? 1u : 0u;
    view.wide = n > lz::kMfClassMax ? 1u : 0u;
    if (n == 0) return LZMA_AMD_OK;

    const unsigned g = grid_for(n, 256);
    uint32_t *inf = info.as<uint32_t>();
    const uint32_t iw = lz::mf_info_words(view.levels), ip = lz::mf_info_prev(view.levels);
    uint32_t *ord = (uint32_t *)vout;
    // level 0: hash-4 buckets
    hipLaunchKernelGGL((mf_key_kernel<KEY4, uint32_t>), dim3(g), dim3(256), 0, st, d_data, n, mask,
                       (uint32_t *)kin);
    if ((rc = sort_by_key(*this, (uint32_t *)kin, (uint32_t *)kout, ord, n, bits_of(mask), st))) return rc;
    hipLaunchKernelGGL(mf_ent_kernel<uint32_t>, dim3(g), dim3(256), 0, st, d_data, n, ord,
                       (const uint32_t *)kout, 0u, ent[0].as<MfEnt>(), inf, iw, 0u, view.wide);
    // level 1: 6-gram groups
    if (view.levels > 1) {
        hipLaunchKernelGGL((mf_key_kernel<KEY6, uint64_t>), dim3(g), dim3(256), 0, st, d_data, n, 0u,
                           (uint64_t *)kin);
        if ((rc = sort_by_key(*this, (uint64_t *)kin, (uint64_t *)kout, ord, n, 48, st))) return rc;
        hipLaunchKernelGGL(mf_ent_kernel<uint64_t>, dim3(g), dim3(256), 0, st, d_data, n, ord,
                           (const uint64_t *)kout, 6u, ent[1].as<MfEnt>(), inf, iw, 1u, view.wide);
    }
    // level 2: 8-gram groups
    if (view.levels > 2) {
        hipLaunchKernelGGL((mf_key_kernel<KEY8, uint64_t>), dim3(g), dim3(256), 0, st, d_data, n, 0u,
                           (uint64_t *)kin);
        if ((rc = sort_by_key(*this, (uint64_t *)kin, (uint64_t *)kout, ord, n, 64, st))) return rc;
        hipLaunchKernelGGL(mf_ent_kernel<uint64_t>, dim3(g), dim3(256), 0, st, d_data, n, ord,
                           (const uint64_t *)kout, 8u, ent[2].as<MfEnt>(), inf, iw, 2u, view.wide);
    }
    // levels 3..: groups of 16, 32, 64, 128 bytes, each from the order of the level before it
    // (see mf_shift_kernel).  `ord` is that order and `sorted` its keys in sorted order.
    const uint64_t *sorted = (const uint64_t *)kout;
    for (uint32_t l = 3; l < view.levels; l++) {
        const uint32_t k = lz::mf_known(l) / 2;
        hipLaunchKernelGGL(mf_head_kernel<uint64_t>, dim3(g), dim3(256), 0, st, sorted, n, (uint32_t *)grp);
        if ((rc = max_scan(*this, (uint32_t *)grp, n, st))) return rc;
        hipLaunchKernelGGL(mf_gid_kernel, dim3(g), dim3(256), 0, st, ord, (const uint32_t *)grp, n, (uint32_t *)gid);
        // keys and values of the sort share the key-input area (32-bit keys leave half of it free)
        uint32_t *k32 = (uint32_t *)kin, *v32 = (uint32_t *)kin + n;
        hipLaunchKernelGGL(mf_shift_kernel, dim3(g), dim3(256), 0, st, ord, (const uint32_t *)gid, n, k, k32, v32);
        // the sorted keys name the 2k-gram groups only together with the bytes k..2k-1 order, so the
        // group boundaries are found from the pair (key, k-gram group of the position + k) below
        if ((rc = sort_vals_by_key(*this, k32, (uint32_t *)kout, v32, ord, n, bits_of((uint64_t)n + k), st)))
            return rc;
        hipLaunchKernelGGL(mf_keydbl_kernel, dim3(g), dim3(256), 0, st, ord, (const uint32_t *)gid, n, k,
                           (uint64_t *)kin);
        hipLaunchKernelGGL(mf_ent_kernel<uint64_t>, dim3(g), dim3(256), 0, st, d_data, n, ord,
                           (const uint64_t *)kin, lz::mf_known(l), ent[l].as<MfEnt>(), inf, iw, l, view.wide);
        sorted = (const uint64_t *)kin;
    }
    // hash-3 and hash-2 heads
    hipLaunchKernelGGL((mf_key_kernel<KEY3, uint32_t>), dim3(g), dim3(256), 0, st, d_data, n, 0u,
                       (uint32_t *)kin);
    if ((rc = sort_by_key(*this, (uint32_t *)kin, (uint32_t *)kout, ord, n, 16, st))) return rc;
    hipLaunchKernelGGL(mf_prev_kernel, dim3(g), dim3(256), 0, st, ord, (uint32_t *)kout, n, inf, iw, ip + 1);
    hipLaunchKernelGGL((mf_key_kernel<KEY2, uint32_t>), dim3(g), dim3(256), 0, st, d_data, n, 0u,
                       (uint32_t *)kin);
    if ((rc = sort_by_key(*this, (uint32_t *)kin, (uint32_t *)kout, ord, n, 10, st))) return rc;
    hipLaunchKernelGGL(mf_prev_kernel, dim3(g), dim3(256), 0, st, ord, (uint32_t *)kout, n, inf, iw, ip);
    LZ_HIP(hipGetLastError());
    return LZMA_AMD_OK;
}

static unsigned scan_blocks(uint32_t n) {
    const uint32_t nruns = (n + kRun - 1) / kRun;
    unsigned blocks = (nruns + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    return blocks ? blocks : 1;
}

int MfResult::reserve(uint32_t n, uint32_t pair_bytes_) {
    if (n <= cap && cap != 0 && pair_bytes_ <= pair_bytes) return LZMA_AMD_OK;
    pair_bytes = pair_bytes_;
    const size_t n1 = n ? n : 1;
    LZ_HIP(meta.alloc(n1 * 8));
    LZ_HIP(ctrl.alloc(3 * 8));
    // first estimate: 5 pairs per position (text has about 4.5) plus one page per row; the scan
    // reports the exact need if this is too small.  LZMA_AMD_ARENA_TEST=1 starts from a
    // deliberately small arena (pages for half of the rows) so that tests reach the grow-and-retry
    // path, and reports each growth on stderr.
    const char *tiny = getenv("LZMA_AMD_ARENA_TEST");
    const uint64_t rows = (uint64_t)scan_blocks((uint32_t)n1) * 16;
    const uint64_t want = (tiny && tiny[0] == '1') ? rows * kMfPagePairs / 2 : (uint64_t)n1 * 5 + rows * kMfPagePairs;
    LZ_HIP(arena.alloc(want * pair_bytes));
    arena_pairs = want;
    cap = (uint32_t)n1;
    return LZMA_AMD_OK;
}

int mf_scan(const MfScratch &s, MfResult &r, const uint64_t *d_job_off, uint64_t base, uint64_t j0,
            uint32_t njobs, hipStream_t st, uint64_t *total_pairs) {
    const uint32_t n = s.view.n;
    if (total_pairs) *total_pairs = 0;
    if (n == 0 || njobs == 0) {
        LZ_HIP(hipStreamSynchronize(st));
        return LZMA_AMD_OK;
    }
    const uint32_t pair_bytes = s.view.packed ? 4u : 8u;
    int rc = r.reserve(n, pair_bytes);
    if (rc) return rc;
    // an arena kept from a call with one-word pairs holds half as many two-word pairs
    r.arena_pairs = r.arena.bytes / pair_bytes;
    const unsigned blocks = scan_blocks(n);
    const uint64_t nrows = (uint64_t)blocks * 16;  // 16-lane rows, each with an arena page of its own
    for (int attempt = 0; attempt < 8; attempt++) {
        LZ_HIP(hipMemsetAsync(r.ctrl.p, 0, 3 * 8, st));
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(blocks * 4), dim3(64), 0, st, s.view, d_job_off, base, j0, njobs,
                               r.meta.as<uint64_t>(), r.arena.as<uint32_t>(), r.arena_pairs,
                               r.ctrl.as<unsigned long long>());
        };
        if (s.view.levels > lz::kMfLevels)
            launch(mf_scan_kernel<lz::kMfMaxLevels, false>);
        else if (n < 32)
            launch(mf_scan_kernel<lz::kMfLevels, true>);
        else
            launch(mf_scan_kernel<lz::kMfLevels, false>);
        LZ_HIP(hipGetLastError());
        unsigned long long h[3];
        LZ_HIP(hipMemcpyAsync(h, r.ctrl.p, sizeof h, hipMemcpyDeviceToHost, st));
        LZ_HIP(hipStreamSynchronize(st));
        if (h[1] == 0) {
            if (total_pairs) *total_pairs = h[2];
            return LZMA_AMD_OK;
        }
        // the emitted count is exact even when the arena overflowed; growing the arena frees
        // the old one, which waits for every kernel on the device
        const uint64_t want = h[2] + h[2] / 8 + nrows * kMfPagePairs;
        if (const char *tiny = getenv("LZMA_AMD_ARENA_TEST"))
            if (tiny[0] == '1') fprintf(stderr, "lzma_amd: arena grown to %llu pairs\n", (unsigned long long)want);
        LZ_HIP(hipDeviceSynchronize());
        LZ_HIP(r.arena.alloc(want * pair_bytes));
        r.arena_pairs = want;
    }
    return fail(LZMA_AMD_ERR_NOMEM, "match finder arena kept overflowing");
}

}  // namespace lzh

namespace {
struct FinderWs {
    int device = -1;
    lzh::MfScratch scratch;
    lzh::MfResult res;
    lzh::DevBuf joff, offs, temp;  // job offsets, pair offsets (if the caller keeps none), scan scratch
};
// never destroyed on its own: at process exit the HIP runtime may already be gone
thread_local FinderWs *g_fws = nullptr;
}  // namespace

void lzh::release_finder_workspace() {
    if (!g_fws || g_fws->device < 0) return;
    (void)hipDeviceSynchronize();
    g_fws->scratch.release();
    g_fws->res.release();
    g_fws->joff.release(), g_fws->offs.release(), g_fws->temp.release();
    g_fws->device = -1;
}

extern "C" int lzma_amd_bt4_matches_dev(const uint8_t *d_data, size_t n, uint32_t dict_size,
                                        uint32_t fb, uint32_t *d_counts, uint64_t *d_offsets,
                                        uint32_t *d_pairs, size_t pairs_cap, uint64_t *d_total_pairs,
                                        void *stream) {
    hipStream_t st = (hipStream_t)stream;
    lzh::tls().ms_total = lzh::tls().ms_finder = lzh::tls().ms_coder = 0.f;
    lzh::tls().coder_launches = 0;
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
    // tables and lists (about 120 bytes per input byte) stay with the calling thread, as the
    // encoder's do: allocating and releasing them costs more than the kernels at these sizes
    int dev = 0;
    LZ_HIP(hipGetDevice(&dev));
    if (!g_fws) g_fws = new FinderWs;
    if (g_fws->device != dev) {
        lzh::release_finder_workspace();
        g_fws->device = dev;
    }
    lzh::MfScratch &ws = g_fws->scratch;
    lzh::MfResult &res = g_fws->res;
    ws.alone = true;
    int rc = ws.build(d_data, N, dict_size, fb, st);
    if (rc) return rc;
    // (everything this call needs beyond the caller's buffers is kept with the tables: no
    // allocation, and so no wait for the device, once a thread has seen its largest stream)
    lzh::DevBuf &joff = g_fws->joff;
    const uint64_t h_joff[2] = {0, n};
    if (!joff.p) LZ_HIP(joff.alloc(16));
    LZ_HIP(hipMemcpyAsync(joff.p, h_joff, 16, hipMemcpyHostToDevice, st));
    uint64_t total = 0;
    t_scan.start(st);
    rc = lzh::mf_scan(ws, res, joff.as<uint64_t>(), 0, 0, 1, st, &total);
    t_scan.stop();
    if (rc) return rc;
    const unsigned g = grid_for(N, 256);
    hipLaunchKernelGGL(mf_counts_kernel, dim3(g), dim3(256), 0, st, res.d_meta(), N, d_counts);
    LZ_HIP(hipMemcpyAsync(d_total_pairs, &total, 8, hipMemcpyHostToDevice, st));
    if ((d_pairs && total <= pairs_cap) || d_offsets) {
        // the offsets go straight into the caller's array if there is one
        uint64_t *offs = d_offsets;
        if (!offs) {
            if (g_fws->offs.bytes < n * 8) LZ_HIP(g_fws->offs.alloc(n * 8));
            offs = g_fws->offs.as<uint64_t>();
        }
        size_t bytes = 0;
        auto in = rocprim::make_transform_iterator(d_counts, [] __device__(uint32_t c) { return (uint64_t)c; });
        LZ_HIP(rocprim::exclusive_scan(nullptr, bytes, in, offs, (uint64_t)0, n, rocprim::plus<uint64_t>(), st));
        lzh::DevBuf &temp = g_fws->temp;
        if (temp.bytes < bytes) LZ_HIP(temp.alloc(bytes));
        bytes = temp.bytes;
        LZ_HIP(rocprim::exclusive_scan(temp.p, bytes, in, offs, (uint64_t)0, n, rocprim::plus<uint64_t>(), st));
        if (d_pairs && total <= pairs_cap)
            hipLaunchKernelGGL(mf_gather_kernel, dim3(g), dim3(256), 0, st, res.d_meta(), res.d_arena(), N,
                               offs, d_pairs, (uint64_t)pairs_cap, ws.view.packed);
    }
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
