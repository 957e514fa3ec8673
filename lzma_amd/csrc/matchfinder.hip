// Exhaustive BT4 match finder for gfx950, position-parallel.
//
// Replaces lzBinTree.getMatches/skip of the reference (lz_bin_tree.go:115-337) run
// at every position of a stream.  Pipeline per batch of positions:
//   1. key kernels: hash-2 / hash-3 / hash-4 slots (lz_bin_tree.go:139-143), the
//      first 8 bytes of every position, and (from the finished 8-gram table) the pair
//      of 8-gram groups that stands for its first 16 bytes;
//   2. stable radix sorts of (key, position) -- rocPRIM device radix sort, the one
//      library primitive used -- giving each bucket's positions in ascending order;
//   3. table kernels: 12-byte entries (position, 8 bytes of its text) in sorted order
//      for the hash-4 buckets, the 8-gram groups and the 16-gram groups, and per
//      position its rank and its group's start in each table and the previous
//      position with the same hash-2 / hash-3 slot;
//   4. the scan kernel (lzma_mf_core.hpp): one wave per position, 64 candidates per
//      step read as consecutive 12-byte entries, record breakers by ballot.  Each
//      wave walks a run of consecutive positions and appends their pairs to a
//      private page of a global arena (one atomic per 8192 pairs); a word per
//      position records where its pairs are.
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

using lz::kNone;
using lz::MfEnt;
using lz::MfInfo;

enum { KEY2 = 2, KEY3 = 3, KEY4 = 4, KEY8 = 8 };

// KEY8: mask is the byte offset of the 8 bytes taken as the key
template <int KIND, class K>
__global__ __launch_bounds__(256) void mf_key_kernel(const uint8_t *__restrict__ data, uint32_t n,
                                                     uint32_t mask, K *__restrict__ keys) {
    __shared__ uint32_t crc[256];
    crc[threadIdx.x] = lz::crc32_entry(threadIdx.x);
    __syncthreads();
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t w = lz::mf_load8(data, n, KIND == KEY8 ? i + mask : i);
        if (KIND == KEY8) {
            keys[i] = (K)w;
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

// 16-gram key of position i: the 8-gram group of i and the 8-gram group of i + 8 (a group's
// start index in the 8-gram table names it), so that one sort orders the positions by their
// first 16 bytes.  The last 8 positions of the text get keys of their own: no search looks at
// them (a search at level 16 needs 16 bytes after its position and only meets older ones).
__global__ __launch_bounds__(256) void mf_key16_kernel(const MfInfo *__restrict__ info, uint32_t n,
                                                       uint64_t *__restrict__ keys) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t hi = info[i].g8;
        const uint32_t lo = i + 8 < n ? info[i + 8].g8 : (0x80000000u | i);
        keys[i] = ((uint64_t)hi << 32) | lo;
    }
}

// ent[j] = {ord[j], 8 bytes at data[ord[j] + sig_off]};  rank of ord[j] = j, its group starts at gstart[j]
template <int WHICH>
__global__ __launch_bounds__(256) void mf_ent_kernel(const uint8_t *__restrict__ data, uint32_t n,
                                                     const uint32_t *__restrict__ ord,
                                                     const uint32_t *__restrict__ gstart,
                                                     uint32_t sig_off, MfEnt *__restrict__ ent,
                                                     MfInfo *__restrict__ info) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const uint32_t p = ord[j];
        const uint64_t sig = lz::mf_load8(data, n, p + sig_off);
        MfEnt e;
        e.pos = p;
        e.sig_lo = (uint32_t)sig, e.sig_hi = (uint32_t)(sig >> 32);
        ent[j] = e;
        const uint32_t g = gstart[j];
        if (WHICH == 4) info[p].rank4 = j, info[p].g4 = g;
        else if (WHICH == 8) info[p].rank8 = j, info[p].g8 = g;
        else info[p].rank16 = j, info[p].g16 = g;
    }
}

// prev of ord[j] = ord[j-1] if it is in the same hash slot AND starts with the same byte, else
// kNone.  The reference looks at the slot's most recent position only and reports it if the
// first bytes agree (lz_bin_tree.go:154-172), so a head whose first byte differs is the same
// as no head; deciding that here (the first byte rides in bits 24..31 of the key, outside the
// sorted bits) saves the scan two random one-byte reads per position.
template <int WHICH>
__global__ __launch_bounds__(256) void mf_prev_kernel(const uint32_t *__restrict__ ord,
                                                      const uint32_t *__restrict__ skey, uint32_t n,
                                                      MfInfo *__restrict__ info) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const uint32_t v = (j > 0 && skey[j - 1] == skey[j]) ? ord[j - 1] : kNone;
        if (WHICH == 2) info[ord[j]].prev2 = v; else info[ord[j]].prev3 = v;
    }
}

unsigned grid_for(uint64_t n, unsigned block, unsigned max_blocks = 256 * 32) {
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    return (unsigned)(g < max_blocks ? g : max_blocks);
}

// Sort configurations.  The library's default for large inputs, the single-pass ("onesweep")
// radix sort, is about three times faster than its merge-sort path here, but makes no progress
// while an encode kernel of another stream is resident (measured: its first pass sits until the
// first encode workgroup exits, about 5 s), which would serialize the finder/encoder pipeline:
// its histogram kernel wants 32 KiB of LDS and its sort kernel 108 registers in each of the four
// waves of a block, and twelve resident encode waves leave a CU 7 KiB and 56 registers per SIMD.
// (A single-wave configuration that fits, 64 threads x 6 items x 6 bits, runs but is slower than
// the merge path.)  The merge path runs next to the encoder unhindered.  So the default is used
// when the finder has the device to itself (MfScratch::alone) and the merge path otherwise.
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

// values out = the positions 0..n-1 ordered by key
template <class K>
int sort_by_key(lzh::MfScratch &s, K *keys_in, K *keys_out, uint32_t *vals_out, uint32_t n,
                unsigned end_bit, hipStream_t st) {
    rocprim::counting_iterator<uint32_t> iota(0);
    return s.alone ? sort_pairs_cfg<AloneCfg>(s, keys_in, keys_out, iota, vals_out, n, end_bit, st)
                   : sort_pairs_cfg<MergeCfg>(s, keys_in, keys_out, iota, vals_out, n, end_bit, st);
}

int max_scan(lzh::MfScratch &s, uint32_t *io, uint32_t n, hipStream_t st) {
    size_t bytes = 0;
    LZ_HIP(rocprim::inclusive_scan(nullptr, bytes, io, io, n, rocprim::maximum<uint32_t>(), st));
    if (bytes > s.temp_bytes) return lzh::fail(LZMA_AMD_ERR_NOMEM, "match finder: scan scratch underestimated");
    bytes = s.temp_bytes;
    LZ_HIP(rocprim::inclusive_scan(s.temp, bytes, io, io, n, rocprim::maximum<uint32_t>(), st));
    return LZMA_AMD_OK;
}

unsigned bits_of(uint32_t mask) {
    unsigned b = 0;
    while (b < 32 && (mask >> b) != 0) b++;
    return b ? b : 1;
}

constexpr uint32_t kRun = 256;     // consecutive positions handled by one wave before it moves on
constexpr uint32_t kRowRun = 64;   // ... of which each 16-lane row takes a quarter

// ctrl[0] = next free arena pair index, ctrl[1] = overflow flag, ctrl[2] = pairs emitted
//
// Each row of a wave walks its own run of consecutive positions and appends their pairs to
// a private page of the arena (one atomic per page); the result word of a position says
// where its pairs are.
__global__ __launch_bounds__(64, 8) void mf_scan_kernel(lz::MfTables t, const uint64_t *__restrict__ job_off,
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
        for (uint32_t k = 0; k < kRowRun; k++) {
            const uint32_t P = p_begin + k;
            const bool act = row_on && P < p_end;
            if (!__any(act)) break;
            lz::MfInfo inf = {};
            if (act) inf = t.info[P];
            uint64_t p0 = 0, p8 = 0, p16 = 0;  // bytes 0..7, 8..15, 16..23 at P
            if (act) {
                p0 = lz::mf_load8(t.data, n, P), p8 = lz::mf_load8(t.data, n, P + 8);
                p16 = lz::mf_load8(t.data, n, P + 16);
            }
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
            lz::MfRowOut out;
            out.dst = arena + 2 * cursor;
            out.on = act && !overflow;
            const uint32_t c = lz::mf_find4(t, P, inf, p0, p8, p16, js, je, act, out, lane);
            if (act) {
                if (idx == 0) meta[P] = lz::mf_meta(cursor, c);
                if (!overflow) cursor += c;
                emitted += c;
            }
        }
    }
    if (idx == 0) {
        if (overflow) atomicAdd(&ctrl[1], 1ull);
        atomicAdd(&ctrl[2], (unsigned long long)emitted);
    }
}

// Stand-alone entry point: pairs of position P to out[offsets[P]...] in position order.
__global__ __launch_bounds__(256) void mf_gather_kernel(const uint64_t *__restrict__ meta,
                                                        const uint32_t *__restrict__ arena, uint32_t n,
                                                        const uint64_t *__restrict__ offsets,
                                                        uint32_t *__restrict__ pairs, uint64_t pairs_cap) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t P = blockIdx.x * blockDim.x + threadIdx.x; P < n; P += stride) {
        const uint64_t m = meta[P], src = lz::mf_meta_off(m), dst = offsets[P];
        const uint32_t c = lz::mf_meta_count(m);
        if (dst + c > pairs_cap) continue;
        for (uint32_t k = 0; k < 2 * c; k++) pairs[2 * dst + k] = arena[2 * src + k];
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
    return t + 256;
}

size_t MfScratch::tmp_bytes(uint32_t n) {
    const size_t n1 = n ? n : 1;
    return 2 * align256(n1 * 8) + 2 * align256(n1 * 4) + align256(lib_temp_bytes((uint32_t)n1));
}

int MfScratch::reserve(uint32_t n) {
    if (n <= cap && cap != 0) return LZMA_AMD_OK;
    const size_t n1 = n ? n : 1;
    LZ_HIP(ent4.alloc(n1 * sizeof(MfEnt)));
    LZ_HIP(ent8.alloc(n1 * sizeof(MfEnt)));
    LZ_HIP(ent16.alloc(n1 * sizeof(MfEnt)));
    LZ_HIP(info.alloc(n1 * sizeof(MfInfo)));
    cap = (uint32_t)n1;
    return LZMA_AMD_OK;
}

int MfScratch::build(const uint8_t *d_data, uint32_t n, uint32_t dict_size, uint32_t fb, hipStream_t st,
                     void *tmp, size_t tmp_cap) {
    int rc = reserve(n);
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
        temp = q;
        temp_bytes = need - (size_t)(q - (uint8_t *)tmp);
    }
    const uint32_t mask = lz::bt4_hash_mask(dict_size);
    view.data = d_data;
    view.n = n;
    view.ent4 = ent4.as<MfEnt>();
    view.ent8 = ent8.as<MfEnt>();
    view.ent16 = ent16.as<MfEnt>();
    view.info = info.as<MfInfo>();
    view.dict = dict_size;
    view.fb = fb;
    view.use8 = fb >= 8;
    view.use16 = fb >= 16;
    if (n == 0) return LZMA_AMD_OK;

    const unsigned g = grid_for(n, 256);
    // hash-4 buckets
    hipLaunchKernelGGL((mf_key_kernel<KEY4, uint32_t>), dim3(g), dim3(256), 0, st, d_data, n, mask,
                       (uint32_t *)kin);
    if ((rc = sort_by_key(*this, (uint32_t *)kin, (uint32_t *)kout, (uint32_t *)vout, n,
                          bits_of(mask), st)))
        return rc;
    hipLaunchKernelGGL(mf_head_kernel<uint32_t>, dim3(g), dim3(256), 0, st, (uint32_t *)kout, n,
                       (uint32_t *)grp);
    if ((rc = max_scan(*this, (uint32_t *)grp, n, st))) return rc;
    hipLaunchKernelGGL(mf_ent_kernel<4>, dim3(g), dim3(256), 0, st, d_data, n, (uint32_t *)vout,
                       (uint32_t *)grp, 0u, ent4.as<MfEnt>(), info.as<MfInfo>());
    // 8-gram groups
    if (view.use8) {
        hipLaunchKernelGGL((mf_key_kernel<KEY8, uint64_t>), dim3(g), dim3(256), 0, st, d_data, n, 0u,
                           (uint64_t *)kin);
        if ((rc = sort_by_key(*this, (uint64_t *)kin, (uint64_t *)kout, (uint32_t *)vout, n, 64,
                              st)))
            return rc;
        hipLaunchKernelGGL(mf_head_kernel<uint64_t>, dim3(g), dim3(256), 0, st, (uint64_t *)kout, n,
                           (uint32_t *)grp);
        if ((rc = max_scan(*this, (uint32_t *)grp, n, st))) return rc;
        hipLaunchKernelGGL(mf_ent_kernel<8>, dim3(g), dim3(256), 0, st, d_data, n, (uint32_t *)vout,
                           (uint32_t *)grp, 8u, ent8.as<MfEnt>(), info.as<MfInfo>());
    }
    // 16-gram groups: one sort by (8-gram group of the position, 8-gram group of the position + 8),
    // both known from the table just built (fb >= 16 implies the 8-gram table)
    if (view.use16) {
        hipLaunchKernelGGL(mf_key16_kernel, dim3(g), dim3(256), 0, st, info.as<MfInfo>(), n, (uint64_t *)kin);
        if ((rc = sort_by_key(*this, (uint64_t *)kin, (uint64_t *)kout, (uint32_t *)vout, n, 64, st)))
            return rc;
        hipLaunchKernelGGL(mf_head_kernel<uint64_t>, dim3(g), dim3(256), 0, st, (uint64_t *)kout, n,
                           (uint32_t *)grp);
        if ((rc = max_scan(*this, (uint32_t *)grp, n, st))) return rc;
        hipLaunchKernelGGL(mf_ent_kernel<16>, dim3(g), dim3(256), 0, st, d_data, n, (uint32_t *)vout,
                           (uint32_t *)grp, 16u, ent16.as<MfEnt>(), info.as<MfInfo>());
    }
    // hash-3 and hash-2 heads
    hipLaunchKernelGGL((mf_key_kernel<KEY3, uint32_t>), dim3(g), dim3(256), 0, st, d_data, n, 0u,
                       (uint32_t *)kin);
    if ((rc = sort_by_key(*this, (uint32_t *)kin, (uint32_t *)kout, (uint32_t *)vout, n, 16, st)))
        return rc;
    hipLaunchKernelGGL(mf_prev_kernel<3>, dim3(g), dim3(256), 0, st, (uint32_t *)vout,
                       (uint32_t *)kout, n, info.as<MfInfo>());
    hipLaunchKernelGGL((mf_key_kernel<KEY2, uint32_t>), dim3(g), dim3(256), 0, st, d_data, n, 0u,
                       (uint32_t *)kin);
    if ((rc = sort_by_key(*this, (uint32_t *)kin, (uint32_t *)kout, (uint32_t *)vout, n, 10, st)))
        return rc;
    hipLaunchKernelGGL(mf_prev_kernel<2>, dim3(g), dim3(256), 0, st, (uint32_t *)vout,
                       (uint32_t *)kout, n, info.as<MfInfo>());
    LZ_HIP(hipGetLastError());
    return LZMA_AMD_OK;
}

static unsigned scan_blocks(uint32_t n) {
    const uint32_t nruns = (n + kRun - 1) / kRun;
    unsigned blocks = (nruns + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    return blocks ? blocks : 1;
}

int MfResult::reserve(uint32_t n) {
    if (n <= cap && cap != 0) return LZMA_AMD_OK;
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
    LZ_HIP(arena.alloc(want * 8));
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
    int rc = r.reserve(n);
    if (rc) return rc;
    const unsigned blocks = scan_blocks(n);
    const uint64_t nrows = (uint64_t)blocks * 16;  // 16-lane rows, each with an arena page of its own
    for (int attempt = 0; attempt < 8; attempt++) {
        LZ_HIP(hipMemsetAsync(r.ctrl.p, 0, 3 * 8, st));
        hipLaunchKernelGGL(mf_scan_kernel, dim3(blocks * 4), dim3(64), 0, st, s.view, d_job_off, base, j0,
                           njobs, r.meta.as<uint64_t>(), r.arena.as<uint32_t>(), r.arena_pairs,
                           r.ctrl.as<unsigned long long>());
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
        LZ_HIP(r.arena.alloc(want * 8));
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
};
// never destroyed on its own: at process exit the HIP runtime may already be gone
thread_local FinderWs *g_fws = nullptr;
}  // namespace

void lzh::release_finder_workspace() {
    if (!g_fws || g_fws->device < 0) return;
    (void)hipDeviceSynchronize();
    g_fws->scratch.release();
    g_fws->res.release();
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
    lzh::DevBuf joff;
    const uint64_t h_joff[2] = {0, n};
    LZ_HIP(joff.alloc(16));
    LZ_HIP(hipMemcpyAsync(joff.p, h_joff, 16, hipMemcpyHostToDevice, st));
    uint64_t total = 0;
    t_scan.start(st);
    rc = lzh::mf_scan(ws, res, joff.as<uint64_t>(), 0, 0, 1, st, &total);
    t_scan.stop();
    if (rc) return rc;
    const unsigned g = grid_for(N, 256);
    hipLaunchKernelGGL(mf_counts_kernel, dim3(g), dim3(256), 0, st, res.d_meta(), N, d_counts);
    LZ_HIP(hipMemcpyAsync(d_total_pairs, &total, 8, hipMemcpyHostToDevice, st));
    lzh::DevBuf offs;
    if ((d_pairs && total <= pairs_cap) || d_offsets) {
        LZ_HIP(offs.alloc(n * 8));
        size_t bytes = 0;
        auto in = rocprim::make_transform_iterator(d_counts, [] __device__(uint32_t c) { return (uint64_t)c; });
        LZ_HIP(rocprim::exclusive_scan(nullptr, bytes, in, offs.as<uint64_t>(), (uint64_t)0, n,
                                       rocprim::plus<uint64_t>(), st));
        lzh::DevBuf temp;
        LZ_HIP(temp.alloc(bytes));
        LZ_HIP(rocprim::exclusive_scan(temp.p, bytes, in, offs.as<uint64_t>(), (uint64_t)0, n,
                                       rocprim::plus<uint64_t>(), st));
        LZ_HIP(hipStreamSynchronize(st));
        if (d_offsets) LZ_HIP(hipMemcpyAsync(d_offsets, offs.p, n * 8, hipMemcpyDeviceToDevice, st));
        if (d_pairs && total <= pairs_cap)
            hipLaunchKernelGGL(mf_gather_kernel, dim3(g), dim3(256), 0, st, res.d_meta(), res.d_arena(), N,
                               offs.as<uint64_t>(), d_pairs, (uint64_t)pairs_cap);
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
