// Host side of the match finder: builds the sorted position tables that
// lzma_mf_core.hpp scans and runs the scan over a batch of jobs.  Shared by the
// stand-alone finder entry point and the encoder.
#pragma once
#include "host_util.hpp"
#include "lzma_mf_core.hpp"

namespace lzh {

constexpr uint32_t kMfPagePairs = 8192;  // arena page: pairs a wave appends to before taking a new page

// Device arrays for one batch of `n` positions (n < 2^32 - 64).
struct MfWorkspace {
    DevBuf ent4, ent8, info;  // tables
    DevBuf meta, arena, ctrl; // results: meta[n] (u64), arena of (len, dist-1) u32 pairs, ctrl words
    uint64_t arena_pairs = 0;
    lz::MfTables view;
    // Builds every table for d_data[0..n) on `st` (synchronizes the stream).  dict_size/fb as
    // given to the reference's newLzBinTree (lz_bin_tree.go:35).
    int build(const uint8_t *d_data, uint32_t n, uint32_t dict_size, uint32_t fb, hipStream_t st);
    // Finds the match lists of every position; jobs are [d_job_off[j]-base, d_job_off[j+1]-base)
    // for j in [j0, j0+njobs).  Results land in meta/arena.  *total_pairs = pairs emitted.
    // Grows the arena and retries if it was too small (synchronizes the stream).
    int scan(const uint64_t *d_job_off, uint64_t base, uint64_t j0, uint32_t njobs, hipStream_t st,
             uint64_t *total_pairs);
    const uint64_t *d_meta() const { return meta.as<uint64_t>(); }
    const uint32_t *d_arena() const { return arena.as<uint32_t>(); }
};

}  // namespace lzh
