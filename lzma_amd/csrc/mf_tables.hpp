// Host side of the match finder: builds the sorted position tables that
// lzma_mf_core.hpp scans and runs the scan over a batch of jobs.  Shared by the
// stand-alone finder entry point and the encoder.
//
// Memory is split in two so that batches can be pipelined: MfScratch holds what only
// the finder itself touches (tables and sort temporaries, reused by every batch) and
// MfResult what the consumer of a batch reads (one per batch in flight).  Nothing is
// allocated or freed between reserve() and the end of a call unless an estimate turns
// out too small: hipFree waits for every kernel on the device, which would serialize
// a pipeline.
#pragma once
#include "host_util.hpp"
#include "lzma_mf_core.hpp"

namespace lzh {

constexpr uint32_t kMfPagePairs = 4096;  // arena page: pairs a 16-lane row appends to before taking a new page

struct MfScratch {
    DevBuf ent[lz::kMfMaxLevels], info; // tables (see lzma_mf_core.hpp)
    DevBuf own_tmp;                 // temporaries, when the caller lends no memory for them
    uint32_t cap = 0;               // positions the tables hold
    uint32_t cap_levels = 0;        // levels they hold
    lz::MfTables view;
    // temporaries of one build: sort keys in/out, sorted positions, group starts, library scratch
    void *kin = nullptr, *kout = nullptr, *vout = nullptr, *grp = nullptr, *gid = nullptr, *temp = nullptr;
    size_t temp_bytes = 0;
    // No encode kernel is resident while this build's sorts run: they may use the library's
    // single-pass radix sort (see matchfinder.hip).  Set by the caller before build().
    bool alone = false;
    // Makes room for the tables of batches of up to n positions (n < 2^32 - 64) searched up to fb bytes.
    int reserve(uint32_t n, uint32_t fb);
    // Bytes of temporaries a build of n positions needs.
    static size_t tmp_bytes(uint32_t n);
    // Makes sure builds of up to n positions find their temporaries without allocating.
    int reserve_tmp(uint32_t n) {
        const size_t need = tmp_bytes(n);
        if (own_tmp.bytes < need) LZ_HIP(own_tmp.alloc(need));
        return LZMA_AMD_OK;
    }
    // Builds every table for d_data[0..n) on `st`; no host synchronization.  dict_size/fb as
    // given to the reference's newLzBinTree (lz_bin_tree.go:35).  The temporaries go to
    // tmp[0..tmp_cap) if that is large enough (memory that is dead until the scan, such as the
    // arena the scan will fill), else to an allocation of their own.
    int build(const uint8_t *d_data, uint32_t n, uint32_t dict_size, uint32_t fb, hipStream_t st,
              void *tmp = nullptr, size_t tmp_cap = 0);
    void release() {
        for (auto &e : ent) e.release();
        info.release(), own_tmp.release();
        cap = 0, cap_levels = 0;
    }
    uint64_t bytes() const {
        uint64_t b = info.bytes + own_tmp.bytes;
        for (const auto &e : ent) b += e.bytes;
        return b;
    }
    // table bytes per position (temporaries not included)
    static uint64_t bytes_per_position(uint32_t fb) {
        const uint32_t levels = lz::mf_levels_for(fb, 1u << 30);
        return levels * sizeof(lz::MfEnt) + lz::mf_info_words(levels) * 4;
    }
};

struct MfResult {
    DevBuf meta, arena, ctrl;  // meta[n] (u64), arena of (len, dist-1) pairs (two words or one packed), control words
    uint64_t arena_pairs = 0;
    uint32_t cap = 0;
    uint32_t pair_bytes = 8;  // 4 when the pairs are packed into one word (lz::mf_pack)
    int reserve(uint32_t n, uint32_t pair_bytes);
    void release() {
        meta.release(), arena.release(), ctrl.release();
        arena_pairs = 0, cap = 0;
    }
    const uint64_t *d_meta() const { return meta.as<uint64_t>(); }
    const uint32_t *d_arena() const { return arena.as<uint32_t>(); }
};

// Finds the match lists of every position of the batch whose tables `s` holds; jobs are
// [d_job_off[j]-base, d_job_off[j+1]-base) for j in [j0, j0+njobs).  Results land in `r`.
// *total_pairs = pairs emitted.  Synchronizes `st`; grows the arena and retries if it was
// too small.
int mf_scan(const MfScratch &s, MfResult &r, const uint64_t *d_job_off, uint64_t base, uint64_t j0,
            uint32_t njobs, hipStream_t st, uint64_t *total_pairs);

// Gives back the tables and lists that lzma_amd_bt4_matches keeps with the calling thread.
void release_finder_workspace();

}  // namespace lzh
