// Host side of the match finder: builds the sorted position tables that
// lzma_mf_core.hpp scans.  Shared by the stand-alone finder and the encoder.
#pragma once
#include "host_util.hpp"
#include "lzma_mf_core.hpp"

namespace lzh {

// Device arrays for one batch of `n` positions (n < 2^31).
struct MfWorkspace {
    DevBuf ord4, skey4, rank4, ord8, rank8, prev2, prev3;
    lz::MfTables view;
    // Builds every table for d_data[0..n) on `st`; asynchronous (no stream sync) apart
    // from temporary allocations.  dict_size/fb as given to the reference's newLzBinTree
    // (lz_bin_tree.go:35).
    int build(const uint8_t *d_data, uint32_t n, uint32_t dict_size, uint32_t fb, hipStream_t st);
};

}  // namespace lzh
