// Device core of the position-parallel exhaustive BT4 match finder.
//
// What the reference computes (lzBinTree.getMatches, lz_bin_tree.go:115-253, with
// cutValue large enough that the walk is never cut) is, for a position P:
//   1. the hash-2 and hash-3 heads give seed pairs of length 2 and 3 (:148-180);
//   2. the walk down the binary tree of P's hash-4 bucket reports every node whose
//      common prefix with P is longer than anything reported so far (:196-250).
// Each bucket's tree is a search tree ordered by suffix in which a parent is always
// newer than its children, so the nodes on P's search path are exactly those with no
// newer node between them and P in suffix order; a node off the path always has a
// newer node on the path with at least as long a common prefix.  Hence the reported
// nodes are the *record breakers* when the bucket's in-window positions are scanned
// from newest to oldest, stopping at the first that reaches lenLimit -- a pure
// function of the window.  That scan is what this file implements, 64 candidates
// at a time, one per lane.
//
// To avoid scanning whole buckets (about 1400 candidates per position on text),
// positions are also grouped by their exact first 8 bytes.  All candidates sharing
// at least 8 bytes with P are in P's 8-gram group; every bucket member newer than
// the newest of them (Q*) shares fewer than 8.  So the bucket is scanned only down
// to Q*, and from Q* on the (much shorter) 8-gram group is scanned instead.
//
// Inputs are arrays produced by stable radix sorts of (key, position):
//   ord4/skey4: positions and hash-4 slots in (slot, position) order; rank4[P]
//   ord8: positions in (first 8 bytes, position) order; rank8[P]
//   prev2/prev3: previous position with the same hash-2 / hash-3 slot
#pragma once
#include "lzma_common.hpp"

namespace lz {

constexpr uint32_t kNone = 0xFFFFFFFFu;

struct MfTables {
    const uint8_t *data;  // all jobs' bytes, positions are offsets into it
    uint32_t n;           // bytes in data
    const uint32_t *ord4, *skey4, *rank4, *ord8, *rank8, *prev2, *prev3;
    uint32_t dict, fb, use8;
};

#if defined(__HIPCC__)

// 8 bytes at data[i..i+8), zero-filled past the end of the buffer.
LZ_D uint64_t mf_load8(const uint8_t *data, uint32_t n, uint32_t i) {
    if (i + 8u <= n && i + 8u >= 8u) {
        uint64_t v;
        __builtin_memcpy(&v, data + i, 8);
        return v;
    }
    uint64_t v = 0;
    for (uint32_t k = 0; k < 8 && i + k < n; k++) v |= (uint64_t)data[i + k] << (8 * k);
    return v;
}

// Common prefix length of the strings at P and Q (Q < P), at most `limit`.
LZ_D uint32_t mf_lcp(const uint8_t *data, uint32_t n, uint32_t P, uint32_t Q, uint32_t limit) {
    uint32_t l = 0;
    while (l < limit) {
        uint64_t x = mf_load8(data, n, P + l) ^ mf_load8(data, n, Q + l);
        if (x != 0) {
            l += (uint32_t)__builtin_ctzll(x) >> 3;
            break;
        }
        l += 8;
    }
    return l < limit ? l : limit;
}

// Match list of position P of the job [job_start, job_end).  Called by all 64 lanes
// of a wave with identical arguments; `out` (LDS or global, 2*(kMatchMax+1) words)
// receives (length, distance-1) pairs; returns the number of pairs.
template <class OutT>
LZ_D uint32_t mf_find(const MfTables &t, uint32_t P, uint32_t job_start, uint32_t job_end,
                      OutT *out, uint32_t lane) {
    const uint32_t left = job_end - P;
    const uint32_t len_limit = left < t.fb ? left : t.fb;
    if (len_limit < 4) return 0;  // lz_bin_tree.go:121-124
    const uint8_t *data = t.data;
    // candidate Q is inside the window iff P - Q <= dict (cyclicBufSize = dict+1, :128-131)
    uint32_t W = P > t.dict ? P - t.dict : 0;
    if (W < job_start) W = job_start;

    const uint32_t r4 = t.rank4[P];
    const uint32_t key = t.skey4[r4];
    uint32_t q4 = kNone;
    if (r4 > 0 && t.skey4[r4 - 1] == key) {
        q4 = t.ord4[r4 - 1];
        if (q4 < job_start) q4 = kNone;
    }
    const uint32_t q2 = t.prev2[P], q3 = t.prev3[P];
    const uint32_t b0 = data[P];
    uint32_t n = 0, max_len = 1, last = q2;
    // seeds, lz_bin_tree.go:154-180
    if (q2 != kNone && q2 >= W && data[q2] == b0) {
        max_len = 2;
        out[0] = 2;
        out[1] = P - q2 - 1;
        n = 1;
    }
    if (q3 != kNone && q3 >= W && data[q3] == b0) {
        if (q3 == q2) n--;
        max_len = 3;
        out[2 * n] = 3;
        out[2 * n + 1] = P - q3 - 1;
        n++;
        last = q3;
    }
    if (n != 0 && last == q4) {
        n--;
        max_len = 1;
    }

    // newest position sharing P's first 8 bytes
    uint32_t qstar = kNone, r8 = 0;
    if (t.use8 && len_limit >= 8) {
        r8 = t.rank8[P];
        if (r8 > 0) {
            uint32_t q = t.ord8[r8 - 1];
            if (q >= W && mf_load8(data, t.n, q) == mf_load8(data, t.n, P)) qstar = q;
        }
    }
    const uint32_t low_a = qstar != kNone ? qstar + 1 : W;

    // ---- scan of the hash-4 bucket, newest first, down to low_a
    for (uint32_t base = r4;; base -= 64) {
        bool valid = base > lane;  // candidate index base-1-lane >= 0
        uint32_t Q = 0;
        if (valid) {
            const uint32_t j = base - 1 - lane;
            Q = t.ord4[j];
            valid = t.skey4[j] == key && Q >= low_a;
        }
        uint32_t l = valid ? mf_lcp(data, t.n, P, Q, len_limit) : 0;
        for (;;) {  // record breakers in lane order
            unsigned long long m = __ballot(valid && l > max_len);
            if (m == 0) break;
            const int f = __builtin_ctzll(m);
            const uint32_t bl = (uint32_t)__builtin_amdgcn_readlane((int)l, f);
            const uint32_t bq = (uint32_t)__builtin_amdgcn_readlane((int)Q, f);
            out[2 * n] = bl;
            out[2 * n + 1] = P - bq - 1;
            n++;
            max_len = bl;
            if (bl == len_limit) return n;  // lz_bin_tree.go:231-235
        }
        if (__ballot(valid) != ~0ull || base <= 64) break;
    }
    if (qstar == kNone) return n;

    // ---- scan of the 8-gram group from Q* on
    for (uint32_t base = r8;; base -= 64) {
        bool valid = base > lane;
        uint32_t Q = 0, l = 0;
        if (valid) {
            Q = t.ord8[base - 1 - lane];
            valid = Q >= W;
        }
        if (valid) {
            l = mf_lcp(data, t.n, P, Q, len_limit);
            valid = l >= 8;  // still the same 8-gram
        }
        for (;;) {
            unsigned long long m = __ballot(valid && l > max_len);
            if (m == 0) break;
            const int f = __builtin_ctzll(m);
            const uint32_t bl = (uint32_t)__builtin_amdgcn_readlane((int)l, f);
            const uint32_t bq = (uint32_t)__builtin_amdgcn_readlane((int)Q, f);
            out[2 * n] = bl;
            out[2 * n + 1] = P - bq - 1;
            n++;
            max_len = bl;
            if (bl == len_limit) return n;
        }
        if (__ballot(valid) != ~0ull || base <= 64) break;
    }
    return n;
}

#endif  // __HIPCC__

}  // namespace lz
