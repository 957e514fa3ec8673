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
// at a time, one per lane, record breakers picked with a ballot.
//
// To avoid scanning whole buckets (about 1400 candidates per position on text),
// positions are also grouped by their exact first 8 and first 16 bytes.  All candidates
// sharing at least 8 bytes with P are in P's 8-gram group; every bucket member newer
// than the newest of them (Q8) shares fewer than 8.  So the bucket is scanned only down
// to Q8; from Q8 on the (much shorter) 8-gram group is scanned instead, and that one
// only down to the newest position sharing 16 bytes (Q16), from where the 16-gram
// group takes over.
//
// All three scans stream: the sorted tables carry, next to each position, the bytes
// needed to compare it (bytes 0..7 in the bucket table, 8..15 in the 8-gram table,
// 16..23 in the 16-gram table), so the lanes read consecutive 12-byte entries instead
// of gathering from the window; only common prefixes of 24 bytes or more touch the
// window itself.  Most scans end within a few candidates, so the first step of each
// reads 16 entries and only the following ones 64.
#pragma once
#include "lzma_common.hpp"

namespace lz {

constexpr uint32_t kNone = 0xFFFFFFFFu;

// One entry of a sorted table: a position and 8 bytes of its text (see above).
struct MfEnt {
    uint32_t pos, sig_lo, sig_hi;
};
// Per position: its index in the three tables with the index of the first entry of
// its bucket / group in each, and the previous position with the same hash-2 / hash-3
// slot (kNone if there is none).
struct __attribute__((aligned(16))) MfInfo {
    uint32_t rank4, g4, rank8, g8;
    uint32_t rank16, g16, prev2, prev3;
};

struct MfTables {
    const uint8_t *data;  // all jobs' bytes, positions are offsets into it
    uint32_t n;           // bytes in data
    const MfEnt *ent4;    // (hash-4 slot, position) order, sig = bytes 0..7
    const MfEnt *ent8;    // (bytes 0..7, position) order, sig = bytes 8..15
    const MfEnt *ent16;   // (bytes 0..15, position) order, sig = bytes 16..23
    const MfInfo *info;
    uint32_t dict, fb, use8, use16;
};

// Result word per position: pair offset in the arena (upper 48 bits) and count.
LZ_HD uint64_t mf_meta(uint64_t off, uint32_t count) { return (off << 16) | count; }
LZ_HD uint64_t mf_meta_off(uint64_t m) { return m >> 16; }
LZ_HD uint32_t mf_meta_count(uint64_t m) { return (uint32_t)(m & 0xFFFF); }

#if defined(__HIPCC__)

// 8 bytes at data[i..i+8), zero-filled past the end of the buffer.
LZ_D uint64_t mf_load8(const uint8_t *data, uint32_t n, uint32_t i) {
    if (i + 8u <= n && i + 8u >= 8u) {
        uint64_t v;
        __builtin_memcpy(&v, data + i, 8);
        return v;
    }
    uint64_t v = 0;
    for (uint32_t k = 0; k < 8 && i < n && i + k < n; k++) v |= (uint64_t)data[i + k] << (8 * k);
    return v;
}

// Common prefix of the strings at P and Q (Q < P) beyond `from` known-equal bytes.
LZ_D uint32_t mf_extend(const uint8_t *data, uint32_t n, uint32_t P, uint32_t Q, uint32_t from,
                        uint32_t limit) {
    uint32_t l = from;
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

LZ_D uint32_t mf_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
LZ_D uint64_t mf_u64(uint64_t v) { return ((uint64_t)mf_u32((uint32_t)(v >> 32)) << 32) | mf_u32((uint32_t)v); }

// One level of the scan: entries tab[top-1], tab[top-2], ... down to tab[bottom], newest
// first, as long as their position is >= low.  `first` holds the 16 newest (lane k: entry
// top-1-k), fetched by the caller.  Candidate lengths are `known` + the bytes sig shares
// with psig, continued in the window beyond known+8 if `deep`.  Record breakers are
// emitted in order.  Returns true if a match of len_limit ended the whole search.
template <class Emit>
LZ_D bool mf_level(const MfTables &t, const MfEnt *tab, uint32_t top, uint32_t bottom, const MfEnt &first,
                   uint32_t low, uint64_t psig, uint32_t known, bool deep, uint32_t P, uint32_t len_limit,
                   uint32_t &n, uint32_t &max_len, Emit &emit, uint32_t lane) {
    const uint32_t count = top - bottom;  // entries older than P in the group
    for (uint32_t done = 0; done < count;) {
        const uint32_t w = done == 0 ? 16u : 64u;
        bool valid = lane < w && done + lane < count;
        uint32_t Q = 0, l = 0;
        if (valid) {
            MfEnt e = first;
            if (done != 0) e = tab[top - 1 - done - lane];
            Q = e.pos;
            valid = Q >= low;
            const uint64_t x = (((uint64_t)e.sig_hi << 32) | e.sig_lo) ^ psig;
            l = known + (x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8);
            if (deep && valid && x == 0 && len_limit > known + 8)
                l = mf_extend(t.data, t.n, P, Q, known + 8, len_limit);
            if (l > len_limit) l = len_limit;
        }
        for (;;) {  // record breakers in lane order
            const unsigned long long m = __ballot(valid && l > max_len);
            if (m == 0) break;
            const int f = __builtin_ctzll(m);
            const uint32_t bl = (uint32_t)__builtin_amdgcn_readlane((int)l, f);
            const uint32_t bq = (uint32_t)__builtin_amdgcn_readlane((int)Q, f);
            emit(n, bl, P - bq - 1);
            n++;
            max_len = bl;
            if (bl == len_limit) return true;  // lz_bin_tree.go:231-235
        }
        // go on only if every entry of this step was a candidate
        if (__ballot(valid) != (w == 16 ? 0xFFFFull : ~0ull)) break;
        done += w;
    }
    return false;
}

// Match list of position P of the job [job_start, job_end).  Called by all 64 lanes
// of a wave with identical arguments (inf = t.info[P] and p0/p8/p16 = bytes 0..7, 8..15,
// 16..23 at P, fetched by the caller ahead of time); emit(k, length, distance-1) is
// called once per pair, k = 0, 1, ... (identically on every lane); returns the number
// of pairs.  A later emit with a smaller k overwrites (the reference drops seed pairs
// that way).
//
// Everything that can be addressed from inf -- the 16 newest entries of the bucket, of
// the 8-gram group and of the 16-gram group, the two seed bytes -- is requested in one
// batch, so a typical position costs one memory round trip.
template <class Emit>
LZ_D uint32_t mf_find(const MfTables &t, uint32_t P, const MfInfo &inf, uint64_t p0, uint64_t p8,
                      uint64_t p16, uint32_t job_start, uint32_t job_end, Emit emit, uint32_t lane) {
    const uint32_t left = job_end - P;
    const uint32_t len_limit = left < t.fb ? left : t.fb;
    if (len_limit < 4) return 0;  // lz_bin_tree.go:121-124
    const uint8_t *data = t.data;
    // candidate Q is inside the window iff P - Q <= dict (cyclicBufSize = dict+1, :128-131)
    uint32_t W = P > t.dict ? P - t.dict : 0;
    if (W < job_start) W = job_start;
    const bool use8 = t.use8 && len_limit >= 8;
    const bool use16 = use8 && t.use16 && len_limit >= 16;

    // ---- one batch of loads
    const uint32_t r4 = inf.rank4, g4 = inf.g4, r8 = inf.rank8, g8 = inf.g8, r16 = inf.rank16, g16 = inf.g16;
    const uint32_t q2 = inf.prev2, q3 = inf.prev3;
    MfEnt a0, b0, c0;
    a0.pos = 0, a0.sig_lo = 0, a0.sig_hi = 0;
    b0 = a0, c0 = a0;
    if (lane < 16 && lane < r4 - g4) a0 = t.ent4[r4 - 1 - lane];
    if (use8 && lane < 16 && lane < r8 - g8) b0 = t.ent8[r8 - 1 - lane];
    if (use16 && lane < 16 && lane < r16 - g16) c0 = t.ent16[r16 - 1 - lane];
    uint32_t b2_v = 0x100, b3_v = 0x100;
    if (q2 != kNone && q2 >= W) b2_v = data[q2];
    if (q3 != kNone && q3 >= W) b3_v = data[q3];

    uint32_t q4 = kNone;
    if (r4 > g4) {
        q4 = (uint32_t)__builtin_amdgcn_readlane((int)a0.pos, 0);
        if (q4 < job_start) q4 = kNone;
    }
    const uint32_t b0byte = (uint32_t)p0 & 0xFF;
    const uint32_t b2 = mf_u32(b2_v), b3 = mf_u32(b3_v);
    uint32_t n = 0, max_len = 1, last = q2;
    // seeds, lz_bin_tree.go:154-180
    if (b2 == b0byte) {
        max_len = 2;
        emit(0u, 2u, P - q2 - 1);
        n = 1;
    }
    if (b3 == b0byte) {
        if (q3 == q2) n--;
        max_len = 3;
        emit(n, 3u, P - q3 - 1);
        n++;
        last = q3;
    }
    if (n != 0 && last == q4) {
        n--;
        max_len = 1;
    }

    // newest positions sharing P's first 8 and first 16 bytes
    uint32_t q8s = kNone, q16s = kNone;
    if (use8 && r8 > g8) {
        const uint32_t q = (uint32_t)__builtin_amdgcn_readlane((int)b0.pos, 0);
        if (q >= W) q8s = q;
    }
    if (use16 && r16 > g16) {
        const uint32_t q = (uint32_t)__builtin_amdgcn_readlane((int)c0.pos, 0);
        if (q >= W) q16s = q;
    }
    // the hash-4 bucket down to Q8, the 8-gram group down to Q16, the 16-gram group
    if (mf_level(t, t.ent4, r4, g4, a0, q8s != kNone ? q8s + 1 : W, p0, 0, false, P, len_limit, n, max_len, emit,
                 lane))
        return n;
    if (q8s == kNone) return n;
    if (mf_level(t, t.ent8, r8, g8, b0, q16s != kNone ? q16s + 1 : W, p8, 8, false, P, len_limit, n, max_len,
                 emit, lane))
        return n;
    if (q16s == kNone) return n;
    mf_level(t, t.ent16, r16, g16, c0, W, p16, 16, true, P, len_limit, n, max_len, emit, lane);
    return n;
}

#endif  // __HIPCC__

}  // namespace lz
