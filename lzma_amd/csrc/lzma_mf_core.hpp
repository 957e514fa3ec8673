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
// function of the window.  That scan is what this file implements, 16 candidates
// at a time for each of four positions per wave (the record breakers found with a
// prefix maximum over the lanes), then 64 at a time for the positions that need more.
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
// window itself.
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
// slot if it starts with the same byte (kNone otherwise).
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

// ---- four positions per wave
// A wave works on four positions at once, one per group of 16 lanes ("row"); within a
// row the lanes hold the 16 candidates of one step.  Values described as "per row" are
// held by every lane of the row.  All cross-lane operations below stay inside a row.

LZ_D uint32_t row_ballot(bool p, uint32_t row_base) { return (uint32_t)(__ballot(p) >> row_base) & 0xFFFFu; }
LZ_D uint32_t row_first(uint32_t v, uint32_t row_base) { return (uint32_t)__shfl((int)v, (int)row_base); }
// value of the lane `k` places below in the row, `fill` where there is none
template <int K>
LZ_D uint32_t row_shr(uint32_t v, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x110 + K, 0xF, 0xF, false);
}
// max of v over the lanes of the row below this one (0 if none)
LZ_D uint32_t row_max_below(uint32_t v) {
    v = max(v, row_shr<1>(v, 0));
    v = max(v, row_shr<2>(v, 0));
    v = max(v, row_shr<4>(v, 0));
    v = max(v, row_shr<8>(v, 0));
    return row_shr<1>(v, 0);
}

// Output of one row: pairs go to dst[2k], dst[2k+1].
struct MfRowOut {
    uint32_t *dst;
    bool on;  // false: count only (arena full)
};

LZ_D uint32_t lane_get(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
LZ_D uint64_t lane_get64(uint64_t v, int src) {
    return ((uint64_t)lane_get((uint32_t)(v >> 32), src) << 32) | lane_get((uint32_t)v, src);
}

// One level of the scan for every row that has `go` set: entries tab[top-1], tab[top-2],
// ... down to tab[bottom], newest first, as long as their position is >= low.  `first`
// holds the 16 newest (lane k of the row: entry top-1-k), fetched by the caller.
// Candidate lengths are `known` + the bytes sig shares with psig, continued in the window
// beyond known+8 if `deep`.  A candidate is reported if it is longer than everything newer.
// Sets fin where a match of len_limit ended the whole search.
//
// Most scans end within the first 16 candidates: that step runs for the four rows at once,
// every lane writing its own pair.  A row that has to go on does so with all 64 lanes of
// the wave, one such row after the other.
LZ_D void mf_level(const MfTables &t, const MfEnt *tab, uint32_t top, uint32_t bottom, const MfEnt &first,
                   uint32_t low, uint64_t psig, uint32_t known, bool deep, uint32_t P, uint32_t len_limit,
                   uint32_t &n, uint32_t &max_len, bool &fin, bool go, const MfRowOut &out, uint32_t lane) {
    const uint32_t idx = lane & 15u, row_base = lane & 48u;
    const uint32_t count = top - bottom;  // entries older than P in the group
    go = go && !fin && count != 0;
    if (!__any(go)) return;
    {   // ---- first step, four rows at once
        bool valid = go && idx < count;
        uint32_t Q = 0, l = 0;
        if (valid) {
            Q = first.pos;
            valid = Q >= low;
            const uint64_t x = (((uint64_t)first.sig_hi << 32) | first.sig_lo) ^ psig;
            l = known + (x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8);
            if (deep && valid && x == 0 && len_limit > known + 8)
                l = mf_extend(t.data, t.n, P, Q, known + 8, len_limit);
            if (l > len_limit) l = len_limit;
        }
        const uint32_t lv = valid ? l : 0;
        const uint32_t before = max(max_len, row_max_below(lv));  // longest among everything newer
        const bool hit = lv > before;
        const uint32_t m = row_ballot(hit, row_base), vm = row_ballot(valid, row_base);
        if (hit && out.on) {
            const uint32_t k = n + (uint32_t)__builtin_popcount(m & ((1u << idx) - 1u));
            out.dst[2 * k] = l;
            out.dst[2 * k + 1] = P - Q - 1;
        }
        // the row's longest candidate so far, from its highest lane
        const uint32_t top_l = (uint32_t)__shfl((int)max(lv, before), (int)(row_base + 15));
        if (go) {
            n += (uint32_t)__builtin_popcount(m);
            if (top_l > max_len) max_len = top_l;
            if (max_len == len_limit && m != 0) fin = true;  // lz_bin_tree.go:231-235
        }
        // go on only if every entry of this step was a candidate
        go = go && !fin && vm == 0xFFFFu && count > 16;
    }
    // ---- rows that go on, one after the other, 64 candidates per step
    unsigned long long pending = __ballot(go && idx == 0);
    while (pending != 0) {
        const int src = __builtin_ctzll(pending);  // first lane of the row
        pending &= pending - 1;
        const uint32_t u_top = lane_get(top, src), u_count = lane_get(count, src), u_low = lane_get(low, src);
        const uint32_t u_P = lane_get(P, src), u_limit = lane_get(len_limit, src);
        const uint64_t u_sig = lane_get64(psig, src);
        uint32_t *u_dst = (uint32_t *)lane_get64((uint64_t)out.dst, src);
        const bool u_on = lane_get((uint32_t)out.on, src) != 0;
        uint32_t u_n = lane_get(n, src), u_max = lane_get(max_len, src);
        bool u_fin = false;
        for (uint32_t done = 16; done < u_count && !u_fin; done += 64) {
            bool valid = done + lane < u_count;
            uint32_t Q = 0, l = 0;
            if (valid) {
                const MfEnt e = tab[u_top - 1 - done - lane];
                Q = e.pos;
                valid = Q >= u_low;
                const uint64_t x = (((uint64_t)e.sig_hi << 32) | e.sig_lo) ^ u_sig;
                l = known + (x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8);
                if (deep && valid && x == 0 && u_limit > known + 8)
                    l = mf_extend(t.data, t.n, u_P, Q, known + 8, u_limit);
                if (l > u_limit) l = u_limit;
            }
            for (;;) {  // record breakers in lane order
                const unsigned long long m = __ballot(valid && l > u_max);
                if (m == 0) break;
                const int f = __builtin_ctzll(m);
                const uint32_t bl = lane_get(l, f), bq = lane_get(Q, f);
                if (lane == 0 && u_on) u_dst[2 * u_n] = bl, u_dst[2 * u_n + 1] = u_P - bq - 1;
                u_n++;
                u_max = bl;
                if (bl == u_limit) {
                    u_fin = true;
                    break;
                }
            }
            if (__ballot(valid) != ~0ull) break;
        }
        if (row_base == (uint32_t)src) n = u_n, max_len = u_max, fin = fin || u_fin;
    }
}

// Match lists of four positions, one per row.  P, inf (= t.info[P]), the text bytes p0 /
// p8 / p16 (bytes 0..7, 8..15, 16..23 at P, zero beyond the buffer) and the job
// [job_start, job_end) are per row;
// rows with act == false take no part.  Returns the row's number of pairs; the pairs
// themselves are written through `out` (a later pair with a smaller index overwrites: the
// reference drops seed pairs that way).
//
// Everything that can be addressed from inf -- the 16 newest entries of the bucket, of
// the 8-gram group and of the 16-gram group -- is requested in one
// batch, so a typical position costs one memory round trip.
LZ_D uint32_t mf_find4(const MfTables &t, uint32_t P, const MfInfo &inf, uint64_t p0, uint64_t p8,
                       uint64_t p16, uint32_t job_start, uint32_t job_end, bool act, const MfRowOut &out,
                       uint32_t lane) {
    const uint32_t idx = lane & 15u, row_base = lane & 48u;
    const uint32_t left = act ? job_end - P : 0;
    const uint32_t len_limit = left < t.fb ? left : t.fb;
    act = act && len_limit >= 4;  // lz_bin_tree.go:121-124
    // candidate Q is inside the window iff P - Q <= dict (cyclicBufSize = dict+1, :128-131)
    uint32_t W = P > t.dict ? P - t.dict : 0;
    if (W < job_start) W = job_start;
    const bool use8 = act && t.use8 && len_limit >= 8;
    const bool use16 = use8 && t.use16 && len_limit >= 16;

    // ---- one batch of loads
    const uint32_t r4 = inf.rank4, g4 = inf.g4, r8 = inf.rank8, g8 = inf.g8, r16 = inf.rank16, g16 = inf.g16;
    const uint32_t q2 = inf.prev2, q3 = inf.prev3;
    MfEnt a0, b0, c0;
    a0.pos = 0, a0.sig_lo = 0, a0.sig_hi = 0;
    b0 = a0, c0 = a0;
    if (act && idx < r4 - g4) a0 = t.ent4[r4 - 1 - idx];
    if (use8 && idx < r8 - g8) b0 = t.ent8[r8 - 1 - idx];
    if (use16 && idx < r16 - g16) c0 = t.ent16[r16 - 1 - idx];
    // inf.prev2 / prev3 are kNone unless that position starts with P's first byte
    const bool hit2 = act && q2 != kNone && q2 >= W, hit3 = act && q3 != kNone && q3 >= W;

    const uint32_t a_new = row_first(a0.pos, row_base), b_new = row_first(b0.pos, row_base),
                   c_new = row_first(c0.pos, row_base);
    uint32_t q4 = kNone;
    if (act && r4 > g4 && a_new >= job_start) q4 = a_new;
    uint32_t n = 0, max_len = 1, last = q2;
    const bool lead = idx == 0 && out.on;
    // seeds, lz_bin_tree.go:154-180
    if (hit2) {
        max_len = 2;
        if (lead) out.dst[0] = 2u, out.dst[1] = P - q2 - 1;
        n = 1;
    }
    if (hit3) {
        if (q3 == q2) n--;
        max_len = 3;
        if (lead) out.dst[2 * n] = 3u, out.dst[2 * n + 1] = P - q3 - 1;
        n++;
        last = q3;
    }
    if (n != 0 && last == q4) {
        n--;
        max_len = 1;
    }

    // newest positions sharing P's first 8 and first 16 bytes
    uint32_t q8s = kNone, q16s = kNone;
    if (use8 && r8 > g8 && b_new >= W) q8s = b_new;
    if (use16 && r16 > g16 && c_new >= W) q16s = c_new;
    // the hash-4 bucket down to Q8, the 8-gram group down to Q16, the 16-gram group
    bool fin = false;
    mf_level(t, t.ent4, r4, g4, a0, q8s != kNone ? q8s + 1 : W, p0, 0, false, P, len_limit, n, max_len, fin, act,
             out, lane);
    mf_level(t, t.ent8, r8, g8, b0, q16s != kNone ? q16s + 1 : W, p8, 8, false, P, len_limit, n, max_len, fin,
             q8s != kNone, out, lane);
    mf_level(t, t.ent16, r16, g16, c0, W, p16, 16, true, P, len_limit, n, max_len, fin, q16s != kNone, out, lane);
    return n;
}

#endif  // __HIPCC__

}  // namespace lz
