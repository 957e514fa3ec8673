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
// positions are also grouped by their exact first 6, 8 and 16 bytes (levels 1..3; the
// hash-4 bucket is level 0), and for searches of 256 bytes and more (redundant data, where
// thousands of positions share 16 bytes) by their first 32, 64 and 128 bytes (levels 4..6).  All candidates sharing at least k bytes with P are in P's
// k-gram group; every member of a coarser group that is newer than the newest of them
// (Qk) shares fewer than k.  So each level is scanned only down to the newest member of
// the next level's group, from where the (much shorter) finer group takes over.
//
// All scans stream: the sorted tables carry, next to each position, the 8 bytes that
// follow the bytes its group shares (bytes 0..7 in the bucket table, 6..13, 8..15 and
// 16..23, ... in the others), so the lanes read consecutive 12-byte entries instead of
// gathering from the window; only common prefixes of 24 bytes or more touch the window
// itself.  A group's oldest entry carries a flag, so a scan knows where its group ends
// without a second table.
#pragma once
#include "lzma_common.hpp"

namespace lz {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kMfLevels = 4;     // levels of an ordinary build
constexpr uint32_t kMfMaxLevels = 7;  // with the deep levels, built for fb >= kMfDeepFb
constexpr uint32_t kMfDeepFb = 256;
// bytes the members of a level's group are known to share
LZ_HD uint32_t mf_known(uint32_t level) {
    return level == 0 ? 0u : (level == 1 ? 6u : (level == 2 ? 8u : 16u << (level - 3)));
}
// levels a build for `fb` over n positions has: a level needs searches that long, and the deep
// ones a text several times their length (their order comes from shifting the previous level's)
LZ_HD uint32_t mf_levels_for(uint32_t fb, uint32_t n) {
    if (n < 32) return 1;
    const uint32_t cap = fb >= kMfDeepFb ? kMfMaxLevels : kMfLevels;
    uint32_t l = 1;
    while (l < cap && fb >= mf_known(l) && (l < kMfLevels || n >= 4 * mf_known(l))) l++;
    return l;
}
constexpr uint32_t kMfFirst = 0x80000000u;  // entry: it is the oldest of its group
// A position's rank word says how many older members its group has, so that a search fetches
// no more entries than can be candidates: tables of up to 2^30 entries keep a class in bits
// 30..31 (0: none, 1: one or two, 2: three or four, 3: more), larger ones a flag in bit 31
// (set: none).
constexpr uint32_t kMfClassMax = 1u << 30;
LZ_HD uint32_t mf_rank_word(uint32_t rank, uint32_t older_capped5, bool wide) {
    if (wide) return rank | (older_capped5 == 0 ? 0x80000000u : 0u);
    const uint32_t cls = older_capped5 == 0 ? 0u : (older_capped5 <= 2 ? 1u : (older_capped5 <= 4 ? 2u : 3u));
    return rank | (cls << 30);
}
LZ_HD uint32_t mf_rank_of(uint32_t w, bool wide) { return wide ? (w & 0x7FFFFFFFu) : (w & 0x3FFFFFFFu); }
// entries worth fetching below the position's own (0: it is the oldest of its group)
LZ_HD uint32_t mf_rank_fetch(uint32_t w, bool wide) {
    if (wide) return (w >> 31) ? 0u : 16u;
    const uint32_t cls = w >> 30;
    return cls == 0 ? 0u : (cls == 1 ? 2u : (cls == 2 ? 4u : 16u));
}

// One entry of a sorted table: a position (bit 31: it is the oldest of its group) and the 8
// bytes of its text that follow the bytes the group shares.
struct MfEnt {
    uint32_t pos, sig_lo, sig_hi;
};
// Per position: its rank word in each level's table (mf_rank_word)
// and the previous position with the same hash-2 / hash-3 slot if it starts with the same byte
// (kNone otherwise).
template <uint32_t NL>
struct __attribute__((aligned(16))) MfInfoT {
    uint32_t rank[NL];
    uint32_t prev2, prev3;
    uint32_t pad[NL == kMfLevels ? 2 : 3];
};
typedef MfInfoT<kMfLevels> MfInfo;        // 32 bytes
typedef MfInfoT<kMfMaxLevels> MfInfoDeep;  // 48 bytes
// words per record, and the words of prev2 / prev3 in it, for a build with `levels` levels
LZ_HD uint32_t mf_info_words(uint32_t levels) { return levels > kMfLevels ? 12u : 8u; }
LZ_HD uint32_t mf_info_prev(uint32_t levels) { return levels > kMfLevels ? kMfMaxLevels : kMfLevels; }

struct MfTables {
    const uint8_t *data;  // all jobs' bytes, positions are offsets into it
    uint32_t n;           // bytes in data
    const MfEnt *ent[kMfMaxLevels];  // level l: (group, position) order, sig = bytes known(l) .. known(l)+7
    const void *info;     // MfInfo per position, MfInfoDeep if more than kMfLevels levels were built
    uint32_t dict, fb;
    uint32_t levels;      // levels built (mf_levels_for)
    uint32_t wide;        // more than 2^30 positions: rank words carry a flag instead of a class
    uint32_t packed;      // pairs are written as one word each (mf_pack) instead of two
};

// A pair in one word, for dictionaries of up to 8 MiB: length - 2 (at most 271) in bits
// 0..8, distance - 1 in bits 9..31.  Halves the match lists, which is what limits how many
// chunks an encode call keeps in flight.
constexpr uint32_t kMfPackDistBits = 23;
LZ_HD bool mf_can_pack(uint32_t dict_size) { return dict_size <= (1u << kMfPackDistBits); }
LZ_HD uint32_t mf_pack(uint32_t len, uint32_t dist) { return (len - 2u) | (dist << 9); }
LZ_HD uint32_t mf_pack_len(uint32_t w) { return (w & 0x1FFu) + 2u; }
LZ_HD uint32_t mf_pack_dist(uint32_t w) { return w >> 9; }

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

// Section timing for diagnostic builds (-DMF_PROFILE): MF_LAP(i) charges the cycles since the
// previous lap to section i of the wave's counters.  Compiled out otherwise.
#if defined(MF_PROFILE)
struct MfProf {
    uint64_t t[8], last;
};
__device__ MfProf *g_mf_prof_;  // set per wave by the kernel
#define MF_PROF_ARG , MfProf &pf
#define MF_PROF_PASS , pf
#define MF_LAP(i) { const uint64_t t_ = __builtin_readcyclecounter(); pf.t[i] += t_ - pf.last; pf.last = t_; }
#else
#define MF_PROF_ARG
#define MF_PROF_PASS
#define MF_LAP(i)
#endif

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

// Output of one row: pair k goes to dst[2k], dst[2k+1], or packed to dst[k].
struct MfRowOut {
    uint32_t *dst;
    bool on;  // false: count only (arena full)
    bool packed;
};
LZ_D void mf_put(uint32_t *dst, bool packed, uint32_t k, uint32_t len, uint32_t dist) {
    if (packed) {
        dst[k] = mf_pack(len, dist);
    } else {
        dst[2 * k] = len;
        dst[2 * k + 1] = dist;
    }
}

LZ_D uint32_t lane_get(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
LZ_D uint64_t lane_get64(uint64_t v, int src) {
    return ((uint64_t)lane_get((uint32_t)(v >> 32), src) << 32) | lane_get((uint32_t)v, src);
}

// What a row knows about its position before any table entry is looked at: the record of
// the position and 24 bytes of its text.
template <uint32_t NL>
struct MfPos {
    MfInfoT<NL> inf;
    uint64_t p0, p8, p16;        // bytes 0..7, 8..15, 16..23 at P (zero beyond the buffer)
    uint64_t pd[NL - kMfLevels + 1];  // deep levels: bytes 32..39, 64..71, 128..135 (one unused word otherwise)
};
// The 16 newest entries of every level's group (lane k of the row: the entry k + 1 places
// below the position's own), requested in one batch.
template <uint32_t NL>
struct MfFront {
    MfEnt e[NL];
};

template <uint32_t NL>
LZ_D void mf_load_pos(const MfTables &t, uint32_t P, bool act, MfPos<NL> &a) {
    a.inf = MfInfoT<NL>{};
    a.p0 = a.p8 = a.p16 = 0;
#pragma unroll
    for (uint32_t i = 0; i < NL - kMfLevels + 1; i++) a.pd[i] = 0;
    if (act) {
        a.inf = ((const MfInfoT<NL> *)t.info)[P];
        a.p0 = mf_load8(t.data, t.n, P), a.p8 = mf_load8(t.data, t.n, P + 8);
        a.p16 = mf_load8(t.data, t.n, P + 16);
#pragma unroll
        for (uint32_t l = kMfLevels; l < NL; l++)
            if (l < t.levels) a.pd[l - kMfLevels] = mf_load8(t.data, t.n, P + mf_known(l));
    }
}
template <uint32_t NL>
LZ_D void mf_load_front(const MfTables &t, const MfPos<NL> &a, bool act, uint32_t idx, MfFront<NL> &f) {
#pragma unroll
    for (uint32_t l = 0; l < NL; l++) {
        f.e[l].pos = 0, f.e[l].sig_lo = 0, f.e[l].sig_hi = 0;
        const uint32_t w = a.inf.rank[l], r = mf_rank_of(w, t.wide != 0);
        // the rank words of levels that were not built are never looked at
        if (act && l < t.levels && idx < mf_rank_fetch(w, t.wide != 0) && idx < r) f.e[l] = t.ent[l][r - 1 - idx];
    }
}

// One level of the scan for every row that has `go` set: entries tab[top-1], tab[top-2],
// ... newest first, down to the oldest of the position's group and as long as their
// position is >= low.  `first` holds the 16 newest (lane k of the row: entry top-1-k),
// fetched by the caller (only the first `fetched` of them where the group is known to be
// that small).  Candidate lengths are `known` + the bytes sig shares with psig,
// continued in the window beyond known+8 if `deep`.  A candidate is reported if it is
// longer than everything newer.  Sets fin where a match of len_limit ended the whole search.
//
// Most scans end within the first 16 candidates: that step runs for the four rows at once,
// every lane writing its own pair.  A row that has to go on does so with all 64 lanes of
// the wave, one such row after the other.
LZ_D void mf_level(const MfTables &t, const MfEnt *tab, uint32_t top, uint32_t fetched, const MfEnt &first, uint32_t low,
                   uint64_t psig, uint32_t known, bool deep, uint32_t P, uint32_t len_limit, uint32_t ceil,
                   uint32_t &n, uint32_t &max_len, bool &fin, bool go, const MfRowOut &out, uint32_t lane MF_PROF_ARG) {
    const uint32_t idx = lane & 15u, row_base = lane & 48u;
    // no candidate of this level is longer than `ceil` (all that share more are in the next
    // level's group): once a match that long is found the level has nothing more to report
    go = go && !fin && top != 0 && max_len < ceil;
    if (!__any(go)) return;
    {   // ---- first step, four rows at once
        // `fetched` < 16 entries were fetched where the group is known to have no more older members
        const bool inb = go && idx < top && idx < fetched;
        // an entry that is the oldest of its group ends the group: older ones are not candidates
        const uint32_t ends = row_ballot(inb && (first.pos & kMfFirst) != 0, row_base);
        bool valid = inb && (ends & ((1u << idx) - 1u)) == 0;
        uint32_t Q = 0, l = 0;
        if (valid) {
            Q = first.pos & ~kMfFirst;
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
            mf_put(out.dst, out.packed, k, l, P - Q - 1);
        }
        // the row's longest candidate so far, from its highest lane
        const uint32_t top_l = (uint32_t)__shfl((int)max(lv, before), (int)(row_base + 15));
        if (go) {
            n += (uint32_t)__builtin_popcount(m);
            if (top_l > max_len) max_len = top_l;
            if (max_len == len_limit && m != 0) fin = true;  // lz_bin_tree.go:231-235
        }
        // go on only if every entry of this step was a candidate and the group goes on
        go = go && !fin && vm == 0xFFFFu && ends == 0 && top > 16 && max_len < ceil;
    }
    MF_LAP(2)
    // ---- rows that go on, one after the other, 64 candidates per step
    unsigned long long pending = __ballot(go && idx == 0);
    while (pending != 0) {
        const int src = __builtin_ctzll(pending);  // first lane of the row
        pending &= pending - 1;
        const uint32_t u_top = lane_get(top, src), u_low = lane_get(low, src);
        const uint32_t u_P = lane_get(P, src), u_limit = lane_get(len_limit, src), u_ceil = lane_get(ceil, src);
        const uint64_t u_sig = lane_get64(psig, src);
        uint32_t *u_dst = (uint32_t *)lane_get64((uint64_t)out.dst, src);
        const bool u_on = lane_get((uint32_t)out.on, src) != 0;
        const bool u_packed = out.packed;  // the same for every row
        uint32_t u_n = lane_get(n, src), u_max = lane_get(max_len, src);
        bool u_fin = false;
        for (uint32_t done = 16; done < u_top && !u_fin; done += 64) {
            const bool inb = done + lane < u_top;
            MfEnt e;
            e.pos = 0, e.sig_lo = 0, e.sig_hi = 0;
            if (inb) e = tab[u_top - 1 - done - lane];
            const unsigned long long ends = __ballot(inb && (e.pos & kMfFirst) != 0);
            bool valid = inb && (ends & ((1ull << lane) - 1ull)) == 0;
            uint32_t Q = 0, l = 0;
            if (valid) {
                Q = e.pos & ~kMfFirst;
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
                if (lane == 0 && u_on) mf_put(u_dst, u_packed, u_n, bl, u_P - bq - 1);
                u_n++;
                u_max = bl;
                if (bl == u_limit) {
                    u_fin = true;
                    break;
                }
            }
            if (__ballot(valid) != ~0ull || ends != 0 || u_max >= u_ceil) break;
        }
        if (row_base == (uint32_t)src) n = u_n, max_len = u_max, fin = fin || u_fin;
    }
    MF_LAP(3)
}

// Match lists of four positions, one per row.  P, `a` (the position's record and text), `f`
// (the newest entries of its groups) and the job [job_start, job_end) are per row; rows with
// act == false take no part.  Returns the row's number of pairs; the pairs themselves are
// written through `out` (a later pair with a smaller index overwrites: the reference drops
// seed pairs that way).
// TINY: the text is too short for any level but the buckets (mf_levels_for), whose candidates
// then share more than the 8 bytes their entries carry.
template <uint32_t NL, bool TINY>
LZ_D uint32_t mf_find4(const MfTables &t, uint32_t P, const MfPos<NL> &a, const MfFront<NL> &f, uint32_t job_start,
                       uint32_t job_end, bool act, const MfRowOut &out, uint32_t lane MF_PROF_ARG) {
    const uint32_t idx = lane & 15u, row_base = lane & 48u;
    const uint32_t left = act ? job_end - P : 0;
    const uint32_t len_limit = left < t.fb ? left : t.fb;
    act = act && len_limit >= 4;  // lz_bin_tree.go:121-124
    // candidate Q is inside the window iff P - Q <= dict (cyclicBufSize = dict+1, :128-131)
    uint32_t W = P > t.dict ? P - t.dict : 0;
    if (W < job_start) W = job_start;

    // inf.prev2 / prev3 are kNone unless that position starts with P's first byte
    const uint32_t q2 = a.inf.prev2, q3 = a.inf.prev3;
    const bool hit2 = act && q2 != kNone && q2 >= W, hit3 = act && q3 != kNone && q3 >= W;

    // per level: the position's rank, whether it has an older member of its group at all, and
    // the newest such member
    uint32_t top[NL], newest[NL], fetched[NL];
    bool has[NL];
#pragma unroll
    for (uint32_t l = 0; l < NL; l++) {
        const uint32_t w = a.inf.rank[l], r = mf_rank_of(w, t.wide != 0);
        // a level is used if it was built and the position has that many bytes to compare
        const bool use = act && l < t.levels && (l == 0 || len_limit >= mf_known(l));
        has[l] = use && mf_rank_fetch(w, t.wide != 0) != 0 && r != 0;
        top[l] = has[l] ? r : 0u;
        fetched[l] = mf_rank_fetch(w, t.wide != 0);
        newest[l] = row_first(f.e[l].pos, row_base) & ~kMfFirst;
    }
    uint32_t q4 = kNone;
    if (has[0] && newest[0] >= job_start) q4 = newest[0];
    uint32_t n = 0, max_len = 1, last = q2;
    const bool lead = idx == 0 && out.on;
    // seeds, lz_bin_tree.go:154-180
    if (hit2) {
        max_len = 2;
        if (lead) mf_put(out.dst, out.packed, 0, 2u, P - q2 - 1);
        n = 1;
    }
    if (hit3) {
        if (q3 == q2) n--;
        max_len = 3;
        if (lead) mf_put(out.dst, out.packed, n, 3u, P - q3 - 1);
        n++;
        last = q3;
    }
    if (n != 0 && last == q4) {
        n--;
        max_len = 1;
    }

    MF_LAP(1)
    // each level down to the newest in-window member of the next level's group
    bool fin = false;
    bool go[NL];
#pragma unroll
    for (uint32_t l = 0; l < NL; l++) go[l] = has[l] && (l == 0 || newest[l] >= W);
#pragma unroll
    for (uint32_t l = 0; l < NL; l++) {
        const uint32_t low = (l + 1 < NL && go[l + 1]) ? newest[l + 1] + 1 : W;
        uint64_t psig;
        if (l == 0) psig = a.p0;
        else if (l == 1) psig = (a.p0 >> 48) | (a.p8 << 16);
        else if (l == 2) psig = a.p8;
        else if (l == 3) psig = a.p16;
        else psig = a.pd[l < kMfLevels ? 0 : l - kMfLevels];
        // candidates of this level share fewer bytes than the next level's groups do
        const uint32_t ceil = (l + 1 < NL && l + 1 < t.levels && len_limit >= mf_known(l + 1))
                                  ? mf_known(l + 1) - 1 : len_limit;
        // candidates can share more than the 8 bytes their entries carry from 16 bytes on (below
        // that the next level, or fb, ends them first)
        mf_level(t, t.ent[l], top[l], fetched[l], f.e[l], low, psig, mf_known(l), l >= 3 || TINY, P, len_limit, ceil,
                 n, max_len, fin, go[l], out, lane MF_PROF_PASS);
    }
    return n;
}

#endif  // __HIPCC__

}  // namespace lz
