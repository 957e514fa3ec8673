// Core of the per-chunk LZMA encoder: price-optimal parse + range coder.
//
// Results follow the reference encoder exactly: getOptimum/backward
// (lzma_encoder.go:331-797), codeOneBlock/flush/writeEndMarker (:836-978), the
// price tables (:294-329, :801-834), the length coder with its cached price table
// (lzma_len_coder.go:48-126), the literal coder (lzma_lit_coder.go:47-102), the
// bit trees (range_bit_tree_coder.go:53-117) and the range encoder
// (range_coder.go:131-246).
//
// What differs is how the work is laid out for a 64-lane wavefront that owns one
// chunk:
//   * the match finder is a functor `mf(pos, pairs)` that returns the match list
//     of any position directly (the lists of the exhaustive BT4 finder are a pure
//     function of the data, so skipped positions cost nothing);
//   * one step of the parse (one position `cur` of getOptimum's main loop) is two
//     lane-parallel phases instead of nested loops:
//       round:   every lane runs the same code on its own task -- lanes 0-3 measure a
//                rep match and price its "rep + literal + rep0" continuation, one lane
//                prices the literal, the others price the "match + literal + rep0"
//                continuation of one pair of the match list each;
//       apply:   one lane per optimum-table entry cur+2.. applies, in the reference's
//                order and with its strict comparisons, every candidate that targets
//                that entry.  (Candidate prices never depend on the entries they
//                update, and for any entry the reference visits its candidates in a
//                fixed order -- continuation of literal, then per rep its lengths and
//                its continuation, then match continuations by pair, then the plain
//                match length -- so applying them per entry gives the same table.)
//   * the non-literal models, the 16-bit price tables and a ring window of the
//     optimum table that follows `cur` live in LDS; the full 4096-entry table sits
//     in HBM, because on text the parse runs to the end of it almost every time: an
//     entry is written there once, when the parse position reaches it (entries beyond
//     the window directly);
//   * match lengths are measured eight bytes at a time;
//   * getPosSlot is computed with a count-leading-zeros instead of a 2 KiB table.
//
// Code outside LZ_PAR_BEGIN/LZ_PAR_END is executed by all lanes with identical
// values (so the compiler keeps it on the scalar unit); code inside runs once per
// lane and only communicates through memory.  The same source compiles on the host
// (LZ_HOSTSIM), where a parallel section is a loop over the 64 lanes.
#pragma once
#include "lzma_common.hpp"

namespace lz {

constexpr uint32_t kLanes = 64;
// The optimum table (4096 entries) lives in HBM; LDS holds a ring window of it that
// follows the parse position `cur`: entries [cur - kRecBack, cur - kRecBack + kRecWin)
// of the path records and [cur - kBksWin + 1, cur] of the rep-distance records.
#ifndef LZ_REC_WIN
#define LZ_REC_WIN 64
#endif
constexpr uint32_t kRecWin = LZ_REC_WIN;           // power of two
#ifndef LZ_REC_BACK
#define LZ_REC_BACK ((LZ_REC_WIN * 3) / 8)
#endif
constexpr uint32_t kRecBack = LZ_REC_BACK;         // entries kept behind cur
constexpr uint32_t kBksWin = kRecWin / 2;          // power of two
// Sizes that follow from the number of fast bytes: a match list holds at most fb - 1
// pairs (lengths 2..fb, strictly increasing), and lengths above fb are never priced.
LZ_HD uint32_t la_max(uint32_t fb) { return fb + 1; }  // continuation records of the pairs
LZ_HD uint32_t mdist_words(uint32_t fb) { return 2 * (fb + 1) + 2; }
// words shared by the continuation records (inside a parse) and the distance-price scratch (between parses)
LZ_HD uint32_t la_words(uint32_t fb) { return 2 * la_max(fb) > kNumFullDist ? 2 * la_max(fb) : kNumFullDist; }
// state-price table: one 16-byte entry per (state, posState)
LZ_HD uint32_t sp_words(uint32_t pb) { return (kNumStates << pb) * 4; }
// uint32 words of caller-provided scratch besides the tables (see Encoder::bind_small)
LZ_HD uint32_t enc_small_words(uint32_t fb, uint32_t pb) {
    return sp_words(pb) + 4 + (2u << pb) + la_words(fb) + mdist_words(fb);
}

// LZ_LDS qualifies pointers into LDS, so that their accesses are ds_* instructions
// and can never be merged with global accesses into flat ones.
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define LZ_LDS __attribute__((address_space(3)))
#else
#define LZ_LDS
#endif

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define LZ_PAR_BEGIN(L) { const uint32_t L = lane_id;
#define LZ_PAR_END } __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#else
#define LZ_PAR_BEGIN(L) for (uint32_t L = 0; L < kLanes; L++) {
#define LZ_PAR_END }
#endif

// Values that differ from lane to lane and are read across lanes: on the GPU a vector
// register, read with readlane (one lane's value for all, LZ_FROM_LANE) or reduced over the
// wave (LZ_WAVE_MAX, LZ_WAVE_COUNT; all 64 lanes must execute these); on the host an array with one element per
// lane.  LZ_LV(name, L) is lane L's element inside a parallel section.
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define LZ_LANEVAR(name) uint32_t name = 0
#define LZ_LANEARG(name) uint32_t name
#define LZ_LV(name, L) name
#define LZ_FROM_LANE(name, k) ((uint32_t)__builtin_amdgcn_readlane((int)(name), (int)(k)))
template <int K>
LZ_D uint32_t lz_row_shr(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + K, 0xF, 0xF, false);
}
LZ_D uint32_t lz_wave_max(uint32_t v) {
    // each shifted value is taken once, by all lanes, before it is compared: a cross-lane move
    // inside a conditional would run with some lanes switched off and read nothing from them
    uint32_t t = lz_row_shr<1>(v);
    v = v > t ? v : t;
    t = lz_row_shr<2>(v);
    v = v > t ? v : t;
    t = lz_row_shr<4>(v);
    v = v > t ? v : t;
    t = lz_row_shr<8>(v);
    v = v > t ? v : t;  // lane 15 of every row: the row's maximum
    const uint32_t a = LZ_FROM_LANE(v, 15), b = LZ_FROM_LANE(v, 31), c = LZ_FROM_LANE(v, 47), d = LZ_FROM_LANE(v, 63);
    const uint32_t x = a > b ? a : b, y = c > d ? c : d;
    return x > y ? x : y;
}
LZ_D uint32_t lz_sum8(uint32_t v) {  // sum of lanes 0..7
    v += lz_row_shr<1>(v);
    v += lz_row_shr<2>(v);
    v += lz_row_shr<4>(v);
    return LZ_FROM_LANE(v, 7);
}
#define LZ_WAVE_MAX(name) lz_wave_max(name)
#define LZ_WAVE_COUNT(name) ((uint32_t)__builtin_popcountll(__ballot((name) != 0)))
#define LZ_SUM8(name) lz_sum8(name)
#define LZ_LANES_BEGIN(L) { const uint32_t L = lane_id;
#define LZ_LANES_END }
#else
#define LZ_LANEVAR(name) uint32_t name[kLanes] = {}
#define LZ_LANEARG(name) const uint32_t *name
#define LZ_LV(name, L) name[L]
#define LZ_FROM_LANE(name, k) name[k]
inline uint32_t lz_wave_max_host(const uint32_t *v) {
    uint32_t m = 0;
    for (uint32_t i = 0; i < kLanes; i++) m = v[i] > m ? v[i] : m;
    return m;
}
#define LZ_WAVE_MAX(name) lz_wave_max_host(name)
inline uint32_t lz_wave_count_host(const uint32_t *v) {
    uint32_t m = 0;
    for (uint32_t i = 0; i < kLanes; i++) m += v[i] != 0;
    return m;
}
#define LZ_WAVE_COUNT(name) lz_wave_count_host(name)
inline uint32_t lz_sum8_host(const uint32_t *v) {
    uint32_t m = 0;
    for (uint32_t i = 0; i < 8; i++) m += v[i];
    return m;
}
#define LZ_SUM8(name) lz_sum8_host(name)
#define LZ_LANES_BEGIN(L) for (uint32_t L = 0; L < kLanes; L++) {
#define LZ_LANES_END }
#endif

// Section timing for diagnostic builds (-DLZ_PROFILE): LZ_LAP(i) charges the cycles since
// the previous lap to section i.  Compiled out otherwise.
#if defined(LZ_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define LZ_LAP(i) { const uint64_t t_ = __builtin_readcyclecounter(); prof[i] += t_ - prof_last; prof_last = t_; }
#else
#define LZ_LAP(i)
#endif

struct EncParams {
    uint32_t fb, lc, lp, pb, dict_size, dist_table_size, eos;
};

// Optimum-table entry beyond the LDS window (`optimal`, lzma_encoder.go:167-181).
// pp packs posPrev (bits 0-11), posPrev2 (12-23), state (24-27), prev1IsChar (28),
// prev2 (29).
struct __attribute__((aligned(16))) Rec {
    uint32_t price, back, pp, back2;
};
struct __attribute__((aligned(16))) Reps {
    uint32_t r[4];
};
struct OptG {
    Rec rec;
    Reps bks;
};
constexpr uint32_t kPpPrev1 = 1u << 28, kPpPrev2 = 1u << 29;

// Price of one bit, range_coder.go:222-246.
LZ_HD void build_prob_prices(uint16_t *pp, uint32_t lane, uint32_t nlanes) {
    for (uint32_t j = lane; j < 512; j += nlanes) {
        if (j == 0) {
            pp[0] = 0;
            continue;
        }
        uint32_t i = 8;  // j in [2^(8-i), 2^(9-i))
        while ((j >> (9 - i)) != 0) i--;
        uint32_t end = 1u << (9 - i);
        pp[j] = (uint16_t)((i << kPriceShift) + (((end - j) << kPriceShift) >> (9 - i - 1)));
    }
}

// getPosSlot / getPosSlot2 (lzma_encoder.go:147-165) without the lookup table.
LZ_HD uint32_t pos_slot(uint32_t pos) {
    if (pos < 4) return pos;
    uint32_t n = 31u - (uint32_t)__builtin_clz(pos);
    return (n << 1) + ((pos >> (n - 1)) & 1);
}

// Layout of the 16-bit price tables in one array.
struct PriceLayout {
    uint32_t len, rep_len, pos_slot, dist, align, total;  // offsets in u16 units
    uint32_t len_stride;                                  // length symbols kept per posState
};
LZ_HD PriceLayout price_layout(uint32_t pb, uint32_t fb) {
    PriceLayout l;
    uint32_t nps = 1u << pb;
    l.len_stride = (fb + 1 - kMatchMin + 1) & ~1u;  // symbols 0 .. fb - kMatchMin, rounded up to even
    l.len = 0;
    l.rep_len = l.len + nps * l.len_stride;
    l.pos_slot = l.rep_len + nps * l.len_stride;
    l.dist = l.pos_slot + (kNumLenToPos << kPosSlotBits);
    l.align = l.dist + (kNumLenToPos * kNumFullDist);
    l.total = l.align + kAlignSize;
    return l;
}

// U(v): v is known to be the same on every lane; on the GPU this moves it to a scalar
// register so that the code depending on it runs on the scalar unit with real branches.
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
LZ_D uint32_t U(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
#else
LZ_HD uint32_t U(uint32_t v) { return v; }
#endif

LZ_HD uint64_t enc_load8(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// Words of the block through which the two waves of a cooperative workgroup synchronise
// (see Encoder::parse_loop).  Counters grow with the parse position inside one parse.
enum : uint32_t {
    SY_FORK = 0,   // parses started (written by wave 0); kSyDone: the chunk is finished
    SY_JOIN,       // parses the second wave has left
    SY_JOIN2,      // ... and the third
    SY_S,          // highest position whose round is done and whose literal candidate is published
    SY_A,          // highest position whose apply phase is done; SY_LEN_END / SY_FAR_HI belong to it
    SY_M,          // highest position whose literal candidate has been merged into the next entry
    SY_EXIT,       // a wave has found the end of the parse: SY_EX_* are valid
    SY_LEN_END, SY_FAR_HI,
    SY_PKT_PRICE, SY_PKT_SP = SY_PKT_PRICE + 2, SY_PKT_EQ = SY_PKT_SP + 2,  // literal candidate, by parity
    SY_NIC = SY_PKT_EQ + 2,                                                  // its outcome, by parity
    SY_EX_CUR = SY_NIC + 2, SY_EX_LONG, SY_EX_LEN, SY_EX_DIST,
    SY_POS0, SY_MF0,
    SY_WORDS = 32
};
constexpr uint32_t kSyDone = 0xFFFFFFFFu;

// Flag accesses between the waves of a workgroup.  Data written before lz_sy_store is visible
// to a wave that has seen the stored value with lz_sy_load.
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
// The waves of a workgroup share their CU's LDS and first-level cache, so a reader needs no
// more than to issue its accesses after it has seen the flag.  A writer whose data is in LDS
// alone needs no wait either (a wave's LDS instructions are performed in order: the data is
// there before the flag); one that has written to HBM as well waits for those stores first
// (hbm_too).
LZ_D uint32_t lz_sy_load(const LZ_LDS uint32_t *p) {
    const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)*(const volatile LZ_LDS uint32_t *)p);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return v;
}
LZ_D void lz_sy_store(LZ_LDS uint32_t *p, uint32_t v, bool one_lane, bool hbm_too = true) {
    if (hbm_too)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    else
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (one_lane) *(volatile LZ_LDS uint32_t *)p = v;
}
LZ_D void lz_sy_pause() { __builtin_amdgcn_s_sleep(1); }
#else
}  // namespace lz
#include <sched.h>
namespace lz {
inline uint32_t lz_sy_load(const uint32_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline void lz_sy_store(uint32_t *p, uint32_t v, bool, bool = true) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
inline void lz_sy_pause() { sched_yield(); }
#endif

// NW > 1: that many waves (two or three) share one chunk (parse_loop); each wave has an Encoder
// of its own over the same LDS and HBM state, with a match-list buffer and continuation
// records of its own.
template <class MF, class LP, uint32_t NW = 1>
struct Encoder {
    static constexpr bool COOP = NW > 1;
    typedef LZ_LDS prob_t *PP;
    typedef LZ_LDS uint32_t *U32P;
    typedef uint16_t WT;
    // ---- memory the caller provides ----
    const uint8_t *data;  // chunk bytes
    uint32_t n;           // chunk length
    uint32_t readable;    // bytes that may be read starting at data (>= n)
    PP probs;             // kNumBaseProbs models
    LP lit;               // 0x300 << (lc+lp) literal models (LDS or HBM)
    LZ_LDS WT *prices;    // price_layout(pb).total entries
    const LZ_LDS WT *pp;  // 512-entry bit-price table
    LZ_LDS Rec *o_rec;    // kRecWin slots: price, backPrev, packed path word, backPrev2
    LZ_LDS Reps *o_bks;   // kBksWin slots: backs0..3
    uint32_t wlo, wcur, far_hi;  // window start, current cur, highest entry written outside the window
    U32P sync;                   // COOP: SY_WORDS words shared by the two waves
    OptG *opt_g;          // kNumOpts entries (HBM), always current
    uint8_t *out;         // payload output (after the header)
    uint64_t out_cap;
    MF mf;
    EncParams ep;
    PriceLayout pl;
    uint32_t lane_id;
#if defined(LZ_PROFILE)
    uint64_t prof[16], prof_last;
#endif

    // ---- prices that depend only on (state, posState)
    struct StatePrices {
        uint32_t im0, rep_match, nmp, rep_len1, prp[4];
    };
    LZ_HD StatePrices sp_direct(uint32_t st, uint32_t ps) const {  // lzma_encoder.go:566-597, per lane
        const uint32_t im_v = probs[P_IS_MATCH + ((st << kPosStatesBitsMax) + ps)], ir_v = probs[P_IS_REP + st];
        const uint32_t g0_v = probs[P_IS_REP_G0 + st], g1_v = probs[P_IS_REP_G1 + st];
        const uint32_t g2_v = probs[P_IS_REP_G2 + st], rl_v = probs[P_IS_REP0_LONG + ((st << kPosStatesBitsMax) + ps)];
        const uint32_t im0 = price0(im_v), im1 = price1(im_v), ir0 = price0(ir_v), ir1 = price1(ir_v);
        const uint32_t g00 = price0(g0_v), g01 = price1(g0_v), g10 = price0(g1_v), g11 = price1(g1_v);
        const uint32_t g20 = price0(g2_v), g21 = price1(g2_v), rl0 = price0(rl_v), rl1 = price1(rl_v);
        StatePrices r;
        r.im0 = im0;
        r.rep_match = im1 + ir1;
        r.nmp = im1 + ir0;
        r.rep_len1 = g00 + rl0;
        r.prp[0] = r.rep_match + g00 + rl1;
        r.prp[1] = r.rep_match + g01 + g10;
        r.prp[2] = r.rep_match + g01 + g11 + g20;
        r.prp[3] = r.rep_match + g01 + g11 + g21;
        return r;
    }
    // The models are frozen while a parse runs (they move when symbols are coded, between
    // parses), so the state prices of all (state, posState) pairs are tabulated once per parse,
    // one pair per lane, and a look-up is a single 16-byte read instead of six models and
    // twelve bit prices.  Entry: {im0 | rep_match << 16, nmp | rep_len1 << 16, prp0 | prp1 << 16,
    // prp2 | prp3 << 16}; every field is a sum of at most five bit prices and fits 16 bits.
    LZ_HD void sp_refresh() {
        const uint32_t total = kNumStates << ep.pb;
        LZ_PAR_BEGIN(L)
        for (uint32_t k = L; k < total; k += kLanes) {
            const StatePrices v = sp_direct(k >> ep.pb, k & pos_state_mask);
            Reps e;
            e.r[0] = v.im0 | (v.rep_match << 16), e.r[1] = v.nmp | (v.rep_len1 << 16);
            e.r[2] = v.prp[0] | (v.prp[1] << 16), e.r[3] = v.prp[2] | (v.prp[3] << 16);
            sp_tab[k] = e;
        }
        LZ_PAR_END
    }
    // single fields, per lane
    LZ_HD uint32_t sp_im0(uint32_t st, uint32_t ps) const { return sp_tab[(st << ep.pb) + ps].r[0] & 0xFFFF; }
    LZ_HD uint32_t sp_prp0(uint32_t st, uint32_t ps) const { return sp_tab[(st << ep.pb) + ps].r[2] & 0xFFFF; }
    // all fields of a wave-uniform (state, posState), in scalar registers
    LZ_HD StatePrices sp_get(uint32_t st, uint32_t ps) const {
        const Reps v = sp_tab[(st << ep.pb) + ps];
        const Reps e = ureps(v);
        StatePrices r;
        r.im0 = e.r[0] & 0xFFFF, r.rep_match = e.r[0] >> 16, r.nmp = e.r[1] & 0xFFFF, r.rep_len1 = e.r[1] >> 16;
        r.prp[0] = e.r[2] & 0xFFFF, r.prp[1] = e.r[2] >> 16, r.prp[2] = e.r[3] & 0xFFFF, r.prp[3] = e.r[3] >> 16;
        return r;
    }

    // ---- range encoder, range_coder.go:131-138 ----
    uint64_t low, out_n;
    uint32_t cache_size, cache, range, overflow;

    // ---- encoder state, lzma_encoder.go:210-272 ----
    uint32_t mfpos;  // positions already handed to the match finder
    uint32_t longest_len, dist_pairs, add_offset, opt_end, opt_cur, longest_found;
    uint32_t align_price_count, match_price_count, pos_state_mask, lit_pos_mask;
    uint32_t now_pos, finished, state, prev_byte, back_res;
    // Small arrays in caller-provided (LDS) memory: never in per-lane scratch.
    LZ_LDS Reps *sp_tab;                             // state prices of every (state, posState), see sp_refresh
    U32P rep_dist;                                   // 4
    U32P len_counter;                                // [2][1 << pb]
    U32P temp_prices;                                // kNumFullDist; shares its words with la_tgt / la_price
    U32P la_tgt, la_price;                           // la_max(fb) each, by pair: target entry (0 = none), price
    U32P mdist;                                      // current match list, (len, dist) pairs

    LZ_HD void bind_small(U32P small, uint32_t fb, uint32_t pb) {
        sp_tab = (LZ_LDS Reps *)small;               // 16-byte aligned: first
        rep_dist = small + sp_words(pb);
        len_counter = rep_dist + 4;
        temp_prices = len_counter + (2u << pb);
        la_tgt = temp_prices;
        la_price = la_tgt + la_max(fb);
        mdist = temp_prices + la_words(fb);
    }

    // ------------------------------------------------------- optimum table access
    // Writes go to HBM and, inside the window, to LDS; reads inside the window come
    // from LDS.  Entry i of the window sits in slot i mod kRecWin.
    // Stores from wave-uniform code are issued by one lane: 64 lanes writing the same
    // address cost the memory pipeline 64 writes.
    LZ_HD bool one() const { return lane_id == 0; }
    // waves at consecutive positions see windows one entry apart in the same ring: their
    // union must fit it, so a cooperative window is shorter by one entry per extra wave
    static constexpr uint32_t kWinSpan = kRecWin - (NW - 1);
    LZ_HD bool in_win(uint32_t i) const { return i - wlo < kWinSpan; }
    LZ_HD bool in_bwin(uint32_t i) const { return wcur - i < kBksWin; }
    LZ_HD Rec grec(uint32_t i) const {
        Rec r;
        if (in_win(i)) r = o_rec[i & (kRecWin - 1)]; else r = opt_g[i].rec;
        return r;
    }
    LZ_HD Reps gbks(uint32_t i) const {
        // the LDS slot is read in any case and replaced under a branch: one pointer chosen between
        // the two memories would make every one of these reads a generic-address load
        Reps r = o_bks[i & (kBksWin - 1)];
        if (!in_bwin(i)) r = opt_g[i].bks;
        return r;
    }
    LZ_HD void sbks(uint32_t i, const Reps &v) {  // only ever written at cur
        opt_g[i].bks = v;
        o_bks[i & (kBksWin - 1)] = v;
    }
    LZ_HD uint32_t oprice(uint32_t i) const {
        return in_win(i) ? o_rec[i & (kRecWin - 1)].price : opt_g[i].rec.price;
    }
    LZ_HD uint32_t opp(uint32_t i) const { return in_win(i) ? o_rec[i & (kRecWin - 1)].pp : opt_g[i].rec.pp; }
    // An entry inside the window lives in LDS alone until the parse position reaches it:
    // only then is it final, and commit_rec() writes it to HBM once (entries ahead of cur
    // are updated many times; writing each update through cost a memory store per lane and
    // step, and every later wait for a load waited for those stores too).  Entries beyond
    // the window are written to HBM directly and fetched when the window reaches them.
    LZ_HD void set_price(uint32_t i, uint32_t v) {
        if (in_win(i)) o_rec[i & (kRecWin - 1)].price = v; else opt_g[i].rec.price = v;
    }
    // price, backPrev and path word in one go (backPrev2 keeps its old value)
    LZ_HD void set_pbp(uint32_t i, uint32_t price, uint32_t back, uint32_t ppw) {
        if (in_win(i)) {
            LZ_LDS Rec *r = o_rec + (i & (kRecWin - 1));
            r->price = price, r->back = back, r->pp = ppw;
        } else {
            opt_g[i].rec.price = price, opt_g[i].rec.back = back, opt_g[i].rec.pp = ppw;
        }
    }
    // entry i (inside the window, wave-uniform value r) is final: to HBM, path word to LDS
    LZ_HD void commit_rec(uint32_t i, const Rec &r) {
        if (one()) {
            opt_g[i].rec = r;
            o_rec[i & (kRecWin - 1)].pp = r.pp;
        }
    }
    // the same for an entry whose LDS copy is complete (the end of a parse)
    LZ_HD void commit_from_window(uint32_t i) {
        if (one()) opt_g[i].rec = o_rec[i & (kRecWin - 1)];
    }
    LZ_HD void u_set_pbp(uint32_t i, uint32_t price, uint32_t back, uint32_t ppw) {
        if (one()) set_pbp(i, price, back, ppw);
    }
    LZ_HD void u_sbks(uint32_t i, const Reps &v) {
        if (one()) sbks(i, v);
    }
    LZ_HD void srec(uint32_t i, const Rec &v) {
        if (in_win(i)) o_rec[i & (kRecWin - 1)] = v; else opt_g[i].rec = v;
    }
    // Uniform helpers: loads are issued first and made uniform afterwards, so that
    // several are in flight at once.
    static LZ_HD Rec urec(Rec r) {
        r.price = U(r.price), r.back = U(r.back), r.pp = U(r.pp), r.back2 = U(r.back2);
        return r;
    }
    static LZ_HD Reps ureps(Reps r) {
        r.r[0] = U(r.r[0]), r.r[1] = U(r.r[1]), r.r[2] = U(r.r[2]), r.r[3] = U(r.r[3]);
        return r;
    }
    LZ_HD uint32_t up0(uint32_t p) const { return U(price0(p)); }
    LZ_HD uint32_t up1(uint32_t p) const { return U(price1(p)); }
    // Window bookkeeping when cur advances to c (by one): the entry that enters the
    // window is fetched from HBM if anything was ever written that far out.
    // The fetch is issued here (into `pend`, a local of the caller) and lands in the window when
    // win_land() is called, after the position's match list has been read: that much work hides
    // most of the memory round trip, and the four registers are live only across it (the
    // kernel has none to spare further on).  Returns the window slot waiting for it, or kNoSlot.
    static constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
    LZ_HD uint32_t win_advance(uint32_t c, Rec &pend) {
        wcur = c;
        uint32_t slot = kNoSlot;
        const uint32_t lo = c > kRecBack ? c - kRecBack : 0;
        if (lo != wlo) {
            wlo = lo;
            const uint32_t e = lo + kWinSpan - 1;
            if (e <= far_hi && e < kNumOpts) {
                if (one()) pend = opt_g[e].rec;
                slot = e & (kRecWin - 1);
            }
        }
        return slot;
    }
    LZ_HD void win_land(uint32_t slot, const Rec &pend) {
        if (slot != kNoSlot && one()) o_rec[slot] = pend;
    }
    // COOP: the window of position c is set when the wave turns to c (win_set) and the entry it
    // gains is fetched when the apply phases of all earlier positions are done (win_enter)
    LZ_HD void win_set(uint32_t c) {
        wcur = c;
        wlo = c > kRecBack ? c - kRecBack : 0;
    }
    // (issued as soon as the earlier apply phases are done, landed with win_land() just before
    // this position's own apply phase: the waits and scalar work in between hide the round trip)
    LZ_HD uint32_t win_enter(uint32_t c, Rec &pend) {
        uint32_t slot = kNoSlot;
        if (c > kRecBack) {
            const uint32_t e = c - kRecBack + kWinSpan - 1;
            if (e <= far_hi && e < kNumOpts) {
                if (one()) pend = opt_g[e].rec;
                slot = e & (kRecWin - 1);
            }
        }
        return slot;
    }
    LZ_HD void note_far(uint32_t hi) {  // an apply phase wrote entries up to hi
        if (hi >= wlo + kWinSpan && hi > far_hi) far_hi = hi;
    }
    // ---- the two waves of a cooperative workgroup
    // Every wait between the waves is bounded (about a second; the longest legitimate wait, for
    // the symbols of one parse to be coded, is a thousand times shorter): a wave that gives up
    // raises kSyAbort in SY_EXIT, which ends every other wait, and the chunk reports an error.
    static constexpr uint32_t kSpinLimit = 1u << 23;
    static constexpr uint32_t kSyAbort = 2;
    uint32_t dead;  // this wave gave up, or saw the other one do so
    LZ_HD void sy_give_up() {
        dead = 1;
        lz_sy_store(sync + SY_EXIT, kSyAbort, one());
    }
    // waits until counter `which` has reached v; false if the parse has ended meanwhile
    LZ_HD bool sy_wait(uint32_t which, uint32_t v) {
        for (uint32_t spins = 0;; spins++) {
            if (lz_sy_load(sync + which) >= v) return true;
            const uint32_t x = lz_sy_load(sync + SY_EXIT);
            if (x != 0) {
                if (x == kSyAbort) dead = 1;
                return false;
            }
            if (spins > kSpinLimit) {
                sy_give_up();
                return false;
            }
            lz_sy_pause();
        }
    }
    // waits until word `which` differs from (differ) or equals (!differ) v; 0 after giving up
    LZ_HD uint32_t sy_wait_word(uint32_t which, uint32_t v, bool differ) {
        for (uint32_t spins = 0;; spins++) {
            const uint32_t f = lz_sy_load(sync + which);
            if ((f != v) == differ) return f;
            if (lz_sy_load(sync + SY_EXIT) == kSyAbort || spins > (differ ? kSpinLimit << 4 : kSpinLimit)) {
                sy_give_up();
                return 0;
            }
            lz_sy_pause();
        }
    }
    LZ_HD void sy_signal(uint32_t which, uint32_t v, bool hbm_too = true) {
        lz_sy_store(sync + which, v, one(), hbm_too);
    }

    // ------------------------------------------------------------ range coder
    LZ_HD void rc_put(uint32_t b) {
        if (out_n < out_cap) {
            if (one()) out[out_n] = (uint8_t)b;
        } else
            overflow = 1;
        out_n++;
    }
    LZ_HD void rc_shift_low() {  // range_coder.go:168-186
        uint32_t hi = (uint32_t)(low >> 32);
        if (hi != 0 || low < 0xFF000000ull) {
            uint32_t t = cache;
            do {
                rc_put(t + hi);
                t = 0xFF;
            } while (--cache_size != 0);
            cache = ((uint32_t)low) >> 24;
        }
        cache_size++;
        low = (uint64_t)(((uint32_t)low) << 8);
    }
    template <class P>
    LZ_HD void rc_bit(P pr, uint32_t idx, uint32_t bit) {  // range_coder.go:205-220
        uint32_t p = U(pr[idx]), bound = (range >> kProbBits) * p;
        uint32_t np;
        if (bit == 0) {
            range = bound;
            np = p + ((kProbTotal - p) >> kMoveBits);
        } else {
            low += bound;
            range -= bound;
            np = p - (p >> kMoveBits);
        }
        if (one()) pr[idx] = (prob_t)np;
        if (range < kTopValue) {
            range <<= 8;
            rc_shift_low();
        }
    }
    LZ_HD void rc_direct(uint32_t v, uint32_t nbits) {  // range_coder.go:188-199
        for (int32_t i = (int32_t)nbits - 1; i >= 0; i--) {
            range >>= 1;
            if ((v >> i) & 1) low += range;
            if (range < kTopValue) {
                range <<= 8;
                rc_shift_low();
            }
        }
    }
    // Up to eight bits whose models are all known before the first is coded (the nodes of one
    // walk down a bit tree): lane i fetches and updates the model of bit i (ix: its index, bt: the
    // bit), one memory round trip for the walk instead of one per bit; the range coder then takes
    // the models from the lanes.  The indices of one call are distinct.
    template <class P>
    LZ_HD void rc_encode_lanes(P pr, LZ_LANEARG(ix), LZ_LANEARG(bt), uint32_t nbits) {
        LZ_LANEVAR(pv);
        LZ_LANES_BEGIN(L)
        if (L < nbits) {
            const uint32_t p = pr[LZ_LV(ix, L)];
            LZ_LV(pv, L) = p;
            pr[LZ_LV(ix, L)] = (prob_t)(LZ_LV(bt, L) ? p - (p >> kMoveBits) : p + ((kProbTotal - p) >> kMoveBits));
        }
        LZ_LANES_END
        for (uint32_t i = 0; i < nbits; i++) {  // range_coder.go:205-220
            const uint32_t p = LZ_FROM_LANE(pv, i), bit = LZ_FROM_LANE(bt, i);
            const uint32_t bound = (range >> kProbBits) * p;
            if (bit == 0) {
                range = bound;
            } else {
                low += bound;
                range -= bound;
            }
            if (range < kTopValue) {
                range <<= 8;
                rc_shift_low();
            }
        }
    }
#ifndef LZ_BATCH_BT
#define LZ_BATCH_BT 0
#endif
#ifndef LZ_BATCH_LIT
#define LZ_BATCH_LIT 1
#endif
    static constexpr bool kBatchBt = COOP || LZ_BATCH_BT, kBatchLit = COOP || LZ_BATCH_LIT;
    template <class P>
    LZ_HD void bt_encode(P pr, uint32_t base, uint32_t nbits, uint32_t sym) {  // nbits <= 8
        if constexpr (!kBatchBt) {
            uint32_t k = 1;
            for (uint32_t i = nbits; i != 0;) {
                i--;
                uint32_t b = (sym >> i) & 1;
                rc_bit(pr, base + k, b);
                k = (k << 1) | b;
            }
            return;
        }
        LZ_LANEVAR(ix);
        LZ_LANEVAR(bt);
        LZ_LANES_BEGIN(L)
        if (L < nbits) {  // depth L: the node is 1 followed by the L highest bits of the symbol
            LZ_LV(ix, L) = base + ((1u << L) | (sym >> (nbits - L)));
            LZ_LV(bt, L) = (sym >> (nbits - 1 - L)) & 1;
        }
        LZ_LANES_END
        rc_encode_lanes(pr, ix, bt, nbits);
    }
    template <class P>
    LZ_HD void bt_rev_encode(P pr, uint32_t base, uint32_t nbits, uint32_t sym) {  // nbits <= 8
        if constexpr (!kBatchBt) {
            uint32_t k = 1;
            for (uint32_t i = 0; i < nbits; i++) {
                uint32_t b = sym & 1;
                rc_bit(pr, base + k, b);
                k = (k << 1) | b;
                sym >>= 1;
            }
            return;
        }
        LZ_LANEVAR(ix);
        LZ_LANEVAR(bt);
        LZ_LANES_BEGIN(L)
        if (L < nbits) {  // depth L: 1 followed by the L lowest bits of the symbol, lowest first
            uint32_t k = 1;
            for (uint32_t t = 0; t < L; t++) k = (k << 1) | ((sym >> t) & 1);
            LZ_LV(ix, L) = base + k;
            LZ_LV(bt, L) = (sym >> L) & 1;
        }
        LZ_LANES_END
        rc_encode_lanes(pr, ix, bt, nbits);
    }

    // ----------------------------------------------------------------- prices
    LZ_HD uint32_t price_bit(uint32_t p, uint32_t bit) const {
        return pp[(((p - bit) ^ (0u - bit)) & (kProbTotal - 1)) >> kMoveReducingBits];
    }
    LZ_HD uint32_t price0(uint32_t p) const { return pp[p >> kMoveReducingBits]; }
    LZ_HD uint32_t price1(uint32_t p) const { return pp[(kProbTotal - p) >> kMoveReducingBits]; }
    template <class P>
    LZ_HD uint32_t bt_price(P pr, uint32_t base, uint32_t nbits, uint32_t sym) const {
        uint32_t res = 0, k = 1;
        for (uint32_t i = nbits; i != 0;) {
            i--;
            uint32_t b = (sym >> i) & 1;
            res += price_bit(pr[base + k], b);
            k = (k << 1) + b;
        }
        return res;
    }
    template <class P>
    LZ_HD uint32_t bt_rev_price(P pr, uint32_t base, uint32_t nbits, uint32_t sym) const {
        uint32_t res = 0, k = 1;
        for (uint32_t i = nbits; i != 0; i--) {
            uint32_t b = sym & 1;
            sym >>= 1;
            res += price_bit(pr[base + k], b);
            k = (k << 1) | b;
        }
        return res;
    }

    // ------------------------------------------------- length coder + its cache
    // which: 0 = match lengths (P_LEN), 1 = rep lengths (P_REP_LEN)
    LZ_HD uint32_t len_base(uint32_t which) const { return which ? P_REP_LEN : P_LEN; }
    LZ_HD uint32_t len_tab(uint32_t which) const { return which ? pl.rep_len : pl.len; }
    // setPrices / updateTable, lzma_len_coder.go:66-88,111-114: one symbol per lane
    LZ_HD void len_set_prices(uint32_t which, uint32_t ps) {
        const uint32_t b = len_base(which), nsym = ep.fb + 1 - kMatchMin;
        const uint32_t c0 = U(probs[b + LEN_CHOICE]), c1 = U(probs[b + LEN_CHOICE + 1]);
        const uint32_t a0 = up0(c0), a1 = up1(c0);
        const uint32_t b0 = a1 + up0(c1);
        const uint32_t b1 = a1 + up1(c1);
        auto pr = prices + len_tab(which) + ps * pl.len_stride;
        LZ_PAR_BEGIN(L)
        for (uint32_t i = L; i < nsym; i += kLanes) {
            uint32_t v;
            if (i < kLenLow)
                v = a0 + bt_price(probs, b + LEN_LOW + (ps << 3), 3, i);
            else if (i < kLenLow + kLenMid)
                v = b0 + bt_price(probs, b + LEN_MID + (ps << 3), 3, i - kLenLow);
            else
                v = b1 + bt_price(probs, b + LEN_HIGH, 8, i - kLenLow - kLenMid);
            pr[i] = (WT)v;
        }
        LZ_PAR_END
        if (one()) len_counter[(which << ep.pb) + ps] = nsym;
    }
    LZ_HD void len_encode(uint32_t which, uint32_t sym, uint32_t ps) {  // :48-63, :120-126
        const uint32_t b = len_base(which);
        if (sym < kLenLow) {
            rc_bit(probs, b + LEN_CHOICE, 0);
            bt_encode(probs, b + LEN_LOW + (ps << 3), 3, sym);
        } else {
            uint32_t s = sym - kLenLow;
            rc_bit(probs, b + LEN_CHOICE, 1);
            if (s < kLenMid) {
                rc_bit(probs, b + LEN_CHOICE + 1, 0);
                bt_encode(probs, b + LEN_MID + (ps << 3), 3, s);
            } else {
                rc_bit(probs, b + LEN_CHOICE + 1, 1);
                bt_encode(probs, b + LEN_HIGH, 8, s - kLenMid);
            }
        }
        const uint32_t left_ = U(len_counter[(which << ep.pb) + ps]) - 1;
        if (one()) len_counter[(which << ep.pb) + ps] = left_;
        if (left_ == 0) len_set_prices(which, ps);
    }
    LZ_HD uint32_t len_price(uint32_t which, uint32_t sym, uint32_t ps) const {
        return prices[len_tab(which) + ps * pl.len_stride + sym];
    }

    // ----------------------------------------------------------------- literals
    LZ_HD uint32_t lit_sub(uint32_t pos, uint32_t prev) const {  // lzma_lit_coder.go:126-128
        return (((pos & lit_pos_mask) << ep.lc) + (prev >> (8 - ep.lc))) * 0x300u;
    }
    LZ_HD void lit_encode(uint32_t sub, uint32_t sym) {  // :47-55
        if constexpr (!kBatchLit) {
            uint32_t ctx = 1;
            for (int i = 7; i >= 0; i--) {
                uint32_t b = (sym >> i) & 1;
                rc_bit(lit, sub + ctx, b);
                ctx = (ctx << 1) | b;
            }
            return;
        }
        LZ_LANEVAR(ix);
        LZ_LANEVAR(bt);
        LZ_LANES_BEGIN(L)
        if (L < 8) {
            LZ_LV(ix, L) = sub + ((0x100u | sym) >> (8 - L));
            LZ_LV(bt, L) = (sym >> (7 - L)) & 1;
        }
        LZ_LANES_END
        rc_encode_lanes(lit, ix, bt, 8);
    }
    LZ_HD void lit_encode_matched(uint32_t sub, uint32_t mb, uint32_t sym) {  // :57-76
        if constexpr (!kBatchLit) {
            uint32_t ctx = 1;
            bool same = true;
            for (int i = 7; i >= 0; i--) {
                uint32_t b = (sym >> i) & 1, st = ctx;
                if (same) {
                    uint32_t m = (mb >> i) & 1;
                    st += (1 + m) << 8;
                    same = (m == b);
                }
                rc_bit(lit, sub + st, b);
                ctx = (ctx << 1) | b;
            }
            return;
        }
        LZ_LANEVAR(ix);
        LZ_LANEVAR(bt);
        LZ_LANES_BEGIN(L)
        if (L < 8) {  // in the matched tables while all higher bits equal the match byte's
            const bool same = ((sym ^ mb) >> (8 - L)) == 0;
            const uint32_t m = (mb >> (7 - L)) & 1;
            LZ_LV(ix, L) = sub + ((0x100u | sym) >> (8 - L)) + (same ? (1 + m) << 8 : 0u);
            LZ_LV(bt, L) = (sym >> (7 - L)) & 1;
        }
        LZ_LANES_END
        rc_encode_lanes(lit, ix, bt, 8);
    }
    // getPrice, :78-102, without the data-dependent loop exit: bit i is priced in the
    // matched tables while all higher bits of the symbol equal those of the match byte.
    LZ_HD uint32_t lit_price(uint32_t sub, bool match_mode, uint32_t mb, uint32_t sym) const {
        uint32_t price = 0;
        const uint32_t diff = match_mode ? (sym ^ mb) : 0xFFu;
#pragma unroll
        for (int i = 7; i >= 0; i--) {
            const uint32_t b = (sym >> i) & 1;
            const uint32_t ctx = (0x100u | sym) >> (i + 1);        // 1 followed by the higher bits
            const bool matched = match_mode && (diff >> (i + 1)) == 0;  // higher bits all equal
            const uint32_t m = (mb >> i) & 1;
            const uint32_t idx = matched ? ((1 + m) << 8) + ctx : ctx;
            price += price_bit(lit[sub + idx], b);
        }
        return price;
    }

    // ------------------------------------------------------ window access helpers
    LZ_HD uint32_t avail() const { return n - mfpos; }  // getNumAvailableBytes
    LZ_HD uint32_t index_byte(int32_t index) const { return data[(int64_t)mfpos + index]; }
    LZ_HD uint32_t uindex_byte(int32_t index) const { return U(index_byte(index)); }
    // getMatchLen, lz_window.go:169-182, eight bytes at a time
    LZ_HD uint32_t match_len(int32_t index, uint32_t distance, uint32_t limit) const {
        const uint32_t p = (uint32_t)((int64_t)mfpos + index);
        const uint32_t left = n - p;
        if (limit > left) limit = left;
        const uint8_t *a = data + p;
        const uint8_t *b = a - (int64_t)distance - 1;
        uint32_t r = 0;
        while (r < limit) {
            if (p + r + 8 <= readable) {
                const uint64_t x = enc_load8(a + r) ^ enc_load8(b + r);
                if (x != 0) {
                    r += (uint32_t)__builtin_ctzll(x) >> 3;
                    break;
                }
                r += 8;
            } else {
                if (a[r] != b[r]) break;
                r++;
            }
        }
        return r < limit ? r : limit;
    }

    LZ_HD uint32_t read_match_distances() {  // lzma_encoder.go:274-285
        uint32_t res = 0;
        dist_pairs = 2 * U(mf(mfpos, mdist));
        mfpos++;
        if (dist_pairs > 0) {
            res = U(mdist[dist_pairs - 2]);
            if (res == ep.fb)
                res += U(match_len((int32_t)res - 1, U(mdist[dist_pairs - 1]), kMatchMax - res));
        }
        add_offset++;
        return res;
    }
    LZ_HD void move_pos(uint32_t num) {  // :287-292; skip() has no observable effect here
        add_offset += num;
        mfpos += num;
    }

#define LZ_IDX(st, ps) (((st) << kPosStatesBitsMax) + (ps))
    LZ_HD uint32_t pure_rep_price(uint32_t ri, uint32_t st, uint32_t ps) const {  // :294-308
        uint32_t p;
        if (ri == 0) {
            p = price0(probs[P_IS_REP_G0 + st]);
            p += price1(probs[P_IS_REP0_LONG + LZ_IDX(st, ps)]);
        } else {
            p = price1(probs[P_IS_REP_G0 + st]);
            if (ri == 1)
                p += price0(probs[P_IS_REP_G1 + st]);
            else {
                p += price1(probs[P_IS_REP_G1 + st]);
                p += price_bit(probs[P_IS_REP_G2 + st], ri - 2);
            }
        }
        return p;
    }
    LZ_HD uint32_t rep_price(uint32_t ri, uint32_t len, uint32_t st, uint32_t ps) const {
        return len_price(1, len - kMatchMin, ps) + pure_rep_price(ri, st, ps);
    }
    LZ_HD uint32_t pos_len_price(uint32_t pos, uint32_t len, uint32_t ps) const {  // :316-325
        uint32_t lps = len_to_pos_state(len), p;
        if (pos < kNumFullDist)
            p = prices[pl.dist + lps * kNumFullDist + pos];
        else
            p = (uint32_t)prices[pl.pos_slot + (lps << kPosSlotBits) + pos_slot(pos)] +
                prices[pl.align + (pos & kAlignMask)];
        return p + len_price(0, len - kMatchMin, ps);
    }
    LZ_HD uint32_t rep_len1_price(uint32_t st, uint32_t ps) const {  // :327-329
        return price0(probs[P_IS_REP_G0 + st]) + price0(probs[P_IS_REP0_LONG + LZ_IDX(st, ps)]);
    }

    // backward (:331-357) walks the whole path, most of which has left the window:
    // it works on the HBM copy alone.  The window is not used again before the next
    // parse starts from entry 0.
    LZ_HD uint32_t backward(uint32_t cur) {
        OptG *o = opt_g;
        opt_end = cur;
        uint32_t w = U(o[cur].rec.pp);
        uint32_t pos_mem = w & 0xFFF, back_mem = U(o[cur].rec.back);
        do {
            const uint32_t w_v = o[cur].rec.pp, b2_v = o[cur].rec.back2;
            const uint32_t pm_v = o[pos_mem].rec.pp, pm1_v = o[pos_mem ? pos_mem - 1 : 0].rec.pp;
            const uint32_t pmb_v = o[pos_mem].rec.back;
            w = U(w_v);
            uint32_t wp = U(pm_v), bp = U(pmb_v);
            if (w & kPpPrev1) {
                bp = 0xFFFFFFFFu;
                wp = (wp & ~(0xFFFu | kPpPrev1)) | (pos_mem - 1);
                if ((w & kPpPrev2) && one()) {
                    o[pos_mem - 1].rec.pp = (U(pm1_v) & ~(0xFFFu | kPpPrev1)) | ((w >> 12) & 0xFFF);
                    o[pos_mem - 1].rec.back = U(b2_v);
                }
            }
            const uint32_t pprev = pos_mem, bc = back_mem;
            back_mem = bp;
            pos_mem = wp & 0xFFF;
            if (one()) {
                o[pprev].rec.back = bc;
                o[pprev].rec.pp = (wp & ~0xFFFu) | cur;
            }
            cur = pprev;
        } while (cur > 0);
        back_res = U(o[0].rec.back);
        opt_cur = U(o[0].rec.pp) & 0xFFF;
        return opt_cur;
    }

    // Uniform inputs of the apply phase, held in registers.
    struct ApplyIn {
        uint32_t cur, old_end, new_end, ps, start_len, new_len, nmp, npairs, first_la;
        uint32_t rl[4], prp[4];  // plain rep candidates: length (< 2 = none), price without length
        uint32_t lt[5], lp[5];   // continuation of the literal [0] and of rep i [1+i]: target (0 = none), price
    };

    // The apply phase (see the header comment): entries cur+2 .. new_end.
    //   old_end: entries beyond it are new and start at infinity
    //   plain match candidates: lengths start_len..new_len over mdist[0..npairs) (new_len 0 = none)
    //   continuations of pairs first_la..npairs-1 are in la_tgt/la_price[5+j];
    //   the path words of a continuation follow from the step it continues.
    LZ_HD void apply(const ApplyIn &A, bool with_la) {
        const uint32_t cur = A.cur, ps = A.ps;
        const uint32_t rl_max = max4(A.rl[0], A.rl[1], A.rl[2], A.rl[3]);
        LZ_PAR_BEGIN(L)
        for (uint32_t x = cur + 2 + L; x <= A.new_end; x += kLanes) {
            const uint32_t len = x - cur;
            uint32_t price = kInfPrice;
            if (x <= A.old_end) price = oprice(x);
            // every table value this entry may need, fetched together
            uint32_t rep_lp = 0, m_price = 0, m_dist = 0;
            if (len <= rl_max) rep_lp = len_price(1, len - kMatchMin, ps);
            const bool plain = A.new_len != 0 && len >= A.start_len && len <= A.new_len;
            if (plain) {
                uint32_t j = 0;
                while (mdist[2 * j] < len) j++;
                m_dist = mdist[2 * j + 1];
                m_price = A.nmp + pos_len_price(m_dist, len, ps);
            }
            bool changed = x > A.old_end;
            uint32_t back = 0, ppw = 0, back2 = 0;
            bool is_la = false;
            if (with_la && A.lt[0] == x && A.lp[0] < price) {
                price = A.lp[0], back = 0, ppw = (cur + 1) | kPpPrev1, back2 = 0, is_la = true, changed = true;
            }
#pragma unroll
            for (uint32_t ri = 0; ri < kNumReps; ri++) {
                const uint32_t lt = A.rl[ri];
                if (lt >= 2 && len <= lt) {
                    const uint32_t p = A.prp[ri] + rep_lp;
                    if (p < price) price = p, back = ri, ppw = cur, is_la = false, changed = true;
                }
                if (with_la && A.lt[1 + ri] == x && A.lp[1 + ri] < price) {
                    price = A.lp[1 + ri], back = 0, back2 = ri;
                    ppw = (cur + lt + 1) | (cur << 12) | kPpPrev1 | kPpPrev2;
                    is_la = true, changed = true;
                }
            }
            if (with_la)
                for (uint32_t j = A.first_la; j < A.npairs; j++)
                    if (la_tgt[j] == x && la_price[j] < price) {
                        price = la_price[j], back = 0, back2 = mdist[2 * j + 1] + kNumReps;
                        ppw = (cur + mdist[2 * j] + 1) | (cur << 12) | kPpPrev1 | kPpPrev2;
                        is_la = true, changed = true;
                    }
            if (plain && m_price < price)
                price = m_price, back = m_dist + kNumReps, ppw = cur, is_la = false, changed = true;
            if (changed) {  // (backPrev2 only counts with the flags of a continuation set)
                Rec r;
                r.price = price, r.back = back, r.pp = ppw, r.back2 = back2;
                srec(x, r);
            }
        }
        LZ_PAR_END
    }
    // The same with the pairs' lengths, continuation targets and continuation prices taken
    // from the lanes that computed them in the round (lane 8 + j: pair j; lists of up to
    // kLanes - 8 pairs): each entry's lane counts the pairs shorter than its length and
    // keeps the cheapest continuation landing on it while the pairs go by one at a time,
    // with no trip through LDS per pair.  Candidates are compared in the order of apply().
    LZ_HD void apply_lanes(const ApplyIn &A, LZ_LANEARG(pl_), LZ_LANEARG(tg_), LZ_LANEARG(pr_)) {
        const uint32_t cur = A.cur, ps = A.ps;
        const uint32_t rl_max = max4(A.rl[0], A.rl[1], A.rl[2], A.rl[3]);
        for (uint32_t xb = cur + 2; xb <= A.new_end; xb += kLanes) {
            LZ_LANEVAR(cnt);
            LZ_LANEVAR(bp);
            LZ_LANEVAR(bj);
            LZ_LANES_BEGIN(L)
            (void)L;
            LZ_LV(cnt, L) = 0, LZ_LV(bp, L) = kInfPrice, LZ_LV(bj, L) = 0;
            LZ_LANES_END
            for (uint32_t j = 0; j < A.npairs; j++) {
                const uint32_t pl = LZ_FROM_LANE(pl_, 8 + j), tg = LZ_FROM_LANE(tg_, 8 + j);
                const uint32_t pr = LZ_FROM_LANE(pr_, 8 + j);
                const bool la = j >= A.first_la && tg != 0;
                LZ_LANES_BEGIN(L)
                const uint32_t x = xb + L;
                if (pl < x - cur) LZ_LV(cnt, L)++;
                if (la && tg == x && pr < LZ_LV(bp, L)) LZ_LV(bp, L) = pr, LZ_LV(bj, L) = j;
                LZ_LANES_END
            }
            LZ_PAR_BEGIN(L)
            const uint32_t x = xb + L;
            if (x <= A.new_end) {
                const uint32_t len = x - cur;
                uint32_t price = kInfPrice;
                if (x <= A.old_end) price = oprice(x);
                uint32_t rep_lp = 0, m_price = 0, m_dist = 0;
                if (len <= rl_max) rep_lp = len_price(1, len - kMatchMin, ps);
                const bool plain = A.new_len != 0 && len >= A.start_len && len <= A.new_len;
                if (plain) {
                    m_dist = mdist[2 * LZ_LV(cnt, L) + 1];
                    m_price = A.nmp + pos_len_price(m_dist, len, ps);
                }
                bool changed = x > A.old_end;
                uint32_t back = 0, ppw = 0, back2 = 0;
                bool is_la = false;
                if (A.lt[0] == x && A.lp[0] < price) {
                    price = A.lp[0], back = 0, ppw = (cur + 1) | kPpPrev1, back2 = 0, is_la = true, changed = true;
                }
#pragma unroll
                for (uint32_t ri = 0; ri < kNumReps; ri++) {
                    const uint32_t lt = A.rl[ri];
                    if (lt >= 2 && len <= lt) {
                        const uint32_t p = A.prp[ri] + rep_lp;
                        if (p < price) price = p, back = ri, ppw = cur, is_la = false, changed = true;
                    }
                    if (A.lt[1 + ri] == x && A.lp[1 + ri] < price) {
                        price = A.lp[1 + ri], back = 0, back2 = ri;
                        ppw = (cur + lt + 1) | (cur << 12) | kPpPrev1 | kPpPrev2;
                        is_la = true, changed = true;
                    }
                }
                if (LZ_LV(bp, L) < price) {
                    const uint32_t j = LZ_LV(bj, L);
                    price = LZ_LV(bp, L), back = 0, back2 = mdist[2 * j + 1] + kNumReps;
                    ppw = (cur + mdist[2 * j] + 1) | (cur << 12) | kPpPrev1 | kPpPrev2;
                    is_la = true, changed = true;
                }
                if (plain && m_price < price)
                    price = m_price, back = m_dist + kNumReps, ppw = cur, is_la = false, changed = true;
                if (changed) {  // (backPrev2 only counts with the flags of a continuation set)
                    Rec r;
                    r.price = price, r.back = back, r.pp = ppw, r.back2 = back2;
                    srec(x, r);
                }
            }
            LZ_PAR_END
        }
    }
    static LZ_HD uint32_t max4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
        const uint32_t x = a > b ? a : b, y = c > d ? c : d;
        return x > y ? x : y;
    }

    // One lane's task in a round of the parse (see get_optimum): which two byte streams it
    // compares (a at a_off, b `dist + 1` before it) and how far.
    struct RoundTask {
        bool is_rep, is_lit, is_pair, active;
        uint32_t a_off, dist, lim1, lim2, lt_pair;
    };
    LZ_HD RoundTask round_task(uint32_t L, uint32_t jbase, bool do_rest, uint32_t npairs, uint32_t r0, uint32_t r1,
                               uint32_t r2, uint32_t r3, uint32_t av, uint32_t avail_full) const {
        const uint32_t fb = ep.fb;
        const bool first = jbase == 0;
        const uint32_t j = jbase + L - 8;  // pair index of lanes >= 8
        RoundTask t;
        t.is_rep = first && L < kNumReps, t.is_lit = first && L == 4;
        t.is_pair = do_rest && L >= 8 && j < npairs;
        t.a_off = mfpos - 1, t.dist = 0, t.lim1 = 0, t.lim2 = 0, t.lt_pair = 0;
        if (t.is_rep) {
            t.dist = L == 0 ? r0 : (L == 1 ? r1 : (L == 2 ? r2 : r3));
            if (do_rest) t.lim1 = av;
        } else if (t.is_lit) {
            t.dist = r0;
            if (do_rest) t.lim2 = avail_full - 1 < fb ? avail_full - 1 : fb;
        } else if (t.is_pair) {
            t.lt_pair = mdist[2 * j], t.dist = mdist[2 * j + 1];
            t.a_off += t.lt_pair;
            if (t.lt_pair < avail_full) t.lim2 = avail_full - 1 - t.lt_pair < fb ? avail_full - 1 - t.lt_pair : fb;
        }
        t.active = t.is_rep || t.is_lit || (t.is_pair && t.lim2 != 0);
        // a cooperative wave may run a position whose entry nothing was written to (parse_loop):
        // its stale distances must not be followed out of the buffer
        if (COOP && t.dist >= t.a_off) t.active = false;
        return t;
    }

    LZ_HD uint32_t get_optimum(uint32_t position) {  // :359-797
        const uint32_t fb = ep.fb;
        if (opt_end != opt_cur) {
            const uint32_t pp_v = opt_g[opt_cur].rec.pp, bk_v = opt_g[opt_cur].rec.back;
            const uint32_t nxt = U(pp_v) & 0xFFF;
            const uint32_t r = nxt - opt_cur;
            back_res = U(bk_v);
            opt_cur = nxt;
            return r;
        }
        opt_end = opt_cur = 0;
        wlo = 0, wcur = 0, far_hi = 0;
        uint32_t len_main;
        if (!longest_found)
            len_main = read_match_distances();
        else {
            len_main = longest_len;
            longest_found = 0;
        }
        uint32_t dpairs = dist_pairs;
        if (avail() + 1 < 2) {
            back_res = 0xFFFFFFFFu;
            return 1;
        }
        // rep lengths and the two bytes the literal needs, in parallel: lanes 0-3 and lane 4
        // hand their results over in x0 / x1
        LZ_LANEVAR(x0);
        LZ_LANEVAR(x1);
        LZ_PAR_BEGIN(L)
        if (L < kNumReps) {
            LZ_LV(x0, L) = match_len(-1, rep_dist[L], kMatchMax);
        } else if (L == 4) {
            LZ_LV(x0, L) = index_byte(-1);
            LZ_LV(x1, L) = index_byte(0 - (int32_t)rep_dist[0] - 1 - 1);
        }
        LZ_PAR_END
        uint32_t r0 = U(rep_dist[0]), r1 = U(rep_dist[1]), r2 = U(rep_dist[2]), r3 = U(rep_dist[3]);  // z.reps
        ApplyIn A;
        A.rl[0] = LZ_FROM_LANE(x0, 0), A.rl[1] = LZ_FROM_LANE(x0, 1);
        A.rl[2] = LZ_FROM_LANE(x0, 2), A.rl[3] = LZ_FROM_LANE(x0, 3);
        // (no indexing by a variable: that would put the whole struct in scratch memory)
        uint32_t rep_max = 0, rep_max_len = A.rl[0];
#pragma unroll
        for (uint32_t i = 1; i < kNumReps; i++)
            if (A.rl[i] > rep_max_len) rep_max = i, rep_max_len = A.rl[i];
        if (rep_max_len >= fb) {
            back_res = rep_max;
            move_pos(rep_max_len - 1);
            return rep_max_len;
        }
        if (len_main >= fb) {
            back_res = U(mdist[dpairs - 1]) + kNumReps;
            move_pos(len_main - 1);
            return len_main;
        }
        const uint32_t cur_byte0 = LZ_FROM_LANE(x0, 4), match_byte0 = LZ_FROM_LANE(x1, 4);
        if (len_main < 2 && cur_byte0 != match_byte0 && rep_max_len < 2) {
            back_res = 0xFFFFFFFFu;
            return 1;
        }
        uint32_t ps = position & pos_state_mask;
        uint32_t rep_len1;
        uint32_t rep_match_price, nmp;
        sp_refresh();
        {
            const StatePrices sp0 = sp_get(state, ps);
            const uint32_t im0 = sp0.im0;
            rep_match_price = sp0.rep_match;
            nmp = sp0.nmp;
            A.prp[0] = sp0.prp[0], A.prp[1] = sp0.prp[1], A.prp[2] = sp0.prp[2], A.prp[3] = sp0.prp[3];
            rep_len1 = sp0.rep_len1;
            uint32_t price1_ = im0 + U(lit_price(lit_sub(position, prev_byte), !st_is_char(state),
                                                 match_byte0, cur_byte0));
            uint32_t back1 = 0xFFFFFFFFu;
            if (match_byte0 == cur_byte0) {
                const uint32_t sp = rep_match_price + rep_len1;
                if (sp < price1_) {
                    price1_ = sp;
                    back1 = 0;
                }
            }
            uint32_t len_end0 = rep_max_len;
            if (len_main > len_end0) len_end0 = len_main;
            if (len_end0 < 2) {
                back_res = back1;
                return 1;
            }
            {
                Rec r0_;
                r0_.price = 0, r0_.back = 0, r0_.pp = state << 24, r0_.back2 = 0;
                commit_rec(0, r0_);
            }
            Reps b0;
            b0.r[0] = r0, b0.r[1] = r1, b0.r[2] = r2, b0.r[3] = r3;
            u_sbks(0, b0);
            u_set_pbp(1, price1_, back1, 0);
        }
        uint32_t len_end = rep_max_len;
        if (len_main > len_end) len_end = len_main;
        {
            A.cur = 0, A.old_end = 1, A.new_end = len_end, A.ps = ps;
            A.start_len = A.rl[0] >= 2 ? A.rl[0] + 1 : 2;
            A.new_len = A.start_len <= len_main ? U(mdist[dpairs - 2]) : 0;
            A.nmp = nmp, A.npairs = dpairs >> 1, A.first_la = 0;
            for (uint32_t i = 0; i < 5; i++) A.lt[i] = 0, A.lp[i] = 0;
            apply(A, false);
            note_far(len_end);
        }

        LoopExit ex;
        if constexpr (COOP) {
            // hand the loop to both waves: this one takes the odd positions
            const uint32_t mf0 = mfpos, add0 = add_offset;
            lz_sy_store(sync + SY_EXIT, 0, one());  // (the waiting wave polls this word)
            if (one()) {
                sync[SY_S] = 0, sync[SY_A] = 0, sync[SY_M] = 0;
                sync[SY_LEN_END] = len_end, sync[SY_FAR_HI] = far_hi;
                sync[SY_POS0] = position, sync[SY_MF0] = mf0;
            }
            const uint32_t seq = lz_sy_load(sync + SY_FORK) + 1;
            sy_signal(SY_FORK, seq);
            const bool mine = parse_loop(1, position, mf0, len_end, ex);
            LZ_LAP(6)
            (void)sy_wait_word(SY_JOIN, seq, false);
            if constexpr (NW > 2) (void)sy_wait_word(SY_JOIN2, seq, false);
            LZ_LAP(15)
            if (dead) {  // the waves lost each other: end the chunk, which reports the error
                back_res = 0xFFFFFFFFu;
                return 1;
            }
            if (!mine) {
                const uint32_t c_v = sync[SY_EX_CUR], l_v = sync[SY_EX_LONG], n_v = sync[SY_EX_LEN], d_v = sync[SY_EX_DIST];
                ex.cur = U(c_v), ex.long_match = U(l_v), ex.longest_len = U(n_v), ex.last_dist = U(d_v);
            }
            // the match finder has been asked for the positions up to the last one parsed
            const uint32_t reads = ex.long_match ? ex.cur : ex.cur - 1;
            mfpos = mf0 + reads, add_offset = add0 + reads;
            if (ex.long_match) {  // the next parse starts from this match
                longest_len = ex.longest_len, longest_found = 1;
                dist_pairs = 2;
                if (one()) mdist[0] = ex.longest_len, mdist[1] = ex.last_dist;
            }
        } else {
            parse_loop(1, position, mfpos, len_end, ex);
        }
        commit_from_window(ex.cur);
        return backward(ex.cur);
    }

    // How the main loop of a parse ended: at the end of the parse range (cur == lenEnd, :509-511)
    // or at a position whose longest match has at least fb bytes (:517-522).
    struct LoopExit {
        uint32_t cur, long_match, longest_len, last_dist;
    };
    // second wave of a cooperative workgroup: takes part in every parse loop wave 0 starts,
    // until wave 0 reports the chunk finished
    LZ_HD void helper_run(U32P small, U32P own, uint32_t wave) {  // wave: 1 or 2
        bind_small(small, ep.fb, ep.pb);
        la_tgt = own, la_price = own + la_max(ep.fb), mdist = own + la_words(ep.fb);
        pos_state_mask = (1u << ep.pb) - 1;
        lit_pos_mask = (1u << ep.lp) - 1;
        pl = price_layout(ep.pb, ep.fb);
        uint32_t seq = 0;
        dead = 0;
        for (;;) {
            // (wave 0 codes the symbols of a parse between two loops: this wait is the long one)
            LZ_LAP(6)
            const uint32_t f = sy_wait_word(SY_FORK, seq, true);
            LZ_LAP(15)
            if (dead || f == kSyDone) return;
            seq = f;
            // (the end of the parse range and the far mark are read with every apply phase's
            // hand-over, never before: by now the first wave may have moved them on)
            const uint32_t p_v = sync[SY_POS0], m_v = sync[SY_MF0];
            far_hi = 0;
            LoopExit ex;
            (void)parse_loop(1 + wave, U(p_v), U(m_v), 0, ex);
            if (dead) return;
            sy_signal(wave == 1 ? SY_JOIN : SY_JOIN2, seq);
        }
    }
    LZ_HD void publish_exit(const LoopExit &ex) {
        if (one()) {
            sync[SY_EX_CUR] = ex.cur, sync[SY_EX_LONG] = ex.long_match;
            sync[SY_EX_LEN] = ex.longest_len, sync[SY_EX_DIST] = ex.last_dist;
        }
        sy_signal(SY_EXIT, 1);
    }

    // Literal and short rep of position cur into entry cur + 1 (:598-624): nxr is that entry
    // before, the result says what to write and whether the literal won (nextIsChar).
    struct NextUpd {
        uint32_t price, back, pp;
        bool changed, next_is_char;
    };
    LZ_HD NextUpd merge_next(const Rec &nxr, uint32_t cur, uint32_t cur_and1, uint32_t short_rep, bool same_byte) const {
        NextUpd u;
        u.price = nxr.price, u.back = 0, u.pp = 0, u.changed = false, u.next_is_char = false;
        if (cur_and1 < u.price) {
            u.price = cur_and1, u.back = 0xFFFFFFFFu, u.pp = cur;
            u.next_is_char = true, u.changed = true;
        }
        if (same_byte) {
            bool blocked;
            if (u.changed)
                blocked = false;  // posPrev == cur is not < cur
            else
                blocked = (nxr.pp & 0xFFF) < cur && nxr.back == 0;
            if (!blocked && short_rep <= u.price) {
                u.price = short_rep, u.back = 0, u.pp = cur;
                u.next_is_char = true, u.changed = true;
            }
        }
        return u;
    }

    // The main loop of a parse (:507-795), positions first_cur, first_cur + step, ...: every
    // position for a wave of its own, every other one in a cooperative workgroup.
    //
    // Several waves (NW = 2 or 3), wave w taking positions w + 1, w + 1 + NW, ...  What position
    // c does falls into three parts:
    //   S(c)  entry c made final, state and reps of the path into c, state prices, and the
    //         literal / short-rep candidate for entry c + 1, priced by eight lanes and published;
    //   R(c)  the match list and the round (rep lengths, continuations);
    //   A(c)  the apply phase over entries c + 2 and beyond.
    // S(c) needs the candidate that S(c - 1) published -- this wave merges it into entry c
    // itself, after the apply phases up to A(c - 2) (it comes after their candidates in the
    // reference's order; A(c - 1), still running, does not touch entry c) -- so the S parts form
    // a chain from wave to wave, and so do the A parts (A(c) changes the entries A(c - 1)
    // changes); R(c) and A(c) of one position run beside S, R and A of the next ones on the other
    // waves.  A(c) also waits for the merge of its own candidate (done by the wave at c + 1),
    // which says whether the literal won (nextIsChar, :601-604 and :635).
    //   The end of the parse range (cur == lenEnd, :509) is known only when A(c - 1) is done, after
    // S(c) and R(c) have run: a wave works on a position before it is known to be parsed at all,
    // and the wave at c + 1 may even start from an entry nothing was written to in this parse
    // (if the parse ends at c).  Such work changes nothing that is read again -- entries beyond
    // the end are never looked at, the window and the models are left alone -- and the few places
    // where a stale entry could lead out of bounds are guarded (round_task, the path entry, the
    // state).  Returns true in the wave that found the end (ex is set), false in the others.
    LZ_HD bool parse_loop(uint32_t first_cur, const uint32_t pos0, const uint32_t mf0, uint32_t len_end,
                          LoopExit &ex) {
        const uint32_t fb = ep.fb;
        constexpr uint32_t kStep = NW;
        LZ_LANEVAR(x0);
        LZ_LANEVAR(x1);
        LZ_LANEVAR(x2);
        LZ_LANEVAR(x3);
        ApplyIn A;
        uint32_t r0, r1, r2, r3, ps, rep_len1, rep_match_price, nmp;
        uint32_t cur = first_cur - kStep;
        for (;;) {
            cur += kStep;
            Rec pend;
            pend.price = pend.back = pend.pp = pend.back2 = 0;
            uint32_t pend_slot = kNoSlot;
            if constexpr (!COOP) {
                if (cur == len_end) {
                    ex.cur = cur, ex.long_match = 0, ex.longest_len = 0, ex.last_dist = 0;
                    return true;
                }
                pend_slot = win_advance(cur, pend);
            } else {
                win_set(cur);
                mfpos = mf0 + cur - 1;
            }
            LZ_LAP(5)
            // COOP: a wave may get here for a position beyond the end of the chunk or of the
            // optimum table; the parse then ends at this position at the latest
            const bool beyond = COOP && (mfpos >= n || cur >= kNumOpts - 1);
            const uint32_t position = pos0 + cur;
            // COOP: what the literal's price needs from the data alone is fetched by lanes 0..7
            // before anything is waited for: the byte, the byte before it, and each lane's model
            // for its bit of the byte when the literal is coded without a match byte
            LZ_LANEVAR(cb8);
            LZ_LANEVAR(sub8);
            LZ_LANEVAR(np8);
            if constexpr (COOP) {
                if (!beyond) {
                    LZ_LANES_BEGIN(L)
                    if (L < 8) {
                        const uint8_t *a = data + mfpos;
                        const uint32_t cb = a[0], sub = lit_sub(position, a[-1]);
                        LZ_LV(cb8, L) = cb, LZ_LV(sub8, L) = sub;
                        LZ_LV(np8, L) = lit[sub + ((0x100u | cb) >> (8 - L))];
                    }
                    LZ_LANES_END
                }
            }
            uint32_t new_len = 0;
            Rec rc_, nxr;
            // three waves have time to spare between their positions: the match list is read
            // before the wave waits; with two, after the merge the other wave waits for
            constexpr bool kListFirst = NW > 2;
            if constexpr (kListFirst) {
                if (!beyond) new_len = read_match_distances();
                LZ_LAP(0)
            }
            if constexpr (COOP) {
                if (!sy_wait(SY_S, cur - 1)) return false;
                // entry cur is final with the apply phases up to cur - 2: this wave's own with two
                // waves, another wave's with three
                if constexpr (NW > 2) {
                    if (cur >= 3 && !sy_wait(SY_A, cur - 2)) return false;
                }
                LZ_LAP(12)
                const Rec rc_v = o_rec[cur & (kRecWin - 1)];
                rc_ = urec(rc_v);
                if (cur > 1) {  // the literal candidate of cur - 1 (entry 1 was set before the loop)
                    const uint32_t par = (cur - 1) & 1;
                    const uint32_t a_v = sync[SY_PKT_PRICE + par], s_v = sync[SY_PKT_SP + par], q_v = sync[SY_PKT_EQ + par];
                    const NextUpd u = merge_next(rc_, cur - 1, U(a_v), U(s_v), U(q_v) != 0);
                    if (u.changed) {
                        rc_.price = u.price, rc_.back = u.back, rc_.pp = u.pp;
                        u_set_pbp(cur, u.price, u.back, u.pp);
                    }
                    if (one()) sync[SY_NIC + par] = u.next_is_char;
                    sy_signal(SY_M, cur - 1, false);
                }
                LZ_LAP(8)
            }
            if constexpr (!kListFirst) {
                if (!beyond) new_len = read_match_distances();
                if constexpr (!COOP) win_land(pend_slot, pend);
                LZ_LAP(0)
            }
            uint32_t dpairs = beyond ? 0 : dist_pairs;
            if constexpr (COOP) {
                if (beyond || new_len >= fb) {
                    if (!sy_wait(SY_A, cur - 1)) return false;
                    const uint32_t e_v = sync[SY_LEN_END];
                    const bool at_end = cur == U(e_v);  // comes first in the reference's order
                    ex.cur = cur, ex.long_match = !at_end, ex.longest_len = new_len;
                    ex.last_dist = at_end ? 0 : U(mdist[dpairs - 1]);
                    publish_exit(ex);
                    return true;
                }
            } else {
                if (new_len >= fb) {
                    longest_len = new_len;
                    longest_found = 1;
                    ex.cur = cur, ex.long_match = 1, ex.longest_len = new_len, ex.last_dist = 0;
                    return true;
                }
            }
            // what is left of the chunk and of the optimum table bounds every length from here (:627, :724-731)
            uint32_t avail_full = kNumOpts - 1 - cur;
            if (avail() + 1 < avail_full) avail_full = avail() + 1;
            const uint32_t av = avail_full < fb ? avail_full : fb;
            const bool do_rest = avail_full >= 2;
            uint32_t npairs = dpairs >> 1;
            if (do_rest && new_len > av) {  // :724-731
                new_len = av;
                uint32_t j = 0;
                while (new_len > U(mdist[2 * j])) j++;
                if (one()) mdist[2 * j] = new_len;
                npairs = j + 1;
            }
            // ---- state and reps of the path that reaches cur (:525-597).  The entry E the
            // path comes from always holds the reps to permute (for a literal or short rep it
            // is cur-1, whose reps are the current ones).
            if constexpr (!COOP) {
                const Rec rc_v = o_rec[cur & (kRecWin - 1)], nxr_v = o_rec[(cur + 1) & (kRecWin - 1)];
                rc_ = urec(rc_v), nxr = urec(nxr_v);
            }
            const uint32_t cw = rc_.pp;
            uint32_t pos_prev = cw & 0xFFF;
            const bool c_p1 = (cw & kPpPrev1) != 0, c_p2 = (cw & kPpPrev2) != 0;
            const uint32_t c_bp = rc_.back;
            uint32_t E = (c_p1 && c_p2) ? ((cw >> 12) & 0xFFF) : (c_p1 ? pos_prev - 1 : pos_prev);
            if constexpr (COOP) E &= kNumOpts - 1;  // (a stale entry, see round_task)
            const uint32_t e_pp_v = opp(E);
            const Reps eb_v = gbks(E);
            const uint32_t e_pp = U(e_pp_v);
            const Reps eb = ureps(eb_v);
            LZ_LAP(8)
            uint32_t st = (e_pp >> 24) & 0xF;
            if constexpr (COOP) st = st < kNumStates ? st : kNumStates - 1;
            if (c_p1) {
                pos_prev--;
                if (c_p2) st = rc_.back2 < kNumReps ? st_rep(st) : st_match(st);
                st = st_char(st);
            }
            if (pos_prev == cur - 1) {
                st = c_bp == 0 ? st_shortrep(st) : st_char(st);
                r0 = eb.r[0], r1 = eb.r[1], r2 = eb.r[2], r3 = eb.r[3];
            } else {
                uint32_t pos;
                if (c_p1 && c_p2) {
                    pos = rc_.back2;
                    st = st_rep(st);
                } else {
                    pos = c_bp;
                    st = pos < kNumReps ? st_rep(st) : st_match(st);
                }
                const uint32_t q0 = eb.r[0], q1 = eb.r[1], q2 = eb.r[2], q3 = eb.r[3];
                if (pos < kNumReps) {
                    if (pos == 0) {
                        r0 = q0, r1 = q1, r2 = q2, r3 = q3;
                    } else if (pos == 1) {
                        r0 = q1, r1 = q0, r2 = q2, r3 = q3;
                    } else if (pos == 2) {
                        r0 = q2, r1 = q0, r2 = q1, r3 = q3;
                    } else {
                        r0 = q3, r1 = q0, r2 = q1, r3 = q2;
                    }
                } else {
                    r0 = pos - kNumReps, r1 = q0, r2 = q1, r3 = q2;
                }
            }
            {
                Rec fr = rc_;
                fr.pp = (cw & ~(0xFu << 24)) | (st << 24);
                commit_rec(cur, fr);
            }
            {
                Reps nb;
                nb.r[0] = r0, nb.r[1] = r1, nb.r[2] = r2, nb.r[3] = r3;
                u_sbks(cur, nb);
            }
            const uint32_t cur_price = rc_.price;
            ps = position & pos_state_mask;
            // every price that depends only on (state, posState): models in one batch, then the
            // price table in one batch
            uint32_t p_im0;
            {
                const StatePrices spc = sp_get(st, ps);
                p_im0 = spc.im0;
                rep_match_price = cur_price + spc.rep_match;
                nmp = cur_price + spc.nmp;
                A.prp[0] = cur_price + spc.prp[0], A.prp[1] = cur_price + spc.prp[1];
                A.prp[2] = cur_price + spc.prp[2], A.prp[3] = cur_price + spc.prp[3];
                rep_len1 = spc.rep_len1;
            }
            LZ_LAP(1)
            uint32_t cur_and1 = 0;
            const uint32_t short_rep = rep_match_price + rep_len1;
            if constexpr (COOP) {
                // The other wave waits for this position's literal / short-rep candidate and for
                // nothing else of it: that comes first, the literal's eight bits priced by
                // eight lanes, and goes out before the round.
                LZ_LANEVAR(lp8);
                LZ_LANEVAR(eq8);
                const uint32_t a_off = mfpos - 1;
                const bool mode = !st_is_char(st);
                LZ_LANES_BEGIN(L)
                if (L < 8) {
                    const uint32_t cb = LZ_LV(cb8, L);
                    const uint32_t mb = r0 < a_off ? data[a_off - r0 - 1] : 0;
                    const uint32_t i = 7 - L;
                    uint32_t model = LZ_LV(np8, L);
                    if (mode && ((cb ^ mb) >> (i + 1)) == 0)  // the higher bits equal the match byte's
                        model = lit[LZ_LV(sub8, L) + ((1 + ((mb >> i) & 1)) << 8) + ((0x100u | cb) >> (i + 1))];
                    LZ_LV(lp8, L) = price_bit(model, (cb >> i) & 1);
                    LZ_LV(eq8, L) = cb == mb;
                }
                LZ_LANES_END
                cur_and1 = cur_price + p_im0 + LZ_SUM8(lp8);
                const uint32_t same = LZ_FROM_LANE(eq8, 0);
                if (one()) {
                    sync[SY_PKT_PRICE + (cur & 1)] = cur_and1, sync[SY_PKT_SP + (cur & 1)] = short_rep;
                    sync[SY_PKT_EQ + (cur & 1)] = same;
                }
                sy_signal(SY_S, cur, false);
            }
            LZ_LAP(9)
            // ---- the round: every lane runs the same code on its own task.
            //   lanes 0-3: rep match L (:661-663) and its continuation (:689-721)
            //   lane 4:    the literal (:599-603) and rep0 after it (:635-638)
            //   lanes 8..: continuation of pair L-8 of the match list (:753-787)
            // A task is two byte streams a (here) and b (dist+1 back): s1 equal bytes, then a
            // literal, then s2 equal bytes.  Sixteen bytes of both are fetched up front, which
            // covers nearly every task in one memory round trip.
            LZ_PAR_BEGIN(L)
            for (uint32_t jbase = 0; jbase == 0 || jbase < npairs; jbase += kLanes - 8) {
                const bool first = jbase == 0;
                const uint32_t j = jbase + L - 8;  // pair index of lanes >= 8
                const RoundTask t = round_task(L, jbase, do_rest, npairs, r0, r1, r2, r3, av, avail_full);
                const bool is_rep = t.is_rep, is_lit = t.is_lit, active = t.active;
                const uint32_t a_off = t.a_off, dist = t.dist, lim1 = t.lim1, lt_pair = t.lt_pair;
                uint32_t lim2 = t.lim2, rprp = 0;
                if (is_rep) rprp = L == 0 ? A.prp[0] : (L == 1 ? A.prp[1] : (L == 2 ? A.prp[2] : A.prp[3]));
                uint32_t s1 = 0, s2 = 0, cur_b = 0, match_b = 0, prev_b = 0;
                bool tail = false;
                if (active) {
                    const uint8_t *a = data + a_off;
                    const uint8_t *b = a - (int64_t)dist - 1;
                    const uint32_t left = n - a_off;  // bytes of the chunk from a on (>= 1)
                    if (a_off + 16 <= readable) {
                        const uint64_t a0 = enc_load8(a), b0 = enc_load8(b), a1 = enc_load8(a + 8), b1 = enc_load8(b + 8);
                        const uint32_t pv = a[-1];
                        const uint64_t x0 = a0 ^ b0, x1 = a1 ^ b1;
#if defined(LZ_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
                        asm volatile("" ::"v"(x0), "v"(x1), "v"(pv));
                        LZ_LAP(11)
#endif
                        // equal bytes from offset `from` inside the 16 fetched (16 - from if all equal)
                        auto run16 = [&](uint32_t from) -> uint32_t {
                            if (from >= 16) return 0;
                            if (from >= 8) {
                                const uint64_t hi = x1 >> (8 * (from - 8));
                                return hi ? (uint32_t)__builtin_ctzll(hi) >> 3 : 16 - from;
                            }
                            const uint64_t lo = x0 >> (8 * from);
                            if (lo) return (uint32_t)__builtin_ctzll(lo) >> 3;
                            return (8 - from) + (x1 ? (uint32_t)__builtin_ctzll(x1) >> 3 : 8);
                        };
                        if (lim1) {
                            const uint32_t lim = lim1 < left ? lim1 : left;
                            s1 = run16(0);
                            if (s1 == 16 && lim > 16) s1 = 16 + match_len((int32_t)(a_off + 16 - mfpos), dist, lim - 16);
                            if (s1 > lim) s1 = lim;
                        }
                        tail = is_rep ? (s1 >= 2 && s1 < avail_full) : true;
                        if (tail) {
                            if (is_rep) lim2 = avail_full - 1 - s1 < fb ? avail_full - 1 - s1 : fb;
                            if (s1 < 16) {
                                const uint32_t sh = 8 * (s1 & 7);
                                cur_b = (uint32_t)((s1 < 8 ? a0 : a1) >> sh) & 0xFF;
                                match_b = (uint32_t)((s1 < 8 ? b0 : b1) >> sh) & 0xFF;
                                const uint32_t shp = 8 * ((s1 - 1) & 7);
                                prev_b = s1 == 0 ? pv : (uint32_t)((s1 <= 8 ? a0 : a1) >> shp) & 0xFF;
                            } else {
                                cur_b = a[s1], match_b = b[s1], prev_b = a[s1 - 1];
                            }
                            if (lim2) {
                                const uint32_t from = s1 + 1;
                                uint32_t lim = left > from ? left - from : 0;
                                if (lim2 < lim) lim = lim2;
                                s2 = run16(from);
                                if (from + s2 >= 16 && lim > s2)
                                    s2 += match_len((int32_t)(a_off + from + s2 - mfpos), dist, lim - s2);
                                if (s2 > lim) s2 = lim;
                            }
                        }
                    } else {  // last bytes of the buffer: byte-wise
                        const int32_t ia = (int32_t)(a_off - mfpos);
                        if (lim1) s1 = match_len(ia, dist, lim1);
                        tail = is_rep ? (s1 >= 2 && s1 < avail_full) : true;
                        if (tail) {
                            if (is_rep) lim2 = avail_full - 1 - s1 < fb ? avail_full - 1 - s1 : fb;
                            cur_b = a[s1], match_b = b[s1], prev_b = a[(int32_t)s1 - 1];
                            if (lim2) s2 = match_len(ia + (int32_t)s1 + 1, dist, lim2);
                        }
                    }
                }
                const uint32_t lt_any = is_rep ? s1 : lt_pair;  // 0 for the literal lane
                uint32_t tgt = 0, la_p = 0;
                if (tail && (is_lit || s2 >= 2)) {
                    const uint32_t lt = lt_any;
                    const uint32_t pos_after = position + lt;
                    const bool mode = is_lit ? !st_is_char(st) : true;
                    const uint32_t litp = lit_price(lit_sub(pos_after, prev_b), mode, match_b, cur_b);
                    if (is_lit) {
                        LZ_LV(x0, L) = cur_b, LZ_LV(x1, L) = match_b, LZ_LV(x2, L) = litp, LZ_LV(x3, L) = s2;
                    } else {
                        const uint32_t s_after = is_rep ? st_rep(st) : st_match(st);
                        const uint32_t s2s = st_char(s_after);
                        const uint32_t psn = pos_after & pos_state_mask, psn2 = (pos_after + 1) & pos_state_mask;
                        // the literal is a non-match in the state after the step, the rep0 after it a
                        // "match, rep, rep0 of more than one byte" in the state after the literal
                        const uint32_t g_im0 = sp_im0(s_after, psn);
                        const uint32_t g_tail = sp_prp0(s2s, psn2);
                        const uint32_t base = is_rep ? rprp + len_price(1, lt - kMatchMin, ps)
                                                     : nmp + pos_len_price(dist, lt, ps);
                        la_p = base + g_im0 + litp + g_tail + len_price(1, s2 - kMatchMin, psn2);
                        tgt = cur + lt + 1 + s2;
                    }
                }
                // lanes 0-3 hand over (rep length, continuation target, its price) in x0..x2; the
                // pair lanes of the first pass their pair's length, target and price in x0..x2
                // (for the end of the parse range below and for the apply phase) besides the
                // arrays a list longer than one pass is applied from
                if (is_rep) {
                    LZ_LV(x0, L) = s1, LZ_LV(x1, L) = tgt, LZ_LV(x2, L) = la_p;
                } else if (!is_lit && do_rest && L >= 8 && j < npairs) {
                    la_tgt[j] = tgt;
                    la_price[j] = la_p;
                    if (first) LZ_LV(x0, L) = lt_pair, LZ_LV(x1, L) = tgt, LZ_LV(x2, L) = la_p;
                }
            }
            LZ_PAR_END

            LZ_LAP(2)
            // ---- literal and short rep into entry cur+1 (:598-624)
            const uint32_t cur_byte = LZ_FROM_LANE(x0, 4), match_byte = LZ_FROM_LANE(x1, 4), lit_l2 = LZ_FROM_LANE(x3, 4);
            bool next_is_char;
            if constexpr (!COOP) {
                cur_and1 = cur_price + p_im0 + LZ_FROM_LANE(x2, 4);
                const NextUpd u = merge_next(nxr, cur, cur_and1, short_rep, match_byte == cur_byte);
                if (u.changed) u_set_pbp(cur + 1, u.price, u.back, u.pp);
                next_is_char = u.next_is_char;
                if (!do_rest) continue;
            } else {
                // is this position parsed at all?  Known when A(cur - 1) is done.
                LZ_LAP(9)
                if (!sy_wait(SY_A, cur - 1)) return false;
                LZ_LAP(13)
                const uint32_t e_v = sync[SY_LEN_END], f_v = sync[SY_FAR_HI];
                len_end = U(e_v), far_hi = U(f_v);
                if (cur == len_end) {
                    ex.cur = cur, ex.long_match = 0, ex.longest_len = 0, ex.last_dist = 0;
                    publish_exit(ex);
                    return true;
                }
                pend_slot = win_enter(cur, pend);
                if (!sy_wait(SY_M, cur)) return false;
                LZ_LAP(14)
                const uint32_t n_v = sync[SY_NIC + (cur & 1)];
                next_is_char = U(n_v) != 0;
                if (!do_rest) {
                    win_land(pend_slot, pend);
                    sy_signal(SY_A, cur);  // len_end and far_hi stay as they are
                    continue;
                }
            }
            LZ_LAP(9)

            // ---- continuation of the literal (:635-658)
            for (uint32_t i = 0; i < 5; i++) A.lt[i] = 0, A.lp[i] = 0;
            // new_end: first the furthest entry this position has a candidate for (the apply phase
            // goes no further: the entries between there and the end of the parse range, reached
            // from earlier positions, have nothing to gain here), then the new end of the range
            uint32_t new_end = 0;
            if (!next_is_char && match_byte != cur_byte) {
                const uint32_t l2 = lit_l2;
                if (l2 >= 2) {
                    const uint32_t s2 = st_char(st);
                    const uint32_t psn = (position + 1) & pos_state_mask;
                    // "match, rep, rep0 of more than one byte" after the literal: the rep0 price of
                    // (state after the literal, its posState) plus the length
                    const uint32_t lenp_v = len_price(1, l2 - kMatchMin, psn);
                    const uint32_t tail = U(sp_prp0(s2, psn)) + U(lenp_v);
                    A.lt[0] = cur + 1 + l2;
                    A.lp[0] = cur_and1 + tail;
                    if (cur + 1 + l2 > new_end) new_end = cur + 1 + l2;
                }
            }
            A.rl[0] = LZ_FROM_LANE(x0, 0), A.rl[1] = LZ_FROM_LANE(x0, 1);
            A.rl[2] = LZ_FROM_LANE(x0, 2), A.rl[3] = LZ_FROM_LANE(x0, 3);
            A.lt[1] = LZ_FROM_LANE(x1, 0), A.lt[2] = LZ_FROM_LANE(x1, 1);
            A.lt[3] = LZ_FROM_LANE(x1, 2), A.lt[4] = LZ_FROM_LANE(x1, 3);
            A.lp[1] = LZ_FROM_LANE(x2, 0), A.lp[2] = LZ_FROM_LANE(x2, 1);
            A.lp[3] = LZ_FROM_LANE(x2, 2), A.lp[4] = LZ_FROM_LANE(x2, 3);
            LZ_LAP(10)
            const uint32_t start_len = A.rl[0] >= 2 ? A.rl[0] + 1 : 2;
            for (uint32_t ri = 0; ri < kNumReps; ri++) {
                const uint32_t lt = A.rl[ri];
                if (lt >= 2 && cur + lt > new_end) new_end = cur + lt;
                if (A.lt[1 + ri] > new_end) new_end = A.lt[1 + ri];
            }
            uint32_t first_la = npairs, plain_len = 0;
            if (new_len >= start_len) {
                plain_len = new_len;
                if (cur + new_len > new_end) new_end = cur + new_len;
                // pairs shorter than start_len take no part (their lengths increase with the index);
                // the first pass of the round left the pairs' lengths and targets in lanes 8..63
                const uint32_t in_lanes = npairs < kLanes - 8 ? npairs : kLanes - 8;
                LZ_LANEVAR(xs);
                LZ_LANEVAR(xm);
                LZ_LANES_BEGIN(L)
                const bool mine = L >= 8 && L - 8 < in_lanes;
                LZ_LV(xs, L) = mine && LZ_LV(x0, L) < start_len;
                LZ_LV(xm, L) = (mine && LZ_LV(x0, L) >= start_len) ? LZ_LV(x1, L) : 0u;
                LZ_LANES_END
                first_la = LZ_WAVE_COUNT(xs);
                const uint32_t tg_max = LZ_WAVE_MAX(xm);
                if (tg_max > new_end) new_end = tg_max;
                if (npairs > in_lanes) {  // lists longer than one pass of the round: the rest from memory
                    if (first_la == in_lanes)
                        while (U(mdist[2 * first_la]) < start_len) first_la++;
                    for (uint32_t j = first_la > in_lanes ? first_la : in_lanes; j < npairs; j++) {
                        const uint32_t tg = U(la_tgt[j]);
                        if (tg > new_end) new_end = tg;
                    }
                }
            }
            A.cur = cur, A.old_end = len_end, A.new_end = new_end, A.ps = ps, A.start_len = start_len;
            if (len_end > new_end) new_end = len_end;
            A.new_len = plain_len, A.nmp = nmp, A.npairs = npairs, A.first_la = first_la;
            LZ_LAP(3)
            if constexpr (COOP) win_land(pend_slot, pend);
            if (npairs <= kLanes - 8)
                apply_lanes(A, x0, x1, x2);
            else
                apply(A, true);
            note_far(new_end);
            LZ_LAP(4)
            len_end = new_end;
            if constexpr (COOP) {
                if (one()) sync[SY_LEN_END] = len_end, sync[SY_FAR_HI] = far_hi;
                sy_signal(SY_A, cur);
            }
        }
    }

    LZ_HD void fill_dist_prices() {  // :801-827
        auto temp = temp_prices;
        LZ_PAR_BEGIN(L)
        for (uint32_t i = kStartPosModel + L; i < kNumFullDist; i += kLanes) {
            uint32_t slot = pos_slot(i), fbits = (slot >> 1) - 1, base = (2 | (slot & 1)) << fbits;
            temp[i] = bt_rev_price(probs, P_POS_SPEC + base - slot - 1, fbits, i - base);
        }
        for (uint32_t k = L; k < kNumLenToPos * ep.dist_table_size; k += kLanes) {
            const uint32_t lps = k / ep.dist_table_size, slot = k % ep.dist_table_size;
            uint32_t v = bt_price(probs, P_POS_SLOT + (lps << kPosSlotBits), kPosSlotBits, slot);
            if (slot >= kEndPosModel) v += (((slot >> 1) - 1) - kAlignBits) << kPriceShift;
            prices[pl.pos_slot + (lps << kPosSlotBits) + slot] = (WT)v;
        }
        LZ_PAR_END
        LZ_PAR_BEGIN(L)
        for (uint32_t k = L; k < kNumLenToPos * kNumFullDist; k += kLanes) {
            const uint32_t lps = k / kNumFullDist, i = k % kNumFullDist;
            auto sp = prices + pl.pos_slot + (lps << kPosSlotBits);
            prices[pl.dist + k] = (WT)(i < kStartPosModel ? sp[i] : sp[pos_slot(i)] + temp[i]);
        }
        LZ_PAR_END
        match_price_count = 0;
    }
    LZ_HD void fill_align_prices() {  // :829-834
        LZ_PAR_BEGIN(L)
        if (L < kAlignSize) prices[pl.align + L] = (WT)bt_rev_price(probs, P_ALIGN, kAlignBits, L);
        LZ_PAR_END
        align_price_count = 0;
    }

    LZ_HD void flush(uint32_t now) {  // :836-857, range_coder.go:158-166
        if (ep.eos) {
            uint32_t ps = now & pos_state_mask;
            rc_bit(probs, P_IS_MATCH + LZ_IDX(state, ps), 1);
            rc_bit(probs, P_IS_REP + state, 0);
            state = st_match(state);
            len_encode(0, 0, ps);
            bt_encode(probs, P_POS_SLOT + (len_to_pos_state(kMatchMin) << kPosSlotBits), kPosSlotBits,
                      (1u << kPosSlotBits) - 1);
            uint32_t fbits = 30, reduced = (1u << fbits) - 1;
            rc_direct(reduced >> kAlignBits, fbits - kAlignBits);
            bt_rev_encode(probs, P_ALIGN, kAlignBits, reduced & kAlignMask);
        }
        for (int i = 0; i < 5; i++) rc_shift_low();
    }

    LZ_HD void code_one_block() {  // :859-978
        finished = 1;
        const uint32_t prev_progress = now_pos;
        if (now_pos == 0) {
            if (avail() == 0) {
                flush(now_pos);
                return;
            }
            (void)read_match_distances();
            rc_bit(probs, P_IS_MATCH + LZ_IDX(state, now_pos & pos_state_mask), 0);
            state = st_char(state);
            uint32_t cb = uindex_byte(0 - (int32_t)add_offset);
            lit_encode(lit_sub(now_pos, prev_byte), cb);
            prev_byte = cb;
            add_offset--;
            now_pos++;
        }
        if (avail() == 0) {
            flush(now_pos);
            return;
        }
        for (;;) {
            LZ_LAP(7)
            uint32_t len = get_optimum(now_pos);
            if constexpr (COOP) {
                if (dead) return;  // finished is set: the chunk ends here with an error status
            }
            LZ_LAP(6)
            uint32_t pos = back_res;
            uint32_t ps = now_pos & pos_state_mask;
            uint32_t cs = LZ_IDX(state, ps);
            if (len == 1 && pos == 0xFFFFFFFFu) {
                rc_bit(probs, P_IS_MATCH + cs, 0);
                uint32_t cb = uindex_byte(0 - (int32_t)add_offset);
                uint32_t sub = lit_sub(now_pos, prev_byte);
                if (!st_is_char(state)) {
                    uint32_t mb = uindex_byte(0 - (int32_t)U(rep_dist[0]) - 1 - (int32_t)add_offset);
                    lit_encode_matched(sub, mb, cb);
                } else
                    lit_encode(sub, cb);
                prev_byte = cb;
                state = st_char(state);
            } else {
                rc_bit(probs, P_IS_MATCH + cs, 1);
                if (pos < kNumReps) {
                    rc_bit(probs, P_IS_REP + state, 1);
                    if (pos == 0) {
                        rc_bit(probs, P_IS_REP_G0 + state, 0);
                        rc_bit(probs, P_IS_REP0_LONG + cs, len == 1 ? 0 : 1);
                    } else {
                        rc_bit(probs, P_IS_REP_G0 + state, 1);
                        if (pos == 1)
                            rc_bit(probs, P_IS_REP_G1 + state, 0);
                        else {
                            rc_bit(probs, P_IS_REP_G1 + state, 1);
                            rc_bit(probs, P_IS_REP_G2 + state, pos - 2);
                        }
                    }
                    if (len == 1)
                        state = st_shortrep(state);
                    else {
                        len_encode(1, len - kMatchMin, ps);
                        state = st_rep(state);
                    }
                    uint32_t d = U(rep_dist[pos]);
                    if (pos != 0) {
                        for (uint32_t i = pos; i >= 1; i--) rep_dist[i] = U(rep_dist[i - 1]);
                        rep_dist[0] = d;
                    }
                } else {
                    rc_bit(probs, P_IS_REP + state, 0);
                    state = st_match(state);
                    len_encode(0, len - kMatchMin, ps);
                    pos -= kNumReps;
                    uint32_t slot = pos_slot(pos);
                    bt_encode(probs, P_POS_SLOT + (len_to_pos_state(len) << kPosSlotBits), kPosSlotBits,
                              slot);
                    if (slot >= kStartPosModel) {
                        uint32_t fbits = (slot >> 1) - 1, base = (2 | (slot & 1)) << fbits;
                        uint32_t reduced = pos - base;
                        if (slot < kEndPosModel)
                            bt_rev_encode(probs, P_POS_SPEC + base - slot - 1, fbits, reduced);
                        else {
                            rc_direct(reduced >> kAlignBits, fbits - kAlignBits);
                            bt_rev_encode(probs, P_ALIGN, kAlignBits, reduced & kAlignMask);
                            align_price_count++;
                        }
                    }
                    for (uint32_t i = kNumReps - 1; i >= 1; i--) rep_dist[i] = U(rep_dist[i - 1]);
                    rep_dist[0] = pos;
                    match_price_count++;
                }
                prev_byte = uindex_byte((int32_t)len - 1 - (int32_t)add_offset);
            }
            add_offset -= len;
            now_pos += len;
            if (add_offset == 0) {
                if (match_price_count >= (1u << 7)) fill_dist_prices();
                if (align_price_count >= kAlignSize) fill_align_prices();
                if (avail() == 0) {
                    flush(now_pos);
                    return;
                }
                if (now_pos - prev_progress >= (1u << 12)) {
                    finished = 0;
                    return;
                }
            }
        }
    }
#undef LZ_IDX

    // Models must already hold kProbInit and pp must be built.  Encodes the whole
    // chunk; out_n receives the payload size (header excluded).
    LZ_HD void run(U32P small) {
        bind_small(small, ep.fb, ep.pb);
        low = 0;
        out_n = 0;
        cache_size = 1;
        cache = 0;
        range = 0xFFFFFFFFu;
        overflow = 0;
        dead = 0;
        mfpos = 0;
        longest_len = dist_pairs = add_offset = opt_end = opt_cur = longest_found = 0;
        now_pos = 0;
        finished = 0;
        state = 0;
        prev_byte = 0;
        back_res = 0;
        for (int i = 0; i < 4; i++) rep_dist[i] = 0;
        pos_state_mask = (1u << ep.pb) - 1;
        lit_pos_mask = (1u << ep.lp) - 1;
        pl = price_layout(ep.pb, ep.fb);
        for (uint32_t ps = 0; ps < (1u << ep.pb); ps++) {
            len_set_prices(0, ps);
            len_set_prices(1, ps);
        }
        fill_dist_prices();
        fill_align_prices();
        do code_one_block();
        while (!finished);
        if constexpr (COOP) sy_signal(SY_FORK, kSyDone);
    }
};

// 13-byte header, lzma_encoder.go:1017-1024.
LZ_HD void write_header(uint8_t *out, const EncParams &ep, int64_t size) {
    out[0] = (uint8_t)((ep.pb * 5 + ep.lp) * 9 + ep.lc);
    for (int i = 0; i < 4; i++) out[1 + i] = (uint8_t)(ep.dict_size >> (8 * i));
    for (int i = 0; i < 8; i++) out[5 + i] = (uint8_t)((uint64_t)size >> (8 * i));
}

}  // namespace lz
