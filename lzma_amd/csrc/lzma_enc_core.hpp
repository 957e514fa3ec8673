// Serial core of the per-chunk LZMA encoder: price-optimal parse + range coder.
//
// Behaviour follows the reference encoder: getOptimum/backward
// (lzma_encoder.go:331-797), codeOneBlock/flush/writeEndMarker (:836-978), the
// price tables (:294-329, :801-834), the length coder with its cached price table
// (lzma_len_coder.go:48-126), the literal coder (lzma_lit_coder.go:47-102), the
// bit trees (range_bit_tree_coder.go:53-117) and the range encoder
// (range_coder.go:131-246).
//
// What differs is where things live and who finds matches:
//   * the match finder is a functor `mf(pos, pairs)` that returns the match list
//     of any position directly (the lists of the exhaustive BT4 finder are a pure
//     function of the data, so skipped positions cost nothing);
//   * models and price tables are meant to sit in LDS, prices are 16-bit, the
//     optimum table (4096 entries) sits in HBM behind the L2;
//   * getPosSlot is computed with a count-leading-zeros instead of a 2 KiB table.
//
// Execution model on the GPU: one wavefront per chunk, all lanes executing this
// code with identical values.  The same source compiles on the host
// (LZ_HOSTSIM) for tests.
#pragma once
#include "lzma_common.hpp"

namespace lz {

constexpr uint32_t kEncSmallWords = 12 + 2 * kPosStatesMax + kNumFullDist;

struct EncParams {
    uint32_t fb, lc, lp, pb, dict_size, dist_table_size, eos;
};

// One entry of the optimum table (`optimal`, lzma_encoder.go:167-181).
struct Opt {
    uint32_t price, back_prev, back_prev2;
    uint32_t backs[4];
    uint16_t pos_prev, pos_prev2;
    uint8_t state, prev1_is_char, prev2, pad;
};

// Price of one bit, range_coder.go:222-246.  The 512-entry table is built by
// build_prob_prices() below.
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
};
LZ_HD PriceLayout price_layout(uint32_t pb) {
    PriceLayout l;
    uint32_t nps = 1u << pb;
    l.len = 0;
    l.rep_len = l.len + nps * kNumLenSyms;
    l.pos_slot = l.rep_len + nps * kNumLenSyms;
    l.dist = l.pos_slot + (kNumLenToPos << kPosSlotBits);
    l.align = l.dist + (kNumLenToPos * kNumFullDist);
    l.total = l.align + kAlignSize;
    return l;
}

template <class MF, class PT, class LT, class WT>
struct Encoder {
    // ---- memory the caller provides ----
    const uint8_t *data;  // chunk bytes
    uint32_t n;           // chunk length
    PT *probs;            // kNumBaseProbs models
    LT *lit;              // 0x300 << (lc+lp) literal models
    WT *prices;           // price_layout(pb).total entries
    const WT *pp;         // 512-entry bit-price table
    Opt *opt;             // kNumOpts entries
    uint32_t *mdist;      // 2 * (kMatchMax + 1) words: current match list (len, dist) pairs
    uint8_t *out;         // payload output (after the header)
    uint64_t out_cap;
    MF mf;
    EncParams ep;
    PriceLayout pl;

    // ---- range encoder, range_coder.go:131-138 ----
    uint64_t low, out_n;
    uint32_t cache_size, cache, range, overflow;

    // ---- encoder state, lzma_encoder.go:210-272 ----
    uint32_t mfpos;  // positions already handed to the match finder
    uint32_t longest_len, dist_pairs, add_offset, opt_end, opt_cur, longest_found;
    uint32_t align_price_count, match_price_count, pos_state_mask, lit_pos_mask;
    uint32_t now_pos, finished, state, prev_byte, back_res;
    // Small runtime-indexed arrays, kept in caller-provided (LDS) memory so that they
    // never end up in per-lane scratch: kEncSmallWords words.
    uint32_t *rep_dist, *reps, *rep_lens;  // 4 each
    uint32_t *len_counter;                 // [2][kPosStatesMax]
    uint32_t *temp_prices;                 // kNumFullDist, scratch of fill_dist_prices

    // ------------------------------------------------------------ range coder
    LZ_HD void rc_put(uint32_t b) {
        if (out_n < out_cap)
            out[out_n] = (uint8_t)b;
        else
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
    LZ_HD void rc_bit(P *pr, uint32_t idx, uint32_t bit) {  // range_coder.go:205-220
        uint32_t p = pr[idx], bound = (range >> kProbBits) * p;
        if (bit == 0) {
            range = bound;
            pr[idx] = (prob_t)(p + ((kProbTotal - p) >> kMoveBits));
        } else {
            low += bound;
            range -= bound;
            pr[idx] = (prob_t)(p - (p >> kMoveBits));
        }
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
    template <class P>
    LZ_HD void bt_encode(P *pr, uint32_t base, uint32_t nbits, uint32_t sym) {
        uint32_t k = 1;
        for (uint32_t i = nbits; i != 0;) {
            i--;
            uint32_t b = (sym >> i) & 1;
            rc_bit(pr, base + k, b);
            k = (k << 1) | b;
        }
    }
    template <class P>
    LZ_HD void bt_rev_encode(P *pr, uint32_t base, uint32_t nbits, uint32_t sym) {
        uint32_t k = 1;
        for (uint32_t i = 0; i < nbits; i++) {
            uint32_t b = sym & 1;
            rc_bit(pr, base + k, b);
            k = (k << 1) | b;
            sym >>= 1;
        }
    }

    // ----------------------------------------------------------------- prices
    LZ_HD uint32_t price_bit(uint32_t p, uint32_t bit) const {
        return pp[(((p - bit) ^ (0u - bit)) & (kProbTotal - 1)) >> kMoveReducingBits];
    }
    LZ_HD uint32_t price0(uint32_t p) const { return pp[p >> kMoveReducingBits]; }
    LZ_HD uint32_t price1(uint32_t p) const { return pp[(kProbTotal - p) >> kMoveReducingBits]; }
    template <class P>
    LZ_HD uint32_t bt_price(const P *pr, uint32_t base, uint32_t nbits, uint32_t sym) const {
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
    LZ_HD uint32_t bt_rev_price(const P *pr, uint32_t base, uint32_t nbits, uint32_t sym) const {
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
    LZ_HD void len_set_prices(uint32_t which, uint32_t ps) {  // lzma_len_coder.go:66-88,111-114
        const uint32_t b = len_base(which), nsym = ep.fb + 1 - kMatchMin;
        uint32_t a0 = price0(probs[b + LEN_CHOICE]), a1 = price1(probs[b + LEN_CHOICE]);
        uint32_t b0 = a1 + price0(probs[b + LEN_CHOICE + 1]);
        uint32_t b1 = a1 + price1(probs[b + LEN_CHOICE + 1]);
        WT *pr = prices + len_tab(which) + ps * kNumLenSyms;
        uint32_t i = 0;
        for (; i < kLenLow && i < nsym; i++) pr[i] = (WT)(a0 + bt_price(probs, b + LEN_LOW + (ps << 3), 3, i));
        for (; i < kLenLow + kLenMid && i < nsym; i++)
            pr[i] = (WT)(b0 + bt_price(probs, b + LEN_MID + (ps << 3), 3, i - kLenLow));
        for (; i < nsym; i++)
            pr[i] = (WT)(b1 + bt_price(probs, b + LEN_HIGH, 8, i - kLenLow - kLenMid));
        len_counter[which * kPosStatesMax + ps] = nsym;
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
        if (--len_counter[which * kPosStatesMax + ps] == 0) len_set_prices(which, ps);
    }
    LZ_HD uint32_t len_price(uint32_t which, uint32_t sym, uint32_t ps) const {
        return prices[len_tab(which) + ps * kNumLenSyms + sym];
    }

    // ----------------------------------------------------------------- literals
    LZ_HD uint32_t lit_sub(uint32_t pos, uint32_t prev) const {  // lzma_lit_coder.go:126-128
        return (((pos & lit_pos_mask) << ep.lc) + (prev >> (8 - ep.lc))) * 0x300u;
    }
    LZ_HD void lit_encode(uint32_t sub, uint32_t sym) {  // :47-55
        uint32_t ctx = 1;
        for (int i = 7; i >= 0; i--) {
            uint32_t b = (sym >> i) & 1;
            rc_bit(lit, sub + ctx, b);
            ctx = (ctx << 1) | b;
        }
    }
    LZ_HD void lit_encode_matched(uint32_t sub, uint32_t mb, uint32_t sym) {  // :57-76
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
    }
    LZ_HD uint32_t lit_price(uint32_t sub, bool match_mode, uint32_t mb, uint32_t sym) const {  // :78-102
        uint32_t price = 0, ctx = 1;
        int i = 7;
        if (match_mode) {
            for (; i >= 0; i--) {
                uint32_t m = (mb >> i) & 1, b = (sym >> i) & 1;
                price += price_bit(lit[sub + ((1 + m) << 8) + ctx], b);
                ctx = (ctx << 1) | b;
                if (m != b) {
                    i--;
                    break;
                }
            }
        }
        for (; i >= 0; i--) {
            uint32_t b = (sym >> i) & 1;
            price += price_bit(lit[sub + ctx], b);
            ctx = (ctx << 1) | b;
        }
        return price;
    }

    // ------------------------------------------------------ window access helpers
    LZ_HD uint32_t avail() const { return n - mfpos; }  // getNumAvailableBytes
    LZ_HD uint32_t index_byte(int32_t index) const { return data[(int64_t)mfpos + index]; }
    LZ_HD uint32_t match_len(int32_t index, uint32_t distance, uint32_t limit) const {
        int64_t p = (int64_t)mfpos + index;  // lz_window.go:169-182
        uint32_t left = n - (uint32_t)p;
        if (limit > left) limit = left;
        const uint8_t *a = data + p;
        const uint8_t *b = a - (int64_t)distance - 1;
        uint32_t r = 0;
        while (r < limit && a[r] == b[r]) r++;
        return r;
    }

    LZ_HD uint32_t read_match_distances() {  // lzma_encoder.go:274-285
        uint32_t res = 0;
        dist_pairs = 2 * mf(mfpos, mdist);
        mfpos++;
        if (dist_pairs > 0) {
            res = mdist[dist_pairs - 2];
            if (res == ep.fb) res += match_len((int32_t)res - 1, mdist[dist_pairs - 1], kMatchMax - res);
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

    LZ_HD uint32_t backward(uint32_t cur) {  // :331-357
        Opt *o = opt;
        opt_end = cur;
        uint32_t pos_mem = o[cur].pos_prev, back_mem = o[cur].back_prev;
        do {
            if (o[cur].prev1_is_char) {
                o[pos_mem].back_prev = 0xFFFFFFFFu;
                o[pos_mem].prev1_is_char = 0;
                o[pos_mem].pos_prev = (uint16_t)(pos_mem - 1);
                if (o[cur].prev2) {
                    o[pos_mem - 1].prev1_is_char = 0;
                    o[pos_mem - 1].pos_prev = o[cur].pos_prev2;
                    o[pos_mem - 1].back_prev = o[cur].back_prev2;
                }
            }
            uint32_t pp_ = pos_mem, bc = back_mem;
            back_mem = o[pp_].back_prev;
            pos_mem = o[pp_].pos_prev;
            o[pp_].back_prev = bc;
            o[pp_].pos_prev = (uint16_t)cur;
            cur = pp_;
        } while (cur > 0);
        back_res = o[0].back_prev;
        opt_cur = o[0].pos_prev;
        return opt_cur;
    }
    LZ_HD void extend_end(uint32_t &len_end, uint32_t upto) {
        while (len_end < upto) opt[++len_end].price = kInfPrice;
    }
    LZ_HD void relax(uint32_t idx, uint32_t price, uint32_t pos_prev, uint32_t back_prev) {
        Opt *t = &opt[idx];
        if (price < t->price) {
            t->price = price;
            t->pos_prev = (uint16_t)pos_prev;
            t->back_prev = back_prev;
            t->prev1_is_char = 0;
        }
    }
    LZ_HD void relax2(uint32_t idx, uint32_t price, uint32_t pos_prev, uint32_t prev2,
                      uint32_t pos_prev2, uint32_t back_prev2) {
        Opt *t = &opt[idx];
        if (price < t->price) {
            t->price = price;
            t->pos_prev = (uint16_t)pos_prev;
            t->back_prev = 0;
            t->prev1_is_char = 1;
            t->prev2 = (uint8_t)prev2;
            if (prev2) {
                t->pos_prev2 = (uint16_t)pos_prev2;
                t->back_prev2 = back_prev2;
            }
        }
    }

    LZ_HD uint32_t get_optimum(uint32_t position) {  // :359-797
        Opt *o = opt;
        const uint32_t fb = ep.fb;
        if (opt_end != opt_cur) {
            uint32_t r = o[opt_cur].pos_prev - opt_cur;
            back_res = o[opt_cur].back_prev;
            opt_cur = o[opt_cur].pos_prev;
            return r;
        }
        opt_end = opt_cur = 0;
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
        uint32_t rep_max = 0;
        for (uint32_t i = 0; i < kNumReps; i++) {
            reps[i] = rep_dist[i];
            rep_lens[i] = match_len(-1, reps[i], kMatchMax);
            if (rep_lens[i] > rep_lens[rep_max]) rep_max = i;
        }
        if (rep_lens[rep_max] >= fb) {
            back_res = rep_max;
            uint32_t r = rep_lens[rep_max];
            move_pos(r - 1);
            return r;
        }
        if (len_main >= fb) {
            back_res = mdist[dpairs - 1] + kNumReps;
            move_pos(len_main - 1);
            return len_main;
        }
        uint32_t cur_byte = index_byte(-1);
        uint32_t match_byte = index_byte(0 - (int32_t)rep_dist[0] - 1 - 1);
        if (len_main < 2 && cur_byte != match_byte && rep_lens[rep_max] < 2) {
            back_res = 0xFFFFFFFFu;
            return 1;
        }
        o[0].state = (uint8_t)state;
        uint32_t ps = position & pos_state_mask;
        o[1].price = price0(probs[P_IS_MATCH + LZ_IDX(state, ps)]) +
                     lit_price(lit_sub(position, prev_byte), !st_is_char(state), match_byte, cur_byte);
        o[1].back_prev = 0xFFFFFFFFu;
        o[1].prev1_is_char = 0;
        uint32_t match_price = price1(probs[P_IS_MATCH + LZ_IDX(state, ps)]);
        uint32_t rep_match_price = match_price + price1(probs[P_IS_REP + state]);
        if (match_byte == cur_byte) {
            uint32_t sp = rep_match_price + rep_len1_price(state, ps);
            if (sp < o[1].price) {
                o[1].price = sp;
                o[1].back_prev = 0;
                o[1].prev1_is_char = 0;
            }
        }
        uint32_t len_end = rep_lens[rep_max];
        if (len_main > len_end) len_end = len_main;
        if (len_end < 2) {
            back_res = o[1].back_prev;
            return 1;
        }
        o[1].pos_prev = 0;
        for (int i = 0; i < 4; i++) o[0].backs[i] = reps[i];
        for (uint32_t l = len_end; l >= 2; l--) o[l].price = kInfPrice;
        for (uint32_t i = 0; i < kNumReps; i++) {
            uint32_t rl = rep_lens[i];
            if (rl < 2) continue;
            uint32_t price = rep_match_price + pure_rep_price(i, state, ps);
            do relax(rl, price + len_price(1, rl - 2, ps), 0, i);
            while (--rl >= 2);
        }
        uint32_t normal_match_price = match_price + price0(probs[P_IS_REP + state]);
        uint32_t len = rep_lens[0] >= 2 ? rep_lens[0] + 1 : 2;
        if (len <= len_main) {
            uint32_t offs = 0;
            while (len > mdist[offs]) offs += 2;
            for (;; len++) {
                uint32_t d = mdist[offs + 1];
                relax(len, normal_match_price + pos_len_price(d, len, ps), 0, d + kNumReps);
                if (len == mdist[offs]) {
                    offs += 2;
                    if (offs == dpairs) break;
                }
            }
        }

        uint32_t cur = 0;
        for (;;) {
            cur++;
            if (cur == len_end) return backward(cur);
            uint32_t new_len = read_match_distances();
            dpairs = dist_pairs;
            if (new_len >= fb) {
                longest_len = new_len;
                longest_found = 1;
                return backward(cur);
            }
            position++;
            uint32_t pos_prev = o[cur].pos_prev, st;
            const uint32_t c_p1 = o[cur].prev1_is_char, c_p2 = o[cur].prev2;
            const uint32_t c_bp = o[cur].back_prev;
            if (c_p1) {
                pos_prev--;
                if (c_p2) {
                    st = o[o[cur].pos_prev2].state;
                    st = o[cur].back_prev2 < kNumReps ? st_rep(st) : st_match(st);
                } else
                    st = o[pos_prev].state;
                st = st_char(st);
            } else
                st = o[pos_prev].state;
            if (pos_prev == cur - 1) {
                st = c_bp == 0 ? st_shortrep(st) : st_char(st);
            } else {
                uint32_t pos;
                if (c_p1 && c_p2) {
                    pos_prev = o[cur].pos_prev2;
                    pos = o[cur].back_prev2;
                    st = st_rep(st);
                } else {
                    pos = c_bp;
                    st = pos < kNumReps ? st_rep(st) : st_match(st);
                }
                const uint32_t q0 = o[pos_prev].backs[0], q1 = o[pos_prev].backs[1],
                               q2 = o[pos_prev].backs[2], q3 = o[pos_prev].backs[3];
                if (pos < kNumReps) {
                    if (pos == 0) {
                        reps[0] = q0, reps[1] = q1, reps[2] = q2, reps[3] = q3;
                    } else if (pos == 1) {
                        reps[0] = q1, reps[1] = q0, reps[2] = q2, reps[3] = q3;
                    } else if (pos == 2) {
                        reps[0] = q2, reps[1] = q0, reps[2] = q1, reps[3] = q3;
                    } else {
                        reps[0] = q3, reps[1] = q0, reps[2] = q1, reps[3] = q2;
                    }
                } else {
                    reps[0] = pos - kNumReps, reps[1] = q0, reps[2] = q1, reps[3] = q2;
                }
            }
            o[cur].state = (uint8_t)st;
            for (int i = 0; i < 4; i++) o[cur].backs[i] = reps[i];
            const uint32_t cur_price = o[cur].price;
            cur_byte = index_byte(-1);
            match_byte = index_byte(0 - (int32_t)reps[0] - 1 - 1);
            ps = position & pos_state_mask;
            const uint32_t is_match_p = probs[P_IS_MATCH + LZ_IDX(st, ps)];
            const uint32_t cur_and1 = cur_price + price0(is_match_p) +
                                      lit_price(lit_sub(position, index_byte(-2)), !st_is_char(st),
                                                match_byte, cur_byte);
            Opt *nx = &o[cur + 1];
            bool next_is_char = false;
            if (cur_and1 < nx->price) {
                nx->price = cur_and1;
                nx->pos_prev = (uint16_t)cur;
                nx->back_prev = 0xFFFFFFFFu;
                nx->prev1_is_char = 0;
                next_is_char = true;
            }
            match_price = cur_price + price1(is_match_p);
            rep_match_price = match_price + price1(probs[P_IS_REP + st]);
            if (match_byte == cur_byte && !(nx->pos_prev < cur && nx->back_prev == 0)) {
                uint32_t sp = rep_match_price + rep_len1_price(st, ps);
                if (sp <= nx->price) {
                    nx->price = sp;
                    nx->pos_prev = (uint16_t)cur;
                    nx->back_prev = 0;
                    nx->prev1_is_char = 0;
                    next_is_char = true;
                }
            }
            uint32_t avail_full = kNumOpts - 1 - cur;
            if (avail() + 1 < avail_full) avail_full = avail() + 1;
            uint32_t av = avail_full;
            if (av < 2) continue;
            if (av > fb) av = fb;
            if (!next_is_char && match_byte != cur_byte) {
                uint32_t t = avail_full - 1 < fb ? avail_full - 1 : fb;
                uint32_t l2 = match_len(0, reps[0], t);
                if (l2 >= 2) {
                    uint32_t s2 = st_char(st);
                    uint32_t psn = (position + 1) & pos_state_mask;
                    uint32_t nrmp = cur_and1 + price1(probs[P_IS_MATCH + LZ_IDX(s2, psn)]) +
                                    price1(probs[P_IS_REP + s2]);
                    uint32_t offset = cur + 1 + l2;
                    extend_end(len_end, offset);
                    relax2(offset, nrmp + rep_price(0, l2, s2, psn), cur + 1, 0, 0, 0);
                }
            }
            uint32_t start_len = 2;
            for (uint32_t ri = 0; ri < kNumReps; ri++) {
                uint32_t lt = match_len(-1, reps[ri], av);
                if (lt < 2) continue;
                const uint32_t lt_keep = lt;
                extend_end(len_end, cur + lt);
                const uint32_t prp = rep_match_price + pure_rep_price(ri, st, ps);
                do relax(cur + lt, prp + len_price(1, lt - kMatchMin, ps), cur, ri);
                while (--lt >= 2);
                lt = lt_keep;
                if (ri == 0) start_len = lt + 1;
                if (lt < avail_full) {
                    uint32_t t = avail_full - 1 - lt < fb ? avail_full - 1 - lt : fb;
                    uint32_t l2 = match_len((int32_t)lt, reps[ri], t);
                    if (l2 >= 2) {
                        uint32_t s2 = st_rep(st);
                        uint32_t psn = (position + lt) & pos_state_mask;
                        uint32_t calcp = prp + len_price(1, lt - kMatchMin, ps) +
                                         price0(probs[P_IS_MATCH + LZ_IDX(s2, psn)]) +
                                         lit_price(lit_sub(position + lt, index_byte((int32_t)lt - 2)), true,
                                                   index_byte((int32_t)lt - 1 - (int32_t)(reps[ri] + 1)),
                                                   index_byte((int32_t)lt - 1));
                        s2 = st_char(s2);
                        psn = (position + lt + 1) & pos_state_mask;
                        uint32_t nmp = calcp + price1(probs[P_IS_MATCH + LZ_IDX(s2, psn)]);
                        uint32_t nrmp = nmp + price1(probs[P_IS_REP + s2]);
                        uint32_t offset = lt + 1 + l2;
                        extend_end(len_end, cur + offset);
                        relax2(cur + offset, nrmp + rep_price(0, l2, s2, psn), cur + lt + 1, 1, cur, ri);
                    }
                }
            }
            if (new_len > av) {
                new_len = av;
                for (dpairs = 0; new_len > mdist[dpairs]; dpairs += 2) {
                }
                mdist[dpairs] = new_len;
                dpairs += 2;
            }
            if (new_len >= start_len) {
                normal_match_price = match_price + price0(probs[P_IS_REP + st]);
                extend_end(len_end, cur + new_len);
                uint32_t offs = 0;
                while (start_len > mdist[offs]) offs += 2;
                for (uint32_t lt = start_len;; lt++) {
                    const uint32_t cb = mdist[offs + 1];
                    uint32_t cp = normal_match_price + pos_len_price(cb, lt, ps);
                    relax(cur + lt, cp, cur, cb + kNumReps);
                    if (lt == mdist[offs]) {
                        if (lt < avail_full) {
                            uint32_t t = avail_full - 1 - lt < fb ? avail_full - 1 - lt : fb;
                            uint32_t l2 = match_len((int32_t)lt, cb, t);
                            if (l2 >= 2) {
                                uint32_t s2 = st_match(st);
                                uint32_t psn = (position + lt) & pos_state_mask;
                                uint32_t calcp =
                                    cp + price0(probs[P_IS_MATCH + LZ_IDX(s2, psn)]) +
                                    lit_price(lit_sub(position + lt, index_byte((int32_t)lt - 2)), true,
                                              index_byte((int32_t)lt - ((int32_t)cb + 1) - 1),
                                              index_byte((int32_t)lt - 1));
                                s2 = st_char(s2);
                                psn = (position + lt + 1) & pos_state_mask;
                                uint32_t nmp = calcp + price1(probs[P_IS_MATCH + LZ_IDX(s2, psn)]);
                                uint32_t nrmp = nmp + price1(probs[P_IS_REP + s2]);
                                uint32_t offset = lt + 1 + l2;
                                extend_end(len_end, cur + offset);
                                relax2(cur + offset, nrmp + rep_price(0, l2, s2, psn), cur + lt + 1, 1,
                                       cur, cb + kNumReps);
                            }
                        }
                        offs += 2;
                        if (offs == dpairs) break;
                    }
                }
            }
        }
    }

    LZ_HD void fill_dist_prices() {  // :801-827
        uint32_t *temp = temp_prices;
        for (uint32_t i = kStartPosModel; i < kNumFullDist; i++) {
            uint32_t slot = pos_slot(i), fbits = (slot >> 1) - 1, base = (2 | (slot & 1)) << fbits;
            temp[i] = bt_rev_price(probs, P_POS_SPEC + base - slot - 1, fbits, i - base);
        }
        for (uint32_t lps = 0; lps < kNumLenToPos; lps++) {
            WT *sp = prices + pl.pos_slot + (lps << kPosSlotBits);
            uint32_t slot;
            for (slot = 0; slot < ep.dist_table_size; slot++)
                sp[slot] = (WT)bt_price(probs, P_POS_SLOT + (lps << kPosSlotBits), kPosSlotBits, slot);
            for (slot = kEndPosModel; slot < ep.dist_table_size; slot++)
                sp[slot] = (WT)(sp[slot] + ((((slot >> 1) - 1) - kAlignBits) << kPriceShift));
            WT *dp = prices + pl.dist + lps * kNumFullDist;
            uint32_t i;
            for (i = 0; i < kStartPosModel; i++) dp[i] = sp[i];
            for (; i < kNumFullDist; i++) dp[i] = (WT)(sp[pos_slot(i)] + temp[i]);
        }
        match_price_count = 0;
    }
    LZ_HD void fill_align_prices() {  // :829-834
        for (uint32_t i = 0; i < kAlignSize; i++)
            prices[pl.align + i] = (WT)bt_rev_price(probs, P_ALIGN, kAlignBits, i);
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
            uint32_t cb = index_byte(0 - (int32_t)add_offset);
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
            uint32_t len = get_optimum(now_pos);
            uint32_t pos = back_res;
            uint32_t ps = now_pos & pos_state_mask;
            uint32_t cs = LZ_IDX(state, ps);
            if (len == 1 && pos == 0xFFFFFFFFu) {
                rc_bit(probs, P_IS_MATCH + cs, 0);
                uint32_t cb = index_byte(0 - (int32_t)add_offset);
                uint32_t sub = lit_sub(now_pos, prev_byte);
                if (!st_is_char(state)) {
                    uint32_t mb = index_byte(0 - (int32_t)rep_dist[0] - 1 - (int32_t)add_offset);
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
                    uint32_t d = rep_dist[pos];
                    if (pos != 0) {
                        for (uint32_t i = pos; i >= 1; i--) rep_dist[i] = rep_dist[i - 1];
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
                    for (uint32_t i = kNumReps - 1; i >= 1; i--) rep_dist[i] = rep_dist[i - 1];
                    rep_dist[0] = pos;
                    match_price_count++;
                }
                prev_byte = index_byte((int32_t)len - 1 - (int32_t)add_offset);
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
    // chunk; returns the payload size (header excluded) or sets overflow.
    LZ_HD void run(uint32_t *small) {
        rep_dist = small;
        reps = small + 4;
        rep_lens = small + 8;
        len_counter = small + 12;
        temp_prices = small + 12 + 2 * kPosStatesMax;
        low = 0;
        out_n = 0;
        cache_size = 1;
        cache = 0;
        range = 0xFFFFFFFFu;
        overflow = 0;
        mfpos = 0;
        longest_len = dist_pairs = add_offset = opt_end = opt_cur = longest_found = 0;
        now_pos = 0;
        finished = 0;
        state = 0;
        prev_byte = 0;
        back_res = 0;
        for (int i = 0; i < 4; i++) rep_dist[i] = reps[i] = rep_lens[i] = 0;
        pos_state_mask = (1u << ep.pb) - 1;
        lit_pos_mask = (1u << ep.lp) - 1;
        pl = price_layout(ep.pb);
        for (uint32_t ps = 0; ps < (1u << ep.pb); ps++) {
            len_set_prices(0, ps);
            len_set_prices(1, ps);
        }
        fill_dist_prices();
        fill_align_prices();
        do code_one_block();
        while (!finished);
    }
};

// 13-byte header, lzma_encoder.go:1017-1024.
LZ_HD void write_header(uint8_t *out, const EncParams &ep, int64_t size) {
    out[0] = (uint8_t)((ep.pb * 5 + ep.lp) * 9 + ep.lc);
    for (int i = 0; i < 4; i++) out[1 + i] = (uint8_t)(ep.dict_size >> (8 * i));
    for (int i = 0; i < 8; i++) out[5 + i] = (uint8_t)((uint64_t)size >> (8 * i));
}

}  // namespace lz
