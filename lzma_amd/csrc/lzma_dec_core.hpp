// Serial core of the batched LZMA decoder: one call decodes one complete .lzma
// stream.  It follows decoder.doDecode (lzma_decoder.go:218-304) with the range
// decoder of range_coder.go:27-105, the bit trees of range_bit_tree_coder.go:19-51,
// the length decoder of lzma_len_coder.go:28-46 and the literal decoder of
// lzma_lit_coder.go:18-44.
//
// Execution model on the GPU: one 64-lane wavefront per stream.  Every lane
// executes this code with identical (wave-uniform) values, so the compiler keeps
// the range-coder state in scalar registers; the non-literal probabilities live in LDS,
// the literal ones in HBM (walked in registers, see below).  The match copy, the literal
// model fetch and the model initialisation use the lanes in parallel.  On the host
// (LZ_HOSTSIM) the same code runs with nlanes == 1.
#pragma once
#include "lzma_common.hpp"

namespace lz {

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
LZ_D uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
#else
LZ_HD uint32_t uni(uint32_t v) { return v; }
#endif

// Range decoder state (rangeDecoder, range_coder.go:27-31).  Input bytes are fetched seven
// at a time into `win`, the next byte in its top eight bits; a single set bit below the
// unread bytes marks their end, so the window is empty when only that bit is left.
struct RangeDec {
    const uint8_t *in;
    uint64_t n;    // stream length in bytes
    uint64_t p;    // offset of the first byte not yet in the window
    uint64_t win;  // unread bytes, first one on top, then the marker bit
    uint32_t range, code;
    uint32_t eof;  // set once a read past the end was attempted
    uint32_t lim;  // the output position may grow up to here; zero once eof is set
};
constexpr uint64_t kWinEmpty = 1ull << 63;

LZ_HD void rd_refill(RangeDec &r) {
    // The eight bytes that end with the last one wanted: near the end of the stream the load
    // starts before `p` (never before the header, which is longer than eight bytes) and the
    // bytes already consumed are shifted out.  Past the end the window delivers zeros.
    const uint64_t left = r.n - r.p;
    const uint32_t take = left < 7 ? (uint32_t)left : 7u;
    uint64_t v = 0;
    if (take) {
        __builtin_memcpy(&v, r.in + r.p + take - 8, 8);
        v = __builtin_bswap64(v);
        // every lane loads the same bytes; saying so keeps the decoder state in scalar registers
        v = (uint64_t)uni((uint32_t)v) | ((uint64_t)uni((uint32_t)(v >> 32)) << 32);
        v <<= 8 * (8 - take);
        r.p += take;
        r.win = v | (0x80ull << (8 * (7 - take)));
    } else {
        r.eof = 1;
        r.lim = 0;
        r.win = 0x80;
    }
}
LZ_HD uint32_t rd_byte(RangeDec &r) {
    if (r.win == kWinEmpty) rd_refill(r);
    uint32_t b = (uint32_t)(r.win >> 56);
    r.win <<= 8;
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    // Keeps the byte a value of its own: merged with the caller's shift of `code` it would
    // become a funnel shift, which only the vector unit has.
    b = uni(b);
    asm("" : "+s"(b));
#endif
    return b;
}
LZ_HD void rd_normalize(RangeDec &r) {
    if (r.range < kTopValue) {
        r.code = (r.code << 8) | rd_byte(r);
        r.range <<= 8;
    }
}
// The arithmetic of decodeBit (range_coder.go:75-105) on a model value `pr`; returns the bit
// and the updated model in `np`.  The update pr + ((2048 - pr) >> 5) of a zero bit equals
// pr - ((pr - 2017) >> 5) with an arithmetic shift, which lets both outcomes share one
// subtract-shift-subtract.
LZ_HD uint32_t rd_step(RangeDec &r, uint32_t pr, uint32_t &np) {
    const uint32_t bound = (r.range >> kProbBits) * pr;
    const bool bit = r.code >= bound;
    const uint32_t rb = r.range - bound, cb = r.code - bound;
    r.range = bit ? rb : bound;
    r.code = bit ? cb : r.code;
    np = pr - (uint32_t)(((int32_t)pr - (bit ? 0 : (int32_t)(kProbTotal - (1u << kMoveBits) + 1))) >> kMoveBits);
    rd_normalize(r);
    return bit ? 1u : 0u;
}
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
// On the GPU the decoder state is wave-uniform and a stream's speed is set by how many
// instructions its wave issues, the scalar ones above all: a CU runs 32 such waves and its
// scalar unit is the first to saturate, the vector unit the second.  The bit step is
// therefore split between the two.  Vector unit: bound = (range >> 11) * model, computed
// where the model is (every lane holds the same one, or in the literal trees each lane its
// own); the model update, from the decoded bit.  Scalar unit, written out by hand: the
// unsigned minimum of code and code - bound is the new code and leaves the decoded bit in
// SCC, which drives the select of the new range and the add-with-carry that shifts the bit
// into the symbol.
static_assert(kProbBits == 11 && kMoveBits == 5 && kProbTotal == 2048, "constants of the bit step");
#define LZ_STEP_SCALAR                                  \
    "s_sub_u32 %[rb], %[range], %[bound]\n\t"           \
    "s_sub_u32 %[cb], %[code], %[bound]\n\t"            \
    "s_min_u32 %[code], %[cb], %[code]\n\t"             \
    "s_cselect_b32 %[range], %[rb], %[bound]\n\t"
// k = 2k + bit
LZ_D void rd_step_tree(RangeDec &r, uint32_t bound, uint32_t &k) {
    // (uni() costs nothing on a value already in a scalar register and keeps the build from
    // failing where the compiler chose a vector register for one)
    uint32_t rb, cb, range = uni(r.range), code = uni(r.code), kk = uni(k);
    asm(LZ_STEP_SCALAR "s_addc_u32 %[k], %[k], %[k]"
        : [rb] "=&s"(rb), [cb] "=&s"(cb), [range] "+s"(range), [code] "+s"(code), [k] "+s"(kk)
        : [bound] "s"(bound)
        : "scc");
    r.range = range, r.code = code, k = kk;
}
// the same, handing out the bit
LZ_D uint32_t rd_step_bit(RangeDec &r, uint32_t bound) {
    uint32_t rb, cb, bit, range = uni(r.range), code = uni(r.code);
    asm(LZ_STEP_SCALAR "s_cselect_b32 %[bit], 1, 0"
        : [rb] "=&s"(rb), [cb] "=&s"(cb), [range] "+s"(range), [code] "+s"(code), [bit] "=&s"(bit)
        : [bound] "s"(bound)
        : "scc");
    r.range = range, r.code = code;
    return bit;
}
// bound of a model held in a vector register
LZ_D uint32_t model_bound(const RangeDec &r, uint32_t pv) {
    return __umul24(pv, r.range >> kProbBits);  // 11 x 21 bits
}
// The model update on the vector unit; `bits` is a scalar whose lowest bit is the decoded one.
// pv + ((2048 - pv) >> 5) for a zero equals pv - ((pv - 2017) >> 5) with an arithmetic shift,
// so both cases are pv - ((pv - off) >> 5) with off = 2017 or 0.
LZ_D uint32_t model_update(uint32_t pv, uint32_t bits) {
    uint32_t m;  // all ones for a one bit; on the vector unit on purpose
    asm("v_bfe_i32 %0, %1, 0, 1" : "=v"(m) : "s"(bits));
    return pv - (uint32_t)((int32_t)(pv - (kProbTotal - 31u) + (m & (kProbTotal - 31u))) >> kMoveBits);
}
// decodeBit, range_coder.go:75-105.
template <class P>
LZ_D uint32_t rd_bit(RangeDec &r, P *probs, uint32_t idx) {
    const uint32_t pv = probs[idx];
    const uint32_t bit = rd_step_bit(r, uni(model_bound(r, pv)));
    probs[idx] = (prob_t)model_update(pv, bit);
    rd_normalize(r);
    return bit;
}
#else
// decodeBit, range_coder.go:75-105.
template <class P>
LZ_HD uint32_t rd_bit(RangeDec &r, P *probs, uint32_t idx) {
    uint32_t np;
    const uint32_t bit = rd_step(r, probs[idx], np);
    probs[idx] = (prob_t)np;
    return bit;
}
#endif
// decodeDirectBits, range_coder.go:57-73.
LZ_HD uint32_t rd_direct(RangeDec &r, uint32_t nbits) {
    uint32_t res = 0;
    for (uint32_t i = nbits; i != 0; i--) {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
        uint32_t cb, range = uni(r.range), code = uni(r.code);  // the minimum leaves the bit in SCC, as above
        res = uni(res);
        asm("s_lshr_b32 %[range], %[range], 1\n\t"
            "s_sub_u32 %[cb], %[code], %[range]\n\t"
            "s_min_u32 %[code], %[cb], %[code]\n\t"
            "s_addc_u32 %[res], %[res], %[res]"
            : [cb] "=&s"(cb), [range] "+s"(range), [code] "+s"(code), [res] "+s"(res)
            :
            : "scc");
        r.range = range, r.code = code;
#else
        r.range >>= 1;
        uint32_t t = (r.code - r.range) >> 31;
        r.code -= r.range & (t - 1);
        res = (res << 1) | (1 - t);
#endif
        rd_normalize(r);
    }
    return res;
}
// rangeBitTreeCoder.decode / reverseDecode, range_bit_tree_coder.go:19-51.
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
template <class P>
LZ_D uint32_t bt_walk(RangeDec &r, P *probs, uint32_t nbits) {  // node index after nbits steps from the root
    uint32_t k = 1;
    for (uint32_t i = nbits; i != 0; i--) {
        P *m = probs + k;
        const uint32_t pv = *m;
        rd_step_tree(r, uni(model_bound(r, pv)), k);
        *m = (prob_t)model_update(pv, k);
        rd_normalize(r);
    }
    return k;
}
template <class P>
LZ_D uint32_t bt_decode(RangeDec &r, P *probs, uint32_t base, uint32_t nbits) {
    return bt_walk(r, probs + base, nbits) - (1u << nbits);
}
template <class P>
LZ_D uint32_t bt_rev_decode(RangeDec &r, P *probs, uint32_t base, uint32_t nbits) {
    // the walk visits the same nodes as the forward tree; its bits, shifted in first-decoded-highest,
    // are wanted lowest first
    const uint32_t k = bt_walk(r, probs + base, nbits);
    return (__builtin_bitreverse32(k) >> (32 - nbits)) & ((1u << nbits) - 1);
}
#else
template <class P>
LZ_HD uint32_t bt_decode(RangeDec &r, P *probs, uint32_t base, uint32_t nbits) {
    uint32_t k = 1;
    for (uint32_t i = nbits; i != 0; i--) k = (k << 1) + rd_bit(r, probs, base + k);
    return k - (1u << nbits);
}
template <class P>
LZ_HD uint32_t bt_rev_decode(RangeDec &r, P *probs, uint32_t base, uint32_t nbits) {
    uint32_t k = 1, res = 0;
    for (uint32_t i = 0; i < nbits; i++) {
        uint32_t b = rd_bit(r, probs, base + k);
        k = (k << 1) + b;
        res |= b << i;
    }
    return res;
}
#endif
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
// ---- literal coder walked in registers (GPU only)
// The literal models are too large to keep every stream's in LDS without giving up most
// of the streams a CU could hold, and decoding a literal from models in HBM one bit at a
// time would cost a memory round trip per bit.  Instead the 256 models of a literal bit
// tree are fetched at once into four registers, model k in lane k & 63 of register k >> 6.
// Bits 1..6 of a symbol use nodes 1..63 (register 0), bit 7 nodes 64..127 (register 1) and
// bit 8 nodes 128..255 (registers 2 and 3), so the register is known when the code is
// compiled.  A bit costs one readlane to fetch the model; every lane then updates its own
// model as if it had been the one read, and the lane that holds it keeps the result.
struct Tree4 {
    uint32_t r0, r1, r2, r3;
};
// the models base .. base + 255 (node k of the tree is model base + k)
template <class L>
LZ_D Tree4 tree_load(const L *lit, uint32_t base, uint32_t lane) {
    const L *p = lit + base + lane;
    Tree4 t;
    t.r0 = p[0];
    t.r1 = p[64];
    t.r2 = p[128];
    t.r3 = p[192];
    return t;
}
template <class L>
LZ_D void tree_store(L *lit, uint32_t base, uint32_t lane, const Tree4 &t) {
    L *p = lit + base + lane;
    p[0] = (L)t.r0;
    p[64] = (L)t.r1;
    p[128] = (L)t.r2;
    p[192] = (L)t.r3;
}
// decodeBit (range_coder.go:75-105) on node `sym` of a tree: `reg` is the register that holds
// the node, `node2_of_lane` twice the node each lane of it holds; sym = 2 sym + bit.
// (v_readlane takes the lane from the low six bits of sym.)
LZ_D void lit_step(RangeDec &r, uint32_t &reg, uint32_t node2_of_lane, uint32_t &sym) {
    // every lane forms the bound of its own model; the one of node `sym` is taken
    const uint32_t bound = (uint32_t)__builtin_amdgcn_readlane((int)model_bound(r, reg), (int)sym);
    rd_step_tree(r, bound, sym);
    const uint32_t np = model_update(reg, sym);
    // the lane whose node was read: node == sym >> 1, tested on the vector unit
    reg = sym - node2_of_lane < 2u ? np : reg;
    rd_normalize(r);
}
// bit number I (0 = first) of a symbol whose bits so far are `sym` (with the leading one)
template <int I>
LZ_D void tree_bit(RangeDec &r, Tree4 &t, uint32_t &sym, uint32_t lane) {
    if (I < 6) lit_step(r, t.r0, 2 * lane, sym);
    else if (I == 6) lit_step(r, t.r1, 2 * lane + 128, sym);
    else if (sym & 64u) lit_step(r, t.r3, 2 * lane + 384, sym);
    else lit_step(r, t.r2, 2 * lane + 256, sym);
}
// litSubCoder.decodeNormal / decodeWithMatchByte, lzma_lit_coder.go:18-44.  After a match the
// tree is chosen by the next bit of the byte the match would have continued with, for as long
// as the decoded bits agree with it (LZ_MATCHED); from the first difference on, and for every
// bit of an ordinary literal, it is the plain tree.  The two match-byte trees are fetched only
// as far as the walk gets: their first six levels at the start, the last two when bit 7 is
// reached with all bits so far in agreement.
template <class L>
LZ_D uint32_t lit_decode_regs(RangeDec &r, L *lit, uint32_t sub, bool matched, uint32_t mb, uint32_t lane) {
    Tree4 a = tree_load(lit, sub, lane);
    uint32_t sym = 1;
    if (matched) {
        L *p0 = lit + sub + 0x100u + lane, *p1 = lit + sub + 0x200u + lane;
        Tree4 b0, b1;
        b0.r0 = p0[0];
        b1.r0 = p1[0];
#define LZ_MATCHED(I, NEXT)                    \
    if (mb & (0x80u >> I)) {                   \
        tree_bit<I>(r, b1, sym, lane);         \
        if (!(sym & 1u)) goto NEXT;            \
    } else {                                   \
        tree_bit<I>(r, b0, sym, lane);         \
        if (sym & 1u) goto NEXT;               \
    }
        LZ_MATCHED(0, m1)
        LZ_MATCHED(1, m2)
        LZ_MATCHED(2, m3)
        LZ_MATCHED(3, m4)
        LZ_MATCHED(4, m5)
        LZ_MATCHED(5, m6)
        b0.r1 = p0[64], b0.r2 = p0[128], b0.r3 = p0[192];
        b1.r1 = p1[64], b1.r2 = p1[128], b1.r3 = p1[192];
        {
            bool plain7 = true;
            LZ_MATCHED(6, deep)
            plain7 = false;
            LZ_MATCHED(7, deep)
        deep:
            p0[64] = (L)b0.r1, p0[128] = (L)b0.r2, p0[192] = (L)b0.r3;
            p1[64] = (L)b1.r1, p1[128] = (L)b1.r2, p1[192] = (L)b1.r3;
            if (plain7) goto m7;
            goto m8;
        }
#undef LZ_MATCHED
    m1: tree_bit<1>(r, a, sym, lane);
    m2: tree_bit<2>(r, a, sym, lane);
    m3: tree_bit<3>(r, a, sym, lane);
    m4: tree_bit<4>(r, a, sym, lane);
    m5: tree_bit<5>(r, a, sym, lane);
    m6: tree_bit<6>(r, a, sym, lane);
    m7: tree_bit<7>(r, a, sym, lane);
    m8: p0[0] = (L)b0.r0;
        p1[0] = (L)b1.r0;
    } else {
        tree_bit<0>(r, a, sym, lane);
        tree_bit<1>(r, a, sym, lane);
        tree_bit<2>(r, a, sym, lane);
        tree_bit<3>(r, a, sym, lane);
        tree_bit<4>(r, a, sym, lane);
        tree_bit<5>(r, a, sym, lane);
        tree_bit<6>(r, a, sym, lane);
        tree_bit<7>(r, a, sym, lane);
    }
    tree_store(lit, sub, lane, a);
    return sym;
}
#endif

// lenCoder.decode, lzma_len_coder.go:28-46.
template <class P>
LZ_HD uint32_t len_decode(RangeDec &r, P *probs, uint32_t base, uint32_t pos_state) {
    if (rd_bit(r, probs, base + LEN_CHOICE) == 0)
        return bt_decode(r, probs, base + LEN_LOW + (pos_state << 3), 3);
    if (rd_bit(r, probs, base + LEN_CHOICE + 1) == 0)
        return kLenLow + bt_decode(r, probs, base + LEN_MID + (pos_state << 3), 3);
    return kLenLow + kLenMid + bt_decode(r, probs, base + LEN_HIGH, 8);
}

struct DecodeResult {
    int32_t status;
    uint64_t out_size;
};

// The output position is kept in 32 bits: a stream may produce at most this many bytes
// (more ends with ST_OUT_CAP).  Leaves room for position + longest match without overflow.
constexpr uint32_t kMaxStreamOut = 0xFFFFFE00u;

// Decodes one stream.  `probs` holds kNumBaseProbs models followed (when
// lit == probs + P_LIT) by the literal coder; `lit` may also point to a separate
// array.  All models must already be initialised to kProbInit.
// lane/nlanes: the calling lane's index and the number of lanes executing this
// call together (64 on the GPU, 1 on the host).
// LIT_REGS (GPU only): the literal models are in HBM and are walked in registers.
template <bool LIT_REGS = false, class P, class L>
LZ_HD DecodeResult decode_stream(const uint8_t *in, uint64_t n, const Header &h, P *probs, L *lit,
                                 uint8_t *out, uint64_t out_cap, uint32_t lane, uint32_t nlanes) {
    DecodeResult res;
    res.status = ST_OK;
    res.out_size = 0;
    // newRangeDecoder reads 5 bytes after the header (range_coder.go:40-55).
    if (n < kHeaderSize + 5) {
        res.status = ST_TRUNCATED;
        return res;
    }
    RangeDec r;
    r.in = in;
    r.n = n;
    r.p = kHeaderSize;
    r.win = kWinEmpty;
    r.range = 0xFFFFFFFFu;
    r.code = 0;
    r.eof = 0;
    // Output may grow up to `lim`.  Running out of input sets it to zero (rd_refill), so the one
    // test made before a symbol is written also catches a truncated stream, where the
    // reference stops with the symbols decoded so far (range_coder.go:82-85).
    r.lim = out_cap < kMaxStreamOut ? (uint32_t)out_cap : kMaxStreamOut;
    for (int i = 0; i < 5; i++) r.code = (r.code << 8) | rd_byte(r);

    // distances from 2^31 on are the end marker or an error (lzma_decoder.go:283-288); folded
    // into the dictionary limit so that one comparison rejects all of them
    const uint32_t dict_check = h.dict_size > 0x80000000u ? 0x80000000u : h.dict_size > 1 ? h.dict_size : 1;
    const uint32_t ps_mask = (1u << h.pb) - 1, lp_mask = (1u << h.lp) - 1, lc = h.lc;
    const int64_t unpack = h.unpack_size;
    const uint32_t last = unpack < 0 || unpack > (int64_t)kMaxStreamOut ? kMaxStreamOut : (uint32_t)unpack;
    uint32_t state = 0, rep0 = 0, rep1 = 0, rep2 = 0, rep3 = 0;
    uint32_t now = 0;
    uint32_t prev = 0;
    uint32_t match_byte = 0;  // the byte the last match would have continued with
    int32_t status = ST_OK;
    bool marker = false;

    // The loop has one exit, its condition: whatever ends the stream early sets `end` to zero.
    // (With several exits the compiler routes each through a selector variable that is tested
    // again and again on the way out, on every symbol.)
    uint32_t end = last;
    while (now < end) {
        uint32_t ps = now & ps_mask;
        if (rd_bit(r, probs, P_IS_MATCH + (state << kPosStatesBitsMax) + ps) == 0) {
            // litCoder.getSubCoder, lzma_lit_coder.go:126-128
            uint32_t sub = (((now & lp_mask) << lc) + (prev >> (8 - lc))) * 0x300u;
            uint32_t sym = 1;
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
            if (LIT_REGS) {
                sym = lit_decode_regs(r, lit, sub, !st_is_char(state), match_byte, lane);
            } else
#endif
            if (!st_is_char(state)) {
                uint32_t mb = match_byte;
                while (sym < 0x100) {
                    uint32_t mbit = (mb >> 7) & 1;
                    mb <<= 1;
                    uint32_t b = rd_bit(r, lit, sub + ((1 + mbit) << 8) + sym);
                    sym = (sym << 1) | b;
                    if (mbit != b) {
                        while (sym < 0x100) sym = (sym << 1) | rd_bit(r, lit, sub + sym);
                        break;
                    }
                }
            } else {
                while (sym < 0x100) sym = (sym << 1) | rd_bit(r, lit, sub + sym);
            }
            prev = sym & 0xFF;
            if (now < r.lim) {
                if (lane == 0) out[now] = (uint8_t)prev;
                now++;
                state = st_char(state);
            } else {
                status = r.eof ? ST_TRUNCATED : ST_OUT_CAP;
                end = 0;
            }
        } else {
            uint32_t len;
            if (rd_bit(r, probs, P_IS_REP + state) == 1) {
                len = 0;
                if (rd_bit(r, probs, P_IS_REP_G0 + state) == 0) {
                    if (rd_bit(r, probs, P_IS_REP0_LONG + (state << kPosStatesBitsMax) + ps) == 0) {
                        state = st_shortrep(state);
                        len = 1;
                    }
                } else {
                    uint32_t d;
                    if (rd_bit(r, probs, P_IS_REP_G1 + state) == 0)
                        d = rep1;
                    else {
                        if (rd_bit(r, probs, P_IS_REP_G2 + state) == 0)
                            d = rep2;
                        else {
                            d = rep3;
                            rep3 = rep2;
                        }
                        rep2 = rep1;
                    }
                    rep1 = rep0;
                    rep0 = d;
                }
                if (len == 0) {
                    len = len_decode(r, probs, P_REP_LEN, ps) + kMatchMin;
                    state = st_rep(state);
                }
            } else {
                rep3 = rep2;
                rep2 = rep1;
                rep1 = rep0;
                len = len_decode(r, probs, P_LEN, ps) + kMatchMin;
                state = st_match(state);
                uint32_t slot =
                    bt_decode(r, probs, P_POS_SLOT + (len_to_pos_state(len) << kPosSlotBits), kPosSlotBits);
                if (slot >= kStartPosModel) {
                    uint32_t nd = (slot >> 1) - 1;
                    rep0 = (2 | (slot & 1)) << nd;
                    if (slot < kEndPosModel) {
                        rep0 += bt_rev_decode(r, probs, P_POS_SPEC + rep0 - slot - 1, nd);
                    } else {
                        rep0 += rd_direct(r, nd - kAlignBits) << kAlignBits;
                        rep0 += bt_rev_decode(r, probs, P_ALIGN, kAlignBits);
                    }
                } else
                    rep0 = slot;
            }
            if (rep0 >= (now < dict_check ? now : dict_check)) {
                // too far back, or (from 2^31 on) the end marker 0xFFFFFFFF
                if (r.eof)
                    status = ST_TRUNCATED;
                else if (rep0 == 0xFFFFFFFFu)
                    marker = true;
                else
                    status = ST_STREAM;
                end = 0;
            } else if (now + len > r.lim) {
                status = r.eof ? ST_TRUNCATED : ST_OUT_CAP;
                end = 0;
            } else {
                // lzOutWindow.copyBlock (lz_window.go:48-64) copies byte by byte, so a
                // source that overlaps the destination repeats with period D = rep0 + 1.
                // Byte i of the copy therefore equals out[now - D + (i mod D)], which only
                // reads bytes written before this match: the lanes can copy in parallel.
                // Element `len` of the same sequence is the byte the match would have continued
                // with, which a literal that follows is coded against (lzma_decoder.go:248-252);
                // it is fetched along with the copy.
                const uint32_t D = rep0 + 1;
                const uint8_t *src = out + (now - D);
                uint8_t *dst = out + now;
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
                // make earlier single-lane literal stores visible to every lane
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                uint32_t v = 0;
                if (len < nlanes && len < D) {  // one pass, nothing repeats
                    if (lane <= len) {
                        v = src[lane];
                        if (lane < len) dst[lane] = (uint8_t)v;
                    }
                } else {
                    for (uint32_t i = lane; i <= len; i += nlanes) {
                        v = src[i < D ? i : i % D];
                        if (i < len) dst[i] = (uint8_t)v;
                    }
                }
                prev = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)((len - 1) & 63u));
                match_byte = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(len & 63u));
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#else
                for (uint32_t i = 0; i < len; i++) dst[i] = src[i < D ? i : i % D];
                prev = dst[len - 1];
                match_byte = src[len < D ? len : len % D];
#endif
                now += len;
            }
        }
    }
    if (status == ST_OK && !marker && now == last && (unpack < 0 || unpack > (int64_t)last)) status = ST_OUT_CAP;  // longer than supported
    res.status = status;
    res.out_size = now;
    return res;
}

}  // namespace lz
