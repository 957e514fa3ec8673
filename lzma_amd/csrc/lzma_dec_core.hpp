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

// Range decoder state (rangeDecoder, range_coder.go:27-31).  Input bytes are
// fetched eight at a time; `p` is the offset of the next unread byte.
struct RangeDec {
    const uint8_t *in;
    uint64_t n;   // stream length in bytes
    uint64_t p;   // next byte to hand out
    uint64_t buf; // up to 8 prefetched bytes, next byte in the low 8 bits
    uint32_t nbuf;
    uint32_t range, code;
    uint32_t eof;  // set once a read past the end was attempted
};

LZ_HD void rd_refill(RangeDec &r) {
    uint64_t left = r.n - r.p;
    if (left >= 8) {
        uint64_t v;
        __builtin_memcpy(&v, r.in + r.p, 8);
        r.buf = v;
        r.nbuf = 8;
    } else {
        uint64_t v = 0;
        for (uint64_t i = 0; i < left; i++) v |= (uint64_t)r.in[r.p + i] << (8 * i);
        r.buf = v;
        r.nbuf = (uint32_t)left;
    }
}
LZ_HD uint32_t rd_byte(RangeDec &r) {
    if (r.nbuf == 0) {
        rd_refill(r);
        if (r.nbuf == 0) {
            r.eof = 1;
            return 0;
        }
    }
    uint32_t b = (uint32_t)r.buf & 0xFF;
    r.buf >>= 8;
    r.nbuf--;
    r.p++;
    return b;
}
LZ_HD void rd_normalize(RangeDec &r) {
    if (r.range < kTopValue) {
        r.code = (r.code << 8) | rd_byte(r);
        r.range <<= 8;
    }
}
// decodeBit, range_coder.go:75-105.
template <class P>
LZ_HD uint32_t rd_bit(RangeDec &r, P *probs, uint32_t idx) {
    uint32_t pr = uni(probs[idx]);
    uint32_t bound = (r.range >> kProbBits) * pr;
    uint32_t bit;
    if (r.code < bound) {
        r.range = bound;
        probs[idx] = (prob_t)(pr + ((kProbTotal - pr) >> kMoveBits));
        bit = 0;
    } else {
        r.range -= bound;
        r.code -= bound;
        probs[idx] = (prob_t)(pr - (pr >> kMoveBits));
        bit = 1;
    }
    rd_normalize(r);
    return bit;
}
// decodeDirectBits, range_coder.go:57-73.
LZ_HD uint32_t rd_direct(RangeDec &r, uint32_t nbits) {
    uint32_t res = 0;
    for (uint32_t i = nbits; i != 0; i--) {
        r.range >>= 1;
        uint32_t t = (r.code - r.range) >> 31;
        r.code -= r.range & (t - 1);
        res = (res << 1) | (1 - t);
        rd_normalize(r);
    }
    return res;
}
// rangeBitTreeCoder.decode / reverseDecode, range_bit_tree_coder.go:19-51.
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
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
// ---- literal coder walked in registers (GPU only)
// The literal models are too large to keep every stream's in LDS without giving up most
// of the streams a CU could hold, and decoding a literal from models in HBM one bit at a
// time would cost a memory round trip per bit.  Instead the 256 models of a literal bit
// tree are fetched at once, two 16-bit models per 32-bit word, word j of the tree in lane
// j and word j + 64 in a second register; the walk picks each model out of the lanes with
// a readlane, the lane that holds it keeps the updated value, and the words that changed
// are stored back when the symbol is complete.
struct TreeW {
    uint32_t w0, w1;
    bool d0, d1;  // this lane's word was updated
};
// the models base .. base + 255 (node k of the tree is model base + k); base is even
template <class L>
LZ_D TreeW tree_load(const L *lit, uint32_t base, uint32_t lane) {
    const uint32_t *wp = (const uint32_t *)(lit + base);
    TreeW t;
    t.w0 = wp[lane];
    t.w1 = wp[lane + 64];
    t.d0 = t.d1 = false;
    return t;
}
template <class L>
LZ_D void tree_store(L *lit, uint32_t base, uint32_t lane, const TreeW &t) {
    uint32_t *wp = (uint32_t *)(lit + base);
    if (t.d0) wp[lane] = t.w0;
    if (t.d1) wp[lane + 64] = t.w1;
}
// decodeBit (range_coder.go:75-105) on node k of a loaded tree
LZ_D uint32_t tree_bit(RangeDec &r, TreeW &t, uint32_t k, uint32_t lane) {
    const uint32_t word = k >> 1, src = word & 63u, sh = (k & 1u) << 4;
    const bool second = word >= 64;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)(second ? t.w1 : t.w0), (int)src);
    const uint32_t pr = (w >> sh) & 0xFFFFu;
    const uint32_t bound = (r.range >> kProbBits) * pr;
    uint32_t bit, np;
    if (r.code < bound) {
        r.range = bound;
        np = pr + ((kProbTotal - pr) >> kMoveBits);
        bit = 0;
    } else {
        r.range -= bound;
        r.code -= bound;
        np = pr - (pr >> kMoveBits);
        bit = 1;
    }
    const uint32_t nw = (w & ~(0xFFFFu << sh)) | (np << sh);
    if (lane == src) {
        if (second) t.w1 = nw, t.d1 = true; else t.w0 = nw, t.d0 = true;
    }
    rd_normalize(r);
    return bit;
}
// litSubCoder.decodeNormal / decodeWithMatchByte, lzma_lit_coder.go:18-44
template <class L>
LZ_D uint32_t lit_decode_regs(RangeDec &r, L *lit, uint32_t sub, bool matched, uint32_t mb, uint32_t lane) {
    TreeW a = tree_load(lit, sub, lane);
    uint32_t sym = 1;
    if (matched) {
        TreeW b0 = tree_load(lit, sub + 0x100u, lane), b1 = tree_load(lit, sub + 0x200u, lane);
        while (sym < 0x100) {
            const uint32_t mbit = (mb >> 7) & 1;
            mb <<= 1;
            const uint32_t b = mbit ? tree_bit(r, b1, sym, lane) : tree_bit(r, b0, sym, lane);
            sym = (sym << 1) | b;
            if (mbit != b) break;
        }
        tree_store(lit, sub + 0x100u, lane, b0);
        tree_store(lit, sub + 0x200u, lane, b1);
    }
    while (sym < 0x100) sym = (sym << 1) | tree_bit(r, a, sym, lane);
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
    r.buf = 0;
    r.nbuf = 0;
    r.range = 0xFFFFFFFFu;
    r.code = 0;
    r.eof = 0;
    for (int i = 0; i < 5; i++) r.code = (r.code << 8) | rd_byte(r);

    const uint32_t dict_check = h.dict_size > 1 ? h.dict_size : 1;
    const uint32_t ps_mask = (1u << h.pb) - 1, lp_mask = (1u << h.lp) - 1, lc = h.lc;
    const int64_t unpack = h.unpack_size;
    uint32_t state = 0, rep0 = 0, rep1 = 0, rep2 = 0, rep3 = 0;
    uint64_t now = 0;
    uint32_t prev = 0;
    int32_t status = ST_OK;

    while (unpack < 0 || (int64_t)now < unpack) {
        uint32_t ps = (uint32_t)now & ps_mask;
        if (rd_bit(r, probs, P_IS_MATCH + (state << kPosStatesBitsMax) + ps) == 0) {
            // litCoder.getSubCoder, lzma_lit_coder.go:126-128
            uint32_t sub = ((((uint32_t)now & lp_mask) << lc) + (prev >> (8 - lc))) * 0x300u;
            uint32_t sym = 1;
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
            if (LIT_REGS) {
                const bool matched = !st_is_char(state);
                uint32_t mb = 0;
                if (matched && (uint64_t)rep0 < now) mb = uni(out[now - 1 - rep0]);
                sym = lit_decode_regs(r, lit, sub, matched, mb, lane);
            } else
#endif
            if (!st_is_char(state)) {
                uint32_t mb = (uint64_t)rep0 < now ? uni(out[now - 1 - rep0]) : 0;
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
            if (r.eof) {
                status = ST_TRUNCATED;
                break;
            }
            prev = sym & 0xFF;
            if (now >= out_cap) {
                status = ST_OUT_CAP;
                break;
            }
            if (lane == 0) out[now] = (uint8_t)prev;
            now++;
            state = st_char(state);
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
                        if ((int32_t)rep0 < 0) {
                            if (r.eof) {
                                status = ST_TRUNCATED;
                                break;
                            }
                            if (rep0 == 0xFFFFFFFFu) break;  // end marker
                            status = ST_STREAM;
                            break;
                        }
                    }
                } else
                    rep0 = slot;
            }
            if (r.eof) {
                status = ST_TRUNCATED;
                break;
            }
            if ((uint64_t)rep0 >= now || rep0 >= dict_check) {
                status = ST_STREAM;
                break;
            }
            if (now + len > out_cap) {
                status = ST_OUT_CAP;
                break;
            }
            // lzOutWindow.copyBlock (lz_window.go:48-64) copies byte by byte, so a
            // source that overlaps the destination repeats with period D = rep0 + 1.
            // Byte i of the copy therefore equals out[now - D + (i mod D)], which only
            // reads bytes written before this match: the lanes can copy in parallel.
            {
                const uint64_t D = (uint64_t)rep0 + 1;
                const uint8_t *src = out + now - D;
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
                // make earlier single-lane literal stores visible to every lane
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#endif
                for (uint32_t i = lane; i < len; i += nlanes) {
                    uint64_t k = i < D ? i : i % D;
                    out[now + i] = src[k];
                }
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#endif
                now += len;
                uint32_t last = len - 1;
                prev = uni(src[last < D ? last : last % D]);
            }
        }
    }
    res.status = status;
    res.out_size = now;
    return res;
}

}  // namespace lz
