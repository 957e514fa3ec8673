// Shared constants and small helpers for the MI355X LZMA kernels.
//
// The format constants restate lzma_decoder.go:57-89, range_coder.go:12-17,116-119,
// lzma_encoder.go:200-208 and lz_bin_tree.go:9-17 of the reference package.
//
// Everything in this header compiles both as gfx950 device code (hipcc) and as
// plain host C++ (g++, with LZ_HOSTSIM defined), so that the serial parts of the
// kernels can be unit-tested on a machine without a GPU.  The host build is a
// test vehicle only; the shipped library contains the device build alone.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LZ_HD __host__ __device__ __forceinline__
#define LZ_D __device__ __forceinline__
#else
#define LZ_HD inline
#define LZ_D inline
#endif

namespace lz {

typedef uint16_t prob_t;

constexpr uint32_t kHeaderSize = 13;
constexpr uint32_t kNumReps = 4;
constexpr uint32_t kNumStates = 12;
constexpr uint32_t kPosSlotBits = 6;
constexpr uint32_t kNumLenToPos = 4;
constexpr uint32_t kMatchMin = 2;
constexpr uint32_t kAlignBits = 4;
constexpr uint32_t kAlignSize = 16;
constexpr uint32_t kAlignMask = 15;
constexpr uint32_t kStartPosModel = 4;
constexpr uint32_t kEndPosModel = 14;
constexpr uint32_t kNumFullDist = 128;
constexpr uint32_t kPosStatesBitsMax = 4;
constexpr uint32_t kPosStatesMax = 16;
constexpr uint32_t kLenLow = 8, kLenMid = 8, kLenHigh = 256;
constexpr uint32_t kNumLenSyms = kLenLow + kLenMid + kLenHigh;  // 272
constexpr uint32_t kMatchMax = kMatchMin + kNumLenSyms - 1;     // 273
constexpr uint32_t kTopValue = 1u << 24;
constexpr uint32_t kProbBits = 11;
constexpr uint32_t kProbTotal = 1u << kProbBits;
constexpr uint32_t kProbInit = kProbTotal >> 1;
constexpr uint32_t kMoveBits = 5;
constexpr uint32_t kMoveReducingBits = 2;
constexpr uint32_t kPriceShift = 6;
constexpr uint32_t kInfPrice = 0x0FFFFFFFu;
constexpr uint32_t kNumOpts = 4096;
constexpr uint32_t kHash2Size = 1u << 10;
constexpr uint32_t kHash3Size = 1u << 16;

// Status codes, identical to include/lzma_amd.h.
enum : int32_t {
    ST_OK = 0, ST_HEADER = 1, ST_STREAM = 2, ST_TRUNCATED = 3, ST_OUT_CAP = 4, ST_ARG = 5,
    ST_NOMEM = 6, ST_HIP = 7, ST_UNSUPPORTED = 8
};

// Probability-model layout shared by encoder and decoder: one flat array of
// 11-bit probabilities.  Sizes follow lzma_decoder.go:330-345 / lzma_encoder.go:1051-1069.
// The literal coder (0x300 << (lc+lp) entries, lzma_lit_coder.go:112-124) comes last.
constexpr uint32_t P_IS_MATCH = 0;                                     // 12<<4 = 192
constexpr uint32_t P_IS_REP = P_IS_MATCH + (kNumStates << kPosStatesBitsMax);
constexpr uint32_t P_IS_REP_G0 = P_IS_REP + kNumStates;
constexpr uint32_t P_IS_REP_G1 = P_IS_REP_G0 + kNumStates;
constexpr uint32_t P_IS_REP_G2 = P_IS_REP_G1 + kNumStates;
constexpr uint32_t P_IS_REP0_LONG = P_IS_REP_G2 + kNumStates;          // 192
constexpr uint32_t P_POS_SLOT = P_IS_REP0_LONG + (kNumStates << kPosStatesBitsMax);  // 4*64
constexpr uint32_t P_POS_SPEC = P_POS_SLOT + (kNumLenToPos << kPosSlotBits);  // 114 (+1 pad)
constexpr uint32_t P_ALIGN = P_POS_SPEC + (kNumFullDist - kEndPosModel) + 2;  // 16
// length coder: choice[2], low[16][8], mid[16][8], high[256]  (lzma_len_coder.go:7-26)
constexpr uint32_t LEN_CHOICE = 0;
constexpr uint32_t LEN_LOW = 2;
constexpr uint32_t LEN_MID = LEN_LOW + (kPosStatesMax << 3);
constexpr uint32_t LEN_HIGH = LEN_MID + (kPosStatesMax << 3);
constexpr uint32_t LEN_CODER_SIZE = LEN_HIGH + 256;                    // 514
constexpr uint32_t P_LEN = P_ALIGN + kAlignSize;
constexpr uint32_t P_REP_LEN = P_LEN + LEN_CODER_SIZE;
constexpr uint32_t P_LIT = P_REP_LEN + LEN_CODER_SIZE;                 // literal coder base
constexpr uint32_t kNumBaseProbs = P_LIT;

// State machine, lzma_decoder.go:114-158.
LZ_HD uint32_t st_char(uint32_t s) { return s < 4 ? 0 : (s < 10 ? s - 3 : s - 6); }
LZ_HD uint32_t st_match(uint32_t s) { return s < 7 ? 7 : 10; }
LZ_HD uint32_t st_rep(uint32_t s) { return s < 7 ? 8 : 11; }
LZ_HD uint32_t st_shortrep(uint32_t s) { return s < 7 ? 9 : 11; }
LZ_HD bool st_is_char(uint32_t s) { return s < 7; }
LZ_HD uint32_t len_to_pos_state(uint32_t len) {
    len -= kMatchMin;
    return len < kNumLenToPos ? len : kNumLenToPos - 1;
}

// props.decodeProps + size field, lzma_decoder.go:175-190, 318-322.
struct Header {
    uint32_t lc, lp, pb, dict_size;
    int64_t unpack_size;
};
LZ_HD int32_t parse_header(const uint8_t *in, uint64_t n, Header &h) {
    if (n < kHeaderSize) return ST_TRUNCATED;
    uint32_t d = in[0];
    if (d > 9 * 5 * 5) return ST_HEADER;
    h.lc = d % 9;
    d /= 9;
    h.pb = d / 5;
    h.lp = d % 5;
    if (h.lc > 8 || h.lp > 4 || h.pb > 4) return ST_HEADER;
    uint32_t ds = 0;
    for (int i = 0; i < 4; i++) ds += (uint32_t)in[1 + i] << (8 * i);
    h.dict_size = ds;
    uint64_t us = 0;
    for (int i = 0; i < 8; i++) us |= (uint64_t)in[5 + i] << (8 * i);
    h.unpack_size = (int64_t)us;
    return ST_OK;
}

// CRC-32 byte table entry (polynomial 0xEDB88320), lz_bin_tree.go:340-355.
LZ_HD uint32_t crc32_entry(uint32_t i) {
    uint32_t r = i;
    for (int j = 0; j < 8; j++) r = (r & 1) ? (r >> 1) ^ 0xEDB88320u : r >> 1;
    return r;
}

// hashMask of lzBinTree for a dictionary size, lz_bin_tree.go:59-71.
LZ_HD uint32_t bt4_hash_mask(uint32_t dict_size) {
    uint32_t hs = dict_size - 1;
    hs |= hs >> 1;
    hs |= hs >> 2;
    hs |= hs >> 4;
    hs |= hs >> 8;
    hs >>= 1;
    hs |= 0xFFFF;
    if (hs > (1u << 24)) hs >>= 1;
    return hs;
}

}  // namespace lz
