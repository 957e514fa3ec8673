/*
 * CPU oracle for the LZMA hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the algorithm of the Go package at
 * /root/reference (itchio/lzma, itself a transliteration of the Java LZMA SDK
 * 4.65).  It exists to check the HIP kernels and to serve as the timed CPU
 * baseline.  Nothing in the shipped library may link against or call it; only
 * tests/, the smoke check and the benchmark's cpu_baseline leg do.
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks this code byte for byte
 * against every known-answer vector the reference's own tests hold
 * (lzma_data_test.go:21-289, data/data.txt <-> data/data.eos.l3.lzma), in both
 * directions.
 */
#ifndef LZMA_ORACLE_H
#define LZMA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes (decoder errors follow lzma_decoder.go:91-95). */
enum {
    LZO_OK = 0,
    LZO_ERR_HEADER = 1,    /* headerError: bad props byte                     */
    LZO_ERR_STREAM = 2,    /* streamError: corrupt range-coded payload        */
    LZO_ERR_TRUNCATED = 3, /* input ended early (io.EOF in the reference)     */
    LZO_ERR_OUT_CAP = 4,   /* caller's output buffer too small                */
    LZO_ERR_ARG = 5,       /* argumentValueError (checkValues, level, size)   */
    LZO_ERR_NOMEM = 6
};

/* Explicit encoder parameters.  The reference only exposes `level`
 * (lzma_encoder.go:97-108); lzo_params_from_level() fills this struct from
 * that table.  cut_value == 0 means the reference default 16 + fb/2
 * (lz_bin_tree.go:41). */
typedef struct {
    uint32_t dict_log;  /* 12..29 */
    uint32_t fb;        /* 5..273, "fast bytes" == matchMaxLen of the finder */
    uint32_t lc, lp, pb;
    uint32_t cut_value; /* 0 = default */
    int32_t eos;        /* 1: size field = -1 and end marker written         */
} lzo_params;

int lzo_params_from_level(int level, int eos, lzo_params *p);

/* Upper bound on the encoded size of n input bytes (header included). */
size_t lzo_encode_bound(size_t n);

/* One .lzma stream (13-byte header + payload) from n bytes.
 * If p->eos == 0 the header's size field is n. */
int lzo_encode(const uint8_t *in, size_t n, const lzo_params *p, uint8_t *out,
               size_t out_cap, size_t *out_n);

/* Decode one .lzma stream.  *out_n receives the number of bytes produced,
 * *in_used (optional) the number of input bytes consumed. */
int lzo_decode(const uint8_t *in, size_t n, uint8_t *out, size_t out_cap,
               size_t *out_n, size_t *in_used);

/* Read just the header: returns LZO_OK / LZO_ERR_HEADER / LZO_ERR_TRUNCATED. */
int lzo_parse_header(const uint8_t *in, size_t n, uint32_t *lc, uint32_t *lp,
                     uint32_t *pb, uint32_t *dict_size, int64_t *unpack_size);

/* BT4 match finder run over every position of `data` in order, as the encoder
 * would (getMatches at each position).  counts[i] = number of (len, dist-1)
 * pairs at position i; pairs are appended to `pairs` as u32 len, u32 dist-1.
 * If pairs == NULL only counts/total are produced.  cut_value 0 = default. */
int lzo_bt4_matches(const uint8_t *data, size_t n, uint32_t dict_size,
                    uint32_t fb, uint32_t cut_value, uint32_t *counts,
                    uint32_t *pairs, size_t pairs_cap_u32, size_t *total_pairs);

#ifdef __cplusplus
}
#endif
#endif
