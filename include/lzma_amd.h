/*
 * lzma_amd -- MI355X (gfx950) LZMA engine, C ABI.
 *
 * This is the drop-in boundary for the hot path of the Go package itchio/lzma
 * (paths below are relative to that package's root).  Each entry point names the
 * reference interface it replaces.  Plain pointers and sizes only; blocking;
 * callable from any thread; nothing is retained after a call returns (cgo
 * pointer rule).
 *
 * Two families of entry points:
 *   - host-buffer calls (lzma_amd_*): take host memory, stage through HBM.
 *   - device calls (lzma_amd_*_dev): inputs/outputs already resident in HBM on
 *     the current HIP device; `stream` is a hipStream_t passed as void*
 *     (NULL = default stream).  They enqueue work and synchronize the stream
 *     before returning unless stated otherwise.
 */
#ifndef LZMA_AMD_H
#define LZMA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Per-call / per-stream status.  Decoder values map onto the reference's error
 * values: HEADER = headerError, STREAM = streamError (lzma_decoder.go:91-95),
 * TRUNCATED = the io.EOF the reference surfaces when the payload ends early
 * (range_coder.go:82-85), ARG = *argumentValueError (lzma_encoder.go:26-34). */
enum {
    LZMA_AMD_OK = 0,
    LZMA_AMD_ERR_HEADER = 1,
    LZMA_AMD_ERR_STREAM = 2,
    LZMA_AMD_ERR_TRUNCATED = 3,
    LZMA_AMD_ERR_OUT_CAP = 4,
    LZMA_AMD_ERR_ARG = 5,
    LZMA_AMD_ERR_NOMEM = 6,
    LZMA_AMD_ERR_HIP = 7,
    LZMA_AMD_ERR_UNSUPPORTED = 8
};

/* Encoder parameters: the fields of the reference's compressionLevel
 * (lzma_encoder.go:85-94) made explicit, because two of the benchmark
 * configurations are not rows of its `levels` table (lzma_encoder.go:97-108).
 * Ranges are those of checkValues (lzma_encoder.go:110-129). */
typedef struct {
    uint32_t dict_log; /* 12..29; dictionary = 1 << dict_log                 */
    uint32_t fb;       /* 5..273; fast bytes = matchMaxLen of the BT4 finder */
    uint32_t lc, lp, pb;
    uint32_t cut_value; /* reserved: the GPU finder is always exhaustive
                           (equivalent to lzBinTree.cutValue = infinity,
                           lz_bin_tree.go:26,41,200); must be 0              */
    int32_t eos;        /* 1: size field -1 and end marker (size == -1 in
                           NewWriterSizeLevel, lzma_encoder.go:1012-1015)    */
} lzma_amd_params;

/* levels[level] of lzma_encoder.go:97-108; returns LZMA_AMD_ERR_ARG outside
 * 1..9 as encoder() does (lzma_encoder.go:997-999). */
int lzma_amd_params_from_level(int level, int eos, lzma_amd_params *p);

/* Number of usable HIP devices (0 if none). */
int lzma_amd_device_count(void);
/* Select the device used by the calling thread's subsequent calls. */
int lzma_amd_set_device(int device);
/* Message for the calling thread's most recent failing call ("" if none). */
const char *lzma_amd_last_error(void);
const char *lzma_amd_version(void);

/* Worst-case encoded size of one chunk of n bytes, header included. */
size_t lzma_amd_encode_bound(size_t n);

/* ---------------------------------------------------------------- match finder
 * Replaces lzBinTree.getMatches called at every position of one stream
 * (lz_bin_tree.go:115-253; skip, :255-337, performs the same insertions), for
 * numHashBytes = 4, with the tree search exhaustive.  For each position i of
 * data[0..n): counts[i] = number of pairs, and the pairs of position i are
 * pairs[2*k], pairs[2*k+1] = (length, distance-1) for k in
 * [offsets[i], offsets[i]+counts[i]), lengths strictly increasing -- the same
 * values and order the reference writes into `distances`.
 * offsets may be NULL.  pairs_cap is in pairs; *total_pairs is always set, and
 * LZMA_AMD_ERR_OUT_CAP is returned if it exceeds pairs_cap (counts still valid).
 */
int lzma_amd_bt4_matches(const uint8_t *data, size_t n, uint32_t dict_size, uint32_t fb,
                         uint32_t *counts, uint64_t *offsets, uint32_t *pairs,
                         size_t pairs_cap, size_t *total_pairs);

/* Same, all pointers in HBM.  d_total_pairs is a device uint64. */
int lzma_amd_bt4_matches_dev(const uint8_t *d_data, size_t n, uint32_t dict_size, uint32_t fb,
                             uint32_t *d_counts, uint64_t *d_offsets, uint32_t *d_pairs,
                             size_t pairs_cap, uint64_t *d_total_pairs, void *stream);

/* --------------------------------------------------------------------- encoder
 * Replaces encoder.encoder() run once per chunk (lzma_encoder.go:989-1107:
 * header, getOptimum/backward optimal parse :331-797, codeOneBlock :859-978,
 * range coder range_coder.go:131-220): chunk c is in[in_off[c]..in_off[c+1])
 * and becomes one complete .lzma stream (13-byte header + payload) written at
 * out + out_off[c], of out_sizes[c] bytes; out_off[c+1]-out_off[c] is the
 * capacity given to chunk c (use lzma_amd_encode_bound).  status[c] is per
 * chunk.  Every chunk is an independent stream, as if passed to its own
 * NewWriterSizeLevel.  The header's size field is the chunk length unless
 * p->eos.  In the device form in_total is the number of bytes addressable from
 * d_in; offsets beyond it are rejected with LZMA_AMD_ERR_ARG.
 */
int lzma_amd_encode_chunks(const uint8_t *in, const uint64_t *in_off, size_t n_chunks,
                           const lzma_amd_params *p, uint8_t *out, const uint64_t *out_off,
                           uint64_t *out_sizes, int32_t *status);

int lzma_amd_encode_chunks_dev(const uint8_t *d_in, const uint64_t *d_in_off, size_t n_chunks,
                               size_t in_total, const lzma_amd_params *p, uint8_t *d_out,
                               const uint64_t *d_out_off, uint64_t *d_out_sizes,
                               int32_t *d_status, void *stream);

/* The encoder keeps its streams and device memory (match-finder tables and match
 * lists, up to about 230 bytes per input byte of the largest batch of chunks seen;
 * batches are at most 1 GiB) with the calling thread from one call to the next,
 * because allocating and releasing that much takes seconds; lzma_amd_bt4_matches
 * keeps its tables and lists (about 120 bytes per input byte) the same way, until
 * the thread's next encoder call.  Every entry point may be called from any thread,
 * but what the threads keep adds up: the environment variable
 * LZMA_AMD_WORKSPACE_GIB bounds what one encoder thread may hold (its batches
 * shrink to fit), for processes that run several encoder threads, or an encoder
 * next to decode calls, on one device.  This call gives back what the calling
 * thread keeps.  No counterpart in the reference (its encoder state lives and dies
 * with each writer, lzma_encoder.go:1120-1136). */
void lzma_amd_release_workspace(void);

/* --------------------------------------------------------------------- decoder
 * Replaces decoder.decoder() run once per stream (lzma_decoder.go:306-349;
 * doDecode :218-304; range decoder range_coder.go:27-105; output window
 * lz_window.go:9-80): stream s is in[in_off[s]..in_off[s+1]), a complete .lzma
 * stream including its 13-byte header; its output goes to out + out_off[s]
 * with capacity out_off[s+1]-out_off[s]; out_sizes[s] receives the number of
 * bytes produced and status[s] the per-stream status.  On a non-OK status the
 * bytes produced before the error are left in place (the reference would have
 * delivered only whole flushed windows of them).  One stream may produce at most
 * 4 GiB - 512 bytes; a longer one ends with LZMA_AMD_ERR_OUT_CAP.
 */
int lzma_amd_decode_batch(const uint8_t *in, const uint64_t *in_off, size_t n_streams,
                          uint8_t *out, const uint64_t *out_off, uint64_t *out_sizes,
                          int32_t *status);

int lzma_amd_decode_batch_dev(const uint8_t *d_in, const uint64_t *d_in_off, size_t n_streams,
                              uint8_t *d_out, const uint64_t *d_out_off, uint64_t *d_out_sizes,
                              int32_t *d_status, void *stream);

/* lzma_amd_decode_batch_dev for callers that cannot allocate device memory: code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc.  It has no
 * counterpart in the reference.  Same arguments, results and statuses as
 * lzma_amd_decode_batch_dev, but what the kernels need beyond the caller's buffers (the
 * workgroups' literal models, the list of streams with lc+lp > 3 that a second launch
 * decodes) lives in a device array of the library's own, about 58 MiB on every GPU the
 * library has run this call on, kept until the process ends; the call itself makes no
 * hipMalloc / hipFree.  Unlike every other entry point it does not wait for the device
 * either: it enqueues a 256-byte memset and two kernels on `stream` and returns, so it can be
 * recorded into a HIP graph; outputs, sizes and statuses are valid once the stream (or the
 * graph launch) has completed.  The array's address is looked up once per device, by
 * lzma_amd_set_device or by the first call on the device: do one of the two before a capture
 * begins.  Limits: at most 2^20 streams per call (more: LZMA_AMD_ERR_ARG); 4096 workgroups in
 * the first launch, so about half the throughput of the allocating call on large batches;
 * streams with lc+lp > 3 (lc = 4 among them, which the allocating call decodes in its wide
 * first launch) share nine workgroups in the second launch: a batch made of such streams
 * decodes at a few hundredths of the allocating call's rate.  lzma_amd_last_timing reports
 * zeros after this call.  Every call on a GPU uses that GPU's one array: calls that overlap in
 * time on the same GPU (from several threads, streams or graph launches) race and must be
 * serialised by the caller. */
int lzma_amd_decode_batch_dev_noalloc(const uint8_t *d_in, const uint64_t *d_in_off, size_t n_streams,
                                      uint8_t *d_out, const uint64_t *d_out_off, uint64_t *d_out_sizes,
                                      int32_t *d_status, void *stream);

/* Header fields of one stream (props.decodeProps, lzma_decoder.go:175-190, and
 * the size field :318-322), host-side helper for sizing decode outputs. */
int lzma_amd_parse_header(const uint8_t *in, size_t n, uint32_t *lc, uint32_t *lp, uint32_t *pb,
                          uint32_t *dict_size, int64_t *unpack_size);

/* Milliseconds of device time (hipEvent) spent in the kernels of the calling
 * thread's most recent *_dev call; any pointer may be NULL.  ms_total is the
 * span from the first to the last kernel; ms_finder and ms_coder are sums over
 * the match-finder batches and over the coder launches.  The encoder overlaps
 * its phases and runs several coder launches at once, so the sums exceed the
 * span.  No counterpart in the reference (measurement only). */
void lzma_amd_last_timing(float *ms_total, float *ms_finder, float *ms_coder);
/* Number of coder kernel launches ms_coder sums over (0 for the finder). */
int lzma_amd_last_coder_launches(void);
/* How many of them ran two waves per chunk (encoder only; see lzma_amd_encode_chunks_dev). */
int lzma_amd_last_coop_launches(void);

#ifdef __cplusplus
}
#endif
#endif
