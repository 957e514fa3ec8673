// Per-chunk LZMA encoder for gfx950: every chunk becomes an independent .lzma
// stream, one 64-lane wavefront (= one workgroup) per chunk.
//
// Replaces encoder.encoder() of the reference (lzma_encoder.go:989-1107) run once
// per chunk.  Two phases per batch of chunks:
//   1. the exhaustive BT4 finder (matchfinder.hip) computes the match list of every
//      position of every chunk at full-chip throughput;
//   2. encode_kernel runs the optimal parse and the range coder (lzma_enc_core.hpp),
//      one wave per chunk, reading the lists as it goes.
// A chunk's wave is a long dependent chain (about 5 s per MiB), and the time of a chain
// hardly changes with the number of chains next to it, so throughput comes from how
// many chains a CU holds: the state kept in LDS is cut to 12.5 KiB at fb 128
// (non-literal models, 16-bit price tables sized by the number of fast bytes, a
// 64-entry window of the optimum table, the current match list), twelve chunks per CU
// (the kernel's registers allow three waves per SIMD); the literal models (12 KiB at
// lc 3) sit in HBM behind the L2, which costs a chain about 2 %.
// The phases of successive batches overlap: see lzma_amd_encode_chunks_dev.
//
// Roofline class: HBM in principle (N bytes in, C bytes out per chunk);
// serial-dependency-bound in practice.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "lzma_enc_core.hpp"
#include "mf_tables.hpp"

namespace {


// Match list of a chunk position, read from the finder's results.  The parse asks
// for consecutive positions almost all the time, so the words of the next position's
// list (one per lane) and the result word of the position after it are requested
// one call ahead and are usually there when needed.
// STEP: distance between the positions one wave asks for (2 when two waves share a chunk).
template <bool PACKED, uint32_t STEP = 1>
struct MfFunctor {
    static __device__ uint64_t uni64(uint64_t v) {
        return ((uint64_t)lz::U((uint32_t)(v >> 32)) << 32) | lz::U((uint32_t)v);
    }
    const uint64_t *meta;
    const uint32_t *arena;
    uint32_t job_start, lane, n_total;
    uint32_t pf_pos;
    uint64_t pf_meta, pf_meta2;
    uint32_t pf_word;
    // words of a list of c pairs, and the list's first word in the arena
    static __device__ uint32_t words(uint32_t c) { return PACKED ? c : 2 * c; }
    static __device__ uint64_t first(uint64_t m) { return PACKED ? lz::mf_meta_off(m) : 2 * lz::mf_meta_off(m); }
    // word k of a list goes to the (length, distance) array the parse reads
    static __device__ void put(LZ_LDS uint32_t *out, uint32_t k, uint32_t w) {
        if (PACKED) {
            out[2 * k] = lz::mf_pack_len(w);
            out[2 * k + 1] = lz::mf_pack_dist(w);
        } else {
            out[k] = w;
        }
    }
    __device__ uint32_t operator()(uint32_t pos, LZ_LDS uint32_t *out) {
        const bool hit = pos == pf_pos;
        const uint64_t m = uni64(hit ? pf_meta : meta[job_start + pos]);
        const uint32_t c = lz::mf_meta_count(m);
        if (hit && words(c) <= 64) {
            if (lane < words(c)) put(out, lane, pf_word);
        } else {
            const uint32_t *src = arena + first(m);
            for (uint32_t k = lane; k < words(c); k += 64) put(out, k, src[k]);
        }
        uint32_t g1 = job_start + pos + STEP, g2 = job_start + pos + 2 * STEP;
        if (g1 >= n_total) g1 = n_total - 1;
        if (g2 >= n_total) g2 = n_total - 1;
        const uint64_t m1 = uni64(hit ? pf_meta2 : meta[g1]);
        const uint32_t c1 = lz::mf_meta_count(m1);
        uint32_t w = 0;
        if (words(c1) <= 64 && lane < words(c1)) w = arena[first(m1) + lane];
        pf_word = w;
        pf_meta = m1;
        pf_meta2 = meta[g2];
        pf_pos = pos + STEP;
        return c;
    }
};

struct LdsLayout {
    uint32_t probs, prices, pp, opt, small, total;  // byte offsets
    uint32_t sync, own1;                            // cooperative workgroups only
    uint32_t lit;                                   // ... with the literal models in LDS
};
// waves > 1: a cooperative workgroup of that many waves, which keeps its literal models
// (lit_models of them) in LDS
__host__ __device__ inline LdsLayout lds_layout(uint32_t pb, uint32_t fb, uint32_t waves = 1, uint32_t lit_models = 0) {
    LdsLayout l;
    uint32_t nprobs = lz::kNumBaseProbs;
    l.probs = 0;
    l.prices = (l.probs + nprobs * 2 + 15) & ~15u;
    l.pp = (l.prices + lz::price_layout(pb, fb).total * 2 + 15) & ~15u;
    l.opt = l.pp + 512 * 2;
    l.small = l.opt + (lz::kRecWin + lz::kBksWin) * 16;
    l.total = l.small + lz::enc_small_words(fb, pb) * 4;
    l.sync = l.own1 = l.total;
    if (waves > 1) {  // the block the waves synchronise through, and the other waves' match lists and records
        l.own1 = l.sync + lz::SY_WORDS * 4;
        l.total = l.own1 + (waves - 1) * (lz::la_words(fb) + lz::mdist_words(fb)) * 4;
    }
    l.lit = (l.total + 15) & ~15u;
    if (lit_models) l.total = l.lit + lit_models * 2;
    return l;
}

// Chunks [c0, c0 + nchunks) of the call; their bytes start at data offset
// in_off[c] - base inside t.data.
// NW > 1: that many waves per chunk (workgroups of 64 * NW threads), for launches that leave
// wave slots free: the parse loop of a chunk runs on all of them (lz::Encoder::parse_loop),
// everything else on the first.  Their literal models live in LDS as well: a device that is not
// full has the LDS to spare, and every literal priced or coded saves a trip to memory.
template <bool PACKED, uint32_t NW>
__global__ __launch_bounds__(64 * NW) void encode_kernel(const uint8_t *__restrict__ data, uint32_t data_n,
                                                    const uint64_t *__restrict__ mf_meta,
                                                    const uint32_t *__restrict__ mf_arena,
                                                    const uint64_t *__restrict__ in_off,
                                                    uint64_t base, uint64_t c0, uint32_t nchunks,
                                                    lz::EncParams ep, uint8_t *__restrict__ out,
                                                    const uint64_t *__restrict__ out_off,
                                                    uint64_t *__restrict__ out_sizes,
                                                    int32_t *__restrict__ status, lz::OptG *opt_scratch,
                                                    lz::prob_t *lit_scratch, uint64_t lit_stride) {
    extern __shared__ __align__(16) uint8_t smem[];
    // A chunk's wave is one long dependent chain; the finder's waves sharing its SIMD are many
    // and interchangeable.  Issue priority to the chain: the finder fills the slots it leaves.
    __builtin_amdgcn_s_setprio(3);
    constexpr bool COOP = NW > 1, LITLDS = NW > 1;
    constexpr uint32_t kThreads = 64 * NW;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = COOP ? lz::U(threadIdx.x >> 6) : 0;
    const uint32_t lit_n = 0x300u << (ep.lc + ep.lp);
    const LdsLayout L = lds_layout(ep.pb, ep.fb, NW, LITLDS ? lit_n : 0u);
    LZ_LDS uint8_t *sm = (LZ_LDS uint8_t *)smem;
    LZ_LDS lz::prob_t *probs = (LZ_LDS lz::prob_t *)(sm + L.probs);
    LZ_LDS uint16_t *prices = (LZ_LDS uint16_t *)(sm + L.prices);
    LZ_LDS uint16_t *pp = (LZ_LDS uint16_t *)(sm + L.pp);
    LZ_LDS uint32_t *optl = (LZ_LDS uint32_t *)(sm + L.opt);
    LZ_LDS uint32_t *small = (LZ_LDS uint32_t *)(sm + L.small);
    LZ_LDS uint32_t *sync = (LZ_LDS uint32_t *)(sm + L.sync);
    typedef typename std::conditional<LITLDS, LZ_LDS lz::prob_t *, lz::prob_t *>::type LitPtr;
    lz::build_prob_prices((uint16_t *)pp, threadIdx.x, kThreads);

    for (uint32_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const uint64_t chunk = c0 + c;
        const uint32_t jb = (uint32_t)(in_off[chunk] - base), je = (uint32_t)(in_off[chunk + 1] - base);
        const uint64_t ob = out_off[chunk], oe = out_off[chunk + 1];
        if (oe - ob < lz::kHeaderSize) {
            if (threadIdx.x == 0) {
                status[chunk] = lz::ST_OUT_CAP;
                out_sizes[chunk] = 0;
            }
            continue;
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < lz::kNumBaseProbs; i += kThreads) probs[i] = lz::kProbInit;
        LitPtr lit;
        if constexpr (LITLDS)
            lit = (LZ_LDS lz::prob_t *)(sm + L.lit);
        else
            lit = lit_scratch + (uint64_t)blockIdx.x * lit_stride;
        for (uint32_t i = threadIdx.x; i < lit_n; i += kThreads) lit[i] = lz::kProbInit;
        if (COOP && threadIdx.x < lz::SY_WORDS) sync[threadIdx.x] = 0;
        __syncthreads();

        lz::Encoder<MfFunctor<PACKED, NW>, LitPtr, NW> e;
        e.data = data + jb;
        e.n = je - jb;
        e.readable = data_n - jb;
        e.probs = probs;
        e.lit = lit;
        e.prices = prices;
        e.pp = pp;
        e.o_rec = (LZ_LDS lz::Rec *)optl;
        e.o_bks = (LZ_LDS lz::Reps *)(optl + 4 * lz::kRecWin);
        e.sync = sync;
        e.opt_g = opt_scratch + (uint64_t)blockIdx.x * lz::kNumOpts;
        e.lane_id = lane;
        e.out = out + ob + lz::kHeaderSize;
        e.out_cap = oe - ob - lz::kHeaderSize;
        e.mf = MfFunctor<PACKED, NW>{mf_meta, mf_arena, jb, lane, data_n, lz::kNone, 0, 0, 0};
        e.ep = ep;
#if defined(LZ_PROFILE)
        for (int i = 0; i < 16; i++) e.prof[i] = 0;
        e.prof_last = __builtin_readcyclecounter();
#endif
        if (COOP && wave != 0) {
            e.helper_run(small, (LZ_LDS uint32_t *)(sm + L.own1) + (wave - 1) * (lz::la_words(ep.fb) + lz::mdist_words(ep.fb)), wave);
#if defined(LZ_PROFILE)
            if (lane == 0 && blockIdx.x == 0)
                printf("PROF helper: mf %llu | wait-S %llu | path %llu | prices %llu | round-loads %llu | round %llu | lit %llu | wait-A %llu | wait-M %llu | litcont %llu | ends %llu | apply %llu | loopctl %llu | other %llu | idle %llu\n",
                       (unsigned long long)e.prof[0], (unsigned long long)e.prof[12], (unsigned long long)e.prof[8], (unsigned long long)e.prof[1],
                       (unsigned long long)e.prof[11], (unsigned long long)e.prof[2], (unsigned long long)e.prof[9], (unsigned long long)e.prof[13],
                       (unsigned long long)e.prof[14], (unsigned long long)e.prof[10], (unsigned long long)e.prof[3], (unsigned long long)e.prof[4],
                       (unsigned long long)e.prof[5], (unsigned long long)e.prof[6], (unsigned long long)e.prof[15]);
#endif
            continue;
        }
        e.run(small);
#if defined(LZ_PROFILE)
        if (lane == 0 && blockIdx.x == 0)
            printf("PROF chunk n=%u: mf %llu | path %llu | prices %llu | round-loads %llu | round %llu | lit %llu | litcont %llu | ends %llu | apply %llu | loopctl %llu | getopt_other %llu | emit %llu\n",
                   e.n, (unsigned long long)e.prof[0], (unsigned long long)e.prof[8], (unsigned long long)e.prof[1],
                   (unsigned long long)e.prof[11], (unsigned long long)e.prof[2], (unsigned long long)e.prof[9], (unsigned long long)e.prof[10],
                   (unsigned long long)e.prof[3], (unsigned long long)e.prof[4], (unsigned long long)e.prof[5],
                   (unsigned long long)e.prof[6], (unsigned long long)e.prof[7]);
        if (COOP && lane == 0 && blockIdx.x == 0)
            printf("PROF wave 0: wait-S %llu | wait-A %llu | wait-M %llu | join %llu\n", (unsigned long long)e.prof[12],
                   (unsigned long long)e.prof[13], (unsigned long long)e.prof[14], (unsigned long long)e.prof[15]);
#endif
        if (lane == 0) {
            lz::write_header(out + ob, ep, ep.eos ? -1 : (int64_t)e.n);
            status[chunk] = e.dead ? lz::ST_HIP : (e.overflow ? lz::ST_OUT_CAP : lz::ST_OK);
            out_sizes[chunk] = lz::kHeaderSize + e.out_n;
        }
    }
}



int check_params(const lzma_amd_params *p) {  // checkValues, lzma_encoder.go:110-129
    if (!p) return lzh::fail(LZMA_AMD_ERR_ARG, "encode_chunks: null params");
    if (p->dict_log < 12 || p->dict_log > 29)
        return lzh::fail(LZMA_AMD_ERR_ARG, "dictionary size out of range: %u", p->dict_log);
    if (p->fb < 5 || p->fb > 273)
        return lzh::fail(LZMA_AMD_ERR_ARG, "number of fast bytes out of range: %u", p->fb);
    if (p->lc > 8) return lzh::fail(LZMA_AMD_ERR_ARG, "number of literal context bits out of range: %u", p->lc);
    if (p->lp > 4) return lzh::fail(LZMA_AMD_ERR_ARG, "number of literal position bits out of range: %u", p->lp);
    if (p->pb > 4) return lzh::fail(LZMA_AMD_ERR_ARG, "number of position bits out of range: %u", p->pb);
    if (p->cut_value != 0)
        return lzh::fail(LZMA_AMD_ERR_ARG, "cut_value must be 0: the finder is always exhaustive");
    return LZMA_AMD_OK;
}

constexpr uint64_t kMaxChunkBytes = 1ull << 30;  // largest single chunk
constexpr uint64_t kMinBatchBytes = 128ull << 20; // a batch is not cut below this for its chunk count
constexpr unsigned kWavesPerSimd = 3;             // what the kernel's register count (136..168) allows
constexpr int kEncStreams = 3;                    // encode launches in flight

// What the launch arithmetic needs to know about the device, read once per device.
struct DeviceShape {
    unsigned cus = 0, lds_per_cu = 0, simds_per_cu = 4;
};
int device_shape(int dev, DeviceShape &d) {
    hipDeviceProp_t pr;
    LZ_HIP(hipGetDeviceProperties(&pr, dev));
    d.cus = (unsigned)pr.multiProcessorCount;
    d.lds_per_cu = (unsigned)pr.maxSharedMemoryPerMultiProcessor;
    if (d.lds_per_cu == 0) d.lds_per_cu = (unsigned)pr.sharedMemPerBlock;
    if (d.cus == 0 || d.lds_per_cu == 0) return lzh::fail(LZMA_AMD_ERR_HIP, "device reports no compute units or no LDS");
    return LZMA_AMD_OK;
}

// Everything one batch in flight owns.
struct Slot {
    lzh::MfResult res;
    lzh::DevBuf opt, lit;
    hipEvent_t e0 = nullptr, e1 = nullptr;  // around the batch's encode launch
    bool busy = false;
    size_t c0 = 0, c1 = 0;                  // the chunks of the batch it holds
};

// The device gives a process four hardware queues: one for the finder and three for
// encode launches, the caller's stream being one of the three (streams beyond that
// would share a queue and run one after the other).
//
// Streams and device memory are kept by the calling thread from one call to the next
// (about 230 bytes per input byte of the largest batch seen, with one-word match pairs):
// allocating and releasing that much costs seconds per call.  The environment variable
// LZMA_AMD_WORKSPACE_GIB bounds what one thread may hold (batches shrink to fit), so that
// several encoder threads, or an encoder and a decoder, can share a device;
// lzma_amd_release_workspace() gives everything back.
struct Workspace {
    int device = -1;
    DeviceShape shape;
    hipStream_t mf = nullptr, enc[kEncStreams] = {};  // enc[0] is the caller's, set per call
    hipEvent_t in_ready = nullptr;
    lzh::MfScratch scratch;
    Slot slots[kEncStreams + 1];
    uint64_t bytes() const {
        uint64_t b = scratch.bytes();
        for (const auto &sl : slots) b += sl.res.meta.bytes + sl.res.arena.bytes + sl.opt.bytes + sl.lit.bytes;
        return b;
    }
    int prepare(int dev) {
        if (device == dev) return LZMA_AMD_OK;
        release();
        device = dev;  // from here on release() undoes whatever was created
        int rc = device_shape(dev, shape);
        if (rc) return rc;
        LZ_HIP(hipStreamCreateWithFlags(&mf, hipStreamNonBlocking));
        for (int i = 1; i < kEncStreams; i++) LZ_HIP(hipStreamCreateWithFlags(&enc[i], hipStreamNonBlocking));
        LZ_HIP(hipEventCreateWithFlags(&in_ready, hipEventDisableTiming));
        for (auto &sl : slots) {
            LZ_HIP(hipEventCreate(&sl.e0));
            LZ_HIP(hipEventCreate(&sl.e1));
        }
        return LZMA_AMD_OK;
    }
    void release() {
        if (device < 0) return;
        (void)hipDeviceSynchronize();
        if (mf) (void)hipStreamDestroy(mf);
        for (int i = 1; i < kEncStreams; i++)
            if (enc[i]) (void)hipStreamDestroy(enc[i]);
        if (in_ready) (void)hipEventDestroy(in_ready);
        mf = nullptr, in_ready = nullptr;
        for (auto &e : enc) e = nullptr;
        for (auto &sl : slots) {
            if (sl.e0) (void)hipEventDestroy(sl.e0);
            if (sl.e1) (void)hipEventDestroy(sl.e1);
            sl.e0 = sl.e1 = nullptr;
            sl.res.release();
            sl.opt.release(), sl.lit.release();
            sl.busy = false;
        }
        scratch.release();
        device = -1;
    }
};
// never destroyed on its own: at process exit the HIP runtime may already be gone
thread_local Workspace *g_ws = nullptr;

template <bool PACKED, uint32_t NW>
int launch_encode(unsigned grid, unsigned lds, hipStream_t es, const uint8_t *data, uint32_t nbytes,
                  const lzh::MfResult &res, const uint64_t *d_in_off, uint64_t base, uint64_t c0, uint32_t nchunks,
                  const lz::EncParams &ep, uint8_t *d_out, const uint64_t *d_out_off, uint64_t *d_out_sizes,
                  int32_t *d_status, lz::OptG *opt, lz::prob_t *lit, uint64_t lit_stride) {
    hipLaunchKernelGGL((encode_kernel<PACKED, NW>), dim3(grid), dim3(64 * NW), lds, es, data, nbytes, res.d_meta(),
                       res.d_arena(), d_in_off, base, c0, nchunks, ep, d_out, d_out_off, d_out_sizes, d_status,
                       opt, lit, lit_stride);
    LZ_HIP(hipGetLastError());
    return LZMA_AMD_OK;
}

}  // namespace

// The call is cut into batches of whole chunks.  The finder works through the batches on
// one stream; the batches' encode launches go to kEncStreams other streams: one of them runs
// beside the finder, the others start together when the finder has done its last batch (see
// "Launch order" below), together as many chunks as the device holds at once.
namespace {
// Host buffers of a call that came through lzma_amd_encode_chunks: the input is copied to the
// device batch by batch on a stream of its own, ahead of the finder (which waits for the batch
// it is about to read, so only the first batch's copy is not hidden behind device work), and
// every batch's output region goes back as soon as its encode launch is done.
struct HostIo {
    const uint8_t *in = nullptr;     // in[0] is byte in_base of the chunk offsets
    uint64_t in_base = 0;
    uint8_t *out = nullptr;          // out[0] is byte out_off[0] of the output offsets
    const uint64_t *out_off = nullptr;
    hipStream_t copy = nullptr;
};
int encode_impl(const uint8_t *d_in, const uint64_t *d_in_off, size_t n_chunks, size_t in_total,
                const lzma_amd_params *p, uint8_t *d_out, const uint64_t *d_out_off, uint64_t *d_out_sizes,
                int32_t *d_status, void *stream, const HostIo *hio);
}  // namespace

extern "C" int lzma_amd_encode_chunks_dev(const uint8_t *d_in, const uint64_t *d_in_off,
                                          size_t n_chunks, size_t in_total, const lzma_amd_params *p,
                                          uint8_t *d_out, const uint64_t *d_out_off,
                                          uint64_t *d_out_sizes, int32_t *d_status, void *stream) {
    return encode_impl(d_in, d_in_off, n_chunks, in_total, p, d_out, d_out_off, d_out_sizes, d_status, stream,
                       nullptr);
}

namespace {
int encode_impl(const uint8_t *d_in, const uint64_t *d_in_off, size_t n_chunks, size_t in_total,
                const lzma_amd_params *p, uint8_t *d_out, const uint64_t *d_out_off, uint64_t *d_out_sizes,
                int32_t *d_status, void *stream, const HostIo *hio) {
    hipStream_t st = (hipStream_t)stream;
    lzh::ThreadState &ts = lzh::tls();
    ts.ms_total = ts.ms_finder = ts.ms_coder = 0.f;
    ts.coder_launches = ts.coop_launches = 0;
    int rc = check_params(p);
    if (rc) return rc;
    if (n_chunks == 0) return LZMA_AMD_OK;
    if (!d_in_off || !d_out || !d_out_off || !d_out_sizes || !d_status)
        return lzh::fail(LZMA_AMD_ERR_ARG, "encode_chunks: null pointer argument");
    std::vector<uint64_t> off(n_chunks + 1);
    LZ_HIP(hipMemcpyAsync(off.data(), d_in_off, (n_chunks + 1) * 8, hipMemcpyDeviceToHost, st));
    LZ_HIP(hipStreamSynchronize(st));
    for (size_t c = 0; c < n_chunks; c++) {
        if (off[c + 1] < off[c]) return lzh::fail(LZMA_AMD_ERR_ARG, "encode_chunks: offsets must be non-decreasing");
        if (off[c + 1] - off[c] > kMaxChunkBytes)
            return lzh::fail(LZMA_AMD_ERR_ARG, "encode_chunks: chunk %zu larger than %llu bytes", c,
                             (unsigned long long)kMaxChunkBytes);
    }
    // in_total is the size of the caller's input buffer: every chunk must lie inside it
    if (off[n_chunks] > (uint64_t)in_total)
        return lzh::fail(LZMA_AMD_ERR_ARG, "encode_chunks: chunk offsets reach %llu, beyond the %zu input bytes",
                         (unsigned long long)off[n_chunks], in_total);
    lz::EncParams ep = {p->fb, p->lc, p->lp, p->pb, 1u << p->dict_log, p->dict_log * 2,
                        (uint32_t)(p->eos != 0)};
    const LdsLayout L = lds_layout(p->pb, p->fb);
    // cooperative workgroups keep their literal models in LDS: coders of up to 0x300 << 4 models
    // (24 KiB); chunks with larger ones are encoded by one wave each
    const bool coop_ok = p->lc + p->lp <= 4;
    const LdsLayout L2 = lds_layout(p->pb, p->fb, 2, 0x300u << (p->lc + p->lp));
    const LdsLayout L3 = lds_layout(p->pb, p->fb, 3, 0x300u << (p->lc + p->lp));
    const uint64_t lit_stride = 0x300ull << (p->lc + p->lp);
    const bool packed = lz::mf_can_pack(ep.dict_size);

    int dev = 0;
    LZ_HIP(hipGetDevice(&dev));
    if (!g_ws) g_ws = new Workspace;
    Workspace &S = *g_ws;
    if ((rc = S.prepare(dev))) {
        S.release();
        return rc;
    }

    // resident workgroups on the whole device: as many per CU as LDS and registers allow
    // (LDS is handed out in 1280-byte units; measured), fewer if the literal models are large.
    // A grid larger than what is really resident would leave workgroups waiting for a whole
    // chunk's time, so the estimate is the minimum of the runtime's and our own.
    const unsigned lds_per_wg = (L.total + 1279u) / 1280u * 1280u;
    unsigned per_cu = std::min(kWavesPerSimd * S.shape.simds_per_cu, S.shape.lds_per_cu / lds_per_wg);
    {
        int api = 0;
        hipError_t e = packed ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, encode_kernel<true, 1>, 64, L.total)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, encode_kernel<false, 1>, 64, L.total);
        if (e == hipSuccess && api > 0) per_cu = std::min(per_cu, (unsigned)api);
    }
    unsigned resident = S.shape.cus * std::max(1u, per_cu);
    const uint64_t lit_budget = 2ull << 30;
    if ((uint64_t)resident * lit_stride * sizeof(lz::prob_t) > lit_budget)
        resident = (unsigned)(lit_budget / (lit_stride * sizeof(lz::prob_t)));

    lzh::release_finder_workspace();  // what a match-finder call of this thread kept is needed here
    int enc_streams = kEncStreams;
    if (const char *e = getenv("LZMA_AMD_ENC_STREAMS")) enc_streams = std::max(1, std::min(atoi(e), (int)kEncStreams));
    // with many chunks a batch holds the resident workgroups' share of one launch, so that
    // the launches in flight fill the device exactly; few or small chunks are batched by size
    // alone (a launch loops over them)
    const size_t batch_chunks = n_chunks > resident ? std::max(1u, resident / enc_streams) : resident;
    // Batch size in bytes: the chunk count above at the average chunk size, within what the
    // device (and the caller's budget) has room for: a batch costs the finder's tables and
    // sort temporaries once, and meta words and pairs for each match-list slot in flight.
    const uint64_t total = off[n_chunks] - off[0];
    const uint64_t avg_chunk = std::max<uint64_t>(1, total / n_chunks);
    uint64_t batch_bytes = std::max<uint64_t>(kMinBatchBytes, std::min<uint64_t>((uint64_t)batch_chunks * avg_chunk, 1ull << 30));
    if (const char *e = getenv("LZMA_AMD_BATCH_MIB")) batch_bytes = (uint64_t)atoi(e) << 20;
    {
        size_t free_b = 0, total_b = 0;
        LZ_HIP(hipMemGetInfo(&free_b, &total_b));
        uint64_t avail = (uint64_t)free_b + S.bytes();  // the kept workspace counts as free: it is reused
        avail -= avail / 16;
        if (const char *e = getenv("LZMA_AMD_WORKSPACE_GIB")) avail = std::min<uint64_t>(avail, (uint64_t)atoll(e) << 30);
        const uint64_t per_pos_finder = lzh::MfScratch::bytes_per_position(ep.fb) + 40;  // tables + sort temporaries
        const uint64_t per_pos_slot = 8 + (packed ? 4 : 8) * 5;
        auto need = [&](uint64_t b) {
            return per_pos_finder * b + (uint64_t)(enc_streams + 1) * (per_pos_slot * b + 600000000ull) + (1ull << 30);
        };
        while (batch_bytes > (64ull << 20) && need(batch_bytes) > avail) batch_bytes -= 64ull << 20;
    }
    std::vector<size_t> cut{0};
    for (size_t c = 0; c < n_chunks;) {
        size_t c1 = c + 1;
        while (c1 < n_chunks && off[c1 + 1] - off[c] <= batch_bytes &&
               (c1 - c < batch_chunks || off[c1] - off[c] < kMinBatchBytes))
            c1++;
        cut.push_back(c1);
        c = c1;
    }
    const size_t nbatches = cut.size() - 1;
    uint64_t max_bytes = 0;
    size_t max_chunks = 0;
    for (size_t b = 0; b < nbatches; b++) {
        max_bytes = std::max<uint64_t>(max_bytes, off[cut[b + 1]] - off[cut[b]]);
        max_chunks = std::max(max_chunks, cut[b + 1] - cut[b]);
    }
    const unsigned in_flight = (unsigned)std::min<size_t>(nbatches, enc_streams);
    // Workgroups per launch: one per chunk up to this many.  For chunks of even size a launch has
    // about resident / in_flight of them anyway; for many uneven ones, workgroups waiting in the
    // dispatcher's queue balance the load better than resident ones looping over a fixed share.
    // Each owns 140 KiB of HBM scratch (optimum table, literal models) per match-list slot.
    unsigned grid_cap = 8192;
    if ((uint64_t)grid_cap * lit_stride * sizeof(lz::prob_t) > lit_budget)
        grid_cap = std::max(1u, (unsigned)(lit_budget / (lit_stride * sizeof(lz::prob_t))));
    const unsigned max_grid = (unsigned)std::min<size_t>(max_chunks, grid_cap);

    // host input: every batch's bytes on the copy stream, an event after each
    std::vector<hipEvent_t> copied;
    struct EventsGuard {
        std::vector<hipEvent_t> &v;
        ~EventsGuard() { for (hipEvent_t e : v) (void)hipEventDestroy(e); }
    } events_guard{copied};
    if (hio && hio->in) {
        copied.resize(nbatches, nullptr);
        for (size_t b = 0; b < nbatches; b++) {
            const uint64_t b0 = off[cut[b]], b1 = off[cut[b + 1]];
            LZ_HIP(hipEventCreateWithFlags(&copied[b], hipEventDisableTiming));
            if (b1 > b0)
                LZ_HIP(hipMemcpyAsync(const_cast<uint8_t *>(d_in) + b0, hio->in + (b0 - hio->in_base), b1 - b0,
                                      hipMemcpyHostToDevice, hio->copy));
            LZ_HIP(hipEventRecord(copied[b], hio->copy));
        }
    }

    S.enc[0] = st;
    // whatever the caller enqueued on its stream (the inputs) comes first
    LZ_HIP(hipEventRecord(S.in_ready, st));
    LZ_HIP(hipStreamWaitEvent(S.mf, S.in_ready, 0));
    for (int i = 1; i < (int)in_flight; i++) LZ_HIP(hipStreamWaitEvent(S.enc[i], S.in_ready, 0));

    // all memory up front: nothing is allocated or freed while launches are in flight
    lzh::MfScratch &scratch = S.scratch;
    if ((rc = scratch.reserve((uint32_t)max_bytes, ep.fb))) return rc;
    const int nslots = (int)std::min<size_t>(nbatches, in_flight + 1);
    Slot *slots = S.slots;
    for (int i = 0; i < nslots; i++) {
        Slot &sl = slots[i];
        sl.busy = false;
        if ((rc = sl.res.reserve((uint32_t)max_bytes, packed ? 4u : 8u))) return rc;
        const size_t opt_bytes = (size_t)max_grid * lz::kNumOpts * sizeof(lz::OptG);
        const size_t lit_bytes = (size_t)max_grid * lit_stride * sizeof(lz::prob_t);
        if (sl.opt.bytes < opt_bytes) LZ_HIP(sl.opt.alloc(opt_bytes));
        if (sl.lit.bytes < lit_bytes) LZ_HIP(sl.lit.alloc(lit_bytes));
    }
    // the sort temporaries go into the arena the scan is about to fill when that is large enough,
    // else into memory of their own
    // (every slot's arena: kept arenas differ in size once one of them has grown)
    size_t min_arena = slots[0].res.arena.bytes;
    for (int i = 1; i < nslots; i++) min_arena = std::min(min_arena, slots[i].res.arena.bytes);
    if (min_arena < lzh::MfScratch::tmp_bytes((uint32_t)max_bytes))
        if ((rc = scratch.reserve_tmp((uint32_t)max_bytes))) return rc;
    lzh::Timer t_all;
    t_all.start(S.mf);
    std::vector<lzh::Timer> t_mf(nbatches);
    float ms_coder = 0.f;
    // an error while launches are in flight: let them finish before the caller gets its buffers
    // back (they write into the caller's outputs and into workspace slots)
    auto bail = [&](int code) -> int {
        (void)hipDeviceSynchronize();
        for (int i = 0; i < nslots; i++) slots[i].busy = false;
        return code;
    };
#define LZ_HIP_BAIL(expr)                                                                           \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return bail(lzh::fail(LZMA_AMD_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                 \
                                  hipGetErrorString(e_), __FILE__, __LINE__));                      \
    } while (0)
    auto retire = [&](Slot &sl) -> int {
        if (!sl.busy) return LZMA_AMD_OK;
        LZ_HIP(hipEventSynchronize(sl.e1));
        float t = 0.f;
        if (hipEventElapsedTime(&t, sl.e0, sl.e1) == hipSuccess) ms_coder += t;
        sl.busy = false;
        if (hio && hio->out) {  // the batch's output region goes home while other launches run
            const uint64_t o0 = hio->out_off[sl.c0], o1 = hio->out_off[sl.c1];
            if (o1 > o0)
                LZ_HIP(hipMemcpyAsync(hio->out + (o0 - hio->out_off[0]), d_out + o0, o1 - o0, hipMemcpyDeviceToHost,
                                      hio->copy));
        }
        return LZMA_AMD_OK;
    };
    // Launch order.  A resident encode wave holds a third of a SIMD's registers, so every launch
    // running beside the finder takes waves away from it: with one launch beside it a finder
    // batch takes 1.4 times as long as alone, with two 2.4 times (measured, 1 MiB chunks).  A
    // chunk's encode time hardly depends on what else runs.  The end of a call is what this
    // costs most: the call ends one chunk time after the finder's last batch.  So once the
    // chunks not yet launched fit the device together (the call's last generation), at most one
    // launch runs beside the finder and the other finished batches wait; they start together when
    // the finder is done, or when every match-list slot is taken.  Before that, every batch is
    // launched as soon as its lists are complete.
    struct Held {
        size_t b;
        uint64_t base;
        uint32_t nbytes, nchunks;
    };
    std::vector<Held> held;
    // (a call whose chunks all fit the device at once ends one chain time after the finder's last
    // batch whatever is launched earlier, so nothing runs beside its finder: 2048 chunks of 1 MiB
    // 9.2 -> 8.5 s, 3072 chunks unchanged)
    int co_run = n_chunks <= resident ? 0 : 1;
    if (const char *e = getenv("LZMA_AMD_CO_RUN")) co_run = std::max(0, atoi(e));
    int coop_mode = -1;  // -1: by the number of chunks, 0: one wave per chunk, 1 or 2: two waves, 3: three
    if (const char *e = getenv("LZMA_AMD_COOP")) coop_mode = atoi(e);
    // workgroups of three and of two waves the device holds at once (the cooperative kernels need
    // more registers than the one-wave kernel: two of their waves fit a SIMD, not three)
    unsigned resident3 = 0, resident2 = 0;
    if (coop_ok) {
        int api = 0;
        hipError_t e = packed ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, encode_kernel<true, 3>, 192, L3.total)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, encode_kernel<false, 3>, 192, L3.total);
        if (e == hipSuccess && api > 0) resident3 = (unsigned)api * S.shape.cus;
        api = 0;
        e = packed ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, encode_kernel<true, 2>, 128, L2.total)
                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, encode_kernel<false, 2>, 128, L2.total);
        if (e == hipSuccess && api > 0) resident2 = (unsigned)api * S.shape.cus;
    }
    // Several waves per chunk when the call has so few chunks that wave slots would stay empty:
    // three if every chunk can have a resident workgroup of three, else two if every chunk can
    // have one of two (1280 and 1536 chunks of 1 MiB, which do not: 7.9 and 8.6 s with two waves,
    // the workgroups beyond the resident ones waiting a chain's time, against 7.8 and 8.3 s with one).  (Not for the last launches of a larger call: they run
    // beside earlier ones, every slot is taken, and two waves hold a chunk's slots for longer
    // in sum than one does -- measured on 4096 chunks: 18.8 s against 15.8 s.)
    unsigned nw_call = 1;
    if (coop_ok && coop_mode != 0) {
        if (coop_mode == 2 || coop_mode == 3)
            nw_call = (unsigned)coop_mode;
        else if (coop_mode == 1)
            nw_call = 2;
        else if (n_chunks <= resident3)
            nw_call = 3;
        else if (n_chunks <= resident2)
            nw_call = 2;
    }
    // the stream each launch goes to: one whose previous launch is done if there is one (a
    // launch queued behind a running one waits a whole chunk time), else in turn
    int stream_slot[kEncStreams];
    for (int &v : stream_slot) v = -1;
    auto pick_stream = [&](size_t b) -> int {
        for (int i = 0; i < (int)in_flight; i++) {
            const int k = (int)((b + i) % in_flight);
            if (stream_slot[k] < 0 || !slots[stream_slot[k]].busy || hipEventQuery(slots[stream_slot[k]].e1) == hipSuccess)
                return k;
        }
        return (int)(b % in_flight);
    };
    auto running = [&]() -> int {
        int n = 0;
        for (int i = 0; i < nslots; i++)
            if (slots[i].busy && hipEventQuery(slots[i].e1) != hipSuccess) n++;
        return n;
    };
    auto launch = [&](const Held &h) -> int {
        Slot &sl = slots[h.b % nslots];
        const int sk = pick_stream(h.b);
        stream_slot[sk] = (int)(h.b % nslots);
        hipStream_t es = S.enc[sk];
        const unsigned grid = std::min(h.nchunks, grid_cap);
        LZ_HIP_BAIL(hipEventRecord(sl.e0, es));
        const unsigned nw = nw_call;
        const bool coop = nw > 1;
        auto fn = packed ? (nw == 3 ? launch_encode<true, 3> : nw == 2 ? launch_encode<true, 2> : launch_encode<true, 1>)
                         : (nw == 3 ? launch_encode<false, 3> : nw == 2 ? launch_encode<false, 2> : launch_encode<false, 1>);
        int r = fn(grid, nw == 3 ? L3.total : nw == 2 ? L2.total : L.total, es, d_in + h.base, h.nbytes, sl.res, d_in_off, h.base,
                   (uint64_t)cut[h.b], h.nchunks, ep, d_out, d_out_off, d_out_sizes, d_status,
                   sl.opt.as<lz::OptG>(), sl.lit.as<lz::prob_t>(), lit_stride);
        if (coop) ts.coop_launches++;
        if (r) return bail(r);
        LZ_HIP_BAIL(hipEventRecord(sl.e1, es));
        sl.busy = true;
        sl.c0 = cut[h.b], sl.c1 = cut[h.b + 1];
        ts.coder_launches++;
        return LZMA_AMD_OK;
    };
    for (size_t b = 0; b < nbatches; b++) {
        Slot &sl = slots[b % nslots];
        // the slot's previous batch must be encoded before its lists go
        for (size_t i = 0; i < held.size(); i++)
            if (held[i].b % nslots == b % nslots) {  // (cannot happen: held batches are launched when the slots run out)
                if ((rc = launch(held[i]))) return rc;
                held.erase(held.begin() + i);
                break;
            }
        if ((rc = retire(sl))) return bail(rc);
        const size_t c0 = cut[b], c1 = cut[b + 1];
        const uint64_t base = off[c0];
        const uint32_t nbytes = (uint32_t)(off[c1] - base);
        const uint32_t nchunks = (uint32_t)(c1 - c0);
        if (!copied.empty()) LZ_HIP_BAIL(hipStreamWaitEvent(S.mf, copied[b], 0));
        t_mf[b].start(S.mf);
        // the sort temporaries live in the arena the scan is about to fill when it is large
        // enough; the first batch's sorts run before any encode launch of this call
        scratch.alone = b == 0;
        if ((rc = scratch.build(d_in + base, nbytes, ep.dict_size, ep.fb, S.mf, sl.res.arena.p,
                                sl.res.arena.bytes)))
            return bail(rc);
        uint64_t total_pairs = 0;
        if ((rc = lzh::mf_scan(scratch, sl.res, d_in_off, base, c0, nchunks, S.mf, &total_pairs)))
            return bail(rc);
        t_mf[b].stop();
        // the finder stream is idle here (mf_scan synchronized it): the lists are complete
        held.push_back({b, base, nbytes, nchunks});
        const bool finder_done = b + 1 == nbatches;
        // the next batch's slot still holds the lists of a batch that waits: the slots have run out
        bool slots_full = false;
        for (const Held &h : held) slots_full |= h.b % nslots == (b + 1) % nslots;
        const bool last_generation = n_chunks - cut[held.front().b] <= resident;
        size_t n_launch = 0;
        // (a call that leaves most of the device empty does not slow the finder down much: its
        // launches go out as their lists are ready)
        if (finder_done || slots_full || !last_generation || nw_call > 1)
            n_launch = held.size();
        else {
            const int r = running();
            n_launch = r < co_run ? std::min<size_t>(held.size(), (size_t)(co_run - r)) : 0;
        }
        for (size_t i = 0; i < n_launch; i++)
            if ((rc = launch(held[i]))) return rc;
        held.erase(held.begin(), held.begin() + n_launch);
    }
    for (int i = 0; i < nslots; i++)
        if ((rc = retire(slots[i]))) return bail(rc);
    t_all.stop();
    LZ_HIP_BAIL(hipStreamSynchronize(S.mf));
    for (int i = 0; i < (int)in_flight; i++) LZ_HIP_BAIL(hipStreamSynchronize(S.enc[i]));
    if (hio) LZ_HIP_BAIL(hipStreamSynchronize(hio->copy));
#undef LZ_HIP_BAIL
    float ms_finder = 0.f;
    for (auto &t : t_mf) ms_finder += t.ms();
    ts.ms_finder = ms_finder;
    ts.ms_coder = ms_coder;
    ts.ms_total = t_all.ms();
    return LZMA_AMD_OK;
}
}  // namespace

extern "C" void lzma_amd_release_workspace(void) {
    if (g_ws) g_ws->release();
    lzh::release_finder_workspace();
    lzh::release_decode_workspace();
    lzh::host_stage().release();
}

extern "C" int lzma_amd_encode_chunks(const uint8_t *in, const uint64_t *in_off, size_t n_chunks,
                                      const lzma_amd_params *p, uint8_t *out, const uint64_t *out_off,
                                      uint64_t *out_sizes, int32_t *status) {
    int rc = check_params(p);
    if (rc) return rc;
    if (n_chunks == 0) return LZMA_AMD_OK;
    if (!in_off || !out || !out_off || !out_sizes || !status)
        return lzh::fail(LZMA_AMD_ERR_ARG, "encode_chunks: null pointer argument");
    for (size_t c = 0; c < n_chunks; c++)
        if (in_off[c + 1] < in_off[c] || out_off[c + 1] < out_off[c])
            return lzh::fail(LZMA_AMD_ERR_ARG, "encode_chunks: offsets must be non-decreasing");
    const uint64_t in_base = in_off[0], in_total = in_off[n_chunks] - in_base;
    const uint64_t out_base = out_off[0], out_total = out_off[n_chunks] - out_base;
    // Device copies of the buffers are kept by the thread between calls; the bytes travel on a
    // stream of their own beside the kernels (HostIo).  Pinned caller buffers (hipHostMalloc /
    // hipHostRegister) make those copies truly asynchronous; pageable ones work, more slowly.
    lzh::HostStage &H = lzh::host_stage();
    if ((rc = H.prepare())) return rc;
    if ((rc = H.reserve(0, in_total)) || (rc = H.reserve(1, (n_chunks + 1) * 8)) || (rc = H.reserve(2, out_total)) ||
        (rc = H.reserve(3, (n_chunks + 1) * 8)) || (rc = H.reserve(4, n_chunks * 8)) || (rc = H.reserve(5, n_chunks * 4)))
        return rc;
    LZ_HIP(hipMemcpyAsync(H.buf[1].p, in_off, (n_chunks + 1) * 8, hipMemcpyHostToDevice, H.stream));
    LZ_HIP(hipMemcpyAsync(H.buf[3].p, out_off, (n_chunks + 1) * 8, hipMemcpyHostToDevice, H.stream));
    LZ_HIP(hipStreamSynchronize(H.stream));
    HostIo hio;
    hio.in = in + in_base, hio.in_base = in_base, hio.out = out + out_base, hio.out_off = out_off, hio.copy = H.stream;
    rc = encode_impl(H.buf[0].as<uint8_t>() - in_base, H.buf[1].as<uint64_t>(), n_chunks, in_base + in_total, p,
                     H.buf[2].as<uint8_t>() - out_base, H.buf[3].as<uint64_t>(), H.buf[4].as<uint64_t>(),
                     H.buf[5].as<int32_t>(), nullptr, &hio);
    if (rc) {
        (void)hipDeviceSynchronize();
        return rc;
    }
    LZ_HIP(hipMemcpyAsync(out_sizes, H.buf[4].p, n_chunks * 8, hipMemcpyDeviceToHost, H.stream));
    LZ_HIP(hipMemcpyAsync(status, H.buf[5].p, n_chunks * 4, hipMemcpyDeviceToHost, H.stream));
    LZ_HIP(hipStreamSynchronize(H.stream));
    return LZMA_AMD_OK;
}
