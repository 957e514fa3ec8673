// Per-chunk LZMA encoder for gfx950: every chunk becomes an independent .lzma
// stream, one 64-lane wavefront (= one workgroup) per chunk.
//
// Replaces encoder.encoder() of the reference (lzma_encoder.go:989-1107) run once
// per chunk.  The optimal parse and the range coder (lzma_enc_core.hpp) run with
// wave-uniform values; the probability models, the 16-bit price tables and the
// current match list live in LDS (about 26 KiB per chunk at lc3 lp0 pb2, six
// chunks resident per CU); the optimum table sits in HBM.  Match lists come from
// the exhaustive BT4 finder (lzma_mf_core.hpp), evaluated by the same wave with all
// 64 lanes whenever the parse asks for a position's matches.
//
// Roofline class: HBM in principle (N bytes in, C bytes out per chunk);
// serial-dependency-bound in practice.
#include <vector>

#include "lzma_enc_core.hpp"
#include "mf_tables.hpp"

namespace {

constexpr uint32_t kLitLdsLog = 3;
constexpr uint32_t kMdistWords = 2 * (lz::kMatchMax + 1) + 2;

// Match list of a chunk position, read from the finder's results.
struct MfFunctor {
    const uint64_t *meta;
    const uint32_t *arena;
    uint32_t job_start, lane;
    __device__ uint32_t operator()(uint32_t pos, uint32_t *out) const {
        const uint64_t m = meta[job_start + pos];
        const uint32_t c = lz::mf_meta_count(m);
        const uint32_t *src = arena + 2 * lz::mf_meta_off(m);
        for (uint32_t k = lane; k < 2 * c; k += 64) out[k] = src[k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        return c;
    }
};

struct LdsLayout {
    uint32_t probs, prices, pp, mdist, small, total;  // byte offsets
};
__host__ __device__ inline LdsLayout lds_layout(uint32_t lc, uint32_t lp, uint32_t pb, bool lit_lds) {
    LdsLayout l;
    uint32_t nprobs = lz::kNumBaseProbs + (lit_lds ? (0x300u << (lc + lp)) : 0);
    l.probs = 0;
    l.prices = (l.probs + nprobs * 2 + 15) & ~15u;
    l.pp = (l.prices + lz::price_layout(pb).total * 2 + 15) & ~15u;
    l.mdist = l.pp + 512 * 2;
    l.small = l.mdist + kMdistWords * 4;
    l.total = l.small + lz::kEncSmallWords * 4;
    return l;
}

// Chunks [c0, c0 + nchunks) of the call; their bytes start at data offset
// in_off[c] - base inside t.data.
template <bool LIT_LDS>
__global__ __launch_bounds__(64) void encode_kernel(const uint8_t *__restrict__ data,
                                                    const uint64_t *__restrict__ mf_meta,
                                                    const uint32_t *__restrict__ mf_arena,
                                                    const uint64_t *__restrict__ in_off,
                                                    uint64_t base, uint64_t c0, uint32_t nchunks,
                                                    lz::EncParams ep, uint8_t *__restrict__ out,
                                                    const uint64_t *__restrict__ out_off,
                                                    uint64_t *__restrict__ out_sizes,
                                                    int32_t *__restrict__ status, lz::Opt *opt_scratch,
                                                    lz::prob_t *lit_scratch, uint64_t lit_stride) {
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t lane = threadIdx.x;
    const LdsLayout L = lds_layout(ep.lc, ep.lp, ep.pb, LIT_LDS);
    lz::prob_t *probs = reinterpret_cast<lz::prob_t *>(smem + L.probs);
    uint16_t *prices = reinterpret_cast<uint16_t *>(smem + L.prices);
    uint16_t *pp = reinterpret_cast<uint16_t *>(smem + L.pp);
    uint32_t *mdist = reinterpret_cast<uint32_t *>(smem + L.mdist);
    uint32_t *small = reinterpret_cast<uint32_t *>(smem + L.small);
    const uint32_t lit_n = 0x300u << (ep.lc + ep.lp);
    lz::build_prob_prices(pp, lane, 64u);

    for (uint32_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const uint64_t chunk = c0 + c;
        const uint32_t jb = (uint32_t)(in_off[chunk] - base), je = (uint32_t)(in_off[chunk + 1] - base);
        const uint64_t ob = out_off[chunk], oe = out_off[chunk + 1];
        if (oe - ob < lz::kHeaderSize) {
            if (lane == 0) {
                status[chunk] = lz::ST_OUT_CAP;
                out_sizes[chunk] = 0;
            }
            continue;
        }
        __syncthreads();
        for (uint32_t i = lane; i < lz::kNumBaseProbs; i += 64) probs[i] = lz::kProbInit;
        lz::prob_t *lit;
        if (LIT_LDS) {
            lit = probs + lz::P_LIT;
        } else {
            lit = lit_scratch + (uint64_t)blockIdx.x * lit_stride;
        }
        for (uint32_t i = lane; i < lit_n; i += 64) lit[i] = lz::kProbInit;
        __syncthreads();

        lz::Encoder<MfFunctor, lz::prob_t, lz::prob_t, uint16_t> e;
        e.data = data + jb;
        e.n = je - jb;
        e.probs = probs;
        e.lit = lit;
        e.prices = prices;
        e.pp = pp;
        e.opt = opt_scratch + (uint64_t)blockIdx.x * lz::kNumOpts;
        e.mdist = mdist;
        e.out = out + ob + lz::kHeaderSize;
        e.out_cap = oe - ob - lz::kHeaderSize;
        e.mf = MfFunctor{mf_meta, mf_arena, jb, lane};
        e.ep = ep;
        e.run(small);
        if (lane == 0) {
            lz::write_header(out + ob, ep, ep.eos ? -1 : (int64_t)e.n);
            status[chunk] = e.overflow ? lz::ST_OUT_CAP : lz::ST_OK;
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

constexpr uint64_t kMaxBatchBytes = 1ull << 30;  // positions per match-finder build

}  // namespace

extern "C" int lzma_amd_encode_chunks_dev(const uint8_t *d_in, const uint64_t *d_in_off,
                                          size_t n_chunks, size_t in_total, const lzma_amd_params *p,
                                          uint8_t *d_out, const uint64_t *d_out_off,
                                          uint64_t *d_out_sizes, int32_t *d_status, void *stream) {
    (void)in_total;
    hipStream_t st = (hipStream_t)stream;
    lzh::tls().ms_total = lzh::tls().ms_finder = lzh::tls().ms_coder = 0.f;
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
        if (off[c + 1] - off[c] > kMaxBatchBytes)
            return lzh::fail(LZMA_AMD_ERR_ARG, "encode_chunks: chunk %zu larger than %llu bytes", c,
                             (unsigned long long)kMaxBatchBytes);
    }
    lz::EncParams ep = {p->fb, p->lc, p->lp, p->pb, 1u << p->dict_log, p->dict_log * 2,
                        (uint32_t)(p->eos != 0)};
    const bool lit_lds = p->lc + p->lp <= kLitLdsLog;
    const LdsLayout L = lds_layout(p->lc, p->lp, p->pb, lit_lds);

    float ms_finder = 0.f, ms_coder = 0.f;
    size_t c0 = 0;
    while (c0 < n_chunks) {
        size_t c1 = c0 + 1;
        while (c1 < n_chunks && off[c1 + 1] - off[c0] <= kMaxBatchBytes) c1++;
        const uint64_t base = off[c0];
        const uint32_t nbytes = (uint32_t)(off[c1] - base);
        const uint32_t nchunks = (uint32_t)(c1 - c0);
        lzh::Timer t_mf, t_enc;
        t_mf.start(st);
        lzh::MfWorkspace ws;
        rc = ws.build(d_in + base, nbytes, ep.dict_size, ep.fb, st);
        if (rc) return rc;
        uint64_t total_pairs = 0;
        rc = ws.scan(d_in_off, base, c0, nchunks, st, &total_pairs);
        if (rc) return rc;
        t_mf.stop();
        unsigned grid = nchunks;
        const unsigned max_grid = lit_lds ? 256u * 16u : 512u;
        if (grid > max_grid) grid = max_grid;
        lzh::DevBuf opt, lit;
        LZ_HIP(opt.alloc((size_t)grid * lz::kNumOpts * sizeof(lz::Opt)));
        const uint64_t lit_stride = 0x300ull << (p->lc + p->lp);
        if (!lit_lds) LZ_HIP(lit.alloc((size_t)grid * lit_stride * sizeof(lz::prob_t)));
        t_enc.start(st);
        if (lit_lds)
            hipLaunchKernelGGL(encode_kernel<true>, dim3(grid), dim3(64), L.total, st, d_in + base, ws.d_meta(),
                               ws.d_arena(), d_in_off,
                               base, (uint64_t)c0, nchunks, ep, d_out, d_out_off, d_out_sizes, d_status,
                               opt.as<lz::Opt>(), (lz::prob_t *)nullptr, (uint64_t)0);
        else
            hipLaunchKernelGGL(encode_kernel<false>, dim3(grid), dim3(64), L.total, st, d_in + base, ws.d_meta(),
                               ws.d_arena(), d_in_off,
                               base, (uint64_t)c0, nchunks, ep, d_out, d_out_off, d_out_sizes, d_status,
                               opt.as<lz::Opt>(), lit.as<lz::prob_t>(), lit_stride);
        t_enc.stop();
        LZ_HIP(hipGetLastError());
        LZ_HIP(hipStreamSynchronize(st));
        ms_finder += t_mf.ms();
        ms_coder += t_enc.ms();
        c0 = c1;
    }
    lzh::tls().ms_finder = ms_finder;
    lzh::tls().ms_coder = ms_coder;
    lzh::tls().ms_total = ms_finder + ms_coder;
    return LZMA_AMD_OK;
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
    lzh::DevBuf d_in, d_io, d_out, d_oo, d_sz, d_st;
    LZ_HIP(d_in.alloc(in_total));
    LZ_HIP(d_io.alloc((n_chunks + 1) * 8));
    LZ_HIP(d_out.alloc(out_total));
    LZ_HIP(d_oo.alloc((n_chunks + 1) * 8));
    LZ_HIP(d_sz.alloc(n_chunks * 8));
    LZ_HIP(d_st.alloc(n_chunks * 4));
    if (in_total) LZ_HIP(hipMemcpy(d_in.p, in + in_base, in_total, hipMemcpyHostToDevice));
    LZ_HIP(hipMemcpy(d_io.p, in_off, (n_chunks + 1) * 8, hipMemcpyHostToDevice));
    LZ_HIP(hipMemcpy(d_oo.p, out_off, (n_chunks + 1) * 8, hipMemcpyHostToDevice));
    rc = lzma_amd_encode_chunks_dev(d_in.as<uint8_t>() - in_base, d_io.as<uint64_t>(), n_chunks,
                                    in_total, p, d_out.as<uint8_t>() - out_base, d_oo.as<uint64_t>(),
                                    d_sz.as<uint64_t>(), d_st.as<int32_t>(), nullptr);
    if (rc) return rc;
    if (out_total) LZ_HIP(hipMemcpy(out + out_base, d_out.p, out_total, hipMemcpyDeviceToHost));
    LZ_HIP(hipMemcpy(out_sizes, d_sz.p, n_chunks * 8, hipMemcpyDeviceToHost));
    LZ_HIP(hipMemcpy(status, d_st.p, n_chunks * 4, hipMemcpyDeviceToHost));
    return LZMA_AMD_OK;
}
