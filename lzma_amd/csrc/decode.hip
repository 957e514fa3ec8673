// Batched LZMA decoder for gfx950: many independent .lzma streams decoded
// concurrently, one 64-lane wavefront (= one workgroup) per stream.
//
// Replaces decoder.decoder()/doDecode of the reference (lzma_decoder.go:218-349)
// applied once per stream.  A stream is a serial chain, and the throughput of the device
// grows almost linearly with the number of streams a CU holds (measured from 2 to 10 per
// CU), so the state kept in LDS is cut to the 3.6 KiB of non-literal models and a CU holds
// 32 streams, as many as it has wave slots (the kernel stays within 64 vector and 96 scalar
// registers for that).  The literal models (12 KiB at lc 3) live in a per-workgroup HBM
// area behind the L2 and are walked in registers, one fetch per literal
// (lzma_dec_core.hpp).  At 32 waves per CU the kernel is bound by instruction issue, so the
// bit step is divided between the scalar and the vector unit by hand (lzma_dec_core.hpp);
// the match copy is done by all 64 lanes.  Each workgroup owns one literal area and loops
// over the streams blockIdx, blockIdx + gridDim, ...  Streams whose literal coder is larger
// than the area (lc+lp > 4) are deferred to a second launch with few workgroups and
// full-size areas.
//
// Roofline class: HBM-bound in principle (reads C compressed bytes, writes N raw
// bytes per stream), instruction-issue-bound in practice.
#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "host_util.hpp"
#include "lzma_dec_core.hpp"

namespace {

constexpr uint32_t kLitSmallLog = 4;                     // lc+lp handled by the first launch
constexpr uint64_t kLitSmall = 0x300ull << kLitSmallLog;  // models per workgroup area, first launch
constexpr uint64_t kLitBig = 0x300ull << 12;              // ... second launch (lc <= 8, lp <= 4)
// Workgroups of the first launch: one per stream up to this many (3 GiB of literal areas).  Far
// more than are resident (32 per CU): streams differ in length, and workgroups that wait in the
// dispatcher's queue balance the load better than resident ones looping over a fixed share.
constexpr uint64_t kMaxGrid = 131072;
constexpr int32_t kPending = -1;

// BIG == false: all streams; those with lc+lp > kLitSmallLog are
// appended to big_list.  BIG == true: persistent grid over big_list.
typedef __attribute__((address_space(3))) lz::prob_t lds_prob_t;

template <bool BIG>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(96))) void decode_kernel(
    const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off, uint64_t n_streams,
    uint8_t *__restrict__ out, const uint64_t *__restrict__ out_off,
    uint64_t *__restrict__ out_sizes, int32_t *__restrict__ status, uint32_t *big_count,
    uint32_t *big_list, lz::prob_t *lit_scratch, uint64_t lit_scratch_stride, uint32_t lit_small_log,
    uint32_t *ticket) {
    __shared__ __align__(16) lz::prob_t s_probs[lz::kNumBaseProbs];
    const uint32_t lane = threadIdx.x;
    lz::prob_t *lit = lit_scratch + (uint64_t)blockIdx.x * lit_scratch_stride;

    // ticket != nullptr: the workgroups (as many as are resident) draw the streams from a
    // counter, so that streams of very different length still keep every workgroup busy while
    // the literal areas in use are those of the resident workgroups only; else workgroup b
    // takes streams b, b + gridDim, ...
    uint64_t first = blockIdx.x, step = gridDim.x;
    uint64_t count = BIG ? (uint64_t)*big_count : n_streams;
    for (uint64_t it = first;; it += step) {
        if (ticket) {
            uint32_t t = 0;
            if (lane == 0) t = atomicAdd(ticket, 1u);
            it = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        }
        if (it >= count) break;
        const uint64_t s = BIG ? (uint64_t)big_list[it] : it;
        const uint64_t ib = in_off[s], ie = in_off[s + 1];
        const uint64_t ob = out_off[s], oe = out_off[s + 1];
        const uint8_t *sin = in + ib;
        const uint64_t n = ie - ib;
        lz::Header h;
        int32_t hs = lz::parse_header(sin, n, h);
        if (hs != lz::ST_OK) {
            if (lane == 0) {
                status[s] = hs;
                out_sizes[s] = 0;
            }
            continue;
        }
        const uint32_t lit_n = 0x300u << (h.lc + h.lp);
        if (!BIG && h.lc + h.lp > lit_small_log) {
            if (lane == 0) {
                uint32_t k = atomicAdd(big_count, 1u);
                big_list[k] = (uint32_t)s;
                status[s] = kPending;
            }
            continue;
        }
        __syncthreads();  // previous stream's models are no longer in use
        for (uint32_t i = lane; i < lz::kNumBaseProbs; i += 64) s_probs[i] = lz::kProbInit;
        {   // two models per store
            uint32_t *lw = (uint32_t *)lit;
            for (uint32_t i = lane; i < lit_n / 2; i += 64) lw[i] = lz::kProbInit * 0x10001u;
        }
        __syncthreads();
        // The model array as an LDS pointer held in a vector register the compiler cannot see
        // through: model addresses are then formed on the vector unit, where LDS wants them,
        // instead of on the scalar unit and copied over.
        lds_prob_t *probs = (lds_prob_t *)s_probs;
        asm volatile("" : "+v"(probs));
        const lz::DecodeResult r = lz::decode_stream<true>(sin, n, h, probs, lit, out + ob, oe - ob, lane, 64u);
        if (lane == 0) {
            status[s] = r.status;
            out_sizes[s] = r.out_size;
        }
    }
}

// What a decode call needs beyond the caller's buffers, kept by the calling thread between
// calls (grow-only; lzma_amd_release_workspace frees it): allocating up to 3 GiB of literal
// areas per call costs more than a small batch takes.
struct DecodeWs {
    lzh::DevBuf big, lit_small, lit_big;
    int device = -1;
    unsigned resident = 0;  // workgroups of the first launch the device holds at once
    void release() {
        big.release(), lit_small.release(), lit_big.release();
        device = -1;
        resident = 0;
    }
};
thread_local DecodeWs *g_dws = nullptr;
}  // namespace

void lzh::release_decode_workspace() {
    if (g_dws) g_dws->release();
}

extern "C" int lzma_amd_decode_batch_dev(const uint8_t *d_in, const uint64_t *d_in_off,
                                         size_t n_streams, uint8_t *d_out,
                                         const uint64_t *d_out_off, uint64_t *d_out_sizes,
                                         int32_t *d_status, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    lzh::tls().ms_total = lzh::tls().ms_finder = lzh::tls().ms_coder = 0.f;
    lzh::tls().coder_launches = 0;
    if (n_streams == 0) return LZMA_AMD_OK;
    if (!d_in || !d_in_off || !d_out || !d_out_off || !d_out_sizes || !d_status)
        return lzh::fail(LZMA_AMD_ERR_ARG, "decode_batch: null pointer argument");
    if (n_streams > 0x7FFFFFFFull)
        return lzh::fail(LZMA_AMD_ERR_ARG, "decode_batch: too many streams in one call");
    int dev = 0;
    LZ_HIP(hipGetDevice(&dev));
    if (!g_dws) g_dws = new DecodeWs;
    if (g_dws->device != dev) {
        g_dws->release();
        g_dws->device = dev;
    }
    lzh::DevBuf &big = g_dws->big;  // [0] = count, [1..] = list of deferred stream indices
    // [0] = count of deferred streams, [1] = the counter the streams are drawn from, [2..] = list
    if (big.bytes < (n_streams + 2) * sizeof(uint32_t)) LZ_HIP(big.alloc((n_streams + 2) * sizeof(uint32_t)));
    LZ_HIP(hipMemsetAsync(big.p, 0, 2 * sizeof(uint32_t), st));
    // as many workgroups as the device holds at once: each owns one literal area, reused stream
    // after stream, so the models in use stay inside the caches (the areas of a grid with one
    // workgroup per stream were written back to memory long after their streams had ended)
    if (g_dws->resident == 0) {
        hipDeviceProp_t pr;
        LZ_HIP(hipGetDeviceProperties(&pr, dev));
        int per_cu = 0;
        LZ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_kernel<false>, 64, 0));
        g_dws->resident = (unsigned)std::max(1, per_cu) * (unsigned)std::max(1, pr.multiProcessorCount);
    }
    unsigned want = g_dws->resident;
    if (const char *e = getenv("LZMA_AMD_DECODE_GRID")) want = (unsigned)std::max(1, atoi(e));
    const unsigned grid = (unsigned)std::min<uint64_t>(n_streams, std::min<uint64_t>(want, kMaxGrid));
    lzh::DevBuf &lit_small = g_dws->lit_small;
    if (lit_small.bytes < (size_t)grid * kLitSmall * sizeof(lz::prob_t))
        LZ_HIP(lit_small.alloc((size_t)grid * kLitSmall * sizeof(lz::prob_t)));
    lzh::Timer tm;
    tm.start(st);
    hipLaunchKernelGGL(decode_kernel<false>, dim3(grid), dim3(64), 0, st, d_in, d_in_off,
                       (uint64_t)n_streams, d_out, d_out_off, d_out_sizes, d_status, big.as<uint32_t>(),
                       big.as<uint32_t>() + 2, lit_small.as<lz::prob_t>(), kLitSmall, kLitSmallLog,
                       big.as<uint32_t>() + 1);
    LZ_HIP(hipGetLastError());
    uint32_t n_big = 0;
    LZ_HIP(hipMemcpyAsync(&n_big, big.p, sizeof n_big, hipMemcpyDeviceToHost, st));
    LZ_HIP(hipStreamSynchronize(st));
    lzh::DevBuf &scratch = g_dws->lit_big;
    if (n_big > 0) {
        const unsigned wgs = n_big < 64 ? n_big : 64;
        if (scratch.bytes < (size_t)wgs * kLitBig * sizeof(lz::prob_t))
            LZ_HIP(scratch.alloc((size_t)wgs * kLitBig * sizeof(lz::prob_t)));
        hipLaunchKernelGGL(decode_kernel<true>, dim3(wgs), dim3(64), 0, st, d_in, d_in_off,
                           (uint64_t)n_streams, d_out, d_out_off, d_out_sizes, d_status,
                           big.as<uint32_t>(), big.as<uint32_t>() + 2, scratch.as<lz::prob_t>(), kLitBig, kLitSmallLog,
                           (uint32_t *)nullptr);
        LZ_HIP(hipGetLastError());
    }
    tm.stop();
    LZ_HIP(hipStreamSynchronize(st));
    lzh::tls().ms_total = lzh::tls().ms_coder = tm.ms();
    lzh::tls().coder_launches = n_big > 0 ? 2 : 1;
    return LZMA_AMD_OK;
}

// ---------------------------------------------------------------- no allocation
// The same decode for callers that cannot allocate (code under HIP graph capture, tight
// serving loops): everything the kernels need beyond the caller's buffers -- the literal
// models of each workgroup and the list of streams deferred to the second launch -- lives
// in a file-scope device array, so the call makes no hipMalloc/hipFree.  The array is
// smaller than what lzma_amd_decode_batch_dev allocates, so fewer workgroups run at once.
// The call only enqueues work on the caller's stream (a clearing memset and two kernels)
// and never waits for the device, so it may be recorded into a graph; the address of the
// array is looked up once per device, by lzma_amd_set_device or by the first call.
namespace {
constexpr uint32_t kNaLitLog = 3;                                // lc+lp handled by the first launch
constexpr uint64_t kNaLitModels = 0x300ull << kNaLitLog;         // models per workgroup area
constexpr uint32_t kNaGrid = 4096;                               // workgroups of the first launch
constexpr uint64_t kNaMaxStreams = 1ull << 20;                   // streams per call
constexpr uint64_t kNaSmallBytes = kNaGrid * kNaLitModels * sizeof(lz::prob_t);  // 48 MiB
constexpr uint64_t kNaBigBytes = kLitBig * sizeof(lz::prob_t);   // one full-size area, 6 MiB
// the second launch runs when the first is done and its areas are dead: they serve again as
// full-size areas, next to the one that follows them
constexpr unsigned kNaBigGrid = (unsigned)(kNaSmallBytes / kNaBigBytes) + 1;  // 9 workgroups
constexpr uint64_t kNaListBytes = (kNaMaxStreams + 64) * sizeof(uint32_t);
constexpr uint64_t kNaScratchBytes = kNaSmallBytes + kNaBigBytes + kNaListBytes;
static __device__ __align__(256) uint8_t scratch[kNaScratchBytes];

constexpr int kNaMaxDevices = 64;
uint8_t *g_na_ptr[kNaMaxDevices] = {};  // the array's address on each device, looked up once
std::mutex g_na_mutex;

// The array's device address on the current device (each device has its own copy at its own
// address); the name `scratch` itself is only a host-side handle.
int na_scratch(uint8_t **out) {
    int dev = 0;
    LZ_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kNaMaxDevices) return lzh::fail(LZMA_AMD_ERR_UNSUPPORTED, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_na_mutex);
    if (!g_na_ptr[dev]) {
        void *p = nullptr;
        LZ_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        g_na_ptr[dev] = (uint8_t *)p;
    }
    *out = g_na_ptr[dev];
    return LZMA_AMD_OK;
}
}  // namespace

// Looks the array up on the current device (a module load: not allowed inside a capture).
int lzh::noalloc_prepare() {
    uint8_t *p = nullptr;
    return na_scratch(&p);
}

extern "C" int lzma_amd_decode_batch_dev_noalloc(const uint8_t *d_in, const uint64_t *d_in_off,
                                                 size_t n_streams, uint8_t *d_out,
                                                 const uint64_t *d_out_off, uint64_t *d_out_sizes,
                                                 int32_t *d_status, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    lzh::tls().ms_total = lzh::tls().ms_finder = lzh::tls().ms_coder = 0.f;  // nothing is timed here
    lzh::tls().coder_launches = 0;
    if (n_streams == 0) return LZMA_AMD_OK;
    if (!d_in || !d_in_off || !d_out || !d_out_off || !d_out_sizes || !d_status)
        return lzh::fail(LZMA_AMD_ERR_ARG, "decode_batch: null pointer argument");
    if (n_streams > kNaMaxStreams)
        return lzh::fail(LZMA_AMD_ERR_ARG, "decode_batch_dev_noalloc: %zu streams, at most %llu per call", n_streams,
                         (unsigned long long)kNaMaxStreams);
    uint8_t *base = nullptr;
    int rc = na_scratch(&base);
    if (rc) return rc;
    lz::prob_t *lit_small = (lz::prob_t *)base;
    lz::prob_t *lit_big = (lz::prob_t *)(base + kNaSmallBytes);
    uint32_t *big = (uint32_t *)(base + kNaSmallBytes + kNaBigBytes);  // [0] = count, [64..] = deferred streams
    // the array keeps the previous call's contents: the count must start at zero (the literal
    // areas are initialised by each workgroup, the list is only read up to the count)
    LZ_HIP(hipMemsetAsync(big, 0, 256, st));
    const unsigned grid = (unsigned)std::min<uint64_t>(n_streams, kNaGrid);
    hipLaunchKernelGGL(decode_kernel<false>, dim3(grid), dim3(64), 0, st, d_in, d_in_off,
                       (uint64_t)n_streams, d_out, d_out_off, d_out_sizes, d_status, big, big + 64, lit_small,
                       kNaLitModels, kNaLitLog, (uint32_t *)nullptr);
    LZ_HIP(hipGetLastError());
    // the second launch reads the count on the device: nine workgroups with full-size areas (the
    // first launch's areas, free again, and the one behind them) share the deferred streams
    // (none: they return at once)
    (void)lit_big;  // = lit_small + 8 full-size areas
    hipLaunchKernelGGL(decode_kernel<true>, dim3(kNaBigGrid), dim3(64), 0, st, d_in, d_in_off, (uint64_t)n_streams,
                       d_out, d_out_off, d_out_sizes, d_status, big, big + 64, lit_small, kLitBig, kNaLitLog,
                       (uint32_t *)nullptr);
    LZ_HIP(hipGetLastError());
    lzh::tls().coder_launches = 2;
    return LZMA_AMD_OK;  // results are there when the stream has run: the caller waits for it
}

extern "C" int lzma_amd_decode_batch(const uint8_t *in, const uint64_t *in_off, size_t n_streams,
                                     uint8_t *out, const uint64_t *out_off, uint64_t *out_sizes,
                                     int32_t *status) {
    if (n_streams == 0) return LZMA_AMD_OK;
    if (!in_off || !out_off || !out_sizes || !status)
        return lzh::fail(LZMA_AMD_ERR_ARG, "decode_batch: null pointer argument");
    const uint64_t in_base = in_off[0], in_total = in_off[n_streams] - in_base;
    const uint64_t out_base = out_off[0], out_total = out_off[n_streams] - out_base;
    for (size_t s = 0; s < n_streams; s++)
        if (in_off[s + 1] < in_off[s] || out_off[s + 1] < out_off[s])
            return lzh::fail(LZMA_AMD_ERR_ARG, "decode_batch: offsets must be non-decreasing");
    // device copies of the buffers are kept by the thread between calls; everything is enqueued
    // on one stream and waited for once
    lzh::HostStage &H = lzh::host_stage();
    int rc = H.prepare();
    if (rc) return rc;
    if ((rc = H.reserve(0, in_total)) || (rc = H.reserve(1, (n_streams + 1) * 8)) || (rc = H.reserve(2, out_total)) ||
        (rc = H.reserve(3, (n_streams + 1) * 8)) || (rc = H.reserve(4, n_streams * 8)) || (rc = H.reserve(5, n_streams * 4)))
        return rc;
    if (in_total) LZ_HIP(hipMemcpyAsync(H.buf[0].p, in + in_base, in_total, hipMemcpyHostToDevice, H.stream));
    LZ_HIP(hipMemcpyAsync(H.buf[1].p, in_off, (n_streams + 1) * 8, hipMemcpyHostToDevice, H.stream));
    LZ_HIP(hipMemcpyAsync(H.buf[3].p, out_off, (n_streams + 1) * 8, hipMemcpyHostToDevice, H.stream));
    // device copies index from the batch's own base
    rc = lzma_amd_decode_batch_dev(H.buf[0].as<uint8_t>() - in_base, H.buf[1].as<uint64_t>(), n_streams,
                                   H.buf[2].as<uint8_t>() - out_base, H.buf[3].as<uint64_t>(),
                                   H.buf[4].as<uint64_t>(), H.buf[5].as<int32_t>(), H.stream);
    if (rc) return rc;
    if (out_total) LZ_HIP(hipMemcpyAsync(out + out_base, H.buf[2].p, out_total, hipMemcpyDeviceToHost, H.stream));
    LZ_HIP(hipMemcpyAsync(out_sizes, H.buf[4].p, n_streams * 8, hipMemcpyDeviceToHost, H.stream));
    LZ_HIP(hipMemcpyAsync(status, H.buf[5].p, n_streams * 4, hipMemcpyDeviceToHost, H.stream));
    LZ_HIP(hipStreamSynchronize(H.stream));
    return LZMA_AMD_OK;
}
