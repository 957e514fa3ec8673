// Host build of the serial cores of the GPU kernels, for unit tests on machines
// without a GPU.  TEST VEHICLE ONLY: the shipped library never contains or calls
// this; it exists so that the kernel logic can be checked against the oracle
// before it is run on a real MI355X.
#define LZ_HOSTSIM 1
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../lzma_amd/csrc/lzma_dec_core.hpp"

extern "C" int hs_decode(const uint8_t *in, size_t n, uint8_t *out, size_t out_cap, size_t *out_n) {
    lz::Header h;
    int32_t rc = lz::parse_header(in, n, h);
    *out_n = 0;
    if (rc) return rc;
    std::vector<lz::prob_t> probs(lz::kNumBaseProbs + ((size_t)0x300 << (h.lc + h.lp)), lz::kProbInit);
    lz::DecodeResult r = lz::decode_stream(in, n, h, probs.data(), probs.data() + lz::P_LIT, out,
                                           out_cap, 0u, 1u);
    *out_n = r.out_size;
    return r.status;
}

// ---------------------------------------------------------------- encoder core
#include "../../lzma_amd/csrc/lzma_enc_core.hpp"

namespace {
// Brute-force exhaustive BT4 model (hash chains + record breakers), host only.
struct ChainMF {
    const uint8_t *d;
    uint32_t n, dict, fb, mask;
    std::vector<int64_t> head2, head3, head4, prev4;
    std::vector<uint32_t> counts, offs, pairs;
    void build() {
        uint32_t crc[256];
        for (int i = 0; i < 256; i++) crc[i] = lz::crc32_entry(i);
        mask = lz::bt4_hash_mask(dict);
        head2.assign(1024, -1);
        head3.assign(65536, -1);
        head4.assign((size_t)mask + 1, -1);
        prev4.assign(n, -1);
        counts.assign(n, 0);
        offs.assign(n, 0);
        for (uint32_t P = 0; P < n; P++) {
            offs[P] = (uint32_t)pairs.size();
            uint32_t lenLimit = n - P < fb ? n - P : fb;
            if (lenLimit < 4) continue;
            const uint8_t *c = d + P;
            uint32_t t = crc[c[0]] ^ c[1], h2 = t & 1023;
            t ^= (uint32_t)c[2] << 8;
            uint32_t h3 = t & 0xFFFF, h4 = (t ^ (crc[c[3]] << 5)) & mask;
            int64_t W = (int64_t)P - dict;
            if (W < 0) W = 0;
            int64_t q2 = head2[h2], q3 = head3[h3], q4 = head4[h4];
            head2[h2] = head3[h3] = head4[h4] = P;
            prev4[P] = q4;
            size_t s0 = pairs.size();
            uint32_t maxLen = 1;
            int64_t last = q2;
            if (q2 >= W && d[q2] == c[0]) { maxLen = 2; pairs.push_back(2); pairs.push_back(P - q2 - 1); }
            if (q3 >= W && d[q3] == c[0]) {
                if (q3 == q2) pairs.resize(pairs.size() - 2);
                maxLen = 3; pairs.push_back(3); pairs.push_back(P - q3 - 1); last = q3;
            }
            if (pairs.size() != s0 && last == q4) { pairs.resize(pairs.size() - 2); maxLen = 1; }
            for (int64_t q = q4; q >= W; q = prev4[q]) {
                uint32_t l = 0;
                while (l < lenLimit && d[q + l] == c[l]) l++;
                if (l > maxLen) {
                    maxLen = l; pairs.push_back(l); pairs.push_back(P - q - 1);
                    if (l == lenLimit) break;
                }
            }
            counts[P] = (uint32_t)(pairs.size() - s0) / 2;
        }
    }
};
struct ChainMFRef {
    const ChainMF *m;
    uint32_t operator()(uint32_t pos, uint32_t *out) const {
        uint32_t c = m->counts[pos];
        memcpy(out, m->pairs.data() + m->offs[pos], c * 8);
        return c;
    }
};
}  // namespace

template <uint32_t NW>
static int hs_encode_impl(const uint8_t *in, size_t n, uint32_t dict_log, uint32_t fb, uint32_t lc,
                          uint32_t lp, uint32_t pb, int eos, uint8_t *out, size_t out_cap,
                          size_t *out_n) {
    std::vector<uint8_t> padded(in, in + n);
    padded.resize(n + 16, 0);
    ChainMF cm;
    cm.d = padded.data();
    cm.n = (uint32_t)n;
    cm.dict = 1u << dict_log;
    cm.fb = fb;
    cm.build();
    lz::EncParams ep = {fb, lc, lp, pb, 1u << dict_log, dict_log * 2, (uint32_t)(eos != 0)};
    std::vector<lz::prob_t> probs(lz::kNumBaseProbs + ((size_t)0x300 << (lc + lp)), lz::kProbInit);
    std::vector<uint16_t> prices(lz::price_layout(pb, fb).total, 0), pp(512);
    std::vector<lz::OptG> opt_g(lz::kNumOpts);
    std::vector<lz::Rec> opt_rec(lz::kRecWin);
    std::vector<lz::Reps> opt_bks(lz::kBksWin);
    std::vector<uint32_t> small(lz::enc_small_words(fb, pb));
    lz::build_prob_prices(pp.data(), 0, 1);
    const size_t own_words = lz::la_words(fb) + lz::mdist_words(fb);
    std::vector<uint32_t> sync(lz::SY_WORDS, 0), own(2 * own_words);
    lz::Encoder<ChainMFRef, lz::prob_t *, NW> e;
    e.sync = sync.data();
    e.data = padded.data();
    e.n = (uint32_t)n;
    e.readable = (uint32_t)n;  // exercises the byte-wise tail of match_len
    e.probs = probs.data();
    e.lit = probs.data() + lz::P_LIT;
    e.prices = prices.data();
    e.pp = pp.data();
    e.o_rec = opt_rec.data();
    e.o_bks = opt_bks.data();
    e.opt_g = opt_g.data();
    e.out = out + lz::kHeaderSize;
    e.out_cap = out_cap - lz::kHeaderSize;
    e.mf = ChainMFRef{&cm};
    e.ep = ep;
    e.lane_id = 0;
    lz::write_header(out, ep, eos ? -1 : (int64_t)n);
    if (NW > 1) {
        // the other waves of a cooperative workgroup are threads over the same memory
        lz::Encoder<ChainMFRef, lz::prob_t *, NW> h1 = e, h2 = e;
        std::thread t1([&] { h1.helper_run(small.data(), own.data(), 1); });
        std::thread t2;
        if (NW > 2) t2 = std::thread([&] { h2.helper_run(small.data(), own.data() + own_words, 2); });
        e.run(small.data());
        t1.join();
        if (NW > 2) t2.join();
    } else {
        e.run(small.data());
    }
    *out_n = lz::kHeaderSize + e.out_n;
    return e.dead ? lz::ST_HIP : (e.overflow ? lz::ST_OUT_CAP : lz::ST_OK);
}

extern "C" int hs_encode(const uint8_t *in, size_t n, uint32_t dict_log, uint32_t fb, uint32_t lc,
                         uint32_t lp, uint32_t pb, int eos, uint8_t *out, size_t out_cap,
                         size_t *out_n) {
    return hs_encode_impl<1>(in, n, dict_log, fb, lc, lp, pb, eos, out, out_cap, out_n);
}
// the same through the cooperative parse loop, one thread per wave (two waves, three waves)
extern "C" int hs_encode_coop(const uint8_t *in, size_t n, uint32_t dict_log, uint32_t fb, uint32_t lc,
                              uint32_t lp, uint32_t pb, int eos, uint8_t *out, size_t out_cap,
                              size_t *out_n) {
    return hs_encode_impl<2>(in, n, dict_log, fb, lc, lp, pb, eos, out, out_cap, out_n);
}
extern "C" int hs_encode_coop3(const uint8_t *in, size_t n, uint32_t dict_log, uint32_t fb, uint32_t lc,
                               uint32_t lp, uint32_t pb, int eos, uint8_t *out, size_t out_cap,
                               size_t *out_n) {
    return hs_encode_impl<3>(in, n, dict_log, fb, lc, lp, pb, eos, out, out_cap, out_n);
}
