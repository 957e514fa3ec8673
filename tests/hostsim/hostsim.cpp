// Host build of the serial cores of the GPU kernels, for unit tests on machines
// without a GPU.  TEST VEHICLE ONLY: the shipped library never contains or calls
// this; it exists so that the kernel logic can be checked against the oracle
// before it is run on a real MI355X.
#define LZ_HOSTSIM 1
#include <cstdlib>
#include <cstring>
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
