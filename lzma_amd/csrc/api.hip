// Miscellaneous C-ABI entry points: errors, device selection, parameters, header.
#include <cstring>

#include "host_util.hpp"
#include "lzma_common.hpp"

namespace lzh {

ThreadState &tls() {
    static thread_local ThreadState s = {{0}, 0.f, 0.f, 0.f, 0};
    return s;
}

HostStage &host_stage() {
    // never destroyed on its own: at process exit the HIP runtime may already be gone
    static thread_local HostStage *h = nullptr;
    if (!h) h = new HostStage;
    return *h;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tls().err, sizeof tls().err, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace lzh

extern "C" {

const char *lzma_amd_last_error(void) { return lzh::tls().err; }
const char *lzma_amd_version(void) { return "lzma_amd 0.1 (gfx950)"; }

int lzma_amd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lzma_amd_set_device(int device) {
    LZ_HIP(hipSetDevice(device));
    // the one thing lzma_amd_decode_batch_dev_noalloc must not do later, inside a graph capture
    return lzh::noalloc_prepare();
}

// levels table, lzma_encoder.go:97-108
int lzma_amd_params_from_level(int level, int eos, lzma_amd_params *p) {
    static const uint32_t dl[10] = {0, 16, 18, 20, 22, 23, 24, 25, 26, 27};
    static const uint32_t fbs[10] = {0, 64, 64, 64, 128, 128, 128, 256, 256, 256};
    if (!p || level < 1 || level > 9) return lzh::fail(LZMA_AMD_ERR_ARG, "level out of range: %d", level);
    p->dict_log = dl[level];
    p->fb = fbs[level];
    p->lc = 3;
    p->lp = 0;
    p->pb = 2;
    p->cut_value = 0;
    p->eos = eos ? 1 : 0;
    return LZMA_AMD_OK;
}

size_t lzma_amd_encode_bound(size_t n) { return n + n / 8 + 128; }

int lzma_amd_parse_header(const uint8_t *in, size_t n, uint32_t *lc, uint32_t *lp, uint32_t *pb,
                          uint32_t *dict_size, int64_t *unpack_size) {
    lz::Header h;
    int32_t rc = lz::parse_header(in, n, h);
    if (rc) return rc;
    if (lc) *lc = h.lc;
    if (lp) *lp = h.lp;
    if (pb) *pb = h.pb;
    if (dict_size) *dict_size = h.dict_size;
    if (unpack_size) *unpack_size = h.unpack_size;
    return LZMA_AMD_OK;
}

void lzma_amd_last_timing(float *ms_total, float *ms_finder, float *ms_coder) {
    if (ms_total) *ms_total = lzh::tls().ms_total;
    if (ms_finder) *ms_finder = lzh::tls().ms_finder;
    if (ms_coder) *ms_coder = lzh::tls().ms_coder;
}

int lzma_amd_last_coder_launches(void) { return lzh::tls().coder_launches; }
int lzma_amd_last_coop_launches(void) { return lzh::tls().coop_launches; }

}  // extern "C"
