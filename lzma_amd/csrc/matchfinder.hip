#include "host_util.hpp"
extern "C" int lzma_amd_bt4_matches_dev(const uint8_t *, size_t, uint32_t, uint32_t, uint32_t *,
                                        uint64_t *, uint32_t *, size_t, uint64_t *, void *) {
    return lzh::fail(LZMA_AMD_ERR_UNSUPPORTED, "bt4_matches: not built yet");
}
extern "C" int lzma_amd_bt4_matches(const uint8_t *, size_t, uint32_t, uint32_t, uint32_t *,
                                    uint64_t *, uint32_t *, size_t, size_t *) {
    return lzh::fail(LZMA_AMD_ERR_UNSUPPORTED, "bt4_matches: not built yet");
}
