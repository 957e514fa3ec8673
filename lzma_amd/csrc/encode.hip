#include "host_util.hpp"
extern "C" int lzma_amd_encode_chunks_dev(const uint8_t *, const uint64_t *, size_t, size_t,
                                          const lzma_amd_params *, uint8_t *, const uint64_t *,
                                          uint64_t *, int32_t *, void *) {
    return lzh::fail(LZMA_AMD_ERR_UNSUPPORTED, "encode_chunks: not built yet");
}
extern "C" int lzma_amd_encode_chunks(const uint8_t *, const uint64_t *, size_t,
                                      const lzma_amd_params *, uint8_t *, const uint64_t *,
                                      uint64_t *, int32_t *) {
    return lzh::fail(LZMA_AMD_ERR_UNSUPPORTED, "encode_chunks: not built yet");
}
