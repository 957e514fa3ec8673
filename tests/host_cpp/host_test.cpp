// Exercises the C++ mirror of the reference API (include/lzma_amd.hpp) the way the
// reference's own tests use the Go API (lzma_encoder_test.go:30-91, lzma_decoder_test.go:14-32).
//
//   host_test args                      argument errors; needs no GPU
//   host_test roundtrip data.txt data.eos.l3.lzma    needs a GPU
#include <cstdio>
#include <fstream>
#include <sstream>

#include "lzma_amd.hpp"

#define CHECK(c)                                                             \
    do {                                                                     \
        if (!(c)) {                                                          \
            std::fprintf(stderr, "FAILED %s (%s:%d)\n", #c, __FILE__, __LINE__); \
            return 1;                                                        \
        }                                                                    \
    } while (0)

static std::string slurp(const char *path) {
    std::ifstream f(path, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

static int test_args() {
    // level out of range: reported by Write and by Close, not by the constructor (TestBoth's bad levels)
    for (int level : {0, 10, -3}) {
        std::ostringstream sink;
        auto w = lzma::NewWriterLevel(sink, level);
        bool thrown = false;
        try {
            w->Write("x", 1);
        } catch (const lzma::ArgumentValueError &e) {
            thrown = e.msg == "level out of range" && e.val == level;
        }
        CHECK(thrown);
        thrown = false;
        try {
            w->Close();
        } catch (const lzma::ArgumentValueError &) {
            thrown = true;
        }
        CHECK(thrown);
        CHECK(sink.str().empty());
    }
    {
        std::ostringstream sink;
        auto w = lzma::NewWriterSizeLevel(sink, -2, 3);
        bool thrown = false;
        try {
            w->Write("x", 1);
        } catch (const lzma::ArgumentValueError &e) {
            thrown = e.msg == "illegal size" && e.val == -2;
        }
        CHECK(thrown);
    }
    lzma_amd_params p;
    CHECK(lzma_amd_params_from_level(5, 1, &p) == LZMA_AMD_OK && p.dict_log == 23 && p.fb == 128 && p.eos == 1);
    CHECK(lzma_amd_params_from_level(0, 1, &p) == LZMA_AMD_ERR_ARG);
    CHECK(lzma::DefaultCompression == 5 && lzma::BestSpeed == 1 && lzma::BestCompression == 9);
    std::puts("args ok");
    return 0;
}

static int test_roundtrip(const char *txt_path, const char *lzma_path) {
    const std::string txt = slurp(txt_path), golden = slurp(lzma_path);
    CHECK(!txt.empty() && !golden.empty());
    // decode the reference's own stream (TestDecoder / BenchmarkDecoder input)
    {
        std::istringstream in(golden);
        auto r = lzma::NewReader(in);
        CHECK(r->ReadAll() == txt);
    }
    // encode at level 3 with end marker: decodes back, size within 1 % of the reference's file
    std::string enc;
    {
        std::ostringstream out;
        auto w = lzma::NewWriterLevel(out, 3);
        w->Write(txt.data(), txt.size() / 2);
        w->Write(txt.data() + txt.size() / 2, txt.size() - txt.size() / 2);
        w->Close();
        enc = out.str();
    }
    CHECK(enc.compare(0, 13, golden, 0, 13) == 0);  // same header: lc3 lp0 pb2, dict 1 MiB, size -1
    CHECK(enc.size() * 100 <= golden.size() * 101 && golden.size() * 100 <= enc.size() * 101);
    {
        std::istringstream in(enc);
        auto r = lzma::NewReader(in);
        std::string back;
        char buf[4096];
        for (size_t k; (k = r->Read(buf, sizeof buf)) != 0;) back.append(buf, k);
        CHECK(back == txt);
    }
    // size-known header, empty input, and a header error
    {
        std::ostringstream out;
        auto w = lzma::NewWriterSize(out, 0);
        w->Close();
        const std::string s = out.str();
        CHECK(s.size() >= 13 && s[5] == 0 && s[12] == 0);
        std::istringstream in(s);
        CHECK(lzma::NewReader(in)->ReadAll().empty());
    }
    {
        std::string bad = golden;
        bad[0] = (char)0xFF;  // properties byte >= 9*5*5
        std::istringstream in(bad);
        bool thrown = false;
        try {
            lzma::NewReader(in)->ReadAll();
        } catch (const lzma::HeaderError &) {
            thrown = true;
        }
        CHECK(thrown);
    }
    // batch form: ragged chunks
    {
        lzma_amd_params p;
        CHECK(lzma_amd_params_from_level(1, 0, &p) == LZMA_AMD_OK);
        std::vector<std::string> chunks = {txt.substr(0, 70000), "", txt.substr(70000, 1), txt.substr(100000, 50000)};
        auto streams = lzma::EncodeChunks(chunks, p);
        auto dec = lzma::DecodeStreams(streams);
        for (size_t i = 0; i < chunks.size(); i++) CHECK(dec[i].status == LZMA_AMD_OK && dec[i].data == chunks[i]);
    }
    std::puts("roundtrip ok");
    return 0;
}

int main(int argc, char **argv) {
    try {
        if (argc >= 2 && std::string(argv[1]) == "args") return test_args();
        if (argc >= 4 && std::string(argv[1]) == "roundtrip") return test_roundtrip(argv[2], argv[3]);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 2;
    }
    std::fprintf(stderr, "usage: host_test args | roundtrip data.txt data.eos.l3.lzma\n");
    return 64;
}
