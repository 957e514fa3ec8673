// Drives the host build of the encoder core (hostsim.cpp: one, two and three "waves" per chunk, the
// cooperative ones as real threads) over a few dozen drawn inputs and parameter sets; built with
// AddressSanitizer and UndefinedBehaviorSanitizer by `make sanitize_check`.  The three widths must
// agree byte for byte, the decoder core must invert them, and the sanitizers must stay silent.
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
extern "C" int hs_encode(const uint8_t *, size_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, int, uint8_t *, size_t, size_t *);
extern "C" int hs_encode_coop3(const uint8_t *, size_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, int, uint8_t *, size_t, size_t *);
extern "C" int hs_decode(const uint8_t *in, size_t n, uint8_t *out, size_t out_cap, size_t *out_n);
extern "C" int hs_encode_coop(const uint8_t *, size_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, int, uint8_t *, size_t, size_t *);
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb");
    std::vector<uint8_t> text(300000);
    text.resize(fread(text.data(), 1, text.size(), f));
    std::mt19937 rng(7);
    int bad = 0;
    for (int it = 0; it < 24; it++) {
        size_t n = it % 6 == 0 ? rng() % 40 : 1000 + rng() % 40000;
        std::vector<uint8_t> in(n);
        int kind = it % 4;
        for (size_t i = 0; i < n; i++)
            in[i] = kind == 0 ? text[(it * 9973 + i) % text.size()] : kind == 1 ? (uint8_t)(rng() & 255) : kind == 2 ? (uint8_t)("abcab"[i % 5] + (rng() % 97 == 0)) : 0;
        uint32_t dict_log = 12 + rng() % 10, fbs[] = {5, 8, 32, 64, 128, 273}, fb = fbs[rng() % 6];
        uint32_t lc = rng() % 5, lp = rng() % 3, pb = rng() % 5; int eos = rng() & 1;
        std::vector<uint8_t> a(n * 2 + 4096), b(a.size()), c(a.size());
        size_t na = 0, nb = 0, nc = 0;
        hs_encode(in.data(), n, dict_log, fb, lc, lp, pb, eos, a.data(), a.size(), &na);
        hs_encode_coop(in.data(), n, dict_log, fb, lc, lp, pb, eos, b.data(), b.size(), &nb);
        hs_encode_coop3(in.data(), n, dict_log, fb, lc, lp, pb, eos, c.data(), c.size(), &nc);
        bool same = na == nb && na == nc && !memcmp(a.data(), b.data(), na) && !memcmp(a.data(), c.data(), na);
        // ... and the decoder core takes the stream back to the input, and a cut copy to a prefix of it
        std::vector<uint8_t> back(n + 512);
        size_t nd = 0;
        const int rc = hs_decode(a.data(), na, back.data(), n + 273, &nd);
        same = same && rc == 0 && nd == n && !memcmp(back.data(), in.data(), n);
        std::vector<uint8_t> cut(a.begin(), a.begin() + na / 2);
        (void)hs_decode(cut.data(), cut.size(), back.data(), n + 273, &nd);
        same = same && nd <= n && !memcmp(back.data(), in.data(), nd);
        if (!same) bad++;
        printf("%d n=%zu kind=%d dict=%u fb=%u -> %zu %s\n", it, n, kind, dict_log, fb, na, same ? "same" : "DIFFERENT");
    }
    return bad != 0;
}
