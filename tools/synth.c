/*
 * Deterministic synthetic inputs for the benchmark and the parity tests.
 * Integer arithmetic only (splitmix64), so every host produces the same bytes.
 *
 *  - Markov text: order-2 byte Markov chain whose transition counts are taken
 *    from a sample text (the reference's data/data.txt).  Generated in
 *    independent 1 MiB blocks so it can be produced in parallel and so that
 *    "block b" is the same bytes whether generated alone or as part of a larger
 *    buffer.
 *  - Redundant log: dpkg-style lines over a small vocabulary (long-match stress).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define GOLDEN 0x9E3779B97F4A7C15ull
#define BLOCK (1u << 20)

static inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
typedef struct { uint64_t s; } rng;
static inline uint64_t rng_next(rng *r) { r->s += GOLDEN; return mix64(r->s); }
static inline uint32_t rng_below(rng *r, uint32_t n) { /* [0,n) */
    return (uint32_t)(((rng_next(r) >> 32) * (uint64_t)n) >> 32);
}

typedef struct {
    uint32_t start[65536 + 1]; /* successors of context c are sym/cum[start[c]..start[c+1]) */
    uint32_t total[65536];
    uint8_t *sym;
    uint32_t *cum; /* inclusive cumulative counts */
    uint32_t restart_ctx;
} markov_model;

markov_model *synth_markov_build(const uint8_t *sample, size_t n) {
    if (n < 3) return NULL;
    markov_model *m = (markov_model *)calloc(1, sizeof *m);
    uint32_t *cnt = (uint32_t *)calloc((size_t)65536 * 256, sizeof(uint32_t));
    if (!m || !cnt) return NULL;
    for (size_t i = 0; i + 2 < n; i++)
        cnt[(((size_t)sample[i] << 8) | sample[i + 1]) * 256 + sample[i + 2]]++;
    size_t nz = 0;
    for (size_t i = 0; i < (size_t)65536 * 256; i++) nz += cnt[i] != 0;
    m->sym = (uint8_t *)malloc(nz);
    m->cum = (uint32_t *)malloc(nz * sizeof(uint32_t));
    uint32_t w = 0;
    for (uint32_t c = 0; c < 65536; c++) {
        m->start[c] = w;
        uint32_t run = 0;
        for (uint32_t s = 0; s < 256; s++) {
            uint32_t k = cnt[(size_t)c * 256 + s];
            if (k) { run += k; m->sym[w] = (uint8_t)s; m->cum[w] = run; w++; }
        }
        m->total[c] = run;
    }
    m->start[65536] = w;
    m->restart_ctx = ((uint32_t)sample[0] << 8) | sample[1];
    free(cnt);
    return m;
}
void synth_markov_free(markov_model *m) {
    if (m) { free(m->sym); free(m->cum); free(m); }
}

static void markov_block(const markov_model *m, uint64_t seed, uint64_t block, uint8_t *out, size_t n) {
    rng r = { mix64(seed ^ (block * GOLDEN + 0x632BE59BD9B4E019ull)) };
    uint32_t ctx = m->restart_ctx;
    for (size_t i = 0; i < n; i++) {
        uint32_t tot = m->total[ctx];
        if (tot == 0) { ctx = m->restart_ctx; tot = m->total[ctx]; }
        uint32_t x = rng_below(&r, tot);
        uint32_t lo = m->start[ctx], hi = m->start[ctx + 1] - 1;
        while (lo < hi) { /* first entry with cum > x */
            uint32_t mid = (lo + hi) >> 1;
            if (m->cum[mid] > x) hi = mid; else lo = mid + 1;
        }
        uint8_t s = m->sym[lo];
        out[i] = s;
        ctx = ((ctx << 8) | s) & 0xFFFF;
    }
}

typedef struct { const markov_model *m; uint64_t seed, first_block; uint8_t *out; size_t n; int tid, nthreads; } mjob;
static void *markov_worker(void *a) {
    mjob *j = (mjob *)a;
    size_t nblocks = (j->n + BLOCK - 1) / BLOCK;
    for (size_t b = (size_t)j->tid; b < nblocks; b += (size_t)j->nthreads) {
        size_t off = b * BLOCK, len = j->n - off < BLOCK ? j->n - off : BLOCK;
        markov_block(j->m, j->seed, j->first_block + b, j->out + off, len);
    }
    return NULL;
}
/* Fills out[0..n) with blocks first_block, first_block+1, ... (1 MiB each). */
void synth_markov_fill(const markov_model *m, uint64_t seed, uint64_t first_block, uint8_t *out,
                       size_t n, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    pthread_t th[64];
    mjob jobs[64];
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (mjob){m, seed, first_block, out, n, t, nthreads};
        pthread_create(&th[t], NULL, markov_worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
}

/* ---- redundant dpkg-style log ---- */
static const char *const VERBS[] = {"status unpacked", "status half-configured", "status installed",
                                    "status half-installed", "status config-files",
                                    "status not-installed", "configure", "install", "upgrade",
                                    "remove", "trigproc", "status triggers-pending"};
static const char *const STEMS[] = {"lib", "python-", "gnome-", "kde", "xserver-xorg-", "linux-",
                                    "perl-", "ruby", "gstreamer0.10-", "ttf-", "openoffice.org-",
                                    "mono-", "texlive-", "erlang-", "ghc6-", "ocaml-"};
static const char *const TAILS[] = {"common", "dev", "data", "doc", "utils", "core", "base", "bin",
                                    "runtime", "plugins", "tools", "dbg", "extra", "gtk", "qt4", "cli"};

static void log_block(uint64_t seed, uint64_t block, uint8_t *out, size_t n) {
    rng r = { mix64(seed ^ (block * GOLDEN + 0x1B873593CC9E2D51ull)) };
    /* 64-entry vocabulary, fixed per seed (not per block) */
    rng vr = { mix64(seed ^ 0x5851F42D4C957F2Dull) };
    char pkg[64][48], ver[64][24];
    for (int i = 0; i < 64; i++) {
        snprintf(pkg[i], sizeof pkg[i], "%s%s", STEMS[rng_below(&vr, 16)], TAILS[rng_below(&vr, 16)]);
        snprintf(ver[i], sizeof ver[i], "%u.%u.%u-%uubuntu%u", rng_below(&vr, 4), rng_below(&vr, 30),
                 rng_below(&vr, 10), 1 + rng_below(&vr, 9), 1 + rng_below(&vr, 5));
    }
    uint32_t t = (uint32_t)(block * 40000u) % 86400u, day = 1 + (uint32_t)((block / 2) % 28);
    size_t w = 0;
    char line[160];
    while (w < n) {
        t += rng_below(&r, 3);
        if (t >= 86400) { t -= 86400; day = day % 28 + 1; }
        uint32_t p = rng_below(&r, 64), v = rng_below(&r, 12);
        /* bursts: the same package usually goes through several states in a row */
        int reps = 1 + (int)rng_below(&r, 4);
        for (int k = 0; k < reps && w < n; k++) {
            int len;
            if (v >= 6 && v <= 9)
                len = snprintf(line, sizeof line, "2010-05-%02u %02u:%02u:%02u %s %s %s %s\n", day,
                               t / 3600, t / 60 % 60, t % 60, VERBS[v], pkg[p], ver[p], ver[p]);
            else
                len = snprintf(line, sizeof line, "2010-05-%02u %02u:%02u:%02u %s %s %s\n", day,
                               t / 3600, t / 60 % 60, t % 60, VERBS[v], pkg[p], ver[p]);
            size_t c = (size_t)len < n - w ? (size_t)len : n - w;
            memcpy(out + w, line, c);
            w += c;
            v = (v + 1) % 6;
        }
    }
}
typedef struct { uint64_t seed, first_block; uint8_t *out; size_t n, block; int tid, nthreads; } ljob;
static void *log_worker(void *a) {
    ljob *j = (ljob *)a;
    size_t nblocks = (j->n + j->block - 1) / j->block;
    for (size_t b = (size_t)j->tid; b < nblocks; b += (size_t)j->nthreads) {
        size_t off = b * j->block, len = j->n - off < j->block ? j->n - off : j->block;
        log_block(j->seed, j->first_block + b, j->out + off, len);
    }
    return NULL;
}
/* Fills out[0..n) with independent log blocks of `block` bytes each. */
void synth_log_fill(uint64_t seed, uint64_t first_block, size_t block, uint8_t *out, size_t n,
                    int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    pthread_t th[64];
    ljob jobs[64];
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (ljob){seed, first_block, out, n, block, t, nthreads};
        pthread_create(&th[t], NULL, log_worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
}

/* Stream sizes for the small-stream batch: log-uniform in [lo, hi]. */
void synth_stream_sizes(uint64_t seed, uint32_t lo, uint32_t hi, uint32_t *sizes, size_t n) {
    rng r = { mix64(seed ^ 0x2545F4914F6CDD1Dull) };
    /* log-uniform via 16.16 fixed-point exponent: size = lo * 2^(x * log2(hi/lo)) with a
     * 64-entry 2^(k/64) table in 1.31 fixed point computed by repeated integer sqrt-free
     * multiplication (integer only, deterministic). */
    uint64_t pow2[65];
    pow2[0] = 1ull << 31;
    /* 2^(1/64) in 1.31 fixed point */
    const uint64_t step = 2170814770ull; /* round(2^(1/64) * 2^31) */
    for (int k = 1; k <= 64; k++) pow2[k] = (pow2[k - 1] * step) >> 31;
    uint32_t octaves = 0;
    while (((uint64_t)lo << (octaves + 1)) <= hi) octaves++;
    for (size_t i = 0; i < n; i++) {
        uint32_t x = rng_below(&r, octaves * 64 + 1);
        uint64_t s = ((uint64_t)lo << (x / 64)) * pow2[x % 64] >> 31;
        if (s > hi) s = hi;
        if (s < lo) s = lo;
        sizes[i] = (uint32_t)s;
    }
}
