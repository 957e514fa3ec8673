/*
 * CPU oracle for the LZMA hot path.  TEST INFRASTRUCTURE ONLY -- see
 * lzma_oracle.h.  Every function names the reference file:line it restates
 * (paths are relative to /root/reference).
 *
 * The whole input is held in one flat array instead of the reference's sliding
 * block buffer (lz_window.go:82-163).  That is behaviour-preserving: the
 * reference never looks further ahead than it has buffered, and it always
 * keeps fb+274 bytes buffered ahead unless the stream has ended, so the only
 * observable quantity -- how many bytes remain -- is the same.
 */
#include "lzma_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- format constants (lzma_decoder.go:57-89, range_coder.go:12-17) ---- */
#define HEADER_SIZE 13
#define N_REPS 4
#define N_STATES 12
#define POS_SLOT_BITS 6
#define N_LEN_TO_POS 4
#define MATCH_MIN 2
#define ALIGN_BITS 4
#define ALIGN_SIZE 16
#define ALIGN_MASK 15
#define START_POS_MODEL 4
#define END_POS_MODEL 14
#define N_FULL_DIST 128
#define POS_STATES_BITS_MAX 4
#define POS_STATES_MAX 16
#define LEN_LOW 8
#define LEN_MID 8
#define LEN_HIGH 256
#define N_LEN_SYMS (LEN_LOW + LEN_MID + LEN_HIGH)
#define MATCH_MAX (MATCH_MIN + N_LEN_SYMS - 1) /* 273 */
#define TOP_VALUE (1u << 24)
#define PROB_BITS 11
#define PROB_TOTAL (1u << PROB_BITS)
#define MOVE_BITS 5
#define MOVE_REDUCING_BITS 2
#define PRICE_SHIFT 6
#define INF_PRICE 0x0FFFFFFFu
#define N_OPTS 4096
#define HASH2_SIZE (1u << 10)
#define HASH3_SIZE (1u << 16)
#define HASH3_OFF HASH2_SIZE
#define FIX_HASH (HASH2_SIZE + HASH3_SIZE)
#define MAX_VAL_NORMALIZE ((1u << 30) - 1)

typedef uint16_t prob_t;

/* ---- constant tables, built once ---- */
static uint32_t g_crc[256];        /* lz_bin_tree.go:340-355 */
static uint32_t g_prob_price[512]; /* range_coder.go:222-234 */
static uint8_t g_fast_pos[2048];   /* lzma_encoder.go:131-145 */
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void build_tables(void) {
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t r = i;
        for (int j = 0; j < 8; j++) r = (r & 1) ? (r >> 1) ^ 0xEDB88320u : r >> 1;
        g_crc[i] = r;
    }
    const uint32_t nbits = PROB_BITS - MOVE_REDUCING_BITS; /* 9 */
    for (int32_t i = (int32_t)nbits - 1; i >= 0; i--) {
        uint32_t start = 1u << (nbits - i - 1), end = 1u << (nbits - i);
        for (uint32_t j = start; j < end; j++)
            g_prob_price[j] = ((uint32_t)i << PRICE_SHIFT) +
                              (((end - j) << PRICE_SHIFT) >> (nbits - i - 1));
    }
    int c = 2;
    g_fast_pos[0] = 0;
    g_fast_pos[1] = 1;
    for (int slot = 2; slot < 22; slot++) {
        int k = 1 << ((slot >> 1) - 1);
        for (int j = 0; j < k; j++, c++) g_fast_pos[c] = (uint8_t)slot;
    }
}

/* range_coder.go:236-246 */
static inline uint32_t price_bit(prob_t p, uint32_t bit) {
    return g_prob_price[((((uint32_t)p - bit) ^ (0u - bit)) & (PROB_TOTAL - 1)) >>
                        MOVE_REDUCING_BITS];
}
static inline uint32_t price0(prob_t p) { return g_prob_price[p >> MOVE_REDUCING_BITS]; }
static inline uint32_t price1(prob_t p) {
    return g_prob_price[(PROB_TOTAL - p) >> MOVE_REDUCING_BITS];
}

/* lzma_encoder.go:147-165 */
static inline uint32_t pos_slot(uint32_t pos) {
    if (pos < (1u << 11)) return g_fast_pos[pos];
    if (pos < (1u << 21)) return (uint32_t)g_fast_pos[pos >> 10] + 20;
    return (uint32_t)g_fast_pos[pos >> 20] + 40;
}
static inline uint32_t pos_slot2(uint32_t pos) {
    if (pos < (1u << 17)) return (uint32_t)g_fast_pos[pos >> 6] + 12;
    if (pos < (1u << 27)) return (uint32_t)g_fast_pos[pos >> 16] + 32;
    return (uint32_t)g_fast_pos[pos >> 26] + 52;
}

/* lzma_decoder.go:114-158 */
static inline uint32_t st_char(uint32_t s) { return s < 4 ? 0 : (s < 10 ? s - 3 : s - 6); }
static inline uint32_t st_match(uint32_t s) { return s < 7 ? 7 : 10; }
static inline uint32_t st_rep(uint32_t s) { return s < 7 ? 8 : 11; }
static inline uint32_t st_shortrep(uint32_t s) { return s < 7 ? 9 : 11; }
static inline int st_is_char(uint32_t s) { return s < 7; }
static inline uint32_t len_to_pos_state(uint32_t len) {
    len -= MATCH_MIN;
    return len < N_LEN_TO_POS ? len : N_LEN_TO_POS - 1;
}

static void init_probs(prob_t *p, size_t n) { /* range_coder.go:107-114 */
    for (size_t i = 0; i < n; i++) p[i] = PROB_TOTAL >> 1;
}

/* ======================================================================
 * Match finder: lzBinTree (lz_bin_tree.go) over a flat input window.
 * Positions are 1-based as in the reference (value 0 == empty link).
 * ==================================================================== */
typedef struct {
    const uint8_t *data;
    uint64_t n;
    int64_t base;          /* absolute index of byte at pos p is base + p   */
    uint32_t pos;          /* iw.pos   (1-based, re-based by normalize)      */
    uint32_t *son, *hash;
    uint32_t cyc_pos, cyc_size, match_max, cut, hash_mask, hash_total;
} bt4;

static void bt4_free(bt4 *m) {
    free(m->son);
    free(m->hash);
    m->son = m->hash = NULL;
}

/* lz_bin_tree.go:35-83 (numHashBytes == 4 only; no level selects bt2) */
static int bt4_init(bt4 *m, const uint8_t *data, uint64_t n, uint32_t dict, uint32_t fb,
                    uint32_t cut) {
    memset(m, 0, sizeof *m);
    m->data = data;
    m->n = n;
    m->base = -1;
    m->pos = 1;
    m->cyc_size = dict + 1;
    m->match_max = fb;
    m->cut = cut ? cut : 16 + (fb >> 1);
    uint32_t hs = dict - 1;
    hs |= hs >> 1;
    hs |= hs >> 2;
    hs |= hs >> 4;
    hs |= hs >> 8;
    hs >>= 1;
    hs |= 0xFFFF;
    if (hs > (1u << 24)) hs >>= 1;
    m->hash_mask = hs;
    m->hash_total = hs + 1 + FIX_HASH;
    m->son = (uint32_t *)calloc((size_t)m->cyc_size * 2, sizeof(uint32_t));
    m->hash = (uint32_t *)calloc(m->hash_total, sizeof(uint32_t));
    if (!m->son || !m->hash) {
        bt4_free(m);
        return LZO_ERR_NOMEM;
    }
    return LZO_OK;
}

static inline uint64_t bt4_avail(const bt4 *m) { /* streamPos - pos */
    return m->n - (uint64_t)(m->base + (int64_t)m->pos);
}

/* lz_bin_tree.go:85-113 */
static void bt4_normalize(bt4 *m) {
    uint32_t sub = m->pos - m->cyc_size;
    for (uint64_t i = 0; i < (uint64_t)m->cyc_size * 2; i++)
        m->son[i] = m->son[i] <= sub ? 0 : m->son[i] - sub;
    for (uint32_t i = 0; i < m->hash_total; i++)
        m->hash[i] = m->hash[i] <= sub ? 0 : m->hash[i] - sub;
    m->base += sub;
    m->pos -= sub;
}
static void bt4_move_pos(bt4 *m) {
    if (++m->cyc_pos >= m->cyc_size) m->cyc_pos = 0;
    m->pos++;
    if (m->pos == MAX_VAL_NORMALIZE) bt4_normalize(m);
}

/* getMatches (lz_bin_tree.go:115-253) when dist != NULL, skip of one position
 * (lz_bin_tree.go:255-337) when dist == NULL.  Returns 2 * number of pairs. */
static uint32_t bt4_step(bt4 *m, uint32_t *dist) {
    uint64_t avail = bt4_avail(m);
    uint32_t len_limit;
    if (avail >= m->match_max) {
        len_limit = m->match_max;
    } else {
        len_limit = (uint32_t)avail;
        if (len_limit < 4) {
            bt4_move_pos(m);
            return 0;
        }
    }
    uint32_t off = 0;
    uint32_t min_pos = m->pos > m->cyc_size ? m->pos - m->cyc_size : 0;
    const uint8_t *buf = m->data + m->base; /* buf[p] = byte at 1-based pos p */
    const uint8_t *cur = buf + m->pos;
    uint32_t max_len = 1;

    uint32_t t = g_crc[cur[0]] ^ cur[1];
    uint32_t h2 = t & (HASH2_SIZE - 1);
    t ^= (uint32_t)cur[2] << 8;
    uint32_t h3 = t & (HASH3_SIZE - 1);
    uint32_t h4 = (t ^ (g_crc[cur[3]] << 5)) & m->hash_mask;

    uint32_t cm = m->hash[FIX_HASH + h4];
    uint32_t cm2 = m->hash[h2];
    uint32_t cm3 = m->hash[HASH3_OFF + h3];
    m->hash[h2] = m->pos;
    m->hash[HASH3_OFF + h3] = m->pos;
    if (dist) {
        if (cm2 > min_pos && buf[cm2] == cur[0]) {
            max_len = 2;
            dist[off++] = max_len;
            dist[off++] = m->pos - cm2 - 1;
        }
        if (cm3 > min_pos && buf[cm3] == cur[0]) {
            if (cm3 == cm2) off -= 2;
            max_len = 3;
            dist[off++] = max_len;
            dist[off++] = m->pos - cm3 - 1;
            cm2 = cm3;
        }
        if (off != 0 && cm2 == cm) {
            off -= 2;
            max_len = 1;
        }
    }
    m->hash[FIX_HASH + h4] = m->pos;

    uint32_t ptr0 = (m->cyc_pos << 1) + 1, ptr1 = m->cyc_pos << 1;
    uint32_t len0 = 0, len1 = 0, count = m->cut;
    for (;;) {
        if (cm <= min_pos || count == 0) {
            m->son[ptr1] = 0;
            m->son[ptr0] = 0;
            break;
        }
        count--;
        uint32_t delta = m->pos - cm;
        uint32_t cp = (delta <= m->cyc_pos ? m->cyc_pos - delta
                                           : m->cyc_pos - delta + m->cyc_size)
                      << 1;
        const uint8_t *pb = buf + cm;
        uint32_t len = len0 < len1 ? len0 : len1;
        if (pb[len] == cur[len]) {
            for (len++; len != len_limit; len++)
                if (pb[len] != cur[len]) break;
            if (dist) {
                if (max_len < len) {
                    max_len = len;
                    dist[off++] = len;
                    dist[off++] = delta - 1;
                    if (len == len_limit) {
                        m->son[ptr1] = m->son[cp];
                        m->son[ptr0] = m->son[cp + 1];
                        break;
                    }
                }
            } else if (len == len_limit) {
                m->son[ptr1] = m->son[cp];
                m->son[ptr0] = m->son[cp + 1];
                break;
            }
        }
        if (pb[len] < cur[len]) {
            m->son[ptr1] = cm;
            ptr1 = cp + 1;
            cm = m->son[ptr1];
            len1 = len;
        } else {
            m->son[ptr0] = cm;
            ptr0 = cp;
            cm = m->son[ptr0];
            len0 = len;
        }
    }
    bt4_move_pos(m);
    return off;
}

/* lz_window.go:165-186, with index relative to iw.pos as in the reference */
static inline uint8_t bt4_index_byte(const bt4 *m, int32_t index) {
    return m->data[m->base + (int64_t)m->pos + index];
}
static uint32_t bt4_match_len(const bt4 *m, int32_t index, uint32_t distance, uint32_t limit) {
    int64_t p = m->base + (int64_t)m->pos + index;
    uint64_t left = m->n - (uint64_t)p;
    if (limit > left) limit = (uint32_t)left;
    distance++;
    const uint8_t *a = m->data + p;
    uint32_t r = 0;
    while (r < limit && a[r] == a[(int64_t)r - (int64_t)distance]) r++;
    return r;
}

int lzo_bt4_matches(const uint8_t *data, size_t n, uint32_t dict_size, uint32_t fb,
                    uint32_t cut_value, uint32_t *counts, uint32_t *pairs,
                    size_t pairs_cap_u32, size_t *total_pairs) {
    pthread_once(&g_once, build_tables);
    if (dict_size < 1 || fb < 5 || fb > MATCH_MAX) return LZO_ERR_ARG;
    bt4 m;
    int rc = bt4_init(&m, data, n, dict_size, fb, cut_value);
    if (rc) return rc;
    uint32_t tmp[MATCH_MAX * 2 + 2];
    size_t w = 0, total = 0;
    for (size_t i = 0; i < n; i++) {
        uint32_t k = bt4_step(&m, tmp);
        if (counts) counts[i] = k / 2;
        total += k / 2;
        if (pairs) {
            if (w + k > pairs_cap_u32) {
                bt4_free(&m);
                return LZO_ERR_OUT_CAP;
            }
            memcpy(pairs + w, tmp, k * sizeof(uint32_t));
            w += k;
        }
    }
    if (total_pairs) *total_pairs = total;
    bt4_free(&m);
    return LZO_OK;
}

/* ======================================================================
 * Range encoder (range_coder.go:131-220)
 * ==================================================================== */
typedef struct {
    uint8_t *out;
    size_t cap, n;
    int overflow;
    uint64_t low;
    uint32_t cache_size, cache, range;
} rc_enc;

static inline void rc_put(rc_enc *r, uint8_t b) {
    if (r->n < r->cap)
        r->out[r->n++] = b;
    else
        r->overflow = 1;
}
static void rc_shift_low(rc_enc *r) {
    uint32_t hi = (uint32_t)(r->low >> 32);
    if (hi != 0 || r->low < 0xFF000000ull) {
        uint32_t t = r->cache;
        do {
            rc_put(r, (uint8_t)(t + hi));
            t = 0xFF;
        } while (--r->cache_size != 0);
        r->cache = (uint32_t)r->low >> 24;
    }
    r->cache_size++;
    r->low = (uint64_t)((uint32_t)r->low << 8);
}
static inline void rc_bit(rc_enc *r, prob_t *p, uint32_t bit) {
    uint32_t pr = *p, bound = (r->range >> PROB_BITS) * pr;
    if (bit == 0) {
        r->range = bound;
        *p = (prob_t)(pr + ((PROB_TOTAL - pr) >> MOVE_BITS));
    } else {
        r->low += bound;
        r->range -= bound;
        *p = (prob_t)(pr - (pr >> MOVE_BITS));
    }
    if (r->range < TOP_VALUE) {
        r->range <<= 8;
        rc_shift_low(r);
    }
}
static void rc_direct(rc_enc *r, uint32_t v, uint32_t nbits) {
    for (int32_t i = (int32_t)nbits - 1; i >= 0; i--) {
        r->range >>= 1;
        if ((v >> i) & 1) r->low += r->range;
        if (r->range < TOP_VALUE) {
            r->range <<= 8;
            rc_shift_low(r);
        }
    }
}
static void rc_flush(rc_enc *r) {
    for (int i = 0; i < 5; i++) rc_shift_low(r);
}

/* bit trees: range_bit_tree_coder.go:53-117 */
static void bt_encode(rc_enc *r, prob_t *m, uint32_t nbits, uint32_t sym) {
    uint32_t k = 1;
    for (uint32_t i = nbits; i != 0;) {
        i--;
        uint32_t b = (sym >> i) & 1;
        rc_bit(r, m + k, b);
        k = (k << 1) | b;
    }
}
static void bt_rev_encode(rc_enc *r, prob_t *m, uint32_t nbits, uint32_t sym) {
    /* m may point one before the first used model: index 0 is never touched */
    uint32_t k = 1;
    for (uint32_t i = 0; i < nbits; i++) {
        uint32_t b = sym & 1;
        rc_bit(r, m + k, b);
        k = (k << 1) | b;
        sym >>= 1;
    }
}
static uint32_t bt_price(const prob_t *m, uint32_t nbits, uint32_t sym) {
    uint32_t res = 0, k = 1;
    for (uint32_t i = nbits; i != 0;) {
        i--;
        uint32_t b = (sym >> i) & 1;
        res += price_bit(m[k], b);
        k = (k << 1) + b;
    }
    return res;
}
static uint32_t bt_rev_price(const prob_t *m, uint32_t nbits, uint32_t sym) {
    uint32_t res = 0, k = 1;
    for (uint32_t i = nbits; i != 0; i--) {
        uint32_t b = sym & 1;
        sym >>= 1;
        res += price_bit(m[k], b);
        k = (k << 1) | b;
    }
    return res;
}

/* ---- length coder with cached price table (lzma_len_coder.go) ---- */
typedef struct {
    prob_t choice[2];
    prob_t low[POS_STATES_MAX][1 << 3];
    prob_t mid[POS_STATES_MAX][1 << 3];
    prob_t high[1 << 8];
    uint32_t prices[N_LEN_SYMS << POS_STATES_BITS_MAX];
    uint32_t counters[POS_STATES_MAX];
    uint32_t table_size;
} len_enc;

static void len_set_prices(len_enc *c, uint32_t ps) { /* :66-88, :111-114 */
    uint32_t a0 = price0(c->choice[0]), a1 = price1(c->choice[0]);
    uint32_t b0 = a1 + price0(c->choice[1]), b1 = a1 + price1(c->choice[1]);
    uint32_t *pr = c->prices + ps * N_LEN_SYMS, n = c->table_size, i = 0;
    for (; i < LEN_LOW && i < n; i++) pr[i] = a0 + bt_price(c->low[ps], 3, i);
    for (; i < LEN_LOW + LEN_MID && i < n; i++) pr[i] = b0 + bt_price(c->mid[ps], 3, i - LEN_LOW);
    for (; i < n; i++) pr[i] = b1 + bt_price(c->high, 8, i - LEN_LOW - LEN_MID);
    c->counters[ps] = n;
}
static void len_init(len_enc *c, uint32_t table_size, uint32_t n_pos_states) { /* :98-109 */
    init_probs(c->choice, 2);
    init_probs(&c->low[0][0], POS_STATES_MAX * 8);
    init_probs(&c->mid[0][0], POS_STATES_MAX * 8);
    init_probs(c->high, 256);
    memset(c->prices, 0, sizeof c->prices);
    memset(c->counters, 0, sizeof c->counters);
    c->table_size = table_size;
    for (uint32_t ps = 0; ps < n_pos_states; ps++) len_set_prices(c, ps);
}
static void len_encode(len_enc *c, rc_enc *r, uint32_t sym, uint32_t ps) { /* :48-63,:120-126 */
    if (sym < LEN_LOW) {
        rc_bit(r, &c->choice[0], 0);
        bt_encode(r, c->low[ps], 3, sym);
    } else {
        uint32_t s = sym - LEN_LOW;
        rc_bit(r, &c->choice[0], 1);
        if (s < LEN_MID) {
            rc_bit(r, &c->choice[1], 0);
            bt_encode(r, c->mid[ps], 3, s);
        } else {
            rc_bit(r, &c->choice[1], 1);
            bt_encode(r, c->high, 8, s - LEN_MID);
        }
    }
    if (--c->counters[ps] == 0) len_set_prices(c, ps);
}
static inline uint32_t len_price(const len_enc *c, uint32_t sym, uint32_t ps) {
    return c->prices[ps * N_LEN_SYMS + sym];
}

/* ---- literal coder (lzma_lit_coder.go) ---- */
static void lit_encode(rc_enc *r, prob_t *p, uint8_t sym) { /* :47-55 */
    uint32_t ctx = 1;
    for (int i = 7; i >= 0; i--) {
        uint32_t b = (sym >> i) & 1;
        rc_bit(r, p + ctx, b);
        ctx = (ctx << 1) | b;
    }
}
static void lit_encode_matched(rc_enc *r, prob_t *p, uint8_t mb, uint8_t sym) { /* :57-76 */
    uint32_t ctx = 1;
    int same = 1;
    for (int i = 7; i >= 0; i--) {
        uint32_t b = (sym >> i) & 1, st = ctx;
        if (same) {
            uint32_t m = (mb >> i) & 1;
            st += (1 + m) << 8;
            same = (m == b);
        }
        rc_bit(r, p + st, b);
        ctx = (ctx << 1) | b;
    }
}
static uint32_t lit_price(const prob_t *p, int match_mode, uint8_t mb, uint8_t sym) { /* :78-102 */
    uint32_t price = 0, ctx = 1;
    int i = 7;
    if (match_mode) {
        for (; i >= 0; i--) {
            uint32_t m = (mb >> i) & 1, b = (sym >> i) & 1;
            price += price_bit(p[((1 + m) << 8) + ctx], b);
            ctx = (ctx << 1) | b;
            if (m != b) {
                i--;
                break;
            }
        }
    }
    for (; i >= 0; i--) {
        uint32_t b = (sym >> i) & 1;
        price += price_bit(p[ctx], b);
        ctx = (ctx << 1) | b;
    }
    return price;
}

/* ======================================================================
 * Encoder (lzma_encoder.go:167-1107)
 * ==================================================================== */
typedef struct {
    uint32_t state, pos_prev2, back_prev2, price, pos_prev, back_prev, backs[4];
    uint8_t prev1_is_char, prev2;
} optimal;

typedef struct {
    rc_enc rc;
    bt4 mf;
    uint32_t fb, lc, lp, pb, dict_size, dist_table_size;
    int eos;
    optimal *opt; /* N_OPTS */
    prob_t is_match[N_STATES << POS_STATES_BITS_MAX], is_rep[N_STATES], is_rep_g0[N_STATES],
        is_rep_g1[N_STATES], is_rep_g2[N_STATES], is_rep0_long[N_STATES << POS_STATES_BITS_MAX];
    prob_t pos_slot_enc[N_LEN_TO_POS][1 << POS_SLOT_BITS];
    prob_t pos_enc[N_FULL_DIST - END_POS_MODEL];
    prob_t align_enc[1 << ALIGN_BITS];
    len_enc len, rep_len;
    prob_t *lit; /* 0x300 << (lc+lp) */
    uint32_t lit_pos_mask;
    uint32_t match_dist[MATCH_MAX * 2 + 2];
    uint32_t longest_len, dist_pairs, add_offset, opt_end, opt_cur;
    int longest_found;
    uint32_t pos_slot_prices[1 << (POS_SLOT_BITS + 2)], dist_prices[N_FULL_DIST << 2],
        align_prices[ALIGN_SIZE], align_price_count, match_price_count;
    uint32_t temp_prices[N_FULL_DIST]; /* package-global in the reference (:799) */
    uint32_t pos_state_mask;
    int64_t now_pos;
    int finished;
    uint32_t state;
    uint8_t prev_byte;
    uint32_t rep_dist[N_REPS], reps[N_REPS], rep_lens[N_REPS], back_res;
} enc;

static inline prob_t *lit_sub(enc *z, uint32_t pos, uint8_t prev) { /* lit_coder.go:126-128 */
    uint32_t idx = ((pos & z->lit_pos_mask) << z->lc) + ((uint32_t)prev >> (8 - z->lc));
    return z->lit + (size_t)idx * 0x300;
}

static uint32_t read_match_distances(enc *z) { /* :274-285 */
    uint32_t res = 0;
    z->dist_pairs = bt4_step(&z->mf, z->match_dist);
    if (z->dist_pairs > 0) {
        res = z->match_dist[z->dist_pairs - 2];
        if (res == z->fb)
            res += bt4_match_len(&z->mf, (int32_t)res - 1, z->match_dist[z->dist_pairs - 1],
                                 MATCH_MAX - res);
    }
    z->add_offset++;
    return res;
}
static void enc_move_pos(enc *z, uint32_t num) { /* :287-292 */
    if (num > 0) {
        z->add_offset += num;
        for (uint32_t i = 0; i < num; i++) bt4_step(&z->mf, NULL);
    }
}
static uint32_t pure_rep_price(enc *z, uint32_t ri, uint32_t st, uint32_t ps) { /* :294-308 */
    uint32_t p;
    if (ri == 0) {
        p = price0(z->is_rep_g0[st]);
        p += price1(z->is_rep0_long[(st << POS_STATES_BITS_MAX) + ps]);
    } else {
        p = price1(z->is_rep_g0[st]);
        if (ri == 1)
            p += price0(z->is_rep_g1[st]);
        else {
            p += price1(z->is_rep_g1[st]);
            p += price_bit(z->is_rep_g2[st], ri - 2);
        }
    }
    return p;
}
static inline uint32_t rep_price(enc *z, uint32_t ri, uint32_t len, uint32_t st, uint32_t ps) {
    return len_price(&z->rep_len, len - MATCH_MIN, ps) + pure_rep_price(z, ri, st, ps);
}
static uint32_t pos_len_price(enc *z, uint32_t pos, uint32_t len, uint32_t ps) { /* :316-325 */
    uint32_t lps = len_to_pos_state(len), p;
    if (pos < N_FULL_DIST)
        p = z->dist_prices[lps * N_FULL_DIST + pos];
    else
        p = z->pos_slot_prices[(lps << POS_SLOT_BITS) + pos_slot2(pos)] +
            z->align_prices[pos & ALIGN_MASK];
    return p + len_price(&z->len, len - MATCH_MIN, ps);
}
static inline uint32_t rep_len1_price(enc *z, uint32_t st, uint32_t ps) { /* :327-329 */
    return price0(z->is_rep_g0[st]) + price0(z->is_rep0_long[(st << POS_STATES_BITS_MAX) + ps]);
}

static uint32_t backward(enc *z, uint32_t cur) { /* :331-357 */
    optimal *o = z->opt;
    z->opt_end = cur;
    uint32_t pos_mem = o[cur].pos_prev, back_mem = o[cur].back_prev;
    do {
        if (o[cur].prev1_is_char) {
            o[pos_mem].back_prev = 0xFFFFFFFFu;
            o[pos_mem].prev1_is_char = 0;
            o[pos_mem].pos_prev = pos_mem - 1;
            if (o[cur].prev2) {
                o[pos_mem - 1].prev1_is_char = 0;
                o[pos_mem - 1].pos_prev = o[cur].pos_prev2;
                o[pos_mem - 1].back_prev = o[cur].back_prev2;
            }
        }
        uint32_t pp = pos_mem, bc = back_mem;
        back_mem = o[pp].back_prev;
        pos_mem = o[pp].pos_prev;
        o[pp].back_prev = bc;
        o[pp].pos_prev = cur;
        cur = pp;
    } while (cur > 0);
    z->back_res = o[0].back_prev;
    z->opt_cur = o[0].pos_prev;
    return z->opt_cur;
}

#define IDX(st, ps) (((st) << POS_STATES_BITS_MAX) + (ps))
static inline void extend_end(enc *z, uint32_t *len_end, uint32_t upto) {
    while (*len_end < upto) z->opt[++*len_end].price = INF_PRICE;
}

static uint32_t get_optimum(enc *z, uint32_t position) { /* :359-797 */
    optimal *o = z->opt;
    const uint32_t fb = z->fb;
    if (z->opt_end != z->opt_cur) {
        uint32_t r = o[z->opt_cur].pos_prev - z->opt_cur;
        z->back_res = o[z->opt_cur].back_prev;
        z->opt_cur = o[z->opt_cur].pos_prev;
        return r;
    }
    z->opt_end = z->opt_cur = 0;
    uint32_t len_main;
    if (!z->longest_found)
        len_main = read_match_distances(z);
    else {
        len_main = z->longest_len;
        z->longest_found = 0;
    }
    uint32_t dist_pairs = z->dist_pairs;
    uint64_t av64 = bt4_avail(&z->mf) + 1;
    if (av64 < 2) {
        z->back_res = 0xFFFFFFFFu;
        return 1;
    }
    uint32_t avail = av64 > MATCH_MAX ? MATCH_MAX : (uint32_t)av64;
    (void)avail;
    uint32_t rep_max = 0;
    for (uint32_t i = 0; i < N_REPS; i++) {
        z->reps[i] = z->rep_dist[i];
        z->rep_lens[i] = bt4_match_len(&z->mf, -1, z->reps[i], MATCH_MAX);
        if (z->rep_lens[i] > z->rep_lens[rep_max]) rep_max = i;
    }
    if (z->rep_lens[rep_max] >= fb) {
        z->back_res = rep_max;
        uint32_t r = z->rep_lens[rep_max];
        enc_move_pos(z, r - 1);
        return r;
    }
    if (len_main >= fb) {
        z->back_res = z->match_dist[dist_pairs - 1] + N_REPS;
        enc_move_pos(z, len_main - 1);
        return len_main;
    }
    uint8_t cur_byte = bt4_index_byte(&z->mf, -1);
    uint8_t match_byte = bt4_index_byte(&z->mf, 0 - (int32_t)z->rep_dist[0] - 1 - 1);
    if (len_main < 2 && cur_byte != match_byte && z->rep_lens[rep_max] < 2) {
        z->back_res = 0xFFFFFFFFu;
        return 1;
    }
    o[0].state = z->state;
    uint32_t ps = position & z->pos_state_mask;
    o[1].price = price0(z->is_match[IDX(z->state, ps)]) +
                 lit_price(lit_sub(z, position, z->prev_byte), !st_is_char(z->state), match_byte,
                           cur_byte);
    o[1].back_prev = 0xFFFFFFFFu;
    o[1].prev1_is_char = 0;

    uint32_t match_price = price1(z->is_match[IDX(z->state, ps)]);
    uint32_t rep_match_price = match_price + price1(z->is_rep[z->state]);
    if (match_byte == cur_byte) {
        uint32_t sp = rep_match_price + rep_len1_price(z, z->state, ps);
        if (sp < o[1].price) {
            o[1].price = sp;
            o[1].back_prev = 0;
            o[1].prev1_is_char = 0;
        }
    }
    uint32_t len_end = z->rep_lens[rep_max];
    if (len_main > len_end) len_end = len_main;
    if (len_end < 2) {
        z->back_res = o[1].back_prev;
        return 1;
    }
    o[1].pos_prev = 0;
    for (int i = 0; i < 4; i++) o[0].backs[i] = z->reps[i];
    {
        uint32_t l = len_end;
        do o[l].price = INF_PRICE;
        while (--l >= 2);
    }
    for (uint32_t i = 0; i < N_REPS; i++) {
        uint32_t rl = z->rep_lens[i];
        if (rl < 2) continue;
        uint32_t price = rep_match_price + pure_rep_price(z, i, z->state, ps);
        do {
            uint32_t cp = price + len_price(&z->rep_len, rl - 2, ps);
            optimal *t = &o[rl];
            if (cp < t->price) {
                t->price = cp;
                t->pos_prev = 0;
                t->back_prev = i;
                t->prev1_is_char = 0;
            }
        } while (--rl >= 2);
    }
    uint32_t normal_match_price = match_price + price0(z->is_rep[z->state]);
    uint32_t len = z->rep_lens[0] >= 2 ? z->rep_lens[0] + 1 : 2;
    if (len <= len_main) {
        uint32_t offs = 0;
        while (len > z->match_dist[offs]) offs += 2;
        for (;; len++) {
            uint32_t d = z->match_dist[offs + 1];
            uint32_t cp = normal_match_price + pos_len_price(z, d, len, ps);
            optimal *t = &o[len];
            if (cp < t->price) {
                t->price = cp;
                t->pos_prev = 0;
                t->back_prev = d + N_REPS;
                t->prev1_is_char = 0;
            }
            if (len == z->match_dist[offs]) {
                offs += 2;
                if (offs == dist_pairs) break;
            }
        }
    }

    uint32_t cur = 0;
    for (;;) {
        cur++;
        if (cur == len_end) return backward(z, cur);
        uint32_t new_len = read_match_distances(z);
        dist_pairs = z->dist_pairs;
        if (new_len >= fb) {
            z->longest_len = new_len;
            z->longest_found = 1;
            return backward(z, cur);
        }
        position++;
        uint32_t pos_prev = o[cur].pos_prev, state;
        if (o[cur].prev1_is_char) {
            pos_prev--;
            if (o[cur].prev2) {
                state = o[o[cur].pos_prev2].state;
                state = o[cur].back_prev2 < N_REPS ? st_rep(state) : st_match(state);
            } else
                state = o[pos_prev].state;
            state = st_char(state);
        } else
            state = o[pos_prev].state;
        if (pos_prev == cur - 1) {
            state = o[cur].back_prev == 0 ? st_shortrep(state) : st_char(state);
        } else {
            uint32_t pos;
            if (o[cur].prev1_is_char && o[cur].prev2) {
                pos_prev = o[cur].pos_prev2;
                pos = o[cur].back_prev2;
                state = st_rep(state);
            } else {
                pos = o[cur].back_prev;
                state = pos < N_REPS ? st_rep(state) : st_match(state);
            }
            const optimal *q = &o[pos_prev];
            if (pos < N_REPS) {
                if (pos == 0) {
                    z->reps[0] = q->backs[0], z->reps[1] = q->backs[1];
                    z->reps[2] = q->backs[2], z->reps[3] = q->backs[3];
                } else if (pos == 1) {
                    z->reps[0] = q->backs[1], z->reps[1] = q->backs[0];
                    z->reps[2] = q->backs[2], z->reps[3] = q->backs[3];
                } else if (pos == 2) {
                    z->reps[0] = q->backs[2], z->reps[1] = q->backs[0];
                    z->reps[2] = q->backs[1], z->reps[3] = q->backs[3];
                } else {
                    z->reps[0] = q->backs[3], z->reps[1] = q->backs[0];
                    z->reps[2] = q->backs[1], z->reps[3] = q->backs[2];
                }
            } else {
                z->reps[0] = pos - N_REPS, z->reps[1] = q->backs[0];
                z->reps[2] = q->backs[1], z->reps[3] = q->backs[2];
            }
        }
        o[cur].state = state;
        for (int i = 0; i < 4; i++) o[cur].backs[i] = z->reps[i];
        uint32_t cur_price = o[cur].price;
        cur_byte = bt4_index_byte(&z->mf, -1);
        match_byte = bt4_index_byte(&z->mf, 0 - (int32_t)z->reps[0] - 1 - 1);
        ps = position & z->pos_state_mask;
        uint32_t cur_and1 = cur_price + price0(z->is_match[IDX(state, ps)]) +
                            lit_price(lit_sub(z, position, bt4_index_byte(&z->mf, -2)),
                                      !st_is_char(state), match_byte, cur_byte);
        optimal *nx = &o[cur + 1];
        int next_is_char = 0;
        if (cur_and1 < nx->price) {
            nx->price = cur_and1;
            nx->pos_prev = cur;
            nx->back_prev = 0xFFFFFFFFu;
            nx->prev1_is_char = 0;
            next_is_char = 1;
        }
        match_price = cur_price + price1(z->is_match[IDX(state, ps)]);
        rep_match_price = match_price + price1(z->is_rep[state]);
        if (match_byte == cur_byte && !(nx->pos_prev < cur && nx->back_prev == 0)) {
            uint32_t sp = rep_match_price + rep_len1_price(z, state, ps);
            if (sp <= nx->price) {
                nx->price = sp;
                nx->pos_prev = cur;
                nx->back_prev = 0;
                nx->prev1_is_char = 0;
                next_is_char = 1;
            }
        }
        uint64_t avf64 = bt4_avail(&z->mf) + 1;
        uint32_t avail_full = N_OPTS - 1 - cur;
        if (avf64 < avail_full) avail_full = (uint32_t)avf64;
        uint32_t av = avail_full;
        if (av < 2) continue;
        if (av > fb) av = fb;
        if (!next_is_char && match_byte != cur_byte) {
            uint32_t t = avail_full - 1 < fb ? avail_full - 1 : fb;
            uint32_t l2 = bt4_match_len(&z->mf, 0, z->reps[0], t);
            if (l2 >= 2) {
                uint32_t s2 = st_char(state);
                uint32_t psn = (position + 1) & z->pos_state_mask;
                uint32_t nrmp = cur_and1 + price1(z->is_match[IDX(s2, psn)]) + price1(z->is_rep[s2]);
                uint32_t offset = cur + 1 + l2;
                extend_end(z, &len_end, offset);
                uint32_t cp = nrmp + rep_price(z, 0, l2, s2, psn);
                optimal *t2 = &o[offset];
                if (cp < t2->price) {
                    t2->price = cp;
                    t2->pos_prev = cur + 1;
                    t2->back_prev = 0;
                    t2->prev1_is_char = 1;
                    t2->prev2 = 0;
                }
            }
        }
        uint32_t start_len = 2;
        for (uint32_t ri = 0; ri < N_REPS; ri++) {
            uint32_t lt = bt4_match_len(&z->mf, -1, z->reps[ri], av);
            if (lt < 2) continue;
            uint32_t lt_keep = lt;
            do {
                extend_end(z, &len_end, cur + lt);
                uint32_t cp = rep_match_price + rep_price(z, ri, lt, state, ps);
                optimal *t2 = &o[cur + lt];
                if (cp < t2->price) {
                    t2->price = cp;
                    t2->pos_prev = cur;
                    t2->back_prev = ri;
                    t2->prev1_is_char = 0;
                }
            } while (--lt >= 2);
            lt = lt_keep;
            if (ri == 0) start_len = lt + 1;
            if (lt < avail_full) {
                uint32_t t = avail_full - 1 - lt < fb ? avail_full - 1 - lt : fb;
                uint32_t l2 = bt4_match_len(&z->mf, (int32_t)lt, z->reps[ri], t);
                if (l2 >= 2) {
                    uint32_t s2 = st_rep(state);
                    uint32_t psn = (position + lt) & z->pos_state_mask;
                    uint32_t calcp =
                        rep_match_price + rep_price(z, ri, lt, state, ps) +
                        price0(z->is_match[IDX(s2, psn)]) +
                        lit_price(lit_sub(z, position + lt, bt4_index_byte(&z->mf, (int32_t)lt - 2)),
                                  1,
                                  bt4_index_byte(&z->mf, (int32_t)lt - 1 - (int32_t)(z->reps[ri] + 1)),
                                  bt4_index_byte(&z->mf, (int32_t)lt - 1));
                    s2 = st_char(s2);
                    psn = (position + lt + 1) & z->pos_state_mask;
                    uint32_t nmp = calcp + price1(z->is_match[IDX(s2, psn)]);
                    uint32_t nrmp = nmp + price1(z->is_rep[s2]);
                    uint32_t offset = lt + 1 + l2;
                    extend_end(z, &len_end, cur + offset);
                    uint32_t cp = nrmp + rep_price(z, 0, l2, s2, psn);
                    optimal *t2 = &o[cur + offset];
                    if (cp < t2->price) {
                        t2->price = cp;
                        t2->pos_prev = cur + lt + 1;
                        t2->back_prev = 0;
                        t2->prev1_is_char = 1;
                        t2->prev2 = 1;
                        t2->pos_prev2 = cur;
                        t2->back_prev2 = ri;
                    }
                }
            }
        }
        if (new_len > av) {
            new_len = av;
            for (dist_pairs = 0; new_len > z->match_dist[dist_pairs]; dist_pairs += 2) {
            }
            z->match_dist[dist_pairs] = new_len;
            dist_pairs += 2;
        }
        if (new_len >= start_len) {
            normal_match_price = match_price + price0(z->is_rep[state]);
            extend_end(z, &len_end, cur + new_len);
            uint32_t offs = 0;
            while (start_len > z->match_dist[offs]) offs += 2;
            for (uint32_t lt = start_len;; lt++) {
                uint32_t cb = z->match_dist[offs + 1];
                uint32_t cp = normal_match_price + pos_len_price(z, cb, lt, ps);
                optimal *t2 = &o[cur + lt];
                if (cp < t2->price) {
                    t2->price = cp;
                    t2->pos_prev = cur;
                    t2->back_prev = cb + N_REPS;
                    t2->prev1_is_char = 0;
                }
                if (lt == z->match_dist[offs]) {
                    if (lt < avail_full) {
                        uint32_t t = avail_full - 1 - lt < fb ? avail_full - 1 - lt : fb;
                        uint32_t l2 = bt4_match_len(&z->mf, (int32_t)lt, cb, t);
                        if (l2 >= 2) {
                            uint32_t s2 = st_match(state);
                            uint32_t psn = (position + lt) & z->pos_state_mask;
                            uint32_t calcp =
                                cp + price0(z->is_match[IDX(s2, psn)]) +
                                lit_price(
                                    lit_sub(z, position + lt, bt4_index_byte(&z->mf, (int32_t)lt - 2)),
                                    1, bt4_index_byte(&z->mf, (int32_t)lt - ((int32_t)cb + 1) - 1),
                                    bt4_index_byte(&z->mf, (int32_t)lt - 1));
                            s2 = st_char(s2);
                            psn = (position + lt + 1) & z->pos_state_mask;
                            uint32_t nmp = calcp + price1(z->is_match[IDX(s2, psn)]);
                            uint32_t nrmp = nmp + price1(z->is_rep[s2]);
                            uint32_t offset = lt + 1 + l2;
                            extend_end(z, &len_end, cur + offset);
                            cp = nrmp + rep_price(z, 0, l2, s2, psn);
                            t2 = &o[cur + offset];
                            if (cp < t2->price) {
                                t2->price = cp;
                                t2->pos_prev = cur + lt + 1;
                                t2->back_prev = 0;
                                t2->prev1_is_char = 1;
                                t2->prev2 = 1;
                                t2->pos_prev2 = cur;
                                t2->back_prev2 = cb + N_REPS;
                            }
                        }
                    }
                    offs += 2;
                    if (offs == dist_pairs) break;
                }
            }
        }
    }
}

static void fill_dist_prices(enc *z) { /* :801-827 */
    for (uint32_t i = START_POS_MODEL; i < N_FULL_DIST; i++) {
        uint32_t slot = pos_slot(i), fbits = (slot >> 1) - 1, base = (2 | (slot & 1)) << fbits;
        /* models start at base-slot-1, index 0 of the sub-tree is never read */
        z->temp_prices[i] = bt_rev_price(z->pos_enc + (int32_t)(base - slot - 1), fbits, i - base);
    }
    for (uint32_t lps = 0; lps < N_LEN_TO_POS; lps++) {
        uint32_t st = lps << POS_SLOT_BITS, slot;
        for (slot = 0; slot < z->dist_table_size; slot++)
            z->pos_slot_prices[st + slot] = bt_price(z->pos_slot_enc[lps], POS_SLOT_BITS, slot);
        for (slot = END_POS_MODEL; slot < z->dist_table_size; slot++)
            z->pos_slot_prices[st + slot] += ((slot >> 1) - 1 - ALIGN_BITS) << PRICE_SHIFT;
        uint32_t st2 = lps * N_FULL_DIST, i;
        for (i = 0; i < START_POS_MODEL; i++) z->dist_prices[st2 + i] = z->pos_slot_prices[st + i];
        for (; i < N_FULL_DIST; i++)
            z->dist_prices[st2 + i] = z->pos_slot_prices[st + pos_slot(i)] + z->temp_prices[i];
    }
    z->match_price_count = 0;
}
static void fill_align_prices(enc *z) { /* :829-834 */
    for (uint32_t i = 0; i < ALIGN_SIZE; i++)
        z->align_prices[i] = bt_rev_price(z->align_enc, ALIGN_BITS, i);
    z->align_price_count = 0;
}

static void write_end_marker(enc *z, uint32_t ps) { /* :836-852 */
    if (!z->eos) return;
    rc_bit(&z->rc, &z->is_match[IDX(z->state, ps)], 1);
    rc_bit(&z->rc, &z->is_rep[z->state], 0);
    z->state = st_match(z->state);
    len_encode(&z->len, &z->rc, 0, ps);
    bt_encode(&z->rc, z->pos_slot_enc[len_to_pos_state(MATCH_MIN)], POS_SLOT_BITS,
              (1u << POS_SLOT_BITS) - 1);
    uint32_t fbits = 30, reduced = (1u << fbits) - 1;
    rc_direct(&z->rc, reduced >> ALIGN_BITS, fbits - ALIGN_BITS);
    bt_rev_encode(&z->rc, z->align_enc, ALIGN_BITS, reduced & ALIGN_MASK);
}
static void enc_flush(enc *z, uint32_t now_pos) { /* :854-857 */
    write_end_marker(z, now_pos & z->pos_state_mask);
    rc_flush(&z->rc);
}

static void code_one_block(enc *z) { /* :859-978 */
    z->finished = 1;
    int64_t prev_progress = z->now_pos;
    if (z->now_pos == 0) {
        if (bt4_avail(&z->mf) == 0) {
            enc_flush(z, (uint32_t)z->now_pos);
            return;
        }
        (void)read_match_distances(z);
        rc_bit(&z->rc, &z->is_match[IDX(z->state, (uint32_t)z->now_pos & z->pos_state_mask)], 0);
        z->state = st_char(z->state);
        uint8_t cb = bt4_index_byte(&z->mf, 0 - (int32_t)z->add_offset);
        lit_encode(&z->rc, lit_sub(z, (uint32_t)z->now_pos, z->prev_byte), cb);
        z->prev_byte = cb;
        z->add_offset--;
        z->now_pos++;
    }
    if (bt4_avail(&z->mf) == 0) {
        enc_flush(z, (uint32_t)z->now_pos);
        return;
    }
    for (;;) {
        uint32_t len = get_optimum(z, (uint32_t)z->now_pos);
        uint32_t pos = z->back_res;
        uint32_t ps = (uint32_t)z->now_pos & z->pos_state_mask;
        uint32_t cs = IDX(z->state, ps);
        if (len == 1 && pos == 0xFFFFFFFFu) {
            rc_bit(&z->rc, &z->is_match[cs], 0);
            uint8_t cb = bt4_index_byte(&z->mf, 0 - (int32_t)z->add_offset);
            prob_t *sub = lit_sub(z, (uint32_t)z->now_pos, z->prev_byte);
            if (!st_is_char(z->state)) {
                uint8_t mb = bt4_index_byte(&z->mf, 0 - (int32_t)z->rep_dist[0] - 1 - (int32_t)z->add_offset);
                lit_encode_matched(&z->rc, sub, mb, cb);
            } else
                lit_encode(&z->rc, sub, cb);
            z->prev_byte = cb;
            z->state = st_char(z->state);
        } else {
            rc_bit(&z->rc, &z->is_match[cs], 1);
            if (pos < N_REPS) {
                rc_bit(&z->rc, &z->is_rep[z->state], 1);
                if (pos == 0) {
                    rc_bit(&z->rc, &z->is_rep_g0[z->state], 0);
                    rc_bit(&z->rc, &z->is_rep0_long[cs], len == 1 ? 0 : 1);
                } else {
                    rc_bit(&z->rc, &z->is_rep_g0[z->state], 1);
                    if (pos == 1)
                        rc_bit(&z->rc, &z->is_rep_g1[z->state], 0);
                    else {
                        rc_bit(&z->rc, &z->is_rep_g1[z->state], 1);
                        rc_bit(&z->rc, &z->is_rep_g2[z->state], pos - 2);
                    }
                }
                if (len == 1)
                    z->state = st_shortrep(z->state);
                else {
                    len_encode(&z->rep_len, &z->rc, len - MATCH_MIN, ps);
                    z->state = st_rep(z->state);
                }
                uint32_t d = z->rep_dist[pos];
                if (pos != 0) {
                    for (uint32_t i = pos; i >= 1; i--) z->rep_dist[i] = z->rep_dist[i - 1];
                    z->rep_dist[0] = d;
                }
            } else {
                rc_bit(&z->rc, &z->is_rep[z->state], 0);
                z->state = st_match(z->state);
                len_encode(&z->len, &z->rc, len - MATCH_MIN, ps);
                pos -= N_REPS;
                uint32_t slot = pos_slot(pos);
                bt_encode(&z->rc, z->pos_slot_enc[len_to_pos_state(len)], POS_SLOT_BITS, slot);
                if (slot >= START_POS_MODEL) {
                    uint32_t fbits = (slot >> 1) - 1, base = (2 | (slot & 1)) << fbits;
                    uint32_t reduced = pos - base;
                    if (slot < END_POS_MODEL)
                        bt_rev_encode(&z->rc, z->pos_enc + (int32_t)(base - slot - 1), fbits, reduced);
                    else {
                        rc_direct(&z->rc, reduced >> ALIGN_BITS, fbits - ALIGN_BITS);
                        bt_rev_encode(&z->rc, z->align_enc, ALIGN_BITS, reduced & ALIGN_MASK);
                        z->align_price_count++;
                    }
                }
                for (uint32_t i = N_REPS - 1; i >= 1; i--) z->rep_dist[i] = z->rep_dist[i - 1];
                z->rep_dist[0] = pos;
                z->match_price_count++;
            }
            z->prev_byte = bt4_index_byte(&z->mf, (int32_t)len - 1 - (int32_t)z->add_offset);
        }
        z->add_offset -= len;
        z->now_pos += len;
        if (z->add_offset == 0) {
            if (z->match_price_count >= (1 << 7)) fill_dist_prices(z);
            if (z->align_price_count >= ALIGN_SIZE) fill_align_prices(z);
            if (bt4_avail(&z->mf) == 0) {
                enc_flush(z, (uint32_t)z->now_pos);
                return;
            }
            if (z->now_pos - prev_progress >= (1 << 12)) {
                z->finished = 0;
                return;
            }
        }
    }
}

/* lzma_encoder.go:97-129 */
int lzo_params_from_level(int level, int eos, lzo_params *p) {
    static const uint32_t dl[10] = {0, 16, 18, 20, 22, 23, 24, 25, 26, 27};
    static const uint32_t fbs[10] = {0, 64, 64, 64, 128, 128, 128, 256, 256, 256};
    if (level < 1 || level > 9) return LZO_ERR_ARG;
    p->dict_log = dl[level];
    p->fb = fbs[level];
    p->lc = 3;
    p->lp = 0;
    p->pb = 2;
    p->cut_value = 0;
    p->eos = eos;
    return LZO_OK;
}

size_t lzo_encode_bound(size_t n) { return n + n / 8 + 128; }

int lzo_encode(const uint8_t *in, size_t n, const lzo_params *p, uint8_t *out, size_t out_cap,
               size_t *out_n) { /* lzma_encoder.go:989-1107 */
    pthread_once(&g_once, build_tables);
    if (p->dict_log < 12 || p->dict_log > 29 || p->fb < 5 || p->fb > 273 || p->lc > 8 ||
        p->lp > 4 || p->pb > 4)
        return LZO_ERR_ARG;
    if (out_cap < HEADER_SIZE) return LZO_ERR_OUT_CAP;
    enc *z = (enc *)calloc(1, sizeof(enc));
    if (!z) return LZO_ERR_NOMEM;
    z->fb = p->fb;
    z->lc = p->lc;
    z->lp = p->lp;
    z->pb = p->pb;
    z->eos = p->eos != 0;
    z->dist_table_size = p->dict_log * 2;
    z->dict_size = 1u << p->dict_log;
    int64_t size = z->eos ? -1 : (int64_t)n;
    out[0] = (uint8_t)((p->pb * 5 + p->lp) * 9 + p->lc);
    for (int i = 0; i < 4; i++) out[1 + i] = (uint8_t)(z->dict_size >> (8 * i));
    for (int i = 0; i < 8; i++) out[5 + i] = (uint8_t)((uint64_t)size >> (8 * i));
    z->rc.out = out + HEADER_SIZE;
    z->rc.cap = out_cap - HEADER_SIZE;
    z->rc.cache_size = 1;
    z->rc.range = 0xFFFFFFFFu;
    int rc = bt4_init(&z->mf, in, n, z->dict_size, p->fb, p->cut_value);
    size_t lit_n = (size_t)0x300 << (p->lc + p->lp);
    z->opt = (optimal *)calloc(N_OPTS, sizeof(optimal));
    z->lit = (prob_t *)malloc(lit_n * sizeof(prob_t));
    if (rc || !z->opt || !z->lit) {
        bt4_free(&z->mf);
        free(z->opt);
        free(z->lit);
        free(z);
        return LZO_ERR_NOMEM;
    }
    init_probs(z->is_match, N_STATES << POS_STATES_BITS_MAX);
    init_probs(z->is_rep, N_STATES);
    init_probs(z->is_rep_g0, N_STATES);
    init_probs(z->is_rep_g1, N_STATES);
    init_probs(z->is_rep_g2, N_STATES);
    init_probs(z->is_rep0_long, N_STATES << POS_STATES_BITS_MAX);
    init_probs(&z->pos_slot_enc[0][0], N_LEN_TO_POS << POS_SLOT_BITS);
    init_probs(z->pos_enc, N_FULL_DIST - END_POS_MODEL);
    init_probs(z->align_enc, 1 << ALIGN_BITS);
    len_init(&z->len, p->fb + 1 - MATCH_MIN, 1u << p->pb);
    len_init(&z->rep_len, p->fb + 1 - MATCH_MIN, 1u << p->pb);
    init_probs(z->lit, lit_n);
    z->lit_pos_mask = (1u << p->lp) - 1;
    z->pos_state_mask = (1u << p->pb) - 1;
    fill_dist_prices(z);
    fill_align_prices(z);
    do code_one_block(z);
    while (!z->finished);
    int ret = z->rc.overflow ? LZO_ERR_OUT_CAP : LZO_OK;
    if (out_n) *out_n = HEADER_SIZE + z->rc.n;
    bt4_free(&z->mf);
    free(z->opt);
    free(z->lit);
    free(z);
    return ret;
}

/* ======================================================================
 * Decoder (lzma_decoder.go:168-349, range_coder.go:27-105, lz_window.go:9-80)
 * ==================================================================== */
typedef struct {
    const uint8_t *in;
    size_t n, p;
    uint32_t range, code;
    int eof;
} rc_dec;

static inline uint32_t rd_byte(rc_dec *r) {
    if (r->p < r->n) return r->in[r->p++];
    r->eof = 1;
    return 0;
}
static inline uint32_t rd_bit(rc_dec *r, prob_t *p) { /* range_coder.go:75-105 */
    uint32_t pr = *p, bound = (r->range >> PROB_BITS) * pr, bit;
    if (r->code < bound) {
        r->range = bound;
        *p = (prob_t)(pr + ((PROB_TOTAL - pr) >> MOVE_BITS));
        bit = 0;
    } else {
        r->range -= bound;
        r->code -= bound;
        *p = (prob_t)(pr - (pr >> MOVE_BITS));
        bit = 1;
    }
    if (r->range < TOP_VALUE) {
        r->code = (r->code << 8) | rd_byte(r);
        r->range <<= 8;
    }
    return bit;
}
static uint32_t rd_direct(rc_dec *r, uint32_t nbits) { /* range_coder.go:57-73 */
    uint32_t res = 0;
    for (uint32_t i = nbits; i != 0; i--) {
        r->range >>= 1;
        uint32_t t = (r->code - r->range) >> 31;
        r->code -= r->range & (t - 1);
        res = (res << 1) | (1 - t);
        if (r->range < TOP_VALUE) {
            r->code = (r->code << 8) | rd_byte(r);
            r->range <<= 8;
        }
    }
    return res;
}
static uint32_t bt_decode(rc_dec *r, prob_t *m, uint32_t nbits) { /* bit_tree_coder.go:19-27 */
    uint32_t k = 1;
    for (uint32_t i = nbits; i != 0; i--) k = (k << 1) + rd_bit(r, m + k);
    return k - (1u << nbits);
}
static uint32_t bt_rev_decode(rc_dec *r, prob_t *m, uint32_t nbits) { /* :29-51 */
    uint32_t k = 1, res = 0;
    for (uint32_t i = 0; i < nbits; i++) {
        uint32_t b = rd_bit(r, m + k);
        k = (k << 1) + b;
        res |= b << i;
    }
    return res;
}

typedef struct {
    prob_t choice[2], low[POS_STATES_MAX][8], mid[POS_STATES_MAX][8], high[256];
} len_dec;
static void len_dec_init(len_dec *c) {
    init_probs(c->choice, 2);
    init_probs(&c->low[0][0], POS_STATES_MAX * 8);
    init_probs(&c->mid[0][0], POS_STATES_MAX * 8);
    init_probs(c->high, 256);
}
static uint32_t len_decode(len_dec *c, rc_dec *r, uint32_t ps) { /* len_coder.go:28-46 */
    if (rd_bit(r, &c->choice[0]) == 0) return bt_decode(r, c->low[ps], 3);
    if (rd_bit(r, &c->choice[1]) == 0) return LEN_LOW + bt_decode(r, c->mid[ps], 3);
    return LEN_LOW + LEN_MID + bt_decode(r, c->high, 8);
}

int lzo_parse_header(const uint8_t *in, size_t n, uint32_t *lc, uint32_t *lp, uint32_t *pb,
                     uint32_t *dict_size, int64_t *unpack_size) { /* lzma_decoder.go:175-190,310-322 */
    if (n < HEADER_SIZE) return LZO_ERR_TRUNCATED;
    uint32_t d = in[0];
    if (d > 9 * 5 * 5) return LZO_ERR_HEADER;
    *lc = d % 9;
    d /= 9;
    *pb = d / 5;
    *lp = d % 5;
    if (*lc > 8 || *lp > 4 || *pb > 4) return LZO_ERR_HEADER;
    uint32_t ds = 0;
    for (int i = 0; i < 4; i++) ds += (uint32_t)in[1 + i] << (8 * i);
    *dict_size = ds;
    uint64_t us = 0;
    for (int i = 0; i < 8; i++) us |= (uint64_t)in[5 + i] << (8 * i);
    *unpack_size = (int64_t)us;
    return LZO_OK;
}

int lzo_decode(const uint8_t *in, size_t n, uint8_t *out, size_t out_cap, size_t *out_n,
               size_t *in_used) {
    uint32_t lc, lp, pb, dict;
    int64_t unpack;
    if (out_n) *out_n = 0;
    int hr = lzo_parse_header(in, n, &lc, &lp, &pb, &dict, &unpack);
    if (hr) return hr;
    rc_dec r = {in, n, HEADER_SIZE, 0xFFFFFFFFu, 0, 0};
    if (n < HEADER_SIZE + 5) return LZO_ERR_TRUNCATED; /* range_coder.go:46-50 */
    for (int i = 0; i < 5; i++) r.code = (r.code << 8) | rd_byte(&r);
    uint32_t dict_check = dict > 1 ? dict : 1;

    prob_t is_match[N_STATES << POS_STATES_BITS_MAX], is_rep[N_STATES], g0[N_STATES], g1[N_STATES],
        g2[N_STATES], rep0_long[N_STATES << POS_STATES_BITS_MAX];
    prob_t pos_slot_d[N_LEN_TO_POS][1 << POS_SLOT_BITS], pos_d[N_FULL_DIST - END_POS_MODEL],
        align_d[1 << ALIGN_BITS];
    len_dec len_c, rep_len_c;
    size_t lit_n = (size_t)0x300 << (lc + lp);
    prob_t *lit = (prob_t *)malloc(lit_n * sizeof(prob_t));
    if (!lit) return LZO_ERR_NOMEM;
    init_probs(lit, lit_n);
    init_probs(is_match, N_STATES << POS_STATES_BITS_MAX);
    init_probs(is_rep, N_STATES);
    init_probs(g0, N_STATES);
    init_probs(g1, N_STATES);
    init_probs(g2, N_STATES);
    init_probs(rep0_long, N_STATES << POS_STATES_BITS_MAX);
    init_probs(&pos_slot_d[0][0], N_LEN_TO_POS << POS_SLOT_BITS);
    init_probs(pos_d, N_FULL_DIST - END_POS_MODEL);
    init_probs(align_d, 1 << ALIGN_BITS);
    len_dec_init(&len_c);
    len_dec_init(&rep_len_c);
    uint32_t ps_mask = (1u << pb) - 1, lp_mask = (1u << lp) - 1;

    uint32_t state = 0, rep0 = 0, rep1 = 0, rep2 = 0, rep3 = 0;
    uint64_t now = 0;
    uint8_t prev = 0;
    int rc = LZO_OK;
    /* The reference keeps a circular window of max(dict,4096) bytes and flushes
     * it to the writer; with the whole output in memory, byte i of the window
     * history is simply out[now-1-dist]. */
    while (unpack < 0 || (int64_t)now < unpack) {
        uint32_t ps = (uint32_t)now & ps_mask;
        if (rd_bit(&r, &is_match[IDX(state, ps)]) == 0) {
            prob_t *sub = lit + (size_t)((((uint32_t)now & lp_mask) << lc) + ((uint32_t)prev >> (8 - lc))) * 0x300;
            uint32_t sym = 1;
            if (!st_is_char(state)) { /* lit_coder.go:27-44 */
                /* getByte(rep0) on a window that has not yet seen rep0+1 bytes
                 * reads the zero-initialised buffer (lz_window.go:74-80). */
                uint32_t mb = (uint64_t)rep0 < now ? out[now - 1 - rep0] : 0;
                while (sym < 0x100) {
                    uint32_t mbit = (mb >> 7) & 1;
                    mb <<= 1;
                    uint32_t b = rd_bit(&r, sub + ((1 + mbit) << 8) + sym);
                    sym = (sym << 1) | b;
                    if (mbit != b) {
                        while (sym < 0x100) sym = (sym << 1) | rd_bit(&r, sub + sym);
                        break;
                    }
                }
            } else {
                while (sym < 0x100) sym = (sym << 1) | rd_bit(&r, sub + sym);
            }
            if (r.eof) { rc = LZO_ERR_TRUNCATED; break; }
            prev = (uint8_t)sym;
            if (now >= out_cap) { rc = LZO_ERR_OUT_CAP; break; }
            out[now++] = prev;
            state = st_char(state);
        } else {
            uint32_t len;
            if (rd_bit(&r, &is_rep[state]) == 1) {
                len = 0;
                if (rd_bit(&r, &g0[state]) == 0) {
                    if (rd_bit(&r, &rep0_long[IDX(state, ps)]) == 0) {
                        state = st_shortrep(state);
                        len = 1;
                    }
                } else {
                    uint32_t d;
                    if (rd_bit(&r, &g1[state]) == 0)
                        d = rep1;
                    else {
                        if (rd_bit(&r, &g2[state]) == 0)
                            d = rep2;
                        else {
                            d = rep3;
                            rep3 = rep2;
                        }
                        rep2 = rep1;
                    }
                    rep1 = rep0;
                    rep0 = d;
                }
                if (len == 0) {
                    len = len_decode(&rep_len_c, &r, ps) + MATCH_MIN;
                    state = st_rep(state);
                }
            } else {
                rep3 = rep2;
                rep2 = rep1;
                rep1 = rep0;
                len = len_decode(&len_c, &r, ps) + MATCH_MIN;
                state = st_match(state);
                uint32_t slot = bt_decode(&r, pos_slot_d[len_to_pos_state(len)], POS_SLOT_BITS);
                if (slot >= START_POS_MODEL) {
                    uint32_t nd = (slot >> 1) - 1;
                    rep0 = (2 | (slot & 1)) << nd;
                    if (slot < END_POS_MODEL)
                        rep0 += bt_rev_decode(&r, pos_d + (int32_t)(rep0 - slot - 1), nd);
                    else {
                        rep0 += rd_direct(&r, nd - ALIGN_BITS) << ALIGN_BITS;
                        rep0 += bt_rev_decode(&r, align_d, ALIGN_BITS);
                        if ((int32_t)rep0 < 0) {
                            if (r.eof) { rc = LZO_ERR_TRUNCATED; break; }
                            if (rep0 == 0xFFFFFFFFu) break; /* end marker */
                            rc = LZO_ERR_STREAM;
                            break;
                        }
                    }
                } else
                    rep0 = slot;
            }
            if (r.eof) { rc = LZO_ERR_TRUNCATED; break; }
            if ((uint64_t)rep0 >= now || rep0 >= dict_check) { rc = LZO_ERR_STREAM; break; }
            if (now + len > out_cap) { rc = LZO_ERR_OUT_CAP; break; }
            for (uint32_t i = 0; i < len; i++, now++) out[now] = out[now - 1 - rep0];
            prev = out[now - 1];
        }
    }
    free(lit);
    if (out_n) *out_n = (size_t)now;
    if (in_used) *in_used = r.p;
    return rc;
}
