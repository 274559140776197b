/*
 * mrz_oracle.c -- CPU restatement of the modern-rzip rzip stage.
 * TEST INFRASTRUCTURE ONLY (see mrz_oracle.h).  Parity: pinned (header).
 *
 * Reference citations are file:line into the reference tree.
 */
#include "mrz_oracle.h"

#include <stdlib.h>
#include <string.h>

#define MIN_MATCH 31        /* MINIMUM_MATCH, src/rzip.c:49 */
#define GREAT_MATCH 1024    /* src/rzip.c:48 */
#define MIB (1048576LL)
#define STREAM_MIN (10 * MIB) /* STREAM_BUFSIZE, include/mrzip_private.h:27 */
#define CHUNK_UNIT (100 * MIB) /* CHUNK_MULTIPLE, src/rzip.c:46 */

/* ------------------------------------------------------------------ */
/* byte buffer                                                         */

static int buf_reserve(mrzo_buf *b, int64_t extra) {
    if (b->len + extra <= b->cap) return 0;
    int64_t nc = b->cap ? b->cap : 256;
    while (nc < b->len + extra) nc += nc / 2 + 64;
    uint8_t *np = (uint8_t *)realloc(b->p, (size_t)nc);
    if (!np) return -1;
    b->p = np;
    b->cap = nc;
    return 0;
}

static int buf_put(mrzo_buf *b, const void *src, int64_t n) {
    if (n <= 0) return 0;
    if (buf_reserve(b, n)) return -1;
    memcpy(b->p + b->len, src, (size_t)n);
    b->len += n;
    return 0;
}

static int buf_put_le(mrzo_buf *b, int64_t v, int width) {
    uint8_t tmp[8];
    for (int i = 0; i < 8; i++) tmp[i] = (uint8_t)((uint64_t)v >> (8 * i));
    return buf_put(b, tmp, width);
}

void mrzo_buf_free(mrzo_buf *b) {
    free(b->p);
    b->p = NULL;
    b->len = b->cap = 0;
}

/* ------------------------------------------------------------------ */
/* hash_index: (random() << 16) ^ random(), glibc TYPE_3, seed 1       */
/* src/rzip.c:669-673                                                  */

void mrzo_hash_index(int64_t H[256]) {
    /* glibc random_r.c: additive feedback generator x^31 + x^3 + 1.
     * srandom(1): r[0] = 1, r[i] = 16807 * r[i-1] mod (2^31 - 1) computed
     * with Schrage's split, then 310 outputs are discarded. */
    int32_t r[31];
    int32_t word = 1;
    r[0] = word;
    for (int i = 1; i < 31; i++) {
        long hi = word / 127773, lo = word % 127773;
        long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        word = (int32_t)w;
        r[i] = word;
    }
    int f = 3, b = 0;
    uint32_t out = 0;
    int produced = -310; /* discard phase */
    int64_t pend = 0;
    int idx = 0, half = 0;
    while (idx < 256) {
        uint32_t s = (uint32_t)r[f] + (uint32_t)r[b];
        r[f] = (int32_t)s;
        out = s >> 1;
        if (++f == 31) f = 0;
        if (++b == 31) b = 0;
        if (produced++ < 0) continue;
        if (!half) { /* left operand of ^ is evaluated first (gcc, clang) */
            pend = (int64_t)out << 16;
            half = 1;
        } else {
            H[idx++] = pend ^ (int64_t)out;
            half = 0;
        }
    }
}

/* ------------------------------------------------------------------ */
/* CRC-32                                                              */

static uint32_t crc_tab[8][256];
static int crc_ready;

static void crc_init(void) {
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        crc_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
        for (int k = 1; k < 8; k++) crc_tab[k][i] = crc_tab[0][crc_tab[k - 1][i] & 0xff] ^ (crc_tab[k - 1][i] >> 8);
    crc_ready = 1;
}

uint32_t mrzo_crc32(uint32_t crc, const uint8_t *p, int64_t n) {
    if (!crc_ready) crc_init();
    uint32_t c = ~crc;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = crc_tab[7][lo & 0xff] ^ crc_tab[6][(lo >> 8) & 0xff] ^ crc_tab[5][(lo >> 16) & 0xff] ^ crc_tab[4][lo >> 24] ^
            crc_tab[3][hi & 0xff] ^ crc_tab[2][(hi >> 8) & 0xff] ^ crc_tab[1][(hi >> 16) & 0xff] ^ crc_tab[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n-- > 0) c = crc_tab[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return ~c;
}

/* ------------------------------------------------------------------ */
/* MD5 (RFC 1321)                                                      */

static const uint32_t md5_k[64] = {
    0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af,
    0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa,
    0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8,
    0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
    0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97,
    0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1,
    0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391
};
static const uint8_t md5_s[64] = { 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,
                                   14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                                   4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21 };

static void md5_block(mrzo_md5 *m, const uint8_t *blk) {
    uint32_t w[16];
    for (int i = 0; i < 16; i++)
        w[i] = (uint32_t)blk[4 * i] | (uint32_t)blk[4 * i + 1] << 8 | (uint32_t)blk[4 * i + 2] << 16 |
               (uint32_t)blk[4 * i + 3] << 24;
    uint32_t a = m->a, b = m->b, c = m->c, d = m->d;
    for (int i = 0; i < 64; i++) {
        uint32_t f;
        int g;
        if (i < 16) {
            f = (b & c) | (~b & d);
            g = i;
        } else if (i < 32) {
            f = (d & b) | (~d & c);
            g = (5 * i + 1) & 15;
        } else if (i < 48) {
            f = b ^ c ^ d;
            g = (3 * i + 5) & 15;
        } else {
            f = c ^ (b | ~d);
            g = (7 * i) & 15;
        }
        uint32_t x = a + f + md5_k[i] + w[g];
        a = d;
        d = c;
        c = b;
        b = b + ((x << md5_s[i]) | (x >> (32 - md5_s[i])));
    }
    m->a += a;
    m->b += b;
    m->c += c;
    m->d += d;
}

void mrzo_md5_init(mrzo_md5 *m) {
    m->a = 0x67452301;
    m->b = 0xefcdab89;
    m->c = 0x98badcfe;
    m->d = 0x10325476;
    m->nbytes = 0;
    m->buflen = 0;
}

void mrzo_md5_update(mrzo_md5 *m, const uint8_t *p, int64_t n) {
    m->nbytes += (uint64_t)n;
    if (m->buflen) {
        int take = 64 - m->buflen;
        if (take > n) take = (int)n;
        memcpy(m->buf + m->buflen, p, (size_t)take);
        m->buflen += take;
        p += take;
        n -= take;
        if (m->buflen == 64) {
            md5_block(m, m->buf);
            m->buflen = 0;
        }
    }
    while (n >= 64) {
        md5_block(m, p);
        p += 64;
        n -= 64;
    }
    if (n > 0) {
        memcpy(m->buf, p, (size_t)n);
        m->buflen = (int)n;
    }
}

void mrzo_md5_final(mrzo_md5 *m, uint8_t out[16]) {
    uint64_t bits = m->nbytes * 8;
    uint8_t pad[72] = { 0x80 };
    int padlen = (m->buflen < 56) ? 56 - m->buflen : 120 - m->buflen;
    uint64_t keep = m->nbytes;
    mrzo_md5_update(m, pad, padlen);
    uint8_t lenb[8];
    for (int i = 0; i < 8; i++) lenb[i] = (uint8_t)(bits >> (8 * i));
    mrzo_md5_update(m, lenb, 8);
    m->nbytes = keep;
    uint32_t v[4] = { m->a, m->b, m->c, m->d };
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 4; k++) out[4 * i + k] = (uint8_t)(v[i] >> (8 * k));
}

/* ------------------------------------------------------------------ */
/* BLAKE2b (RFC 7693; common/blake2b.c:85-201)                          */

static const uint64_t b2_iv[8] = { 0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                   0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                   0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL };
/* message schedule; rounds 10 and 11 reuse rows 0 and 1 */
static const uint8_t b2_sigma[10][16] = {
    { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }, { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 },
    { 11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4 }, { 7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8 },
    { 9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13 }, { 2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9 },
    { 12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11 }, { 13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10 },
    { 6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5 }, { 10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0 }
};

static inline uint64_t ror64(uint64_t x, int c) { return (x >> c) | (x << (64 - c)); }

static void b2_compress(mrzo_blake2b *s, const uint8_t *blk, int last) {
    uint64_t m[16], v[16];
    for (int i = 0; i < 16; i++) {
        uint64_t w = 0;
        for (int k = 7; k >= 0; k--) w = (w << 8) | blk[8 * i + k];
        m[i] = w;
    }
    for (int i = 0; i < 8; i++) {
        v[i] = s->h[i];
        v[i + 8] = b2_iv[i];
    }
    v[12] ^= s->t[0];
    v[13] ^= s->t[1];
    if (last) v[14] = ~v[14];
    static const uint8_t quad[8][4] = { { 0, 4, 8, 12 }, { 1, 5, 9, 13 }, { 2, 6, 10, 14 }, { 3, 7, 11, 15 },
                                        { 0, 5, 10, 15 }, { 1, 6, 11, 12 }, { 2, 7, 8, 13 }, { 3, 4, 9, 14 } };
    for (int r = 0; r < 12; r++) {
        const uint8_t *sg = b2_sigma[r % 10];
        for (int g = 0; g < 8; g++) {
            uint64_t *a = &v[quad[g][0]], *b = &v[quad[g][1]], *c = &v[quad[g][2]], *d = &v[quad[g][3]];
            *a += *b + m[sg[2 * g]];
            *d = ror64(*d ^ *a, 32);
            *c += *d;
            *b = ror64(*b ^ *c, 24);
            *a += *b + m[sg[2 * g + 1]];
            *d = ror64(*d ^ *a, 16);
            *c += *d;
            *b = ror64(*b ^ *c, 63);
        }
    }
    for (int i = 0; i < 8; i++) s->h[i] ^= v[i] ^ v[i + 8];
}

void mrzo_blake2b_init(mrzo_blake2b *s, size_t outlen) {
    memset(s, 0, sizeof(*s));
    for (int i = 0; i < 8; i++) s->h[i] = b2_iv[i];
    s->outlen = (uint8_t)outlen;
    s->h[0] ^= 0x01010000ULL ^ (uint64_t)s->outlen; /* common/blake2b.c:89 */
}

static void b2_count(mrzo_blake2b *s, uint64_t inc) {
    s->t[0] += inc;
    if (s->t[0] < inc) s->t[1]++;
}

void mrzo_blake2b_update(mrzo_blake2b *s, const void *in, size_t n) {
    const uint8_t *p = (const uint8_t *)in;
    /* a full buffer is only compressed once more input is known to follow
     * (the final block must be flagged), common/blake2b.c:161-184 */
    while (n > 0) {
        if (s->buflen == 128) {
            b2_count(s, 128);
            b2_compress(s, s->buf, 0);
            s->buflen = 0;
        }
        size_t take = 128 - s->buflen;
        if (take > n) take = n;
        memcpy(s->buf + s->buflen, p, take);
        s->buflen += take;
        p += take;
        n -= take;
    }
}

void mrzo_blake2b_final(mrzo_blake2b *s, uint8_t *out) {
    b2_count(s, s->buflen);
    memset(s->buf + s->buflen, 0, 128 - s->buflen);
    b2_compress(s, s->buf, 1);
    uint8_t full[64];
    for (int i = 0; i < 8; i++)
        for (int k = 0; k < 8; k++) full[8 * i + k] = (uint8_t)(s->h[i] >> (8 * k));
    memcpy(out, full, s->outlen);
}

/* ------------------------------------------------------------------ */
/* matcher                                                              */

typedef struct {
    int64_t offset;
    int64_t t;
} slot_t; /* struct hash_entry, src/rzip.c:59-62 */

struct mrzo_matcher {
    int level;
    unsigned mb_used, initial_freq, max_chain;
    int64_t H[256];
    slot_t *tab;
    int bits;
    int64_t nslots, slot_mask;
    int64_t count, limit;
    int64_t min_mask, clean_ptr, last_match;
    int64_t victim_round;
    mrzo_stats st;
    int bytewise; /* compare one byte at a time, as single_match_len does (src/rzip.c:378), not eight */
};

/* levels[], src/rzip.c:65-73 */
static const unsigned level_rows[10][3] = { { 1, 4, 1 },  { 2, 4, 2 },  { 4, 4, 2 },  { 8, 4, 2 },   { 16, 4, 3 },
                                            { 32, 4, 4 }, { 32, 2, 6 }, { 64, 1, 16 }, { 64, 1, 32 }, { 64, 1, 128 } };

mrzo_matcher *mrzo_matcher_new(int level) {
    if (level < 0 || level > 9) return NULL;
    mrzo_matcher *m = (mrzo_matcher *)calloc(1, sizeof(*m));
    if (!m) return NULL;
    m->level = level;
    m->mb_used = level_rows[level][0];
    m->initial_freq = level_rows[level][1];
    m->max_chain = level_rows[level][2];
    mrzo_hash_index(m->H);
    /* table geometry, src/rzip.c:521-530 */
    int64_t want = (int64_t)m->mb_used * (MIB / (int64_t)sizeof(slot_t));
    for (m->bits = 0; (1LL << m->bits) < want; m->bits++) {
    }
    m->nslots = 1LL << m->bits;
    m->slot_mask = m->nslots - 1;
    m->limit = m->nslots / 3 * 2;
    m->tab = (slot_t *)calloc((size_t)m->nslots, sizeof(slot_t));
    if (!m->tab) {
        free(m);
        return NULL;
    }
    return m;
}

void mrzo_matcher_free(mrzo_matcher *m) {
    if (!m) return;
    free(m->tab);
    free(m);
}

int64_t mrzo_matcher_get_victim_round(const mrzo_matcher *m) { return m->victim_round; }
void mrzo_matcher_set_victim_round(mrzo_matcher *m, int64_t v) { m->victim_round = v; }
void mrzo_matcher_set_bytewise(mrzo_matcher *m, int on) { m->bytewise = on ? 1 : 0; }
const mrzo_stats *mrzo_matcher_stats(const mrzo_matcher *m) { return &m->st; }
int64_t mrzo_matcher_min_mask(const mrzo_matcher *m) { return m->min_mask; }
int64_t mrzo_matcher_hash_count(const mrzo_matcher *m) { return m->count; }
const void *mrzo_matcher_table(const mrzo_matcher *m, int64_t *nslots) {
    if (nslots) *nslots = m->nslots;
    return m->tab;
}

void mrzo_matcher_distrib(const mrzo_matcher *m, int64_t *total, int64_t *primary) {
    int64_t tot = 0, pri = 0;
    for (int64_t i = 0; i < m->nslots; i++) {
        if (!(m->tab[i].offset | m->tab[i].t)) continue;
        tot++;
        if ((m->tab[i].t & m->slot_mask) == i) pri++;
    }
    *total = tot;
    *primary = pri;
}

static inline int slot_empty(const slot_t *s) { return !(s->offset | s->t); } /* src/rzip.c:230 */

/* number of trailing one bits + 1, the quantity lesser_bitness compares
 * (ffsll of the complement), src/rzip.c:248-252 */
static inline int ones_rank(int64_t t) { return __builtin_ffsll(~t); }

/* insert_hash, src/rzip.c:256-301.  The reference recurses when the new tag
 * outranks an occupant (the occupant is re-inserted first, then its slot is
 * taken).  All probe walks of one such cascade run before any slot is written
 * and the writes then land innermost-first, so the cascade is restated as
 * walk/collect followed by a reverse write-back. */
static void table_insert(mrzo_matcher *m, int64_t t, int64_t offset) {
    struct {
        int64_t h, t, off;
    } pend[64];
    int np = 0;
    const int64_t better = (m->min_mask << 1) | 1; /* minimum_bitness, :239-244 */
    for (;;) {
        int64_t h = t & m->slot_mask, victim_h = 0, round = 0;
        int displaced = 0;
        slot_t *s = &m->tab[h];
        while (!slot_empty(s)) {
            if ((s->t & better) != better) { /* due for culling: overwrite, :267-270 */
                m->count--;
                break;
            }
            if (ones_rank(s->t) < ones_rank(t)) { /* :275-278 */
                displaced = 1;
                break;
            }
            if (s->t == t) { /* :282-292 */
                if (round == m->victim_round) victim_h = h;
                if (++round == (int64_t)m->max_chain) {
                    h = victim_h;
                    m->count--;
                    if (++m->victim_round == (int64_t)m->max_chain) m->victim_round = 0;
                    break;
                }
            }
            h = (h + 1) & m->slot_mask;
            s = &m->tab[h];
        }
        pend[np].h = h;
        pend[np].t = t;
        pend[np].off = offset;
        np++;
        if (!displaced) break;
        t = m->tab[h].t;
        offset = m->tab[h].offset;
    }
    while (np-- > 0) {
        m->tab[pend[np].h].t = pend[np].t;
        m->tab[pend[np].h].offset = pend[np].off;
    }
}

/* clean_one_from_hash, src/rzip.c:305-328: returns the new insert mask */
static int64_t table_cull_one(mrzo_matcher *m) {
    for (;;) {
        const int64_t better = (m->min_mask << 1) | 1;
        for (; m->clean_ptr < m->nslots; m->clean_ptr++) {
            slot_t *s = &m->tab[m->clean_ptr];
            if (slot_empty(s)) continue;
            if ((s->t & better) != better) {
                s->offset = 0;
                s->t = 0;
                m->count--;
                return better;
            }
        }
        m->min_mask = better;
        m->clean_ptr = 0;
    }
}

/* single_match_len, src/rzip.c:372-397 */
static int64_t extend_match(const uint8_t *buf, int64_t p0, int64_t op, int64_t end, int64_t last_match, int64_t *rev,
                            int bytewise) {
    if (op >= p0) return 0;
    int64_t p = p0, q = op;
    while (!bytewise && p + 8 <= end) {
        uint64_t a, b;
        memcpy(&a, buf + p, 8);
        memcpy(&b, buf + q, 8);
        if (a != b) {
            int k = __builtin_ctzll(a ^ b) >> 3;
            p += k;
            q += k;
            goto fwd_done;
        }
        p += 8;
        q += 8;
    }
    while (p < end && buf[p] == buf[q]) {
        p++;
        q++;
    }
fwd_done:;
    int64_t len = p - p0;
    p = p0;
    q = op;
    int64_t floor = last_match > 0 ? last_match : 0;
    while (p > floor && q > 0 && buf[q - 1] == buf[p - 1]) {
        q--;
        p--;
    }
    *rev = p0 - p;
    len += *rev;
    return len < MIN_MATCH ? 0 : len;
}

/* find_best_match, src/rzip.c:426-462 */
static int64_t table_lookup(mrzo_matcher *m, const uint8_t *buf, int64_t t, int64_t p, int64_t end, int64_t *offset,
                            int64_t *reverse) {
    int64_t best = 0;
    *reverse = 0;
    int64_t h = t & m->slot_mask;
    const slot_t *s = &m->tab[h];
    while (!slot_empty(s)) {
        if (s->t == t) {
            int64_t rev = 0;
            int64_t ml = extend_match(buf, p, s->offset, end, m->last_match, &rev, m->bytewise);
            if (ml) {
                if (ml > best) {
                    best = ml;
                    *offset = s->offset - rev;
                    *reverse = rev;
                }
                m->st.tag_hits++;
            } else
                m->st.tag_misses++;
        }
        h = (h + 1) & m->slot_mask;
        s = &m->tab[h];
    }
    return best;
}

/* put_literal, src/rzip.c:213-227 (+ write_sbstream :197-211) */
static int emit_literal(mrzo_matcher *m, const uint8_t *buf, mrzo_buf *s0, mrzo_buf *s1, int64_t from, int64_t to) {
    do {
        int64_t len = to - from;
        if (len > 0xFFFF) len = 0xFFFF;
        m->st.literals++;
        m->st.literal_bytes += len;
        uint8_t hdr[3] = { 0, (uint8_t)len, (uint8_t)(len >> 8) };
        if (buf_put(s0, hdr, 3)) return -1;
        if (len && buf_put(s1, buf + from, len)) return -1;
        from += len;
    } while (to > from);
    return 0;
}

/* put_match, src/rzip.c:179-194 */
static int emit_match(mrzo_matcher *m, mrzo_buf *s0, int cb, int64_t p, int64_t offset, int64_t len) {
    do {
        int64_t n = len > 0xFFFF ? 0xFFFF : len;
        uint8_t hdr[3] = { 1, (uint8_t)n, (uint8_t)(n >> 8) };
        if (buf_put(s0, hdr, 3)) return -1;
        if (buf_put_le(s0, p - offset, cb)) return -1;
        m->st.matches++;
        m->st.match_bytes += n;
        len -= n;
        p += n;
        offset += n;
    } while (len);
    return 0;
}

int mrzo_chunk_bytes(int64_t chunk_size) {
    int bits = 8;
    while (chunk_size >> bits > 0) bits++;
    return bits / 8 + (bits % 8 ? 1 : 0);
}

/* hash_search, src/rzip.c:507-667 */
int mrzo_rzip_chunk(mrzo_matcher *m, const uint8_t *buf, int64_t n, int cb, mrzo_buf *s0, mrzo_buf *s1, uint32_t *crc_out) {
    memset(m->tab, 0, (size_t)m->nslots * sizeof(slot_t)); /* :518-519 */
    int64_t tag_mask = (1LL << m->initial_freq) - 1;
    m->min_mask = tag_mask;
    m->clean_ptr = 0;
    m->count = 0;
    m->last_match = 0;

    int64_t p = 0;
    const int64_t end = n - MIN_MATCH;
    int64_t cur_p = 0, cur_ofs = 0, cur_len = 0;
    int64_t t = 0;
    const int64_t *H = m->H;

    if (end > 0) /* single_full_tag, :348-358 */
        for (int i = 0; i < MIN_MATCH; i++) t ^= H[buf[i]];

    while (p < end) {
        p++;
        t ^= H[buf[p - 1]] ^ H[buf[p + MIN_MATCH - 1]]; /* single_next_tag, :330-337 */
        if ((t & m->min_mask) != m->min_mask) continue; /* :573 */

        int64_t reverse = 0, offset = 0;
        int64_t mlen = table_lookup(m, buf, t, p, end, &offset, &reverse);

        if ((t & tag_mask) == tag_mask) { /* :579-584 */
            m->st.inserts++;
            m->count++;
            table_insert(m, t, p);
            if (m->count > m->limit) tag_mask = table_cull_one(m);
        }

        if (mlen > cur_len) { /* :586-590 */
            cur_p = p - reverse;
            cur_len = mlen;
            cur_ofs = offset;
        }

        if ((cur_len >= GREAT_MATCH || p >= cur_p + MIN_MATCH) && cur_len >= MIN_MATCH) { /* :592-599 */
            if (m->last_match < cur_p && emit_literal(m, buf, s0, s1, m->last_match, cur_p)) return -1;
            if (emit_match(m, s0, cb, cur_p, cur_ofs, cur_len)) return -1;
            m->last_match = cur_p + cur_len;
            cur_p = p = m->last_match;
            cur_len = 0;
            t = 0;
            for (int i = 0; i < MIN_MATCH; i++) t ^= H[buf[p + i]];
        }
    }

    if (m->last_match < n && emit_literal(m, buf, s0, s1, m->last_match, n)) return -1; /* :619 */

    /* per-chunk CRC-32 over the chunk bytes; libgcrypt hands the digest out
     * most-significant byte first and the reference stores those four bytes
     * verbatim (:662-665). */
    uint32_t crc = mrzo_crc32(0, buf, n);
    if (emit_literal(m, buf, s0, s1, 0, 0)) return -1; /* terminator :664 */
    uint8_t cb4[4] = { (uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc };
    if (buf_put(s0, cb4, 4)) return -1;
    if (crc_out) *crc_out = crc;
    return 0;
}

/* ------------------------------------------------------------------ */
/* sizing rules                                                         */

static int64_t page_floor(int64_t v, int64_t page) { /* round_to_page, src/util.c:166-169 */
    v -= v % page;
    return v ? v : page;
}

static int64_t page_ceil(int64_t v, int64_t page) { /* round_up_page, src/util.c:171-176 */
    int64_t r = v % page;
    return r ? v + page - r : v;
}

int64_t mrzo_plan(const mrzo_params *prm, int64_t st_size, int64_t *stream_bufsize) {
    const int64_t page = prm->page_size ? prm->page_size : 4096;
    /* setup_ram, src/util.c:156-164 (file -> file) */
    int64_t usable = prm->ramsize / 3;
    /* src/rzip.c:881-888 */
    int64_t max_chunk;
    if (prm->unlimited)
        max_chunk = st_size;
    else if (prm->window)
        max_chunk = prm->window * CHUNK_UNIT;
    else
        max_chunk = prm->ramsize / 3 * 2;
    if (max_chunk < st_size) max_chunk = page_floor(max_chunk, page);
    if (stream_bufsize) {
        /* open_stream_out with NO_COMPRESS: testbufs 1, threads 1,
         * src/stream.c:797-805,878-881,913-914 */
        int64_t first_chunk = st_size < max_chunk ? st_size : max_chunk;
        int64_t chunk_limit = first_chunk < page ? page : first_chunk;
        int64_t limit = usable;
        if (st_size > 0 && st_size < limit)
            limit = st_size > STREAM_MIN ? st_size : STREAM_MIN;
        else if (limit > chunk_limit)
            limit = chunk_limit;
        *stream_bufsize = page_ceil(limit, page);
    }
    return max_chunk;
}

/* ------------------------------------------------------------------ */
/* the stream sink for -n: src/stream.c:1115-1305 (block writer),        */
/* :1307-1349 (flush), :1574-1590 (write_stream), :1623-1648 (close)     */

typedef struct {
    mrzo_buf *out;
    int64_t bufsize;
    int cb;
    int64_t initial_pos, cur_pos;
    int64_t last_head[2];
    int blocks; /* blocks written so far in this chunk */
    int eof;
    int64_t chunk_field;
    mrzo_buf sbuf[2];
} sink_t;

static void poke_le(mrzo_buf *b, int64_t at, int64_t v, int width) {
    for (int i = 0; i < width; i++) b->p[at + i] = (uint8_t)((uint64_t)v >> (8 * i));
}

static int sink_emit_block(sink_t *k, int sno) {
    mrzo_buf *o = k->out;
    const int wl = k->cb;
    if (!k->blocks++) { /* chunk header + per-stream heads, :1199-1244 */
        uint8_t two[2] = { (uint8_t)k->cb, (uint8_t)k->eof };
        if (buf_put(o, two, 2) || buf_put_le(o, k->chunk_field, wl)) return -1;
        k->initial_pos = o->len;
        for (int j = 0; j < 2; j++) {
            k->last_head[j] = k->cur_pos + 1 + 2 * wl;
            uint8_t ct = 3; /* CTYPE_NONE */
            if (buf_put(o, &ct, 1) || buf_put_le(o, 0, wl) || buf_put_le(o, 0, wl) || buf_put_le(o, 0, wl)) return -1;
            k->cur_pos += 1 + 3 * wl;
        }
    }
    /* link the previous head of this stream to the new block, :1249-1257 */
    poke_le(o, k->initial_pos + k->last_head[sno], k->cur_pos, wl);
    k->last_head[sno] = k->cur_pos + 1 + 2 * wl;
    uint8_t ct = 3;
    int64_t len = k->sbuf[sno].len;
    if (buf_put(o, &ct, 1) || buf_put_le(o, len, wl) || buf_put_le(o, len, wl) || buf_put_le(o, 0, wl)) return -1;
    k->cur_pos += 1 + 3 * wl;
    if (buf_put(o, k->sbuf[sno].p, len)) return -1;
    k->cur_pos += len;
    k->sbuf[sno].len = 0;
    return 0;
}

static int sink_write(sink_t *k, int sno, const uint8_t *p, int64_t n) {
    while (n) {
        int64_t room = k->bufsize - k->sbuf[sno].len;
        int64_t take = room < n ? room : n;
        if (buf_put(&k->sbuf[sno], p, take)) return -1;
        p += take;
        n -= take;
        if (k->sbuf[sno].len == k->bufsize && sink_emit_block(k, sno)) return -1;
    }
    return 0;
}

/* Replays one chunk's record stream through the sink so that "buffer full"
 * flushes of the two streams interleave exactly as in the reference, where
 * stream-1 bytes of a literal are appended right after its 3-byte header. */
static int sink_chunk(sink_t *k, int64_t chunk_size, int cb, int eof, int64_t page, const uint8_t *s0, int64_t n0,
                      const uint8_t *s1, int64_t n1) {
    k->cb = cb;
    k->eof = eof;
    k->chunk_field = chunk_size < page ? page : chunk_size; /* src/stream.c:779-780 */
    k->cur_pos = 0;
    k->blocks = 0;
    k->sbuf[0].len = k->sbuf[1].len = 0;
    int64_t i = 0, j = 0;
    while (i < n0) {
        if (i + 3 > n0) return -2;
        int head = s0[i];
        int64_t len = s0[i + 1] | (int64_t)s0[i + 2] << 8;
        if (head == 0) {
            if (sink_write(k, 0, s0 + i, 3)) return -1;
            i += 3;
            if (len == 0) { /* terminator: CRC follows */
                if (i + 4 != n0) return -2;
                if (sink_write(k, 0, s0 + i, 4)) return -1;
                i += 4;
                break;
            }
            if (j + len > n1) return -2;
            if (sink_write(k, 1, s1 + j, len)) return -1;
            j += len;
        } else {
            if (i + 3 + cb > n0) return -2;
            if (sink_write(k, 0, s0 + i, 3 + cb)) return -1;
            i += 3 + cb;
        }
    }
    if (j != n1) return -2;
    /* close_stream_out: both streams are flushed even when empty */
    if (sink_emit_block(k, 0) || sink_emit_block(k, 1)) return -1;
    return 0;
}

static void fill_magic(uint8_t *mg, const mrzo_params *prm, int64_t st_size) { /* write_magic, src/mrzip.c:127-188 */
    memset(mg, 0, 20);
    mg[0] = 'M';
    mg[1] = 'R';
    mg[2] = 'Z';
    mg[3] = 'I';
    mg[4] = 0;
    mg[5] = 9;
    for (int i = 0; i < 8; i++) mg[6 + i] = (uint8_t)((uint64_t)st_size >> (8 * i));
    mg[14] = 1; /* MD5 */
    mg[18] = (uint8_t)((prm->level << 4) + prm->level);
}

int mrzo_frame(const mrzo_params *prm, int64_t st_size, const mrzo_chunk_streams *chunks, int nchunks,
               const uint8_t md5[16], mrzo_buf *out) {
    const int64_t page = prm->page_size ? prm->page_size : 4096;
    sink_t k;
    memset(&k, 0, sizeof(k));
    k.out = out;
    mrzo_plan(prm, st_size, &k.bufsize);
    uint8_t mg[20] = { 0 };
    if (buf_put(out, mg, 20)) return -1;
    int rc = 0;
    for (int c = 0; c < nchunks && !rc; c++) {
        int cb = mrzo_chunk_bytes(chunks[c].chunk_size);
        rc = sink_chunk(&k, chunks[c].chunk_size, cb, c == nchunks - 1, page, chunks[c].s0, chunks[c].s0_len, chunks[c].s1,
                        chunks[c].s1_len);
    }
    mrzo_buf_free(&k.sbuf[0]);
    mrzo_buf_free(&k.sbuf[1]);
    if (rc) return rc;
    if (buf_put(out, md5, 16)) return -1;
    fill_magic(out->p, prm, st_size);
    return 0;
}

int mrzo_compress(const mrzo_params *prm, const uint8_t *in, int64_t n, mrzo_buf *out, mrzo_stats *stats,
                  uint8_t md5_out[16]) {
    mrzo_matcher *m = mrzo_matcher_new(prm->level);
    if (!m) return -1;
    int64_t max_chunk = mrzo_plan(prm, n, NULL);
    int cap = 8, nch = 0;
    mrzo_chunk_streams *cs = (mrzo_chunk_streams *)calloc((size_t)cap, sizeof(*cs));
    mrzo_buf *bufs = (mrzo_buf *)calloc((size_t)cap * 2, sizeof(*bufs));
    int rc = 0;
    int64_t left = n, pos = 0;
    int pass = 0;
    while (!pass || left > 0) { /* chunk loop, src/rzip.c:915-1061 */
        int64_t csz = max_chunk < left ? max_chunk : left;
        if (nch == cap) {
            cap *= 2;
            cs = (mrzo_chunk_streams *)realloc(cs, (size_t)cap * sizeof(*cs));
            bufs = (mrzo_buf *)realloc(bufs, (size_t)cap * 2 * sizeof(*bufs));
            memset(bufs + nch * 2, 0, (size_t)(cap - nch) * 2 * sizeof(*bufs));
        }
        rc = mrzo_rzip_chunk(m, in + pos, csz, mrzo_chunk_bytes(csz), &bufs[2 * nch], &bufs[2 * nch + 1], NULL);
        if (rc) break;
        cs[nch].chunk_size = csz;
        nch++;
        pos += csz;
        left -= csz;
        pass++;
    }
    uint8_t md5[16];
    if (!rc) {
        for (int c = 0; c < nch; c++) {
            cs[c].s0 = bufs[2 * c].p;
            cs[c].s0_len = bufs[2 * c].len;
            cs[c].s1 = bufs[2 * c + 1].p;
            cs[c].s1_len = bufs[2 * c + 1].len;
        }
        mrzo_md5 h;
        mrzo_md5_init(&h);
        mrzo_md5_update(&h, in, n);
        mrzo_md5_final(&h, md5);
        if (md5_out) memcpy(md5_out, md5, 16);
        rc = mrzo_frame(prm, n, cs, nch, md5, out);
    }
    if (stats) *stats = m->st;
    for (int c = 0; c < 2 * cap; c++) mrzo_buf_free(&bufs[c]);
    free(bufs);
    free(cs);
    mrzo_matcher_free(m);
    return rc;
}

/* `mrzip -n` reading STDIN (src/rzip.c:700-732 mmap_stdin, :915-1061 the chunk loop with STDIN set, src/util.c:156-164
 * setup_ram).  What differs from a file: the size is not known, so every chunk is max_mmap = min(page-rounded maxram,
 * max_chunk) bytes (maxram = ramsize / 3, or ramsize / 6 when the output goes to STDOUT too) until read() returns 0;
 * the chunk in which that happens is shrunk and carries the eof flag -- when the input length is a multiple of the
 * chunk size that is one more, EMPTY chunk; the stream block size is fixed at the first open_stream_out from the bytes
 * read so far (src/stream.c:797-914 with control->st_size = first chunk); and the size field of the magic header is the
 * total only if it is known when the header is written: at the end for a file (src/mrzip.c:1132), but at the first
 * block of the first chunk for STDOUT (src/stream.c:1202-1205, write_magic :137-140: only `if (control->eof)`).
 * `in`/`n` stand for what read() delivers.  Parity of this mode is pinned by reading only: the survey's golden vectors
 * are file -> file runs. */
int mrzo_compress_stream(const mrzo_params *prm, const uint8_t *in, int64_t n, int to_stdout, mrzo_buf *out,
                         mrzo_stats *stats, uint8_t md5_out[16], int *nchunks_out) {
    const int64_t page = prm->page_size ? prm->page_size : 4096;
    if (prm->unlimited) return -3; /* -U takes the window from the file size: 0 for STDIN */
    const int64_t maxram = prm->ramsize / (to_stdout ? 6 : 3);
    int64_t max_mmap = page_floor(maxram, page);
    const int64_t max_chunk = prm->window ? prm->window * CHUNK_UNIT : prm->ramsize / 3 * 2;
    if (max_mmap > max_chunk) max_mmap = max_chunk;
    mrzo_matcher *m = mrzo_matcher_new(prm->level);
    if (!m) return -1;
    sink_t k;
    memset(&k, 0, sizeof(k));
    k.out = out;
    uint8_t mg[20] = { 0 };
    int rc = buf_put(out, mg, 20) ? -1 : 0;
    int64_t pos = 0, st_size = 0;
    int eof = 0, nch = 0;
    int64_t magic_size = 0;
    mrzo_buf s0 = { 0, 0, 0 }, s1 = { 0, 0, 0 };
    while (!rc && !eof) {
        /* mmap_stdin: fill the chunk; a read that returns 0 ends the input */
        int64_t csz = max_mmap;
        if (n - pos < csz) { /* the read after the last byte returns 0 */
            csz = n - pos;
            eof = 1;
        }
        st_size += csz;
        if (!nch) { /* first open_stream_out: src/stream.c:803-914 with NO_COMPRESS, one thread */
            const int64_t chunk_limit = csz < page ? page : csz;
            int64_t limit = maxram;
            if (st_size > 0 && st_size < limit)
                limit = st_size > STREAM_MIN ? st_size : STREAM_MIN;
            else if (limit > chunk_limit)
                limit = chunk_limit;
            k.bufsize = page_ceil(limit, page);
            if (to_stdout && eof) magic_size = st_size; /* write_magic at the first block: size known only at eof */
        }
        s0.len = s1.len = 0;
        const int cb = mrzo_chunk_bytes(csz);
        rc = mrzo_rzip_chunk(m, in + pos, csz, cb, &s0, &s1, NULL);
        if (!rc) rc = sink_chunk(&k, csz, cb, eof, page, s0.p, s0.len, s1.p, s1.len);
        pos += csz;
        nch++;
    }
    mrzo_buf_free(&s0);
    mrzo_buf_free(&s1);
    mrzo_buf_free(&k.sbuf[0]);
    mrzo_buf_free(&k.sbuf[1]);
    if (!rc) {
        uint8_t md5[16];
        mrzo_md5 h;
        mrzo_md5_init(&h);
        mrzo_md5_update(&h, in, n);
        mrzo_md5_final(&h, md5);
        if (md5_out) memcpy(md5_out, md5, 16);
        if (buf_put(out, md5, 16)) rc = -1;
        if (!rc) fill_magic(out->p, prm, to_stdout ? magic_size : st_size);
    }
    if (stats) *stats = m->st;
    if (nchunks_out) *nchunks_out = nch;
    mrzo_matcher_free(m);
    return rc;
}

/* ------------------------------------------------------------------ */
/* decoder (CTYPE_NONE only)                                            */

static int64_t peek_le(const uint8_t *p, int width) {
    uint64_t v = 0;
    for (int i = width - 1; i >= 0; i--) v = (v << 8) | p[i];
    return (int64_t)v;
}

/* gathers one logical stream by following the block chain,
 * src/stream.c:1412-1571 (fill_buffer) */
static int gather_stream(const uint8_t *mrz, int64_t n, int64_t initial_pos, int64_t head_at, int cb, mrzo_buf *dst,
                         int64_t *end_max) {
    int64_t at = head_at;
    for (;;) {
        if (at + 1 + 3 * cb > n) return -3;
        int ctype = mrz[at];
        int64_t c_len = peek_le(mrz + at + 1, cb);
        int64_t u_len = peek_le(mrz + at + 1 + cb, cb);
        int64_t next = peek_le(mrz + at + 1 + 2 * cb, cb);
        if (ctype != 3 || c_len != u_len) return -4;
        int64_t pay = at + 1 + 3 * cb;
        if (pay + c_len > n) return -3;
        if (buf_put(dst, mrz + pay, c_len)) return -1;
        if (pay + c_len > *end_max) *end_max = pay + c_len;
        if (!next) return 0;
        at = initial_pos + next;
    }
}

int mrzo_decompress(const uint8_t *mrz, int64_t n, mrzo_buf *out) {
    if (n < 20 || memcmp(mrz, "MRZI", 4)) return -2;
    const int64_t expect = peek_le(mrz + 6, 8);
    const int hashed = mrz[14];
    int64_t at = 20 + mrz[19];
    int rc = 0;
    for (;;) { /* runzip_chunk, src/runzip.c:226-330 */
        if (at + 2 > n) return -3;
        int cb = mrz[at];
        int eof = mrz[at + 1];
        if (cb < 1 || cb > 8) return -4;
        at += 2 + cb;
        const int64_t initial_pos = at;
        int64_t end_max = initial_pos + 2 * (1 + 3 * cb);
        mrzo_buf s0 = { 0 }, s1 = { 0 };
        rc = gather_stream(mrz, n, initial_pos, initial_pos, cb, &s0, &end_max);
        if (!rc) rc = gather_stream(mrz, n, initial_pos, initial_pos + 1 + 3 * cb, cb, &s1, &end_max);
        const int64_t chunk_start = out->len;
        int64_t i = 0, j = 0;
        int done = 0;
        while (!rc && !done) {
            if (i + 3 > s0.len) {
                rc = -5;
                break;
            }
            int head = s0.p[i];
            int64_t len = peek_le(s0.p + i + 1, 2);
            i += 3;
            if (!head && !len) {
                done = 1;
            } else if (!head) { /* unzip_literal, :120-157 */
                if (j + len > s1.len)
                    rc = -5;
                else if (buf_put(out, s1.p + j, len))
                    rc = -1;
                j += len;
            } else { /* unzip_match, :159-207 */
                if (i + cb > s0.len) {
                    rc = -5;
                    break;
                }
                int64_t dist = peek_le(s0.p + i, cb);
                i += cb;
                if (dist < 1 || dist > out->len || buf_reserve(out, len)) {
                    rc = -5;
                    break;
                }
                /* the first min(len, dist) history bytes are replicated */
                int64_t span = len < dist ? len : dist;
                int64_t from = out->len - dist;
                for (int64_t w = 0; w < len; w++) out->p[out->len + w] = out->p[from + (w % span)];
                out->len += len;
            }
        }
        if (!rc) {
            if (i + 4 > s0.len)
                rc = -5;
            else {
                uint32_t want = (uint32_t)s0.p[i] << 24 | (uint32_t)s0.p[i + 1] << 16 | (uint32_t)s0.p[i + 2] << 8 | s0.p[i + 3];
                if (want != mrzo_crc32(0, out->p + chunk_start, out->len - chunk_start)) rc = -6;
            }
        }
        mrzo_buf_free(&s0);
        mrzo_buf_free(&s1);
        if (rc) return rc;
        at = end_max;
        if (eof) break;
    }
    if (out->len != expect) return -7;
    if (hashed == 1) {
        if (at + 16 > n) return -3;
        mrzo_md5 h;
        uint8_t d[16];
        mrzo_md5_init(&h);
        mrzo_md5_update(&h, out->p, out->len);
        mrzo_md5_final(&h, d);
        if (memcmp(d, mrz + at, 16)) return -8;
    }
    return 0;
}
