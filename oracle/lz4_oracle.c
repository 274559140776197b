/*
 * lz4_oracle.c -- restatement of what LZ4_compress_default() returns, and of
 * the reference's compressibility gate built on it.
 * TEST INFRASTRUCTURE ONLY (see mrz_oracle.h).
 *
 * lz4 is an un-vendored git submodule of the reference (vendor/lz4 is empty,
 * .gitmodules:1-11; pin unknowable), so the algorithm is restated from the
 * published LZ4 block format and the fast-compressor design of liblz4 1.9.3
 * (the version installed in this image as liblz4.so.1): single-probe hash
 * table, skip acceleration, one-position look-back, immediate re-test after a
 * match.  Parity is pinned byte-for-byte against liblz4.so.1 1.9.3 in
 * tests/test_lz4_oracle.py; the reference's call sites are
 * src/stream.c:1705 (gate) and :291,:298 (lz4 back-end).
 */
#include "mrz_oracle.h"

#include <stdlib.h>
#include <string.h>

enum {
    LZ_MINMATCH = 4,
    LZ_LASTLITERALS = 5,
    LZ_MFLIMIT = 12,
    LZ_MINLEN = 13,
    LZ_SKIP_TRIGGER = 6,
    LZ_MAXDIST = 65535,
    LZ_HASHLOG = 12,
    LZ_64K_LIMIT = 65536 + 11
};

static inline uint32_t rd32(const uint8_t *p) {
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}
static inline uint64_t rd64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}

/* inputs shorter than 64 KiB use 16-bit table cells, a 13-bit 4-byte hash and
 * no distance check; longer ones 32-bit cells and a 12-bit 5-byte hash */
static inline uint32_t hash_small(const uint8_t *p) { return (rd32(p) * 2654435761U) >> (32 - (LZ_HASHLOG + 1)); }
static inline uint32_t hash_large(const uint8_t *p) {
    return (uint32_t)(((rd64(p) << 24) * 889523592379ULL) >> (64 - LZ_HASHLOG));
}

static inline int count_equal(const uint8_t *a, const uint8_t *b, const uint8_t *alimit) {
    const uint8_t *a0 = a;
    while (a + 8 <= alimit) {
        uint64_t d = rd64(a) ^ rd64(b);
        if (d) return (int)(a - a0) + (__builtin_ctzll(d) >> 3);
        a += 8;
        b += 8;
    }
    while (a < alimit && *a == *b) {
        a++;
        b++;
    }
    return (int)(a - a0);
}

int mrzo_lz4_compress(const uint8_t *src, int n, uint8_t *dst, int cap) {
    if (n < 0 || (uint32_t)n > 0x7E000000u) return 0;
    if (n == 0) {
        if (cap <= 0) return 0;
        if (dst) dst[0] = 0;
        return 1;
    }
    const int bound = n + n / 255 + 16;
    const int limited = cap < bound;
    const int small = n < LZ_64K_LIMIT;
    uint32_t *tab = (uint32_t *)calloc(small ? 8192 : 4096, sizeof(uint32_t));
    if (!tab) return 0;
#define HASH(q) (small ? hash_small(q) : hash_large(q))
#define OUT(b)                              \
    do {                                    \
        if (dst) dst[op] = (uint8_t)(b);    \
        op++;                               \
    } while (0)

    const uint8_t *ip = src, *anchor = src;
    const uint8_t *const iend = src + n;
    const uint8_t *const mfl1 = iend - LZ_MFLIMIT + 1;
    const uint8_t *const mlimit = iend - LZ_LASTLITERALS;
    int64_t op = 0;
    const int64_t olimit = cap;
    int result = 0;

    if (n < LZ_MINLEN) goto tail;

    tab[HASH(ip)] = 0;
    ip++;
    uint32_t fwd_h = HASH(ip);

    for (;;) {
        const uint8_t *match;
        int64_t token;
        { /* search */
            const uint8_t *fwd = ip;
            int step = 1, nb = 1 << LZ_SKIP_TRIGGER;
            for (;;) {
                uint32_t h = fwd_h;
                uint32_t cur = (uint32_t)(fwd - src);
                uint32_t mi = tab[h];
                ip = fwd;
                fwd += step;
                step = nb++ >> LZ_SKIP_TRIGGER;
                if (fwd > mfl1) goto tail;
                match = src + mi;
                fwd_h = HASH(fwd);
                tab[h] = cur;
                if (!small && mi + LZ_MAXDIST < cur) continue;
                if (rd32(match) == rd32(ip)) break;
            }
        }
        while (ip > anchor && match > src && ip[-1] == match[-1]) { /* catch up */
            ip--;
            match--;
        }
        { /* literals */
            unsigned lit = (unsigned)(ip - anchor);
            token = op++;
            if (limited && op + lit + (2 + 1 + LZ_LASTLITERALS) + lit / 255 > olimit) goto out;
            if (lit >= 15) {
                if (dst) dst[token] = 15 << 4;
                int r = (int)lit - 15;
                for (; r >= 255; r -= 255) OUT(255);
                OUT(r);
            } else if (dst)
                dst[token] = (uint8_t)(lit << 4);
            if (dst) memcpy(dst + op, anchor, lit);
            op += lit;
        }
    next_match:
        OUT((ip - match) & 0xff);
        OUT(((ip - match) >> 8) & 0xff);
        {
            unsigned mc = (unsigned)count_equal(ip + LZ_MINMATCH, match + LZ_MINMATCH, mlimit);
            ip += (size_t)mc + LZ_MINMATCH;
            if (limited && op + (1 + LZ_LASTLITERALS) + (mc + 240) / 255 > olimit) goto out;
            if (mc >= 15) {
                if (dst) dst[token] += 15;
                mc -= 15;
                while (mc >= 255) {
                    OUT(255);
                    mc -= 255;
                }
                OUT(mc);
            } else if (dst)
                dst[token] += (uint8_t)mc;
        }
        anchor = ip;
        if (ip >= mfl1) break;
        tab[HASH(ip - 2)] = (uint32_t)(ip - 2 - src);
        { /* immediate re-test at the new position */
            uint32_t h = HASH(ip);
            uint32_t cur = (uint32_t)(ip - src);
            uint32_t mi = tab[h];
            match = src + mi;
            tab[h] = cur;
            if ((small || mi + LZ_MAXDIST >= cur) && rd32(match) == rd32(ip)) {
                token = op++;
                if (dst) dst[token] = 0;
                goto next_match;
            }
        }
        fwd_h = HASH(++ip);
    }

tail : {
    size_t last = (size_t)(iend - anchor);
    if (limited && op + (int64_t)last + 1 + (int64_t)((last + 255 - 15) / 255) > olimit) goto out;
    if (last >= 15) {
        size_t acc = last - 15;
        OUT(15 << 4);
        for (; acc >= 255; acc -= 255) OUT(255);
        OUT(acc);
    } else
        OUT(last << 4);
    if (dst) memcpy(dst + op, anchor, last);
    op += (int64_t)last;
    result = (int)op;
}
out:
    free(tab);
    return result;
#undef HASH
#undef OUT
}

int mrzo_lz4_compressed_size(const uint8_t *src, int n, int dst_cap) { return mrzo_lz4_compress(src, n, NULL, dst_cap); }

/* lz4_compresses, src/stream.c:1685-1733 */
int mrzo_lz4_compresses(const uint8_t *s_buf, int64_t s_len, int threshold) {
    int test_len = (int)s_len;
    int in_len = test_len < 10 * 1048576 ? test_len : 10 * 1048576;
    int buftest = in_len;
    double pct = 101;
    while (test_len > 0) {
        int r = mrzo_lz4_compressed_size(s_buf, in_len, in_len + 1);
        if (r > 0) {
            pct = 100 * ((double)r / (double)in_len);
            if (r < in_len * ((double)threshold / 100)) break;
        }
        test_len -= in_len;
        if (test_len > 0) {
            buftest += in_len;
            if (buftest < 10 * 1048576) buftest <<= 1;
            in_len = test_len < buftest ? test_len : buftest;
        }
    }
    return (int)(pct > threshold ? 0 : pct < 1 ? pct + 1 : pct);
}
