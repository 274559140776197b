/*
 * mrz_oracle.h -- CPU restatement of the modern-rzip "rzip stage" hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product path: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link, load or call this code, and there only as the
 * checker / CPU baseline.  The product (libmrzgpu.so) never falls back to it.
 *
 * Every function cites the reference file:line (paths are relative to the
 * reference tree) whose behaviour it restates.  It is an independent
 * restatement of the algorithm, not a copy of the source.
 *
 * Parity status: PINNED against the six whole-file golden vectors of
 * SURVEY.md section 8c (sha256 of the complete `mrzip -n -L7` output + the
 * seven -vv statistics counters, produced by the reference binary during the
 * survey) -- see tests/test_oracle_golden.py.  The reference has no tests or
 * fixtures of its own.  BLAKE2b is additionally pinned against
 * oracle/_ref/libblake2b_ref.so (the reference's own common/blake2b.c compiled
 * in place) and RFC 7693 known answers.  LZ4 sizes are pinned against the
 * system liblz4.so.1 1.9.3 (lz4 is an un-vendored submodule of the reference).
 */
#ifndef MRZ_ORACLE_H
#define MRZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The seven counters the reference prints at -vv (src/rzip.c:1108-1115,
 * struct rzip_state.stats include/mrzip_private.h:407-415). */
typedef struct {
    int64_t inserts;
    int64_t literals;
    int64_t literal_bytes;
    int64_t matches;
    int64_t match_bytes;
    int64_t tag_hits;
    int64_t tag_misses;
} mrzo_stats;

/* growable byte buffer owned by the oracle (free with mrzo_buf_free) */
typedef struct {
    uint8_t *p;
    int64_t len;
    int64_t cap;
} mrzo_buf;

void mrzo_buf_free(mrzo_buf *b);

/* ---- primitives ------------------------------------------------------- */

/* glibc TYPE_3 random() at its default seed (1), restated so the oracle does
 * not depend on process-global libc state.  src/rzip.c:669-673 */
void mrzo_hash_index(int64_t H[256]);

/* CRC-32 (IEEE 802.3, reflected, init/xorout 0xffffffff) == libgcrypt
 * GCRY_MD_CRC32; update form: pass previous return value (start with 0). */
uint32_t mrzo_crc32(uint32_t crc, const uint8_t *p, int64_t n);

/* MD5 (RFC 1321) == libgcrypt GCRY_MD_MD5 (src/main.c:66, default -H 1) */
typedef struct {
    uint32_t a, b, c, d;
    uint64_t nbytes;
    uint8_t buf[64];
    int buflen;
} mrzo_md5;
void mrzo_md5_init(mrzo_md5 *m);
void mrzo_md5_update(mrzo_md5 *m, const uint8_t *p, int64_t n);
void mrzo_md5_final(mrzo_md5 *m, uint8_t out[16]);

/* BLAKE2b, unkeyed (common/blake2b.c:85-201) */
typedef struct {
    uint64_t h[8];
    uint64_t t[2];
    uint8_t buf[128];
    size_t buflen;
    size_t outlen;
} mrzo_blake2b;
void mrzo_blake2b_init(mrzo_blake2b *s, size_t outlen);
void mrzo_blake2b_update(mrzo_blake2b *s, const void *in, size_t n);
void mrzo_blake2b_final(mrzo_blake2b *s, uint8_t *out);

/* ---- the rzip matcher (src/rzip.c:230-667) ---------------------------- */

typedef struct mrzo_matcher mrzo_matcher;

/* level 1..9 -> levels[] row (src/rzip.c:65-73).  The matcher object keeps
 * what the reference keeps for the life of one rzip_fd call (hash table
 * allocation, hash_index, stats) plus the process-lifetime static
 * victim_round (src/rzip.c:259), exposed so callers can chain it. */
mrzo_matcher *mrzo_matcher_new(int level);
void mrzo_matcher_free(mrzo_matcher *m);
int64_t mrzo_matcher_get_victim_round(const mrzo_matcher *m);
void mrzo_matcher_set_victim_round(mrzo_matcher *m, int64_t v);
/* timing variant: extend matches one byte at a time like single_match_len (src/rzip.c:378) instead of eight -- the
 * same results, the reference's inner loop shape (bench.py reports both as cpu_baseline) */
void mrzo_matcher_set_bytewise(mrzo_matcher *m, int on);
const mrzo_stats *mrzo_matcher_stats(const mrzo_matcher *m);
/* table introspection (show_distrib, src/rzip.c:464-485) */
void mrzo_matcher_distrib(const mrzo_matcher *m, int64_t *total, int64_t *primary);
/* final state of the last chunk (for kernel-vs-oracle state parity tests) */
int64_t mrzo_matcher_min_mask(const mrzo_matcher *m);
int64_t mrzo_matcher_hash_count(const mrzo_matcher *m);
const void *mrzo_matcher_table(const mrzo_matcher *m, int64_t *nslots);

/* One chunk: hash_search (src/rzip.c:507-667).  Appends the control records
 * to s0 (ending with the 00 00 00 terminator and the 4 CRC bytes) and the
 * literal bytes to s1.  chunk_bytes = width of match distances. */
int mrzo_rzip_chunk(mrzo_matcher *m, const uint8_t *buf, int64_t n, int chunk_bytes, mrzo_buf *s0, mrzo_buf *s1,
                    uint32_t *crc_out);

/* chunk_bytes rule, src/rzip.c:1006-1008 */
int mrzo_chunk_bytes(int64_t chunk_size);

/* ---- whole-file `mrzip -n` (file in, file out; not stdin/stdout) ------- */

typedef struct {
    int level;          /* -L / -R, 1..9 (default 7) */
    int64_t window;     /* -w, units of 100 MiB; 0 = unset */
    int unlimited;      /* -U */
    int64_t ramsize;    /* -m in bytes (reference: units of 100 MiB) */
    int64_t page_size;  /* 4096 */
} mrzo_params;

/* src/mrzip.c:1053-1163 compress_file + src/rzip.c:807-1132 rzip_fd +
 * src/stream.c:771-938,1115-1305,1307-1349,1574-1648 (sink, -n only) +
 * src/mrzip.c:127-188 write_magic.  `out` receives the complete archive. */
int mrzo_compress(const mrzo_params *prm, const uint8_t *in, int64_t n, mrzo_buf *out, mrzo_stats *stats,
                  uint8_t md5_out[16]);

/* the same for input read from STDIN (src/rzip.c:700-732,915-1061; src/util.c:156-164): chunks of maxram bytes until
 * read() returns 0, eof flag on the chunk that saw it (an empty one when the length is a multiple of the chunk size),
 * stream block size from the first chunk, magic size field per src/mrzip.c:137-140.  to_stdout: output to STDOUT too. */
int mrzo_compress_stream(const mrzo_params *prm, const uint8_t *in, int64_t n, int to_stdout, mrzo_buf *out,
                         mrzo_stats *stats, uint8_t md5_out[16], int *nchunks_out);

/* Chunking / sizing rules alone (src/rzip.c:875-894, src/util.c:156-176,
 * src/stream.c:797-914 for -n).  Returns max_chunk; *stream_bufsize gets the
 * per-stream block size for a file of st_size bytes. */
int64_t mrzo_plan(const mrzo_params *prm, int64_t st_size, int64_t *stream_bufsize);

/* Frame pre-computed per-chunk streams into a complete archive exactly as the
 * reference's sink would (used to check the GPU path's streams end-to-end). */
typedef struct {
    int64_t chunk_size;
    const uint8_t *s0;
    int64_t s0_len;
    const uint8_t *s1;
    int64_t s1_len;
} mrzo_chunk_streams;
int mrzo_frame(const mrzo_params *prm, int64_t st_size, const mrzo_chunk_streams *chunks, int nchunks,
               const uint8_t md5[16], mrzo_buf *out);

/* Decoder: src/runzip.c:112-330 + src/stream.c:941-1080 (CTYPE_NONE blocks
 * only).  Verifies per-chunk CRC32 and trailing MD5; 0 on success. */
int mrzo_decompress(const uint8_t *mrz, int64_t n, mrzo_buf *out);

/* ---- LZ4 compressibility gate (src/stream.c:1685-1733) ---------------- */

/* size LZ4_compress_default(src, dst, n, n+1) would return (0 if it does not
 * fit), liblz4 1.9.3 algorithm (LZ4_compress_fast, acceleration 1). */
int mrzo_lz4_compressed_size(const uint8_t *src, int n, int dst_cap);
/* writes the actual LZ4 block as well (dst may be NULL => size only) */
int mrzo_lz4_compress(const uint8_t *src, int n, uint8_t *dst, int dst_cap);
int mrzo_lz4_compresses(const uint8_t *s_buf, int64_t s_len, int threshold);

/* ---- rs-mrzip encoder (rs-mrzip/reed-solomon.c:115-141,311-321; rs-mrzip.c:119-158) ---- */
void mrzo_rs_parity(const uint8_t data[223], uint8_t parity[32]);
int64_t mrzo_rs_encoded_size(int64_t n);
int mrzo_rs_encode(const uint8_t *in, int64_t n, uint8_t *out);
void mrzo_rs_tables(uint8_t exp_[256], uint8_t log_[256], uint8_t tal[256], uint8_t tal1[256], uint8_t gen_index[33]);

#ifdef __cplusplus
}
#endif
#endif
