/* tests/c/integration_snippet.c -- TEST: the code of INTEGRATION.md section 1, verbatim, inside a function, so that
 * gcc -fsyntax-only can check it against include/mrzgpu.h and the reference prototypes (tests/c/ref_decls.h). */
#include "ref_decls.h"
#include "mrzgpu.h"

/* BEGIN SNIPPET (kept identical to INTEGRATION.md section 1 by tests/test_abi.py) */
/* once per rzip_fd call, next to init_hash_indexes (src/rzip.c:901) */
static mrz_ctx *gpu;
static int64_t gpu_victim_round; /* == the `static i64 victim_round` of insert_hash (src/rzip.c:259) */

static void gpu_open(rzip_control *control) {
    if (mrz_open(&gpu, 0, control->rzip_compression_level, control->max_chunk)) fatal("no MI355X\n");
}

/* in rzip_chunk, instead of hash_search(control, st, pct_base, pct_multiple) (src/rzip.c:763-792) */
static void gpu_hash_search(rzip_control *control, struct rzip_state *st) {
    mrz_chunk_result r;
    int rc = mrz_rzip_chunk(gpu, control->sb.buf_low, st->chunk_size, MRZ_MEM_HOST, st->chunk_bytes,
                            &gpu_victim_round, &r);
    if (rc) fatal("mrz_rzip_chunk: %s\n", mrz_strerror(rc));
    uchar *s0 = malloc(r.s0_len);
    if (!s0 || mrz_fetch_streams(gpu, s0, NULL)) fatal("mrz_fetch_streams\n"); /* stream 1 is not needed: see below */

    /* Replay the records into the reference's own sink, exactly as put_literal / put_match would
       (src/rzip.c:179-227): a literal's header goes through write_stream (include/stream.h:41), its bytes through
       write_sbstream (src/rzip.c:197-211), which copies them out of the mmap'd input at their own offset and calls
       flush_buffer (include/stream.h:40) whenever a stream buffer fills -- so the "buffer full" flushes of the two
       streams interleave as in the reference and the block chain comes out byte-identical. */
    i64 pos = 0; /* input offset the records have accounted for */
    for (i64 i = 0; i < r.s0_len;) {
        int head = s0[i];
        i64 len = s0[i + 1] | (i64)s0[i + 2] << 8;
        if (!head) {
            write_stream(control, st->ss, 0, s0 + i, 3);
            i += 3;
            if (!len) { /* terminator + CRC (src/rzip.c:664-665) */
                write_stream(control, st->ss, 0, s0 + i, 4);
                break;
            }
            write_sbstream(control, st->ss, 1, pos, len);
        } else {
            write_stream(control, st->ss, 0, s0 + i, 3 + st->chunk_bytes);
            i += 3 + st->chunk_bytes;
        }
        pos += len;
    }
    free(s0);
    st->stats.inserts += r.stats.inserts; /* ... the seven counters printed at -vv, src/rzip.c:1108-1115 */
    st->stats.matches += r.stats.matches;
    st->stats.match_bytes += r.stats.match_bytes;
    st->stats.literals += r.stats.literals;
    st->stats.literal_bytes += r.stats.literal_bytes;
    st->stats.tag_hits += r.stats.tag_hits;
    st->stats.tag_misses += r.stats.tag_misses;
}
/* at the end of rzip_fd: mrz_close(gpu); */
/* END SNIPPET */

void integration_snippet_anchor(rzip_control *c, struct rzip_state *s) {
    gpu_open(c);
    gpu_hash_search(c, s);
    mrz_close(gpu);
}
