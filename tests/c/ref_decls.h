/* tests/c/ref_decls.h -- TEST ONLY: prototypes of the reference interfaces that INTEGRATION.md section 1 touches,
 * restated from include/stream.h:39-44, include/mrzip_private.h (rzip_control, rzip_state, stream_info: only the
 * fields the snippet names) and src/rzip.c:197 -- so that the snippet can be syntax-checked by gcc.  Nothing here is
 * compiled into the product, and nothing of the reference is built with it. */
#include <stdint.h>
#include <stdlib.h>
typedef int64_t i64;
typedef unsigned char uchar;
struct stream { uchar *buf; i64 buflen; };
struct stream_info { struct stream *s; i64 bufsize; };
struct sliding_buffer { uchar *buf_low; };
typedef struct rzip_control {
    int rzip_compression_level;
    i64 max_chunk;
    struct sliding_buffer sb;
} rzip_control;
struct rzip_state {
    void *ss;
    i64 chunk_size;
    int chunk_bytes;
    struct { i64 inserts, literals, literal_bytes, matches, match_bytes, tag_hits, tag_misses; } stats;
};
void write_stream(rzip_control *control, void *ss, int streamno, uchar *p, i64 len);                /* include/stream.h:41 */
void flush_buffer(rzip_control *control, struct stream_info *sinfo, int stream);                    /* include/stream.h:40 */
static inline void write_sbstream(rzip_control *control, void *ss, int stream, i64 p, i64 len) {     /* src/rzip.c:197 */
    (void)control; (void)ss; (void)stream; (void)p; (void)len;
}
void fatal(const char *fmt, ...);
