/*
 * mrzgpu.h -- C ABI of libmrzgpu.so: the MI355X (gfx950) implementation of the
 * modern-rzip "rzip stage" hot path.  Plain C, plain pointers and sizes; no
 * torch / C++ types cross this boundary.  Every entry point names the
 * reference interface (file:line in the reference tree) it replaces.
 *
 * Error model: every function returns 0 on success or a negative MRZ_E_* code
 * (the library never calls exit(); the reference's fatal() policy,
 * include/util.h:50-62, stays with the host program).  mrz_strerror() gives
 * text.  There is NO CPU fallback: without a usable HIP device mrz_open()
 * fails with MRZ_E_NODEVICE.
 *
 * Threading: one mrz_ctx per host thread / HIP stream.  A ctx owns its device
 * buffers (hash table, tag/bitmap scratch, event list, output streams) and one
 * HIP stream; calls on different ctxs may run concurrently.
 */
#ifndef MRZGPU_H
#define MRZGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRZ_ABI_VERSION 2

enum {
    MRZ_OK = 0,
    MRZ_E_ARG = -1,      /* bad argument */
    MRZ_E_NODEVICE = -2, /* no HIP device / HIP runtime failure at open */
    MRZ_E_NOMEM = -3,    /* device or host allocation failed */
    MRZ_E_HIP = -4,      /* HIP runtime error during a call (see mrz_last_hip_error) */
    MRZ_E_OVERFLOW = -5, /* internal capacity exceeded (should not happen) */
    MRZ_E_STATE = -6,    /* call order violated (e.g. fetch before a chunk ran) */
    MRZ_E_CORRUPT = -7,  /* runzip: invalid record stream / archive (the reference fatal()s "corrupt archive") */
    MRZ_E_UNSUPPORTED = -8 /* runzip: block type other than CTYPE_NONE (back-end codecs are host code) */
};

/* where a caller-supplied buffer lives */
enum { MRZ_MEM_HOST = 0, MRZ_MEM_DEVICE = 1 };

typedef struct mrz_ctx mrz_ctx;

/* The seven counters of struct rzip_state.stats
 * (include/mrzip_private.h:407-415, printed at src/rzip.c:1108-1115). */
typedef struct {
    int64_t inserts;
    int64_t literals;
    int64_t literal_bytes;
    int64_t matches;
    int64_t match_bytes;
    int64_t tag_hits;
    int64_t tag_misses;
} mrz_stats;

/* Result of one chunk (what hash_search leaves in the two streams). */
typedef struct {
    int64_t s0_len;      /* control stream bytes incl. 00 00 00 terminator + 4 CRC bytes */
    int64_t s1_len;      /* literal stream bytes */
    uint32_t crc32;      /* CRC-32 of the chunk (src/rzip.c:662) */
    uint32_t reserved;
    const uint8_t *d_s0; /* DEVICE pointers, valid until the next chunk / mrz_close */
    const uint8_t *d_s1;
    mrz_stats stats;     /* counters of THIS chunk (the reference accumulates them per file) */
    /* final matcher state, for state-parity tests and -vv style reporting */
    int64_t min_mask;    /* rzip_state.minimum_tag_mask at chunk end */
    int64_t hash_count;  /* rzip_state.hash_count at chunk end */
    int64_t n_events;    /* emitted matches before 0xFFFF splitting */
} mrz_chunk_result;

/* per-kernel device times of the last chunk, milliseconds (HIP events on the
 * ctx stream; filled only when profiling was enabled with mrz_set_profiling) */
typedef struct {
    float tagscan_ms;   /* sum over segments */
    float sequencer_ms; /* sum over segments */
    float encode_ms;
    float crc_ms;
    float total_ms;     /* first launch -> last kernel done */
    int32_t n_segments; /* sequencer launches (segments an emitted match has covered are not launched) */
    int32_t n_narrow;   /* ... of which ran on the narrow engine (mrz_seq_narrow.hip) */
} mrz_timings;

/* ---- context ----------------------------------------------------------- */

/* Replaces the per-file set-up half of rzip_fd (src/rzip.c:836-913): level ->
 * levels[] row (:65-73,896), hash_index (:669-673,901), hash table allocation
 * (:521-530).  max_chunk sizes the device scratch; a larger chunk later makes
 * the ctx grow.  device = HIP ordinal. */
int mrz_open(mrz_ctx **out, int device, int level, int64_t max_chunk);
void mrz_close(mrz_ctx *ctx);

int mrz_abi_version(void);
const char *mrz_strerror(int code);
/* last hipError_t seen by this ctx (0 = hipSuccess) and its text */
int mrz_last_hip_error(const mrz_ctx *ctx, const char **text);

/* The HIP stream the ctx launches on, as an opaque handle (hipStream_t), so a
 * host framework can order its own work or record events against it. */
void *mrz_stream(const mrz_ctx *ctx);
int mrz_synchronize(mrz_ctx *ctx);
int mrz_set_profiling(mrz_ctx *ctx, int enable);
/* Helper workgroups per sequencer launch (the compare farm; default: the GPU's CU count - 16, or the
 * MRZ_FARM_WGS environment variable).  Several ctxs (or processes) sharing one GPU should split the CUs
 * between them, e.g. 224 / number of streams; 0 = no helpers (everything on the sequencer's own CU);
 * negative = back to the default.  Independent streams then overlap on the device: measured 3.1x aggregate
 * for 4 streams.  (In one process this needs GPU_MAX_HW_QUEUES >= 2 x ctxs when the HIP runtime initialises;
 * the library sets 8 at load time unless the variable is already set.) */
int mrz_set_farm_helpers(mrz_ctx *ctx, int n);
int mrz_get_timings(const mrz_ctx *ctx, mrz_timings *out);

/* Progress of the chunk in flight (within-chunk hand-off to the back-end: the reference flushes a stream buffer the
 * moment it fills DURING hash_search, src/rzip.c:197-211 -> flush_buffer src/stream.c:1347).  mrz_rzip_chunk calls
 * `fn` on the calling thread every time a segment of the chunk has been sequenced and once more when the whole chunk
 * has: the first n_events matches of the chunk are final, and so is every byte before last_match (literal or
 * matched).  Inside `fn` mrz_fetch_events may be called; the device keeps working on the segments already queued.
 * A non-zero return aborts the chunk with MRZ_E_STATE.  fn = NULL switches it off. */
typedef struct {
    int64_t p;   /* position of the match in the chunk */
    int64_t ofs; /* position it copies from (distance = p - ofs) */
    int64_t len; /* before splitting into 0xFFFF pieces (put_match, src/rzip.c:179-194) */
} mrz_match;
typedef int (*mrz_progress_fn)(void *user, int64_t n_events, int64_t last_match, int chunk_done);
int mrz_set_progress(mrz_ctx *ctx, mrz_progress_fn fn, void *user);
/* copies matches [first, first + count) of the chunk in flight (or of the last chunk) to host memory */
int mrz_fetch_events(mrz_ctx *ctx, int64_t first, int64_t count, mrz_match *host_dst);

/* ---- one window over several GPUs (second row; BASELINE configs[3]) --------------------------
 * What shards inside ONE chunk is the front end: every rank owns a contiguous byte range of the window (plus a
 * 30-byte halo) and computes the tags of its positions (single_full_tag / next_tag, src/rzip.c:330-358) and the
 * candidate bitmap for the mask the matcher has reached (:573); the exact matcher itself stays on one rank, which
 * takes the segments' tags from their owners instead of scanning them.
 *
 * mrz_window_scan (any rank): tags and bitmap of positions [seg_start, seg_start + seg_len) of a chunk of chunk_n
 * bytes, given the rank's bytes [range_start, range_start + range_len) -- which must reach 48 bytes past the
 * segment's last position (30 for the tags, the rest because the kernel loads 16-byte pieces) or to the end of the
 * chunk; seg_len a multiple of 4096 unless the segment is the chunk's last.  min_mask: the matcher's minimum_tag_mask as last heard of
 * (an older, looser mask only makes the bitmap a superset, which the matcher re-checks); p_done: positions at or
 * before it need no tags.  tags_out: seg_len x int64, bitmap_out: ceil(seg_len / 64) x uint64, host memory.
 *
 * mrz_set_tag_provider (the matcher's rank): mrz_rzip_chunk then calls `fn` for every segment it is about to
 * sequence instead of scanning it; `fn` has to fill the two device buffers (seg_len x int64 tags, seg_len / 64
 * words of bitmap; e.g. hipMemcpyAsync on `stream`, or a blocking copy) and return 0.  fn = NULL: scan locally.
 * mrz_set_segment_positions: positions per segment launch (a multiple of 4096, at most the default of 16 Mi). */
int mrz_window_scan(mrz_ctx *ctx, const void *range_bytes, int64_t range_len, int where, int64_t range_start,
                    int64_t chunk_n, int64_t seg_start, int64_t seg_len, int64_t min_mask, int64_t p_done,
                    int64_t *tags_out, uint64_t *bitmap_out);
typedef int (*mrz_tag_provider_fn)(void *user, int64_t seg_index, int64_t seg_start, int64_t seg_len, int64_t min_mask,
                                   int64_t p_done, int64_t *d_tags, uint64_t *d_bitmap, void *stream);
int mrz_set_tag_provider(mrz_ctx *ctx, mrz_tag_provider_fn fn, void *user);
int mrz_set_segment_positions(mrz_ctx *ctx, int64_t positions);
/* host -> device copy on the ctx stream (what a tag provider without a HIP runtime of its own fills the buffers with) */
int mrz_copy_to_device(mrz_ctx *ctx, void *dst_device, const void *src_host, int64_t n);

/* ---- the rzip stage ---------------------------------------------------- */

/* Replaces rzip_chunk -> hash_search (src/rzip.c:763-792,507-667) for one
 * chunk: tag scan (:330-358), table look-up / insert / cull (:232-328,426-462),
 * lazy match selection and emission (:548-599), record encoding
 * (:163-227), per-chunk CRC-32 (:488-505,601-666).
 *   chunk, n        the chunk bytes (host or device memory per `where`)
 *   chunk_bytes     width of match distances (src/rzip.c:1006-1008); use
 *                   mrz_chunk_bytes(n)
 *   victim_round    in/out: the process-lifetime `static victim_round`
 *                   (src/rzip.c:259); pass 0 for the first chunk of a process
 *                   and chain the returned value into the next chunk.
 * The two output streams stay on the device (res->d_s0 / d_s1); copy them out
 * with mrz_fetch_streams.  Returns after all device work has completed. */
int mrz_rzip_chunk(mrz_ctx *ctx, const void *chunk, int64_t n, int where, int chunk_bytes, int64_t *victim_round,
                   mrz_chunk_result *res);

/* Copies the last chunk's streams to host memory (either may be NULL). */
int mrz_fetch_streams(mrz_ctx *ctx, uint8_t *s0_host, uint8_t *s1_host);

/* src/rzip.c:1006-1008 */
int mrz_chunk_bytes(int64_t chunk_size);

/* Debug / parity: copy the hash table (nslots * 16 bytes, struct hash_entry
 * layout {i64 offset; i64 tag}, src/rzip.c:59-62) of the last chunk to host. */
int64_t mrz_table_slots(const mrz_ctx *ctx);
int mrz_fetch_table(mrz_ctx *ctx, void *host_dst);

/* ---- CRC-32 alone (libgcrypt GCRY_MD_CRC32 as used at src/rzip.c:494,648) -- */
int mrz_crc32(mrz_ctx *ctx, const void *buf, int64_t n, int where, uint32_t *crc_out);

/* ---- LZ4 compressibility gate ------------------------------------------ */

/* Replaces `static int lz4_compresses(rzip_control*, uchar *s_buf, i64 s_len)`
 * (src/stream.c:116,1685-1733) for a BATCH of stream blocks: result[i] is
 * exactly what the reference returns for block i (0 = incompressible, else
 * 1..100 = percent), with control->threshold = `threshold`.
 * bufs[i]/lens[i]: the blocks (all host or all device per `where`). */
int mrz_lz4_compresses_batch(mrz_ctx *ctx, const void *const *bufs, const int64_t *lens, int count, int where,
                             int threshold, int *results);
/* single-block convenience with the reference's exact shape */
int mrz_lz4_compresses(mrz_ctx *ctx, const void *s_buf, int64_t s_len, int where, int threshold, int *result);
/* the size LZ4_compress_default(src, dst, n, n + 1) returns (0 = does not fit),
 * src/stream.c:1705 -- exposed for parity tests */
int mrz_lz4_sizes(mrz_ctx *ctx, const void *const *bufs, const int *lens, int count, int where, int *sizes);

/* ---- BLAKE2b (co-resident checksum kernel) ------------------------------ */

/* Streaming triple mirroring common/blake2b.h:47-49
 * (blake2b_init / blake2b_update / blake2b_final); the state lives on the
 * device and the kernel runs on a second low-priority stream of the ctx so it
 * overlaps the rzip kernels.  Unlike blake2b_update, the call returns BEFORE the bytes have been read: host input is
 * copied into a staging buffer first (reusable on return), DEVICE input must stay valid and unmodified until
 * mrz_blake2b_final has returned.  Device input is ordered after everything already queued on the ctx stream
 * (mrz_stream); producers on other streams have to be ordered by the caller. */
typedef struct mrz_blake2b mrz_blake2b;
int mrz_blake2b_init(mrz_ctx *ctx, mrz_blake2b **st, size_t outlen);
int mrz_blake2b_update(mrz_blake2b *st, const void *in, size_t inlen, int where);
int mrz_blake2b_final(mrz_blake2b *st, void *out_host, size_t outlen); /* frees st */

/* Many independent messages at once (ar-mrzip hashes every file separately,
 * ar-mrzip/ar-mrzip.cpp:139-171): digest i (outlen bytes) is written to
 * out_host + i*outlen. */
int mrz_blake2b_batch(mrz_ctx *ctx, const void *const *msgs, const int64_t *lens, int count, int where, size_t outlen,
                      uint8_t *out_host);

/* ---- rs-mrzip encoder (the "next" row after the rzip stage) ------------------ */

/* Replaces encode() of rs-mrzip (rs-mrzip/rs-mrzip.c:119-158): for every 223-byte row
 * the CCSDS RS(255,223) dual-basis parity (rse32, rs-mrzip/reed-solomon.c:115-141), bursts
 * of 8176 rows interleaved column-major (scatter, :311-321), then BLAKE2b-512 of the padded
 * rows and the 4-byte {k_i, k_j} trailer.  `out` receives exactly what `rs-mrzip` writes to
 * stdout for `n` bytes of stdin: mrz_rs_encoded_size(n) bytes. */
int64_t mrz_rs_encoded_size(int64_t n);
int mrz_rs_encode(mrz_ctx *ctx, const void *in, int64_t n, int where, void *out, int out_where, int64_t out_cap);

/* Replaces decode() of rs-mrzip (rs-mrzip/rs-mrzip.c:37-117): de-interleave every burst (gather,
 * rs-mrzip/reed-solomon.c:323-333), correct every 255-byte codeword (rsd32, :143-309: syndromes, Berlekamp-Massey,
 * Chien search, Forney; up to 16 byte errors per codeword), check the BLAKE2b-512 of the decoded rows against the
 * trailer and strip the zero padding ({k_i, k_j}).  `in`: what `rs-mrzip` (or mrz_rs_encode) wrote, possibly damaged;
 * out_host: room for mrz_rs_encoded_size's input, i.e. (n / 2084880) * 1823248 bytes; *out_len: the bytes `rs-mrzip -d`
 * writes to stdout.  rep: what it prints -- corrected byte errors, uncorrectable codewords (left as they are, like
 * the reference), whether the checksum matched, whether the trailer was missing (then nothing is stripped). */
typedef struct {
    int64_t corrected;
    int64_t uncorrectable;
    int32_t checksum_ok;
    int32_t truncated;
} mrz_rs_report;
int mrz_rs_decode(mrz_ctx *ctx, const void *in, int64_t n, int where, void *out_host, int64_t out_cap, int64_t *out_len,
                  mrz_rs_report *rep);

/* ---- runzip: decoder of the two rzip streams of a chunk ---------- */

/* Replaces the record loop of runzip_chunk (src/runzip.c:277-308) with unzip_literal (:120-157) and
 * unzip_match (:159-207), given the two (already back-end-decompressed) streams of one chunk:
 *   s0/s0_len   stream 0: control records, the 00 00 00 terminator and the 4 stored CRC bytes
 *   s1/s1_len   stream 1: literal bytes
 * Writes the chunk's bytes to `out` (host or device per out_where) and their count to *out_len;
 * MRZ_E_ARG with *out_len set when out_cap is too small.  *crc_calc is the CRC-32 of the output
 * (src/runzip.c:310), *crc_stored the one stored behind the terminator; comparing them is the
 * caller's business (the reference only does when the archive carries no hash, :311-321).
 * Invalid records (distance 0 or beyond the history, empty match, truncated stream, literals beyond
 * stream 1) give MRZ_E_CORRUPT where the reference fatal()s. */
int mrz_runzip_chunk(mrz_ctx *ctx, const void *s0, int64_t s0_len, const void *s1, int64_t s1_len, int where,
                     int chunk_bytes, void *out, int out_where, int64_t out_cap, int64_t *out_len, uint32_t *crc_calc,
                     uint32_t *crc_stored);

#ifdef __cplusplus
}
#endif
#endif
