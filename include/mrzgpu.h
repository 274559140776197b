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

#define MRZ_ABI_VERSION 4

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
    int32_t n_deep;     /* ... and on the deep engine (mrz_seq_deep.hip); the rest on the wide engine */
    int32_t reserved;
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
 * for 4 streams.  (In one process this needs the HIP runtime to open enough hardware queues: the HOST PROGRAM has to
 * export GPU_MAX_HW_QUEUES >= 2 x ctxs before the runtime initialises -- the library does not touch the environment.) */
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

/* ---- the front end's candidate list; one window over several GPUs (second row; BASELINE configs[3]) ----
 * The tag of every position (single_full_tag / next_tag, src/rzip.c:330-358) is computed by a parallel front end, and
 * what the exact matcher consumes is the position-ordered list of the CANDIDATES -- the positions whose tag passes the
 * matcher's minimum_tag_mask (:573), 16 bytes {position, tag} each -- of one stretch of the chunk after another.
 * A front-end pass looks at no more than `segment positions` positions and stops where its list is full (`candidate
 * capacity` entries); the next pass goes on from there.
 *
 * What shards inside ONE chunk is this front end: every rank owns a contiguous byte range of the window (plus a
 * 48-byte halo) and scans the stretches that lie in it; the exact matcher itself stays on one rank, which takes the
 * lists from their owners instead of scanning.
 *
 * mrz_window_scan (any rank): one pass over positions [seg_start, seg_start + max_span) of a chunk of chunk_n bytes,
 * given the rank's bytes [range_start, range_start + range_len) -- which must reach 48 bytes past the last position
 * looked at (30 for the tags, the rest because the kernels load 16-byte pieces) or to the end of the chunk; seg_start
 * and max_span multiples of 4096.  min_mask: the matcher's minimum_tag_mask as last heard of (an older, looser mask
 * only makes the list a superset, which the matcher re-checks); p_done: positions at or before it are left out.
 * Outputs, in host or device memory per out_where: cand_out, room for `cap` entries; tile_off_out, max_span / 4096 + 1
 * ints (list offset of every 4096-position tile); bitmap_out, max_span / 8 bytes (1 pass bit per position).
 * *scan_next: where the pass stopped (the next one starts there; a multiple of 4096); *n_cand: entries written.
 *
 * mrz_set_cand_provider (the matcher's rank): mrz_rzip_chunk then calls `fn` for every stretch it is about to
 * sequence instead of scanning it; `fn` has to fill the three DEVICE buffers (as mrz_window_scan lays them out; e.g.
 * by hipMemcpyAsync on `stream`, by a peer copy, or mrz_copy_to_device), set *scan_next and *n_cand, and return 0.
 * fn = NULL: scan locally. */
typedef struct {
    int64_t pos; /* position in the chunk */
    int64_t tag; /* its tag (a 47-bit value, typedef i64 tag, include/mrzip_private.h:375) */
} mrz_candidate;
int mrz_window_scan(mrz_ctx *ctx, const void *range_bytes, int64_t range_len, int where, int64_t range_start,
                    int64_t chunk_n, int64_t seg_start, int64_t max_span, int64_t min_mask, int64_t p_done, int64_t cap,
                    mrz_candidate *cand_out, int32_t *tile_off_out, void *bitmap_out, int out_where, int64_t *scan_next,
                    int64_t *n_cand);
typedef int (*mrz_cand_provider_fn)(void *user, int64_t seg_start, int64_t max_span, int64_t min_mask, int64_t p_done,
                                    int64_t cap, mrz_candidate *d_cand, int32_t *d_tile_off, void *d_bitmap,
                                    int64_t *scan_next, int64_t *n_cand, void *stream);
int mrz_set_cand_provider(mrz_ctx *ctx, mrz_cand_provider_fn fn, void *user);
/* positions a front-end pass looks at, at most (a multiple of 4096; default 2^30), and entries its list holds (>= 4096;
 * default 8 Mi = 128 MiB) */
int mrz_set_segment_positions(mrz_ctx *ctx, int64_t positions);
int mrz_set_candidate_capacity(mrz_ctx *ctx, int64_t entries);
/* host -> device / device -> device copy on the ctx stream (what a provider without a HIP runtime of its own fills the
 * buffers with); returns when the source may be reused */
int mrz_copy_to_device(mrz_ctx *ctx, void *dst_device, const void *src_host, int64_t n);
int mrz_copy_device(mrz_ctx *ctx, void *dst_device, const void *src_device, int64_t n);
/* The window's BYTES when its ranges live on several GPUs (one process per GPU).  The reference maps the whole file and
 * single_match_len compares anywhere in it (src/rzip.c:372-397; -U: one chunk = the file, :881-882).  Here every rank
 * keeps its range in its own HBM as a shareable physical allocation (HIP virtual memory management):
 *   mrz_window_part_create   allocates `bytes` (a multiple of mrz_window_granularity) on `device`, maps it for the
 *                            owner (*dptr: where the owner writes its range) and exports it as a POSIX file descriptor
 *                            (*fd, owned by the part) that the host program hands to the other processes (SCM_RIGHTS);
 *   mrz_window_map_create    imports n_parts descriptors (its own among them) and maps them back to back, in the order
 *                            given, into ONE virtual address range on `device`: *dptr + position is the window's byte,
 *                            whichever GPU holds it -- loads that fall into another rank's range go over xGMI.  The
 *                            matcher's rank passes it to mrz_rzip_chunk as the chunk (MRZ_MEM_DEVICE): match extension,
 *                            compare farm, literal gather and CRC read the bytes where they lie, nothing is gathered;
 *                            the other ranks pass their range of it (plus the 48-byte halo, which is the next rank's
 *                            memory) to mrz_window_scan.
 * Sizes are the mapped sizes of the parts (multiples of the granularity); the descriptors stay the caller's. */
typedef struct mrz_window_part mrz_window_part;
typedef struct mrz_window_map mrz_window_map;
int64_t mrz_window_granularity(int device); /* > 0, or a negative MRZ_E_ code */
int mrz_window_part_create(int device, int64_t bytes, mrz_window_part **out, void **dptr, int *fd);
void mrz_window_part_destroy(mrz_window_part *part);
int mrz_window_map_create(int device, int n_parts, const int *fds, const int64_t *sizes, mrz_window_map **out, void **dptr);
void mrz_window_map_destroy(mrz_window_map *map);
/* Which of the GPU's 8 XCDs carries this ctx's sequencer workgroups (0..7; default 0): block index mod 8 under the
 * round-robin placement of workgroups.  The exact matcher of one chunk is one dependency chain on one XCD; ctxs that run
 * concurrently (independent streams, chunks of one file) should get different values so that each has an XCD's
 * L2 and CUs to itself. */
int mrz_set_xcd(mrz_ctx *ctx, int xcd);

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
