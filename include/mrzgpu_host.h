/*
 * mrzgpu_host.h -- host driver of libmrzgpu.so: the `mrzip -n` file path built
 * on the C ABI of mrzgpu.h.  It mirrors, for the rzip-only (-n) mode,
 *   void rzip_fd(rzip_control *control, int fd_in, int fd_out)   (include/rzip.h:25, src/rzip.c:807-1132)
 * together with what compress_file (src/mrzip.c:1053-1163) and the stream sink
 * (src/stream.c:771-938 open_stream_out, :1574-1590 write_stream, :1307-1349
 * flush_buffer, :1115-1305 compthread block writer, :1623-1648 close_stream_out)
 * and write_magic (src/mrzip.c:127-188) add around it, so that the bytes
 * written are identical to the reference's output file.
 *
 * mrz_control carries exactly the rzip_control fields that path reads
 * (include/mrzip_private.h:419-524); names follow the reference.
 */
#ifndef MRZGPU_HOST_H
#define MRZGPU_HOST_H

#include "mrzgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int rzip_compression_level; /* -R / -L, 1..9 (src/main.c:575) */
    int compression_level;      /* -L; only recorded in magic[18] (src/mrzip.c:171) */
    int64_t window;             /* -w, units of 100 MiB; 0 = unset (src/rzip.c:883-884) */
    int unlimited;              /* FLAG_UNLIMITED, -U (src/rzip.c:881-882) */
    int64_t ramsize;            /* control->ramsize in bytes (-m * 100 MiB or sysconf) */
    int64_t page_size;          /* control->page_size (4096) */
    int hash_code;              /* control->hash_code; only 1 (MD5, the default) is supported */
    int device;                 /* HIP device ordinal */
    int lz4_test;               /* FLAG_THRESHOLD / LZ4_TEST (-T switches it off, src/main.c:499-510): the pipeline
                                 * asks the LZ4 compressibility gate about every block */
    int threshold;              /* control->threshold, per cent (default 100, src/mrzip.c:1376) */
} mrz_control;

/* As `mrzip -n -L<level> [-w|-U] -m<ramsize>` would write it.  fd_in a regular file: the FILE form of the chunk loop
 * (size from fstat, chunks of max_chunk bytes read one at a time, src/rzip.c:864-866,915-1061), fd_out must seek.
 * fd_in anything else (a pipe): the STDIN form, see mrz_rzip_stream; the output counts as STDOUT if it cannot seek.
 * The archive is written chunk by chunk.  stats (may be NULL) receives the totals printed at -vv. */
int mrz_rzip_fd(const mrz_control *control, int fd_in, int fd_out, mrz_stats *stats);

/* The STDIN form of rzip_fd's chunk loop (mmap_stdin, src/rzip.c:700-732; the loop :915-1061 with STDIN set;
 * setup_ram, src/util.c:156-164): the size is unknown, so every chunk is max_mmap = min(page-rounded maxram,
 * max_chunk) bytes, maxram = ramsize / 3 (ramsize / 6 with to_stdout), until read() returns 0; the chunk in which
 * that happens is shrunk and carries the eof flag -- one more, EMPTY chunk when the input length is a multiple of
 * the chunk size; the stream block size is fixed at the first chunk from the bytes read so far (src/stream.c:803-914).
 * to_stdout = 0: the header is written last, with the total size (src/mrzip.c:1132).  to_stdout = 1: the header goes
 * out first (src/stream.c:1202-1205) and holds the size only if the input ended within the first chunk
 * (src/mrzip.c:137-140); nothing is ever sought, so fd_out may be a pipe.  -U (unlimited) is refused: it takes the
 * window from the file size. */
int mrz_rzip_stream(const mrz_control *control, int fd_in, int fd_out, int to_stdout, mrz_stats *stats);
/* the same with `in` standing for what read() would deliver (tests) */
int mrz_rzip_stream_buffer(const mrz_control *control, const void *in, int64_t n, int to_stdout, void **out,
                           int64_t *out_len, mrz_stats *stats, uint8_t *md5_out);

/* memory -> memory variant of the same path.  *out is malloc'd by the library
 * (release with mrz_free); md5_out (may be NULL) gets the 16 trailing bytes. */
int mrz_rzip_buffer(const mrz_control *control, const void *in, int64_t n, void **out, int64_t *out_len,
                    mrz_stats *stats, uint8_t *md5_out);
void mrz_free(void *p);

/* chunking / block sizing rules alone (src/rzip.c:875-894, src/util.c:156-176,
 * src/stream.c:797-914 for -n): returns max_chunk, *stream_bufsize optional */
int64_t mrz_plan(const mrz_control *control, int64_t st_size, int64_t *stream_bufsize);

/* ---- back-end hand-off --------------------------------------------
 * What the reference's sink does between rzip and the back-end codecs: every stream buffer that fills
 * (stream_bufsize bytes, src/stream.c:878-914) is handed over as one block the moment it fills, DURING
 * hash_search (write_sbstream src/rzip.c:197-211 -> flush_buffer src/stream.c:1307-1349 -> compthread
 * :1115-1305); the output keeps flush order; a back-end that honours LZ4_TEST first asks lz4_compresses
 * (src/stream.c:1685-1733, called from lzma/zpaq/bzip3_compress_buf :249,167,124) whether the block is worth
 * compressing.  mrz_rzip_pipeline runs the GPU rzip stage over the chunks of `in`; as soon as a segment of a
 * chunk has been sequenced the host encodes the records that have become final (put_match / put_literal,
 * src/rzip.c:179-227) and cuts blocks; a consumer thread runs the LZ4 gate on the device for every block
 * (control->lz4_test) and then calls `fn` -- once per block, in flush order, while the GPU is still working on
 * the rest of the same chunk.  `fn` is where a back-end would compress (and frame) the block.  A non-zero
 * return of `fn` aborts the run and is returned. */
typedef struct {
    int chunk_index;      /* 0, 1, ... */
    int stream;           /* 0 = control records, 1 = literal bytes */
    int chunk_bytes;      /* width of the header fields of this chunk's blocks (src/rzip.c:1006-1008) */
    int eof;              /* this is the file's last chunk (src/rzip.c:1049) */
    int64_t chunk_size;
    int first_of_chunk;   /* first block of the chunk: the chunk header precedes it in the file */
    int lz4_verdict;      /* lz4_compresses' answer for this block: 0 = does not compress (store it), 1..100 = per
                           * cent of its size; -1 = not asked (control->lz4_test == 0, or fewer than 64 bytes:
                           * compthread only compresses blocks of >= 64 bytes, src/stream.c:1147) */
    int64_t input_final;  /* bytes of the chunk that were final (sequenced) when the block was cut */
} mrz_block_info;
typedef int (*mrz_block_fn)(void *user, const mrz_block_info *info, const uint8_t *payload, int64_t len);
int mrz_rzip_pipeline(const mrz_control *control, const void *in, int64_t n, mrz_block_fn fn, void *user,
                      mrz_stats *stats, uint8_t *md5_out);

/* `mrzip -d` of a whole -n archive held in memory: runzip_fd (src/runzip.c:332-437) over
 * runzip_chunk (:226-330), the block chains of the two streams (fill_buffer, src/stream.c:1412-1571;
 * CTYPE_NONE blocks only -- anything a back-end codec wrote gives MRZ_E_UNSUPPORTED) and the final
 * hash check (:384-412; per-chunk CRC instead when the archive carries no hash, :311-321).
 * *out is malloc'd by the library (mrz_free).  Record decoding runs on the GPU (mrz_runzip_chunk). */
int mrz_runzip_buffer(int device, const void *mrz, int64_t n, void **out, int64_t *out_len);

#ifdef __cplusplus
}
#endif
#endif
