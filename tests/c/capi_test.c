/* tests/c/capi_test.c -- TEST: a plain C (gcc, C99) caller of libmrzgpu.so, the way the reference -- a C program --
 * would bind it: mrz_open -> mrz_rzip_chunk -> mrz_fetch_streams -> mrz_close, and mrz_rzip_fd on a real file and
 * on a pipe, each compared with the oracle (liboracle.so: the checker, linked by this test only).
 *   gcc -std=c99 -Iinclude -Ioracle tests/c/capi_test.c -o capi_test -Lmodern-rzip_amd -lmrzgpu -Loracle -loracle */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/types.h>
#include <sys/wait.h>

#include "mrzgpu.h"
#include "mrzgpu_host.h"
#include "mrz_oracle.h"

#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) {                                                     \
            fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #c); \
            exit(1);                                                    \
        }                                                               \
    } while (0)

static uint64_t rng = 88172645463325252ull;
static uint8_t next_byte(void) {
    rng ^= rng << 13;
    rng ^= rng >> 7;
    rng ^= rng << 17;
    return (uint8_t)(rng >> 24);
}

int main(void) {
    /* a stream with literals, short and long matches: noise, a repeat of it with one byte changed, text-like filler */
    const int64_t n = 3 * 1000 * 1000 + 123;
    uint8_t *in = malloc((size_t)n);
    CHECK(in);
    for (int64_t i = 0; i < 1000000; i++) in[i] = next_byte();
    memcpy(in + 1000000, in, 1000000);
    in[1500000] ^= 0x55;
    for (int64_t i = 2000000; i < n; i++) in[i] = (uint8_t)("the quick brown fox "[i % 20] + (i % 7919 == 0));

    CHECK(mrz_abi_version() == MRZ_ABI_VERSION);

    /* ---- per chunk: what rzip_chunk would call instead of hash_search --------------------------------------- */
    mrz_ctx *ctx = NULL;
    CHECK(mrz_open(&ctx, 0, 7, n) == MRZ_OK);
    const int cb = mrz_chunk_bytes(n);
    int64_t vr = 3;
    mrz_chunk_result r;
    CHECK(mrz_rzip_chunk(ctx, in, n, MRZ_MEM_HOST, cb, &vr, &r) == MRZ_OK);
    uint8_t *s0 = malloc((size_t)r.s0_len), *s1 = malloc((size_t)r.s1_len + 1);
    CHECK(s0 && s1);
    CHECK(mrz_fetch_streams(ctx, s0, s1) == MRZ_OK);

    mrzo_matcher *om = mrzo_matcher_new(7);
    CHECK(om);
    mrzo_matcher_set_victim_round(om, 3);
    mrzo_buf o0 = { 0, 0, 0 }, o1 = { 0, 0, 0 };
    uint32_t ocrc = 0;
    CHECK(mrzo_rzip_chunk(om, in, n, mrzo_chunk_bytes(n), &o0, &o1, &ocrc) == 0);
    CHECK(cb == mrzo_chunk_bytes(n));
    CHECK(r.s0_len == o0.len && r.s1_len == o1.len);
    CHECK(!memcmp(s0, o0.p, (size_t)o0.len) && !memcmp(s1, o1.p, (size_t)o1.len));
    CHECK(r.crc32 == ocrc && vr == mrzo_matcher_get_victim_round(om));
    const mrzo_stats *os = mrzo_matcher_stats(om);
    CHECK(r.stats.inserts == os->inserts && r.stats.matches == os->matches && r.stats.tag_hits == os->tag_hits &&
          r.stats.tag_misses == os->tag_misses && r.stats.literal_bytes == os->literal_bytes);
    CHECK(r.stats.matches >= 2);
    mrz_close(ctx);
    mrzo_matcher_free(om);

    /* ---- whole file: mrz_rzip_fd on a regular file, then on a pipe (the STDIN form) ---------------------------- */
    mrz_control ctl;
    memset(&ctl, 0, sizeof(ctl));
    ctl.rzip_compression_level = ctl.compression_level = 7;
    ctl.ramsize = 3 * 1024 * 1024; /* chunks of 2 MiB (file) / 1 MiB (STDIN): several chunks */
    ctl.page_size = 4096;
    ctl.hash_code = 1;
    mrzo_params prm = { 7, 0, 0, ctl.ramsize, 4096 };

    char tin[] = "/tmp/mrz_capi_in_XXXXXX", tout[] = "/tmp/mrz_capi_out_XXXXXX";
    int fi = mkstemp(tin), fo = mkstemp(tout);
    CHECK(fi >= 0 && fo >= 0);
    CHECK(write(fi, in, (size_t)n) == (ssize_t)n && lseek(fi, 0, SEEK_SET) == 0);
    mrz_stats st;
    CHECK(mrz_rzip_fd(&ctl, fi, fo, &st) == MRZ_OK);
    mrzo_buf want = { 0, 0, 0 };
    uint8_t md5[16];
    CHECK(mrzo_compress(&prm, in, n, &want, NULL, md5) == 0);
    uint8_t *got = malloc((size_t)want.len + 1);
    CHECK(lseek(fo, 0, SEEK_END) == (off_t)want.len && lseek(fo, 0, SEEK_SET) == 0);
    CHECK(read(fo, got, (size_t)want.len) == (ssize_t)want.len && !memcmp(got, want.p, (size_t)want.len));
    close(fi);

    int pfd[2];
    CHECK(pipe(pfd) == 0);
    pid_t kid = fork();
    CHECK(kid >= 0);
    if (!kid) { /* the writer end of the pipe: `cat file |` */
        close(pfd[0]);
        for (int64_t at = 0; at < n;) {
            ssize_t w = write(pfd[1], in + at, (size_t)(n - at > 70001 ? 70001 : n - at));
            if (w <= 0) _exit(2);
            at += w;
        }
        _exit(0);
    }
    close(pfd[1]);
    CHECK(ftruncate(fo, 0) == 0 && lseek(fo, 0, SEEK_SET) == 0);
    CHECK(mrz_rzip_fd(&ctl, pfd[0], fo, &st) == MRZ_OK);
    int wst = 0;
    CHECK(waitpid(kid, &wst, 0) == kid && WIFEXITED(wst) && WEXITSTATUS(wst) == 0);
    mrzo_buf wants = { 0, 0, 0 };
    int nch = 0;
    CHECK(mrzo_compress_stream(&prm, in, n, 0, &wants, NULL, md5, &nch) == 0 && nch >= 3);
    got = realloc(got, (size_t)wants.len + 1);
    CHECK(lseek(fo, 0, SEEK_END) == (off_t)wants.len && lseek(fo, 0, SEEK_SET) == 0);
    CHECK(read(fo, got, (size_t)wants.len) == (ssize_t)wants.len && !memcmp(got, wants.p, (size_t)wants.len));
    close(fo);
    unlink(tin);
    unlink(tout);

    /* and back: mrzip -d of that archive through the GPU decoder */
    void *back = NULL;
    int64_t back_len = 0;
    CHECK(mrz_runzip_buffer(0, wants.p, wants.len, &back, &back_len) == MRZ_OK);
    CHECK(back_len == n && !memcmp(back, in, (size_t)n));
    mrz_free(back);
    printf("capi_test ok: %lld bytes, %lld matches, file %lld B, stdin-form %lld B in %d chunks\n", (long long)n,
           (long long)r.stats.matches, (long long)want.len, (long long)wants.len, nch);
    return 0;
}
