// mrz_host.hip -- host driver (include/mrzgpu_host.h): the `mrzip -n` file path
// around the GPU rzip stage.  Host code only: chunk loop, -n stream sink and
// block framing, magic header, whole-file MD5 on a helper thread (the
// reference also keeps the checksums off the matcher's thread,
// src/rzip.c:488-505).  MD5 is a strictly serial hash; it overlaps the GPU work.
#include <errno.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <new>
#include <stdexcept>

#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

// joins a thread when the scope is left, whichever way (an allocation that throws between the thread's start and its
// join would otherwise destroy a joinable std::thread: std::terminate instead of MRZ_E_NOMEM at the C ABI)
struct mrz_join_guard {
    std::thread &t;
    explicit mrz_join_guard(std::thread &th) : t(th) {}
    ~mrz_join_guard() {
        if (t.joinable()) t.join();
    }
};
#include <vector>

#include "../../include/mrzgpu_host.h"
#include "mrz_ctx.h"

namespace {

const int64_t kMiB = 1048576;
const int64_t kStreamMin = 10 * kMiB;   // STREAM_BUFSIZE, include/mrzip_private.h:27
const int64_t kChunkUnit = 100 * kMiB;  // CHUNK_MULTIPLE, src/rzip.c:46

// ---- MD5 (RFC 1321; libgcrypt GCRY_MD_MD5 in the reference) ---------------
struct Md5 {
    uint32_t h[4];
    uint64_t total;
    uint8_t pend[64];
    size_t npend;
    Md5() : total(0), npend(0) {
        h[0] = 0x67452301u;
        h[1] = 0xefcdab89u;
        h[2] = 0x98badcfeu;
        h[3] = 0x10325476u;
    }
    static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
    void block(const uint8_t *p) {
        static const uint32_t K[64] = {
            0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
            0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
            0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
            0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
            0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
            0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
            0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
            0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391 };
        static const int S[4][4] = { { 7, 12, 17, 22 }, { 5, 9, 14, 20 }, { 4, 11, 16, 23 }, { 6, 10, 15, 21 } };
        uint32_t m[16];
        memcpy(m, p, 64);  // little-endian host
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
        for (int i = 0; i < 64; i++) {
            const int r = i >> 4;
            uint32_t f;
            int g;
            switch (r) {
                case 0: f = d ^ (b & (c ^ d)); g = i; break;
                case 1: f = c ^ (d & (b ^ c)); g = (5 * i + 1) & 15; break;
                case 2: f = b ^ c ^ d; g = (3 * i + 5) & 15; break;
                default: f = c ^ (b | ~d); g = (7 * i) & 15; break;
            }
            const uint32_t nb = b + rol(a + f + K[i] + m[g], S[r][i & 3]);
            a = d;
            d = c;
            c = b;
            b = nb;
        }
        h[0] += a;
        h[1] += b;
        h[2] += c;
        h[3] += d;
    }
    void update(const uint8_t *p, size_t n) {
        total += n;
        if (npend) {
            size_t take = 64 - npend;
            if (take > n) take = n;
            memcpy(pend + npend, p, take);
            npend += take;
            p += take;
            n -= take;
            if (npend < 64) return;
            block(pend);
            npend = 0;
        }
        for (; n >= 64; p += 64, n -= 64) block(p);
        if (n) {
            memcpy(pend, p, n);
            npend = n;
        }
    }
    void finish(uint8_t out[16]) {
        const uint64_t bits = total * 8;
        uint8_t tail[128];
        size_t k = npend;
        memcpy(tail, pend, k);
        tail[k++] = 0x80;
        while (k % 64 != 56) tail[k++] = 0;
        for (int i = 0; i < 8; i++) tail[k++] = (uint8_t)(bits >> (8 * i));
        for (size_t o = 0; o < k; o += 64) block(tail + o);
        memcpy(out, h, 16);
    }
};

int64_t page_floor(int64_t v, int64_t page) {  // round_to_page, src/util.c:166-169
    v -= v % page;
    return v ? v : page;
}
int64_t page_ceil(int64_t v, int64_t page) {  // round_up_page, src/util.c:171-176
    const int64_t r = v % page;
    return r ? v + page - r : v;
}

// Replays a chunk's records into a two-stream writer so that "buffer full" events of the two streams
// interleave as in the reference: the stream-1 bytes of a literal follow its 3-byte header (put_literal,
// src/rzip.c:213-227).  W::write(stream, bytes, n).
template <class W>
int mrz_replay_records(W &w, int cb, const uint8_t *s0, int64_t n0, const uint8_t *s1, int64_t n1) {
    int64_t i = 0, j = 0;
    while (i < n0) {
        if (i + 3 > n0) return MRZ_E_STATE;
        const int head = s0[i];
        const int64_t len = s0[i + 1] | (int64_t)s0[i + 2] << 8;
        if (head == 0) {
            w.write(0, s0 + i, 3);
            i += 3;
            if (len == 0) {
                if (i + 4 != n0) return MRZ_E_STATE;
                w.write(0, s0 + i, 4);
                i += 4;
                break;
            }
            if (j + len > n1) return MRZ_E_STATE;
            w.write(1, s1 + j, len);
            j += len;
        } else {
            if (i + 3 + cb > n0) return MRZ_E_STATE;
            w.write(0, s0 + i, 3 + cb);
            i += 3 + cb;
        }
    }
    return j == n1 ? MRZ_OK : MRZ_E_STATE;
}

// ---- the -n stream sink ----------------------------------------------------
// Two logical streams are cut into blocks whenever a stream buffer of
// `bufsize` bytes fills (write_stream / flush_buffer), blocks are appended to
// the file in flush order, each with a {ctype, c_len, u_len, next} header whose
// `next` field is patched when the following block of the same stream lands
// (compthread, src/stream.c:1199-1293).
struct Sink {
    std::vector<uint8_t> &out;
    int64_t bufsize;
    int cb = 1;
    int eof = 0;
    int64_t size_field = 0;
    int64_t initial_pos = 0, cur_pos = 0;
    int64_t last_head[2] = { 0, 0 };
    int blocks = 0;
    std::vector<uint8_t> sbuf[2];

    Sink(std::vector<uint8_t> &o, int64_t bs) : out(o), bufsize(bs) {}

    void put_val(int64_t v, int width) {
        for (int i = 0; i < width; i++) out.push_back((uint8_t)((uint64_t)v >> (8 * i)));
    }
    void begin_chunk(int64_t chunk_size, int chunk_bytes, int is_last, int64_t page) {
        cb = chunk_bytes;
        eof = is_last;
        size_field = chunk_size < page ? page : chunk_size;  // src/stream.c:779-780
        cur_pos = 0;
        blocks = 0;
        sbuf[0].clear();
        sbuf[1].clear();
    }
    void emit_block(int s) {
        if (!blocks++) {
            out.push_back((uint8_t)cb);
            out.push_back((uint8_t)eof);
            put_val(size_field, cb);
            initial_pos = (int64_t)out.size();
            for (int j = 0; j < 2; j++) {
                last_head[j] = cur_pos + 1 + 2 * cb;
                out.push_back(3);  // CTYPE_NONE
                put_val(0, cb);
                put_val(0, cb);
                put_val(0, cb);
                cur_pos += 1 + 3 * cb;
            }
        }
        for (int i = 0; i < cb; i++)
            out[(size_t)(initial_pos + last_head[s] + i)] = (uint8_t)((uint64_t)cur_pos >> (8 * i));
        last_head[s] = cur_pos + 1 + 2 * cb;
        const int64_t len = (int64_t)sbuf[s].size();
        out.push_back(3);
        put_val(len, cb);
        put_val(len, cb);
        put_val(0, cb);
        cur_pos += 1 + 3 * cb;
        out.insert(out.end(), sbuf[s].begin(), sbuf[s].end());
        cur_pos += len;
        sbuf[s].clear();
    }
    void write(int s, const uint8_t *p, int64_t n) {
        while (n) {
            const int64_t room = bufsize - (int64_t)sbuf[s].size();
            const int64_t take = room < n ? room : n;
            sbuf[s].insert(sbuf[s].end(), p, p + take);
            p += take;
            n -= take;
            if ((int64_t)sbuf[s].size() == bufsize) emit_block(s);
        }
    }
    int feed(const uint8_t *s0, int64_t n0, const uint8_t *s1, int64_t n1) {
        const int rc = mrz_replay_records(*this, cb, s0, n0, s1, n1);
        if (rc) return rc;
        emit_block(0);  // close_stream_out flushes both, even when empty
        emit_block(1);
        return MRZ_OK;
    }
};

// ---- where the chunks come from, where the archive goes ---------------------------------------------------
// The chunk loop of rzip_fd (src/rzip.c:915-1061) in its two forms: a FILE, whose size is known -- chunks of
// max_chunk bytes, the last one carries the eof flag (:1049) -- and STDIN (mmap_stdin :700-732) -- chunks of
// max_mmap bytes until read() returns 0; the chunk in which that happens is shrunk and carries the eof flag,
// which is one more, empty chunk when the input length is a multiple of the chunk size.
struct Source {
    virtual ~Source() {}
    // next chunk into `buf` (resized); *eof: the reference's control->eof for this chunk; returns an MRZ_ code
    virtual int next(std::vector<uint8_t> &buf, int *eof) = 0;
    virtual bool more() const = 0;  // while (!pass || len > 0 || (STDIN && !stdin_eof))
};

struct Reader {  // bytes from memory or from a file descriptor
    const uint8_t *mem = nullptr;
    int64_t mem_n = 0, mem_at = 0;
    int fd = -1;
    // read(): up to `want` bytes, 0 at the end of the input, < 0 on error
    int64_t get(uint8_t *dst, int64_t want) {
        if (fd < 0) {
            const int64_t k = mem_n - mem_at < want ? mem_n - mem_at : want;
            if (k > 0) memcpy(dst, mem + mem_at, (size_t)k);
            mem_at += k;
            return k;
        }
        for (;;) {
            const ssize_t r = read(fd, dst, (size_t)(want > (1ll << 30) ? (1ll << 30) : want));
            if (r < 0 && errno == EINTR) continue;
            return (int64_t)r;
        }
    }
    // fill dst[0..want) unless the input ends first; returns the bytes got, *hit_end if a read returned 0
    int64_t fill(uint8_t *dst, int64_t want, bool *hit_end) {
        int64_t got = 0;
        *hit_end = false;
        while (got < want) {
            const int64_t r = get(dst + got, want - got);
            if (r < 0) return -1;
            if (r == 0) {
                *hit_end = true;
                break;
            }
            got += r;
        }
        return got;
    }
};

struct FileSource : Source {  // size known up front
    Reader rd;
    int64_t left, max_chunk;
    int pass = 0;
    FileSource(const Reader &r, int64_t n, int64_t mc) : rd(r), left(n), max_chunk(mc) {}
    bool more() const override { return !pass || left > 0; }
    int next(std::vector<uint8_t> &buf, int *eof) override {
        const int64_t csz = max_chunk < left ? max_chunk : left;
        buf.resize((size_t)csz);
        bool hit;
        if (rd.fill(buf.data(), csz, &hit) != csz) return MRZ_E_ARG;  // the file shrank under us
        *eof = csz == left;
        left -= csz;
        pass++;
        return MRZ_OK;
    }
};

struct StdinSource : Source {  // size unknown: mmap_stdin
    Reader rd;
    int64_t max_mmap;
    bool seen_eof = false;
    StdinSource(const Reader &r, int64_t mm) : rd(r), max_mmap(mm) {}
    bool more() const override { return !seen_eof; }
    int next(std::vector<uint8_t> &buf, int *eof) override {
        buf.resize((size_t)max_mmap);
        bool hit;
        const int64_t got = rd.fill(buf.data(), max_mmap, &hit);
        if (got < 0) return MRZ_E_ARG;
        buf.resize((size_t)got);  // "Shrinking chunk"
        seen_eof = hit;
        *eof = hit ? 1 : 0;
        return MRZ_OK;
    }
};

struct Out {  // the archive: memory, or a file descriptor written chunk by chunk
    std::vector<uint8_t> *vec = nullptr;
    int fd = -1;
    int put(const uint8_t *p, size_t n) {
        if (vec) {
            vec->insert(vec->end(), p, p + n);
            return MRZ_OK;
        }
        size_t off = 0;
        while (off < n) {
            const ssize_t w = write(fd, p + off, n - off);
            if (w < 0) {
                if (errno == EINTR) continue;
                return MRZ_E_ARG;
            }
            off += (size_t)w;
        }
        return MRZ_OK;
    }
    int put_at0(const uint8_t *p, size_t n) {  // fdout_seekto(0) + put_fdout, src/mrzip.c:176-178
        if (vec) {
            memcpy(vec->data(), p, n);
            return MRZ_OK;
        }
        size_t off = 0;
        while (off < n) {
            const ssize_t w = pwrite(fd, p + off, n - off, (off_t)off);
            if (w < 0) {
                if (errno == EINTR) continue;
                return MRZ_E_ARG;
            }
            off += (size_t)w;
        }
        return MRZ_OK;
    }
};

void fill_magic(uint8_t mg[20], const mrz_control *ctl, int64_t st_size) {  // write_magic, src/mrzip.c:127-188
    memset(mg, 0, 20);
    mg[0] = 'M';
    mg[1] = 'R';
    mg[2] = 'Z';
    mg[3] = 'I';
    mg[4] = 0;  // MRZIP_MAJOR
    mg[5] = 9;  // MRZIP_MINOR
    for (int i = 0; i < 8; i++) mg[6 + i] = (uint8_t)((uint64_t)st_size >> (8 * i));
    mg[14] = 1;  // hash_code: MD5
    mg[18] = (uint8_t)((ctl->rzip_compression_level << 4) + ctl->compression_level);
}

// stdin_mode: 0 file (st_size known), 1 STDIN -> file, 2 STDIN -> STDOUT
int run_chunks(const mrz_control *ctl, Source &src, int stdin_mode, int64_t st_size_known, int64_t open_chunk, Out &out,
               mrz_stats *stats, uint8_t *md5_out) {
    const int64_t page = ctl->page_size > 0 ? ctl->page_size : 4096;
    mrz_ctx *ctx = nullptr;
    int rc = mrz_open(&ctx, ctl->device, ctl->rzip_compression_level, open_chunk);
    if (rc) return rc;
    mrz_stats total;
    memset(&total, 0, sizeof(total));
    Md5 md5h;  // whole-input hash (src/rzip.c:1069-1090), fed chunk by chunk on a helper thread beside the GPU
    uint8_t md5[16], mg[20];
    std::vector<uint8_t> chunk, piece, s0, s1;
    int64_t bufsize = 0, st_size = 0, victim_round = 0;  // victim_round: first rzip_fd call of the process
    int nch = 0;
    try {
        if (stdin_mode != 2) {  // header placeholder (compress_file, src/mrzip.c:1126)
            memset(mg, 0, 20);
            rc = out.put(mg, 20);
        }
        Sink sink(piece, 0);
        while (!rc && src.more()) {
            int eof = 0;
            rc = src.next(chunk, &eof);
            if (rc) break;
            const int64_t csz = (int64_t)chunk.size();
            st_size += csz;
            if (!nch) {
                // the block size is fixed at the first open_stream_out (src/stream.c:803-914, NO_COMPRESS, one thread)
                if (stdin_mode) {
                    const int64_t usable = ctl->ramsize / (stdin_mode == 2 ? 6 : 3);  // setup_ram, src/util.c:156-164
                    const int64_t chunk_limit = csz < page ? page : csz;
                    int64_t limit = usable;
                    if (st_size > 0 && st_size < limit)
                        limit = st_size > kStreamMin ? st_size : kStreamMin;
                    else if (limit > chunk_limit)
                        limit = chunk_limit;
                    bufsize = page_ceil(limit, page);
                } else
                    mrz_plan(ctl, st_size_known, &bufsize);
                sink.bufsize = bufsize;
                if (stdin_mode == 2) {
                    // STDOUT: the first block writes the magic header (src/stream.c:1202-1205); the size is only in
                    // it if the input has already ended (write_magic, src/mrzip.c:137-140)
                    fill_magic(mg, ctl, eof ? st_size : 0);
                    rc = out.put(mg, 20);
                    if (rc) break;
                }
            }
            std::thread hasher([&]() { md5h.update(chunk.data(), (size_t)csz); });
            mrz_join_guard hasher_guard(hasher);
            const int cb = mrz_chunk_bytes(csz);
            mrz_chunk_result res;
            rc = mrz_rzip_chunk(ctx, chunk.data(), csz, MRZ_MEM_HOST, cb, &victim_round, &res);
            if (!rc) {
                s0.resize((size_t)res.s0_len);
                s1.resize((size_t)res.s1_len);
                rc = mrz_fetch_streams(ctx, s0.data(), s1.data());
            }
            hasher.join();
            if (rc) break;
            total.inserts += res.stats.inserts;
            total.literals += res.stats.literals;
            total.literal_bytes += res.stats.literal_bytes;
            total.matches += res.stats.matches;
            total.match_bytes += res.stats.match_bytes;
            total.tag_hits += res.stats.tag_hits;
            total.tag_misses += res.stats.tag_misses;
            piece.clear();
            sink.begin_chunk(csz, cb, eof, page);
            rc = sink.feed(s0.data(), res.s0_len, s1.data(), res.s1_len);
            if (!rc) rc = out.put(piece.data(), piece.size());
            nch++;
        }
        if (!rc) {
            md5h.finish(md5);
            rc = out.put(md5, 16);
        }
        if (!rc && stdin_mode != 2) {  // compress_file writes the header last (src/mrzip.c:1132)
            fill_magic(mg, ctl, st_size);
            rc = out.put_at0(mg, 20);
        }
    } catch (const std::bad_alloc &) {
        rc = MRZ_E_NOMEM;
    }
    mrz_close(ctx);
    if (rc) return rc;
    if (stats) *stats = total;
    if (md5_out) memcpy(md5_out, md5, 16);
    return MRZ_OK;
}

int check_control(const mrz_control *ctl) {
    if (!ctl || ctl->rzip_compression_level < 1 || ctl->rzip_compression_level > 9) return MRZ_E_ARG;
    if (ctl->hash_code != 1 && ctl->hash_code != 0) return MRZ_E_ARG;  // MD5 only (reference default)
    return MRZ_OK;
}

// chunk size of STDIN mode: max_mmap, src/rzip.c:875-894
int64_t stdin_chunk(const mrz_control *ctl, int to_stdout) {
    const int64_t page = ctl->page_size > 0 ? ctl->page_size : 4096;
    int64_t max_mmap = page_floor(ctl->ramsize / (to_stdout ? 6 : 3), page);
    const int64_t max_chunk = ctl->window ? ctl->window * kChunkUnit : ctl->ramsize / 3 * 2;
    return max_mmap < max_chunk ? max_mmap : max_chunk;
}

int run_file(const mrz_control *ctl, const Reader &rd, int64_t n, Out &out, mrz_stats *stats, uint8_t *md5_out) {
    int rc = check_control(ctl);
    if (rc || n < 0) return MRZ_E_ARG;
    const int64_t max_chunk = mrz_plan(ctl, n, nullptr);
    FileSource src(rd, n, max_chunk);
    return run_chunks(ctl, src, 0, n, max_chunk < n ? max_chunk : n, out, stats, md5_out);
}

int run_stdin(const mrz_control *ctl, const Reader &rd, int to_stdout, Out &out, mrz_stats *stats, uint8_t *md5_out) {
    int rc = check_control(ctl);
    if (rc) return rc;
    if (ctl->unlimited) return MRZ_E_ARG;  // -U takes the window from the file size, which STDIN does not have
    const int64_t mm = stdin_chunk(ctl, to_stdout);
    if (mm <= 0) return MRZ_E_ARG;
    StdinSource src(rd, mm);
    return run_chunks(ctl, src, to_stdout ? 2 : 1, 0, mm, out, stats, md5_out);
}

int vec_to_malloc(std::vector<uint8_t> &buf, void **out, int64_t *out_len) {
    void *p = malloc(buf.size() ? buf.size() : 1);
    if (!p) return MRZ_E_NOMEM;
    memcpy(p, buf.data(), buf.size());
    *out = p;
    *out_len = (int64_t)buf.size();
    return MRZ_OK;
}

}  // namespace

extern "C" int64_t mrz_plan(const mrz_control *ctl, int64_t st_size, int64_t *stream_bufsize) {
    const int64_t page = ctl->page_size > 0 ? ctl->page_size : 4096;
    const int64_t usable = ctl->ramsize / 3;  // setup_ram, src/util.c:156-164 (file -> file)
    int64_t max_chunk;                        // src/rzip.c:881-888
    if (ctl->unlimited)
        max_chunk = st_size;
    else if (ctl->window)
        max_chunk = ctl->window * kChunkUnit;
    else
        max_chunk = ctl->ramsize / 3 * 2;
    if (max_chunk < st_size) max_chunk = page_floor(max_chunk, page);
    if (stream_bufsize) {  // open_stream_out with NO_COMPRESS, src/stream.c:797-805,878-881,913-914
        const int64_t first = st_size < max_chunk ? st_size : max_chunk;
        const int64_t chunk_limit = first < page ? page : first;
        int64_t limit = usable;
        if (st_size > 0 && st_size < limit)
            limit = st_size > kStreamMin ? st_size : kStreamMin;
        else if (limit > chunk_limit)
            limit = chunk_limit;
        *stream_bufsize = page_ceil(limit, page);
    }
    return max_chunk;
}

extern "C" void mrz_free(void *p) { free(p); }

extern "C" int mrz_rzip_buffer(const mrz_control *ctl, const void *in, int64_t n, void **out, int64_t *out_len,
                               mrz_stats *stats, uint8_t *md5_out) {
    if (!out || !out_len || (n > 0 && !in)) return MRZ_E_ARG;
    try {
        std::vector<uint8_t> buf;
        Reader rd;
        rd.mem = (const uint8_t *)in;
        rd.mem_n = n;
        Out o;
        o.vec = &buf;
        const int rc = run_file(ctl, rd, n, o, stats, md5_out);
        if (rc) return rc;
        return vec_to_malloc(buf, out, out_len);
    } catch (const std::bad_alloc &) {
        return MRZ_E_NOMEM;
    }
}

extern "C" int mrz_rzip_stream_buffer(const mrz_control *ctl, const void *in, int64_t n, int to_stdout, void **out,
                                      int64_t *out_len, mrz_stats *stats, uint8_t *md5_out) {
    if (!out || !out_len || (n > 0 && !in)) return MRZ_E_ARG;
    try {
        std::vector<uint8_t> buf;
        Reader rd;
        rd.mem = (const uint8_t *)in;
        rd.mem_n = n;
        Out o;
        o.vec = &buf;
        const int rc = run_stdin(ctl, rd, to_stdout, o, stats, md5_out);
        if (rc) return rc;
        return vec_to_malloc(buf, out, out_len);
    } catch (const std::bad_alloc &) {
        return MRZ_E_NOMEM;
    }
}

extern "C" int mrz_rzip_stream(const mrz_control *ctl, int fd_in, int fd_out, int to_stdout, mrz_stats *stats) {
    try {
        Reader rd;
        rd.fd = fd_in;
        Out o;
        o.fd = fd_out;
        return run_stdin(ctl, rd, to_stdout, o, stats, nullptr);
    } catch (const std::bad_alloc &) {
        return MRZ_E_NOMEM;
    }
}

extern "C" int mrz_rzip_fd(const mrz_control *ctl, int fd_in, int fd_out, mrz_stats *stats) {
    // a regular file is compressed as a FILE (size from fstat, src/rzip.c:864-866), anything else -- a pipe, a
    // terminal -- the way the reference compresses STDIN; the output counts as STDOUT when it cannot seek
    struct stat sb;
    if (fstat(fd_in, &sb)) return MRZ_E_ARG;
    const bool out_seeks = lseek(fd_out, 0, SEEK_CUR) != (off_t)-1;
    if (!S_ISREG(sb.st_mode)) return mrz_rzip_stream(ctl, fd_in, fd_out, out_seeks ? 0 : 1, stats);
    if (!out_seeks) return MRZ_E_ARG;  // file -> STDOUT keeps the output in a temporary buffer upstream: not mirrored
    try {
        const off_t at = lseek(fd_in, 0, SEEK_CUR);
        const int64_t n = (int64_t)sb.st_size - (at > 0 ? (int64_t)at : 0);
        Reader rd;
        rd.fd = fd_in;
        Out o;
        o.fd = fd_out;
        return run_file(ctl, rd, n < 0 ? 0 : n, o, stats, nullptr);
    } catch (const std::bad_alloc &) {
        return MRZ_E_NOMEM;
    }
}

// ---- back-end hand-off pipeline -------------------------------------------------
// The reference hands every full stream buffer to a compthread and keeps the output in flush order
// (flush_buffer src/stream.c:1307-1349, compthread :1115-1305, the output_thread ticket :1199-1201).  Here the
// GPU thread produces the two streams of chunk k+1 while a consumer thread cuts chunk k into blocks of
// stream_bufsize bytes and hands them, in exactly that flush order, to the caller's function -- the place
// where a back-end codec would compress the block.
namespace {

struct Block {
    mrz_block_info info;
    std::vector<uint8_t> payload;
};

// write_stream / flush_buffer without the file: every full stream buffer becomes one Block for the consumer
struct BlockCutter {
    int64_t bufsize = 0;
    mrz_block_info info;
    std::vector<uint8_t> sbuf[2];
    std::mutex *mu = nullptr;
    std::condition_variable *cv = nullptr;
    std::deque<Block *> *queue = nullptr;
    const int *abort_rc = nullptr;
    void flush(int s) {
        Block *b = new Block;
        b->info = info;
        b->info.stream = s;
        b->payload.swap(sbuf[s]);
        info.first_of_chunk = 0;
        {
            std::unique_lock<std::mutex> lk(*mu);
            cv->wait(lk, [&] { return queue->size() < 4 || *abort_rc; });  // back-pressure: a few blocks in flight
            queue->push_back(b);
        }
        cv->notify_all();
    }
    void write(int s, const uint8_t *p, int64_t n) {
        while (n) {
            const int64_t room = bufsize - (int64_t)sbuf[s].size();
            const int64_t take = room < n ? room : n;
            sbuf[s].insert(sbuf[s].end(), p, p + take);
            p += take;
            n -= take;
            if ((int64_t)sbuf[s].size() == bufsize) flush(s);
        }
    }
};

// put_literal (src/rzip.c:213-227): per <= 0xFFFF piece a {00, len} header on stream 0, then its bytes on stream 1
void put_literal(BlockCutter &w, const uint8_t *buf, int64_t from, int64_t to) {
    do {
        int64_t len = to - from;
        if (len > 0xFFFF) len = 0xFFFF;
        const uint8_t hdr[3] = { 0, (uint8_t)len, (uint8_t)(len >> 8) };
        w.write(0, hdr, 3);
        if (len) w.write(1, buf + from, len);
        from += len;
    } while (to > from);
}

// put_match (src/rzip.c:179-194): {01, len, dist} per <= 0xFFFF piece, dist = p - offset
void put_match(BlockCutter &w, int cb, int64_t p, int64_t ofs, int64_t len) {
    const int64_t dist = p - ofs;
    do {
        const int64_t n = len > 0xFFFF ? 0xFFFF : len;
        uint8_t rec[3 + 8] = { 1, (uint8_t)n, (uint8_t)(n >> 8) };
        for (int i = 0; i < cb; i++) rec[3 + i] = (uint8_t)((uint64_t)dist >> (8 * i));
        w.write(0, rec, 3 + cb);
        len -= n;
    } while (len);
}

// state of the chunk in flight, driven by mrz_rzip_chunk's progress calls
struct ChunkFeed {
    mrz_ctx *ctx;
    const uint8_t *buf;
    int64_t n;
    int cb;
    BlockCutter *cut;
    int64_t ev_done = 0, lit_from = 0;  // matches encoded so far; end of the last encoded match
    std::vector<mrz_match> ev;
    const int *abort_rc;
};

int chunk_progress(void *user, int64_t n_events, int64_t last_match, int chunk_done) {
    ChunkFeed *f = (ChunkFeed *)user;
    if (*f->abort_rc) return 1;
    try {
        if (n_events > f->ev_done) {
            const int64_t cnt = n_events - f->ev_done;
            f->ev.resize((size_t)cnt);
            if (mrz_fetch_events(f->ctx, f->ev_done, cnt, f->ev.data())) return 1;
            f->cut->info.input_final = last_match;
            for (int64_t i = 0; i < cnt; i++) {  // hash_search's emit, src/rzip.c:593-595
                const mrz_match &e = f->ev[(size_t)i];
                if (f->lit_from < e.p) put_literal(*f->cut, f->buf, f->lit_from, e.p);
                put_match(*f->cut, f->cb, e.p, e.ofs, e.len);
                f->lit_from = e.p + e.len;
            }
            f->ev_done = n_events;
        }
        (void)chunk_done;
    } catch (const std::bad_alloc &) {
        return 1;
    }
    return 0;
}

}  // namespace

static int rzip_pipeline_impl(const mrz_control *ctl, const uint8_t *in, int64_t n, mrz_block_fn fn, void *user,
                              mrz_stats *stats, uint8_t *md5_out) {
    int64_t bufsize = 0;
    const int64_t max_chunk = mrz_plan(ctl, n, &bufsize);

    uint8_t md5[16];
    std::thread hasher([&]() {  // whole-file hash beside everything else (src/rzip.c:1069-1090)
        Md5 h;
        h.update(in, (size_t)n);
        h.finish(md5);
    });
    mrz_join_guard hasher_guard(hasher);

    // consumer: the LZ4 gate (on its own ctx / stream) and the caller's function, block by block, in flush order
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Block *> queue;
    bool producer_done = false;
    int consumer_rc = 0;
    mrz_ctx *gate = nullptr;
    int rc = MRZ_OK;
    if (ctl->lz4_test) rc = mrz_open(&gate, ctl->device, ctl->rzip_compression_level, 0);
    std::thread consumer([&]() {
        for (;;) {
            Block *b = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !queue.empty() || producer_done; });
                if (queue.empty()) break;
                b = queue.front();
                queue.pop_front();
            }
            cv.notify_all();
            if (!consumer_rc) {
                b->info.lz4_verdict = -1;
                int r = 0;
                if (gate && (int64_t)b->payload.size() >= 64) {  // compthread: `c_len >= 64`, src/stream.c:1147
                    int verdict = 0;
                    r = mrz_lz4_compresses(gate, b->payload.data(), (int64_t)b->payload.size(), MRZ_MEM_HOST,
                                           ctl->threshold > 0 ? ctl->threshold : 100, &verdict);
                    b->info.lz4_verdict = verdict;
                }
                if (!r) r = fn(user, &b->info, b->payload.data(), (int64_t)b->payload.size());
                if (r) {
                    std::lock_guard<std::mutex> lk(mu);
                    consumer_rc = r;
                }
                if (r) cv.notify_all();
            }
            delete b;
        }
    });
    // (the same for the consumer, which first has to be told that nothing more is coming)
    struct consumer_guard_t {
        std::mutex &mu;
        std::condition_variable &cv;
        bool &done;
        std::thread &t;
        ~consumer_guard_t() {
            {
                std::lock_guard<std::mutex> lk(mu);
                done = true;
            }
            cv.notify_all();
            if (t.joinable()) t.join();
        }
    } consumer_guard{ mu, cv, producer_done, consumer };

    mrz_ctx *ctx = nullptr;
    if (!rc) rc = mrz_open(&ctx, ctl->device, ctl->rzip_compression_level, max_chunk < n ? max_chunk : n);
    mrz_stats total;
    memset(&total, 0, sizeof(total));
    int64_t victim_round = 0, left = n, pos = 0;
    int pass = 0;
    BlockCutter cut;
    cut.bufsize = bufsize;
    cut.mu = &mu;
    cut.cv = &cv;
    cut.queue = &queue;
    cut.abort_rc = &consumer_rc;
    try {
        while (!rc && (!pass || left > 0)) {  // chunk loop, src/rzip.c:915-1061
            const int64_t csz = max_chunk < left ? max_chunk : left;
            const int cb = mrz_chunk_bytes(csz);
            memset(&cut.info, 0, sizeof(cut.info));
            cut.info.chunk_index = pass;
            cut.info.chunk_bytes = cb;
            cut.info.eof = csz == left;
            cut.info.chunk_size = csz;
            cut.info.first_of_chunk = 1;
            cut.info.lz4_verdict = -1;
            ChunkFeed feed;
            feed.ctx = ctx;
            feed.buf = in + pos;
            feed.n = csz;
            feed.cb = cb;
            feed.cut = &cut;
            feed.abort_rc = &consumer_rc;
            mrz_set_progress(ctx, chunk_progress, &feed);
            mrz_chunk_result res;
            rc = mrz_rzip_chunk(ctx, in + pos, csz, MRZ_MEM_HOST, cb, &victim_round, &res);
            mrz_set_progress(ctx, nullptr, nullptr);
            if (rc) break;
            // the tail of hash_search (src/rzip.c:619,664-665): trailing literal, terminator, CRC (most significant
            // byte first); then close_stream_out flushes both streams, even when empty (src/stream.c:1623-1648)
            cut.info.input_final = csz;
            if (feed.lit_from < csz) put_literal(cut, in + pos, feed.lit_from, csz);
            const uint8_t term[7] = { 0, 0, 0, (uint8_t)(res.crc32 >> 24), (uint8_t)(res.crc32 >> 16),
                                      (uint8_t)(res.crc32 >> 8), (uint8_t)res.crc32 };
            cut.write(0, term, 7);
            cut.flush(0);
            cut.flush(1);
            total.inserts += res.stats.inserts;
            total.literals += res.stats.literals;
            total.literal_bytes += res.stats.literal_bytes;
            total.matches += res.stats.matches;
            total.match_bytes += res.stats.match_bytes;
            total.tag_hits += res.stats.tag_hits;
            total.tag_misses += res.stats.tag_misses;
            pos += csz;
            left -= csz;
            pass++;
        }
    } catch (const std::bad_alloc &) {
        rc = MRZ_E_NOMEM;
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        producer_done = true;
    }
    cv.notify_all();
    consumer.join();
    hasher.join();
    for (Block *b : queue) delete b;
    if (ctx) mrz_close(ctx);
    if (gate) mrz_close(gate);
    if (consumer_rc) rc = consumer_rc;
    if (rc) return rc;
    if (stats) *stats = total;
    if (md5_out) memcpy(md5_out, md5, 16);
    return MRZ_OK;
}

extern "C" int mrz_rzip_pipeline(const mrz_control *ctl, const void *in_v, int64_t n, mrz_block_fn fn, void *user,
                                 mrz_stats *stats, uint8_t *md5_out) {
    if (!ctl || !fn || n < 0 || (n > 0 && !in_v)) return MRZ_E_ARG;
    if (ctl->rzip_compression_level < 1 || ctl->rzip_compression_level > 9) return MRZ_E_ARG;
    try {
        return rzip_pipeline_impl(ctl, (const uint8_t *)in_v, n, fn, user, stats, md5_out);
    } catch (const std::bad_alloc &) {
        return MRZ_E_NOMEM;
    }
}

// ---- decompress side: `mrzip -d` of a -n archive (runzip_fd, src/runzip.c:332-437) -----------------
namespace {

int64_t peek_le(const uint8_t *p, int nbytes) {
    uint64_t v = 0;
    for (int i = 0; i < nbytes; i++) v |= (uint64_t)p[i] << (8 * i);
    return (int64_t)v;
}

// follows one stream's block chain and concatenates the payloads (fill_buffer, src/stream.c:1412-1571);
// only CTYPE_NONE (3) blocks: the back-end codecs stay host code outside this library
int gather_stream(const uint8_t *mrz, int64_t n, int64_t initial_pos, int64_t head_at, int cb, std::vector<uint8_t> &dst,
                  int64_t *end_max) {
    int64_t at = head_at;
    for (;;) {
        if (at + 1 + 3 * cb > n) return MRZ_E_CORRUPT;
        const int ctype = mrz[at];
        const int64_t c_len = peek_le(mrz + at + 1, cb), u_len = peek_le(mrz + at + 1 + cb, cb);
        const int64_t next = peek_le(mrz + at + 1 + 2 * cb, cb);
        if (ctype != 3) return MRZ_E_UNSUPPORTED;
        if (c_len != u_len || c_len < 0) return MRZ_E_CORRUPT;
        const int64_t pay = at + 1 + 3 * cb;
        if (c_len > n - pay) return MRZ_E_CORRUPT;  // (compared without adding: a length field may be 2^63 - 1)
        dst.insert(dst.end(), mrz + pay, mrz + pay + c_len);
        if (pay + c_len > *end_max) *end_max = pay + c_len;
        if (!next) return MRZ_OK;
        if (next < 0 || next > n - initial_pos || initial_pos + next <= at) return MRZ_E_CORRUPT;  // chains only run forward
        at = initial_pos + next;
    }
}

}  // namespace

static int runzip_buffer_impl(int device, const void *mrz_v, int64_t n, void **out, int64_t *out_len) {
    if (!mrz_v || !out || !out_len) return MRZ_E_ARG;
    const uint8_t *mrz = (const uint8_t *)mrz_v;
    if (n < 20 || memcmp(mrz, "MRZI", 4)) return MRZ_E_CORRUPT;  // read_magic, src/mrzip.c:225-321
    if (mrz[15]) return MRZ_E_UNSUPPORTED;                        // encrypted
    const int64_t expected = peek_le(mrz + 6, 8);
    const int hash_code = mrz[14];
    if (hash_code != 0 && hash_code != 1) return MRZ_E_UNSUPPORTED;  // MD5 (default) or CRC only
    // an archive written to STDOUT in several chunks carries no size (src/mrzip.c:137-140): grow chunk by chunk
    const bool size_known = expected > 0;
    int64_t cap = size_known ? expected : 0;
    uint8_t *res = (uint8_t *)malloc((size_t)(cap > 0 ? cap : 1));
    if (!res) return MRZ_E_NOMEM;
    mrz_ctx *ctx = nullptr;
    int rc = mrz_open(&ctx, device, 7, 0);
    int64_t at = 20 + mrz[19], total = 0;
    std::vector<uint8_t> s0, s1;
    while (!rc) {  // runzip_chunk, src/runzip.c:226-330
        if (at + 2 > n) {
            rc = MRZ_E_CORRUPT;
            break;
        }
        const int cb = mrz[at], eof = mrz[at + 1];
        if (cb < 1 || cb > 8 || at + 2 + cb > n) {
            rc = MRZ_E_CORRUPT;
            break;
        }
        at += 2 + cb;  // chunk_bytes, eof, chunk size (the field is chunk_bytes wide: a chunk below one page does
                       // not fit its own size there, src/stream.c:779-780,1224 -- so it is not used for sizing)
        const int64_t initial_pos = at;
        int64_t end_max = initial_pos + 2 * (1 + 3 * cb);
        s0.clear();
        s1.clear();
        rc = gather_stream(mrz, n, initial_pos, initial_pos, cb, s0, &end_max);
        if (!rc) rc = gather_stream(mrz, n, initial_pos, initial_pos + 1 + 3 * cb, cb, s1, &end_max);
        if (rc) break;
        if (!size_known) {
            // bytes this chunk decodes to: the lengths of its records (host side: 3 or 3 + cb bytes each)
            int64_t need = 0, i = 0;
            const int64_t n0 = (int64_t)s0.size();
            bool ended = false;
            while (i + 3 <= n0) {
                const int64_t len = s0[(size_t)i + 1] | (int64_t)s0[(size_t)i + 2] << 8;
                if (s0[(size_t)i] == 0) {
                    if (len == 0) {
                        ended = true;
                        break;
                    }
                    i += 3;
                } else
                    i += 3 + cb;
                need += len;
            }
            if (!ended) {
                rc = MRZ_E_CORRUPT;
                break;
            }
            if (total + need > cap) {
                uint8_t *r2 = (uint8_t *)realloc(res, (size_t)(total + need > 0 ? total + need : 1));
                if (!r2) {
                    rc = MRZ_E_NOMEM;
                    break;
                }
                res = r2;
                cap = total + need;
            }
        }
        int64_t got = 0;
        uint32_t crc_calc = 0, crc_stored = 0;
        rc = mrz_runzip_chunk(ctx, s0.data(), (int64_t)s0.size(), s1.data(), (int64_t)s1.size(), MRZ_MEM_HOST, cb,
                              res + total, MRZ_MEM_HOST, cap - total, &got, &crc_calc, &crc_stored);
        if (rc == MRZ_E_ARG) rc = MRZ_E_CORRUPT;  // more output than the header promised
        if (rc) break;
        if (!hash_code && crc_calc != crc_stored) {  // "Bad checksum", src/runzip.c:317-320 (only without a hash)
            rc = MRZ_E_CORRUPT;
            break;
        }
        total += got;
        at = end_max;
        if (eof) break;
    }
    if (ctx) mrz_close(ctx);
    if (!rc && size_known && total != expected) rc = MRZ_E_CORRUPT;
    if (!rc && hash_code == 1) {  // src/runzip.c:384-412
        uint8_t d[16];
        Md5 h;
        h.update(res, (size_t)total);
        h.finish(d);
        if (at + 16 > n || memcmp(d, mrz + at, 16)) rc = MRZ_E_CORRUPT;
    }
    if (rc) {
        free(res);
        return rc;
    }
    *out = res;
    *out_len = total;
    return MRZ_OK;
}

extern "C" int mrz_runzip_buffer(int device, const void *mrz_v, int64_t n, void **out, int64_t *out_len) {
    try {
        return runzip_buffer_impl(device, mrz_v, n, out, out_len);
    } catch (const std::bad_alloc &) {  // the header promises that the library never takes the host process down
        return MRZ_E_NOMEM;
    } catch (const std::length_error &) {
        return MRZ_E_CORRUPT;
    }
}
