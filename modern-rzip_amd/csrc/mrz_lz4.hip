// mrz_lz4.hip -- the per-block LZ4 compressibility gate.
//
// Replaces lz4_compresses (src/stream.c:1685-1733), which asks "what would
// LZ4_compress_default(buf, tmp, in_len, in_len + 1) return?" for a growing
// prefix of each stream block and never looks at the compressed bytes.  Only
// the SIZE matters, so the kernel runs the LZ4 fast-compressor state machine
// (liblz4 1.9.3 LZ4_compress_fast, acceleration 1: single-probe hash table,
// skip-strength 6, one-position look-back, immediate re-test after a match)
// and counts output bytes without producing them.
//
// One wavefront per block, hash table (8192 x u32) in LDS.  The state machine
// is sequential, the steps are wave-wide:
//   * search: 64 consecutive probes at once (positions from the skip schedule by
//     a wave prefix sum); "what did the table hold when probe i ran" is the old
//     LDS value unless an earlier probe of the batch hashed to the same cell
//     (forwarded in registers); ballot picks the first hit / end-of-input; only
//     the probes that really ran commit their table writes (last writer per cell);
//   * catch-up and match extension: 64 lanes x 1 B backwards, 64 lanes x 16 B
//     forwards with ballot + ffs;
// many blocks run concurrently (one per wave; the gate sees every block of a
// chunk's two streams).  Bit-exact with liblz4 1.9.3 sizes.
//
// Bound: latency of dependent L2 reads per match; HBM traffic = bytes tested.
#include <string.h>

#include "mrz_ctx.h"
#include "mrz_device.h"

#define MRZ_LZ_MFLIMIT 12
#define MRZ_LZ_LASTLITERALS 5
#define MRZ_LZ_MINLEN 13
#define MRZ_LZ_MAXDIST 65535
#define MRZ_LZ_64K_LIMIT (65536 + 11)
#define MRZ_LZ_STREAM_MIN (10 * 1048576)
#ifndef MRZ_LZ_FIRST_WIDTH
#define MRZ_LZ_FIRST_WIDTH 16  // probes in the first batch of a search
#endif

__device__ __forceinline__ uint32_t mrz_lz_hash(const uint8_t *p, bool small) {
    if (small) return (mrz_ld4(p) * 2654435761u) >> (32 - 13);
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return (uint32_t)(((v << 24) * 889523592379ULL) >> (64 - 12));
}

// number of equal bytes of a[0..) and b[0..) with a limited to [.., alimit)
__device__ static int64_t mrz_lz_count(const uint8_t *src, int64_t a, int64_t b, int64_t alimit, int lane) {
    const int64_t maxf = alimit - a;
    if (maxf <= 0) return 0;
    for (int64_t base = 0;; base += 1024) {
        const int64_t off = base + lane * 16;
        int lane_len = 0;
        bool full = false;
        if (off < maxf) {
            const int64_t rem = maxf - off;
            const int lim = rem < 16 ? (int)rem : 16;
            int d;
            if (rem >= 16)
                d = mrz_first_diff16(mrz_ld16(src + a + off), mrz_ld16(src + b + off));
            else {
                d = 0;
                while (d < lim && src[a + off + d] == src[b + off + d]) d++;
            }
            lane_len = d < lim ? d : lim;
            full = lane_len == 16;
        }
        const mrz_u64 stop = __ballot(!full);
        if (stop) {
            const int fl = __ffsll((long long)stop) - 1;
            return base + (int64_t)fl * 16 + mrz_lane_read(lane_len, fl);
        }
    }
}

// what LZ4_compress_default(src, dst, n, cap) returns (0 = does not fit)
__device__ static int mrz_lz4_size_wave(const uint8_t *__restrict__ src, int n, int cap, uint32_t *tab, int lane) {
    if (n <= 0) return n == 0 && cap > 0 ? 1 : 0;
    const bool limited = cap < n + n / 255 + 16;
    const bool small = n < MRZ_LZ_64K_LIMIT;
    for (int i = lane; i < 8192; i += 64) tab[i] = 0;
    const int64_t mfl1 = (int64_t)n - MRZ_LZ_MFLIMIT + 1;
    const int64_t mlimit = (int64_t)n - MRZ_LZ_LASTLITERALS;
    const int64_t olimit = cap;
    int64_t ip = 0, anchor = 0, op = 0;
    bool to_tail = n < MRZ_LZ_MINLEN;

    if (!to_tail) {
        if (lane == 0) tab[mrz_lz_hash(src, small)] = 0;
        ip = 1;
    }
    while (!to_tail) {
        // ---- search: probes k = 0,1,2,.. at positions given by the skip schedule
        int64_t pos0 = ip;
        int k0 = 0;
        int64_t match = 0;
        bool found = false;
        // compressible data finds its match within a few probes: the first batch of a search is 16 probes wide
        // (the all-pairs forwarding below costs `width` shuffle rounds), later ones 64
        int width = MRZ_LZ_FIRST_WIDTH;
        while (true) {
            const int k = k0 + lane;
            const bool active = lane < width;
            const int my_step = !active ? 0 : (k == 0 ? 1 : (63 + k) >> 6);
            const int incl = mrz_wave_incl_sum(my_step, lane);
            const int64_t pos = pos0 + (incl - my_step);
            const int64_t nxt = pos + my_step;
            const bool runs = active && nxt <= mfl1;  // this probe gets past `if (fwd > mflimitPlusOne) goto _last_literals`
            uint32_t h = 0xffffffffu;  // inactive lanes never equal a real cell index
            if (runs) h = mrz_lz_hash(src + pos, small);
            // table value seen by this probe: old cell unless an earlier probe of the batch wrote it
            uint32_t mi = runs ? tab[h] : 0;
            int next_same = 64;
            for (int j = 0; j < width; j++) {
                const uint32_t hj = (uint32_t)__shfl((int)h, j, MRZ_WAVE);
                const uint32_t pj = (uint32_t)__shfl((int)(uint32_t)pos, j, MRZ_WAVE);
                if (hj == h) {
                    if (j < lane) mi = pj;
                    if (j > lane && next_same == 64) next_same = j;
                }
            }
            bool hit = false;
            if (runs) {
                const bool near = small || ((int64_t)mi + MRZ_LZ_MAXDIST >= pos);
                hit = near && mrz_ld4(src + mi) == mrz_ld4(src + pos);
            }
            const mrz_u64 m_end = __ballot(active && !runs);
            const mrz_u64 m_hit = __ballot(hit);
            const int first_end = m_end ? __ffsll((long long)m_end) - 1 : width;
            const int first_hit = m_hit ? __ffsll((long long)m_hit) - 1 : 64;
            // probes [0, last_run] executed their table update
            int last_run;
            if (first_hit < first_end)
                last_run = first_hit;
            else
                last_run = first_end - 1;
            if (lane <= last_run && next_same > last_run) tab[h] = (uint32_t)pos;
            if (first_hit < first_end) {
                ip = mrz_bcast64(pos, first_hit);
                match = (int64_t)(uint32_t)mrz_lane_read((int)mi, first_hit);
                found = true;
                break;
            }
            if (first_end < width) break;  // ran out of input: last literals
            pos0 = mrz_bcast64(nxt, width - 1);
            k0 += width;
            width = 64;
        }
        if (!found) {
            to_tail = true;
            break;
        }
        // ---- catch up (look back over equal bytes) -------------------------
        {
            int64_t room = ip - anchor;
            if (match < room) room = match;  // match > lowLimit(=0)
            int64_t back = 0;
            while (back < room) {
                const int64_t j = back + lane;
                const bool eq = j < room && src[ip - 1 - j] == src[match - 1 - j];
                const mrz_u64 ne = __ballot(!eq);
                if (ne) {
                    back += __ffsll((long long)ne) - 1;
                    break;
                }
                back += 64;
            }
            ip -= back;
            match -= back;
        }
        // ---- literal run accounting ---------------------------------------
        {
            const int64_t lit = ip - anchor;
            op += 1;  // token
            if (limited && op + lit + (2 + 1 + MRZ_LZ_LASTLITERALS) + lit / 255 > olimit) return 0;
            if (lit >= 15) op += (lit - 15) / 255 + 1;
            op += lit;
        }
        // ---- match(es): _next_match loop ------------------------------------
        while (true) {
            op += 2;  // offset
            int64_t mc = mrz_lz_count(src, ip + 4, match + 4, mlimit, lane);
            ip += mc + 4;
            if (limited && op + (1 + MRZ_LZ_LASTLITERALS) + (mc + 240) / 255 > olimit) return 0;
            if (mc >= 15) {
                mc -= 15;
                op += mc / 255 + 1;
            }
            anchor = ip;
            if (ip >= mfl1) {
                to_tail = true;
                break;
            }
            // fill table with ip-2, then test ip immediately
            const uint32_t h2 = mrz_lz_hash(src + ip - 2, small);
            const uint32_t h = mrz_lz_hash(src + ip, small);
            uint32_t mi = 0;
            if (lane == 0) {
                tab[h2] = (uint32_t)(ip - 2);
                mi = tab[h];
                tab[h] = (uint32_t)ip;
            }
            mi = (uint32_t)mrz_lane_read((int)mi, 0);
            if ((small || (int64_t)mi + MRZ_LZ_MAXDIST >= ip) && mrz_ld4(src + mi) == mrz_ld4(src + ip)) {
                op += 1;  // token of a zero-literal sequence
                match = mi;
                continue;
            }
            ip++;
            break;
        }
    }
    // ---- last literals ------------------------------------------------------
    const int64_t last = (int64_t)n - anchor;
    if (limited && op + last + 1 + (last + 255 - 15) / 255 > olimit) return 0;
    op += 1;
    if (last >= 15) op += (last - 15) / 255 + 1;
    op += last;
    return (int)op;
}

// sizes[i] = LZ4_compress_default size of block i with dst capacity lens[i] + 1
__global__ __launch_bounds__(64) void mrz_lz4_sizes_kernel(const uint8_t *const *__restrict__ bufs,
                                                           const int *__restrict__ lens, int count,
                                                           int *__restrict__ sizes) {
    __shared__ uint32_t tab[8192];
    const int i = blockIdx.x;
    if (i >= count) return;
    const int r = mrz_lz4_size_wave(bufs[i], lens[i], lens[i] + 1, tab, threadIdx.x);
    if (threadIdx.x == 0) sizes[i] = r;
}

// lz4_compresses (src/stream.c:1685-1733) for block blockIdx.x
__global__ __launch_bounds__(64) void mrz_lz4_gate_kernel(const uint8_t *const *__restrict__ bufs,
                                                          const int64_t *__restrict__ lens, int count, int threshold,
                                                          int *__restrict__ results) {
    __shared__ uint32_t tab[8192];
    const int i = blockIdx.x;
    if (i >= count) return;
    const uint8_t *s_buf = bufs[i];
    int test_len = (int)lens[i];
    int in_len = test_len < MRZ_LZ_STREAM_MIN ? test_len : MRZ_LZ_STREAM_MIN;
    int buftest = in_len;
    double pct = 101;
    while (test_len > 0) {
        const int r = mrz_lz4_size_wave(s_buf, in_len, in_len + 1, tab, threadIdx.x);
        if (r > 0) {
            pct = 100 * ((double)r / (double)in_len);
            if (r < in_len * ((double)threshold / 100)) break;
        }
        test_len -= in_len;
        if (test_len > 0) {
            buftest += in_len;
            if (buftest < MRZ_LZ_STREAM_MIN) buftest <<= 1;
            in_len = test_len < buftest ? test_len : buftest;
        }
    }
    if (threadIdx.x == 0) results[i] = (int)(pct > threshold ? 0 : pct < 1 ? pct + 1 : pct);
}

// ---- host side -----------------------------------------------------------------
// scratch layout: ptrs[count] (8 B) | lens[count] (8 B) | results[count] (4 B, padded) | staged block bytes
static int mrz_lz4_prepare(mrz_ctx *ctx, const void *const *bufs, const int64_t *lens, int count, int where,
                           const uint8_t ***d_ptrs, int64_t **d_lens, int **d_res) {
    int64_t total = 0;
    for (int i = 0; i < count; i++) {
        if (lens[i] < 0 || lens[i] > 0x7E000000ll || (lens[i] && !bufs[i])) return MRZ_E_ARG;
        total += (lens[i] + 31) & ~15ll;
    }
    const int64_t hdr = (int64_t)count * 16 + (((int64_t)count * 4 + 15) & ~15ll);
    const int64_t need = hdr + (where == MRZ_MEM_HOST ? total : 0) + 64;
    if (need > ctx->lz4_scratch_cap || !ctx->lz4_scratch) {
        if (ctx->lz4_scratch) hipFree(ctx->lz4_scratch);
        ctx->lz4_scratch = nullptr;
        ctx->lz4_scratch_cap = 0;
        void *p = nullptr;
        if (hipMalloc(&p, (size_t)need) != hipSuccess) return MRZ_E_NOMEM;
        ctx->lz4_scratch = p;
        ctx->lz4_scratch_cap = need;
    }
    uint8_t *base = (uint8_t *)ctx->lz4_scratch;
    *d_ptrs = (const uint8_t **)base;
    *d_lens = (int64_t *)(base + (int64_t)count * 8);
    *d_res = (int *)(base + (int64_t)count * 16);
    uint8_t *d_data = base + hdr;
    const uint8_t **h_ptrs = (const uint8_t **)malloc((size_t)count * sizeof(void *));
    if (!h_ptrs) return MRZ_E_NOMEM;
    hipError_t e = hipSuccess;
    int64_t off = 0;
    for (int i = 0; i < count && e == hipSuccess; i++) {
        if (where == MRZ_MEM_HOST) {
            h_ptrs[i] = d_data + off;
            if (lens[i]) e = hipMemcpyAsync(d_data + off, bufs[i], (size_t)lens[i], hipMemcpyHostToDevice, ctx->stream);
            off += (lens[i] + 31) & ~15ll;
        } else
            h_ptrs[i] = (const uint8_t *)bufs[i];
    }
    if (e == hipSuccess) e = hipMemcpyAsync(*d_ptrs, h_ptrs, (size_t)count * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(*d_lens, lens, (size_t)count * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // h_ptrs / lens are host temporaries
    free(h_ptrs);
    if (e != hipSuccess) {
        ctx->last_err = e;
        return MRZ_E_HIP;
    }
    return MRZ_OK;
}

extern "C" int mrz_lz4_compresses_batch(mrz_ctx *ctx, const void *const *bufs, const int64_t *lens, int count,
                                        int where, int threshold, int *results) {
    if (!ctx || count < 0 || (count && (!bufs || !lens || !results))) return MRZ_E_ARG;
    if (where != MRZ_MEM_HOST && where != MRZ_MEM_DEVICE) return MRZ_E_ARG;
    if (!count) return MRZ_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint8_t **d_ptrs;
    int64_t *d_lens;
    int *d_res;
    int rc = mrz_lz4_prepare(ctx, bufs, lens, count, where, &d_ptrs, &d_lens, &d_res);
    if (rc) return rc;
    hipLaunchKernelGGL(mrz_lz4_gate_kernel, dim3((unsigned)count), dim3(64), 0, ctx->stream,
                       (const uint8_t *const *)d_ptrs, (const int64_t *)d_lens, count, threshold, d_res);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(results, d_res, (size_t)count * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MRZ_OK;
}

extern "C" int mrz_lz4_compresses(mrz_ctx *ctx, const void *s_buf, int64_t s_len, int where, int threshold,
                                  int *result) {
    const void *bufs[1] = { s_buf };
    return mrz_lz4_compresses_batch(ctx, bufs, &s_len, 1, where, threshold, result);
}

extern "C" int mrz_lz4_sizes(mrz_ctx *ctx, const void *const *bufs, const int *lens, int count, int where,
                             int *sizes) {
    if (!ctx || count < 0 || (count && (!bufs || !lens || !sizes))) return MRZ_E_ARG;
    if (where != MRZ_MEM_HOST && where != MRZ_MEM_DEVICE) return MRZ_E_ARG;
    if (!count) return MRZ_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int64_t *l64 = (int64_t *)malloc((size_t)count * 8);
    if (!l64) return MRZ_E_NOMEM;
    for (int i = 0; i < count; i++) l64[i] = lens[i];
    const uint8_t **d_ptrs;
    int64_t *d_lens;
    int *d_res;
    int rc = mrz_lz4_prepare(ctx, bufs, l64, count, where, &d_ptrs, &d_lens, &d_res);
    free(l64);
    if (rc) return rc;
    // the sizes kernel wants 32-bit lengths: reuse the results area after a conversion on the host side
    int *h32 = (int *)malloc((size_t)count * sizeof(int));
    if (!h32) return MRZ_E_NOMEM;
    memcpy(h32, lens, (size_t)count * sizeof(int));
    int *d_len32 = (int *)d_lens;  // 8 B per entry reserved, 4 B used
    hipError_t e = hipMemcpyAsync(d_len32, h32, (size_t)count * sizeof(int), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    free(h32);
    if (e != hipSuccess) {
        ctx->last_err = e;
        return MRZ_E_HIP;
    }
    hipLaunchKernelGGL(mrz_lz4_sizes_kernel, dim3((unsigned)count), dim3(64), 0, ctx->stream,
                       (const uint8_t *const *)d_ptrs, (const int *)d_len32, count, d_res);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(sizes, d_res, (size_t)count * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MRZ_OK;
}
