// mrz_runzip.hip -- the decoder of the two rzip streams of a chunk.
//
// Restates the record loop of runzip_chunk (src/runzip.c:277-308) with unzip_literal (:120-157)
// and unzip_match (:159-207): stream 0 is a sequence of records {head:u8, len:u16le} (+ a
// chunk_bytes-wide distance when head != 0), ended by a zero-length literal and the stored
// CRC-32 (most significant byte first, src/rzip.c:662-665); a literal copies `len` bytes of
// stream 1, a match copies `len` bytes of history starting `dist` back, the first
// min(len, dist) history bytes repeated when len > dist (:182-199).
//
// Four passes on the GPU:
//   1. parse, speculative: stream 0 is cut into 1 KiB tiles; records are 3 or 3 + chunk_bytes
//      bytes, so a tile can only be entered at offsets 0 .. 2 + chunk_bytes.  One thread per
//      (tile, entry offset) walks the tile and notes where it leaves it and what it saw
//      (records, output bytes, literal bytes);
//   2. scan: one workgroup composes the per-tile entry->exit tables (associative), which
//      gives every tile its true entry offset and the running record / output / literal
//      counts in front of it;
//   3. parse, for real: one thread per tile writes its records {out_pos, literal offset or
//      distance} and validates them;
//   4. decode: the output is cut into tiles (8-128 KiB, by record density) claimed in order by persistent workgroups.
//      A match needs its history: bytes of earlier tiles are awaited through per-tile done
//      flags (release / acquire at agent scope -- tiles are written on different XCDs), bytes
//      of the same tile are ordered by workgroup barriers.  A workgroup only ever waits for
//      tiles claimed before its own, i.e. by workgroups that are already running.
// Then the CRC-32 kernel runs over the output.  Bound: HBM for literal-heavy input; on highly
// repetitive input the copy chain (tile k needs tile k-1) is latency-bound.
#include <stdio.h>
#include <string.h>

#include <vector>

#include "mrz_ctx.h"
#include "mrz_device.h"

#define MRZ_UZ_PT 1024        // stream-0 parse tile
#define MRZ_UZ_PHMAX 11       // entry offsets: record size <= 3 + 8
#define MRZ_UZ_END 255        // parse state: terminator seen
#define MRZ_UZ_ERR 254        // parse state: record runs past the end of stream 0
#define MRZ_UZ_TSHIFT_MIN 13  // output tile of the decode pass: 8 KiB (many short records) .. 128 KiB (few long ones)
#define MRZ_UZ_TSHIFT_MAX 17
#ifndef MRZ_UZ_DEC_THREADS
#define MRZ_UZ_DEC_THREADS 512
#endif
#define MRZ_UZ_THREADS 256
#define MRZ_UZ_SCAN_THREADS 128  // run tables of the scan live in LDS: 128 x 12 x 25 B
#define MRZ_UZ_MATCH (1ull << 63)
#define MRZ_UZ_SPIN_LIMIT (1ll << 30)

struct mrz_urec {
    int64_t out_pos;          // first output byte of the record
    unsigned long long src;   // literal: offset in stream 1; match: MRZ_UZ_MATCH | distance
};

struct mrz_uz_hdr {
    int64_t nrec, out_total, lit_total, term_pos;
    int final_state;
    int error;                // 1 = invalid record (distance 0 / beyond history, empty match), 2 = a wait gave up
    unsigned long long next_tile;   // decode pass: tile claim counter
    unsigned long long prefix_done; // decode pass: every tile below this index is complete
};

struct mrz_uz_tile {          // per parse tile and entry offset
    unsigned char exit_state[MRZ_UZ_PHMAX + 1];
    unsigned nrec[MRZ_UZ_PHMAX + 1];
    int64_t outb[MRZ_UZ_PHMAX + 1];
    int64_t litb[MRZ_UZ_PHMAX + 1];
};

struct mrz_uz_base {          // per parse tile, after the scan
    int64_t rec_base, out_base, lit_base;
    int entry;                // true entry offset, or MRZ_UZ_END / MRZ_UZ_ERR
};

// walk one parse tile from `pos`; sink(kind, len, dist_pos) is called per record when WRITE
template <bool WRITE>
__device__ __forceinline__ int mrz_uz_walk(const uint8_t *__restrict__ s0, int64_t s0_len, int cb, int64_t pos,
                                           int64_t tile_end, unsigned *nrec_out, int64_t *outb_out, int64_t *litb_out,
                                           mrz_urec *__restrict__ rec, int64_t rec_at, int64_t out_at, int64_t lit_at,
                                           mrz_uz_hdr *hdr) {
    unsigned nrec = 0;
    int64_t outb = 0, litb = 0;
    int state = -1;
    while (pos < tile_end) {
        if (pos + 3 > s0_len) {
            state = MRZ_UZ_ERR;
            break;
        }
        const int head = s0[pos];
        const int64_t len = (int64_t)s0[pos + 1] | ((int64_t)s0[pos + 2] << 8);
        if (!head && !len) {  // terminator; the stored CRC follows
            state = pos + 7 <= s0_len ? MRZ_UZ_END : MRZ_UZ_ERR;
            if (WRITE && state == MRZ_UZ_END) {
                hdr->term_pos = pos;
                rec[rec_at + nrec].out_pos = out_at + outb;
                rec[rec_at + nrec].src = 0;
            }
            break;
        }
        if (!head) {
            if (WRITE) {
                rec[rec_at + nrec].out_pos = out_at + outb;
                rec[rec_at + nrec].src = (unsigned long long)(lit_at + litb);
            }
            litb += len;
            pos += 3;
        } else {
            if (pos + 3 + cb > s0_len) {
                state = MRZ_UZ_ERR;
                break;
            }
            if (WRITE) {
                unsigned long long dist = 0;
                for (int k = 0; k < cb; k++) dist |= (unsigned long long)s0[pos + 3 + k] << (8 * k);
                // unzip_match: n = MIN(len, offset) < 1 is "corrupt archive" (:176-177); history starts at 0
                if (dist < 1 || dist > (unsigned long long)(out_at + outb) || len < 1) hdr->error = 1;
                rec[rec_at + nrec].out_pos = out_at + outb;
                rec[rec_at + nrec].src = MRZ_UZ_MATCH | dist;
            }
            pos += 3 + cb;
        }
        outb += len;
        nrec++;
    }
    *nrec_out = nrec;
    *outb_out = outb;
    *litb_out = litb;
    return state >= 0 ? state : (int)(pos - tile_end);
}

__global__ __launch_bounds__(MRZ_UZ_THREADS) void mrz_uz_parse1_kernel(const uint8_t *__restrict__ s0, int64_t s0_len,
                                                                       int cb, int64_t ntiles,
                                                                       mrz_uz_tile *__restrict__ tiles) {
    const int ph = 3 + cb;
    const int64_t id = (int64_t)blockIdx.x * MRZ_UZ_THREADS + threadIdx.x;
    const int64_t t = id / ph;
    const int c = (int)(id % ph);
    if (t >= ntiles) return;
    unsigned nrec;
    int64_t outb, litb;
    int64_t tile_end = (t + 1) * MRZ_UZ_PT;
    if (tile_end > s0_len) tile_end = s0_len;
    int st;
    st = mrz_uz_walk<false>(s0, s0_len, cb, t * MRZ_UZ_PT + c, tile_end, &nrec, &outb, &litb, nullptr, 0, 0, 0, nullptr);
    // a walk that reaches the end of stream 0 without a terminator is an error
    if (st < MRZ_UZ_ERR && tile_end == s0_len) st = MRZ_UZ_ERR;
    tiles[t].exit_state[c] = (unsigned char)st;
    tiles[t].nrec[c] = nrec;
    tiles[t].outb[c] = outb;
    tiles[t].litb[c] = litb;
}

// one workgroup: thread i composes the tables of tiles [i*per, (i+1)*per) for every entry state, thread 0 chains the
// runs, then every thread replays its run from its true entry state and writes the per-tile bases
__global__ __launch_bounds__(MRZ_UZ_SCAN_THREADS) void mrz_uz_scan_kernel(const mrz_uz_tile *__restrict__ tiles,
                                                                     int64_t ntiles, int cb,
                                                                     mrz_uz_base *__restrict__ bases,
                                                                     mrz_uz_hdr *__restrict__ hdr) {
    __shared__ unsigned char r_exit[MRZ_UZ_SCAN_THREADS][MRZ_UZ_PHMAX + 1];
    __shared__ int64_t r_nrec[MRZ_UZ_SCAN_THREADS][MRZ_UZ_PHMAX + 1];
    __shared__ int64_t r_outb[MRZ_UZ_SCAN_THREADS][MRZ_UZ_PHMAX + 1];
    __shared__ int64_t r_litb[MRZ_UZ_SCAN_THREADS][MRZ_UZ_PHMAX + 1];
    __shared__ int s_entry[MRZ_UZ_SCAN_THREADS];
    __shared__ int64_t s_rec[MRZ_UZ_SCAN_THREADS], s_out[MRZ_UZ_SCAN_THREADS], s_lit[MRZ_UZ_SCAN_THREADS];
    const int ph = 3 + cb;
    const int i = threadIdx.x;
    const int64_t per = (ntiles + MRZ_UZ_SCAN_THREADS - 1) / MRZ_UZ_SCAN_THREADS;
    const int64_t t0 = (int64_t)i * per;
    int64_t t1 = t0 + per;
    if (t1 > ntiles) t1 = ntiles;
    for (int c = 0; c < ph; c++) {
        int st = c;
        int64_t nr = 0, ob = 0, lb = 0;
        for (int64_t t = t0; t < t1 && st < MRZ_UZ_ERR; t++) {
            nr += tiles[t].nrec[st];
            ob += tiles[t].outb[st];
            lb += tiles[t].litb[st];
            st = tiles[t].exit_state[st];
        }
        r_exit[i][c] = (unsigned char)st;
        r_nrec[i][c] = nr;
        r_outb[i][c] = ob;
        r_litb[i][c] = lb;
    }
    __syncthreads();
    if (i == 0) {
        int st = 0;
        int64_t nr = 0, ob = 0, lb = 0;
        for (int k = 0; k < MRZ_UZ_SCAN_THREADS; k++) {
            s_entry[k] = st;
            s_rec[k] = nr;
            s_out[k] = ob;
            s_lit[k] = lb;
            if (st < MRZ_UZ_ERR && (int64_t)k * per < ntiles) {
                nr += r_nrec[k][st];
                ob += r_outb[k][st];
                lb += r_litb[k][st];
                st = r_exit[k][st];
            }
        }
        hdr->nrec = nr;
        hdr->out_total = ob;
        hdr->lit_total = lb;
        hdr->final_state = ntiles ? st : MRZ_UZ_ERR;
    }
    __syncthreads();
    int st = s_entry[i];
    int64_t nr = s_rec[i], ob = s_out[i], lb = s_lit[i];
    for (int64_t t = t0; t < t1; t++) {
        bases[t].entry = st;
        bases[t].rec_base = nr;
        bases[t].out_base = ob;
        bases[t].lit_base = lb;
        if (st < MRZ_UZ_ERR) {
            nr += tiles[t].nrec[st];
            ob += tiles[t].outb[st];
            lb += tiles[t].litb[st];
            st = tiles[t].exit_state[st];
        }
    }
}

__global__ __launch_bounds__(MRZ_UZ_THREADS) void mrz_uz_parse2_kernel(const uint8_t *__restrict__ s0, int64_t s0_len,
                                                                       int cb, int64_t ntiles,
                                                                       const mrz_uz_base *__restrict__ bases,
                                                                       mrz_urec *__restrict__ rec,
                                                                       mrz_uz_hdr *__restrict__ hdr) {
    const int64_t t = (int64_t)blockIdx.x * MRZ_UZ_THREADS + threadIdx.x;
    if (t >= ntiles) return;
    const mrz_uz_base b = bases[t];
    if (b.entry >= MRZ_UZ_ERR) return;
    int64_t tile_end = (t + 1) * MRZ_UZ_PT;
    if (tile_end > s0_len) tile_end = s0_len;
    unsigned nrec;
    int64_t outb, litb;
    mrz_uz_walk<true>(s0, s0_len, cb, t * MRZ_UZ_PT + b.entry, tile_end, &nrec, &outb, &litb, rec, b.rec_base, b.out_base,
                      b.lit_base, hdr);
}

// workgroup-wide copy of n bytes (both sides arbitrarily aligned): 16 B per thread and step
__device__ __forceinline__ void mrz_uz_copy(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, int64_t n) {
    for (int64_t o = (int64_t)threadIdx.x * 16; o < n; o += (int64_t)MRZ_UZ_DEC_THREADS * 16) {
        if (o + 16 <= n) {
            const uint4 v = mrz_ld16(src + o);
            __builtin_memcpy(dst + o, &v, 16);
        } else
            for (int64_t k = o; k < n; k++) dst[k] = src[k];
    }
}

__device__ __forceinline__ void mrz_uz_acquire() {
#ifdef __HIP_DEVICE_COMPILE__
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}
__device__ __forceinline__ void mrz_uz_release() {
#ifdef __HIP_DEVICE_COMPILE__
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
}

__global__ __launch_bounds__(MRZ_UZ_DEC_THREADS) void mrz_uz_decode_kernel(const mrz_urec *__restrict__ rec, int64_t nrec,
                                                                       const uint8_t *__restrict__ s1,
                                                                       uint8_t *__restrict__ out, int64_t out_total,
                                                                       int64_t ntiles, int tshift,
                                                                       unsigned *__restrict__ done,
                                                                       mrz_uz_hdr *__restrict__ hdr) {
    const int64_t TO = (int64_t)1 << tshift;
    __shared__ long long s_tile, s_first, s_prefix;
    __shared__ int s_fail;
    long long known = 0;  // every tile below this index is known to be complete (and acquired)
    unsigned long long vmask = 0;  // bit i: tile known + i has been seen complete and acquired as well
    while (true) {
        if (threadIdx.x == 0) {
            const long long t = (long long)__hip_atomic_fetch_add(&hdr->next_tile, 1ull, __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_AGENT);
            s_tile = t;
            s_fail = 0;
            if (t < ntiles) {
                // last record that starts at or before the tile's first byte
                const int64_t T0 = (t << tshift);
                int64_t lo = 0, hi = nrec - 1;
                while (lo < hi) {
                    const int64_t mid = (lo + hi + 1) >> 1;
                    if (rec[mid].out_pos <= T0)
                        lo = mid;
                    else
                        hi = mid - 1;
                }
                s_first = lo;
            }
        }
        __syncthreads();
        const long long t = s_tile;
        if (t >= ntiles) return;
        const int64_t T0 = (t << tshift);
        const int64_t T1 = T0 + TO < out_total ? T0 + TO : out_total;
        int64_t written_to = T0;  // bytes of this tile below this are written (maybe not yet ordered)
        bool unordered = false;   // stores since the last barrier
        for (int64_t r = s_first; r < nrec; r++) {
            const int64_t o = rec[r].out_pos;
            if (o >= T1) break;
            const int64_t o_end = rec[r + 1].out_pos;
            const unsigned long long src = rec[r].src;
            const int64_t x0 = o > T0 ? o : T0, x1 = o_end < T1 ? o_end : T1;
            if (x1 <= x0) continue;
            if (!(src & MRZ_UZ_MATCH)) {
                mrz_uz_copy(out + x0, s1 + (int64_t)src + (x0 - o), x1 - x0);
            } else {
                const int64_t dist = (int64_t)(src & ~MRZ_UZ_MATCH);
                const int64_t len = o_end - o, span = len < dist ? len : dist, from = o - dist;
                // history needed by this piece
                const int64_t h0 = span == len ? from + (x0 - o) : from;
                const int64_t h1 = span == len ? from + (x1 - o) : from + span;
                if (h0 < T0) {
                    const long long need_hi = (long long)(((h1 < T0 ? h1 : T0) - 1) >> tshift);
                    const long long need_lo = (long long)(h0 >> tshift) > known ? (long long)(h0 >> tshift) : known;
                    bool seen = need_hi < known;
                    if (!seen && need_hi - known < 64) {
                        const unsigned long long hi_bits = need_hi - known == 63 ? ~0ull : ((1ull << (need_hi - known + 1)) - 1ull);
                        const unsigned long long needmask = hi_bits & ~((1ull << (need_lo - known)) - 1ull);
                        seen = (vmask & needmask) == needmask;
                    }
                    if (!seen) {
                        if (threadIdx.x == 0) {
                            long long spins = 0;
                            // relaxed polls (an acquire load would invalidate the L2 on every iteration); one acquire
                            // fence by every thread follows the barrier below
                            long long p = (long long)__hip_atomic_load(&hdr->prefix_done, __ATOMIC_RELAXED,
                                                                       __HIP_MEMORY_SCOPE_AGENT);
                            for (long long q = (long long)(h0 >> tshift) > p ? (long long)(h0 >> tshift) : p;
                                 q <= need_hi; q++) {
                                while (!__hip_atomic_load(&done[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                                    if (spins++ > MRZ_UZ_SPIN_LIMIT) {
                                        s_fail = 1;
                                        break;
                                    }
                                    __builtin_amdgcn_s_sleep(1);
                                }
                                if (s_fail) break;
                            }
                            s_prefix = p;
                        }
                        __syncthreads();
                        mrz_uz_acquire();
                        if (s_fail) {
                            if (threadIdx.x == 0) hdr->error = 2;
                            return;
                        }
                        // remember the prefix and, in a 64-tile window above it, the single tiles just seen
                        const long long p = s_prefix > known ? s_prefix : known;
                        vmask = p - known >= 64 ? 0ull : vmask >> (p - known);
                        known = p;
                        for (long long qq = need_lo > known ? need_lo : known; qq <= need_hi && qq - known < 64; qq++)
                            vmask |= 1ull << (qq - known);
                        unordered = false;
                    }
                }
                if (h1 > T0 && unordered) {  // history inside this tile: order the earlier stores
                    __syncthreads();
                    unordered = false;
                }
                if (span == len)
                    mrz_uz_copy(out + x0, out + from + (x0 - o), x1 - x0);
                else  // the first `dist` history bytes repeat (src/runzip.c:182-199)
                    for (int64_t x = x0 + threadIdx.x; x < x1; x += MRZ_UZ_DEC_THREADS) out[x] = out[from + (x - o) % span];
            }
            written_to = x1;
            unordered = true;
        }
        (void)written_to;
        __syncthreads();
        if (threadIdx.x == 0) {
            mrz_uz_release();
            __hip_atomic_store(&done[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // after the release fence
            // push the completed prefix forward as far as the flags allow
            unsigned long long p = __hip_atomic_load(&hdr->prefix_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while ((long long)p < ntiles && __hip_atomic_load(&done[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                unsigned long long expect = p;
                if (__hip_atomic_compare_exchange_strong(&hdr->prefix_done, &expect, p + 1, __ATOMIC_RELAXED,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                    p = p + 1;
                else
                    p = expect;
            }
        }
        __syncthreads();
    }
}

static int uz_grow(mrz_ctx *ctx, int64_t want) {
    return mrz_grow(ctx, (uint8_t **)&ctx->rz_scratch, &ctx->rz_scratch_cap, want);
}

extern "C" int mrz_runzip_chunk(mrz_ctx *ctx, const void *s0, int64_t s0_len, const void *s1, int64_t s1_len, int where,
                                int chunk_bytes, void *out, int out_where, int64_t out_cap, int64_t *out_len,
                                uint32_t *crc_calc, uint32_t *crc_stored) {
    if (!ctx || !s0 || s0_len < 7 || s1_len < 0 || (s1_len > 0 && !s1) || chunk_bytes < 1 || chunk_bytes > 8 || !out_len)
        return MRZ_E_ARG;
    if ((where != MRZ_MEM_HOST && where != MRZ_MEM_DEVICE) || (out_where != MRZ_MEM_HOST && out_where != MRZ_MEM_DEVICE))
        return MRZ_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int64_t ntiles = (s0_len + MRZ_UZ_PT - 1) / MRZ_UZ_PT;
    const int64_t max_rec = s0_len / 3 + 2;

    // scratch layout (256-B aligned pieces)
    auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
    const int64_t o_hdr = 0;
    const int64_t o_tiles = o_hdr + al(sizeof(mrz_uz_hdr));
    const int64_t o_bases = o_tiles + al(ntiles * (int64_t)sizeof(mrz_uz_tile));
    const int64_t o_rec = o_bases + al(ntiles * (int64_t)sizeof(mrz_uz_base));
    const int64_t o_s0 = o_rec + al(max_rec * (int64_t)sizeof(mrz_urec));
    const int64_t o_s1 = o_s0 + (where == MRZ_MEM_HOST ? al(s0_len + 64) : 0);
    const int64_t o_end = o_s1 + (where == MRZ_MEM_HOST ? al(s1_len + 64) : 0);
    int rc = uz_grow(ctx, o_end);
    if (rc) return rc;
    uint8_t *base = (uint8_t *)ctx->rz_scratch;
    mrz_uz_hdr *d_hdr = (mrz_uz_hdr *)(base + o_hdr);
    mrz_uz_tile *d_tiles = (mrz_uz_tile *)(base + o_tiles);
    mrz_uz_base *d_bases = (mrz_uz_base *)(base + o_bases);
    mrz_urec *d_rec = (mrz_urec *)(base + o_rec);
    const uint8_t *d_s0 = (const uint8_t *)s0, *d_s1 = (const uint8_t *)s1;
    if (where == MRZ_MEM_HOST) {
        HIPCHK(ctx, hipMemcpyAsync(base + o_s0, s0, (size_t)s0_len, hipMemcpyHostToDevice, s));
        if (s1_len) HIPCHK(ctx, hipMemcpyAsync(base + o_s1, s1, (size_t)s1_len, hipMemcpyHostToDevice, s));
        d_s0 = base + o_s0;
        d_s1 = base + o_s1;
    }
    HIPCHK(ctx, hipMemsetAsync(d_hdr, 0, sizeof(mrz_uz_hdr), s));
    const int ph = 3 + chunk_bytes;
    {
        const int64_t threads = ntiles * ph;
        hipLaunchKernelGGL(mrz_uz_parse1_kernel, dim3((unsigned)((threads + MRZ_UZ_THREADS - 1) / MRZ_UZ_THREADS)),
                           dim3(MRZ_UZ_THREADS), 0, s, d_s0, s0_len, chunk_bytes, ntiles, d_tiles);
        hipLaunchKernelGGL(mrz_uz_scan_kernel, dim3(1), dim3(MRZ_UZ_SCAN_THREADS), 0, s, d_tiles, ntiles, chunk_bytes, d_bases,
                           d_hdr);
        hipLaunchKernelGGL(mrz_uz_parse2_kernel, dim3((unsigned)((ntiles + MRZ_UZ_THREADS - 1) / MRZ_UZ_THREADS)),
                           dim3(MRZ_UZ_THREADS), 0, s, d_s0, s0_len, chunk_bytes, ntiles, d_bases, d_rec, d_hdr);
        HIPCHK(ctx, hipGetLastError());
    }
    mrz_uz_hdr h;
    HIPCHK(ctx, hipMemcpyAsync(&h, d_hdr, sizeof(h), hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    if (h.final_state != MRZ_UZ_END || h.error || h.lit_total > s1_len) return MRZ_E_CORRUPT;
    *out_len = h.out_total;
    if (h.out_total > out_cap || (h.out_total > 0 && !out)) return MRZ_E_ARG;

    uint8_t *d_out = (uint8_t *)out;
    if (out_where == MRZ_MEM_HOST) {
        rc = mrz_grow(ctx, &ctx->d_rz_out, &ctx->rz_out_cap, h.out_total + 64);
        if (rc) return rc;
        d_out = ctx->d_rz_out;
    }
    // tile size from the record density: about 32 records per tile, 8 KiB .. 128 KiB
    int tshift = MRZ_UZ_TSHIFT_MIN;
    {
        const int64_t target = h.out_total / (h.nrec > 0 ? h.nrec : 1) * 32;
        while (tshift < MRZ_UZ_TSHIFT_MAX && ((int64_t)1 << tshift) < target) tshift++;
    }
    const int64_t otiles = (h.out_total + ((int64_t)1 << tshift) - 1) >> tshift;
    if (otiles) {
        rc = mrz_grow(ctx, &ctx->d_rz_done, &ctx->rz_done_cap, otiles);
        if (rc) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->d_rz_done, 0, (size_t)otiles * sizeof(unsigned), s));
        int cus = 0;
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
        int64_t grid = (int64_t)(cus > 0 ? cus : 64) * 2;
        if (grid > otiles) grid = otiles;
        hipLaunchKernelGGL(mrz_uz_decode_kernel, dim3((unsigned)grid), dim3(MRZ_UZ_DEC_THREADS), 0, s, d_rec, h.nrec + 1,
                           d_s1, d_out, h.out_total, otiles, tshift, ctx->d_rz_done, d_hdr);
        HIPCHK(ctx, hipGetLastError());
    }
    // CRC-32 of the output (gcry_md_read at src/runzip.c:310) and the stored one behind the terminator
    rc = mrz_grow(ctx, &ctx->d_crc_parts, &ctx->crc_parts_cap, mrz_crc32_parts_needed(h.out_total));
    if (rc) return rc;
    HIPCHK(ctx, mrz_launch_crc32(s, d_out, h.out_total, ctx->d_crc_tables, ctx->d_crc_parts, ctx->d_crc_out));
    uint32_t crc = 0;
    uint8_t stored[4];
    HIPCHK(ctx, hipMemcpyAsync(&crc, ctx->d_crc_out, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(stored, d_s0 + h.term_pos + 3, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(&h, d_hdr, sizeof(h), hipMemcpyDeviceToHost, s));
    if (out_where == MRZ_MEM_HOST && h.out_total)
        HIPCHK(ctx, hipMemcpyAsync(out, d_out, (size_t)h.out_total, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    if (h.error) return h.error == 2 ? MRZ_E_STATE : MRZ_E_CORRUPT;
    if (crc_calc) *crc_calc = crc;
    if (crc_stored)
        *crc_stored = (uint32_t)stored[0] << 24 | (uint32_t)stored[1] << 16 | (uint32_t)stored[2] << 8 | stored[3];
    return MRZ_OK;
}
