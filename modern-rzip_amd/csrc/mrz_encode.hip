// mrz_encode.hip -- parallel back-end of the rzip stage: turns the sequencer's
// event list into the two output streams.
//
// Restates put_match / put_literal / put_header / put_vchars / write_sbstream
// (src/rzip.c:163-227) and the tail of hash_search (:619,:664-665):
//   stream 0: literal run -> {00, len:u16le} per <=0xFFFF piece,
//             match       -> {01, len:u16le, dist:u{chunk_bytes}le} per <=0xFFFF piece
//             (dist = p - offset is the same for every piece of one match),
//             end         -> 00 00 00 + CRC-32 (most significant byte first)
//   stream 1: the literal bytes, concatenated.
// Item i (0 <= i < E) is "the literal run before match i, then match i"; item E
// is the trailing literal run.  Sizes are prefix-summed (two-level block scan),
// then every item writes its records at its own offset and the literal bytes
// are gathered by a grid-wide copy (16 B per thread, unaligned source).
//
// Bound: HBM (reads <= N literal bytes once, writes them once).
#include "mrz_device.h"

#define MRZ_ENC_THREADS 256

__device__ __forceinline__ void mrz_item(const mrz_event *ev, int64_t E, int64_t n, int64_t i, int64_t *lit_from,
                                         int64_t *lit_len, int64_t *mp, int64_t *mofs, int64_t *mlen) {
    const int64_t prev_end = i ? ev[i - 1].p + ev[i - 1].len : 0;
    *lit_from = prev_end;
    if (i < E) {
        *lit_len = ev[i].p - prev_end;
        *mp = ev[i].p;
        *mofs = ev[i].ofs;
        *mlen = ev[i].len;
    } else {
        *lit_len = n - prev_end;
        *mp = *mofs = *mlen = 0;
    }
}

__device__ __forceinline__ int64_t mrz_pieces(int64_t len) { return (len + 0xFFFE) / 0xFFFF; }

// block-wide exclusive scan of two int64 values; returns block totals
__device__ static void mrz_block_scan2(int64_t &a, int64_t &b, int64_t &ta, int64_t &tb) {
    __shared__ int64_t sa[MRZ_ENC_THREADS], sb[MRZ_ENC_THREADS];
    const int tid = threadIdx.x;
    sa[tid] = a;
    sb[tid] = b;
    __syncthreads();
    for (int d = 1; d < MRZ_ENC_THREADS; d <<= 1) {
        int64_t va = 0, vb = 0;
        if (tid >= d) {
            va = sa[tid - d];
            vb = sb[tid - d];
        }
        __syncthreads();
        sa[tid] += va;
        sb[tid] += vb;
        __syncthreads();
    }
    ta = sa[MRZ_ENC_THREADS - 1];
    tb = sb[MRZ_ENC_THREADS - 1];
    const int64_t ia = sa[tid] - a, ib = sb[tid] - b;
    __syncthreads();
    a = ia;
    b = ib;
}

// pass 1: per-block totals of (stream-0 bytes, stream-1 bytes)
__global__ __launch_bounds__(MRZ_ENC_THREADS) void mrz_enc_size_kernel(const mrz_event *__restrict__ ev, int64_t E,
                                                                       int64_t n, int cb,
                                                                       int64_t *__restrict__ block_s0,
                                                                       int64_t *__restrict__ block_s1) {
    const int64_t i = (int64_t)blockIdx.x * MRZ_ENC_THREADS + threadIdx.x;
    int64_t c0 = 0, c1 = 0;
    if (i <= E) {
        int64_t lf, ll, mp, mo, ml;
        mrz_item(ev, E, n, i, &lf, &ll, &mp, &mo, &ml);
        c0 = mrz_pieces(ll) * 3 + mrz_pieces(ml) * (3 + cb);
        c1 = ll;
    }
    int64_t ta, tb;
    mrz_block_scan2(c0, c1, ta, tb);
    if (threadIdx.x == 0) {
        block_s0[blockIdx.x] = ta;
        block_s1[blockIdx.x] = tb;
    }
}

// pass 2: one workgroup turns the block totals into exclusive offsets and
// publishes the grand totals
__global__ __launch_bounds__(MRZ_ENC_THREADS) void mrz_enc_scan_kernel(int64_t *__restrict__ block_s0,
                                                                       int64_t *__restrict__ block_s1,
                                                                       int64_t nblocks,
                                                                       mrz_enc_totals *__restrict__ totals) {
    int64_t run0 = 0, run1 = 0;
    for (int64_t base = 0; base < nblocks; base += MRZ_ENC_THREADS) {
        const int64_t i = base + threadIdx.x;
        int64_t a = i < nblocks ? block_s0[i] : 0, b = i < nblocks ? block_s1[i] : 0;
        int64_t ta, tb;
        mrz_block_scan2(a, b, ta, tb);
        if (i < nblocks) {
            block_s0[i] = run0 + a;
            block_s1[i] = run1 + b;
        }
        run0 += ta;
        run1 += tb;
    }
    if (threadIdx.x == 0) {
        totals->s0_len = run0;
        totals->s1_len = run1;
    }
}

#define MRZ_ENC_OWN_RECORDS 4

// record r of an item (nl literal records, then nm match records) that starts at offset o0 of stream 0
__device__ __forceinline__ void mrz_put_record(uint8_t *__restrict__ s0, int64_t o0, int64_t r, int64_t nl, int64_t nm,
                                               int64_t ll, int64_t ml, int64_t dist, int cb) {
    if (r < nl) {
        const int64_t piece = r == nl - 1 ? ll - r * 0xFFFF : 0xFFFF;
        const int64_t o = o0 + r * 3;
        s0[o] = 0;
        s0[o + 1] = (uint8_t)piece;
        s0[o + 2] = (uint8_t)(piece >> 8);
    } else {
        const int64_t k = r - nl;
        const int64_t piece = k == nm - 1 ? ml - k * 0xFFFF : 0xFFFF;
        const int64_t o = o0 + nl * 3 + k * (3 + cb);
        s0[o] = 1;
        s0[o + 1] = (uint8_t)piece;
        s0[o + 2] = (uint8_t)(piece >> 8);
        for (int j = 0; j < cb; j++) s0[o + 3 + j] = (uint8_t)((uint64_t)dist >> (8 * j));
    }
}

// pass 3: records into stream 0, per-item stream-1 offsets, statistics
__global__ __launch_bounds__(MRZ_ENC_THREADS) void mrz_enc_write_kernel(const mrz_event *__restrict__ ev, int64_t E,
                                                                        int64_t n, int cb,
                                                                        const int64_t *__restrict__ block_s0,
                                                                        const int64_t *__restrict__ block_s1,
                                                                        uint8_t *__restrict__ s0,
                                                                        int64_t *__restrict__ lit_off,
                                                                        mrz_enc_totals *__restrict__ totals,
                                                                        uint32_t crc) {
    const int64_t i = (int64_t)blockIdx.x * MRZ_ENC_THREADS + threadIdx.x;
    int64_t c0 = 0, c1 = 0;
    int64_t lf = 0, ll = 0, mp = 0, mo = 0, ml = 0;
    if (i <= E) {
        mrz_item(ev, E, n, i, &lf, &ll, &mp, &mo, &ml);
        c0 = mrz_pieces(ll) * 3 + mrz_pieces(ml) * (3 + cb);
        c1 = ll;
    }
    int64_t ta, tb;
    mrz_block_scan2(c0, c1, ta, tb);
    const bool live = i <= E;
    int64_t o0 = block_s0[blockIdx.x] + c0;
    const int64_t o1 = block_s1[blockIdx.x] + c1;
    if (live) lit_off[i] = o1;
    if (!live) ll = ml = 0;
    // item i = nl literal records, then nm match records (put_literal / put_match split at 0xFFFF, :183,:217)
    const int64_t nl = mrz_pieces(ll), nm = mrz_pieces(ml);
    const int64_t nlit = nl, nmat = nm;
    const int64_t dist = mp - mo;
    // the first few records by the owning thread; an item with more of them (a match of gigabytes has tens of
    // thousands) is finished by the whole workgroup below
    for (int64_t r = 0; r < nl + nm && r < MRZ_ENC_OWN_RECORDS; r++) mrz_put_record(s0, o0, r, nl, nm, ll, ml, dist, cb);
    {
        __shared__ int64_t sh_o0[MRZ_ENC_THREADS], sh_ll[MRZ_ENC_THREADS], sh_ml[MRZ_ENC_THREADS], sh_dist[MRZ_ENC_THREADS];
        __shared__ int sh_any;
        const int tid = threadIdx.x;
        if (tid == 0) sh_any = 0;
        __syncthreads();
        sh_o0[tid] = o0;
        sh_ll[tid] = ll;
        sh_ml[tid] = (nl + nm > MRZ_ENC_OWN_RECORDS) ? ml : -1;  // -1: nothing left to do for this item
        sh_dist[tid] = dist;
        if (nl + nm > MRZ_ENC_OWN_RECORDS) sh_any = 1;
        __syncthreads();
        if (sh_any)
            for (int t = 0; t < MRZ_ENC_THREADS; t++) {
                const int64_t tml = sh_ml[t];
                if (tml < 0) continue;
                const int64_t tll = sh_ll[t], tnl = mrz_pieces(tll), tnm = mrz_pieces(tml);
                for (int64_t r = MRZ_ENC_OWN_RECORDS + tid; r < tnl + tnm; r += MRZ_ENC_THREADS)
                    mrz_put_record(s0, sh_o0[t], r, tnl, tnm, tll, tml, sh_dist[t], cb);
            }
    }
    o0 += nl * 3 + nm * (3 + cb);
    // statistics (st->stats.* at src/rzip.c:188-189,219-220): reduced over the wave first
    {
        int64_t v0 = nlit, v1 = nlit ? ll : 0, v2 = nmat, v3 = nmat ? ml : 0;
        for (int d = 32; d >= 1; d >>= 1) {
            v0 += mrz_shfl_xor64(v0, d);
            v1 += mrz_shfl_xor64(v1, d);
            v2 += mrz_shfl_xor64(v2, d);
            v3 += mrz_shfl_xor64(v3, d);
        }
        if ((threadIdx.x & 63) == 0) {
            if (v0) atomicAdd((unsigned long long *)&totals->literals, (unsigned long long)v0);
            if (v1) atomicAdd((unsigned long long *)&totals->literal_bytes, (unsigned long long)v1);
            if (v2) atomicAdd((unsigned long long *)&totals->matches, (unsigned long long)v2);
            if (v3) atomicAdd((unsigned long long *)&totals->match_bytes, (unsigned long long)v3);
        }
    }
    if (live && i == E) {
        // terminator literal + CRC (src/rzip.c:664-665); o0 == total s0_len here
        lit_off[E + 1] = o1 + ll;
        s0[o0] = 0;
        s0[o0 + 1] = 0;
        s0[o0 + 2] = 0;
        s0[o0 + 3] = (uint8_t)(crc >> 24);
        s0[o0 + 4] = (uint8_t)(crc >> 16);
        s0[o0 + 5] = (uint8_t)(crc >> 8);
        s0[o0 + 6] = (uint8_t)crc;
    }
}

// pass 4: gather the literal bytes into stream 1 (write_sbstream, :197-211)
__global__ __launch_bounds__(MRZ_ENC_THREADS) void mrz_literal_gather_kernel(const uint8_t *__restrict__ buf,
                                                                             const mrz_event *__restrict__ ev,
                                                                             int64_t E,
                                                                             const int64_t *__restrict__ lit_off,
                                                                             int64_t s1_len,
                                                                             uint8_t *__restrict__ s1) {
    __shared__ int64_t s_lo, s_hi;
    const int64_t blk_o = (int64_t)blockIdx.x * MRZ_ENC_THREADS * 16;
    // narrow the item range for this block: items lo..hi cover [blk_o, blk_o + 4096)
    if (threadIdx.x < 2) {
        int64_t target = threadIdx.x == 0 ? blk_o : blk_o + (int64_t)MRZ_ENC_THREADS * 16 - 1;
        if (target >= s1_len) target = s1_len - 1;
        int64_t lo = 0, hi = E;  // largest i with lit_off[i] <= target
        while (lo < hi) {
            const int64_t mid = (lo + hi + 1) >> 1;
            if (lit_off[mid] <= target)
                lo = mid;
            else
                hi = mid - 1;
        }
        if (threadIdx.x == 0)
            s_lo = lo;
        else
            s_hi = lo;
    }
    __syncthreads();
    const int64_t o = blk_o + (int64_t)threadIdx.x * 16;
    if (o >= s1_len) return;
    int64_t lo = s_lo, hi = s_hi;
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (lit_off[mid] <= o)
            lo = mid;
        else
            hi = mid - 1;
    }
    int64_t i = lo;
    // skip empty runs so that lit_off[i] <= o < lit_off[i+1]
    while (lit_off[i + 1] <= o) i++;
    int64_t from = (i ? ev[i - 1].p + ev[i - 1].len : 0) + (o - lit_off[i]);
    if (o + 16 <= lit_off[i + 1]) {
        *reinterpret_cast<uint4 *>(s1 + o) = mrz_ld16(buf + from);
        return;
    }
    const int64_t stop = (o + 16 < s1_len) ? o + 16 : s1_len;
    int64_t run_end = lit_off[i + 1];
    for (int64_t w = o; w < stop; w++) {
        while (w >= run_end) {
            i++;
            run_end = lit_off[i + 1];
            from = i ? ev[i - 1].p + ev[i - 1].len : 0;
        }
        s1[w] = buf[from++];
    }
}

extern "C" hipError_t mrz_launch_enc_size(hipStream_t stream, const mrz_event *ev, int64_t E, int64_t n, int cb,
                                          int64_t *block_s0, int64_t *block_s1, mrz_enc_totals *totals) {
    const int64_t nblocks = (E + 1 + MRZ_ENC_THREADS - 1) / MRZ_ENC_THREADS;
    hipLaunchKernelGGL(mrz_enc_size_kernel, dim3((unsigned)nblocks), dim3(MRZ_ENC_THREADS), 0, stream, ev, E, n, cb,
                       block_s0, block_s1);
    hipLaunchKernelGGL(mrz_enc_scan_kernel, dim3(1), dim3(MRZ_ENC_THREADS), 0, stream, block_s0, block_s1, nblocks,
                       totals);
    return hipGetLastError();
}

extern "C" hipError_t mrz_launch_enc_write(hipStream_t stream, const uint8_t *buf, const mrz_event *ev, int64_t E,
                                           int64_t n, int cb, const int64_t *block_s0, const int64_t *block_s1,
                                           uint8_t *s0, uint8_t *s1, int64_t s1_len, int64_t *lit_off,
                                           mrz_enc_totals *totals, uint32_t crc) {
    const int64_t nblocks = (E + 1 + MRZ_ENC_THREADS - 1) / MRZ_ENC_THREADS;
    hipLaunchKernelGGL(mrz_enc_write_kernel, dim3((unsigned)nblocks), dim3(MRZ_ENC_THREADS), 0, stream, ev, E, n, cb,
                       block_s0, block_s1, s0, lit_off, totals, crc);
    if (s1_len > 0) {
        const int64_t gblocks = (s1_len + MRZ_ENC_THREADS * 16 - 1) / (MRZ_ENC_THREADS * 16);
        hipLaunchKernelGGL(mrz_literal_gather_kernel, dim3((unsigned)gblocks), dim3(MRZ_ENC_THREADS), 0, stream, buf,
                           ev, E, lit_off, s1_len, s1);
    }
    return hipGetLastError();
}
