// mrz_tagscan.hip -- front end of the rzip stage: the 31-byte XOR tag of every position of a stretch of the chunk,
// filtered by the matcher's mask and COMPACTED into a position-ordered candidate list.
//
// Restates single_full_tag / single_next_tag (src/rzip.c:330-358):
//     tag(p) = XOR_{i=0..30} hash_index[buf[p+i]]
// and the `continue` of hash_search's loop (:573): only positions with (tag & minimum_tag_mask) == minimum_tag_mask
// execute the loop body at all.  The reference rolls the tag one byte at a time on the CPU; here every position is
// independent, and what the sequencer is handed is not a tag per position but the list of the positions that pass:
// 16 bytes {position, tag} per candidate -- 2^-k of the positions under a k-bit mask, so that a window of hundreds
// of GiB costs the sequencer what its O(log N) mask levels cost and nothing per byte.
//
// Three kernels per pass, over tiles of 4096 positions (one 256-thread workgroup each, 16 positions per thread):
//   mark   stages the tile (+30 byte halo) in LDS with 16-byte coalesced loads, computes the 16 tags of every thread
//          (one 31-term XOR + 15 two-term rolls out of an LDS copy of hash_index), writes 16 pass bits per thread
//          (bitmap: 1 bit per position) and the tile's candidate count;
//   scan   exclusive prefix sum of the tile counts (tile_off) and the cut: the pass ends with the last tile whose
//          candidates still fit the list (capacity `cap`), so a stretch whose tags all pass (a run of one byte) cannot
//          overflow anything -- the rest is the next pass's;
//   emit   tiles with candidates recompute the tags of their marked positions and write the list entries at
//          tile_off[tile] + rank inside the tile.
// Where a pass begins is DEVICE state (mrz_seq_state.scan_next, and the matcher's position: tiles an emitted match
// has covered are skipped), so the host queues passes without knowing their geometry; it only bounds their span.
// The bitmap and tile_off stay: rank(position) = tile_off[tile] + popcount(bits of the tile below the position) is the
// sequencer's lower bound into the list (one round trip).
//
// Bound: HBM.  Algorithmic bytes per position: 1 B read by mark (+ 1 B again by emit for tiles that hold candidates),
// 0.125 B of bitmap written, 16 B per candidate written.
#include "mrz_device.h"

#define MRZ_TS_THREADS 256
#define MRZ_TS_PER_THREAD 16
static_assert(MRZ_TS_THREADS * MRZ_TS_PER_THREAD == MRZ_TILE, "one workgroup per tile");
#define MRZ_TS_TILES_PER_WG 8  // consecutive tiles a workgroup takes (hash_index is loaded into LDS once; 4096 positions
                               // per workgroup made the launch, not the loads, the bound: 16.8 M workgroups per 64 GiB)

// where a pass begins: behind the previous one, and not before the tile that holds the matcher's next position
__device__ __forceinline__ int64_t mrz_fe_base(const mrz_seq_state *st) {
    int64_t b = st->scan_next;
    const int64_t pt = ((st->p + 1) >> MRZ_TILE_SHIFT) << MRZ_TILE_SHIFT;
    return pt > b ? pt : b;
}

// stage bytes [tile_pos, tile_pos + TILE + 32) of the chunk (zero beyond n) into sB
__device__ __forceinline__ void mrz_fe_stage(const uint8_t *__restrict__ buf, int64_t n, int64_t tile_pos, uint8_t *sB, int tid) {
    for (int i = tid; i < (MRZ_TILE + 32) / 16; i += MRZ_TS_THREADS) {
        const int64_t g = tile_pos + (int64_t)i * 16;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g + 16 <= n)
            v = mrz_ld16(buf + g);
        else if (g < n) {
            uint8_t tmp[16];
            for (int k = 0; k < 16; k++) tmp[k] = (g + k < n) ? buf[g + k] : (uint8_t)0;
            __builtin_memcpy(&v, tmp, 16);
        }
        *reinterpret_cast<uint4 *>(&sB[i * 16]) = v;
    }
}

// workgroup-wide inclusive prefix sum of one int per thread (256 threads); *total = the sum
__device__ __forceinline__ int mrz_fe_block_incl(int v, int *wsum, int tid, int *total) {
    const int lane = tid & 63, wave = tid >> 6;
    const int incl = mrz_wave_incl_sum(v, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int add = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < MRZ_TS_THREADS / 64; w++) {
        const int x = wsum[w];
        tot += x;
        if (w < wave) add += x;
    }
    *total = tot;
    return incl + add;
}

__global__ __launch_bounds__(MRZ_TS_THREADS) void mrz_fe_mark_kernel(const uint8_t *__restrict__ buf, int64_t n,
                                                                     const int64_t *__restrict__ hash_index,
                                                                     const mrz_seq_state *__restrict__ st, int max_tiles,
                                                                     mrz_fe_hdr *__restrict__ hdr,
                                                                     uint16_t *__restrict__ bitmap16,
                                                                     int *__restrict__ tile_cnt, int *__restrict__ grp_cnt) {
    __shared__ int64_t sH[256];
    __shared__ __attribute__((aligned(16))) uint8_t sB[MRZ_TILE + 48];
    __shared__ int wsum[MRZ_TS_THREADS / 64];

    const int tid = threadIdx.x;
    const int64_t end = n - MRZ_MIN_MATCH;  // last position that has a tag
    const int64_t base = mrz_fe_base(st);
    const int64_t p_done = st->p;
    const int64_t mask = st->min_mask;
    int64_t want = (st->finished || st->error || end < base) ? 0 : ((end - base) >> MRZ_TILE_SHIFT) + 1;
    if (want > max_tiles) want = max_tiles;
    const int ntiles = (int)want;
    if (blockIdx.x == 0 && tid == 0) {
        hdr->base = base;
        hdr->mask = mask;
        hdr->p_done = p_done;
        hdr->ntiles = ntiles;
        hdr->T = 0;
    }
    sH[tid] = hash_index[tid];
    for (int tile = (int)blockIdx.x * MRZ_TS_TILES_PER_WG; tile < ((int)blockIdx.x + 1) * MRZ_TS_TILES_PER_WG && tile < ntiles; tile++) {
    const int64_t tile_pos = base + ((int64_t)tile << MRZ_TILE_SHIFT);
    // the sequencer never looks at positions <= its current p
    if (tile_pos + MRZ_TILE <= p_done + 1) {
        bitmap16[(int64_t)tile * MRZ_TS_THREADS + tid] = 0;
        if (tid == 0) tile_cnt[tile] = 0;
        continue;
    }
    __syncthreads();  // (the previous tile's readers of sB / wsum are done)
    mrz_fe_stage(buf, n, tile_pos, sB, tid);
    __syncthreads();

    const int64_t p0 = tile_pos + (int64_t)tid * MRZ_TS_PER_THREAD;
    const uint8_t *b = &sB[tid * MRZ_TS_PER_THREAD];
    int64_t t = 0;
#pragma unroll
    for (int i = 0; i < MRZ_MIN_MATCH; i++) t ^= sH[b[i]];
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < MRZ_TS_PER_THREAD; k++) {
        if (k) t ^= sH[b[k - 1]] ^ sH[b[k + MRZ_MIN_MATCH - 1]];
        const bool valid = (p0 + k <= end) && (p0 + k > p_done);
        if (valid && (t & mask) == mask) bits |= 1u << k;
    }
    bitmap16[(int64_t)tile * MRZ_TS_THREADS + tid] = (uint16_t)bits;
    int total;
    (void)mrz_fe_block_incl(__popc(bits), wsum, tid, &total);
    if (tid == 0) {
        tile_cnt[tile] = total;
        if (total) atomicAdd(&grp_cnt[tile / MRZ_FE_GROUP], total);
    }
    }
}

// one workgroup per group of 256 tiles: the group's offset (sum of the groups before it), the tiles' offsets inside
// it, and -- by the one thread that sees it -- where the pass is cut
__global__ __launch_bounds__(MRZ_FE_GROUP) void mrz_fe_scan_kernel(mrz_seq_state *__restrict__ st, mrz_fe_hdr *__restrict__ hdr,
                                                                   int64_t n, int64_t cap, const int *__restrict__ tile_cnt,
                                                                   const int *__restrict__ grp_cnt, int *__restrict__ tile_off) {
    __shared__ int wsum[MRZ_FE_GROUP / 64];
    __shared__ int wsum2[MRZ_FE_GROUP / 64];
    const int tid = threadIdx.x;
    const int g = (int)blockIdx.x;
    const int ntiles = hdr->ntiles;
    const int64_t base = hdr->base;
    const int64_t end = n - MRZ_MIN_MATCH;
    if (ntiles == 0) {
        if (g == 0 && tid == 0) {
            hdr->T = 0;
            st->seg_start = base;
            st->seg_end = base;
            st->n_cand = 0;
            st->scan_next = base;
            st->list_mask = hdr->mask;
            tile_off[0] = 0;
        }
        return;
    }
    const int ngroups = (ntiles + MRZ_FE_GROUP - 1) / MRZ_FE_GROUP;
    if (g >= ngroups) return;
    int part = 0;
    for (int i = tid; i < g; i += MRZ_FE_GROUP) part += grp_cnt[i];
    int goff;
    (void)mrz_fe_block_incl(part, wsum, tid, &goff);
    const int tile = g * MRZ_FE_GROUP + tid;
    const int c = tile < ntiles ? tile_cnt[tile] : 0;
    int gtotal;
    const int incl = mrz_fe_block_incl(c, wsum2, tid, &gtotal);
    const int64_t off = (int64_t)goff + incl - c;
    if (tile <= ntiles) tile_off[tile] = (int)off;
    if (tile == ntiles - 1 && tid == MRZ_FE_GROUP - 1 && g + 1 == ngroups) tile_off[ntiles] = (int)(off + c);  // (a full last group)
    if (tile < ntiles) {
        const bool fits = off + c <= cap;
        int T = -1;
        int64_t cnt = 0;
        if (!fits && (tile == 0 || off <= cap)) {  // the first tile that does not fit: the pass ends before it
            T = tile;
            cnt = off;
        } else if (fits && tile == ntiles - 1) {
            T = ntiles;
            cnt = off + c;
        }
        if (T >= 0) {
            if (T == 0) T = 1, cnt = off + c;  // (cannot happen: cap >= one tile; never leave a pass empty)
            const int64_t stop = base + ((int64_t)T << MRZ_TILE_SHIFT);
            hdr->T = T;
            st->seg_start = base;
            st->seg_end = stop < end + 1 ? stop : end + 1;
            st->n_cand = cnt;
            st->scan_next = stop;
            st->list_mask = hdr->mask;
        }
    }
}

__global__ __launch_bounds__(MRZ_TS_THREADS) void mrz_fe_emit_kernel(const uint8_t *__restrict__ buf, int64_t n,
                                                                     const int64_t *__restrict__ hash_index,
                                                                     const mrz_fe_hdr *__restrict__ hdr,
                                                                     const uint16_t *__restrict__ bitmap16,
                                                                     const int *__restrict__ tile_cnt,
                                                                     const int *__restrict__ tile_off,
                                                                     mrz_cand *__restrict__ cand) {
    __shared__ int64_t sH[256];
    __shared__ __attribute__((aligned(16))) uint8_t sB[MRZ_TILE + 48];
    __shared__ int wsum[MRZ_TS_THREADS / 64];
    const int tid = threadIdx.x;
    const int T = hdr->T;
    sH[tid] = hash_index[tid];
    for (int tile = (int)blockIdx.x * MRZ_TS_TILES_PER_WG; tile < ((int)blockIdx.x + 1) * MRZ_TS_TILES_PER_WG && tile < T; tile++) {
    if (tile_cnt[tile] == 0) continue;
    const int64_t tile_pos = hdr->base + ((int64_t)tile << MRZ_TILE_SHIFT);
    const uint32_t bits = bitmap16[(int64_t)tile * MRZ_TS_THREADS + tid];
    __syncthreads();  // (the previous tile's readers of sB / wsum are done)
    mrz_fe_stage(buf, n, tile_pos, sB, tid);
    __syncthreads();
    int total;
    const int cnt = __popc(bits);
    const int incl = mrz_fe_block_incl(cnt, wsum, tid, &total);
    if (!bits) continue;
    const int64_t p0 = tile_pos + (int64_t)tid * MRZ_TS_PER_THREAD;
    const uint8_t *b = &sB[tid * MRZ_TS_PER_THREAD];
    int64_t t = 0;
#pragma unroll
    for (int i = 0; i < MRZ_MIN_MATCH; i++) t ^= sH[b[i]];
    mrz_cand *dst = cand + tile_off[tile] + (incl - cnt);
    int at = 0;
#pragma unroll
    for (int k = 0; k < MRZ_TS_PER_THREAD; k++) {
        if (k) t ^= sH[b[k - 1]] ^ sH[b[k + MRZ_MIN_MATCH - 1]];
        if ((bits >> k) & 1u) {
            mrz_cand e;
            e.off = p0 + k;
            e.t = t;
            dst[at++] = e;
        }
    }
    }
}

// launcher (host): one pass over at most max_tiles tiles from where the device state says the last one ended
extern "C" hipError_t mrz_launch_frontend(hipStream_t stream, const uint8_t *buf, int64_t n, const int64_t *hash_index,
                                          mrz_seq_state *st, int max_tiles, int64_t cap, mrz_fe_hdr *hdr, uint16_t *bitmap16,
                                          int *tile_cnt, int *tile_off, int *grp_cnt, mrz_cand *cand) {
    if (max_tiles <= 0) return hipErrorInvalidValue;
    const int ngroups = (max_tiles + MRZ_FE_GROUP - 1) / MRZ_FE_GROUP;
    hipError_t e = hipMemsetAsync(grp_cnt, 0, (size_t)ngroups * sizeof(int), stream);
    if (e != hipSuccess) return e;
    const unsigned wgs = (unsigned)((max_tiles + MRZ_TS_TILES_PER_WG - 1) / MRZ_TS_TILES_PER_WG);
    hipLaunchKernelGGL(mrz_fe_mark_kernel, dim3(wgs), dim3(MRZ_TS_THREADS), 0, stream, buf, n, hash_index, st,
                       max_tiles, hdr, bitmap16, tile_cnt, grp_cnt);
    hipLaunchKernelGGL(mrz_fe_scan_kernel, dim3((unsigned)ngroups), dim3(MRZ_FE_GROUP), 0, stream, st, hdr, n, cap, tile_cnt,
                       grp_cnt, tile_off);
    hipLaunchKernelGGL(mrz_fe_emit_kernel, dim3(wgs), dim3(MRZ_TS_THREADS), 0, stream, buf, n, hash_index, hdr,
                       bitmap16, tile_cnt, tile_off, cand);
    return hipGetLastError();
}
