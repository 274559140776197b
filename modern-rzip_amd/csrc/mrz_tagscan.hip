// mrz_tagscan.hip -- front-end of the rzip stage: the 31-byte XOR tag of every
// position of one segment of the chunk, plus a candidate bitmap.
//
// Restates single_full_tag / single_next_tag (src/rzip.c:330-358):
//     tag(p) = XOR_{i=0..30} hash_index[buf[p+i]]
// The reference rolls this one byte at a time on the CPU; here every position
// is independent: a 256-thread workgroup stages a 4 KiB tile (+30 byte halo)
// in LDS with 16-byte coalesced loads, each thread produces the tags of 16
// consecutive positions (one full 31-term XOR, then 15 two-term rolls) out of
// an LDS copy of hash_index, and writes
//     tags[pos - seg_start]            dense, 8 B per position, coalesced 128 B per thread
//     bitmap16[(pos - seg_start)/16]   bit set iff (tag & mask) == mask
// where mask = the matcher's current minimum_tag_mask (src/rzip.c:573), read
// from the device-resident matcher state so later segments are filtered with
// the mask the sequencer has reached.  Masks only ever tighten, so a bitmap
// made with an older mask is a superset the sequencer re-checks.
//
// Bound: HBM (reads 1 B, writes 8.125 B per position).
#include "mrz_device.h"

#define MRZ_TS_THREADS 256
#define MRZ_TS_PER_THREAD 16
#define MRZ_TS_TILE (MRZ_TS_THREADS * MRZ_TS_PER_THREAD)

__global__ __launch_bounds__(MRZ_TS_THREADS) void mrz_tagscan_kernel(const uint8_t *__restrict__ buf, int64_t n,
                                                                     int64_t seg_start, int64_t seg_len,
                                                                     const int64_t *__restrict__ hash_index,
                                                                     const mrz_seq_state *__restrict__ st,
                                                                     int64_t *__restrict__ tags,
                                                                     uint16_t *__restrict__ bitmap16) {
    __shared__ int64_t sH[256];
    __shared__ __attribute__((aligned(16))) uint8_t sB[MRZ_TS_TILE + 48];

    const int tid = threadIdx.x;
    const int64_t tile_rel = (int64_t)blockIdx.x * MRZ_TS_TILE;
    const int64_t tile_pos = seg_start + tile_rel;
    const int64_t end = n - MRZ_MIN_MATCH;  // last position that has a tag
    // the sequencer never looks at positions <= its current p
    if (tile_pos + MRZ_TS_TILE <= st->p) return;
    const int64_t mask = st->min_mask;

    sH[tid] = hash_index[tid];
    // stage tile bytes [tile_pos, tile_pos + TILE + 32) clipped to n
    for (int i = tid; i < (MRZ_TS_TILE + 32) / 16; i += MRZ_TS_THREADS) {
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
    __syncthreads();

    const int64_t p0 = tile_pos + (int64_t)tid * MRZ_TS_PER_THREAD;
    const int64_t rel0 = tile_rel + (int64_t)tid * MRZ_TS_PER_THREAD;
    if (rel0 >= seg_len) return;
    const uint8_t *b = &sB[tid * MRZ_TS_PER_THREAD];

    int64_t t = 0;
#pragma unroll
    for (int i = 0; i < MRZ_MIN_MATCH; i++) t ^= sH[b[i]];

    int64_t out[MRZ_TS_PER_THREAD];
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < MRZ_TS_PER_THREAD; k++) {
        if (k) t ^= sH[b[k - 1]] ^ sH[b[k + MRZ_MIN_MATCH - 1]];
        const bool valid = (p0 + k <= end) && (rel0 + k < seg_len);
        out[k] = valid ? t : 0;
        if (valid && (t & mask) == mask) bits |= 1u << k;
    }
    int64_t *dst = tags + rel0;
    if (rel0 + MRZ_TS_PER_THREAD <= seg_len) {
#pragma unroll
        for (int k = 0; k < MRZ_TS_PER_THREAD; k += 2)
            *reinterpret_cast<longlong2 *>(dst + k) = make_longlong2(out[k], out[k + 1]);
    } else {
        for (int k = 0; k < MRZ_TS_PER_THREAD; k++)
            if (rel0 + k < seg_len) dst[k] = out[k];
    }
    bitmap16[rel0 / MRZ_TS_PER_THREAD] = (uint16_t)bits;
}

// launcher (host)
extern "C" hipError_t mrz_launch_tagscan(hipStream_t stream, const uint8_t *buf, int64_t n, int64_t seg_start,
                                         int64_t seg_len, const int64_t *hash_index, const mrz_seq_state *st,
                                         int64_t *tags, uint16_t *bitmap16) {
    if (seg_len <= 0) return hipSuccess;
    const int64_t tiles = (seg_len + MRZ_TS_TILE - 1) / MRZ_TS_TILE;
    hipLaunchKernelGGL(mrz_tagscan_kernel, dim3((unsigned)tiles), dim3(MRZ_TS_THREADS), 0, stream, buf, n, seg_start,
                       seg_len, hash_index, st, tags, bitmap16);
    return hipGetLastError();
}
