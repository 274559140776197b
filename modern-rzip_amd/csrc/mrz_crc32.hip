// mrz_crc32.hip -- CRC-32 (IEEE 802.3, reflected; libgcrypt GCRY_MD_CRC32) of a
// chunk at HBM speed.
//
// The reference feeds the chunk to libgcrypt page by page from a helper thread
// (src/rzip.c:488-505,601-666) -- a serial byte-table loop.  CRC is linear over
// GF(2), so the chunk is cut into 64 KiB wave tiles and every lane of a wave
// runs its own interleaved sub-stream:
//   lane l owns the 16-byte groups g = l + 64*k of the tile (so every wave load
//   is 64 x 16 B = 1 KiB, fully coalesced);
//   inside a group the four little-endian words go through the usual
//   slice-by-4 step  R4(v) = T3[v0]^T2[v1]^T1[v2]^T0[v3];
//   the last word of every group but the final one uses a second table set
//   J(v) = "R4 followed by 1008 zero bytes", which jumps the lane state over the
//   63 groups owned by the other lanes;
//   at tile end lane l still has to be advanced over the 16*(63-l) bytes that
//   follow its final group: one GF(2) multiply by x^(128*(63-l)) mod P, then a
//   butterfly XOR over the wave.
// Each wave then advances its tile remainder over the bytes that follow the tile
// (multiply by x^(8*after), one bit of `after` per lane, butterfly product); a small
// second kernel XORs the tiles and the byte tail, adds the 0xffffffff init advanced
// over the whole length and complements.
//
// Bound: HBM read (1 B per byte); LDS table look-ups (8 per 16 B per lane) are
// the on-chip cost.
#include "mrz_device.h"

#define MRZ_CRC_THREADS 256
#define MRZ_CRC_GROUPS 64                       // 16-byte groups per lane per tile
#define MRZ_CRC_TILE (64 * 16 * MRZ_CRC_GROUPS) // 64 KiB per wave
#define MRZ_CRC_POLY 0xEDB88320u

// tables: [0..3] = slice-by-4 T0..T3, [4..7] = jump tables J0..J3, then
// lane_shift[64] = x^(128*(63-l)) mod P, then x2n[32] = x^(2^k) mod P
struct mrz_crc_tables {
    uint32_t t[8][256];
    uint32_t lane_shift[64];
    uint32_t x2n[32];
};

// a(x) * b(x) mod P in the reflected representation (bit 31 = x^0)
__host__ __device__ static inline uint32_t mrz_gf_mul(uint32_t a, uint32_t b) {
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1) ? (b >> 1) ^ MRZ_CRC_POLY : b >> 1;
    }
    return p;
}

// x^(n * 2^k) mod P
__host__ __device__ static inline uint32_t mrz_gf_xpow(const uint32_t *x2n, uint64_t n, unsigned k) {
    uint32_t p = 1u << 31;
    while (n) {
        if (n & 1) p = mrz_gf_mul(x2n[k & 31], p);
        n >>= 1;
        k++;
    }
    return p;
}

extern "C" void mrz_crc_build_tables(mrz_crc_tables *tb) {
    uint32_t t0[256];
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? MRZ_CRC_POLY ^ (c >> 1) : c >> 1;
        t0[i] = c;
    }
    for (uint32_t i = 0; i < 256; i++) {
        tb->t[0][i] = t0[i];
        for (int k = 1; k < 4; k++) tb->t[k][i] = (tb->t[k - 1][i] >> 8) ^ t0[tb->t[k - 1][i] & 0xff];
    }
    // jump tables: R4 of the byte, then 1008 zero bytes
    for (int k = 0; k < 4; k++)
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = tb->t[k][i];
            for (int z = 0; z < 1008; z++) c = t0[c & 0xff] ^ (c >> 8);
            tb->t[4 + k][i] = c;
        }
    // x^(2^k): x^1 is bit 30
    uint32_t p = 1u << 30;
    tb->x2n[0] = p;
    for (int k = 1; k < 32; k++) tb->x2n[k] = p = mrz_gf_mul(p, p);
    for (int l = 0; l < 64; l++) tb->lane_shift[l] = mrz_gf_xpow(tb->x2n, (uint64_t)16 * (63 - l), 3);
}

__device__ __forceinline__ uint32_t mrz_crc_step(const uint32_t (*t)[256], uint32_t v) {
    return t[3][v & 0xff] ^ t[2][(v >> 8) & 0xff] ^ t[1][(v >> 16) & 0xff] ^ t[0][v >> 24];
}

// one wave per 64 KiB tile; parts[tile] = raw remainder of the tile
__global__ __launch_bounds__(MRZ_CRC_THREADS) void mrz_crc_tiles_kernel(const uint8_t *__restrict__ buf, int64_t n,
                                                                        int64_t ntiles,
                                                                        const mrz_crc_tables *__restrict__ tb,
                                                                        uint32_t *__restrict__ parts) {
    __shared__ uint32_t st[8][256];
    for (int i = threadIdx.x; i < 8 * 256; i += MRZ_CRC_THREADS) (&st[0][0])[i] = (&tb->t[0][0])[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * (MRZ_CRC_THREADS / 64) + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const uint8_t *src = buf + tile * MRZ_CRC_TILE + lane * 16;
    uint32_t c = 0;
    uint4 nxt = mrz_ld16(src);
    for (int g = 0; g < MRZ_CRC_GROUPS; g++) {
        const uint4 w = nxt;
        if (g + 1 < MRZ_CRC_GROUPS) nxt = mrz_ld16(src + (int64_t)(g + 1) * 1024);
        c = mrz_crc_step(st, c ^ w.x);
        c = mrz_crc_step(st, c ^ w.y);
        c = mrz_crc_step(st, c ^ w.z);
        c = (g + 1 < MRZ_CRC_GROUPS) ? mrz_crc_step(st + 4, c ^ w.w) : mrz_crc_step(st, c ^ w.w);
    }
    c = mrz_gf_mul(tb->lane_shift[lane], c);
    for (int d = 32; d >= 1; d >>= 1) c ^= (uint32_t)__shfl_xor((int)c, d, MRZ_WAVE);
    // advance the tile remainder over the bytes that follow the tile: x^(8*after) is the product of
    // x^(2^(k+3)) over the set bits k of `after`; every lane contributes one bit, butterfly product
    const uint64_t after = (uint64_t)(n - (tile + 1) * MRZ_CRC_TILE);
    uint32_t f = ((after >> lane) & 1) ? tb->x2n[(lane + 3) & 31] : (1u << 31);
    for (int d = 32; d >= 1; d >>= 1) f = mrz_gf_mul(f, (uint32_t)__shfl_xor((int)f, d, MRZ_WAVE));
    if (lane == 0) parts[tile] = mrz_gf_mul(f, c);
}

// final combine: tiles + byte tail -> CRC-32.  One workgroup.
__global__ __launch_bounds__(MRZ_CRC_THREADS) void mrz_crc_final_kernel(const uint8_t *__restrict__ buf, int64_t n,
                                                                        int64_t ntiles,
                                                                        const mrz_crc_tables *__restrict__ tb,
                                                                        const uint32_t *__restrict__ parts,
                                                                        uint32_t *__restrict__ crc_out) {
    __shared__ uint32_t red[MRZ_CRC_THREADS];
    const int tid = threadIdx.x;
    uint32_t acc = 0;
    // tiles: already advanced to the end of the buffer by the tile kernel
    for (int64_t i = tid; i < ntiles; i += MRZ_CRC_THREADS) acc ^= parts[i];
    // tail (< 64 KiB): each thread a contiguous sub-slice, bytewise
    const int64_t tail0 = ntiles * MRZ_CRC_TILE;
    const int64_t tail = n - tail0;
    if (tail > 0) {
        const int64_t per = (tail + MRZ_CRC_THREADS - 1) / MRZ_CRC_THREADS;
        const int64_t lo = tail0 + (int64_t)tid * per;
        int64_t hi = lo + per;
        if (hi > n) hi = n;
        if (lo < hi) {
            uint32_t c = 0;
            for (int64_t k = lo; k < hi; k++) c = tb->t[0][(c ^ buf[k]) & 0xff] ^ (c >> 8);
            acc ^= mrz_gf_mul(mrz_gf_xpow(tb->x2n, (uint64_t)(n - hi), 3), c);
        }
    }
    red[tid] = acc;
    __syncthreads();
    for (int d = MRZ_CRC_THREADS / 2; d >= 1; d >>= 1) {
        if (tid < d) red[tid] ^= red[tid + d];
        __syncthreads();
    }
    if (tid == 0) {
        const uint32_t init = mrz_gf_mul(mrz_gf_xpow(tb->x2n, (uint64_t)n, 3), 0xFFFFFFFFu);
        *crc_out = ~(red[0] ^ init);
    }
}

extern "C" hipError_t mrz_launch_crc32(hipStream_t stream, const uint8_t *buf, int64_t n, const mrz_crc_tables *tb,
                                       uint32_t *parts, uint32_t *crc_out) {
    const int64_t ntiles = n / MRZ_CRC_TILE;
    if (ntiles > 0) {
        const int wpb = MRZ_CRC_THREADS / 64;
        const int64_t nblocks = (ntiles + wpb - 1) / wpb;
        hipLaunchKernelGGL(mrz_crc_tiles_kernel, dim3((unsigned)nblocks), dim3(MRZ_CRC_THREADS), 0, stream, buf, n,
                           ntiles, tb, parts);
    }
    hipLaunchKernelGGL(mrz_crc_final_kernel, dim3(1), dim3(MRZ_CRC_THREADS), 0, stream, buf, n, ntiles, tb, parts,
                       crc_out);
    return hipGetLastError();
}

extern "C" int64_t mrz_crc32_parts_needed(int64_t n) { return n / MRZ_CRC_TILE + 1; }

extern "C" size_t mrz_crc_tables_size(void) { return sizeof(mrz_crc_tables); }
