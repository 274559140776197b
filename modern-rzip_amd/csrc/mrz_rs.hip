// mrz_rs.hip -- rs-mrzip's encoder on the GPU: CCSDS RS(255,223) parity in Berlekamp's dual
// basis for every 223-byte row, and the burst interleave, fused.
//
// Restates rse32 (rs-mrzip/reed-solomon.c:115-141), scatter (:311-321) and the burst loop of
// encode() (rs-mrzip/rs-mrzip.c:119-158).  The reference encodes one row after another on one
// core; the 8176 rows of a burst (and all bursts) are independent, so:
//   * a 128-thread workgroup takes 112 consecutive rows (8176 = 73 x 112): their 24,976 input
//     bytes are loaded coalesced (16 B per lane) into an LDS image with 260-byte rows (4 x 65: a
//     row's bytes can be read a dword at a time, and one column of 64 rows spans all 64 banks);
//   * thread t runs the 32-byte LFSR of row t: per data byte one dual->conventional look-up (four
//     at a time, ahead of the steps that use them) and one 32-byte row of the precomputed
//     "feedback x generator" table (256 x 32 B in LDS, its halves in two arrays so that each
//     ds_read_b128 spreads over all banks) XORed into the shifted register held in 8 dwords --
//     the log/antilog arithmetic of the reference (:122-134) is folded into that table;
//   * the 32 parity bytes go back through the conventional->dual table into columns 223..254 of
//     the LDS image, and the image is written out transposed (column c of row r at
//     c * 8176 + r, :311-321) as 16-byte words: 16 rows per lane, 112 contiguous bytes per column.
// The data columns are written unchanged (taltab o tal1tab = identity, :118,138).
// Tables are generated on the host from the field polynomial, the generator roots and the 8
// dual-basis images; nothing is copied from the reference.
//
// Bound: 223 B read + 255 B written per row (478 B per row) against HBM; measured, the kernel is bound by the LFSR's
// LDS traffic instead (32 B of table per data byte: without the LFSR the same kernel moves 3.1 TB/s, with it 1.3).
#include <string.h>

#include <thread>
#include <vector>

#include "mrz_ctx.h"
#include "mrz_device.h"

#define MRZ_RS_ROWS 8176  // BLK_LEN, rs-mrzip/reed-solomon.h:31
#define MRZ_RS_K 223
#define MRZ_RS_N 255
#define MRZ_RS_TILE 112   // rows per workgroup; 8176 = 73 * 112
#define MRZ_RS_THREADS 128
#define MRZ_RS_PITCH 260  // bytes between the rows of the LDS image: 4 x 65 -- a row's data can be read a dword at a time, and
                          // the dwords (and bytes) of one column of 64 consecutive rows lie on 64 different banks

struct mrz_rs_tables {
    uint8_t fbgen[256][32];  // fbgen[f][j] = f * g_j in GF(256), conventional basis
    uint8_t tal[256];        // conventional -> dual basis
    uint8_t tal1[256];       // dual -> conventional
    uint8_t ex[256];         // alpha^i (index -> polynomial form), ex[255] = 0     (Alpha_to, reed-solomon.c:28)
    uint8_t lg[256];         // log_alpha (polynomial -> index form), lg[0] = 255   (Index_of, :43)
};

static void mrz_rs_build_tables(mrz_rs_tables *T) {
    uint8_t ex[256], lg[256];
    unsigned v = 1;
    for (int i = 0; i < 255; i++) {  // GF(2^8), p(x) = x^8 + x^7 + x^2 + x + 1, alpha = 2
        ex[i] = (uint8_t)v;
        lg[v] = (uint8_t)i;
        v <<= 1;
        if (v & 0x100) v ^= 0x187;
    }
    auto mul = [&](uint8_t a, uint8_t b) -> uint8_t { return (!a || !b) ? 0 : ex[(lg[a] + lg[b]) % 255]; };
    uint8_t g[33] = { 1 };  // g(x) = prod_{j=112..143} (x - alpha^(11 j))
    for (int j = 112, deg = 0; j <= 143; j++, deg++) {
        const uint8_t root = ex[(11 * j) % 255];
        g[deg + 1] = 0;
        for (int k = deg + 1; k > 0; k--) g[k] = g[k - 1] ^ mul(g[k], root);
        g[0] = mul(g[0], root);
    }
    for (int f = 0; f < 256; f++)
        for (int j = 0; j < 32; j++) T->fbgen[f][j] = mul((uint8_t)f, g[j]);
    static const uint8_t basis[8] = { 0x8d, 0xef, 0xec, 0x86, 0xfa, 0x99, 0xaf, 0x7b };
    for (int i = 0; i < 256; i++) {
        uint8_t t = 0;
        for (int k = 0; k < 8; k++)
            if (i & (1 << k)) t ^= basis[7 - k];
        T->tal[i] = t;
    }
    for (int i = 0; i < 256; i++) T->tal1[T->tal[i]] = (uint8_t)i;
    for (int i = 0; i < 255; i++) {
        T->ex[i] = ex[i];
        T->lg[ex[i]] = (uint8_t)i;
    }
    T->ex[255] = 0;
    T->lg[0] = 255;
}

// grid.x = bursts * 73; each workgroup: 112 rows of one burst
__global__ __launch_bounds__(MRZ_RS_THREADS) void mrz_rs_encode_kernel(const uint8_t *__restrict__ in, int64_t n,
                                                                       const mrz_rs_tables *__restrict__ T,
                                                                       uint8_t *__restrict__ out) {
    // (the two halves of a table row in arrays of their own: a 16-byte read of row f then lands on banks 4 (f % 16)..+3,
    // all 64 banks in use -- with 32-byte rows each of the two reads had half of the banks to itself)
    __shared__ __attribute__((aligned(16))) uint8_t s_fb_lo[256][16], s_fb_hi[256][16];
    __shared__ uint8_t s_tal[256], s_tal1[256];
    __shared__ __attribute__((aligned(16))) uint8_t s_img[MRZ_RS_TILE * MRZ_RS_PITCH + 16];

    const int tid = threadIdx.x;
    for (int i = tid; i < 256 * 32 / 16; i += MRZ_RS_THREADS) {
        const uint4 v = reinterpret_cast<const uint4 *>(&T->fbgen[0][0])[i];
        *reinterpret_cast<uint4 *>((i & 1) ? &s_fb_hi[i >> 1][0] : &s_fb_lo[i >> 1][0]) = v;
    }
    for (int i = tid; i < 256; i += MRZ_RS_THREADS) {
        s_tal[i] = T->tal[i];
        s_tal1[i] = T->tal1[i];
    }
    const int64_t burst = blockIdx.x / (MRZ_RS_ROWS / MRZ_RS_TILE);
    const int tile = blockIdx.x % (MRZ_RS_ROWS / MRZ_RS_TILE);
    const int64_t row0 = burst * MRZ_RS_ROWS + (int64_t)tile * MRZ_RS_TILE;  // global row index
    const int64_t in0 = row0 * MRZ_RS_K;
    // stage 112 x 223 input bytes (zero beyond n, rs-mrzip.c:132-133) into 255-byte LDS rows
    const int tile_bytes = MRZ_RS_TILE * MRZ_RS_K;
    for (int x = tid * 16; x < tile_bytes; x += MRZ_RS_THREADS * 16) {
        uint8_t tmp[16];
        const int64_t gpos = in0 + x;
        if (gpos + 16 <= n) {
            const uint4 v = mrz_ld16(in + gpos);
            __builtin_memcpy(tmp, &v, 16);
        } else {
            for (int k = 0; k < 16; k++) tmp[k] = (gpos + k < n) ? in[gpos + k] : (uint8_t)0;
        }
        int r = x / MRZ_RS_K, c = x % MRZ_RS_K;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            s_img[r * MRZ_RS_PITCH + c] = tmp[k];  // (112 x 223 = 1561 x 16: no piece straddles the tile's end)
            if (++c == MRZ_RS_K) {
                c = 0;
                r++;
            }
        }
    }
    __syncthreads();
#ifndef MRZ_RS_SKIP_LFSR  // (-DMRZ_RS_SKIP_LFSR / -DMRZ_RS_SKIP_SCATTER: ablation builds for tools/probe_rs.py, never shipped)
    if (tid < MRZ_RS_TILE) {
        // rse32: bb[j] = bb[j-1] ^ g_j * feedback, bb[0] = g_0 * feedback  (:120-135)
        uint32_t b[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        // The data bytes come a dword at a time and their dual->conventional look-ups four at a time: none of that
        // waits for the register, so a step's critical path is the one trip to the LDS for its table row.
        const uint32_t *row32 = reinterpret_cast<const uint32_t *>(&s_img[tid * MRZ_RS_PITCH]);
        auto step = [&](uint32_t d) {
            const uint32_t fb = d ^ (b[7] >> 24);
#pragma unroll
            for (int k = 7; k > 0; k--) b[k] = (b[k] << 8) | (b[k - 1] >> 24);
            b[0] <<= 8;
            const uint4 g0 = *reinterpret_cast<const uint4 *>(&s_fb_lo[fb][0]);
            const uint4 g1 = *reinterpret_cast<const uint4 *>(&s_fb_hi[fb][0]);
            b[0] ^= g0.x;
            b[1] ^= g0.y;
            b[2] ^= g0.z;
            b[3] ^= g0.w;
            b[4] ^= g1.x;
            b[5] ^= g1.y;
            b[6] ^= g1.z;
            b[7] ^= g1.w;
        };
        {  // bytes 222, 221, 220 (223 = 4 x 55 + 3; the top byte of this dword is the first parity column)
            const uint32_t w = row32[MRZ_RS_K / 4];
            const uint32_t d2 = s_tal1[(w >> 16) & 0xff], d1 = s_tal1[(w >> 8) & 0xff], d0 = s_tal1[w & 0xff];
            step(d2);
            step(d1);
            step(d0);
        }
#pragma unroll 2
        for (int j = MRZ_RS_K / 4 - 1; j >= 0; j--) {
            const uint32_t w = row32[j];
            const uint32_t d3 = s_tal1[w >> 24], d2 = s_tal1[(w >> 16) & 0xff], d1 = s_tal1[(w >> 8) & 0xff],
                           d0 = s_tal1[w & 0xff];
            step(d3);
            step(d2);
            step(d1);
            step(d0);
        }
        uint8_t *par = &s_img[tid * MRZ_RS_PITCH + MRZ_RS_K];
#pragma unroll
        for (int j = 0; j < 32; j++) par[j] = s_tal[(b[j >> 2] >> (8 * (j & 3))) & 0xff];  // :138
    }
#endif
    __syncthreads();
    // scatter (:311-321): dst[c * 8176 + r] = row r, column c.  A column's 112 bytes of this tile are 7 x 16 B: a lane
    // gathers 16 rows of one column from the image (the odd row pitch keeps the 16 byte reads of neighbouring lanes on
    // different banks) and stores them as one 16-byte word (8176, 112 and the burst size are multiples of 16)
#ifndef MRZ_RS_SKIP_SCATTER
    uint8_t *dst = out + burst * (int64_t)MRZ_RS_N * MRZ_RS_ROWS + (int64_t)tile * MRZ_RS_TILE;
    const int segs = MRZ_RS_TILE / 16;  // 7
    for (int idx = tid; idx < MRZ_RS_N * segs; idx += MRZ_RS_THREADS) {
        const int c = idx / segs, u = idx % segs;
        const uint8_t *p = &s_img[(16 * u) * MRZ_RS_PITCH + c];
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            w[k] = (uint32_t)p[(4 * k) * MRZ_RS_PITCH] | (uint32_t)p[(4 * k + 1) * MRZ_RS_PITCH] << 8 |
                   (uint32_t)p[(4 * k + 2) * MRZ_RS_PITCH] << 16 | (uint32_t)p[(4 * k + 3) * MRZ_RS_PITCH] << 24;
        uint4 v;
        v.x = w[0];
        v.y = w[1];
        v.z = w[2];
        v.w = w[3];
        *reinterpret_cast<uint4 *>(dst + (int64_t)c * MRZ_RS_ROWS + 16 * u) = v;
    }
#endif
}

// ---- BLAKE2b-512 on the host (the trailer hash of rs-mrzip.c:138,148 is one serial chain over
// the whole padded stream; it runs on a host thread while the GPU encodes) ---------------------
namespace {
struct HostB2 {
    uint64_t h[8], t0 = 0, t1 = 0;
    uint8_t buf[128];
    size_t buflen = 0;
    static uint64_t ror(uint64_t x, int c) { return (x >> c) | (x << (64 - c)); }
    HostB2() {
        static const uint64_t iv[8] = { 0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                        0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                        0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL };
        for (int i = 0; i < 8; i++) h[i] = iv[i];
        h[0] ^= 0x01010000ULL ^ 64;
    }
    void compress(const uint8_t *blk, bool last) {
        static const uint64_t iv[8] = { 0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                        0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                        0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL };
        static const uint8_t sg[10][16] = {
            { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }, { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 },
            { 11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4 }, { 7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8 },
            { 9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13 }, { 2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9 },
            { 12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11 }, { 13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10 },
            { 6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5 }, { 10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0 }
        };
        uint64_t m[16], v[16];
        memcpy(m, blk, 128);
        for (int i = 0; i < 8; i++) {
            v[i] = h[i];
            v[i + 8] = iv[i];
        }
        v[12] ^= t0;
        v[13] ^= t1;
        if (last) v[14] = ~v[14];
        static const int q[8][4] = { { 0, 4, 8, 12 }, { 1, 5, 9, 13 }, { 2, 6, 10, 14 }, { 3, 7, 11, 15 },
                                     { 0, 5, 10, 15 }, { 1, 6, 11, 12 }, { 2, 7, 8, 13 }, { 3, 4, 9, 14 } };
        for (int r = 0; r < 12; r++) {
            const uint8_t *s = sg[r % 10];
            for (int g = 0; g < 8; g++) {
                uint64_t &a = v[q[g][0]], &b = v[q[g][1]], &c = v[q[g][2]], &d = v[q[g][3]];
                a += b + m[s[2 * g]];
                d = ror(d ^ a, 32);
                c += d;
                b = ror(b ^ c, 24);
                a += b + m[s[2 * g + 1]];
                d = ror(d ^ a, 16);
                c += d;
                b = ror(b ^ c, 63);
            }
        }
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
    }
    void count(uint64_t inc) {
        t0 += inc;
        if (t0 < inc) t1++;
    }
    void update(const uint8_t *p, size_t n) {
        while (n) {
            if (buflen == 128) {
                count(128);
                compress(buf, false);
                buflen = 0;
            }
            size_t take = 128 - buflen;
            if (take > n) take = n;
            memcpy(buf + buflen, p, take);
            buflen += take;
            p += take;
            n -= take;
        }
    }
    void zeros(uint64_t n) {
        static const uint8_t z[4096] = { 0 };
        while (n) {
            const size_t take = n > sizeof(z) ? sizeof(z) : (size_t)n;
            update(z, take);
            n -= take;
        }
    }
    void final(uint8_t out[64]) {
        count(buflen);
        memset(buf + buflen, 0, 128 - buflen);
        compress(buf, true);
        memcpy(out, h, 64);
    }
};
}  // namespace

extern "C" int64_t mrz_rs_encoded_size(int64_t n) {
    if (n < 0) return MRZ_E_ARG;
    const int64_t burst_in = (int64_t)MRZ_RS_K * MRZ_RS_ROWS;
    return (n / burst_in + 1) * (int64_t)MRZ_RS_N * MRZ_RS_ROWS + 64 + 4;  // feof() needs a short read (rs-mrzip.c:125)
}

extern "C" int mrz_rs_encode(mrz_ctx *ctx, const void *in, int64_t n, int where, void *out, int out_where,
                             int64_t out_cap) {
    if (!ctx || n < 0 || (n > 0 && !in) || !out) return MRZ_E_ARG;
    const int64_t total = mrz_rs_encoded_size(n);
    if (out_cap < total) return MRZ_E_ARG;
    if (out_where != MRZ_MEM_HOST && out_where != MRZ_MEM_DEVICE) return MRZ_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t burst_in = (int64_t)MRZ_RS_K * MRZ_RS_ROWS, burst_out = (int64_t)MRZ_RS_N * MRZ_RS_ROWS;
    const int64_t nbursts = n / burst_in + 1;

    // the serial trailer hash runs on a host thread (host input) or after a device->host copy
    std::vector<uint8_t> host_copy;
    const uint8_t *h_in = (const uint8_t *)in;
    const uint8_t *d_in = nullptr;
    int rc = mrz_stage_input(ctx, in, n, where, &d_in);
    if (rc) return rc;
    if (where == MRZ_MEM_DEVICE) {
        host_copy.resize((size_t)n);
        if (n) HIPCHK(ctx, hipMemcpy(host_copy.data(), in, (size_t)n, hipMemcpyDeviceToHost));
        h_in = host_copy.data();
    }
    uint8_t digest[64];
    std::thread hasher([&]() {
        HostB2 b;
        b.update(h_in, (size_t)n);
        b.zeros((uint64_t)(nbursts * burst_in - n));  // the padded rows are hashed too (rs-mrzip.c:132-138)
        b.final(digest);
    });

    if (!ctx->d_rs_tables) {
        mrz_rs_tables *T = (mrz_rs_tables *)malloc(sizeof(mrz_rs_tables));
        void *p = nullptr;
        hipError_t e = T ? hipMalloc(&p, sizeof(mrz_rs_tables)) : hipErrorOutOfMemory;
        if (e == hipSuccess) {
            mrz_rs_build_tables(T);
            e = hipMemcpy(p, T, sizeof(mrz_rs_tables), hipMemcpyHostToDevice);
        }
        free(T);
        if (e != hipSuccess) {
            hasher.join();
            ctx->last_err = e;
            return MRZ_E_NOMEM;
        }
        ctx->d_rs_tables = p;
    }
    uint8_t *d_out = (uint8_t *)out;
    if (out_where == MRZ_MEM_HOST) {
        rc = mrz_grow(ctx, &ctx->d_rs_out, &ctx->rs_out_cap, nbursts * burst_out);
        if (rc) {
            hasher.join();
            return rc;
        }
        d_out = ctx->d_rs_out;
    }
    hipError_t e = hipSuccess;
    hipEvent_t ea = nullptr, eb = nullptr;
    if (ctx->profiling) {
        hipEventCreate(&ea);
        hipEventCreate(&eb);
        hipEventRecord(ea, ctx->stream);
    }
    hipLaunchKernelGGL(mrz_rs_encode_kernel, dim3((unsigned)(nbursts * (MRZ_RS_ROWS / MRZ_RS_TILE))), dim3(MRZ_RS_THREADS),
                       0, ctx->stream, d_in, n, (const mrz_rs_tables *)ctx->d_rs_tables, d_out);
    e = hipGetLastError();
    if (ctx->profiling) hipEventRecord(eb, ctx->stream);
    if (e == hipSuccess && out_where == MRZ_MEM_HOST)
        e = hipMemcpyAsync(out, d_out, (size_t)(nbursts * burst_out), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (ctx->profiling) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ea, eb) == hipSuccess) ctx->timings.encode_ms = ms;  // reported via mrz_get_timings
        hipEventDestroy(ea);
        hipEventDestroy(eb);
    }
    hasher.join();
    if (e != hipSuccess) {
        ctx->last_err = e;
        return MRZ_E_HIP;
    }
    // trailer: BLAKE2b-512 + k_i, k_j (first short row and its length, rs-mrzip.c:128-136,148-157)
    const int64_t rem = n - (nbursts - 1) * burst_in;
    const unsigned k_i = (unsigned)(rem / MRZ_RS_K), k_j = (unsigned)(rem % MRZ_RS_K);
    uint8_t tail[68];
    memcpy(tail, digest, 64);
    tail[64] = (uint8_t)(k_i & 0xff);
    tail[65] = (uint8_t)(k_i >> 8);
    tail[66] = (uint8_t)(k_j & 0xff);
    tail[67] = (uint8_t)(k_j >> 8);
    if (out_where == MRZ_MEM_HOST)
        memcpy((uint8_t *)out + nbursts * burst_out, tail, 68);
    else
        HIPCHK(ctx, hipMemcpy((uint8_t *)out + nbursts * burst_out, tail, 68, hipMemcpyHostToDevice));
    return MRZ_OK;
}


// ---- rs-mrzip decoder (rs-mrzip/rs-mrzip.c:37-117 decode(), reed-solomon.c:143-309 rsd32, :323-333 gather) ------
// One lane per codeword: the 8176 rows of a burst are independent.  A workgroup takes 128 rows: the interleaved
// input is read column by column (byte c of row r at c * 8176 + r: coalesced across the lanes), converted to the
// conventional basis (tal1tab) into an LDS image, and the 32 syndromes are accumulated on the way.  Rows whose
// syndromes vanish -- all of them on undamaged input -- are done; the others run Berlekamp-Massey, the Chien search
// and Forney's formula exactly as rsd32 does (same field, same roots alpha^(11 (112 + i)), same order of operations,
// so that miscorrections and "uncorrectable" verdicts agree too), each lane on its own row.  The image goes back
// through taltab and out as 223 data bytes per row, row-major.
#define MRZ_RSD_ROWS 128
#define MRZ_RSD_STRIDE 260  // bytes per LDS row (65 words: lanes of a wave hit different banks)

__device__ static int mrz_rsd_slow(uint8_t *data, const int *s_in, const uint8_t *ex, const uint8_t *lg) {
    // s_in[1..32]: syndromes in index form (255 = zero).  no_eras = 0 (rs-mrzip never passes erasures).
    int lambda[33], b[33], t[33], omega[33], reg[33], root[32], loc[32], s[33];
    for (int i = 1; i <= 32; i++) s[i] = s_in[i];
    for (int i = 0; i < 33; i++) lambda[i] = 0;
    lambda[0] = 1;
    for (int i = 0; i < 33; i++) b[i] = lg[lambda[i]];
    int r = 0, el = 0;
    while (++r <= 32) {  // Berlekamp-Massey, :203-232
        int discr = 0;
        for (int i = 0; i < r; i++)
            if (lambda[i] != 0 && s[r - i] != 255) discr ^= ex[(lg[lambda[i]] + s[r - i]) % 255];
        discr = lg[discr];
        if (discr == 255) {
            for (int i = 32; i > 0; i--) b[i] = b[i - 1];
            b[0] = 255;
        } else {
            t[0] = lambda[0];
            for (int i = 0; i < 32; i++) t[i + 1] = b[i] != 255 ? (lambda[i + 1] ^ ex[(discr + b[i]) % 255]) : lambda[i + 1];
            if (2 * el <= r - 1) {
                el = r - el;
                for (int i = 0; i <= 32; i++) b[i] = lambda[i] == 0 ? 255 : (lg[lambda[i]] - discr + 255) % 255;
            } else {
                for (int i = 32; i > 0; i--) b[i] = b[i - 1];
                b[0] = 255;
            }
            for (int i = 0; i < 33; i++) lambda[i] = t[i];
        }
    }
    int deg_lambda = 0;
    for (int i = 0; i < 33; i++) {
        lambda[i] = lg[lambda[i]];
        if (lambda[i] != 255) deg_lambda = i;
    }
    for (int i = 1; i <= 32; i++) reg[i] = lambda[i];
    int count = 0;
    for (int i = 1, k = 139; i <= 255; i++, k = (k + 139) % 255) {  // Chien search, :244-258
        int q = 1;
        for (int j = deg_lambda; j > 0; j--)
            if (reg[j] != 255) {
                reg[j] = (reg[j] + j) % 255;
                q ^= ex[reg[j]];
            }
        if (q != 0) continue;
        root[count] = i;
        loc[count] = k;
        if (++count == deg_lambda) break;
    }
    if (deg_lambda != count) return -1;  // uncorrectable, :259-264
    int deg_omega = 0;
    for (int i = 0; i < 32; i++) {  // omega(x) = s(x) lambda(x) mod x^32, :267-276
        int tmp = 0;
        for (int j = deg_lambda < i ? deg_lambda : i; j >= 0; j--)
            if (s[i + 1 - j] != 255 && lambda[j] != 255) tmp ^= ex[(s[i + 1 - j] + lambda[j]) % 255];
        if (tmp != 0) deg_omega = i;
        omega[i] = lg[tmp];
    }
    omega[32] = 255;
    for (int j = count - 1; j >= 0; j--) {  // Forney, :280-301
        int num1 = 0;
        for (int i = deg_omega; i >= 0; i--)
            if (omega[i] != 255) num1 ^= ex[(omega[i] + i * root[j]) % 255];
        const int num2 = ex[(root[j] * 111) % 255];
        int den = 0;
        for (int i = (deg_lambda < 31 ? deg_lambda : 31) & ~1; i >= 0; i -= 2)
            if (lambda[i + 1] != 255) den ^= ex[(lambda[i + 1] + i * root[j]) % 255];
        if (den == 0) return -1;  // (what has been applied so far stays applied, as in the reference)
        if (num1 != 0) data[loc[j]] ^= ex[(lg[num1] + lg[num2] + 255 - lg[den]) % 255];
    }
    return count;
}

__global__ __launch_bounds__(MRZ_RSD_ROWS) void mrz_rs_decode_kernel(const uint8_t *__restrict__ in, int64_t nbursts,
                                                                     const mrz_rs_tables *__restrict__ T,
                                                                     uint8_t *__restrict__ out, int *__restrict__ counts) {
    __shared__ uint8_t s_ex[256], s_lg[256], s_tal[256], s_tal1[256];
    __shared__ __attribute__((aligned(4))) uint8_t s_img[MRZ_RSD_ROWS * MRZ_RSD_STRIDE];
    const int tid = threadIdx.x;
    for (int i = tid; i < 256; i += MRZ_RSD_ROWS) {
        s_ex[i] = T->ex[i];
        s_lg[i] = T->lg[i];
        s_tal[i] = T->tal[i];
        s_tal1[i] = T->tal1[i];
    }
    __syncthreads();
    const int tiles = (MRZ_RS_ROWS + MRZ_RSD_ROWS - 1) / MRZ_RSD_ROWS;  // 64: the last one holds 112 rows
    const int64_t burst = blockIdx.x / tiles;
    const int row0 = (int)(blockIdx.x % tiles) * MRZ_RSD_ROWS;
    const int nrows = MRZ_RS_ROWS - row0 < MRZ_RSD_ROWS ? MRZ_RS_ROWS - row0 : MRZ_RSD_ROWS;
    const uint8_t *src = in + burst * (int64_t)MRZ_RS_N * MRZ_RS_ROWS + row0;
    uint8_t *row = &s_img[tid * MRZ_RSD_STRIDE];
    int count = 0;
    if (tid < nrows) {
        // gather + dual -> conventional + syndromes: s[i] = sum_j data[j] alpha^((111 + i) 11 j), :156-166
        int s[33], pw[33];
#pragma unroll
        for (int i = 1; i <= 32; i++) {
            s[i] = 0;
            pw[i] = 0;
        }
        for (int c = 0; c < MRZ_RS_N; c++) {
            const int d = s_tal1[src[(int64_t)c * MRZ_RS_ROWS + tid]];
            row[c] = (uint8_t)d;
            if (d != 0) {
                const int lgd = s_lg[d];
#pragma unroll
                for (int i = 1; i <= 32; i++) {
                    int e = lgd + pw[i];
                    e = e >= 255 ? e - 255 : e;
                    s[i] ^= s_ex[e];
                }
            }
#pragma unroll
            for (int i = 1; i <= 32; i++) {  // exponent of column c + 1
                int e = pw[i] + ((111 + i) * 11) % 255;
                pw[i] = e >= 255 ? e - 255 : e;
            }
        }
        int syn_error = 0;
#pragma unroll
        for (int i = 1; i <= 32; i++) {
            syn_error |= s[i];
            s[i] = s_lg[s[i]];
        }
        if (syn_error) count = mrz_rsd_slow(row, s, s_ex, s_lg);
        counts[burst * MRZ_RS_ROWS + row0 + tid] = count;
    }
    __syncthreads();
    // the data bytes of the rows, back in the dual basis (taltab, :305), row-major
    uint8_t *dst = out + (burst * MRZ_RS_ROWS + row0) * (int64_t)MRZ_RS_K;
    for (int idx = tid; idx < nrows * MRZ_RS_K; idx += MRZ_RSD_ROWS) {
        const int r = idx / MRZ_RS_K, c = idx % MRZ_RS_K;
        dst[idx] = s_tal[s_img[r * MRZ_RSD_STRIDE + c]];
    }
    (void)nbursts;
}

extern "C" int mrz_rs_decode(mrz_ctx *ctx, const void *in, int64_t n, int where, void *out_host, int64_t out_cap,
                             int64_t *out_len, mrz_rs_report *rep) {
    if (!ctx || !in || !out_host || !out_len || n < 0) return MRZ_E_ARG;
    const int64_t burst_in = (int64_t)MRZ_RS_K * MRZ_RS_ROWS, burst_out = (int64_t)MRZ_RS_N * MRZ_RS_ROWS;
    const int64_t nbursts = n / burst_out;
    const int64_t tail = n - nbursts * burst_out;
    if (nbursts < 1) return MRZ_E_CORRUPT;
    if (out_cap < nbursts * burst_in) return MRZ_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint8_t *d_in = nullptr;
    int rc = mrz_stage_input(ctx, in, n, where, &d_in);
    if (rc) return rc;
    if (!ctx->d_rs_tables) {
        mrz_rs_tables *T = (mrz_rs_tables *)malloc(sizeof(mrz_rs_tables));
        void *p = nullptr;
        hipError_t e = T ? hipMalloc(&p, sizeof(mrz_rs_tables)) : hipErrorOutOfMemory;
        if (e == hipSuccess) {
            mrz_rs_build_tables(T);
            e = hipMemcpy(p, T, sizeof(mrz_rs_tables), hipMemcpyHostToDevice);
        }
        free(T);
        if (e != hipSuccess) {
            ctx->last_err = e;
            return MRZ_E_NOMEM;
        }
        ctx->d_rs_tables = p;
    }
    // device output: rows x 223 bytes, then one count per row
    const int64_t rows = nbursts * MRZ_RS_ROWS;
    rc = mrz_grow(ctx, &ctx->d_rs_out, &ctx->rs_out_cap, rows * MRZ_RS_K + rows * 4 + 16);
    if (rc) return rc;
    uint8_t *d_out = ctx->d_rs_out;
    int *d_counts = (int *)(d_out + ((rows * MRZ_RS_K + 15) / 16) * 16);
    const int tiles = (MRZ_RS_ROWS + MRZ_RSD_ROWS - 1) / MRZ_RSD_ROWS;
    hipLaunchKernelGGL(mrz_rs_decode_kernel, dim3((unsigned)(nbursts * tiles)), dim3(MRZ_RSD_ROWS), 0, ctx->stream, d_in,
                       nbursts, (const mrz_rs_tables *)ctx->d_rs_tables, d_out, d_counts);
    HIPCHK(ctx, hipGetLastError());
    std::vector<int> counts((size_t)rows);
    HIPCHK(ctx, hipMemcpyAsync(out_host, d_out, (size_t)(rows * MRZ_RS_K), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(counts.data(), d_counts, (size_t)rows * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    mrz_rs_report r;
    memset(&r, 0, sizeof(r));
    for (int64_t i = 0; i < rows; i++) {  // rs-mrzip.c:103-108
        if (counts[(size_t)i] > 0) r.corrected += counts[(size_t)i];
        if (counts[(size_t)i] == -1) r.uncorrectable++;
    }
    int64_t produced = rows * MRZ_RS_K;
    if (tail == 64 + 4) {
        // trailer: BLAKE2b-512 of every 223-byte row as decoded, then the first short row and its length (:70-95)
        uint8_t trailer[68], digest[64];
        if (where == MRZ_MEM_HOST)
            memcpy(trailer, (const uint8_t *)in + nbursts * burst_out, 68);
        else
            HIPCHK(ctx, hipMemcpy(trailer, (const uint8_t *)in + nbursts * burst_out, 68, hipMemcpyDeviceToHost));
        HostB2 b;
        b.update((const uint8_t *)out_host, (size_t)produced);
        b.final(digest);
        r.checksum_ok = memcmp(digest, trailer, 64) == 0;
        const int64_t k_i = trailer[64] | trailer[65] << 8, k_j = trailer[66] | trailer[67] << 8;
        if (k_i < MRZ_RS_ROWS) {
            const int64_t cut = (nbursts - 1) * burst_in + k_i * MRZ_RS_K + (k_j < MRZ_RS_K ? k_j : MRZ_RS_K);
            if (cut < produced) produced = cut;
        }
    } else
        r.truncated = 1;  // "file truncated. can't validate the checksum or remove superfluous 0x00 padding" (:58-68)
    *out_len = produced;
    if (rep) *rep = r;
    return MRZ_OK;
}
