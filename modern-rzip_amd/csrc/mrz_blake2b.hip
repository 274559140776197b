// mrz_blake2b.hip -- BLAKE2b (RFC 7693) as the co-resident checksum kernel.
//
// Restates common/blake2b.c:85-201 (unkeyed, outlen parameter only, 12 rounds,
// 128-byte blocks, parameter word 0x01010000 ^ outlen at :89).  The hash is a
// strict serial chain per message, so the parallel axis is messages:
//   * mrz_blake2b_batch: one lane per message (ar-mrzip hashes every file on its
//     own, ar-mrzip/ar-mrzip.cpp:139-171), 64 messages per wave;
//   * the streaming triple (common/blake2b.h:47-49) keeps its state in device
//     memory and runs a one-lane kernel per update on the ctx's low-priority
//     side stream, so it shares the GPU with the rzip kernels (rs-mrzip hashes
//     the stream it encodes, rs-mrzip/rs-mrzip.c:119-158).
// Bound: integer ALU latency of one lane (the G chain), not HBM.
#include <string.h>

#include "mrz_ctx.h"
#include "mrz_device.h"

struct mrz_b2_state {
    uint64_t h[8];
    uint64_t t[2];
    uint8_t buf[128];
    uint64_t buflen;
    uint64_t outlen;
};

__device__ static const uint64_t k_b2_iv[8] = { 0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                                0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                                0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL };
__device__ static const uint8_t k_b2_sigma[12][16] = {
    { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }, { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 },
    { 11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4 }, { 7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8 },
    { 9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13 }, { 2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9 },
    { 12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11 }, { 13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10 },
    { 6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5 }, { 10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0 },
    { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }, { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 }
};

__device__ __forceinline__ uint64_t mrz_ror64(uint64_t x, int c) { return (x >> c) | (x << (64 - c)); }

#define MRZ_B2_G(a, b, c, d, x, y)   \
    do {                             \
        a = a + b + (x);             \
        d = mrz_ror64(d ^ a, 32);    \
        c = c + d;                   \
        b = mrz_ror64(b ^ c, 24);    \
        a = a + b + (y);             \
        d = mrz_ror64(d ^ a, 16);    \
        c = c + d;                   \
        b = mrz_ror64(b ^ c, 63);    \
    } while (0)

// blake2b_compress, common/blake2b.c:123-156
__device__ static void mrz_b2_compress(uint64_t h[8], const uint64_t t[2], const uint8_t *blk, bool last) {
    uint64_t m[16], v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        uint64_t w;
        __builtin_memcpy(&w, blk + 8 * i, 8);
        m[i] = w;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        v[i] = h[i];
        v[i + 8] = k_b2_iv[i];
    }
    v[12] ^= t[0];
    v[13] ^= t[1];
    if (last) v[14] = ~v[14];
#pragma unroll
    for (int r = 0; r < 12; r++) {
        const uint8_t *s = k_b2_sigma[r];
        MRZ_B2_G(v[0], v[4], v[8], v[12], m[s[0]], m[s[1]]);
        MRZ_B2_G(v[1], v[5], v[9], v[13], m[s[2]], m[s[3]]);
        MRZ_B2_G(v[2], v[6], v[10], v[14], m[s[4]], m[s[5]]);
        MRZ_B2_G(v[3], v[7], v[11], v[15], m[s[6]], m[s[7]]);
        MRZ_B2_G(v[0], v[5], v[10], v[15], m[s[8]], m[s[9]]);
        MRZ_B2_G(v[1], v[6], v[11], v[12], m[s[10]], m[s[11]]);
        MRZ_B2_G(v[2], v[7], v[8], v[13], m[s[12]], m[s[13]]);
        MRZ_B2_G(v[3], v[4], v[9], v[14], m[s[14]], m[s[15]]);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}

__device__ __forceinline__ void mrz_b2_count(uint64_t t[2], uint64_t inc) {
    t[0] += inc;
    t[1] += (t[0] < inc);
}

// whole message in one go: init + update + final (common/blake2b.c:85-92,161-201)
__device__ static void mrz_b2_oneshot(const uint8_t *msg, int64_t len, uint64_t outlen, uint8_t *digest) {
    uint64_t h[8], t[2] = { 0, 0 };
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = k_b2_iv[i];
    h[0] ^= 0x01010000ULL ^ (uint64_t)(uint8_t)outlen;
    int64_t off = 0;
    // every block except the last one (the last keeps 1..128 bytes; 0 only for an empty message)
    while (len - off > 128) {
        mrz_b2_count(t, 128);
        mrz_b2_compress(h, t, msg + off, false);
        off += 128;
    }
    uint8_t lastblk[128];
    const int64_t rem = len - off;
    for (int i = 0; i < 128; i++) lastblk[i] = i < rem ? msg[off + i] : (uint8_t)0;
    mrz_b2_count(t, (uint64_t)rem);
    mrz_b2_compress(h, t, lastblk, true);
    for (uint64_t i = 0; i < outlen; i++) digest[i] = (uint8_t)(h[i >> 3] >> (8 * (i & 7)));
}

__global__ __launch_bounds__(64) void mrz_blake2b_batch_kernel(const uint8_t *const *__restrict__ msgs,
                                                               const int64_t *__restrict__ lens, int count,
                                                               uint64_t outlen, uint8_t *__restrict__ digests) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    mrz_b2_oneshot(msgs[i], lens[i], outlen, digests + (size_t)i * outlen);
}

// blake2b_update (common/blake2b.c:161-184): one lane advances the state
__global__ __launch_bounds__(64) void mrz_blake2b_update_kernel(mrz_b2_state *__restrict__ st,
                                                                const uint8_t *__restrict__ in, uint64_t inlen) {
    if (threadIdx.x != 0 || inlen == 0) return;
    uint64_t h[8], t[2];
    for (int i = 0; i < 8; i++) h[i] = st->h[i];
    t[0] = st->t[0];
    t[1] = st->t[1];
    uint64_t left = st->buflen;
    const uint64_t fill = 128 - left;
    if (inlen > fill) {
        for (uint64_t i = 0; i < fill; i++) st->buf[left + i] = in[i];
        mrz_b2_count(t, 128);
        mrz_b2_compress(h, t, st->buf, false);
        in += fill;
        inlen -= fill;
        left = 0;
        while (inlen > 128) {
            mrz_b2_count(t, 128);
            mrz_b2_compress(h, t, in, false);
            in += 128;
            inlen -= 128;
        }
    }
    for (uint64_t i = 0; i < inlen; i++) st->buf[left + i] = in[i];
    st->buflen = left + inlen;
    for (int i = 0; i < 8; i++) st->h[i] = h[i];
    st->t[0] = t[0];
    st->t[1] = t[1];
}

// blake2b_final (common/blake2b.c:186-201)
__global__ __launch_bounds__(64) void mrz_blake2b_final_kernel(mrz_b2_state *__restrict__ st,
                                                               uint8_t *__restrict__ digest) {
    if (threadIdx.x != 0) return;
    uint64_t h[8], t[2];
    for (int i = 0; i < 8; i++) h[i] = st->h[i];
    t[0] = st->t[0];
    t[1] = st->t[1];
    mrz_b2_count(t, st->buflen);
    for (uint64_t i = st->buflen; i < 128; i++) st->buf[i] = 0;
    mrz_b2_compress(h, t, st->buf, true);
    for (uint64_t i = 0; i < 64; i++) digest[i] = (uint8_t)(h[i >> 3] >> (8 * (i & 7)));
}

// ---- host side ---------------------------------------------------------------
struct mrz_blake2b {
    mrz_ctx *ctx;
    mrz_b2_state *d_state;
    uint8_t *d_digest;
    uint8_t *d_in;
    int64_t in_cap;
    size_t outlen;
};

static int mrz_side_stream(mrz_ctx *ctx) {
    if (ctx->side_stream) return MRZ_OK;
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);  // lo = least urgent
    HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->side_stream, hipStreamNonBlocking, lo));
    return MRZ_OK;
}

extern "C" int mrz_blake2b_init(mrz_ctx *ctx, mrz_blake2b **out, size_t outlen) {
    if (!ctx || !out || outlen < 1 || outlen > 64) return MRZ_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = mrz_side_stream(ctx);
    if (rc) return rc;
    mrz_blake2b *s = (mrz_blake2b *)calloc(1, sizeof(*s));
    if (!s) return MRZ_E_NOMEM;
    s->ctx = ctx;
    s->outlen = outlen;
    void *p = nullptr;
    if (hipMalloc(&p, sizeof(mrz_b2_state)) != hipSuccess) {
        free(s);
        return MRZ_E_NOMEM;
    }
    s->d_state = (mrz_b2_state *)p;
    if (hipMalloc(&p, 64) != hipSuccess) {
        hipFree(s->d_state);
        free(s);
        return MRZ_E_NOMEM;
    }
    s->d_digest = (uint8_t *)p;
    mrz_b2_state init;
    memset(&init, 0, sizeof(init));
    static const uint64_t iv[8] = { 0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                    0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                    0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL };
    for (int i = 0; i < 8; i++) init.h[i] = iv[i];
    init.outlen = (uint8_t)outlen;
    init.h[0] ^= 0x01010000ULL ^ init.outlen;  // common/blake2b.c:89
    if (hipMemcpy(s->d_state, &init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess) {
        hipFree(s->d_state);
        hipFree(s->d_digest);
        free(s);
        return MRZ_E_HIP;
    }
    *out = s;
    return MRZ_OK;
}

extern "C" int mrz_blake2b_update(mrz_blake2b *s, const void *in, size_t inlen, int where) {
    if (!s || (inlen && !in)) return MRZ_E_ARG;
    if (!inlen) return MRZ_OK;
    mrz_ctx *ctx = s->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint8_t *d = (const uint8_t *)in;
    if (where == MRZ_MEM_HOST) {
        // the previous update may still be reading the staging buffer
        HIPCHK(ctx, hipStreamSynchronize(ctx->side_stream));
        int rc = mrz_grow(ctx, &s->d_in, &s->in_cap, (int64_t)inlen);
        if (rc) return rc;
        HIPCHK(ctx, hipMemcpyAsync(s->d_in, in, inlen, hipMemcpyHostToDevice, ctx->side_stream));
        d = s->d_in;
    } else if (where != MRZ_MEM_DEVICE)
        return MRZ_E_ARG;
    else {
        // device input: whatever the ctx stream has queued so far (e.g. the kernels that produce d_s0 / d_s1 of a
        // chunk) comes first; work queued on other streams is the caller's to order (mrzgpu.h)
        hipEvent_t ev = nullptr;
        HIPCHK(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e = hipEventRecord(ev, ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->side_stream, ev, 0);
        hipEventDestroy(ev);
        if (e != hipSuccess) {
            ctx->last_err = e;
            return MRZ_E_HIP;
        }
    }
    hipLaunchKernelGGL(mrz_blake2b_update_kernel, dim3(1), dim3(64), 0, ctx->side_stream, s->d_state, d,
                       (uint64_t)inlen);
    HIPCHK(ctx, hipGetLastError());
    return MRZ_OK;
}

extern "C" int mrz_blake2b_final(mrz_blake2b *s, void *out_host, size_t outlen) {
    if (!s || !out_host || outlen < s->outlen) return MRZ_E_ARG;
    mrz_ctx *ctx = s->ctx;
    hipError_t e = hipSetDevice(ctx->device);
    uint8_t full[64];
    if (e == hipSuccess) {
        hipLaunchKernelGGL(mrz_blake2b_final_kernel, dim3(1), dim3(64), 0, ctx->side_stream, s->d_state, s->d_digest);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(full, s->d_digest, 64, hipMemcpyDeviceToHost, ctx->side_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->side_stream);
    if (e == hipSuccess) memcpy(out_host, full, s->outlen);
    hipFree(s->d_state);
    hipFree(s->d_digest);
    if (s->d_in) hipFree(s->d_in);
    free(s);
    if (e != hipSuccess) {
        ctx->last_err = e;
        return MRZ_E_HIP;
    }
    return MRZ_OK;
}

extern "C" int mrz_blake2b_batch(mrz_ctx *ctx, const void *const *msgs, const int64_t *lens, int count, int where,
                                 size_t outlen, uint8_t *out_host) {
    if (!ctx || count < 0 || outlen < 1 || outlen > 64 || (count && (!msgs || !lens || !out_host))) return MRZ_E_ARG;
    if (!count) return MRZ_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = mrz_side_stream(ctx);
    if (rc) return rc;
    hipStream_t s = ctx->side_stream;
    // layout of the scratch: ptrs[count] | lens[count] | digests[count*outlen] | (host mode) message bytes
    int64_t total = 0;
    for (int i = 0; i < count; i++) {
        if (lens[i] < 0 || (lens[i] && !msgs[i])) return MRZ_E_ARG;
        total += (lens[i] + 15) & ~15ll;
    }
    const int64_t hdr = (int64_t)count * 16 + (((int64_t)count * (int64_t)outlen + 15) & ~15ll);
    const int64_t need = hdr + (where == MRZ_MEM_HOST ? total : 0) + 64;
    if (need > ctx->b2_scratch_cap || !ctx->b2_scratch) {
        if (ctx->b2_scratch) hipFree(ctx->b2_scratch);
        ctx->b2_scratch = nullptr;
        ctx->b2_scratch_cap = 0;
        void *p = nullptr;
        if (hipMalloc(&p, (size_t)need) != hipSuccess) return MRZ_E_NOMEM;
        ctx->b2_scratch = p;
        ctx->b2_scratch_cap = need;
    }
    uint8_t *base = (uint8_t *)ctx->b2_scratch;
    const uint8_t **d_ptrs = (const uint8_t **)base;
    int64_t *d_lens = (int64_t *)(base + (int64_t)count * 8);
    uint8_t *d_dig = base + (int64_t)count * 16;
    uint8_t *d_data = base + hdr;
    const uint8_t **h_ptrs = (const uint8_t **)malloc((size_t)count * sizeof(void *));
    if (!h_ptrs) return MRZ_E_NOMEM;
    hipError_t e = hipSuccess;
    int64_t off = 0;
    for (int i = 0; i < count && e == hipSuccess; i++) {
        if (where == MRZ_MEM_HOST) {
            h_ptrs[i] = d_data + off;
            if (lens[i]) e = hipMemcpyAsync(d_data + off, msgs[i], (size_t)lens[i], hipMemcpyHostToDevice, s);
            off += (lens[i] + 15) & ~15ll;
        } else
            h_ptrs[i] = (const uint8_t *)msgs[i];
    }
    if (e == hipSuccess) e = hipMemcpyAsync(d_ptrs, h_ptrs, (size_t)count * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_lens, lens, (size_t)count * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(mrz_blake2b_batch_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, s,
                           (const uint8_t *const *)d_ptrs, (const int64_t *)d_lens, count, (uint64_t)outlen, d_dig);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_host, d_dig, (size_t)count * outlen, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    free(h_ptrs);
    if (e != hipSuccess) {
        ctx->last_err = e;
        return MRZ_E_HIP;
    }
    return MRZ_OK;
}
