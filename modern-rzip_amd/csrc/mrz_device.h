// mrz_device.h -- small device helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mrz_common.h"

typedef unsigned long long mrz_u64;

// 16-byte load from an arbitrarily aligned address (gfx950 global memory
// handles unaligned dwordx4; the builtin memcpy keeps the compiler honest
// about the alignment it may assume).
__device__ __forceinline__ uint4 mrz_ld16(const uint8_t *p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

__device__ __forceinline__ uint2 mrz_ld8(const uint8_t *p) {
    uint2 v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

__device__ __forceinline__ uint32_t mrz_ld4(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// index of the first differing byte of two 16-byte pieces (16 = identical)
__device__ __forceinline__ int mrz_first_diff16(uint4 a, uint4 b) {
    uint32_t d0 = a.x ^ b.x, d1 = a.y ^ b.y, d2 = a.z ^ b.z, d3 = a.w ^ b.w;
    if (d0) return (__ffs((int)d0) - 1) >> 3;
    if (d1) return 4 + ((__ffs((int)d1) - 1) >> 3);
    if (d2) return 8 + ((__ffs((int)d2) - 1) >> 3);
    if (d3) return 12 + ((__ffs((int)d3) - 1) >> 3);
    return 16;
}

// number of equal bytes counted from byte 15 downwards (16 = identical)
__device__ __forceinline__ int mrz_top_equal16(uint4 a, uint4 b) {
    uint32_t d0 = a.x ^ b.x, d1 = a.y ^ b.y, d2 = a.z ^ b.z, d3 = a.w ^ b.w;
    if (d3) return __clz((int)d3) >> 3;
    if (d2) return 4 + (__clz((int)d2) >> 3);
    if (d1) return 8 + (__clz((int)d1) >> 3);
    if (d0) return 12 + (__clz((int)d0) >> 3);
    return 16;
}

// The state machines in this library keep their control values wave-uniform.
// Reading them back through v_readlane / v_readfirstlane tells the compiler so:
// the values live in SGPRs and branches on them are scalar branches instead of
// EXEC-masked vector code.
__device__ __forceinline__ int mrz_lane_read(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ int mrz_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t mrz_uni64(int64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
// wave-uniform broadcast of a 64-bit value held by lane `src` (src uniform)
__device__ __forceinline__ int64_t mrz_bcast64(int64_t v, int src) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), src);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// per-lane source index (unlike mrz_bcast64, whose source is wave-uniform)
__device__ __forceinline__ int64_t mrz_shfl64(int64_t v, int src) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)(uint64_t)v, src, MRZ_WAVE);
    const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)((uint64_t)v >> 32), src, MRZ_WAVE);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ int64_t mrz_shfl_xor64(int64_t v, int d) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)(uint64_t)v, d, MRZ_WAVE);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)((uint64_t)v >> 32), d, MRZ_WAVE);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// mask with the low `k` bits set, k in [0, 64]
__device__ __forceinline__ mrz_u64 mrz_low_mask(int k) { return k >= 64 ? ~0ull : ((1ull << k) - 1ull); }

// 0-based position of the k-th (k >= 0) set bit of m; m must have > k bits set
__device__ __forceinline__ int mrz_nth_set(mrz_u64 m, int k) {
    for (int i = 0; i < k; i++) m &= m - 1;
    return __ffsll((long long)m) - 1;
}

// trailing-ones rank used by lesser_bitness (src/rzip.c:248-252): ffsll(~t)
__device__ __forceinline__ int mrz_ones_rank(int64_t t) { return __ffsll((long long)~t); }

// inclusive prefix sum over the wave.  On the device: DPP row shifts inside the 16-lane rows, then the two row
// broadcasts (6 VALU instructions, no LDS crossbar); the emulator build takes the shuffle form.
__device__ __forceinline__ int mrz_wave_incl_sum(int v, int lane) {
#ifdef __HIP_DEVICE_COMPILE__
    (void)lane;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
#else
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl(v, lane - d >= 0 ? lane - d : lane, MRZ_WAVE);
        if (lane >= d) v += o;
    }
    return v;
#endif
}

// inclusive prefix maximum of unsigned 64-bit keys over the wave (0 is the neutral element)
__device__ __forceinline__ mrz_u64 mrz_wave_incl_max64(mrz_u64 v, int lane) {
#ifdef __HIP_DEVICE_COMPILE__
    (void)lane;
#define MRZ_DPP_MAX64(ctrl, rmask, bc)                                                                        \
    do {                                                                                                      \
        const uint32_t lo__ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, ctrl, rmask, 0xf, bc);        \
        const uint32_t hi__ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), ctrl, rmask, 0xf, bc); \
        const mrz_u64 o__ = ((mrz_u64)hi__ << 32) | lo__;                                                     \
        v = o__ > v ? o__ : v;                                                                                \
    } while (0)
    MRZ_DPP_MAX64(0x111, 0xf, true);
    MRZ_DPP_MAX64(0x112, 0xf, true);
    MRZ_DPP_MAX64(0x114, 0xf, true);
    MRZ_DPP_MAX64(0x118, 0xf, true);
    MRZ_DPP_MAX64(0x142, 0xa, false);
    MRZ_DPP_MAX64(0x143, 0xc, false);
#undef MRZ_DPP_MAX64
    return v;
#else
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const mrz_u64 o = (mrz_u64)mrz_shfl64((int64_t)v, lane - d >= 0 ? lane - d : lane);
        if (lane >= d && o > v) v = o;
    }
    return v;
#endif
}

// position of the k-th (0-based) set bit of w; w must have more than k bits set
__device__ __forceinline__ int mrz_select64(mrz_u64 w, int k) {
    int pos = 0;
    uint32_t x = (uint32_t)w;
    int c = __popc(x);
    if (k >= c) {
        k -= c;
        pos = 32;
        x = (uint32_t)(w >> 32);
    }
    c = __popc(x & 0xffffu);
    if (k >= c) {
        k -= c;
        pos += 16;
        x >>= 16;
    }
    x &= 0xffffu;
    c = __popc(x & 0xffu);
    if (k >= c) {
        k -= c;
        pos += 8;
        x >>= 8;
    }
    x &= 0xffu;
    c = __popc(x & 0xfu);
    if (k >= c) {
        k -= c;
        pos += 4;
        x >>= 4;
    }
    x &= 0xfu;
    c = __popc(x & 3u);
    if (k >= c) {
        k -= c;
        pos += 2;
        x >>= 2;
    }
    x &= 3u;
    return pos + (k >= (int)(x & 1u) ? 1 : 0);
}
