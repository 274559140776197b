// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE: a minimal wave64 SIMT
// emulator so the gfx950 kernel sources under modern-rzip_amd/csrc can be
// compiled with g++ and executed on the CPU (where sanitizers work and where
// the per-round CPU test tier runs).  The product never uses this; it is only
// reached by building with -Itests/emu, which shadows <hip/hip_runtime.h>.
//
// Model: every thread of a workgroup is a ucontext fiber; workgroups run one
// after another.  Wave collectives (__ballot/__shfl*) and __syncthreads are
// rendezvous points between fibers.  Only uniform (all-live-lanes) collectives
// are supported, which is how the kernels are written.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <ucontext.h>

#include <functional>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint2 { uint32_t x, y; };
struct __attribute__((aligned(16))) uint4 { uint32_t x, y, z, w; };
struct __attribute__((aligned(16))) longlong2 { long long x, y; };
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { uint4 v = { a, b, c, d }; return v; }
static inline uint2 make_uint2(uint32_t a, uint32_t b) { uint2 v = { a, b }; return v; }
static inline longlong2 make_longlong2(long long a, long long b) { longlong2 v = { a, b }; return v; }

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef struct emu_stream *hipStream_t;
typedef struct emu_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

namespace emu {
struct Idx { unsigned x, y, z; };
struct Fiber;
extern Fiber *cur;
extern Idx cur_tid, cur_bid, cur_bdim, cur_gdim;
void launch(dim3 grid, dim3 block, const std::function<void()> &body);
unsigned long long ballot(int pred);
int shfl(int v, int src);
int shfl_xor(int v, int mask);
void syncthreads();
int first_live_lane();
void yield();
void request_coresident();
}  // namespace emu

#define threadIdx (emu::cur_tid)
#define blockIdx (emu::cur_bid)
#define blockDim (emu::cur_bdim)
#define gridDim (emu::cur_gdim)

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })

static inline void __syncthreads() { emu::syncthreads(); }
static inline unsigned long long __ballot(int pred) { return emu::ballot(pred); }
static inline int __shfl(int v, int src, int width = 64) { (void)width; return emu::shfl(v, src); }
static inline int __shfl_xor(int v, int m, int width = 64) { (void)width; return emu::shfl_xor(v, m); }
static inline int __builtin_amdgcn_readlane(int v, int src) { return emu::shfl(v, src); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return emu::shfl(v, emu::first_live_lane()); }
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 3
template <typename T> static inline T __hip_atomic_load(T *p, int, int) { return *(volatile T *)p; }
template <typename T> static inline void __hip_atomic_store(T *p, T v, int, int) { *(volatile T *)p = v; }
template <typename T> static inline T __hip_atomic_fetch_add(T *p, T v, int, int) { T o = *p; *p = o + v; return o; }
template <typename T> static inline bool __hip_atomic_compare_exchange_strong(T *p, T *expect, T v, int, int, int) { if (*p == *expect) { *p = v; return true; } *expect = *p; return false; }
static inline void __builtin_amdgcn_s_sleep(int) { emu::yield(); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }
static inline unsigned atomicAdd(unsigned *p, unsigned v) { unsigned o = *p; *p += v; return o; }
static inline int atomicAdd(int *p, int v) { int o = *p; *p += v; return o; }
static inline unsigned atomicMax(unsigned *p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }
static inline unsigned atomicCAS(unsigned *p, unsigned cmp, unsigned v) { unsigned o = *p; if (o == cmp) *p = v; return o; }
static inline unsigned long long atomicMin(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; if (v < o) *p = v; return o; }
static inline unsigned long long atomicCAS(unsigned long long *p, unsigned long long cmp, unsigned long long v) { unsigned long long o = *p; if (o == cmp) *p = v; return o; }
static inline unsigned atomicMin(unsigned *p, unsigned v) { unsigned o = *p; if (v < o) *p = v; return o; }
static inline int atomicMin(int *p, int v) { int o = *p; if (v < o) *p = v; return o; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }

// ---- runtime API subset -----------------------------------------------------
struct emu_event { double t; };
static inline double emu_now_ms() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1 };
static inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 2; return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <typename T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { *p = malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = (hipStream_t)malloc(8); return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { return hipStreamCreate(s); }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreate(s); }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)calloc(1, sizeof(emu_event)); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
#define hipEventDisableTiming 2
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = emu_now_ms(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess (emulator)" : "emulated HIP error"; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return hipSuccess; }
#define hipStreamNonBlocking 1

// ---- virtual memory management subset (mrz_window.hip): physical allocation = memfd, mapping = mmap -------------
// What HIP's VMM gives on the GPU (shareable physical allocations mapped back to back into one reserved address range,
// across processes) is what memfd + MAP_FIXED give on the CPU: the multi-process window tests run on it unchanged.
#include <sys/mman.h>
#include <unistd.h>
typedef struct { int fd; size_t size; } *hipMemGenericAllocationHandle_t;
enum hipMemAllocationType { hipMemAllocationTypePinned = 1 };
enum hipMemLocationType { hipMemLocationTypeDevice = 1 };
enum hipMemAllocationHandleType { hipMemHandleTypePosixFileDescriptor = 1 };
enum hipMemAllocationGranularity_flags { hipMemAllocationGranularityRecommended = 1 };
enum hipMemAccessFlags { hipMemAccessFlagsProtReadWrite = 3 };
struct hipMemLocation { hipMemLocationType type; int id; };
struct hipMemAllocationProp { hipMemAllocationType type; hipMemAllocationHandleType requestedHandleType; hipMemLocation location; };
struct hipMemAccessDesc { hipMemLocation location; hipMemAccessFlags flags; };
static inline hipError_t hipMemGetAllocationGranularity(size_t *g, const hipMemAllocationProp *, hipMemAllocationGranularity_flags) { *g = 65536; return hipSuccess; }
static inline hipError_t hipMemCreate(hipMemGenericAllocationHandle_t *h, size_t size, const hipMemAllocationProp *, unsigned long long) {
    const int fd = memfd_create("mrz_emu_window", 0);
    if (fd < 0 || ftruncate(fd, (off_t)size) != 0) { if (fd >= 0) close(fd); return hipErrorOutOfMemory; }
    *h = (hipMemGenericAllocationHandle_t)malloc(sizeof(**h));
    (*h)->fd = fd;
    (*h)->size = size;
    return hipSuccess;
}
static inline hipError_t hipMemRelease(hipMemGenericAllocationHandle_t h) { if (h) { close(h->fd); free(h); } return hipSuccess; }
static inline hipError_t hipMemExportToShareableHandle(void *out, hipMemGenericAllocationHandle_t h, hipMemAllocationHandleType, unsigned long long) {
    const int d = dup(h->fd);
    if (d < 0) return hipErrorInvalidValue;
    *(int *)out = d;
    return hipSuccess;
}
static inline hipError_t hipMemImportFromShareableHandle(hipMemGenericAllocationHandle_t *h, void *os_handle, hipMemAllocationHandleType) {
    const int d = dup(*(int *)os_handle);
    if (d < 0) return hipErrorInvalidValue;
    const off_t size = lseek(d, 0, SEEK_END);
    if (size <= 0) { close(d); return hipErrorInvalidValue; }
    *h = (hipMemGenericAllocationHandle_t)malloc(sizeof(**h));
    (*h)->fd = d;
    (*h)->size = (size_t)size;
    return hipSuccess;
}
static inline hipError_t hipMemAddressReserve(void **p, size_t size, size_t, void *, unsigned long long) {
    void *a = mmap(nullptr, size, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (a == MAP_FAILED) return hipErrorOutOfMemory;
    *p = a;
    return hipSuccess;
}
static inline hipError_t hipMemAddressFree(void *p, size_t size) { munmap(p, size); return hipSuccess; }
static inline hipError_t hipMemMap(void *p, size_t size, size_t, hipMemGenericAllocationHandle_t h, unsigned long long) {
    if (size > h->size) return hipErrorInvalidValue;
    return mmap(p, size, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, h->fd, 0) == MAP_FAILED ? hipErrorInvalidValue : hipSuccess;
}
static inline hipError_t hipMemUnmap(void *p, size_t size) {
    return mmap(p, size, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED, -1, 0) == MAP_FAILED ? hipErrorInvalidValue : hipSuccess;
}
static inline hipError_t hipMemSetAccess(void *, size_t, const hipMemAccessDesc *, size_t) { return hipSuccess; }
