// pointer-chase latency probe: one wave, dependent 16-B loads, footprints 1 MiB .. 4 GiB
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ void chase(const unsigned long long *buf, unsigned long long start, int iters, unsigned long long *out) {
    unsigned long long idx = start;
    long long t0 = __builtin_amdgcn_s_memtime();
    long long c0 = clock64();
    for (int i = 0; i < iters; i++) idx = buf[idx * 2];  // 16-byte records, first word = next index
    long long t1 = __builtin_amdgcn_s_memtime();
    long long c1 = clock64();
    if (threadIdx.x == 0) { out[0] = idx; out[1] = (unsigned long long)(t1 - t0); out[2] = (unsigned long long)(c1 - c0); }
}
int main() {
    const size_t sizes[] = { 1ull << 20, 16ull << 20, 64ull << 20, 1ull << 30, 4ull << 30 };
    for (size_t sz : sizes) {
        size_t n = sz / 16;
        std::vector<unsigned long long> h(n * 2);
        // random cyclic permutation (Sattolo)
        std::vector<unsigned long long> perm(n);
        for (size_t i = 0; i < n; i++) perm[i] = i;
        unsigned long long s = 88172645463325252ull;
        for (size_t i = n - 1; i > 0; i--) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; size_t j = s % i; std::swap(perm[i], perm[j]); }
        for (size_t i = 0; i < n; i++) h[perm[i] * 2] = perm[(i + 1) % n];
        unsigned long long *d, *o;
        hipMalloc(&d, sz); hipMalloc(&o, 64);
        hipMemcpy(d, h.data(), sz, hipMemcpyHostToDevice);
        const int iters = 20000;
        for (int rep = 0; rep < 2; rep++) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, 0ull, iters, o);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            unsigned long long r[3]; hipMemcpy(r, o, 24, hipMemcpyDeviceToHost);
            printf("footprint %6zu MiB rep %d: %.1f ns/load, %.1f memtime ticks/load, %.1f clock64/load\n", sz >> 20, rep, ms * 1e6 / iters, (double)r[1] / iters, (double)r[2] / iters);
        }
        hipFree(d); hipFree(o);
    }
    return 0;
}
