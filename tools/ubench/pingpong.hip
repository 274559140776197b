// cross-CU hand-off latency: block 0 posts a round number, NP blocks poll it (agent-scope loads), the first NA of
// them acknowledge in their own word, block 0 polls the acks.  Reports time per round (two hops).
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ unsigned long long ld(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// layout: flag lines at m[0 + 16*k] (k < nlines), acks at m[1024 + 16*b]
__global__ void pingpong(unsigned long long *m, int iters, int na, int nlines, int sleep, int astride, unsigned long long *out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b == 0) {
        long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 1; i <= iters; i++) {
            if (lane < nlines) st(&m[16 * lane], (unsigned long long)i);
            while (true) {
                bool ok = true;
                unsigned long long v[4] = { (unsigned long long)i, (unsigned long long)i, (unsigned long long)i, (unsigned long long)i };
#pragma unroll
                for (int j = 0; j < 4; j++) if (lane + 64 * j < na) v[j] = ld(&m[1024 + astride * (lane + 64 * j)]);
#pragma unroll
                for (int j = 0; j < 4; j++) ok = ok && v[j] == (unsigned long long)i;
                if (__ballot(!ok) == 0) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        long long t1 = __builtin_amdgcn_s_memtime();
        if (lane < nlines) st(&m[16 * lane], ~0ull);
        if (lane == 0) out[0] = (unsigned long long)(t1 - t0);
    } else {
        const int me = b - 1;
        const unsigned long long *f = &m[16 * (me % nlines)];
        unsigned long long seen = 0;
        long long spins = 0;
        while (spins++ < (1ll << 26)) {
            unsigned long long v = ld(f);
            v = __shfl(v, 0);
            if (v == ~0ull) break;
            if (v != seen) {
                seen = v;
                if (me < na && lane == 0) st(&m[1024 + astride * me], v);
            } else if (sleep == 2) __builtin_amdgcn_s_sleep(2);
            else if (sleep == 8) __builtin_amdgcn_s_sleep(8);
        }
    }
}
int main() {
    unsigned long long *m, *o;
    hipMalloc(&m, 1 << 20); hipMalloc(&o, 64);
    const int iters = 20000;
    int cfgs[][5] = { {1,1,1,2,16}, {16,16,1,2,16}, {16,16,1,2,1}, {64,64,1,2,16}, {64,64,1,2,1}, {128,128,1,2,16}, {128,128,1,2,1}, {240,240,1,2,16}, {240,240,1,2,1}, {240,16,1,2,1} };
    for (auto &c : cfgs) {
        hipMemset(m, 0, 1 << 20);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(pingpong, dim3(1 + c[0]), dim3(64), 0, 0, m, iters, c[1], c[2], c[3], c[4], o);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long r; hipMemcpy(&r, o, 8, hipMemcpyDeviceToHost);
        printf("pollers %3d ackers %3d flag-lines %2d sleep %d ack-stride %2d words: %.0f ns/round (%.0f ticks)\n", c[0], c[1], c[2], c[3], c[4], ms * 1e6 / iters, (double)r / iters);
    }
    return 0;
}
