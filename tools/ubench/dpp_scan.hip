#include <hip/hip_runtime.h>
#include "../../modern-rzip_amd/csrc/mrz_device.h"
__global__ void k(const int *in, int *out, unsigned long long *o2) {
  int lane = threadIdx.x & 63;
  out[threadIdx.x] = mrz_wave_incl_sum(in[threadIdx.x], lane);
  o2[threadIdx.x] = mrz_wave_incl_max64((unsigned long long)(unsigned)in[threadIdx.x] * 1000003ull, lane);
}
int main() {
  int h[128], r[128]; unsigned long long r2[128];
  for (int i = 0; i < 128; i++) h[i] = (i * 37 + 11) % 101;
  int *d, *o; unsigned long long *o2; hipMalloc(&d, 512); hipMalloc(&o, 512); hipMalloc(&o2, 1024);
  hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
  k<<<1, 128>>>(d, o, o2); hipMemcpy(r, o, 512, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 2; w++) { int acc = 0; unsigned long long mx = 0; for (int i = 0; i < 64; i++) { acc += h[w*64+i]; unsigned long long kk = (unsigned long long)h[w*64+i]*1000003ull; if (kk > mx) mx = kk; if (r[w*64+i] != acc || r2[w*64+i] != mx) bad++; } }
  printf("dpp scan %s (%d bad)\n", bad ? "FAIL" : "OK", bad); return bad != 0;
}
