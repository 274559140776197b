#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int *o) {
  if (threadIdx.x == 0) {
    int x = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));  // HW_REG_XCC_ID, bits [3:0]
    o[blockIdx.x] = x;
  }
}
int main() {
  int *d; int h[64];
  hipMalloc(&d, 256); k<<<64, 64>>>(d); hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; i++) printf("%d%c", h[i], i % 8 == 7 ? '\n' : ' ');
  return 0;
}
