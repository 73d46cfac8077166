// measurement only: what v + row_ror:4 + row_ror:8 leaves in each lane (expected: the sum of the 4 lanes 4 apart)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
  float v = (float)(1 << (threadIdx.x & 15));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, true));
  out[threadIdx.x] = v;
}
int main() {
  float* d; hipMalloc(&d, 64 * 4); k<<<1, 64>>>(d); float h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
  for (int i = 0; i < 16; ++i) printf("%d:%.0f ", i, h[i]);
  printf("\n");
  return 0;
}
