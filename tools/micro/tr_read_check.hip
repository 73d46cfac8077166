// ds_read_b64_tr_b16 semantics check (gfx950): a [4 slots][8 rows][72-column stride] 16-bit image read as the A operand
// of v_mfma_f32_16x16x32_bf16 (lane (m = l & 15, kq = l >> 4) needs rows 0..7 of slot kq at column m): two transposed
// reads per lane, lane 4q + p of a 16-lane group supplying the address of row q, columns 4p .. 4p + 3.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/tr_read_check.hip -o /tmp/tr_read_check && /tmp/tr_read_check
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short buf[4][8][72];
  for (int i = threadIdx.x; i < 4 * 8 * 72; i += 64) (&buf[0][0][0])[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, q = j >> 2, p = j & 3;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)&buf[g][q][4 * p]);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)&buf[g][4 + q][4 * p]);
  for (int e = 0; e < 4; ++e) { out[lane * 8 + e] = (unsigned short)a[e]; out[lane * 8 + 4 + e] = (unsigned short)b[e]; }
}
int main() {
  unsigned short h[4 * 8 * 72], o[64 * 8];
  for (int g = 0; g < 4; ++g) for (int r = 0; r < 8; ++r) for (int c = 0; c < 72; ++c) h[(g * 8 + r) * 72 + c] = (unsigned short)(g * 1000 + r * 100 + c);
  unsigned short *di, *dout;
  hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 8; ++r) { const int want = (l >> 4) * 1000 + r * 100 + (l & 15); if (o[l * 8 + r] != want) { if (bad < 8) printf("lane %d row %d: got %d want %d\n", l, r, o[l * 8 + r], want); ++bad; } }
  printf("%s (%d mismatches)\n", bad ? "MISMATCH" : "ok: lane (col, slot) receives rows 0..7 of its column", bad);
  return bad != 0;
}
