// spinner.hip - a stand-in for a foreign kernel (e.g. an RCCL collective waiting for a late rank) that stays resident on some
// CUs while a persistent XCD-local kernel is launched: `wgs` workgroups of `threads` threads, `lds` bytes of LDS each, ~`regs`
// live VGPRs per lane, spinning for `cycles` shader clocks (bounded).  tools/coresident_probe.py
#include <hip/hip_runtime.h>
template <int NV>
__global__ void spin_kernel(long long cycles, float* sink, unsigned xcd_mask = 0xffu) {
  extern __shared__ float lds[];
  // (xcd_mask: workgroups that land on an XCD outside the mask leave at once - the tile-queue GEMM of DESIGN 6 that runs on
  // the XCDs a small-batch persistent kernel leaves idle)
  if (!((xcd_mask >> (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u)) & 1u)) return;
  float v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = threadIdx.x * 0.001f + i;
  lds[threadIdx.x] = v[0];
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = v[i] * 1.0001f + lds[(threadIdx.x + i) & 63];
    __builtin_amdgcn_s_sleep(8);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += v[i];
  if (s == 12345.678f) sink[0] = s;
}
extern "C" int spin_launch(int wgs, int threads, int lds, int regs, long long cycles, float* sink, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (regs <= 32) hipLaunchKernelGGL((spin_kernel<16>), dim3(wgs), dim3(threads), lds, st, cycles, sink);
  else if (regs <= 72) hipLaunchKernelGGL((spin_kernel<56>), dim3(wgs), dim3(threads), lds, st, cycles, sink);
  else hipLaunchKernelGGL((spin_kernel<110>), dim3(wgs), dim3(threads), lds, st, cycles, sink);
  return (int)hipGetLastError();
}
extern "C" int spin_launch_mask(int wgs, int threads, int lds, int regs, long long cycles, float* sink, unsigned mask, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (regs <= 32) hipLaunchKernelGGL((spin_kernel<16>), dim3(wgs), dim3(threads), lds, st, cycles, sink, mask);
  else if (regs <= 72) hipLaunchKernelGGL((spin_kernel<56>), dim3(wgs), dim3(threads), lds, st, cycles, sink, mask);
  else hipLaunchKernelGGL((spin_kernel<110>), dim3(wgs), dim3(threads), lds, st, cycles, sink, mask);
  return (int)hipGetLastError();
}
