// Measurement only (not part of the product library): how fast does ONE workgroup per CU (4 waves) drain a 256 x 128 fp32
// accumulator tile to HBM, by store pattern?  Built and driven by tools/store_probe.py.
//   0: the MFMA 32x32 C layout as gemm_bfs_kernel stores it now: lane = column, 16 dword stores per block (2 rows x 128 B each)
//   1: the transposed layout (operands swapped in the MFMA): lane = row, 4 dwordx4 stores per block (32 rows x 32 B each)
//   2: row-contiguous dwordx4: 4 rows x 256 B per instruction (what a transpose through LDS would give)
//   3: pattern 0 through buffer stores with the row in the SGPR offset
#include <hip/hip_runtime.h>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(256, 1) void store_probe(float* C, int ldc, int tiles_n, int ntiles, int reps) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, kh = lane >> 5;
  float acc[8][16];
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[b][e] = (float)(t + b * 16 + e);
  for (int r = 0; r < reps; ++r) {
    const int tile = (blockIdx.x + r * gridDim.x) % ntiles;
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    float* base = C + ((int64_t)tm * 256 + wm * 128) * ldc + tn * 128 + wn * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int b = i * 2 + j;
        if (PAT == 0) {
          float* p = base + (int64_t)(i * 32 + 4 * kh) * ldc + j * 32 + l31;
#pragma unroll
          for (int e = 0; e < 16; ++e) p[(unsigned)((e & 3) + 8 * (e >> 2)) * (unsigned)ldc] = acc[b][e];
        } else if (PAT == 1) {
          float* p = base + (int64_t)(i * 32 + l31) * ldc + j * 32 + 4 * kh;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f4*>(p + 8 * q) = (f4){acc[b][4 * q], acc[b][4 * q + 1], acc[b][4 * q + 2], acc[b][4 * q + 3]};
        } else if (PAT == 2) {
          // block b = 32 rows x 32 columns re-imagined as rows of 64 floats: 16 lanes x 16 B per row, 4 rows per instruction
          float* p = base + (int64_t)(i * 32 + (lane >> 4)) * ldc + (lane & 15) * 4;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f4*>(p + (int64_t)(4 * q + 16 * j) * ldc) = (f4){acc[b][4 * q], acc[b][4 * q + 1], acc[b][4 * q + 2], acc[b][4 * q + 3]};
        }
      }
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) asm volatile("" : "+v"(acc[b][e]));
  }
}

extern "C" int store_probe_launch(int pat, float* C, int M, int N, int grid, int reps, void* stream) {
  const int tiles_n = N / 128, ntiles = (M / 256) * tiles_n;
  if (pat == 0) hipLaunchKernelGGL(store_probe<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, C, N, tiles_n, ntiles, reps);
  else if (pat == 1) hipLaunchKernelGGL(store_probe<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, C, N, tiles_n, ntiles, reps);
  else hipLaunchKernelGGL(store_probe<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, C, N, tiles_n, ntiles, reps);
  return (int)hipGetLastError();
}
