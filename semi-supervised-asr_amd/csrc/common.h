// common.h — shared device helpers for libasr_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/asr_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ASR_CHECK_LAUNCH()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

static inline bool asr_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

__device__ __forceinline__ float asr_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// ---------------------------------------------------------------------------------------
// Skinny MFMA tile: partial[MT*16 rows][16 cols] = A[rows, K] * Bt[16 rows of Bt, K]^T.
// K is split over the 4 waves of a 256-thread workgroup in 16-wide chunks; v_mfma_f32_16x16x4
// operands come straight from global/L2 as float4 with a K-permutation inside each chunk
// (lane quad q holds k = 16c+4q..+3 for BOTH operands, element e feeds MFMA e — any
// consistent assignment of k to MFMA slots is a valid contraction order).
//   A operand lane map (16x16x4 f32): a = A[row = lane&15][k-slot = lane>>4]
//   B operand lane map              : b = B[k-slot = lane>>4][col = lane&15]
//   C/D: lane holds rows 4*(lane>>4)+i (i=0..3) of column lane&15.
// Rows >= nrows are clamped on load (caller masks the store).  `amask` (optional) multiplies
// A elementwise for k >= mask_from (dropout on part of the operand).
// Result lands in LDS red[w][MT*16][17] per wave; caller syncs and sums the 4 waves.
// ---------------------------------------------------------------------------------------
#define SK_LDS_STRIDE 17
template <int MT>
__device__ __forceinline__ void skinny_partial(const float* __restrict__ A, int64_t lda, int64_t row0,
                                               int64_t nrows, const float* __restrict__ Bt, int64_t ldb,
                                               int64_t bt_row0, int64_t bt_nrows, int K,
                                               const float* __restrict__ amask,
                                               int64_t ldmask, int mask_from, float* red /* [4][MT*16][17] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  f32x4 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int64_t br = bt_row0 + r < bt_nrows ? bt_row0 + r : bt_nrows - 1;  // clamp; caller masks the store
  const float* brow = Bt + br * ldb;
  int64_t arow[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    int64_t rr = row0 + m * 16 + r;
    arow[m] = rr < nrows ? rr : nrows - 1;
  }
  const int nchunk = K >> 4;
  for (int cidx = wave; cidx < nchunk; cidx += 4) {
    const int k0 = (cidx << 4) + (q << 2);
    const float4 bv = *reinterpret_cast<const float4*>(brow + k0);
    float4 av[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) av[m] = *reinterpret_cast<const float4*>(A + arow[m] * lda + k0);
    if (amask != nullptr && k0 >= mask_from) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float4 mv = *reinterpret_cast<const float4*>(amask + arow[m] * ldmask + (k0 - mask_from));
        av[m].x *= mv.x; av[m].y *= mv.y; av[m].z *= mv.z; av[m].w *= mv.w;
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m].x, bv.x, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m].y, bv.y, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m].z, bv.z, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m].w, bv.w, acc[m], 0, 0, 0);
    }
  }
  float* mine = red + wave * (MT * 16 * SK_LDS_STRIDE);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) mine[(m * 16 + q * 4 + i) * SK_LDS_STRIDE + r] = acc[m][i];
}

// sum of the 4 waves' partials for (row, col) after a __syncthreads()
template <int MT>
__device__ __forceinline__ float skinny_reduced(const float* red, int row, int col) {
  const int o = row * SK_LDS_STRIDE + col;
  const int st = MT * 16 * SK_LDS_STRIDE;
  return (red[o] + red[o + st]) + (red[o + 2 * st] + red[o + 3 * st]);
}
