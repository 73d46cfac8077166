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
// Hardware-transcendental forms for the pointwise phase of the persistent recurrences, which sits on the serial
// critical path of every time step (the libm forms cost ~250 VALU instructions per cell update, ~0.4 us per step).
// v_exp_f32 / v_rcp_f32 are accurate to ~1-2 ulp; both forms saturate correctly at +-inf.
__device__ __forceinline__ float asr_fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float asr_fast_tanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
}

// Counter-based dropout: keep(element) is a pure function of (seed, flat element index), so the forward and the
// backward regenerate the same mask instead of writing / reading it (229 MB per cfg-2 step).  Two rounds of the
// "lowbias32" integer mixer over the 64-bit index and the 64-bit seed; an element is kept when the hash is >= thresh
// = p * 2^32.  Kept elements are scaled by 1/(1-p) (inverted dropout, as torch.nn.Dropout).
__device__ __forceinline__ unsigned asr_mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool asr_drop_keep(unsigned long long seed, long long idx, unsigned thresh) {
  unsigned h = asr_mix32((unsigned)idx ^ (unsigned)seed);
  h = asr_mix32(h + (unsigned)((unsigned long long)idx >> 32) * 0x9e3779b9u + (unsigned)(seed >> 32));
  return h >= thresh;
}
static inline unsigned asr_drop_thresh(float p) {
  const double t = (double)p * 4294967296.0;
  return t <= 0.0 ? 0u : (t >= 4294967295.0 ? 4294967295u : (unsigned)t);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// Sum over each aligned group of 16 lanes with DPP modifiers on the adds (xor 1, xor 2 by quad permutation; then the
// mirrored half row and the mirrored row bring the other quad / other half once the lower levels are uniform).
// Every lane of the group ends with the total.  4 VALU instructions; a __shfl_xor is a ds_bpermute round trip each.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
  return v;
}
// Four independent 16-lane sums, stage by stage: consecutive DPP instructions never depend on each other, so the
// compiler does not have to pad the VALU-write -> DPP-read hazard with s_nop.
__device__ __forceinline__ void row16_sum4(float (&v)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k)
    v[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[k]), 0xB1, 0xf, 0xf, true));
#pragma unroll
  for (int k = 0; k < 4; ++k)
    v[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[k]), 0x4E, 0xf, 0xf, true));
#pragma unroll
  for (int k = 0; k < 4; ++k)
    v[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[k]), 0x141, 0xf, 0xf, true));
#pragma unroll
  for (int k = 0; k < 4; ++k)
    v[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[k]), 0x140, 0xf, 0xf, true));
}
// Whole-wave sum / max from the row totals: 4 DPP steps inside each row of 16, then the four row totals are read
// with v_readlane (every lane of a row holds its total).  ~12 instructions instead of 6 ds_bpermute round trips.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  const float r = row16_sum(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 0)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 16)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 32)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 48));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false)));
  return v;
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  const float r = row16_max(v);
  return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 0)),
                     __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 16))),
               fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 32)),
                     __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 48))));
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
// Rows >= nrows are clamped on load (caller masks the store).
// The chain is L2-latency bound, so the loop is software pipelined: two statically named register
// groups of SK_GROUP chunks; group g+1's loads are issued before group g's MFMAs.  Loads are
// unconditional and nothing touches a loaded value before its MFMA (a branch around a load, or a
// select right after it, makes hipcc drain vmcnt there and serialises the L2 round trips); the tail
// group clamps its chunk index and zeroes the B operand at MFMA time.
// Result lands in LDS red[w][MT*16][17] per wave; caller syncs and sums the 4 waves.
// ---------------------------------------------------------------------------------------
#define SK_LDS_STRIDE 17
#define SK_GROUP 8

template <int MT>
struct SkinnyRegs {
  float4 b[SK_GROUP];
  float4 a[SK_GROUP][MT];
};

template <int MT, int NW>
__device__ __forceinline__ void skinny_fetch(SkinnyRegs<MT>& rg, int i0, int nmine, int wave, int q, int cbase,
                                             const float* __restrict__ brow, const float* __restrict__ A,
                                             int64_t lda, const int64_t (&arow)[MT]) {
#pragma unroll
  for (int u = 0; u < SK_GROUP; ++u) {
    int i = i0 + u < nmine ? i0 + u : nmine - 1;
#ifndef ASR_NO_ROTATE
    i += cbase;                       // per-workgroup rotation of the K traversal (cbase < nmine):
    i = i < nmine ? i : i - nmine;    // all workgroups share operand A; staggering avoids L2-channel hot spots
#endif
    const int k0 = ((wave + NW * i) << 4) + (q << 2);
#ifdef ASR_ABLATE_NOLOAD
    rg.b[u] = make_float4(k0 * 1e-9f, 1.f, 2.f, 3.f);
#pragma unroll
    for (int m = 0; m < MT; ++m) rg.a[u][m] = make_float4(1.f, k0 * 1e-9f, 2.f, 3.f);
#else
    rg.b[u] = *reinterpret_cast<const float4*>(brow + k0);
#pragma unroll
    for (int m = 0; m < MT; ++m) rg.a[u][m] = *reinterpret_cast<const float4*>(A + arow[m] * lda + k0);
#endif
  }
}

template <int MT, bool FULL>
__device__ __forceinline__ void skinny_mma(const SkinnyRegs<MT>& rg, int i0, int nmine, f32x4 (&acc)[MT]) {
#pragma unroll
  for (int u = 0; u < SK_GROUP; ++u) {
    float4 b = rg.b[u];
    if (!FULL && i0 + u >= nmine) b = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef ASR_ABLATE_NOMMA
#pragma unroll
    for (int m = 0; m < MT; ++m) { acc[m][0] += rg.a[u][m].x * b.x + rg.a[u][m].y * b.y + rg.a[u][m].z * b.z + rg.a[u][m].w * b.w; }
    continue;
#endif
    // alternate the accumulators so consecutive MFMAs are independent (40-cycle dependent latency)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(rg.a[u][m].x, b.x, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(rg.a[u][m].y, b.y, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(rg.a[u][m].z, b.z, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(rg.a[u][m].w, b.w, acc[m], 0, 0, 0);
  }
}

template <int MT>
__device__ __forceinline__ void skinny_mma_any(const SkinnyRegs<MT>& rg, int i0, int nmine, f32x4 (&acc)[MT]) {
  if (i0 + SK_GROUP <= nmine) skinny_mma<MT, true>(rg, i0, nmine, acc);
  else skinny_mma<MT, false>(rg, i0, nmine, acc);
}

// skinny_partial_rows: the rows of A are given per lane - arow[m] = the row of A that feeds tile row m * 16 + (lane & 15)
// (any rows: the packed-row layout of the LSTM kernels has no common stride between batch rows).
template <int MT, int NW = 4>
__device__ __forceinline__ void skinny_partial_rows(const float* __restrict__ A, int64_t lda, const int64_t (&arow)[MT],
                                                    const float* __restrict__ Bt, int64_t ldb,
                                                    int64_t bt_row0, int64_t bt_nrows, int K,
                                                    float* red /* [NW][MT*16][17] */) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  f32x4 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int64_t br = bt_row0 + r < bt_nrows ? bt_row0 + r : bt_nrows - 1;  // clamp; caller masks the store
  const float* brow = Bt + br * ldb;
  const int nchunk = K >> 4;
  const int nmine = (nchunk - wave + NW - 1) / NW;   // this wave owns chunks wave, wave+NW, ...
  const int cbase = nmine > 0 ? (int)((blockIdx.x * 5u + blockIdx.y * 3u + blockIdx.z * 7u) % (unsigned)nmine) : 0;
  if (nmine > 0) {                              // wave-uniform
    SkinnyRegs<MT> ra, rb;
    skinny_fetch<MT, NW>(ra, 0, nmine, wave, q, cbase, brow, A, lda, arow);
    for (int i0 = 0; i0 < nmine; i0 += 2 * SK_GROUP) {
      const bool second = i0 + SK_GROUP < nmine;
      if (second) skinny_fetch<MT, NW>(rb, i0 + SK_GROUP, nmine, wave, q, cbase, brow, A, lda, arow);
      skinny_mma_any<MT>(ra, i0, nmine, acc);
      if (i0 + 2 * SK_GROUP < nmine) skinny_fetch<MT, NW>(ra, i0 + 2 * SK_GROUP, nmine, wave, q, cbase, brow, A, lda, arow);
      if (second) skinny_mma_any<MT>(rb, i0 + SK_GROUP, nmine, acc);
    }
  }
  float* mine = red + wave * (MT * 16 * SK_LDS_STRIDE);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) mine[(m * 16 + q * 4 + i) * SK_LDS_STRIDE + r] = acc[m][i];
}

template <int MT, int NW = 4>
__device__ __forceinline__ void skinny_partial(const float* __restrict__ A, int64_t lda, int64_t row0,
                                               int64_t nrows, const float* __restrict__ Bt, int64_t ldb,
                                               int64_t bt_row0, int64_t bt_nrows, int K,
                                               float* red /* [NW][MT*16][17] */) {
  const int r = threadIdx.x & 15;
  int64_t arow[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    int64_t rr = row0 + m * 16 + r;
    arow[m] = rr < nrows ? rr : nrows - 1;               // clamp; caller masks the store
  }
  skinny_partial_rows<MT, NW>(A, lda, arow, Bt, ldb, bt_row0, bt_nrows, K, red);
}

// sum of the NW waves' partials for (row, col) after a __syncthreads()
template <int MT, int NW = 4>
__device__ __forceinline__ float skinny_reduced(const float* red, int row, int col) {
  const int o = row * SK_LDS_STRIDE + col;
  const int st = MT * 16 * SK_LDS_STRIDE;
  float v = (red[o] + red[o + st]) + (red[o + 2 * st] + red[o + 3 * st]);
  if (NW == 8) v += (red[o + 4 * st] + red[o + 5 * st]) + (red[o + 6 * st] + red[o + 7 * st]);
  return v;
}
