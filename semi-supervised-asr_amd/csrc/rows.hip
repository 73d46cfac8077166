// rows.hip — packed rows of the encoder (include/asr_hip.h, "Packed rows"): the padded batch-major tensors at the two
// ends of the encoder <-> the padding-free row layout its kernels run on.
//   asr_rows_pack_f32        x [B][T][C] (zero padded, model.py:79's input)  -> rows [R][C]
//   asr_rows_unpack_fwd_f32  rows [R][C] -> enc_h [B][T][C]; frames past an utterance's length hold what the reference's
//                            last projection makes of a zero frame: dropout(relu(bias))
//   asr_rows_unpack_bwd_f32  the inverse scatter + the gradient of that fill vector
// Pure HBM streaming, one float4 per lane.
#include "common.h"

namespace {

// grid (ceil(T / 4), B): block (bx, b) moves times 4 bx .. 4 bx + 3 of utterance b
__global__ void rows_pack_kernel(int T, int C4, const float4* __restrict__ x, const int32_t* __restrict__ lens,
                                 const int32_t* __restrict__ rowbase, const int32_t* __restrict__ rowext,
                                 float4* __restrict__ out, int ext_max) {
  const int b = blockIdx.y;
  const int len = lens[b], ext = rowext[b];
  const int64_t base = rowbase[b];
  for (int tt = 0; tt < 4; ++tt) {
    const int t = 4 * blockIdx.x + tt;
    if (t >= ext) return;
    for (int c = threadIdx.x; c < C4; c += blockDim.x)
      out[(base + t) * C4 + c] = (t < len && t < T) ? x[((int64_t)b * T + t) * C4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

__device__ __forceinline__ float4 pad_mask(const float4* mask, int64_t i4, unsigned long long seed, unsigned thresh, float scale) {
  if (mask) return mask[i4];
  if (!thresh) return make_float4(1.f, 1.f, 1.f, 1.f);
  return make_float4(asr_drop_keep(seed, 4 * i4, thresh) ? scale : 0.f, asr_drop_keep(seed, 4 * i4 + 1, thresh) ? scale : 0.f,
                     asr_drop_keep(seed, 4 * i4 + 2, thresh) ? scale : 0.f, asr_drop_keep(seed, 4 * i4 + 3, thresh) ? scale : 0.f);
}

__global__ void rows_unpack_fwd_kernel(int T, int C4, const float4* __restrict__ rows, const int32_t* __restrict__ lens,
                                       const int32_t* __restrict__ rowbase, const float4* __restrict__ fill, int fill_relu,
                                       const float4* __restrict__ mask, unsigned long long seed, unsigned thresh, float scale,
                                       float4* __restrict__ out) {
  const int b = blockIdx.y;
  const int len = lens[b];
  const int64_t base = rowbase[b];
  for (int tt = 0; tt < 4; ++tt) {
    const int t = 4 * blockIdx.x + tt;
    if (t >= T) return;
    for (int c = threadIdx.x; c < C4; c += blockDim.x) {
      const int64_t o = ((int64_t)b * T + t) * C4 + c;
      float4 v;
      if (t < len) v = rows[(base + t) * C4 + c];
      else {
        v = fill ? fill[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (fill_relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        const float4 m = pad_mask(mask, o, seed, thresh, scale);
        v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
      }
      out[o] = v;
    }
  }
}

// valid frames back to their rows; the padding rows of every block (lens[b] <= t < rowext[b]) get zeros
__global__ void rows_unpack_bwd_kernel(int T, int C4, const float4* __restrict__ dout, const int32_t* __restrict__ lens,
                                       const int32_t* __restrict__ rowbase, const int32_t* __restrict__ rowext,
                                       float4* __restrict__ drows) {
  const int b = blockIdx.y;
  const int len = lens[b], ext = rowext[b];
  const int64_t base = rowbase[b];
  for (int tt = 0; tt < 4; ++tt) {
    const int t = 4 * blockIdx.x + tt;
    if (t >= ext) return;
    for (int c = threadIdx.x; c < C4; c += blockDim.x)
      drows[(base + t) * C4 + c] = (t < len && t < T) ? dout[((int64_t)b * T + t) * C4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// dfill[c] += sum over the padded frames of utterance b of dout * mask.  One block per utterance, FL frame lanes of C4 threads
// each (lane f takes the padded frames len + f, + FL, ...), folded through LDS: ONE atomic per element and utterance.  (More
// blocks per utterance make it slower, not faster: the atomics of a column meet in one L2 line - 8 blocks per utterance
// measured 24 us against 17 for one.)
__global__ __launch_bounds__(512) void rows_fill_grad_kernel(int T, int C4, int FL, const float4* __restrict__ dout,
                                                             const int32_t* __restrict__ lens, const float4* __restrict__ mask,
                                                             unsigned long long seed, unsigned thresh, float scale,
                                                             float* __restrict__ dfill, const float4* __restrict__ relu_of) {
  extern __shared__ float4 fold[];                         // [FL][C4]
  const int b = blockIdx.x;
  const int len = lens[b];
  const int c = threadIdx.x % C4, f = threadIdx.x / C4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (f < FL) {
    for (int t = len + f; t < T; t += FL) {
      const int64_t o = ((int64_t)b * T + t) * C4 + c;
      const float4 g = dout[o], m = pad_mask(mask, o, seed, thresh, scale);
      acc.x += g.x * m.x; acc.y += g.y * m.y; acc.z += g.z * m.z; acc.w += g.w * m.w;
    }
    fold[f * C4 + c] = acc;
  }
  __syncthreads();
  if (f == 0 && len < T) {
    for (int k = 1; k < FL; ++k) {
      const float4 v = fold[k * C4 + c];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const float4 r = relu_of ? relu_of[c] : make_float4(1.f, 1.f, 1.f, 1.f);
    if (r.x > 0.f) atomicAdd(dfill + 4 * c, acc.x);
    if (r.y > 0.f) atomicAdd(dfill + 4 * c + 1, acc.y);
    if (r.z > 0.f) atomicAdd(dfill + 4 * c + 2, acc.z);
    if (r.w > 0.f) atomicAdd(dfill + 4 * c + 3, acc.w);
  }
}

int threads_for(int C4) { return C4 >= 256 ? 256 : (C4 >= 128 ? 128 : 64); }

}  // namespace

extern "C" int asr_rows_pack_f32(int B, int T, int C, const float* x, const int32_t* lens, const int32_t* rowbase,
                                 const int32_t* rowext, int ext_max, float* rows, asr_stream_t stream) {
  if (!x || !lens || !rowbase || !rowext || !rows || B <= 0 || T <= 0 || C <= 0 || ext_max <= 0) return ASR_E_ARG;
  if (C % 4) return ASR_E_SHAPE;
  if (!asr_aligned16(x) || !asr_aligned16(rows)) return ASR_E_ALIGN;
  hipLaunchKernelGGL(rows_pack_kernel, dim3((ext_max + 3) / 4, B), dim3(threads_for(C / 4)), 0, (hipStream_t)stream, T, C / 4,
                     (const float4*)x, lens, rowbase, rowext, (float4*)rows, ext_max);
  ASR_CHECK_LAUNCH();
  return 0;
}

extern "C" int asr_rows_unpack_fwd_f32(int B, int T, int C, const float* rows, const int32_t* lens, const int32_t* rowbase,
                                       const float* fill, int fill_relu, const float* mask, uint64_t seed, float p,
                                       float* out, asr_stream_t stream) {
  if (!rows || !lens || !rowbase || !out || B <= 0 || T <= 0 || C <= 0 || p < 0.f || p >= 1.f) return ASR_E_ARG;
  if (C % 4) return ASR_E_SHAPE;
  if (!asr_aligned16(rows) || !asr_aligned16(out) || (fill && !asr_aligned16(fill)) || (mask && !asr_aligned16(mask))) return ASR_E_ALIGN;
  hipLaunchKernelGGL(rows_unpack_fwd_kernel, dim3((T + 3) / 4, B), dim3(threads_for(C / 4)), 0, (hipStream_t)stream, T, C / 4,
                     (const float4*)rows, lens, rowbase, (const float4*)fill, fill_relu, (const float4*)mask, seed,
                     mask ? 0u : asr_drop_thresh(p), 1.0f / (1.0f - p), (float4*)out);
  ASR_CHECK_LAUNCH();
  return 0;
}

extern "C" int asr_rows_unpack_bwd_f32(int B, int T, int C, const float* dout, const int32_t* lens, const int32_t* rowbase,
                                       const int32_t* rowext, int ext_max, const float* mask, uint64_t seed, float p,
                                       float* drows, float* dfill, const float* relu_of, asr_stream_t stream) {
  if (!dout || !lens || !rowbase || !rowext || !drows || B <= 0 || T <= 0 || C <= 0 || ext_max <= 0 || p < 0.f || p >= 1.f) return ASR_E_ARG;
  if (C % 4 || (dfill && C / 4 > 512)) return ASR_E_SHAPE;          // (checked before anything is launched: no half-done result)
  if (!asr_aligned16(dout) || !asr_aligned16(drows) || (mask && !asr_aligned16(mask)) || (relu_of && !asr_aligned16(relu_of)))
    return ASR_E_ALIGN;
  hipLaunchKernelGGL(rows_unpack_bwd_kernel, dim3((ext_max + 3) / 4, B), dim3(threads_for(C / 4)), 0, (hipStream_t)stream, T, C / 4,
                     (const float4*)dout, lens, rowbase, rowext, (float4*)drows);
  if (dfill) {
    const int C4 = C / 4;
    int FL = 512 / C4;                                     // frame lanes: as many as 512 threads hold, at most 8
    if (FL > 8) FL = 8;
    hipLaunchKernelGGL(rows_fill_grad_kernel, dim3(B), dim3(FL * C4), (size_t)FL * C4 * sizeof(float4), (hipStream_t)stream, T, C4,
                       FL, (const float4*)dout, lens, (const float4*)mask, seed, mask ? 0u : asr_drop_thresh(p), 1.0f / (1.0f - p),
                       dfill, (const float4*)relu_of);
  }
  ASR_CHECK_LAUNCH();
  return 0;
}
