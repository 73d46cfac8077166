// pyramid.hip — pyramidal time-subsample / pair-concat (model.py:85-92), time-major.
//   fwd: out[t'][b] = [ in[2t'][b] | in[2t'+1][b] ],  odd T: the missing frame replicates in[T-1]
//   bwd: inverse scatter; the replicated frame's gradient folds into din[T-1]
// Pure HBM streaming: one float4 per lane, fully coalesced on both sides, grid-stride capped at
// 2048 workgroups (256 CUs x 8).  Optional dropout mask (input-shaped, pre-scaled) fused in.
#include "common.h"

namespace {

__global__ void pyramid_fwd_kernel(int T, int B, int C4, const float4* __restrict__ in,
                                   const float4* __restrict__ mask, float4* __restrict__ out, int64_t total,
                                   unsigned long long seed, unsigned thresh, float scale) {
  const int64_t rowlen = 2 * (int64_t)C4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t rb = i / rowlen;                 // (t', b)
    const int col = (int)(i - rb * rowlen);
    const int64_t t2 = rb / B, b = rb - t2 * B;
    const int half = col >= C4;
    int64_t tin = 2 * t2 + half;
    if (tin >= T) tin = T - 1;
    const int64_t src = (tin * B + b) * C4 + (col - half * C4);
    float4 v = in[src];
    if (mask) {
      const float4 m = mask[src];
      v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
    } else if (thresh) {                           // seeded dropout: the mask of input element 4*src + k
      v.x = asr_drop_keep(seed, 4 * src, thresh) ? v.x * scale : 0.f;
      v.y = asr_drop_keep(seed, 4 * src + 1, thresh) ? v.y * scale : 0.f;
      v.z = asr_drop_keep(seed, 4 * src + 2, thresh) ? v.z * scale : 0.f;
      v.w = asr_drop_keep(seed, 4 * src + 3, thresh) ? v.w * scale : 0.f;
    }
    out[i] = v;
  }
}

__global__ void pyramid_bwd_kernel(int T, int B, int C4, const float4* __restrict__ dout,
                                   const float4* __restrict__ mask, float4* __restrict__ din, int64_t total,
                                   unsigned long long seed, unsigned thresh, float scale) {
  const int T2 = (T + 1) / 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t rb = i / C4;                     // (t, b)
    const int col = (int)(i - rb * C4);
    const int64_t t = rb / B, b = rb - t * B;
    float4 v = dout[((t >> 1) * B + b) * (2 * (int64_t)C4) + (t & 1) * C4 + col];
    if ((T & 1) && t == T - 1) {
      const float4 r = dout[(((int64_t)T2 - 1) * B + b) * (2 * (int64_t)C4) + C4 + col];
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (mask) {
      const float4 m = mask[i];
      v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
    } else if (thresh) {
      v.x = asr_drop_keep(seed, 4 * i, thresh) ? v.x * scale : 0.f;
      v.y = asr_drop_keep(seed, 4 * i + 1, thresh) ? v.y * scale : 0.f;
      v.z = asr_drop_keep(seed, 4 * i + 2, thresh) ? v.z * scale : 0.f;
      v.w = asr_drop_keep(seed, 4 * i + 3, thresh) ? v.w * scale : 0.f;
    }
    din[i] = v;
  }
}

}  // namespace

static int pyramid_fwd_impl(int T, int B, int C, const float* in, const float* mask, float* out, unsigned long long seed,
                            float p, asr_stream_t stream) {
  if (!in || !out || T <= 0 || B <= 0 || C <= 0) return ASR_E_ARG;
  if (C % 4) return ASR_E_SHAPE;
  if (!asr_aligned16(in) || !asr_aligned16(out) || (mask && !asr_aligned16(mask))) return ASR_E_ALIGN;
  const int T2 = (T + 1) / 2;
  const int64_t total = (int64_t)T2 * B * (2 * C / 4);
  const int64_t nb = (total + 255) / 256;
  hipLaunchKernelGGL(pyramid_fwd_kernel, dim3((unsigned)(nb > 2048 ? 2048 : nb)), dim3(256), 0, (hipStream_t)stream, T,
                     B, C / 4, (const float4*)in, (const float4*)mask, (float4*)out, total, seed, asr_drop_thresh(p),
                     p < 1.f ? 1.0f / (1.0f - p) : 0.f);
  ASR_CHECK_LAUNCH();
  return 0;
}

static int pyramid_bwd_impl(int T, int B, int C, const float* dout, const float* mask, float* din, unsigned long long seed,
                            float p, asr_stream_t stream) {
  if (!dout || !din || T <= 0 || B <= 0 || C <= 0) return ASR_E_ARG;
  if (C % 4) return ASR_E_SHAPE;
  if (!asr_aligned16(dout) || !asr_aligned16(din) || (mask && !asr_aligned16(mask))) return ASR_E_ALIGN;
  const int64_t total = (int64_t)T * B * (C / 4);
  const int64_t nb = (total + 255) / 256;
  hipLaunchKernelGGL(pyramid_bwd_kernel, dim3((unsigned)(nb > 2048 ? 2048 : nb)), dim3(256), 0, (hipStream_t)stream, T,
                     B, C / 4, (const float4*)dout, (const float4*)mask, (float4*)din, total, seed, asr_drop_thresh(p),
                     p < 1.f ? 1.0f / (1.0f - p) : 0.f);
  ASR_CHECK_LAUNCH();
  return 0;
}

extern "C" int asr_pyramid_concat_fwd(int T, int B, int C, const float* in, const float* mask, float* out,
                                      asr_stream_t stream) {
  return pyramid_fwd_impl(T, B, C, in, mask, out, 0ull, 0.f, stream);
}
extern "C" int asr_pyramid_concat_bwd(int T, int B, int C, const float* dout, const float* mask, float* din,
                                      asr_stream_t stream) {
  return pyramid_bwd_impl(T, B, C, dout, mask, din, 0ull, 0.f, stream);
}
// Same with the dropout mask regenerated from (seed, element index of the [T][B][C] input) instead of read: see
// asr_dropout_seeded_f32 for the mask definition.
extern "C" int asr_pyramid_concat_fwd_seeded(int T, int B, int C, const float* in, uint64_t seed, float p, float* out,
                                             asr_stream_t stream) {
  if (p < 0.f || p >= 1.f) return ASR_E_ARG;
  return pyramid_fwd_impl(T, B, C, in, nullptr, out, seed, p, stream);
}
extern "C" int asr_pyramid_concat_bwd_seeded(int T, int B, int C, const float* dout, uint64_t seed, float p, float* din,
                                             asr_stream_t stream) {
  if (p < 0.f || p >= 1.f) return ASR_E_ARG;
  return pyramid_bwd_impl(T, B, C, dout, nullptr, din, seed, p, stream);
}
