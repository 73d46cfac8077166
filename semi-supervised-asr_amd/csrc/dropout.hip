// dropout.hip — counter-based inverted dropout (torch.nn.Dropout of model.py:82,95,285) without a stored mask:
//   mask(i) = keep(seed, i) / (1 - p),  keep = hash(seed, i) >= p * 2^32  (common.h: asr_drop_keep)
// over the flat element index i of the tensor the dropout applies to.  Forward and backward regenerate it.
#include "common.h"

namespace {

// op 0: x[i] *= mask(i)                         (in place)
// op 1: out[i] = mask(i)                        (materialised mask: tests, and the decoder's [L][B][O+E] operand mask)
// op 2: out[i] = g[i] * mask(i) * (y[i] > 0)    (backward of relu -> dropout; y = the dropped-out output)
template <int OP>
__global__ void dropout_kernel(int64_t n4, float4* __restrict__ x, const float4* __restrict__ g,
                               const float4* __restrict__ y, unsigned long long seed, unsigned thresh, float scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float m[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = (!thresh || asr_drop_keep(seed, 4 * i + k, thresh)) ? scale : 0.f;
    float4 v;
    if (OP == 0) {
      v = x[i];
      v.x *= m[0]; v.y *= m[1]; v.z *= m[2]; v.w *= m[3];
    } else if (OP == 1) {
      v = make_float4(m[0], m[1], m[2], m[3]);
    } else {
      const float4 gv = g[i], yv = y[i];
      v.x = yv.x > 0.f ? gv.x * m[0] : 0.f; v.y = yv.y > 0.f ? gv.y * m[1] : 0.f;
      v.z = yv.z > 0.f ? gv.z * m[2] : 0.f; v.w = yv.w > 0.f ? gv.w * m[3] : 0.f;
    }
    x[i] = v;
  }
}

template <int OP>
int launch(int64_t n, float* x, const float* g, const float* y, uint64_t seed, float p, hipStream_t stream) {
  if (!x || n <= 0 || p < 0.f || p >= 1.f) return ASR_E_ARG;
  if (n % 4) return ASR_E_SHAPE;
  if (!asr_aligned16(x) || (g && !asr_aligned16(g)) || (y && !asr_aligned16(y))) return ASR_E_ALIGN;
  const int64_t n4 = n / 4, nb = (n4 + 255) / 256;
  hipLaunchKernelGGL((dropout_kernel<OP>), dim3((unsigned)(nb > 2048 ? 2048 : nb)), dim3(256), 0, stream, n4, (float4*)x,
                     (const float4*)g, (const float4*)y, (unsigned long long)seed, asr_drop_thresh(p), 1.0f / (1.0f - p));
  ASR_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int asr_dropout_seeded_f32(int64_t n, float* x, uint64_t seed, float p, asr_stream_t stream) {
  return launch<0>(n, x, nullptr, nullptr, seed, p, (hipStream_t)stream);
}
extern "C" int asr_dropout_mask_f32(int64_t n, float* mask, uint64_t seed, float p, asr_stream_t stream) {
  return launch<1>(n, mask, nullptr, nullptr, seed, p, (hipStream_t)stream);
}
extern "C" int asr_relu_dropout_bwd_f32(int64_t n, const float* grad, const float* y, uint64_t seed, float p, float* out,
                                        asr_stream_t stream) {
  if (!grad || !y) return ASR_E_ARG;
  return launch<2>(n, out, grad, y, seed, p, (hipStream_t)stream);
}
