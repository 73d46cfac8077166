// optim.hip — fused gradient-norm + clip + Adam(amsgrad, L2 weight decay) on a flat fp32 buffer.
// Replaces torch.nn.utils.clip_grad_norm_ + torch.optim.Adam(amsgrad=True).step as called at
// solver.py:152-153,384-385 (and 171-173,296-297 for the judge, without amsgrad/weight decay).
// HBM streaming: float4 per lane, grid-stride; the clip coefficient is read from a device scalar so the
// step needs no host synchronisation.
#include "common.h"

namespace {

__global__ void sumsq_kernel(int64_t n4, int64_t n, const float* __restrict__ g, float* __restrict__ out) {
  float s = 0.f;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = g4[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int64_t i = n4 * 4; i < n; ++i) s += g[i] * g[i];
  s = wave_sum(s);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (part[0] + part[1]) + (part[2] + part[3]));
}

// Gradient gather: the tensors autograd produced -> their slices of the flat buffer, and (optionally) the sum of their
// squares on the way - the two passes in front of the update of a one-process step (torch._foreach_copy_ + asr_sumsq_f32)
// as one.  Jobs travel by value; a block owns GATHER_CHUNK consecutive elements of one job.
constexpr int GATHER_JOBS = ASR_GATHER_MAX_JOBS, GATHER_U = 8, GATHER_CHUNK = 256 * 4 * GATHER_U;
struct GatherJobs {
  int n;
  int first[GATHER_JOBS + 1];          // first block of job j
  const float* src[GATHER_JOBS];
  int64_t dst[GATHER_JOBS];            // element offset in the flat buffer
  int64_t count[GATHER_JOBS];
};
__global__ __launch_bounds__(256) void gather_sumsq_kernel(GatherJobs t, float* __restrict__ flat, float* __restrict__ sumsq) {
  int lo = 0, hi = t.n - 1;            // the job of this block: last j with first[j] <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (t.first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const int j = lo;
  const int64_t e0 = (int64_t)(blockIdx.x - t.first[j]) * GATHER_CHUNK;
  const int64_t e1 = e0 + GATHER_CHUNK < t.count[j] ? e0 + GATHER_CHUNK : t.count[j];
  const float* __restrict__ s = t.src[j];
  float* __restrict__ d = flat + t.dst[j];
  float acc = 0.f;
  if ((((uintptr_t)s) & 15) == 0 && (t.dst[j] & 3) == 0 && e1 - e0 == GATHER_CHUNK) {
    // a whole chunk: all of a thread's loads in flight before its first store (the first version, a loop of load -> store
    // per float4, moved 3.3 TB/s where torch's multi-tensor copy moves 6)
    const float4* s4 = reinterpret_cast<const float4*>(s + e0);
    float4* d4 = reinterpret_cast<float4*>(d + e0);
    float4 v[GATHER_U];
#pragma unroll
    for (int u = 0; u < GATHER_U; ++u) v[u] = s4[threadIdx.x + 256 * u];
#pragma unroll
    for (int u = 0; u < GATHER_U; ++u) {
      d4[threadIdx.x + 256 * u] = v[u];
      acc += v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w;
    }
  } else if ((((uintptr_t)s) & 15) == 0 && (t.dst[j] & 3) == 0) {
    const int64_t n4 = (e1 - e0) >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(s + e0);
    float4* d4 = reinterpret_cast<float4*>(d + e0);
    for (int64_t i = threadIdx.x; i < n4; i += 256) {
      const float4 v = s4[i];
      d4[i] = v;
      acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (int64_t i = e0 + 4 * n4 + threadIdx.x; i < e1; i += 256) { const float v = s[i]; d[i] = v; acc += v * v; }
  } else {
    for (int64_t i = e0 + threadIdx.x; i < e1; i += 256) { const float v = s[i]; d[i] = v; acc += v * v; }
  }
  if (sumsq) {
    acc = wave_sum(acc);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sumsq, (part[0] + part[1]) + (part[2] + part[3]));
  }
}

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float* vmax, float coef, float lr_c1,
                                         float b1, float b2, float eps, float wd, float rs_c2) {
  g = g * coef + wd * p;
  m = b1 * m + (1.f - b1) * g;
  v = b2 * v + (1.f - b2) * g * g;
  float vv = v;
  if (vmax) { vv = fmaxf(*vmax, v); *vmax = vv; }
  p -= lr_c1 * m / (sqrtf(vv) * rs_c2 + eps);
}

__global__ void adam_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, float* __restrict__ vmax, const float* __restrict__ gnorm_sq,
                            float max_norm, float lr_c1, float b1, float b2, float eps, float wd, float rs_c2,
                            const unsigned* __restrict__ skip, float* __restrict__ zero_word) {
  // (the accumulator the NEXT step's asr_sumsq_f32 adds into: zeroed here, also by a skipped update, so that no fill launch
  // precedes the norm - the caller alternates between two words)
  if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0.f;
  // device-side predicate: a non-zero word (the abort latch of the persistent kernels, or the all-reduced latch sum of a
  // data-parallel step) turns the whole update into a no-op, so the host can launch it without reading the latch first
  if (skip && *skip != 0u) return;
  float coef = 1.f;
  if (gnorm_sq) {
    const float nrm = sqrtf(*gnorm_sq);
    coef = fminf(1.f, max_norm / (nrm + 1e-6f));
  }
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    adam_one(p[i], g[i], m[i], v[i], vmax ? vmax + i : nullptr, coef, lr_c1, b1, b2, eps, wd, rs_c2);
}

}  // namespace

extern "C" int asr_sumsq_f32(int64_t n, const float* g, float* out, asr_stream_t stream) {
  if (!g || !out || n <= 0) return ASR_E_ARG;
  if (!asr_aligned16(g)) return ASR_E_ALIGN;
  const int64_t n4 = n / 4;
  int64_t nb = (n4 + 255) / 256;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)(nb > 1024 ? 1024 : nb)), dim3(256), 0, (hipStream_t)stream, n4, n, g,
                     out);
  ASR_CHECK_LAUNCH();
  return 0;
}

extern "C" int asr_adam_clip_f32(int64_t n, float* p, const float* g, float* m, float* v, float* vmax,
                                 const float* gnorm_sq, float max_norm, float lr, float beta1, float beta2, float eps,
                                 float weight_decay, float bias_c1, float bias_c2, const void* skip_if_nonzero,
                                 float* zero_word, asr_stream_t stream) {
  if (!p || !g || !m || !v || n <= 0) return ASR_E_ARG;
  const int64_t nb = (n + 255) / 256;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(nb > 2048 ? 2048 : nb)), dim3(256), 0, (hipStream_t)stream, n, p, g,
                     m, v, vmax, gnorm_sq, max_norm, lr / bias_c1, beta1, beta2, eps, weight_decay,
                     1.0f / sqrtf(bias_c2), (const unsigned*)skip_if_nonzero, zero_word);
  ASR_CHECK_LAUNCH();
  return 0;
}

extern "C" int asr_gather_sumsq_f32(int njobs, const float* const* src, const int64_t* dst_offset, const int64_t* count,
                                    float* flat, float* sumsq, asr_stream_t stream) {
  if (njobs <= 0 || !src || !dst_offset || !count || !flat) return ASR_E_ARG;
  for (int j0 = 0; j0 < njobs; j0 += GATHER_JOBS) {
    GatherJobs t;
    t.n = njobs - j0 < GATHER_JOBS ? njobs - j0 : GATHER_JOBS;
    int64_t at = 0;
    for (int j = 0; j < GATHER_JOBS; ++j) {
      t.first[j] = (int)at;
      if (j < t.n) {
        if (!src[j0 + j] || count[j0 + j] <= 0 || dst_offset[j0 + j] < 0) return ASR_E_ARG;
        t.src[j] = src[j0 + j]; t.dst[j] = dst_offset[j0 + j]; t.count[j] = count[j0 + j];
        at += (count[j0 + j] + GATHER_CHUNK - 1) / GATHER_CHUNK;
        if (at > 0x7fffffff) return ASR_E_SHAPE;
      } else {
        t.src[j] = nullptr; t.dst[j] = 0; t.count[j] = 0;
      }
    }
    t.first[GATHER_JOBS] = (int)at;
    hipLaunchKernelGGL(gather_sumsq_kernel, dim3((unsigned)at), dim3(256), 0, (hipStream_t)stream, t, flat, sumsq);
  }
  ASR_CHECK_LAUNCH();
  return 0;
}
