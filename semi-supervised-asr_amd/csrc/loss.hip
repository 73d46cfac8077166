// loss.hip — label log-probabilities of the decoder output with label smoothing, forward and backward.
// Replaces log_softmax -> gather -> (1-ls) lp + ls * sum_v labeldist_v logp_v (model.py:354-366) and its autograd
// graph (~35 small launches per step) by one kernel each way.  One wave per (step, utterance) row; V is small.
#include "common.h"

namespace {

// y[row] = (1-ls) * logp[row][idx[row]] + ls * sum_v dist[v] * logp[row][v]
__global__ __launch_bounds__(256) void label_logprob_fwd_kernel(int64_t rows, int V, const float* __restrict__ z,
                                                                int64_t ld, const int64_t* __restrict__ idx,
                                                                const float* __restrict__ dist, float ls,
                                                                float* __restrict__ y, float* __restrict__ total,
                                                                float total_scale, int64_t* __restrict__ amax) {
  // a wave walks rows blockIdx.x * 4 + wave, + 4 gridDim.x, ...; `total` (optional) += total_scale * the sum of all y: one
  // atomic per block; `amax` (optional) = the row's argmax (lowest index on ties)
  __shared__ float part[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const float* zr = z + row * ld;
    float mx = -INFINITY;
    int at = V;
    for (int v = lane; v < V; v += 64) {
      const float zv = zr[v];
      if (zv > mx || (zv == mx && v < at)) at = v;       // (first maximum of this lane's elements)
      mx = fmaxf(mx, zv);
    }
    const float lane_mx = mx;
    mx = wave_max(mx);
    if (amax) {                                            // lowest index among the lanes that hold the maximum
      int cand = lane_mx == mx ? at : V;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(cand, o);
        cand = other < cand ? other : cand;
      }
      if (lane == 0) amax[row] = cand < V ? cand : 0;
    }
    float se = 0.f, sd = 0.f, sdz = 0.f;
    for (int v = lane; v < V; v += 64) {
      const float zv = zr[v];
      se += expf(zv - mx);
      if (dist) { sd += dist[v]; sdz += dist[v] * zv; }
    }
    se = wave_sum(se);
    const float lse = mx + logf(se);
    if (dist) { sd = wave_sum(sd); sdz = wave_sum(sdz); }
    if (lane == 0) {
      const float lp = zr[idx[row]] - lse;
      const float out = dist ? (1.f - ls) * lp + ls * (sdz - sd * lse) : lp;
      y[row] = out;
      acc += out;
    }
  }
  if (total) {
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(total, total_scale * ((part[0] + part[1]) + (part[2] + part[3])));
  }
}

// dz[row][v] = g[row] * ((1-ls) (delta(v, idx) - p_v) + ls (dist_v - p_v sum(dist)))
__global__ __launch_bounds__(256) void label_logprob_bwd_kernel(int64_t rows, int V, const float* __restrict__ z,
                                                                int64_t ld, const int64_t* __restrict__ idx,
                                                                const float* __restrict__ dist, float ls,
                                                                const float* __restrict__ g, int64_t gstride,
                                                                float gscale, float* __restrict__ dz, int64_t lddz) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* zr = z + row * ld;
  float mx = -INFINITY;
  for (int v = lane; v < V; v += 64) mx = fmaxf(mx, zr[v]);
  mx = wave_max(mx);
  float se = 0.f, sd = 0.f;
  for (int v = lane; v < V; v += 64) {
    se += expf(zr[v] - mx);
    if (dist) sd += dist[v];
  }
  se = wave_sum(se);
  if (dist) sd = wave_sum(sd);
  const float inv = 1.0f / se, gr = gscale * g[row * gstride];
  const int64_t ix = idx[row];
  const float a = dist ? (1.f - ls) : 1.f, b = dist ? ls : 0.f;
  for (int v = lane; v < V; v += 64) {
    const float p = expf(zr[v] - mx) * inv;
    float d = a * ((v == ix ? 1.f : 0.f) - p);
    if (dist) d += b * (dist[v] - p * sd);
    dz[row * lddz + v] = gr * d;
  }
}

}  // namespace

extern "C" int asr_label_logprob_fwd(int64_t rows, int V, const float* logits, int64_t ld, const int64_t* index,
                                     const float* labeldist, float ls_weight, float* out, float* total,
                                     float total_scale, int64_t* argmax, asr_stream_t stream) {
  if (rows <= 0 || V <= 0 || !logits || !index || !out) return ASR_E_ARG;
  const int64_t blocks = (rows + 3) / 4;
  hipLaunchKernelGGL(label_logprob_fwd_kernel, dim3((unsigned)(total && blocks > 512 ? 512 : blocks)), dim3(256), 0,
                     (hipStream_t)stream, rows, V, logits, ld, index, labeldist, ls_weight, out, total, total_scale, argmax);
  ASR_CHECK_LAUNCH();
  return 0;
}

extern "C" int asr_label_logprob_bwd(int64_t rows, int V, const float* logits, int64_t ld, const int64_t* index,
                                     const float* labeldist, float ls_weight, const float* grad_out,
                                     int64_t grad_stride, float grad_scale, float* dlogits, int64_t lddz,
                                     asr_stream_t stream) {
  if (rows <= 0 || V <= 0 || !logits || !index || !grad_out || !dlogits) return ASR_E_ARG;
  hipLaunchKernelGGL(label_logprob_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rows, V,
                     logits, ld, index, labeldist, ls_weight, grad_out, grad_stride, grad_scale, dlogits, lddz);
  ASR_CHECK_LAUNCH();
  return 0;
}
