// feedback.hip — the free-running decoder's output -> next-input coupling (model.py:329-351), one kernel per step each
// way.  Forward: logit_s = W_out [z_s, c_s] + b, prediction = argmax, and the embedding that feeds step s+1: the teacher
// token's row, the predicted token's row, or the "smooth" embedding softmax(scaling * logit_s) @ E (model.py:341).
// Backward (smooth mode): gradient of that embedding back into logit_s and on into [z_s, c_s].  These replace, per
// decoder step, a softmax + three tile GEMMs on 32 x 34 operands + half a dozen elementwise launches (~190 us of
// launches) by two launches of a few microseconds.  One workgroup per utterance; V <= 128.
#include "common.h"

namespace {

constexpr int FB_VMAX = 128;

struct FeedbackFwdArgs {
  int B, V, E, DO;
  const float* x;          // [B, ldx]: [z_s, c_s] at columns 0..DO-1
  int64_t ldx;
  const float* w_out;      // [V, DO]
  const float* b_out;      // [V]
  const float* emb;        // [V, E]
  float* logits;           // [B, V]
  int64_t* pred;           // [B]
  int mode;                // 0 predicted token, 1 smooth, 2 teacher token, 3 no next input (last step)
  float scaling;
  const int64_t* tok;      // mode 2: token of utterance b at tok[b * tok_stride]
  int64_t tok_stride;
  int64_t* fed;            // [B]: token whose embedding feeds the next step (-1: smooth)
  float* probs;            // mode 1: [B, V] softmax(scaling * logits), kept for the backward
  float* xe;               // next step's embedding slot [B, ldx] (already offset to the column)
  float* xde;              // same slot of the dropped-out input, or null
  const float* mask;       // [B, ldm] dropout multipliers of the embedding columns, or null
  int64_t ldm;
};

__global__ __launch_bounds__(256) void feedback_fwd_kernel(FeedbackFwdArgs a) {
  __shared__ float lg[FB_VMAX];
  __shared__ float pr[FB_VMAX];
  __shared__ int best;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xr = a.x + (int64_t)b * a.ldx;
  // logits: one wave per output, lanes stride the D+O inputs (coalesced rows of W_out)
  for (int v = wave; v < a.V; v += 4) {
    const float* wr = a.w_out + (int64_t)v * a.DO;
    float s = 0.f;
    for (int j = lane; j < a.DO; j += 64) s += wr[j] * xr[j];
    s = wave_sum(s);
    if (lane == 0) {
      s += a.b_out ? a.b_out[v] : 0.f;
      lg[v] = s;
      a.logits[(int64_t)b * a.V + v] = s;
    }
  }
  __syncthreads();
  if (wave == 0) {
    // argmax (lowest index among equal maxima) and, for the smooth input, softmax(scaling * logit)
    float mv = -INFINITY;
    int mi = 0x7fffffff;
    for (int v = lane; v < a.V; v += 64)
      if (lg[v] > mv) { mv = lg[v]; mi = v; }
    const float mx = wave_max(mv);
    int cand = mv == mx ? mi : 0x7fffffff;
    for (int off = 32; off; off >>= 1) cand = min(cand, __shfl_xor(cand, off, 64));
    if (cand >= a.V) cand = 0;          // all-NaN row: keep the index in range
    if (lane == 0) { best = cand; a.pred[b] = cand; }
    if (a.mode == 1) {
      float sm = -INFINITY;
      for (int v = lane; v < a.V; v += 64) sm = fmaxf(sm, a.scaling * lg[v]);
      sm = wave_max(sm);
      float se = 0.f;
      for (int v = lane; v < a.V; v += 64) { const float e = expf(a.scaling * lg[v] - sm); pr[v] = e; se += e; }
      se = wave_sum(se);
      const float inv = 1.0f / se;
      for (int v = lane; v < a.V; v += 64) {
        const float p = pr[v] * inv;
        pr[v] = p;
        a.probs[(int64_t)b * a.V + v] = p;
      }
    }
  }
  __syncthreads();
  if (a.mode == 3) return;
  int64_t token = -1;
  if (a.mode == 0) token = best;
  if (a.mode == 2) {
    token = a.tok[(int64_t)b * a.tok_stride];
    token = token < 0 ? 0 : (token >= a.V ? a.V - 1 : token);
  }
  if (tid == 0) a.fed[b] = token;
  for (int e = tid; e < a.E; e += 256) {
    float v;
    if (a.mode == 1) {
      v = 0.f;
      for (int u = 0; u < a.V; ++u) v += pr[u] * a.emb[(int64_t)u * a.E + e];
    } else {
      v = a.emb[token * a.E + e];
    }
    a.xe[(int64_t)b * a.ldx + e] = v;
    if (a.xde) a.xde[(int64_t)b * a.ldx + e] = v * a.mask[(int64_t)b * a.ldm + e];
  }
}

struct FeedbackBwdArgs {
  int B, V, E, DO;
  const float* demb;       // [B, ldg]: gradient of the embedding input of step s (already offset to its columns)
  float* gtop;             // [B, ldg]: gradient of [z_{s-1}, c_{s-1}], accumulated
  int64_t ldg;
  const float* probs;      // [B, V] of step s-1
  const float* emb;        // [V, E]
  const float* w_out;      // [V, DO]
  float scaling;
  float* dlog;             // [B, V]: total gradient of logit_{s-1}, accumulated (feeds the deferred dW_out, db_out)
};

__global__ __launch_bounds__(256) void feedback_bwd_kernel(FeedbackBwdArgs a) {
  __shared__ float de[512];
  __shared__ float dl[FB_VMAX];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < a.E; e += 256) de[e] = a.demb[(int64_t)b * a.ldg + e];
  __syncthreads();
  // dp[v] = demb . E[v]; dl = scaling * p * (dp - sum_u p_u dp_u)
  for (int v = wave; v < a.V; v += 4) {
    const float* er = a.emb + (int64_t)v * a.E;
    float s = 0.f;
    for (int e = lane; e < a.E; e += 64) s += er[e] * de[e];
    s = wave_sum(s);
    if (lane == 0) dl[v] = s;
  }
  __syncthreads();
  if (wave == 0) {
    float s = 0.f;
    for (int v = lane; v < a.V; v += 64) s += a.probs[(int64_t)b * a.V + v] * dl[v];
    s = wave_sum(s);
    for (int v = lane; v < a.V; v += 64) {
      const float d = a.scaling * a.probs[(int64_t)b * a.V + v] * (dl[v] - s);
      dl[v] = d;
      a.dlog[(int64_t)b * a.V + v] += d;
    }
  }
  __syncthreads();
  for (int j = tid; j < a.DO; j += 256) {
    float s = 0.f;
    for (int v = 0; v < a.V; ++v) s += dl[v] * a.w_out[(int64_t)v * a.DO + j];
    a.gtop[(int64_t)b * a.ldg + j] += s;
  }
}

}  // namespace

extern "C" int asr_dec_feedback_fwd(int B, int V, int E, int DO, const float* x, int64_t ldx, const float* w_out,
                                    const float* b_out, const float* emb, float* logits, int64_t* pred, int mode,
                                    float scaling, const int64_t* tok, int64_t tok_stride, int64_t* fed, float* probs,
                                    float* x_emb_next, float* xd_emb_next, const float* mask, int64_t ldm,
                                    asr_stream_t stream) {
  if (B <= 0 || V <= 0 || V > FB_VMAX || E <= 0 || DO <= 0 || !x || !w_out || !emb || !logits || !pred) return ASR_E_ARG;
  if (mode < 0 || mode > 3) return ASR_E_ARG;
  if (mode != 3 && (!fed || !x_emb_next)) return ASR_E_ARG;
  if (mode == 1 && !probs) return ASR_E_ARG;
  if (mode == 2 && !tok) return ASR_E_ARG;
  if (xd_emb_next && !mask) return ASR_E_ARG;
  FeedbackFwdArgs a{B, V, E, DO, x, ldx, w_out, b_out, emb, logits, pred, mode, scaling, tok, tok_stride, fed, probs,
                    x_emb_next, xd_emb_next, mask, ldm};
  hipLaunchKernelGGL(feedback_fwd_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, a);
  ASR_CHECK_LAUNCH();
  return 0;
}

extern "C" int asr_dec_feedback_bwd(int B, int V, int E, int DO, const float* demb, float* gtop, int64_t ldg,
                                    const float* probs, const float* emb, const float* w_out, float scaling,
                                    float* dlog, asr_stream_t stream) {
  if (B <= 0 || V <= 0 || V > FB_VMAX || E <= 0 || E > 512 || DO <= 0 || !demb || !gtop || !probs || !emb || !w_out ||
      !dlog)
    return ASR_E_ARG;
  FeedbackBwdArgs a{B, V, E, DO, demb, gtop, ldg, probs, emb, w_out, scaling, dlog};
  hipLaunchKernelGGL(feedback_bwd_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, a);
  ASR_CHECK_LAUNCH();
  return 0;
}
