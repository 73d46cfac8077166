// decoder.hip — decoder time step: LSTMCell + location-aware attention, forward and backward.
// Replaces Decoder.forward_step (model.py:283-294) and AttLoc.forward (model.py:139-173):
//   e[b,t] = gvec . tanh(P[b,t,:] + W_dec z_b + W_att (F * w_prev_b)[t,:])
//   w      = softmax(scaling * e) over ALL Tp frames (unmasked; scaling 2.0, F4)
//   ctx    = sum_t w[b,t] Q[b,t,:] + b_o,   Q = enc_h W_o^T hoisted out of the loop
// Kernels (forward):  cell (lstm.hip) -> skinny GEMM W_dec z -> att_score_fwd -> att_softmax_ctx_fwd
// Kernels (backward): att_dw -> att_score_bwd -> att_conv_bwd -> skinny GEMM (dz += dD W_dec)
//                     -> cell_bwd pointwise -> skinny GEMM (G[s] += dgates Wcat)
// All reductions inside a wave use 64-lane shuffles; cross-wave sums go through LDS; cross-workgroup
// sums are written as partial slabs and added by the consumer kernel of the next launch (no atomics,
// results are bitwise reproducible run to run).
#include "common.h"
#include "graphs.h"

int asr_skinny_launch(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* Bt, int64_t ldb,
                      float* C, int64_t ldc, const float* bias, int accumulate, const float* mask, int64_t ldmask,
                      int64_t mask_from, hipStream_t stream);
int asr_cell_fwd_launch(int B, int D, int KX, const float* Xs, const float* wcat, const float* bcat, float* gates,
                        const float* cprev, float* cout, float* zout, float* zout2, hipStream_t stream);
int asr_cell_bwd_launch(int B, int D, int KX, const float* Gnext, const float* gates, const float* cst,
                        const float* cprev, float* dcell, float* dgates, hipStream_t stream);

namespace {

constexpr int FRAMES_PER_WG = 16;  // 4 waves x 4 frames
constexpr int CMAX = 16;           // max conv channels held in registers by the backward kernel
constexpr int ATILE = 64;          // attention-dim columns per workgroup in the backward score kernel

__device__ __forceinline__ float fast_tanh(float x) {
  // 1 - 2/(1+e^{2x}); saturates correctly at +-inf; abs error ~1e-7
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
}

// NOTE on load scheduling (applies to every kernel below): global loads are issued in batches, are never
// guarded by a branch (indices are clamped instead, invalid lanes get weight 0 / a predicated store) and are
// hoisted above the LDS staging so that each kernel pays ~one memory round trip.  A guarded load consumed
// right away costs a full HBM/Infinity-Cache latency per loop iteration (35 us kernels in the first version).

// Copy n floats global -> LDS with all of a thread's loads of a batch issued before any store (one memory round
// trip per PER*256 elements instead of one per element).
template <int PER, int NT = 256>
__device__ __forceinline__ void stage_copy(float* __restrict__ lds, const float* __restrict__ g, int n) {
  for (int base = 0; base < n; base += NT * PER) {
    float v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = base + k * NT + (int)threadIdx.x;
      v[k] = g[i < n ? i : n - 1];
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = base + k * NT + (int)threadIdx.x;
      if (i < n) lds[i] = v[k];
    }
  }
}

// previous attention weights of utterance b with a zero halo of K frames on both sides
__device__ __forceinline__ void stage_wprev(float* __restrict__ wp, const float* __restrict__ wrow, int Tp, int K) {
  const int n = Tp + 2 * K;
  for (int base = 0; base < n; base += 512) {
    float v[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int fr = base + k * 256 + (int)threadIdx.x - K;
      v[k] = wrow[fr < 0 ? 0 : (fr >= Tp ? Tp - 1 : fr)];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = base + k * 256 + (int)threadIdx.x;
      if (i < n) wp[i] = (i - K >= 0 && i - K < Tp) ? v[k] : 0.f;
    }
  }
}

// ------------------------------------------------------------------ forward: energies
// grid (ceil(Tp/16), B), 256 threads.  dynamic LDS: wp[Tp+2K] | Fs[C][2K+1] | fs[16][C] | Ut[C][A]
// Each wave owns 4 frames; a lane owns 4 consecutive attention-dim columns of each 256-wide chunk.
constexpr int SC_WAVES = 8;                          // waves per workgroup of the score kernels
constexpr int SC_FPW = FRAMES_PER_WG / SC_WAVES;     // frames per wave (2)
constexpr int SC_NT = SC_WAVES * 64;
struct ScoreRegs {
  float4 p[SC_FPW];   // P[b, frame i, a..a+3]
  float4 d, g;        // Dproj[b, a..a+3], gvec[a..a+3]
};

__device__ __forceinline__ void score_load(ScoreRegs& r, const float* __restrict__ P, const float* __restrict__ Dp,
                                           const float* __restrict__ gvec, int b, int Tp, int A, int tw, int a) {
  const int ac = a < A ? a : 0;
#pragma unroll
  for (int i = 0; i < SC_FPW; ++i) {
    const int t = tw + i < Tp ? tw + i : Tp - 1;
    r.p[i] = *reinterpret_cast<const float4*>(P + ((int64_t)b * Tp + t) * A + ac);
  }
  r.d = *reinterpret_cast<const float4*>(Dp + (int64_t)b * A + ac);
  r.g = *reinterpret_cast<const float4*>(gvec + ac);
}

__device__ __forceinline__ void score_compute(const ScoreRegs& r, const float* __restrict__ Ut,
                                              const float* __restrict__ fs4 /* this wave's frames x C */,
                                              float* __restrict__ S, int b, int Tp, int A, int C, int tw, int a,
                                              float (&part)[SC_FPW]) {
  if (a >= A) return;    // wave-uniform except in the last chunk; no loads below depend on it
  float4 ucol[CMAX];
#pragma unroll
  for (int ch = 0; ch < CMAX; ++ch)
    ucol[ch] = ch < C ? *reinterpret_cast<const float4*>(Ut + ch * A + a) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < SC_FPW; ++i) {
    float4 u = r.p[i];
    u.x += r.d.x; u.y += r.d.y; u.z += r.d.z; u.w += r.d.w;
    const float* f = fs4 + i * C;
#pragma unroll
    for (int ch = 0; ch < CMAX; ++ch) {
      if (ch < C) {
        const float fv = f[ch];
        u.x += ucol[ch].x * fv; u.y += ucol[ch].y * fv; u.z += ucol[ch].z * fv; u.w += ucol[ch].w * fv;
      }
    }
    float4 sv;
    sv.x = fast_tanh(u.x); sv.y = fast_tanh(u.y); sv.z = fast_tanh(u.z); sv.w = fast_tanh(u.w);
#if !(defined(ASR_ABL) && (ASR_ABL & 8))
    if (tw + i < Tp) *reinterpret_cast<float4*>(S + ((int64_t)b * Tp + tw + i) * A + a) = sv;
#endif
    part[i] += r.g.x * sv.x + r.g.y * sv.y + r.g.z * sv.z + r.g.w * sv.w;
  }
}

__global__ __launch_bounds__(SC_NT) void att_score_fwd_kernel(int B, int Tp, int A, int C, int K,
                                                            const float* __restrict__ P,
                                                            const float* __restrict__ Dp,
                                                            const float* __restrict__ wprev,
                                                            const float* __restrict__ convw,
                                                            const float* __restrict__ wattT /* [C][A] */,
                                                            const float* __restrict__ gvec, float* __restrict__ S,
                                                            float* __restrict__ fconv, float* __restrict__ energy) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int taps = 2 * K + 1;
  float* Ut = sm;                                   // [C][A]  (16-byte aligned: first)
  float* wp = Ut + C * A;
  float* Fs = wp + (FRAMES_PER_WG * ((Tp + FRAMES_PER_WG - 1) / FRAMES_PER_WG) + 2 * K + 4);
  float* fs = Fs + 16 * ((taps + 3) & ~3);
  float* cred = fs + FRAMES_PER_WG * C;               // [SC_WAVES][16][17] conv partial tiles
  const int b = blockIdx.y, t0 = blockIdx.x * FRAMES_PER_WG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tw = t0 + wave * SC_FPW;
  // prefetch the first two 256-column chunks of this wave's frames (all of A when A <= 512)
  ScoreRegs r0, r1;
  score_load(r0, P, Dp, gvec, b, Tp, A, tw, lane * 4);
  score_load(r1, P, Dp, gvec, b, Tp, A, tw, 256 + lane * 4);
  // wp: previous weights with a zero halo, long enough for the last (partial) frame tile; Fs: [16][taps4] with
  // zero rows/taps beyond C / 2K+1 so the conv is a plain 16 x 16 x taps4 product
  const int taps4 = (taps + 3) & ~3;
  const int wlen = FRAMES_PER_WG * ((Tp + FRAMES_PER_WG - 1) / FRAMES_PER_WG) + 2 * K + 4;
  for (int base = 0; base < wlen; base += SC_NT) {
    const int i = base + tid;
    const int fr = i - K;
    const float v = wprev[(int64_t)b * Tp + (fr < 0 ? 0 : (fr >= Tp ? Tp - 1 : fr))];
    if (i < wlen) wp[i] = (fr >= 0 && fr < Tp) ? v : 0.f;
  }
  for (int base = 0; base < 16 * taps4; base += SC_NT * 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = base + k * SC_NT + tid;
      const int ch = i / taps4, j = i - ch * taps4;
      v[k] = convw[(ch < C && j < taps) ? ch * taps + j : 0];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = base + k * SC_NT + tid;
      const int ch = i / taps4, j = i - ch * taps4;
      if (i < 16 * taps4) Fs[i] = (ch < C && j < taps) ? v[k] : 0.f;
    }
  }
#if !(defined(ASR_ABL) && (ASR_ABL & 1))
  stage_copy<10, SC_NT>(Ut, wattT, A * C);
#endif
  __syncthreads();
  // location conv f[tl][ch] = sum_j F[ch][j] wp[t0+tl+j] as a Toeplitz product on the f32 MFMA, K (taps) split
  // over the waves: A[row=tl][k=j] = wp[t0+tl+j], B[k=j][col=ch] = F[ch][j]; D: lane holds rows 4*(lane>>4)+i of
  // column lane&15.  Partial tiles are summed through LDS (cred).
  {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15, q = lane >> 4;
    const float* ap = wp + t0 + r + q;
    const float* bp = Fs + r * taps4 + q;
    for (int j = 4 * wave; j < taps4; j += 4 * SC_WAVES) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[j], bp[j], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) cred[(wave * 16 + 4 * q + i) * 17 + r] = acc[i];
  }
  __syncthreads();
  if (tid < FRAMES_PER_WG * 16) {
    const int tl = tid >> 4, ch = tid & 15;
    float v = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < SC_WAVES; ++w2) v += cred[(w2 * 16 + tl) * 17 + ch];
    if (ch < C) {
      fs[tl * C + ch] = v;
      if (t0 + tl < Tp) fconv[((int64_t)b * C + ch) * Tp + t0 + tl] = v;
    }
  }
  __syncthreads();
  float part[SC_FPW];
#pragma unroll
  for (int i = 0; i < SC_FPW; ++i) part[i] = 0.f;
  const float* fs4 = fs + wave * SC_FPW * C;
#if defined(ASR_ABL) && (ASR_ABL & 4)
  part[0] = r0.p[0].x + r1.p[3].w + r0.d.x + r1.g.y;
#else
  score_compute(r0, Ut, fs4, S, b, Tp, A, C, tw, lane * 4, part);
  score_compute(r1, Ut, fs4, S, b, Tp, A, C, tw, 256 + lane * 4, part);
#endif
  for (int a0 = 512; a0 < A; a0 += 512) {           // attention dims beyond 512: same code, not prefetched
    score_load(r0, P, Dp, gvec, b, Tp, A, tw, a0 + lane * 4);
    score_load(r1, P, Dp, gvec, b, Tp, A, tw, a0 + 256 + lane * 4);
    score_compute(r0, Ut, fs4, S, b, Tp, A, C, tw, a0 + lane * 4, part);
    score_compute(r1, Ut, fs4, S, b, Tp, A, C, tw, a0 + 256 + lane * 4, part);
  }
#pragma unroll
  for (int i = 0; i < SC_FPW; ++i) {
    const float e = wave_sum(part[i]);
    if (lane == 0 && tw + i < Tp) energy[(int64_t)b * Tp + tw + i] = e;
  }
}

// ------------------------------------------------------------------ forward: softmax + context
// grid (ceil(O/256), B), 256 threads; dynamic LDS: ws[Tp] | red[4][256]
constexpr int CTX_BATCH = 8;
__device__ __forceinline__ void ctx_load(float4 (&q)[CTX_BATCH], const float* __restrict__ qb, int O, int Tp, int wave,
                                         int i0) {
#pragma unroll
  for (int u = 0; u < CTX_BATCH; ++u) {
    const int t = wave + 4 * (i0 + u);
    q[u] = *reinterpret_cast<const float4*>(qb + (int64_t)(t < Tp ? t : Tp - 1) * O);
  }
}
__device__ __forceinline__ void ctx_fma(const float4 (&q)[CTX_BATCH], const float* __restrict__ wsm, int Tp, int wave,
                                        int i0, float4& acc) {
#pragma unroll
  for (int u = 0; u < CTX_BATCH; ++u) {
    const int t = wave + 4 * (i0 + u);
    const float w = t < Tp ? wsm[t] : 0.f;
    acc.x += w * q[u].x; acc.y += w * q[u].y; acc.z += w * q[u].z; acc.w += w * q[u].w;
  }
}

__global__ __launch_bounds__(256) void att_softmax_ctx_fwd_kernel(int B, int Tp, int O, float scaling,
                                                                  const float* __restrict__ energy,
                                                                  const float* __restrict__ Q,
                                                                  const float* __restrict__ bo,
                                                                  float* __restrict__ wout, float* __restrict__ ctx,
                                                                  int64_t ldctx, float* __restrict__ ctxd,
                                                                  const float* __restrict__ dmask, int64_t ldmask) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wsm = sm;
  float* red = sm + ((Tp + 3) & ~3);
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int o = blockIdx.x * 256 + lane * 4;
  const int oc = o < O ? o : 0;
  const float* qb = Q + (int64_t)b * Tp * O + oc;
  const int nmine = (Tp - wave + 3) >> 2;                  // frames wave, wave+4, ...
  float4 qa[CTX_BATCH], qc[CTX_BATCH];
  ctx_load(qa, qb, O, Tp, wave, 0);                        // in flight during the softmax
  // every wave computes the softmax statistics redundantly (Tp is ~100): no block-level reduction
  float ev[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int t = lane + 64 * k;
    ev[k] = scaling * energy[(int64_t)b * Tp + (t < Tp ? t : Tp - 1)];
  }
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (lane + 64 * k < Tp) mx = fmaxf(mx, ev[k]);
  for (int t = lane + 256; t < Tp; t += 64) mx = fmaxf(mx, scaling * energy[(int64_t)b * Tp + t]);
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (lane + 64 * k < Tp) sum += expf(ev[k] - mx);
  for (int t = lane + 256; t < Tp; t += 64) sum += expf(scaling * energy[(int64_t)b * Tp + t] - mx);
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  if (wave == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int t = lane + 64 * k;
      if (t < Tp) {
        const float w = expf(ev[k] - mx) * inv;
        wsm[t] = w;
        if (blockIdx.x == 0) wout[(int64_t)b * Tp + t] = w;
      }
    }
    for (int t = lane + 256; t < Tp; t += 64) {
      const float w = expf(scaling * energy[(int64_t)b * Tp + t] - mx) * inv;
      wsm[t] = w;
      if (blockIdx.x == 0) wout[(int64_t)b * Tp + t] = w;
    }
  }
  __syncthreads();
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i0 = 0; i0 < nmine; i0 += 2 * CTX_BATCH) {
    const bool second = i0 + CTX_BATCH < nmine;
    if (second) ctx_load(qc, qb, O, Tp, wave, i0 + CTX_BATCH);
    ctx_fma(qa, wsm, Tp, wave, i0, acc);
    if (i0 + 2 * CTX_BATCH < nmine) ctx_load(qa, qb, O, Tp, wave, i0 + 2 * CTX_BATCH);
    if (second) ctx_fma(qc, wsm, Tp, wave, i0 + CTX_BATCH, acc);
  }
  *reinterpret_cast<float4*>(red + wave * 256 + lane * 4) = acc;
  __syncthreads();
  if (wave == 0 && o < O) {
    float4 r = *reinterpret_cast<const float4*>(red + lane * 4);
#pragma unroll
    for (int w2 = 1; w2 < 4; ++w2) {
      const float4 x = *reinterpret_cast<const float4*>(red + w2 * 256 + lane * 4);
      r.x += x.x; r.y += x.y; r.z += x.z; r.w += x.w;
    }
    const float4 bb = *reinterpret_cast<const float4*>(bo + o);
    r.x += bb.x; r.y += bb.y; r.z += bb.z; r.w += bb.w;
    *reinterpret_cast<float4*>(ctx + (int64_t)b * ldctx + o) = r;
    if (ctxd) {   // dropout-masked copy feeding the next step's cell (model.py:284-285)
      const float4 mk = *reinterpret_cast<const float4*>(dmask + (int64_t)b * ldmask + o);
      r.x *= mk.x; r.y *= mk.y; r.z *= mk.z; r.w *= mk.w;
      *reinterpret_cast<float4*>(ctxd + (int64_t)b * ldctx + o) = r;
    }
  }
}

// ------------------------------------------------------------------ backward: d(w) before the softmax
// dwraw[b,t] = Q[b,t,:] . dctx[b,:] + sum_c dwext[c][b][t] (+ dws[b,t]);  grid (ceil(Tp/16), B)
__global__ __launch_bounds__(256) void att_dw_kernel(int B, int Tp, int O, int C, const float* __restrict__ Q,
                                                     const float* __restrict__ dctx, int64_t lddctx,
                                                     const float* __restrict__ dwext,
                                                     const float* __restrict__ dws, float* __restrict__ dwraw) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tw = blockIdx.x * FRAMES_PER_WG + wave * 4;
  float part[4] = {0.f, 0.f, 0.f, 0.f};
  // side terms: lane ch < C fetches channel ch's conv-path partial for each of the 4 frames
  float ext[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = tw + i < Tp ? tw + i : Tp - 1;
    float v = 0.f;
    if (dwext) {
      const float x = dwext[((int64_t)(lane < C ? lane : 0) * B + b) * Tp + t];
      v = lane < C ? x : 0.f;
    }
    if (dws) {
      const float x = dws[(int64_t)b * Tp + t];
      v += lane == 0 ? x : 0.f;
    }
    ext[i] = v;
  }
  for (int o0 = 0; o0 < O; o0 += 512) {
    float4 g[2], q[2][4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int o = o0 + 256 * k + lane * 4;
      const int oc = o < O ? o : 0;
      g[k] = *reinterpret_cast<const float4*>(dctx + (int64_t)b * lddctx + oc);
      if (o >= O) g[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int t = tw + i < Tp ? tw + i : Tp - 1;
        q[k][i] = *reinterpret_cast<const float4*>(Q + ((int64_t)b * Tp + t) * O + oc);
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        part[i] += q[k][i].x * g[k].x + q[k][i].y * g[k].y + q[k][i].z * g[k].z + q[k][i].w * g[k].w;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float v = wave_sum(part[i] + ext[i]);
    if (lane == 0 && tw + i < Tp) dwraw[(int64_t)b * Tp + tw + i] = v;
  }
}

// ------------------------------------------------------------------ backward: scores
// grid (ceil(A/64), B), 256 threads.  Each workgroup owns 64 attention-dim columns of one utterance for
// ALL frames, so dD, dgvec, dW_att partials are local; d(conv output) partials go to a slab per tile.
// dynamic LDS: de[Tp] | fsm[C][Tp] | Us[64][C] | du[Tp][65] | red[4][64][2+CMAX]
constexpr int SB_BATCH = 16;   // frames per wave fetched in one batch (Tp <= 128 -> a single memory round trip)
struct ScoreBwdRegs {
  float s[SB_BATCH], dp[SB_BATCH];
};
__device__ __forceinline__ void sbwd_load(ScoreBwdRegs& r, const float* __restrict__ S, const float* __restrict__ dP,
                                          int64_t base /* (b*Tp)*A + a */, int A, int Tp, int wave, int i0) {
#pragma unroll
  for (int u = 0; u < SB_BATCH; ++u) {
    const int t = wave + SC_WAVES * (i0 + u);
    const int64_t off = base + (int64_t)(t < Tp ? t : Tp - 1) * A;
    r.s[u] = S[off];
    r.dp[u] = dP[off];
  }
}

// grid (ceil(A/64), B), 512 threads (8 waves; frames dealt round-robin to the waves).  Each workgroup owns 64
// attention-dim columns of one utterance for ALL frames, so dD, dgvec, dW_att are local; d(conv output) partials
// go to a slab per tile.  The two contractions over the du tile run on the f32 MFMA:
//   dW_att[a][ch] = sum_t du[t][a] f[ch][t]          (M = 64 columns, N = 16 channels, K = Tp)
//   df[ch][t]     = sum_a U[a][ch]  du[t][a]          (M = Tp frames,  N = 16 channels, K = 64)
// dynamic LDS: de[TpP] | fsm[16][TpK] | Us[64][16] | du[TpM][65] | red[8][64][2] | ured[2][64][17]
__global__ __launch_bounds__(SC_NT) void att_score_bwd_kernel(int B, int Tp, int A, int C, float scaling,
                                                            const float* __restrict__ wcur,
                                                            const float* __restrict__ dwraw,
                                                            const float* __restrict__ S,
                                                            const float* __restrict__ fconv,
                                                            const float* __restrict__ watt,
                                                            const float* __restrict__ gvec, float* __restrict__ dP,
                                                            float* __restrict__ dD, float* __restrict__ dgvec_part,
                                                            float* __restrict__ dwatt_part,
                                                            float* __restrict__ dfpart) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int TpP = (Tp + 3) & ~3;          // padded lengths: K of the dW_att product (multiple of 4)
  const int TpM = (Tp + 15) & ~15;        // M of the df product (multiple of 16)
  float* de = sm;
  float* fsm = de + TpP;                  // [16][TpP], zero beyond C / Tp
  float* Us = fsm + 16 * TpP;             // [64][16], zero beyond A / C
  float* du = Us + ATILE * 16;            // [TpM][65], zero rows beyond Tp
  float* red = du + TpM * 65;             // [8][64][2]
  float* ured = red + SC_WAVES * 64 * 2;  // [2][64][17]
  const int b = blockIdx.y, tile = blockIdx.x, a0 = tile * ATILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int a = a0 + lane;
  const bool live = a < A;
  const int64_t base = (int64_t)b * Tp * A + (live ? a : 0);
  const int nmine = (Tp - wave + SC_WAVES - 1) / SC_WAVES;
  ScoreBwdRegs ra;
  sbwd_load(ra, S, dP, base, A, Tp, wave, 0);            // in flight during the staging below
  const float gv = gvec[live ? a : 0];
  // softmax backward (each wave redundantly reduces the dot product)
  float wv[4], dv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int t = lane + 64 * k;
    const int tc = t < Tp ? t : Tp - 1;
    wv[k] = wcur[(int64_t)b * Tp + tc];
    dv[k] = dwraw[(int64_t)b * Tp + tc];
  }
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (lane + 64 * k < Tp) dot += wv[k] * dv[k];
  for (int t = lane + 256; t < Tp; t += 64) dot += wcur[(int64_t)b * Tp + t] * dwraw[(int64_t)b * Tp + t];
  dot = wave_sum(dot);
  if (wave == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (lane + 64 * k < Tp) de[lane + 64 * k] = scaling * wv[k] * (dv[k] - dot);
    for (int t = lane + 256; t < Tp; t += 64)
      de[t] = scaling * wcur[(int64_t)b * Tp + t] * (dwraw[(int64_t)b * Tp + t] - dot);
  }
  // fsm[ch][t] (zero padded), Us[al][ch] (zero padded), zero rows of du beyond Tp
  for (int base2 = 0; base2 < 16 * TpP; base2 += SC_NT * 4) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = base2 + k * SC_NT + tid;
      const int ch = i / TpP, t = i - ch * TpP;
      v[k] = fconv[((int64_t)b * C + (ch < C ? ch : 0)) * Tp + (t < Tp ? t : 0)];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = base2 + k * SC_NT + tid;
      const int ch = i / TpP, t = i - ch * TpP;
      if (i < 16 * TpP) fsm[i] = (ch < C && t < Tp) ? v[k] : 0.f;
    }
  }
  {
    float v[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = k * SC_NT + tid;               // ATILE*16 = 1024 = 2 * SC_NT
      const int al = i >> 4, ch = i & 15;
      v[k] = watt[(int64_t)(a0 + al < A ? a0 + al : 0) * C + (ch < C ? ch : 0)];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = k * SC_NT + tid;
      const int al = i >> 4, ch = i & 15;
      Us[i] = (a0 + al < A && ch < C) ? v[k] : 0.f;
    }
  }
  for (int i = Tp * 65 + tid; i < TpM * 65; i += SC_NT) du[i] = 0.f;
  __syncthreads();
  float dD_acc = 0.f, dg_acc = 0.f;
  auto consume = [&](const ScoreBwdRegs& r, int i0) {
#pragma unroll
    for (int u = 0; u < SB_BATCH; ++u) {
      const int t = wave + SC_WAVES * (i0 + u);
      if (t < Tp) {                                        // wave-uniform
        const float sv = r.s[u];
        const float det = de[t];
        const float duv = live ? det * gv * (1.f - sv * sv) : 0.f;
        if (live) dP[base + (int64_t)t * A] = r.dp[u] + duv;
        dD_acc += duv;
        dg_acc += live ? det * sv : 0.f;
        du[t * 65 + lane] = duv;
      }
    }
  };
  consume(ra, 0);
  for (int i0 = SB_BATCH; i0 < nmine; i0 += SB_BATCH) {   // only for Tp > 128
    sbwd_load(ra, S, dP, base, A, Tp, wave, i0);
    consume(ra, i0);
  }
  red[(wave * 64 + lane) * 2] = dD_acc;
  red[(wave * 64 + lane) * 2 + 1] = dg_acc;
  __syncthreads();
  if (wave == 0 && live) {
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < SC_WAVES; ++w2) {
      t0 += red[(w2 * 64 + lane) * 2];
      t1 += red[(w2 * 64 + lane) * 2 + 1];
    }
    dD[(int64_t)b * A + a] = t0;
    dgvec_part[(int64_t)b * A + a] += t1;
  }
  const int r = lane & 15, q = lane >> 4;
  // dW_att tile: wave -> (M tile m = wave&3, K half kh = wave>>2); A[row=a][k=t] = du[t][a], B[k=t][col=ch] = fsm[ch][t]
  {
    const int m = wave & 3, kh = wave >> 2;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int khalf = ((TpP >> 2) + 1) >> 1;                 // k-steps per half
    const int s0 = kh * khalf, s1 = (s0 + khalf) < (TpP >> 2) ? (s0 + khalf) : (TpP >> 2);
    for (int st = s0; st < s1; ++st) {
      const int t = 4 * st + q;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(du[t * 65 + 16 * m + r], fsm[r * TpP + t], acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) ured[(kh * 64 + 16 * m + 4 * q + i) * 17 + r] = acc[i];
  }
  // df tiles: M tile = 16 frames, K = 64 columns; A[row=t][k=al] = du[t][al], B[k=al][col=ch] = Us[al][ch]
  for (int mt = wave; mt < (TpM >> 4); mt += SC_WAVES) {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int st = 0; st < ATILE / 4; ++st) {
      const int al = 4 * st + q;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(du[(16 * mt + r) * 65 + al], Us[al * 16 + r], acc, 0, 0, 0);
    }
    if (r < C) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int t = 16 * mt + 4 * q + i;
        if (t < Tp) dfpart[(((int64_t)tile * B + b) * C + r) * Tp + t] = acc[i];
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < ATILE * C; i += SC_NT) {
    const int al = i / C, ch = i - al * C;
    if (a0 + al < A)
      dwatt_part[((int64_t)b * A + a0 + al) * C + ch] += ured[al * 17 + ch] + ured[(64 + al) * 17 + ch];
  }
}

// ------------------------------------------------------------------ backward: location conv
// grid (C, B), 256 threads.  dynamic LDS: df[Tp] | wp[Tp+2K] | Fs[2K+1]
__global__ __launch_bounds__(256) void att_conv_bwd_kernel(int B, int Tp, int C, int K, int ntile,
                                                           const float* __restrict__ dfpart,
                                                           const float* __restrict__ wprev,
                                                           const float* __restrict__ convw,
                                                           float* __restrict__ dwext,
                                                           float* __restrict__ dconv_part) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int taps = 2 * K + 1;
  float* df = sm;
  float* wp = df + ((Tp + 3) & ~3);
  float* Fs = wp + ((Tp + 2 * K + 3) & ~3);
  const int ch = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  for (int t = tid; t < Tp; t += 256) {
    float v = 0.f;
    for (int tl0 = 0; tl0 < ntile; tl0 += 8) {             // 8 tile partials per batch, unconditional loads
      float x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int tl = tl0 + k < ntile ? tl0 + k : ntile - 1;
        x[k] = dfpart[(((int64_t)tl * B + b) * C + ch) * Tp + t];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) v += tl0 + k < ntile ? x[k] : 0.f;
    }
    df[t] = v;
  }
  stage_wprev(wp, wprev + (int64_t)b * Tp, Tp, K);
  stage_copy<1>(Fs, convw + ch * taps, taps);
  // accumulate-in-place operand fetched early (one value per tap owned by this thread)
  float dcv = 0.f;
  const bool tapmine = tid < taps;
  if (tapmine) dcv = dconv_part[((int64_t)b * C + ch) * taps + tid];
  __syncthreads();
  // f[t] = sum_j F[j] wprev[t + j - K]  =>  dwprev[t'] = sum_t F[t' - t + K] df[t]
  for (int base = 0; base < Tp * 4; base += 256) {
    const int idx = base + tid;
    const int tq = idx >> 2, part = idx & 3;
    float v = 0.f;
    if (tq < Tp) {
      const int lo = tq - K > 0 ? tq - K : 0;            // t' - t + K <= 2K
      const int hi = tq + K < Tp - 1 ? tq + K : Tp - 1;  // t' - t + K >= 0
      float v1 = 0.f, v2 = 0.f, v3 = 0.f;
      int t = lo + part;
      for (; t + 12 <= hi; t += 16) {
        v += Fs[tq - t + K] * df[t];
        v1 += Fs[tq - t - 4 + K] * df[t + 4];
        v2 += Fs[tq - t - 8 + K] * df[t + 8];
        v3 += Fs[tq - t - 12 + K] * df[t + 12];
      }
      for (; t <= hi; t += 4) v += Fs[tq - t + K] * df[t];
      v += (v1 + v2) + v3;
    }
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    if (tq < Tp && part == 0) dwext[((int64_t)ch * B + b) * Tp + tq] = v;
  }
  // dF[j] += sum_t df[t] wprev[t + j - K]
  if (tapmine) {
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int t = 0;
    for (; t + 3 < Tp; t += 4) {
      v0 += df[t] * wp[t + tid];
      v1 += df[t + 1] * wp[t + 1 + tid];
      v2 += df[t + 2] * wp[t + 2 + tid];
      v3 += df[t + 3] * wp[t + 3 + tid];
    }
    for (; t < Tp; ++t) v0 += df[t] * wp[t + tid];
    dconv_part[((int64_t)b * C + ch) * taps + tid] = dcv + ((v0 + v1) + (v2 + v3));
  }
  for (int j = tid + 256; j < taps; j += 256) {            // taps beyond 256 (K > 127)
    float v = 0.f;
    for (int t = 0; t < Tp; ++t) v += df[t] * wp[t + j];
    dconv_part[((int64_t)b * C + ch) * taps + j] += v;
  }
}

int check_fwd(const asr_dec_fwd_t* p) {
  if (!p || !p->P || !p->Q || !p->bo || !p->wcat || !p->bcat || !p->wdec || !p->convw || !p->watt || !p->wattT || !p->gvec ||
      !p->w0 || !p->X || !p->gates || !p->cstate || !p->Dproj || !p->fconv || !p->S || !p->energy || !p->ws)
    return ASR_E_ARG;
  if (p->B <= 0 || p->nb <= 0 || p->nb > p->B || p->Tp <= 0 || p->L <= 0) return ASR_E_ARG;
  if (p->D % 16 || p->A % 16 || p->O % 4 || (p->D + p->O + p->E) % 16 || p->C > CMAX || p->C <= 0) return ASR_E_SHAPE;
  return 0;
}

}  // namespace

static int dec_step_fwd_impl(const asr_dec_fwd_t* p, int s, hipStream_t stream) {
  const int B = p->B, nb = p->nb, Tp = p->Tp, A = p->A, D = p->D, O = p->O, E = p->E, C = p->C, K = p->K;
  const int KX = D + O + E;
  const float* Xs = p->X + (int64_t)s * B * KX;
  float* Xn = p->X + (int64_t)(s + 1) * B * KX;
  const bool drop = p->xmask != nullptr;
  if (drop && !p->Xd) return ASR_E_ARG;
  float* Xdn = drop ? p->Xd + (int64_t)(s + 1) * B * KX : nullptr;
  int rc = 0;
#ifdef ASR_ONLY
  const int only = ASR_ONLY;
#else
  const int only = 0;
#endif
  if (only == 0 || only == 1)
  rc = asr_cell_fwd_launch(nb, D, KX, drop ? p->Xd + (int64_t)s * B * KX : Xs, p->wcat, p->bcat,
                               p->gates + (int64_t)s * B * 4 * D, s > 0 ? p->cstate + (int64_t)(s - 1) * B * D : nullptr,
                               p->cstate + (int64_t)s * B * D, Xn, Xdn, stream);
  if (rc) return rc;
  float* Dp = p->Dproj + (int64_t)s * B * A;
  if (only == 0 || only == 2)
  rc = asr_skinny_launch(nb, A, D, Xn, KX, p->wdec, D, Dp, A, nullptr, 0, nullptr, 0, 0, stream);
  if (rc) return rc;
  const float* wprev = s > 0 ? p->ws + (int64_t)(s - 1) * B * Tp : p->w0;
  const int taps = 2 * K + 1;
  const size_t lds1 = sizeof(float) * ((size_t)(FRAMES_PER_WG * ((Tp + FRAMES_PER_WG - 1) / FRAMES_PER_WG) + 2 * K + 4) +
                                       (size_t)16 * ((taps + 3) & ~3) + FRAMES_PER_WG * C + (size_t)C * A +
                                       (size_t)SC_WAVES * 16 * 17);
  if (only == 0 || only == 3)
  hipLaunchKernelGGL(att_score_fwd_kernel, dim3((Tp + FRAMES_PER_WG - 1) / FRAMES_PER_WG, nb), dim3(SC_NT), lds1, stream,
                     B, Tp, A, C, K, p->P, Dp, wprev, p->convw, p->wattT, p->gvec, p->S + (int64_t)s * B * Tp * A,
                     p->fconv + (int64_t)s * B * C * Tp, p->energy + (int64_t)s * B * Tp);
  const size_t lds2 = sizeof(float) * ((size_t)((Tp + 3) & ~3) + 4 * 256);
  if (only == 0 || only == 4)
  hipLaunchKernelGGL(att_softmax_ctx_fwd_kernel, dim3((O + 255) / 256, nb), dim3(256), lds2, stream, B, Tp, O,
                     p->scaling, p->energy + (int64_t)s * B * Tp, p->Q, p->bo, p->ws + (int64_t)s * B * Tp, Xn + D,
                     (int64_t)KX, (drop && s + 1 < p->L) ? Xdn + D : nullptr,
                     (drop && s + 1 < p->L) ? p->xmask + (int64_t)(s + 1) * B * (O + E) : nullptr, (int64_t)(O + E));
  ASR_CHECK_LAUNCH();
  return 0;
}

extern "C" int asr_dec_step_fwd(const asr_dec_fwd_t* p, int s, asr_stream_t stream_) {
  int rc = check_fwd(p);
  if (rc) return rc;
  if (s < 0 || s >= p->L) return ASR_E_ARG;
  return dec_step_fwd_impl(p, s, (hipStream_t)stream_);
}

// Attention part of step s only (AttLoc.forward, model.py:139-173): the caller has placed the decoder state z in
// X[s+1][:, 0:D]; runs mlp_dec -> energies -> softmax + context, leaving mlp_o(context) in X[s+1][:, D:D+O] and the
// weights in ws[s].
extern "C" int asr_att_step_fwd(const asr_dec_fwd_t* p, int s, asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_fwd(p);
  if (rc) return rc;
  if (s < 0 || s >= p->L) return ASR_E_ARG;
  const int B = p->B, nb = p->nb, Tp = p->Tp, A = p->A, D = p->D, O = p->O, E = p->E, C = p->C, K = p->K;
  const int KX = D + O + E;
  float* Xn = p->X + (int64_t)(s + 1) * B * KX;
  float* Dp = p->Dproj + (int64_t)s * B * A;
  rc = asr_skinny_launch(nb, A, D, Xn, KX, p->wdec, D, Dp, A, nullptr, 0, nullptr, 0, 0, stream);
  if (rc) return rc;
  const float* wprev = s > 0 ? p->ws + (int64_t)(s - 1) * B * Tp : p->w0;
  const int taps = 2 * K + 1;
  const size_t lds1 = sizeof(float) * ((size_t)(FRAMES_PER_WG * ((Tp + FRAMES_PER_WG - 1) / FRAMES_PER_WG) + 2 * K + 4) +
                                       (size_t)16 * ((taps + 3) & ~3) + FRAMES_PER_WG * C + (size_t)C * A +
                                       (size_t)SC_WAVES * 16 * 17);
  hipLaunchKernelGGL(att_score_fwd_kernel, dim3((Tp + FRAMES_PER_WG - 1) / FRAMES_PER_WG, nb), dim3(SC_NT), lds1, stream,
                     B, Tp, A, C, K, p->P, Dp, wprev, p->convw, p->wattT, p->gvec, p->S + (int64_t)s * B * Tp * A,
                     p->fconv + (int64_t)s * B * C * Tp, p->energy + (int64_t)s * B * Tp);
  const size_t lds2 = sizeof(float) * ((size_t)((Tp + 3) & ~3) + 4 * 256);
  hipLaunchKernelGGL(att_softmax_ctx_fwd_kernel, dim3((O + 255) / 256, nb), dim3(256), lds2, stream, B, Tp, O,
                     p->scaling, p->energy + (int64_t)s * B * Tp, p->Q, p->bo, p->ws + (int64_t)s * B * Tp, Xn + D,
                     (int64_t)KX, nullptr, nullptr, (int64_t)0);
  ASR_CHECK_LAUNCH();
  return 0;
}

extern "C" int asr_dec_seq_fwd(const asr_dec_fwd_t* p, int s_begin, int s_end, void* graphs, asr_stream_t stream_) {
  int rc = check_fwd(p);
  if (rc) return rc;
  if (s_begin < 0 || s_end > p->L || s_begin > s_end) return ASR_E_ARG;
  struct { int kind, s0, s1; asr_dec_fwd_t f; } key;
  memset(&key, 0, sizeof(key));
  key.kind = 3; key.s0 = s_begin; key.s1 = s_end; key.f = *p;
  return asr_graph_run((AsrGraphCache*)graphs, &key, sizeof(key), (hipStream_t)stream_, [&](hipStream_t stream) -> int {
    for (int s = s_begin; s < s_end; ++s) {
      int r = dec_step_fwd_impl(p, s, stream);
      if (r) return r;
    }
    return 0;
  });
}

static int check_bwd(const asr_dec_bwd_t* q) {
  if (!q) return ASR_E_ARG;
  int rc = check_fwd(&q->f);
  if (rc) return rc;
  if (!q->wcatT || !q->wdecT || !q->G || !q->dwext || !q->dwraw || !q->dfpart || !q->dP || !q->dgates || !q->dD ||
      !q->dcell || !q->dgvec_part || !q->dwatt_part || !q->dconv_part)
    return ASR_E_ARG;
  return 0;
}

static int dec_step_bwd_impl(const asr_dec_bwd_t* q, int s, hipStream_t stream) {
  const asr_dec_fwd_t* p = &q->f;
  const int B = p->B, nb = p->nb, Tp = p->Tp, A = p->A, D = p->D, O = p->O, E = p->E, C = p->C, K = p->K;
  const int KX = D + O + E, taps = 2 * K + 1;
  const int ntile = (A + ATILE - 1) / ATILE;
  float* Gn = q->G + (int64_t)(s + 1) * B * KX;
  float* Gs = q->G + (int64_t)s * B * KX;
  const int tgrid = (Tp + FRAMES_PER_WG - 1) / FRAMES_PER_WG;
#ifdef ASR_ONLYB
  const int onlyb = ASR_ONLYB;
#else
  const int onlyb = 0;
#endif
  if (onlyb == 0 || onlyb == 1)
  hipLaunchKernelGGL(att_dw_kernel, dim3(tgrid, nb), dim3(256), 0, stream, B, Tp, O, C, p->Q, Gn + D, (int64_t)KX,
                     s + 1 < p->L ? q->dwext : nullptr, q->dws ? q->dws + (int64_t)s * B * Tp : nullptr, q->dwraw);
  const int TpP = (Tp + 3) & ~3;
  const int TpM = (Tp + 15) & ~15;
  const size_t lds2 = sizeof(float) * ((size_t)TpP + (size_t)16 * TpP + ATILE * 16 + (size_t)TpM * 65 +
                                       SC_WAVES * 64 * 2 + 2 * 64 * 17);
  float* dDs = q->dD + (int64_t)s * B * A;
  if (onlyb == 0 || onlyb == 2)
  hipLaunchKernelGGL(att_score_bwd_kernel, dim3(ntile, nb), dim3(SC_NT), lds2, stream, B, Tp, A, C, p->scaling,
                     p->ws + (int64_t)s * B * Tp, q->dwraw, p->S + (int64_t)s * B * Tp * A,
                     p->fconv + (int64_t)s * B * C * Tp, p->watt, p->gvec, q->dP, dDs, q->dgvec_part, q->dwatt_part,
                     q->dfpart);
  const float* wprev = s > 0 ? p->ws + (int64_t)(s - 1) * B * Tp : p->w0;
  const size_t lds3 = sizeof(float) * ((size_t)TpP + (size_t)((Tp + 2 * K + 3) & ~3) + taps);
  if (onlyb == 0 || onlyb == 3)
  hipLaunchKernelGGL(att_conv_bwd_kernel, dim3(C, nb), dim3(256), lds3, stream, B, Tp, C, K, ntile, q->dfpart, wprev,
                     p->convw, q->dwext, q->dconv_part);
  ASR_CHECK_LAUNCH();
  // dz_s += dD W_dec
  int rc = 0;
  if (onlyb == 0 || onlyb == 4)
  rc = asr_skinny_launch(nb, D, A, dDs, A, q->wdecT, A, Gn, KX, nullptr, 1, nullptr, 0, 0, stream);
  if (rc) return rc;
  float* dg = q->dgates + (int64_t)s * B * 4 * D;
  if (onlyb == 0 || onlyb == 5)
  rc = asr_cell_bwd_launch(nb, D, KX, Gn, p->gates + (int64_t)s * B * 4 * D, p->cstate + (int64_t)s * B * D,
                           s > 0 ? p->cstate + (int64_t)(s - 1) * B * D : nullptr, q->dcell, dg, stream);
  if (rc) return rc;
  const float* xm = p->xmask ? p->xmask + (int64_t)s * B * (O + E) : nullptr;
  if (!(onlyb == 0 || onlyb == 6)) return 0;
  return asr_skinny_launch(nb, KX, 4 * D, dg, 4 * D, q->wcatT, 4 * D, Gs, KX, nullptr, 1, xm, O + E, D, stream);
}

extern "C" int asr_dec_step_bwd(const asr_dec_bwd_t* q, int s, asr_stream_t stream_) {
  int rc = check_bwd(q);
  if (rc) return rc;
  if (s < 0 || s >= q->f.L) return ASR_E_ARG;
  return dec_step_bwd_impl(q, s, (hipStream_t)stream_);
}

extern "C" int asr_dec_seq_bwd(const asr_dec_bwd_t* q, int s_begin, int s_end, void* graphs, asr_stream_t stream_) {
  int rc = check_bwd(q);
  if (rc) return rc;
  if (s_begin < 0 || s_end > q->f.L || s_begin > s_end) return ASR_E_ARG;
  struct { int kind, s0, s1; asr_dec_bwd_t b; } key;
  memset(&key, 0, sizeof(key));
  key.kind = 4; key.s0 = s_begin; key.s1 = s_end; key.b = *q;
  return asr_graph_run((AsrGraphCache*)graphs, &key, sizeof(key), (hipStream_t)stream_, [&](hipStream_t stream) -> int {
    for (int s = s_end - 1; s >= s_begin; --s) {
      int r = dec_step_bwd_impl(q, s, stream);
      if (r) return r;
    }
    return 0;
  });
}
