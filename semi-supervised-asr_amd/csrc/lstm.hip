// lstm.hip — fused LSTM time-step kernels for gfx950.
//   encoder / judge recurrence: asr_lstm_seq_fwd / asr_lstm_seq_bwd  (replaces torch.nn.LSTM on a
//     PackedSequence, model.py:79-81 and 466-467,515-519)
//   decoder cell: cell_fwd_launch / cell_bwd_launch used by decoder.hip (replaces torch.nn.LSTMCell,
//     model.py:262,286)
// One launch per time step; both directions of a layer share the launch (blockIdx.y).  A workgroup owns
// 4 hidden units x 4 gates (16 gate-interleaved rows of W) x <=32 batch rows in the forward, and 16
// hidden units in the backward; the recurrent product runs on v_mfma_f32_16x16x4_f32 with K split
// over the 4 waves and reduced through LDS, followed by the pointwise gate math in the same kernel.
#include "common.h"
#include "graphs.h"

namespace {

// ------------------------------------------------------------------ forward step (encoder/judge)
// Workgroup = 4 hidden units x 4 gates (16 gate-interleaved rows of W_hh) x MT*16 batch rows.
// The pointwise operands (x-projection, c_{t-1}, length) are fetched BEFORE the recurrent product so their
// HBM/Infinity-Cache miss latency hides under the weight stream + MFMA phase.
template <int MT>
__global__ __launch_bounds__(256) void enc_step_fwd_kernel(int T, int B, int nb, int H, int ndir,
                                                           float* __restrict__ gates,
                                                           const float* __restrict__ w_hh,
                                                           const int32_t* __restrict__ lens, float* __restrict__ y,
                                                           float* __restrict__ c, int s,
                                                           const int32_t* __restrict__ rowbase,
                                                           const int32_t* __restrict__ rowext) {
  // rowbase / rowext: NULL = time-major rows (t * B + b); else packed rows, batch row b at rowbase[b] + t for t < rowext[b]
  // (include/asr_hip.h): times >= rowext[b] do not exist - loads are clamped to the block's last row (dead values), nothing
  // is stored
  __shared__ float red[4 * MT * 16 * SK_LDS_STRIDE];
  const int j = blockIdx.x, d = blockIdx.y;
  const int64_t row0 = (int64_t)blockIdx.z * (MT * 16);
  const int t = d == 0 ? s : T - 1 - s;
  const int tp = d == 0 ? t - 1 : t + 1;
  const int64_t ldy = (int64_t)ndir * H;
  const int e = threadIdx.x;
  const int row = e >> 2, u = e & 3;
  const int64_t b = row0 + row;
  const bool mine = e < MT * 16 * 4 && b < nb;
  const int unit = 4 * j + u;
  float4 gx = make_float4(0.f, 0.f, 0.f, 0.f);
  float cp = 0.f;
  int len = 0;
  float4* gp = nullptr;
  int64_t so = 0;
  bool exists = false;
  auto row_of = [&](int64_t bb, int tt) -> int64_t {
    if (!rowbase) return (int64_t)tt * B + bb;
    const int ext = rowext[bb];
    return (int64_t)rowbase[bb] + (tt < ext ? tt : ext - 1);
  };
  if (mine) {
    exists = !rowbase || t < rowext[b];
    const int64_t rt = row_of(b, t);
    gp = reinterpret_cast<float4*>(gates + (rt * ndir + d) * 4 * H + unit * 4);
    so = rt * ldy + d * H + unit;
    gx = *gp;
    len = lens[b];
    if (s > 0) cp = c[row_of(b, tp) * ldy + d * H + unit];
  }
  if (s > 0) {
    int64_t arow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int64_t rr = row0 + m * 16 + (threadIdx.x & 15);
      arow[m] = row_of(rr < nb ? rr : nb - 1, tp);
    }
    skinny_partial_rows<MT>(y + d * H, ldy, arow, w_hh + (int64_t)d * 4 * H * H, H, (int64_t)16 * j, (int64_t)4 * H, H, red);
  }
  __syncthreads();
  if (mine) {
    float pre[4] = {0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) pre[g] = skinny_reduced<MT>(red, row, u * 4 + g);
    }
    const float gi = asr_sigmoid(pre[0] + gx.x), gf = asr_sigmoid(pre[1] + gx.y);
    const float gg = tanhf(pre[2] + gx.z), go = asr_sigmoid(pre[3] + gx.w);
    float cn = gf * cp + gi * gg;
    float hn = go * tanhf(cn);
    if (t >= len) { cn = 0.f; hn = 0.f; }
    if (exists) {
      *gp = make_float4(gi, gf, gg, go);
      c[so] = cn;
      y[so] = hn;
    }
    if (rowbase && s == 0) {          // packed rows: the block's padding rows behind the T steps of the sequence
      for (int tt = T; tt < rowext[b]; ++tt) {
        const int64_t o = ((int64_t)rowbase[b] + tt) * ldy + d * H + unit;
        c[o] = 0.f;
        y[o] = 0.f;
      }
    }
  }
}

// ------------------------------------------------------------------ backward step (encoder/judge)
// Workgroup = UNITS hidden units of one direction x MT*16 batch rows.  Phase 1: dh_rec = dG[t_next] W_hh
// (K = 4H) for its units; phase 2: pointwise LSTM backward at time t, dG[t] written in place.
// W_hh is re-streamed from Infinity Cache every launch (L2 does not survive the kernel boundary) at the
// per-CU fabric share, so UNITS is small enough to spread that stream over the whole chip.
template <int MT, int UNITS, int NW>
__global__ __launch_bounds__(NW * 64) void enc_step_bwd_kernel(int T, int B, int nb, int H, int ndir,
                                                           float* __restrict__ gates,
                                                           const float* __restrict__ w_hhT,
                                                           const int32_t* __restrict__ lens,
                                                           const float* __restrict__ dy, const float* __restrict__ c,
                                                           float* __restrict__ dcarry, int s,
                                                           const int32_t* __restrict__ rowbase,
                                                           const int32_t* __restrict__ rowext) {
  __shared__ float red[NW * MT * 16 * SK_LDS_STRIDE];
  auto row_of = [&](int64_t bb, int tt) -> int64_t {     // see enc_step_fwd_kernel
    if (!rowbase) return (int64_t)tt * B + bb;
    const int ext = rowext[bb];
    return (int64_t)rowbase[bb] + (tt < ext ? tt : ext - 1);
  };
  constexpr int NT = NW * 64;
  constexpr int NE = (MT * 16 * UNITS + NT - 1) / NT;   // pointwise elements per thread
  const int j = blockIdx.x, d = blockIdx.y;
  const int64_t row0 = (int64_t)blockIdx.z * (MT * 16);
  const int t = d == 0 ? T - 1 - s : s;
  const int tn = d == 0 ? t + 1 : t - 1;   // step handled by the previous launch
  const int tp = d == 0 ? t - 1 : t + 1;   // forward-time predecessor (owner of c_prev)
  const bool has_prev = d == 0 ? (t > 0) : (t < T - 1);
  const int64_t ldy = (int64_t)ndir * H, ldg = (int64_t)ndir * 4 * H;
  // prefetch the pointwise operands
  float dyv[NE], ctv[NE], cpv[NE], dcv[NE];
  float4 av[NE];
  int lenv[NE];
  bool live[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = threadIdx.x + NT * i;
    const int row = e / UNITS, u = e % UNITS;
    const int64_t b = row0 + row;
    live[i] = e < MT * 16 * UNITS && b < nb;
    dyv[i] = ctv[i] = cpv[i] = dcv[i] = 0.f;
    av[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    lenv[i] = 0;
    if (live[i]) {
      const int unit = UNITS * j + u;
      const int64_t rt = row_of(b, t);
      const int64_t so = rt * ldy + d * H + unit;
      dyv[i] = dy[so];
      av[i] = *reinterpret_cast<const float4*>(gates + rt * ldg + (int64_t)d * 4 * H + unit * 4);
      ctv[i] = c[so];
      if (has_prev) cpv[i] = c[row_of(b, tp) * ldy + d * H + unit];
      dcv[i] = dcarry[b * ldy + d * H + unit];
      lenv[i] = lens[b];
    }
  }
  if (s > 0) {
    int64_t arow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int64_t rr = row0 + m * 16 + (threadIdx.x & 15);
      arow[m] = row_of(rr < nb ? rr : nb - 1, tn);
    }
    skinny_partial_rows<MT, NW>(gates + (int64_t)d * 4 * H, ldg, arow, w_hhT + (int64_t)d * H * 4 * H, (int64_t)4 * H,
                                (int64_t)UNITS * j, (int64_t)UNITS * (j + 1), 4 * H, red);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    if (!live[i]) continue;
    const int e = threadIdx.x + NT * i;
    const int row = e / UNITS, u = e % UNITS;
    const int64_t b = row0 + row;
    const int unit = UNITS * j + u;
    float dh = dyv[i];
    if (s > 0) dh += skinny_reduced<MT, NW>(red, row, u);
    const float4 a = av[i];  // i f g o
    const float tc = tanhf(ctv[i]);
    const float dc = dcv[i] + dh * a.w * (1.f - tc * tc);
    float4 da;
    da.x = dc * a.z * a.x * (1.f - a.x);
    da.y = dc * cpv[i] * a.y * (1.f - a.y);
    da.z = dc * a.x * (1.f - a.z * a.z);
    da.w = dh * tc * a.w * (1.f - a.w);
    float dcn = dc * a.y;
    if (t >= lenv[i]) { da = make_float4(0.f, 0.f, 0.f, 0.f); dcn = 0.f; }
    if (!rowbase || t < rowext[b])
      *reinterpret_cast<float4*>(gates + row_of(b, t) * ldg + (int64_t)d * 4 * H + unit * 4) = da;
    if (rowbase && s == 0) {          // packed rows: dG of the block's padding rows behind the T steps of the sequence
      for (int tt = T; tt < rowext[b]; ++tt)
        *reinterpret_cast<float4*>(gates + ((int64_t)rowbase[b] + tt) * ldg + (int64_t)d * 4 * H + unit * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    dcarry[b * ldy + d * H + unit] = dcn;
  }
}

// ------------------------------------------------------------------ decoder cell forward
// gates = [z_prev | ctx_prev | emb] Wcat^T + bcat  (K = KX), then the same pointwise update.
template <int MT>
__global__ __launch_bounds__(256) void cell_fwd_kernel(int B, int D, int KX, const float* __restrict__ Xs,
                                                       const float* __restrict__ wcat,
                                                       const float* __restrict__ bcat,
                                                       float* __restrict__ gates, const float* __restrict__ cprev,
                                                       float* __restrict__ cout, float* __restrict__ zout,
                                                       float* __restrict__ zout2) {
  __shared__ float red[4 * MT * 16 * SK_LDS_STRIDE];
  const int j = blockIdx.x;
  const int64_t row0 = (int64_t)blockIdx.z * (MT * 16);
  const int e = threadIdx.x;
  // epilogue operands first: their latency hides under the gate product
  float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
  float cp = 0.f;
  {
    const int row = e >> 2, u = e & 3;
    const int64_t b = row0 + row;
    if (e < MT * 16 * 4 && b < B) {
      bb = *reinterpret_cast<const float4*>(bcat + (4 * j + u) * 4);
      if (cprev) cp = cprev[b * D + 4 * j + u];
    }
  }
  skinny_partial<MT>(Xs, KX, row0, B, wcat, KX, (int64_t)16 * j, (int64_t)4 * D, KX, red);
  __syncthreads();
  if (e < MT * 16 * 4) {
    const int row = e >> 2, u = e & 3;
    const int64_t b = row0 + row;
    if (b < B) {
      const int unit = 4 * j + u;
      const float gi = asr_sigmoid(skinny_reduced<MT>(red, row, u * 4 + 0) + bb.x);
      const float gf = asr_sigmoid(skinny_reduced<MT>(red, row, u * 4 + 1) + bb.y);
      const float gg = tanhf(skinny_reduced<MT>(red, row, u * 4 + 2) + bb.z);
      const float go = asr_sigmoid(skinny_reduced<MT>(red, row, u * 4 + 3) + bb.w);
      const float cn = gf * cp + gi * gg;
      *reinterpret_cast<float4*>(gates + (b * 4 * D) + unit * 4) = make_float4(gi, gf, gg, go);
      cout[b * D + unit] = cn;
      const float zn = go * tanhf(cn);
      zout[b * KX + unit] = zn;
      if (zout2) zout2[b * KX + unit] = zn;   // the dropout-masked operand copy shares the recurrent state
    }
  }
}

// decoder cell backward, pointwise part: dgates from dz (= G[s+1][:,0:D]) and the dc carry
__global__ void cell_bwd_kernel(int B, int D, int KX, const float* __restrict__ Gnext,
                                const float* __restrict__ gates, const float* __restrict__ cst,
                                const float* __restrict__ cprev, float* __restrict__ dcell,
                                float* __restrict__ dgates) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * D) return;
  const int b = idx / D, unit = idx % D;
  const float dh = Gnext[(int64_t)b * KX + unit];
  const float4 a = *reinterpret_cast<const float4*>(gates + (int64_t)b * 4 * D + unit * 4);
  const float ct = cst[idx];
  const float cp = cprev ? cprev[idx] : 0.f;
  const float tc = tanhf(ct);
  const float dc = dcell[idx] + dh * a.w * (1.f - tc * tc);
  float4 da;
  da.x = dc * a.z * a.x * (1.f - a.x);
  da.y = dc * cp * a.y * (1.f - a.y);
  da.z = dc * a.x * (1.f - a.z * a.z);
  da.w = dh * tc * a.w * (1.f - a.w);
  *reinterpret_cast<float4*>(dgates + (int64_t)b * 4 * D + unit * 4) = da;
  dcell[idx] = dc * a.y;
}

}  // namespace

int asr_cell_fwd_launch(int B, int D, int KX, const float* Xs, const float* wcat, const float* bcat, float* gates,
                        const float* cprev, float* cout, float* zout, float* zout2, hipStream_t stream) {
  if (D % 16 || KX % 16) return ASR_E_SHAPE;
  if (B <= 32)   // 16-row workgroups: twice the workgroups, half the shared-operand bytes each
    hipLaunchKernelGGL((cell_fwd_kernel<1>), dim3(D / 4, 1, (B + 15) / 16), dim3(256), 0, stream, B, D, KX, Xs, wcat, bcat, gates,
                       cprev, cout, zout, zout2);
  else
    hipLaunchKernelGGL((cell_fwd_kernel<2>), dim3(D / 4, 1, (B + 31) / 32), dim3(256), 0, stream, B, D, KX, Xs, wcat,
                       bcat, gates, cprev, cout, zout, zout2);
  ASR_CHECK_LAUNCH();
  return 0;
}

int asr_cell_bwd_launch(int B, int D, int KX, const float* Gnext, const float* gates, const float* cst,
                        const float* cprev, float* dcell, float* dgates, hipStream_t stream) {
  hipLaunchKernelGGL(cell_bwd_kernel, dim3((B * D + 255) / 256), dim3(256), 0, stream, B, D, KX, Gnext, gates, cst,
                     cprev, dcell, dgates);
  ASR_CHECK_LAUNCH();
  return 0;
}

extern "C" int asr_lstm_seq_fwd(int T, int B, int nb, int H, int ndir, float* gates, const float* w_hh,
                                const int32_t* lens, const int32_t* rowbase, const int32_t* rowext, float* y, float* c,
                                void* graphs, asr_stream_t stream_) {
  hipStream_t stream0 = (hipStream_t)stream_;
  if (!gates || !w_hh || !lens || !y || !c || T <= 0 || B <= 0 || H <= 0 || nb <= 0 || nb > B) return ASR_E_ARG;
  if ((rowbase == nullptr) != (rowext == nullptr)) return ASR_E_ARG;
  if (H % 16 || (ndir != 1 && ndir != 2)) return ASR_E_SHAPE;
  if (!asr_aligned16(gates) || !asr_aligned16(w_hh) || !asr_aligned16(y)) return ASR_E_ALIGN;
  struct { int kind, T, B, nb, H, ndir; const void *a, *b, *c, *d, *e, *f, *g; } key = {1, T, B, nb, H, ndir, gates, w_hh,
                                                                                     lens, y, c, rowbase, rowext};
  return asr_graph_run((AsrGraphCache*)graphs, &key, sizeof(key), stream0, [&](hipStream_t stream) -> int {
    for (int s = 0; s < T; ++s) {
      if (nb <= 16)
        hipLaunchKernelGGL((enc_step_fwd_kernel<1>), dim3(H / 4, ndir, 1), dim3(256), 0, stream, T, B, nb, H, ndir,
                           gates, w_hh, lens, y, c, s, rowbase, rowext);
      else
        hipLaunchKernelGGL((enc_step_fwd_kernel<2>), dim3(H / 4, ndir, (nb + 31) / 32), dim3(256), 0, stream, T, B, nb,
                           H, ndir, gates, w_hh, lens, y, c, s, rowbase, rowext);
    }
    ASR_CHECK_LAUNCH();
    return 0;
  });
}

#ifndef ASR_BWD_UNITS
#define ASR_BWD_UNITS 8
#endif
#ifndef ASR_BWD_MT
#define ASR_BWD_MT 1
#endif
#ifndef ASR_BWD_NW
#define ASR_BWD_NW 4
#endif

extern "C" int asr_lstm_seq_bwd(int T, int B, int nb, int H, int ndir, float* gates, const float* w_hhT,
                                const int32_t* lens, const int32_t* rowbase, const int32_t* rowext, const float* dy,
                                const float* c, float* dcarry, void* graphs, asr_stream_t stream_) {
  hipStream_t stream0 = (hipStream_t)stream_;
  if (!gates || !w_hhT || !lens || !dy || !c || !dcarry || T <= 0 || B <= 0 || H <= 0 || nb <= 0 || nb > B)
    return ASR_E_ARG;
  if ((rowbase == nullptr) != (rowext == nullptr)) return ASR_E_ARG;
  if (H % 16 || (ndir != 1 && ndir != 2)) return ASR_E_SHAPE;
  if (!asr_aligned16(gates) || !asr_aligned16(w_hhT)) return ASR_E_ALIGN;
  constexpr int U = ASR_BWD_UNITS, MTB = ASR_BWD_MT, NWB = ASR_BWD_NW;
  struct { int kind, T, B, nb, H, ndir; const void *a, *b, *c, *d, *e, *f, *g, *h; } key = {2, T, B, nb, H, ndir, gates, w_hhT,
                                                                                         lens, dy, c, dcarry, rowbase, rowext};
  return asr_graph_run((AsrGraphCache*)graphs, &key, sizeof(key), stream0, [&](hipStream_t stream) -> int {
    for (int s = 0; s < T; ++s) {
      hipLaunchKernelGGL((enc_step_bwd_kernel<MTB, U, NWB>), dim3(H / U, ndir, (nb + MTB * 16 - 1) / (MTB * 16)),
                         dim3(NWB * 64), 0, stream, T, B, nb, H, ndir, gates, w_hhT, lens, dy, c, dcarry, s, rowbase, rowext);
    }
    ASR_CHECK_LAUNCH();
    return 0;
  });
}

extern "C" void* asr_graphs_create(int max_entries) {
  AsrGraphCache* gc = new AsrGraphCache();
  if (max_entries > 0) gc->max_entries = (size_t)max_entries;
  return gc;
}

extern "C" void asr_graphs_destroy(void* graphs) {
  AsrGraphCache* gc = (AsrGraphCache*)graphs;
  if (!gc) return;
  for (auto& e : gc->entries)
    if (e.exec) (void)hipGraphExecDestroy(e.exec);
  delete gc;
}

extern "C" int asr_graphs_stats(void* graphs, int64_t* hits, int64_t* captures, int64_t* eager) {
  AsrGraphCache* gc = (AsrGraphCache*)graphs;
  if (!gc) return ASR_E_ARG;
  if (hits) *hits = (int64_t)gc->hits;
  if (captures) *captures = (int64_t)gc->captures;
  if (eager) *eager = (int64_t)gc->eager;
  return 0;
}
