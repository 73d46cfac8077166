// pack.hip — parameter layout conversion between the reference's torch layout (gate-major rows i,f,g,o) and the
// kernels' gate-interleaved layout (row = unit*4 + gate), forward (weights) and backward (gradients).
// One launch per LSTM layer / decoder cell instead of ~20 torch gathers, concatenations and copies each way
// (torch.nn.LSTM / LSTMCell parameters, model.py:67-68,262).
#include "common.h"

namespace {

struct LstmPackArgs {
  int H, I, ndir;
  const float* w_ih[2];
  const float* w_hh[2];
  const float* b_ih[2];
  const float* b_hh[2];
  float* w_ih_cat;   // [ndir*4H][I]
  float* w_hh_il;    // [ndir][4H][H]
  float* bias;       // [ndir*4H]  = b_ih + b_hh
};

// grid = ndir*4H (one interleaved row each), 128 threads
__global__ void lstm_pack_kernel(LstmPackArgs a) {
  const int r = blockIdx.x, H = a.H, I = a.I;
  const int d = r / (4 * H), ri = r - d * 4 * H, u = ri >> 2, g = ri & 3;
  const int src = g * H + u;
  const float* wi = a.w_ih[d] + (int64_t)src * I;
  float* oi = a.w_ih_cat + (int64_t)r * I;
  for (int k = threadIdx.x; k < I; k += blockDim.x) oi[k] = wi[k];
  const float* wh = a.w_hh[d] + (int64_t)src * H;
  float* oh = a.w_hh_il + ((int64_t)d * 4 * H + ri) * H;
  for (int k = threadIdx.x; k < H; k += blockDim.x) oh[k] = wh[k];
  if (threadIdx.x == 0) a.bias[r] = a.b_ih[d][src] + a.b_hh[d][src];
}

struct LstmUnpackArgs {
  int H, I, ndir;
  const float* dw_ih_cat;   // [ndir*4H][I] interleaved rows
  const float* dw_hh_il;    // [ndir][4H][H]
  const float* db_il;       // [ndir*4H]
  float* dw_ih[2];
  float* dw_hh[2];
  float* db[2];
  float* db2[2];            // optional second copy of the bias gradient (b_ih and b_hh receive the same values)
};

// grid = ndir*4H (one torch-layout row each)
__global__ void lstm_unpack_kernel(LstmUnpackArgs a) {
  const int r = blockIdx.x, H = a.H, I = a.I;
  const int d = r / (4 * H), rt = r - d * 4 * H, g = rt / H, u = rt - g * H;
  const int ri = u * 4 + g;
  const float* si = a.dw_ih_cat + ((int64_t)d * 4 * H + ri) * I;
  float* oi = a.dw_ih[d] + (int64_t)rt * I;
  for (int k = threadIdx.x; k < I; k += blockDim.x) oi[k] = si[k];
  const float* sh = a.dw_hh_il + ((int64_t)d * 4 * H + ri) * H;
  float* oh = a.dw_hh[d] + (int64_t)rt * H;
  for (int k = threadIdx.x; k < H; k += blockDim.x) oh[k] = sh[k];
  if (threadIdx.x == 0) {
    const float v = a.db_il[d * 4 * H + ri];
    a.db[d][rt] = v;
    if (a.db2[d]) a.db2[d][rt] = v;
  }
}

// decoder cell: wcat[u*4+g] = [w_hh[src][0:D] | w_ih[src][E:E+O] | w_ih[src][0:E]], bcat = b_ih + b_hh
__global__ void cell_pack_kernel(int D, int O, int E, const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                 const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                 float* __restrict__ wcat, float* __restrict__ bcat) {
  const int r = blockIdx.x, u = r >> 2, g = r & 3, src = g * D + u, KX = D + O + E;
  float* o = wcat + (int64_t)r * KX;
  const float* hh = w_hh + (int64_t)src * D;
  const float* ih = w_ih + (int64_t)src * (E + O);
  for (int k = threadIdx.x; k < KX; k += blockDim.x)
    o[k] = k < D ? hh[k] : (k < D + O ? ih[E + (k - D)] : ih[k - D - O]);
  if (threadIdx.x == 0) bcat[r] = b_ih[src] + b_hh[src];
}

__global__ void cell_unpack_kernel(int D, int O, int E, const float* __restrict__ dwcat, const float* __restrict__ db_il,
                                   float* __restrict__ dw_ih, float* __restrict__ dw_hh, float* __restrict__ db) {
  const int rt = blockIdx.x, g = rt / D, u = rt - g * D, ri = u * 4 + g, KX = D + O + E;
  const float* s = dwcat + (int64_t)ri * KX;
  float* hh = dw_hh + (int64_t)rt * D;
  float* ih = dw_ih + (int64_t)rt * (E + O);
  for (int k = threadIdx.x; k < KX; k += blockDim.x) {
    const float v = s[k];
    if (k < D) hh[k] = v;
    else if (k < D + O) ih[E + (k - D)] = v;
    else ih[k - D - O] = v;
  }
  if (threadIdx.x == 0) db[rt] = db_il[ri];
}

}  // namespace

extern "C" int asr_lstm_pack_f32(int H, int I, int ndir, const float* const* w_ih, const float* const* w_hh,
                                 const float* const* b_ih, const float* const* b_hh, float* w_ih_cat, float* w_hh_il,
                                 float* bias, asr_stream_t stream) {
  if (H <= 0 || I <= 0 || (ndir != 1 && ndir != 2) || !w_ih || !w_hh || !b_ih || !b_hh || !w_ih_cat || !w_hh_il || !bias)
    return ASR_E_ARG;
  LstmPackArgs a;
  a.H = H; a.I = I; a.ndir = ndir;
  for (int d = 0; d < 2; ++d) {
    const int s = d < ndir ? d : 0;
    if (!w_ih[s] || !w_hh[s] || !b_ih[s] || !b_hh[s]) return ASR_E_ARG;
    a.w_ih[d] = w_ih[s]; a.w_hh[d] = w_hh[s]; a.b_ih[d] = b_ih[s]; a.b_hh[d] = b_hh[s];
  }
  a.w_ih_cat = w_ih_cat; a.w_hh_il = w_hh_il; a.bias = bias;
  hipLaunchKernelGGL(lstm_pack_kernel, dim3(ndir * 4 * H), dim3(128), 0, (hipStream_t)stream, a);
  ASR_CHECK_LAUNCH();
  return 0;
}

extern "C" int asr_lstm_unpack2_f32(int H, int I, int ndir, const float* dw_ih_cat, const float* dw_hh_il,
                                    const float* db_il, float* const* dw_ih, float* const* dw_hh, float* const* db,
                                    float* const* db2, asr_stream_t stream) {
  if (H <= 0 || I <= 0 || (ndir != 1 && ndir != 2) || !dw_ih_cat || !dw_hh_il || !db_il || !dw_ih || !dw_hh || !db)
    return ASR_E_ARG;
  LstmUnpackArgs a;
  a.H = H; a.I = I; a.ndir = ndir; a.dw_ih_cat = dw_ih_cat; a.dw_hh_il = dw_hh_il; a.db_il = db_il;
  for (int d = 0; d < 2; ++d) {
    const int s = d < ndir ? d : 0;
    if (!dw_ih[s] || !dw_hh[s] || !db[s]) return ASR_E_ARG;
    a.dw_ih[d] = dw_ih[s]; a.dw_hh[d] = dw_hh[s]; a.db[d] = db[s];
    a.db2[d] = db2 ? db2[s] : nullptr;
  }
  hipLaunchKernelGGL(lstm_unpack_kernel, dim3(ndir * 4 * H), dim3(128), 0, (hipStream_t)stream, a);
  ASR_CHECK_LAUNCH();
  return 0;
}

extern "C" int asr_lstm_unpack_f32(int H, int I, int ndir, const float* dw_ih_cat, const float* dw_hh_il,
                                   const float* db_il, float* const* dw_ih, float* const* dw_hh, float* const* db,
                                   asr_stream_t stream) {
  return asr_lstm_unpack2_f32(H, I, ndir, dw_ih_cat, dw_hh_il, db_il, dw_ih, dw_hh, db, nullptr, stream);
}

// teacher-forced decoder input: X[s][b] = [0 (z_{s-1}, ctx_{s-1}: written by the recurrence) | emb[tok[b][s]]], Xd = X with
// the dropout mask on the embedding part, fed[s][b] = tok[b][s]; slab L (the recurrence's last z / ctx) all zero.
// One launch for what was two fills, a transpose copy, a gather, a multiply and two strided copies (model.py:301-306, 337).
__global__ void dec_prepare_kernel(int L, int B, int DO4, int E4, const long long* __restrict__ tok, int64_t tok_rs,
                                   const float4* __restrict__ emb, const float* __restrict__ xmask, int OE,
                                   float4* __restrict__ X, float4* __restrict__ Xd, long long* __restrict__ fed) {
  const int KX4 = DO4 + E4;
  const int64_t n = (int64_t)(L + 1) * B * KX4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % KX4);
    const int64_t sb = i / KX4;
    const int b = (int)(sb % B), s = (int)(sb / B);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f), vd = v;
    if (c4 >= DO4 && s < L) {
      const long long t = tok[(int64_t)b * tok_rs + s];
      const int e4 = c4 - DO4;
      v = emb[t * E4 + e4];
      vd = v;
      if (xmask) {
        const float4 m = *reinterpret_cast<const float4*>(xmask + sb * OE + (OE - 4 * E4) + 4 * e4);
        vd = make_float4(v.x * m.x, v.y * m.y, v.z * m.z, v.w * m.w);
      }
      if (e4 == 0) fed[sb] = t;
    }
    X[i] = v;
    if (Xd) Xd[i] = vd;
  }
}

extern "C" int asr_dec_prepare_f32(int L, int B, int D, int O, int E, const long long* tokens, int64_t tok_row_stride,
                                   const float* emb_w, const float* xmask, float* X, float* Xd, long long* fed,
                                   asr_stream_t stream) {
  if (L <= 0 || B <= 0 || D <= 0 || O < 0 || E <= 0 || !tokens || !emb_w || !X || !fed) return ASR_E_ARG;
  if ((D + O) % 4 || E % 4 || (xmask && (O + E) % 4)) return ASR_E_SHAPE;
  if (!asr_aligned16(emb_w) || !asr_aligned16(X) || (Xd && !asr_aligned16(Xd)) || (xmask && !asr_aligned16(xmask))) return ASR_E_ALIGN;
  const int64_t n = (int64_t)(L + 1) * B * ((D + O + E) / 4);
  const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(dec_prepare_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L, B, (D + O) / 4, E / 4, tokens,
                     tok_row_stride, reinterpret_cast<const float4*>(emb_w), xmask, O + E, reinterpret_cast<float4*>(X),
                     reinterpret_cast<float4*>(Xd), fed);
  ASR_CHECK_LAUNCH();
  return 0;
}

extern "C" int asr_cell_pack_f32(int D, int O, int E, const float* w_ih, const float* w_hh, const float* b_ih,
                                 const float* b_hh, float* wcat, float* bcat, asr_stream_t stream) {
  if (D <= 0 || O < 0 || E < 0 || !w_ih || !w_hh || !b_ih || !b_hh || !wcat || !bcat) return ASR_E_ARG;
  hipLaunchKernelGGL(cell_pack_kernel, dim3(4 * D), dim3(128), 0, (hipStream_t)stream, D, O, E, w_ih, w_hh, b_ih, b_hh,
                     wcat, bcat);
  ASR_CHECK_LAUNCH();
  return 0;
}

extern "C" int asr_cell_unpack_f32(int D, int O, int E, const float* dwcat, const float* db_il, float* dw_ih,
                                   float* dw_hh, float* db, asr_stream_t stream) {
  if (D <= 0 || O < 0 || E < 0 || !dwcat || !db_il || !dw_ih || !dw_hh || !db) return ASR_E_ARG;
  hipLaunchKernelGGL(cell_unpack_kernel, dim3(4 * D), dim3(128), 0, (hipStream_t)stream, D, O, E, dwcat, db_il, dw_ih,
                     dw_hh, db);
  ASR_CHECK_LAUNCH();
  return 0;
}
