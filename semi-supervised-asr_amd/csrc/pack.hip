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

// Several layers in one launch (asr_lstm_pack_multi_f32 / asr_lstm_unpack_multi_f32): the jobs travel by value, a block finds
// its job from the prefix sums of their row counts.
constexpr int PACK_MAX_JOBS = ASR_PACK_MAX_LAYERS;
struct LstmPackJobs {
  int n;
  int first[PACK_MAX_JOBS + 1];      // first block of job j; first[n] = grid size
  LstmPackArgs job[PACK_MAX_JOBS];
};

// one row of n floats, 16 bytes per lane when the rows allow it (n % 4 == 0 and both rows 16-byte aligned: with scalar copies
// the three-layer pack moved 2.7 TB/s)
__device__ __forceinline__ void copy_row(float* __restrict__ dst, const float* __restrict__ src, int n) {
  if ((n & 3) == 0 && ((((uintptr_t)dst) | ((uintptr_t)src)) & 15) == 0) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int k = threadIdx.x; k < (n >> 2); k += blockDim.x) d4[k] = s4[k];
  } else {
    for (int k = threadIdx.x; k < n; k += blockDim.x) dst[k] = src[k];
  }
}

__device__ __forceinline__ void lstm_pack_row(const LstmPackArgs& a, int r) {
  const int H = a.H, I = a.I;
  const int d = r / (4 * H), ri = r - d * 4 * H, u = ri >> 2, g = ri & 3;
  const int src = g * H + u;
  copy_row(a.w_ih_cat + (int64_t)r * I, a.w_ih[d] + (int64_t)src * I, I);
  copy_row(a.w_hh_il + ((int64_t)d * 4 * H + ri) * H, a.w_hh[d] + (int64_t)src * H, H);
  if (threadIdx.x == 0) a.bias[r] = a.b_ih[d][src] + a.b_hh[d][src];
}

// grid = ndir*4H (one interleaved row each), 128 threads
__global__ void lstm_pack_kernel(LstmPackArgs a) { lstm_pack_row(a, blockIdx.x); }

__global__ void lstm_pack_multi_kernel(LstmPackJobs t) {
  int j = 0;
  while (j + 1 < t.n && (int)blockIdx.x >= t.first[j + 1]) ++j;
  lstm_pack_row(t.job[j], blockIdx.x - t.first[j]);
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

struct LstmUnpackJobs {
  int n;
  int first[PACK_MAX_JOBS + 1];
  LstmUnpackArgs job[PACK_MAX_JOBS];
};

__device__ __forceinline__ void lstm_unpack_row(const LstmUnpackArgs& a, int r) {
  const int H = a.H, I = a.I;
  const int d = r / (4 * H), rt = r - d * 4 * H, g = rt / H, u = rt - g * H;
  const int ri = u * 4 + g;
  copy_row(a.dw_ih[d] + (int64_t)rt * I, a.dw_ih_cat + ((int64_t)d * 4 * H + ri) * I, I);
  copy_row(a.dw_hh[d] + (int64_t)rt * H, a.dw_hh_il + ((int64_t)d * 4 * H + ri) * H, H);
  if (threadIdx.x == 0) {
    const float v = a.db_il[d * 4 * H + ri];
    a.db[d][rt] = v;
    if (a.db2[d]) a.db2[d][rt] = v;
  }
}

// grid = ndir*4H (one torch-layout row each)
__global__ void lstm_unpack_kernel(LstmUnpackArgs a) { lstm_unpack_row(a, blockIdx.x); }

__global__ void lstm_unpack_multi_kernel(LstmUnpackJobs t) {
  int j = 0;
  while (j + 1 < t.n && (int)blockIdx.x >= t.first[j + 1]) ++j;
  lstm_unpack_row(t.job[j], blockIdx.x - t.first[j]);
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

// Everything the decoder's sequence kernels read in their own layout, one launch (asr_dec_pack_f32): the blocks of
// cell_pack_kernel, then 32 x 32 tile transposes through LDS - wcatT [KX][4D] (straight from the torch layout), wdecT [D][A],
// wattT [C][A] - the copies the backward (wcatT, wdecT) and the per-step forward (wattT) want.
struct DecPackArgs {
  int D, O, E, A, C;
  const float *w_ih, *w_hh, *b_ih, *b_hh, *wdec, *watt;
  float *wcat, *bcat, *wcatT, *wdecT, *wattT;
  int first_catT, first_decT, first_attT, tiles_catT_r, tiles_decT_r, tiles_attT_r;   // block ranges, tiles per source row block
};

// dst[c][r] = src(r, c) for the 32 x 32 tile (tr, tc); src(r, c) given by a functor; 128 threads
template <class Src>
__device__ __forceinline__ void transpose_tile(Src src, int R, int Cn, int tr, int tc, float* __restrict__ dst, float (*tile)[33]) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 4) {
    const int r = 32 * tr + i, c = 32 * tc + tx;
    tile[i][tx] = (r < R && c < Cn) ? src(r, c) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 4) {
    const int c = 32 * tc + i, r = 32 * tr + tx;
    if (c < Cn && r < R) dst[(int64_t)c * R + r] = tile[tx][i];
  }
}

__global__ __launch_bounds__(128) void dec_pack_kernel(DecPackArgs a) {
  __shared__ float tile[32][33];
  const int D = a.D, O = a.O, E = a.E, KX = D + O + E;
  const int bid = blockIdx.x;
  auto cat_at = [&](int r, int k) -> float {            // element (r, k) of the interleaved wcat, from the torch layout
    const int u = r >> 2, g = r & 3, src = g * D + u;
    return k < D ? a.w_hh[(int64_t)src * D + k]
                 : (k < D + O ? a.w_ih[(int64_t)src * (E + O) + E + (k - D)] : a.w_ih[(int64_t)src * (E + O) + (k - D - O)]);
  };
  if (bid < a.first_catT) {                              // one interleaved row of wcat / one element of bcat
    const int r = bid, u = r >> 2, g = r & 3, src = g * D + u;
    float* o = a.wcat + (int64_t)r * KX;
    for (int k = threadIdx.x; k < KX; k += blockDim.x) o[k] = cat_at(r, k);
    if (threadIdx.x == 0) a.bcat[r] = a.b_ih[src] + a.b_hh[src];
  } else if (bid < a.first_decT) {
    const int t = bid - a.first_catT;
    transpose_tile(cat_at, 4 * D, KX, t % a.tiles_catT_r, t / a.tiles_catT_r, a.wcatT, tile);
  } else if (bid < a.first_attT) {
    const int t = bid - a.first_decT;
    const float* w = a.wdec;
    transpose_tile([&](int r, int c) { return w[(int64_t)r * D + c]; }, a.A, D, t % a.tiles_decT_r, t / a.tiles_decT_r, a.wdecT, tile);
  } else {
    const int t = bid - a.first_attT;
    const float* w = a.watt;
    const int C = a.C;
    transpose_tile([&](int r, int c) { return w[(int64_t)r * C + c]; }, a.A, C, t % a.tiles_attT_r, t / a.tiles_attT_r, a.wattT, tile);
  }
}

// out_i[j] = sum_r src_i[r * n_i + j] for up to four [rows][n_i] matrices (the per-utterance partial gradients of the decoder
// backward): one launch instead of one reduction each.  Block = 256 columns of one part.
struct ColsumParts {
  int nparts, rows;
  int first[5];
  const float* src[4];
  float* dst[4];
  int n[4];
};
__global__ __launch_bounds__(256) void colsum_parts_kernel(ColsumParts a) {
  int i = 0;
  while (i + 1 < a.nparts && (int)blockIdx.x >= a.first[i + 1]) ++i;
  const int j = 256 * (blockIdx.x - a.first[i]) + threadIdx.x;
  if (j >= a.n[i]) return;
  const float* s = a.src[i] + j;
  float v0 = 0.f, v1 = 0.f;
  int r = 0;
  for (; r + 1 < a.rows; r += 2) { v0 += s[(int64_t)r * a.n[i]]; v1 += s[(int64_t)(r + 1) * a.n[i]]; }
  if (r < a.rows) v0 += s[(int64_t)r * a.n[i]];
  a.dst[i][j] = v0 + v1;
}

__global__ void cell_unpack_kernel(int D, int O, int E, const float* __restrict__ dwcat, const float* __restrict__ db_il,
                                   float* __restrict__ dw_ih, float* __restrict__ dw_hh, float* __restrict__ db,
                                   float* __restrict__ db2) {
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
  if (threadIdx.x == 0) {
    db[rt] = db_il[ri];
    if (db2) db2[rt] = db_il[ri];
  }
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

// Gradient of the embedding rows behind dec_prepare_kernel (autograd of nn.Embedding, model.py:337): demb[tok[r]][:] +=
// g[r][:] for r < rows, g row-strided (the embedding columns of the decoder's dX buffer, read where they lie).  Real label
// matrices are SKEWED - the <EOS> padding makes one of the V rows the target of a third of all adds - so one global atomic
// per (row, element) serialises on that row's addresses (52 us inside a cfg-2 step, 13 with uniform random tokens).  A
// workgroup therefore folds its rows into an LDS table [V][E] (ds_add_f32) and adds the table's non-zero entries to global
// memory once.  Per row lane the loop is a chain of dependent loads (token -> gradient row): four rows per trip, all eight
// loads in flight before the first add (one row per trip: 24 us).
__global__ __launch_bounds__(256) void embedding_grad_kernel(int64_t rows, int E4, int V, const long long* __restrict__ tok,
                                                             const float* __restrict__ g, int64_t ldg, float* __restrict__ demb) {
  extern __shared__ float table[];                       // [V][4 E4]
  const int E = 4 * E4;
  for (int i = threadIdx.x; i < V * E; i += blockDim.x) table[i] = 0.f;
  __syncthreads();
  const int lanes = blockDim.x / E4;                     // rows in flight per trip and block (x 4)
  const int c = threadIdx.x % E4, rl = threadIdx.x / E4;
  if (rl < lanes) {
    const int64_t stride = (int64_t)gridDim.x * lanes;
    for (int64_t r0 = (int64_t)blockIdx.x * lanes + rl; r0 < rows; r0 += 4 * stride) {
      long long t[4];
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = r0 + u * stride;
        t[u] = r < rows ? tok[r] : -1;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = r0 + u * stride;
        const bool ok = t[u] >= 0 && t[u] < V;            // (-1: a step that was not fed a token)
        v[u] = *reinterpret_cast<const float4*>(g + (ok ? r : 0) * ldg + 4 * c);
        if (!ok) t[u] = -1;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (t[u] < 0) continue;
        float* d = table + t[u] * E + 4 * c;
        atomicAdd(d, v[u].x); atomicAdd(d + 1, v[u].y); atomicAdd(d + 2, v[u].z); atomicAdd(d + 3, v[u].w);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < V * E; i += blockDim.x) {
    const float v = table[i];
    if (v != 0.f) atomicAdd(demb + i, v);
  }
}

extern "C" int asr_embedding_grad_f32(int64_t rows, int E, int V, const long long* tokens, const float* grad, int64_t ldg,
                                      float* demb, asr_stream_t stream) {
  if (rows <= 0 || E <= 0 || V <= 0 || !tokens || !grad || !demb) return ASR_E_ARG;
  if (E % 4 || ldg % 4 || E / 4 > 256 || (size_t)V * E * sizeof(float) > 64 * 1024) return ASR_E_SHAPE;
  if (!asr_aligned16(grad)) return ASR_E_ALIGN;
  const int lanes = 256 / (E / 4);
  int64_t blocks = (rows + (int64_t)lanes * 8 - 1) / ((int64_t)lanes * 8);       // >= 8 rows (two trips) per row lane
  if (blocks > 64) blocks = 64;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(embedding_grad_kernel, dim3((unsigned)blocks), dim3(256), (size_t)V * E * sizeof(float), (hipStream_t)stream,
                     rows, E / 4, V, tokens, grad, ldg, demb);
  ASR_CHECK_LAUNCH();
  return 0;
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
                                   float* dw_hh, float* db, float* db2, asr_stream_t stream) {
  if (D <= 0 || O < 0 || E < 0 || !dwcat || !db_il || !dw_ih || !dw_hh || !db) return ASR_E_ARG;
  hipLaunchKernelGGL(cell_unpack_kernel, dim3(4 * D), dim3(128), 0, (hipStream_t)stream, D, O, E, dwcat, db_il, dw_ih,
                     dw_hh, db, db2);
  ASR_CHECK_LAUNCH();
  return 0;
}

extern "C" int asr_dec_pack_f32(int D, int O, int E, int A, int C, const float* w_ih, const float* w_hh, const float* b_ih,
                                const float* b_hh, const float* wdec, const float* watt, float* wcat, float* bcat,
                                float* wcatT, float* wdecT, float* wattT, asr_stream_t stream) {
  if (D <= 0 || O < 0 || E < 0 || A <= 0 || C <= 0 || !w_ih || !w_hh || !b_ih || !b_hh || !wcat || !bcat) return ASR_E_ARG;
  if ((wdecT && !wdec) || (wattT && !watt)) return ASR_E_ARG;
  DecPackArgs a;
  a.D = D; a.O = O; a.E = E; a.A = A; a.C = C;
  a.w_ih = w_ih; a.w_hh = w_hh; a.b_ih = b_ih; a.b_hh = b_hh; a.wdec = wdec; a.watt = watt;
  a.wcat = wcat; a.bcat = bcat; a.wcatT = wcatT; a.wdecT = wdecT; a.wattT = wattT;
  const int KX = D + O + E;
  auto tiles = [](int n) { return (n + 31) / 32; };
  a.tiles_catT_r = tiles(4 * D); a.tiles_decT_r = tiles(A); a.tiles_attT_r = tiles(A);
  a.first_catT = 4 * D;
  a.first_decT = a.first_catT + (wcatT ? a.tiles_catT_r * tiles(KX) : 0);
  a.first_attT = a.first_decT + (wdecT ? a.tiles_decT_r * tiles(D) : 0);
  const int total = a.first_attT + (wattT ? a.tiles_attT_r * tiles(C) : 0);
  hipLaunchKernelGGL(dec_pack_kernel, dim3(total), dim3(128), 0, (hipStream_t)stream, a);
  ASR_CHECK_LAUNCH();
  return 0;
}

extern "C" int asr_colsum_parts_f32(int nparts, int rows, const float* const* src, const int32_t* n, float* const* dst,
                                    asr_stream_t stream) {
  if (nparts <= 0 || nparts > 4 || rows <= 0 || !src || !n || !dst) return ASR_E_ARG;
  ColsumParts a;
  a.nparts = nparts; a.rows = rows;
  int at = 0;
  for (int i = 0; i < 4; ++i) {
    a.first[i] = at;
    if (i < nparts) {
      if (!src[i] || !dst[i] || n[i] <= 0) return ASR_E_ARG;
      a.src[i] = src[i]; a.dst[i] = dst[i]; a.n[i] = n[i];
      at += (n[i] + 255) / 256;
    } else {
      a.src[i] = nullptr; a.dst[i] = nullptr; a.n[i] = 0;
    }
  }
  a.first[4] = at;
  hipLaunchKernelGGL(colsum_parts_kernel, dim3(at), dim3(256), 0, (hipStream_t)stream, a);
  ASR_CHECK_LAUNCH();
  return 0;
}

// several LSTM layers in one launch; jobs[j] as the arguments of asr_lstm_pack_f32 / asr_lstm_unpack2_f32
extern "C" int asr_lstm_pack_multi_f32(int nlayers, const asr_lstm_pack_job_t* jobs, asr_stream_t stream) {
  if (nlayers <= 0 || nlayers > PACK_MAX_JOBS || !jobs) return ASR_E_ARG;
  LstmPackJobs t;
  t.n = nlayers;
  int at = 0;
  for (int j = 0; j < PACK_MAX_JOBS; ++j) {
    t.first[j] = at;
    if (j >= nlayers) { t.job[j] = t.job[0]; continue; }
    const asr_lstm_pack_job_t& q = jobs[j];
    if (q.H <= 0 || q.I <= 0 || (q.ndir != 1 && q.ndir != 2) || !q.w_ih_cat || !q.w_hh_il || !q.bias) return ASR_E_ARG;
    LstmPackArgs& a = t.job[j];
    a.H = q.H; a.I = q.I; a.ndir = q.ndir;
    for (int d = 0; d < 2; ++d) {
      const int s = d < q.ndir ? d : 0;
      if (!q.w_ih[s] || !q.w_hh[s] || !q.b_ih[s] || !q.b_hh[s]) return ASR_E_ARG;
      a.w_ih[d] = q.w_ih[s]; a.w_hh[d] = q.w_hh[s]; a.b_ih[d] = q.b_ih[s]; a.b_hh[d] = q.b_hh[s];
    }
    a.w_ih_cat = q.w_ih_cat; a.w_hh_il = q.w_hh_il; a.bias = q.bias;
    at += q.ndir * 4 * q.H;
  }
  t.first[PACK_MAX_JOBS] = at;
  hipLaunchKernelGGL(lstm_pack_multi_kernel, dim3(at), dim3(128), 0, (hipStream_t)stream, t);
  ASR_CHECK_LAUNCH();
  return 0;
}

extern "C" int asr_lstm_unpack_multi_f32(int nlayers, const asr_lstm_unpack_job_t* jobs, asr_stream_t stream) {
  if (nlayers <= 0 || nlayers > PACK_MAX_JOBS || !jobs) return ASR_E_ARG;
  LstmUnpackJobs t;
  t.n = nlayers;
  int at = 0;
  for (int j = 0; j < PACK_MAX_JOBS; ++j) {
    t.first[j] = at;
    if (j >= nlayers) { t.job[j] = t.job[0]; continue; }
    const asr_lstm_unpack_job_t& q = jobs[j];
    if (q.H <= 0 || q.I <= 0 || (q.ndir != 1 && q.ndir != 2) || !q.dw_ih_cat || !q.dw_hh_il || !q.db_il) return ASR_E_ARG;
    LstmUnpackArgs& a = t.job[j];
    a.H = q.H; a.I = q.I; a.ndir = q.ndir; a.dw_ih_cat = q.dw_ih_cat; a.dw_hh_il = q.dw_hh_il; a.db_il = q.db_il;
    for (int d = 0; d < 2; ++d) {
      const int s = d < q.ndir ? d : 0;
      if (!q.dw_ih[s] || !q.dw_hh[s] || !q.db[s]) return ASR_E_ARG;
      a.dw_ih[d] = q.dw_ih[s]; a.dw_hh[d] = q.dw_hh[s]; a.db[d] = q.db[s]; a.db2[d] = q.db2[s];
    }
    at += q.ndir * 4 * q.H;
  }
  t.first[PACK_MAX_JOBS] = at;
  hipLaunchKernelGGL(lstm_unpack_multi_kernel, dim3(at), dim3(128), 0, (hipStream_t)stream, t);
  ASR_CHECK_LAUNCH();
  return 0;
}
