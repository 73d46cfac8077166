// dec_persist.hip — persistent, XCD-local decoder sequence kernel (teacher-forced forward).
//
// Replaces the per-step launch chain of asr_dec_seq_fwd (cell -> W_dec z -> energies -> softmax + context; 4
// kernels x L steps, ~31 us per step of which ~20 us are kernel boundaries) by ONE launch for all L steps of
// Decoder.forward (model.py:324-351) on the teacher-forced path.
//
//   * Utterances are independent, so each of the 8 XCDs owns 4 batch rows for the whole sequence and every
//     exchange stays inside that XCD's L2 (persist.h).  A batch of 32 fills the 8 XCDs x 32 CUs = 256 CUs.
//   * Cell: CU j owns D/32 units (all four gates).  Its 4*D/32 rows of W_cat (K = D+O+E) stay in REGISTERS for
//     the whole sequence (8 waves split K: 144 VGPRs at cfg-2); the product runs on v_mfma_f32_4x4x1 with the 4
//     batch rows as the B operand, K-partials are summed through LDS by the pointwise threads.
//   * Attention energies: CU j owns A/32 attention columns for all 4 rows x T' frames.  Its slice of
//     P = mlp_enc(enc_h) never changes during the sequence and lives in registers; W_dec (A/32 x D) too.
//     Partial energies (sum over the CU's columns) are exchanged.
//   * Softmax / context / location conv: CU j = (row j>>3, part j&7) sums the 32 partial energies of its row,
//     runs the unmasked softmax (temperature F4), forms O/8 context outputs from its LDS-resident
//     slice of Q = enc_h W_o^T, and computes the location-conv features of 16 frames for the NEXT step on the
//     f32 MFMA (Toeplitz product) while the z hand-off is in flight.
//   * Hand-offs per step on the critical path: ctx_{s-1} -> cell, z_s -> W_dec z_s, partial energies -> softmax.
//     The conv features ride a fourth exchange that is published ~2 us before it is needed.
// Everything the backward needs (gates, c, S = tanh(..), conv features, attention weights, X/Xd) is written to
// the same buffers as the per-step path, so asr_dec_seq_bwd runs unchanged on the result.
#include "persist.h"

#ifndef ASR_DP_ABL
#define ASR_DP_ABL 0
#endif
#ifndef ASR_DP_FULL
#define ASR_DP_FULL false
#endif

namespace {

constexpr int DP_NT = 512;          // 8 waves
constexpr int DP_TPM = 128;         // max encoder frames T'
constexpr int DP_FPC = 16;          // conv frames per CU (8 CUs per row)
constexpr int DP_KMAX = 100;        // max conv half width
constexpr int DP_TAPS4 = 208;       // LDS row length of the filter image (>= 2*KMAX+1 rounded to 4)
constexpr int DP_WLEN = DP_TPM + 2 * DP_KMAX + 8;
constexpr int DP_NP = DP_TPM / 8;   // (row, frame) pairs per score thread
// exchange layout per group, in floats
constexpr int DX_Z = 0;                            // [2][4][512]
constexpr int DX_C = DX_Z + 2 * 4 * 512;           // [2][4][512]
constexpr int DX_F = DX_C + 2 * 4 * 512;           // [2][4][16][TPM]
constexpr int DX_E = DX_F + 2 * 4 * 16 * DP_TPM;   // [2][32][4][TPM]
constexpr int DX_GROUP = DX_E + 2 * 32 * 4 * DP_TPM;

struct DecPersistArgs {
  int B, nb, Tp, C, K, L;
  float scaling;
  const float *P, *Q, *bo, *wcat, *bcat, *wdec, *convw, *watt, *gvec, *w0, *xmask;
  float *X, *Xd, *gates, *cstate, *fconv, *S, *energy, *ws;
  float* xch;
  unsigned* ctrl;
};

__device__ __forceinline__ float dp_tanh(float x) {   // same formula as decoder.hip:fast_tanh
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
}

template <int DD, int AA, int OO, int EE>
struct DecDims {
  static constexpr int KX = DD + OO + EE;
  static constexpr int KXW = KX / 8;       // K columns of the cell product per wave
  static constexpr int DU = DD / 32;       // cell units per CU
  static constexpr int AU = AA / 32;       // attention columns per CU
  static constexpr int OQ = OO / 8;        // context outputs per CU
  static constexpr int DKW = DD / 8;       // W_dec product: k's per wave
  static constexpr int DKQ = DD / 32;      // ... per (wave, k-sub)
  static constexpr int XS = KX + 4;        // padded LDS row of the cell operand
  static constexpr int NZ = (2 * DD + DP_NT - 1) / DP_NT;     // 8-byte pairs per thread when gathering z
  static constexpr int NC = (2 * OO + DP_NT - 1) / DP_NT;
  static constexpr int NE = (4 * EE + DP_NT - 1) / DP_NT;     // embedding values per thread
  static constexpr size_t lds_floats = 4 * XS + 8 * 64 * 5 + 64 + 4 * 16 * DP_TPM + DP_TPM * OQ + 16 * DP_TAPS4 +
                                       DP_WLEN + 8 * 16 * 17 + 8 * DP_TPM + DP_TPM + 8 * 64 + 32 * 4 * 64 + 8;
  static_assert(KX % 32 == 0 && DD % 32 == 0 && AA % 32 == 0 && OO % 8 == 0, "slice sizes");
  static_assert(DU <= 16 && AU <= 16 && OQ <= 64 && DD <= 512 && OO <= 512, "per-CU slices must fit the mappings");
};

template <int DD, int AA, int OO, int EE>
__global__ __launch_bounds__(DP_NT) void dec_persist_fwd_kernel(DecPersistArgs a) {
  using DM = DecDims<DD, AA, OO, EE>;
  constexpr int KX = DM::KX, KXW = DM::KXW, DU = DM::DU, AU = DM::AU, OQ = DM::OQ, DKW = DM::DKW, DKQ = DM::DKQ;
  constexpr int XS = DM::XS, NZ = DM::NZ, NC = DM::NC, NE = DM::NE;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* xs = sm;                            // [4][XS]      cell operand [z | ctx(masked) | emb(masked)] per row
  float* part = xs + 4 * XS;                 // [8][64][5]   K-partials (cell product, then W_dec product)
  float* dps = part + 8 * 64 * 5;            // [4][16]      W_dec z slice
  float* fs = dps + 64;                      // [4][16][TPM] conv features of all rows
  float* Qs = fs + 4 * 16 * DP_TPM;          // [TPM][OQ]    Q slice of this CU's row
  float* Fs = Qs + DP_TPM * OQ;              // [16][TAPS4]  conv filters, zero padded
  float* wp = Fs + 16 * DP_TAPS4;            // [WLEN]       previous attention weights of this CU's row, zero halo
  float* cred = wp + DP_WLEN;                // [8][16][17]  conv partial tiles
  float* epart = cred + 8 * 16 * 17;         // [8][TPM]     partial energy sums
  float* wsm = epart + 8 * DP_TPM;           // [TPM]        attention weights of this step
  float* cpart = wsm + DP_TPM;               // [8][64]      context partials
  float* Ps = cpart + 8 * 64;                // [32 tiles][4 rows][64 lanes]  P slice in the score-lane layout
  int* role = reinterpret_cast<int*>(Ps + 32 * 4 * 64);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int g, slice;
  take_role(a.ctrl, role, g, slice);
  if (slice < 0) return;
  const int r0 = 4 * g;
  if (r0 >= a.nb) return;                    // this group has no rows (nobody waits for it)
  const int Tp = a.Tp, C = a.C, K = a.K, B = a.B, L = a.L, nb = a.nb;
  const int TpP = (Tp + 3) & ~3;
  const int taps = 2 * K + 1, taps4 = (taps + 3) & ~3;
  const bool drop = a.xmask != nullptr;
  const float* Xin = drop ? a.Xd : a.X;
  float* xg = a.xch + (int64_t)g * DX_GROUP;
  bool aborted = false;

  // ---------------------------------------------------------------- per-role constants
  // attention row / part of this CU
  const int ar = slice >> 3, aq = slice & 7;
  const int ab = r0 + ar;
  const bool ab_ok = ab < nb;
  const int abc = ab_ok ? ab : r0;
  // cell weights: lane owns gate-interleaved row 4*DU*slice + lane, wave owns K range [wave*KXW, +KXW)
  float wreg[KXW];
  {
    const int wrow = lane < 4 * DU ? lane : 0;
    const float* wr = a.wcat + (int64_t)(4 * DU * slice + wrow) * KX + wave * KXW;
#pragma unroll
    for (int k4 = 0; k4 < KXW / 4; ++k4) {
      const float4 v = *reinterpret_cast<const float4*>(wr + 4 * k4);
      wreg[4 * k4] = v.x; wreg[4 * k4 + 1] = v.y; wreg[4 * k4 + 2] = v.z; wreg[4 * k4 + 3] = v.w;
    }
  }
  // W_dec slice: MFMA block = 4*ks + ag; lane 4*blk+i holds column AU*slice + 4*ag + i, k = wave*DKW + ks*DKQ + q
  float wdreg[DKQ];
  {
    const int blk = lane >> 2, ks = blk >> 2, ag = blk & 3, al = 4 * ag + (lane & 3);
    const bool ok = al < AU;
    const float* wr = a.wdec + (int64_t)(AU * slice + (ok ? al : 0)) * DD + wave * DKW + ks * DKQ;
#pragma unroll
    for (int q = 0; q < DKQ; ++q) wdreg[q] = ok ? wr[q] : 0.f;
  }
  // pointwise threads of the cell: tid < 4*DU -> (unit tid>>2, row tid&3)
  const bool pw_thread = tid < 4 * DU;
  const int punit = DU * slice + (pw_thread ? (tid >> 2) : 0);
  const int pb = r0 + (tid & 3);
  const bool pb_ok = pw_thread && pb < nb;
  float4 pbias = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pw_thread) pbias = *reinterpret_cast<const float4*>(a.bcat + punit * 4);
  float c_prev = 0.f;
  // score lanes: the contraction over conv channels runs on the 16x16x4 MFMA: tile = 16 (row, frame) pairs
  // (frames 4*tile .. 4*tile+3 x 4 rows) x this CU's 16 attention columns; wave w owns tiles w, w+8, w+16, w+24.
  // D layout: lane (q = lane>>4, col = lane&15) holds pairs (row i, frame 4*tile + q), i = 0..3, of column col.
  const int a_l = lane & 15, sq = lane >> 4;
  const bool sc_ok = a_l < AU;
  const int acol = AU * slice + (sc_ok ? a_l : 0);
  const float gv = sc_ok ? a.gvec[acol] : 0.f;
  float ub[4];                               // B operand: U[col][c = 4*kk + q]
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int c = 4 * kk + sq;
    ub[kk] = (c < C && sc_ok) ? a.watt[(int64_t)acol * C + (c < C ? c : 0)] : 0.f;
  }
  // context threads: o_l = tid & 63, frames wave, wave+8, ...
  const int o_l = tid & 63;
  const bool ctx_thread = tid < OQ;
  const float bo_v = ctx_thread ? a.bo[OQ * aq + tid] : 0.f;

  // ---------------------------------------------------------------- LDS images
  for (int i = tid; i < 4 * XS; i += DP_NT) xs[i] = 0.f;                 // z_{-1} = 0, ctx_{-1} = 0
  for (int i = tid; i < 4 * 16 * DP_TPM; i += DP_NT) fs[i] = 0.f;        // channels >= C stay zero
  for (int i = tid; i < Tp * OQ; i += DP_NT) {
    const int t = i / OQ, o = i - t * OQ;
    Qs[i] = a.Q[((int64_t)abc * Tp + t) * OO + OQ * aq + o];
  }
  for (int i = tid; i < 16 * DP_TAPS4; i += DP_NT) {
    const int ch = i / DP_TAPS4, j = i - ch * DP_TAPS4;
    Fs[i] = (ch < C && j < taps) ? a.convw[ch * taps + j] : 0.f;
  }
  for (int i = tid; i < DP_WLEN; i += DP_NT) {
    const int fr = i - K;
    wp[i] = (fr >= 0 && fr < Tp) ? a.w0[(int64_t)abc * Tp + fr] : 0.f;
  }
  // P slice (constant over the sequence) in the layout the score lanes read: tile = wave + 8*it, row i, lane
#pragma unroll
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tile = wave + 8 * it, t = 4 * tile + sq, b = r0 + i;
      Ps[(tile * 4 + i) * 64 + lane] = a.P[((int64_t)(b < nb ? b : r0) * Tp + (t < Tp ? t : Tp - 1)) * AA + acol];
    }
  // embedding part of the first step's operand (already masked in Xd)
  float emb_next[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int id = tid + DP_NT * i, er = id / EE, ee = id - er * EE;
    const int eb = r0 + er;
    emb_next[i] = (er < 4) ? Xin[((int64_t)0 * B + (eb < nb ? eb : r0)) * KX + DD + OO + ee] : 0.f;
  }
  float mask_next = 1.f;                      // dropout mask of ctx_s as consumed by step s+1's cell
  if (drop && ctx_thread && L > 1) mask_next = a.xmask[((int64_t)1 * B + abc) * (OO + EE) + OQ * aq + tid];
  __syncthreads();

  for (int s = 0; s < L; ++s) {
    // Per-thread indices are re-derived from an opaque copy of the thread id every step: otherwise the compiler hoists
    // ~60 loop-invariant addresses/predicates out of the loop and, with 144 VGPRs pinned by the weights, spills them.
    int zv;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zv));
    const int tid_ = tid + zv, lane_ = lane + zv;
    const bool pw_thread_ = tid_ < 4 * DU;
    const int punit_ = DU * slice + (pw_thread_ ? (tid_ >> 2) : 0);
    const int pb_ = r0 + (tid_ & 3);
    const bool pb_ok_ = pw_thread_ && pb_ < nb;
    const int a_l_ = lane_ & 15, sq_ = lane_ >> 4;
    const bool sc_ok_ = a_l_ < AU;
    const int acol_ = AU * slice + (sc_ok_ ? a_l_ : 0);
    const int o_l_ = tid_ & 63;
    const bool ctx_thread_ = tid_ < OQ;
    const unsigned bit = tag_bit_of_step(s);
    const int par = s & 1;
    // ------------------------------------------------------------ (1) cell operand: ctx_{s-1} (exchange) + emb_s
    if (s > 0) {
      const float* cx = xg + DX_C + ((s - 1) & 1) * 4 * 512;
      const u64* p[NC];
      u64 v[NC];
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int id = tid_ + DP_NT * i;                 // pair id over [4][OO/2]
        const int row = (2 * id) / OO, o = 2 * id - row * OO;
        p[i] = reinterpret_cast<const u64*>(cx + ((2 * id < 4 * OO) ? row * 512 + o : 0));
      }
      poll_pairs<NC, ASR_DP_FULL>(p, tag_bit_of_step(s - 1), v, a.ctrl, aborted, 11u);
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int id = tid_ + DP_NT * i;
        const int row = (2 * id) / OO, o = 2 * id - row * OO;
        if (2 * id < 4 * OO) { xs[row * XS + DD + o] = pair_lo(v[i]); xs[row * XS + DD + o + 1] = pair_hi(v[i]); }
      }
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int id = tid_ + DP_NT * i, er = id / EE, ee = id - er * EE;
      if (er < 4) xs[er * XS + DD + OO + ee] = emb_next[i];
    }
    if (s + 1 < L) {
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int id = tid_ + DP_NT * i, er = id / EE, ee = id - er * EE;
        const int eb = r0 + er;
        emb_next[i] = (er < 4) ? Xin[((int64_t)(s + 1) * B + (eb < nb ? eb : r0)) * KX + DD + OO + ee] : 0.f;
      }
    }
    const float mask_cur = mask_next;
    if (drop && ctx_thread_ && s + 2 < L)
      mask_next = a.xmask[((int64_t)(s + 2) * B + abc) * (OO + EE) + OQ * aq + tid_];
    __syncthreads();
    // ------------------------------------------------------------ (2) gates = x Wcat^T on the 4x4x1 MFMA
    {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* xr = xs + (lane_ & 3) * XS + wave * KXW;
#pragma unroll
      for (int k4 = 0; k4 < ((ASR_DP_ABL & 2) ? 1 : KXW / 4); ++k4) {
        const float4 b = *reinterpret_cast<const float4*>(xr + 4 * k4);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * k4], b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * k4 + 1], b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * k4 + 2], b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * k4 + 3], b.w, acc, 0, 0, 0);
        if ((k4 & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // keep at most 4 operand reads in flight (VGPRs)
      }
      float* pp = part + (wave * 64 + lane_) * 5;
#pragma unroll
      for (int i = 0; i < 4; ++i) pp[i] = acc[i];
    }
    __syncthreads();
    // ------------------------------------------------------------ (3) pointwise LSTM update, publish z_s
    if (pw_thread_) {
      float pre[4] = {pbias.x, pbias.y, pbias.z, pbias.w};
#pragma unroll
      for (int w2 = 0; w2 < 8; ++w2)
#pragma unroll
        for (int i = 0; i < 4; ++i) pre[i] += part[(w2 * 64 + tid_) * 5 + i];
      const float gi = asr_sigmoid(pre[0]), gf = asr_sigmoid(pre[1]);
      const float gg = tanhf(pre[2]), go = asr_sigmoid(pre[3]);
      const float cn = gf * c_prev + gi * gg;
      float zn = go * tanhf(cn);
      if (aborted || flag_load(a.ctrl + 8) != 0u) zn = __builtin_nanf("");
      c_prev = cn;
      if (pb_ok_ && !(ASR_DP_ABL & 32)) {
        *reinterpret_cast<float4*>(a.gates + ((int64_t)s * B + pb_) * 4 * DD + punit_ * 4) = make_float4(gi, gf, gg, go);
        a.cstate[((int64_t)s * B + pb_) * DD + punit_] = cn;
        a.X[((int64_t)(s + 1) * B + pb_) * KX + punit_] = zn;
        if (drop) a.Xd[((int64_t)(s + 1) * B + pb_) * KX + punit_] = zn;
      }
      word_store(xg + DX_Z + par * 4 * 512 + (tid_ & 3) * 512 + punit_, zn, bit);
    }
    // ------------------------------------------------------------ (3b) location conv of w_{s-1} -> f_s (16 frames)
    {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int r = lane_ & 15, q = lane_ >> 4;
      const float* ap = wp + aq * DP_FPC + r + q;
      const float* bp = Fs + r * DP_TAPS4 + q;
      for (int j = 4 * wave; j < ((ASR_DP_ABL & 8) ? 4 : taps4); j += 32) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[j], bp[j], acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) cred[(wave * 16 + 4 * q + i) * 17 + r] = acc[i];
    }
    __syncthreads();
    if (tid_ < 256) {
      const int tl = tid_ >> 4, ch = tid_ & 15;
      float v = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < 8; ++w2) v += cred[(w2 * 16 + tl) * 17 + ch];
      const int t = aq * DP_FPC + tl;
      if (ch < C && t < TpP) {
        if (t >= Tp) v = 0.f;
        if (ab_ok && t < Tp) a.fconv[(((int64_t)s * B + ab) * C + ch) * Tp + t] = v;
        word_store(xg + DX_F + ((par * 4 + ar) * 16 + ch) * DP_TPM + t, v, bit);
      }
    }
    // ------------------------------------------------------------ (4)+(5) z_s and f_s (exchange) -> W_dec z_s for AU columns
    {
      // one poll for both: z_s pairs over [4][DD/2] and the conv features f_s (published ~2 us ago) as pairs
      // id = tid + 512 i over [4 rows][C][TpP/2]; small-integer divisions via exact float reciprocals
      const float* zx = xg + DX_Z + par * 4 * 512;
      const float* fx = xg + DX_F + par * 4 * 16 * DP_TPM;
      const int hp = TpP >> 1;
      const float rhp = 1.0f / (float)hp, rC = 1.0f / (float)C;
      int foff[4];
      const u64* p[NZ + 4];
      u64 v[NZ + 4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int id = tid_ + DP_NT * i;
        const int rc = (int)(((float)id + 0.5f) * rhp), t2 = id - rc * hp;
        const int row = (int)(((float)rc + 0.5f) * rC), c = rc - row * C;
        foff[i] = rc < 4 * C ? (row * 16 + c) * DP_TPM + 2 * t2 : -1;
        p[i] = reinterpret_cast<const u64*>(fx + (foff[i] < 0 ? 0 : foff[i]));
      }
#pragma unroll
      for (int i = 0; i < NZ; ++i) {
        const int id = tid_ + DP_NT * i;
        const int row = (2 * id) / DD, d = 2 * id - row * DD;
        p[4 + i] = reinterpret_cast<const u64*>(zx + ((2 * id < 4 * DD) ? row * 512 + d : 0));
      }
      poll_pairs<NZ + 4, ASR_DP_FULL>(p, bit, v, a.ctrl, aborted, 12u);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (foff[i] >= 0) { fs[foff[i]] = pair_lo(v[i]); fs[foff[i] + 1] = pair_hi(v[i]); }
#pragma unroll
      for (int i = 0; i < NZ; ++i) {
        const int id = tid_ + DP_NT * i;
        const int row = (2 * id) / DD, d = 2 * id - row * DD;
        if (2 * id < 4 * DD) { xs[row * XS + d] = pair_lo(v[4 + i]); xs[row * XS + d + 1] = pair_hi(v[4 + i]); }
      }
    }
    __syncthreads();
    {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* zr = xs + (lane_ & 3) * XS + wave * DKW + (lane_ >> 4) * DKQ;
#pragma unroll
      for (int q = 0; q < DKQ; ++q) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wdreg[q], zr[q], acc, 0, 0, 0);
      float* pp = part + (wave * 64 + lane_) * 5;
#pragma unroll
      for (int i = 0; i < 4; ++i) pp[i] = acc[i];
    }
    __syncthreads();
    if (tid_ < 64) {
      // (column al = tid_>>2, row = tid_&3): lanes 4*(4*ks + ag) + row, register al&3, all ks, all waves
      const int al = tid_ >> 2, row = tid_ & 3, ag = al >> 2, ii = al & 3;
      float v = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < 8; ++w2)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) v += part[(w2 * 64 + 4 * (4 * ks + ag) + row) * 5 + ii];
      dps[row * 16 + al] = v;
    }
    __syncthreads();
    // ------------------------------------------------------------ (6) energies: partial sums over this CU's columns
    {
      const int m = lane_ & 15;                       // A operand: pair m of the tile = (row m&3, frame 4*tile + (m>>2))
      const int nkk = (C + 3) >> 2;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int tile = wave + 8 * it;
        if (4 * tile < TpP && !(ASR_DP_ABL & 4)) {                        // wave-uniform
          f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
          const float* fa = fs + ((m & 3) * 16 + sq_) * DP_TPM + 4 * tile + (m >> 2);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            if (kk < nkk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[4 * kk * DP_TPM], ub[kk], acc, 0, 0, 0);
          const int t = 4 * tile + sq_;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float sv = dp_tanh(acc[i] + Ps[(tile * 4 + i) * 64 + lane_] + dps[i * 16 + a_l_]);
            const int b = r0 + i;
            if (sc_ok_ && b < nb && t < Tp && !(ASR_DP_ABL & 16)) a.S[(((int64_t)s * B + b) * Tp + t) * AA + acol_] = sv;
            float pe = sc_ok_ ? gv * sv : 0.f;
            pe += __shfl_xor(pe, 1, 64);
            pe += __shfl_xor(pe, 2, 64);
            pe += __shfl_xor(pe, 4, 64);
            pe += __shfl_xor(pe, 8, 64);
            if (a_l_ == 0) word_store(xg + DX_E + ((par * 32 + slice) * 4 + i) * DP_TPM + t, t < Tp ? pe : 0.f, bit);
          }
        }
      }
    }
    // ------------------------------------------------------------ (7) full energies of this CU's row -> softmax
    {
      const float* ex = xg + DX_E + par * 32 * 4 * DP_TPM + ar * DP_TPM;
      const int t2 = lane_;                           // pair (2*t2, 2*t2+1); producers 4*wave .. 4*wave+3
      const bool ok = 2 * t2 < TpP;
      const u64* p[4];
      u64 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        p[i] = reinterpret_cast<const u64*>(ex + (4 * wave + i) * 4 * DP_TPM + (ok ? 2 * t2 : 0));
      poll_pairs<4, ASR_DP_FULL>(p, bit, v, a.ctrl, aborted, 14u);
      float e0 = 0.f, e1 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { e0 += pair_lo(v[i]); e1 += pair_hi(v[i]); }
      if (ok) { epart[wave * DP_TPM + 2 * t2] = e0; epart[wave * DP_TPM + 2 * t2 + 1] = e1; }
    }
    __syncthreads();
    {
      // every wave runs the (tiny) softmax redundantly; wave 0 keeps the results
      float ev[2], wv[2];
      float mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int t = lane_ + 64 * k;
        float e = 0.f;
        if (t < Tp) {
#pragma unroll
          for (int w2 = 0; w2 < 8; ++w2) e += epart[w2 * DP_TPM + t];
        }
        ev[k] = e;
        if (t < Tp) mx = fmaxf(mx, a.scaling * e);
      }
      mx = wave_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int t = lane_ + 64 * k;
        wv[k] = t < Tp ? expf(a.scaling * ev[k] - mx) : 0.f;
        sum += wv[k];
      }
      sum = wave_sum(sum);
      const float inv = 1.0f / sum;
      if (wave == 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int t = lane_ + 64 * k;
          if (t < Tp) {
            const float w = wv[k] * inv;
            wsm[t] = w;
            wp[K + t] = w;                           // operand of the next step's location conv
            if (aq == 0 && ab_ok) {
              a.ws[((int64_t)s * B + ab) * Tp + t] = w;
              a.energy[((int64_t)s * B + ab) * Tp + t] = ev[k];
            }
          }
        }
      }
    }
    __syncthreads();
    // ------------------------------------------------------------ (8) context slice, publish (masked) for the cell
    {
      float acc = 0.f;
      if (o_l_ < OQ)
        for (int t = wave; t < Tp; t += 8) acc += wsm[t] * Qs[t * OQ + o_l_];
      cpart[wave * 64 + o_l_] = acc;
    }
    __syncthreads();
    if (ctx_thread_) {
      float v = bo_v;
#pragma unroll
      for (int w2 = 0; w2 < 8; ++w2) v += cpart[w2 * 64 + tid_];
      const int o = OQ * aq + tid_;
      const float vm = drop ? v * mask_cur : v;
      if (ab_ok) {
        a.X[((int64_t)(s + 1) * B + ab) * KX + DD + o] = v;
        if (drop && s + 1 < L) a.Xd[((int64_t)(s + 1) * B + ab) * KX + DD + o] = vm;
      }
      word_store(xg + DX_C + par * 4 * 512 + ar * 512 + o, vm, bit);
    }
  }
}

template <int DD, int AA, int OO, int EE>
int launch_dec_fwd(const DecPersistArgs& a, hipStream_t stream) {
  using DM = DecDims<DD, AA, OO, EE>;
  const size_t lds = DM::lds_floats * sizeof(float);     // > 80 KB: one workgroup per CU
  static_assert(DM::lds_floats * sizeof(float) > 82 * 1024 && DM::lds_floats * sizeof(float) <= 160 * 1024, "LDS budget");
  hipError_t e = hipFuncSetAttribute((const void*)dec_persist_fwd_kernel<DD, AA, OO, EE>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((dec_persist_fwd_kernel<DD, AA, OO, EE>), dim3(256), dim3(DP_NT), lds, stream, a);
  return 0;
}

}  // namespace

bool asr_persist_device_ok();

// Whole teacher-forced decoder sequence (steps 0..L-1) in one launch per block of 32 rows.  Same operands and
// results as asr_dec_seq_fwd(p, 0, L) except that Dproj is not written.  Returns ASR_E_SHAPE when the fast path
// does not apply (the caller then uses asr_dec_seq_fwd).  xch >= 2 MB, ctrl >= 64 B (zeroed here on the stream).
extern "C" int asr_dec_seq_fwd_persist(const asr_dec_fwd_t* p, void* xch, void* ctrl, asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!p || !xch || !ctrl || !p->P || !p->Q || !p->bo || !p->wcat || !p->bcat || !p->wdec || !p->convw || !p->watt ||
      !p->gvec || !p->w0 || !p->X || !p->gates || !p->cstate || !p->fconv || !p->S || !p->energy || !p->ws)
    return ASR_E_ARG;
  if (p->B <= 0 || p->nb <= 0 || p->nb > p->B || p->Tp <= 0 || p->L <= 0) return ASR_E_ARG;
  if (p->xmask && !p->Xd) return ASR_E_ARG;
  const bool cfg2 = p->D == 512 && p->A == 512 && p->O == 512 && p->E == 128;
  const bool cfg1 = p->D == 320 && p->A == 320 && p->O == 320 && p->E == 128;
  if (!cfg1 && !cfg2) return ASR_E_SHAPE;
  const int TpP = (p->Tp + 3) & ~3;
  if (p->Tp > DP_TPM || p->C <= 0 || p->C > 16 || p->K < 0 || p->K > DP_KMAX || 2 * p->C * TpP > 4 * DP_NT ||
      p->nb > 128)
    return ASR_E_SHAPE;
  if (!asr_persist_device_ok()) return ASR_E_SHAPE;
  const int B = p->B, Tp = p->Tp, A = p->A, D = p->D, O = p->O, E = p->E, C = p->C, KX = D + O + E;
  for (int rb = 0; rb < p->nb; rb += 32) {
    hipError_t e = hipMemsetAsync(ctrl, 0, 16 * sizeof(unsigned), stream);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(xch, 0, (size_t)8 * DX_GROUP * sizeof(float), stream);
    if (e != hipSuccess) return (int)e;
    DecPersistArgs a;
    a.B = B; a.nb = p->nb - rb < 32 ? p->nb - rb : 32; a.Tp = Tp; a.C = C; a.K = p->K; a.L = p->L;
    a.scaling = p->scaling;
    a.P = p->P + (int64_t)rb * Tp * A; a.Q = p->Q + (int64_t)rb * Tp * O; a.bo = p->bo; a.wcat = p->wcat;
    a.bcat = p->bcat; a.wdec = p->wdec; a.convw = p->convw; a.watt = p->watt; a.gvec = p->gvec;
    a.w0 = p->w0 + (int64_t)rb * Tp; a.xmask = p->xmask ? p->xmask + (int64_t)rb * (O + E) : nullptr;
    a.X = p->X + (int64_t)rb * KX; a.Xd = p->Xd ? p->Xd + (int64_t)rb * KX : nullptr;
    a.gates = p->gates + (int64_t)rb * 4 * D; a.cstate = p->cstate + (int64_t)rb * D;
    a.fconv = p->fconv + (int64_t)rb * C * Tp; a.S = p->S + (int64_t)rb * Tp * A;
    a.energy = p->energy + (int64_t)rb * Tp; a.ws = p->ws + (int64_t)rb * Tp;
    a.xch = (float*)xch; a.ctrl = (unsigned*)ctrl;
    const int rc = cfg2 ? launch_dec_fwd<512, 512, 512, 128>(a, stream) : launch_dec_fwd<320, 320, 320, 128>(a, stream);
    if (rc) return rc;
  }
  ASR_CHECK_LAUNCH();
  return 0;
}
