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
// measurement only: per-phase shader-clock stamps of workgroup (group 0, slice 0), steps 8..15, into ctrl[16..]
#if defined(ASR_DP_TRACE2)
// every slice of group 0, steps 8..15, on the chip-wide 100 MHz clock (comparable across CUs): [slice][step][mark] in a
// 32 KB trace buffer behind the control words (tools/dec_trace2.py: who waits for whom)
#define DP_MARK(k) do { if (tid == 0 && g == 0 && TRS >= 8 && TRS < 16) \
    ((unsigned long long*)(a.ctrl + 16))[(slice * 8 + (TRS - 8)) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#elif defined(ASR_DP_TRACE)
#define DP_MARK(k) do { if (tid == 0 && g == 0 && slice == 0 && TRS >= 8 && TRS < 16) \
    ((unsigned long long*)(a.ctrl + 16))[(TRS - 8) * 16 + (k)] = clock64(); } while (0)
#else
#define DP_MARK(k) do {} while (0)
#endif

#ifndef ASR_DP_ABL
#define ASR_DP_ABL 0
#endif
#ifndef ASR_DP_FULL
#define ASR_DP_FULL false
#endif

namespace {

constexpr int DP_NT = 512;          // 8 waves
constexpr int DP_TPM = 128;         // max encoder frames T' of the 4-rows-per-group geometry (and of the backward kernel)
constexpr int DP_FPC = 16;          // conv frames per CU (8 CUs per row)
constexpr int DP_KMAX = 100;        // max conv half width
constexpr int DP_TAPS4 = 208;       // LDS row length of the filter image (>= 2*KMAX+1 rounded to 4)
// Geometry of a group (= XCD).  RG utterances per group, each served by PPR = 32 / RG CUs ("parts"): a part owns 16 conv
// frames, O / PPR context outputs and (like every CU) D/32 cell units and A/32 attention columns of all RG rows.
//   RG = 4, TPM = 128: a batch of 32 fills the chip (T' <= 102 with 10 conv channels: cfg-2, cfg-1);
//   RG = 2, TPM = 256: 16 CUs per utterance, so the T'-sized LDS images (P slice, Q slice, conv features) keep their
//                      size at twice the frames (T' <= 256: cfg-5's T' = 200); 16 utterances per launch.
// The row axis of the exchange and LDS layouts keeps 4 slots in both geometries (rows >= RG are never published to
// other CUs and never read from them).
template <int RG_, int TPM_>
struct DecGeo {
  static constexpr int RG = RG_, TPM = TPM_;
  static constexpr int PPR = 32 / RG;                 // parts (CUs) per row
  static constexpr int LR = RG == 4 ? 3 : 4;          // log2(PPR)
  static constexpr int FPT = 16 / RG;                 // frames per score tile (16 (row, frame) pairs)
  static constexpr int NFQ = RG == 4 ? 2 : 3;         // conv-feature quads per thread in the z / f poll
  static constexpr int FROWS = RG == 4 ? 16 : 12;     // rows of the conv filter image (channels, zero padded)
  static constexpr int WLEN = TPM + 2 * DP_KMAX + 8;
  // exchange layout per group, in floats
  static constexpr int X_Z = 0;                            // [2][4][512]
  static constexpr int X_C = X_Z + 2 * 4 * 512;            // [2][4][512]
  static constexpr int X_F = X_C + 2 * 4 * 512;            // [2][4][16][TPM]
  static constexpr int X_E = X_F + 2 * 4 * 16 * TPM;       // [2][32][4][TPM]
  static constexpr int X_U = X_E + 2 * 32 * 4 * TPM;       // [2][4][512]  free-running only: ctx before the dropout mask
  static constexpr int X_M = X_U + 2 * 4 * 512;            // [2][4][128]  free-running only: embedding input of the step
  static constexpr int X_S = X_M + 2 * 4 * 128;            // [2][2]       free-running only: all rows of the group at <EOS>
  static constexpr int X_GROUP = X_S + 2 * 2 + 4;
  static_assert(PPR * DP_FPC == TPM && 32 * FPT == TPM, "16 conv frames per part, 32 score tiles");
};

struct DecPersistArgs {
  int B, nb, Tp, C, K, L;
  float scaling;
  const float *P, *Q, *bo, *wcat, *bcat, *wdec, *convw, *watt, *gvec, *w0, *xmask;
  float *X, *Xd, *gates, *cstate, *fconv, *S, *energy, *ws;
  float* xch;
  unsigned* ctrl;
  // free-running feedback (kernel template FB): 1 = embedding of the predicted token, 2 = smooth embedding
  int fb_mode, V, eos;     // eos >= 0: a group stops once all of its rows have emitted eos (decoding without autograd)
  float fb_scale;
  const float *w_out, *b_out, *emb;
  float *logits, *probs;
  long long *pred, *fed;
  // scheduled sampling (fb_mode 1): step s >= 1 is fed tok[b][s] where tf[s] != 0, its own prediction elsewhere
  const long long* tok;
  long long ldtok;
  const unsigned char* tf;
};

__device__ __forceinline__ float dp_tanh(float x) {   // same formula as decoder.hip:fast_tanh
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
}

template <int DD, int AA, int OO, int EE, int RG = 4, int TPM = DP_TPM>
struct DecDims {
  using GEO = DecGeo<RG, TPM>;
  static constexpr int KX = DD + OO + EE;
  static constexpr int KXW = KX / 8;       // K columns of the cell product per wave
  static constexpr int DU = DD / 32;       // cell units per CU
  static constexpr int AU = AA / 32;       // attention columns per CU
  static constexpr int OQ = OO / GEO::PPR; // context outputs per CU
  static constexpr int DKW = DD / 8;       // W_dec product: k's per wave
  static constexpr int DKQ = DD / 32;      // ... per (wave, k-sub)
  static constexpr int XS = KX + 4;        // padded LDS row of the cell operand
  static constexpr int NZ = (2 * DD + DP_NT - 1) / DP_NT;     // 8-byte pairs per thread when gathering z
  static constexpr int NC = (2 * OO + DP_NT - 1) / DP_NT;
  static constexpr int NE = (4 * EE + DP_NT - 1) / DP_NT;     // embedding values per thread
  static constexpr size_t lds_floats = 4 * XS + 8 * 64 * 5 + 64 + RG * 16 * TPM + TPM * OQ + GEO::FROWS * DP_TAPS4 +
                                       GEO::WLEN + 8 * 16 * 17 + 8 * TPM + TPM + 8 * 64 + 32 * 4 * 64 + 8;
  static_assert(KX % 32 == 0 && DD % 32 == 0 && AA % 32 == 0 && OO % 8 == 0 && (EE & (EE - 1)) == 0, "slice sizes");
  static_assert(DU <= 16 && AU <= 16 && OQ <= 64 && DD <= 512 && OO <= 512, "per-CU slices must fit the mappings");
};

// FB = free-running decode (model.py:334-341): the embedding input of step s > 0 is not read from X but made from the
// logits of step s-1 inside the kernel.  Slice 0 of every group does that for the group's 4 rows at the top of the
// step (it holds z_{s-1} and receives ctx_{s-1} like every CU, plus the unmasked ctx when dropout is on): logits =
// W_out [z, ctx] + b (W_out streamed from L2, one wave per output), argmax / softmax(scale * logits), embedding row or
// p @ E, dropout mask, -> X/Xd, logits, pred, fed, probs in global memory and one more exchange (DX_M) from which all
// 32 CUs take the 4 x 128 embedding values.  The logits of the last step are left to the caller.
template <int DD, int AA, int OO, int EE, bool FB, int RG = 4, int TPM = DP_TPM, bool FAULT = false>
__global__ __launch_bounds__(DP_NT) void dec_persist_fwd_kernel(DecPersistArgs a) {
  using DM = DecDims<DD, AA, OO, EE, RG, TPM>;
  using GEO = DecGeo<RG, TPM>;
  constexpr int PPR = GEO::PPR, LR = GEO::LR, FPT = GEO::FPT, NFQ = GEO::NFQ, FROWS = GEO::FROWS, DP_WLEN = GEO::WLEN;
  constexpr int DX_Z = GEO::X_Z, DX_C = GEO::X_C, DX_F = GEO::X_F, DX_E = GEO::X_E, DX_U = GEO::X_U, DX_M = GEO::X_M;
  constexpr int DX_S = GEO::X_S, DX_GROUP = GEO::X_GROUP;
  // free-running feedback: written for the 4 row slots of the exchange / LDS layouts; with RG = 2 (T' <= 256) rows 2, 3 of
  // every slot are dead weight: never polled (nobody publishes them), never stored to memory, counted as finished
  constexpr int KX = DM::KX, KXW = DM::KXW, DU = DM::DU, AU = DM::AU, OQ = DM::OQ, DKW = DM::DKW, DKQ = DM::DKQ;
  constexpr int XS = DM::XS, NZ = DM::NZ, NC = DM::NC, NE = DM::NE;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* xs = sm;                            // [4][XS]      cell operand [z | ctx(masked) | emb(masked)] per row
  float* part = xs + 4 * XS;                 // [8][64][5]   K-partials (cell product, then W_dec product)
  float* dps = part + 8 * 64 * 5;            // [4][16]      W_dec z slice
  float* fs = dps + 64;                      // [RG][16][TPM] conv features of all rows
  float* Qs = fs + RG * 16 * TPM;            // [TPM][OQ]    Q slice of this CU's row
  float* Fs = Qs + TPM * OQ;                 // [FROWS][TAPS4] conv filters, zero padded
  float* wp = Fs + FROWS * DP_TAPS4;         // [WLEN]       previous attention weights of this CU's row, zero halo
  float* cred = wp + DP_WLEN;                // [8][16][17]  conv partial tiles
  float* epart = cred + 8 * 16 * 17;         // [8][TPM]     partial energy sums
  float* wsm = epart + 8 * TPM;              // [TPM]        attention weights of this step
  float* cpart = wsm + TPM;                  // [8][64]      context partials
  float* Ps = cpart + 8 * 64;                // [32 tiles][4 rows][64 lanes]  P slice in the score-lane layout
  int* role = reinterpret_cast<int*>(Ps + 32 * 4 * 64);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int g, slice;
  take_role(a.ctrl, role, g, slice);
  if (slice < 0) return;
  constexpr unsigned spin_limit = FAULT ? DEBUG_SPIN_LIMIT : SPIN_LIMIT;     // (FAULT: persist.h - tests of the abort path)
  const int r0 = RG * g;
  if (r0 >= a.nb) return;                    // this group has no rows (nobody waits for it)
  const int Tp = a.Tp, C = a.C, K = a.K, B = a.B, L = a.L, nb = a.nb;
  const int TpP = (Tp + 3) & ~3;
  const int taps = 2 * K + 1, taps4 = (taps + 3) & ~3;
  const bool drop = a.xmask != nullptr;
  const float* Xin = drop ? a.Xd : a.X;
  float* xg = a.xch + (int64_t)g * DX_GROUP;
  const __amdgpu_buffer_rsrc_t xrs = make_xch_rsrc(a.xch);
  bool aborted = false;

  // ---------------------------------------------------------------- per-role constants
  // attention row / part of this CU
  const int ar = slice >> LR, aq = slice & (PPR - 1);
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
  const bool pb_ok = pw_thread && (tid & 3) < RG && pb < nb;
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
  for (int i = tid; i < RG * 16 * TPM; i += DP_NT) fs[i] = 0.f;          // channels >= C stay zero
  for (int i = tid; i < Tp * OQ; i += DP_NT) {
    const int t = i / OQ, o = i - t * OQ;
    Qs[i] = a.Q[((int64_t)abc * Tp + t) * OO + OQ * aq + o];
  }
  for (int i = tid; i < FROWS * DP_TAPS4; i += DP_NT) {
    const int ch = i / DP_TAPS4, j = i - ch * DP_TAPS4;
    Fs[i] = (ch < C && j < taps) ? a.convw[ch * taps + j] : 0.f;
  }
  for (int i = tid; i < DP_WLEN; i += DP_NT) {
    const int fr = i - K;
    wp[i] = (fr >= 0 && fr < Tp) ? a.w0[(int64_t)abc * Tp + fr] : 0.f;
  }
  // P slice (constant over the sequence) in the layout the score lanes read: tile = wave + 8*it, row i, lane
  // (pair 4 sq + i of a tile = row (4 sq + i) % RG, frame FPT tile + (4 sq + i) / RG: the D layout of the score MFMA)
#pragma unroll
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tile = wave + 8 * it, t = FPT * tile + (4 * sq + i) / RG, b = r0 + (4 * sq + i) % RG;
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
  unsigned rows_done = 0u;                    // FB, slice 0: bit i = row i of the group has emitted <EOS>
  float mask_next = 1.f;                      // dropout mask of ctx_s as consumed by step s+1's cell
  if (drop && ctx_thread && L > 1) mask_next = a.xmask[((int64_t)1 * B + abc) * (OO + EE) + OQ * aq + tid];
  __syncthreads();

  for (int s = 0; s < L; ++s) {
    // Per-thread indices are re-derived from an opaque copy of the thread id every step: otherwise the compiler hoists
    // ~60 loop-invariant addresses/predicates out of the loop and, with 144 VGPRs pinned by the weights, spills them.
    int zv;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zv));
    const int tid_ = tid + zv, lane_ = lane + zv;
    const unsigned abort_seen = tid < 64 ? flag_load(a.ctrl + 8) : 0u;   // sampled early, consumed by the pointwise phase
    const bool pw_thread_ = tid_ < 4 * DU;
    const int punit_ = DU * slice + (pw_thread_ ? (tid_ >> 2) : 0);
    const int pb_ = r0 + (tid_ & 3);
    const bool pb_ok_ = pw_thread_ && (tid_ & 3) < RG && pb_ < nb;
    const int a_l_ = lane_ & 15, sq_ = lane_ >> 4;
    const bool sc_ok_ = a_l_ < AU;
    const int acol_ = AU * slice + (sc_ok_ ? a_l_ : 0);
    const int o_l_ = tid_ & 63;
    const bool ctx_thread_ = tid_ < OQ;
    const unsigned bit = tag_bit_of_step(s);
    const int par = s & 1;
#define TRS s
    DP_MARK(0);
    // ------------------------------------------------------------ (1) cell operand: ctx_{s-1} (exchange) + emb_s
    if (s > 0) {
      // quads over [4][OO/4] (16-byte sc1 buffer loads: half the load instructions of the pair poll)
      constexpr int NQ = (OO + DP_NT - 1) / DP_NT;
      const unsigned cbase = (unsigned)((xg - a.xch) + DX_C + ((s - 1) & 1) * 4 * 512) * 4u;
      unsigned off[NQ];
      u4v v[NQ];
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int id = tid_ + DP_NT * i;                 // quad id over [4][OO/4]
        const int row = (4 * id) / OO, o = 4 * id - row * OO;
        off[i] = cbase + (unsigned)((4 * id < RG * OO) ? row * 512 + o : 0) * 4u;      // rows >= RG are never published
      }
      poll_quads<NQ, true>(xrs, off, tag_bit_of_step(s - 1), v, a.ctrl, aborted, 11u, spin_limit);
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int id = tid_ + DP_NT * i;
        const int row = (4 * id) / OO, o = 4 * id - row * OO;
        if (4 * id < RG * OO)
          *reinterpret_cast<float4*>(xs + row * XS + DD + o) = make_float4(__uint_as_float(v[i].x), __uint_as_float(v[i].y),
                                                                          __uint_as_float(v[i].z), __uint_as_float(v[i].w));
      }
    }
    if (FB && s > 0) {
      if (slice == 0) {
        float* cu = part;                      // [4][512] unmasked ctx_{s-1} (scratch: free until phase 2)
        float* lgs = cred;                     // [4][64] logits      (scratch: free until phase 3b)
        float* prs = cred + 256;               // [4][64] probabilities / the chosen token
        if (drop) {
          const float* ux = xg + DX_U + ((s - 1) & 1) * 4 * 512;
          const u64* p[NC];
          u64 v[NC];
#pragma unroll
          for (int i = 0; i < NC; ++i) {
            const int id = tid_ + DP_NT * i;
            const int row = (2 * id) / OO, o = 2 * id - row * OO;
            p[i] = reinterpret_cast<const u64*>(ux + ((2 * id < RG * OO) ? row * 512 + o : 0));
          }
          poll_pairs<NC, ASR_DP_FULL>(p, tag_bit_of_step(s - 1), v, a.ctrl, aborted, 16u, spin_limit);
#pragma unroll
          for (int i = 0; i < NC; ++i) {
            const int id = tid_ + DP_NT * i;
            const int row = (2 * id) / OO, o = 2 * id - row * OO;
            if (2 * id < RG * OO) { cu[row * 512 + o] = pair_lo(v[i]); cu[row * 512 + o + 1] = pair_hi(v[i]); }
          }
        }
        __syncthreads();
        const float* cb = drop ? cu : xs + DD;
        const int cst = drop ? 512 : XS;
        for (int v = wave; v < a.V; v += 8) {
          // one output per wave and trip: the lanes stride the D+O inputs in float4 (coalesced rows of W_out from L2)
          const float4* wr = reinterpret_cast<const float4*>(a.w_out + (int64_t)v * (DD + OO));
          constexpr int NJZ = (DD / 4 + 63) / 64, NJC = (OO / 4 + 63) / 64;
          float4 wz[NJZ], wc[NJC];
#pragma unroll
          for (int j = 0; j < NJZ; ++j) {
            const int q = lane_ + 64 * j;
            wz[j] = wr[q < DD / 4 ? q : 0];
          }
#pragma unroll
          for (int j = 0; j < NJC; ++j) {
            const int q = lane_ + 64 * j;
            wc[j] = wr[DD / 4 + (q < OO / 4 ? q : 0)];
          }
          float ac[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < NJZ; ++j) {
            const int q = lane_ + 64 * j;
            if (q < DD / 4) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float4 x = *reinterpret_cast<const float4*>(xs + i * XS + 4 * q);
                ac[i] += wz[j].x * x.x + wz[j].y * x.y + wz[j].z * x.z + wz[j].w * x.w;
              }
            }
          }
#pragma unroll
          for (int j = 0; j < NJC; ++j) {
            const int q = lane_ + 64 * j;
            if (q < OO / 4) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float4 x = *reinterpret_cast<const float4*>(cb + i * cst + 4 * q);
                ac[i] += wc[j].x * x.x + wc[j].y * x.y + wc[j].z * x.z + wc[j].w * x.w;
              }
            }
          }
          const float bv = a.b_out ? a.b_out[v] : 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float t = wave_sum_dpp(ac[i]);
            if (lane_ == 0) lgs[i * 64 + v] = t + bv;
          }
        }
        __syncthreads();
        if (wave < 4) {
          const int b = r0 + wave;
          const bool bok = wave < RG && b < nb, lv = lane_ < a.V;
          const float l = lv ? lgs[wave * 64 + lane_] : -INFINITY;
          const float mx = wave_max_dpp(l);
          const unsigned long long hit = __ballot(lv && l == mx);
          const int am = hit ? __ffsll(hit) - 1 : 0;          // lowest index among the maxima; all-NaN row: 0
          // scheduled sampling: the teacher's token where the host drew "teacher" for this step (model.py:328-333)
          int feed = am;
          if (a.tok != nullptr && a.tf[s] != 0) feed = (int)a.tok[(int64_t)(bok ? b : r0) * a.ldtok + s];
          if (lv && bok) a.logits[((int64_t)(s - 1) * B + b) * a.V + lane_] = l;
          if (lane_ == 0 && bok) {
            a.pred[(int64_t)(s - 1) * B + b] = am;
            a.fed[(int64_t)s * B + b] = a.fb_mode == 2 ? -1 : feed;
          }
          if (a.fb_mode == 2) {
            const float sl = lv ? a.fb_scale * l : -INFINITY;
            const float smx = wave_max_dpp(sl);
            const float e = lv ? expf(sl - smx) : 0.f;
            const float p = e / wave_sum_dpp(e);
            if (lv) {
              prs[wave * 64 + lane_] = p;
              if (bok) a.probs[((int64_t)(s - 1) * B + b) * a.V + lane_] = p;
            }
          } else if (lane_ == 0) {
            prs[wave * 64] = __int_as_float(feed);
          }
          if (lane_ == 0) cred[512 + wave] = __int_as_float((am == a.eos || !bok) ? 1 : 0);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) rows_done |= (unsigned)__float_as_int(cred[512 + i]) << i;
        if (tid_ == 0) {
          const float stop = (a.eos >= 0 && rows_done == 0xFu) ? 2.0f : 0.f;      // LSB carries the tag
          word_store(xg + DX_S + ((s - 1) & 1) * 2, stop, tag_bit_of_step(s - 1));
          word_store(xg + DX_S + ((s - 1) & 1) * 2 + 1, stop, tag_bit_of_step(s - 1));
        }
        {
          const int row = tid_ >> 7, e = tid_ & 127, b = r0 + row;
          const bool bok = row < RG && b < nb;
          const int bc = bok ? b : r0;
          float v = 0.f;
          if (a.fb_mode == 2) {
            for (int u = 0; u < a.V; ++u) v += prs[row * 64 + u] * a.emb[u * EE + e];
          } else {
            v = a.emb[__float_as_int(prs[row * 64]) * EE + e];
          }
          const float mk = drop ? a.xmask[((int64_t)s * B + bc) * (OO + EE) + OO + e] : 1.f;
          const float vm = v * mk;
          if (bok) {
            a.X[((int64_t)s * B + b) * KX + DD + OO + e] = v;
            if (drop) a.Xd[((int64_t)s * B + b) * KX + DD + OO + e] = vm;
          }
          word_store(xg + DX_M + ((s - 1) & 1) * 512 + row * 128 + e, vm, tag_bit_of_step(s - 1));
        }
      }
      {
        const int id = tid_ & 255;                       // 256 pairs over [4][128]; the upper half mirrors
        // (this exchange is first used at step 1: slot and tag follow s - 1, so that the zeroed buffer reads invalid)
        const u64* p1[2] = {reinterpret_cast<const u64*>(xg + DX_S + ((s - 1) & 1) * 2),
                            reinterpret_cast<const u64*>(xg + DX_M + ((s - 1) & 1) * 512 + 2 * id)};
        u64 v1[2];
        poll_pairs<2, true>(p1, tag_bit_of_step(s - 1), v1, a.ctrl, aborted, 17u, spin_limit);
        if (tid_ < 256) {
          const int row = id >> 6, e2 = 2 * (id & 63);
          xs[row * XS + DD + OO + e2] = pair_lo(v1[1]);
          xs[row * XS + DD + OO + e2 + 1] = pair_hi(v1[1]);
        }
        // every row of this group has emitted <EOS>: all 32 CUs read the same word and leave together (the caller
        // pre-fills the outputs of the steps that are not run)
        if (pair_lo(v1[0]) >= 1.0f && !aborted) break;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int id = tid_ + DP_NT * i, er = id / EE, ee = id - er * EE;
        if (er < 4) xs[er * XS + DD + OO + ee] = emb_next[i];
      }
    }
    {
      // next step's operands: UNGUARDED loads from clamped addresses (a load under a lane predicate or a uniform branch
      // is waited for at the join: a memory round trip on the serial chain of every step); invalid lanes discard
      const int sn = s + 1 < L ? s + 1 : s;
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int id = tid_ + DP_NT * i, er = (id / EE) & 3, ee = id & (EE - 1);
        const int eb = r0 + er;
        emb_next[i] = Xin[((int64_t)sn * B + (eb < nb ? eb : r0)) * KX + DD + OO + ee];
      }
    }
    const float mask_cur = mask_next;
    {
      const int sm = s + 2 < L ? s + 2 : (L > 1 ? 1 : 0);
      const float* mp = drop ? a.xmask + ((int64_t)sm * B + abc) * (OO + EE) + OQ * aq : a.bo + OQ * aq;
      mask_next = mp[tid_ < OQ ? tid_ : 0];                    // only the context threads (tid < OQ) use it
    }
    __syncthreads();
    DP_MARK(1);
    // ------------------------------------------------------------ (2) gates = x Wcat^T on the 4x4x1 MFMA
    {
      // two partial accumulators (even / odd k), summed once: a chain of dependent MFMAs on one accumulator waits
      // for the predecessor's passes every time
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, accb = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* xr = xs + (lane_ & 3) * XS + wave * KXW;
#pragma unroll
      for (int k4 = 0; k4 < ((ASR_DP_ABL & 2) ? 1 : KXW / 4); ++k4) {
        const float4 b = *reinterpret_cast<const float4*>(xr + 4 * k4);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * k4], b.x, acc, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * k4 + 1], b.y, accb, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * k4 + 2], b.z, acc, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * k4 + 3], b.w, accb, 0, 0, 0);
        if ((k4 & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // keep at most 4 operand reads in flight (VGPRs)
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += accb[i];
      float* pp = part + (wave * 64 + lane_) * 5;
#pragma unroll
      for (int i = 0; i < 4; ++i) pp[i] = acc[i];
    }
    __syncthreads();
    DP_MARK(2);
    // ------------------------------------------------------------ (3) pointwise LSTM update, publish z_s
    if (pw_thread_) {
      float pre[4] = {pbias.x, pbias.y, pbias.z, pbias.w};
#pragma unroll
      for (int w2 = 0; w2 < 8; ++w2)
#pragma unroll
        for (int i = 0; i < 4; ++i) pre[i] += part[(w2 * 64 + tid_) * 5 + i];
      const float gi = asr_fast_sigmoid(pre[0]), gf = asr_fast_sigmoid(pre[1]);
      const float gg = asr_fast_tanh(pre[2]), go = asr_fast_sigmoid(pre[3]);
      const float cn = gf * c_prev + gi * gg;
      float zn = go * asr_fast_tanh(cn);
      if (aborted || abort_seen != 0u) zn = __builtin_nanf("");
      c_prev = cn;
      if (pb_ok_ && !(ASR_DP_ABL & 32)) {
        *reinterpret_cast<float4*>(a.gates + ((int64_t)s * B + pb_) * 4 * DD + punit_ * 4) = make_float4(gi, gf, gg, go);
        a.cstate[((int64_t)s * B + pb_) * DD + punit_] = cn;
        a.X[((int64_t)(s + 1) * B + pb_) * KX + punit_] = zn;
        if (drop) a.Xd[((int64_t)(s + 1) * B + pb_) * KX + punit_] = zn;
      }
      if (!(FAULT && g == 0 && slice == 1 && s >= 1))              // (FAULT: a producer that went silent)
        word_store(xg + DX_Z + par * 4 * 512 + (tid_ & 3) * 512 + punit_, zn, bit);
    }
    DP_MARK(3);
    // ------------------------------------------------------------ (3b) location conv of w_{s-1} -> f_s (16 frames)
    {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int r = lane_ & 15, q = lane_ >> 4;
      const float* ap = wp + aq * DP_FPC + r + q;
      const float* bp = Fs + (r < FROWS ? r : FROWS - 1) * DP_TAPS4 + q;      // rows >= C are zero
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
        word_store(xg + DX_F + ((par * 4 + ar) * 16 + ch) * TPM + t, v, bit);
      }
    }
    DP_MARK(4);
    // ------------------------------------------------------------ (4)+(5) z_s and f_s (exchange) -> W_dec z_s for AU columns
    {
      // one poll for both, in 16-byte quads: z_s over [4][DD/4] and the conv features f_s (published ~2 us ago) over
      // [4 rows][C][TpP/4], id = tid + 512 i; small-integer divisions via exact float reciprocals
      constexpr int NZQ = (DD + DP_NT - 1) / DP_NT;
      const unsigned zbase = (unsigned)((xg - a.xch) + DX_Z + par * 4 * 512) * 4u;
      const unsigned fbase = (unsigned)((xg - a.xch) + DX_F + par * 4 * 16 * TPM) * 4u;
      const int hq = TpP >> 2;
      const float rhq = 1.0f / (float)hq, rC = 1.0f / (float)C;
      int foff[NFQ];
      unsigned off[NZQ + NFQ];
      u4v v[NZQ + NFQ];
#pragma unroll
      for (int i = 0; i < NFQ; ++i) {
        const int id = tid_ + DP_NT * i;
        const int rc = (int)(((float)id + 0.5f) * rhq), t4 = id - rc * hq;
        const int row = (int)(((float)rc + 0.5f) * rC), c = rc - row * C;
        foff[i] = rc < RG * C ? (row * 16 + c) * TPM + 4 * t4 : -1;
        off[i] = fbase + (unsigned)(foff[i] < 0 ? 0 : foff[i]) * 4u;
      }
#pragma unroll
      for (int i = 0; i < NZQ; ++i) {
        const int id = tid_ + DP_NT * i;
        const int row = (4 * id) / DD, d = 4 * id - row * DD;
        off[NFQ + i] = zbase + (unsigned)((4 * id < 4 * DD) ? row * 512 + d : 0) * 4u;
      }
      poll_quads<NZQ + NFQ, true>(xrs, off, bit, v, a.ctrl, aborted, 12u, spin_limit);
      DP_MARK(5);
#pragma unroll
      for (int i = 0; i < NFQ; ++i)
        if (foff[i] >= 0)
          *reinterpret_cast<float4*>(fs + foff[i]) = make_float4(__uint_as_float(v[i].x), __uint_as_float(v[i].y),
                                                                 __uint_as_float(v[i].z), __uint_as_float(v[i].w));
#pragma unroll
      for (int i = 0; i < NZQ; ++i) {
        const int id = tid_ + DP_NT * i;
        const int row = (4 * id) / DD, d = 4 * id - row * DD;
        if (4 * id < 4 * DD)
          *reinterpret_cast<float4*>(xs + row * XS + d) = make_float4(__uint_as_float(v[NFQ + i].x), __uint_as_float(v[NFQ + i].y),
                                                                     __uint_as_float(v[NFQ + i].z), __uint_as_float(v[NFQ + i].w));
      }
    }
    __syncthreads();
    {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, accb = (f32x4){0.f, 0.f, 0.f, 0.f};   // even / odd k partials
      const float* zr = xs + (lane_ & 3) * XS + wave * DKW + (lane_ >> 4) * DKQ;
#pragma unroll
      for (int q = 0; q < DKQ; q += 2) {
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wdreg[q], zr[q], acc, 0, 0, 0);
        if (q + 1 < DKQ) accb = __builtin_amdgcn_mfma_f32_4x4x1f32(wdreg[q + 1], zr[q + 1], accb, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += accb[i];
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
    DP_MARK(6);
    // ------------------------------------------------------------ (6) energies: partial sums over this CU's columns
    {
      // A operand: pair m of the tile = (row m % RG, frame FPT tile + m / RG); D: lane (q, col) holds pairs 4 q + i
      const int m = lane_ & 15;
      const int nkk = (C + 3) >> 2;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int tile = wave + 8 * it;
        if (FPT * tile < TpP && !(ASR_DP_ABL & 4)) {                      // wave-uniform
          f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
          const float* fa = fs + ((m % RG) * 16 + sq_) * TPM + FPT * tile + m / RG;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            if (kk < nkk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[4 * kk * TPM], ub[kk], acc, 0, 0, 0);
          float pe[4], sv[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            sv[i] = dp_tanh(acc[i] + Ps[(tile * 4 + i) * 64 + lane_] + dps[((4 * sq_ + i) % RG) * 16 + a_l_]);
            pe[i] = sc_ok_ ? gv * sv[i] : 0.f;
          }
          // uniform row bases + one 32-bit lane offset per pair (no 64-bit index math per element)
          if (sc_ok_ && !(ASR_DP_ABL & 16)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int ri = (4 * sq_ + i) % RG, ti = FPT * tile + (4 * sq_ + i) / RG;
              if (ti < Tp && r0 + ri < nb) (a.S + ((int64_t)s * B + r0 + ri) * Tp * AA)[ti * AA + acol_] = sv[i];
            }
          }
          row16_sum4(pe);
          if (a_l_ < 4) {      // lane i of each 16-lane group publishes pair i (every lane of the group holds all four sums)
            const float pv_ = a_l_ == 0 ? pe[0] : a_l_ == 1 ? pe[1] : a_l_ == 2 ? pe[2] : pe[3];
            const int ri = (4 * sq_ + a_l_) % RG, ti = FPT * tile + (4 * sq_ + a_l_) / RG;
            word_store(xg + DX_E + ((par * 32 + slice) * 4 + ri) * TPM + ti, ti < Tp ? pv_ : 0.f, bit);
          }
        }
      }
    }
    DP_MARK(7);
    // ------------------------------------------------------------ (7) full energies of this CU's row -> softmax
    {
      const float* ex = xg + DX_E + par * 32 * 4 * TPM + ar * TPM;
      // pair (2 t2, 2 t2 + 1) of frames per lane and pass; producers 4*wave .. 4*wave+3
#pragma unroll
      for (int pp = 0; pp < TPM / 128; ++pp) {
        const int t2 = lane_ + 64 * pp;
        const bool ok = 2 * t2 < TpP;
        const u64* p[4];
        u64 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          p[i] = reinterpret_cast<const u64*>(ex + (4 * wave + i) * 4 * TPM + (ok ? 2 * t2 : 0));
        poll_pairs<4, ASR_DP_FULL>(p, bit, v, a.ctrl, aborted, 14u, spin_limit);
        float e0 = 0.f, e1 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { e0 += pair_lo(v[i]); e1 += pair_hi(v[i]); }
        if (ok) { epart[wave * TPM + 2 * t2] = e0; epart[wave * TPM + 2 * t2 + 1] = e1; }
      }
      DP_MARK(8);
    }
    __syncthreads();
    {
      // every wave runs the (tiny) softmax redundantly; wave 0 keeps the results
      float ev[TPM / 64], wv[TPM / 64];
      float mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < TPM / 64; ++k) {
        const int t = lane_ + 64 * k;
        float e = 0.f;
        if (t < Tp) {
#pragma unroll
          for (int w2 = 0; w2 < 8; ++w2) e += epart[w2 * TPM + t];
        }
        ev[k] = e;
        if (t < Tp) mx = fmaxf(mx, a.scaling * e);
      }
      mx = wave_max_dpp(mx);
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < TPM / 64; ++k) {
        const int t = lane_ + 64 * k;
        wv[k] = t < Tp ? __expf(a.scaling * ev[k] - mx) : 0.f;
        sum += wv[k];
      }
      sum = wave_sum_dpp(sum);
      const float inv = 1.0f / sum;
      if (wave == 0) {
#pragma unroll
        for (int k = 0; k < TPM / 64; ++k) {
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
    DP_MARK(9);
    // ------------------------------------------------------------ (8) context slice, publish (masked) for the cell
    {
      // frames wave, wave+8, ...: 4 per trip with the 8 LDS reads issued before the FMAs (a one-frame loop waits for
      // its two reads every iteration); out-of-range frames are clamped and weighted 0
      float acc = 0.f;
      const int oc = o_l_ < OQ ? o_l_ : 0;
      for (int t0 = wave; t0 < Tp; t0 += 32) {
        float wv[4], qv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int t = t0 + 8 * u < Tp ? t0 + 8 * u : t0;
          wv[u] = wsm[t];
          qv[u] = Qs[t * OQ + oc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += t0 + 8 * u < Tp ? wv[u] * qv[u] : 0.f;
      }
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
      if (FB && drop) word_store(xg + DX_U + par * 4 * 512 + ar * 512 + o, v, bit);
    }
    DP_MARK(10);
#undef TRS
  }
}

template <int DD, int AA, int OO, int EE, bool FB, int RG = 4, int TPM = DP_TPM, bool FAULT = false>
int launch_dec_fwd(const DecPersistArgs& a, hipStream_t stream) {
  using DM = DecDims<DD, AA, OO, EE, RG, TPM>;
  const size_t lds = DM::lds_floats * sizeof(float);     // > 80 KB: one workgroup per CU
  static_assert(DM::lds_floats * sizeof(float) > 82 * 1024 && DM::lds_floats * sizeof(float) <= 160 * 1024, "LDS budget");
  static_assert(!FB || (EE == 128 && DD % 64 == 0 && OO % 64 == 0), "free-running feedback mapping");
  hipError_t e = hipFuncSetAttribute((const void*)dec_persist_fwd_kernel<DD, AA, OO, EE, FB, RG, TPM, FAULT>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((dec_persist_fwd_kernel<DD, AA, OO, EE, FB, RG, TPM, FAULT>), dim3(256), dim3(DP_NT), lds, stream, a);
  return 0;
}


// ======================================================================================================= backward
// Persistent backward of the same sequence (reverse order s = L-1 .. 0), same XCD-local organisation:
//   (a) CU (row r, part q) reads the TOTAL d(ctx_s) of its row from the exchange and forms dw_raw for its 16 frames
//       (Q slice of those frames in LDS) + the location-conv path kept from the previous iteration;      -> XW
//   (c) every CU reads dw_raw of the 4 rows (400 words), softmax backward -> de;
//   (d) score backward for this CU's A/32 attention columns x 4 rows x all frames: du = de g (1 - S^2);
//       dP accumulates in LDS for the whole sequence, dD = sum_t du                                       -> XD
//       dgvec / dW_att accumulate in registers (dW_att on the 16x16x4 MFMA);
//   (e) d(conv features) of the CU's row WITHOUT a cross-CU reduction: df[c][t] = de[t] (UG[c] - M[c][t]) with
//       M[c][t] = sum_a U[a][c] g[a] S[t][a]^2 precomputed by att_m_kernel from forward data; conv backward gives
//       the dw path of the next iteration (local) and dconv (registers);
//   (f) dz_s = G_out + dX_z(previous iteration, local) + dD W_dec for the CU's D/32 units, LSTM cell backward,
//       dgates -> global (deferred weight gradients) and                                                   -> XG
//   (g) dX = dgates W_cat for the CU's z and ctx columns (W_cat^T slice in registers, 4x4x1 MFMA); the z part
//       stays in the CU, the (masked) ctx part + G_out is the total d(ctx_{s-1})                           -> XC
// The embedding part of dX is not recurrent: one GEMM after the kernel (asr_dec_seq_bwd_persist).
// exchange layout per group (floats); the row axis keeps 4 slots in both geometries (see DecGeo)
template <int TPM>
struct BwdX {
  static constexpr int C_ = 0;                      // [2][4][512]
  static constexpr int W_ = C_ + 2 * 4 * 512;       // [2][4][TPM]
  static constexpr int D_ = W_ + 2 * 4 * TPM;       // [2][4][512]
  static constexpr int G_ = D_ + 2 * 4 * 512;       // [2][4][2048]
  // free-running (smooth feedback) backward only:
  static constexpr int P_ = G_ + 2 * 4 * 2048;      // [2][32 source CUs][4][36]  partial d(probabilities) of the CU's embedding columns
  static constexpr int L_ = P_ + 2 * 32 * 4 * 36;   // [2][4][36]                 d(logits_{s-1}) through the smooth embedding
  static constexpr int GROUP = L_ + 2 * 4 * 36;
};

struct DecPersistBwdArgs {
  int B, nb, Tp, C, K, L;
  float scaling;
  const float *Q, *wcatT, *wdecT, *convw, *watt, *gvec, *w0, *xmask;
  const float *gates, *cstate, *S, *fconv, *ws, *Mf, *dws;
  float *G, *dgates, *dD, *dP, *dgvec_part, *dwatt_part, *dconv_part;
  float* dbg;          // measurement builds only (ASR_DP_DEBUG): d(conv features) of the last step, [B][C][Tp]
  float* xch;
  unsigned* ctrl;
  // smooth-embedding feedback (kernel template FB): emb_s = softmax(fb_scale * logit_{s-1}) @ emb couples step s to the
  // logits of step s - 1 (model.py:341).  probs [L-1][B][V] saved by the forward; dlfb [L][B][V] receives the gradient
  // that reaches logit_{s-1} through it (the caller adds it to the upstream d(logits) for the output-layer weights)
  int V;
  float fb_scale;
  const float *w_out, *emb, *probs;
  float* dlfb;
};

template <int DD, int AA, int OO>
struct DecBwdDims {
  static constexpr int GK = 4 * DD;        // K of the dX product
  static constexpr int GKW = GK / 8;       // per wave
  static constexpr int GKS = GK / 16;      // per (wave, k-sub): registers per lane
  static constexpr int GS = GK + 4;        // padded LDS row of the gathered dgates
  static constexpr int DU = DD / 32, AU = AA / 32, OU = OO / 32;
  static constexpr int AKW = AA / 8, AQ = AA / 32;   // dz product: k's per wave / per (wave, k-sub)
  static constexpr int DS = AA + 4;        // padded LDS row of the gathered dD
  static_assert(DD % 32 == 0 && AA % 32 == 0 && OO % 32 == 0 && GKS % 4 == 0, "slice sizes");
  static_assert(DU <= 16 && AU <= 16 && OU <= 16 && DD <= 512 && OO <= 512 && AA <= 512, "mappings");
};

// LDS plan of the backward kernel (floats), sized from the run-time T', C, K
struct BwdLds {
  int dgs, part, qs, dcx, fs, dps, Fs, dfh, wph, des, dwr, wsl, dds, ddp, dwext, ugs, wos, total;
  int dfs_stride, taps4;
};
template <int DD, int AA, int OO, int RG = 4, int TPM = DP_TPM>
__host__ __device__ inline BwdLds bwd_lds_plan(int Tp, int C, int K, int fbV = 0) {
  using BM = DecBwdDims<DD, AA, OO>;
  const int TpP = (Tp + 3) & ~3;
  BwdLds l;
  int o = 0;
  l.taps4 = ((2 * K + 1) + 3) & ~3;
  // row stride of dfh == 13 (mod 32): the Toeplitz products read dfh[c][4 i + const] for 10 channels x 4 offsets in one
  // wave instruction; a stride that is a multiple of 16 puts every second channel on the same LDS bank (5-way conflict)
  l.dfs_stride = TpP + 2 * K + 4;
  l.dfs_stride += (13 - (l.dfs_stride & 31) + 32) & 31;
  l.dgs = o; o += RG * BM::GS;
  l.part = o; o += 8 * 64 * 5;
  l.qs = o; o += 16 * OO;
  l.dcx = o; o += OO;
  l.fs = o; o += RG * C * TpP;
  l.dps = o; o += ((TpP + 16 / RG - 1) / (16 / RG)) * 4 * 64;       // [tiles of 16 / RG frames][4 pairs][64 lanes]
  l.Fs = o; o += C * (l.taps4 + 8);     // rows zero padded by 8: the Toeplitz products read up to 6 taps past the end
  l.dfh = o; o += C * l.dfs_stride;
  l.wph = o; o += l.dfs_stride;
  l.des = o; o += RG * TPM;
  l.dwr = o; o += RG * TPM;
  l.wsl = o; o += RG * TPM;
  l.dds = o; o += RG * BM::DS;
  l.ddp = o; o += 8 * 4 * 16;
  l.dwext = o; o += 16;
  l.ugs = o; o += 16;
  l.wos = o; o += fbV * 32;            // free-running backward: W_out[v][my 16 z + 16 ctx columns]
  l.total = o + 8;
  return l;
}

// FB = backward of a free-running sequence with the smooth-embedding feedback (the semi-supervised generator step,
// solver.py:465-470): per step, after the dgates of the group's rows have been gathered for the dX product,
//   (g2) CU j forms d(emb_s) for ITS E/32 embedding columns from the gathered dgates (K = 4D; its slice of W_cat^T streamed
//        from L2, 16 values per thread), applies the dropout mask, stores it (the embedding-weight gradient is a GEMM
//        after the kernel) and publishes its share of d(probabilities): pdp[row][v] = sum_{e mine} d(emb)[row][e] E[v][e];
//   (g3) slice 0 sums the 32 shares, runs the softmax backward with the saved probabilities,
//        dl = k p (dp - sum_u p_u dp_u), stores dl (-> dlfb) and publishes it;
//   (g4) every CU adds dl W_out for its 16 z / 16 ctx columns (W_out slice in LDS) to the dX it hands to step s - 1.
// Two more hand-offs per step.  The row axis of every feedback buffer keeps 4 slots in both geometries: with 2 rows per
// group (T' <= 256) rows 2, 3 alias rows 0, 1 and are never stored.
template <int DD, int AA, int OO, int EE, int RG = 4, int TPM = DP_TPM, bool FB = false>
__global__ __launch_bounds__(DP_NT) void dec_persist_bwd_kernel(DecPersistBwdArgs a) {
  static_assert(!FB || EE == 128, "the feedback backward is written for E = 128");
  using BM = DecBwdDims<DD, AA, OO>;
  using GEO = DecGeo<RG, TPM>;
  using BX = BwdX<TPM>;
  constexpr int PPR = GEO::PPR, LR = GEO::LR, FPT = GEO::FPT;
  constexpr int BX_C = BX::C_, BX_W = BX::W_, BX_D = BX::D_, BX_G = BX::G_, BX_GROUP = BX::GROUP;
  constexpr int NFR = RG == 4 ? 8 : 10;      // conv features of the RG rows per thread (ids over [RG][C][TpP])
  constexpr int NMR = RG == 4 ? 2 : 5;       // M values per thread (ids over [C][TpP])
  constexpr int LT = RG == 4 ? 7 : 8;        // log2(TPM)
  static_assert((1 << LT) == TPM && RG * TPM == DP_NT, "one thread per (row, frame) of the attention weights");
  constexpr int KX = DD + OO + EE;
  constexpr int GK = BM::GK, GKW = BM::GKW, GKS = BM::GKS, GS = BM::GS, DU = BM::DU, AU = BM::AU, OU = BM::OU;
  constexpr int AKW = BM::AKW, AQ = BM::AQ, DS = BM::DS;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Tp = a.Tp, C = a.C, K = a.K, B = a.B, L = a.L, nb = a.nb;
  const int TpP = (Tp + 3) & ~3;
  const BwdLds ld = bwd_lds_plan<DD, AA, OO, RG, TPM>(Tp, C, K, FB ? a.V : 0);
  constexpr int BX_P = BX::P_, BX_L = BX::L_;
  float* wos = sm + ld.wos;      // [V][32]          FB: W_out[v][z columns of my units | my ctx columns]
  float* dgs = sm + ld.dgs;      // [4][GS]          gathered dgates of the 4 rows
  float* part = sm + ld.part;    // [8][64][5]       K-partials of both MFMA products
  float* Qs = sm + ld.qs;        // [16][OO]         Q rows of this CU's 16 frames
  float* dcx = sm + ld.dcx;      // [OO]             total d(ctx_s) of this CU's row
  float* fs = sm + ld.fs;        // [4][C][TpP]      conv features f_s of the 4 rows
  float* dPs = sm + ld.dps;      // [TpP/4][4][64]   dP accumulators in the score-lane layout
  float* Fs = sm + ld.Fs;        // [C][taps4 + 8]
  float* dfh = sm + ld.dfh;      // [C][dfs_stride]  d(conv features) of this CU's row, zero halo of K
  float* wph = sm + ld.wph;      // [dfs_stride]     w_{s-1} of this CU's row, zero halo of K
  float* des = sm + ld.des;      // [4][TPM]         d(energy)
  float* dwr = sm + ld.dwr;      // [4][TPM]         dw_raw
  float* wsl = sm + ld.wsl;      // [4][TPM]         w_s
  float* dDs = sm + ld.dds;      // [4][DS]          gathered dD
  float* dDp = sm + ld.ddp;      // [8][4][16]       per-wave dD partials
  float* dwext = sm + ld.dwext;  // [16]             conv-path dw of this CU's frames (for the next iteration)
  float* ugs = sm + ld.ugs;      // [16]             UG[c] = sum_a U[a][c] g[a]
  int* role = reinterpret_cast<int*>(sm + ld.total - 8);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int g, slice;
  take_role(a.ctrl, role, g, slice);
  if (slice < 0) return;
  const int r0 = RG * g;
  if (r0 >= nb) return;
  const int taps = 2 * K + 1, taps4 = ld.taps4, DFS = ld.dfs_stride;
  const bool drop = a.xmask != nullptr;
  float* xg = a.xch + (int64_t)g * BX_GROUP;
  const __amdgpu_buffer_rsrc_t xrs = make_xch_rsrc(a.xch);
  bool aborted = false;
  const int ar = slice >> LR, aq = slice & (PPR - 1);
  const int ab = r0 + ar;
  const bool ab_ok = ab < nb;
  const int abc = ab_ok ? ab : r0;

  // ---------------------------------------------------------------- weights in registers
  // dX product: block = 2*cg + ks; column group cg < 4: z columns DU*slice + 4cg + i, cg >= 4: ctx columns
  // D + OU*slice + 4(cg-4) + i; k = wave*GKW + ks*GKS + q
  float wx[GKS];
  {
    const int blk = lane >> 2, i = lane & 3, cg = blk >> 1, ks = blk & 1;
    const int cl = 4 * (cg & 3) + i;
    const bool ok = cg < 4 ? cl < DU : cl < OU;
    const int col = cg < 4 ? DU * slice + (ok ? cl : 0) : DD + OU * slice + (ok ? cl : 0);
    const float* wr = a.wcatT + (int64_t)col * GK + wave * GKW + ks * GKS;
#pragma unroll
    for (int q4 = 0; q4 < GKS / 4; ++q4) {
      const float4 v = *reinterpret_cast<const float4*>(wr + 4 * q4);
      wx[4 * q4] = ok ? v.x : 0.f; wx[4 * q4 + 1] = ok ? v.y : 0.f;
      wx[4 * q4 + 2] = ok ? v.z : 0.f; wx[4 * q4 + 3] = ok ? v.w : 0.f;
    }
  }
  // dz product: block = 4*ks + ug; lane 4*blk+i holds unit DU*slice + 4ug + i, k (= attention column) = wave*AKW + ks*AQ + q
  float wd[AQ];
  {
    const int blk = lane >> 2, ks = blk >> 2, ug = blk & 3, ul = 4 * ug + (lane & 3);
    const bool ok = ul < DU;
    const float* wr = a.wdecT + (int64_t)(DU * slice + (ok ? ul : 0)) * AA + wave * AKW + ks * AQ;
#pragma unroll
    for (int q = 0; q < AQ; ++q) wd[q] = ok ? wr[q] : 0.f;
  }
  // ---------------------------------------------------------------- LDS images
  for (int i = tid; i < ld.total - 8; i += DP_NT) sm[i] = 0.f;      // accumulators, halos, padding
  __syncthreads();
  for (int i = tid; i < 16 * OO; i += DP_NT) {
    const int tl = i / OO, o = i - tl * OO;
    const int t = 16 * aq + tl;
    Qs[i] = a.Q[((int64_t)abc * Tp + (t < Tp ? t : Tp - 1)) * OO + o];
  }
  const int FSS = taps4 + 8;
  for (int i = tid; i < C * FSS; i += DP_NT) {
    const int ch = i / FSS, j = i - ch * FSS;
    Fs[i] = j < taps ? a.convw[ch * taps + j] : 0.f;
  }
  if (tid < 16) {
    float v = 0.f;
    if (tid < C)
      for (int aa = 0; aa < AA; ++aa) v += a.watt[(int64_t)aa * C + tid] * a.gvec[aa];
    ugs[tid] = v;
  }
  float ereg[4] = {0.f, 0.f, 0.f, 0.f};     // FB: E[v][my 4 embedding columns] of thread (row = tid / 36, v = tid % 36)
  if (FB) {
    for (int i = tid; i < a.V * 32; i += DP_NT) {
      const int v = i >> 5, ci = i & 31;
      const bool ok = ci < 16 ? ci < DU : ci - 16 < OU;
      const int col = ci < 16 ? DU * slice + ci : DD + OU * slice + ci - 16;
      wos[i] = ok ? a.w_out[(int64_t)v * (DD + OO) + col] : 0.f;
    }
    if (tid < 144 && tid % 36 < a.V) {
#pragma unroll
      for (int e = 0; e < 4; ++e) ereg[e] = a.emb[(tid % 36) * EE + 4 * slice + e];
    }
  }
  // score lanes (as in the forward kernel): column a_l = lane&15 of this CU's slice, frame quarter lane>>4
  const float gv = (lane & 15) < AU ? a.gvec[AU * slice + (lane & 15)] : 0.f;
  // persistent accumulators
  f32x4 acc_watt = (f32x4){0.f, 0.f, 0.f, 0.f};   // dW_att[a = 4*(lane>>4)+i][c = lane&15] partial of this wave
  float dgl = 0.f;                                // dgvec partial of this lane's column
  f32x4 acc_cv0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc_cv1 = (f32x4){0.f, 0.f, 0.f, 0.f};   // dconv tiles aq, aq+8 (see (e))
  float dcarry = 0.f, dxz = 0.f;
  // conv-weight gradient ownership: this CU owns taps idx = aq*NCJ + tid (tid < NCJ) of its row's C*taps
  // pointwise threads of the cell (tid < 4*DU): unit tid>>2, row tid&3
  const bool pw_thread = tid < 4 * DU;
  const int punit = DU * slice + (pw_thread ? (tid >> 2) : 0);
  const int pb = r0 + (tid & 3);
  const bool pb_ok = pw_thread && (tid & 3) < RG && pb < nb;
  const int pbc = ((tid & 3) < RG && pb < nb) ? pb : r0;
  // ctx-column threads of the dX result (64 <= tid < 64 + 4*OU): column OU*slice + ((tid-64)>>2), row tid&3
  const bool cx_thread = tid >= 64 && tid < 64 + 4 * OU;
  const int cxcol = OU * slice + (cx_thread ? ((tid - 64) >> 2) : 0);

  // ---------------------------------------------------------------- prefetch registers (data of iteration n)
  float sreg[16];      // S[s][pair 4 q4 + i of tile wave + 8 it][acol]: row (4 q4 + i) % RG, frame FPT tile + (4 q4 + i) / RG
  float freg[NFR];     // conv features of the RG rows (ids tid + 512 k over [RG][C][TpP])
  float mreg[NMR];     // M[s][row ar][c][t] (ids tid + 512 k over [C][TpP])
  float wsreg = 0.f, wpreg = 0.f, dwsreg = 0.f;
  float4 ga = make_float4(0.f, 0.f, 0.f, 0.f);
  float ct = 0.f, cp = 0.f, gz = 0.f, gc = 0.f, xm = 1.f;
  // forward data of step s (no dependence on the recurrence), issued ~one iteration before use.  Indices are derived
  // from an opaque copy of the thread id (see the forward kernel) and addresses are uniform base + 32-bit lane offset.
  auto prefetchA = [&](int s_, int zq) {
    const int s = (ASR_DP_ABL & 64) ? (s_ > 0 ? 1 : 0) : s_;     // bit 64 (measurement): always the same, cached rows
    const int tidq = tid + zq, laneq = lane + zq;
    const int q4q = laneq >> 4, alq = laneq & 15;
    const int acolq = AU * slice + (alq < AU ? alq : 0);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int t = FPT * (wave + 8 * it) + (4 * q4q + i) / RG;
        const int b = r0 + (4 * q4q + i) % RG;
        const float* sb = a.S + ((int64_t)s * B + (b < nb ? b : r0)) * Tp * AA;
        sreg[it * 4 + i] = sb[(t < Tp ? t : Tp - 1) * AA + acolq];
      }
    }
    const float rct = 1.0f / (float)(C * TpP), rtp = 1.0f / (float)TpP;
    const float* fb = a.fconv + (int64_t)s * B * C * Tp;
#pragma unroll
    for (int k = 0; k < NFR; ++k) {
      const int id = tidq + DP_NT * k;
      const int row = (int)(((float)id + 0.5f) * rct), rem = id - row * C * TpP;
      const int c = (int)(((float)rem + 0.5f) * rtp), t = rem - c * TpP;
      const int b = r0 + (row < RG ? row : 0);
      freg[k] = fb[((b < nb ? b : r0) * C + (row < RG ? c : 0)) * Tp + (t < Tp ? t : Tp - 1)];
    }
    const float* mb = a.Mf + ((int64_t)s * B + abc) * C * Tp;
#pragma unroll
    for (int k = 0; k < NMR; ++k) {
      const int id = tidq + DP_NT * k;
      const int c = (int)(((float)id + 0.5f) * rtp), t = id - c * TpP;
      mreg[k] = mb[(c < C ? c : 0) * Tp + (t < Tp && c < C ? t : Tp - 1)];
    }
    {
      const int row = tidq >> LT, t = tidq & (TPM - 1), b = r0 + row;      // w_s of the RG rows: RG x TPM threads
      wsreg = a.ws[((int64_t)s * B + (b < nb ? b : r0)) * Tp + (t < Tp ? t : Tp - 1)];
      const float* wprev = s > 0 ? a.ws + ((int64_t)(s - 1) * B + abc) * Tp : a.w0 + (int64_t)abc * Tp;
      wpreg = wprev[t < Tp ? t : Tp - 1];                            // (threads tid < TPM use it)
      const int tq = 16 * aq + (tidq >> 5);
      dwsreg = a.dws ? a.dws[((int64_t)s * B + abc) * Tp + (tq < Tp ? tq : Tp - 1)] : 0.f;
    }
  };
  auto prefetchB = [&](int s_, int zq) {
    const int s = (ASR_DP_ABL & 64) ? (s_ > 0 ? 1 : 0) : s_;
    const int tidq = tid + zq;
    const int pbq = r0 + (tidq & 3);
    const int pbcq = pbq < nb ? pbq : r0;
    if (tidq < 4 * DU) {
      const int un = DU * slice + (tidq >> 2);
      ga = *reinterpret_cast<const float4*>(a.gates + ((int64_t)s * B + pbcq) * 4 * DD + un * 4);
      ct = a.cstate[((int64_t)s * B + pbcq) * DD + un];
      cp = s > 0 ? a.cstate[((int64_t)(s - 1) * B + pbcq) * DD + un] : 0.f;
      gz = a.G[((int64_t)(s + 1) * B + pbcq) * KX + un];
    }
    if (tidq >= 64 && tidq < 64 + 4 * OU) {
      const int col = OU * slice + ((tidq - 64) >> 2);
      gc = a.G[((int64_t)s * B + pbcq) * KX + DD + col];            // G_out part of d(ctx_{s-1}) (row tid&3)
      xm = drop ? a.xmask[((int64_t)s * B + pbcq) * (OO + EE) + col] : 1.f;
    }
  };
  {
    int z0;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z0));
    prefetchA(L - 1, z0);
    prefetchB(L - 1, z0);
  }
  // prologue: the total d(ctx_{L-1}) is the output-layer gradient alone
  if (cx_thread)
    word_store(xg + BX_C + 0 * 4 * 512 + (tid & 3) * 512 + cxcol,
               a.G[((int64_t)L * B + pbc) * KX + DD + cxcol], tag_bit_of_step(0));
  __syncthreads();

  for (int n = 0; n < L; ++n) {
    const int s = L - 1 - n;
    const unsigned bit = tag_bit_of_step(n);
    const int slot = n & 1;
    int zv;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zv));
    const int tid_ = tid + zv, lane_ = lane + zv;
    const unsigned abort_seen = tid < 64 ? flag_load(a.ctrl + 8) : 0u;   // sampled early, consumed by the pointwise phase
    const int a_l_ = lane_ & 15, q4_ = lane_ >> 4;
    const bool sc_ok_ = a_l_ < AU;
#define TRS n
    DP_MARK(0);
    // ------------------------------------------------------------ (0) prefetched forward data -> LDS
    {
      const float rct = 1.0f / (float)(C * TpP), rtp = 1.0f / (float)TpP;
#pragma unroll
      for (int k = 0; k < NFR; ++k) {
        const int id = tid_ + DP_NT * k;
        if (id < RG * C * TpP) {
          const int row = (int)(((float)id + 0.5f) * rct), rem = id - row * C * TpP;
          const int c = (int)(((float)rem + 0.5f) * rtp), t = rem - c * TpP;
          fs[id] = t < Tp ? freg[k] : 0.f;
        }
      }
      const int t = tid_ & (TPM - 1);
      wsl[tid_] = t < Tp ? wsreg : 0.f;                           // [row = tid >> log2(TPM)][t]
      if (tid_ < TPM && t < Tp) wph[K + t] = wpreg;
    }
    DP_MARK(1);
    // ------------------------------------------------------------ (a) total d(ctx_s) of my row -> dw_raw of my frames
    {
      const float* cx = xg + BX_C + slot * 4 * 512 + ar * 512;
      const u64* p[1];
      u64 v[1];
      p[0] = reinterpret_cast<const u64*>(cx + (tid_ < OO / 2 ? 2 * tid_ : 0));
      poll_pairs<1, true>(p, bit, v, a.ctrl, aborted, 21u);
      if (tid_ < OO / 2) { dcx[2 * tid_] = pair_lo(v[0]); dcx[2 * tid_ + 1] = pair_hi(v[0]); }
    }
    __syncthreads();
    {
      const int tl = tid_ >> 5, op = tid_ & 31;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < OO / 32; ++k) acc += Qs[tl * OO + op + 32 * k] * dcx[op + 32 * k];
      acc = row16_sum(acc);
      acc += __shfl_xor(acc, 16, 64);
      const int t = 16 * aq + tl;
      if (op == 0 && t < TpP) word_store(xg + BX_W + (slot * 4 + ar) * TPM + t, t < Tp ? acc + dwext[tl] + dwsreg : 0.f, bit);
#ifdef ASR_DP_DEBUG
      if (op == 0 && t < Tp && n == 1 && ab_ok) a.dbg[(int64_t)3 * B * C * Tp + (int64_t)ab * Tp + t] = dwext[tl];
#endif
    }
    DP_MARK(2);
    // ------------------------------------------------------------ (c) dw_raw of the 4 rows -> softmax backward
    {
      const int hp = TpP >> 1;
      const int row = (int)(((float)tid_ + 0.5f) * (1.0f / (float)hp)), t2 = tid_ - row * hp;
      const bool ok = row < RG;                     // rows >= RG are never published
      const u64* p[1];
      u64 v[1];
      p[0] = reinterpret_cast<const u64*>(xg + BX_W + slot * 4 * TPM + (ok ? row * TPM + 2 * t2 : 0));
      poll_pairs<1, true>(p, bit, v, a.ctrl, aborted, 22u);
      if (ok) { dwr[row * TPM + 2 * t2] = pair_lo(v[0]); dwr[row * TPM + 2 * t2 + 1] = pair_hi(v[0]); }
    }
    __syncthreads();
    if (wave < RG) {
      float w[TPM / 64], dv[TPM / 64], dot = 0.f;
#pragma unroll
      for (int k = 0; k < TPM / 64; ++k) {
        const int t = lane_ + 64 * k;
        w[k] = t < Tp ? wsl[wave * TPM + t] : 0.f;
        dv[k] = t < Tp ? dwr[wave * TPM + t] : 0.f;
        dot += w[k] * dv[k];
      }
      dot = wave_sum_dpp(dot);
#pragma unroll
      for (int k = 0; k < TPM / 64; ++k)     // rows beyond the batch contribute nothing to the sequence-long accumulators
        des[wave * TPM + lane_ + 64 * k] = r0 + wave < nb ? a.scaling * w[k] * (dv[k] - dot) : 0.f;
    }
    __syncthreads();
    DP_MARK(3);
    // ------------------------------------------------------------ (d) score backward for my attention columns
    {
      float dDl[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int tile = wave + 8 * it;
        if (FPT * tile < TpP && !(ASR_DP_ABL & 256)) {              // bit 256 (measurement): no score backward
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            // pair 4 q4 + i of the tile: row ri (a compile-time constant: 4 q4 is a multiple of RG), frame t
            const int ri = i % RG, t = FPT * tile + (4 * q4_ + i) / RG;
            const float sv = sreg[it * 4 + i];
            const float det = des[ri * TPM + t];          // des is defined (0 beyond T') for every t < TPM
            const float duv = sc_ok_ ? det * gv * (1.f - sv * sv) : 0.f;
            dPs[(tile * 4 + i) * 64 + lane_] += duv;
            dDl[ri] += duv;
            dgl += sc_ok_ ? det * sv : 0.f;
            // dW_att[a][c] += sum over the k-group of this MFMA: A[m = a][k = q4] = du, B[k = q4][n = c] = f[row][c][frame]
            const float fb = fs[(ri * C + (a_l_ < C ? a_l_ : 0)) * TpP + ((RG == 4 || t < TpP) ? t : 0)];
            acc_watt = __builtin_amdgcn_mfma_f32_16x16x4f32(duv, a_l_ < C ? fb : 0.f, acc_watt, 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = dDl[i];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (q4_ == 0) dDp[(wave * 4 + i) * 16 + a_l_] = v;
      }
    }
    __syncthreads();
    if (tid_ < 64) {
      const int row = tid_ >> 4, al = tid_ & 15;
      float v = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < 8; ++w2) v += dDp[(w2 * 4 + row) * 16 + al];
      const int b = r0 + row;
      if (al < AU) {
        if (row < RG && b < nb && !(ASR_DP_ABL & 2048)) a.dD[((int64_t)s * B + b) * AA + AU * slice + al] = v;
        word_store(xg + BX_D + (slot * 4 + row) * 512 + AU * slice + al, v, bit);
      }
    }
    DP_MARK(4);
    // ------------------------------------------------------------ (e) d(conv features) of my row, conv backward
    {
      const float rtp = 1.0f / (float)TpP;
#pragma unroll
      for (int k = 0; k < NMR; ++k) {
        const int id = tid_ + DP_NT * k;
        if (id < C * TpP) {
          const int c = (int)(((float)id + 0.5f) * rtp), t = id - c * TpP;
          if (t < Tp) dfh[c * DFS + K + t] = des[ar * TPM + t] * (ugs[c] - mreg[k]);
#ifdef ASR_DP_DEBUG
          if (t < Tp && n == 0 && aq == 0 && ab_ok) {
            a.dbg[((int64_t)ab * C + c) * Tp + t] = des[ar * TPM + t] * (ugs[c] - mreg[k]);
            a.dbg[(int64_t)B * C * Tp + ((int64_t)ab * C + c) * Tp + t] = mreg[k];
            a.dbg[(int64_t)2 * B * C * Tp + ((int64_t)ab * C + c) * Tp + t] = des[ar * TPM + t];
          }
#endif
        }
      }
    }
    __syncthreads();
    {
      DP_MARK(10);
      // Both conv-backward contractions are Toeplitz products on the 4x4x1 MFMA with block = channel:
      //   D[c][i][jj] += A[c][i] * B[c][jj], one instruction per summation index, K range dealt round-robin to waves.
      // (1) dw path of the next iteration, dwext[t'] = sum_c sum_j F[c][j] df[c][t' - j + K] for t' = t0 + 4i + jj:
      //     A = dfh[c][t0 + 4i + 2K - j], B = F[c][j + jj]   (substituting j -> j + jj keeps A independent of jj)
      const int cb = lane_ >> 2, li = lane_ & 3;
      const bool cok = cb < C;
      const int cc = cok ? cb : 0;
      {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, accb = (f32x4){0.f, 0.f, 0.f, 0.f};   // alternating partials
        // The summation index runs from -3: output jj needs the taps j + jj >= 0, i.e. j >= -jj (their operands are d(conv
        // features) of frames beyond t0 + 4i + K, which are zero only while T' <= K + 1 - the case of every test of
        // round 1).  js = j + 4 >= 0 is the loop variable; taps with j + jj < 0 are masked.
        const float* ap = dfh + cc * DFS + 16 * aq + 4 * li + 2 * K + 4;
        const float* bp = Fs + cc * FSS + li - 4;
        const int jend = (ASR_DP_ABL & 128) ? 1 : taps + 3 + 4;      // bit 128 (measurement): one trip of each Toeplitz product
        for (int j = wave; j < jend; j += 32) {      // 4 taps per trip: 8 LDS reads in flight, then 4 MFMAs
          float av[4], bv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int jc = j + 8 * u < jend ? j + 8 * u : j;
            av[u] = ap[-jc];
            bv[u] = bp[jc];
          }
#pragma unroll
          for (int u = 0; u < 4; u += 2) {
            const int j0 = j + 8 * u, j1 = j + 8 * (u + 1);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32((cok && j0 < jend) ? av[u] : 0.f, j0 - 4 + li >= 0 ? bv[u] : 0.f, acc, 0, 0, 0);
            accb = __builtin_amdgcn_mfma_f32_4x4x1f32((cok && j1 < jend) ? av[u + 1] : 0.f, j1 - 4 + li >= 0 ? bv[u + 1] : 0.f, accb, 0, 0, 0);
          }
        }
        float* pp = part + (wave * 64 + lane_) * 5;
#pragma unroll
        for (int i = 0; i < 4; ++i) pp[i] = acc[i] + accb[i];
      }
      // (2) dconv[c][j0 + 4i + jj] += sum_u w_{s-1}[u + j0 + 4i - K] df[c][u - jj]: A = wph[u + j0 + 4i], B = dfh[c][K + u - jj];
      //     this CU owns the 16-tap tiles aq and aq + PPR of its row; accumulators live in registers for the whole sequence
      {
        const float* bq = dfh + cc * DFS + K - li;
        const float* a0 = wph + 16 * aq + 4 * li;
        const bool two = 16 * (aq + PPR) < taps;
        const int uend = (ASR_DP_ABL & 128) ? 1 : Tp + 3;
        for (int u = wave; u < uend; u += 32) {
          float bv[4], a0v[4], a1v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int uc = u + 8 * k < uend ? u + 8 * k : u;
            bv[k] = bq[uc];
            a0v[k] = a0[uc];
            a1v[k] = a0[two ? uc + 16 * PPR : uc];
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float b = (cok && u + 8 * k < uend) ? bv[k] : 0.f;
            acc_cv0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0v[k], b, acc_cv0, 0, 0, 0);
            acc_cv1 = __builtin_amdgcn_mfma_f32_4x4x1f32(two ? a1v[k] : 0.f, b, acc_cv1, 0, 0, 0);
          }
        }
      }
      DP_MARK(11);
      __syncthreads();
      if (tid_ < 128) {      // (wave w, output o = 4i + jj): sum over channels
        const int w2 = tid_ >> 4, o = tid_ & 15, i = o >> 2, jj = o & 3;
        float v = 0.f;
        for (int c = 0; c < C; ++c) v += part[(w2 * 64 + 4 * c + jj) * 5 + i];
        dDp[w2 * 16 + o] = v;
      }
      __syncthreads();
      if (tid_ < 16) {
        float v = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 8; ++w2) v += dDp[w2 * 16 + tid_];
        dwext[tid_] = 16 * aq + tid_ < Tp ? v : 0.f;
      }
      DP_MARK(12);
    }
    DP_MARK(5);
    // ------------------------------------------------------------ (f) dz_s for my units, LSTM cell backward
    {
      // dD of the 4 rows in 16-byte quads over [4][AA/4]
      constexpr int ND = (AA + DP_NT - 1) / DP_NT;
      const unsigned dbase = (unsigned)((xg - a.xch) + BX_D + slot * 4 * 512) * 4u;
      unsigned off[ND];
      u4v v[ND];
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int id = tid_ + DP_NT * i;
        const int row = (4 * id) / AA, c4 = 4 * id - row * AA;
        off[i] = dbase + (unsigned)((4 * id < 4 * AA) ? row * 512 + c4 : 0) * 4u;
      }
      poll_quads<ND, true>(xrs, off, bit, v, a.ctrl, aborted, 23u);
      DP_MARK(6);
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int id = tid_ + DP_NT * i;
        const int row = (4 * id) / AA, c4 = 4 * id - row * AA;
        if (4 * id < RG * AA)
          *reinterpret_cast<float4*>(dDs + row * DS + c4) = make_float4(__uint_as_float(v[i].x), __uint_as_float(v[i].y),
                                                                       __uint_as_float(v[i].z), __uint_as_float(v[i].w));
      }
    }
    __syncthreads();
    {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, accb = (f32x4){0.f, 0.f, 0.f, 0.f};   // even / odd k partials
      const float* dr = dDs + ((lane_ & 3) % RG) * DS + wave * AKW + (lane_ >> 4) * AQ;      // rows >= RG alias (results unused)
#pragma unroll
      for (int q = 0; q < ((ASR_DP_ABL & 1024) ? 2 : AQ); q += 2) {      // bit 1024 (measurement): one pair of the dz product
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wd[q], dr[q], acc, 0, 0, 0);
        if (q + 1 < AQ) accb = __builtin_amdgcn_mfma_f32_4x4x1f32(wd[q + 1], dr[q + 1], accb, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += accb[i];
      float* pp = part + (wave * 64 + lane_) * 5;
#pragma unroll
      for (int i = 0; i < 4; ++i) pp[i] = acc[i];
    }
    __syncthreads();
    if (tid_ < 4 * DU) {
      const int ul = tid_ >> 2, row = tid_ & 3, ug = ul >> 2, ii = ul & 3;
      float dh = gz + dxz;
#pragma unroll
      for (int w2 = 0; w2 < 8; ++w2)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) dh += part[(w2 * 64 + 4 * (4 * ks + ug) + row) * 5 + ii];
      const float tc = asr_fast_tanh(ct);
      const float dc = dcarry + dh * ga.w * (1.f - tc * tc);
      float4 da;
      da.x = dc * ga.z * ga.x * (1.f - ga.x);
      da.y = dc * cp * ga.y * (1.f - ga.y);
      da.z = dc * ga.x * (1.f - ga.z * ga.z);
      da.w = dh * tc * ga.w * (1.f - ga.w);
      dcarry = dc * ga.y;
      if (aborted || abort_seen != 0u) da.x = __builtin_nanf("");
      const int un = DU * slice + ul, b = r0 + row;
      if (row < RG && b < nb && !(ASR_DP_ABL & 2048)) *reinterpret_cast<float4*>(a.dgates + ((int64_t)s * B + b) * GK + un * 4) = da;   // bit 2048: no global stores
      float* dst = xg + BX_G + (slot * 4 + row) * 2048 + un * 4;
      word_store(dst, da.x, bit); word_store(dst + 1, da.y, bit);
      word_store(dst + 2, da.z, bit); word_store(dst + 3, da.w, bit);
    }
    DP_MARK(7);
    if (s == 0) break;
    // ------------------------------------------------------------ (g) dX = dgates W_cat for my z / ctx columns
    {
      // the dgates of the 4 rows in 16-byte quads over [4][GK/4]: 4 loads per thread in one poll (8 pairs in two before)
      constexpr int NG = (GK + DP_NT - 1) / DP_NT;
      const unsigned gbase = (unsigned)((xg - a.xch) + BX_G + slot * 4 * 2048) * 4u;
      unsigned off[NG];
      u4v v[NG];
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const int id = tid_ + DP_NT * i;
        const int row = (4 * id) / GK, c4 = 4 * id - row * GK;
        off[i] = gbase + (unsigned)((4 * id < 4 * GK) ? row * 2048 + c4 : 0) * 4u;
      }
      poll_quads<NG, true>(xrs, off, bit, v, a.ctrl, aborted, 24u);
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const int id = tid_ + DP_NT * i;
        const int row = (4 * id) / GK, c4 = 4 * id - row * GK;
        if (4 * id < RG * GK)
          *reinterpret_cast<float4*>(dgs + row * GS + c4) = make_float4(__uint_as_float(v[i].x), __uint_as_float(v[i].y),
                                                                       __uint_as_float(v[i].z), __uint_as_float(v[i].w));
      }
    }
    __syncthreads();
    // FB: this thread's 16 values of W_cat^T for (embedding column 4 slice + (wave & 3), k chunk 64 (wave >> 2) + lane), the
    // dropout mask of d(emb) and (slice 0) the saved probabilities: L2 / HBM loads issued here, consumed after the dX product
    float4 wfe[4];
    float fb_mask = 1.f, fb_prob = 0.f;
    if (FB) {
      const int kcg = 64 * (wave >> 2) + lane_;                       // k chunk of 16; chunks beyond 4D / 16 idle (D < 512)
      const float4* wp = reinterpret_cast<const float4*>(a.wcatT + (int64_t)(DD + OO + 4 * slice + (wave & 3)) * GK +
                                                          (16 * kcg < GK ? 16 * kcg : 0));
#pragma unroll
      for (int i = 0; i < 4; ++i) wfe[i] = wp[i];
      const int mrow = tid_ & 3, mb = r0 + mrow;
      if (drop && tid_ < 16)
        fb_mask = a.xmask[((int64_t)s * B + (mb < nb ? mb : r0)) * (OO + EE) + OO + 4 * slice + (tid_ >> 2)];
      if (slice == 0 && wave < RG) {
        const int pb2 = r0 + wave;
        fb_prob = (lane_ < a.V && pb2 < nb) ? a.probs[((int64_t)(s - 1) * B + pb2) * a.V + lane_] : 0.f;
      }
    }
    // Forward data of the next iteration (independent of the recurrence), issued right after the last poll of this
    // iteration: vmcnt retires in order, so these HBM first-touch loads would hold back any poll issued behind them;
    // from here the next poll is ~2 us away and the data is consumed ~4 us later.
    prefetchA(s - 1, zv);
    {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, accb = (f32x4){0.f, 0.f, 0.f, 0.f};   // even / odd k partials
      DP_MARK(8);
      const float* gr = dgs + ((lane_ & 3) % RG) * GS + wave * GKW + ((lane_ >> 2) & 1) * GKS;   // rows >= RG alias
#pragma unroll
      for (int q4 = 0; q4 < ((ASR_DP_ABL & 512) ? 1 : GKS / 4); ++q4) {      // bit 512 (measurement): one quad of the dX product
        const float4 b = *reinterpret_cast<const float4*>(gr + 4 * q4);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wx[4 * q4], b.x, acc, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_4x4x1f32(wx[4 * q4 + 1], b.y, accb, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wx[4 * q4 + 2], b.z, acc, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_4x4x1f32(wx[4 * q4 + 3], b.w, accb, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += accb[i];
      float* pp = part + (wave * 64 + lane_) * 5;
#pragma unroll
      for (int i = 0; i < 4; ++i) pp[i] = acc[i];
    }
    __syncthreads();
    float vdx = 0.f, vfb = 0.f;               // dX of (column ci = tid >> 2, row = tid & 3), threads tid < 128; FB: + dl W_out
    if (tid_ < 128) {
      // (column index ci = tid>>2: 0..15 z, 16..31 ctx; row = tid&3): lanes 4*(2*cg + ks) + row, register ci&3
      const int ci = tid_ >> 2, row = tid_ & 3, cg = ci >> 2, ii = ci & 3;
#pragma unroll
      for (int w2 = 0; w2 < 8; ++w2)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) vdx += part[(w2 * 64 + 4 * (2 * cg + ks) + row) * 5 + ii];
    }
    if (FB) {
      // ---- (g2) d(emb_s) of my 4 embedding columns, my share of d(probabilities)
      float* fbr = dDp;                        // [8 waves][4 rows] wave sums; [32..47] d(emb)[row][ec]; [64..207] dl of the 4 rows
      {
        float pr[4] = {0.f, 0.f, 0.f, 0.f};
        const int kcg = 64 * (wave >> 2) + lane_;
        const bool kok = 16 * kcg < GK;
        const float* dg = dgs + (kok ? 16 * kcg : 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float4 x = *reinterpret_cast<const float4*>(dg + (r % RG) * GS + 4 * i);      // rows >= RG alias (never stored)
            pr[r] += wfe[i].x * x.x + wfe[i].y * x.y + wfe[i].z * x.z + wfe[i].w * x.w;
          }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = wave_sum_dpp(kok ? pr[r] : 0.f);
          if (lane_ == 0) fbr[wave * 4 + r] = t;
        }
      }
      __syncthreads();
      if (tid_ < 16) {
        const int row = tid_ & 3, ec = tid_ >> 2, b = r0 + row;
        const float v = (fbr[ec * 4 + row] + fbr[(ec + 4) * 4 + row]) * fb_mask;
        if (row < RG && b < nb) a.G[((int64_t)s * B + b) * KX + DD + OO + 4 * slice + ec] = v;
        fbr[32 + row * 4 + ec] = v;
      }
      __syncthreads();
      if (tid_ < 144) {
        const int row = tid_ / 36;
        const float* dm = fbr + 32 + row * 4;
        const float pdp = dm[0] * ereg[0] + dm[1] * ereg[1] + dm[2] * ereg[2] + dm[3] * ereg[3];
        word_store(xg + BX_P + (slot * 32 + slice) * 144 + tid_, pdp, bit);
      }
      // ---- (g3) slice 0: sum of the 32 shares, softmax backward, publish dl
      if (slice == 0) {
        constexpr int NPQ = 3;                 // 32 x 36 quads over 512 threads
        const unsigned pbase = (unsigned)((xg - a.xch) + BX_P + slot * 32 * 144) * 4u;
        unsigned off[NPQ];
        u4v v[NPQ];
#pragma unroll
        for (int i = 0; i < NPQ; ++i) {
          const int q = tid_ + DP_NT * i;
          off[i] = pbase + (unsigned)(q < 32 * 36 ? q : 0) * 16u;
        }
        poll_quads<NPQ, true>(xrs, off, bit, v, a.ctrl, aborted, 25u);
        // every thread is done with the gathered dgates: dgs becomes scratch for the 32 x 144 shares - with two rows per
        // group that is more than the two gathered rows hold (4 608 > 4 104 floats) and the tail lands in `part`, which
        // follows dgs in the LDS plan and is idle here (its K-partials of the dX product were summed before (g2))
        static_assert(32 * 144 <= RG * BM::GS + 8 * 64 * 5, "scratch of the feedback shares");
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NPQ; ++i) {
          const int q = tid_ + DP_NT * i;
          if (q < 32 * 36)
            *reinterpret_cast<float4*>(dgs + 4 * q) = make_float4(__uint_as_float(v[i].x), __uint_as_float(v[i].y),
                                                                  __uint_as_float(v[i].z), __uint_as_float(v[i].w));
        }
        __syncthreads();
        if (wave < 4) {
          float dp = 0.f;
          if (lane_ < 36) {
#pragma unroll 8
            for (int src = 0; src < 32; ++src) dp += dgs[src * 144 + wave * 36 + lane_];
          }
          const float pdot = wave_sum_dpp(fb_prob * dp);
          const float dl = a.fb_scale * fb_prob * (dp - pdot);          // 0 beyond V and for rows beyond the batch
          const int b = r0 + wave;
          if (lane_ < a.V && wave < RG && b < nb) a.dlfb[((int64_t)(s - 1) * B + b) * a.V + lane_] = dl;
          if (lane_ < 36) word_store(xg + BX_L + slot * 144 + wave * 36 + lane_, dl, bit);
        }
      }
      // ---- (g4) dl W_out for my columns
      {
        const unsigned lbase = (unsigned)((xg - a.xch) + BX_L + slot * 144) * 4u;
        unsigned off[1] = {lbase + (unsigned)(tid_ < 36 ? tid_ : 0) * 16u};
        u4v v[1];
        poll_quads<1, true>(xrs, off, bit, v, a.ctrl, aborted, 26u);
        if (tid_ < 36)
          *reinterpret_cast<float4*>(fbr + 64 + 4 * tid_) = make_float4(__uint_as_float(v[0].x), __uint_as_float(v[0].y),
                                                                         __uint_as_float(v[0].z), __uint_as_float(v[0].w));
      }
      __syncthreads();
      if (tid_ < 128) {
        const int ci = tid_ >> 2, row = tid_ & 3;
        const float* dl = fbr + 64 + row * 36;
        float acc = 0.f;
        for (int v = 0; v < a.V; ++v) acc += dl[v] * wos[v * 32 + ci];
        vfb = acc;
      }
    }
    if (tid_ < 128) {
      const int ci = tid_ >> 2, row = tid_ & 3;
      const float v = vdx;
      if (ci < 16) {
        dxz = v + vfb;                                            // gradient wrt z_{s-1} of my unit (same thread as in (f))
      } else if (ci - 16 < OU) {
        const int col = OU * slice + ci - 16, b = r0 + row;
        const float tot = gc + v * xm + vfb;                  // total d(ctx_{s-1}) = output layer (+ feedback) + masked cell input
        if (row < RG && b < nb) a.G[((int64_t)s * B + b) * KX + DD + col] = tot;
        word_store(xg + BX_C + (((n + 1) & 1) * 4 + row) * 512 + col, tot, tag_bit_of_step(n + 1));
      }
    }
    DP_MARK(9);
#undef TRS
    prefetchB(s - 1, zv);
  }
  // ---------------------------------------------------------------- epilogue: sequence-long accumulators
  __syncthreads();
  int ze;
  asm volatile("v_mov_b32 %0, 0" : "=v"(ze));
  const int lane_e = lane + ze, a_le = lane_e & 15, q4e = lane_e >> 4;
  const bool sc_oke = a_le < AU;
  const int acole = AU * slice + (sc_oke ? a_le : 0);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int tile = wave + 8 * it;
    if (FPT * tile < TpP) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int b = r0 + (4 * q4e + i) % RG, t = FPT * tile + (4 * q4e + i) / RG;
        if (sc_oke && b < nb && t < Tp) a.dP[((int64_t)b * Tp + t) * AA + acole] = dPs[(tile * 4 + i) * 64 + lane_e];
      }
    }
  }
  if (sc_oke) atomicAdd(a.dgvec_part + (int64_t)r0 * AA + acole, dgl);
  {
    const int c = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int al = 4 * (lane >> 4) + i;
      if (c < C && al < AU) atomicAdd(a.dwatt_part + ((int64_t)r0 * AA + AU * slice + al) * C + c, acc_watt[i]);
    }
  }
  {
    // lane (channel c = lane>>2, jj = lane&3), register i: tap j0 + 4i + jj of tiles aq (j0 = 16 aq) and aq + 8; every wave
    // holds a partial sum over its share of the frames
    const int c = lane_e >> 2, jj = lane_e & 3;
    if (c < C && ab_ok) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j0 = 16 * aq + 4 * i + jj, j1 = j0 + 16 * PPR;
        if (j0 < taps) atomicAdd(a.dconv_part + ((int64_t)ab * C + c) * taps + j0, acc_cv0[i]);
        if (j1 < taps) atomicAdd(a.dconv_part + ((int64_t)ab * C + c) * taps + j1, acc_cv1[i]);
      }
    }
  }
}

// M[s][b][c][t] = sum_a U[a][c] g[a] S[s][b][t][a]^2  (forward data only).  256 threads; a workgroup stages UG once and
// then walks (b, s) pairs with a grid stride (one pair per workgroup re-staged 32 KB of UG 3 232 times); wave w owns
// the 16-frame tiles w, w+4, ...; contraction on the 16x16x4 MFMA with the k-permutation of common.h (a lane's float4
// of S feeds 4 MFMAs, alternating between two accumulators).  LDS: UG[a][16] = U[a][c] g[a], zero beyond C.
template <int AA>
__global__ __launch_bounds__(256) void att_m_kernel(int B, int nb, int L, int Tp, int C, const float* __restrict__ S,
                                                    const float* __restrict__ watt, const float* __restrict__ gvec,
                                                    float* __restrict__ Mf) {
  __shared__ __attribute__((aligned(16))) float UG[AA * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // unguarded, batched loads (a load under `c < C ?` is waited for on the spot: 32 serial L2 round trips per workgroup)
  for (int i0 = 0; i0 < AA * 16; i0 += 256 * 8) {
    float wv[8], gq[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + 256 * k + tid, aa = (i >> 4) < AA ? (i >> 4) : AA - 1, c = i & 15;
      wv[k] = watt[(int64_t)aa * C + (c < C ? c : 0)];
      gq[k] = gvec[aa];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + 256 * k + tid;
      if (i < AA * 16) UG[i] = (i & 15) < C ? wv[k] * gq[k] : 0.f;
    }
  }
  __syncthreads();
  const int r = lane & 15, q = lane >> 4;
  // work items = (pair, 16-frame tile), dealt to the WAVES of the grid: with the tiles of one pair dealt to the four waves of
  // a workgroup, T' = 100 (seven tiles) ran 2 : 2 : 2 : 1
  const int ntile = (Tp + 15) / 16;
  for (int64_t item = (int64_t)blockIdx.x * 4 + wave; item < (int64_t)nb * L * ntile; item += (int64_t)gridDim.x * 4) {
    const int pair = (int)(item / ntile), tile = (int)(item - (int64_t)pair * ntile);
    const int s = pair / nb, b = pair - s * nb;
    const float* Sb = S + ((int64_t)s * B + b) * Tp * AA;
    {
      const int t = 16 * tile + r;
      const float* row = Sb + (int64_t)(t < Tp ? t : Tp - 1) * AA + 4 * q;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, accb = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
      for (int kq = 0; kq < AA / 16; ++kq) {
        const float4 sv = *reinterpret_cast<const float4*>(row + 16 * kq);
        const float* ub = UG + (16 * kq + 4 * q) * 16 + r;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sv.x * sv.x, ub[0], acc, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_16x16x4f32(sv.y * sv.y, ub[16], accb, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sv.z * sv.z, ub[32], acc, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_16x16x4f32(sv.w * sv.w, ub[48], accb, 0, 0, 0);
      }
      // D: lane holds frames 16*tile + 4q + i of channel r
      if (r < C) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int tt = 16 * tile + 4 * q + i;
          if (tt < Tp) Mf[(((int64_t)s * B + b) * C + r) * Tp + tt] = acc[i] + accb[i];
        }
      }
    }
  }
}

// G[s][b][D+O+e] *= xmask[s][b][O+e]  (dropout on the embedding part of the cell input, model.py:284-285)
__global__ void mask_emb_kernel(int L, int B, int nb, int D, int O, int E, const float* __restrict__ xmask,
                                float* __restrict__ G) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)L * nb * E) return;
  const int e = (int)(idx % E);
  const int64_t rb = idx / E;
  const int b = (int)(rb % nb), s = (int)(rb / nb);
  G[((int64_t)s * B + b) * (D + O + E) + D + O + e] *= xmask[((int64_t)s * B + b) * (O + E) + O + e];
}

template <int DD, int AA, int OO, int EE, int RG = 4, int TPM = DP_TPM, bool FB = false>
int launch_dec_bwd(const DecPersistBwdArgs& a, hipStream_t stream) {
  const BwdLds ld = bwd_lds_plan<DD, AA, OO, RG, TPM>(a.Tp, a.C, a.K, FB ? a.V : 0);
  size_t lds = (size_t)ld.total * sizeof(float);
  if (lds > 160 * 1024) return ASR_E_SHAPE;                          // must fit ...
  if (lds <= 82 * 1024) lds = 82 * 1024 + 64;                        // ... and must force one workgroup per CU
  hipError_t e = hipFuncSetAttribute((const void*)dec_persist_bwd_kernel<DD, AA, OO, EE, RG, TPM, FB>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((dec_persist_bwd_kernel<DD, AA, OO, EE, RG, TPM, FB>), dim3(256), dim3(DP_NT), lds, stream, a);
  return 0;
}

}  // namespace

bool asr_persist_device_ok();

namespace {

int dec_fwd_persist_impl(const asr_dec_fwd_t* p, const asr_dec_feedback_t* f, void* xch, void* ctrl, hipStream_t stream,
                         bool fault = false) {
  if (!p || !xch || !ctrl || !p->P || !p->Q || !p->bo || !p->wcat || !p->bcat || !p->wdec || !p->convw || !p->watt ||
      !p->gvec || !p->w0 || !p->X || !p->gates || !p->cstate || !p->fconv || !p->S || !p->energy || !p->ws)
    return ASR_E_ARG;
  if (p->B <= 0 || p->nb <= 0 || p->nb > p->B || p->Tp <= 0 || p->L <= 0) return ASR_E_ARG;
  if (p->xmask && !p->Xd) return ASR_E_ARG;
  if (f) {
    if ((f->mode != 1 && f->mode != 2) || !f->w_out || !f->emb || !f->logits || !f->pred || !f->fed || f->V <= 0)
      return ASR_E_ARG;
    if (f->mode == 2 && !f->probs) return ASR_E_ARG;
    if (f->tokens && (f->mode != 1 || !f->teacher || f->ld_tokens < p->L || f->eos >= 0)) return ASR_E_ARG;
    if (f->V > 64) return ASR_E_SHAPE;
  }
  const bool cfg2 = p->D == 512 && p->A == 512 && p->O == 512 && p->E == 128;
  const bool cfg1 = p->D == 320 && p->A == 320 && p->O == 320 && p->E == 128;
  if (!cfg1 && !cfg2) return ASR_E_SHAPE;
  const int TpP = (p->Tp + 3) & ~3;
  if (p->Tp <= 0 || p->C <= 0 || p->C > 12 || p->K < 0 || p->K > DP_KMAX) return ASR_E_SHAPE;
  // geometry: 4 utterances per group while the conv features of 4 rows fit the z / f poll (2 quads per thread: T' <= 102
  // at 10 channels), else 2 utterances per group on 16 CUs each (T' <= 256), teacher-forced and free-running alike
  using G4 = DecGeo<4, DP_TPM>;
  using G2 = DecGeo<2, 256>;
  const bool geo4 = p->Tp <= G4::TPM && 4 * p->C * (TpP / 4) <= G4::NFQ * DP_NT;
  const bool geo2 = !geo4 && p->Tp <= G2::TPM && 2 * p->C * (TpP / 4) <= G2::NFQ * DP_NT;
  if (!geo4 && !geo2) return ASR_E_SHAPE;
  if (!asr_persist_device_ok()) return ASR_E_SHAPE;
  const int B = p->B, Tp = p->Tp, A = p->A, D = p->D, O = p->O, E = p->E, C = p->C, KX = D + O + E;
  const int rows_per_launch = geo4 ? 32 : 16;
  for (int rb = 0; rb < p->nb; rb += rows_per_launch) {
    hipError_t e = persist_reset(xch, ctrl, (size_t)8 * (geo4 ? G4::X_GROUP : G2::X_GROUP) * sizeof(float), stream);
    if (e != hipSuccess) return (int)e;
    DecPersistArgs a;
    a.B = B; a.nb = p->nb - rb < rows_per_launch ? p->nb - rb : rows_per_launch; a.Tp = Tp; a.C = C; a.K = p->K; a.L = p->L;
    a.scaling = p->scaling;
    a.P = p->P + (int64_t)rb * Tp * A; a.Q = p->Q + (int64_t)rb * Tp * O; a.bo = p->bo; a.wcat = p->wcat;
    a.bcat = p->bcat; a.wdec = p->wdec; a.convw = p->convw; a.watt = p->watt; a.gvec = p->gvec;
    a.w0 = p->w0 + (int64_t)rb * Tp; a.xmask = p->xmask ? p->xmask + (int64_t)rb * (O + E) : nullptr;
    a.X = p->X + (int64_t)rb * KX; a.Xd = p->Xd ? p->Xd + (int64_t)rb * KX : nullptr;
    a.gates = p->gates + (int64_t)rb * 4 * D; a.cstate = p->cstate + (int64_t)rb * D;
    a.fconv = p->fconv + (int64_t)rb * C * Tp; a.S = p->S + (int64_t)rb * Tp * A;
    a.energy = p->energy + (int64_t)rb * Tp; a.ws = p->ws + (int64_t)rb * Tp;
    a.xch = (float*)xch; a.ctrl = persist_launch_words(ctrl);
    a.fb_mode = 0; a.V = 0; a.eos = -1; a.fb_scale = 1.f; a.w_out = a.b_out = a.emb = nullptr; a.logits = a.probs = nullptr;
    a.pred = a.fed = nullptr;
    a.tok = nullptr; a.ldtok = 0; a.tf = nullptr;
    int rc;
    if (f) {
      a.fb_mode = f->mode; a.V = f->V; a.eos = f->eos; a.fb_scale = f->scaling; a.w_out = f->w_out; a.b_out = f->b_out; a.emb = f->emb;
      a.logits = f->logits + (int64_t)rb * f->V; a.probs = f->probs ? f->probs + (int64_t)rb * f->V : nullptr;
      a.pred = (long long*)f->pred + rb; a.fed = (long long*)f->fed + rb;
      if (f->tokens) { a.tok = (const long long*)f->tokens + (int64_t)rb * f->ld_tokens; a.ldtok = f->ld_tokens; a.tf = f->teacher; }
      if (geo4) rc = cfg2 ? launch_dec_fwd<512, 512, 512, 128, true>(a, stream) : launch_dec_fwd<320, 320, 320, 128, true>(a, stream);
      else rc = cfg2 ? launch_dec_fwd<512, 512, 512, 128, true, 2, 256>(a, stream)
                     : launch_dec_fwd<320, 320, 320, 128, true, 2, 256>(a, stream);
    } else if (fault) {          // the FAULT instantiation (persist.h) exists for the cfg-2 widths in the 4-row geometry only
      if (!geo4 || !cfg2) return ASR_E_SHAPE;
      rc = launch_dec_fwd<512, 512, 512, 128, false, 4, DP_TPM, true>(a, stream);
    } else if (geo4) {
      rc = cfg2 ? launch_dec_fwd<512, 512, 512, 128, false>(a, stream) : launch_dec_fwd<320, 320, 320, 128, false>(a, stream);
    } else {
      rc = cfg2 ? launch_dec_fwd<512, 512, 512, 128, false, 2, 256>(a, stream)
                : launch_dec_fwd<320, 320, 320, 128, false, 2, 256>(a, stream);
    }
    if (rc) return rc;
  }
  ASR_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// Whole teacher-forced decoder sequence (steps 0..L-1) in one launch per block of 32 rows.  Same operands and
// results as asr_dec_seq_fwd(p, 0, L) except that Dproj is not written.  Returns ASR_E_SHAPE when the fast path
// does not apply (the caller then uses asr_dec_seq_fwd).  xch / ctrl: asr_persist_scratch_bytes() (zeroed here on the stream: up to 3.6 MB and the 64 bytes of per-launch words).
extern "C" int asr_dec_seq_fwd_persist(const asr_dec_fwd_t* p, void* xch, void* ctrl, asr_stream_t stream_) {
  return dec_fwd_persist_impl(p, nullptr, xch, ctrl, (hipStream_t)stream_);
}
// Test entry: the same launch on the FAULT instantiation of the kernel (csrc/persist.h) - a producer goes silent, the
// bounded spins expire in a millisecond, the launch aborts by itself.  cfg-2 widths, T' <= 102; ASR_E_SHAPE otherwise.
extern "C" int asr_dec_seq_fwd_persist_fault(const asr_dec_fwd_t* p, void* xch, void* ctrl, asr_stream_t stream_) {
  return dec_fwd_persist_impl(p, nullptr, xch, ctrl, (hipStream_t)stream_, true);
}

// Free-running variant: steps 1..L-1 take their embedding input from the previous step's logits (f->mode 1: row of the
// argmax token, 2: softmax(f->scaling * logits) @ emb).  The caller provides X[0] / Xd[0] (embedding of <BOS>) and
// fed[0]; the kernel writes logits[s], pred[s] for s < L-1, fed[s] and the embedding columns of X[s] / Xd[s] for
// s >= 1, probs[s] for s < L-1 (mode 2).  logits / pred of the last step: asr_dec_feedback_fwd(mode 3) on X[L].
extern "C" int asr_dec_seq_fwd_persist_free(const asr_dec_fwd_t* p, const asr_dec_feedback_t* f, void* xch, void* ctrl,
                                            asr_stream_t stream_) {
  if (!f) return ASR_E_ARG;
  return dec_fwd_persist_impl(p, f, xch, ctrl, (hipStream_t)stream_);
}

// Persistent fast path of asr_dec_seq_bwd(q, 0, L) for sequences produced by the teacher-forced forward (either
// forward path): same results in G[:, :, D:], dgates, dD, dP and the partial-sum buffers (dgvec_part / dwatt_part /
// dconv_part hold the sums in other rows than the per-step path does; the caller reduces over rows either way).
// mbuf: scratch [L][B][C][Tp].  dwext / dwraw / dfpart / dcell of q are not used.  Returns ASR_E_SHAPE when the
// fast path does not apply.
static int dec_bwd_persist_impl(const asr_dec_bwd_t* q, const asr_dec_feedback_bwd_t* fb, float* mbuf, void* xch, void* ctrl,
                                asr_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!q || !mbuf || !xch || !ctrl) return ASR_E_ARG;
  if (fb && (!fb->w_out || !fb->emb || !fb->probs || !fb->dlfb || fb->V <= 0)) return ASR_E_ARG;
  if (fb && (fb->V > 36 || q->f.E != 128)) return ASR_E_SHAPE;
  const asr_dec_fwd_t* p = &q->f;
  if (!p->Q || !p->wcat || !p->convw || !p->watt || !p->gvec || !p->w0 || !p->gates || !p->cstate || !p->fconv ||
      !p->S || !p->ws || !q->wcatT || !q->wdecT || !q->G || !q->dgates || !q->dD || !q->dP || !q->dgvec_part ||
      !q->dwatt_part || !q->dconv_part)
    return ASR_E_ARG;
  if (p->B <= 0 || p->nb <= 0 || p->nb > p->B || p->Tp <= 0 || p->L <= 0) return ASR_E_ARG;
  const bool cfg2 = p->D == 512 && p->A == 512 && p->O == 512 && p->E == 128;
  const bool cfg1 = p->D == 320 && p->A == 320 && p->O == 320 && p->E == 128;
  if (!cfg1 && !cfg2) return ASR_E_SHAPE;
  const int TpP = (p->Tp + 3) & ~3;
  if (p->C <= 0 || p->C > 16 || p->K < 0 || p->K > DP_KMAX) return ASR_E_SHAPE;
  // geometry as in the forward: 4 utterances per group while the per-thread prefetch registers cover the conv features
  // and M of 4 rows (T' <= 102 at 10 channels), else 2 utterances per group on 16 CUs each (T' <= 256)
  const bool geo4 = p->Tp <= DP_TPM && p->C * TpP <= 2 * DP_NT && 4 * p->C * TpP <= 8 * DP_NT;
  const bool geo2 = !geo4 && p->Tp <= 256 && p->C * TpP <= 5 * DP_NT && 2 * p->C * TpP <= 10 * DP_NT;
  if (!geo4 && !geo2) return ASR_E_SHAPE;
  if (!asr_persist_device_ok()) return ASR_E_SHAPE;
  const int B = p->B, Tp = p->Tp, A = p->A, D = p->D, O = p->O, E = p->E, C = p->C, KX = D + O + E;
  const int taps = 2 * p->K + 1;
  const int rows_per_launch = geo4 ? 32 : 16;
  for (int rb = 0; rb < p->nb; rb += rows_per_launch) {
    const int nbb = p->nb - rb < rows_per_launch ? p->nb - rb : rows_per_launch;
    hipError_t e = persist_reset(xch, ctrl, (size_t)8 * (geo4 ? BwdX<DP_TPM>::GROUP : BwdX<256>::GROUP) * sizeof(float), stream);
    if (e != hipSuccess) return (int)e;
    if (cfg2)
      hipLaunchKernelGGL((att_m_kernel<512>), dim3(nbb * p->L < 2048 ? nbb * p->L : 2048), dim3(256), 0, stream, B, nbb, p->L, Tp, C,
                         p->S + (int64_t)rb * Tp * A, p->watt, p->gvec, mbuf + (int64_t)rb * C * Tp);
    else
      hipLaunchKernelGGL((att_m_kernel<320>), dim3(nbb * p->L < 2048 ? nbb * p->L : 2048), dim3(256), 0, stream, B, nbb, p->L, Tp, C,
                         p->S + (int64_t)rb * Tp * A, p->watt, p->gvec, mbuf + (int64_t)rb * C * Tp);
    DecPersistBwdArgs a;
    a.B = B; a.nb = nbb; a.Tp = Tp; a.C = C; a.K = p->K; a.L = p->L; a.scaling = p->scaling;
    a.Q = p->Q + (int64_t)rb * Tp * O; a.wcatT = q->wcatT; a.wdecT = q->wdecT; a.convw = p->convw; a.watt = p->watt;
    a.gvec = p->gvec; a.w0 = p->w0 + (int64_t)rb * Tp; a.xmask = p->xmask ? p->xmask + (int64_t)rb * (O + E) : nullptr;
    a.gates = p->gates + (int64_t)rb * 4 * D; a.cstate = p->cstate + (int64_t)rb * D;
    a.S = p->S + (int64_t)rb * Tp * A; a.fconv = p->fconv + (int64_t)rb * C * Tp; a.ws = p->ws + (int64_t)rb * Tp;
    a.Mf = mbuf + (int64_t)rb * C * Tp; a.dws = q->dws ? q->dws + (int64_t)rb * Tp : nullptr;
    a.dbg = q->dfpart;
    a.G = q->G + (int64_t)rb * KX; a.dgates = q->dgates + (int64_t)rb * 4 * D; a.dD = q->dD + (int64_t)rb * A;
    a.dP = q->dP + (int64_t)rb * Tp * A; a.dgvec_part = q->dgvec_part + (int64_t)rb * A;
    a.dwatt_part = q->dwatt_part + (int64_t)rb * A * C; a.dconv_part = q->dconv_part + (int64_t)rb * C * taps;
    a.xch = (float*)xch; a.ctrl = persist_launch_words(ctrl);
    a.V = 0; a.fb_scale = 1.f; a.w_out = a.emb = a.probs = nullptr; a.dlfb = nullptr;
    int rc;
    if (fb) {
      a.V = fb->V; a.fb_scale = fb->scaling; a.w_out = fb->w_out; a.emb = fb->emb;
      a.probs = fb->probs + (int64_t)rb * fb->V; a.dlfb = fb->dlfb + (int64_t)rb * fb->V;
      rc = geo4 ? (cfg2 ? launch_dec_bwd<512, 512, 512, 128, 4, DP_TPM, true>(a, stream)
                        : launch_dec_bwd<320, 320, 320, 128, 4, DP_TPM, true>(a, stream))
                : (cfg2 ? launch_dec_bwd<512, 512, 512, 128, 2, 256, true>(a, stream)
                        : launch_dec_bwd<320, 320, 320, 128, 2, 256, true>(a, stream));
    } else {
      rc = geo4 ? (cfg2 ? launch_dec_bwd<512, 512, 512, 128>(a, stream) : launch_dec_bwd<320, 320, 320, 128>(a, stream))
                : (cfg2 ? launch_dec_bwd<512, 512, 512, 128, 2, 256>(a, stream)
                        : launch_dec_bwd<320, 320, 320, 128, 2, 256>(a, stream));
    }
    if (rc) return rc;
  }
  ASR_CHECK_LAUNCH();
  // embedding part of dX: G[s][:, D+O:] += dgates[s] Wcat[:, D+O:] (masked), batched over the steps.  Teacher-forced: not
  // recurrent, all L steps here.  Free-running: the kernel formed it for steps >= 1 (it feeds the previous step's logits);
  // step 0 (the <BOS> embedding) is left for here.
  // (fp32-equivalent bf16x6 products, as everything else the decoder kernels compute is exact fp32)
  const int Lg = fb ? 1 : p->L;
  int rc;
  if (p->nb == B)      // all rows: one GEMM over the L*B rows, K = 4D split so that the few output tiles fill the chip
    rc = asr_gemm_f32(0, 0, (int64_t)Lg * B, E, 4 * D, q->dgates, 4 * D, p->wcat + D + O, KX, q->G + D + O, KX,
                      nullptr, 0, 1, 1, 0, 0, 0, 16, ASR_ARITH_BF16X6, stream_);
  else
    rc = asr_gemm_f32(0, 0, p->nb, E, 4 * D, q->dgates, 4 * D, p->wcat + D + O, KX, q->G + D + O, KX, nullptr, 0, 1,
                      Lg, (int64_t)B * 4 * D, 0, (int64_t)B * KX, 4, ASR_ARITH_BF16X6, stream_);
  if (rc) return rc;
  if (p->xmask) {
    const int64_t n = (int64_t)Lg * p->nb * E;
    hipLaunchKernelGGL(mask_emb_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, Lg, B, p->nb, D, O, E,
                       p->xmask, q->G);
    ASR_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int asr_dec_seq_bwd_persist(const asr_dec_bwd_t* q, float* mbuf, void* xch, void* ctrl, asr_stream_t stream) {
  return dec_bwd_persist_impl(q, nullptr, mbuf, xch, ctrl, stream);
}

// Backward of a FREE-RUNNING sequence with the smooth-embedding feedback (asr_dec_seq_fwd_persist_free, mode 2): as
// asr_dec_seq_bwd_persist, plus the gradient that every step sends into the previous step's logits through
// emb_s = softmax(k logit_{s-1}) @ E (kernel template FB).  fb->dlfb [L][B][V] (zero-filled by the caller) receives that
// gradient per step; G[s][:, D+O:] holds d(emb_s) for every s.  Both geometries (4 rows per group for T' <= 100 at 10
// channels, 2 rows per group up to T' = 256); beyond that ASR_E_SHAPE and the caller uses the per-step kernels +
// asr_dec_feedback_bwd.
extern "C" int asr_dec_seq_bwd_persist_free(const asr_dec_bwd_t* q, const asr_dec_feedback_bwd_t* fb, float* mbuf, void* xch,
                                            void* ctrl, asr_stream_t stream) {
  if (!fb) return ASR_E_ARG;
  return dec_bwd_persist_impl(q, fb, mbuf, xch, ctrl, stream);
}
