// gemm.hip — fp32 GEMMs on the f32-input MFMA of gfx950.
//   asr_gemm_f32        128x128x32 LDS-tiled, v_mfma_f32_32x32x2_f32, 4 waves x (2x2) 32x32 tiles,
//                       register-prefetched next K tile, optional split-K with f32 atomics.
//   asr_gemm_skinny_f32 M = batch rows of a sequential chain; 16 columns per workgroup.
//   asr_colsum_f32      bias gradients.
// Replaces the torch mm/addmm/bmm calls behind nn.Linear / nn.LSTM input projections on the
// reference path (model.py:67-68,80,93-94,144,163,293 and their autograd backward).
#include <cstdlib>
#include <type_traits>
#include "common.h"

#ifndef ASR_GEMM_BF3_TOUCH      /* L2 warm-up distance of the split-bf16 kernel in K tiles (2, 4, 6, 10 measured within 5 %: tools/gemm_cold_sweep.py; the kernel is bound by operand traffic at 32 flop/byte per 128x128 tile, not by latency) */
#define ASR_GEMM_BF3_TOUCH 2
#endif
#ifndef ASR_GB_ABL               /* measurement only: 1 no products, 2 no operand loads after the first tile, 4 no LDS staging, 8 no epilogue stores */
#define ASR_GB_ABL 0
#endif
#ifndef ASR_GLDS_CLOBBER_M0
#define ASR_GLDS_CLOBBER_M0 0
#endif
#ifndef ASR_GEMM_SETPRIO
#define ASR_GEMM_SETPRIO 1
#endif
namespace {

constexpr int BM = 128, BN = 128, BK = 32;
#ifndef ASR_GEMM_SMALL_OCC      /* waves per SIMD the 64 x 64-tile kernel is compiled for (= workgroups per CU) */
#define ASR_GEMM_SMALL_OCC 1      /* (3: 168 registers with 11-21 spilled, 4: 52-113 spilled; left to itself hipcc takes 195-229 = two workgroups per CU) */
#endif
#ifndef ASR_GEMM_TOUCH
#define ASR_GEMM_TOUCH 2      /* L2 warm-up distance in K tiles (0 = off); 2 measured best of 2,3,5,8 */
#endif

// A "stored matrix" view: element (r, c) at p[r*ld + c], valid for r < R, c < Cn.
struct MatView {
  const float* p;
  int64_t ld, R, Cn;
  bool vec;  // ld % 4 == 0 and base 16B aligned -> float4 loads allowed
};

__device__ __forceinline__ float4 load4_guard(const MatView& m, int64_t r, int64_t c) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r >= m.R) return v;
  const float* q = m.p + r * m.ld + c;
  if (m.vec && c + 3 < m.Cn) return *reinterpret_cast<const float4*>(q);
  if (c < m.Cn) v.x = q[0];
  if (c + 1 < m.Cn) v.y = q[1];
  if (c + 2 < m.Cn) v.z = q[2];
  if (c + 3 < m.Cn) v.w = q[3];
  return v;
}

// LDS image of an operand tile is always [BK][rows(+pad)] (k-major planes, rows contiguous) so
// that the MFMA fragment read (32 consecutive rows at one k) is a conflict-free ds_read_b32.
//   KC operand (k contiguous in memory, e.g. A[M][K]):   stride 129, scattered b32 writes
//   MC operand (row index contiguous, e.g. A stored [K][M]): stride 128, b128 writes
template <bool KC>
struct TileCfg {
  static constexpr int S = KC ? 129 : 128;
};

// Fetch this thread's 4 float4 pieces of a 128 x 32 operand tile (rows x k) into registers.
template <bool KC>
__device__ __forceinline__ void tile_fetch(const MatView& m, int64_t row0, int64_t k0, float4 (&v)[4]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = t + 256 * i;
    if (KC) {  // stored [rows][K]: 8 float4 per row
      const int row = f >> 3, kc4 = f & 7;
      v[i] = load4_guard(m, row0 + row, k0 + 4 * kc4);
    } else {   // stored [K][rows]: 32 float4 per k
      const int k = f >> 5, m4 = f & 31;
      v[i] = load4_guard(m, k0 + k, row0 + 4 * m4);
    }
  }
}

template <bool KC>
__device__ __forceinline__ void tile_store(float* lds, const float4 (&v)[4]) {
  constexpr int S = TileCfg<KC>::S;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = t + 256 * i;
    if (KC) {
      const int row = f >> 3, kc4 = f & 7;
      float* d = lds + (4 * kc4) * S + row;
      d[0] = v[i].x; d[S] = v[i].y; d[2 * S] = v[i].z; d[3 * S] = v[i].w;
    } else {
      const int k = f >> 5, m4 = f & 31;
      *reinterpret_cast<float4*>(lds + k * S + 4 * m4) = v[i];
    }
  }
}

struct GemmArgs {
  MatView A, B;  // A: KC -> [M][K] else [K][M];  B: KC -> [N][K] else [K][N]
  float* C;
  int64_t ldc, M, N, K;
  const float* bias;
  int relu, accumulate, split_k;
  int64_t sA, sB, sC;
  int tiles_m, tiles_n;
  // gemm_bf3_kernel<..., 64, QUEUE = true> (asr_gemm_side_f32): the ticket counter the workgroups draw their (tile, K slice)
  // from, and the XCDs they may run on (bit x = XCC id x; a workgroup that lands elsewhere leaves at once)
  unsigned* queue;
  unsigned xcd_mask;
};


// Tile id -> (tile row, tile column).  Consecutive ids walk down GM tile rows, then step one tile column (groups of GM
// rows x all columns): the 32 workgroups an XCD runs at one time (its CUs take consecutive ids of the XCD's contiguous
// chunk) then cover a GM x (32 / GM) block of tiles instead of one tile row x 32 columns, i.e. 32 / GM + GM operand
// panels per K tile in the XCD's L2 instead of 33.  The operand fetch of the split-bf16 kernels runs at what ONE CU can
// pull through its memory path (~11 B / cycle from the Infinity Cache, ~30 from L2; they need 16 - 21 B / cycle), so L2
// hits are what the main loop's speed is made of.
#ifndef ASR_GEMM_GM
#define ASR_GEMM_GM 4
#endif
__device__ __forceinline__ void tile_coords(int tid, int tiles_m, int tiles_n, int gm_, int& tm, int& tn) {
  const int per = gm_ * tiles_n;
  const int grp = tid / per, in = tid - grp * per;
  const int first = grp * gm_;
  const int rows = tiles_m - first < gm_ ? tiles_m - first : gm_;
  tm = first + in % rows;
  tn = in / rows;
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  constexpr int SA = TileCfg<AKC>::S, SB = TileCfg<BKC>::S;
  __shared__ __attribute__((aligned(16))) float smem[BK * SA + BK * SB];
  float* As = smem;
  float* Bs = smem + BK * SA;

  // XCD-aware tile order: consecutive tile ids share an A row-panel; the hardware deals
  // workgroups round-robin over the 8 XCDs, so give each XCD a contiguous chunk of ids.
  const int ntile = g.tiles_m * g.tiles_n;
  int tid = blockIdx.x;
  {
    const int q = ntile >> 3, rmd = ntile & 7, xcd = tid & 7, idx = tid >> 3;
    tid = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + idx;
  }
  const int tm = tid / g.tiles_n, tn = tid % g.tiles_n;
  const int z = blockIdx.y;
  const int bz = z / g.split_k, kz = z % g.split_k;

  MatView A = g.A, B = g.B;
  A.p += bz * g.sA;
  B.p += bz * g.sB;
  float* C = g.C + bz * g.sC;

  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int64_t ktiles = (g.K + BK - 1) / BK;
  const int64_t per = (ktiles + g.split_k - 1) / g.split_k;
  const int64_t kt_begin = kz * per;
  const int64_t kt_end = kt_begin + per < ktiles ? kt_begin + per : ktiles;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // Fast operand path for interior tiles and full K tiles: one uniform base pointer per operand (advanced by the K
  // offset) + a 32-bit per-thread element offset computed once, i.e. 8 unguarded float4 loads per K tile and no
  // address arithmetic.  The guarded path costs ~120 VALU instructions and ~30 branches per K tile and wave, which
  // sit between the barrier and the first MFMA.  Edge tiles and the K tail keep the guarded path.
  const bool fast_a = A.vec && (AKC ? (m0 + BM <= A.R) : (m0 + BM <= A.Cn)) && A.R * A.ld < (int64_t)1 << 30;
  const bool fast_b = B.vec && (BKC ? (n0 + BN <= B.R) : (n0 + BN <= B.Cn)) && B.R * B.ld < (int64_t)1 << 30;
  const bool fast = fast_a && fast_b;
  unsigned offa[4], offb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = threadIdx.x + 256 * i;
    offa[i] = AKC ? (unsigned)((f >> 3) * A.ld + 4 * (f & 7)) : (unsigned)((f >> 5) * A.ld + 4 * (f & 31));
    offb[i] = BKC ? (unsigned)((f >> 3) * B.ld + 4 * (f & 7)) : (unsigned)((f >> 5) * B.ld + 4 * (f & 31));
  }
  const float* basea = AKC ? A.p + m0 * A.ld : A.p + m0;     // + k0 (KC) or + k0*ld (MC)
  const float* baseb = BKC ? B.p + n0 * B.ld : B.p + n0;
  float4 ra[4], rb[4];
  auto fetch = [&](int64_t kt2) {
    const int64_t k0 = kt2 * BK;
    if (fast && k0 + BK <= g.K) {
      const float* pa = basea + (AKC ? k0 : k0 * A.ld);
      const float* pb = baseb + (BKC ? k0 : k0 * B.ld);
#pragma unroll
      for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const float4*>(pa + offa[i]);
#pragma unroll
      for (int i = 0; i < 4; ++i) rb[i] = *reinterpret_cast<const float4*>(pb + offb[i]);
    } else {
      tile_fetch<AKC>(A, m0, k0, ra);
      tile_fetch<BKC>(B, n0, k0, rb);
    }
  };
  if (kt_begin < kt_end) fetch(kt_begin);
  // L2 warm-up: the register prefetch runs one K tile ahead, which does not cover an HBM first touch (operands of the
  // train step are cold: +15-25 % time on the weight-gradient shapes).  Each thread therefore also touches ONE
  // 128-byte line of the tile two further ahead (256 threads = the 2 x 128 lines of an A and a B tile); the value
  // is discarded (ASR_GEMM_TOUCH).
  const int tt = threadIdx.x & 127;
  const bool touch_a = threadIdx.x < 128;
  float touched = 0.f;
  const bool do_touch = kt_end - kt_begin > 16;      // short K loops (K <= 512) measured 5 % slower with it
  auto touch_tile = [&](int64_t ktt) {
    const MatView& m = touch_a ? A : B;
    const bool kc = touch_a ? AKC : BKC;
    const int64_t r0t = touch_a ? m0 : n0;
    int64_t rr, cc;
    if (kc) { rr = r0t + tt; cc = ktt * BK; }                       // [rows][K]: one 128-byte row segment per row
    else { rr = ktt * BK + (tt >> 2); cc = r0t + 32 * (tt & 3); }   // [K][rows]: four segments per k row
    rr = rr < m.R ? rr : m.R - 1;
    cc = cc < m.Cn ? cc : m.Cn - 1;
    touched = m.p[rr * m.ld + cc];
  };
  for (int64_t kt = kt_begin; kt < kt_end; ++kt) {
    tile_store<AKC>(As, ra);
    tile_store<BKC>(Bs, rb);
    __syncthreads();
    if (kt + 1 < kt_end) fetch(kt + 1);
#if ASR_GEMM_TOUCH
    asm volatile("" ::"v"(touched));                 // retire the previous touch (issued one tile ago)
    if (do_touch && kt + ASR_GEMM_TOUCH < kt_end) touch_tile(kt + ASR_GEMM_TOUCH);
#endif
    const float* ap = As + kh * SA + wm * 64 + l31;
    const float* bp = Bs + kh * SB + wn * 64 + l31;
    // transA (weight-gradient) shapes: raise the wave priority over the MFMA block, so that the co-resident
    // workgroup's loads do not take issue slots between the products: +4-7 % on those shapes with cold operands, but
    // -4 % on the NN / NT shapes, which keep the default
    if (ASR_GEMM_SETPRIO && !AKC) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const float a0 = ap[(2 * kk) * SA], a1 = ap[(2 * kk) * SA + 32];
      const float b0 = bp[(2 * kk) * SB], b1 = bp[(2 * kk) * SB + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (ASR_GEMM_SETPRIO && !AKC) __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }

  // epilogue: lane owns column n, rows (e&3) + 8*(e>>2) + 4*kh of each 32x32 tile
  if (m0 + BM <= g.M && n0 + BN <= g.N) {
    // interior tile: no bounds checks, 32-bit row offsets from one base pointer per 32x32 tile (the generic path
    // below costs ~30 VALU instructions per element, which is visible for short K)
    const unsigned ldc = (unsigned)g.ldc;
    const bool split = g.split_k > 1, accum = g.accumulate != 0, relu = g.relu != 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t n = n0 + wn * 64 + j * 32 + l31;
      const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float* base = C + (m0 + wm * 64 + i * 32 + 4 * kh) * g.ldc + n;
        if (split) {
#pragma unroll
          for (int e = 0; e < 16; ++e) atomicAdd(base + (unsigned)((e & 3) + 8 * (e >> 2)) * ldc, acc[i][j][e]);
        } else if (accum) {
          float old[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) old[e] = base[(unsigned)((e & 3) + 8 * (e >> 2)) * ldc];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float v = acc[i][j][e] + bv + old[e];
            if (relu) v = fmaxf(v, 0.f);
            base[(unsigned)((e & 3) + 8 * (e >> 2)) * ldc] = v;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float v = acc[i][j][e] + bv;
            if (relu) v = fmaxf(v, 0.f);
            base[(unsigned)((e & 3) + 8 * (e >> 2)) * ldc] = v;
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t n = n0 + wn * 64 + j * 32 + l31;
    if (n >= g.N) continue;
    const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        if (m >= g.M) continue;
        float v = acc[i][j][e];
        float* dst = C + m * g.ldc + n;
        if (g.split_k > 1) {
          atomicAdd(dst, v);
        } else {
          v += bv;
          if (g.accumulate) v += *dst;
          if (g.relu) v = fmaxf(v, 0.f);
          *dst = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ split-bf16 product
// Same tiling, operand fetch, split-K and epilogue as gemm_f32_kernel; the product runs on v_mfma_f32_32x32x16_bf16
// with both fp32 operands split in two bf16 terms while they are staged into LDS (x = hi + lo, hi = the upper 16 bits
// of the fp32 word, lo = bf16(x - hi) rounded to nearest: 16 significand bits) and three products hi*hi + hi*lo + lo*hi
// accumulated in fp32.  Dropped: lo*lo and the rounding of lo, <= 2^-16 relative per product (fp32: 2^-24), far inside
// the 1e-3 parity gate (tests/test_hip_parity.py::test_gemm_variants, test_cfg2_against_golden).  The bf16 pipe does
// 16x the MACs per cycle of the fp32 MFMA, so a K tile of 32 costs 24 MFMAs of 32 cycles per wave instead of 64 of 64.
// LDS images: per operand hi and lo, [128 rows][32 k] bf16 with 80-byte rows; a fragment (row l & 31, 8 consecutive k)
// is one ds_read_b128.  Operands stored k-contiguous ([rows][K]) are staged as before (a thread's float4 = 4 k of one
// row); operands stored row-contiguous ([K][rows]) use a 4 (k) x 4 (rows) register block per thread, read as four
// float4 at consecutive k, so that both kinds end up as 8-byte LDS writes of 4 consecutive k.
typedef __bf16 gbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gbf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned gu32x4 __attribute__((ext_vector_type(4)));
constexpr int BS = 40;                       // LDS row stride in bf16 (80 bytes)

__device__ __forceinline__ unsigned bf3g_hi2(float a, float b) {      // {hi(a), hi(b)}: upper halves of the two words
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ unsigned bf3g_lo2(float a, float b) {      // {lo(a), lo(b)}, rounded to nearest
  const float la = a - __uint_as_float(__float_as_uint(a) & 0xffff0000u);
  const float lb = b - __uint_as_float(__float_as_uint(b) & 0xffff0000u);
  const gbf16x2 p = {(__bf16)la, (__bf16)lb};
  return __builtin_bit_cast(unsigned, p);
}

// Three-term split (NT = 3, "bf16x6"): a = bf16(x), b = bf16(x - a), c = bf16(x - a - b), every conversion rounded to
// nearest (v_cvt_pk_bf16_f32).  Both differences are exact in fp32 and c needs at most 8 significand bits, so
// a + b + c == x exactly: the three terms are a lossless re-encoding of the fp32 operand.  With the six products
// aa' + ab' + ba' + ac' + ca' + bb' accumulated in fp32 the dropped terms (bc', cb', cc') are <= 2^-25 |x x'| each
// (|b| <= 2^-9 |x|, |c| <= 2^-17 |x|), i.e. below the rounding of an fp32 product: fp32-equivalent arithmetic on the bf16
// pipe at 6 MFMAs per product (the fp32-input MFMA costs the time of 16).  5.5 VALU instructions per element.
// (|x| within one bf16 ulp of FLT_MAX would round a to infinity; no operand of this path comes near.)
template <int NT>
__device__ __forceinline__ void bfn_split2(float x, float y, unsigned (&t)[NT]) {   // t[k] = {term_k(x), term_k(y)}
  if constexpr (NT == 2) {
    t[0] = bf3g_hi2(x, y);
    t[1] = bf3g_lo2(x, y);
  } else {
    const gbf16x2 pa = {(__bf16)x, (__bf16)y};
    const unsigned ua = __builtin_bit_cast(unsigned, pa);
    const float rx = x - __uint_as_float(ua << 16), ry = y - __uint_as_float(ua & 0xffff0000u);
    const gbf16x2 pb = {(__bf16)rx, (__bf16)ry};
    const unsigned ub = __builtin_bit_cast(unsigned, pb);
    const float sx = rx - __uint_as_float(ub << 16), sy = ry - __uint_as_float(ub & 0xffff0000u);
    t[0] = ua;
    t[1] = ub;
    t[2] = bf3g_hi2(sx, sy);           // exact: sx, sy have at most 8 significand bits
  }
}

// this thread's TS / 32 float4 pieces of a TS (rows) x 32 (k) operand tile, TS = 128 or 64.  KC (k-contiguous rows):
// piece i = 4 k of row (t + 256 i) >> 3.  MC (row-contiguous): k block t & 7, rows 4 (t >> 3) .. + 3 - at TS = 128 all four
// k of the block, at TS = 64 (16 row quads x 8 k blocks = 128 threads' worth) the two k 2 (t >> 7), + 1 of it
template <bool KC, int TS>
__device__ __forceinline__ void tile_fetch_bf3(const MatView& m, int64_t row0, int64_t k0, float4 (&v)[TS / 32]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < TS / 32; ++i) {
    if (KC) {
      const int f = t + 256 * i;
      v[i] = load4_guard(m, row0 + (f >> 3), k0 + 4 * (f & 7));
    } else if (TS == 128) {
      v[i] = load4_guard(m, k0 + 4 * (t & 7) + i, row0 + 4 * (t >> 3));
    } else {
      v[i] = load4_guard(m, k0 + 4 * (t & 7) + 2 * (t >> 7) + i, row0 + 4 * ((t >> 3) & 15));
    }
  }
}

template <bool KC, int NT, int TS>
__device__ __forceinline__ void tile_store_bf3(unsigned short* img, const float4 (&v)[TS / 32]) {   // NT images, TS * BS apart
  const int t = threadIdx.x;
  if (KC) {
#pragma unroll
    for (int i = 0; i < TS / 32; ++i) {
      const int f = t + 256 * i;
      const int o = (f >> 3) * BS + 4 * (f & 7);
      unsigned p0[NT], p1[NT];
      bfn_split2<NT>(v[i].x, v[i].y, p0);
      bfn_split2<NT>(v[i].z, v[i].w, p1);
#pragma unroll
      for (int k = 0; k < NT; ++k) *reinterpret_cast<uint2*>(img + k * TS * BS + o) = make_uint2(p0[k], p1[k]);
    }
  } else if constexpr (TS == 64) {
    // two consecutive k of four rows: one packed split and a 4-byte store per row and term
    const int o = 4 * ((t >> 3) & 15) * BS + 4 * (t & 7) + 2 * (t >> 7);
    const float r[4][2] = {{v[0].x, v[1].x}, {v[0].y, v[1].y}, {v[0].z, v[1].z}, {v[0].w, v[1].w}};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned p0[NT];
      bfn_split2<NT>(r[j][0], r[j][1], p0);
#pragma unroll
      for (int k = 0; k < NT; ++k) *reinterpret_cast<unsigned*>(img + k * TS * BS + o + j * BS) = p0[k];
    }
  } else {
    const int o = 4 * (t >> 3) * BS + 4 * (t & 7);
    const float r[4][4] = {{v[0].x, v[1].x, v[2].x, v[3].x}, {v[0].y, v[1].y, v[2].y, v[3].y},
                           {v[0].z, v[1].z, v[2].z, v[3].z}, {v[0].w, v[1].w, v[2].w, v[3].w}};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned p0[NT], p1[NT];
      bfn_split2<NT>(r[j][0], r[j][1], p0);
      bfn_split2<NT>(r[j][2], r[j][3], p1);
#pragma unroll
      for (int k = 0; k < NT; ++k) *reinterpret_cast<uint2*>(img + k * TS * BS + o + j * BS) = make_uint2(p0[k], p1[k]);
    }
  }
}

// TS = 128: the 128 x 128 tile (a wave owns 64 x 64).  TS = 64: a 64 x 64 tile (a wave owns one 32 x 32 block) for the
// products too small to fill the chip with 128 x 128 tiles - the decoder-side projections, their gradients, the output
// layer: [3 200, 512] outputs are 100 large tiles, each a serial walk of 16-64 K tiles alone on its CU; as 400 small tiles
// every CU holds one or two workgroups whose K tiles cost a quarter (tools/gemm_shapes.py).
//
// QUEUE (TS = 64 only; asr_gemm_side_f32): the instantiation that runs BESIDE the persistent XCD-local kernels of a small
// batch (8 utterances keep the LSTM / decoder chains on four of the eight XCDs).  Three things make it a good
// neighbour: it FITS next to a persistent workgroup on a CU (<= 128 VGPRs by its launch bounds - the LSTM backward leaves 144
// per SIMD -, 30 KB of LDS beside their 82), so a persistent launch never waits for one of its workgroups to find room; a
// workgroup that lands on an XCD outside g.xcd_mask leaves at once, so the chain's L2 sees none of its operand traffic; and
// a workgroup computes ONE (tile, K slice) - drawn from the ticket counter g.queue, because which workgroups survive the mask
// is not known at launch - and ends, so the critical-path kernels of the main stream find CUs at tile granularity.  One K tile
// in flight instead of three (registers); the partial products meet in atomics (C starts from zero or accumulates).
template <bool AKC, bool BKC, int NT, int TS = 128, bool QUEUE = false>
__global__ __launch_bounds__(256, QUEUE ? 4 : (TS == 64 ? ASR_GEMM_SMALL_OCC : 1)) void gemm_bf3_kernel(GemmArgs g) {
  static_assert(!QUEUE || TS == 64, "the queue instantiation is the 64 x 64 tile");
  // NT images (split terms, most significant first) per operand: 40 KB for two terms, 60 KB for three (TS = 128)
  constexpr int BM = TS, BN = TS, NP = TS / 32, NB = TS / 64;      // tile, float4 pieces per thread and operand, blocks per wave and side
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * NT * BM * BS];
  unsigned short* Ai = smem;
  unsigned short* Bi = smem + NT * BM * BS;

  const int ntile = g.tiles_m * g.tiles_n;
  int tid = blockIdx.x;
  int z = blockIdx.y;
  if constexpr (QUEUE) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;      // XCC_ID
    if (!((g.xcd_mask >> xcc) & 1u)) return;
    __shared__ unsigned ticket;
    if (threadIdx.x == 0) ticket = atomicAdd(g.queue, 1u);
    __syncthreads();
    const unsigned tk = ticket;
    if (tk >= (unsigned)ntile * gridDim.y) return;       // gridDim.y = batch * split_k; the x extent is oversubscribed
    tid = (int)(tk % (unsigned)ntile);
    z = (int)(tk / (unsigned)ntile);
  } else {
    const int q = ntile >> 3, rmd = ntile & 7, xcd = tid & 7, idx = tid >> 3;
    tid = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + idx;
  }
  const int tm = tid / g.tiles_n, tn = tid % g.tiles_n;
  const int bz = z / g.split_k, kz = z % g.split_k;

  MatView A = g.A, B = g.B;
  A.p += bz * g.sA;
  B.p += bz * g.sB;
  float* C = g.C + bz * g.sC;

  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int64_t ktiles = (g.K + BK - 1) / BK;
  const int64_t per = (ktiles + g.split_k - 1) / g.split_k;
  const int64_t kt_begin = kz * per;
  const int64_t kt_end = kt_begin + per < ktiles ? kt_begin + per : ktiles;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;

  f32x16 acc[NB][NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // interior tiles and full K tiles: uniform base + 32-bit per-thread offsets (see gemm_f32_kernel)
  const bool fast_a = A.vec && (AKC ? (m0 + BM <= A.R) : (m0 + BM <= A.Cn)) && A.R * A.ld < (int64_t)1 << 30;
  const bool fast_b = B.vec && (BKC ? (n0 + BN <= B.R) : (n0 + BN <= B.Cn)) && B.R * B.ld < (int64_t)1 << 30;
  const bool fast = fast_a && fast_b;
  unsigned offa[NP], offb[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int f = threadIdx.x + 256 * i;
    const int t = threadIdx.x;
    const int krow = TS == 128 ? 4 * (t & 7) + i : 4 * (t & 7) + 2 * (t >> 7) + i, rq = TS == 128 ? (t >> 3) : ((t >> 3) & 15);
    offa[i] = AKC ? (unsigned)((f >> 3) * A.ld + 4 * (f & 7)) : (unsigned)(krow * A.ld + 4 * rq);
    offb[i] = BKC ? (unsigned)((f >> 3) * B.ld + 4 * (f & 7)) : (unsigned)(krow * B.ld + 4 * rq);
  }
  const float* basea = AKC ? A.p + m0 * A.ld : A.p + m0;     // + k0 (KC) or + k0*ld (MC)
  const float* baseb = BKC ? B.p + n0 * B.ld : B.p + n0;
  float4 ra[NP], rb[NP];
  // Two copies of the K loop, chosen once per workgroup.  With the bare loads and the guarded loads as two branches of
  // one fetch, hipcc loads into temporaries and copies them into (ra, rb) at the end of the branch - an s_waitcnt vmcnt
  // right behind the loads, i.e. a full memory round trip exposed on every K tile (the kernel ran at 150 TF-equivalent
  // on every large shape, and removing EITHER the loads, the staging or the products made the rest free).
  auto fetch_fast = [&](int64_t kt2) {
    const int64_t k0 = kt2 * BK;
    const float* pa = basea + (AKC ? k0 : k0 * A.ld);
    const float* pb = baseb + (BKC ? k0 : k0 * B.ld);
#pragma unroll
    for (int i = 0; i < NP; ++i) ra[i] = *reinterpret_cast<const float4*>(pa + offa[i]);
#pragma unroll
    for (int i = 0; i < NP; ++i) rb[i] = *reinterpret_cast<const float4*>(pb + offb[i]);
  };
  auto fetch_guard = [&](int64_t kt2) {
    const int64_t k0 = kt2 * BK;
    tile_fetch_bf3<AKC, TS>(A, m0, k0, ra);
    tile_fetch_bf3<BKC, TS>(B, n0, k0, rb);
  };
  const int tt = threadIdx.x & 127;
  const bool touch_a = threadIdx.x < 128;
  float touched = 0.f;
  const bool do_touch = TS == 128 && kt_end - kt_begin > 2 * ASR_GEMM_BF3_TOUCH;   // (the small tiles run several workgroups per CU instead)
  auto touch_tile = [&](int64_t ktt) {
    const MatView& m = touch_a ? A : B;
    const bool kc = touch_a ? AKC : BKC;
    const int64_t r0t = touch_a ? m0 : n0;
    int64_t rr, cc;
    if (kc) { rr = r0t + tt; cc = ktt * BK; }
    else { rr = ktt * BK + (tt >> 2); cc = r0t + 32 * (tt & 3); }
    rr = rr < m.R ? rr : m.R - 1;
    cc = cc < m.Cn ? cc : m.Cn - 1;
    touched = m.p[rr * m.ld + cc];
  };
  // fragment of row tile i at k-step ks: row wm*64 + 32 i + (l & 31), k = 16 ks + 8 (l >> 5) .. + 7
  const int ao = (wm * (TS / 2) + l31) * BS + 8 * kh, bo = (wn * (TS / 2) + l31) * BS + 8 * kh;
  auto multiply = [&]() {
#pragma unroll
    for (int ks = 0; ks < ((ASR_GB_ABL & 1) ? 0 : BK / 16); ++ks) {
      gu32x4 af[NB][NT], bf[NB][NT];
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int k = 0; k < NT; ++k) {
          af[i][k] = *reinterpret_cast<const gu32x4*>(Ai + k * BM * BS + ao + 32 * i * BS + 16 * ks);
          bf[i][k] = *reinterpret_cast<const gu32x4*>(Bi + k * BM * BS + bo + 32 * i * BS + 16 * ks);
        }
#define BF3G(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gbf16x8, a_), __builtin_bit_cast(gbf16x8, b_), c_, 0, 0, 0)
      // term pairs (p, q) with p + q < NT: 3 products for two terms, 6 for three
#pragma unroll
      for (int o = 0; o < NT; ++o)
#pragma unroll
        for (int p = 0; p <= o; ++p)
#pragma unroll
          for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) BF3G(af[i][p], bf[j][o - p], acc[i][j]);
#undef BF3G
    }
  };
  // full K tiles of interior workgroups run the bare-load loop; a K tail (K % 32 != 0, e.g. the 80-dim features) and edge
  // workgroups take the guarded loop, which does not overlap its loads (one tile in the K = 80 case)
  const int64_t kt_full = g.K / BK < kt_end ? g.K / BK : kt_end;
  const int64_t kt_fast_end = (fast && kt_full > kt_begin) ? kt_full : kt_begin;
  if constexpr (TS == 64) {
    // Small tiles: a K tile's 12 MFMAs per wave (384 cycles) cannot cover a memory round trip, so the operands of PD = 3
    // K tiles are in flight in registers (12 float4 per thread).  The steady state runs in groups of PD tiles with
    // unconditional fetches - a conditional one makes hipcc wait for ALL outstanding loads (vmcnt(0)) at the next store -
    // and the last < 2 PD tiles take the conditional form.
    constexpr int PD = QUEUE ? 1 : 3;
    float4 qa[PD][NP], qb[PD][NP];
    auto fetch_set = [&](int64_t kt2, float4 (&xa)[NP], float4 (&xb)[NP]) __attribute__((always_inline)) {
      const int64_t k0 = kt2 * BK;
      const float* pa = basea + (AKC ? k0 : k0 * A.ld);
      const float* pb = baseb + (BKC ? k0 : k0 * B.ld);
#pragma unroll
      for (int i = 0; i < NP; ++i) xa[i] = *reinterpret_cast<const float4*>(pa + offa[i]);
#pragma unroll
      for (int i = 0; i < NP; ++i) xb[i] = *reinterpret_cast<const float4*>(pb + offb[i]);
    };
    int64_t kt = kt_begin;
    if (kt < kt_fast_end) {
#pragma unroll
      for (int d = 0; d < PD; ++d)
        if (kt + d < kt_fast_end) fetch_set(kt + d, qa[d], qb[d]);
      for (; kt + 2 * PD <= kt_fast_end; kt += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
          __builtin_amdgcn_sched_barrier(0);        // (keeps the split of a later set - and the wait for its loads - out of this step)
          tile_store_bf3<AKC, NT, TS>(Ai, qa[d]);
          tile_store_bf3<BKC, NT, TS>(Bi, qb[d]);
          __syncthreads();
          fetch_set(kt + d + PD, qa[d], qb[d]);
          multiply();
          __syncthreads();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // tail: PD .. 2 PD - 1 tiles (or fewer than PD when the whole slice is short); tile kt + d sits in set d % PD
#pragma unroll
      for (int d = 0; d < 2 * PD - 1; ++d) {
        if (kt + d < kt_fast_end) {
          tile_store_bf3<AKC, NT, TS>(Ai, qa[d % PD]);
          tile_store_bf3<BKC, NT, TS>(Bi, qb[d % PD]);
          __syncthreads();
          if (kt + d + PD < kt_fast_end) fetch_set(kt + d + PD, qa[d % PD], qb[d % PD]);
          multiply();
          __syncthreads();
        }
      }
    }
  } else if (kt_begin < kt_fast_end) {
    fetch_fast(kt_begin);
    for (int64_t kt = kt_begin; kt < kt_fast_end; ++kt) {
#if ASR_GB_ABL & 4
      if (kt == kt_begin) {
        tile_store_bf3<AKC, NT, TS>(Ai, ra);
        tile_store_bf3<BKC, NT, TS>(Bi, rb);
      } else {
#pragma unroll
        for (int i = 0; i < NP; ++i) asm volatile("" ::"v"(ra[i].x), "v"(ra[i].y), "v"(ra[i].z), "v"(ra[i].w), "v"(rb[i].x), "v"(rb[i].y), "v"(rb[i].z), "v"(rb[i].w));
      }
#else
      tile_store_bf3<AKC, NT, TS>(Ai, ra);
      tile_store_bf3<BKC, NT, TS>(Bi, rb);
#endif
      __syncthreads();
      if (kt + 1 < kt_fast_end && !(ASR_GB_ABL & 2)) fetch_fast(kt + 1);
#if ASR_GEMM_TOUCH
      asm volatile("" ::"v"(touched));
      if (do_touch && kt + ASR_GEMM_BF3_TOUCH < kt_end) touch_tile(kt + ASR_GEMM_BF3_TOUCH);
#endif
      multiply();
      __syncthreads();
    }
  }
  if (kt_fast_end < kt_end) {
    fetch_guard(kt_fast_end);
    for (int64_t kt = kt_fast_end; kt < kt_end; ++kt) {
      tile_store_bf3<AKC, NT, TS>(Ai, ra);
      tile_store_bf3<BKC, NT, TS>(Bi, rb);
      __syncthreads();
      if (kt + 1 < kt_end) fetch_guard(kt + 1);
#if ASR_GEMM_TOUCH
      asm volatile("" ::"v"(touched));
      if (do_touch && kt + ASR_GEMM_BF3_TOUCH < kt_end) touch_tile(kt + ASR_GEMM_BF3_TOUCH);
#endif
      multiply();
      __syncthreads();
    }
  }

#if ASR_GB_ABL & 8   /* measurement: no epilogue stores (one lane keeps the accumulators alive) */
  if (threadIdx.x + blockIdx.x + blockIdx.y != 0 || g.M != 1) {
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(acc[i][j][e]));
    return;
  }
#endif
  // epilogue (as gemm_f32_kernel: the C/D lane map of the 32x32 MFMAs does not depend on the input type)
  if (m0 + BM <= g.M && n0 + BN <= g.N) {
    const unsigned ldc = (unsigned)g.ldc;
    const bool split = QUEUE || g.split_k > 1, accum = g.accumulate != 0, relu = g.relu != 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int64_t n = n0 + wn * (TS / 2) + j * 32 + l31;
      const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        float* base = C + (m0 + wm * (TS / 2) + i * 32 + 4 * kh) * g.ldc + n;
        if (split) {
#pragma unroll
          for (int e = 0; e < 16; ++e) atomicAdd(base + (unsigned)((e & 3) + 8 * (e >> 2)) * ldc, acc[i][j][e]);
        } else if (accum) {
          float old[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) old[e] = base[(unsigned)((e & 3) + 8 * (e >> 2)) * ldc];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float v = acc[i][j][e] + bv + old[e];
            if (relu) v = fmaxf(v, 0.f);
            base[(unsigned)((e & 3) + 8 * (e >> 2)) * ldc] = v;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float v = acc[i][j][e] + bv;
            if (relu) v = fmaxf(v, 0.f);
            base[(unsigned)((e & 3) + 8 * (e >> 2)) * ldc] = v;
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int64_t n = n0 + wn * (TS / 2) + j * 32 + l31;
    if (n >= g.N) continue;
    const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t m = m0 + wm * (TS / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        if (m >= g.M) continue;
        float v = acc[i][j][e];
        float* dst = C + m * g.ldc + n;
        if (QUEUE || g.split_k > 1) {
          atomicAdd(dst, v);
        } else {
          v += bv;
          if (g.accumulate) v += *dst;
          if (g.relu) v = fmaxf(v, 0.f);
          *dst = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ split-bf16, wide tile
// gemm_bf3w_kernel: the same arithmetic on a 256 x 128 output tile per workgroup of 8 waves, operands brought in by
// LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass) into a ring of three fp32 stages of
// 32 k (48 KB each, 96 KB in flight per CU while the third is multiplied), one raw s_barrier per stage and counted
// s_waitcnt vmcnt.  The bf16 split happens on the fragments, after the LDS read (2.5 VALU per element in the shadow of
// the MFMAs).  Waves: 2 (M) x 2 (N) x 2 (K halves of a stage); a wave owns a 128 x 64 accumulator (4 x 2 blocks of
// 32 x 32) over its 16 k of every stage, and the two K halves are summed through LDS once, each half then writing 64
// of the 128 rows.  Per stage and wave: 12 ds_read_b128 (or 48 ds_read_b32), 120 VALU, 24 MFMAs of 32 cycles.
//   k-contiguous operands ([rows][K]): a 1 KB piece = 8 rows x 128 B; the 16-byte granule q of row r sits at slot
//     q ^ ((r >> 1) & 7) of its row (the swizzle is applied to the SOURCE address of the DMA lane: the LDS side of a
//     DMA is lane-linear), which makes the fragment's two ds_read_b128 conflict-free over the instruction's lane groups;
//   row-contiguous operands ([K][rows]): a piece = one k (A) or two (B), image [k][rows], fragments by ds_read_b32.
// Shapes: K % 32 == 0, 16-byte aligned rows, M and N free (edge tiles clamp their DMA sources and guard their stores);
// everything else stays on gemm_bf3_kernel.
constexpr int WM = 256, WN = 128, WK = 32;
constexpr int W_A_FLOATS = WM * WK, W_B_FLOATS = WN * WK, W_STAGE_FLOATS = W_A_FLOATS + W_B_FLOATS, W_STAGES = 3;
typedef __attribute__((address_space(3))) void* lds_vptr;

// One LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to lds_dst + 16 * lane.  Issued as inline asm
// on purpose: for the builtin hipcc puts an s_waitcnt vmcnt(0) in front of the next ds_read of the same array (it cannot
// tell the stages of the ring apart), which drains the two stages in flight every iteration.  The kernel counts vmcnt
// itself; no other VMEM load is outstanding while DMAs are.
__device__ __forceinline__ void glds16(const float* src, float* lds_dst) {
  const unsigned dst = (unsigned)(uintptr_t)(lds_vptr)lds_dst;
#if ASR_GLDS_CLOBBER_M0     /* measurement: m0 on the clobber list instead (hipcc warns: reserved register) */
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory", "m0");
#else
  unsigned keep;                                   // m0 is saved and restored: it may not appear in a clobber list
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
#endif
}

#ifndef ASR_GW_ABL      /* measurement only: 1 no DMA inside the stage loop, 2 no split arithmetic (raw bits as hi / lo), 4 no fragment reads after the prologue */
#define ASR_GW_ABL 0
#endif
__device__ __forceinline__ void bf3w_split(const float (&v)[8], gu32x4& hi, gu32x4& lo) {
#if ASR_GW_ABL & 2
#pragma unroll
  for (int p = 0; p < 4; ++p) { hi[p] = __float_as_uint(v[p]); lo[p] = __float_as_uint(v[4 + p]); }
  return;
#endif
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    hi[p] = bf3g_hi2(v[2 * p], v[2 * p + 1]);
    lo[p] = bf3g_lo2(v[2 * p], v[2 * p + 1]);
  }
}

// the 8 consecutive k of this lane for one 32-row block of an operand stage
template <bool KC, int ROWS>
__device__ __forceinline__ void bf3w_frag(const float* st, int row, int kq0s, int kq1s, int k0, gu32x4& hi, gu32x4& lo) {
  float v[8];
  if (KC) {
    const float4 x = *reinterpret_cast<const float4*>(st + row * WK + kq0s);
    const float4 y = *reinterpret_cast<const float4*>(st + row * WK + kq1s);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = st[(k0 + e) * ROWS + row];
  }
  bf3w_split(v, hi, lo);
}

template <int I0>
__device__ __forceinline__ void bf3w_send(float* xs, const f32x16 (&acc)[4][2], int lane) {
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) xs[((b * 2 + j) * 16 + e) * 64 + lane] = acc[I0 + b][j][e];
}

template <int I0, bool EDGE>
__device__ __forceinline__ void bf3w_finish(const float* xr, f32x16 (&acc)[4][2], const GemmArgs& g, float* C, int64_t mrow0,
                                            int64_t ncol0, int lane) {
  const int l31 = lane & 31, kh = lane >> 5;
  const unsigned ldc = (unsigned)g.ldc;
  const bool split = g.split_k > 1, accum = g.accumulate != 0, relu = g.relu != 0;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t n = ncol0 + j * 32 + l31;
      const int64_t mb = mrow0 + (I0 + b) * 32 + 4 * kh;
      const bool nok = !EDGE || n < g.N;
      const float bv = (g.bias && nok) ? g.bias[n] : 0.f;
      float* base = C + mb * g.ldc + n;
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = acc[I0 + b][j][e] + xr[((b * 2 + j) * 16 + e) * 64 + lane];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2);
        if (EDGE && !(nok && mb + r < g.M)) continue;
        float* dst = base + (unsigned)r * ldc;
        if (split) {
          atomicAdd(dst, v[e]);
        } else {
          float o = v[e] + bv;
          if (accum) o += *dst;
          if (relu) o = fmaxf(o, 0.f);
          *dst = o;
        }
      }
    }
}

#ifdef ASR_GW_TRACE   /* measurement builds only (tools/gemm_wide_trace.py): shader-clock stamps of waves 0 and 4 of workgroup 0 */
__device__ unsigned long long asr_gw_trace_buf[2 * 64 * 8];
#define GW_MARK(m) do { if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (wave & 3) == 0 && st < 63) \
    asr_gw_trace_buf[((wave >> 2) * 64 + st) * 8 + (m)] = clock64(); } while (0)
#define GW_COARSE(m) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) asr_gw_trace_buf[63 * 8 + (m)] = clock64(); } while (0)
#else
#define GW_MARK(m) do {} while (0)
#define GW_COARSE(m) do {} while (0)
#endif

template <bool AKC, bool BKC>
__global__ __launch_bounds__(512) void gemm_bf3w_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(1024))) float smem[W_STAGES * W_STAGE_FLOATS];

  const int ntile = g.tiles_m * g.tiles_n;
  int tid = blockIdx.x;
  {
    const int q = ntile >> 3, rmd = ntile & 7, xcd = tid & 7, idx = tid >> 3;
    tid = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + idx;
  }
  const int tm = tid / g.tiles_n, tn = tid % g.tiles_n;
  const int z = blockIdx.y;
  const int bz = z / g.split_k, kz = z % g.split_k;
  const float* Ap = g.A.p + bz * g.sA;
  const float* Bp = g.B.p + bz * g.sB;
  float* C = g.C + bz * g.sC;
  const int64_t m0 = (int64_t)tm * WM, n0 = (int64_t)tn * WN;
  const int64_t ktiles = g.K / WK;
  const int64_t per = (ktiles + g.split_k - 1) / g.split_k;
  const int64_t kt_begin = kz * per;
  const int64_t kt_end = kt_begin + per < ktiles ? kt_begin + per : ktiles;
  const int S = (int)(kt_end - kt_begin);
  if (S <= 0) return;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;

  // ---- DMA source offsets of this lane (floats, relative to the operand's first element at the stage's first k).  Rows
  // and columns past the matrix edge are clamped to the last valid ones: what they deliver only reaches accumulator
  // rows / columns that the guarded epilogue does not store.
  const int64_t lda = g.A.ld, ldb = g.B.ld;
  unsigned offa[4], offb[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int p = 4 * wave + q;
    if (AKC) {
      const int r = 8 * p + (lane >> 3);
      const int64_t row = m0 + r < g.M ? m0 + r : g.M - 1;
      offa[q] = (unsigned)(row * lda + 4 * ((lane & 7) ^ ((r >> 1) & 7)));
    } else {
      const int64_t col = m0 + 4 * lane + 4 <= g.M ? m0 + 4 * lane : g.M - 4;
      offa[q] = (unsigned)(p * lda + col);
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int p = 2 * wave + q;
    if (BKC) {
      const int r = 8 * p + (lane >> 3);
      const int64_t row = n0 + r < g.N ? n0 + r : g.N - 1;
      offb[q] = (unsigned)(row * ldb + 4 * ((lane & 7) ^ ((r >> 1) & 7)));
    } else {
      const int64_t col = n0 + 4 * (lane & 31) + 4 <= g.N ? n0 + 4 * (lane & 31) : g.N - 4;
      offb[q] = (unsigned)((2 * p + (lane >> 5)) * ldb + col);
    }
  }
  const float* basea = AKC ? Ap + kt_begin * WK : Ap + kt_begin * WK * lda;
  const float* baseb = BKC ? Bp + kt_begin * WK : Bp + kt_begin * WK * ldb;
  const int64_t stepa = AKC ? WK : WK * lda, stepb = BKC ? WK : WK * ldb;
  auto issue = [&](int st, int buf) {
    float* sb = smem + buf * W_STAGE_FLOATS;
    const float* pa = basea + st * stepa;
    const float* pb = baseb + st * stepb;
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(pa + offa[q], sb + (4 * wave + q) * 256);
#pragma unroll
    for (int q = 0; q < 2; ++q) glds16(pb + offb[q], sb + W_A_FLOATS + (2 * wave + q) * 256);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment addressing: granules (4 kg + 2 kh, + 1) of the lane's row, swizzled by (row >> 1) & 7 = (l31 >> 1) & 7
  const int sw = (l31 >> 1) & 7;
  const int kq0s = 4 * ((4 * kg + 2 * kh) ^ sw), kq1s = kq0s ^ 4;
  const int kfirst = 16 * kg + 8 * kh;
  const int arow = wm * 128 + l31, brow = wn * 64 + l31;

  // Software pipeline inside every wave (all 8 waves in step, one barrier per stage, in its middle):
  //   H1(s): 12 MFMAs  A rows 0-63 x B of stage s          || read + split A rows 64-127 of stage s
  //   mid(s): this wave's DMAs of stage s + 1 have landed (counted vmcnt, stage s + 2 stays in flight), barrier
  //   H2(s): 12 MFMAs  A rows 64-127 x B of stage s        || read + split B and A rows 0-63 of stage s + 1,
  //                                                            issue the six DMAs of stage s + 3 into stage s's buffer
  // (stage s's buffer is last read in H1(s), before the barrier).  Measured before this: with the fragment reads, the
  // split and the MFMAs of a stage back to back (first version) a stage took 3 800 cycles per SIMD for 1 536 cycles of
  // matrix pipe, and with the two K halves running in opposite phases (second version) 5 250: tools/gemm_wide_trace.py
  // showed 1 040-1 550 cycles per fragment phase (three exposed LDS round trips + 120 VALU) and 125 cycles per DMA issue.
  gu32x4 bh[2][2], bl[2][2], a0h[2], a0l[2], a1h[2], a1l[2];
#define BF3W(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gbf16x8, a_), __builtin_bit_cast(gbf16x8, b_), c_, 0, 0, 0)
  auto stage = [&](int st, int buf, auto ptag, auto ftag) {
    constexpr int P = decltype(ptag)::value;
    constexpr bool FULL = decltype(ftag)::value;       // stages st + 1 and st + 3 exist: no conditionals in the body
    const float* sa = smem + buf * W_STAGE_FLOATS;
    const int buf1 = buf == 2 ? 0 : buf + 1;
    const float* sa1 = smem + buf1 * W_STAGE_FLOATS;
    // ---- H1
    GW_MARK(0);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 2; ++j) BF3W(a0h[a], bh[P][j], acc[a][j]);
#pragma unroll
    for (int a = 0; a < 2; ++a) if (!(ASR_GW_ABL & 4) || st == 0) bf3w_frag<AKC, WM>(sa, arow + 32 * (2 + a), kq0s, kq1s, kfirst, a1h[a], a1l[a]);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 2; ++j) BF3W(a0h[a], bl[P][j], acc[a][j]);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 2; ++j) BF3W(a0l[a], bh[P][j], acc[a][j]);
    // ---- mid
    const bool next = FULL || st + 1 < S;
    GW_MARK(1);
    if (next) {
      if (FULL || st + 2 < S) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      GW_MARK(2);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");      // the raw barrier has no memory semantics in the IR: keep the LDS reads of stage s + 1 behind it
    }
    GW_MARK(3);
    // ---- H2
    const bool dma = (FULL || st + 3 < S) && !(ASR_GW_ABL & 1);
    float* sb = smem + buf * W_STAGE_FLOATS;           // stage st + 3 goes where stage st was
    const float* pa = basea + (st + 3) * stepa;
    const float* pb = baseb + (st + 3) * stepb;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 2; ++j) BF3W(a1h[a], bh[P][j], acc[2 + a][j]);
    if (dma) {
      glds16(pa + offa[0], sb + (4 * wave + 0) * 256);
      glds16(pa + offa[1], sb + (4 * wave + 1) * 256);
    }
    if (next && (!(ASR_GW_ABL & 4) || st < 2)) {
#pragma unroll
      for (int j = 0; j < 2; ++j) bf3w_frag<BKC, WN>(sa1 + W_A_FLOATS, brow + 32 * j, kq0s, kq1s, kfirst, bh[1 - P][j], bl[1 - P][j]);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 2; ++j) BF3W(a1h[a], bl[P][j], acc[2 + a][j]);
    if (dma) {
      glds16(pa + offa[2], sb + (4 * wave + 2) * 256);
      glds16(pa + offa[3], sb + (4 * wave + 3) * 256);
    }
    if (next && (!(ASR_GW_ABL & 4) || st < 2)) {
#pragma unroll
      for (int a = 0; a < 2; ++a) bf3w_frag<AKC, WM>(sa1, arow + 32 * a, kq0s, kq1s, kfirst, a0h[a], a0l[a]);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 2; ++j) BF3W(a1l[a], bh[P][j], acc[2 + a][j]);
    if (dma) {
      glds16(pb + offb[0], sb + W_A_FLOATS + (2 * wave + 0) * 256);
      glds16(pb + offb[1], sb + W_A_FLOATS + (2 * wave + 1) * 256);
    }
    GW_MARK(4);
  };
#undef BF3W
  typedef std::integral_constant<int, 0> P0;
  typedef std::integral_constant<int, 1> P1;
  typedef std::integral_constant<bool, true> Full;
  typedef std::integral_constant<bool, false> Tail;

  GW_COARSE(0);
  issue(0, 0);
  if (S > 1) issue(1, 1);
  if (S > 2) issue(2, 2);
  if (S > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (S > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int j = 0; j < 2; ++j) bf3w_frag<BKC, WN>(smem + W_A_FLOATS, brow + 32 * j, kq0s, kq1s, kfirst, bh[0][j], bl[0][j]);
#pragma unroll
  for (int a = 0; a < 2; ++a) bf3w_frag<AKC, WM>(smem, arow + 32 * a, kq0s, kq1s, kfirst, a0h[a], a0l[a]);
  int st = 0, buf = 0;
  GW_COARSE(1);
  for (; st + 4 < S; st += 2) {                  // both stages of the pair have st + 3 < S
    stage(st, buf, P0(), Full());
    buf = buf == 2 ? 0 : buf + 1;
    stage(st + 1, buf, P1(), Full());
    buf = buf == 2 ? 0 : buf + 1;
  }
  for (; st < S; st += 2) {                      // st stays even: the parity of the B registers is a compile-time tag
    stage(st, buf, P0(), Tail());
    buf = buf == 2 ? 0 : buf + 1;
    if (st + 1 < S) {
      stage(st + 1, buf, P1(), Tail());
      buf = buf == 2 ? 0 : buf + 1;
    }
  }

  // ---- sum the two K halves: each wave hands the 64 rows it does not write to its partner (same wm, wn)
  GW_COARSE(2);
  __syncthreads();
  const int w4 = wave & 3;
  float* xs = smem + ((w4 * 2 + (1 - kg)) * 64) * 64;       // read by the partner
  const float* xr = smem + ((w4 * 2 + kg) * 64) * 64;       // written by the partner
  if (kg == 0) bf3w_send<2>(xs, acc, lane); else bf3w_send<0>(xs, acc, lane);
  __syncthreads();
  const int64_t mrow0 = m0 + wm * 128, ncol0 = n0 + wn * 64;
  GW_COARSE(3);
  if (m0 + WM <= g.M && n0 + WN <= g.N) {
    if (kg == 0) bf3w_finish<0, false>(xr, acc, g, C, mrow0, ncol0, lane);
    else bf3w_finish<2, false>(xr, acc, g, C, mrow0, ncol0, lane);
  } else {
    if (kg == 0) bf3w_finish<0, true>(xr, acc, g, C, mrow0, ncol0, lane);
    else bf3w_finish<2, true>(xr, acc, g, C, mrow0, ncol0, lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GW_COARSE(4);
}

// ------------------------------------------------------------------------------------ three-term split, wide tile
// gemm_bf6w_kernel: gemm_bf3w_kernel's tile, DMA ring and layouts with every operand split in THREE bf16 terms on the
// fragments (bfn_split2<3>: a + b + c == x exactly) and SIX products per product - fp32-equivalent arithmetic, the
// default of the train step.  Per stage and wave: 12 ds_read_b128, ~270 VALU, 48 MFMAs of 32 cycles.  Three terms of the
// 4 + 2 fragment blocks of a stage plus gemm_bf3w_kernel's look-ahead would need 96 fragment registers next to the 128
// accumulators; the stage is therefore cut in four quarters of 12 MFMAs (A half x B block) and every quarter fetches
// exactly the one operand the next quarter changes, which caps the live fragments at 72 registers:
//   Q1: A rows 0-63   x B cols 0-31  of stage s   || read + split B cols 32-63 of stage s
//   Q2: A rows 0-63   x B cols 32-63              || read + split A rows 64-127 of stage s      (last reads of stage s)
//   mid: own DMAs of stage s + 1 landed (counted vmcnt), barrier
//   Q3: A rows 64-127 x B cols 32-63              || read + split A rows 0-63 of stage s + 1, DMAs of stage s + 3 (A)
//   Q4: A rows 64-127 x B cols 0-31               || read + split B cols 0-31 of stage s + 1 (into the registers B cols
//                                                    32-63 just left: the two B register sets swap roles every stage,
//                                                    compile-time tag P), DMAs of stage s + 3 (B)
template <bool KC, int ROWS>
__device__ __forceinline__ void bf6w_frag(const float* st, int row, int kq0s, int kq1s, int k0, gu32x4 (&t)[3]) {
  float v[8];
  if (KC) {
    const float4 x = *reinterpret_cast<const float4*>(st + row * WK + kq0s);
    const float4 y = *reinterpret_cast<const float4*>(st + row * WK + kq1s);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = st[(k0 + e) * ROWS + row];
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned q[3];
    bfn_split2<3>(v[2 * p], v[2 * p + 1], q);
    t[0][p] = q[0]; t[1][p] = q[1]; t[2][p] = q[2];
  }
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(512) void gemm_bf6w_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(1024))) float smem[W_STAGES * W_STAGE_FLOATS];

  const int ntile = g.tiles_m * g.tiles_n;
  int tid = blockIdx.x;
  {
    const int q = ntile >> 3, rmd = ntile & 7, xcd = tid & 7, idx = tid >> 3;
    tid = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + idx;
  }
  const int tm = tid / g.tiles_n, tn = tid % g.tiles_n;
  const int z = blockIdx.y;
  const int bz = z / g.split_k, kz = z % g.split_k;
  const float* Ap = g.A.p + bz * g.sA;
  const float* Bp = g.B.p + bz * g.sB;
  float* C = g.C + bz * g.sC;
  const int64_t m0 = (int64_t)tm * WM, n0 = (int64_t)tn * WN;
  const int64_t ktiles = g.K / WK;
  const int64_t per = (ktiles + g.split_k - 1) / g.split_k;
  const int64_t kt_begin = kz * per;
  const int64_t kt_end = kt_begin + per < ktiles ? kt_begin + per : ktiles;
  const int S = (int)(kt_end - kt_begin);
  if (S <= 0) return;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;

  // DMA source offsets of this lane: as in gemm_bf3w_kernel (edge rows / columns clamped, k-contiguous pieces swizzled)
  const int64_t lda = g.A.ld, ldb = g.B.ld;
  unsigned offa[4], offb[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int p = 4 * wave + q;
    if (AKC) {
      const int r = 8 * p + (lane >> 3);
      const int64_t row = m0 + r < g.M ? m0 + r : g.M - 1;
      offa[q] = (unsigned)(row * lda + 4 * ((lane & 7) ^ ((r >> 1) & 7)));
    } else {
      const int64_t col = m0 + 4 * lane + 4 <= g.M ? m0 + 4 * lane : g.M - 4;
      offa[q] = (unsigned)(p * lda + col);
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int p = 2 * wave + q;
    if (BKC) {
      const int r = 8 * p + (lane >> 3);
      const int64_t row = n0 + r < g.N ? n0 + r : g.N - 1;
      offb[q] = (unsigned)(row * ldb + 4 * ((lane & 7) ^ ((r >> 1) & 7)));
    } else {
      const int64_t col = n0 + 4 * (lane & 31) + 4 <= g.N ? n0 + 4 * (lane & 31) : g.N - 4;
      offb[q] = (unsigned)((2 * p + (lane >> 5)) * ldb + col);
    }
  }
  const float* basea = AKC ? Ap + kt_begin * WK : Ap + kt_begin * WK * lda;
  const float* baseb = BKC ? Bp + kt_begin * WK : Bp + kt_begin * WK * ldb;
  const int64_t stepa = AKC ? WK : WK * lda, stepb = BKC ? WK : WK * ldb;
  auto issue = [&](int st, int buf) {
    float* sb = smem + buf * W_STAGE_FLOATS;
    const float* pa = basea + st * stepa;
    const float* pb = baseb + st * stepb;
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(pa + offa[q], sb + (4 * wave + q) * 256);
#pragma unroll
    for (int q = 0; q < 2; ++q) glds16(pb + offb[q], sb + W_A_FLOATS + (2 * wave + q) * 256);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int sw = (l31 >> 1) & 7;
  const int kq0s = 4 * ((4 * kg + 2 * kh) ^ sw), kq1s = kq0s ^ 4;
  const int kfirst = 16 * kg + 8 * kh;
  const int arow = wm * 128 + l31, brow = wn * 64 + l31;

  gu32x4 bq[2][3], a0[2][3], a1[2][3];
#define BF6W(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gbf16x8, a_), __builtin_bit_cast(gbf16x8, b_), c_, 0, 0, 0)
  // the six products of (A half, B block): term pairs in order of significance sum; consecutive MFMAs alternate between
  // the two accumulators of the quarter
#define BF6W_PAIR(ah_, b_, i0_, j_, p_, q_) do { BF6W(ah_[0][p_], b_[q_], acc[i0_][j_]); BF6W(ah_[1][p_], b_[q_], acc[i0_ + 1][j_]); } while (0)
  auto stage = [&](int st, int buf, auto ptag, auto ftag) {
    constexpr int P = decltype(ptag)::value;
    constexpr bool FULL = decltype(ftag)::value;       // stages st + 1 and st + 3 exist: no conditionals in the body
    const float* sa = smem + buf * W_STAGE_FLOATS;
    const int buf1 = buf == 2 ? 0 : buf + 1;
    const float* sa1 = smem + buf1 * W_STAGE_FLOATS;
    gu32x4 (&b0)[3] = bq[P];
    gu32x4 (&b1)[3] = bq[1 - P];
    const bool next = FULL || st + 1 < S;
    // ---- Q1
    BF6W_PAIR(a0, b0, 0, 0, 0, 0);
    BF6W_PAIR(a0, b0, 0, 0, 0, 1);
    bf6w_frag<BKC, WN>(sa + W_A_FLOATS, brow + 32, kq0s, kq1s, kfirst, b1);
    BF6W_PAIR(a0, b0, 0, 0, 1, 0);
    BF6W_PAIR(a0, b0, 0, 0, 0, 2);
    BF6W_PAIR(a0, b0, 0, 0, 2, 0);
    BF6W_PAIR(a0, b0, 0, 0, 1, 1);
    // ---- Q2
    BF6W_PAIR(a0, b1, 0, 1, 0, 0);
    bf6w_frag<AKC, WM>(sa, arow + 64, kq0s, kq1s, kfirst, a1[0]);
    BF6W_PAIR(a0, b1, 0, 1, 0, 1);
    BF6W_PAIR(a0, b1, 0, 1, 1, 0);
    bf6w_frag<AKC, WM>(sa, arow + 96, kq0s, kq1s, kfirst, a1[1]);
    BF6W_PAIR(a0, b1, 0, 1, 0, 2);
    BF6W_PAIR(a0, b1, 0, 1, 2, 0);
    BF6W_PAIR(a0, b1, 0, 1, 1, 1);
    // ---- mid
    if (next) {
      if (FULL || st + 2 < S) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    // ---- Q3
    const bool dma = FULL || st + 3 < S;
    float* sb = smem + buf * W_STAGE_FLOATS;           // stage st + 3 goes where stage st was
    const float* pa = basea + (st + 3) * stepa;
    const float* pb = baseb + (st + 3) * stepb;
    BF6W_PAIR(a1, b1, 2, 1, 0, 0);
    if (dma) {
      glds16(pa + offa[0], sb + (4 * wave + 0) * 256);
      glds16(pa + offa[1], sb + (4 * wave + 1) * 256);
    }
    if (next) bf6w_frag<AKC, WM>(sa1, arow, kq0s, kq1s, kfirst, a0[0]);
    BF6W_PAIR(a1, b1, 2, 1, 0, 1);
    BF6W_PAIR(a1, b1, 2, 1, 1, 0);
    if (dma) {
      glds16(pa + offa[2], sb + (4 * wave + 2) * 256);
      glds16(pa + offa[3], sb + (4 * wave + 3) * 256);
    }
    if (next) bf6w_frag<AKC, WM>(sa1, arow + 32, kq0s, kq1s, kfirst, a0[1]);
    BF6W_PAIR(a1, b1, 2, 1, 0, 2);
    BF6W_PAIR(a1, b1, 2, 1, 2, 0);
    BF6W_PAIR(a1, b1, 2, 1, 1, 1);
    // ---- Q4 (b1 is dead: its registers receive B cols 0-31 of the next stage)
    BF6W_PAIR(a1, b0, 2, 0, 0, 0);
    if (dma) {
      glds16(pb + offb[0], sb + W_A_FLOATS + (2 * wave + 0) * 256);
      glds16(pb + offb[1], sb + W_A_FLOATS + (2 * wave + 1) * 256);
    }
    BF6W_PAIR(a1, b0, 2, 0, 0, 1);
    if (next) bf6w_frag<BKC, WN>(sa1 + W_A_FLOATS, brow, kq0s, kq1s, kfirst, b1);
    BF6W_PAIR(a1, b0, 2, 0, 1, 0);
    BF6W_PAIR(a1, b0, 2, 0, 0, 2);
    BF6W_PAIR(a1, b0, 2, 0, 2, 0);
    BF6W_PAIR(a1, b0, 2, 0, 1, 1);
  };
  typedef std::integral_constant<int, 0> P0;
  typedef std::integral_constant<int, 1> P1;
  typedef std::integral_constant<bool, true> Full;
  typedef std::integral_constant<bool, false> Tail;

  issue(0, 0);
  if (S > 1) issue(1, 1);
  if (S > 2) issue(2, 2);
  if (S > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (S > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  bf6w_frag<BKC, WN>(smem + W_A_FLOATS, brow, kq0s, kq1s, kfirst, bq[0]);
  bf6w_frag<AKC, WM>(smem, arow, kq0s, kq1s, kfirst, a0[0]);
  bf6w_frag<AKC, WM>(smem, arow + 32, kq0s, kq1s, kfirst, a0[1]);
  int st = 0, buf = 0;
  for (; st + 4 < S; st += 2) {                  // both stages of the pair have st + 3 < S
    stage(st, buf, P0(), Full());
    buf = buf == 2 ? 0 : buf + 1;
    stage(st + 1, buf, P1(), Full());
    buf = buf == 2 ? 0 : buf + 1;
  }
  for (; st < S; st += 2) {                      // st stays even: the parity of the B registers is a compile-time tag
    stage(st, buf, P0(), Tail());
    buf = buf == 2 ? 0 : buf + 1;
    if (st + 1 < S) {
      stage(st + 1, buf, P1(), Tail());
      buf = buf == 2 ? 0 : buf + 1;
    }
  }
#undef BF6W_PAIR
#undef BF6W

  // ---- sum the two K halves and store: as gemm_bf3w_kernel
  __syncthreads();
  const int w4 = wave & 3;
  float* xs = smem + ((w4 * 2 + (1 - kg)) * 64) * 64;       // read by the partner
  const float* xr = smem + ((w4 * 2 + kg) * 64) * 64;       // written by the partner
  if (kg == 0) bf3w_send<2>(xs, acc, lane); else bf3w_send<0>(xs, acc, lane);
  __syncthreads();
  const int64_t mrow0 = m0 + wm * 128, ncol0 = n0 + wn * 64;
  if (m0 + WM <= g.M && n0 + WN <= g.N) {
    if (kg == 0) bf3w_finish<0, false>(xr, acc, g, C, mrow0, ncol0, lane);
    else bf3w_finish<2, false>(xr, acc, g, C, mrow0, ncol0, lane);
  } else {
    if (kg == 0) bf3w_finish<0, true>(xr, acc, g, C, mrow0, ncol0, lane);
    else bf3w_finish<2, true>(xr, acc, g, C, mrow0, ncol0, lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------ split-bf16, one wave per SIMD
// gemm_bfs_kernel: 256 x 128 output tile per workgroup of FOUR waves - one per SIMD, up to 512 registers each - and the
// whole pipeline of a K tile inside the instruction stream of every wave.  Why: two waves on one SIMD do not overlap one's
// split arithmetic with the other's MFMAs (measured with a two-group ping-pong variant of gemm_bf3_kernel: the staging
// phase takes 2 900 - 4 400 cycles beside the partner's matrix phase, 900 alone), while VALU / LDS / VMEM instructions of
// the SAME wave hide in its own MFMA gaps as long as there are at most ~5 per 32-cycle MFMA.  So:
//   * operands come in through registers (global_load_dwordx4, two K tiles ahead: 96 staging registers), are split ONCE
//     per workgroup into NT bf16 images in LDS (5.5 VALU per element; per wave and K tile 264 VALU + 36 ds_write_b64 next
//     to 96 MFMAs of 32 cycles = 3.6 fillers per MFMA gap, where the fragment-splitting wide kernel has 5.6 per gap and
//     two waves per SIMD), and every wave reads its fragments (128 x 64 accumulator = 4 x 2 blocks of 32 x 32) by
//     ds_read_b128 from the images;
//   * images are double-buffered (2 x 72 KB with three terms), one s_barrier per K tile, in its middle:
//       part A of K tile s: 48 MFMAs on k 0-15 (fragment set F0)  || read F1 = k 16-31 of image s
//                                                                  || split + write units 6-11 of image s + 1, reload their registers with K tile s + 3
//       barrier (image s + 1 complete; image s no longer read)
//       part B:             48 MFMAs on k 16-31 (F1)               || read F0 = k 0-15 of image s + 1
//                                                                  || split + write units 0-5 of image s + 2 (into image s's buffer), reload with K tile s + 4
//   * image rows are 64 bytes (32 k), the 16-byte chunk c of row r sits at chunk c ^ ((r >> 2) & 3): fragment reads and
//     8-byte writes are conflict-free without padding.
// Shapes: K % 32 == 0, 16-byte aligned rows, M and N free (edge tiles clamp their source rows / columns and guard their
// stores), row-contiguous operands need a multiple of 4 rows: the wide kernels' conditions.
constexpr int SM = 256, SN = 128, SK = 32;
constexpr int SP_A = SM * SK, SP_B = SN * SK;          // bf16 elements of one image
#ifndef ASR_GS_ABL      /* measurement only: 1 no products, 2 no split arithmetic / image writes after the prologue, 4 no operand loads after the prologue, 8 no epilogue stores */
#define ASR_GS_ABL 0
#endif
#ifndef ASR_GS_PAT      /* filler pattern inside a slot (measurement) */
#define ASR_GS_PAT 0
#endif
#ifndef ASR_GS_NOBAR    /* measurement only (wrong results): no barrier in the K loop */
#define ASR_GS_NOBAR 0
#endif
#ifndef ASR_GS_SCHED    /* 1: pin the MFMA / filler interleave with sched_group_barrier */
#define ASR_GS_SCHED 3
#endif

typedef unsigned gu32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) gu32x2* lds_u2ptr;
typedef __attribute__((address_space(3))) const gu32x4* lds_q4ptr;
typedef __attribute__((address_space(3))) char* lds_cptr;
// an LDS byte address held in a 32-bit register -> a typed LDS pointer.  Through uintptr_t: the host pass of the HIP
// compile parses this code with 64-bit pointers and (rightly) objects to a cast from a narrower integer.
template <typename P>
__device__ __forceinline__ P lds_at(unsigned a) { return (P)(uintptr_t)a; }

// split four consecutive k of one row and store them into the NT images at LDS byte address a, a + plane, a + 2 plane
template <int NT>
__device__ __forceinline__ void bfs_write4(unsigned a, int plane, float x0, float x1, float x2, float x3) {
  unsigned p0[NT], p1[NT];
#if ASR_GS_ABL & 32      /* measurement: no split arithmetic (raw bits as terms) */
  p0[0] = __float_as_uint(x0); p0[1] = __float_as_uint(x1); p1[0] = __float_as_uint(x2); p1[1] = __float_as_uint(x3);
  if constexpr (NT > 2) { p0[2] = p0[0] ^ p1[1]; p1[2] = p0[1] ^ p1[0]; }
#else
  bfn_split2<NT>(x0, x1, p0);
  bfn_split2<NT>(x2, x3, p1);
#endif
#if ASR_GS_ABL & 16      /* measurement: no image writes (the split stays) */
  asm volatile("" ::"v"(p0[0]), "v"(p0[1]), "v"(p1[0]), "v"(p1[1]));
  if constexpr (NT > 2) asm volatile("" ::"v"(p0[2]), "v"(p1[2]));
  return;
#endif
  *lds_at<lds_u2ptr>(a) = gu32x2{p0[0], p1[0]};
  *lds_at<lds_u2ptr>(a + plane) = gu32x2{p0[1], p1[1]};
  if constexpr (NT > 2) *lds_at<lds_u2ptr>(a + 2 * plane) = gu32x2{p0[2], p1[2]};
}

template <bool AKC, bool BKC, int NT, bool KT>
__global__ __launch_bounds__(256, 1) void gemm_bfs_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * NT * (SP_A + SP_B)];
  constexpr int PLA = 2 * SP_A, PLB = 2 * SP_B;        // bytes of one A / B image
  constexpr int BUFB = NT * (PLA + PLB);               // bytes of one buffer: NT A images, then NT B images

  const int ntile = g.tiles_m * g.tiles_n;
  int tid = blockIdx.x;
  {
    const int q = ntile >> 3, rmd = ntile & 7, xcd = tid & 7, idx = tid >> 3;
    tid = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + idx;
  }
  int tm, tn;
  tile_coords(tid, g.tiles_m, g.tiles_n, ASR_GEMM_GM, tm, tn);
  const int z = blockIdx.y;
  const int bz = z / g.split_k, kz = z % g.split_k;
  float* C = g.C + bz * g.sC;
  const int64_t m0 = (int64_t)tm * SM, n0 = (int64_t)tn * SN;
  const int64_t ktiles = KT ? (g.K + SK - 1) / SK : g.K / SK;         // KT: K % 32 != 0 (K % 4 == 0): the last K tile is masked
  const int64_t per = (ktiles + g.split_k - 1) / g.split_k;
  const int64_t kt_begin = kz * per;
  const int64_t kt_end = kt_begin + per < ktiles ? kt_begin + per : ktiles;
  const int S = (int)(kt_end - kt_begin);
  if (S <= 0) return;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;

  // ---- operands through buffer resources: wave-uniform base (batch element, first k of this K slice) + this thread's
  // byte offset of each of its 8 + 4 float4 pieces (VGPR, computed once) + the K tile's byte offset (SGPR): no address
  // arithmetic in the loop.  Rows and columns past the edge are clamped to the last valid ones (they only reach
  // accumulators the guarded epilogue drops).
  const int64_t lda = g.A.ld, ldb = g.B.ld;
  const float* Ab = g.A.p + bz * g.sA + (AKC ? kt_begin * SK : kt_begin * SK * lda);
  const float* Bb = g.B.p + bz * g.sB + (BKC ? kt_begin * SK : kt_begin * SK * ldb);
  // the resources end with the operand: a fetch behind it (the K tail of the KT instantiations) returns zeros - the range
  // check covers the VGPR offset only, so those instantiations carry the K tile's offset there (load_piece)
  const int64_t enda = ((AKC ? (g.M - 1) * lda + g.K : (g.K - 1) * lda + g.M) - (AKC ? kt_begin * SK : kt_begin * SK * lda)) * 4;
  const int64_t endb = ((BKC ? (g.N - 1) * ldb + g.K : (g.K - 1) * ldb + g.N) - (BKC ? kt_begin * SK : kt_begin * SK * ldb)) * 4;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ab), 0, (int)(enda < 0x7ffffff0 ? enda : 0x7ffffff0), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bb), 0, (int)(endb < 0x7ffffff0 ? endb : 0x7ffffff0), 0x00020000);
  const int kloc = (int)(g.K - kt_begin * SK) - 4 * (t & 7);           // this lane's four k of K tile st are valid iff 32 st < kloc
  unsigned offa[8], offb[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (AKC) {
      const int r = (t >> 3) + 32 * i;
      const int64_t row = m0 + r < g.M ? m0 + r : g.M - 1;
      offa[i] = (unsigned)(row * lda + 4 * (t & 7)) * 4u;
    } else {
      const int64_t c0 = m0 + 128 * (i >> 2) + 4 * (t >> 3);
      const int64_t col = c0 + 4 <= g.M ? c0 : g.M - 4;
      offa[i] = (unsigned)((4 * (t & 7) + (i & 3)) * lda + col) * 4u;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (BKC) {
      const int r = (t >> 3) + 32 * i;
      const int64_t row = n0 + r < g.N ? n0 + r : g.N - 1;
      offb[i] = (unsigned)(row * ldb + 4 * (t & 7)) * 4u;
    } else {
      const int64_t c0 = n0 + 4 * (t >> 3);
      const int64_t col = c0 + 4 <= g.N ? c0 : g.N - 4;
      offb[i] = (unsigned)((4 * (t & 7) + i) * ldb + col) * 4u;
    }
  }
  const int stepa = (int)((AKC ? SK : SK * lda) * 4), stepb = (int)((BKC ? SK : SK * ldb) * 4);     // bytes per K tile

  // ---- LDS addresses (bytes).  Image rows are 64 bytes; the 16-byte chunk c of row r sits at chunk c ^ ((r >> 2) & 3).
  // One base per (buffer, operand, access kind) in a VGPR, everything else is an immediate offset of the DS instruction.
  const int kq = t & 7;
  const int woa = AKC ? (t >> 3) * 64 + ((((kq >> 1) ^ ((t >> 5) & 3))) << 4) + (kq & 1) * 8          // + 2048 i
                      : 4 * (t >> 3) * 64 + ((((kq >> 1) ^ ((t >> 3) & 3))) << 4) + (kq & 1) * 8;     // + 64 j + 8192 h
  const int wob = BKC ? (t >> 3) * 64 + ((((kq >> 1) ^ ((t >> 5) & 3))) << 4) + (kq & 1) * 8
                      : 4 * (t >> 3) * 64 + ((((kq >> 1) ^ ((t >> 3) & 3))) << 4) + (kq & 1) * 8;
  const int fsw = (l31 >> 2) & 3;       // fragment rows: wm*128 + 32 i + l31 (A), wn*64 + 32 j + l31 (B); chunk 2 ks + kh
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_cptr)smem;
  unsigned wA[2], wB[2], rA[2][2], rB[2][2];           // [buffer], [buffer][k half]
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    wA[b] = lds0 + b * BUFB + woa;
    wB[b] = lds0 + b * BUFB + NT * PLA + wob;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      rA[b][ks] = lds0 + b * BUFB + (wm * 128 + l31) * 64 + (((2 * ks + kh) ^ fsw) << 4);
      rB[b][ks] = lds0 + b * BUFB + NT * PLA + (wn * 64 + l31) * 64 + (((2 * ks + kh) ^ fsw) << 4);
      asm volatile("" : "+v"(rA[b][ks]), "+v"(rB[b][ks]));
    }
    asm volatile("" : "+v"(wA[b]), "+v"(wB[b]));
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  gu32x4 RA[2][8], RB[2][4];                 // staging registers (raw fp32 bits) of K tiles of even / odd parity
  gu32x4 FA[2][4][NT], FB[2][2][NT];         // fragment sets F0 (k 0-15), F1 (k 16-31)

  // load piece u (0-7: A, 8-11: B) of K tile st into register set Q
  auto load_piece = [&](auto qtag, auto utag, int st) __attribute__((always_inline)) {
    constexpr int Q = decltype(qtag)::value, U = decltype(utag)::value;
    if (ASR_GS_ABL & 4) return;
    if constexpr (KT) {          // the last K tile reaches behind the operand: offset in the VGPR, where the range check sees it
      if constexpr (U < 8) RA[Q][U] = __builtin_amdgcn_raw_buffer_load_b128(rsA, offa[U] + (unsigned)(st * stepa), 0, 0);
      else RB[Q][U - 8] = __builtin_amdgcn_raw_buffer_load_b128(rsB, offb[U - 8] + (unsigned)(st * stepb), 0, 0);
    } else {                     // every K tile is inside the operand (rows / columns are clamped in offa / offb)
      if constexpr (U < 8) RA[Q][U] = __builtin_amdgcn_raw_buffer_load_b128(rsA, offa[U], st * stepa, 0);
      else RB[Q][U - 8] = __builtin_amdgcn_raw_buffer_load_b128(rsB, offb[U - 8], st * stepb, 0);
    }
  };
  // split + write unit u of the K tile held in register set Q into buffer BUF; units 0-5: A 0-3, B 0-1; 6-11: A 4-7, B 2-3
  auto unit = [&](auto qtag, auto utag, auto btag, int stu) __attribute__((always_inline)) {
    constexpr int Q = decltype(qtag)::value, U = decltype(utag)::value, BUF = decltype(btag)::value;
    const bool kok = !KT || 32 * stu < kloc;
#define BFS_F(v_, e_) (kok ? __uint_as_float((v_)[e_]) : 0.f)
    constexpr bool isA = (U % 6) < 4;
    constexpr int idx = isA ? (U / 6) * 4 + (U % 6) : (U / 6) * 2 + (U % 6) - 4;      // A piece 0-7 / B piece 0-3
    if (ASR_GS_ABL & 2) return;
    if constexpr (isA) {
      if constexpr (AKC) {
        const gu32x4 v = RA[Q][idx];
        bfs_write4<NT>(wA[BUF] + 2048 * idx, PLA, BFS_F(v, 0), BFS_F(v, 1), BFS_F(v, 2), BFS_F(v, 3));
      } else {
        constexpr int h = idx >> 2, j = idx & 3;
        bfs_write4<NT>(wA[BUF] + 64 * j + 8192 * h, PLA, BFS_F(RA[Q][4 * h], j), BFS_F(RA[Q][4 * h + 1], j), BFS_F(RA[Q][4 * h + 2], j),
                       BFS_F(RA[Q][4 * h + 3], j));
      }
    } else {
      if constexpr (BKC) {
        const gu32x4 v = RB[Q][idx];
        bfs_write4<NT>(wB[BUF] + 2048 * idx, PLB, BFS_F(v, 0), BFS_F(v, 1), BFS_F(v, 2), BFS_F(v, 3));
      } else {
        constexpr int j = idx;
        bfs_write4<NT>(wB[BUF] + 64 * j, PLB, BFS_F(RB[Q][0], j), BFS_F(RB[Q][1], j), BFS_F(RB[Q][2], j), BFS_F(RB[Q][3], j));
      }
    }
  };
#undef BFS_F
  // after unit u of set Q has been written: reload the registers it (and, for row-contiguous operands, its group) used
  auto reload = [&](auto qtag, auto utag, int st) __attribute__((always_inline)) {
    constexpr int Q = decltype(qtag)::value, U = decltype(utag)::value;
    constexpr bool isA = (U % 6) < 4;
    constexpr int idx = isA ? (U / 6) * 4 + (U % 6) : (U / 6) * 2 + (U % 6) - 4;
    typedef std::integral_constant<int, Q> QT;
    if constexpr (isA) {
      if constexpr (AKC) load_piece(QT(), std::integral_constant<int, idx>(), st);
      else if constexpr ((idx & 3) == 3) {
        load_piece(QT(), std::integral_constant<int, idx - 3>(), st);
        load_piece(QT(), std::integral_constant<int, idx - 2>(), st);
        load_piece(QT(), std::integral_constant<int, idx - 1>(), st);
        load_piece(QT(), std::integral_constant<int, idx>(), st);
      }
    } else {
      if constexpr (BKC) load_piece(QT(), std::integral_constant<int, 8 + idx>(), st);
      else if constexpr (idx == 3) {
        load_piece(QT(), std::integral_constant<int, 8>(), st);
        load_piece(QT(), std::integral_constant<int, 9>(), st);
        load_piece(QT(), std::integral_constant<int, 10>(), st);
        load_piece(QT(), std::integral_constant<int, 11>(), st);
      }
    }
  };
  // fragment reads of slot q (0-3: A block q, 4-5: B block q - 4) of k half KS from buffer BUF
  auto frag_read = [&](auto kstag, auto qtag, auto btag) __attribute__((always_inline)) {
    constexpr int KS = decltype(kstag)::value, Qs = decltype(qtag)::value, BUF = decltype(btag)::value;
    if constexpr (Qs < 4) {
      const unsigned a = rA[BUF][KS] + Qs * 2048;
      FA[KS][Qs][0] = *lds_at<lds_q4ptr>(a);
      FA[KS][Qs][1] = *lds_at<lds_q4ptr>(a + PLA);
      if constexpr (NT > 2) FA[KS][Qs][2] = *lds_at<lds_q4ptr>(a + 2 * PLA);
    } else {
      const unsigned a = rB[BUF][KS] + (Qs - 4) * 2048;
      FB[KS][Qs - 4][0] = *lds_at<lds_q4ptr>(a);
      FB[KS][Qs - 4][1] = *lds_at<lds_q4ptr>(a + PLB);
      if constexpr (NT > 2) FB[KS][Qs - 4][2] = *lds_at<lds_q4ptr>(a + 2 * PLB);
    }
  };
#define BFS(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gbf16x8, a_), __builtin_bit_cast(gbf16x8, b_), c_, 0, 0, 0)
  // the 8 MFMAs of term pair number tp (in order of significance sum) of k half KS
  auto products = [&](auto kstag, auto tptag) __attribute__((always_inline)) {
    constexpr int KS = decltype(kstag)::value, TP = decltype(tptag)::value;
    constexpr int o = TP == 0 ? 0 : TP < 3 ? 1 : 2, p = TP == 0 ? 0 : TP < 3 ? TP - 1 : TP - 3;
    if (ASR_GS_ABL & 1) return;
    if constexpr (o < NT) {
      BFS(FA[KS][0][p], FB[KS][0][o - p], acc[0][0]); BFS(FA[KS][0][p], FB[KS][1][o - p], acc[0][1]);
      BFS(FA[KS][1][p], FB[KS][0][o - p], acc[1][0]); BFS(FA[KS][1][p], FB[KS][1][o - p], acc[1][1]);
      BFS(FA[KS][2][p], FB[KS][0][o - p], acc[2][0]); BFS(FA[KS][2][p], FB[KS][1][o - p], acc[2][1]);
      BFS(FA[KS][3][p], FB[KS][0][o - p], acc[3][0]); BFS(FA[KS][3][p], FB[KS][1][o - p], acc[3][1]);
    }
  };
  // One half of a K tile: 48 MFMAs of k half KS, slot by slot with the fragment reads of the other k half (from buffer RD),
  // six split + write units (register set UQ, first unit U0 = 0 or 6, into buffer WR) and the reloads of their registers
  // with K tile ld_st.  Each slot's instruction mix is pinned: 1 MFMA : 3 VALU, LDS reads early, LDS writes late.
  auto half = [&](auto kstag, auto uqtag, auto u0tag, auto rdtag, bool do_rd, auto wrtag, bool do_wr, int stu, int ld_st, bool do_ld)
      __attribute__((always_inline)) {
    constexpr int KS = decltype(kstag)::value, UQ = decltype(uqtag)::value, U0 = decltype(u0tag)::value;
    typedef std::integral_constant<int, KS> KST;
    typedef std::integral_constant<int, 1 - KS> KSN;
    typedef std::integral_constant<int, UQ> UQT;
#define BFS_SLOT(q_)                                                                                     \
    products(KST(), std::integral_constant<int, q_>());                                                  \
    if (do_rd) frag_read(KSN(), std::integral_constant<int, q_>(), rdtag);                               \
    if (do_wr) unit(UQT(), std::integral_constant<int, U0 + q_>(), wrtag, stu);                               \
    if (do_ld) reload(UQT(), std::integral_constant<int, U0 + q_>(), ld_st);                             \
    if (ASR_GS_SCHED & 2) {                                                                              \
      _Pragma("unroll") for (int m_ = 0; m_ < 8; ++m_) {                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                               \
        if (ASR_GS_PAT == 0) {                                                                           \
          if (m_ < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                 \
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                             \
          if (m_ >= 5) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                \
        } else if (ASR_GS_PAT == 1) {                                                                    \
          if (m_ < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                 \
          if (m_ < 6) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                 \
          if (m_ >= 5) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                \
        } else if (ASR_GS_PAT == 2) {                                                                    \
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                             \
        } else if (ASR_GS_PAT == 3) {                                                                    \
          if (m_ == 0) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);                                \
          if (m_ < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                 \
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                             \
          if (m_ >= 4 && m_ < 7) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                      \
        }                                                                                                \
      }                                                                                                  \
      if (ASR_GS_PAT != 3) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);                            \
    }                                                                                                    \
    if (ASR_GS_SCHED & 1) __builtin_amdgcn_sched_barrier(0);
    BFS_SLOT(0) BFS_SLOT(1) BFS_SLOT(2) BFS_SLOT(3) BFS_SLOT(4) BFS_SLOT(5)
#undef BFS_SLOT
  };
  auto barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  typedef std::integral_constant<int, 6> I6;

  // ---- prologue: K tiles 0 and 1 into the registers, image 0 whole, units 0-5 of image 1
  {
#define BFS_LD(q_, u_, st_) load_piece(std::integral_constant<int, q_>(), std::integral_constant<int, u_>(), st_)
    BFS_LD(0, 0, 0); BFS_LD(0, 1, 0); BFS_LD(0, 2, 0); BFS_LD(0, 3, 0); BFS_LD(0, 4, 0); BFS_LD(0, 5, 0);
    BFS_LD(0, 6, 0); BFS_LD(0, 7, 0); BFS_LD(0, 8, 0); BFS_LD(0, 9, 0); BFS_LD(0, 10, 0); BFS_LD(0, 11, 0);
    if (S > 1) {
      BFS_LD(1, 0, 1); BFS_LD(1, 1, 1); BFS_LD(1, 2, 1); BFS_LD(1, 3, 1); BFS_LD(1, 4, 1); BFS_LD(1, 5, 1);
      BFS_LD(1, 6, 1); BFS_LD(1, 7, 1); BFS_LD(1, 8, 1); BFS_LD(1, 9, 1); BFS_LD(1, 10, 1); BFS_LD(1, 11, 1);
    }
#undef BFS_LD
#define BFS_UN(q_, u_, st_) do { unit(std::integral_constant<int, q_>(), std::integral_constant<int, u_>(), std::integral_constant<int, q_>(), q_);   \
      if ((st_) < S) reload(std::integral_constant<int, q_>(), std::integral_constant<int, u_>(), st_); } while (0)
    BFS_UN(0, 0, 2); BFS_UN(0, 1, 2); BFS_UN(0, 2, 2); BFS_UN(0, 3, 2); BFS_UN(0, 4, 2); BFS_UN(0, 5, 2);
    BFS_UN(0, 6, 2); BFS_UN(0, 7, 2); BFS_UN(0, 8, 2); BFS_UN(0, 9, 2); BFS_UN(0, 10, 2); BFS_UN(0, 11, 2);
    if (S > 1) {
      BFS_UN(1, 0, 3); BFS_UN(1, 1, 3); BFS_UN(1, 2, 3); BFS_UN(1, 3, 3); BFS_UN(1, 4, 3); BFS_UN(1, 5, 3);
    }
#undef BFS_UN
    barrier();
    frag_read(I0(), std::integral_constant<int, 0>(), I0()); frag_read(I0(), std::integral_constant<int, 1>(), I0());
    frag_read(I0(), std::integral_constant<int, 2>(), I0()); frag_read(I0(), std::integral_constant<int, 3>(), I0());
    frag_read(I0(), std::integral_constant<int, 4>(), I0()); frag_read(I0(), std::integral_constant<int, 5>(), I0());
  }
  // K tile st (parity P = st & 1; image st in buffer P): part A, barrier, part B
  auto ktile = [&](int st, auto ptag, auto ftag) __attribute__((always_inline)) {
    constexpr int P = decltype(ptag)::value;
    constexpr bool FULL = decltype(ftag)::value;            // K tiles st + 1 ... st + 4 exist: no conditionals in the body
    typedef std::integral_constant<int, P> PT;
    typedef std::integral_constant<int, 1 - P> PN;
    const bool n1 = FULL || st + 1 < S, n2 = FULL || st + 2 < S;
    half(I0(), PN(), I6(), PT(), true, PN(), n1, st + 1, st + 3, FULL || st + 3 < S);
    if (n1 && !ASR_GS_NOBAR) barrier();
    half(I1(), PT(), I0(), PN(), n1, PT(), n2, st + 2, st + 4, FULL || st + 4 < S);
  };
  typedef std::integral_constant<bool, true> Full;
  typedef std::integral_constant<bool, false> Tail;
  int st = 0;
  for (; st + 5 < S; st += 2) {
    ktile(st, I0(), Full());
    ktile(st + 1, I1(), Full());
  }
  for (; st < S; st += 2) {
    ktile(st, I0(), Tail());
    if (st + 1 < S) ktile(st + 1, I1(), Tail());
  }
#undef BFS

#if ASR_GS_ABL & 8
  if (threadIdx.x + blockIdx.x + blockIdx.y != 0 || g.M != 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(acc[i][j][e]));
    return;
  }
#endif
  // ---- epilogue: lane owns column n, rows (e & 3) + 8 (e >> 2) + 4 kh of each 32 x 32 block
  const int64_t mrow0 = m0 + wm * 128, ncol0 = n0 + wn * 64;
  const unsigned ldc = (unsigned)g.ldc;
  const bool split = g.split_k > 1, accum = g.accumulate != 0, relu = g.relu != 0;
  const bool edge = !(m0 + SM <= g.M && n0 + SN <= g.N);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t n = ncol0 + j * 32 + l31;
    const bool nok = !edge || n < g.N;
    const float bv = (g.bias && nok) ? g.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t mb = mrow0 + i * 32 + 4 * kh;
      float* base = C + mb * g.ldc + n;
      if (!edge) {
        if (split) {
#pragma unroll
          for (int e = 0; e < 16; ++e) atomicAdd(base + (unsigned)((e & 3) + 8 * (e >> 2)) * ldc, acc[i][j][e]);
        } else if (accum) {
          float old[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) old[e] = base[(unsigned)((e & 3) + 8 * (e >> 2)) * ldc];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float v = acc[i][j][e] + bv + old[e];
            if (relu) v = fmaxf(v, 0.f);
            base[(unsigned)((e & 3) + 8 * (e >> 2)) * ldc] = v;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float v = acc[i][j][e] + bv;
            if (relu) v = fmaxf(v, 0.f);
            base[(unsigned)((e & 3) + 8 * (e >> 2)) * ldc] = v;
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = (e & 3) + 8 * (e >> 2);
          if (!(nok && mb + r < g.M)) continue;
          float* dst = base + (unsigned)r * ldc;
          if (split) {
            atomicAdd(dst, acc[i][j][e]);
          } else {
            float o = acc[i][j][e] + bv;
            if (accum) o += *dst;
            if (relu) o = fmaxf(o, 0.f);
            *dst = o;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------ split-bf16, short K, weights stationary
// gemm_bfk_kernel: C[M, N] = A[M, K] B[N, K]^T (+ bias) for a SHORT contraction, K = 16 KS <= 96 - the layer-0 input
// projection of the encoder (K = 80 features, M = T B = 25 600 rows, N = 8H = 4 096: 16.8 GFLOP against 420 MB of output,
// i.e. bound by the OUTPUT STREAM - the chip stores 6.3-7.2 TB/s, ~10-13 B / cycle and CU when all 256 store at once (tools/store_probe.hip) - not by the matrix pipe).
// The general kernels spend such a product on prologues: three K tiles, then the tile's output with nothing to overlap it.
// Here a workgroup (8 waves = 2 (M) x 4 (N), two per SIMD: while one waits in the store queue the other issues) keeps ONE
// 128-column tile of B for its whole life - split once, every wave's 32 x K slice of it as fragments in 12 KS registers -
// and walks down the M tiles of its column: per tile of 128 rows the wave's 12 KS MFMAs (64 x 32 accumulator) run over
// the stores of the PREVIOUS tile's accumulators (two accumulator sets), the split + image writes of the NEXT tile's A
// rows (two LDS images of 128 x K) and the loads of the one after that; one barrier per tile.
// Workgroup w: XCD w % 8 = group of M tiles (tiles g, g + G, ...), w / G = column tile: the 32 column tiles that read one
// A tile run on one XCD at about the same time (one HBM read per XCD).  Image rows are padded to 2 K + 16 bytes:
// 11 r mod 16 is a bijection, so a fragment read's 16 lanes hit 16 different bank quads.
#ifndef ASR_GK_ABL      /* measurement only: 1 no products, 2 no split / image writes in the loop, 8 no output stores */
#define ASR_GK_ABL 0
#endif
template <int KS> struct BfkDims {
  static constexpr int K = 16 * KS, ROWB = 2 * K + 16, PLANE = 128 * ROWB, HP = K / 16;   // HP float4 pieces per thread and tile
};

// PLAIN: every tile is interior (M % 128 == 0, N % 128 == 0) and the epilogue is bias only - straight-line stores, no branches
template <int NT, int KS, bool PLAIN>
__global__ __launch_bounds__(512) void gemm_bfk_kernel(GemmArgs g, int groups) {
  typedef BfkDims<KS> D;
  constexpr int ROWB = D::ROWB, PLANE = D::PLANE, HP = D::HP, BUFB = NT * PLANE;
  static_assert(KS >= 4 && KS <= 6, "the previous tile's accumulators are stored under k steps 0..3");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * NT * D::PLANE];

  const int grp = blockIdx.x % groups, tn = blockIdx.x / groups;
  const int bz = blockIdx.y;
  const int tiles_m = (int)((g.M + 127) / 128);
  const int ntile = grp < tiles_m ? (tiles_m - grp + groups - 1) / groups : 0;
  if (ntile <= 0) return;
  const int64_t n0 = (int64_t)tn * 128;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;               // 2 (M) x 4 (N) waves: 64 rows x 32 columns each
  const int l31 = lane & 31, kh = lane >> 5;
  const int64_t lda = g.A.ld, ldb = g.B.ld;
  const float* Ab = g.A.p + bz * g.sA;
  const float* Bb = g.B.p + bz * g.sB + n0 * ldb;
  float* C = g.C + bz * g.sC;
  // resources end with the operands: rows past M / N read as zeros.  The hardware's range check covers the VGPR offset
  // plus the instruction offset only - NOT the SGPR offset - so everything that can leave the operand (the M tile of A)
  // goes into the VGPR offset (load_a); B's rows past N are inside pb0.
  const int64_t enda = ((g.M - 1) * lda + g.K) * 4, endb = ((g.N - 1 - n0) * ldb + g.K) * 4;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ab), 0, (int)(enda < 0x7ffffff0 ? enda : 0x7ffffff0), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bb), 0, (int)(endb < 0x7ffffff0 ? endb : 0x7ffffff0), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(C, 0, 0x7ffffff0, 0x00020000);
  // piece u (of HP) of a tile: row t >> 2, float4 number (t & 3) HP + u of the row's K / 4: ONE base offset per operand in a
  // VGPR, the rest are immediate offsets of the buffer / DS instructions
  const unsigned pa0 = (unsigned)((t >> 2) * lda + 4 * (t & 3) * HP) * 4u;
  const unsigned pb0 = (unsigned)((t >> 2) * ldb + 4 * (t & 3) * HP) * 4u;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_cptr)smem;
  unsigned wimg[2], rimg[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    wimg[b] = lds0 + b * BUFB + (unsigned)((t >> 2) * ROWB + 8 * (t & 3) * HP);
    rimg[b] = lds0 + b * BUFB + (wm * 64 + l31) * ROWB + 16 * kh;
    asm volatile("" : "+v"(wimg[b]), "+v"(rimg[b]));
  }
  const int tile_step = (int)(128 * lda * 4);            // bytes between consecutive M tiles

  gu32x4 RA[2][HP];               // staging registers of two A tiles (tile parity)
  gu32x4 FB[KS][NT];              // this wave's B fragments (32 columns): [k step][term]
  gu32x4 FA[2][2][NT];            // A fragments of one k step, double buffered: [slot][32-row block][term]
  f32x16 acc[2][2];               // [set][row block]

  auto load_a = [&](auto qtag, auto utag, int mt) __attribute__((always_inline)) {
    constexpr int Q = decltype(qtag)::value, U = decltype(utag)::value;
    RA[Q][U] = __builtin_amdgcn_raw_buffer_load_b128(rsA, pa0 + (unsigned)(mt * tile_step) + 16 * U, 0, 0);
  };
  auto unit = [&](auto qtag, auto utag, auto btag) __attribute__((always_inline)) {       // split + write piece U of staged tile Q into image BUF
    constexpr int Q = decltype(qtag)::value, U = decltype(utag)::value, BUF = decltype(btag)::value;
    const gu32x4 v = RA[Q][U];
    bfs_write4<NT>(wimg[BUF] + 8 * U, PLANE, __uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
  };
  auto read_fa = [&](auto slottag, auto kstag, auto btag) __attribute__((always_inline)) {
    constexpr int SL = decltype(slottag)::value, KSI = decltype(kstag)::value, BUF = decltype(btag)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned a = rimg[BUF] + i * 32 * ROWB + 32 * KSI;
      FA[SL][i][0] = *lds_at<lds_q4ptr>(a);
      FA[SL][i][1] = *lds_at<lds_q4ptr>(a + PLANE);
      if constexpr (NT > 2) FA[SL][i][2] = *lds_at<lds_q4ptr>(a + 2 * PLANE);
    }
  };
#define BFK(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gbf16x8, a_), __builtin_bit_cast(gbf16x8, b_), c_, 0, 0, 0)
  auto products = [&](auto slottag, auto kstag, auto settag) __attribute__((always_inline)) {
    constexpr int SL = decltype(slottag)::value, KSI = decltype(kstag)::value, P = decltype(settag)::value;
    if (ASR_GK_ABL & 1) return;
#pragma unroll
    for (int o = 0; o < NT; ++o)
#pragma unroll
      for (int p = 0; p <= o; ++p) {
        BFK(FA[SL][0][p], FB[KSI][o - p], acc[P][0]);
        BFK(FA[SL][1][p], FB[KSI][o - p], acc[P][1]);
      }
  };
#undef BFK
  const bool accum = g.accumulate != 0, relu = g.relu != 0;
  const int64_t ncol = n0 + wn * 32 + l31;
  const float bv = (g.bias && ncol < g.N) ? g.bias[ncol] : 0.f;
  const unsigned cvo = (unsigned)((4 * kh) * g.ldc + ncol) * 4u;        // byte offset of (row 4 kh, this lane's column)
  const int ldcb = (int)(g.ldc * 4);
  // store half HF (accumulator elements 8 HF .. 8 HF + 7) of row block I of accumulator set P = tile mt's rows wm*64 + 32 I ..
  auto store_half = [&](auto settag, auto itag, auto hftag, int mt) __attribute__((always_inline)) {
    constexpr int P = decltype(settag)::value, I = decltype(itag)::value, HF = decltype(hftag)::value;
    if (ASR_GK_ABL & 8) {
#pragma unroll
      for (int e = 8 * HF; e < 8 * HF + 8; ++e) asm volatile("" ::"v"(acc[P][I][e]));
    } else if constexpr (PLAIN) {
      // rows through the SGPR offset of the buffer store, this lane's column in the VGPR offset: no address arithmetic
      const int row0 = (mt * 128 + wm * 64 + I * 32) * ldcb;
#pragma unroll
      for (int e = 8 * HF; e < 8 * HF + 8; ++e)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[P][I][e] + bv), rsC, cvo, row0 + ((e & 3) + 8 * (e >> 2)) * ldcb, 0);
    } else {
      const int64_t mb = (int64_t)mt * 128 + wm * 64 + I * 32 + 4 * kh;
      float* base = C + mb * g.ldc + ncol;
#pragma unroll
      for (int e = 8 * HF; e < 8 * HF + 8; ++e) {
        const int r = (e & 3) + 8 * (e >> 2);
        if (ncol < g.N && mb + r < g.M) {
          float* dst = base + (unsigned)r * (unsigned)g.ldc;
          float v = acc[P][I][e] + bv;
          if (accum) v += *dst;
          if (relu) v = fmaxf(v, 0.f);
          *dst = v;
        }
      }
    }
#pragma unroll
    for (int e = 8 * HF; e < 8 * HF + 8; ++e) acc[P][I][e] = 0.f;
  };
  auto barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
#define BFK_FOR_PIECES(X) do { X(0); X(1); X(2); X(3); if constexpr (HP > 4) { X(4); } if constexpr (HP > 5) { X(5); } } while (0)

  // ---- prologue: B tile -> image 1 -> fragments in registers; A tile 0 -> image 0; A tile 1 staged
#define X_LDB(u_) RA[1][u_] = __builtin_amdgcn_raw_buffer_load_b128(rsB, pb0 + 16 * (u_), 0, 0)
  BFK_FOR_PIECES(X_LDB);
#undef X_LDB
#define X_UNB(u_) unit(I1(), std::integral_constant<int, u_>(), I1())
  BFK_FOR_PIECES(X_UNB);
#undef X_UNB
#define X_LDA0(u_) load_a(I0(), std::integral_constant<int, u_>(), grp)
  BFK_FOR_PIECES(X_LDA0);
#undef X_LDA0
  barrier();
  {
    const unsigned bbase = lds0 + BUFB + (wn * 32 + l31) * ROWB + 16 * kh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const unsigned a = bbase + 32 * ks;
      FB[ks][0] = *lds_at<lds_q4ptr>(a);
      FB[ks][1] = *lds_at<lds_q4ptr>(a + PLANE);
      if constexpr (NT > 2) FB[ks][2] = *lds_at<lds_q4ptr>(a + 2 * PLANE);
    }
  }
#define X_UNA0(u_) unit(I0(), std::integral_constant<int, u_>(), I0())
  BFK_FOR_PIECES(X_UNA0);
#undef X_UNA0
  if (ntile > 1) {          // tile k's rows are staged in register set k & 1
#define X_LDA1(u_) load_a(I1(), std::integral_constant<int, u_>(), grp + groups)
    BFK_FOR_PIECES(X_LDA1);
#undef X_LDA1
  }
  barrier();                                            // image 0 complete; every wave has its B fragments (image 1 is free)
  read_fa(I0(), I0(), I0());
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[p][i][e] = 0.f;

  // ---- one M tile: k steps 0 .. KS-1 on accumulator set P / image P, around them the previous tile's stores (set 1 - P),
  // the next tile's image (1 - P) and the loads of the tile after it
  auto mtile = [&](int k, auto ptag) __attribute__((always_inline)) {
    constexpr int P = decltype(ptag)::value;
    typedef std::integral_constant<int, P> PT;
    typedef std::integral_constant<int, 1 - P> PN;
    const int mt = grp + k * groups;
    const bool has_prev = k > 0, has_next = k + 1 < ntile, has_next2 = k + 2 < ntile;
    // The loads of tile k + 2 go first, into the register set tile k's rows have left (the current tile's image was written
    // during the previous tile): vmcnt retires loads and stores in issue order, so a wait for these loads also waits for
    // every store issued before them - placed ahead of this tile's stores they only wait for stores of a tile ago.
    if (has_next2 && !(ASR_GK_ABL & 2)) {
#define X_LDA2(u_) load_a(PT(), std::integral_constant<int, u_>(), mt + 2 * groups)
      BFK_FOR_PIECES(X_LDA2);
#undef X_LDA2
    }
    // the HP units of the next tile are spread over k steps 0 .. KS-2; the barrier sits in front of the last k step
#define BFK_UNIT(u_) { unit(PN(), std::integral_constant<int, u_>(), PN()); }
#define BFK_STEP(ks_)                                                                                                   \
    {                                                                                                                   \
      typedef std::integral_constant<int, ((ks_) + P * KS) & 1> SL;      /* fragment slots alternate across tiles too (odd KS) */ \
      typedef std::integral_constant<int, ((ks_) + 1 + P * KS) & 1> SN;                                                 \
      if ((ks_) == KS - 1 && has_next) barrier();                                                                       \
      products(SL(), std::integral_constant<int, ks_>(), PT());                                                         \
      if ((ks_) + 1 < KS) read_fa(SN(), std::integral_constant<int, ((ks_) + 1 < KS ? (ks_) + 1 : 0)>(), PT());        \
      else if (has_next) read_fa(SN(), I0(), PN());                                                                     \
      if constexpr ((ks_) < KS - 1) {          /* (constexpr: the discarded branch would index past RA / acc) */         \
        if (has_next && !(ASR_GK_ABL & 2)) {                                                                            \
          constexpr int u0 = (ks_) * HP / (KS - 1), u1 = ((ks_) + 1) * HP / (KS - 1);                                   \
          if constexpr (u0 < u1) BFK_UNIT(u0)                                                                           \
          if constexpr (u0 + 1 < u1) BFK_UNIT(u0 + 1)                                                                   \
          if constexpr (u0 + 2 < u1) BFK_UNIT(u0 + 2)                                                                   \
        }                                                                                                               \
      }                                                                                                                 \
      if constexpr ((ks_) < 4) {                                                                                        \
        if (has_prev) store_half(PN(), std::integral_constant<int, ((ks_) / 2)>(), std::integral_constant<int, (ks_) & 1>(), mt - groups); \
      }                                                                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                                \
    }
    BFK_STEP(0) BFK_STEP(1) BFK_STEP(2) BFK_STEP(3)
    if constexpr (KS > 4) BFK_STEP(4)
    if constexpr (KS > 5) BFK_STEP(5)
#undef BFK_STEP
#undef BFK_UNIT
  };
  int k = 0;
  for (; k + 1 < ntile; k += 2) {
    mtile(k, I0());
    mtile(k + 1, I1());
  }
  if (k < ntile) mtile(k, I0());
  // the last tile's accumulators
  {
    const int klast = ntile - 1, mt = grp + klast * groups;
    if (klast & 1) {
      store_half(I1(), I0(), I0(), mt); store_half(I1(), I0(), I1(), mt); store_half(I1(), I1(), I0(), mt); store_half(I1(), I1(), I1(), mt);
    } else {
      store_half(I0(), I0(), I0(), mt); store_half(I0(), I0(), I1(), mt); store_half(I0(), I1(), I0(), mt); store_half(I0(), I1(), I1(), mt);
    }
  }
#undef BFK_FOR_PIECES
}

// bias (+ReLU) pass after a split-K product (the atomics cannot carry an epilogue); with drop.thresh != 0 the seeded
// dropout mask over the element index m N + n (asr_dropout_seeded_f32's) goes into the same pass (asr_gemm_drop_f32)
struct DropEpi {
  unsigned long long seed;
  unsigned thresh;
  float scale;
};
__global__ void bias_act_kernel(float* C, int64_t ldc, int64_t M, int64_t N, int64_t sC, const float* __restrict__ bias,
                                int relu, DropEpi drop) {
  float* p = C + blockIdx.z * sC;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < M * N; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / N, n = i % N;
    float v = p[m * ldc + n] + (bias ? bias[n] : 0.f);
    if (relu) v = fmaxf(v, 0.f);
    if (drop.thresh) v = asr_drop_keep(drop.seed, i, drop.thresh) ? v * drop.scale : 0.f;
    p[m * ldc + n] = v;
  }
}

__global__ void zero_rows_kernel(float* C, int64_t ldc, int64_t M, int64_t N, int64_t sC) {
  float* p = C + blockIdx.z * sC;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < M * N; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / N, n = i % N;
    p[m * ldc + n] = 0.f;
  }
}

// ---------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(256) void skinny_kernel(int64_t M, int64_t N, int K, const float* A, int64_t lda,
                                                     const float* Bt, int64_t ldb, float* C, int64_t ldc,
                                                     const float* bias, int accumulate, const float* mask,
                                                     int64_t ldmask, int64_t mask_from, int ksplit) {
  __shared__ float red[4 * MT * 16 * SK_LDS_STRIDE];
  const int64_t n0 = (int64_t)blockIdx.x * 16;
  const int64_t row0 = (int64_t)blockIdx.y * (MT * 16);
  // ksplit > 1 (only with accumulate): grid.z slices K, partial products are added with f32 atomics
  const int kper = K / ksplit;
  A += (int64_t)blockIdx.z * kper;
  Bt += (int64_t)blockIdx.z * kper;
  K = kper;
  // epilogue operands (bias, dropout mask, accumulate-into value) are fetched before the product
  constexpr int NE = MT;                      // MT*256 outputs / 256 threads
  float bv[NE], mv[NE], cv[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = threadIdx.x + 256 * i;
    const int row = e >> 4, col = e & 15;
    const int64_t m = row0 + row, n = n0 + col;
    const bool ok = m < M && n < N;
    const int64_t mc = ok ? m : 0, nc = ok ? n : 0;
    bv[i] = bias ? bias[nc] : 0.f;
    mv[i] = (mask && nc >= mask_from) ? mask[mc * ldmask + (nc - mask_from)] : 1.f;
    cv[i] = (accumulate && ksplit == 1) ? C[mc * ldc + nc] : 0.f;
  }
  skinny_partial<MT>(A, lda, row0, M, Bt, ldb, n0, N, K, red);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = threadIdx.x + 256 * i;
    const int row = e >> 4, col = e & 15;
    const int64_t m = row0 + row, n = n0 + col;
    if (m >= M || n >= N) continue;
    const float v = (skinny_reduced<MT>(red, row, col) + bv[i]) * mv[i];
    if (ksplit > 1) atomicAdd(C + m * ldc + n, v);
    else C[m * ldc + n] = v + cv[i];
  }
}

__global__ void colsum_kernel(int64_t M, int64_t N, const float* X, int64_t ldx, float* out) {
  // block = 256 threads = 64 columns x 4 row groups; grid.y strides over 512-row chunks
  __shared__ float part[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * 64 + cx;
  const int64_t mbeg = (int64_t)blockIdx.y * 512;
  const int64_t mend = mbeg + 512 < M ? mbeg + 512 : M;
  float s = 0.f;
  if (n < N)
    for (int64_t m = mbeg + ry; m < mend; m += 4) s += X[m * ldx + n];
  part[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && n < N) atomicAdd(out + n, (part[0][cx] + part[1][cx]) + (part[2][cx] + part[3][cx]));
}

// float4 variant (N, ldx multiples of 4, 16-byte aligned base): block = 16 column quads x 16 row groups over a
// 256-row chunk, 16 independent 16-byte loads per thread; ~3x the bandwidth of the scalar kernel, whose 512-row
// chunks also left half the CUs without a workgroup on the [12800 x 512] bias gradients
__global__ __launch_bounds__(256) void colsum4_kernel(int64_t M, int64_t N, const float* __restrict__ X, int64_t ldx,
                                                      float* out) {
  __shared__ float4 part[16][16];
  const int cq = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int64_t n = (int64_t)blockIdx.x * 64 + 4 * cq;
  const int64_t mbeg = (int64_t)blockIdx.y * 256;
  const int64_t mend = mbeg + 256 < M ? mbeg + 256 : M;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n < N) {
    const float* p = X + n;
#pragma unroll 4
    for (int64_t m = mbeg + ry; m < mend; m += 16) {
      const float4 v = *reinterpret_cast<const float4*>(p + m * ldx);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  part[ry][cq] = s;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int q = threadIdx.x >> 2, e = threadIdx.x & 3;
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += reinterpret_cast<const float*>(&part[r][q])[e];
    const int64_t col = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (col < N) atomicAdd(out + col, t);
  }
}

}  // namespace

extern "C" int asr_abi_version(void) { return ASR_ABI_VERSION; }

#ifdef ASR_GW_TRACE
extern "C" int asr_gw_trace_read(void* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(asr_gw_trace_buf), sizeof(unsigned long long) * 2 * 64 * 8);
}
#endif

// Split-K factor of the 128 x 128 kernels when the caller passes split_k <= 0 (partials are added with f32 atomics; a
// bias / ReLU epilogue then needs a second pass over C).  256 CUs hold two 128 x 128 workgroups each, so the target is
// ~512 workgroups: floor(512 / tiles), at most 8 (16 for <= 32 tiles), every K slice at least 256 long.  Measured with
// cold operands (tools/gemm_cold_split_sweep.py): 400 tiles -> 1, 200 -> 2, 144 -> 3, 128 -> 4, 64 -> 8.
static int narrow_split_k(int64_t M, int64_t N, int64_t K, int batch, bool epilogue, int ts = 128) {
  const int64_t tiles = ((M + ts - 1) / ts) * ((N + ts - 1) / ts) * batch;
  const int64_t target = 512;       // (64 x 64 tiles, tools/gemm_small_sweep.py: 400 tiles unsplit 28 us, split in two 41)
  if (tiles >= target || K < 512) return 1;
  int64_t sk = tiles <= 32 ? 16 : 8;
  if (target / tiles < sk) sk = target / tiles;
  if (K / 256 < sk) sk = K / 256;
  if (sk < 1) sk = 1;
  if (epilogue && sk > 1 && tiles > 64 && K < 2048) return 1;     // (K >= 2048: 100 tiles x 64 serial K tiles 134 us, split in 4-5 + the two passes 80)
  return (int)sk;
}

template <int NT, int TS = 128>
static void launch_narrow_split(bool akc, bool bkc, dim3 grid, hipStream_t stream, const GemmArgs& g) {
  const dim3 block(256);
  if (akc && bkc) hipLaunchKernelGGL((gemm_bf3_kernel<true, true, NT, TS>), grid, block, 0, stream, g);
  else if (akc && !bkc) hipLaunchKernelGGL((gemm_bf3_kernel<true, false, NT, TS>), grid, block, 0, stream, g);
  else if (!akc && bkc) hipLaunchKernelGGL((gemm_bf3_kernel<false, true, NT, TS>), grid, block, 0, stream, g);
  else hipLaunchKernelGGL((gemm_bf3_kernel<false, false, NT, TS>), grid, block, 0, stream, g);
}

static int gemm_impl(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                     const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int relu,
                     int accumulate, int batch, int64_t sA, int64_t sB, int64_t sC, int split_k, int arith,
                     DropEpi drop, hipStream_t stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0) return ASR_E_ARG;
  if (drop.thresh && (batch != 1 || accumulate)) return ASR_E_ARG;
  const DropEpi no_drop = {0ull, 0u, 1.f};
  // the pass behind the product: the late epilogue of a split product and / or the dropout mask
  auto pass_behind = [&](bool late) {
    if (!late && !drop.thresh) return;
    dim3 eg((unsigned)((M * N + 255) / 256 > 2048 ? 2048 : (M * N + 255) / 256), 1, batch);
    hipLaunchKernelGGL(bias_act_kernel, eg, dim3(256), 0, stream, C, ldc, M, N, sC, late ? bias : nullptr, late ? relu : 0,
                       drop.thresh ? drop : no_drop);
  };
  const int ar = arith & ASR_ARITH_MASK;
  if (ar != ASR_ARITH_F32 && ar != ASR_ARITH_BF16X6 && ar != ASR_ARITH_BF16X3) return ASR_E_ARG;
  const bool auto_split = split_k <= 0;           // the kernel chooses; split_k == 1 is honoured as "unsplit" (run-to-run
                                                  // deterministic: no atomics), split_k > 1 as given on the 128 x 128 kernels
  GemmArgs g;
  const bool akc = !transA, bkc = transB != 0;
  g.A.p = A; g.A.ld = lda;
  if (akc) { g.A.R = M; g.A.Cn = K; } else { g.A.R = K; g.A.Cn = M; }
  g.A.vec = (lda % 4 == 0) && asr_aligned16(A) && (sA % 4 == 0);
  g.B.p = B; g.B.ld = ldb;
  if (bkc) { g.B.R = N; g.B.Cn = K; } else { g.B.R = K; g.B.Cn = N; }
  g.B.vec = (ldb % 4 == 0) && asr_aligned16(B) && (sB % 4 == 0);
  g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  g.accumulate = accumulate;
  g.sA = sA; g.sB = sB; g.sC = sC;
  g.queue = nullptr; g.xcd_mask = 0xffu;
  const bool epi = bias || relu;
  // Wide-tile LDS-DMA kernels (gemm_bf3w_kernel / gemm_bf6w_kernel) for conforming shapes: any M, N (edge tiles clamp their
  // DMA rows / columns and guard the stores); K % 32 == 0; row-contiguous operands need a multiple of 4 rows; per-lane
  // offsets are 32-bit.  Where they are used (tools/gemm_shapes.py, cold operands, one cfg-2 step): the weight-gradient
  // products (both operands row-contiguous, long K) and every K >= 2048 product; K = 512 projections time the same on both
  // kernels or worse on the wide one (one 8-wave workgroup per CU cannot hide its prologue and its 128 KB of output behind
  // another workgroup) and stay on the 128 x 128 kernel.  ASR_GEMM_TILE_WIDE sends every conforming shape there,
  // ASR_GEMM_TILE_NARROW none (tests, measurements).
  const bool base_shape = g.A.vec && g.B.vec && (akc ? M : K) * lda < ((int64_t)1 << 31) &&
                          (bkc ? N : K) * ldb < ((int64_t)1 << 31) && (akc || (M % 4 == 0 && M >= 4)) &&
                          (bkc || (N % 4 == 0 && N >= 4)) && M >= 64 && N >= 64;
  const bool wide_shape = K % WK == 0 && base_shape;
  const int64_t wtiles = ((M + WM - 1) / WM) * ((N + WN - 1) / WN) * batch;
  const bool may_split = auto_split && !(epi && accumulate);
  const bool wide_pays = (!akc && !bkc && K >= 1024) || (K >= 2048 && ((may_split && !epi) || wtiles >= 150));
  // Short contraction with k-contiguous operands (the layer-0 input projection, K = 80): the weights-stationary kernel
  if (ar != ASR_ARITH_F32 && akc && bkc && K == 80 && g.A.vec && g.B.vec && M * lda < ((int64_t)1 << 29) && N * ldb < ((int64_t)1 << 29) &&
      M >= 1024 && N >= 128 && (auto_split || split_k == 1) && !(arith & (ASR_GEMM_TILE_NARROW | ASR_GEMM_TILE_WIDE | ASR_GEMM_TILE_SP | ASR_GEMM_TILE_SMALL))) {
    const int tiles_n = (int)((N + 127) / 128);
    if (tiles_n <= 256) {
      const int groups = 8 * (tiles_n >= 32 ? 1 : 32 / tiles_n);
      g.bias = bias; g.relu = relu; g.split_k = 1;
      dim3 grid(tiles_n * groups, batch, 1), b4(512);
      const bool plain = M % 128 == 0 && N % 128 == 0 && !relu && !accumulate && M * ldc < ((int64_t)1 << 29);
      if (ar == ASR_ARITH_BF16X6) {
        if (plain) hipLaunchKernelGGL((gemm_bfk_kernel<3, 5, true>), grid, b4, 0, stream, g, groups);
        else hipLaunchKernelGGL((gemm_bfk_kernel<3, 5, false>), grid, b4, 0, stream, g, groups);
      } else {
        if (plain) hipLaunchKernelGGL((gemm_bfk_kernel<2, 5, true>), grid, b4, 0, stream, g, groups);
        else hipLaunchKernelGGL((gemm_bfk_kernel<2, 5, false>), grid, b4, 0, stream, g, groups);
      }
      pass_behind(false);
      ASR_CHECK_LAUNCH();
      return 0;
    }
  }
  // Products too small to fill the chip with 128 x 128 tiles take 64 x 64 tiles (gemm_bf3_kernel<..., 64>): at most
  // ASR_GEMM_SMALL_MAX workgroups of large tiles (the decoder-side projections and their gradients, the output layer); ASR_GEMM_TILE_SMALL
  // forces them (tests, measurements), ASR_GEMM_TILE_NARROW the 128 x 128 tile.
  static const int64_t small_max = [] { const char* f = getenv("ASR_GEMM_SMALL_MAX"); return f ? (int64_t)atoll(f) : (int64_t)256; }();   // measurement
  const int64_t tiles128 = ((M + BM - 1) / BM) * ((N + BN - 1) / BN) * batch;
  // ... i.e. when the large tiles, K split included, would be at most one workgroup per CU (tools/gemm_small_sweep.py:
  // [3 200 x 512] x K 512 28-31 us against 42-47, the output layer 31-37 against 46-48; a long K on 64 large tiles -
  // 512 workgroups after the split - stays on the large tiles: 76 against 83)
  const int64_t wgs128 = tiles128 * (auto_split ? ((epi && accumulate) ? 1 : narrow_split_k(M, N, K, batch, epi, 128)) : (split_k > 0 ? split_k : 1));
  const bool small = ar != ASR_ARITH_F32 && ((arith & ASR_GEMM_TILE_SMALL) || (wgs128 <= small_max && !(arith & ASR_GEMM_TILE_NARROW)));
  // The one-wave-per-SIMD kernel (gemm_bfs_kernel, same tile and K split policy): faster than both others on every shape
  // with enough work to fill the chip a few times (tools/gemm_shapes.py: 256 x 128 tiles x K tiles >= ~5 000; below
  // that its 4-wave workgroups cannot hide their prologue and the 128 x 128 kernel's two workgroups per CU win); its
  // buffer addressing wants byte offsets below 2^31.  ASR_GEMM_TILE_SP forces it for conforming shapes.
  const bool sp_shape = base_shape && (K % WK == 0 || (K % 4 == 0 && K > WK)) &&       // a K tail is masked (K = 80: the features)
                        (akc ? M : K) * lda < ((int64_t)1 << 29) && (bkc ? N : K) * ldb < ((int64_t)1 << 29);
  static const int64_t sp_min_units = [] { const char* f = getenv("ASR_GEMM_SP_MIN"); return f ? (int64_t)atoll(f) : (int64_t)5000; }();   // measurement
  // (K = 80: 223 us against the 128 x 128 kernel's 195 - three K tiles are all prologue; a LONG K with a masked tail - the
  // weight-gradient products over the packed rows of the encoder, K = sum of the utterances' extents - pays like any other)
  const bool sp_pays = (K % WK == 0 || K >= 64 * WK) && wtiles * (K / WK) >= sp_min_units;
  const bool use_sp = ar != ASR_ARITH_F32 && sp_shape && (auto_split || split_k == 1) && (!small || (arith & ASR_GEMM_TILE_SP)) &&
                      ((arith & ASR_GEMM_TILE_SP) || (sp_pays && !(arith & (ASR_GEMM_TILE_NARROW | ASR_GEMM_TILE_WIDE | ASR_GEMM_TILE_SMALL))));
  const bool wide = use_sp || (ar != ASR_ARITH_F32 && !(arith & (ASR_GEMM_TILE_NARROW | ASR_GEMM_TILE_SP | ASR_GEMM_TILE_SMALL)) && wide_shape &&
                               ((wide_pays && !small) || (arith & ASR_GEMM_TILE_WIDE)) && (auto_split || split_k == 1));
  if (wide) {
    // its own K split: 256 workgroup slots (one 8-wave workgroup per CU), cost in units of one stage = rounds x (stages
    // per slice + a fixed prologue / epilogue share) + what the atomics and the zero pass of a split cost per MB of output
    // (tools/gemm_wide_split.py: 12800 x 512 x 4096 takes 215 us unsplit on 200 of the 256 CUs, 256 us split in two)
    const int64_t tiles = wtiles, stages = (K + WK - 1) / WK;
    // A product with a bias / ReLU epilogue may be split as well: the atomics cannot carry the epilogue, so it becomes a
    // pass of its own over C (bias_act_kernel, as on the 128 x 128 kernels) - worth it where few tiles walk a long K
    // (the [T B / 4, 2048] x [2048, 512] projection of the top encoder layer: 52 tiles x 64 K tiles; tools/gemm_epi_split_sweep.py)
    int best = 1;
    if (may_split) {
      double best_cost = 1e30;
      for (int sk = 1; sk <= 16 && stages / sk >= 8; ++sk) {
        const int64_t wgs = tiles * sk, rounds = (wgs + 255) / 256;
        const double out_mb = (double)M * N * batch * 4.0 / 1048576.0;
        const double cost = (double)rounds * ((double)((stages + sk - 1) / sk) + 6.0) +
                            (sk > 1 ? 0.47 * sk * out_mb + (epi ? 3.0 + 0.6 * out_mb : 0.0) : 0.0);
        if (cost < best_cost) { best_cost = cost; best = sk; }
      }
    }
    static const int forced_sk = [] { const char* f = getenv("ASR_GEMM_WIDE_SK"); return f ? atoi(f) : 0; }();   // measurement
    if (forced_sk >= 1 && forced_sk <= stages && may_split) best = forced_sk;
    const bool late_epi = best > 1 && epi;
    g.bias = late_epi ? nullptr : bias; g.relu = late_epi ? 0 : relu;
    g.split_k = best;
    g.tiles_m = (int)((M + WM - 1) / WM);
    g.tiles_n = (int)((N + WN - 1) / WN);
    if (best > 1 && !accumulate && !(arith & ASR_GEMM_C_ZEROED)) {
      dim3 zg((unsigned)((M * N + 255) / 256 > 2048 ? 2048 : (M * N + 255) / 256), 1, batch);
      hipLaunchKernelGGL(zero_rows_kernel, zg, dim3(256), 0, stream, C, ldc, M, N, sC);
    }
    dim3 grid(g.tiles_m * g.tiles_n, batch * best, 1), block(512);
    if (use_sp) {
      const dim3 b4(256);
#define SP_LAUNCH(nt_, kt_)                                                                                             \
      do {                                                                                                              \
        if (akc && bkc) hipLaunchKernelGGL((gemm_bfs_kernel<true, true, nt_, kt_>), grid, b4, 0, stream, g);            \
        else if (akc && !bkc) hipLaunchKernelGGL((gemm_bfs_kernel<true, false, nt_, kt_>), grid, b4, 0, stream, g);     \
        else if (!akc && bkc) hipLaunchKernelGGL((gemm_bfs_kernel<false, true, nt_, kt_>), grid, b4, 0, stream, g);     \
        else hipLaunchKernelGGL((gemm_bfs_kernel<false, false, nt_, kt_>), grid, b4, 0, stream, g);                     \
      } while (0)
      if (ar == ASR_ARITH_BF16X6) { if (K % WK) SP_LAUNCH(3, true); else SP_LAUNCH(3, false); }
      else { if (K % WK) SP_LAUNCH(2, true); else SP_LAUNCH(2, false); }
#undef SP_LAUNCH
    } else if (ar == ASR_ARITH_BF16X6) {
      if (akc && bkc) hipLaunchKernelGGL((gemm_bf6w_kernel<true, true>), grid, block, 0, stream, g);
      else if (akc && !bkc) hipLaunchKernelGGL((gemm_bf6w_kernel<true, false>), grid, block, 0, stream, g);
      else if (!akc && bkc) hipLaunchKernelGGL((gemm_bf6w_kernel<false, true>), grid, block, 0, stream, g);
      else hipLaunchKernelGGL((gemm_bf6w_kernel<false, false>), grid, block, 0, stream, g);
    } else {
      if (akc && bkc) hipLaunchKernelGGL((gemm_bf3w_kernel<true, true>), grid, block, 0, stream, g);
      else if (akc && !bkc) hipLaunchKernelGGL((gemm_bf3w_kernel<true, false>), grid, block, 0, stream, g);
      else if (!akc && bkc) hipLaunchKernelGGL((gemm_bf3w_kernel<false, true>), grid, block, 0, stream, g);
      else hipLaunchKernelGGL((gemm_bf3w_kernel<false, false>), grid, block, 0, stream, g);
    }
    pass_behind(late_epi);
    ASR_CHECK_LAUNCH();
    return 0;
  }
  const int ts = small ? 64 : 128;
  if (auto_split) split_k = (epi && accumulate) ? 1 : narrow_split_k(M, N, K, batch, epi, ts);
  // split-K with an epilogue: the product is formed without it (atomics) and a second pass applies bias / ReLU
  const bool late_epilogue = split_k > 1 && epi;
  if (late_epilogue && accumulate) return ASR_E_SHAPE;
  g.bias = late_epilogue ? nullptr : bias;
  g.relu = late_epilogue ? 0 : relu;
  g.tiles_m = (int)((M + ts - 1) / ts);
  g.tiles_n = (int)((N + ts - 1) / ts);
  const int64_t ktiles = (K + BK - 1) / BK;
  if (split_k > ktiles) split_k = (int)ktiles;
  g.split_k = split_k;
  if (split_k > 1 && !accumulate && !(arith & ASR_GEMM_C_ZEROED)) {
    dim3 zg((unsigned)((M * N + 255) / 256 > 2048 ? 2048 : (M * N + 255) / 256), 1, batch);
    hipLaunchKernelGGL(zero_rows_kernel, zg, dim3(256), 0, stream, C, ldc, M, N, sC);
  }
  dim3 grid(g.tiles_m * g.tiles_n, batch * split_k, 1), block(256);
  if (ar == ASR_ARITH_BF16X6 && small) launch_narrow_split<3, 64>(akc, bkc, grid, stream, g);
  else if (ar == ASR_ARITH_BF16X3 && small) launch_narrow_split<2, 64>(akc, bkc, grid, stream, g);
  else if (ar == ASR_ARITH_BF16X6) launch_narrow_split<3>(akc, bkc, grid, stream, g);
  else if (ar == ASR_ARITH_BF16X3) launch_narrow_split<2>(akc, bkc, grid, stream, g);
  else if (akc && bkc) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, stream, g);
  else if (akc && !bkc) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, stream, g);
  else if (!akc && bkc) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, stream, g);
  else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, stream, g);
  pass_behind(late_epilogue);
  ASR_CHECK_LAUNCH();
  return 0;
}

extern "C" int asr_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                            const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int relu,
                            int accumulate, int batch, int64_t sA, int64_t sB, int64_t sC, int split_k, int arith,
                            asr_stream_t stream) {
  const DropEpi none = {0ull, 0u, 1.f};
  return gemm_impl(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, relu, accumulate, batch, sA, sB, sC, split_k, arith,
                   none, (hipStream_t)stream);
}

// C = dropout(act(A B + bias)): asr_gemm_f32 (batch 1, no accumulate) with the seeded mask of asr_dropout_seeded_f32 over the
// element index m N + n of C - in the bias / ReLU pass where the product was split over K, else in a pass of its own.
extern "C" int asr_gemm_drop_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                                 const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int relu,
                                 int split_k, int arith, uint64_t seed, float p, asr_stream_t stream) {
  if (p < 0.f || p >= 1.f) return ASR_E_ARG;
  const DropEpi d = {(unsigned long long)seed, asr_drop_thresh(p), 1.0f / (1.0f - p)};
  return gemm_impl(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, relu, 0, 1, 0, 0, 0, split_k, arith, d,
                   (hipStream_t)stream);
}

// C += op(A) op(B) on the XCDs of `xcd_mask` only, by workgroups that fit beside a persistent XCD-local kernel: the
// weight-gradient products of a small batch, issued on a side stream while the recurrence of the layer below runs on the
// other XCDs (gemm_bf3_kernel<..., 64, QUEUE>).  The partial products of all K slices are ADDED to C with atomics:
// C holds zeros (a plain product) or what the product accumulates onto.  `queue`: one zeroed 32-bit word of the caller's
// (the ticket counter; consumed).  Two launches: the masked one, oversubscribed by 8 / popcount(mask) (a workgroup that
// lands on an excluded XCD leaves at once and draws no ticket), and an unmasked sweep behind it that draws whatever tickets
// are left - none, when workgroups are dealt round robin over the XCDs; the result does not depend on that.
extern "C" int asr_gemm_side_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                                 const float* B, int64_t ldb, float* C, int64_t ldc, int batch, int64_t sA, int64_t sB,
                                 int64_t sC, int arith, unsigned xcd_mask, unsigned* queue, asr_stream_t stream_) {
  if (!A || !B || !C || !queue || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || !(xcd_mask & 0xffu)) return ASR_E_ARG;
  const int ar = arith & ASR_ARITH_MASK;
  if (ar != ASR_ARITH_BF16X6 && ar != ASR_ARITH_BF16X3) return ASR_E_SHAPE;      // (the fp32-input MFMA kernel has no such form)
  hipStream_t stream = (hipStream_t)stream_;
  GemmArgs g;
  const bool akc = !transA, bkc = transB != 0;
  g.A.p = A; g.A.ld = lda;
  if (akc) { g.A.R = M; g.A.Cn = K; } else { g.A.R = K; g.A.Cn = M; }
  g.A.vec = (lda % 4 == 0) && asr_aligned16(A) && (sA % 4 == 0);
  g.B.p = B; g.B.ld = ldb;
  if (bkc) { g.B.R = N; g.B.Cn = K; } else { g.B.R = K; g.B.Cn = N; }
  g.B.vec = (ldb % 4 == 0) && asr_aligned16(B) && (sB % 4 == 0);
  g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  g.bias = nullptr; g.relu = 0; g.accumulate = 1;
  g.sA = sA; g.sB = sB; g.sC = sC;
  g.tiles_m = (int)((M + 63) / 64);
  g.tiles_n = (int)((N + 63) / 64);
  g.queue = queue; g.xcd_mask = xcd_mask & 0xffu;
  const int nx = __builtin_popcount(g.xcd_mask);
  // K slices: about four tickets per CU of the allowed XCDs (a workgroup is 1 of up to 4 on its CU), at least 8 K tiles each
  const int64_t tiles = (int64_t)g.tiles_m * g.tiles_n * batch, ktiles = (K + BK - 1) / BK;
  int64_t sk = (4 * 32 * nx + tiles - 1) / tiles;
  if (sk > ktiles / 8) sk = ktiles / 8;
  if (sk < 1) sk = 1;
  if (sk > 64) sk = 64;
  g.split_k = (int)sk;
  const int64_t tickets = (int64_t)g.tiles_m * g.tiles_n * batch * sk;
  if (tickets > 0x3fffffff) return ASR_E_SHAPE;
  const unsigned gy = (unsigned)(batch * sk);
  const int64_t per_y = (int64_t)g.tiles_m * g.tiles_n;                 // tickets per unit of grid.y
  const unsigned gx = (unsigned)((per_y * 8 + nx - 1) / nx + 8);      // x extent oversubscribed: 1 in 8 / nx workgroups survives
  auto launch = [&](dim3 grid) {
    const dim3 block(256);
#define SIDE_LAUNCH(nt_)                                                                                                     \
    do {                                                                                                                     \
      if (akc && bkc) hipLaunchKernelGGL((gemm_bf3_kernel<true, true, nt_, 64, true>), grid, block, 0, stream, g);           \
      else if (akc && !bkc) hipLaunchKernelGGL((gemm_bf3_kernel<true, false, nt_, 64, true>), grid, block, 0, stream, g);    \
      else if (!akc && bkc) hipLaunchKernelGGL((gemm_bf3_kernel<false, true, nt_, 64, true>), grid, block, 0, stream, g);    \
      else hipLaunchKernelGGL((gemm_bf3_kernel<false, false, nt_, 64, true>), grid, block, 0, stream, g);                    \
    } while (0)
    if (ar == ASR_ARITH_BF16X6) SIDE_LAUNCH(3); else SIDE_LAUNCH(2);
#undef SIDE_LAUNCH
  };
  launch(dim3(gx, gy, 1));
  // the sweep: every XCD, one workgroup per ticket that could be left at most - all of them leave at once in the expected case
  g.xcd_mask = 0xffu;
  launch(dim3((unsigned)per_y, gy, 1));
  ASR_CHECK_LAUNCH();
  return 0;
}

int asr_skinny_launch(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* Bt, int64_t ldb,
                      float* C, int64_t ldc, const float* bias, int accumulate, const float* mask, int64_t ldmask,
                      int64_t mask_from, hipStream_t stream) {
  if (!A || !Bt || !C || M <= 0 || N <= 0 || K <= 0) return ASR_E_ARG;
  if (K % 16 || lda % 4 || ldb % 4) return ASR_E_SHAPE;
  if (!asr_aligned16(A) || !asr_aligned16(Bt)) return ASR_E_ALIGN;
  const unsigned nb = (unsigned)((N + 15) / 16);
  // long-K accumulate products (the decoder's dX = dgates Wcat, K = 4D) are split over K: the consumer is a later
  // launch, so partial sums can simply be added atomically into C
  int ksplit = 1;
  if (accumulate && !bias && K >= 1024 && K % 64 == 0) {
    ksplit = (int)(K / 512);
    while ((K / 16) % ksplit) --ksplit;
  }
  if (M <= 32) {   // 16-row workgroups (MT = 1) also for 17..32 rows: more workgroups, fewer bytes each
    hipLaunchKernelGGL((skinny_kernel<1>), dim3(nb, (unsigned)((M + 15) / 16), ksplit), dim3(256), 0, stream, M, N,
                       (int)K, A, lda, Bt, ldb, C, ldc, bias, accumulate, mask, ldmask, mask_from, ksplit);
  } else {
    hipLaunchKernelGGL((skinny_kernel<2>), dim3(nb, (unsigned)((M + 31) / 32), ksplit), dim3(256), 0, stream, M, N,
                       (int)K, A, lda, Bt, ldb, C, ldc, bias, accumulate, mask, ldmask, mask_from, ksplit);
  }
  ASR_CHECK_LAUNCH();
  return 0;
}

extern "C" int asr_gemm_skinny_f32(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* Bt,
                                   int64_t ldb, float* C, int64_t ldc, const float* bias, int accumulate,
                                   const float* mask, int64_t ldmask, int64_t mask_from, asr_stream_t stream) {
  return asr_skinny_launch(M, N, K, A, lda, Bt, ldb, C, ldc, bias, accumulate, mask, ldmask, mask_from,
                           (hipStream_t)stream);
}

extern "C" int asr_colsum_f32(int64_t M, int64_t N, const float* X, int64_t ldx, float* out, int accumulate,
                              asr_stream_t stream) {
  if (!X || !out || M <= 0 || N <= 0) return ASR_E_ARG;
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(out, 0, (size_t)N * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  if (N % 4 == 0 && ldx % 4 == 0 && asr_aligned16(X))
    hipLaunchKernelGGL(colsum4_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((M + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, M, N, X, ldx, out);
  else
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((M + 511) / 512)), dim3(256), 0,
                       (hipStream_t)stream, M, N, X, ldx, out);
  ASR_CHECK_LAUNCH();
  return 0;
}
