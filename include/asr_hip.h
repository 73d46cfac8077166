/*
 * asr_hip.h — C ABI of libasr_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for
 * the seq2seq-ASR training hot path of jjery2243542/semi-supervised-ASR.
 *
 * The reference has no FFI: the path sits behind torch modules (model.py) that dispatch
 * stock operators.  Each entry point below names the reference operator call it replaces
 * (file:line into the reference) — that is the "interface" a maintainer would bind.
 *
 * Conventions (all entry points):
 *   - plain device pointers + explicit sizes; contiguous row-major fp32; int32 lengths;
 *     base pointers 16-byte aligned;
 *   - the CALLER allocates every buffer (outputs and workspaces); nothing is allocated,
 *     freed or synchronised inside; work is enqueued on `stream` (a hipStream_t);
 *   - return 0 on success, a negative ASR_E_* code for a bad argument, or a positive
 *     hipError_t if a launch failed.  Nothing throws across the ABI;
 *   - re-entrant; no global mutable state.
 *
 * Layout vocabulary:
 *   time-major      activations are [T][B][...] so one time step is one contiguous slab;
 *   gate-interleave the 4H gate axis is ordered unit-major: index = unit*4 + gate, gate in
 *                   (i,f,g,o) (torch order).  Host code permutes W_ih/W_hh/bias
 *                   rows once per step; a 4-unit slice of all four gates is 16 contiguous
 *                   floats (64 B).
 */
#ifndef ASR_HIP_H
#define ASR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASR_ABI_VERSION 5

#define ASR_E_ARG    (-1)  /* null pointer / non-positive size */
#define ASR_E_SHAPE  (-2)  /* size not supported by the kernel (see each function) */
#define ASR_E_ALIGN  (-3)  /* pointer or leading dimension not 16-byte aligned */

typedef void* asr_stream_t; /* hipStream_t */

int asr_abi_version(void);

/* Product arithmetic of the MFMA kernels: an explicit argument (`arith`) of asr_gemm_f32 and of the persistent LSTM
 * recurrences - there is no process-wide switch.  All operands, accumulators and results are fp32 in every mode; what
 * the mode selects is how a product x * y of two fp32 operands is formed:
 *   ASR_ARITH_F32     on the fp32-input MFMA (v_mfma_f32_32x32x2_f32 / 4x4x1): the exact fp32 product, 157 TF peak.
 *                     (In the persistent recurrences the RECURRENT operand - h_{t-1}, the exchanged partial sums - crosses
 *                     CUs as fp32 words whose mantissa LSB carries the hand-off's validity tag (csrc/persist.h): the
 *                     product is exact, that operand has 23 mantissa bits.  Worst cfg-2 / cfg-5 gradient element under
 *                     this mode: 1.7e-4 / 4.6e-5 of its tensor's scale.  asr_dec_seq_bwd_persist* take no arith argument: the
 *                     decoder kernels are exact fp32 throughout except the embedding part of dX, dgates W_cat[:, D+O:],
 *                     which they always form with bf16x6 products.)
 *   ASR_ARITH_BF16X6  fp32-equivalent on the bf16 MFMA: each operand is re-encoded LOSSLESSLY as three bf16 terms
 *                     (x = a + b + c exactly: 3 x 8 significand bits, every split rounded to nearest) and the six products
 *                     aa' + ab' + ba' + ac' + ca' + bb' are accumulated in fp32.  Dropped: bc' + cb' + cc' <= 2^-24 |x y|,
 *                     below the rounding of the fp32 product itself.  2.7x the fp32 pipe's MAC rate.  The host code's
 *                     default (hip_backend.ARITH).
 *   ASR_ARITH_BF16X3  two terms (16 significand bits), three products: <= 2^-15 relative per product.  Fastest, and NOT a
 *                     precision the reference has: OUTSIDE the 1e-3 parity gate on long recurrences (2.2e-3 on a sampled
 *                     layer-0 dW_ih element at cfg-5, 6.8e-4 at cfg-2; tests/test_big_configs_gpu.py holds it to 5e-3) -
 *                     never the default.
 * Flags OR-ed into `arith` select a kernel where several implement the same arithmetic (tests, measurements):
 *   ASR_GEMM_TILE_NARROW / ASR_GEMM_TILE_WIDE / ASR_GEMM_TILE_SP   asr_gemm_f32: only the 128 x 128 kernel / the 256 x 128
 *                     LDS-DMA kernel / the 256 x 128 one-wave-per-SIMD kernel for every conforming shape; by default
 *                     each is used for the shapes it pays on (csrc/gemm.hip: asr_gemm_f32);
 *   ASR_GEMM_TILE_SMALL   asr_gemm_f32: 64 x 64 tiles on the 128 x 128 kernel's code (by default for products whose large tiles,
 *                     K split included, would be at most one workgroup per CU: decoder-side projections, output layer);
 *   ASR_GEMM_C_ZEROED     asr_gemm_f32 / asr_gemm_drop_f32: a promise, not a selector - C holds zeros already (a slice of a
 *                     buffer the caller zeroes once per step), so a product the library splits over K needs no zero pass
 *                     of its own in front of the atomics.  Ignored with accumulate != 0.
 *   ASR_LSTM_BWD_GATHER   asr_lstm_seq_bwd_persist: the gathered-dG kernel instead of the one with exchanged partials.
 *   ASR_DEBUG_FAULT       asr_lstm_seq_fwd_persist (ASR_ARITH_BF16X6, H = 512 only; ASR_E_SHAPE otherwise): TESTS ONLY - the
 *                         FAULT instantiation of the kernel: slice 1 of group 0 stops publishing after its first step and
 *                         every wait gives up after 4 096 attempts, so the launch runs the kernels' own abort path (bounded
 *                         spin expires -> abort word + latch + code -> NaN poison -> every workgroup drains) in a
 *                         millisecond.  asr_dec_seq_fwd_persist_fault is the decoder's. */
#define ASR_ARITH_F32        0
#define ASR_ARITH_BF16X6     1
#define ASR_ARITH_BF16X3     2
#define ASR_ARITH_MASK       0xff
#define ASR_GEMM_TILE_NARROW 0x100
#define ASR_GEMM_TILE_WIDE   0x200
#define ASR_GEMM_TILE_SP     0x800
#define ASR_GEMM_TILE_SMALL  0x1000
#define ASR_GEMM_C_ZEROED    0x2000
#define ASR_LSTM_BWD_GATHER  0x400
#define ASR_DEBUG_FAULT      0x10000

/* ---------------------------------------------------------------------------------------
 * Graph memo for the per-time-step launch chains (asr_lstm_seq_*, asr_dec_seq_*).  The `graphs`
 * argument of those calls may be NULL (eager launches) or a handle from asr_graphs_create(): the
 * chain is then stream-captured the second time the same argument tuple is seen and replayed
 * with one hipGraphLaunch afterwards (kernel arguments are baked, so a different buffer address
 * is a different key and simply runs eagerly).  One handle per launching host thread; the handle
 * is the only state and is owned by the caller.
 * ------------------------------------------------------------------------------------- */
void* asr_graphs_create(int max_entries);
void asr_graphs_destroy(void* graphs);
int asr_graphs_stats(void* graphs, int64_t* hits, int64_t* captures, int64_t* eager);

/* ---------------------------------------------------------------------------------------
 * Dense fp32 GEMM on the MFMA (product arithmetic: `arith`, see above).
 *   C[M,N] (ldc) = op(A)[M,K] * op(B)[K,N]  (+ bias[N]) (relu) (+ C if accumulate)
 * Row-major.  transA=0: A is [M][K] (lda>=K); transA=1: A is [K][M] (lda>=M).
 *             transB=0: B is [K][N] (ldb>=N); transB=1: B is [N][K] (ldb>=K).
 * batch>1 runs `batch` independent GEMMs with element strides sA,sB,sC.
 * split_k <= 0: the library chooses a K split; split_k == 1: unsplit (no atomics: run-to-run deterministic);
 * split_k > 1 splits K over grid.z and accumulates with fp32 atomics (C is zero-filled on
 * the stream first unless accumulate!=0); with a K split, bias/relu are applied by a second
 * pass over C (not combinable with accumulate).
 * Replaces torch.nn.Linear / mm / bmm on the path: the LSTM input-gate product inside
 * torch.nn.LSTM (model.py:67-68,80), project_layer (model.py:93-94), mlp_enc
 * (model.py:144), output_layer (model.py:293) and every autograd mm behind them.
 * ------------------------------------------------------------------------------------- */
int asr_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K,
                 const float* A, int64_t lda, const float* B, int64_t ldb,
                 float* C, int64_t ldc, const float* bias, int relu, int accumulate,
                 int batch, int64_t sA, int64_t sB, int64_t sC, int split_k, int arith,
                 asr_stream_t stream);
/* C = dropout(act(op(A) op(B) + bias)): asr_gemm_f32 (batch 1, no accumulate) followed by the seeded mask of
 * asr_dropout_seeded_f32 over the element index m N + n of C (project_layer -> relu -> dropout, model.py:93-95): in the
 * bias / ReLU pass where the product was split over K, in a pass of its own otherwise. */
int asr_gemm_drop_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                      const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int relu, int split_k,
                      int arith, uint64_t seed, float p, asr_stream_t stream);

/* C[b] += op(A[b]) op(B[b]) on the XCDs of `xcd_mask` only (bit x = XCC id x), by workgroups built to run BESIDE the
 * persistent XCD-local kernels below: a batch of <= 8 utterances keeps the LSTM / decoder recurrences on four of the eight
 * XCDs (group g of a persistent launch = XCC id g; groups without rows leave at once), and the weight-gradient products
 * (the autograd mm's of torch.nn.LSTM / Linear behind model.py:67-68,93-94,262-263 - off the backward's critical path) run
 * on the other four, issued on a side stream under the recurrence of the layer below.  64 x 64 tiles, <= 128 VGPRs and 30 KB
 * of LDS (a workgroup fits next to a persistent one on a CU, so no persistent launch waits for room), one (tile, K slice)
 * per workgroup drawn from the ticket counter `queue` (ONE zeroed 32-bit word of the caller's, consumed by the call).  The
 * K slices are ADDED to C with atomics: C holds zeros for a plain product.  Arguments as asr_gemm_f32 (strides in elements,
 * negative batch strides allowed); arith: ASR_ARITH_BF16X6 / _BF16X3 (ASR_E_SHAPE for ASR_ARITH_F32: the caller runs
 * asr_gemm_f32 instead).  The result does not depend on where the hardware places workgroups (a second, unmasked launch
 * draws whatever tickets the masked one left). */
int asr_gemm_side_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                      const float* B, int64_t ldb, float* C, int64_t ldc, int batch, int64_t sA, int64_t sB, int64_t sC,
                      int arith, unsigned xcd_mask, unsigned* queue, asr_stream_t stream);

/* Skinny GEMM for the sequential chains (M = batch rows, tens not thousands):
 *   C[M,N] (ldc) (+)= A[M,K] (lda) * Bt[N,K]^T (ldb)  (+ bias[N]) (* mask[M,N] from col mask_from)
 * 16 output columns per workgroup, K split over the 4 waves, 16x16x4 f32 MFMA.
 * K % 16 == 0, lda/ldb % 4 == 0.  Replaces mlp_dec (model.py:163), output_layer per step
 * (model.py:293) and the dX products of LSTMCell/mlp_dec backward. */
int asr_gemm_skinny_f32(int64_t M, int64_t N, int64_t K,
                        const float* A, int64_t lda, const float* Bt, int64_t ldb,
                        float* C, int64_t ldc, const float* bias, int accumulate,
                        const float* mask, int64_t ldmask, int64_t mask_from,
                        asr_stream_t stream);

/* Column sums: out[N] (+)= sum_m X[m][n]   (bias gradients). */
int asr_colsum_f32(int64_t M, int64_t N, const float* X, int64_t ldx, float* out,
                   int accumulate, asr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Fused LSTM sequence, all time steps of one layer, both directions in one call.
 * Replaces torch.nn.LSTM(bidirectional) on a PackedSequence + pad_packed_sequence
 * (model.py:79-81) and the 2-layer judge LSTM (model.py:466-467,515-519) — restated as a
 * masked recurrence on the padded time-major tensor: at t >= lens[b] the
 * state and the output are 0.
 *
 *   gates [T][B][ndir][4H]  in : x_t W_ih^T + b_ih + b_hh, gate-interleaved
 *                           out: activated gates (i,f,g,o) — saved for backward
 *   w_hh  [ndir][4H][H]     rows gate-interleaved
 *   lens  [B] int32 (device)
 *   y     [T][B][ndir*H]    hidden states; direction d occupies columns [d*H,(d+1)*H)
 *   c     [T][B][ndir*H]    cell states (saved for backward)
 * B is the batch STRIDE of the buffers, nb <= B the number of rows this call processes: utterances are
 * independent, so a caller may run disjoint row groups (pointers pre-offset to the group's first row)
 * concurrently on different streams to overlap the latency-bound chains.
 * Direction 0 runs t = 0..T-1, direction 1 (if ndir==2) runs t = T-1..0.
 * One kernel launch per time step covers both directions: workgroup = (4 hidden units x
 * 4 gates = 16 gate rows) x (<=32 batch rows); h_{t-1} W_hh^T on the 16x16x4 f32 MFMA
 * with K split over the 4 waves, partials reduced through LDS, then sigmoid/tanh/state
 * update.  H % 16 == 0.
 *
 * PACKED ROWS (ABI 4; rowbase / rowext, both NULL = the time-major layout above).  pack_padded_sequence spares the
 * reference's LSTM the padded frames (model.py:79-81); here the whole encoder runs without them: batch row b owns the
 * rows rowbase[b] .. rowbase[b] + rowext[b] - 1 of gates / y / c / dy (then [R][ndir][4H] and [R][ndir*H], R = the sum
 * of the extents), time t at row rowbase[b] + t.  lens[b] < rowext[b]: the rows lens[b] .. rowext[b] - 1 are padding
 * INSIDE the block - the kernels write y = c = 0 (forward) and dG = 0 (backward) there, as they do for the padded times
 * of the time-major layout, also behind the T steps of the call (T >= max lens suffices) - and times >= rowext[b] do not
 * exist (nothing is read into a result or written).  With at
 * least one padding row behind every utterance the products that pair a row with its time neighbour (dW_hh = sum_t
 * dG_t^T h_{t-1}) stay ONE row-shifted GEMM over all R rows: the neighbour across a block boundary is a zero row.
 * Blocks whose extents halve from layer to layer (rowext_l = 2 rowext_{l+1}) make the pyramid's pair-concat the
 * B = 1 case of asr_pyramid_concat_* over the R rows.  rowbase / rowext: int32 [B] on the device.
 * ------------------------------------------------------------------------------------- */
int asr_lstm_seq_fwd(int T, int B, int nb, int H, int ndir, float* gates, const float* w_hh,
                     const int32_t* lens, const int32_t* rowbase, const int32_t* rowext, float* y, float* c,
                     void* graphs, asr_stream_t stream);

/* Persistent fast path of asr_lstm_seq_fwd (same arguments and results; csrc/lstm_persist.hip): ONE launch runs
 * all T steps, each XCD owns a (direction, 8- or 4-row) group, W_hh stays in registers, h_t is exchanged inside the
 * XCD as LSB-tagged fp32 words.  Applies when H is 128, 256, 320, 512 (or 640 with a bf16 arithmetic) on an 8 x 32-CU
 * device, any nb (row blocks of 32 * (8 / ndir) run as consecutive launches; at H = 512 under the bf16 arithmetics every
 * block of 64 * (8 / ndir) rows runs 16 rows per group in one launch); otherwise returns ASR_E_SHAPE and the
 * caller uses asr_lstm_seq_fwd.  `arith`: product arithmetic of h W_hh^T (ASR_ARITH_*, see above).
 * xch and ctrl are caller-allocated scratch shared by ALL persistent entry points (LSTM and decoder): at least
 * asr_persist_scratch_bytes() says (10 MB of exchange - the H = 640 backward's partial sums are the largest user - and a
 * 128-byte control block); smaller buffers are silently overrun by the pre-launch zero fill and the kernels' atomics.
 * ctrl = [16 latch words | 16 per-launch words]: the
 * per-launch words and the used part of xch are zeroed on the stream before every launch (one fill when ctrl sits exactly
 * 128 bytes in front of xch, else two).  A kernel that aborts (bounded spin expired / unexpected placement) poisons its
 * outputs with NaN and sets per-launch word 8 (code in word 9) AND latch word 0 (code in latch word 1).  The library
 * never clears the latch words: a sequence operator is several launches, and the caller looks once, after the last
 * one, and clears the latch itself. */
int asr_persist_scratch_bytes(int64_t* xch_bytes, int64_t* ctrl_bytes);   /* minimum sizes of the scratch pair; returns 0 */
/* rowbase / rowext: packed rows (see asr_lstm_seq_fwd), NULL = time-major.  With packed rows T is the number of STEPS to
 * run, at least the longest of the nb rows (it may be less than their extents: the kernels zero a block's padding rows
 * behind the last step themselves).  lens_host: optional HOST copy of lens (packed rows only): every row block then runs
 * max(lens of its rows) steps instead of T (a batch of 256 length-sorted utterances on one GPU: the later blocks are
 * shorter). */
int asr_lstm_seq_fwd_persist(int T, int B, int nb, int H, int ndir, float* gates, const float* w_hh,
                             const int32_t* lens, const int32_t* rowbase, const int32_t* rowext,
                             const int32_t* lens_host, float* y, float* c, void* xch, void* ctrl, int arith,
                             asr_stream_t stream);

/* Backward through the same recurrence.
 *   gates [T][B][ndir][4H]  in : activated gates from the forward; out: dL/d(pre-activation)
 *                           (= gradient of the x-projection, gate-interleaved)
 *   w_hhT [ndir][H][4H]     transpose of the gate-interleaved w_hh
 *   dy    [T][B][ndir*H]    upstream gradient of y
 *   c     forward cell states;  dcarry [B][ndir*H] zero-initialised scratch (dL/dc carry)
 * dW_hh is NOT produced here: the caller forms sum_t dG_t^T h_{t-1} with one asr_gemm_f32
 * (transA=1) over the whole sequence after this call. */
int asr_lstm_seq_bwd(int T, int B, int nb, int H, int ndir, float* gates, const float* w_hhT,
                     const int32_t* lens, const int32_t* rowbase, const int32_t* rowext, const float* dy, const float* c,
                     float* dcarry, void* graphs, asr_stream_t stream);
/* Persistent fast path of asr_lstm_seq_bwd (same conditions / scratch / abort convention as asr_lstm_seq_fwd_persist;
 * H in {128, 256, 320, 512}, and 640 under ASR_ARITH_BF16X6).  With a bf16 arithmetic and H in {128, 256, 512} - and
 * H = 320 (10 units per CU in 12 slots) / H = 640 (20 units per CU, its own kernel) under ASR_ARITH_BF16X6 - the CUs of a
 * group exchange partial sums of dh_rec, laid out [8 groups][2][32 dest][32 src]
 * [8 rows][slots per CU] floats in xch (8 MB at H = 512); otherwise (H = 320 with two terms, ASR_ARITH_F32,
 * ASR_LSTM_BWD_GATHER) every CU gathers the step's dG tile.  Exchanged words carry a 1-bit tag in the
 * mantissa LSB; the in-place dG is what the pointwise update produced.  `arith` selects the product arithmetic of
 * dG W_hh and of the fused dW_hh (the gathered-dG kernels always form dW_hh on the fp32 MFMA).
 * If y (forward hidden states) and dw_hh ([ndir][4H][H], gate-interleaved, zero-filled or holding a running sum)
 * are given, the recurrent weight gradient sum_t dG_t^T h_{t-1} is accumulated into dw_hh inside the kernel
 * (fp32 atomics across the row groups) and the caller skips that GEMM.  If db ([ndir][4H], gate-interleaved,
 * zero-filled) is given, the bias gradient sum_{t,b} dG is accumulated into it as well.
 * With packed rows (rowbase / rowext / lens_host as in asr_lstm_seq_fwd_persist) y / dw_hh are ignored: dW_hh is the
 * caller's row-shifted product over all R rows. */
int asr_lstm_seq_bwd_persist(int T, int B, int nb, int H, int ndir, float* gates, const float* w_hhT,
                             const int32_t* lens, const int32_t* rowbase, const int32_t* rowext,
                             const int32_t* lens_host, const float* dy, const float* c, const float* y,
                             float* dw_hh, float* db, void* xch, void* ctrl, int arith, asr_stream_t stream);
/* Does asr_lstm_seq_bwd_persist(_w) with this (H, arith) accumulate dW_hh itself when given y and dw_hh?  1 yes; 0 no -
 * the ASR_ARITH_BF16X6 exchanged-partials kernels (H in {128, 256, 320, 512, 640}) leave dW_hh = sum_t dG_t^T h_{t-1} to the caller
 * (one batched asr_gemm_f32 over the two directions: with six products per product the fused form costs more time on the
 * kernel's serial chain than the GEMM does) and ignores y / dw_hh; -1 no persistent backward for this H / arith.  The
 * bias gradient db is accumulated by every persistent backward kernel. */
int asr_lstm_bwd_persist_fuses_dw(int H, int arith);
/* asr_lstm_seq_bwd_persist with W_hh in the FORWARD layout (w_hh_il [ndir][4H][H], gate-interleaved: the array
 * asr_lstm_seq_fwd_persist consumed) instead of its transpose: the exchanged-partials kernel reads its slice once per
 * launch.  Returns ASR_E_SHAPE where that kernel does not apply; the caller then forms w_hhT and calls
 * asr_lstm_seq_bwd_persist / asr_lstm_seq_bwd. */
int asr_lstm_seq_bwd_persist_w(int T, int B, int nb, int H, int ndir, float* gates, const float* w_hh_il,
                               const int32_t* lens, const int32_t* rowbase, const int32_t* rowext,
                               const int32_t* lens_host, const float* dy, const float* c, const float* y, float* dw_hh,
                               float* db, void* xch, void* ctrl, int arith, asr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * The two ends of the packed-row encoder (csrc/rows.hip).  C % 4 == 0; lens / rowbase / rowext int32 [B] on the device;
 * ext_max = max_b rowext[b] (host).
 *   asr_rows_pack_f32        x [B][T][C], the collated batch (dataloader.py:6-12) -> rows [R][C]: row rowbase[b] + t =
 *                            x[b][t] for t < lens[b], zeros for lens[b] <= t < rowext[b]
 *   asr_rows_unpack_fwd_f32  rows [R][C] -> out [B][T][C] (the encoder output the decoder reads, model.py:109-112):
 *                            out[b][t] = rows[rowbase[b] + t] for t < lens[b]; the frames behind an utterance hold what the
 *                            reference's last projection makes of an all-zero frame, dropout(relu(bias)) (model.py:93-95,
 *                            SURVEY F2): fill [C] (NULL = zeros) times the dropout mask - `mask` [B][T][C] given, or
 *                            regenerated from (seed, p) over the element index of out (asr_dropout_seeded_f32; p = 0: none).
 *                            fill_relu != 0: `fill` is the projection's bias itself and the kernel takes relu(fill)
 *   asr_rows_unpack_bwd_f32  drows[rowbase[b] + t] = dout[b][t] for t < lens[b], zeros on the block's padding rows;
 *                            dfill [C] (NULL, or zero-filled by the caller) += sum of dout * mask over the padded frames
 *                            (relu_of != NULL: only where relu_of[c] > 0 - the gradient of the bias behind that relu)
 * ------------------------------------------------------------------------------------- */
int asr_rows_pack_f32(int B, int T, int C, const float* x, const int32_t* lens, const int32_t* rowbase,
                      const int32_t* rowext, int ext_max, float* rows, asr_stream_t stream);
int asr_rows_unpack_fwd_f32(int B, int T, int C, const float* rows, const int32_t* lens, const int32_t* rowbase,
                            const float* fill, int fill_relu, const float* mask, uint64_t seed, float p, float* out,
                            asr_stream_t stream);
int asr_rows_unpack_bwd_f32(int B, int T, int C, const float* dout, const int32_t* lens, const int32_t* rowbase,
                            const int32_t* rowext, int ext_max, const float* mask, uint64_t seed, float p, float* drows,
                            float* dfill, const float* relu_of, asr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Pyramidal pair-concat (model.py:85-92), time-major:
 *   in [T][B][C] -> out [ceil(T/2)][B][2C], out[t'] = [in[2t'] | in[2t'+1]]; for odd T the
 *   missing last frame replicates in[T-1].  Optional elementwise mask (dropout, already
 *   scaled by 1/(1-p)) of the input shape is applied on the fly.  float4 coalesced.
 * Backward: din[t] = dout[t/2][.., (t%2)*C ..] (* mask), the replicated frame's gradient
 * folded into din[T-1].
 * ------------------------------------------------------------------------------------- */
int asr_pyramid_concat_fwd(int T, int B, int C, const float* in, const float* mask,
                           float* out, asr_stream_t stream);
int asr_pyramid_concat_bwd(int T, int B, int C, const float* dout, const float* mask,
                           float* din, asr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Counter-based dropout (the shared torch.nn.Dropout of model.py:73,82,95): inverted dropout whose mask is a pure
 * function of (seed, flat element index), mask(i) = [hash(seed, i) >= p * 2^32] / (1 - p), so nothing is stored
 * between forward and backward.  p in [0, 1); n % 4 == 0; pointers 16-byte aligned.
 *   asr_dropout_seeded_f32      x[i] *= mask(i), in place (after relu(x W^T + b), model.py:94-95)
 *   asr_relu_dropout_bwd_f32    out[i] = grad[i] * mask(i) * (y[i] > 0), y = the dropped-out relu output
 *   asr_dropout_mask_f32        mask[i] = mask(i) (tests; the decoder's [L][B][O+E] operand mask)
 *   asr_pyramid_concat_*_seeded the pair-concat kernels with mask(i) over the [T][B][C] input regenerated in flight
 * ------------------------------------------------------------------------------------- */
int asr_dropout_seeded_f32(int64_t n, float* x, uint64_t seed, float p, asr_stream_t stream);
int asr_relu_dropout_bwd_f32(int64_t n, const float* grad, const float* y, uint64_t seed, float p, float* out,
                             asr_stream_t stream);
int asr_dropout_mask_f32(int64_t n, float* mask, uint64_t seed, float p, asr_stream_t stream);
int asr_pyramid_concat_fwd_seeded(int T, int B, int C, const float* in, uint64_t seed, float p, float* out,
                                  asr_stream_t stream);
int asr_pyramid_concat_bwd_seeded(int T, int B, int C, const float* dout, uint64_t seed, float p, float* din,
                                  asr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Decoder step = LSTMCell + location-aware attention (Decoder.forward_step model.py:283-294,
 * AttLoc.forward model.py:139-173).  Per-sequence constants:
 *   P   [B][Tp][A]   mlp_enc(enc_h)                     (model.py:144)
 *   Q   [B][Tp][O]   enc_h W_o^T  — mlp_o hoisted out of the step loop by linearity:
 *                    mlp_o(sum_t w_t h_t) = sum_t w_t (W_o h_t) + b_o   (model.py:171-172)
 *   X   [L+1][B][KX] step inputs, KX = D + O + E: X[s] = [ z_{s-1} | ctx_{s-1} | emb_s ]
 *                    (cell_inp of model.py:284 plus the recurrent z); the step writes z_s
 *                    and ctx_s into X[s+1].
 *   wcat [4D][KX]    [W_hh | W_ih(ctx cols) | W_ih(emb cols)], rows gate-interleaved
 *   bcat [4D]        b_ih + b_hh, gate-interleaved
 * asr_dec_step_fwd(s) enqueues: fused cell (skinny MFMA gate GEMM + activations), mlp_dec
 * skinny GEMM, attention score kernel (201-tap location conv, tanh energy), softmax +
 * context kernel.  Softmax runs over ALL Tp frames with temperature `scaling` (SURVEY
 * F1/F4).  w_prev for s==0 is the caller-provided uniform-over-valid-frames row block.
 * Saved for backward: gates[s], cstate[s], S[s] (tanh values), fconv[s], ws[s], energies.
 * ------------------------------------------------------------------------------------- */
typedef struct {
  int B, nb, Tp, A, D, O, E, C, K; /* B = batch stride of every buffer, nb <= B rows processed by this call
                                       (pointers pre-offset to the group's first row); K = conv half width */
  int L;                        /* number of steps buffers are sized for */
  float scaling;
  /* per-sequence inputs */
  const float* P;     /* [B][Tp][A] */
  const float* Q;     /* [B][Tp][O] */
  const float* bo;    /* [O] mlp_o bias */
  const float* wcat;  /* [4D][KX] */
  const float* bcat;  /* [4D] */
  const float* wdec;  /* [A][D]  mlp_dec.weight */
  const float* convw; /* [C][2K+1] loc_conv.weight */
  const float* watt;  /* [A][C]  mlp_att.weight */
  const float* wattT; /* [C][A]  its transpose (prepared by the caller once per sequence) */
  const float* gvec;  /* [A] */
  const float* w0;    /* [B][Tp] initial attention weights (model.py:151-153) */
  const float* xmask; /* [L][B][O+E] dropout mask for the (ctx|emb) part of X[s], or NULL */
  /* state / saved buffers */
  float* X;       /* [L+1][B][KX] */
  float* Xd;      /* [L+1][B][KX] dropout-masked copy of X (the cell's operand) when xmask != NULL, else NULL;
                     the caller fills its emb columns, the kernels fill z and ctx*mask */
  float* gates;   /* [L][B][4D] */
  float* cstate;  /* [L][B][D] */
  float* Dproj;   /* [L][B][A]   mlp_dec(z_s) */
  float* fconv;   /* [L][B][C][Tp] */
  float* S;       /* [L][B][Tp][A] */
  float* energy;  /* [L][B][Tp] */
  float* ws;      /* [L][B][Tp] attention weights */
} asr_dec_fwd_t;

int asr_dec_step_fwd(const asr_dec_fwd_t* p, int s, asr_stream_t stream);
/* attention part of step s only (AttLoc.forward model.py:139-173): z is read from X[s+1][:,0:D]; writes
 * mlp_o(context) to X[s+1][:,D:D+O] and the weights to ws[s] */
int asr_att_step_fwd(const asr_dec_fwd_t* p, int s, asr_stream_t stream);
int asr_dec_seq_fwd(const asr_dec_fwd_t* p, int s_begin, int s_end, void* graphs, asr_stream_t stream);
/* Persistent fast path of asr_dec_seq_fwd(p, 0, L): all L teacher-forced steps in one launch per 32 rows (each XCD
 * owns 4 utterances; W_cat, W_dec and the P slice stay in registers, exchanges stay in the XCD's L2).  Same results
 * except that Dproj is not written.  Returns ASR_E_SHAPE (-2) when it does not apply ((D,A,O,E) other than
 * (512,512,512,128) / (320,320,320,128), Tp > 128, C > 16, K > 100, not an 8 x 32-CU device): use asr_dec_seq_fwd.
 * xch and ctrl: the scratch pair of asr_lstm_seq_fwd_persist (sizes: asr_persist_scratch_bytes(); the decoder kernels
 * zero up to 3.6 MB of xch and use all 128 bytes of ctrl); abort convention as asr_lstm_seq_fwd_persist. */
int asr_dec_seq_fwd_persist(const asr_dec_fwd_t* p, void* xch, void* ctrl, asr_stream_t stream);
/* TESTS ONLY: asr_dec_seq_fwd_persist on the FAULT instantiation of its kernel (see ASR_DEBUG_FAULT): the launch aborts by
 * itself within a millisecond.  cfg-2 widths (512, 512, 512, 128), T' <= 102; ASR_E_SHAPE otherwise. */
int asr_dec_seq_fwd_persist_fault(const asr_dec_fwd_t* p, void* xch, void* ctrl, asr_stream_t stream);
/* Free-running variant (greedy / smooth-embedding decode, model.py:334-341; solver.py:230-231,466-470): the embedding
 * input of step s >= 1 is made inside the kernel from the logits of step s-1: mode 1 = emb[argmax], mode 2 =
 * softmax(scaling * logits) @ emb.  The caller fills X[0] / Xd[0] (embedding columns of <BOS>) and fed[0].  Written:
 * logits[s] [B][V] and pred[s] [B] (int64) for s < L-1; fed[s] [B] (int64, -1 in mode 2) and the embedding columns of
 * X[s] / Xd[s] for 1 <= s < L; probs[s] [B][V] for s < L-1 (mode 2, for the backward).  The last step's logits / pred
 * come from asr_dec_feedback_fwd(mode 3) on X[L].  V <= 64.  Same applicability rule, scratch and abort convention as
 * asr_dec_seq_fwd_persist. */
typedef struct {
  int mode;
  int V;
  int eos;             /* >= 0: a group of 4 utterances stops once each has emitted this token (decoding without a
                          backward: solver.py:212-242 strips everything after the first <EOS> anyway); the caller
                          pre-fills pred / logits of the steps that are then not run.  -1: run all L steps. */
  float scaling;
  const float* w_out;  /* [V][D+O] */
  const float* b_out;  /* [V] or NULL */
  const float* emb;    /* [V][E] */
  float* logits;       /* [L][B][V] */
  float* probs;        /* [L-1][B][V], mode 2 */
  int64_t* pred;       /* [L][B] */
  int64_t* fed;        /* [L][B] */
  /* scheduled sampling (mode 1, eos = -1; Decoder.forward with ys and tf_rate < 1, model.py:328-333): step s >= 1 is fed
   * tokens[b][s] where teacher[s] != 0 (the host's per-step draw) and its own argmax elsewhere; NULL: never the teacher */
  const int64_t* tokens;        /* [B][ld_tokens], ld_tokens >= L, or NULL */
  int64_t ld_tokens;
  const unsigned char* teacher; /* [L] */
} asr_dec_feedback_t;
int asr_dec_seq_fwd_persist_free(const asr_dec_fwd_t* p, const asr_dec_feedback_t* f, void* xch, void* ctrl,
                                 asr_stream_t stream);

/* Backward of one decoder step (reverse order s = L-1..0).
 *   G     [L+1][B][KX]  gradient wrt X; on entry G[s+1][:, 0:D+O] holds every other
 *                       contribution to d(z_s, ctx_s) (output layer); the step adds its own
 *                       and accumulates dX[s] into G[s].
 *   dwext [C][B][Tp]    partial d(w_s) coming from step s+1's location conv (in: consumed,
 *                       out: overwritten with step s's partials for step s-1).  Zero it
 *                       before the first (s = L-1) call.
 *   dws   [L][B][Tp]    optional upstream gradient of the returned attention weights
 *   dP    [B][Tp][A]    accumulated (+=);  dQw: dQ is formed by the caller as a batched
 *                       GEMM of ws and dctx after the loop.
 *   dgates[L][B][4D], dD [L][B][A]  saved per step for the deferred weight-gradient GEMMs
 *   dgvec_part [B][A], dwatt_part [B][A][C], dconv_part [B][C][2K+1]: per-utterance
 *                       partial sums accumulated over steps (caller zero-fills, reduces
 *                       over B afterwards).
 *   wcatT [KX][4D], wdecT [D][A]: transposes prepared by the caller.
 */
typedef struct {
  asr_dec_fwd_t f;
  const float* wcatT;
  const float* wdecT;
  const float* dws;   /* may be NULL */
  float* G;
  float* dwext;
  float* dwraw;       /* [B][Tp] scratch */
  float* dfpart;      /* [A/64 tiles][B][C][Tp] scratch */
  float* dP;
  float* dgates;
  float* dD;
  float* dcell;       /* [B][D] carry of dL/dc, zero-filled by the caller */
  float* dgvec_part;
  float* dwatt_part;
  float* dconv_part;
} asr_dec_bwd_t;

int asr_dec_step_bwd(const asr_dec_bwd_t* p, int s, asr_stream_t stream);
int asr_dec_seq_bwd(const asr_dec_bwd_t* p, int s_begin, int s_end, void* graphs, asr_stream_t stream);
/* Persistent fast path of asr_dec_seq_bwd(p, 0, L) (same applicability rule and scratch convention as
 * asr_dec_seq_fwd_persist; the sequence must have been teacher-forced).  mbuf: caller-allocated scratch
 * [L][B][C][Tp].  Results as the per-step path in G[:, :, D:], dgates, dD, dP; dgvec_part / dwatt_part / dconv_part
 * receive the same totals over rows (placed in other rows: reduce over B as usual); dwext, dwraw, dfpart and dcell
 * are not used.  The embedding part of dX is formed by one batched GEMM after the recurrence. */
int asr_dec_seq_bwd_persist(const asr_dec_bwd_t* p, float* mbuf, void* xch, void* ctrl, asr_stream_t stream);

/* Persistent backward of a FREE-RUNNING sequence with the smooth-embedding feedback (forward:
 * asr_dec_seq_fwd_persist_free, mode 2; Decoder.forward with ys=None, smooth=True: model.py:334-341 - the unlabeled
 * decode of the semi-supervised generator step, solver.py:465-470).  Step s's embedding input is
 * softmax(scaling * logit_{s-1}) @ emb, so d(emb_s) flows into logit_{s-1} and from there into [z_{s-1}, ctx_{s-1}]:
 * the kernel carries that path inside the recurrence (two more XCD-local hand-offs per step).
 *   probs [L-1][B][V]  the probabilities the forward saved;  w_out [V][D+O];  emb [V][E]
 *   dlfb  [L][B][V]    out (zero-filled by the caller): gradient reaching logit_s through the feedback; the caller adds
 *                      it to the upstream d(logits) before forming the output-layer weight gradients
 * On return G[s][:, D+O:] holds d(emb_s) (dropout-masked) for every step.  V <= 36, E = 128, the 4-row geometry of
 * asr_dec_seq_bwd_persist (T' <= 100 at 10 conv channels); otherwise ASR_E_SHAPE (per-step kernels +
 * asr_dec_feedback_bwd). */
typedef struct {
  int V;
  float scaling;
  const float* w_out;
  const float* emb;
  const float* probs;
  float* dlfb;
} asr_dec_feedback_bwd_t;
int asr_dec_seq_bwd_persist_free(const asr_dec_bwd_t* p, const asr_dec_feedback_bwd_t* fb, float* mbuf, void* xch,
                                 void* ctrl, asr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Parameter layout conversion: torch layout of nn.LSTM / nn.LSTMCell (gate-major rows i,f,g,o; model.py:67-68,262)
 * <-> the kernels' gate-interleaved rows (unit*4+gate).  w_ih/w_hh/b_ih/b_hh: arrays of `ndir` device pointers.
 *   asr_lstm_pack_f32    -> w_ih_cat [ndir*4H][I], w_hh_il [ndir][4H][H], bias [ndir*4H] = b_ih + b_hh
 *   asr_lstm_unpack_f32  gradients in the interleaved layout -> per-direction torch-layout dw_ih [4H][I],
 *                        dw_hh [4H][H], db [4H] (the gradient of b_ih and of b_hh)
 *   asr_cell_pack_f32    decoder cell: wcat [4D][D+O+E] = [w_hh | w_ih[:, E:E+O] | w_ih[:, :E]] interleaved, bcat
 *   asr_cell_unpack_f32  dwcat, db (interleaved) -> dw_ih [4D][E+O], dw_hh [4D][D], db [4D] (+ db2: a second copy, or NULL)
 *   asr_dec_pack_f32     asr_cell_pack_f32 + the transposed images the decoder kernels read, one launch: wcatT [D+O+E][4D]
 *                        (the backward's dX product), wdecT [D][A] of mlp_dec.weight [A][D], wattT [C][A] of mlp_att.weight
 *                        [A][C] (each output NULL = not wanted)
 *   asr_lstm_pack_multi_f32 / asr_lstm_unpack_multi_f32   asr_lstm_pack_f32 / asr_lstm_unpack2_f32 for up to
 *                        ASR_PACK_MAX_LAYERS layers in one launch (the encoder's three, the judge's two)
 *   asr_colsum_parts_f32 dst[i][j] = sum_r src[i][r * n[i] + j], r < rows, for up to four matrices: the decoder backward's
 *                        per-utterance partial gradients (gvec, mlp_att.weight, loc_conv.weight) in one launch
 * ------------------------------------------------------------------------------------- */
#define ASR_PACK_MAX_LAYERS 4
typedef struct {
  int H, I, ndir;
  const float* w_ih[2];
  const float* w_hh[2];
  const float* b_ih[2];
  const float* b_hh[2];
  float* w_ih_cat;
  float* w_hh_il;
  float* bias;
} asr_lstm_pack_job_t;
typedef struct {
  int H, I, ndir;
  const float* dw_ih_cat;
  const float* dw_hh_il;
  const float* db_il;
  float* dw_ih[2];
  float* dw_hh[2];
  float* db[2];
  float* db2[2];   /* second copy of the bias gradient (b_hh), or NULL */
} asr_lstm_unpack_job_t;
int asr_lstm_pack_multi_f32(int nlayers, const asr_lstm_pack_job_t* jobs, asr_stream_t stream);
int asr_lstm_unpack_multi_f32(int nlayers, const asr_lstm_unpack_job_t* jobs, asr_stream_t stream);
int asr_dec_pack_f32(int D, int O, int E, int A, int C, const float* w_ih, const float* w_hh, const float* b_ih,
                     const float* b_hh, const float* wdec, const float* watt, float* wcat, float* bcat, float* wcatT,
                     float* wdecT, float* wattT, asr_stream_t stream);
int asr_colsum_parts_f32(int nparts, int rows, const float* const* src, const int32_t* n, float* const* dst,
                         asr_stream_t stream);
int asr_lstm_pack_f32(int H, int I, int ndir, const float* const* w_ih, const float* const* w_hh,
                      const float* const* b_ih, const float* const* b_hh, float* w_ih_cat, float* w_hh_il,
                      float* bias, asr_stream_t stream);
int asr_lstm_unpack_f32(int H, int I, int ndir, const float* dw_ih_cat, const float* dw_hh_il, const float* db_il,
                        float* const* dw_ih, float* const* dw_hh, float* const* db, asr_stream_t stream);
/* as asr_lstm_unpack_f32, with an optional second set of bias-gradient outputs (db2[d], may be NULL): nn.LSTM has two
 * bias vectors per direction that receive the same gradient, and autograd would otherwise copy the shared tensor. */
int asr_lstm_unpack2_f32(int H, int I, int ndir, const float* dw_ih_cat, const float* dw_hh_il, const float* db_il,
                         float* const* dw_ih, float* const* dw_hh, float* const* db, float* const* db2,
                         asr_stream_t stream);

/* Teacher-forced decoder input in one launch (model.py:301-306,337: pad_list + embedding + dropout of the decoder input):
 * X [L+1][B][D+O+E] = [0 | 0 | emb_w[tokens[b][s]]] (slab L all zero), Xd (nullable) the same with xmask
 * [L][B][O+E] applied to the embedding part, fed [L][B] = tokens[b][s].  tokens: int64, element (b, s) at
 * tokens[b * tok_row_stride + s].  (D+O) % 4 == 0, E % 4 == 0, 16-byte aligned buffers. */
int asr_dec_prepare_f32(int L, int B, int D, int O, int E, const long long* tokens, int64_t tok_row_stride,
                        const float* emb_w, const float* xmask, float* X, float* Xd, long long* fed, asr_stream_t stream);
/* The gradient of those embedding rows (autograd of nn.Embedding, model.py:337): demb [V][E] (caller-zeroed or accumulating)
 * += grad[r][0..E) for every r < rows with 0 <= tokens[r] < V (a step that was not fed a token carries -1); grad row-strided
 * (ldg floats: the embedding columns of the decoder's dX buffer, read where they lie).  E % 4 == 0, ldg % 4 == 0, V * E * 4 <=
 * 64 KB (the table a workgroup folds its rows into: label matrices are skewed towards <EOS>); ASR_E_SHAPE otherwise - the
 * caller then takes its own index_add. */
int asr_embedding_grad_f32(int64_t rows, int E, int V, const long long* tokens, const float* grad, int64_t ldg,
                           float* demb, asr_stream_t stream);
int asr_cell_pack_f32(int D, int O, int E, const float* w_ih, const float* w_hh, const float* b_ih,
                      const float* b_hh, float* wcat, float* bcat, asr_stream_t stream);
int asr_cell_unpack_f32(int D, int O, int E, const float* dwcat, const float* db_il, float* dw_ih, float* dw_hh,
                        float* db, float* db2, asr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Label log-probabilities with label smoothing (Decoder.forward, model.py:354-366):
 *   out[r] = (1-ls) * log_softmax(logits[r])[index[r]] + ls * sum_v labeldist[v] * log_softmax(logits[r])[v]
 * (labeldist NULL: plain gather of the log-softmax).  rows = L*B; index is int64 (torch long).  total (NULL, or one float
 * the caller zeroed) += total_scale * sum_r out[r]: the training loss -mean(log_probs) (solver.py:377) is that sum times a
 * constant, so with total_scale = -1 / (B L) `total` IS the loss and neither a reduction nor a multiply follows.  argmax
 * (NULL, or int64 [rows]) receives argmax_v logits[r][v] (lowest index on ties: the `prediction` output of model.py:346).
 * The backward writes d(logits) for an upstream gradient grad_scale * grad_out[r * grad_stride] (grad_stride 0: ONE device
 * scalar for every row - the gradient of `total`, grad_scale = the forward's total_scale).
 * ------------------------------------------------------------------------------------- */
int asr_label_logprob_fwd(int64_t rows, int V, const float* logits, int64_t ld, const int64_t* index,
                          const float* labeldist, float ls_weight, float* out, float* total, float total_scale,
                          int64_t* argmax, asr_stream_t stream);
int asr_label_logprob_bwd(int64_t rows, int V, const float* logits, int64_t ld, const int64_t* index,
                          const float* labeldist, float ls_weight, const float* grad_out, int64_t grad_stride,
                          float grad_scale, float* dlogits, int64_t lddz, asr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Free-running decoder feedback (Decoder.forward loop, model.py:329-351): one launch per decoder step each way for
 * what sits between two steps when the next input is not a stored teacher token.
 *   forward : logits[b] = w_out [z_s, c_s] + b_out (x: row-strided [B, DO] slice of the step-input buffer, ldx);
 *             pred[b] = argmax (int64, lowest index on ties); then the embedding input of step s+1 into x_emb_next
 *             (same buffer layout, ldx) by mode:  0 = emb[pred[b]]  (greedy, model.py:336-337)
 *                                                 1 = softmax(scaling * logits[b]) @ emb  (smooth, model.py:341), the
 *                                                     probabilities are kept in probs [B, V] for the backward
 *                                                 2 = emb[tok[b * tok_stride]]  (teacher token of a tf draw, 331-333)
 *                                                 3 = nothing (last step).
 *             fed[b] = the token used (-1 for mode 1); xd_emb_next (optional) = x_emb_next * mask (dropout
 *             multipliers [B, ldm] of the embedding columns).  V <= 128.
 *   backward (mode 1 only): with demb = d loss / d x_emb of step s (row-strided, ldg) and p = probs of step s-1:
 *             dl = scaling * p * (demb emb^T - sum_v p_v (demb emb^T)_v);  dlog[b] += dl  (gradient of logits_{s-1},
 *             from which the caller takes dW_out, db_out once per sequence);  gtop[b] += dl w_out  (gradient of
 *             [z_{s-1}, c_{s-1}], same buffer layout as demb).
 * ------------------------------------------------------------------------------------- */
int asr_dec_feedback_fwd(int B, int V, int E, int DO, const float* x, int64_t ldx, const float* w_out,
                         const float* b_out, const float* emb, float* logits, int64_t* pred, int mode, float scaling,
                         const int64_t* tok, int64_t tok_stride, int64_t* fed, float* probs, float* x_emb_next,
                         float* xd_emb_next, const float* mask, int64_t ldm, asr_stream_t stream);
int asr_dec_feedback_bwd(int B, int V, int E, int DO, const float* demb, float* gtop, int64_t ldg, const float* probs,
                         const float* emb, const float* w_out, float scaling, float* dlog, asr_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Optimiser on a flat fp32 buffer (solver.py:152-153,384-385: clip_grad_norm_ + Adam(amsgrad,
 * weight_decay).step).  asr_sumsq_f32 adds sum(g^2) into the device scalar out[0] (caller zeroes
 * it); asr_adam_clip_f32 scales g by min(1, max_norm/(sqrt(*gnorm_sq)+1e-6)) (skipped when
 * gnorm_sq is NULL), folds weight_decay*p into g, updates m, v, the AMSGrad max (skipped when vmax
 * is NULL) and p.  bias_c1 = 1-beta1^t, bias_c2 = 1-beta2^t are passed by the host.  skip_if_nonzero (may be
 * NULL): a 4-byte device word; when it is not zero at the time the kernel runs, nothing is updated - the abort latch
 * of the persistent kernels (or its all-reduced sum), so that the step can be enqueued before the host has read it.
 * zero_word (may be NULL): one float the kernel sets to zero whether or not the update is skipped - the accumulator the
 * NEXT step's asr_sumsq_f32 adds into (a caller that alternates between two words needs no fill launch per step).
 * ------------------------------------------------------------------------------------- */
int asr_sumsq_f32(int64_t n, const float* g, float* out, asr_stream_t stream);
/* The gradient gather in front of them in a one-process step: flat[dst_offset[j] .. + count[j]) = src[j][0 .. count[j]) for
 * njobs contiguous tensors (what autograd left in the parameters' .grad) and, if sumsq != NULL, sumsq[0] += the sum of their
 * squares - torch._foreach_copy_ and asr_sumsq_f32 in one pass over the gradients (the norm of clip_grad_norm_,
 * solver.py:384, is taken where they are read anyway).  Jobs beyond ASR_GATHER_MAX_JOBS take further launches. */
#define ASR_GATHER_MAX_JOBS 64
int asr_gather_sumsq_f32(int njobs, const float* const* src, const int64_t* dst_offset, const int64_t* count, float* flat,
                         float* sumsq, asr_stream_t stream);
int asr_adam_clip_f32(int64_t n, float* p, const float* g, float* m, float* v, float* vmax,
                      const float* gnorm_sq, float max_norm, float lr, float beta1, float beta2, float eps,
                      float weight_decay, float bias_c1, float bias_c2, const void* skip_if_nonzero, float* zero_word,
                      asr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ASR_HIP_H */
