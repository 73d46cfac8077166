"""bench.py — utterances/sec of a full train step (forward + loss + backward + gradient all-reduce + clip +
Adam) of the seq2seq ASR hot path on N MI355X GPUs, plus the dominant kernel's roofline fraction, the per-layer
encoder gate-GEMM fractions and a CPU baseline (the oracle port) timed on the host cores of the same box.

    python bench.py --gpus 1 --steps 10 --warmup 3                                  # cfg-2, the headline line
    python bench.py --config cfg5            |  --config cfg1  |  --frames 400       # the other BASELINE.json shapes
    python bench.py --scaling strong --global-batch 256                              # fixed total work over N GPUs
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Default workload = BASELINE.json configs[1] ("cfg-2"): 3x512 pyramidal BiLSTM encoder, 512 LSTM decoder with
location-aware attention, V=34, batch 32 per GPU, 80-dim x 800-frame synthetic fbank (ragged lengths
U[0.6T, T], longest pinned to T; label length 0.125 T_i), dropout 0.3 as in config.yaml, fp32.
Weak scaling (default): 32 utterances per GPU; at N=8 the global batch is 256 (configs[2]'s shape).
Strong scaling: --global-batch utterances in total, split over the ranks (N=1 runs all of them on one GPU).
Every rank pads its strided shard to the global T_max / olength (exact-parity mode).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "semi-supervised-asr_amd"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

import synth

# BASELINE.json configs: (model dims, utterances per GPU, frames); dropout 0.3 = config.yaml
CONFIGS = {
    "cfg1": dict(model=dict(synth.CFG1, dropout_rate=0.3), batch=4, frames=200,
                 name="cfg-1: config.yaml-shaped 1x128 pBiLSTM enc / 320 LSTM dec (BASELINE.json configs[0])"),
    "cfg2": dict(model=dict(synth.CFG2, dropout_rate=0.3), batch=32, frames=800,
                 name="cfg-2: 3x512 pBiLSTM enc / 1x512 LSTM dec + location attention (BASELINE.json configs[1])"),
    "cfg5": dict(model=dict(synth.CFG2, dropout_rate=0.3), batch=8, frames=1600,
                 name="cfg-5: cfg-2's model on long utterances (BASELINE.json configs[4])"),
}
HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TF = 157.3                       # fp32-input MFMA, dense (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TF = 2500.0                     # bf16 MFMA, dense
# Product arithmetic (include/asr_hip.h ASR_ARITH_*; --arith).  The default, bf16x6, re-encodes every fp32 operand
# losslessly as three bf16 terms and spends SIX bf16 MFMA products per algorithmic product (fp32-equivalent); bf16x3 (two
# terms, three products, 16 significand bits) is the non-default fast mode; f32 is the fp32-input MFMA.  `achieved`
# counts ALGORITHMIC flops, so the roof a kernel is priced against is the peak of the pipe it runs on divided by the
# products it spends per product.
PRODUCTS_PER_PRODUCT = {"bf16x6": 6.0, "bf16x3": 3.0, "f32": 1.0}


def arith_peak_tf(name):
    return MFMA_F32_PEAK_TF if name == "f32" else MFMA_BF16_PEAK_TF / PRODUCTS_PER_PRODUCT[name]


ARITH_TEXT = {
    "bf16x6": "fp32-equivalent: fp32 operands, accumulators and results; every MFMA product on the bf16 pipe with both "
              "operands re-encoded losslessly as three bf16 terms (a + b + c == x exactly) and six products per product; "
              "the dropped terms are <= 2^-24 |x y|, below the rounding of the fp32 product (tests: bf16x6 error vs float64 "
              "within 2x of the fp32-input MFMA kernel's, cfg-2 / cfg-5 parity vs the reference under this arithmetic AND "
              "under the exact fp32-input MFMA)",
    "bf16x3": "NOT the reference's precision: fp32 operands and accumulators, products on the bf16 pipe with two bf16 terms "
              "per operand (16 significand bits) and three products per product, <= 2^-15 relative each",
    "f32": "fp32 operands, accumulators and results; products on the fp32-input MFMA (v_mfma_f32_32x32x2_f32 / 4x4x1) - except ONE "
           "product of the decoder's backward, the embedding part of dX (dgates W_cat[:, D+O:], [L*B, 4D] x [4D, E]), which "
           "asr_dec_seq_bwd_persist* always forms with the fp32-equivalent bf16x6 products",
}
CFG2 = CONFIGS["cfg2"]["model"]                 # used by tools/


def global_batch(n_utt, t_max, seed):
    """The synthetic batch generator of the benchmark (tools/ use it too)."""
    return synth.ragged_batch(n_utt, t_max, CFG2["input_dim"], CFG2["output_dim"], seed)


def fwd_flops_per_utt(c, t_frames, l_plus_1):
    """SURVEY 8d F_fwd (algorithmic, per utterance) for model dims c."""
    H, I = c["enc_hidden_dim"], c["input_dim"]
    f, t = 0.0, t_frames
    for layer in range(c["enc_n_layers"]):
        idim = I if layer == 0 else H
        f += 2.0 * t * 2 * 4 * H * (idim + H)
        t2 = (t + 1) // 2 if c["subsample"][layer] > 1 else t
        f += 2.0 * t2 * (4 * H if c["subsample"][layer] > 1 else 2 * H) * H
        t = t2
    tp = t
    A, D, O, E, V = c["att_dim"], c["dec_hidden_dim"], c["att_odim"], c["embedding_dim"], c["output_dim"]
    C, K = c["conv_channels"], c["conv_kernel_size"]
    f += 2.0 * tp * H * A
    f += l_plus_1 * (2.0 * 4 * D * (E + O + D) + 2 * D * A + 2 * C * (2 * K + 1) * tp + 2 * C * A * tp + 2 * A * tp
                     + 2 * tp * H + 2 * H * O + 2 * (D + O) * V)
    return f


def usable_cpus():
    """CPU share of this container: affinity mask, capped by the cgroup quota and by 16 (the GPU box gives one
    GPU's job 16 cores; os.cpu_count() reports the whole host and oversubscribes badly)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def note(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def cpu_baseline(cfg, xs, lens, ys, label, min_steps):
    """The oracle (CPU port of the reference path, oracle/asr_oracle.py) running the SAME train step on the SAME
    synthetic batch the GPU was timed on (identical batches; 1 step at the cfg-2 / cfg-5 shapes, >= 3 at
    cfg-1), on the host cores of this box."""
    from oracle import asr_oracle as O
    ncores = usable_cpus()
    note("cpu baseline: %s on %d threads" % (label, ncores))
    torch.set_num_threads(ncores)
    mcfg = dict(cfg, labeldist=synth.labeldist(cfg["output_dim"], 5))
    sd = O.make_leaf_state(synth.e2e_weights(cfg, 99))
    opt = O.AdamAmsgrad(O.unique_param_names(sd), lr=5e-4, weight_decay=1e-6)
    xs_t, ys_t = torch.from_numpy(xs), [torch.from_numpy(y) for y in ys]
    t0 = time.perf_counter()
    steps = 0
    while True:
        O.sup_train_step(sd, mcfg, opt, xs_t, lens, ys_t, max_grad_norm=5.0)
        steps += 1
        el = time.perf_counter() - t0
        if steps >= min_steps and (el > 20.0 or steps >= 10):
            break
    return dict(value=len(lens) * steps / el, unit="utterances/sec", cores=ncores, kind="port", seconds=el,
                sample="%d full train step(s) of oracle/asr_oracle.py on the whole %s batch the GPU ran (%d utterances, "
                       "T=%d, dropout %.1f, clip 5 + Adam(amsgrad)), %d torch CPU threads.  A port, and FASTER than the "
                       "reference's own CPU path: the oracle unbinds the time axis, which removes the O(T^2) zero-fill of "
                       "the reference's packed-LSTM backward (SURVEY 6: 0.54 utt/s for the real reference at this shape "
                       "on 8 threads)"
                       % (steps, label, len(lens), max(lens), cfg["dropout_rate"], ncores))


def encoder_layer_frames(c, t_frames):
    out, t = [], t_frames
    for layer in range(c["enc_n_layers"]):
        out.append(t)
        t = (t + 1) // 2 if c["subsample"][layer] > 1 else t
    return out, t


def tn_gemm_shapes(c, B, t_frames, olength, fused_dw_hh=True, rows=None):
    """The weight-gradient (transA) GEMM launches of ONE train step: (M, N, K, batch).  dW_hh is among them only when the
    persistent LSTM backward kernel does not accumulate it itself (bf16x6: one GEMM per layer, batched over the directions).
    rows: the packed row counts of the encoder layers (+ the output), hb.RowLayout.rows - K of the encoder's products; None:
    the padded counts T_l * B."""
    H, I = c["enc_hidden_dim"], c["input_dim"]
    D, O, E, A, V = c["dec_hidden_dim"], c["att_odim"], c["embedding_dim"], c["att_dim"], c["output_dim"]
    shapes = []
    frames, tp = encoder_layer_frames(c, t_frames)
    for layer, t in enumerate(frames):
        idim = I if layer == 0 else H
        k_in = rows[layer] if rows is not None else t * B
        shapes.append((8 * H, idim, k_in, 1))                        # dW_ih (both directions)
        if not fused_dw_hh and t > 1:
            shapes.append((4 * H, H, k_in if rows is not None else (t - 1) * B, 2))     # dW_hh = dG^T h_prev, batched over the directions
        sub = c["subsample"][layer] > 1
        shapes.append((H, 4 * H if sub else 2 * H, rows[layer + 1] if rows is not None else ((t + 1) // 2 if sub else t) * B, 1))   # dW of the projection
    L = olength
    shapes += [(A, H, tp * B, 1), (O, H, tp * B, 1)]                 # mlp_enc, mlp_o (hoisted)
    shapes += [(V, D + O, L * B, 1), (4 * D, D + O + E, L * B, 1), (A, D, L * B, 1)]   # output layer, cell, mlp_dec
    shapes.append((tp, O, L, B))                                     # dQ, batched over utterances
    return shapes


def _time_events(fn, reps):
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def mfma_busy_table():
    """MFMA-pipe busy fractions per kernel family from the committed counter pass (tools/pmc_mfma.py ->
    profiles/r06_pmc_mfma.json (r05 ... r03 when absent): SQ_VALU_MFMA_BUSY_CYCLES against SQ_BUSY_CYCLES-derived kernel cycles), keyed for the rows
    of this bench; {} when the file is absent or was collected under another arithmetic."""
    import hip_backend as hb
    try:
        for name in ("r06_pmc_mfma.json", "r05_pmc_mfma.json", "r04_pmc_mfma.json", "r03_pmc_mfma.json"):
            path = os.path.join(ROOT, "profiles", name)
            if os.path.exists(path):
                with open(path) as f:
                    d = json.load(f)
                return d.get(hb.arith_name(), {}).get("bench_keys", {})
        return {}
    except (OSError, ValueError):
        return {}


def encoder_gate_gemms(dev, c, B, t_frames, lens=None):
    """north_star: ">= 40 % MFMA utilisation on the encoder gate GEMM".  Per encoder layer, on operands of the step's
    shapes that were evicted from the caches before each launch (a 512 MB fill in between: the train step finds them in
    HBM too): the input-gate projection [T*B, in] x [in, 8H] (forward), its two backward GEMMs (dX = dG W_ih,
    dW_ih = dG^T X), and the recurrent products of the persistent kernels (h W_hh^T forward; dG W_hh + dG^T h backward)
    timed as whole kernels.  Each entry: achieved algorithmic TFLOP/s; `frac` = fraction of the roof of the pipe and
    arithmetic it runs on (bf16 peak / products per product: 416.7 TF for bf16x6); `x_f32_mfma_peak` = the same throughput
    as a MULTIPLE of the 157.3 TF fp32-input MFMA peak (a speed ratio against the pipe the reference's arithmetic would
    otherwise need, not a utilisation); `mfma_busy_frac` = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE-
    equivalent cycles) of the same kernel from the committed PMC pass (profiles/r05_pmc_mfma.json) when it exists."""
    import hip_backend as hb
    an = hb.arith_name()
    peak = arith_peak_tf(an)
    busy = mfma_busy_table()
    lib = hb.load()
    H, I = c["enc_hidden_dim"], c["input_dim"]
    frames, _ = encoder_layer_frames(c, t_frames)
    # the encoder runs on PACKED rows (hb.RowLayout): M of layer l = the sum of the utterances' extents, not T_l * B
    layout = hb.RowLayout([int(v) for v in lens], [c["subsample"][i] for i in range(c["enc_n_layers"])], dev) \
        if (lens is not None and hb.USE_PACKED_ROWS) else None
    flush = torch.empty(128 * 1024 * 1024, device=dev)
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def cold(fn):
        best = None
        for _ in range(3):
            flush.fill_(1.0)
            e0.record(stream)
            fn()
            e1.record(stream)
            torch.cuda.synchronize()
            dt = e0.elapsed_time(e1) * 1e-3
            best = dt if best is None else min(best, dt)
        return best

    out = []
    g = torch.Generator().manual_seed(7)
    xch, ctrl = hb.persist_scratch(dev)
    for layer, T in enumerate(frames):
        idim = I if layer == 0 else H
        M = layout.rows[layer] if layout is not None else T * B
        x = torch.randn(M, idim, device=dev)
        w = torch.randn(8 * H, idim, device=dev) / np.sqrt(idim)
        dG = torch.randn(M, 8 * H, device=dev)
        gates = torch.empty(M, 8 * H, device=dev)
        bias = torch.zeros(8 * H, device=dev)
        rows = []
        for op, fn, flops in (
                ("in-proj fwd  [R=%d,%d]x[%d,8H]" % (M, idim, idim), lambda: hb.gemm(x, w, trans_b=True, bias=bias, out=gates),
                 2.0 * M * idim * 8 * H),
                ("dX = dG W_ih [R,8H]x[8H,%d]" % idim, lambda: hb.gemm(dG, w), 2.0 * M * idim * 8 * H),
                ("dW_ih = dG^T X [8H,R]x[R,%d]" % idim, lambda: hb.gemm(dG, x, trans_a=True), 2.0 * M * idim * 8 * H)):
            if layer == 0 and op.startswith("dX"):
                continue                                  # the features need no gradient
            dt = cold(fn)
            rows.append(dict(layer=layer, op=op, us=dt * 1e6, tflops=flops / dt / 1e12, frac=flops / dt / 1e12 / peak,
                             x_f32_mfma_peak=flops / dt / 1e12 / MFMA_F32_PEAK_TF,
                             mfma_busy_frac=busy.get("gemm/%d/%s" % (layer, op.split(" ")[0]))))
        del x, w, dG, gates
        # recurrent products: the persistent kernels of this layer (both directions)
        lens_t = torch.full((B,), T, dtype=torch.int32, device=dev)     # (time-major, every row full length: us per time step)
        gts = (torch.randn(T, B, 2, 4 * H, generator=g) * 0.5).to(dev)
        whh = (torch.randn(2, 4 * H, H, generator=g) / np.sqrt(H)).to(dev)
        y = torch.empty(T, B, 2 * H, device=dev)
        cst = torch.empty(T, B, 2 * H, device=dev)
        torch.cuda.synchronize()
        e0.record(stream)
        rc = lib.asr_lstm_seq_fwd_persist(T, B, B, H, 2, hb.ptr(gts), hb.ptr(whh), hb.ptr(lens_t), None, None, None, hb.ptr(y), hb.ptr(cst),
                                          hb.c_p(xch.data_ptr()), hb.c_p(ctrl.data_ptr()), hb.current_arith(), hb.stream())
        e1.record(stream)
        torch.cuda.synchronize()
        if rc == 0:
            dt = e0.elapsed_time(e1) * 1e-3
            fl = T * 2.0 * B * 4 * H * H * 2
            rows.append(dict(layer=layer, op="recurrent fwd h W_hh^T (persistent kernel, T=%d)" % T, us=dt * 1e6,
                             us_per_time_step=dt / T * 1e6, tflops=fl / dt / 1e12, frac=fl / dt / 1e12 / peak,
                             x_f32_mfma_peak=fl / dt / 1e12 / MFMA_F32_PEAK_TF, mfma_busy_frac=busy.get("lstm_fwd")))
            dy = (torch.randn(T, B, 2 * H, generator=g) * 0.01).to(dev)
            dw = torch.zeros(2, 4 * H, H, device=dev)
            db = torch.zeros(2 * 4 * H, device=dev)
            whT = whh.transpose(1, 2).contiguous()
            torch.cuda.synchronize()
            e0.record(stream)
            rc = lib.asr_lstm_seq_bwd_persist(T, B, B, H, 2, hb.ptr(gts), hb.ptr(whT), hb.ptr(lens_t), None, None, None, hb.ptr(dy), hb.ptr(cst),
                                              hb.ptr(y), hb.ptr(dw), hb.ptr(db), hb.c_p(xch.data_ptr()),
                                              hb.c_p(ctrl.data_ptr()), hb.current_arith(), hb.stream())
            e1.record(stream)
            torch.cuda.synchronize()
            hb.check(rc, "asr_lstm_seq_bwd_persist")
            dt = e0.elapsed_time(e1) * 1e-3
            fused = lib.asr_lstm_bwd_persist_fuses_dw(H, hb.current_arith()) == 1
            nprod = 2 if fused else 1          # bf16x6: dW_hh is a GEMM after the kernel, the kernel's flops are dh_rec alone
            rows.append(dict(layer=layer, op="recurrent bwd dG W_hh%s (persistent kernel, T=%d)" % (" + dG^T h" if fused else "", T),
                             us=dt * 1e6, us_per_time_step=dt / T * 1e6, tflops=nprod * fl / dt / 1e12,
                             frac=nprod * fl / dt / 1e12 / peak, x_f32_mfma_peak=nprod * fl / dt / 1e12 / MFMA_F32_PEAK_TF,
                             mfma_busy_frac=busy.get("lstm_bwd")))
        out += rows
    return out


def kernel_roofline(dev, c, B, t_frames, olength, lens_batch=None):
    """Roofline of the dominant kernel of the train step, timed live with HIP events on the launch stream.

    Two kernels compete for "dominant by total time per step" (profiles/*_bench_kernel_stats*.csv), so both are
    replayed on synthetic operands of the step's exact shapes and the one with the larger per-step total is reported
    as the roofline (the other goes to `also`):
      * lstm_persist_bwd_rs_kernel<H> (lstm_persist_bwd_kernel<H> where the exchanged-partials form does not apply): one
        launch per encoder layer and row block; algorithmic flops per time step =
        2*B*4H*H*ndir for dh_rec = dG W_hh, plus the same again where the kernel fuses dW_hh += dG^T h (bf16x3 / f32; under
        the default bf16x6 that product is a batched GEMM after the kernel and counts with the GEMMs).
      * gemm_bf3w_kernel<false,false> (gemm_bf3_kernel / gemm_f32_kernel with the wide tile or the split products switched
        off): the transA GEMMs that form the remaining weight gradients.
    `achieved` = algorithmic flops of those launches / their total time; us_per_launch is directly comparable with
    rocprofv3's average duration for that kernel name."""
    import hip_backend as hb
    lib = hb.load()
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    # ---- transA GEMMs
    packed = lens_batch is not None and hb.USE_PACKED_ROWS
    layout = hb.RowLayout([int(v) for v in lens_batch], [c["subsample"][i] for i in range(c["enc_n_layers"])], dev) if packed else None
    shapes = tn_gemm_shapes(c, B, t_frames, olength,
                            fused_dw_hh=(not packed) and lib.asr_lstm_bwd_persist_fuses_dw(c["enc_hidden_dim"], hb.current_arith()) != 0,
                            rows=layout.rows if packed else None)
    bufs = []
    for (M, N, K, batch) in shapes:
        bufs.append((torch.randn(batch * K, M, device=dev) if batch == 1 else torch.randn(K, batch, M, device=dev),
                     torch.randn(batch * K, N, device=dev) if batch == 1 else torch.randn(K, batch, N, device=dev),
                     torch.empty(batch * M, N, device=dev)))

    def run_all():
        for (M, N, K, batch), (a_, b_, o_) in zip(shapes, bufs):
            if batch == 1:
                hb.gemm(a_, b_, trans_a=True, out=o_)
            else:
                hb.gemm_batched(a_, b_, o_, True, False, M, N, K, batch * M, batch * N, N, batch, M, N, M * N)

    gemm_s = _time_events(run_all, 3)
    gemm_flops = sum(2.0 * M * N * K * batch for (M, N, K, batch) in shapes)
    del bufs
    an = hb.arith_name()
    busy = mfma_busy_table()
    gpeak = arith_peak_tf(an)
    gname = {"f32": "gemm_f32_kernel<false,false>", "bf16x6": "gemm_bfs_kernel<false,false,3,*> (one wave per SIMD; small shapes: gemm_bf3_kernel)",
             "bf16x3": "gemm_bfs_kernel<false,false,2,*>"}[an]
    gemm = dict(bound="mfma", kernel="%s (transA weight-gradient GEMMs of one step)" % gname,
                achieved=gemm_flops / gemm_s / 1e12, peak=gpeak, unit="TFLOP/s",
                peak_is="%s MFMA dense peak / %d products per product" % ("fp32-input" if an == "f32" else "bf16", PRODUCTS_PER_PRODUCT[an]),
                frac=gemm_flops / gemm_s / 1e12 / gpeak, x_f32_mfma_peak=gemm_flops / gemm_s / 1e12 / MFMA_F32_PEAK_TF,
                mfma_busy_frac=busy.get("gemm_tn"), traffic=None, launches_per_step=len(shapes),
                us_per_launch=gemm_s / len(shapes) * 1e6, ms_per_step=gemm_s * 1e3)

    # ---- persistent LSTM backward, the encoder layers, AS THE STEP RUNS IT: packed rows (hb.RowLayout) with the lengths of the
    # bench's batch - the `..., true>` instantiation rocprofv3 lists.  Algorithmic flops: the valid (utterance, frame) pairs
    # only, 2 * 4H * H per pair and direction (the kernel's MFMAs also run over the dead rows of a group whose other rows are
    # still alive; those are not counted).  Without lengths (or ASR_ENCODER_ROWS=padded): time-major, every row T frames.
    H = c["enc_hidden_dim"]
    g = torch.Generator().manual_seed(3)
    layers, _ = encoder_layer_frames(c, t_frames)
    w = (torch.randn(2, H, 4 * H, generator=g) / np.sqrt(H)).to(dev)
    R0 = layout.rows[0] if packed else layers[0] * B
    gates0 = (torch.rand(R0, 2, 4 * H, generator=g) * 0.8 + 0.1).to(dev)
    gates = torch.empty_like(gates0)
    dy = (torch.randn(R0, 2 * H, generator=g) * 0.01).to(dev)
    cst = torch.randn(R0, 2 * H, generator=g).to(dev)
    y = torch.tanh(torch.randn(R0, 2 * H, generator=g)).to(dev)
    dw = torch.zeros(2, 4 * H, H, device=dev)
    xch, ctrl = hb.persist_scratch(dev)
    lstm_s, lstm_flops, launches, steps_total = 0.0, 0.0, 0, 0
    fused = (not packed) and lib.asr_lstm_bwd_persist_fuses_dw(H, hb.current_arith()) == 1
    db = torch.zeros(2 * 4 * H, device=dev)
    rows_per_launch = 16 if B <= 16 else 32
    lens_full = torch.empty(B, dtype=torch.int32, device=dev)
    for li, T in enumerate(layers):
        if packed:
            rows = hb.LayerRows(layout, li)
            T, lens_d, rb, re_, rh = rows.T, rows.lens, hb.ptr(rows.base), hb.ptr(rows.ext), rows.host_ptr()
            valid = float(layout.lens[li].sum())
        else:
            lens_full.fill_(T)
            lens_d, rb, re_, rh, valid = lens_full, None, None, None, float(T * B)
        best = None
        for _rep in range(3):                      # first pass warms clocks / code; report the best of the next two
            gates.copy_(gates0)
            torch.cuda.synchronize()
            e0.record(stream)
            rc = lib.asr_lstm_seq_bwd_persist(T, B, B, H, 2, hb.ptr(gates), hb.ptr(w), hb.ptr(lens_d), rb, re_, rh, hb.ptr(dy),
                                              hb.ptr(cst), hb.ptr(y), hb.ptr(dw), hb.ptr(db), hb.c_p(xch.data_ptr()),
                                              hb.c_p(ctrl.data_ptr()), hb.current_arith(), hb.stream())
            e1.record(stream)
            torch.cuda.synchronize()
            if rc == -2:
                return dict(gemm, note="persistent LSTM kernels do not cover H=%d" % H)
            hb.check(rc, "asr_lstm_seq_bwd_persist")
            if _rep > 0:
                dt = e0.elapsed_time(e1) * 1e-3
                best = dt if best is None else min(best, dt)
        lstm_s += best
        lstm_flops += valid * (2.0 if fused else 1.0) * (2.0 * 4 * H * H * 2)
        launches += (B + rows_per_launch - 1) // rows_per_launch
        steps_total += T * ((B + rows_per_launch - 1) // rows_per_launch)
    rs = an != "f32" and H in (128, 256, 512)
    lpeak = gpeak
    lstm = dict(bound="mfma", kernel="%s<%d> (dG recurrence%s, %d encoder layers%s)"
                                     % ("lstm_persist_bwd_rs_kernel" if rs else "lstm_persist_bwd_kernel", H,
                                        " + fused dW_hh" if fused else "; dW_hh is a batched GEMM after it", len(layers),
                                        "; packed rows, the batch's own lengths: the `true` instantiation" if packed else ""),
                algorithmic_flops_per_time_step=(2.0 if fused else 1.0) * (2.0 * B * 4 * H * H * 2),
                algorithmic_flops_counted="valid (utterance, frame) pairs of the batch: %.0f of the %d x T the kernel's row groups walk"
                                          % (lstm_flops / ((2.0 if fused else 1.0) * (2.0 * 4 * H * H * 2)), B) if packed else "B x T",
                limiter="dependent chain: per time step two barriers and one L2 hand-off of partial sums between the 32 CUs "
                        "of an XCD; the MFMA floor of the step is ~0.3-0.6 us",
                achieved=lstm_flops / lstm_s / 1e12, peak=lpeak, unit="TFLOP/s", peak_is=gemm["peak_is"],
                frac=lstm_flops / lstm_s / 1e12 / lpeak, x_f32_mfma_peak=lstm_flops / lstm_s / 1e12 / MFMA_F32_PEAK_TF,
                mfma_busy_frac=busy.get("lstm_bwd"), traffic=None, launches_per_step=launches,
                us_per_launch=lstm_s / launches * 1e6, ms_per_step=lstm_s * 1e3,
                us_per_time_step=lstm_s / max(1, steps_total) * 1e6,
                aborted=bool(hb.persist_aborted(dev)))
    # HBM-side traffic of that kernel from the committed PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs of
    # tools/pmc_probe.py, FETCH doubled as the gfx950 guide prescribes); measured at H=512, 8-row groups
    if H == 512 and B >= 32:
        for name in ("r06_pmc_lstm_persist.json", "r05_pmc_lstm_persist.json", "r04_pmc_lstm_persist.json", "r03_pmc_lstm_persist.json", "r02_pmc_lstm_persist.json", "r01_pmc_lstm_persist.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    pmc = json.load(f)["lstm_persist_bwd_kernel<512>"]
                scale = (B / 32.0) * steps_total / launches          # PMC bytes are per time step of a 32-row launch
                lstm["traffic"] = pmc["hbm_side_bytes_per_time_step"] * scale
                lstm["traffic_unit"] = "bytes/launch (PMC bytes per time step x mean T of the launches), " + name
                lstm["algorithmic_bytes_per_launch"] = pmc["algorithmic_bytes_per_time_step"] * scale
                break
            except (OSError, KeyError, ValueError):
                continue
    first, second = (lstm, gemm) if lstm_s >= gemm_s else (gemm, lstm)
    first["also"] = second
    return first


def row_geometry(n_rows, H, ndir=2):
    """How the persistent LSTM kernels walk a batch of n_rows utterances on ONE GPU (csrc/lstm_persist.hip launchers): every
    launch is one traversal of the layer's serial chain.  Explains the N = 1 base of a strong-scaling curve."""
    per8, fwd, left = 8 * (8 // ndir), [], n_rows
    while left > 0:
        if H == 512 and left >= 16 * (8 // ndir):
            fwd.append("%d rows (16 per XCD group)" % (16 * (8 // ndir)))
            left -= 16 * (8 // ndir)
        else:
            rows = min(left, per8)
            fwd.append("%d rows (%d per XCD group)" % (rows, 4 if n_rows <= 4 * (8 // ndir) else 8))
            left -= rows
    nb = (n_rows + per8 - 1) // per8
    return dict(lstm_forward_passes=len(fwd), lstm_forward=fwd, lstm_backward_passes=nb,
                lstm_backward="%d x %d-row passes" % (nb, min(n_rows, per8)),
                note="packed rows: a pass runs max(len of ITS rows) steps, so the later passes of a length-sorted batch are shorter")


VOCAB = ["<PAD>", "<BOS>", "<EOS>"] + ["c%02d" % i for i in range(29)] + ["<space>", "<NOISE>"]       # V = 34


def make_solver(model_cfg, batch_per_gpu, frames, workdir, **overrides):
    """The product's Solver (semi-supervised-asr_amd/solver.py) on a synthetic corpus written to `workdir`: the same
    object main.py builds from config.yaml, with the benchmark's model sizes and the deterministic synthetic weights.
    The corpus only feeds what the constructor derives from data (vocabulary, label distributions, length proportion);
    the timed batches are the synthetic batches of this file, already resident in HBM."""
    import contextlib
    import pickle
    import yaml
    from dataset import synthetic_utterances
    from solver import Solver
    os.makedirs(workdir, exist_ok=True)
    vocab = {sym: i for i, sym in enumerate(VOCAB)}
    assert len(vocab) == model_cfg["output_dim"]
    for name, n, seed in (("train", 16, 1), ("dev", 4, 2)):
        with open(os.path.join(workdir, name + ".pkl"), "wb") as f:
            pickle.dump(synthetic_utterances(n, model_cfg["input_dim"], len(vocab), max(64, frames // 4), seed), f)
    with open(os.path.join(workdir, "vocab_dict.pkl"), "wb") as f:
        pickle.dump(vocab, f)
    with open(os.path.join(workdir, "non_lang_syms.pkl"), "wb") as f:
        pickle.dump(["<NOISE>", "<PAD>", "<BOS>", "<EOS>"], f)
    with open(os.path.join(ROOT, "semi-supervised-asr_amd", "config.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg.update({k: v for k, v in model_cfg.items() if k in cfg})
    cfg.update(logdir=os.path.join(workdir, "log"), model_dir=workdir, model_name="m", load_model_path=os.path.join(workdir, "m"),
               load_judge_path=os.path.join(workdir, "m"), dataset_root_dir=workdir,
               vocab_path=os.path.join(workdir, "vocab_dict.pkl"), non_lang_syms_path=os.path.join(workdir, "non_lang_syms.pkl"),
               labeled_set="train", unlabeled_speech_set="train", unlabeled_text_set="train", dev_set="dev", test_set="dev",
               min_feature_length=8, batch_size=batch_per_gpu, shuffle=False, learning_rate=5e-4, weight_decay=1e-6,
               max_grad_norm=5)
    cfg.update(overrides)
    with contextlib.redirect_stdout(sys.stderr):             # the Solver prints its config and modules: stdout is the JSON line's
        solver = Solver(cfg)
    mcfg = dict(model_cfg)
    solver.model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.e2e_weights(mcfg, 99).items()})
    solver.model.train()
    return solver


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg2")
    ap.add_argument("--frames", type=int, default=None, help="T of the synthetic 80 x T batches (200/400/800/1600)")
    ap.add_argument("--batch-per-gpu", type=int, default=None)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--global-batch", type=int, default=256, help="total utterances with --scaling strong")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-layer-gemms", action="store_true")
    ap.add_argument("--no-workloads", action="store_true", help="skip the cfg-1 / cfg-5 / ssl / judge / decode sub-objects")
    ap.add_argument("--dropout", type=float, default=None)
    ap.add_argument("--arith", choices=["bf16x6", "f32", "bf16x3"], default="bf16x6",
                    help="product arithmetic of the MFMA kernels (default: bf16x6 = fp32-equivalent, see ARITH_TEXT)")
    ap.add_argument("--no-also", action="store_true", help="skip the extra timings under the other arithmetics")
    ap.add_argument("--epoch-only", type=int, default=0, metavar="STEPS",
                    help="only the epoch workload (Solver.sup_train_one_epoch through the input pipeline), for profiling")
    args = ap.parse_args()
    # stdout carries ONE line, the JSON record.  Libraries write there too - RCCL prints a five-line version banner from C when a
    # communicator comes up, on every rank - so file descriptor 1 is pointed at stderr for the whole run and the record goes
    # out through a private duplicate of the original descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import contextlib
    import tempfile
    import __graft_entry__ as entry
    entry.build()
    import parallel
    import torch.distributed as dist

    import hip_backend as hb
    hb.ARITH[0] = hb.ARITH_NAMES[args.arith]
    rank, world, local = parallel.init_distributed()
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback in the product path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    spec = CONFIGS[args.config]
    cfg = dict(spec["model"])
    if args.dropout is not None:
        cfg["dropout_rate"] = args.dropout
    t_frames = args.frames or spec["frames"]
    if args.scaling == "strong":
        n_global = args.global_batch
        assert n_global % world == 0, "--global-batch must divide by the number of GPUs"
    else:
        n_global = (args.batch_per_gpu or spec["batch"]) * world
    torch.manual_seed(1000 + rank)                     # per-rank dropout streams; weights below are shared
    tmp = tempfile.mkdtemp(prefix="asr_bench_r%d_" % rank)
    if args.epoch_only:
        assert world == 1
        res = epoch_workload(dev, tmp, args.epoch_only, args.config)
        note("epoch: %.2f ms/step" % res["ms_per_step"])
        os.write(json_fd, (json.dumps(res) + "\n").encode())
        return

    # ---- the timed object is the product's Solver: one step = Solver.sup_train_one_iteration on the (global) batch, i.e.
    # forward on this rank's strided shard, loss, zero_grad, backward, ONE all-reduce of the flat gradient buffer, clip + Adam,
    # and the host read of loss + abort latch (the reference's loss.item(), solver.py:379).  That read is pipelined: the
    # Adam kernel checks the latch on the device - under data parallelism the latch summed over the ranks by the step's
    # all-reduce - and the host reads step i's record while step i + 1 runs (Solver._step, parallel.DpPipeline)
    solver = make_solver(cfg, n_global // world, t_frames, os.path.join(tmp, "main"))
    xs, lens, ys = synth.ragged_batch(n_global, t_frames, cfg["input_dim"], cfg["output_dim"], 1234)
    xs_d = torch.from_numpy(np.ascontiguousarray(xs)).to(dev)        # inputs resident in HBM before timing (every rank
    ys_d = [torch.from_numpy(y).to(dev) for y in ys]                  # holds the global batch, as the Solver's loaders give it)
    info = dict(t_max=int(max(lens)), olength=max(int(y.shape[0]) for y in ys) + 1)
    b_local = len(parallel.shard_indices(n_global, rank, world))
    last = {}

    def step():
        last["loss"] = solver.sup_train_one_iteration(xs_d, lens, ys_d, 1.0)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n, fn=None):
        """n steps bracketed by barrier + synchronize on both sides; wall seconds, max over ranks."""
        fn = fn or step
        fence()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        solver.flush()                                # the host record (loss + abort latch) of the last step too
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    if rank == 0:
        note("%s, %d utterances per GPU x %d GPU(s), T=%d: warmup" % (args.config, b_local, world, t_frames))
    seen_world = None
    persist_at_start = bool(hb.USE_PERSIST)           # (False: several ranks share this card - parallel.init_distributed)
    hb.PERSIST_RETRY_STEPS = 0                        # a timed region is ONE path: no return to the persistent kernels after an
                                                      # abort inside this process (the product's default is a 200-step probation)
    if world > 1:
        # communicator set-up (lazy in the first collective) and rank alignment before the first step; what every rank then
        # believes the job to be goes into the record (config.per_rank)
        dist.all_reduce(torch.zeros(1, device=dev))
        fence()
        mine = torch.tensor([float(dist.get_world_size()), float(dist.get_rank()), float(torch.cuda.current_device())], device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        seen_world = [[int(v) for v in t.tolist()] for t in allr]
        note("rank %d of %d on cuda:%d (%s), backend %s, exchange: %s, step pipeline: %s"
             % (rank, dist.get_world_size(), torch.cuda.current_device(), torch.cuda.get_device_name(dev), dist.get_backend(),
                "one all-reduce of the flat buffer after the backward pass",
                "host reads a step's record one step late (parallel.DpPipeline)" if int(solver.config.get("pipeline_steps", 1)) > 0
                else "the same path, every record read at once (pipeline_steps: 0)"))
    # Rehearsal of the product's abort handling (Solver._recover in one process, parallel.DpPipeline's coordinated
    # repeat under data parallelism): ASR_BENCH_INJECT_ABORT=warmup|timed sets the sticky latch from the host once, as an
    # aborting persistent kernel would (on the last rank only).
    inject = os.environ.get("ASR_BENCH_INJECT_ABORT", "")
    for i in range(args.warmup):
        if inject == "warmup" and i == 0 and rank == world - 1:
            hb.persist_scratch(dev)
            hb.persist_abort_flag(dev).fill_(1)
        with contextlib.redirect_stdout(sys.stderr):
            step()
    fence()
    hb.LAUNCHES.clear()
    if rank == 0:
        note("timing %d steps (%s)" % (args.steps, args.arith))
    persist_before = bool(hb.USE_PERSIST)
    if inject == "timed" and rank == world - 1:
        hb.persist_scratch(dev)
        hb.persist_abort_flag(dev).fill_(1)
    with contextlib.redirect_stdout(sys.stderr):
        el = timed(args.steps)
    retimed = False
    if persist_before and not hb.USE_PERSIST:
        # a persistent kernel aborted inside the timed steps and the Solver repeated that step on the per-step kernels
        # (every rank alike): the number above mixes two paths - time again, on the path the run is on now
        note("persistent kernels were left during the timed steps (abort handled by the Solver): timing again on the per-step kernels")
        hb.LAUNCHES.clear()
        with contextlib.redirect_stdout(sys.stderr):
            el = timed(args.steps)
        retimed = True
    final_loss = float(last["loss"])
    paths = {k: v // max(1, args.steps) for k, v in sorted(hb.LAUNCHES.items())}
    # which stage of the fallback ladder this rank is on after the timed steps
    stage = ("persistent kernels" if hb.USE_PERSIST else
             ("per-step kernels from the start (several ranks share this card)" if not persist_at_start else
              "per-step kernels: left the persistent kernels after an abort in the %s steps" % ("timed" if persist_before else "warm-up")))
    note("rank %d after the timed steps: %.2f ms/step (max over ranks), abort latch %s, %s, sequence-operator paths per step %s"
         % (rank, el / args.steps * 1e3, bool(hb.persist_aborted(dev)), stage, paths))
    per_rank = None
    if world > 1:
        mine = torch.tensor([float(hb.persist_aborted(dev)), float(bool(hb.USE_PERSIST)), float(persist_before), float(persist_at_start)], device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [dict(rank=i, world_size_seen=seen_world[i][0], rank_seen=seen_world[i][1], device=seen_world[i][2],
                         abort_latch=bool(t[0].item()), persistent_kernels=bool(t[1].item()),
                         fallback_stage="none" if t[1].item() else (
                             "per-step kernels from the start (several ranks share this card)" if not t[3].item() else
                             "repeated a step on the per-step kernels during the %s steps" % ("timed" if t[2].item() else "warm-up")))
                    for i, t in enumerate(allr)]
    # the gradient all-reduce alone (68.7 MB at cfg-2), outside the timed region
    allreduce_ms = None
    opt = solver.gen_opt
    if world > 1:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.all_reduce(opt.buf.flat_g)
        fence()
        e0.record()
        for _ in range(5):
            dist.all_reduce(opt.buf.flat_g)
        e1.record()
        fence()
        allreduce_ms = e0.elapsed_time(e1) / 5
        ar = torch.tensor([allreduce_ms], device=dev)
        ar_all = [torch.zeros_like(ar) for _ in range(world)]
        dist.all_gather(ar_all, ar)
        for i, t in enumerate(ar_all):
            per_rank[i]["allreduce_ms"] = float(t.item())
        note("rank %d: all-reduce of the flat gradient buffer alone %.3f ms" % (rank, allreduce_ms))
    # the same steps under the other arithmetics (labelled sub-objects; never the headline)
    also = {}
    if not args.no_also:
        for other in ("bf16x3", "f32"):
            if other == args.arith:
                continue
            with hb.arith(other), contextlib.redirect_stdout(sys.stderr):
                step()
                el_o = timed(args.steps)
            also[other] = dict(ms_per_step=el_o / args.steps * 1e3, value=n_global * args.steps / el_o,
                               unit="utterances/sec", arithmetic=ARITH_TEXT[other])
    # N > 1: the bucketed exchange issued from inside the backward pass (opt-in in the product: config key dp_overlap /
    # ASR_DP_OVERLAP=1), as a labelled sub-object next to the default (one collective)
    also_overlapped = None
    if world > 1 and not args.no_also:
        opt.buf.enable_overlap()
        with contextlib.redirect_stdout(sys.stderr):
            step()
            el_v = timed(args.steps)
        also_overlapped = dict(ms_per_step=el_v / args.steps * 1e3, value=n_global * args.steps / el_v, unit="utterances/sec",
                               exchange="%d buckets of >= 16 MB issued from inside the backward pass (fixed order), awaited "
                                        "before the clip" % len(opt.buf.buckets),
                               abort_latch_any_rank=None, persistent_kernels=bool(hb.USE_PERSIST))
        flag = torch.tensor([float(hb.persist_aborted(dev))], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        also_overlapped["abort_latch_any_rank"] = bool(flag.item())
        opt.buf.disable_overlap()

    out = None
    if rank == 0:
        ms = el / args.steps * 1e3
        value = n_global * args.steps / el
        f_train = 3.0 * sum(fwd_flops_per_utt(cfg, l, max(2, int(0.125 * l)) + 1) for l in lens) / len(lens)
        f_train_padded = 3.0 * fwd_flops_per_utt(cfg, info["t_max"], info["olength"])
        out = {
            "metric": "utterances/sec (train step)", "value": value, "unit": "utterances/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None,
            "dtype": {"bf16x6": "f32 (fp32-equivalent bf16x6 products)", "f32": "f32", "bf16x3": "f32 (bf16x3 products: 16-bit significands)"}[args.arith],
            "data": "synthetic", "arithmetic": ARITH_TEXT[args.arith],
            "config": {"workload": "%s: batch %d per GPU, 80x%d synthetic fbank, RAGGED lengths U[0.6T, T] (mean %.0f frames per "
                                   "utterance, longest %d; the fixed-length reading is workloads.cfg2_fixed_T), V=%d, L+1=%d, "
                                   "dropout %.2f, Adam(amsgrad)+clip 5; %s"
                                   % (args.config, b_local, t_frames, float(np.mean(lens)), int(max(lens)), cfg["output_dim"],
                                      info["olength"], cfg["dropout_rate"], spec["name"]),
                       "mean_frames_per_utterance": float(np.mean(lens)),
                       "config": args.config, "global_batch": n_global, "frames": t_frames,
                       "parallelism": "dp%d" % world, "pad_mode": "global-exact", "launch_mode": "eager",
                       "timed_call": "Solver.sup_train_one_iteration (semi-supervised-asr_amd/solver.py): forward, loss, zero_grad, "
                                     "backward, gradient exchange, clip + Adam (device-side check of the abort latch), host read of "
                                     "loss + abort latch %sone step late (pipeline_steps = 1), the last one inside the timed "
                                     "region; the (global) batch is resident in HBM"
                                     % ("(summed over the ranks by the all-reduce) " if world > 1 else ""),
                       "exchange": "one all-reduce of the flat gradient buffer after the backward pass (the Solver's default)"
                                   if world > 1 else "none (one process)",
                       "persistent_kernels": bool(hb.USE_PERSIST), "retimed_after_abort": retimed,
                       "backend": (dist.get_backend() if world > 1 else None),
                       "step_pipeline": "host reads a step's record one step late; the update is predicated on the device on the "
                                        "abort latch%s" % (" summed over the ranks by the all-reduce (parallel.DpPipeline)" if world > 1 else ""),
                       "row_geometry_per_gpu": row_geometry(b_local, cfg["enc_hidden_dim"]),
                       "per_rank": per_rank, "sequence_op_paths_per_step": paths},
            "loss": final_loss, "allreduce_ms": allreduce_ms,
            "allreduce": "one collective over the flat gradient buffer after the backward pass",
            "model_tflops": value * f_train / 1e12,
            "model_tflops_incl_padded_frames": value * f_train_padded / 1e12,
        }
        if also_overlapped is not None:
            out["also_overlapped"] = also_overlapped
        if "bf16x3" in also:
            out["also_split_bf16_x3"] = also["bf16x3"]
        if "f32" in also:
            out["also_f32_mfma"] = also["f32"]
        note("%.1f utt/s, %.2f ms/step" % (value, ms))
    # ---- the other workloads of BASELINE.json through the same Solver methods (N = 1 only; ~2 s each)
    del solver, opt
    torch.cuda.empty_cache()
    if world == 1 and not args.no_workloads and rank == 0:
        out["workloads"] = other_workloads(dev, tmp, args.config, t_frames)
        if args.config == "cfg2" and t_frames == spec["frames"] and args.scaling == "weak" and not args.batch_per_gpu:
            note("workload epoch")
            ep = epoch_workload(dev, tmp)
            ep["vs_resident_batch"] = ep["ms_per_step"] / out["ms_per_step"]
            out["workloads"]["epoch"] = ep
    if rank == 0:
        note("measuring dominant kernel")
        lens_local = [lens[i] for i in parallel.shard_indices(n_global, rank, world)]
        out["roofline"] = kernel_roofline(dev, cfg, b_local, t_frames, info["olength"], lens_local)
        if not args.no_layer_gemms:
            out["encoder_gate_gemm"] = encoder_gate_gemms(dev, cfg, b_local, t_frames, lens_local)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, xs, lens, ys, args.config + " T=%d" % t_frames,
                                               3 if args.config == "cfg1" else 1)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
    if dist.is_available() and dist.is_initialized():      # (also the one-rank group of ASR_FORCE_DP=1)
        dist.destroy_process_group()
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


def epoch_workload(dev, tmp, steps=60, config="cfg2"):
    """The product's own training loop, input pipeline included: Solver.sup_train_one_epoch over a loader of `steps` batches
    whose utterances have exactly the lengths of the headline batch (ragged 0.6T..T, seed 1234; features and labels differ
    from batch to batch), so that its ms/step is comparable with the resident-batch `ms_per_step`.  Every batch is collated
    one step ahead into pinned memory and uploaded on a side stream (feed.DeviceFeed); per-step logging as in the
    product.  One warm-up epoch of 3 batches, then the timed epoch (its first batch's collate + upload is inside)."""
    import contextlib
    import hip_backend as hb
    from dataset import DictDataset
    spec = CONFIGS[config]
    c, B, T = dict(spec["model"]), spec["batch"], spec["frames"]
    sv = make_solver(c, B, T, os.path.join(tmp, "epoch"))
    _, lens, ys = synth.ragged_batch(B, T, c["input_dim"], c["output_dim"], 1234)
    rs = np.random.RandomState(77)
    feats = [[rs.normal(0.0, 1.0, size=(l, c["input_dim"])).astype(np.float32) for l in lens] for _ in range(4)]

    def corpus(n_batches):
        data = {}
        for i in range(n_batches):
            for j, l in enumerate(lens):
                data["b%03du%02d" % (i, j)] = dict(feature=feats[i % 4][j],
                                                    token_ids=rs.randint(3, c["output_dim"], size=(len(ys[j]),)).tolist())
        return DictDataset(data, config=None, sort=False)

    def loop(n_batches, epoch):
        sv.train_lab_dataset = corpus(n_batches)
        sv.train_lab_loader = sv._loader(sv.train_lab_dataset, B, False, False)
        if os.environ.get("ASR_BENCH_GC_FREEZE", "1") != "0":
            sv.settle_host_memory()                # what Solver.__init__ does after loading its corpora
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(sys.stderr):
            mean_loss = sv.sup_train_one_epoch(epoch, 1.0)
            sv.flush()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n_batches * 1e3, mean_loss

    loop(3, 0)
    hb.LAUNCHES.clear()
    ms, mean_loss = loop(steps, 1)
    paths = {k: v // steps for k, v in sorted(hb.LAUNCHES.items())}
    return dict(call="Solver.sup_train_one_epoch", steps=steps, ms_per_step=ms, value=B / ms * 1e3, unit="utterances/sec",
                workload="%s: an epoch of %d batches of %d utterances with the headline batch's lengths, through the product's "
                         "loader -> feed.DeviceFeed (collated one step ahead into pinned memory by a background thread, "
                         "uploaded on a side stream, labels as one packed int64 tensor) -> sup_train_one_iteration, per-step "
                         "logging on" % (spec["name"], steps, B),
                mean_loss=float(mean_loss), sequence_op_paths=paths)


def other_workloads(dev, tmp, main_config, main_frames):
    """cfg-1, cfg-5 (cfg-2 when the headline is another one), the semi-supervised generator iteration of cfg-4, the judge
    step and the validation decode - each through the Solver method the product runs (sup_train_one_iteration,
    gen_train_one_iteration, judge_train_one_iteration, _greedy), 2 warm-up + 5 timed calls, with the sequence-operator paths
    the calls took."""
    import contextlib
    import hip_backend as hb

    def run(fn, flush, warm=2, n=5):
        with contextlib.redirect_stdout(sys.stderr):
            for _ in range(warm):
                fn()
            hb.LAUNCHES.clear()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                r = fn()
            flush()
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        return ms, r, {k: v // n for k, v in sorted(hb.LAUNCHES.items())}

    res = {}
    for name in ("cfg1", "cfg2", "cfg5"):
        spec = CONFIGS[name]
        if name == main_config and main_frames == spec["frames"]:
            continue
        note("workload %s" % name)
        c, B, T = dict(spec["model"]), spec["batch"], spec["frames"]
        sv = make_solver(c, B, T, os.path.join(tmp, name))
        xs, lens, ys = synth.ragged_batch(B, T, c["input_dim"], c["output_dim"], 1234)
        xs_d, ys_d = torch.from_numpy(xs).to(dev), [torch.from_numpy(y).to(dev) for y in ys]
        ms, loss, paths = run(lambda: sv.sup_train_one_iteration(xs_d, lens, ys_d, 1.0), sv.flush)
        res[name] = dict(call="Solver.sup_train_one_iteration", workload="%s, batch %d, 80x%d" % (spec["name"], B, T),
                         ms_per_step=ms, value=B / ms * 1e3, unit="utterances/sec", loss=float(loss), sequence_op_paths=paths)
        del sv, xs_d, ys_d
        torch.cuda.empty_cache()
        if name == "cfg5":
            # its dominant kernel against its roof, as for the headline (B = 8: the chains run on four of the eight XCDs, and
            # the weight-gradient products of the layers above run beside them on the other four - ops._SideStream)
            import ops
            res[name]["roofline"] = kernel_roofline(dev, c, B, T, max(int(y.shape[0]) for y in ys) + 1, lens)
            res[name]["weight_gradients_beside_the_chains"] = dict(
                enabled=bool(ops._SIDE.enabled and hb.idle_xcd_mask(B)), idle_xcd_mask=hb.idle_xcd_mask(B),
                products_on_the_side_stream_so_far=ops._SIDE.launches)
    # BASELINE.json quotes cfg-2 on "80x800" batches and north_star on {200 ... 1600} frames: the headline batch is the ragged
    # reading of that (U[0.6T, T]); here the FIXED-length reading - every utterance exactly T frames, T / 8 labels -
    # and the ragged batches at the other frame counts, all through the same Solver method
    spec = CONFIGS["cfg2"]
    c, B = dict(spec["model"]), spec["batch"]
    sv = make_solver(c, B, spec["frames"], os.path.join(tmp, "sweep"))
    for name, T, fixed in (("cfg2_fixed_T", spec["frames"], True), ("cfg2_T200", 200, False), ("cfg2_T400", 400, False),
                           ("cfg2_T1600", 1600, False)):
        if main_config == "cfg2" and main_frames == T and not fixed:
            continue
        note("workload %s" % name)
        if fixed:
            rs = np.random.RandomState(1234)
            lens = [T] * B
            xs = rs.normal(0, 1, size=(B, T, c["input_dim"])).astype(np.float32)
            ys = [rs.randint(3, c["output_dim"], size=(T // 8,)).astype(np.int64) for _ in range(B)]
        else:
            xs, lens, ys = synth.ragged_batch(B, T, c["input_dim"], c["output_dim"], 1234)
        xs_d, ys_d = torch.from_numpy(xs).to(dev), [torch.from_numpy(y).to(dev) for y in ys]
        ms, loss, paths = run(lambda: sv.sup_train_one_iteration(xs_d, lens, ys_d, 1.0), sv.flush)
        res[name] = dict(call="Solver.sup_train_one_iteration",
                         workload="cfg-2's model, batch %d, 80x%d, %s (mean %.0f frames, %.0f labels per utterance)"
                                  % (B, T, "every utterance exactly T frames" if fixed else "ragged 0.6T..T",
                                     float(np.mean(lens)), float(np.mean([len(y) for y in ys]))),
                         ms_per_step=ms, value=B / ms * 1e3, unit="utterances/sec", loss=float(loss), sequence_op_paths=paths)
        del xs_d, ys_d
    del sv
    torch.cuda.empty_cache()
    # cfg-4's iteration at cfg-2's shape: 32 labeled + 32 unlabeled utterances of T = 800, judge 2 x 640 (config.yaml).
    # Random weights + random labels drift to an all-<EOS> hypothesis within a few steps (mask sum 0, the reference's 0 / 0):
    # a negligible learning rate and an <EOS> bias keep the timed iterations in the regime a trained model is in.
    note("workload ssl / judge / decode")
    spec = CONFIGS["cfg2"]
    c, B, T = dict(spec["model"]), spec["batch"], spec["frames"]
    sv = make_solver(c, B, T, os.path.join(tmp, "ssl"), learning_rate=1e-8, g_learning_rate=1e-8)
    sv.model.decoder.output_layer.bias.data[2] = -10.0
    xs, lens, ys = synth.ragged_batch(B, T, c["input_dim"], c["output_dim"], 1234)
    uxs, ulens, _ = synth.ragged_batch(B, T, c["input_dim"], c["output_dim"], 4321)
    xs_d, ys_d, uxs_d = torch.from_numpy(xs).to(dev), [torch.from_numpy(y).to(dev) for y in ys], torch.from_numpy(uxs).to(dev)
    ms, meta, paths = run(lambda: sv.gen_train_one_iteration(xs_d, lens, ys_d, uxs_d, ulens), sv.flush)
    res["ssl"] = dict(call="Solver.gen_train_one_iteration", workload="cfg-4 iteration: %d labeled + %d unlabeled utterances of 80x%d, "
                      "smooth-embedding free-running decode of %d steps, judge 2x640" % (B, B, T, int(uxs.shape[1] * sv.proportion)),
                      ms_per_step=ms, value=2 * B / ms * 1e3, unit="utterances/sec (labeled + unlabeled)",
                      losses={k: float(v) for k, v in meta.items()}, sequence_op_paths=paths)
    ms, meta, paths = run(lambda: sv.judge_train_one_iteration(ys_d), sv.flush)
    res["judge"] = dict(call="Solver.judge_train_one_iteration", workload="2x640 LM, %d transcripts, %d steps" % (B, max(len(y) for y in ys) + 5),
                        ms_per_step=ms, value=B / ms * 1e3, unit="transcripts/sec", losses={k: float(v) for k, v in meta.items()},
                        sequence_op_paths=paths)
    sv.model.eval()
    ms, _, paths = run(lambda: sv._greedy(xs_d, lens), sv.flush)
    sv.model.train()
    res["decode"] = dict(call="Solver._greedy", workload="validation decode: cfg-2 model, %d utterances of 80x%d, greedy, max_dec_timesteps %d "
                         "(random weights never emit <EOS>: no early stop)" % (B, T, sv.config["max_dec_timesteps"]),
                         ms_per_batch=ms, value=B / ms * 1e3, unit="utterances/sec", sequence_op_paths=paths)
    del sv
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    main()
