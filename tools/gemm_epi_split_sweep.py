"""Measurement: the bias(+ReLU) projections of the encoder (NT, K = 2048 / 512) on the 128 x 128 kernel by K split (cold
operands).  A split product pays a zero pass and a bias/ReLU pass over C; an unsplit one runs 100-400 workgroups of
64 serial K tiles.  SWEEP_SP_ONLY=1 with ASR_GEMM_WIDE_SK=n (read once per process): the 256 x 128 kernel with a forced split."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch
import __graft_entry__ as entry
entry.build()
import hip_backend as hb
dev = torch.device('cuda')
flush = torch.empty(256 * 1024 * 1024, device=dev)
shapes = [(3200, 512, 2048, True, True), (6400, 512, 2048, True, True), (12800, 512, 2048, True, True), (3200, 512, 512, True, False),
          (3232, 34, 1024, True, False), (25600, 512, 2048, True, True)]
for (M, N, K, bias, relu) in shapes:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); bv = torch.randn(N, device=dev) if bias else None
    out = torch.zeros(M, N, device=dev)
    line = []
    modes = (('bf16x6', 0), ('bf16x6+sp', 0)) if os.environ.get('SWEEP_SP_ONLY') else (('bf16x6', 0), ('bf16x6+sp', 0), ('bf16x6+narrow', 1), ('bf16x6+narrow', 2), ('bf16x6+narrow', 3), ('bf16x6+narrow', 4), ('bf16x6+narrow', 5), ('bf16x6+narrow', 8))
    for mode, sk in modes:
        ts = []
        for _ in range(4):
            flush.fill_(1.0); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); hb.gemm(A, B, trans_b=True, bias=bv, relu=relu, out=out, split_k=sk, arith=mode); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        line.append('%s sk%d %5.0f' % (mode.replace('bf16x6', '').replace('+', '') or 'policy', sk, sorted(ts)[1]))
    print('NT %5d x %4d x %4d %s%s: ' % (M, N, K, 'b' if bias else '-', 'r' if relu else '-') + ' | '.join(line), flush=True)
