"""Measurement: the small products of a cfg-2 step (decoder-side projections, output layer, their gradients, the top
encoder layer) on 128 x 128 tiles against 64 x 64 tiles (asr_gemm_f32: ASR_GEMM_TILE_SMALL), cold operands, by K split."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch
import __graft_entry__ as entry
entry.build()
import hip_backend as hb
dev = torch.device('cuda')
flush = torch.empty(256 * 1024 * 1024, device=dev)
# (ta, tb, M, N, K, bias, relu)
shapes = [(0, 1, 3200, 512, 512, 1, 0), (0, 1, 3200, 512, 512, 0, 0), (0, 0, 3200, 512, 512, 0, 0), (1, 0, 512, 512, 3200, 0, 0),
          (1, 0, 512, 512, 3232, 0, 0), (1, 0, 34, 1024, 3232, 0, 0), (0, 1, 3232, 34, 1024, 1, 0), (0, 0, 3232, 1024, 34, 0, 0),
          (1, 0, 512, 2048, 3200, 0, 0), (0, 0, 3200, 2048, 512, 0, 0), (0, 1, 3200, 512, 2048, 1, 1), (1, 0, 512, 2048, 6400, 0, 0),
          (0, 0, 6400, 2048, 512, 0, 0), (0, 1, 6400, 512, 2048, 1, 1), (1, 0, 2048, 1152, 3232, 0, 0),
          (0, 1, 800, 128, 80, 1, 0), (0, 1, 800, 512, 256, 1, 1), (1, 0, 512, 128, 800, 0, 0)]
tot = {}
for (ta, tb, M, N, K, bias, relu) in shapes:
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev)
    bv = torch.randn(N, device=dev) if bias else None
    out = torch.zeros(M, N, device=dev)
    line = []
    for mode, sk in (('bf16x6+narrow', 0), ('bf16x6', 0), ('bf16x6+small', 0), ('bf16x6+small', 1), ('bf16x6+small', 2), ('bf16x6+small', 4), ('bf16x6+small', 8), ('bf16x6+small', 16)):
        ts = []
        for _ in range(4):
            flush.fill_(1.0); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); hb.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), bias=bv, relu=bool(relu), out=out, split_k=sk, arith=mode); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        t = sorted(ts)[1]
        tot[(mode, sk)] = tot.get((mode, sk), 0.0) + t
        line.append('%s/%d %4.0f' % (mode.replace('bf16x6', '').replace('+', '') or 'policy', sk, t))
    print('%s%s %5d x %4d x %5d %s%s: ' % ('T' if ta else 'N', 'T' if tb else 'N', M, N, K, 'b' if bias else '-', 'r' if relu else '-') + ' | '.join(line), flush=True)
print('sums:', {('%s/%d' % k): round(v) for k, v in tot.items()})
