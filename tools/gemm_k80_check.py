"""gemm_bfk_kernel (K = 80, weights stationary): float64 check over edges / epilogues, then cold timing against the 128 x 128 kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch, hip_backend as hb
dev = torch.device('cuda')
bad = 0
for arith in ('bf16x6', 'bf16x3'):
    for (M, N) in [(1024, 128), (1100, 200), (25600, 4096), (3000, 4000), (2048, 130), (1025, 257)]:
        g = torch.Generator().manual_seed(M + N)
        A = torch.randn(M, 80, generator=g); B = torch.randn(N, 80, generator=g); bias = torch.randn(N, generator=g); acc0 = torch.randn(M, N, generator=g)
        ref = A.double() @ B.double().t()
        sc = float(ref.abs().max())
        for what, kw, want in (('plain', {}, ref), ('bias+relu', dict(bias=bias.to(dev), relu=True), torch.relu(ref + bias)),
                               ('accumulate', dict(out=acc0.clone().to(dev), accumulate=True), ref + acc0)):
            out = hb.gemm(A.to(dev), B.to(dev), trans_b=True, arith=arith, **kw)
            e = float((out.double().cpu() - want).abs().max()) / sc
            if not e < (3e-6 if arith == 'bf16x6' else 3e-4):
                bad += 1; print('MISMATCH', arith, M, N, what, e)
    # strided views
    Aw = torch.randn(2000, 96, device=dev); Bw = torch.randn(300, 88, device=dev); outw = torch.zeros(2000, 320, device=dev)
    hb.gemm(Aw[:, 8:88], Bw[:, 4:84], trans_b=True, out=outw[:, 8:308], arith=arith)
    ref = Aw[:, 8:88].double().cpu() @ Bw[:, 4:84].double().cpu().t()
    e = float((outw[:, 8:308].double().cpu() - ref).abs().max()) / float(ref.abs().max())
    if not e < 3e-4 or float(outw[:, :8].abs().max()) != 0 or float(outw[:, 308:].abs().max()) != 0:
        bad += 1; print('MISMATCH strided', arith, e)
print('k80 check:', 'ok' if not bad else '%d mismatches' % bad)
M, N, K = 25600, 4096, 80
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev); bias = torch.randn(N, device=dev)
flush = torch.empty(256 * 1024 * 1024, device=dev)
for mode in ('bf16x6', 'bf16x6+narrow', 'bf16x6+sp'):
    ts = []
    for rep in range(4):
        flush.fill_(1.0); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); hb.gemm(A, B, trans_b=True, bias=bias, out=C, arith=mode); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print('%-16s NT 25600x4096x80 + bias: %.0f us (%.0f TF, %.2f TB/s of output)' % (mode, min(ts[1:]), 2e-6 * M * N * K / min(ts[1:]), M * N * 4 / min(ts[1:]) / 1e6))
