"""Race screen of the 256 x 128 LDS-DMA GEMM kernel (its ring synchronisation is hand-counted vmcnt + raw barriers): the
same product many times, each result compared with a float64 reference; unsplit products must also be bit-identical
from run to run.  Other work (a big copy) is interleaved so that the DMA timing varies."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch, hip_backend as hb
dev = torch.device('cuda')
N_REP = int(sys.argv[1]) if len(sys.argv) > 1 else 40
hb.set_split_bf16((hb.set_split_bf16(-1) & 7) | hb.SPLIT_GEMM | hb.SPLIT_GEMM_WIDE | hb.SPLIT_GEMM_WIDE_ALL)
noise = torch.empty(64 * 1024 * 1024, device=dev)
shapes = [(1, 0, 4096, 512, 12800), (0, 0, 12800, 512, 4096), (0, 1, 12800, 512, 2048), (1, 0, 4096, 80, 6400), (0, 1, 3232, 1152, 2048),
          (1, 1, 1000, 200, 512), (0, 0, 300, 192, 96), (0, 1, 256, 128, 32)]
bad = 0
for ta, tb, M, N, K in shapes:
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g).to(dev); B = torch.randn((N, K) if tb else (K, N), generator=g).to(dev)
    ref = ((A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())).float()
    scale = float(ref.abs().max())
    first = None; worst = 0.0; differ = 0
    zb = torch.zeros(N, device=dev)               # a bias keeps the kernel from choosing a K split of its own (atomics: order varies)
    for r in range(N_REP):
        if r % 3 == 1: noise.fill_(float(r))
        out = hb.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), bias=zb, split_k=1) if r % 2 else hb.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb))
        err = float((out - ref).abs().max()) / scale
        worst = max(worst, err)
        if r % 2:                                   # unsplit: deterministic
            if first is None: first = out.clone()
            elif not torch.equal(first, out): differ += 1
    ok = worst < 3e-5 and differ == 0
    bad += not ok
    print('%s%s M%6d N%5d K%6d: %d runs, worst error %.2e of the scale, %d unsplit runs differ bitwise  %s' % (
        'T' if ta else 'N', 'T' if tb else 'N', M, N, K, N_REP, worst, differ, 'ok' if ok else 'FAIL'), flush=True)
sys.exit(1 if bad else 0)
