"""Race screen of the 256 x 128 LDS-DMA GEMM kernels (gemm_bf6w_kernel / gemm_bf3w_kernel: their ring synchronisation is
hand-counted vmcnt + raw barriers): the same product many times, each result compared with a float64 reference; unsplit
products (split_k = 1) must also be bit-identical from run to run.  Other work (a big copy) is interleaved so that the DMA
timing varies.    python3 tools/gemm_wide_stress.py [runs] [bf16x6|bf16x3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch, hip_backend as hb
dev = torch.device('cuda')
N_REP = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ARITH = (sys.argv[2] if len(sys.argv) > 2 else 'bf16x6')
MODE = ARITH + '+wide'
TOL = 1e-5 if ARITH == 'bf16x6' else 3e-5
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
    for r in range(N_REP):
        if r % 3 == 1: noise.fill_(float(r))
        out = hb.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), split_k=1, arith=MODE) if r % 2 else hb.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), arith=MODE)
        err = float((out - ref).abs().max()) / scale
        worst = max(worst, err)
        if r % 2:                                   # unsplit: deterministic
            if first is None: first = out.clone()
            elif not torch.equal(first, out): differ += 1
    ok = worst < TOL and differ == 0
    bad += not ok
    print(ARITH + ' %s%s M%6d N%5d K%6d: %d runs, worst error %.2e of the scale, %d unsplit runs differ bitwise  %s' % (
        'T' if ta else 'N', 'T' if tb else 'N', M, N, K, N_REP, worst, differ, 'ok' if ok else 'FAIL'), flush=True)
sys.exit(1 if bad else 0)
