"""Ping-pong GEMM kernel (gemm_bfp_kernel): float64 check on a few shapes (edges, K tail, split-K, accumulate, epilogue),
then cold-operand timing of the large GEMM shapes of the cfg-2 step under every kernel choice, for the shipped library and
any variant under scratchlibs/ (tools/mkvar.sh): python3 tools/gemm_pp_bench.py [bf16x6] [modes...]"""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch, hip_backend as hb
dev = torch.device('cuda')
base = sys.argv[1] if len(sys.argv) > 1 else 'bf16x6'
modes = sys.argv[2:] or ['', '+narrow', '+wide', '+sp']

if not os.environ.get('PP_NOCHECK'):
    bad = 0
    for (ta, tb, M, N, K) in [(0, 1, 256, 128, 32), (0, 1, 512, 256, 96), (0, 0, 300, 200, 1030), (1, 0, 257, 130, 70), (1, 1, 384, 128, 64),
                              (0, 1, 1000, 200, 512), (1, 0, 4096, 80, 3200), (0, 0, 3232, 1152, 2048), (0, 1, 128, 128, 512), (0, 1, 129, 64, 80), (0, 1, 2560, 512, 80), (1, 0, 512, 80, 2560), (0, 0, 300, 256, 72), (1, 1, 260, 128, 100)]:
        g = torch.Generator().manual_seed(M * 5 + N * 3 + K)
        A = torch.randn((K, M) if ta else (M, K), generator=g); B = torch.randn((N, K) if tb else (K, N), generator=g)
        bias = torch.randn(N, generator=g); acc0 = torch.randn(M, N, generator=g)
        ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
        sc = float(ref.abs().max())
        for what, kw, want in (('plain', {}, ref), ('unsplit', dict(split_k=1), ref), ('split3', dict(split_k=3), ref),
                               ('bias+relu', dict(bias=bias.to(dev), relu=True), torch.relu(ref + bias)),
                               ('accumulate', dict(out=acc0.clone().to(dev), accumulate=True), ref + acc0)):
            out = hb.gemm(A.to(dev), B.to(dev), trans_a=bool(ta), trans_b=bool(tb), arith=base + '+sp', **kw)
            e = float((out.double().cpu() - want).abs().max()) / sc
            if not e < 2e-6:
                bad += 1
                print('MISMATCH', ta, tb, M, N, K, what, e)
    print('sp float64 check:', 'ok' if not bad else '%d mismatches' % bad)

SH = [('NT', 12800, 4096, 512), ('NN', 12800, 512, 4096), ('TN', 4096, 512, 12800), ('NT', 12800, 512, 2048), ('TN', 512, 2048, 12800),
      ('NN', 12800, 2048, 512), ('NT', 6400, 4096, 512), ('NN', 6400, 512, 4096), ('TN', 4096, 512, 6400), ('NT', 25600, 4096, 80), ('TN', 4096, 80, 25600)]
paths = [hb.LIB_PATH] + sorted(glob.glob(ROOT + '/scratchlibs/lib_*.so'))
libs = {p: ctypes.CDLL(p) for p in paths}
flush = torch.empty(256 * 1024 * 1024, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
res = {}
for (lay, M, N, K) in SH:
    ta, tb = lay[0] == 'T', lay[1] == 'T'
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    for rep in range(3):
        for p in paths:
            for m in modes:
                ar = hb._arith_code(base + m)
                flush.fill_(1.0); torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = libs[p].asr_gemm_f32(int(ta), int(tb), ctypes.c_int64(M), ctypes.c_int64(N), ctypes.c_int64(K), P(A), ctypes.c_int64(A.shape[1]), P(B),
                                          ctypes.c_int64(B.shape[1]), P(C), ctypes.c_int64(N), None, 0, 0, 1, ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0), 0, ar, hb.stream())
                e1.record(); torch.cuda.synchronize(); assert rc == 0, rc
                if rep:
                    k = (os.path.basename(p), m, (lay, M, N, K))
                    res[k] = min(res.get(k, 1e9), e0.elapsed_time(e1) * 1e3)
for p in paths:
    for m in modes:
        row = [(k[2], v) for k, v in res.items() if k[0] == os.path.basename(p) and k[1] == m]
        print('%-22s %-8s' % (os.path.basename(p), m or 'policy'), ' '.join('%s%dx%dx%d %.0f(%.0fTF)' % (s[0], s[1], s[2], s[3], v, 2e-6 * s[1] * s[2] * s[3] / v) for s, v in row),
              '| sum %.0f us' % sum(v for _, v in row))
