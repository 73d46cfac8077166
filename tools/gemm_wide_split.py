"""Wide-kernel K split sweep (ASR_GEMM_WIDE_SK, one process per value) on the shapes whose split the cost model decides."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
    import torch, hip_backend as hb
    dev = torch.device('cuda'); flush = torch.empty(256 * 1024 * 1024, device=dev)
    hb.set_split_bf16((hb.set_split_bf16(-1) & 7) | 56)
    shapes = [('l1 dX NN', 0, 0, 12800, 512, 4096), ('l2 dX NN', 0, 0, 6400, 512, 4096), ('l0 proj NT', 0, 1, 12800, 512, 2048),
              ('l1 in-proj NT', 0, 1, 12800, 4096, 512), ('l1 dW_ih TN', 1, 0, 4096, 512, 12800), ('l0 dW_ih TN', 1, 0, 4096, 80, 25600),
              ('l0 dproj dX NN', 0, 0, 12800, 2048, 512), ('dwcat TN', 1, 0, 2048, 1152, 3232)]
    out = []
    for name, ta, tb, M, N, K in shapes:
        A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev); C = torch.empty(M, N, device=dev)
        ts = []
        for _ in range(5):
            flush.fill_(1.0); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); hb.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), out=C, split_k=2); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        out.append('%s %.0f' % (name, sorted(ts)[1]))
    print(' | '.join(out))
else:
    for sk in sys.argv[1:] or ['0', '1', '2', '3', '4', '5', '8', '16']:
        env = dict(os.environ)
        if sk != '0': env['ASR_GEMM_WIDE_SK'] = sk
        r = subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True)
        print('sk %2s (0 = cost model): %s' % (sk, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]), flush=True)
