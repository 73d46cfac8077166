"""Split-K sweep with COLD operands (see gemm_cold_sweep.py) over finer split factors for the large GEMMs of the cfg-2 step."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch, hip_backend as hb
dev = torch.device('cuda')
flush = torch.empty(256 * 1024 * 1024, device=dev)
shapes = [('l1 dW_ih TN', 1, 0, 4096, 512, 12800), ('l2 dW_ih TN', 1, 0, 4096, 512, 6400), ('l0 dW_proj TN', 1, 0, 512, 2048, 12800),
          ('l1 dW_proj TN', 1, 0, 512, 2048, 6400), ('l1 dX NN', 0, 0, 12800, 512, 4096), ('l2 dX NN', 0, 0, 6400, 512, 4096),
          ('l0 proj NT', 0, 1, 12800, 512, 2048), ('l1 proj NT', 0, 1, 6400, 512, 2048), ('dwcat TN', 1, 0, 2048, 1152, 3232)]
for name, ta, tb, M, N, K in shapes:
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev)
    out = torch.empty(M, N, device=dev); res = []
    for sk in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16):
        ts = []
        for _ in range(5):
            flush.fill_(1.0); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); hb.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), out=out, split_k=sk); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res.append('%d:%.0f' % (sk, sorted(ts)[1] * 1e3))
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    print('%-15s tiles %4d auto %2d | us %s' % (name, tiles, hb.auto_split_k(M, N, K), '  '.join(res)), flush=True)
