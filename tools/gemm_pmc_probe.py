"""Workload for counter passes over ONE GEMM kernel choice: python3 tools/gemm_pmc_probe.py bf16x6+sp [nt|nn|tn ...]
(rocprofv3 --pmc ... --kernel-trace --output-format csv -d OUT -o g -- python3 tools/gemm_pmc_probe.py MODE SHAPES)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch, hip_backend as hb
dev = torch.device('cuda')
mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16x6'
shapes = {'nt': (0, 1, 12800, 4096, 512), 'tn': (1, 0, 4096, 512, 12800), 'nn': (0, 0, 12800, 512, 4096), 'nt80': (0, 1, 25600, 4096, 80)}
for name in (sys.argv[2:] or ['nt']):
    ta, tb, M, N, K = shapes[name]
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev)
    out = torch.empty(M, N, device=dev)
    for _ in range(3):
        hb.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), out=out, arith=mode, split_k=1)
    torch.cuda.synchronize()
print('done')
