"""The layer-0 input projection [M, 80] x [80, 4096] + bias (gemm_bfk_kernel) over M: is the time linear in the rows?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch
import __graft_entry__ as entry
entry.build()
import hip_backend as hb
dev = torch.device('cuda', 0)
flush = torch.empty(128 * 1024 * 1024, device=dev)
w = torch.randn(4096, 80, device=dev); b = torch.randn(4096, device=dev)
for M in (25600, 22016, 21888, 21760, 21856, 20480, 16384, 12800):
    x = torch.randn(M, 80, device=dev); out = torch.empty(M + 1, 4096, device=dev)[:M]
    ts = []
    for cold in (True, False):
        best = 1e9
        for _ in range(5):
            if cold: flush.fill_(1.0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); hb.gemm(x, w, trans_b=True, bias=b, out=out); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3)
        ts.append(best)
    print("M %6d (%3d tiles of 128, M %% 128 = %3d): cold %6.1f us  warm %6.1f us  (%.2f / %.2f TB/s of output)" % (M, (M + 127) // 128, M % 128, ts[0], ts[1], M * 4096 * 4 / ts[0] / 1e6, M * 4096 * 4 / ts[1] / 1e6))
