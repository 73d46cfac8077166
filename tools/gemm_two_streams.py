"""Do two GEMMs on two streams share the chip when the first leaves CUs idle?  dX = dG W_ih (NN 12800 x 512 x 4096: 200 workgroups
of the one-wave-per-SIMD kernel on 256 CUs) beside dW_ih = dG^T x (TN 4096 x 512 x 12800: 256 workgroups): one after the other
on one stream against both at once on two streams."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch, hip_backend as hb
dev = torch.device('cuda')
dG = torch.randn(12800, 4096, device=dev); W = torch.randn(4096, 512, device=dev); x = torch.randn(12800, 512, device=dev)
dX = torch.empty(12800, 512, device=dev); dW = torch.empty(4096, 512, device=dev)
side = torch.cuda.Stream()
flush = torch.empty(256 * 1024 * 1024, device=dev)
def seq():
    hb.gemm(dG, W, out=dX); hb.gemm(dG, x, trans_a=True, out=dW)
def par():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        hb.gemm(dG, x, trans_a=True, out=dW)
        ev2 = torch.cuda.Event(); ev2.record()
    hb.gemm(dG, W, out=dX)
    torch.cuda.current_stream().wait_event(ev2)
for name, fn in (('one stream', seq), ('two streams', par), ('one stream', seq), ('two streams', par)):
    ts = []
    for rep in range(4):
        flush.fill_(1.0); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print('%-12s dX + dW: %.0f us' % (name, min(ts[1:])))
