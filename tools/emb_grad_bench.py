"""Times asr_embedding_grad_f32 at the cfg-2 shape (3232 rows, E = 128, V = 34, rows strided by 1152 floats) against
index_add_ on a contiguous copy, with uniform and with skewed tokens."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch
import hip_backend as hb
dev = torch.device('cuda')
rows, E, V, KX = 3232, 128, 34, 1152
buf = torch.randn(rows, KX, device=dev); grad = buf[:, KX - E:]
tok = torch.randint(0, V, (rows,), device=dev)
acc = torch.zeros(V, E, device=dev)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
for name, tk in (("uniform tokens", tok), ("a third <EOS> (a padded label matrix)", torch.where(torch.rand(rows, device=dev) < 0.35, torch.full_like(tok, 2), tok))):
    print('%-40s kernel %.1f us | index_add_(contiguous copy) %.1f us' % (
        name, t(lambda: hb.embedding_grad(tk, grad, acc)), t(lambda: acc.index_add_(0, tk, grad.reshape(rows, E)))))
