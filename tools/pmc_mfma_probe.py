"""Workload for the MFMA-utilisation counter pass:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d OUT -o m \
        -- python3 tools/pmc_mfma_probe.py [bf16x6|bf16x3|f32]
Runs, under the given arithmetic: two full cfg-2 train steps (B = 32, T = 800; every kernel of the step shows up under its
own name), then - in this fixed order, which tools/pmc_mfma_summary.py relies on - the encoder gate GEMMs of the three
layers alone: per layer in-proj fwd, dX = dG W_ih (layers 1, 2 only), dW_ih = dG^T X."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd', ROOT + '/tests/golden']
import numpy as np, torch
import synth, hip_backend as hb, model as M, parallel
from parallel import FlatAdam
arith = sys.argv[1] if len(sys.argv) > 1 else 'bf16x6'
hb.ARITH[0] = hb.ARITH_NAMES[arith]
dev = torch.device('cuda')
cfg = dict(synth.CFG2, dropout_rate=0.3)
B, T = 32, 800
net = M.E2E(labeldist=synth.labeldist(cfg['output_dim'], 5), **cfg)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.e2e_weights(cfg, 99).items()})
net = net.to(dev).train()
opt = FlatAdam(net, lr=5e-4, weight_decay=1e-6, amsgrad=True, max_grad_norm=5.0)
xs, lens, ys = synth.ragged_batch(B, T, cfg['input_dim'], cfg['output_dim'], 1234)
xs_d, ys_d = torch.from_numpy(xs).to(dev), [torch.from_numpy(y).to(dev) for y in ys]
for _ in range(2):
    _, lp, _, _ = net(xs_d, lens, ys_d, tf_rate=1.0)
    loss = -lp.mean()
    opt.zero_grad(); loss.backward(); opt.step()
torch.cuda.synchronize()
print('loss', float(loss), 'aborted', hb.persist_aborted(dev))
H, I = cfg['enc_hidden_dim'], cfg['input_dim']
t = T
for layer in range(3):
    idim = I if layer == 0 else H
    Mrows = t * B
    x = torch.randn(Mrows, idim, device=dev); w = torch.randn(8 * H, idim, device=dev) / np.sqrt(idim)
    dG = torch.randn(Mrows, 8 * H, device=dev); gates = torch.empty(Mrows, 8 * H, device=dev); bias = torch.zeros(8 * H, device=dev)
    hb.gemm(x, w, trans_b=True, bias=bias, out=gates)
    if layer > 0:
        hb.gemm(dG, w)
    hb.gemm(dG, x, trans_a=True)
    torch.cuda.synchronize()
    t = (t + 1) // 2
print('done')
