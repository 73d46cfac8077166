"""Runs only the persistent LSTM sequence kernels (cfg-2 layer-0 shape, T=800, B=32, H=512, both directions, with
the fused dW_hh) so that a rocprofv3 --pmc pass stays short:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 tools/pmc_probe.py
and again with WRITE_SIZE (the two counters do not fit one pass).  `python3 tools/pmc_probe.py step` runs the
per-time-step fallback kernels instead."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch, numpy as np, hip_backend as hb
mode = sys.argv[1] if len(sys.argv) > 1 else 'persist'
dev = torch.device('cuda'); lib = hb.load()
H, B, T = 512, 32, int(sys.argv[2]) if len(sys.argv) > 2 else 800
g = torch.Generator().manual_seed(3)
gates = (torch.rand(T, B, 2, 4 * H, generator=g) * 0.8 + 0.1).to(dev)
w = (torch.randn(2, H, 4 * H, generator=g) / np.sqrt(H)).to(dev); wf = (torch.randn(2, 4 * H, H, generator=g) / np.sqrt(H)).to(dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
dy = (torch.randn(T, B, 2 * H, generator=g) * 0.01).to(dev); c = torch.randn(T, B, 2 * H, generator=g).to(dev)
y = torch.empty(T, B, 2 * H, device=dev); c2 = torch.empty(T, B, 2 * H, device=dev)
dcarry = torch.zeros(B, 2 * H, device=dev); dw = torch.zeros(2, 4 * H, H, device=dev)
g2 = gates.clone()
if mode == 'persist' and os.environ.get('PMC_ROWS', 'packed') == 'packed':
    # what the train step runs: PACKED rows (the `..., true>` instantiations) with the lengths of the bench's batch
    sys.path.insert(0, ROOT + '/tests/golden')
    import synth
    _, lens_b, _ = synth.ragged_batch(B, T, 80, 34, 1234)
    layout = hb.RowLayout([int(v) for v in lens_b], [2, 2, 2], dev)
    rows = hb.LayerRows(layout, 0)
    R = rows.R
    gates = (torch.rand(R, 2, 4 * H, generator=g) * 0.8 + 0.1).to(dev); g2 = gates.clone()
    dy = (torch.randn(R, 2 * H, generator=g) * 0.01).to(dev); c = torch.randn(R, 2 * H, generator=g).to(dev)
    y = torch.empty(R, 2 * H, device=dev); c2 = torch.empty(R, 2 * H, device=dev)
    xch, ctrl = hb.persist_scratch(dev)
    x_, c_ = hb.c_p(xch.data_ptr()), hb.c_p(ctrl.data_ptr())
    rb, re_, rh = hb.ptr(rows.base), hb.ptr(rows.ext), rows.host_ptr()
    hb.check(lib.asr_lstm_seq_fwd_persist(rows.T, B, B, H, 2, hb.ptr(g2), hb.ptr(wf), hb.ptr(rows.lens), rb, re_, rh, hb.ptr(y), hb.ptr(c2), x_, c_, hb.current_arith(), hb.stream()), 'fwd')
    hb.check(lib.asr_lstm_seq_bwd_persist(rows.T, B, B, H, 2, hb.ptr(gates), hb.ptr(w), hb.ptr(rows.lens), rb, re_, rh, hb.ptr(dy), hb.ptr(c), hb.ptr(y), hb.ptr(dw), None, x_, c_, hb.current_arith(), hb.stream()), 'bwd')
    print('packed rows: R = %d, valid (utterance, frame) pairs = %d' % (R, int(sum(lens_b))))
elif mode == 'persist':
    xch, ctrl = hb.persist_scratch(dev)
    x_, c_ = hb.c_p(xch.data_ptr()), hb.c_p(ctrl.data_ptr())
    hb.check(lib.asr_lstm_seq_fwd_persist(T, B, B, H, 2, hb.ptr(g2), hb.ptr(wf), hb.ptr(lens), None, None, None, hb.ptr(y), hb.ptr(c2), x_, c_, hb.current_arith(), hb.stream()), 'fwd')
    hb.check(lib.asr_lstm_seq_bwd_persist(T, B, B, H, 2, hb.ptr(gates), hb.ptr(w), hb.ptr(lens), None, None, None, hb.ptr(dy), hb.ptr(c), hb.ptr(y), hb.ptr(dw), None, x_, c_, hb.current_arith(), hb.stream()), 'bwd')
else:
    hb.check(lib.asr_lstm_seq_fwd(T, B, B, H, 2, hb.ptr(g2), hb.ptr(wf), hb.ptr(lens), None, None, hb.ptr(y), hb.ptr(c2), None, hb.stream()), 'fwd')
    hb.check(lib.asr_lstm_seq_bwd(T, B, B, H, 2, hb.ptr(gates), hb.ptr(w), hb.ptr(lens), None, None, hb.ptr(dy), hb.ptr(c), hb.ptr(dcarry), None, hb.stream()), 'bwd')
torch.cuda.synchronize()
print('done', 'aborted' if mode == 'persist' and hb.persist_aborted(dev) else '')
