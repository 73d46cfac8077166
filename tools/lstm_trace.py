"""Per-phase timing of the persistent LSTM kernels from shader-clock stamps (library built with -DASR_LP_TRACE in
scratchlibs/lib_lptrace.so): mean cycles between marks over time steps 8..15 of workgroup (group 0, slice 0)."""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch, numpy as np
import hip_backend as hb
dev = torch.device('cuda')
H, B, T = 512, int(os.environ.get("PB_B", "32")), 64
g = torch.Generator().manual_seed(3)
gates0 = (torch.randn(T, B, 2, 4 * H, generator=g) * 0.5).to(dev)
wf = (torch.randn(2, 4 * H, H, generator=g) / np.sqrt(H)).to(dev)
w = (torch.randn(2, H, 4 * H, generator=g) / np.sqrt(H)).to(dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
y = torch.empty(T, B, 2 * H, device=dev); c = torch.empty(T, B, 2 * H, device=dev)
gact = (torch.rand(T, B, 2, 4 * H, generator=g) * 0.8 + 0.1).to(dev)
dy = (torch.randn(T, B, 2 * H, generator=g) * 0.01).to(dev); cc = torch.randn(T, B, 2 * H, generator=g).to(dev)
yy = torch.tanh(torch.randn(T, B, 2 * H, generator=g)).to(dev); dw = torch.zeros(2, 4 * H, H, device=dev)
db = torch.zeros(2 * 4 * H, device=dev)
xch, ctrl = hb.persist_scratch(dev, trace=True)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = hb.stream()
l = ctypes.CDLL(ROOT + '/scratchlibs/' + (sys.argv[1] if len(sys.argv) > 1 else 'lib_lptrace.so'))
def report(name, n):
    t = ctrl[32:32 + 8 * 16 * 2].cpu().numpy().view(np.int64).reshape(8, 16)[:, :n]
    print(name, 'cycles/step %.0f' % (t[1:, 0] - t[:-1, 0]).mean(), ' deltas:', ' '.join('%d:%.0f' % (i + 1, x) for i, x in enumerate(np.diff(t, axis=1).mean(0))),
          ' tail->next top: %.0f' % (t[1:, 0] - t[:-1, n - 1]).mean())
for _ in range(2):
    ga = gates0.clone(); ctrl.zero_()
    assert l.asr_lstm_seq_fwd_persist(T, B, B, H, 2, P(ga), P(wf), P(lens), None, None, None, P(y), P(c), P(xch), P(ctrl), hb.current_arith(), st) == 0
    torch.cuda.synchronize(); report('fwd', 7)
    t = ctrl[32:32 + 8 * 16 * 2].cpu().numpy().view(np.int64).reshape(8, 16)
    print('   fwd poll split: top->sentinel ok %.0f | sentinel ok->tile gathered %.0f | failed polls %.1f' % ((t[:, 7] - t[:, 0]).mean(), (t[:, 1] - t[:, 7]).mean(), t[:, 8].mean()))
    t7 = ctrl[32 + 256:32 + 256 + 8 * 16 * 2].cpu().numpy().view(np.int64).reshape(8, 16)
    print('   wave 7 relative to wave 0 top of the same step: top %.0f | sentinel ok %.0f | gathered %.0f | MFMA done %.0f | barrier passed %.0f ; wave 0: gathered %.0f | MFMA done %.0f | barrier %.0f | published %.0f' % (
        (t7[:, 0] - t[:, 0]).mean(), (t7[:, 7] - t[:, 0]).mean(), (t7[:, 1] - t[:, 0]).mean(), (t7[:, 2] - t[:, 0]).mean(), (t7[:, 4] - t[:, 0]).mean(),
        (t[:, 1] - t[:, 0]).mean(), (t[:, 2] - t[:, 0]).mean(), (t[:, 4] - t[:, 0]).mean(), (t[:, 6] - t[:, 0]).mean()))
for _ in range(2):
    gb = gact.clone(); ctrl.zero_()
    assert l.asr_lstm_seq_bwd_persist(T, B, B, H, 2, P(gb), P(w), P(lens), None, None, None, P(dy), P(cc), P(yy), P(dw), P(db), P(xch), P(ctrl), hb.current_arith(), st) == 0
    torch.cuda.synchronize(); report('bwd', 6)
    t = ctrl[32:32 + 8 * 16 * 2].cpu().numpy().view(np.int64).reshape(8, 16)
    print('   bwd pointwise split: partial sums %.0f | math+tags %.0f | publish %.0f | bulk store+db %.0f' % ((t[:, 9] - t[:, 4]).mean(), (t[:, 10] - t[:, 9]).mean(), (t[:, 11] - t[:, 10]).mean(), (t[:, 5] - t[:, 11]).mean()))
    t7 = ctrl[32 + 256:32 + 256 + 8 * 16 * 2].cpu().numpy().view(np.int64).reshape(8, 16)
    names = {0: 'top', 1: 'gathered', 2: 'dh MFMA done', 3: 'partials written', 4: 'barrier passed', 9: 'partials summed', 10: 'math+tags', 11: 'published', 5: 'stores done (dW starts)'}
    for wname, tt in (('wave 0', t), ('wave 7', t7)):
        print('   %s, cycles after wave 0 top of the same step: ' % wname + ' | '.join('%s %.0f' % (names[k], (tt[:, k] - t[:, 0]).mean()) for k in (0, 1, 2, 3, 4, 9, 10, 11, 5) if tt[:, k].min() > 0) +
              ' | next top %.0f' % (tt[1:, 0] - t[:-1, 0]).mean())

# exchanged-partials backward (lstm_persist_bwd_rs_kernel): its own marks
if len(sys.argv) > 2 and sys.argv[2] == 'rs':
    names = ['top', 'gathered', 'reduced+prefetch issued', 'barrier A', 'pointwise done', 'barrier B', 'dh MFMA done', 'published', 'dW done']
    for wname, tt in (('wave 0', t), ('wave 7', t7)):
        print('   rs %s, cycles after wave 0 top of the same step: ' % wname + ' | '.join('%s %.0f' % (names[k], (tt[:, k] - t[:, 0]).mean()) for k in range(9)) +
              ' | next top %.0f' % (tt[1:, 0] - t[:-1, 0]).mean())
