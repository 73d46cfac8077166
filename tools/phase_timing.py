import sys, time, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, ROOT+'/semi-supervised-asr_amd', ROOT+'/tests/golden']
import numpy as np, torch
import bench, synth
import model as M, parallel
from parallel import FlatAdam
def log(*a):
    print(*a, flush=True)
dev=torch.device('cuda')
Bn=int(sys.argv[1]) if len(sys.argv)>1 else 32
T=int(sys.argv[2]) if len(sys.argv)>2 else 800
cfg=dict(bench.CFG2, dropout_rate=0.0)
net=M.E2E(labeldist=synth.labeldist(34,5), **cfg)
net.load_state_dict({k: torch.from_numpy(v) for k,v in synth.e2e_weights(cfg,99).items()})
net=net.to(dev).train()
log('model built')
xs,lens,ys=bench.global_batch(Bn,T,1234)
xs_d=torch.from_numpy(xs).to(dev); ys_d=[torch.from_numpy(y).to(dev) for y in ys]
opt=FlatAdam(net, lr=5e-4, weight_decay=1e-6, amsgrad=True, max_grad_norm=5.0)
def sync(): torch.cuda.synchronize()
for it in range(3):
    t0=time.time()
    enc_h, enc_lens = net.encoder(xs_d, lens); sync(); t1=time.time(); log('enc fwd', t1-t0)
    out = net.decoder(enc_h, enc_lens, ys_d); sync(); t2=time.time(); log('dec fwd', t2-t1)
    loss=-out[1].mean()
    opt.zero_grad(); loss.backward(); sync(); t3=time.time(); log('bwd', t3-t2)
    opt.step(); sync(); t4=time.time(); log('opt', t4-t3, 'loss', float(loss))
