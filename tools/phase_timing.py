"""Per-phase wall time of one cfg-2 train step, and host-enqueue time vs GPU-complete time."""
import sys, time, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, ROOT+'/semi-supervised-asr_amd', ROOT+'/tests/golden']
import numpy as np, torch
import bench, synth
import model as M, parallel
from parallel import FlatAdam
def log(*a): print(*a, flush=True)
dev=torch.device('cuda')
Bn=int(sys.argv[1]) if len(sys.argv)>1 else 32
T=int(sys.argv[2]) if len(sys.argv)>2 else 800
drop=float(sys.argv[3]) if len(sys.argv)>3 else 0.3
cfg=dict(bench.CFG2, dropout_rate=drop)
net=M.E2E(labeldist=synth.labeldist(34,5), **cfg)
net.load_state_dict({k: torch.from_numpy(v) for k,v in synth.e2e_weights(cfg,99).items()})
net=net.to(dev).train()
xs,lens,ys=bench.global_batch(Bn,T,1234)
xs_d=torch.from_numpy(xs).to(dev); ys_d=[torch.from_numpy(y).to(dev) for y in ys]
opt=FlatAdam(net, lr=5e-4, weight_decay=1e-6, amsgrad=True, max_grad_norm=5.0)
def sync(): torch.cuda.synchronize()
for it in range(4):
    sync(); t0=time.time()
    enc_h, enc_lens = net.encoder(xs_d, lens); h1=time.time(); sync(); t1=time.time()
    out = net.decoder(enc_h, enc_lens, ys_d); h2=time.time(); sync(); t2=time.time()
    loss=-out[1].mean()
    opt.zero_grad(); loss.backward(); h3=time.time(); sync(); t3=time.time()
    opt.step(); sync(); t4=time.time()
    log('it%d enc fwd %.1f ms (host %.1f) | dec fwd %.1f (host %.1f) | bwd %.1f (host %.1f) | opt %.2f | total %.1f'%(
        it,(t1-t0)*1e3,(h1-t0)*1e3,(t2-t1)*1e3,(h2-t1)*1e3,(t3-t2)*1e3,(h3-t2)*1e3,(t4-t3)*1e3,(t4-t0)*1e3))
# host-only enqueue time of a full step (no intermediate syncs)
for it in range(3):
    sync(); t0=time.time()
    _,lp,_,_=net(xs_d,lens,ys_d); loss=-lp.mean(); opt.zero_grad(); loss.backward(); opt.step()
    h=time.time(); sync(); t1=time.time()
    log('full step: host enqueue %.1f ms, gpu done %.1f ms'%((h-t0)*1e3,(t1-t0)*1e3))
import hip_backend as hb
log('graph stats', hb.graph_stats())
