"""Runs only the dominant kernel chain (LSTM backward step kernel, layer-0 shape) so that a rocprofv3 --pmc pass
stays short:  rocprofv3 --pmc FETCH_SIZE -- python3 tools/pmc_probe.py   (and again with WRITE_SIZE)."""
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, ROOT+'/semi-supervised-asr_amd']
import torch, numpy as np, hip_backend as hb
dev=torch.device('cuda'); lib=hb.load()
H,B,T=512,32,int(sys.argv[1]) if len(sys.argv)>1 else 64
g=torch.Generator().manual_seed(3)
gates=(torch.rand(T,B,2,4*H,generator=g)*0.8+0.1).to(dev)
w=(torch.randn(2,H,4*H,generator=g)/np.sqrt(H)).to(dev); wf=(torch.randn(2,4*H,H,generator=g)/np.sqrt(H)).to(dev)
lens=torch.full((B,),T,dtype=torch.int32,device=dev)
dy=torch.randn(T,B,2*H,generator=g).to(dev); c=torch.randn(T,B,2*H,generator=g).to(dev)
y=torch.empty(T,B,2*H,device=dev); c2=torch.empty(T,B,2*H,device=dev)
dcarry=torch.zeros(B,2*H,device=dev)
g2=gates.clone()
hb.check(lib.asr_lstm_seq_fwd(T,B,B,H,2,hb.ptr(g2),hb.ptr(wf),hb.ptr(lens),hb.ptr(y),hb.ptr(c2),None,hb.stream()),'fwd')
hb.check(lib.asr_lstm_seq_bwd(T,B,B,H,2,hb.ptr(gates),hb.ptr(w),hb.ptr(lens),hb.ptr(dy),hb.ptr(c),hb.ptr(dcarry),None,hb.stream()),'bwd')
torch.cuda.synchronize()
print('done')
