"""Does a GEMM stream overlap with a latency-bound LSTM step chain on another stream?"""
import sys, os, time, ctypes
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, ROOT+'/semi-supervised-asr_amd']
import torch, numpy as np, hip_backend as hb
dev=torch.device('cuda'); lib=hb.load()
H,B,T=512,32,800
g=torch.Generator().manual_seed(3)
gates0=(torch.rand(T,B,2,4*H,generator=g)*0.8+0.1).to(dev); gates=gates0.clone()
wf=(torch.randn(2,4*H,H,generator=g)/np.sqrt(H)).to(dev)
lens=torch.full((B,),T,dtype=torch.int32,device=dev)
y=torch.empty(T,B,2*H,device=dev); c=torch.empty(T,B,2*H,device=dev)
A=torch.randn(4096,12800,device=dev); Bm=torch.randn(12800,512,device=dev); out=torch.empty(4096,512,device=dev)
side=torch.cuda.Stream()
def chain(st): hb.check(lib.asr_lstm_seq_fwd(T,B,B,H,2,hb.ptr(gates),hb.ptr(wf),hb.ptr(lens),None,None,hb.ptr(y),hb.ptr(c),None,ctypes.c_void_p(st.cuda_stream)),'x')
def gemms(n):
    for _ in range(n): hb.gemm(A,Bm,out=out,split_k=4)
def t(fn):
    torch.cuda.synchronize(); t0=time.time(); fn(); torch.cuda.synchronize(); return (time.time()-t0)*1e3
main=torch.cuda.current_stream()
for r in range(3):
    tc=t(lambda: chain(main))
    tg=t(lambda: gemms(8))
    def both():
        side.wait_stream(main)
        with torch.cuda.stream(side): gemms(8)
        chain(main)
        main.wait_stream(side)
    tb=t(both)
    print('chain %.2f ms | 8 gemms %.2f ms | concurrent %.2f ms (sum %.2f)'%(tc,tg,tb,tc+tg),flush=True)
