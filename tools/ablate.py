import ctypes, sys, os, torch, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs={v: ctypes.CDLL(os.path.join(ROOT,'scratch','libs','lib_%s.so'%v)) for v in sys.argv[1:]}
dev=torch.device('cuda')
H,B,T=512,32,400
g=torch.Generator().manual_seed(3)
gates0=(torch.rand(T,B,2,4*H,generator=g)*0.8+0.1).to(dev)
w=(torch.randn(2,H,4*H,generator=g)/np.sqrt(H)).to(dev)
wf=(torch.randn(2,4*H,H,generator=g)/np.sqrt(H)).to(dev)
lens=torch.full((B,),T,dtype=torch.int32,device=dev)
dy=torch.randn(T,B,2*H,generator=g).to(dev); c=torch.randn(T,B,2*H,generator=g).to(dev)
y=torch.empty(T,B,2*H,device=dev); c2=torch.empty(T,B,2*H,device=dev)
dcarry=torch.zeros(B,2*H,device=dev)
P=lambda t: ctypes.c_void_p(t.data_ptr())
st=ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn):
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)*1e3/T
for rnd in range(3):
    for v,l in libs.items():
        gates=gates0.clone(); dcarry.zero_()
        tb=timeit(lambda: l.asr_lstm_seq_bwd(T,B,B,H,2,P(gates),P(w),P(lens),P(dy),P(c),P(dcarry),st))
        gates=gates0.clone()
        tf=timeit(lambda: l.asr_lstm_seq_fwd(T,B,B,H,2,P(gates),P(wf),P(lens),P(y),P(c2),st))
        print(rnd, v, 'bwd us/step %.2f  fwd us/step %.2f'%(tb,tf), flush=True)
