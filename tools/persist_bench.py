import ctypes, sys, os, glob, torch, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev=torch.device('cuda')
H,B,T=512,32,400
g=torch.Generator().manual_seed(3)
gates0=(torch.randn(T,B,2,4*H,generator=g)*0.5).to(dev)
wf=(torch.randn(2,4*H,H,generator=g)/np.sqrt(H)).to(dev)
lens=torch.full((B,),T,dtype=torch.int32,device=dev)
y=torch.empty(T,B,2*H,device=dev); c=torch.empty(T,B,2*H,device=dev)
y2=torch.empty(T,B,2*H,device=dev); c2=torch.empty(T,B,2*H,device=dev)
xch=torch.zeros(2*8*8*2048,dtype=torch.int64,device=dev); ctrl=torch.zeros(16,dtype=torch.int32,device=dev)
P=lambda t: ctypes.c_void_p(t.data_ptr())
st=ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn):
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); rc=fn(); e1.record(); torch.cuda.synchronize(); assert rc==0, rc; return e0.elapsed_time(e1)*1e3/T
for path in sorted(glob.glob(ROOT+'/scratchlibs/lib_*.so')):
    l=ctypes.CDLL(path)
    for r in range(3):
        ga=gates0.clone(); gb=gates0.clone()
        tp=timeit(lambda: l.asr_lstm_seq_fwd_persist(T,B,B,H,2,P(ga),P(wf),P(lens),P(y),P(c),P(xch),P(ctrl),st))
        ab=int(ctrl[8].item()); err=int(ctrl[9].item())
        te=timeit(lambda: l.asr_lstm_seq_fwd(T,B,B,H,2,P(gb),P(wf),P(lens),P(y2),P(c2),None,st))
        diff=float((y-y2).abs().max())
        print('%-14s persist %.2f us/step (abort %d err %d) | per-step %.2f us/step | max diff %.2e'%(os.path.basename(path),tp,ab,err,te,diff),flush=True)
# ---- backward
w=(torch.randn(2,H,4*H,generator=g)/np.sqrt(H)).to(dev)
gact=(torch.rand(T,B,2,4*H,generator=g)*0.8+0.1).to(dev)
dy=torch.randn(T,B,2*H,generator=g).to(dev); cc=torch.randn(T,B,2*H,generator=g).to(dev)
dcarry=torch.zeros(B,2*H,device=dev)
l=ctypes.CDLL(ROOT+'/semi-supervised-asr_amd/lib/libasr_hip.so')
for r in range(3):
    ga=gact.clone(); gb=gact.clone(); dcarry.zero_()
    tp=timeit(lambda: l.asr_lstm_seq_bwd_persist(T,B,B,H,2,P(ga),P(w),P(lens),P(dy),P(cc),P(xch),P(ctrl),st))
    ab=int(ctrl[8].item()); err=int(ctrl[9].item())
    te=timeit(lambda: l.asr_lstm_seq_bwd(T,B,B,H,2,P(gb),P(w),P(lens),P(dy),P(cc),P(dcarry),None,st))
    diff=float((ga-gb).abs().max()); scale=float(gb.abs().max())
    print('BWD persist %.2f us/step (abort %d err %d) | per-step %.2f us/step | max diff %.2e (scale %.2e)'%(tp,ab,err,te,diff,scale),flush=True)
