"""Time single decoder-step kernels in a dependent chain (L launches) for ablated builds in scratchlibs/."""
import sys, os, ctypes, glob
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, ROOT+'/semi-supervised-asr_amd']
import torch, hip_backend as hb, ops
dev=torch.device('cuda')
B,Tp,A,D,O,E,C,K,L=32,100,512,512,512,128,10,100,101
ws=ops._dec_workspace(B,Tp,A,D,O,E,C,K,L,False,dev,False)
for k,v in ws.items():
    if v is not None: v.normal_(0,0.1)
ws['w0'].fill_(1.0/Tp)
bo=torch.zeros(O,device=dev); wdec=torch.randn(A,D,device=dev)*0.04; watt=torch.randn(A,C,device=dev)*0.3
d=dict(B=B,Tp=Tp,A=A,D=D,O=O,E=E,C=C,K=K,L=L,KX=D+O+E,scaling=2.0,bo=bo,wdec=wdec,watt=watt)
d.update({k: ws[k] for k in ("P","Q","wcat","bcat","convw","gvec","wattT","w0","xmask","X","Xd","gates","cstate","Dproj","fconv","S","energy","ws")})
fs=ops._dec_fwd_struct(d,0,B)
st=ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for path in sorted(glob.glob(ROOT+'/scratchlibs/lib_*.so')):
    l=ctypes.CDLL(path); l.asr_dec_seq_fwd.restype=ctypes.c_int
    l.asr_dec_seq_fwd.argtypes=[ctypes.POINTER(hb.DecFwd),ctypes.c_int,ctypes.c_int,ctypes.c_void_p,ctypes.c_void_p]
    best=1e9
    for r in range(3):
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(); rc=l.asr_dec_seq_fwd(ctypes.byref(fs),0,L,None,st); e1.record(); torch.cuda.synchronize()
        assert rc==0, rc
        best=min(best,e0.elapsed_time(e1)*1e3/L)
    print('%-20s %.2f us/launch'%(os.path.basename(path),best),flush=True)

# ---- backward kernels (libs named lib_b*)
wsb=ops._dec_workspace(B,Tp,A,D,O,E,C,K,L,False,dev,True)
for k,v in wsb.items():
    if v is not None: v.normal_(0,0.1)
wk=dict(wsb); wk['dws']=None
d2=dict(d); d2.update({k: wsb[k] for k in ("P","Q","wcat","bcat","convw","gvec","wattT","w0","xmask","X","Xd","gates","cstate","Dproj","fconv","S","energy","ws")})
d2['ws'].copy_(torch.softmax(torch.randn(L,B,Tp,device=dev),-1))
bs=ops._dec_bwd_struct(d2,wk,0,B)
for path in sorted(glob.glob(ROOT+'/scratchlibs/libb_*.so')):
    l=ctypes.CDLL(path); l.asr_dec_seq_bwd.restype=ctypes.c_int
    l.asr_dec_seq_bwd.argtypes=[ctypes.POINTER(hb.DecBwd),ctypes.c_int,ctypes.c_int,ctypes.c_void_p,ctypes.c_void_p]
    best=1e9
    for r in range(3):
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(); rc=l.asr_dec_seq_bwd(ctypes.byref(bs),0,L,None,st); e1.record(); torch.cuda.synchronize()
        assert rc==0, rc
        best=min(best,e0.elapsed_time(e1)*1e3/L)
    print('%-20s %.2f us/launch'%(os.path.basename(path),best),flush=True)
