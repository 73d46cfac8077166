"""Split-K sweep with COLD operands (a 1 GiB buffer is rewritten before every timed launch, so A/B come from HBM as
they do inside the train step) for the large GEMM shapes of the cfg-2 step."""
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, ROOT+'/semi-supervised-asr_amd']
import ctypes, glob
import torch, hip_backend as hb
dev=torch.device('cuda')
flush=torch.empty(256*1024*1024, device=dev)
shapes=[('l1 dW_ih TN',1,0,4096,512,12800),('l2 dW_ih TN',1,0,4096,512,6400),('l0 dW_ih TN',1,0,4096,80,25600),
 ('l0 dW_proj TN',1,0,512,2048,12800),('l1 dX NN',0,0,12800,512,4096),('l2 dX NN',0,0,6400,512,4096),
 ('l0 dproj dX NN',0,0,12800,2048,512),('l1 in-proj NT',0,1,12800,4096,512),('l2 in-proj NT',0,1,6400,4096,512),('l0 dW_hh-like TN',1,0,2048,512,25600),('l0 proj NT',0,1,12800,512,2048),('dwcat TN',1,0,2048,1152,3232)]
def run(name,ta,tb,M,N,K):
    global LIBS
    A=torch.randn((K,M) if ta else (M,K),device=dev); B=torch.randn((N,K) if tb else (K,N),device=dev)
    out=torch.empty(M,N,device=dev); res=[]
    for lib in LIBS:
     for mode in MODES:
      sk = hb.auto_split_k(M,N,K)
      if True:
        ts=[]
        use_lib(lib)
        hb.set_split_bf16((hb.set_split_bf16(-1) & 7) | mode)
        for _ in range(5):
            flush.fill_(1.0); torch.cuda.synchronize()
            e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
            e0.record(); hb.gemm(A,B,trans_a=bool(ta),trans_b=bool(tb),out=out,split_k=sk); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms=sorted(ts)[1]
        res.append('%s/%d sk%d: %.0f us / %.0f TF'%(os.path.basename(lib),mode,sk,ms*1e3,2.0*M*N*K/ms/1e9))
    print('%-15s M%6d N%5d K%6d auto %2d | us/TF %s'%(name,M,N,K,hb.auto_split_k(M,N,K),'  '.join(res)),flush=True)
_loaded={}
def use_lib(path):
    import hip_backend
    if path not in _loaded:
        hip_backend._lib=None; hip_backend.LIB_PATH=path; _loaded[path]=hip_backend.load()
    hip_backend._lib=_loaded[path]
MODES=[int(x) for x in os.environ.get('MODES','24,8').split(',')]
LIBS=[hb.LIB_PATH]+sorted(glob.glob(ROOT+'/scratchlibs/lib_*.so'))
for s in shapes: run(*s)
