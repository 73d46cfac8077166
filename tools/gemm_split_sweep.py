"""Sweep split_k for the train step's no-epilogue GEMM shapes (cfg-2)."""
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, ROOT+'/semi-supervised-asr_amd']
import torch, hip_backend as hb
dev=torch.device('cuda')
shapes=[('l1 dX NN',0,0,12800,512,4096),('l2 dX NN',0,0,6400,512,4096),('l0 dproj dX NN',0,0,12800,2048,512),('l1 dproj dX',0,0,6400,2048,512),
 ('l2 dproj dX',0,0,3200,2048,512),
 ('l1 dW_ih TN',1,0,4096,512,12800),('l2 dW_ih TN',1,0,4096,512,6400),('l0 dW_ih TN',1,0,4096,80,25600),('l0 dW_proj TN',1,0,512,2048,12800),
 ('l1 dW_proj TN',1,0,512,2048,6400),('l2 dW_proj',1,0,512,2048,3200)]
def run(name,ta,tb,M,N,K):
    A=torch.randn((K,M) if ta else (M,K),device=dev); B=torch.randn((N,K) if tb else (K,N),device=dev)
    out=torch.empty(M,N,device=dev); res=[]
    for sk in (1,2,3,4,6,8,16):
        if sk > K//64: continue
        for _ in range(2): hb.gemm(A,B,trans_a=bool(ta),trans_b=bool(tb),out=out,split_k=sk)
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        n=5; e0.record()
        for _ in range(n): hb.gemm(A,B,trans_a=bool(ta),trans_b=bool(tb),out=out,split_k=sk)
        e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/n
        res.append('%d:%.0fus/%.0fTF'%(sk,ms*1e3,2.0*M*N*K/ms/1e9))
    print('%-16s M%6d N%5d K%6d auto %2d | %s'%(name,M,N,K,hb.auto_split_k(M,N,K),'  '.join(res)),flush=True)
for s in shapes: run(*s)
