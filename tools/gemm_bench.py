"""Time asr_gemm_f32 on the train step's real shapes (cfg-2) and report TFLOP/s vs the 157.3 TF f32-MFMA peak."""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, ROOT+'/semi-supervised-asr_amd']
import torch, hip_backend as hb
dev=torch.device('cuda')
shapes=[ # name, ta, tb, M, N, K, split
 ('l0 in-proj  NT', 0,1, 25600,4096,80,1), ('l1 in-proj  NT',0,1,12800,4096,512,1), ('l2 in-proj  NT',0,1,6400,4096,512,1),
 ('l0 proj     NT',0,1,12800,512,2048,1), ('l1 dX       NN',0,0,12800,512,4096,1), ('l0 dproj dX NN',0,0,12800,2048,512,1),
 ('l1 dW_ih    TN',1,0,4096,512,12800,None), ('l0 dW_ih    TN',1,0,4096,80,25600,None), ('l0 dW_hh    TN',1,0,2048,512,25600,None),
 ('l0 dW_proj  TN',1,0,512,2048,12800,None), ('square 4096 NT',0,1,4096,4096,4096,1), ('square 4096 NN',0,0,4096,4096,4096,1)]
def run(name,ta,tb,M,N,K,split):
    A=torch.randn((K,M) if ta else (M,K),device=dev); B=torch.randn((N,K) if tb else (K,N),device=dev)
    out=torch.empty(M,N,device=dev)
    sk = hb.auto_split_k(M,N,K) if split is None else split
    for _ in range(2): hb.gemm(A,B,trans_a=bool(ta),trans_b=bool(tb),out=out,split_k=sk)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    n=5; e0.record()
    for _ in range(n): hb.gemm(A,B,trans_a=bool(ta),trans_b=bool(tb),out=out,split_k=sk)
    e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/n
    print('%-16s M%6d N%5d K%6d split %2d  %8.1f us  %6.1f TF'%(name,M,N,K,sk,ms*1e3,2.0*M*N*K/ms/1e9),flush=True)
for mode, name in ((24, 'split-bf16 products, 256x128 LDS-DMA kernel where the shape conforms'), (8, 'split-bf16 products, 128x128 register-staged kernel'), (0, 'fp32-input MFMA')):
    hb.set_split_bf16((hb.set_split_bf16(-1) & 7) | mode)
    print('---', name)
    for s in shapes: run(*s)
