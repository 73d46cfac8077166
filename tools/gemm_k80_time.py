import ctypes, glob, os, sys
ROOT='/root/repo'
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch, hip_backend as hb
dev = torch.device('cuda')
M, N, K = 25600, 4096, 80
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev); bias = torch.randn(N, device=dev)
flush = torch.empty(256 * 1024 * 1024, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
for p in [hb.LIB_PATH] + sorted(glob.glob(ROOT + '/scratchlibs/lib_*.so')):
    lib = ctypes.CDLL(p); ts = []
    for rep in range(4):
        flush.fill_(1.0); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.asr_gemm_f32(0, 1, ctypes.c_int64(M), ctypes.c_int64(N), ctypes.c_int64(K), P(A), ctypes.c_int64(K), P(B), ctypes.c_int64(K), P(C), ctypes.c_int64(N), P(bias), 0, 0, 1,
                              ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0), 0, 1, hb.stream())
        e1.record(); torch.cuda.synchronize(); assert rc == 0
        ts.append(e0.elapsed_time(e1) * 1e3)
    print('%-20s %.0f us' % (os.path.basename(p), min(ts[1:])))
