"""Cold-operand timing of the big GEMM shapes of the cfg-2 step under one arithmetic, for the shipped library and any
variant under scratchlibs/ (tools/mkvar.sh): python3 tools/gemm_x6_bench.py [bf16x6]"""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/semi-supervised-asr_amd']
import torch, hip_backend as hb
dev = torch.device('cuda')
ar = hb._arith_code(sys.argv[1] if len(sys.argv) > 1 else 'bf16x6')
SH = [('NT', 12800, 4096, 512), ('NN', 12800, 512, 4096), ('TN', 4096, 512, 12800), ('NT', 12800, 512, 2048), ('TN', 512, 2048, 12800),
      ('NN', 12800, 2048, 512), ('TN', 2048, 512, 25568), ('NT', 25600, 4096, 80)]
paths = [hb.LIB_PATH] + sorted(glob.glob(ROOT + '/scratchlibs/lib_*.so'))
libs = {p: ctypes.CDLL(p) for p in paths}
flush = torch.empty(256 * 1024 * 1024, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
res = {p: [] for p in paths}
for (lay, M, N, K) in SH:
    ta, tb = lay[0] == 'T', lay[1] == 'T'
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    for rep in range(3):
        for p in paths:
            flush.fill_(1.0); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = libs[p].asr_gemm_f32(int(ta), int(tb), ctypes.c_int64(M), ctypes.c_int64(N), ctypes.c_int64(K), P(A), ctypes.c_int64(A.shape[1]), P(B),
                                      ctypes.c_int64(B.shape[1]), P(C), ctypes.c_int64(N), None, 0, 0, 1, ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0), 0, ar, hb.stream())
            e1.record(); torch.cuda.synchronize(); assert rc == 0, rc
            if rep: res[p].append(((lay, M, N, K), e0.elapsed_time(e1) * 1e3))
for p in paths:
    best = {}
    for k, t in res[p]: best[k] = min(best.get(k, 1e9), t)
    print('%-24s' % os.path.basename(p), ' '.join('%s%dx%dx%d %.0f' % (k[0], k[1], k[2], k[3], v) for k, v in best.items()), '| sum %.0f us' % sum(best.values()))
