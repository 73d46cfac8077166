"""ctypes binding of libasr_hip.so (C ABI declared in include/asr_hip.h).

There is NO CPU fallback: every wrapper needs device ("cuda" == HIP on ROCm)
tensors and raises if the library is absent or a tensor is on the host.
PyTorch is only the allocator / stream provider here.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libasr_hip.so")
ABI_VERSION = 1

EXPORTS = (
    "asr_abi_version", "asr_gemm_f32", "asr_gemm_skinny_f32", "asr_colsum_f32",
    "asr_lstm_seq_fwd", "asr_lstm_seq_bwd", "asr_pyramid_concat_fwd", "asr_pyramid_concat_bwd",
    "asr_dec_step_fwd", "asr_dec_seq_fwd", "asr_dec_step_bwd", "asr_dec_seq_bwd",
    "asr_adam_clip_f32", "asr_sumsq_f32",
)

_lib = None

c_i, c_i64, c_f, c_p = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


class DecFwd(ctypes.Structure):
    """asr_dec_fwd_t"""
    _fields_ = [(n, c_i) for n in ("B", "Tp", "A", "D", "O", "E", "C", "K", "L")] + [("scaling", c_f)] + \
               [(n, c_p) for n in ("P", "Q", "bo", "wcat", "bcat", "wdec", "convw", "watt", "gvec", "w0", "xmask",
                                   "X", "Xd", "gates", "cstate", "Dproj", "fconv", "S", "energy", "ws")]


class DecBwd(ctypes.Structure):
    """asr_dec_bwd_t"""
    _fields_ = [("f", DecFwd)] + \
               [(n, c_p) for n in ("wcatT", "wdecT", "dws", "G", "dwext", "dwraw", "dfpart", "dP", "dgates", "dD",
                                   "dcell", "dgvec_part", "dwatt_part", "dconv_part")]


def load():
    """Load the shared library once; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libasr_hip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). This package has no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name in EXPORTS:
        getattr(lib, name).restype = c_i
    lib.asr_gemm_f32.argtypes = [c_i, c_i, c_i64, c_i64, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_i, c_i,
                                 c_i, c_i64, c_i64, c_i64, c_i, c_p]
    lib.asr_gemm_skinny_f32.argtypes = [c_i64, c_i64, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_i, c_p,
                                        c_i64, c_i64, c_p]
    lib.asr_colsum_f32.argtypes = [c_i64, c_i64, c_p, c_i64, c_p, c_i, c_p]
    lib.asr_lstm_seq_fwd.argtypes = [c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]
    lib.asr_lstm_seq_bwd.argtypes = [c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]
    lib.asr_pyramid_concat_fwd.argtypes = [c_i, c_i, c_i, c_p, c_p, c_p, c_p]
    lib.asr_pyramid_concat_bwd.argtypes = [c_i, c_i, c_i, c_p, c_p, c_p, c_p]
    lib.asr_dec_step_fwd.argtypes = [ctypes.POINTER(DecFwd), c_i, c_p]
    lib.asr_dec_seq_fwd.argtypes = [ctypes.POINTER(DecFwd), c_i, c_i, c_p]
    lib.asr_dec_step_bwd.argtypes = [ctypes.POINTER(DecBwd), c_i, c_p]
    lib.asr_dec_seq_bwd.argtypes = [ctypes.POINTER(DecBwd), c_i, c_i, c_p]
    lib.asr_adam_clip_f32.argtypes = [c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_f, c_f, c_f, c_f, c_f,
                                      c_f, c_p]
    lib.asr_sumsq_f32.argtypes = [c_i64, c_p, c_p, c_p]
    if lib.asr_abi_version() != ABI_VERSION:
        raise RuntimeError("libasr_hip.so ABI %d != expected %d" % (lib.asr_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def _dev(t, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU: the HIP path has no CPU fallback" % name)
    if t.dtype not in (torch.float32, torch.int32):
        raise RuntimeError("%s must be float32/int32, got %s" % (name, t.dtype))
    return t


def ptr(t):
    return None if t is None else c_p(_dev(t).data_ptr())


def stream():
    return c_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with code %d (negative: ASR_E_*; positive: hipError_t)" % (what, rc))


def _rowmajor(t):
    """2-D view with unit column stride -> (tensor, leading dimension)."""
    assert t.dim() == 2 and (t.stride(1) == 1 or t.shape[1] == 1), "need row-major 2-D view"
    return t, (t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1))


def auto_split_k(M, N, K, batch=1):
    tiles = ((M + 127) // 128) * ((N + 127) // 128) * batch
    if tiles >= 192 or K < 1024:
        return 1
    return int(max(1, min(32, 512 // tiles, K // 256)))


def gemm(A, B, trans_a=False, trans_b=False, bias=None, relu=False, out=None, accumulate=False, split_k=None):
    """out[M,N] = op(A) op(B) (+bias)(relu)(+out).  A, B, out are 2-D row-major views (row stride free)."""
    A, lda = _rowmajor(_dev(A, "A"))
    B, ldb = _rowmajor(_dev(B, "B"))
    M, K = (A.shape[1], A.shape[0]) if trans_a else A.shape
    K2, N = (B.shape[1], B.shape[0]) if trans_b else B.shape
    assert K == K2, "inner dimensions differ: %d vs %d" % (K, K2)
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    out, ldc = _rowmajor(out)
    assert out.shape == (M, N)
    if split_k is None:
        split_k = 1 if (bias is not None or relu) else auto_split_k(M, N, K)
    check(load().asr_gemm_f32(int(trans_a), int(trans_b), M, N, K, ptr(A), lda, ptr(B), ldb, ptr(out), ldc,
                              ptr(bias), int(relu), int(accumulate), 1, 0, 0, 0, split_k, stream()), "asr_gemm_f32")
    return out


def gemm_batched(A, B, out, trans_a, trans_b, M, N, K, lda, ldb, ldc, batch, sA, sB, sC, accumulate=False):
    """Raw batched form (pointer + strides); tensors only provide the base pointers."""
    check(load().asr_gemm_f32(int(trans_a), int(trans_b), M, N, K, ptr(A), lda, ptr(B), ldb, ptr(out), ldc, None, 0,
                              int(accumulate), batch, sA, sB, sC, 1, stream()), "asr_gemm_f32(batched)")
    return out


def gemm_skinny(A, Bt, bias=None, out=None, accumulate=False):
    """out[M,N] (+)= A[M,K] Bt[N,K]^T (+bias) for small M (the sequential chains)."""
    A, lda = _rowmajor(_dev(A, "A"))
    Bt, ldb = _rowmajor(_dev(Bt, "Bt"))
    M, K = A.shape
    N = Bt.shape[0]
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    out, ldc = _rowmajor(out)
    check(load().asr_gemm_skinny_f32(M, N, K, ptr(A), lda, ptr(Bt), ldb, ptr(out), ldc, ptr(bias), int(accumulate),
                                     None, 0, 0, stream()), "asr_gemm_skinny_f32")
    return out


def colsum(X, out=None, accumulate=False):
    X, ldx = _rowmajor(_dev(X, "X"))
    M, N = X.shape
    if out is None:
        out = torch.empty(N, device=X.device, dtype=torch.float32)
    check(load().asr_colsum_f32(M, N, ptr(X), ldx, ptr(out), int(accumulate), stream()), "asr_colsum_f32")
    return out


def lstm_seq_fwd(gates, w_hh, lens, y, c):
    T, B, ndir, H4 = gates.shape
    check(load().asr_lstm_seq_fwd(T, B, H4 // 4, ndir, ptr(gates), ptr(w_hh), ptr(lens), ptr(y), ptr(c), stream()),
          "asr_lstm_seq_fwd")


def lstm_seq_bwd(gates, w_hhT, lens, dy, c, dcarry):
    T, B, ndir, H4 = gates.shape
    check(load().asr_lstm_seq_bwd(T, B, H4 // 4, ndir, ptr(gates), ptr(w_hhT), ptr(lens), ptr(dy), ptr(c),
                                  ptr(dcarry), stream()), "asr_lstm_seq_bwd")


def pyramid_fwd(x, mask, out):
    T, B, C = x.shape
    check(load().asr_pyramid_concat_fwd(T, B, C, ptr(x), ptr(mask), ptr(out), stream()), "asr_pyramid_concat_fwd")


def pyramid_bwd(dout, mask, din):
    T, B, C = din.shape
    check(load().asr_pyramid_concat_bwd(T, B, C, ptr(dout), ptr(mask), ptr(din), stream()), "asr_pyramid_concat_bwd")
