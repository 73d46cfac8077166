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
ABI_VERSION = 5

EXPORTS = (
    "asr_abi_version", "asr_persist_scratch_bytes", "asr_gemm_f32", "asr_gemm_skinny_f32", "asr_colsum_f32",
    "asr_lstm_seq_fwd", "asr_lstm_seq_fwd_persist", "asr_lstm_seq_bwd", "asr_lstm_seq_bwd_persist", "asr_lstm_seq_bwd_persist_w", "asr_lstm_bwd_persist_fuses_dw", "asr_pyramid_concat_fwd", "asr_pyramid_concat_bwd",
    "asr_pyramid_concat_fwd_seeded", "asr_pyramid_concat_bwd_seeded", "asr_rows_pack_f32", "asr_rows_unpack_fwd_f32", "asr_rows_unpack_bwd_f32", "asr_dropout_seeded_f32", "asr_relu_dropout_bwd_f32",
    "asr_dropout_mask_f32",
    "asr_dec_step_fwd", "asr_att_step_fwd", "asr_dec_seq_fwd", "asr_dec_seq_fwd_persist", "asr_dec_seq_fwd_persist_fault", "asr_dec_seq_fwd_persist_free", "asr_dec_step_bwd", "asr_dec_seq_bwd", "asr_dec_seq_bwd_persist", "asr_dec_seq_bwd_persist_free",
    "asr_lstm_pack_f32", "asr_lstm_unpack_f32", "asr_lstm_unpack2_f32", "asr_dec_prepare_f32", "asr_cell_pack_f32", "asr_cell_unpack_f32",
    "asr_lstm_pack_multi_f32", "asr_lstm_unpack_multi_f32", "asr_dec_pack_f32", "asr_colsum_parts_f32", "asr_gemm_drop_f32", "asr_gemm_side_f32", "asr_embedding_grad_f32",
    "asr_label_logprob_fwd", "asr_label_logprob_bwd", "asr_dec_feedback_fwd", "asr_dec_feedback_bwd",
    "asr_adam_clip_f32", "asr_sumsq_f32", "asr_gather_sumsq_f32", "asr_graphs_create", "asr_graphs_destroy", "asr_graphs_stats",
)

_lib = None

c_i, c_i64, c_f, c_p = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


PACK_MAX_LAYERS = 4      # ASR_PACK_MAX_LAYERS


class LstmPackJob(ctypes.Structure):
    """asr_lstm_pack_job_t"""
    _fields_ = [("H", c_i), ("I", c_i), ("ndir", c_i), ("w_ih", c_p * 2), ("w_hh", c_p * 2), ("b_ih", c_p * 2),
                ("b_hh", c_p * 2), ("w_ih_cat", c_p), ("w_hh_il", c_p), ("bias", c_p)]


class LstmUnpackJob(ctypes.Structure):
    """asr_lstm_unpack_job_t"""
    _fields_ = [("H", c_i), ("I", c_i), ("ndir", c_i), ("dw_ih_cat", c_p), ("dw_hh_il", c_p), ("db_il", c_p),
                ("dw_ih", c_p * 2), ("dw_hh", c_p * 2), ("db", c_p * 2), ("db2", c_p * 2)]


class DecFeedback(ctypes.Structure):
    """asr_dec_feedback_t"""
    _fields_ = [("mode", c_i), ("V", c_i), ("eos", c_i), ("scaling", c_f), ("w_out", c_p), ("b_out", c_p), ("emb", c_p), ("logits", c_p),
                ("probs", c_p), ("pred", c_p), ("fed", c_p), ("tokens", c_p), ("ld_tokens", ctypes.c_int64), ("teacher", c_p)]


class DecFeedbackBwd(ctypes.Structure):
    """asr_dec_feedback_bwd_t"""
    _fields_ = [("V", c_i), ("scaling", c_f), ("w_out", c_p), ("emb", c_p), ("probs", c_p), ("dlfb", c_p)]


class DecFwd(ctypes.Structure):
    """asr_dec_fwd_t"""
    _fields_ = [(n, c_i) for n in ("B", "nb", "Tp", "A", "D", "O", "E", "C", "K", "L")] + [("scaling", c_f)] + \
               [(n, c_p) for n in ("P", "Q", "bo", "wcat", "bcat", "wdec", "convw", "watt", "wattT", "gvec", "w0", "xmask",
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
    lib.asr_graphs_create.restype = c_p
    lib.asr_graphs_create.argtypes = [c_i]
    lib.asr_graphs_destroy.restype = None
    lib.asr_graphs_destroy.argtypes = [c_p]
    lib.asr_graphs_stats.argtypes = [c_p, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]
    lib.asr_persist_scratch_bytes.argtypes = [ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]
    lib.asr_gemm_f32.argtypes = [c_i, c_i, c_i64, c_i64, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_i, c_i,
                                 c_i, c_i64, c_i64, c_i64, c_i, c_i, c_p]
    lib.asr_gemm_skinny_f32.argtypes = [c_i64, c_i64, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_i, c_p,
                                        c_i64, c_i64, c_p]
    lib.asr_colsum_f32.argtypes = [c_i64, c_i64, c_p, c_i64, c_p, c_i, c_p]
    lib.asr_lstm_seq_fwd.argtypes = [c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]
    lib.asr_lstm_seq_fwd_persist.argtypes = [c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p]
    lib.asr_lstm_seq_bwd_persist.argtypes = [c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                             c_i, c_p]
    lib.asr_lstm_seq_bwd_persist_w.argtypes = lib.asr_lstm_seq_bwd_persist.argtypes
    lib.asr_lstm_bwd_persist_fuses_dw.argtypes = [c_i, c_i]
    lib.asr_lstm_seq_bwd.argtypes = [c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]
    lib.asr_rows_pack_f32.argtypes = [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_p]
    lib.asr_rows_unpack_fwd_f32.argtypes = [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p, ctypes.c_uint64, c_f, c_p, c_p]
    lib.asr_rows_unpack_bwd_f32.argtypes = [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p, ctypes.c_uint64, c_f, c_p, c_p, c_p, c_p]
    lib.asr_pyramid_concat_fwd.argtypes = [c_i, c_i, c_i, c_p, c_p, c_p, c_p]
    lib.asr_pyramid_concat_bwd.argtypes = [c_i, c_i, c_i, c_p, c_p, c_p, c_p]
    c_u64 = ctypes.c_uint64
    lib.asr_pyramid_concat_fwd_seeded.argtypes = [c_i, c_i, c_i, c_p, c_u64, c_f, c_p, c_p]
    lib.asr_pyramid_concat_bwd_seeded.argtypes = [c_i, c_i, c_i, c_p, c_u64, c_f, c_p, c_p]
    lib.asr_dropout_seeded_f32.argtypes = [c_i64, c_p, c_u64, c_f, c_p]
    lib.asr_relu_dropout_bwd_f32.argtypes = [c_i64, c_p, c_p, c_u64, c_f, c_p, c_p]
    lib.asr_dropout_mask_f32.argtypes = [c_i64, c_p, c_u64, c_f, c_p]
    lib.asr_dec_step_fwd.argtypes = [ctypes.POINTER(DecFwd), c_i, c_p]
    lib.asr_att_step_fwd.argtypes = [ctypes.POINTER(DecFwd), c_i, c_p]
    lib.asr_dec_seq_fwd.argtypes = [ctypes.POINTER(DecFwd), c_i, c_i, c_p, c_p]
    lib.asr_dec_seq_fwd_persist.argtypes = [ctypes.POINTER(DecFwd), c_p, c_p, c_p]
    lib.asr_dec_seq_fwd_persist_fault.argtypes = lib.asr_dec_seq_fwd_persist.argtypes
    lib.asr_dec_seq_fwd_persist_free.argtypes = [ctypes.POINTER(DecFwd), ctypes.POINTER(DecFeedback), c_p, c_p, c_p]
    lib.asr_dec_step_bwd.argtypes = [ctypes.POINTER(DecBwd), c_i, c_p]
    lib.asr_dec_seq_bwd.argtypes = [ctypes.POINTER(DecBwd), c_i, c_i, c_p, c_p]
    lib.asr_dec_seq_bwd_persist.argtypes = [ctypes.POINTER(DecBwd), c_p, c_p, c_p, c_p]
    lib.asr_dec_seq_bwd_persist_free.argtypes = [ctypes.POINTER(DecBwd), ctypes.POINTER(DecFeedbackBwd), c_p, c_p, c_p, c_p]
    lib.asr_adam_clip_f32.argtypes = [c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_p, c_p]
    lib.asr_sumsq_f32.argtypes = [c_i64, c_p, c_p, c_p]
    lib.asr_gather_sumsq_f32.argtypes = [c_i, ctypes.POINTER(c_p), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), c_p, c_p, c_p]
    lib.asr_label_logprob_fwd.argtypes = [c_i64, c_i, c_p, c_i64, c_p, c_p, c_f, c_p, c_p, c_f, c_p, c_p]
    lib.asr_label_logprob_bwd.argtypes = [c_i64, c_i, c_p, c_i64, c_p, c_p, c_f, c_p, c_i64, c_f, c_p, c_i64, c_p]
    lib.asr_dec_feedback_fwd.argtypes = [c_i, c_i, c_i, c_i, c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_p, c_i64,
                                         c_p, c_p, c_p, c_p, c_p, c_i64, c_p]
    lib.asr_dec_feedback_bwd.argtypes = [c_i, c_i, c_i, c_i, c_p, c_p, c_i64, c_p, c_p, c_p, c_f, c_p, c_p]
    pp = ctypes.POINTER(c_p)
    lib.asr_lstm_pack_f32.argtypes = [c_i, c_i, c_i, pp, pp, pp, pp, c_p, c_p, c_p, c_p]
    lib.asr_lstm_unpack_f32.argtypes = [c_i, c_i, c_i, c_p, c_p, c_p, pp, pp, pp, c_p]
    lib.asr_lstm_unpack2_f32.argtypes = [c_i, c_i, c_i, c_p, c_p, c_p, pp, pp, pp, pp, c_p]
    lib.asr_dec_prepare_f32.argtypes = [c_i, c_i, c_i, c_i, c_i, c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_p]
    lib.asr_cell_pack_f32.argtypes = [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]
    lib.asr_cell_unpack_f32.argtypes = [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]
    lib.asr_lstm_pack_multi_f32.argtypes = [c_i, ctypes.POINTER(LstmPackJob), c_p]
    lib.asr_lstm_unpack_multi_f32.argtypes = [c_i, ctypes.POINTER(LstmUnpackJob), c_p]
    lib.asr_dec_pack_f32.argtypes = [c_i, c_i, c_i, c_i, c_i] + [c_p] * 12
    lib.asr_colsum_parts_f32.argtypes = [c_i, c_i, pp, ctypes.POINTER(ctypes.c_int32), pp, c_p]
    lib.asr_embedding_grad_f32.argtypes = [c_i64, c_i, c_i, c_p, c_p, c_i64, c_p, c_p]
    lib.asr_gemm_drop_f32.argtypes = [c_i, c_i, c_i64, c_i64, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_i, c_i, c_i,
                                      ctypes.c_uint64, c_f, c_p]
    lib.asr_gemm_side_f32.argtypes = [c_i, c_i, c_i64, c_i64, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_i64, c_i, c_i64, c_i64, c_i64,
                                      c_i, ctypes.c_uint, c_p, c_p]
    if lib.asr_abi_version() != ABI_VERSION:
        raise RuntimeError("libasr_hip.so ABI %d != expected %d" % (lib.asr_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


# Product arithmetic of the MFMA kernels (include/asr_hip.h: ASR_ARITH_*): an explicit argument of every C-ABI call.
# This module only holds the host code's DEFAULT for calls that do not name one: bf16x6 (three-term split, six products:
# fp32-equivalent), overridable with ASR_ARITH=f32|bf16x6|bf16x3 or `with hb.arith("f32"):`.
ARITH_F32, ARITH_BF16X6, ARITH_BF16X3 = 0, 1, 2
GEMM_TILE_NARROW, GEMM_TILE_WIDE, LSTM_BWD_GATHER, GEMM_TILE_SP, GEMM_TILE_SMALL = 0x100, 0x200, 0x400, 0x800, 0x1000
DEBUG_FAULT = 0x10000        # ASR_DEBUG_FAULT: the persistent LSTM forward launch aborts by itself (tests of the abort path)
# tests of the abort path: True routes the next teacher-forced persistent decoder launch to asr_dec_seq_fwd_persist_fault
DEC_FAULT = [False]
ARITH_NAMES = {"f32": ARITH_F32, "bf16x6": ARITH_BF16X6, "bf16x3": ARITH_BF16X3}
ARITH_LABEL = {ARITH_F32: "f32", ARITH_BF16X6: "bf16x6", ARITH_BF16X3: "bf16x3"}


def _arith_code(a):
    if isinstance(a, str):
        code = 0
        for part in a.lower().split("+"):
            code |= {"narrow": GEMM_TILE_NARROW, "wide": GEMM_TILE_WIDE, "gather": LSTM_BWD_GATHER, "sp": GEMM_TILE_SP,
                     "small": GEMM_TILE_SMALL, "fault": DEBUG_FAULT}.get(part, 0) or \
                    ARITH_NAMES[part]
        return code
    return int(a)


ARITH = [_arith_code(os.environ.get("ASR_ARITH", "bf16x6"))]


def current_arith():
    return ARITH[0]


def arith_name(a=None):
    return ARITH_LABEL[(ARITH[0] if a is None else _arith_code(a)) & 0xff]


class arith(object):
    """Context manager: host-side default arithmetic for the calls inside (name, code, or name+flag: "bf16x3+wide")."""

    def __init__(self, a):
        self.code = _arith_code(a)

    def __enter__(self):
        self.old = ARITH[0]
        ARITH[0] = self.code
        return self

    def __exit__(self, *exc):
        ARITH[0] = self.old
        return False


def _dev(t, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU: the HIP path has no CPU fallback" % name)
    if t.dtype not in (torch.float32, torch.int32):
        raise RuntimeError("%s must be float32/int32, got %s" % (name, t.dtype))
    return t


def ptr(t):
    return None if t is None else c_p(_dev(t).data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """The current HIP stream of the current device as the C ABI takes it.  Asked ~85 times per train step (once per launch):
    the raw handle straight from torch's C layer (0.3 us) instead of a torch.cuda.Stream object per call (2.3 us each - 0.2
    ms of a cfg-1 step whose 2.4 ms ARE its host time)."""
    if _raw_stream is not None and _raw_device is not None:
        return c_p(_raw_stream(_raw_device()))
    return c_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with code %d (negative: ASR_E_*; positive: hipError_t)" % (what, rc))


def _rowmajor(t):
    """2-D view with unit column stride -> (tensor, leading dimension)."""
    assert t.dim() == 2 and (t.stride(1) == 1 or t.shape[1] == 1), "need row-major 2-D view"
    return t, (t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1))


C_ZEROED = 0x2000          # ASR_GEMM_C_ZEROED


def gemm(A, B, trans_a=False, trans_b=False, bias=None, relu=False, out=None, accumulate=False, split_k=None, arith=None,
         drop=None, out_zeroed=False):
    """out[M,N] = op(A) op(B) (+bias)(relu)(+out).  A, B, out are 2-D row-major views (row stride free).
    split_k None: the library chooses (asr_gemm_f32 with split_k = 0); 1: unsplit, run-to-run deterministic.
    drop (SeededMask): the seeded dropout mask over out's element index behind the epilogue (asr_gemm_drop_f32).
    out_zeroed: `out` holds zeros (a slice of the step's arena): a product split over K needs no zero pass of its own."""
    A, lda = _rowmajor(_dev(A, "A"))
    B, ldb = _rowmajor(_dev(B, "B"))
    M, K = (A.shape[1], A.shape[0]) if trans_a else A.shape
    K2, N = (B.shape[1], B.shape[0]) if trans_b else B.shape
    assert K == K2, "inner dimensions differ: %d vs %d" % (K, K2)
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    out, ldc = _rowmajor(out)
    assert out.shape == (M, N)
    code = (ARITH[0] if arith is None else _arith_code(arith)) | (C_ZEROED if out_zeroed else 0)
    if drop is not None:
        assert not accumulate and ldc == N
        check(load().asr_gemm_drop_f32(int(trans_a), int(trans_b), M, N, K, ptr(A), lda, ptr(B), ldb, ptr(out), ldc,
                                       ptr(bias), int(relu), 0 if split_k is None else int(split_k), code, drop.seed,
                                       float(drop.p), stream()), "asr_gemm_drop_f32")
        return out
    check(load().asr_gemm_f32(int(trans_a), int(trans_b), M, N, K, ptr(A), lda, ptr(B), ldb, ptr(out), ldc,
                              ptr(bias), int(relu), int(accumulate), 1, 0, 0, 0, 0 if split_k is None else int(split_k),
                              code, stream()), "asr_gemm_f32")
    return out


def gemm_batched(A, B, out, trans_a, trans_b, M, N, K, lda, ldb, ldc, batch, sA, sB, sC, accumulate=False, arith=None,
                 split_k=None, a_off=0, b_off=0):
    """Raw batched form (pointer + strides in elements; strides may be negative); tensors only provide the base pointers
    (a_off / b_off: element offsets of the first operand elements inside A / B)."""
    check(load().asr_gemm_f32(int(trans_a), int(trans_b), M, N, K, _off(A, a_off), lda, _off(B, b_off), ldb, ptr(out), ldc,
                              None, 0, int(accumulate), batch, sA, sB, sC, 0 if split_k is None else int(split_k),
                              ARITH[0] if arith is None else _arith_code(arith), stream()), "asr_gemm_f32(batched)")
    return out


def gemm_side(A, B, out, queue, xcd_mask, trans_a=False, trans_b=False, arith=None):
    """out[M,N] += op(A) op(B) on the XCDs of `xcd_mask` only, by workgroups that fit beside a persistent kernel
    (asr_gemm_side_f32); `queue`: a zeroed int32 / float32 word (tensor), consumed.  False when the arithmetic has no such
    kernel (the fp32-input MFMA): the caller runs gemm() instead."""
    code = ARITH[0] if arith is None else _arith_code(arith)
    if (code & 0xff) == ARITH_F32:
        return False
    A, lda = _rowmajor(_dev(A, "A"))
    B, ldb = _rowmajor(_dev(B, "B"))
    M, K = (A.shape[1], A.shape[0]) if trans_a else A.shape
    K2, N = (B.shape[1], B.shape[0]) if trans_b else B.shape
    assert K == K2 and out.shape == (M, N)
    out, ldc = _rowmajor(out)
    check(load().asr_gemm_side_f32(int(trans_a), int(trans_b), M, N, K, ptr(A), lda, ptr(B), ldb, ptr(out), ldc, 1, 0, 0, 0,
                                   code, int(xcd_mask) & 0xff, c_p(queue.data_ptr()), stream()), "asr_gemm_side_f32")
    return True


def gemm_side_batched(A, B, out, queue, xcd_mask, trans_a, trans_b, M, N, K, lda, ldb, ldc, batch, sA, sB, sC, arith=None,
                      a_off=0, b_off=0):
    """The raw batched form of gemm_side (as gemm_batched)."""
    code = ARITH[0] if arith is None else _arith_code(arith)
    if (code & 0xff) == ARITH_F32:
        return False
    check(load().asr_gemm_side_f32(int(trans_a), int(trans_b), M, N, K, _off(A, a_off), lda, _off(B, b_off), ldb, ptr(out), ldc,
                                   batch, sA, sB, sC, code, int(xcd_mask) & 0xff, c_p(queue.data_ptr()), stream()),
          "asr_gemm_side_f32(batched)")
    return True


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


FEED_PREDICTED, FEED_SMOOTH, FEED_TEACHER, FEED_NONE = 0, 1, 2, 3


def _lptr(t):
    return None if t is None else c_p(t.data_ptr())


def dec_feedback_fwd(x_top, w_out, b_out, emb_w, logits, pred, mode, scaling=1.0, tok=None, fed=None, probs=None,
                     x_emb_next=None, xd_emb_next=None, mask=None):
    """One decoder step's output side (model.py:329-351): logits = x_top w_out^T + b, pred = argmax, and the next
    step's embedding input (teacher token / predicted token / softmax(scaling*logits) @ E), with its dropped-out copy.
    x_top [B, D+O] and the embedding slots are row-strided views of the step input buffer; pred/fed/tok are int64."""
    _dev(x_top, "x_top")
    B, DO = x_top.shape
    V, E = emb_w.shape
    ldx = x_top.stride(0)
    assert x_top.stride(1) == 1 and w_out.is_contiguous() and emb_w.is_contiguous() and logits.is_contiguous()
    tok_stride = 0
    if tok is not None:
        assert tok.dim() == 1 and tok.dtype == torch.long and tok.shape[0] == B
        tok_stride = tok.stride(0)
    for t in (pred, fed):
        assert t is None or (t.dtype == torch.long and t.is_contiguous() and t.numel() == B)
    ldm = 0
    if x_emb_next is not None:
        assert x_emb_next.stride(0) == ldx and x_emb_next.stride(1) == 1 and x_emb_next.shape == (B, E)
    if xd_emb_next is not None:
        assert xd_emb_next.stride(0) == ldx and xd_emb_next.stride(1) == 1 and mask.stride(1) == 1
        ldm = mask.stride(0)
    check(load().asr_dec_feedback_fwd(B, V, E, DO, ptr(x_top), ldx, ptr(w_out), ptr(b_out), ptr(emb_w), ptr(logits),
                                      _lptr(pred), int(mode), float(scaling), _lptr(tok), tok_stride, _lptr(fed),
                                      ptr(probs), ptr(x_emb_next), ptr(xd_emb_next), ptr(mask), ldm, stream()),
          "asr_dec_feedback_fwd")


def dec_feedback_bwd(demb, gtop, probs, emb_w, w_out, scaling, dlog):
    """Backward of the smooth embedding feedback: demb [B,E] (grad of step s's embedding input) -> dlog [B,V] (+=, grad
    of logits_{s-1}) -> gtop [B,D+O] (+=, grad of [z_{s-1}, c_{s-1}])."""
    _dev(demb, "demb")
    B, E = demb.shape
    V, DO = w_out.shape
    ldg = demb.stride(0)
    assert demb.stride(1) == 1 and gtop.stride(1) == 1 and gtop.stride(0) == ldg and gtop.shape == (B, DO)
    assert probs.is_contiguous() and dlog.is_contiguous() and w_out.is_contiguous() and emb_w.is_contiguous()
    check(load().asr_dec_feedback_bwd(B, V, E, DO, ptr(demb), ptr(gtop), ldg, ptr(probs), ptr(emb_w), ptr(w_out),
                                      float(scaling), ptr(dlog), stream()), "asr_dec_feedback_bwd")


def colsum(X, out=None, accumulate=False):
    X, ldx = _rowmajor(_dev(X, "X"))
    M, N = X.shape
    if out is None:
        out = torch.empty(N, device=X.device, dtype=torch.float32)
    check(load().asr_colsum_f32(M, N, ptr(X), ldx, ptr(out), int(accumulate), stream()), "asr_colsum_f32")
    return out


# ---------------------------------------------------------------------------------------------------
# Utterance-group concurrency: the recurrent chains are latency-bound (one small kernel per time step), and
# utterances are independent, so disjoint row groups run on separate HIP streams and overlap each other's
# launch-boundary / memory latency.  Each group's launch loop runs in its own host thread (ctypes drops the
# GIL inside the C call), so the host enqueue rate scales with the number of groups as well.
# Measured on MI355X: graph-replayed kernels run ~0.8 us slower each than eager launches and
# two concurrent row groups disturb each other (per-kernel cache maintenance), so the defaults are one group and
# eager launches; graphs pay off only when the host cannot keep up (bench.py picks the faster mode in warm-up).
GROUPS = int(os.environ.get("ASR_ROW_GROUPS", "1"))
USE_GRAPHS = os.environ.get("ASR_GRAPHS", "0") != "0"
_side_streams = {}
_pool = None
_graph_handles = {}


def graphs_for(group_index):
    """hipGraph memo handle of a row group (one per launching thread; see include/asr_hip.h)."""
    if not USE_GRAPHS:
        return None
    key = (torch.cuda.current_device(), group_index)
    h = _graph_handles.get(key)
    if h is None:
        h = c_p(load().asr_graphs_create(96))
        _graph_handles[key] = h
    return h


def graph_stats():
    out = {}
    for key, h in _graph_handles.items():
        a, b, c = c_i64(0), c_i64(0), c_i64(0)
        load().asr_graphs_stats(h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        out[key] = dict(hits=a.value, captures=b.value, eager=c.value)
    return out


def row_groups(B):
    """Split B rows into <= GROUPS contiguous groups whose sizes are multiples of 16 where possible."""
    g = max(1, min(GROUPS, B // 16))
    if g <= 1:
        return [(0, B)]
    per = ((B + g - 1) // g + 15) // 16 * 16
    out, b0 = [], 0
    while b0 < B:
        out.append((b0, min(per, B - b0)))
        b0 += per
    return out


def run_grouped(groups, fn):
    """fn(group_index, (b0, nb), stream_ptr) for every group.  Every group runs on its own non-default HIP stream
    (stream capture for the graph memo is illegal on the legacy default stream), forked from and joined back
    into the current stream.  With several groups the launch loops run concurrently on a thread pool."""
    global _pool
    if len(groups) == 1 and not USE_GRAPHS:
        fn(0, groups[0], stream())          # plain in-order launches on the current stream (fastest on the GPU side)
        return
    main = torch.cuda.current_stream()
    dev = main.device
    side = _side_streams.setdefault(dev, [])
    while len(side) < len(groups):
        side.append(torch.cuda.Stream(device=dev))
    used = side[: len(groups)]
    for st in used:
        st.wait_stream(main)
    if len(groups) == 1:
        fn(0, groups[0], c_p(used[0].cuda_stream))
    else:
        from concurrent.futures import ThreadPoolExecutor
        if _pool is None:
            _pool = ThreadPoolExecutor(max_workers=8)
        futs = [_pool.submit(fn, gi, groups[gi], c_p(used[gi].cuda_stream)) for gi in range(1, len(groups))]
        fn(0, groups[0], c_p(used[0].cuda_stream))
        for f in futs:
            f.result()
    for st in used:
        main.wait_stream(st)


_pinned_ring = {}
_upload_streams = {}
# The side stream costs the host ~25 us per upload (stream switch, event record / wait): worth it when the GPU is the
# bottleneck (the copy leaves the compute stream), a loss when the host is (cfg-1: 2.01 -> 2.32 ms per step).  E2E.forward
# switches it on for batches of at least UPLOAD_SIDE_MIN_FRAMES padded frames (B * T); ASR_UPLOAD_STREAM=0/1 forces it.
UPLOAD_SIDE_STREAM = [os.environ.get("ASR_UPLOAD_STREAM", "0") == "1"]
UPLOAD_SIDE_MIN_FRAMES = 4096


def upload_side_stream_for(n_frames):
    if "ASR_UPLOAD_STREAM" not in os.environ:
        UPLOAD_SIDE_STREAM[0] = int(n_frames) >= UPLOAD_SIDE_MIN_FRAMES


def _to_device(arr, torch_dtype, device):
    """Host array -> device tensor without blocking the host and without a place in the compute stream: staged through a
    small ring of pinned buffers and copied on a side stream (the compute stream only waits for the copy's event - the
    host runs ahead of the GPU, so the copy has long finished when the stream gets there; as an in-stream copy each of
    these small uploads cost ~10 us of the step)."""
    key = (str(device), str(torch_dtype), arr.size)
    ring = _pinned_ring.get(key)
    if ring is None:            # (not setdefault(key, dict(...)): its default is built - eight pinned allocations - on EVERY call)
        ring = _pinned_ring[key] = dict(bufs=[torch.empty(arr.size, dtype=torch_dtype).pin_memory() for _ in range(8)], i=0)
    buf = ring["bufs"][ring["i"] % 8]
    ring["i"] += 1
    buf.copy_(torch.from_numpy(arr.reshape(-1)))
    dev = torch.device(device)
    if dev.type != "cuda" or not UPLOAD_SIDE_STREAM[0]:
        return buf.to(device, non_blocking=True).view(arr.shape)
    side = _upload_streams.get(str(dev))
    if side is None:
        side = _upload_streams.setdefault(str(dev), torch.cuda.Stream(device=dev))
    main = torch.cuda.current_stream(dev)
    with torch.cuda.stream(side):
        out = buf.to(dev, non_blocking=True)
    main.wait_stream(side)
    out.record_stream(main)
    return out.view(arr.shape)


def to_device_i32(values, device):
    """Host ints -> int32 device tensor (see _to_device)."""
    import numpy as np
    return _to_device(np.asarray(values, dtype=np.int32), torch.int32, device)


def to_device_i64(values, device):
    """Host ints -> int64 device tensor (index tensors: no conversion kernel on the device)."""
    import numpy as np
    return _to_device(np.asarray(values, dtype=np.int64), torch.int64, device)


def to_device_f32(array, device):
    """Host float32 array -> device tensor (see _to_device)."""
    import numpy as np
    return _to_device(np.ascontiguousarray(array, dtype=np.float32), torch.float32, device)


def _ptr_array(tensors):
    return (c_p * len(tensors))(*[_dev(t).data_ptr() for t in tensors])


def lstm_pack(params, ndir, w_ih_cat, w_hh_il, bias):
    """params: per direction (w_ih, w_hh, b_ih, b_hh) in torch layout -> interleaved kernel layout (one launch)."""
    H, I = params[1].shape[1], params[0].shape[1]
    ps = [p if p.is_contiguous() else p.contiguous() for p in params]
    check(load().asr_lstm_pack_f32(H, I, ndir, _ptr_array(ps[0::4]), _ptr_array(ps[1::4]), _ptr_array(ps[2::4]),
                                   _ptr_array(ps[3::4]), ptr(w_ih_cat), ptr(w_hh_il), ptr(bias), stream()),
          "asr_lstm_pack_f32")


def lstm_unpack(H, I, ndir, dw_ih_cat, dw_hh_il, db_il, two_biases=False):
    """Interleaved gradients -> [dw_ih, dw_hh, db] per direction in torch layout (one launch).  two_biases: also a second,
    independent copy of every bias gradient (for b_hh: autograd clones a tensor returned for two parameters)."""
    dev = dw_ih_cat.device
    f32 = dict(device=dev, dtype=torch.float32)
    dw_ih = [torch.empty(4 * H, I, **f32) for _ in range(ndir)]
    dw_hh = [torch.empty(4 * H, H, **f32) for _ in range(ndir)]
    db = [torch.empty(4 * H, **f32) for _ in range(ndir)]
    db2 = [torch.empty(4 * H, **f32) for _ in range(ndir)] if two_biases else None
    check(load().asr_lstm_unpack2_f32(H, I, ndir, ptr(dw_ih_cat), ptr(dw_hh_il), ptr(db_il), _ptr_array(dw_ih),
                                      _ptr_array(dw_hh), _ptr_array(db), _ptr_array(db2) if two_biases else None, stream()),
          "asr_lstm_unpack2_f32")
    return (dw_ih, dw_hh, db, db2) if two_biases else (dw_ih, dw_hh, db)


def lstm_pack_multi(layers, ndir):
    """layers: per layer the list of per-direction (w_ih, w_hh, b_ih, b_hh) in torch layout -> per layer (w_ih_cat
    [ndir*4H, I], w_hh_il [ndir, 4H, H], bias [ndir*4H]) in the kernels' gate-interleaved layout; one launch for up to
    PACK_MAX_LAYERS layers."""
    outs, keep = [], []
    jobs = (LstmPackJob * len(layers))()
    for j, params in enumerate(layers):
        H, I = params[1].shape[1], params[0].shape[1]
        ps = [p if p.is_contiguous() else p.contiguous() for p in params]
        keep.append(ps)
        f32 = dict(device=ps[0].device, dtype=torch.float32)
        out = (torch.empty(ndir * 4 * H, I, **f32), torch.empty(ndir, 4 * H, H, **f32), torch.empty(ndir * 4 * H, **f32))
        outs.append(out)
        job = jobs[j]
        job.H, job.I, job.ndir = H, I, ndir
        for d in range(ndir):
            job.w_ih[d], job.w_hh[d] = _dev(ps[4 * d]).data_ptr(), _dev(ps[4 * d + 1]).data_ptr()
            job.b_ih[d], job.b_hh[d] = _dev(ps[4 * d + 2]).data_ptr(), _dev(ps[4 * d + 3]).data_ptr()
        job.w_ih_cat, job.w_hh_il, job.bias = out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr()
    lib = load()
    for j0 in range(0, len(layers), PACK_MAX_LAYERS):
        n = min(PACK_MAX_LAYERS, len(layers) - j0)
        check(lib.asr_lstm_pack_multi_f32(n, ctypes.cast(ctypes.byref(jobs, j0 * ctypes.sizeof(LstmPackJob)),
                                                         ctypes.POINTER(LstmPackJob)), stream()), "asr_lstm_pack_multi_f32")
    return outs


def lstm_unpack_multi(grads, dims, ndir):
    """grads: per layer (dw_ih_cat, dw_hh_il, db_il) in the interleaved layout (a layer whose three are all None gets no
    gradients), dims: per layer (H, I) -> per layer [per direction dw_ih, dw_hh, db_ih, db_hh] flattened in torch layout
    (b_ih and b_hh get their own tensors); one launch."""
    live = [j for j, g in enumerate(grads) if g[0] is not None]
    outs, keep = [None] * len(grads), []
    jobs = (LstmUnpackJob * max(len(live), 1))()
    for k, j in enumerate(live):
        H, I = dims[j]
        dw_ih, dw_hh, db = [g if g.is_contiguous() else g.contiguous() for g in grads[j]]
        keep.append((dw_ih, dw_hh, db))
        f32 = dict(device=dw_ih.device, dtype=torch.float32)
        o = []
        job = jobs[k]
        job.H, job.I, job.ndir = H, I, ndir
        job.dw_ih_cat, job.dw_hh_il, job.db_il = _dev(dw_ih).data_ptr(), _dev(dw_hh).data_ptr(), _dev(db).data_ptr()
        for d in range(ndir):
            t = [torch.empty(4 * H, I, **f32), torch.empty(4 * H, H, **f32), torch.empty(4 * H, **f32), torch.empty(4 * H, **f32)]
            job.dw_ih[d], job.dw_hh[d], job.db[d], job.db2[d] = [x.data_ptr() for x in t]
            o += t
        outs[j] = o
    lib = load()
    for k0 in range(0, len(live), PACK_MAX_LAYERS):
        n = min(PACK_MAX_LAYERS, len(live) - k0)
        check(lib.asr_lstm_unpack_multi_f32(n, ctypes.cast(ctypes.byref(jobs, k0 * ctypes.sizeof(LstmUnpackJob)),
                                                           ctypes.POINTER(LstmUnpackJob)), stream()), "asr_lstm_unpack_multi_f32")
    return outs


def dec_prepare(tokens, emb_w, xmask, X, Xd, fed, L, B, D, O, E):
    """Teacher-forced decoder input (embedding gather into X / Xd, zero recurrent slots, fed = tokens^T) in one launch.
    tokens [B, >= L] int64 on the device (row stride free)."""
    assert tokens.dtype == torch.long and tokens.is_cuda and tokens.stride(1) == 1 and tokens.shape[1] >= L
    assert X.is_contiguous() and (Xd is None or Xd.is_contiguous()) and fed.is_contiguous() and emb_w.is_contiguous()
    check(load().asr_dec_prepare_f32(L, B, D, O, E, c_p(tokens.data_ptr()), tokens.stride(0), ptr(emb_w),
                                     ptr(xmask) if xmask is not None else None, ptr(X), ptr(Xd) if Xd is not None else None,
                                     c_p(fed.data_ptr()), stream()), "asr_dec_prepare_f32")


def cell_pack(w_ih, w_hh, b_ih, b_hh, D, O, E, wcat, bcat):
    check(load().asr_cell_pack_f32(D, O, E, ptr(w_ih.contiguous()), ptr(w_hh.contiguous()), ptr(b_ih.contiguous()),
                                   ptr(b_hh.contiguous()), ptr(wcat), ptr(bcat), stream()), "asr_cell_pack_f32")


def dec_pack(w_ih, w_hh, b_ih, b_hh, wdec, watt, D, O, E, A, C, wcat, bcat, wcatT=None, wdecT=None, wattT=None):
    """cell_pack + the transposed images wcatT [KX, 4D], wdecT [D, A], wattT [C, A] (each optional) in one launch."""
    check(load().asr_dec_pack_f32(D, O, E, A, C, ptr(w_ih.contiguous()), ptr(w_hh.contiguous()), ptr(b_ih.contiguous()),
                                  ptr(b_hh.contiguous()), ptr(wdec.contiguous()), ptr(watt.contiguous()), ptr(wcat), ptr(bcat),
                                  ptr(wcatT), ptr(wdecT), ptr(wattT), stream()), "asr_dec_pack_f32")


def cell_unpack(dwcat, db_il, D, O, E):
    """-> dw_ih, dw_hh, db_ih, db_hh (the two bias gradients are equal, in tensors of their own)."""
    f32 = dict(device=dwcat.device, dtype=torch.float32)
    dw_ih, dw_hh, db, db2 = (torch.empty(4 * D, E + O, **f32), torch.empty(4 * D, D, **f32), torch.empty(4 * D, **f32),
                             torch.empty(4 * D, **f32))
    check(load().asr_cell_unpack_f32(D, O, E, ptr(dwcat), ptr(db_il), ptr(dw_ih), ptr(dw_hh), ptr(db), ptr(db2), stream()),
          "asr_cell_unpack_f32")
    return dw_ih, dw_hh, db, db2


def embedding_grad(tokens, grad, demb):
    """demb [V, E] += grad[r] for the token of row r (tokens [rows] int64, -1 = none; grad [rows, E] row-strided view).
    False when the layout is outside the kernel's (the caller then uses index_add_)."""
    V, E = demb.shape
    rows = tokens.numel()
    if (grad.stride(1) != 1 or grad.shape != (rows, E) or not tokens.is_contiguous() or not demb.is_contiguous()
            or E % 4 or grad.stride(0) % 4 or V * E * 4 > 65536 or E // 4 > 256 or grad.data_ptr() % 16):
        return False
    check(load().asr_embedding_grad_f32(rows, E, V, c_p(tokens.data_ptr()), ptr(grad), grad.stride(0), ptr(demb), stream()),
          "asr_embedding_grad_f32")
    return True


def gather_sumsq(srcs, offsets, flat, sumsq=None):
    """flat[offsets[j] : + srcs[j].numel()] = srcs[j] (contiguous fp32 device tensors) in one launch; sumsq (1-element
    tensor or None) += the sum of their squares."""
    n = len(srcs)
    check(load().asr_gather_sumsq_f32(n, _ptr_array(srcs), (c_i64 * n)(*[int(o) for o in offsets]),
                                      (c_i64 * n)(*[int(t.numel()) for t in srcs]), ptr(flat), ptr(sumsq), stream()),
          "asr_gather_sumsq_f32")


def colsum_parts(parts):
    """parts: up to four [rows, ...] contiguous tensors with the same `rows` -> their sums over dim 0, one launch."""
    rows = parts[0].shape[0]
    outs = [torch.empty(p.shape[1:], device=p.device, dtype=torch.float32) for p in parts]
    n = (ctypes.c_int32 * len(parts))(*[int(p.numel() // rows) for p in parts])
    assert all(p.is_contiguous() and p.shape[0] == rows for p in parts) and len(parts) <= 4
    check(load().asr_colsum_parts_f32(len(parts), rows, _ptr_array(parts), n, _ptr_array(outs), stream()), "asr_colsum_parts_f32")
    return outs


def _off(t, elems):
    return c_p(_dev(t).data_ptr() + 4 * int(elems))


# the encoder on packed rows (RowLayout below; model.pBLSTM) - "padded": the time-major padded tensors (measurement)
USE_PACKED_ROWS = os.environ.get("ASR_ENCODER_ROWS", "packed") != "padded"
USE_PERSIST = os.environ.get("ASR_PERSIST", "1") != "0"
USE_PERSIST_DEC = USE_PERSIST and os.environ.get("ASR_PERSIST_DEC", "1") != "0"   # persistent decoder forward
USE_PERSIST_DEC_BWD = USE_PERSIST and os.environ.get("ASR_PERSIST_DEC_BWD", "1") != "0"   # ... and backward
# free-running decode: fused logits/argmax/next-embedding kernel per step (off: the same steps through torch glue)
USE_FEEDBACK_KERNEL = os.environ.get("ASR_FEEDBACK_KERNEL", "1") != "0"
# greedy decode without autograd: a group of 4 utterances stops once all of them have emitted <EOS> (what follows the
# first <EOS> is stripped by the CER path; the predictions of the skipped steps read <EOS>)
DECODE_EARLY_STOP = os.environ.get("ASR_DECODE_EARLY_STOP", "1") != "0"
_persist_scratch = {}
# exchange area of the persistent kernels: the largest user is the LSTM backward with exchanged dh partials,
# [8 groups][2 parities][32 dest][32 src][8 rows][H/32 units] floats = 8 MB at H = 512 (each launcher zeroes what it uses)
XCH_BYTES = 8 * 2 * 32 * 32 * 8 * 20 * 4      # the largest user: exchanged dh partials at H = 640 (20 units per CU), 10 MB
                                              # (= asr_persist_scratch_bytes(); persist_scratch checks it against the library)

# Which path every sequence operator actually took, per process: "<op>_persist" counts launches of the persistent
# XCD-local kernels, "<op>_step" counts sequences that ran on the per-step kernels instead (persistent path switched off,
# or the launcher answered ASR_E_SHAPE = -2: unsupported sizes / not an 8 x 32-CU device).  Ops: lstm_fwd, lstm_bwd,
# dec_fwd, dec_bwd, dec_free (free-running decode forward).  Tests assert on these so that a silent fallback cannot
# pass for the fast path; `with require_persistent():` turns a fallback into an error.
import collections
LAUNCHES = collections.Counter()
_REQUIRE_PERSIST = [os.environ.get("ASR_REQUIRE_PERSIST", "0") != "0"]


def count_path(op, persistent, why=""):
    LAUNCHES[op + ("_persist" if persistent else "_step")] += 1
    if not persistent and _REQUIRE_PERSIST[0]:
        raise RuntimeError("%s fell back to the per-step kernels (%s) while the persistent path was required" % (op, why))


class require_persistent(object):
    """Context manager: any sequence operator that does not run on its persistent kernel raises."""

    def __enter__(self):
        self._old = _REQUIRE_PERSIST[0]
        _REQUIRE_PERSIST[0] = True
        return LAUNCHES

    def __exit__(self, *exc):
        _REQUIRE_PERSIST[0] = self._old
        return False


def _scratch_key(device):
    """One scratch pair per physical device: "cuda" and "cuda:0" name the same one."""
    d = torch.device(device)
    return "%s:%d" % (d.type, torch.cuda.current_device() if d.index is None else d.index) if d.type == "cuda" else str(d)


def persist_scratch(device, trace=False):
    """(xch, ctrl) scratch of the persistent kernels, one pair per device (calls are stream-ordered).  ctrl = 32 int32 words:
    [0] abort latch, [1] its code (set by any aborting launch, cleared only by persist_clear_abort), [16..31] the
    per-launch words the library zeroes before every launch (csrc/persist.h).  trace=True: a separate 4 KB control buffer
    whose words behind the per-launch block receive the clock stamps of the measurement builds (tools/)."""
    if trace:
        tkey = str(device) + "/trace"
        if tkey not in _persist_scratch:
            _persist_scratch[tkey] = (torch.zeros(XCH_BYTES // 8, dtype=torch.int64, device=device),
                                      torch.zeros(16384, dtype=torch.int32, device=device))     # 64 KB: tools/dec_trace2.py
        return _persist_scratch[tkey]
    key = _scratch_key(device)
    if key not in _persist_scratch:
        xb, cb = c_i64(0), c_i64(0)
        load().asr_persist_scratch_bytes(ctypes.byref(xb), ctypes.byref(cb))
        assert xb.value <= XCH_BYTES and cb.value <= 128, "libasr_hip.so wants a larger persistent scratch than this host code allocates"
        # one allocation: [128-byte control block | 10 MB exchange] so that the pre-launch reset is a single fill
        # (persist.h: persist_reset)
        base = torch.zeros(16 + XCH_BYTES // 8, dtype=torch.int64, device=device)
        _persist_scratch[key] = (base[16:], base[:16].view(torch.int32), base)
    return _persist_scratch[key][:2]


def persist_abort_code(device):
    """Which wait gave up first since the latch was cleared (ctrl[1]; 2 = unexpected workgroup placement, others = the
    poll site).  Synchronises."""
    key = _scratch_key(device)
    return int(_persist_scratch[key][1][1].item()) if key in _persist_scratch else 0


def persist_clear_abort(device):
    """Clear the abort latch (start of a step / after the caller has dealt with an abort); stream-ordered, no sync."""
    key = _scratch_key(device)
    if key in _persist_scratch:
        _persist_scratch[key][1][:2].zero_()


# An abort of the persistent kernels is a PLACEMENT problem - a workgroup that did not get its CU because another process's
# kernels held it, a bounded spin that expired behind one - and those pass.  So leaving the persistent kernels is a probation,
# not a verdict: after PERSIST_RETRY_STEPS train steps on the per-step kernels (3x slower at cfg-2) they are tried again,
# twice as late after every further abort (200, 400, ... capped at 64x).  0: never again (the behaviour up to round 4).
# Env ASR_PERSIST_RETRY_STEPS, config key `persist_retry_steps` (not a reference key).
PERSIST_RETRY_STEPS = int(os.environ.get("ASR_PERSIST_RETRY_STEPS", "200"))
_PROBATION = dict(wanted=None, aborts=0, steps=0, retry_at=None)


def disable_persistent(device=None, permanent=False):
    """Route every sequence operator of this process to the per-step HIP kernels (after an abort, or - permanent - when
    several processes share one GPU) and clear the abort latch.  Unless permanent, persistent_step_tick() brings the
    persistent kernels back after a probation (see PERSIST_RETRY_STEPS)."""
    global USE_PERSIST, USE_PERSIST_DEC, USE_PERSIST_DEC_BWD
    st = _PROBATION
    if st["wanted"] is None:
        st["wanted"] = (USE_PERSIST, USE_PERSIST_DEC, USE_PERSIST_DEC_BWD)       # what this process started with
    if permanent:
        st["wanted"] = (False, False, False)
    USE_PERSIST = USE_PERSIST_DEC = USE_PERSIST_DEC_BWD = False
    if not permanent:                 # (ranks sharing a card leave by decision, not after an abort: nothing to count, and a
        st["aborts"] += 1             #  later abort's probation starts at the base length)
    if permanent or PERSIST_RETRY_STEPS <= 0 or not any(st["wanted"]):
        st["retry_at"] = None
    else:
        st["retry_at"] = st["steps"] + PERSIST_RETRY_STEPS * (1 << min(st["aborts"] - 1, 6))
    if device is not None:
        persist_clear_abort(device)


def idle_xcd_mask(nbatch):
    """The XCDs NO persistent kernel of a step uses when a rank's batch has `nbatch` utterances (bit x = XCC id x), 0 when
    there are fewer than four.  Group g of a persistent launch is XCC id g; the LSTM kernels use ndir * ceil(nb / 4) groups
    for nb <= 16 rows (4-row groups, csrc/lstm_persist.hip: rows_per_group), the decoder kernels ceil(nb / 2) or ceil(nb / 4)
    (csrc/dec_persist.hip: DecGeo): eight utterances or fewer stay on XCDs 0-3.  Not with the per-step kernels (they use the
    whole chip) or with a forced row geometry (ASR_LSTM_ROWS: measurements)."""
    if not (USE_PERSIST and USE_PERSIST_DEC and USE_PERSIST_DEC_BWD) or os.environ.get("ASR_LSTM_ROWS"):
        return 0
    if _SIDE_FORCE_MASK is not None:           # measurement: ASR_SIDE_FORCE_MASK=0xff puts the products beside the chains of ANY batch
        return _SIDE_FORCE_MASK if int(nbatch) > 0 else 0
    return 0xF0 if 0 < int(nbatch) <= 8 else 0


_SIDE_FORCE_MASK = int(os.environ["ASR_SIDE_FORCE_MASK"], 0) & 0xff if os.environ.get("ASR_SIDE_FORCE_MASK") else None


def persistent_step_tick():
    """Called once per train step (Solver._step).  True when this call ended a probation: the sequence operators of the
    step that follows run on the persistent kernels again (an abort there is found and repeated like any other)."""
    global USE_PERSIST, USE_PERSIST_DEC, USE_PERSIST_DEC_BWD
    st = _PROBATION
    st["steps"] += 1
    if st["retry_at"] is None or st["steps"] < st["retry_at"]:
        return False
    USE_PERSIST, USE_PERSIST_DEC, USE_PERSIST_DEC_BWD = st["wanted"]
    st["retry_at"] = None
    return True


def persistent_probation():
    """(aborts so far, train steps until the persistent kernels are tried again or None)."""
    st = _PROBATION
    return st["aborts"], (None if st["retry_at"] is None else max(0, st["retry_at"] - st["steps"]))


def persist_aborted(device):
    """True if ANY persistent launch on `device` aborted since the latch was last cleared (synchronises; for tests /
    end-of-step checks).  The latch is sticky across launches: the per-launch abort word is zeroed before every launch,
    and a sequence operator is several launches."""
    key = _scratch_key(device)
    return key in _persist_scratch and int(_persist_scratch[key][1][0].item()) != 0


def persist_abort_flag(device):
    """The abort latch as a 1-element int32 device tensor (no sync): lets a data-parallel step add it to the values its
    all-reduce carries."""
    return persist_scratch(device)[1][:1]


class RowLayout(object):
    """Packed rows of the encoder (include/asr_hip.h, "PACKED ROWS"): per layer l = 0 .. n (n = the encoder output), batch
    row b owns ext[l][b] rows starting at base[l][b]; time t of utterance b sits at row base[l][b] + t.  The extents halve
    with the pyramid (ext[l] = 2 ext[l + 1] where layer l subsamples), so the pair-concat is a reshape of the row matrix,
    and every block ends with at least one padding row (ext > len).  Built on the host from the utterance lengths and
    uploaded once per batch (one non-blocking copy): `dev` [n + 1][3][B] int32 = (lens, base, ext) per layer."""

    ROW_QUANTUM = 16

    def __init__(self, ilens, subsample, device, t_pad=None):
        import numpy as np
        n = len(subsample)
        lens = [np.asarray([int(v) for v in ilens], dtype=np.int64)]
        pads = [int(max(ilens)) if t_pad is None else int(t_pad[0])]
        for i in range(n):
            lens.append((lens[-1] + 1) // 2 if subsample[i] > 1 else lens[-1].copy())
            pads.append(((pads[-1] + 1) // 2 if subsample[i] > 1 else pads[-1]) if t_pad is None else int(t_pad[i + 1]))
        ext = [None] * (n + 1)
        ext[n] = lens[n] + 1
        # The row counts are the M of the forward products and the K of the weight-gradient products.  The output's count is
        # rounded up to a multiple of ROW_QUANTUM = 16 (the extents double from there: 32 / 64 / 128 rows for a three-layer
        # pyramid), so that K % 32 == 0 where the GEMM's fast kernels want it and the layer-0 input projection has whole
        # 128-row tiles (its plain epilogue: 97 against 137 us at cfg-2).  The extra padding rows go to the shortest utterances
        # first (round robin); the recurrences run max(lens) steps whatever the extents are.
        odd = (-int(ext[n].sum())) % self.ROW_QUANTUM
        order = np.argsort(lens[n], kind="stable")
        while odd > 0:
            take = min(odd, len(ilens))
            ext[n][order[:take]] += 1
            odd -= take
        for i in range(n - 1, -1, -1):
            ext[i] = ext[i + 1] * 2 if subsample[i] > 1 else ext[i + 1].copy()
        self.B, self.n = len(ilens), n
        self.lens = [l.astype(np.int32) for l in lens]          # host copies (the launchers take rowext_host)
        self.ext = [np.ascontiguousarray(e.astype(np.int32)) for e in ext]
        self.base = [np.concatenate([[0], np.cumsum(e)[:-1]]).astype(np.int32) for e in ext]
        self.rows = [int(e.sum()) for e in ext]                 # R per layer
        self.steps = [int(l.max()) for l in lens]               # time steps the recurrence of layer l has to run (the kernels
                                                                # zero the padding rows of a block behind them themselves)
        self.t_pad = pads                                       # padded time extent per layer (the global one of a shard)
        table = np.stack([np.stack([self.lens[i], self.base[i], self.ext[i]]) for i in range(n + 1)])
        self.dev = to_device_i32(table, device)                 # [n + 1][3][B]

    def lens_dev(self, l):
        return self.dev[l, 0]

    def base_dev(self, l):
        return self.dev[l, 1]

    def ext_dev(self, l):
        return self.dev[l, 2]

    def replicated_rows(self, l):
        """Output rows of layer l's pair-concat whose second half is the reference's replicate-padded frame (model.py:85-88:
        an odd padded length T; only utterances of exactly that length see a non-zero replica): [(row of the pair)]."""
        import numpy as np
        T = self.t_pad[l]
        if T % 2 == 0:
            return []
        return [int((self.base[l][b] + T - 1) // 2) for b in np.nonzero(self.lens[l] == T)[0]]


class LayerRows(object):
    """One layer's view of a RowLayout: what the LSTM launchers need."""

    def __init__(self, layout, l):
        self.layout, self.l = layout, l
        self.B, self.R, self.T = layout.B, layout.rows[l], layout.steps[l]
        self.lens, self.base, self.ext = layout.lens_dev(l), layout.base_dev(l), layout.ext_dev(l)
        self.lens_host, self.ext_max = layout.lens[l], int(layout.ext[l].max())

    def host_ptr(self):
        return c_p(self.lens_host.ctypes.data)


def lstm_seq_fwd(gates, w_hh, lens, y, c, use_graphs=True, rows=None):
    """rows: a LayerRows - gates / y / c are then row matrices [R, 1, ...] in the packed layout (lens = rows.lens)."""
    T, B, ndir, H4 = gates.shape
    H = H4 // 4
    lib = load()
    rb, re, rh = (ptr(rows.base), ptr(rows.ext), rows.host_ptr()) if rows is not None else (None, None, None)
    if rows is not None:
        assert B == 1 and T == rows.R
        T, B = rows.T, rows.B
    if USE_PERSIST:
        xch, ctrl = persist_scratch(gates.device)
        rc = lib.asr_lstm_seq_fwd_persist(T, B, B, H, ndir, ptr(gates), ptr(w_hh), ptr(lens), rb, re, rh, ptr(y), ptr(c),
                                          c_p(xch.data_ptr()), c_p(ctrl.data_ptr()), ARITH[0], stream())
        if rc == 0:
            count_path("lstm_fwd", True)
            return
        if rc != -2:                      # anything but ASR_E_SHAPE is an error
            check(rc, "asr_lstm_seq_fwd_persist")
    count_path("lstm_fwd", False, "H=%d B=%d ndir=%d persist=%s" % (H, B, ndir, USE_PERSIST))
    groups = row_groups(B)
    gh = [graphs_for(i) if use_graphs else None for i in range(len(groups))]     # created on the calling thread

    def one(gi, grp, st):
        b0, nb = grp
        if rows is not None:                               # packed rows: the row maps are absolute, only they advance
            check(lib.asr_lstm_seq_fwd(T, B, nb, H, ndir, ptr(gates), ptr(w_hh), _off(lens, b0), _off(rows.base, b0),
                                       _off(rows.ext, b0), ptr(y), ptr(c), gh[gi], st), "asr_lstm_seq_fwd")
            return
        check(lib.asr_lstm_seq_fwd(T, B, nb, H, ndir, _off(gates, b0 * ndir * H4), ptr(w_hh), _off(lens, b0), None, None,
                                   _off(y, b0 * ndir * H), _off(c, b0 * ndir * H), gh[gi], st), "asr_lstm_seq_fwd")

    run_grouped(groups, one)


def lstm_seq_bwd(gates, w_hhT, lens, dy, c, dcarry, y=None, dw_hh=None, db=None, w_hh=None, rows=None):
    """-> (fused_dw, fused_db): whether dW_hh and the bias gradient `db` were accumulated by the persistent kernel itself
    (db: by every persistent backward kernel; dW_hh: by all of them except the bf16x6 exchanged-partials kernel, see
    asr_lstm_bwd_persist_fuses_dw - the caller then forms it with a GEMM).
    w_hhT: the transposed recurrent weights [ndir][H][4H], or a callable producing them on demand; w_hh: the forward
    layout [ndir][4H][H] - when given, the kernel that can read it directly is tried first and the transpose is only
    formed if that kernel does not apply."""
    T, B, ndir, H4 = gates.shape
    H = H4 // 4
    lib = load()
    ar = ARITH[0]
    rb, re, rh = (ptr(rows.base), ptr(rows.ext), rows.host_ptr()) if rows is not None else (None, None, None)
    if rows is not None:                                   # packed rows (see lstm_seq_fwd): dW_hh is always the caller's
        assert B == 1 and T == rows.R
        T, B = rows.T, rows.B
    fuses = USE_PERSIST and rows is None and lib.asr_lstm_bwd_persist_fuses_dw(H, ar) == 1 and y is not None and dw_hh is not None
    yk, dwk = (y, dw_hh) if fuses else (None, None)
    if USE_PERSIST and w_hh is not None:
        xch, ctrl = persist_scratch(gates.device)
        rc = lib.asr_lstm_seq_bwd_persist_w(T, B, B, H, ndir, ptr(gates), ptr(w_hh), ptr(lens), rb, re, rh, ptr(dy), ptr(c),
                                            ptr(yk), ptr(dwk), ptr(db), c_p(xch.data_ptr()), c_p(ctrl.data_ptr()), ar, stream())
        if rc == 0:
            count_path("lstm_bwd", True)
            return fuses, db is not None
        if rc != -2:
            check(rc, "asr_lstm_seq_bwd_persist_w")
    if callable(w_hhT):
        w_hhT = w_hhT()
    if USE_PERSIST:
        xch, ctrl = persist_scratch(gates.device)
        rc = lib.asr_lstm_seq_bwd_persist(T, B, B, H, ndir, ptr(gates), ptr(w_hhT), ptr(lens), rb, re, rh, ptr(dy), ptr(c),
                                          ptr(yk), ptr(dwk), ptr(db), c_p(xch.data_ptr()), c_p(ctrl.data_ptr()), ar, stream())
        if rc == 0:
            count_path("lstm_bwd", True)
            return fuses, db is not None
        if rc != -2:
            check(rc, "asr_lstm_seq_bwd_persist")
    count_path("lstm_bwd", False, "H=%d B=%d ndir=%d persist=%s" % (H, B, ndir, USE_PERSIST))
    groups = row_groups(B)
    gh = [graphs_for(i) for i in range(len(groups))]

    def one(gi, grp, st):
        b0, nb = grp
        if rows is not None:
            check(lib.asr_lstm_seq_bwd(T, B, nb, H, ndir, ptr(gates), ptr(w_hhT), _off(lens, b0), _off(rows.base, b0),
                                       _off(rows.ext, b0), ptr(dy), ptr(c), _off(dcarry, b0 * ndir * H), gh[gi], st),
                  "asr_lstm_seq_bwd")
            return
        check(lib.asr_lstm_seq_bwd(T, B, nb, H, ndir, _off(gates, b0 * ndir * H4), ptr(w_hhT), _off(lens, b0), None, None,
                                   _off(dy, b0 * ndir * H), _off(c, b0 * ndir * H), _off(dcarry, b0 * ndir * H), gh[gi],
                                   st), "asr_lstm_seq_bwd")

    run_grouped(groups, one)
    return False, False


# dropout masks regenerated inside the consuming kernels from a seed (off: materialised fp32 masks, as injected by tests)
USE_SEEDED_DROPOUT = os.environ.get("ASR_SEEDED_DROPOUT", "1") != "0"


class SeededMask(object):
    """Inverted-dropout mask given by (seed, p) over a tensor shape: mask(i) = keep(seed, i) / (1 - p) (dropout.hip)."""

    def __init__(self, shape, p, device, seed=None):
        self.shape, self.p, self.device = tuple(int(v) for v in shape), float(p), device
        # seeds come from torch's CPU generator: reproducible under torch.manual_seed, no device sync
        self.seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if seed is None else int(seed)

    def tensor(self):
        n = 1
        for v in self.shape:
            n *= v
        pad = torch.empty((n + 3) // 4 * 4, device=self.device, dtype=torch.float32)
        check(load().asr_dropout_mask_f32(pad.numel(), ptr(pad), self.seed, self.p, stream()), "asr_dropout_mask_f32")
        return pad[:n].view(self.shape)


def dropout_seeded_(x, m):
    """x *= mask in place (x contiguous, numel % 4 == 0)."""
    check(load().asr_dropout_seeded_f32(x.numel(), ptr(x), m.seed, m.p, stream()), "asr_dropout_seeded_f32")
    return x


def relu_dropout_bwd(grad, y, seed, p):
    """grad * mask * (y > 0) in one pass (p == 0: the plain relu gradient)."""
    out = torch.empty_like(y)
    check(load().asr_relu_dropout_bwd_f32(y.numel(), ptr(grad), ptr(y), int(seed), float(p), ptr(out), stream()),
          "asr_relu_dropout_bwd_f32")
    return out


def pyramid_fwd(x, mask, out):
    T, B, C = x.shape
    if isinstance(mask, SeededMask):
        check(load().asr_pyramid_concat_fwd_seeded(T, B, C, ptr(x), mask.seed, mask.p, ptr(out), stream()),
              "asr_pyramid_concat_fwd_seeded")
        return
    check(load().asr_pyramid_concat_fwd(T, B, C, ptr(x), ptr(mask), ptr(out), stream()), "asr_pyramid_concat_fwd")


def pyramid_bwd(dout, mask, din):
    T, B, C = din.shape
    if isinstance(mask, SeededMask):
        check(load().asr_pyramid_concat_bwd_seeded(T, B, C, ptr(dout), mask.seed, mask.p, ptr(din), stream()),
              "asr_pyramid_concat_bwd_seeded")
        return
    check(load().asr_pyramid_concat_bwd(T, B, C, ptr(dout), ptr(mask), ptr(din), stream()), "asr_pyramid_concat_bwd")


def rows_pack(x, rows):
    """x [B, T, C] (zero-padded batch, dataloader.py:6-12) -> [R, C] in the packed layout of `rows` (a LayerRows)."""
    B, T, C = x.shape
    out = torch.empty(rows.R, C, device=x.device, dtype=torch.float32)
    check(load().asr_rows_pack_f32(B, T, C, ptr(x), ptr(rows.lens), ptr(rows.base), ptr(rows.ext), rows.ext_max, ptr(out), stream()),
          "asr_rows_pack_f32")
    return out


def rows_unpack_fwd(packed, rows, T, fill, mask, fill_relu=False):
    """[R, C] -> [B, T, C]; frames behind an utterance = fill * mask (mask: None, a [B, T, C] tensor or a SeededMask);
    fill_relu: relu(fill) * mask (fill = the last projection's bias as it is)."""
    C = packed.shape[1]
    out = torch.empty(rows.B, T, C, device=packed.device, dtype=torch.float32)
    seeded = isinstance(mask, SeededMask)
    check(load().asr_rows_unpack_fwd_f32(rows.B, T, C, ptr(packed), ptr(rows.lens), ptr(rows.base), ptr(fill),
                                         1 if fill_relu else 0,
                                         None if (mask is None or seeded) else ptr(mask), mask.seed if seeded else 0,
                                         mask.p if seeded else 0.0, ptr(out), stream()), "asr_rows_unpack_fwd_f32")
    return out


def rows_unpack_bwd(dout, rows, C, mask, want_fill, relu_of=None, dfill=None):
    """-> (drows [R, C], dfill [C] or None).  relu_of: the fill vector in front of its relu (fill_relu of the forward): its
    gradient is masked by relu_of > 0.  dfill: a zeroed accumulator to use instead of a fresh one."""
    B, T, _ = dout.shape
    drows = torch.empty(rows.R, C, device=dout.device, dtype=torch.float32)
    if not want_fill:
        dfill = None
    elif dfill is None:
        dfill = torch.zeros(C, device=dout.device, dtype=torch.float32)
    seeded = isinstance(mask, SeededMask)
    check(load().asr_rows_unpack_bwd_f32(B, T, C, ptr(dout), ptr(rows.lens), ptr(rows.base), ptr(rows.ext), rows.ext_max,
                                         None if (mask is None or seeded) else ptr(mask), mask.seed if seeded else 0,
                                         mask.p if seeded else 0.0, ptr(drows), ptr(dfill), ptr(relu_of), stream()),
          "asr_rows_unpack_bwd_f32")
    return drows, dfill
