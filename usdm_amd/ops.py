"""Thin per-kernel wrappers over the C-ABI (torch tensors in, device pointers out).

These exist for the parity tests and for the Python drop-ins; they add no arithmetic.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import BF16, F32, GemmArgs, check, lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _dt(t):
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"unsupported dtype {t.dtype}")


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.UsdmError("usdm_amd kernels run on the GPU only (no CPU fallback); got a CPU tensor")


def gemm(A, W, *, M, N, Kc, taps=1, lda=None, rowsA=None, a_row_mul=1, a_row_off=0, a_row_step=0,
         a_tap_stride=0, ldw=None, groups=1, batch=1, a_gstride=0, w_gstride=0, a_bstride=0, c_gcol=0,
         c_bstride=0, bias=None, alpha=1.0, act=0, round_bf16=False, residual=None, ldr=0,
         out32=None, out16=None, ldc=None, c_row_mul=1, c_row_off=0, transpose_out=False,
         qkv=None):
    """Raw launch of usdm_gemm; see include/usdm_hip.h for the meaning of every field."""
    _need_cuda(A, W, bias, residual, out32, out16)
    a = GemmArgs()
    a.dtype = _dt(A)
    assert W.dtype == A.dtype
    a.M, a.N, a.taps, a.Kc = M, N, taps, Kc
    a.A, a.lda = _ptr(A), (lda if lda is not None else A.stride(-2))
    a.rowsA = rowsA if rowsA is not None else M
    a.a_row_mul, a.a_row_off, a.a_row_step, a.a_tap_stride = a_row_mul, a_row_off, a_row_step, a_tap_stride
    a.W, a.ldw = _ptr(W), (ldw if ldw is not None else taps * Kc)
    a.groups, a.batch = groups, batch
    a.a_gstride, a.w_gstride, a.a_bstride, a.c_gcol, a.c_bstride = a_gstride, w_gstride, a_bstride, c_gcol, c_bstride
    a.bias, a.alpha, a.act, a.round_bf16 = _ptr(bias), alpha, act, int(round_bf16)
    a.residual = _ptr(residual)
    a.res_dtype = _dt(residual) if residual is not None else F32
    a.ldr = ldr
    a.C32, a.C16 = _ptr(out32), _ptr(out16)
    a.ldc = ldc if ldc is not None else N
    a.c_row_mul, a.c_row_off, a.transpose_out = c_row_mul, c_row_off, int(transpose_out)
    if qkv is not None:
        a.epi = _lib.EPI_QKV_HEADS
        a.qkv_S, a.qkv_Spad, a.qkv_H, a.qkv_D = qkv["S"], qkv["Spad"], qkv["H"], qkv["D"]
        a.qkv_q, a.qkv_k, a.qkv_v = _ptr(qkv["q"]), _ptr(qkv["k"]), _ptr(qkv["v"])
    check(lib.usdm_gemm(C.byref(a), _stream()), "usdm_gemm")
