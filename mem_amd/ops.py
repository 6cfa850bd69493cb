"""Thin ctypes wrappers over the ViT kernels of libmemhip.so (include/memhip.h).

These take torch CUDA tensors only as (pointer, shape) carriers; all arithmetic happens in the
hand-written HIP kernels.  Everything is enqueued on torch's current stream.
"""
import ctypes as C

import torch

from ._lib import check, declare, f32, i32, i64, lib, ptr, stream_ptr, sz, vp

EPI_BIAS_BF16, EPI_BIAS_GELU, EPI_RESIDUAL, EPI_DGELU, EPI_F32, EPI_PATCH_EMBED = range(6)


class GemmArgs(C.Structure):
    """== memhip_gemm_args_t."""
    _fields_ = [("A", vp), ("B", vp), ("lda", i64), ("ldb", i64),
                ("M", i32), ("N", i32), ("K", i32), ("epilogue", i32),
                ("out0", vp), ("ldo0", i64), ("out1", vp), ("ldo1", i64),
                ("bias", vp), ("vec1", vp), ("resid", vp), ("ldr", i64),
                ("aux", vp), ("ldaux", i64), ("rowmask", vp), ("keep_prob", f32),
                ("colscale", f32), ("colscale_n", i32), ("rows_per_sample", i32), ("accumulate", i32)]


declare({"memhip_gemm_bf16_nt": (i32, [C.POINTER(GemmArgs), vp])})


def _p(t):
    return None if t is None else t.data_ptr()


def gemm_nt(A, B, M, N, K, epi, out0=None, out1=None, bias=None, vec1=None, resid=None, aux=None,
            rowmask=None, keep_prob=1.0, colscale=1.0, colscale_n=0, rows_per_sample=1, accumulate=False,
            lda=None, ldb=None, ldo0=None, ldo1=None, ldr=None, ldaux=None):
    """C[M,N] = A[M,K] @ B[N,K]^T with a fused epilogue.  A/B bf16, row-major, K contiguous."""
    a = GemmArgs()
    a.A, a.B = _p(A), _p(B)
    a.lda = A.stride(0) if lda is None else lda
    a.ldb = B.stride(0) if ldb is None else ldb
    a.M, a.N, a.K, a.epilogue = M, N, K, epi
    a.out0 = _p(out0)
    a.ldo0 = (out0.stride(0) if out0 is not None else 0) if ldo0 is None else ldo0
    a.out1 = _p(out1)
    a.ldo1 = (out1.stride(0) if out1 is not None else 0) if ldo1 is None else ldo1
    a.bias, a.vec1, a.resid = _p(bias), _p(vec1), _p(resid)
    a.ldr = (resid.stride(0) if resid is not None else 0) if ldr is None else ldr
    a.aux = _p(aux)
    a.ldaux = (aux.stride(0) if (aux is not None and aux.dim() > 1) else 0) if ldaux is None else ldaux
    a.rowmask = _p(rowmask)
    a.keep_prob, a.colscale, a.colscale_n = keep_prob, colscale, colscale_n
    a.rows_per_sample, a.accumulate = rows_per_sample, int(accumulate)
    check(lib.memhip_gemm_bf16_nt(C.byref(a), stream_ptr()), "gemm_bf16_nt")
