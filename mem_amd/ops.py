"""Thin ctypes wrappers over the ViT kernels of libmemhip.so (include/memhip.h).

These take torch CUDA tensors only as (pointer, shape) carriers; all arithmetic happens in the
hand-written HIP kernels.  Everything is enqueued on torch's current stream.
"""
import ctypes as C

import torch

from ._lib import check, declare, f32, i32, i64, lib, ptr, stream_ptr, sz, vp

EPI_BIAS_BF16, EPI_BIAS_GELU, EPI_RESIDUAL, EPI_DGELU, EPI_F32, EPI_PATCH_EMBED, EPI_BIAS_GELU_DG, EPI_MUL_AUX = range(8)


class GemmArgs(C.Structure):
    """== memhip_gemm_args_t."""
    _fields_ = [("A", vp), ("B", vp), ("lda", i64), ("ldb", i64),
                ("M", i32), ("N", i32), ("K", i32), ("epilogue", i32),
                ("out0", vp), ("ldo0", i64), ("out1", vp), ("ldo1", i64),
                ("bias", vp), ("vec1", vp), ("resid", vp), ("ldr", i64),
                ("aux", vp), ("ldaux", i64), ("rowmask", vp), ("keep_prob", f32),
                ("colscale", f32), ("colscale_n", i32), ("rows_per_sample", i32), ("accumulate", i32),
                ("colsum", vp), ("sample_map", vp), ("colsum_copies", i32), ("reserved0", i32)]


declare({"memhip_gemm_bf16_nt": (i32, [C.POINTER(GemmArgs), vp])})


def _p(t):
    return None if t is None else t.data_ptr()


# Optional live timing of every GEMM launch with HIP events on the launch stream (bench.py's
# roofline leg): set GEMM_TIMER to a list; entries are (start_event, end_event, flops, epilogue).
GEMM_TIMER = None
# Events for the timer, created AND recorded once before the timed region (bench.py): creating / first-recording a few
# hundred timing events inside the region grows the runtime's signal pool there, seen as ~14 ms stalls some steps later.
GEMM_EVENT_POOL = []


def _timer_event():
    return GEMM_EVENT_POOL.pop() if GEMM_EVENT_POOL else torch.cuda.Event(enable_timing=True)


def gemm_nt(A, B, M, N, K, epi, out0=None, out1=None, bias=None, vec1=None, resid=None, aux=None,
            rowmask=None, keep_prob=1.0, colscale=1.0, colscale_n=0, rows_per_sample=1, accumulate=False, colsum=None,
            lda=None, ldb=None, ldo0=None, ldo1=None, ldr=None, ldaux=None, sample_map=None, colsum_copies=0):
    """C[M,N] = A[M,K] @ B[N,K]^T with a fused epilogue.  A/B bf16, row-major, K contiguous.
    colsum_copies > 1: `colsum` is a zeroed [copies, N] workspace, folded into the bias gradient by colsum_fold."""
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
    a.colsum = _p(colsum)
    a.sample_map = _p(sample_map)
    a.colsum_copies = colsum_copies
    if GEMM_TIMER is None:
        check(lib.memhip_gemm_bf16_nt(C.byref(a), stream_ptr()), "gemm_bf16_nt")
    else:
        e0, e1 = _timer_event(), _timer_event()
        e0.record()
        check(lib.memhip_gemm_bf16_nt(C.byref(a), stream_ptr()), "gemm_bf16_nt")
        e1.record()
        GEMM_TIMER.append((e0, e1, 2.0 * M * N * K, epi))


f64 = C.c_double
declare({
    "memhip_layernorm_fwd": (i32, [vp, i64, vp, i32, i32, vp, vp, f32, vp, i64, vp, vp, vp]),
    "memhip_layernorm_bwd": (i32, [vp, i64, vp, i64, vp, i32, i32, vp, vp, vp, vp, i64, i32, vp, vp, vp]),
    "memhip_layernorm_bwd_branch": (i32, [vp, i64, vp, i64, i32, i32, vp, vp, vp, vp, i64, vp, vp, vp, i64, vp, vp, f32,
                                          i32, vp, i64, vp, vp, vp]),
    "memhip_layerscale_grad": (i32, [vp, i64, vp, i64, vp, vp, vp, i32, i32, vp, vp]),
    "memhip_branch_bwd": (i32, [vp, i64, vp, i64, vp, vp, f32, i32, i32, i32, vp, i64, vp, vp, vp]),
    "memhip_branch_bwd_map": (i32, [vp, i64, vp, i64, vp, vp, f32, i32, i32, i32, vp, i64, vp, vp, vp, vp]),
    "memhip_layernorm_bwd_branch_map": (i32, [vp, i64, vp, i64, i32, i32, vp, vp, vp, vp, i64, vp, vp, vp, i64, vp, vp, f32,
                                              i32, vp, i64, vp, vp, vp, vp, vp]),
    "memhip_embed_bwd": (i32, [vp, i64, vp, i32, i32, i32, vp, i64, vp, vp, vp]),
    "memhip_cross_entropy": (i32, [vp, i64, vp, i32, i32, f32, vp, vp, i32, vp, vp]),
    "memhip_attn_tokens_padded": (i32, [i32]),
    "memhip_relpos_gather": (i32, [vp, vp, i32, i32, i32, vp, vp, vp]),
    "memhip_attn_fwd": (i32, [vp, i64, i32, i32, i32, i32, vp, i32, i32, vp, i64, vp, vp]),
    "memhip_attn_delta": (i32, [vp, vp, i64, i64, i32, vp, vp]),
    "memhip_attn_bwd": (i32, [vp, i64, vp, i64, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, i64, vp, vp, vp, vp]),
    "memhip_attn_bwd_out": (i32, [vp, i64, vp, i64, vp, i64, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, i64, vp, vp, vp, vp]),
    "memhip_attn_bwd_workspace": (i64, [i32, i32, i32, i32, i32]),
    "memhip_attn_bwd_ws": (i32, [vp, i64, vp, i64, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, i64, vp, vp, vp, vp, i64, vp]),
    "memhip_attn_bwd_out_ws": (i32, [vp, i64, vp, i64, vp, i64, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, i64, vp, vp, vp,
                                      vp, i64, vp]),
    "memhip_cast_f32_bf16": (i32, [vp, vp, i64, vp]),
    "memhip_copy_samples_f32": (i32, [vp, vp, vp, i32, i64, vp]),
    "memhip_zero": (i32, [vp, i64, vp]),
    "memhip_stream_reserve_cus": (i32, [vp, i32]),
    "memhip_zero_ranges": (i32, [vp, vp, i32, i64, vp]),
    "memhip_transpose_cast_f32_bf16": (i32, [vp, i64, i32, i32, vp, i64, vp]),
    "memhip_transpose_bf16": (i32, [vp, i64, i32, i32, vp, i64, i32, vp, i32, i32, vp, i32, i32, vp]),
    "memhip_im2col_bf16": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "memhip_fill_cls": (i32, [vp, i64, i32, i32, i32, vp, vp]),
    "memhip_gemv_bf16_acc": (i32, [vp, i64, i32, i32, vp, vp, vp, vp, vp]),
    "memhip_conv2d_nhwc_bf16": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "memhip_nchw_to_padded_nhwc4": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "memhip_argmax_rows_bf16": (i32, [vp, i64, i32, i32, vp, vp]),
    "memhip_conv2d_nhwc_f32": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "memhip_nchw_to_padded_nhwc4_f32": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "memhip_argmax_rows_f32": (i32, [vp, i64, i32, i32, vp, vp, vp]),
    "memhip_argmax_rows_f32_ex": (i32, [vp, i64, i32, i32, vp, vp, vp, vp, i32, vp]),
    "memhip_tok_flag_samples": (i32, [vp, vp, i32, i32, f32, vp, vp, vp, vp]),
    "memhip_tok_gather_images_f32": (i32, [vp, i32, i32, i32, vp, vp, vp, vp, i32, i32, vp, vp, vp]),
    "memhip_conv2d_nhwc_f32_dyn": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "memhip_tok_scatter_ids": (i32, [vp, vp, vp, i32, i32, i32, vp, vp]),
    "memhip_grad_norm_workspace": (sz, []),
    "memhip_grad_norm": (i32, [vp, i64, vp, vp, sz, vp]),
    "memhip_adamw": (i32, [vp, vp, vp, vp, i64, vp, f64, f64, f64, f64, f64, i32, vp, f64, vp]),
    "memhip_transpose_cast_batched": (i32, [vp, vp, i32, i32, vp]),
    "memhip_adamw_groups": (i32, [vp, vp, vp, vp, i64, vp, vp, i32, f64, f64, f64, i32, vp, f64, vp]),
})


def layernorm_fwd(x, gamma, beta, y, mean, rstd, R, D, eps=1e-6, row_idx=None):
    check(lib.memhip_layernorm_fwd(ptr(x), x.stride(0), ptr(row_idx), R, D, ptr(gamma), ptr(beta), eps,
                                   ptr(y), y.stride(0), ptr(mean), ptr(rstd), stream_ptr()), "layernorm_fwd")


def layernorm_bwd(dy, x, gamma, mean, rstd, dres, dgamma, dbeta, R, D, accumulate=True, row_idx=None):
    check(lib.memhip_layernorm_bwd(ptr(dy), dy.stride(0), ptr(x), x.stride(0), ptr(row_idx), R, D, ptr(gamma),
                                   ptr(mean), ptr(rstd), ptr(dres), dres.stride(0), int(accumulate),
                                   ptr(dgamma), ptr(dbeta), stream_ptr()), "layernorm_bwd")


def layernorm_bwd_branch(dy, x, gamma, mean, rstd, dres, dgamma, dbeta, R, D, y_b, gamma_b, dy_b, dgamma_b, dbias_b,
                         rowmask=None, keep_prob=1.0, rows_per_sample=1, in_map=None, out_map=None):
    """layernorm_bwd(accumulate=True) + the branch_bwd that reads the updated dres, in one pass.  in_map / out_map (i32
    [samples], -1 = dropped): work-skipping stochastic depth, dy / mean / rstd and dy_b then hold kept samples only."""
    check(lib.memhip_layernorm_bwd_branch_map(ptr(dy), dy.stride(0), ptr(x), x.stride(0), R, D, ptr(gamma), ptr(mean),
                                              ptr(rstd), ptr(dres), dres.stride(0), ptr(dgamma), ptr(dbeta), ptr(y_b),
                                              y_b.stride(0) if y_b is not None else 0, ptr(gamma_b), ptr(rowmask), keep_prob,
                                              rows_per_sample, ptr(dy_b), dy_b.stride(0), ptr(dgamma_b), ptr(dbias_b),
                                              ptr(in_map), ptr(out_map), stream_ptr()), "layernorm_bwd_branch")


def layerscale_grad(W16, dW, bias, dbias, gamma, N, K, dgamma):
    """dgamma[N] = (rowdot(W16, dW) + bias * dbias) / gamma  (overwrites dgamma)."""
    check(lib.memhip_layerscale_grad(ptr(W16), W16.stride(0), ptr(dW), dW.stride(0), ptr(bias), ptr(dbias), ptr(gamma),
                                     N, K, ptr(dgamma), stream_ptr()), "layerscale_grad")


def branch_bwd(dx, y, gamma, dy, dgamma, dbias, M, D, rowmask=None, keep_prob=1.0, rows_per_sample=1, out_map=None):
    check(lib.memhip_branch_bwd_map(ptr(dx), dx.stride(0), ptr(y), y.stride(0) if y is not None else 0, ptr(gamma), ptr(rowmask),
                                    keep_prob, rows_per_sample, M, D, ptr(dy), dy.stride(0), ptr(dgamma), ptr(dbias),
                                    ptr(out_map), stream_ptr()), "branch_bwd")


def gemv_acc(W, N, K, x, y, x_acc=None, zero=None):
    """y[N] += W[N,K] (bf16) @ x[K]; optionally x_acc += x and zero[:] = 0."""
    check(lib.memhip_gemv_bf16_acc(ptr(W), W.stride(0), N, K, ptr(x), ptr(y), ptr(x_acc), ptr(zero), stream_ptr()),
          "gemv_bf16_acc")


def conv2d_nhwc(x_pad, weight, bias, out, B, H, W, Cin, Cout, ksize, stride, pad, relu=False, add=None, out_padded=True,
                n_active=None):
    """x_pad [B,H+2,W+2,Cin] -> out [B,Ho+2,Wo+2,Cout] interior (or dense [B*Ho*Wo,Cout]); bf16 or fp32 by x_pad.dtype.
    n_active (fp32 only): device int32 [1], the number of live samples of the capacity B."""
    if x_pad.dtype == torch.float32:
        assert weight.dtype == torch.float32 and out.dtype == torch.float32 and (add is None or add.dtype == torch.float32)
        if n_active is not None:
            check(lib.memhip_conv2d_nhwc_f32_dyn(ptr(x_pad), ptr(weight), ptr(bias), ptr(add), ptr(out), B, H, W, Cin, Cout,
                                                 ksize, stride, pad, int(relu), int(out_padded), ptr(n_active), stream_ptr()),
                  "conv2d_nhwc_f32_dyn")
            return
        check(lib.memhip_conv2d_nhwc_f32(ptr(x_pad), ptr(weight), ptr(bias), ptr(add), ptr(out), B, H, W, Cin, Cout, ksize,
                                         stride, pad, int(relu), int(out_padded), stream_ptr()), "conv2d_nhwc_f32")
        return
    check(lib.memhip_conv2d_nhwc_bf16(ptr(x_pad), ptr(weight), ptr(bias), ptr(add), ptr(out), B, H, W, Cin, Cout, ksize,
                                      stride, pad, int(relu), int(out_padded), stream_ptr()), "conv2d_nhwc_bf16")


def nchw_to_padded_nhwc4(x, out, mean=None, std=None):
    B, Cc, H, W = x.shape
    if out.dtype == torch.float32:
        check(lib.memhip_nchw_to_padded_nhwc4_f32(ptr(x), B, Cc, H, W, ptr(mean), ptr(std), ptr(out), stream_ptr()),
              "nchw_to_padded_nhwc4_f32")
        return
    check(lib.memhip_nchw_to_padded_nhwc4(ptr(x), B, Cc, H, W, ptr(mean), ptr(std), ptr(out), stream_ptr()),
          "nchw_to_padded_nhwc4")


def argmax_rows(logits, M, N, ids, gap=None, rms=None, n_samples=None, rows_per_sample=0):
    if logits.dtype == torch.float32:
        if rms is not None or n_samples is not None:
            check(lib.memhip_argmax_rows_f32_ex(ptr(logits), logits.stride(0), M, N, ptr(ids), ptr(gap), ptr(rms), ptr(n_samples),
                                                rows_per_sample, stream_ptr()), "argmax_rows_f32_ex")
            return
        check(lib.memhip_argmax_rows_f32(ptr(logits), logits.stride(0), M, N, ptr(ids), ptr(gap), stream_ptr()),
              "argmax_rows_f32")
        return
    check(lib.memhip_argmax_rows_bf16(ptr(logits), logits.stride(0), M, N, ptr(ids), stream_ptr()), "argmax_rows_bf16")


def tok_flag_samples(gap, rms, B, hw, kappa, lst, count, stats=None):
    check(lib.memhip_tok_flag_samples(ptr(gap), ptr(rms), B, hw, float(kappa), ptr(lst), ptr(count), ptr(stats), stream_ptr()),
          "tok_flag_samples")


def tok_gather_images(x, mean, std, lst, count, offset, R, out, n_round):
    _, Cc, H, W = x.shape
    check(lib.memhip_tok_gather_images_f32(ptr(x), Cc, H, W, ptr(mean), ptr(std), ptr(lst), ptr(count), offset, R, ptr(out),
                                           ptr(n_round), stream_ptr()), "tok_gather_images_f32")


def tok_scatter_ids(ids_in, lst, n_round, offset, R, hw, ids_out):
    check(lib.memhip_tok_scatter_ids(ptr(ids_in), ptr(lst), ptr(n_round), offset, R, hw, ptr(ids_out), stream_ptr()),
          "tok_scatter_ids")


def embed_bwd(dx, mask_u8, B, L, D, dy, dcls, dmask_token):
    check(lib.memhip_embed_bwd(ptr(dx), dx.stride(0), ptr(mask_u8), B, L, D, ptr(dy), dy.stride(0), ptr(dcls),
                               ptr(dmask_token), stream_ptr()), "embed_bwd")


def cross_entropy(logits, labels, M, V, grad_scale, row_loss, row_correct, out2, write_grad=True):
    check(lib.memhip_cross_entropy(ptr(logits), logits.stride(0), ptr(labels), M, V, grad_scale, ptr(row_loss),
                                   ptr(row_correct), int(write_grad), ptr(out2), stream_ptr()), "cross_entropy")


def attn_tokens_padded(T):
    return lib.memhip_attn_tokens_padded(T)


def relpos_gather(table, index_i32, T, TP, heads, bias_pad, biasT_pad=None):
    check(lib.memhip_relpos_gather(ptr(table), ptr(index_i32), T, TP, heads, ptr(bias_pad), ptr(biasT_pad),
                                   stream_ptr()), "relpos_gather")


def _timed(code, flops, fn):
    """Run fn() between two HIP events on the launch stream when the per-launch timer is armed (bench.py)."""
    if GEMM_TIMER is None:
        fn()
        return
    e0, e1 = _timer_event(), _timer_event()
    e0.record()
    fn()
    e1.record()
    GEMM_TIMER.append((e0, e1, flops, code))


def attn_fwd(qkv, B, T, D, heads, table, window, out, lse):
    _timed(200, 4.0 * B * T * T * D, lambda: check(
        lib.memhip_attn_fwd(ptr(qkv), qkv.stride(0), B, T, D, heads, ptr(table), window[0], window[1], ptr(out),
                            out.stride(0), ptr(lse), stream_ptr()), "attn_fwd"))


def attn_delta(dout, out, rows, heads, delta):
    check(lib.memhip_attn_delta(ptr(dout), ptr(out), out.stride(0), rows, heads, ptr(delta), stream_ptr()),
          "attn_delta")


def attn_bwd_workspace(B, T, heads, window):
    """Bytes of scratch the dS-storing backward of the long-window kernels wants for this shape (0: no such form)."""
    return int(lib.memhip_attn_bwd_workspace(B, T, heads, window[0], window[1]))


def attn_bwd(qkv, dout, lse, delta, table, window, B, T, D, heads, scale, dqkv, dtable, dq_bias=None, dv_bias=None, out=None, ws=None):
    """out = the forward output: rowsum(dout * out) is computed by the library (inside the fused 14 x 14 kernel when it
    applies); without it `delta` must have been filled by attn_delta.  ws = a uint8 scratch tensor of at least
    attn_bwd_workspace(...) bytes: the long-window backward then stores dS instead of computing it twice."""
    wsp, wsn = (ptr(ws), ws.numel() * ws.element_size()) if ws is not None else (None, 0)
    if out is not None:
        _timed(201, 10.0 * B * T * T * D, lambda: check(
            lib.memhip_attn_bwd_out_ws(ptr(qkv), qkv.stride(0), ptr(dout), dout.stride(0), ptr(out), out.stride(0), ptr(lse),
                                       ptr(delta), ptr(table), window[0], window[1], B, T, D, heads, scale, ptr(dqkv),
                                       dqkv.stride(0), ptr(dtable), ptr(dq_bias), ptr(dv_bias), wsp, wsn, stream_ptr()), "attn_bwd_out"))
        return
    _timed(201, 10.0 * B * T * T * D, lambda: check(
        lib.memhip_attn_bwd_ws(ptr(qkv), qkv.stride(0), ptr(dout), dout.stride(0), ptr(lse), ptr(delta), ptr(table),
                               window[0], window[1], B, T, D, heads, scale, ptr(dqkv), dqkv.stride(0), ptr(dtable),
                               ptr(dq_bias), ptr(dv_bias), wsp, wsn, stream_ptr()), "attn_bwd"))


def residual_rows(x, rows_i32, y, gamma, rowkeep, keep_prob, R, D, out):
    """out[i] = x[rows[i]] + drop_path(gamma * y[i]) (the RESIDUAL epilogue's arithmetic on compact rows)."""
    check(lib.memhip_residual_rows(ptr(x), x.stride(0), ptr(rows_i32), ptr(y), y.stride(0), ptr(gamma), ptr(rowkeep), keep_prob,
                                   R, D, ptr(out), out.stride(0), stream_ptr()), "residual_rows")


def scatter_rows(src, rows_i32, R, D, dst):
    check(lib.memhip_scatter_rows_f32(ptr(src), src.stride(0), ptr(rows_i32), R, D, ptr(dst), dst.stride(0), stream_ptr()),
          "scatter_rows")


def copy_samples(src, dst, ids_i32, n, n_per_sample):
    check(lib.memhip_copy_samples_f32(ptr(src), ptr(dst), ptr(ids_i32), n, n_per_sample, stream_ptr()), "copy_samples")


def stream_reserve_cus(stream, cus):
    """Launches on `stream` (a torch.cuda.Stream) size their persistent grids for `cus` CUs fewer (0 clears it)."""
    check(lib.memhip_stream_reserve_cus(C.c_void_p(stream.cuda_stream), int(cus)), "stream_reserve_cus")


def zero_(t):
    """t.zero_() by the library's fill kernel (t contiguous, 16-byte aligned, a multiple of 16 bytes)."""
    nb = t.numel() * t.element_size()
    assert t.is_contiguous() and nb % 16 == 0
    check(lib.memhip_zero(ptr(t), nb, stream_ptr()), "zero")


def zero_ranges(base, ranges_dev, n, total_bytes):
    check(lib.memhip_zero_ranges(ptr(base), ptr(ranges_dev), n, total_bytes, stream_ptr()), "zero_ranges")


def cast_f32_bf16(src, dst, n):
    check(lib.memhip_cast_f32_bf16(ptr(src), ptr(dst), n, stream_ptr()), "cast")


def transpose_cast(src_f32, R, Cc, dst_bf16, ldout=None):
    check(lib.memhip_transpose_cast_f32_bf16(ptr(src_f32), src_f32.stride(0), R, Cc, ptr(dst_bf16),
                                             dst_bf16.stride(0) if ldout is None else ldout, stream_ptr()),
          "transpose_cast")


def transpose_cast_batched(desc_dev, prefix_dev, n, total_tiles):
    check(lib.memhip_transpose_cast_batched(ptr(desc_dev), ptr(prefix_dev), n, total_tiles, stream_ptr()),
          "transpose_cast_batched")


def transpose_bf16(src, R, Cc, dst, R_pad, colsum0=None, c0=(0, 0), colsum1=None, c1=(0, 0)):
    check(lib.memhip_transpose_bf16(ptr(src), src.stride(0), R, Cc, ptr(dst), dst.stride(0), R_pad, ptr(colsum0),
                                    c0[0], c0[1], ptr(colsum1), c1[0], c1[1], stream_ptr()), "transpose_bf16")


def im2col(x, B, Cc, H, W, ph, pw, out):
    check(lib.memhip_im2col_bf16(ptr(x), B, Cc, H, W, ph, pw, ptr(out), stream_ptr()), "im2col")


def fill_cls(x, B, T, D, cls):
    check(lib.memhip_fill_cls(ptr(x), x.stride(0), B, T, D, ptr(cls), stream_ptr()), "fill_cls")


def grad_norm(g, n, norm_out, ws):
    check(lib.memhip_grad_norm(ptr(g), n, ptr(norm_out), ptr(ws), ws.numel() * ws.element_size(), stream_ptr()),
          "grad_norm")


def adamw(p, g, m, v, n, wd_flags, lr, beta1, beta2, eps, wd, step, gnorm=None, max_norm=0.0):
    check(lib.memhip_adamw(ptr(p), ptr(g), ptr(m), ptr(v), n, ptr(wd_flags), lr, beta1, beta2, eps, wd, step,
                           ptr(gnorm), max_norm if max_norm else 0.0, stream_ptr()), "adamw")


def adamw_groups(p, g, m, v, n, group_of_chunk, group_table, n_groups, beta1, beta2, eps, step, gnorm=None, max_norm=0.0):
    check(lib.memhip_adamw_groups(ptr(p), ptr(g), ptr(m), ptr(v), n, ptr(group_of_chunk), ptr(group_table), n_groups, beta1,
                                  beta2, eps, step, ptr(gnorm), max_norm if max_norm else 0.0, stream_ptr()), "adamw_groups")


declare({
    "memhip_gemm_bf16_tn": (i32, [vp, i64, vp, i64, i32, i32, i32, vp, i64, i32, vp]),
    "memhip_gemm_bf16_tn_ws": (i32, [vp, i64, vp, i64, i32, i32, i32, vp, i64, i32, vp, sz, vp]),
    "memhip_gemm_bf16_tn_workspace": (sz, [i32, i32, i32]),
    "memhip_gemm_bf16_tn_group_workspace": (sz, [vp, i32]),
    "memhip_gemm_bf16_tn_group": (i32, [vp, i32, i32, vp, sz, vp]),
    "memhip_colsum_bf16": (i32, [vp, i64, i32, i32, vp, vp]),
    "memhip_colsum_fold": (i32, [vp, i32, i32, vp, vp]),
    "memhip_residual_rows": (i32, [vp, i64, vp, vp, i64, vp, vp, f32, i32, i32, vp, i64, vp]),
    "memhip_scatter_rows_f32": (i32, [vp, i64, vp, i32, i32, vp, i64, vp]),
})


_TN_WS_CACHE = {}       # (the library plans up to 32 split counts per query, and the engine asks in front of every weight-gradient launch;
                        # the answer depends on the shape only: it is sized for every CU of the device whatever a stream has reserved)


def gemm_tn_workspace(R, N, K):
    key = (R, N, K)
    v = _TN_WS_CACHE.get(key)
    if v is None:
        v = _TN_WS_CACHE[key] = int(lib.memhip_gemm_bf16_tn_workspace(R, N, K))
    return v


def gemm_tn(A, B, R, N, K, out, accumulate=True, workspace=None):
    """out[N,K] (+)= A[R,N]^T @ B[R,K]  (weight gradient; A = dY, B = X, token-major bf16).  `workspace`:
    optional uint8 scratch tensor of >= gemm_tn_workspace(R,N,K) bytes (plain-store partial tiles)."""
    ws, wsb = (ptr(workspace), workspace.numel() * workspace.element_size()) if workspace is not None else (None, 0)
    if GEMM_TIMER is None:
        check(lib.memhip_gemm_bf16_tn_ws(ptr(A), A.stride(0), ptr(B), B.stride(0), R, N, K, ptr(out), out.stride(0),
                                         int(accumulate), ws, wsb, stream_ptr()), "gemm_bf16_tn")
    else:
        e0, e1 = _timer_event(), _timer_event()
        e0.record()
        check(lib.memhip_gemm_bf16_tn_ws(ptr(A), A.stride(0), ptr(B), B.stride(0), R, N, K, ptr(out), out.stride(0),
                                         int(accumulate), ws, wsb, stream_ptr()), "gemm_bf16_tn")
        e1.record()
        GEMM_TIMER.append((e0, e1, 2.0 * R * N * K, 100))


class TnProblem(C.Structure):
    """memhip_tn_problem_t (include/memhip.h)"""
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("B", C.c_void_p), ("ldb", C.c_int64), ("out", C.c_void_p),
                ("ldo", C.c_int64), ("R", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("reserved0", C.c_int32)]


def _tn_problems(problems):
    arr = (TnProblem * len(problems))()
    for q, (A, B, R, N, K, out) in zip(arr, problems):
        q.A, q.lda, q.B, q.ldb, q.out, q.ldo = A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), out.data_ptr(), out.stride(0)
        q.R, q.N, q.K, q.reserved0 = R, N, K, 0
    return arr


def gemm_tn_group_workspace(shapes):
    """bytes of scratch for gemm_tn_group over products of the shapes [(R, N, K), ...]"""
    key = tuple(shapes)
    if key in _TN_WS_CACHE:
        return _TN_WS_CACHE[key]
    arr = (TnProblem * len(shapes))()
    for q, (R, N, K) in zip(arr, shapes):
        q.R, q.N, q.K, q.lda, q.ldb, q.ldo = R, N, K, N, K, K
    v = _TN_WS_CACHE[key] = int(lib.memhip_gemm_bf16_tn_group_workspace(arr, len(shapes)))
    return v


def gemm_tn_group(problems, accumulate=True, workspace=None):
    """The weight gradients [(A = dY, B = X, R, N, K, out f32 [N, K]), ...] of up to four layers as ONE launch
    (memhip_gemm_bf16_tn_group); each product has the contract of gemm_tn."""
    ws, wsb = (ptr(workspace), workspace.numel() * workspace.element_size()) if workspace is not None else (None, 0)
    arr = _tn_problems(problems)
    if GEMM_TIMER is None:
        check(lib.memhip_gemm_bf16_tn_group(arr, len(problems), int(accumulate), ws, wsb, stream_ptr()), "gemm_bf16_tn_group")
    else:
        e0, e1 = _timer_event(), _timer_event()
        e0.record()
        check(lib.memhip_gemm_bf16_tn_group(arr, len(problems), int(accumulate), ws, wsb, stream_ptr()), "gemm_bf16_tn_group")
        e1.record()
        GEMM_TIMER.append((e0, e1, sum(2.0 * R * N * K for _, _, R, N, K, _ in problems), 100 + len(problems)))   # 102..104: a group


def colsum_fold(ws, copies, N, out):
    """out[N] += the `copies` accumulator copies in ws (a GEMM with colsum_copies > 1 filled them); ws is zeroed again."""
    check(lib.memhip_colsum_fold(ptr(ws), copies, N, ptr(out), stream_ptr()), "colsum_fold")


def colsum_bf16(x, R, Cc, out):
    check(lib.memhip_colsum_bf16(ptr(x), x.stride(0), R, Cc, ptr(out), stream_ptr()), "colsum_bf16")


# ---------------------------------------------------------------- fp32 parity mode (csrc/fp32_path.hip)
declare({
    "memhip_f32_gemm_nt": (i32, [C.POINTER(GemmArgs), vp]),
    "memhip_f32_transpose": (i32, [vp, i64, i32, i32, vp, i64, vp]),
    "memhip_f32_layernorm_fwd": (i32, [vp, i64, vp, i32, i32, vp, vp, f32, vp, i64, vp, vp, vp]),
    "memhip_f32_layernorm_bwd": (i32, [vp, i64, vp, i64, vp, i32, i32, vp, vp, vp, vp, i64, i32, vp, vp, vp]),
    "memhip_f32_branch_bwd": (i32, [vp, i64, vp, i64, vp, vp, f32, i32, i32, i32, vp, i64, vp, vp, vp]),
    "memhip_f32_embed_bwd": (i32, [vp, i64, vp, i32, i32, i32, vp, i64, vp, vp, vp]),
    "memhip_f32_cross_entropy": (i32, [vp, i64, vp, i32, i32, f32, vp, vp, i32, vp, vp]),
    "memhip_f32_colsum": (i32, [vp, i64, i32, i32, vp, vp]),
    "memhip_f32_im2col": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "memhip_f32_attn_fwd": (i32, [vp, i64, i32, i32, i32, i32, vp, vp, vp, i64, vp, vp]),
    "memhip_f32_attn_bwd": (i32, [vp, i64, vp, i64, i32, i32, i32, i32, f32, vp, vp, vp, i64, vp, vp]),
})


def f32_gemm_nt(A, B, M, N, K, epi, out0=None, out1=None, bias=None, vec1=None, resid=None, aux=None, rowmask=None,
                keep_prob=1.0, colscale=1.0, colscale_n=0, rows_per_sample=1, accumulate=False, colsum=None, ldaux=None):
    """fp32 twin of gemm_nt (A, B, out0, out1, DGELU aux are fp32)."""
    a = GemmArgs()
    a.A, a.B, a.lda, a.ldb = _p(A), _p(B), A.stride(0), B.stride(0)
    a.M, a.N, a.K, a.epilogue = M, N, K, epi
    a.out0, a.ldo0 = _p(out0), (out0.stride(0) if out0 is not None else 0)
    a.out1, a.ldo1 = _p(out1), (out1.stride(0) if out1 is not None else 0)
    a.bias, a.vec1, a.resid = _p(bias), _p(vec1), _p(resid)
    a.ldr = resid.stride(0) if resid is not None else 0
    a.aux = _p(aux)
    a.ldaux = (aux.stride(0) if (aux is not None and aux.dim() > 1) else 0) if ldaux is None else ldaux
    a.rowmask = _p(rowmask)
    a.keep_prob, a.colscale, a.colscale_n = keep_prob, colscale, colscale_n
    a.rows_per_sample, a.accumulate = rows_per_sample, int(accumulate)
    a.colsum = _p(colsum)
    check(lib.memhip_f32_gemm_nt(C.byref(a), stream_ptr()), "f32_gemm_nt")


def f32_transpose(src, R, Cc, dst):
    check(lib.memhip_f32_transpose(ptr(src), src.stride(0), R, Cc, ptr(dst), dst.stride(0), stream_ptr()), "f32_transpose")


def f32_layernorm_fwd(x, gamma, beta, y, mean, rstd, R, D, eps=1e-6, row_idx=None):
    check(lib.memhip_f32_layernorm_fwd(ptr(x), x.stride(0), ptr(row_idx), R, D, ptr(gamma), ptr(beta), eps, ptr(y),
                                       y.stride(0), ptr(mean), ptr(rstd), stream_ptr()), "f32_layernorm_fwd")


def f32_layernorm_bwd(dy, x, gamma, mean, rstd, dres, dgamma, dbeta, R, D, accumulate=True, row_idx=None):
    check(lib.memhip_f32_layernorm_bwd(ptr(dy), dy.stride(0), ptr(x), x.stride(0), ptr(row_idx), R, D, ptr(gamma), ptr(mean),
                                       ptr(rstd), ptr(dres), dres.stride(0), int(accumulate), ptr(dgamma), ptr(dbeta),
                                       stream_ptr()), "f32_layernorm_bwd")


def f32_branch_bwd(dx, y, gamma, dy, dgamma, dbias, M, D, rowmask=None, keep_prob=1.0, rows_per_sample=1):
    check(lib.memhip_f32_branch_bwd(ptr(dx), dx.stride(0), ptr(y), y.stride(0) if y is not None else 0, ptr(gamma), ptr(rowmask),
                                    keep_prob, rows_per_sample, M, D, ptr(dy), dy.stride(0), ptr(dgamma), ptr(dbias),
                                    stream_ptr()), "f32_branch_bwd")


def f32_embed_bwd(dx, mask_u8, B, L, D, dy, dcls, dmask_token):
    check(lib.memhip_f32_embed_bwd(ptr(dx), dx.stride(0), ptr(mask_u8), B, L, D, ptr(dy), dy.stride(0), ptr(dcls),
                                   ptr(dmask_token), stream_ptr()), "f32_embed_bwd")


def f32_cross_entropy(logits, labels, M, V, grad_scale, row_loss, row_correct, out2, write_grad=True):
    check(lib.memhip_f32_cross_entropy(ptr(logits), logits.stride(0), ptr(labels), M, V, grad_scale, ptr(row_loss),
                                       ptr(row_correct), int(write_grad), ptr(out2), stream_ptr()), "f32_cross_entropy")


def f32_colsum(x, R, Cc, out):
    check(lib.memhip_f32_colsum(ptr(x), x.stride(0), R, Cc, ptr(out), stream_ptr()), "f32_colsum")


def f32_im2col(x, B, Cc, H, W, ph, pw, out):
    check(lib.memhip_f32_im2col(ptr(x), B, Cc, H, W, ph, pw, ptr(out), stream_ptr()), "f32_im2col")


def f32_attn_fwd(qkv, B, T, D, heads, table, index, out, lse=None):
    check(lib.memhip_f32_attn_fwd(ptr(qkv), qkv.stride(0), B, T, D, heads, ptr(table), ptr(index), ptr(out), out.stride(0),
                                  ptr(lse), stream_ptr()), "f32_attn_fwd")


def f32_attn_bwd(qkv, dout, B, T, D, heads, scale, table, index, dqkv, dtable):
    check(lib.memhip_f32_attn_bwd(ptr(qkv), qkv.stride(0), ptr(dout), dout.stride(0), B, T, D, heads, scale, ptr(table),
                                  ptr(index), ptr(dqkv), dqkv.stride(0), ptr(dtable), stream_ptr()), "f32_attn_bwd")


# ---------------------------------------------------------------- MAE plumbing (csrc/fp32_path.hip)
declare({
    "memhip_mae_enc_assemble": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "memhip_mae_enc_assemble_bwd": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp]),
    "memhip_mae_dec_assemble": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "memhip_mae_dec_assemble_bwd": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp]),
    "memhip_mae_loss": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
})


def mae_enc_assemble(xe, pos, cls, ids_keep, B, L, K, D, out):
    check(lib.memhip_mae_enc_assemble(ptr(xe), ptr(pos), ptr(cls), ptr(ids_keep), B, L, K, D, ptr(out), stream_ptr()), "mae_enc_assemble")


def mae_enc_assemble_bwd(dx, ids_keep, B, L, K, D, dxe, dcls):
    check(lib.memhip_mae_enc_assemble_bwd(ptr(dx), ptr(ids_keep), B, L, K, D, ptr(dxe), ptr(dcls), stream_ptr()), "mae_enc_assemble_bwd")


def mae_dec_assemble(y, mask_token, dpos, ids_restore, B, L, K, D, out):
    check(lib.memhip_mae_dec_assemble(ptr(y), ptr(mask_token), ptr(dpos), ptr(ids_restore), B, L, K, D, ptr(out), stream_ptr()),
          "mae_dec_assemble")


def mae_dec_assemble_bwd(dxd, ids_restore, B, L, K, D, dy, dmask_token):
    check(lib.memhip_mae_dec_assemble_bwd(ptr(dxd), ptr(ids_restore), B, L, K, D, ptr(dy), ptr(dmask_token), stream_ptr()),
          "mae_dec_assemble_bwd")


def mae_loss(pred, img, mask, B, Cc, H, W, patch, only_masked, row_loss, dpred, scratch2):
    check(lib.memhip_mae_loss(ptr(pred), ptr(img), ptr(mask), B, Cc, H, W, patch, int(only_masked), ptr(row_loss), ptr(dpred),
                              ptr(scratch2), stream_ptr()), "mae_loss")


# ---------------------------------------------------------------- fp16 x 2 tokenizer mode (csrc/conv_f16x2.hip)
declare({
    "memhip_conv2d_nhwc_f16x2": (i32, [vp, i64, vp, i64, vp, vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32,
                                       i32, vp]),
    "memhip_nchw_to_padded_nhwc4_f16x2": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, i64, vp]),
})


def conv2d_nhwc_f16x2(x2, w2, bias, out, B, H, W, Cin, Cout, ksize, stride, pad, relu=False, add2=None, out_padded=True):
    """x2 fp16 [2,B,H+2,W+2,Cin] (hi / lo planes), w2 fp16 [2,Cout,K] -> out fp16 [2,B,Ho+2,Wo+2,Cout] interior, or a dense
    fp32 [B*Ho*Wo, Cout] matrix when `out` is float32."""
    out_f32 = out.dtype == torch.float32
    check(lib.memhip_conv2d_nhwc_f16x2(ptr(x2), x2.stride(0), ptr(w2), w2.stride(0), ptr(bias), ptr(add2),
                                       add2.stride(0) if add2 is not None else 0, ptr(out), 0 if out_f32 else out.stride(0),
                                       B, H, W, Cin, Cout, ksize, stride, pad, int(relu), int(out_padded and not out_f32),
                                       int(out_f32), stream_ptr()), "conv2d_nhwc_f16x2")


def nchw_to_padded_nhwc4_f16x2(x, out2, mean=None, std=None):
    B, Cc, H, W = x.shape
    check(lib.memhip_nchw_to_padded_nhwc4_f16x2(ptr(x), B, Cc, H, W, ptr(mean), ptr(std), ptr(out2), out2.stride(0), stream_ptr()),
          "nchw_to_padded_nhwc4_f16x2")
