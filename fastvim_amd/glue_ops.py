"""Host wrappers of csrc/glue.hip: the small memory-bound ops around the backbone (patch unfold, patch projection with
the bias / position-embedding epilogue, token mean pool, the stochastic-depth table, loss-gradient scaling, bias
gradients).  Each is one HIP launch where the eager expression is a string of library kernels."""
import ctypes

import torch

from . import _lib as L
from .mixer_ops import reduce_partials


def patch_unfold(x, ph, pw, out_dtype):
    """(B, C, H, W) -> (B, gh*gw, C*ph*pw) in ``out_dtype``: the operand of the k == stride patch-embed conv as a GEMM
    (models/fastvim.py:95), unfolded and cast in one pass."""
    B, C, H, W = x.shape
    x = x.contiguous()
    out = torch.empty(B, (H // ph) * (W // pw), C * ph * pw, device=x.device, dtype=out_dtype)
    rc = L.lib().fv_patch_unfold(L.ptr(x), L.i32(L.dtype_code(x.dtype)), L.ptr(out), L.i32(L.dtype_code(out_dtype)),
                                 L.i32(B), L.i32(C), L.i32(H), L.i32(W), L.i32(ph), L.i32(pw), L.stream_of(x))
    L.check(rc, "patch_unfold")
    return out


def patch_unfold_ok(x, ph, pw):
    return (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16) and pw % 8 == 0
            and x.shape[2] % ph == 0 and x.shape[3] % pw == 0 and x.shape[1] * ph * pw * 16 * 4 <= 64 * 1024)


def gemm_rowbias(a, w, table):
    """fp32 (M, N) = bf16_round(a (M, K) @ w (N, K)^T) + table[m mod period]: bf16 operands, table (period, N) fp32."""
    M, K = a.shape
    N = w.shape[0]
    period = table.shape[0]
    c = torch.empty(M, N, device=a.device, dtype=torch.float32)
    rc = L.lib().fv_gemm_bf16_rowbias(L.ptr(a), L.ptr(w), L.ptr(c), L.ptr(table), L.i32(period), L.i32(M), L.i32(N), L.i32(K),
                                      ctypes.c_long(a.stride(0)), ctypes.c_long(w.stride(0)), ctypes.c_long(N), L.stream_of(a))
    L.check(rc, "gemm_bf16_rowbias")
    return c


class MeanPoolFn(torch.autograd.Function):
    """x (B, L, D) -> (B, D): ``x.mean(dim=1)`` (models/fastvim.py:529-531) and its adjoint, one launch each."""

    @staticmethod
    def forward(ctx, x):
        L.require_gpu(x)
        B, Ltok, D = x.shape
        xc = x.contiguous()
        out = torch.empty(B, D, device=x.device, dtype=x.dtype)
        rc = L.lib().fv_mean_pool_fwd(L.ptr(xc), L.ptr(out), L.i32(B), L.i32(Ltok), L.i32(D), L.i32(L.dtype_code(x.dtype)),
                                      L.stream_of(xc))
        L.check(rc, "mean_pool_fwd")
        ctx.shape = (B, Ltok, D)
        return out

    @staticmethod
    def backward(ctx, g):
        B, Ltok, D = ctx.shape
        g = g.contiguous()
        dx = torch.empty(B, Ltok, D, device=g.device, dtype=g.dtype)
        rc = L.lib().fv_mean_pool_bwd(L.ptr(g), L.ptr(dx), L.i32(B), L.i32(Ltok), L.i32(D), L.i32(L.dtype_code(g.dtype)),
                                      L.stream_of(g))
        L.check(rc, "mean_pool_bwd")
        return dx


def mean_pool_ok(x):
    return x.is_cuda and x.dim() == 3 and x.dtype in (torch.float32, torch.bfloat16) and x.shape[2] % 4 == 0


def droppath_table_(table, keep, inv_keep):
    """In place: U[0,1) draws (mods, batch) -> floor(keep + U) * inv_keep, one row of per-sample scales per module."""
    mods, batch = table.shape
    rc = L.lib().fv_droppath_table(L.ptr(table), L.ptr(keep), L.ptr(inv_keep), L.i32(mods), L.i32(batch), L.stream_of(table))
    L.check(rc, "droppath_table")
    return table


def scale_cast(x, scale, out_dtype):
    """x (fp32) * scale (0-dim / 1-element fp32 device tensor), cast to ``out_dtype``."""
    x = x.contiguous()
    y = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    rc = L.lib().fv_scale_cast(L.ptr(x), L.ptr(scale), L.ptr(y), L.i32(L.dtype_code(out_dtype)), ctypes.c_size_t(x.numel()),
                               L.stream_of(x))
    L.check(rc, "scale_cast")
    return y


def column_sum(x, out=None, accumulate=False):
    """Sum over the rows of a (rows, cols) fp32 / bf16 matrix -> (cols,) fp32 (fixed order); ``out`` + ``accumulate`` add
    into a gradient buffer."""
    rows, cols = x.shape
    x = x.contiguous()
    if out is None:
        out = torch.empty(cols, device=x.device, dtype=torch.float32)
        accumulate = False
    rc = L.lib().fv_column_sum(L.ptr(x), L.i32(L.dtype_code(x.dtype)), L.ptr(out), L.i32(rows), L.i32(cols), L.i32(accumulate),
                               L.stream_of(x))
    L.check(rc, "column_sum")
    return out
