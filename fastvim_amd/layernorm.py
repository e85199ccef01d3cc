"""Fused residual-add + RMSNorm/LayerNorm operator API: mirror of
mamba-1p1p1/mamba_ssm/ops/triton/layernorm.py:402-536 (``LayerNormFn``, ``layer_norm_fn``,
``rms_norm_fn``, ``RMSNorm``) on the HIP kernels of csrc/norm.hip.

Same signatures and dtype rules (``y`` in ``x.dtype``; ``residual_out`` in ``residual.dtype``, or
fp32 when ``residual_in_fp32``, else ``x.dtype``).  Two keyword-only extensions used by
``fastvim_amd.fastvim.Block``: ``row_scale`` (B,) folds timm DropPath's per-sample scale of ``x``
into the add, ``out_dtype`` folds the cast to the mixer's compute dtype into the store.
"""
import ctypes

import torch

from . import _lib as L
from .mixer_ops import reduce_partials


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False,
                is_rms_norm=False, row_scale=None, out_dtype=None):
        L.require_gpu(x, weight, residual)
        x_shape_og = x.shape
        N = x.shape[-1]
        x2 = x.reshape(-1, N).contiguous()
        M = x2.shape[0]
        res2 = None
        if residual is not None:
            if residual.shape != x_shape_og:
                raise RuntimeError("layer_norm_fn: residual must have x's shape")
            res2 = residual.reshape(-1, N).contiguous()
        w_param = weight
        weight = weight.float().contiguous()
        bias = bias.float().contiguous() if bias is not None else None
        res_dtype = residual.dtype if residual is not None else (torch.float32 if residual_in_fp32 else x.dtype)
        y_dtype = x.dtype if out_dtype is None else out_dtype
        y = torch.empty(M, N, device=x.device, dtype=y_dtype)
        # residual_out doubles as the saved normalisation input (layernorm.py:442)
        res_out = torch.empty(M, N, device=x.device, dtype=res_dtype)
        mean = None if is_rms_norm else torch.empty(M, device=x.device, dtype=torch.float32)
        rstd = torch.empty(M, device=x.device, dtype=torch.float32)
        rows_per_scale = 1
        if row_scale is not None:
            row_scale = row_scale.float().contiguous()
            if M % row_scale.numel():
                raise RuntimeError("layer_norm_fn: row_scale does not divide the row count")
            rows_per_scale = M // row_scale.numel()
        rc = L.lib().fv_add_norm_fwd(
            L.ptr(x2), L.i32(L.dtype_code(x2.dtype)), L.ptr(res2), L.i32(L.dtype_code(res_dtype) if res2 is not None else 0),
            L.ptr(weight), L.ptr(bias), L.ptr(row_scale), L.i32(rows_per_scale), L.ptr(y), L.i32(L.dtype_code(y_dtype)),
            L.ptr(res_out), L.i32(L.dtype_code(res_dtype)), L.ptr(mean), L.ptr(rstd), L.i32(M), L.i32(N),
            ctypes.c_float(eps), L.i32(is_rms_norm), L.stream_of(x2))
        L.check(rc, "add_norm_fwd")
        ctx.save_for_backward(res_out, weight, bias, mean, rstd, row_scale)
        ctx.x_shape_og = x_shape_og
        ctx.is_rms_norm = is_rms_norm
        ctx.has_residual = residual is not None
        ctx.prenorm = prenorm
        ctx.x_dtype = x.dtype
        ctx.res_in_dtype = residual.dtype if residual is not None else None
        ctx.rows_per_scale = rows_per_scale
        ctx.x_needs_grad = x.requires_grad
        ctx.w_direct = (w_param.grad if (getattr(w_param, "_fv_direct", False) and w_param.grad is not None
                                         and w_param.grad.is_contiguous() and w_param.dtype == torch.float32) else None)
        y = y.reshape(x_shape_og)
        return y if not prenorm else (y, res_out.reshape(x_shape_og))

    @staticmethod
    def backward(ctx, dy, *args):
        r, weight, bias, mean, rstd, row_scale = ctx.saved_tensors
        M, N = r.shape
        dy = dy.reshape(M, N).contiguous()
        dres_out = None
        if ctx.prenorm and args[0] is not None:
            dres_out = args[0].reshape(M, N).contiguous()
        dev = r.device
        dx = torch.empty(M, N, device=dev, dtype=ctx.x_dtype)
        dres_in = torch.empty(M, N, device=dev, dtype=ctx.res_in_dtype) if ctx.has_residual else None
        lib = L.lib()
        nb = lib.fv_add_norm_blocks(L.i32(M))
        pw = torch.empty(nb, N, device=dev, dtype=torch.float32)
        pb = torch.empty(nb, N, device=dev, dtype=torch.float32) if bias is not None else None
        rc = lib.fv_add_norm_bwd(
            L.ptr(dy), L.i32(L.dtype_code(dy.dtype)), L.ptr(dres_out),
            L.i32(L.dtype_code(dres_out.dtype) if dres_out is not None else 0), L.ptr(r), L.i32(L.dtype_code(r.dtype)),
            L.ptr(weight), L.ptr(mean), L.ptr(rstd), L.ptr(row_scale), L.i32(ctx.rows_per_scale), L.ptr(dx),
            L.i32(L.dtype_code(dx.dtype)), L.ptr(dres_in), L.i32(L.dtype_code(dres_in.dtype) if dres_in is not None else 0),
            L.ptr(pw), L.ptr(pb), L.i32(M), L.i32(N), L.i32(ctx.is_rms_norm), L.stream_of(r))
        L.check(rc, "add_norm_bwd")
        if ctx.w_direct is not None:
            reduce_partials(pw, nb, out=ctx.w_direct.view(-1), accumulate=True)
            dw = None
        else:
            dw = reduce_partials(pw, nb)
        db = reduce_partials(pb, nb) if pb is not None else None
        return (dx.reshape(ctx.x_shape_og), dw, db,
                dres_in.reshape(ctx.x_shape_og) if ctx.has_residual else None,
                None, None, None, None, None, None)


def layer_norm_fn(x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False,
                  is_rms_norm=False, *, row_scale=None, out_dtype=None):
    return LayerNormFn.apply(x, weight, bias, residual, eps, prenorm, residual_in_fp32, is_rms_norm,
                             row_scale, out_dtype)


def rms_norm_fn(x, weight, bias, residual=None, prenorm=False, residual_in_fp32=False, eps=1e-6, *,
                row_scale=None, out_dtype=None):
    return LayerNormFn.apply(x, weight, bias, residual, eps, prenorm, residual_in_fp32, True,
                             row_scale, out_dtype)


class RMSNorm(torch.nn.Module):
    def __init__(self, hidden_size, eps=1e-5, device=None, dtype=None):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        self.eps = eps
        self.weight = torch.nn.Parameter(torch.empty(hidden_size, **factory_kwargs))
        self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.ones_(self.weight)

    def forward(self, x, residual=None, prenorm=False, residual_in_fp32=False):
        return rms_norm_fn(x, self.weight, self.bias, residual=residual, eps=self.eps, prenorm=prenorm,
                           residual_in_fp32=residual_in_fp32)
