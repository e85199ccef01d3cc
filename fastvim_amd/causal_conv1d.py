"""``causal_conv1d_fn``: the op surface of PyPI causal-conv1d 1.1.3.post1 that the reference imports
(``from causal_conv1d import causal_conv1d_fn``; mamba_ssm/modules/mamba_simple_faster.py:17-20,
called at :274-285 and from the fused autograd functions,
mamba_ssm/ops/selective_scan_interface.py:496-498, 640-642, 751-753), on the gfx950 HIP kernels of
csrc/conv_bdl.hip through the C ABI (fv_causal_conv1d_fwd / _bwd).

x: (batch, dim, seqlen); weight: (dim, width), width in 2..4; bias: (dim,) or None;
activation in {None, "silu", "swish"}; returns (batch, dim, seqlen) in x's dtype.  Differentiable
w.r.t. x, weight and bias; gradients are deterministic (fixed-order reductions).
"""
import torch

from . import _lib as L
from .mixer_ops import reduce_partials


def _conv_fwd(x, w32, b32, silu):
    """The forward kernel launch: x (B, D, L) contiguous, w32 (D, W) / b32 (D,) fp32."""
    B, D, Lq = x.shape
    y = torch.empty_like(x)
    rc = L.lib().fv_causal_conv1d_fwd(L.ptr(x), L.ptr(w32), L.ptr(b32), L.ptr(y), L.i32(B), L.i32(D), L.i32(Lq),
                                      L.i32(w32.shape[1]), L.i32(silu), L.i32(L.dtype_code(x.dtype)), L.stream_of(x))
    L.check(rc, "causal_conv1d_fwd")
    return y


def _conv_bwd(x, w32, b32, dy, silu):
    """The backward kernel launch: (dx in x's dtype, dw (D, W) fp32, db (D,) fp32), the weight / bias sums over the batch
    in fixed order."""
    B, D, Lq = x.shape
    W = w32.shape[1]
    dy = dy.to(x.dtype)
    if dy.stride(2) != 1 or dy.stride(1) != Lq or dy.stride(0) != D * Lq:
        dy = dy.contiguous()
    dx = torch.empty_like(x)
    part = torch.empty(B, D, 5, device=x.device, dtype=torch.float32)
    rc = L.lib().fv_causal_conv1d_bwd(L.ptr(x), L.ptr(w32), L.ptr(b32), L.ptr(dy), L.ptr(dx), L.ptr(part), L.i32(B),
                                      L.i32(D), L.i32(Lq), L.i32(W), L.i32(silu), L.i32(L.dtype_code(x.dtype)),
                                      L.stream_of(x))
    L.check(rc, "causal_conv1d_bwd")
    red = reduce_partials(part.view(B, D * 5), B).view(D, 5)        # fixed-order sum over the batch
    return dx, red[:, 4 - W:4], red[:, 4]


class CausalConv1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias=None, seq_idx=None, activation=None):
        if activation not in (None, "silu", "swish"):
            raise NotImplementedError("activation must be None, silu, or swish")
        if seq_idx is not None:
            raise NotImplementedError("causal_conv1d_fn: seq_idx is not used by the FastVim path")
        L.require_gpu(x)
        if x.dim() != 3 or weight.dim() != 2 or weight.shape[0] != x.shape[1]:
            raise RuntimeError(f"causal_conv1d_fn: x (B, D, L) / weight (D, W) expected, got {tuple(x.shape)} / {tuple(weight.shape)}")
        if x.stride(2) != 1 or x.stride(1) != x.shape[2]:
            x = x.contiguous()
        w32 = weight.detach().float().contiguous()
        b32 = bias.detach().float().contiguous() if bias is not None else None
        silu = activation in ("silu", "swish")
        y = _conv_fwd(x, w32, b32, silu)
        ctx.save_for_backward(x, w32, b32 if b32 is not None else x.new_empty(0))
        ctx.silu, ctx.has_bias = silu, bias is not None
        ctx.w_dtype = weight.dtype
        ctx.b_dtype = bias.dtype if bias is not None else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w32, b32 = ctx.saved_tensors
        dx, dw, db = _conv_bwd(x, w32, b32 if ctx.has_bias else None, dy, ctx.silu)
        return dx, dw.to(ctx.w_dtype), db.to(ctx.b_dtype) if ctx.has_bias else None, None, None


def causal_conv1d_fn(x, weight, bias=None, seq_idx=None, activation=None):
    return CausalConv1dFn.apply(x, weight, bias, seq_idx, activation)
