"""Training losses of the reference recipe on the HIP path.

``SoftTargetCrossEntropy`` is a drop-in for ``timm.loss.SoftTargetCrossEntropy`` as the reference trainer uses it under
mixup / label smoothing (imagenet_classification/supervised_imagenet.py:83, 109-115):
``loss = mean_b sum_c -target[b, c] * log_softmax(x[b])[c]``.  Value and gradient come from ONE fused launch
(csrc/loss.hip, C ABI ``fv_soft_target_ce``) instead of the eleven small kernels of the eager expression and its
autograd.  ``target`` is treated as a constant (no gradient), as in the reference's use.
"""
import torch
import torch.nn as nn

from . import _lib as L


class _SoftTargetCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target):
        L.require_gpu(x, target)
        if x.dim() != 2 or target.shape != x.shape:
            raise RuntimeError(f"SoftTargetCrossEntropy: logits {tuple(x.shape)} and target {tuple(target.shape)} must be (B, C)")
        B, C = x.shape
        xc = x.contiguous()
        if xc.dtype not in (torch.float32, torch.bfloat16):
            xc = xc.float()
        t = target.detach().float().contiguous()
        rows = torch.empty(B, device=x.device, dtype=torch.float32)
        loss = torch.empty(1, device=x.device, dtype=torch.float32)
        dx = torch.empty(B, C, device=x.device, dtype=torch.float32)
        rc = L.lib().fv_soft_target_ce(L.ptr(xc), L.i32(L.dtype_code(xc.dtype)), L.ptr(t), L.ptr(rows), L.ptr(loss), L.ptr(dx),
                                       L.i32(B), L.i32(C), L.stream_of(xc))
        L.check(rc, "soft_target_ce")
        ctx.save_for_backward(dx)
        ctx.x_dtype = x.dtype
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        if ctx.x_dtype in (torch.float32, torch.bfloat16) and g.dtype == torch.float32 and g.numel() == 1:
            from .glue_ops import scale_cast
            return scale_cast(dx, g, ctx.x_dtype), None           # scale by the upstream scalar and cast, one launch
        return (dx * g).to(ctx.x_dtype), None


class SoftTargetCrossEntropy(nn.Module):
    """``forward(x, target)``: x (B, C) logits (fp32 or bf16), target (B, C) soft labels -> scalar fp32 loss."""

    def forward(self, x, target):
        return _SoftTargetCEFn.apply(x, target)
