"""Fused residual-add + RMSNorm/LayerNorm oracle (test infrastructure; see oracle/__init__.py).

Follows the reference's Triton op contract:
* kernel math ``r = x + residual`` in fp32, ``y = r * rstd * w (+ b)`` or
  ``(r - mean) * rstd * w + b`` with ``rstd = 1/sqrt(var + eps)``
  (mamba-1p1p1/mamba_ssm/ops/triton/layernorm.py:96-121);
* dtype rules of ``LayerNormFn.forward`` (layernorm.py:415-450): ``residual_out``
  is stored in ``residual.dtype`` if a residual is given, else fp32 when
  ``residual_in_fp32``, else ``x.dtype``; ``y`` is stored in ``x.dtype``;
* pure references ``rms_norm_ref`` / ``layer_norm_ref`` (layernorm.py:18-49).
"""
import torch


def fused_add_norm_oracle(x, weight, bias=None, residual=None, eps=1e-6, prenorm=False,
                          residual_in_fp32=False, is_rms_norm=True, row_scale=None,
                          compute_dtype=torch.float64):
    """``row_scale`` (B,) optionally scales ``x`` per sample before the add: this is
    DropPath applied to the mixer output (models/fastvim.py:182-190, timm DropPath)."""
    cd = compute_dtype
    xf = x.to(cd)
    if row_scale is not None:
        xf = xf * row_scale.to(cd).view(-1, *([1] * (x.dim() - 1)))
    r = xf + residual.to(cd) if residual is not None else xf
    res_dtype = (residual.dtype if residual is not None
                 else (torch.float32 if residual_in_fp32 else x.dtype))
    if is_rms_norm:
        rstd = torch.rsqrt(r.square().mean(-1, keepdim=True) + eps)
        y = r * rstd * weight.to(cd)
    else:
        mu = r.mean(-1, keepdim=True)
        rstd = torch.rsqrt((r - mu).square().mean(-1, keepdim=True) + eps)
        y = (r - mu) * rstd * weight.to(cd)
    if bias is not None:
        y = y + bias.to(cd)
    y = y.to(x.dtype)
    return (y, r.to(res_dtype)) if prenorm else y
