"""Import the insitro/FastVim reference on CPU (build container only).

Tool used by ``gen_golden.py`` to capture golden vectors.  It reads
``/root/reference`` by path; neither this helper's targets nor the reference
travel to the GPU box -- only the ``.pt`` fixtures it produces do.

The reference imports CUDA extensions, timm and mm* at module import time.
Those are replaced by ``sys.modules`` stubs:

* ``causal_conv1d.causal_conv1d_fn`` -> the reference's own fallback formula
  ``F.conv1d(x, w[:, None, :], bias, padding=W-1, groups=D)[..., :L]`` + SiLU
  (reference: mamba-1p1p1/mamba_ssm/modules/mamba_simple.py:302-303).
* ``selective_scan_fn`` -> the reference's ``selective_scan_ref``
  (mamba-1p1p1/mamba_ssm/ops/selective_scan_interface.py:126-206).
* ``rms_norm_fn`` / ``layer_norm_fn`` -> the reference's ``rms_norm_ref`` /
  ``layer_norm_ref`` (mamba_ssm/ops/triton/layernorm.py:18-49) wrapped with the
  ``prenorm`` / ``residual_in_fp32`` handling of ``LayerNormFn.forward``
  (layernorm.py:402-450).
"""
import sys
import types

REF_ROOT = "/root/reference"

sys.dont_write_bytecode = True


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_loaded = None


def load_reference():
    """Returns a namespace with the reference modules, importing once."""
    global _loaded
    if _loaded is not None:
        return _loaded
    import torch
    import torch.nn as nn
    import torch.nn.functional as F

    # ---- CUDA extension stubs -------------------------------------------
    _mod("causal_conv1d_cuda")
    _mod("selective_scan_cuda")

    def causal_conv1d_fn(x, weight, bias=None, activation=None, seq_idx=None):
        assert activation in (None, "silu", "swish")
        D, W = weight.shape
        L = x.shape[-1]
        y = F.conv1d(x, weight[:, None, :], bias, padding=W - 1, groups=D)[..., :L]
        return F.silu(y) if activation is not None else y

    _mod("causal_conv1d", causal_conv1d_fn=causal_conv1d_fn, causal_conv1d_update=None)

    # ---- timm / mm* stubs -----------------------------------------------
    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0, scale_by_keep=True):
            super().__init__()
            self.drop_prob = drop_prob
            self.scale_by_keep = scale_by_keep

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            shape = (x.shape[0],) + (1,) * (x.ndim - 1)
            mask = x.new_empty(shape).bernoulli_(keep)
            if keep > 0.0 and self.scale_by_keep:
                mask.div_(keep)
            return x * mask

    def to_2tuple(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v, v)

    def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

    def lecun_normal_(t):
        fan_in = nn.init._calculate_fan_in_and_fan_out(t)[0]
        std = (1.0 / fan_in) ** 0.5 / 0.87962566103423978
        return nn.init.trunc_normal_(t, std=std, a=-2 * std, b=2 * std)

    timm = _mod("timm")
    timm.layers = _mod(
        "timm.layers",
        DropPath=DropPath,
        lecun_normal_=lecun_normal_,
        to_2tuple=to_2tuple,
        trunc_normal_=trunc_normal_,
    )
    timm.models = _mod("timm.models", register_model=lambda f: f)
    timm.models.layers = _mod(
        "timm.models.layers",
        DropPath=DropPath,
        lecun_normal_=lecun_normal_,
        to_2tuple=to_2tuple,
        trunc_normal_=trunc_normal_,
    )
    _mod("timm.models.registry", register_model=lambda f: f)
    _mod(
        "timm.models.vision_transformer",
        _cfg=lambda **kw: dict(kw),
        _load_weights=lambda *a, **k: None,
        VisionTransformer=object,
    )

    class _Registry:
        def register_module(self, *a, **k):
            return lambda cls: cls

    _mod("mmdet")
    _mod("mmdet.registry", MODELS=_Registry())
    _mod("mmseg")
    _mod("mmseg.models")
    _mod("mmseg.models.builder", BACKBONES=_Registry())

    for p in (REF_ROOT + "/mamba-1p1p1", REF_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)

    import mamba_ssm  # noqa: F401  (reference package)
    from mamba_ssm.ops import selective_scan_interface as ssi
    from mamba_ssm.ops.triton import layernorm as ln

    # route the CUDA op to the reference's own pure-PyTorch spec
    ssi.selective_scan_fn = ssi.selective_scan_ref

    def _fused_add_norm(x, weight, bias, residual, eps, prenorm, residual_in_fp32, is_rms):
        # LayerNormFn.forward dtype contract (layernorm.py:415-450): residual_out
        # takes residual's dtype, or fp32 if residual_in_fp32, else x's dtype;
        # y takes x's dtype; statistics in fp32.
        x_dtype = x.dtype
        res_dtype = (
            residual.dtype
            if residual is not None
            else (torch.float32 if residual_in_fp32 else x_dtype)
        )
        r = x.float() + (residual.float() if residual is not None else 0.0)
        res_out = r.to(res_dtype)
        ref = ln.rms_norm_ref if is_rms else ln.layer_norm_ref
        y = ref(r, weight, bias, eps=eps, upcast=True).to(x_dtype)
        return (y, res_out) if prenorm else y

    def rms_norm_fn(x, weight, bias, residual=None, prenorm=False, residual_in_fp32=False, eps=1e-6):
        return _fused_add_norm(x, weight, bias, residual, eps, prenorm, residual_in_fp32, True)

    def layer_norm_fn(x, weight, bias, residual=None, eps=1e-6, prenorm=False,
                      residual_in_fp32=False, is_rms_norm=False):
        return _fused_add_norm(x, weight, bias, residual, eps, prenorm, residual_in_fp32, is_rms_norm)

    class RMSNorm(ln.RMSNorm):
        def forward(self, x, residual=None, prenorm=False, residual_in_fp32=False):
            return rms_norm_fn(x, self.weight, self.bias, residual=residual, eps=self.eps,
                               prenorm=prenorm, residual_in_fp32=residual_in_fp32)

    from mamba_ssm.modules import mamba_simple_faster as msf
    import models.fastvim as fastvim

    from mamba_ssm.modules import mamba_simple as ms            # Vim mixer
    ms.selective_scan_fn = ssi.selective_scan_ref
    ms.causal_conv1d_fn = causal_conv1d_fn
    import models.vim as vim
    for m in (vim, ms):
        m.rms_norm_fn = rms_norm_fn
        m.layer_norm_fn = layer_norm_fn
        m.RMSNorm = RMSNorm
    from mamba_ssm.modules import mamba_simple_channel_faster as mscf
    import importlib
    chan = importlib.import_module("models.channel_wise_tokenization.models_channel_mamba_faster")

    from mamba_ssm.modules import mamba_simple_masked_faster as msmf     # MAE masked mixer
    msmf.causal_conv1d_fn = causal_conv1d_fn

    mae = importlib.import_module("models.mae.models_mamba_faster_mae_vimdecoder")

    from mamba_ssm.modules import mamba_simple_channel_faster_2dcompress as mscf2    # 2-D compress channel mixer / model
    chan2 = importlib.import_module("models.channel_wise_tokenization.models_channel_mamba_faster_2dcompress")

    # the two un-pooled baselines of the other task families (round 5): Vim-encoder MAE, ChannelVim with a middle class token
    mae_vim = importlib.import_module("models.mae.fastvim_mae")
    chan_vim = importlib.import_module("models.channel_wise_tokenization.models_channel_mamba")

    for m in (fastvim, msf, chan, mscf, mae, chan2, mscf2, mae_vim, chan_vim):
        m.rms_norm_fn = rms_norm_fn
        m.layer_norm_fn = layer_norm_fn
        m.RMSNorm = RMSNorm

    ns = types.SimpleNamespace(
        ssi=ssi, ln=ln, msf=msf, fastvim=fastvim, mscf=mscf, chan=chan, mscf2=mscf2, chan2=chan2, ms=ms, vim=vim, msmf=msmf, mae=mae, mae_vim=mae_vim, chan_vim=chan_vim, rms_norm_fn=rms_norm_fn,
        layer_norm_fn=layer_norm_fn, RMSNorm=RMSNorm, causal_conv1d_fn=causal_conv1d_fn,
    )
    _loaded = ns
    return ns
