"""FastVim backbone: mirror of models/fastvim.py (``PatchEmbed`` :25-103, ``Block`` :106-217,
``create_block`` :220-291, init fns :295-339, ``VisionMamba`` :342-557, factories :696-967).

Same constructor kwargs, attribute names, ``state_dict`` keys and factory entry points, so it drops
into the reference's Lightning / Hydra ``_target_`` training loops.  Differences are internal:

* ``Block`` never transposes the token grid; odd layers pass ``transposed_grid=True`` to the mixer,
  whose kernels walk the grid with swapped token strides (models/fastvim.py:192-210 does two
  full-length copies per odd layer);
* DropPath on the mixer branch is a per-sample scale folded into the fused add+RMSNorm kernel;
* timm / mmdet / mmseg are not imported (``MM_FastVim`` det/seg glue is out of scope).
"""
import math
from functools import partial
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from .layernorm import RMSNorm, layer_norm_fn, rms_norm_fn
from . import glue_ops as G
from . import mamba_simple_faster as msf
from .mamba_simple_faster import (ChainedBlockFn, LinearFn, Mamba, OutProjAddNormFn, _compute_dtype, _direct_grad, _shadow,
                                  linear_dgrad, linear_wgrad, out_proj_add_norm_ok)
from .mixer_ops import reduce_partials


class _EmbedEpilogueFn(torch.autograd.Function):
    """out (fp32) = lin (B, L, D; bf16/fp32) + bias (D) + pos (1, L, D): the tail of PatchEmbed plus the
    ``x + pos_embed`` of models/fastvim.py:500.  Backward sums with the fixed-order HIP reduction
    (d pos = sum over batch, d bias = sum over batch and tokens) instead of torch's multi-block sum."""

    @staticmethod
    def forward(ctx, lin, bias, pos):
        ctx.lin_dtype = lin.dtype
        ctx.has = (bias is not None, pos is not None)
        pb = bias.float() if bias is not None else None
        if pos is not None:
            pb = pos.float() if pb is None else pos.float() + pb          # (1, L, D): tiny
        if pb is None:
            return lin.float()
        return torch.add(pb, lin)          # ONE full-length pass: fp32 (1, L, D) + bf16 (B, L, D) promotes to fp32

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        B, Ltok, D = g.shape
        dpos = dbias = None
        if ctx.has[1] or ctx.has[0]:
            per_tok = reduce_partials(g.view(B, Ltok * D), B).view(1, Ltok, D)      # sum over batch
            if ctx.has[1]:
                dpos = per_tok
            if ctx.has[0]:
                dbias = reduce_partials(per_tok.view(Ltok, D), Ltok)                  # then over tokens
        return g.to(ctx.lin_dtype), dbias, dpos


class _PatchProjFn(torch.autograd.Function):
    """patches (B, L, K) bf16 -> fp32 (B, L, D) = bf16_round(patches @ W^T) + bias (D) + pos (1, L, D): the patch
    projection (models/fastvim.py:95), the conv bias and the ``x + pos_embed`` of :500 in ONE GEMM whose epilogue adds a
    per-token table -- bit for bit what ``LinearFn`` followed by ``_EmbedEpilogueFn`` returns, without their two
    full-length passes.  Backward: the fixed-order batch / token sums go straight into the flat gradients when the
    parameters have them."""

    @staticmethod
    def forward(ctx, patches, W, bias, pos, cdt):
        B, Ltok, K = patches.shape
        D = W.shape[0]
        with torch.autocast("cuda", enabled=False):
            if pos is not None:
                table = pos.float().reshape(Ltok, D)
                if bias is not None:
                    table = table + bias.float()
            else:
                table = bias.float().reshape(1, D)
            y = G.gemm_rowbias(patches.view(B * Ltok, K), _shadow(W, cdt).reshape(D, K), table.contiguous())
        ctx.save_for_backward(patches, W, bias, pos)
        ctx.cdt = cdt
        return y.view(B, Ltok, D)

    @staticmethod
    def backward(ctx, g):
        patches, W, bias, pos = ctx.saved_tensors
        B, Ltok, K = patches.shape
        D = W.shape[0]
        with torch.autocast("cuda", enabled=False):
            g = g.contiguous()
            dpos = dbias = None
            per_tok = reduce_partials(g.view(B, Ltok * D), B)                               # sum over batch, (Ltok*D,)
            if pos is not None and ctx.needs_input_grad[3]:
                gd = _direct_grad(pos)
                if gd is not None:
                    reduce_partials(per_tok.view(1, Ltok * D), 1, out=gd.view(-1), accumulate=True)
                else:
                    dpos = per_tok.view(1, Ltok, D)
            if bias is not None and ctx.needs_input_grad[2]:
                gd = _direct_grad(bias)
                if gd is not None:
                    reduce_partials(per_tok.view(Ltok, D), Ltok, out=gd.view(-1), accumulate=True)
                else:
                    dbias = reduce_partials(per_tok.view(Ltok, D), Ltok)                    # then over tokens
            g_c = g.view(B * Ltok, D).to(ctx.cdt)
            dpatches = None
            if ctx.needs_input_grad[0]:      # the image asks for its gradient (saliency maps, adversarial inputs)
                dpatches = linear_dgrad(g_c, _shadow(W, ctx.cdt).reshape(D, K)).view(B, Ltok, K)
            dW = linear_wgrad(g_c, patches.view(B * Ltok, K), W)
            if dW is not None:
                dW = dW.view(W.shape)
        return dpatches, dW, dbias, dpos, None


def to_2tuple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def lecun_normal_(tensor):
    # timm.layers.lecun_normal_: variance_scaling_(scale=1, mode="fan_in", distribution="truncated_normal")
    fan_in = nn.init._calculate_fan_in_and_fan_out(tensor)[0]
    std = math.sqrt(1.0 / fan_in) / 0.87962566103423978
    return nn.init.trunc_normal_(tensor, std=std, a=-2 * std, b=2 * std)


class DropPath(nn.Module):
    """Stochastic depth per sample (timm.layers.DropPath).  ``row_scale`` returns the (B,) scale
    vector instead of multiplying, so the fused add+norm kernel can apply it."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    @staticmethod
    def predraw(mods, batch, device):
        """Draw the per-sample keep masks of several DropPath modules in one shot (4 small launches per
        step instead of 2 per layer): Bernoulli(keep) == floor(keep + U[0,1)).  Each module's next
        ``row_scale`` call consumes its row."""
        mods = [m for m in mods if isinstance(m, DropPath) and m.drop_prob > 0.0 and m.training]
        if not mods:
            return
        key = (tuple(m.drop_prob for m in mods), str(device))
        cache = DropPath._keep_cache
        if key not in cache:
            keep = torch.tensor([1.0 - m.drop_prob for m in mods], dtype=torch.float32, device=device)[:, None]
            inv = torch.tensor([1.0 / (1.0 - m.drop_prob) if (m.scale_by_keep and m.drop_prob < 1.0) else 1.0
                                for m in mods], dtype=torch.float32, device=device)[:, None]
            cache[key] = (keep, inv)
        keep, inv = cache[key]
        table = torch.rand(len(mods), batch, device=device, dtype=torch.float32)
        if table.is_cuda:
            G.droppath_table_(table, keep, inv)          # floor(keep + U) * inv in one launch
        else:
            table.add_(keep).floor_().mul_(inv)
        for i, m in enumerate(mods):
            m.__dict__["_pre"] = table[i]

    _keep_cache = {}

    def row_scale(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return None
        pre = self.__dict__.pop("_pre", None)
        if pre is not None and pre.shape[0] == x.shape[0] and pre.device == x.device:
            return pre
        keep = 1.0 - self.drop_prob
        mask = torch.empty(x.shape[0], device=x.device, dtype=torch.float32).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return mask

    def forward(self, x):
        s = self.row_scale(x)
        if s is None:
            return x
        return x * s.to(x.dtype).view(-1, *([1] * (x.ndim - 1)))

    def extra_repr(self):
        return f"drop_prob={round(self.drop_prob, 3):0.3f}"


class PatchEmbed(nn.Module):
    """2D Image to Patch Embedding (models/fastvim.py:25-103)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True,
                 strict_img_size=True, dynamic_img_pad=False, scanpath_type="rowwise"):
        super().__init__()
        self.img_size = to_2tuple(img_size)
        self.patch_size = to_2tuple(patch_size)
        gh, gw = self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1]
        if scanpath_type == "colwise":      # Pool_row in the paper
            self.grid_size = (gw, gh)
        elif scanpath_type == "rowwise":    # Pool_col in the paper
            self.grid_size = (gh, gw)
        else:
            raise ValueError(scanpath_type)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.scanpath_type = scanpath_type
        self.flatten = flatten
        self.strict_img_size = strict_img_size
        self.dynamic_img_pad = dynamic_img_pad
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x, pos_embed=None):
        """``pos_embed`` (1, L, D), optional: added in the same epilogue as the conv bias (only for the
        row-major, flattened token order)."""
        B, C, H, W = x.shape
        if self.strict_img_size:
            assert H == self.img_size[0], f"Input height ({H}) doesn't match model ({self.img_size[0]})."
            assert W == self.img_size[1], f"Input width ({W}) doesn't match model ({self.img_size[1]})."
        elif not self.dynamic_img_pad:
            assert H % self.patch_size[0] == 0, f"Input height ({H}) should be divisible by patch size ({self.patch_size[0]})."
            assert W % self.patch_size[1] == 0, f"Input width ({W}) should be divisible by patch size ({self.patch_size[1]})."
        if self.dynamic_img_pad:
            pad_h = (self.patch_size[0] - H % self.patch_size[0]) % self.patch_size[0]
            pad_w = (self.patch_size[1] - W % self.patch_size[1]) % self.patch_size[1]
            if pad_h or pad_w:          # F.pad with zero pads still clones the batch (77 MB at 128 x 3 x 224 x 224 fp32)
                x = F.pad(x, (0, pad_w, 0, pad_h))
        # k == stride conv == one GEMM over non-overlapping patches: (B*gh*gw, C*ph*pw) x (C*ph*pw, D).
        # (MIOpen resolves this bf16 conv to naive kernels on gfx950; the GEMM form is also what the
        # patch-embed MFMA kernel consumes.)
        ph, pw = self.patch_size
        gh, gw = x.shape[2] // ph, x.shape[3] // pw
        cdt = _compute_dtype(x)
        # unfold and cast in ONE strided copy (fp32 image read once, compute-dtype patches written once); the GEMM then
        # needs no second pass over the patches
        # (the kernel has no autograd node: an image that requires grad takes the strided copy below, whose adjoint --
        # the fold -- autograd knows)
        want_dx = x.requires_grad and torch.is_grad_enabled()
        if G.patch_unfold_ok(x, ph, pw) and cdt in (torch.float32, torch.bfloat16) and not want_dx:
            patches = G.patch_unfold(x, ph, pw, cdt)                # one HIP launch through LDS: 16-byte accesses both ways
        else:
            patches = torch.empty(B, gh * gw, C * ph * pw, device=x.device, dtype=cdt)
            patches.view(B, gh, gw, C, ph, pw).copy_(x.reshape(B, C, gh, ph, gw, pw).permute(0, 2, 4, 1, 3, 5))
        D = self.proj.weight.shape[0]
        if (x.is_cuda and cdt == torch.bfloat16 and self.flatten and self.scanpath_type != "colwise"
                and (pos_embed is not None or self.proj.bias is not None) and (C * ph * pw) % 8 == 0 and D % 8 == 0
                and isinstance(self.norm, nn.Identity)):
            # projection + conv bias + position embedding in one GEMM (row-periodic table in the epilogue)
            return _PatchProjFn.apply(patches, self.proj.weight, self.proj.bias, pos_embed, cdt)
        x = LinearFn.apply(patches, self.proj.weight, cdt)          # (B, gh*gw, D), weight viewed (D, C*ph*pw)
        if self.scanpath_type == "colwise":
            x = x.reshape(B, gh, gw, -1).transpose(1, 2).reshape(B, gh * gw, -1)
        if pos_embed is not None or self.proj.bias is not None:
            x = _EmbedEpilogueFn.apply(x, self.proj.bias, pos_embed)
        if not self.flatten:
            assert pos_embed is None
            g0, g1 = (gh, gw) if self.scanpath_type != "colwise" else (gw, gh)
            x = x.transpose(1, 2).reshape(B, -1, g0, g1)
        return self.norm(x)


class Block(nn.Module):
    """Add -> (RMS/Layer)Norm -> Mixer, returning (hidden_states, residual)  (models/fastvim.py:106-217)."""

    def __init__(self, dim, mixer_cls, norm_cls=nn.LayerNorm, fused_add_norm=False, residual_in_fp32=False,
                 drop_path=0.0, rotate_every_block=True, layer_idx=None, token_size=None):
        super().__init__()
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.mixer = mixer_cls(dim)
        self.norm = norm_cls(dim)
        self.rotate_every_block = rotate_every_block
        self.layer_idx = layer_idx
        self.token_size = token_size
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        if self.fused_add_norm:
            assert isinstance(self.norm, (nn.LayerNorm, RMSNorm)), \
                "Only LayerNorm and RMSNorm are supported for fused_add_norm"

    def forward(self, hidden_states: Tensor, residual: Optional[Tensor] = None, inference_params=None):
        cdt = _compute_dtype(hidden_states)
        is_rms = isinstance(self.norm, RMSNorm)
        if self.fused_add_norm:
            scale = None
            if residual is not None and isinstance(self.drop_path, DropPath):
                scale = self.drop_path.row_scale(hidden_states)
            hidden_states, residual = layer_norm_fn(
                hidden_states, self.norm.weight, self.norm.bias, residual=residual, eps=self.norm.eps,
                prenorm=True, residual_in_fp32=self.residual_in_fp32, is_rms_norm=is_rms,
                row_scale=scale, out_dtype=cdt)
        else:
            residual = hidden_states if residual is None else residual + self.drop_path(hidden_states)
            hidden_states = layer_norm_fn(residual.to(self.norm.weight.dtype), self.norm.weight, self.norm.bias,
                                          eps=self.norm.eps, is_rms_norm=is_rms, out_dtype=cdt)
            if self.residual_in_fp32:
                residual = residual.to(torch.float32)
        # odd layers pool across the other grid axis: the mixer (built with the swapped token_size)
        # reads the un-transposed tokens through swapped strides
        rot = self.rotate_every_block is True and self.layer_idx % 2 != 0
        hidden_states = self._mix(hidden_states, inference_params, rot)
        return hidden_states, residual

    def _mix(self, hidden_states, inference_params, rot):
        return self.mixer(hidden_states, inference_params=inference_params, transposed_grid=rot)

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        raise NotImplementedError("FastVim mixers have no inference cache")


def create_block(d_model, ssm_cfg=None, norm_epsilon=1e-5, drop_path=0.0, rms_norm=False, residual_in_fp32=False,
                 fused_add_norm=False, layer_idx=None, device=None, dtype=None, init_layer_scale=None,
                 scanpath_type="rowwise", use_norm_after_ssm=True, rotate_every_block=True,
                 collapse_method="mean", token_size=None, use_our_selective_scan=False, scaling_factor=1):
    if ssm_cfg is None:
        ssm_cfg = {}
    factory_kwargs = {"device": device, "dtype": dtype}
    rot = rotate_every_block is True and layer_idx % 2 != 0
    mixer_cls = partial(
        Mamba, layer_idx=layer_idx, init_layer_scale=init_layer_scale, scanpath_type=scanpath_type,
        use_norm_after_ssm=use_norm_after_ssm,
        token_size=[token_size[1], token_size[0]] if rot else list(token_size),   # models/fastvim.py:244-274
        collapse_method=collapse_method, use_our_selective_scan=use_our_selective_scan,
        scaling_factor=scaling_factor, **ssm_cfg, **factory_kwargs)
    norm_cls = partial(nn.LayerNorm if not rms_norm else RMSNorm, eps=norm_epsilon, **factory_kwargs)
    block = Block(d_model, mixer_cls, norm_cls=norm_cls, drop_path=drop_path, fused_add_norm=fused_add_norm,
                  residual_in_fp32=residual_in_fp32, rotate_every_block=rotate_every_block, layer_idx=layer_idx,
                  token_size=token_size)
    block.layer_idx = layer_idx
    return block


def _init_weights(module, n_layer, initializer_range=0.02, rescale_prenorm_residual=True, n_residuals_per_layer=1):
    """models/fastvim.py:295-324 (GPT-2 style scaled init of the residual-branch output projections)."""
    if isinstance(module, nn.Linear):
        if module.bias is not None and not getattr(module.bias, "_no_reinit", False):
            nn.init.zeros_(module.bias)
    elif isinstance(module, nn.Embedding):
        nn.init.normal_(module.weight, std=initializer_range)
    if rescale_prenorm_residual:
        for name, p in module.named_parameters():
            if name in ["out_proj.weight", "fc2.weight"]:
                nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                with torch.no_grad():
                    p /= math.sqrt(n_residuals_per_layer * n_layer)


def segm_init_weights(m):
    if isinstance(m, nn.Linear):
        trunc_normal_(m.weight, std=0.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.Conv2d):
        lecun_normal_(m.weight)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, (nn.LayerNorm, nn.GroupNorm, nn.BatchNorm2d)):
        nn.init.zeros_(m.bias)
        nn.init.ones_(m.weight)


class VisionMamba(nn.Module):
    def __init__(self, img_size=224, patch_size=16, stride=16, depth=24, embed_dim=192, channels=3,
                 num_classes=1000, ssm_cfg=None, drop_rate=0.0, drop_path_rate=0.1, norm_epsilon: float = 1e-5,
                 rms_norm: bool = True, initializer_cfg=None, fused_add_norm=False, residual_in_fp32=False,
                 device=None, dtype=None, final_pool_type="none", if_abs_pos_embed=True, init_layer_scale=None,
                 embed_layer=PatchEmbed, scanpath_type="rowwise", use_norm_after_ssm=True,
                 rotate_every_block=True, collapse_method="mean", use_our_selective_scan=False,
                 scaling_factor=1, **kwargs):
        factory_kwargs = {"device": device, "dtype": dtype}
        kwargs.update(factory_kwargs)
        super().__init__()
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.final_pool_type = final_pool_type
        self.if_abs_pos_embed = if_abs_pos_embed
        self.rotate_every_block = rotate_every_block
        self.num_classes = num_classes
        self.d_model = self.num_features = self.embed_dim = embed_dim
        self.patch_size = patch_size
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=channels,
                                      embed_dim=embed_dim, strict_img_size=False, dynamic_img_pad=True,
                                      scanpath_type=scanpath_type)
        self.num_patches = self.patch_embed.num_patches
        self.token_size = self.patch_embed.grid_size
        if if_abs_pos_embed:
            self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches, self.embed_dim))
            self.pos_drop = nn.Dropout(p=drop_rate)
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()

        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]   # stochastic depth decay rule
        inter_dpr = [0.0] + dpr                                              # layer i uses inter_dpr[i] (:433)
        self.drop_path = DropPath(drop_path_rate) if drop_path_rate > 0.0 else nn.Identity()
        self.layers = nn.ModuleList([
            create_block(embed_dim, ssm_cfg=ssm_cfg, norm_epsilon=norm_epsilon, rms_norm=rms_norm,
                         residual_in_fp32=residual_in_fp32, fused_add_norm=fused_add_norm, layer_idx=i,
                         drop_path=inter_dpr[i], init_layer_scale=init_layer_scale, scanpath_type=scanpath_type,
                         use_norm_after_ssm=use_norm_after_ssm, rotate_every_block=rotate_every_block,
                         collapse_method=collapse_method, token_size=self.token_size,
                         use_our_selective_scan=use_our_selective_scan, scaling_factor=scaling_factor,
                         **factory_kwargs)
            for i in range(depth)])
        self.norm_f = (nn.LayerNorm if not rms_norm else RMSNorm)(embed_dim, eps=norm_epsilon, **factory_kwargs)

        self.patch_embed.apply(segm_init_weights)
        self.head.apply(segm_init_weights)
        if if_abs_pos_embed:
            trunc_normal_(self.pos_embed, std=0.02)
        self.apply(partial(_init_weights, n_layer=depth, **(initializer_cfg if initializer_cfg is not None else {})))

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed"}

    # forward_features (models/fastvim.py:484-546) in three pieces, so that a training step can be cut into segments of
    # layers (fastvim_amd/pipeline.py: gradient buckets exchanged while the next segment's backward runs)
    def _embed(self, x):
        B, _, H, W = x.shape
        if self.if_abs_pos_embed:
            H, W = math.ceil(H / self.patch_size), math.ceil(W / self.patch_size)
            hw = (H, W) if self.patch_embed.scanpath_type == "rowwise" else (W, H)
            if hw[0] != self.token_size[0] or hw[1] != self.token_size[1]:
                # the reference's resize call is broken for this case: construct
                # the model with the target img_size instead
                raise RuntimeError(f"input grid {H}x{W} differs from the model's {self.token_size}; "
                                   "build VisionMamba with the matching img_size")
            x = self.patch_embed(x, pos_embed=self.pos_embed)     # x + pos_embed (:500) in the epilogue
            x = self.pos_drop(x)
        else:
            H, W = math.ceil(H / self.patch_size), math.ceil(W / self.patch_size)
            x = self.patch_embed(x)
        if self.training:
            DropPath.predraw([l.drop_path for l in self.layers] + [self.drop_path], x.shape[0], x.device)
        return x, (H, W)

    def _chainable(self, hidden_states, lo, hi, inference_params, out_indices, carried=False):
        """Blocks lo .. hi-1 can run with every interior ``out_proj`` fused into the next block's add + RMSNorm
        (``carried``: the run is part of a longer chain -- ``_run_layers_open`` -- and may be a single block)."""
        if inference_params is not None or out_indices is not None or hi - lo < (1 if carried else 2) or not hidden_states.is_cuda:
            return False
        if _compute_dtype(hidden_states) != torch.bfloat16 or self.embed_dim != 192:
            return False
        for blk in self.layers[lo:hi]:
            mx = blk.mixer
            if (type(blk) is not Block or type(mx) is not Mamba or not blk.fused_add_norm or not blk.residual_in_fp32
                    or not isinstance(blk.norm, RMSNorm) or blk.norm.bias is not None or mx.init_layer_scale is not None
                    or mx.out_proj.bias is not None or mx.out_proj.weight.shape[1] % 64 or mx.d_conv != 4 or mx.d_state != 16):
                return False
        return True

    def _run_layers_chained(self, hidden_states, residual, lo, hi, pend=None, close=True):
        """Same values as the plain loop: block i's mixer stops at its gated activations, and ``out_proj`` runs inside the
        GEMM that also does block i+1's DropPath scale + residual add + RMSNorm; from the second block on a block is one
        autograd node (``ChainedBlockFn``), whose backward also fuses the in_proj data gradient into the norm's adjoint.
        The range is closed by a plain ``out_proj``.  Parameter gradients equal the plain loop's bit for bit, except the
        norm weights', whose row sums are grouped differently (fixed order, ~1e-7 relative)."""
        cdt = _compute_dtype(hidden_states if pend is None else pend[0])
        # pend: (gated activations, out_proj weight) of the previous block -- None at the start of a chain, or carried in
        # from the previous run of blocks (``_run_layers_open``: the segmented training step cuts there)
        # pack: the previous block's gated activations have NOT been computed yet -- their combine (expand + LayerNorm +
        # gate) is deferred into the next block's out_proj + add + norm launch (``fv_mixer_combine_out_proj_addnorm``);
        # whoever else consumes them (the end of the run, a cut, a block the fused launch is not built for) resolves it
        pack = None
        for layer_idx in range(lo, hi):
            blk = self.layers[layer_idx]
            scale = None
            if residual is not None and isinstance(blk.drop_path, DropPath):
                scale = blk.drop_path.row_scale(hidden_states if pend is None else pend[0])
            rot = blk.rotate_every_block is True and blk.layer_idx % 2 != 0
            if pend is not None and out_proj_add_norm_ok(pend[0], pend[1], residual, blk.norm.weight, cdt):
                # previous out_proj + this block's add + norm + mixer as ONE autograd node: its backward can hand the
                # in_proj data gradient straight to the norm's adjoint
                ChainedBlockFn.pending_in = pack
                # a deferred combine leaves saved tensors UNWRITTEN until the next block's launch fills them: anything that
                # reads saved tensors when they are saved (torch.autograd.graph.save_on_cpu, non-reentrant checkpointing --
                # both are saved-tensor hooks) would copy garbage, so deferral is off while such hooks are active.  The
                # hand-over (pending_in / pending_out / defer) is class state of one host thread: single-threaded by design
                ChainedBlockFn.defer = msf.COMBINE_IN_OUT_PROJ and torch._C._autograd._top_saved_tensors_default_hooks(True) is None
                ChainedBlockFn.pending_out = None
                try:
                    g, residual = ChainedBlockFn.apply(pend[0], pend[1], residual, blk.norm.weight, float(blk.norm.eps), scale,
                                                       *blk.mixer.mixer_fn_args(cdt, rot, defer_out_proj=True))
                except BaseException:
                    ChainedBlockFn.pending_out = None      # (a failed apply must not leave its pack for the next caller)
                    raise
                finally:
                    ChainedBlockFn.pending_in, ChainedBlockFn.defer = None, False
                pack, ChainedBlockFn.pending_out = ChainedBlockFn.pending_out, None
                pend = (g, blk.mixer.out_proj.weight)
                continue
            else:
                msf.resolve_combine(pack)
                pack = None
                if pend is not None:
                    hidden_states = LinearFn.apply(pend[0], pend[1], cdt)
                hidden_states, residual = layer_norm_fn(
                    hidden_states, blk.norm.weight, blk.norm.bias, residual=residual, eps=blk.norm.eps, prenorm=True,
                    residual_in_fp32=blk.residual_in_fp32, is_rms_norm=True, row_scale=scale, out_dtype=cdt)
            pend = (blk.mixer(hidden_states, transposed_grid=rot, defer_out_proj=True), blk.mixer.out_proj.weight)
        msf.resolve_combine(pack)      # the run ends (or is cut) here: the last block's combine as its own launch
        if not close:
            return None, residual, pend
        return LinearFn.apply(pend[0], pend[1], cdt), residual

    def _run_layers_open(self, hidden_states, residual, lo, hi, pend=None):
        """``_run_layers`` that may leave the run's last ``out_proj`` to whoever continues: returns (hidden_states,
        residual, pend) with exactly one of hidden_states / pend set; pass ``pend`` back in for the next run and finish
        with ``_close_run``.  A run cut this way computes exactly what the uncut chain computes."""
        probe = hidden_states if pend is None else pend[0]
        if self._chainable(probe, lo, hi, None, None, carried=True):       # an open run may be a single block
            return self._run_layers_chained(hidden_states, residual, lo, hi, pend=pend, close=False)
        if pend is not None:
            hidden_states = self._close_run(pend)
        hidden_states, residual = self._run_layers(hidden_states, residual, lo, hi)
        return hidden_states, residual, None

    def _close_run(self, pend):
        return LinearFn.apply(pend[0], pend[1], _compute_dtype(pend[0]))

    def _run_layers(self, hidden_states, residual, lo, hi, inference_params=None, out_indices=None, outs=None):
        if self._chainable(hidden_states, lo, hi, inference_params, out_indices):
            return self._run_layers_chained(hidden_states, residual, lo, hi)
        for layer_idx in range(lo, hi):
            hidden_states, residual = self.layers[layer_idx](hidden_states, residual, inference_params=inference_params)
            if out_indices is not None and layer_idx in out_indices:
                outs.append(hidden_states)
        return hidden_states, residual

    def _final(self, hidden_states, residual):
        is_rms = isinstance(self.norm_f, RMSNorm)
        if not self.fused_add_norm:
            residual = hidden_states if residual is None else residual + self.drop_path(hidden_states)
            hidden_states = layer_norm_fn(residual.to(self.norm_f.weight.dtype), self.norm_f.weight, self.norm_f.bias,
                                          eps=self.norm_f.eps, is_rms_norm=is_rms)
        else:
            scale = self.drop_path.row_scale(hidden_states) if isinstance(self.drop_path, DropPath) else None
            hidden_states = layer_norm_fn(hidden_states, self.norm_f.weight, self.norm_f.bias, eps=self.norm_f.eps,
                                          residual=residual, prenorm=False, residual_in_fp32=self.residual_in_fp32,
                                          is_rms_norm=is_rms, row_scale=scale)
        if self.final_pool_type == "none":
            return hidden_states[:, -1, :]
        elif self.final_pool_type == "mean":
            if G.mean_pool_ok(hidden_states):
                return G.MeanPoolFn.apply(hidden_states)
            return hidden_states.mean(dim=1)
        elif self.final_pool_type in ("max", "all"):
            return hidden_states
        raise NotImplementedError

    def _head(self, x):
        if isinstance(self.head, nn.Linear) and x.is_cuda:
            # F.linear(x, head.weight, head.bias) (models/fastvim.py:541) through the MFMA GEMM: at batch 128 the
            # library picks a one-workgroup kernel for this 128 x 1000 x 192 problem (25 us)
            x = LinearFn.apply(x, self.head.weight, _compute_dtype(x), self.head.bias)
        else:
            x = self.head(x)
        if self.final_pool_type == "max":
            x = x.max(dim=1)[0]
        return x

    def forward_features(self, x, inference_params=None, out_indices=None):
        hidden_states, (H, W) = self._embed(x)
        outs = []
        hidden_states, residual = self._run_layers(hidden_states, None, 0, len(self.layers), inference_params,
                                                   out_indices, outs)
        if out_indices is not None:
            assert len(outs) == len(out_indices)
            return outs, (H, W)
        return self._final(hidden_states, residual)

    def forward(self, x, return_features=False, inference_params=None):
        half = msf.half_io(x) or self.pos_embed_dtype_is_half()
        x = self.forward_features(x, inference_params)
        if not return_features:
            x = self._head(x)
        # fp16 regime (fp16 autocast / a .half() model): computed in fp32 inside, fp16 at the boundary like the reference
        return x.to(torch.float16) if half and torch.is_tensor(x) and x.is_floating_point() else x

    def pos_embed_dtype_is_half(self):
        p = next(self.parameters(), None)
        return p is not None and p.dtype == torch.float16


class MM_FastVim(VisionMamba):
    """Multi-scale feature backbone for detection / segmentation heads (models/fastvim.py:560-691), without the
    mmdet / mmseg registry decorators: ``forward(x)`` returns the LayerNorm-ed hidden states of ``out_indices`` as
    (B, C, H, W) maps; ``load_pretrained`` reads a Lightning checkpoint (``state_dict_ema`` preferred, ``backbone.``
    prefix stripped, square ``pos_embed`` bicubically resized to this model's grid)."""

    def __init__(self, img_size=224, patch_size=16, stride=16, in_chans=3, embed_dim=192, depth=24, pretrained=None,
                 out_indices=(5, 11, 17, 23), scanpath_type="rowwise", load_ema=True, **kwargs):
        super().__init__(img_size, patch_size, stride, depth, embed_dim, in_chans, scanpath_type=scanpath_type, **kwargs)
        self.load_ema = load_ema
        self.scanpath_type = scanpath_type
        self.out_indices = list(out_indices)
        for i in range(len(self.out_indices)):
            self.add_module(f"outnorm_{i}", nn.LayerNorm(self.embed_dim))
        del self.head
        del self.norm_f
        self.load_pretrained(pretrained)

    def load_pretrained(self, pretrained):
        if pretrained is None:
            return None
        ckpt = torch.load(pretrained, map_location="cpu")
        state_dict = ckpt["state_dict_ema"] if (self.load_ema and "state_dict_ema" in ckpt) else ckpt["state_dict"]
        sd = {k.replace("backbone.", ""): v for k, v in state_dict.items()}
        if "pos_embed" in sd:
            pos_size = int(math.sqrt(sd["pos_embed"].shape[1]))
            sd["pos_embed"] = self.resize_pos_embed(sd["pos_embed"], self.token_size, (pos_size, pos_size), "bicubic",
                                                    self.scanpath_type)
        if "patch_embed.proj.weight" in sd and \
                self.patch_embed.patch_size[-1] != sd["patch_embed.proj.weight"].shape[-1]:
            sd.pop("patch_embed.proj.weight")
            sd.pop("patch_embed.proj.bias", None)
        return self.load_state_dict(sd, strict=False)

    @staticmethod
    def resize_pos_embed(pos_embed, input_shape, pos_shape, mode, scanpath_type):
        """(1, L, C) position table of a ``pos_shape`` grid -> ``input_shape`` grid (models/fastvim.py:645-680)."""
        assert pos_embed.ndim == 3, "shape of pos_embed must be [B, L, C]"
        pos_h, pos_w = pos_shape
        w = pos_embed.reshape(1, pos_h, pos_w, pos_embed.shape[2]).permute(0, 3, 1, 2)
        if scanpath_type == "colwise":
            w = w.transpose(2, 3)
        w = F.interpolate(w, size=tuple(input_shape), mode=mode, align_corners=False)
        if scanpath_type == "colwise":
            w = w.transpose(2, 3)
        return torch.flatten(w, 2).transpose(1, 2)

    def forward(self, x):
        C = self.embed_dim
        outs, (H, W) = self.forward_features(x, out_indices=self.out_indices)
        outs = [getattr(self, f"outnorm_{i}")(o.float()) for i, o in enumerate(outs)]
        outs = [o.view(-1, H, W, C).permute(0, 3, 1, 2).contiguous() for o in outs]
        return outs[0] if len(self.out_indices) == 1 else outs


def _factory(embed_dim, depth, img_size, patch_size, stride, kwargs):
    return VisionMamba(img_size=img_size, patch_size=patch_size, stride=stride, embed_dim=embed_dim, depth=depth,
                       rms_norm=True, residual_in_fp32=True, fused_add_norm=True, final_pool_type="mean",
                       if_abs_pos_embed=True, **kwargs)


def _load_transfer_checkpoint(model, path, img_size, patch_size):
    """Checkpoint contract of models/fastvim.py:779-815: Lightning ``state_dict`` with a ``backbone.``
    prefix; square ``pos_embed`` bicubically resized to the new grid."""
    checkpoint = torch.load(path, map_location="cpu")["state_dict"]
    sd = {k.replace("backbone.", ""): v for k, v in checkpoint.items()}
    if "pos_embed" in sd:
        orig = int(math.sqrt(sd["pos_embed"].shape[1]))
        new = int(img_size // patch_size)
        if orig != new:
            e = sd["pos_embed"].shape[-1]
            t = sd["pos_embed"].reshape(-1, orig, orig, e).permute(0, 3, 1, 2)
            t = F.interpolate(t, size=(new, new), mode="bicubic", align_corners=False)
            sd["pos_embed"] = t.permute(0, 2, 3, 1).flatten(1, 2)
    return model.load_state_dict(sd, strict=False)


def vim_tiny_patch16_224_final_pool_mean_abs_pos_embed_with_noclstok_div2(
        pretrained=False, img_size=224, patch_size=16, stride=16, **kwargs):
    assert not pretrained, "no network: load weights with load_state_dict"
    return _factory(192, 24, img_size, patch_size, stride, kwargs)


def vim_small_patch16_224_final_pool_mean_abs_pos_embed_with_noclstok_div2(
        pretrained=False, img_size=224, patch_size=16, stride=16, **kwargs):
    assert not pretrained, "no network: load weights with load_state_dict"
    return _factory(384, 24, img_size, patch_size, stride, kwargs)


def vim_base_patch16_224_final_pool_mean_abs_pos_embed_with_noclstok_div2(
        pretrained=False, img_size=224, patch_size=16, stride=16, pretrained_checkpoint_path=None, **kwargs):
    assert not pretrained, "no network: load weights with load_state_dict"
    model = _factory(768, 24, img_size, patch_size, stride, kwargs)
    if pretrained_checkpoint_path is not None:
        _load_transfer_checkpoint(model, pretrained_checkpoint_path, img_size, patch_size)
    return model


def vim_large_patch16_224_final_pool_mean_abs_pos_embed_with_noclstok_div2(
        pretrained=False, img_size=224, patch_size=16, stride=16, pretrained_checkpoint_path=None, **kwargs):
    assert not pretrained, "no network: load weights with load_state_dict"
    model = _factory(1024, 48, img_size, patch_size, stride, kwargs)
    if pretrained_checkpoint_path is not None:
        _load_transfer_checkpoint(model, pretrained_checkpoint_path, img_size, patch_size)
    return model


def vim_huge_patch14_224_final_pool_mean_abs_pos_embed_with_noclstok_div2(
        pretrained=False, img_size=224, patch_size=14, stride=14, pretrained_checkpoint_path=None, **kwargs):
    assert not pretrained, "no network: load weights with load_state_dict"
    model = _factory(1280, 64, img_size, patch_size, stride, kwargs)
    if pretrained_checkpoint_path is not None:
        _load_transfer_checkpoint(model, pretrained_checkpoint_path, img_size, patch_size)
    return model


# short aliases for the three configs BASELINE.json names
FastVimT = vim_tiny_patch16_224_final_pool_mean_abs_pos_embed_with_noclstok_div2
FastVimS = vim_small_patch16_224_final_pool_mean_abs_pos_embed_with_noclstok_div2
FastVimB = vim_base_patch16_224_final_pool_mean_abs_pos_embed_with_noclstok_div2
