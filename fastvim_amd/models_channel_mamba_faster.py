"""FastChannelVim backbone: mirror of models/channel_wise_tokenization/models_channel_mamba_faster.py
(``PatchEmbedPerChannel`` :22-203, ``Block`` :206-336, ``create_block`` :339-408, ``VisionMamba``
:458-682, factory :685-706).  Same constructor kwargs, attribute names, ``state_dict`` keys and entry
point.  ``scan_order="Channel-First"`` (the default, what the S/16 entry point uses) runs without any token copy;
``"Spatial-First"`` (the reference's ablation: tokens ordered (channel, row, col)) permutes the embedded tokens once and
transposes rotated layers physically, as the reference does.

Internals follow fastvim_amd/fastvim.py: the shared Conv3d(1, d, (1, p, p)) is one MFMA GEMM over
per-channel patches written directly in (row, col, channel) token order; bias + channel embedding +
repeat-interleaved ``pos_embed`` are one epilogue with fixed-order gradient reductions; odd layers
read the un-transposed tokens through swapped cell strides.
"""
import random
from functools import partial
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from .fastvim import (DropPath, _compute_dtype, _init_weights, segm_init_weights as _segm_2d,
                      to_2tuple, trunc_normal_)
from .layernorm import RMSNorm, layer_norm_fn
from .mamba_simple_channel_faster import Mamba
from .mamba_simple_faster import LinearFn, linear_module
from .mixer_ops import reduce_partials


class _ChannelEmbedEpilogueFn(torch.autograd.Function):
    """out (fp32, (B, P, C, D)) = lin + bias (D) + chan (Bc, C, D) + pos (1, P, D): the tail of
    PatchEmbedPerChannel (:188-191) plus ``x + repeat_interleave(pos_embed, C, 1)`` (:626-627).
    Backward reduces with the fixed-order HIP reduction: batch first, then channels (d pos),
    positions (d chan), both (d bias)."""

    @staticmethod
    def forward(ctx, lin, bias, chan, pos):
        ctx.lin_dtype = lin.dtype
        ctx.has = (bias is not None, pos is not None)
        ctx.chan_batched = chan.shape[0] != 1
        add = chan.float()[:, None]                                   # (Bc, 1, C, D)
        if bias is not None:
            add = add + bias.float()
        if pos is not None:
            add = add + pos.float()[:, :, None]                       # (Bc, P, C, D): per-sample only under channel sampling
        return torch.add(add, lin)       # one full-length pass; fp32 + bf16 promotes to fp32

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        B, P, C, D = g.shape
        per_tok = reduce_partials(g.view(B, P * C * D), B).view(P, C * D)                 # sum over batch
        if ctx.chan_batched:          # explicit per-sample channel ids: keep the batch axis
            dchan = torch.stack([reduce_partials(g[b].view(P, C * D), P).view(C, D) for b in range(B)])
            d_cd = reduce_partials(dchan.view(B, C * D), B).view(C, D)
        else:
            d_cd = reduce_partials(per_tok, P).view(C, D)                                  # sum over positions
            dchan = d_cd.view(1, C, D)
        dbias = reduce_partials(d_cd, C) if ctx.has[0] else None
        dpos = None
        if ctx.has[1]:
            by_c = per_tok.view(P, C, D).transpose(0, 1).contiguous().view(C, P * D)
            dpos = reduce_partials(by_c, C).view(1, P, D)                                  # sum over channels
        return g.to(ctx.lin_dtype), dbias, dchan, dpos


class PatchEmbedPerChannel(nn.Module):
    """Per-channel patch embedding with a shared projection, a channel-embedding table and
    hierarchical channel sampling (models_channel_mamba_faster.py:22-203)."""

    def __init__(self, img_size: int = 224, patch_size: int = 16, stride: int = 16, in_chans: int = 8,
                 embed_dim: int = 768, hcs: bool = True, scan_order: str = "Channel-First", sort_channels=True,
                 scanpath_type="rowwise", flatten=True):
        super().__init__()
        if scan_order not in ("Channel-First", "Spatial-First"):
            raise ValueError(scan_order)
        if stride != patch_size:
            raise NotImplementedError("PatchEmbedPerChannel: stride == patch_size (non-overlapping patches)")
        self.img_size = to_2tuple(img_size)
        self.patch_size = to_2tuple(patch_size)
        gh, gw = self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1]
        if scanpath_type == "colwise":
            self.grid_size = (gw, gh)
        elif scanpath_type == "rowwise":
            self.grid_size = (gh, gw)
        else:
            raise ValueError(scanpath_type)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.scanpath_type = scanpath_type
        self.flatten = flatten
        self.hcs = hcs
        self.scan_order = scan_order
        self.sort_channels = sort_channels
        # all channels share the filter: channel axis is the depth axis of a Conv3d (:117-125)
        self.proj = nn.Conv3d(1, embed_dim, kernel_size=(1, patch_size, patch_size), stride=(1, stride, stride))
        self.channel_embed = nn.Embedding(in_chans, embed_dim)

    def forward(self, x: Tensor, input_channel_order: Optional[Tensor] = None, pos_embed=None):
        """Returns (tokens (B, P*C', D) Channel-First, C', h, w, channels) like the reference (:203).
        ``pos_embed`` (1, P, D), optional: added (repeat-interleaved per channel) in the same epilogue."""
        B, num_channels, h, w = x.shape
        if input_channel_order is None:
            chan = self.channel_embed.weight[:num_channels][None]                # (1, C, D), == embedding(arange)
        else:
            chan = self.channel_embed(input_channel_order)                       # (B, C, D)
        # same python-RNG consumption as the reference (:167-185)
        if self.training and self.hcs:
            c_new = random.randint(1, num_channels)
            channels = random.sample(range(num_channels), k=c_new)
            if self.sort_channels is True:
                channels.sort()
            num_channels = c_new
            x = x[:, channels, :, :]
            chan = chan[:, channels]
        else:
            channels = random.sample(range(num_channels), k=num_channels)
            channels.sort()
        ph, pw = self.patch_size
        gh, gw = h // ph, w // pw
        C = num_channels
        cdt = _compute_dtype(x)
        p6 = x.reshape(B, C, gh, ph, gw, pw)
        if self.scanpath_type == "colwise":                                      # :193-194
            patches = p6.permute(0, 4, 2, 1, 3, 5)
            g0, g1 = gw, gh
        else:
            patches = p6.permute(0, 2, 4, 1, 3, 5)
            g0, g1 = gh, gw
        patches = patches.reshape(B, g0 * g1 * C, ph * pw)                       # (row, col, channel) order
        lin = LinearFn.apply(patches, self.proj.weight, cdt)                     # weight viewed (D, ph*pw)
        D = lin.shape[-1]
        out = _ChannelEmbedEpilogueFn.apply(lin.view(B, g0 * g1, C, D), self.proj.bias, chan, pos_embed)
        if self.scan_order == "Spatial-First":                                   # tokens ordered (channel, row, col)
            out = out.transpose(1, 2)                                            # (B, C, P, D); pos_embed[p] per (c, p) (:620-623)
            if self.flatten:
                out = out.reshape(B, C * g0 * g1, D)
            else:
                assert pos_embed is None
                out = out.reshape(B, C, g0, g1, D).permute(0, 4, 1, 2, 3)        # B D C H' W'
        elif self.flatten:
            out = out.view(B, g0 * g1 * C, D)
        else:
            assert pos_embed is None
            out = out.view(B, g0, g1, C, D).permute(0, 4, 1, 2, 3)               # B D H' W' C (:196-198)
        return out, num_channels, h, w, channels


class Block(nn.Module):
    """Add -> (RMS/Layer)Norm -> channel Mixer (models_channel_mamba_faster.py:206-336)."""

    def __init__(self, dim, mixer_cls, norm_cls=nn.LayerNorm, fused_add_norm=False, residual_in_fp32=False,
                 drop_path=0.0, rotate_every_block=True, layer_idx=None, token_size=None, scan_order=None,
                 max_tokens_per_patch=None):
        super().__init__()
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.mixer = mixer_cls(dim)
        self.norm = norm_cls(dim)
        self.rotate_every_block = rotate_every_block
        self.layer_idx = layer_idx
        self.token_size = token_size
        self.scan_order = scan_order
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        if self.fused_add_norm:
            assert isinstance(self.norm, (nn.LayerNorm, RMSNorm)), \
                "Only LayerNorm and RMSNorm are supported for fused_add_norm"

    def forward(self, hidden_states: Tensor, tokens_per_patch: int, residual: Optional[Tensor] = None,
                inference_params=None):
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
        rot = self._rotated()
        if rot and self._physical_transpose():                                   # :298-329 as written: two copies
            B, M, _ = hidden_states.shape
            T0, T1 = self.token_size
            if self.scan_order == "Spatial-First":
                hidden_states = hidden_states.reshape(B, tokens_per_patch, T0, T1, -1).transpose(2, 3).reshape(B, M, -1)
            else:
                hidden_states = hidden_states.reshape(B, T0, T1, tokens_per_patch, -1).transpose(1, 2).reshape(B, M, -1)
            hidden_states = self.mixer(hidden_states, tokens_per_patch, inference_params=inference_params)
            if self.scan_order == "Spatial-First":
                hidden_states = hidden_states.reshape(B, tokens_per_patch, T1, T0, -1).transpose(2, 3).reshape(B, M, -1)
            else:
                hidden_states = hidden_states.reshape(B, T1, T0, tokens_per_patch, -1).transpose(1, 2).reshape(B, M, -1)
        else:                                                                    # Channel-First: swapped cell strides, no copy
            hidden_states = self.mixer(hidden_states, tokens_per_patch, inference_params=inference_params,
                                       transposed_grid=rot)
        return hidden_states, residual

    def _rotated(self):
        return self.rotate_every_block is True and self.layer_idx % 2 != 0

    def _physical_transpose(self):
        return self.scan_order == "Spatial-First"

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        raise NotImplementedError("FastVim mixers have no inference cache")


def create_block(d_model, ssm_cfg=None, norm_epsilon=1e-5, drop_path=0.0, rms_norm=False, residual_in_fp32=False,
                 fused_add_norm=False, layer_idx=None, device=None, dtype=None, init_layer_scale=None,
                 scanpath_type="rowwise", use_norm_after_ssm=True, rotate_every_block=True,
                 collapse_method="mean", token_size=None, scan_order=None, max_tokens_per_patch=None,
                 mixer_type=Mamba, block_type=None, rotated=None):
    """``mixer_type`` / ``block_type`` / ``rotated(layer_idx)``: hooks of the 2-D compress variant (its own mixer class,
    Block subclass and three-layer rotation cycle)."""
    if ssm_cfg is None:
        ssm_cfg = {}
    factory_kwargs = {"device": device, "dtype": dtype}
    rot = rotate_every_block is True and (layer_idx % 2 != 0 if rotated is None else rotated(layer_idx))
    mixer_cls = partial(
        mixer_type, layer_idx=layer_idx, init_layer_scale=init_layer_scale, scanpath_type=scanpath_type,
        use_norm_after_ssm=use_norm_after_ssm,
        token_size=[token_size[1], token_size[0]] if rot else list(token_size),   # :363-388
        collapse_method=collapse_method, scan_order=scan_order, **ssm_cfg, **factory_kwargs)
    norm_cls = partial(nn.LayerNorm if not rms_norm else RMSNorm, eps=norm_epsilon, **factory_kwargs)
    block = (block_type or Block)(d_model, mixer_cls, norm_cls=norm_cls, drop_path=drop_path, fused_add_norm=fused_add_norm,
                  residual_in_fp32=residual_in_fp32, rotate_every_block=rotate_every_block, layer_idx=layer_idx,
                  token_size=token_size, scan_order=scan_order, max_tokens_per_patch=max_tokens_per_patch)
    block.layer_idx = layer_idx
    return block


def segm_init_weights(m):
    """:443-455 (the reference's Conv2d branch does not match the Conv3d projection, which therefore
    keeps torch's default init there; same here)."""
    _segm_2d(m)


class VisionMamba(nn.Module):
    _create_block = staticmethod(create_block)

    def __init__(self, img_size=224, patch_size=16, stride=16, depth=24, embed_dim=192, channels=3,
                 num_classes=1000, ssm_cfg=None, drop_rate=0.0, drop_path_rate=0.1, norm_epsilon: float = 1e-5,
                 rms_norm: bool = False, initializer_cfg=None, fused_add_norm=False, residual_in_fp32=False,
                 device=None, dtype=None, final_pool_type="mean", if_abs_pos_embed=True, init_layer_scale=None,
                 scan_order="Channel-First", hcs=True, sort_channels=True, scanpath_type="rowwise",
                 use_norm_after_ssm=True, rotate_every_block=True, collapse_method="mean", **kwargs):
        factory_kwargs = {"device": device, "dtype": dtype}
        kwargs.update(factory_kwargs)
        super().__init__()
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.final_pool_type = final_pool_type
        self.if_abs_pos_embed = if_abs_pos_embed
        self.rotate_every_block = rotate_every_block
        self.channels = channels
        self.num_classes = num_classes
        self.d_model = self.num_features = self.embed_dim = embed_dim
        self.scan_order = scan_order
        self.patch_size = patch_size
        self.patch_embed = PatchEmbedPerChannel(img_size=img_size, patch_size=patch_size, stride=stride,
                                                in_chans=channels, embed_dim=embed_dim, hcs=hcs,
                                                scan_order=scan_order, sort_channels=sort_channels,
                                                scanpath_type=scanpath_type)
        self.num_patches = self.patch_embed.num_patches
        self.token_size = self.patch_embed.grid_size
        if if_abs_pos_embed:
            self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches, self.embed_dim))
            self.pos_drop = nn.Dropout(p=drop_rate)
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        inter_dpr = [0.0] + dpr
        self.drop_path = DropPath(drop_path_rate) if drop_path_rate > 0.0 else nn.Identity()
        self.layers = nn.ModuleList([
            self._create_block(embed_dim, ssm_cfg=ssm_cfg, norm_epsilon=norm_epsilon, rms_norm=rms_norm,
                         residual_in_fp32=residual_in_fp32, fused_add_norm=fused_add_norm, layer_idx=i,
                         drop_path=inter_dpr[i], init_layer_scale=init_layer_scale, scanpath_type=scanpath_type,
                         use_norm_after_ssm=use_norm_after_ssm, rotate_every_block=rotate_every_block,
                         collapse_method=collapse_method, token_size=self.token_size, scan_order=self.scan_order,
                         max_tokens_per_patch=self.channels, **factory_kwargs)
            for i in range(depth)])
        self.norm_f = (nn.LayerNorm if not rms_norm else RMSNorm)(embed_dim, eps=norm_epsilon, **factory_kwargs)
        self.patch_embed.apply(segm_init_weights)
        self.head.apply(segm_init_weights)
        if if_abs_pos_embed:
            trunc_normal_(self.pos_embed, std=0.02)
        self.apply(partial(_init_weights, n_layer=depth, **(initializer_cfg if initializer_cfg is not None else {})))

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "pos_embed_obj", "cls_token", "dist_token", "cls_token_head", "cls_token_tail"}

    # forward_features (models_channel_mamba_faster.py:607-667) in the four pieces the segmented data-parallel step cuts at
    # (fastvim_amd/pipeline.py: gradient buckets exchanged under the backward of the remaining runs of blocks)
    def _embed(self, x):
        if self.if_abs_pos_embed:
            x, tokens_per_patch, h, w, _ = self.patch_embed(x, pos_embed=self.pos_embed)   # :617-627
            x = self.pos_drop(x)
        else:
            x, tokens_per_patch, h, w, _ = self.patch_embed(x)
        self._tokens_per_patch = tokens_per_patch       # what every block of this forward pass is called with
        if self.training:
            DropPath.predraw([l.drop_path for l in self.layers] + [self.drop_path], x.shape[0], x.device)
        return x, (h, w)

    def _run_layers(self, hidden_states, residual, lo, hi, inference_params=None):
        for layer in self.layers[lo:hi]:
            hidden_states, residual = layer(hidden_states, self._tokens_per_patch, residual,
                                            inference_params=inference_params)
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
            return hidden_states.mean(dim=1)
        elif self.final_pool_type in ("max", "all"):
            return hidden_states
        raise NotImplementedError

    def _head(self, x):
        x = linear_module(self.head, x)
        if self.final_pool_type == "max":
            x = x.max(dim=1)[0]
        return x

    def forward_features(self, x, inference_params=None):
        hidden_states, _ = self._embed(x)
        hidden_states, residual = self._run_layers(hidden_states, None, 0, len(self.layers), inference_params)
        return self._final(hidden_states, residual)

    def forward(self, x, return_features=False, inference_params=None):
        x = self.forward_features(x, inference_params)
        if return_features:
            return x
        return self._head(x)


def channelvim_small_patch16_224_final_pool_mean_abs_pos_embed_with_noclstok_div2(
        pretrained=False, patch_size=16, stride=16, if_abs_pos_embed=True, **kwargs):
    """FastChannelVim-S/16 (models_channel_mamba_faster.py:685-706)."""
    if pretrained:
        raise RuntimeError("no pretrained FastChannelVim weights are published (reference url is 'to.do')")
    model = VisionMamba(patch_size=patch_size, stride=stride, if_abs_pos_embed=if_abs_pos_embed, embed_dim=384,
                        depth=24, rms_norm=True, residual_in_fp32=True, fused_add_norm=True, **kwargs)
    model.default_cfg = {}
    return model
