"""Vim baseline backbone: mirror of models/vim.py (``PatchEmbed`` :25-84, ``Block`` :87-172,
``create_block`` :175-212, ``VisionMamba`` :263-508, factories :642-775) -- same constructor kwargs, attribute
names, ``state_dict`` keys (incl. ``cls_token``) and entry points ``vim_{tiny,small,base,large}_patch16_224_
final_pool_mean_abs_pos_embed_with_midclstok_div2``.

The blocks run the un-pooled Vim mixer (fastvim_amd/mamba_simple.py) on the fused HIP kernels, so the paper's
FastVim-vs-Vim comparison (README.md:15) can be reproduced on one code base.  ``MM_Vim`` (mmdet / mmseg wrapper)
is not provided.
"""
import math
from functools import partial

import torch
import torch.nn as nn

from .fastvim import (Block as _FastVimBlock, DropPath, PatchEmbed as _PatchEmbed, _init_weights, segm_init_weights,
                      trunc_normal_)
from .layernorm import RMSNorm, layer_norm_fn
from .mamba_simple import Mamba
from .mamba_simple_faster import linear_module


class PatchEmbed(_PatchEmbed):
    """2D Image to Patch Embedding (models/vim.py:25-84): row-major tokens, no scan-path option."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True,
                 strict_img_size=True, dynamic_img_pad=False):
        super().__init__(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                         norm_layer=norm_layer, flatten=flatten, strict_img_size=strict_img_size,
                         dynamic_img_pad=dynamic_img_pad, scanpath_type="rowwise")


class Block(_FastVimBlock):
    """Add -> (RMS/Layer)Norm -> Mixer (models/vim.py:87-172); no grid rotation in Vim."""

    def __init__(self, dim, mixer_cls, norm_cls=nn.LayerNorm, fused_add_norm=False, residual_in_fp32=False,
                 drop_path=0.0):
        super().__init__(dim, mixer_cls, norm_cls=norm_cls, fused_add_norm=fused_add_norm,
                         residual_in_fp32=residual_in_fp32, drop_path=drop_path, rotate_every_block=False,
                         layer_idx=0, token_size=None)

    def _mix(self, hidden_states, inference_params, rot):
        return self.mixer(hidden_states, inference_params=inference_params)


def create_block(d_model, ssm_cfg=None, norm_epsilon=1e-5, drop_path=0.0, rms_norm=False, residual_in_fp32=False,
                 fused_add_norm=False, layer_idx=None, device=None, dtype=None, use_norm_after_ssm=True,
                 init_layer_scale=None):
    if ssm_cfg is None:
        ssm_cfg = {}
    factory_kwargs = {"device": device, "dtype": dtype}
    mixer_cls = partial(Mamba, layer_idx=layer_idx, init_layer_scale=init_layer_scale,
                        use_norm_after_ssm=use_norm_after_ssm, **ssm_cfg, **factory_kwargs)
    norm_cls = partial(nn.LayerNorm if not rms_norm else RMSNorm, eps=norm_epsilon, **factory_kwargs)
    block = Block(d_model, mixer_cls, norm_cls=norm_cls, drop_path=drop_path, fused_add_norm=fused_add_norm,
                  residual_in_fp32=residual_in_fp32)
    block.layer_idx = layer_idx
    return block


class VisionMamba(nn.Module):
    def __init__(self, img_size=224, patch_size=16, stride=16, depth=24, embed_dim=192, channels=3,
                 num_classes=1000, ssm_cfg=None, drop_rate=0.0, drop_path_rate=0.1, norm_epsilon: float = 1e-5,
                 rms_norm: bool = True, initializer_cfg=None, fused_add_norm=False, residual_in_fp32=False,
                 device=None, dtype=None, final_pool_type="none", if_abs_pos_embed=True, if_cls_token=True,
                 init_layer_scale=None, use_middle_cls_token=True, use_norm_after_ssm=True, embed_layer=PatchEmbed,
                 **kwargs):
        factory_kwargs = {"device": device, "dtype": dtype}
        kwargs.update(factory_kwargs)
        super().__init__()
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.final_pool_type = final_pool_type
        self.if_abs_pos_embed = if_abs_pos_embed
        self.if_cls_token = if_cls_token
        self.use_middle_cls_token = use_middle_cls_token
        self.num_tokens = 1 if if_cls_token else 0
        self.num_classes = num_classes
        self.d_model = self.num_features = self.embed_dim = embed_dim
        self.patch_size = patch_size
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=channels,
                                      embed_dim=embed_dim, strict_img_size=False, dynamic_img_pad=True)
        self.num_patches = self.patch_embed.num_patches
        self.token_size = self.patch_embed.grid_size
        if if_cls_token:
            self.cls_token = nn.Parameter(torch.zeros(1, 1, self.embed_dim))
        if if_abs_pos_embed:
            self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches + self.num_tokens, self.embed_dim))
            self.pos_drop = nn.Dropout(p=drop_rate)
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        inter_dpr = [0.0] + dpr
        self.drop_path = DropPath(drop_path_rate) if drop_path_rate > 0.0 else nn.Identity()
        self.layers = nn.ModuleList([
            create_block(embed_dim, ssm_cfg=ssm_cfg, norm_epsilon=norm_epsilon, rms_norm=rms_norm,
                         residual_in_fp32=residual_in_fp32, fused_add_norm=fused_add_norm, layer_idx=i,
                         drop_path=inter_dpr[i], use_norm_after_ssm=use_norm_after_ssm,
                         init_layer_scale=init_layer_scale, **factory_kwargs)
            for i in range(depth)])
        self.norm_f = (nn.LayerNorm if not rms_norm else RMSNorm)(embed_dim, eps=norm_epsilon, **factory_kwargs)
        self.patch_embed.apply(segm_init_weights)
        self.head.apply(segm_init_weights)
        if if_abs_pos_embed:
            trunc_normal_(self.pos_embed, std=0.02)
        if if_cls_token:
            trunc_normal_(self.cls_token, std=0.02)
        self.apply(partial(_init_weights, n_layer=depth, **(initializer_cfg if initializer_cfg is not None else {})))

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "cls_token", "dist_token", "cls_token_head", "cls_token_tail"}

    def forward_features(self, x, inference_params=None, out_indices=None):
        B, _, H, W = x.shape
        x = self.patch_embed(x)                        # (B, M, D) fp32 (conv bias added in the epilogue)
        M = x.shape[1]
        token_position = 0
        if self.if_cls_token:                          # :417-433
            cls_token = self.cls_token.expand(B, -1, -1).to(x.dtype)
            if self.use_middle_cls_token:
                token_position = M // 2
                x = torch.cat((x[:, :token_position, :], cls_token, x[:, token_position:, :]), dim=1)
            else:
                x = torch.cat((cls_token, x), dim=1)
        if self.if_abs_pos_embed:
            Hg, Wg = math.ceil(H / self.patch_size), math.ceil(W / self.patch_size)
            if Hg != self.token_size[0] or Wg != self.token_size[1]:
                raise RuntimeError(f"input grid {Hg}x{Wg} differs from the model's {self.token_size}; "
                                   "build VisionMamba with the matching img_size")
            x = x + self.pos_embed
            x = self.pos_drop(x)
        outs = []
        residual = None
        hidden_states = x
        if self.training:
            DropPath.predraw([l.drop_path for l in self.layers] + [self.drop_path], x.shape[0], x.device)
        for layer_idx, layer in enumerate(self.layers):
            hidden_states, residual = layer(hidden_states, residual, inference_params=inference_params)
            if out_indices is not None and layer_idx in out_indices:
                outs.append(hidden_states)
        if out_indices is not None:
            assert len(outs) == len(out_indices)
            return outs, (math.ceil(H / self.patch_size), math.ceil(W / self.patch_size))
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
        if self.if_cls_token:                          # :478-480
            return hidden_states[:, token_position, :]
        if self.final_pool_type == "none":
            return hidden_states[:, -1, :]
        elif self.final_pool_type == "mean":
            return hidden_states.mean(dim=1)
        elif self.final_pool_type in ("max", "all"):
            return hidden_states
        raise NotImplementedError

    def forward(self, x, return_features=False, inference_params=None):
        x = self.forward_features(x, inference_params)
        if return_features:
            return x
        x = linear_module(self.head, x)
        if self.final_pool_type == "max":
            x = x.max(dim=1)[0]
        return x


class MM_Vim(VisionMamba):
    """Multi-scale feature backbone on the Vim baseline for detection / segmentation heads (models/vim.py:511-639),
    without the mmdet / mmseg registry decorators.  Built WITHOUT a class token (dense prediction); ``if_cls_token``
    only says whether the checkpoint being loaded has one in the middle of its ``pos_embed``, which is then cut out
    before the table is bicubically resized to this model's grid (:560-583).  ``forward(x)`` returns the LayerNorm-ed
    hidden states of ``out_indices`` as (B, C, H, W) maps."""

    def __init__(self, img_size=224, patch_size=16, stride=16, in_chans=3, embed_dim=192, depth=24, if_cls_token=True,
                 use_middle_cls_token=False, pretrained=None, out_indices=(5, 11, 17, 23), load_ema=True, **kwargs):
        super().__init__(img_size, patch_size, stride, depth, embed_dim, in_chans, if_cls_token=False,
                         use_middle_cls_token=False, **kwargs)
        self.remove_cls_token = if_cls_token
        self.load_ema = load_ema
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
            pe = sd["pos_embed"]
            if self.remove_cls_token:
                pos_size = int(math.sqrt(pe.shape[1] - 1))
                mid = (pe.shape[1] - 1) // 2
                pe = torch.cat([pe[:, :mid, :], pe[:, mid + 1:, :]], dim=1)
            else:
                pos_size = int(math.sqrt(pe.shape[1]))
            sd["pos_embed"] = self.resize_pos_embed(pe, self.token_size, (pos_size, pos_size), "bicubic")
        return self.load_state_dict(sd, strict=False)

    @staticmethod
    def resize_pos_embed(pos_embed, input_shape, pos_shape, mode):
        """(1, L, C) position table of a ``pos_shape`` grid -> ``input_shape`` grid (models/vim.py:591-623)."""
        assert pos_embed.ndim == 3, "shape of pos_embed must be [B, L, C]"
        pos_h, pos_w = pos_shape
        w = pos_embed.reshape(1, pos_h, pos_w, pos_embed.shape[2]).permute(0, 3, 1, 2)
        w = torch.nn.functional.interpolate(w, size=tuple(input_shape), mode=mode, align_corners=False)
        return torch.flatten(w, 2).transpose(1, 2)

    def forward(self, x):
        C = self.embed_dim
        outs, (H, W) = self.forward_features(x, out_indices=self.out_indices)
        outs = [getattr(self, f"outnorm_{i}")(o.float()) for i, o in enumerate(outs)]
        outs = [o.view(-1, H, W, C).permute(0, 3, 1, 2).contiguous() for o in outs]
        return outs[0] if len(self.out_indices) == 1 else outs


def _factory(embed_dim, depth, patch_size, stride, kwargs):
    model = VisionMamba(patch_size=patch_size, stride=stride, embed_dim=embed_dim, depth=depth, rms_norm=True,
                        residual_in_fp32=True, fused_add_norm=True, final_pool_type="mean", if_abs_pos_embed=True,
                        if_cls_token=True, use_middle_cls_token=True, **kwargs)
    model.default_cfg = {}
    return model


def vim_tiny_patch16_224_final_pool_mean_abs_pos_embed_with_midclstok_div2(pretrained=False, patch_size=16, stride=16,
                                                                           **kwargs):
    assert not pretrained, "no pretrained weights are bundled"
    return _factory(192, 24, patch_size, stride, kwargs)


def vim_small_patch16_224_final_pool_mean_abs_pos_embed_with_midclstok_div2(pretrained=False, patch_size=16, stride=16,
                                                                            **kwargs):
    assert not pretrained, "no pretrained weights are bundled"
    return _factory(384, 24, patch_size, stride, kwargs)


def vim_base_patch16_224_final_pool_mean_abs_pos_embed_with_midclstok_div2(pretrained=False, patch_size=16, stride=16,
                                                                           **kwargs):
    assert not pretrained, "no pretrained weights are bundled"
    return _factory(768, 24, patch_size, stride, kwargs)


def vim_large_patch16_224_final_pool_mean_abs_pos_embed_with_midclstok_div2(pretrained=False, patch_size=16, stride=16,
                                                                            **kwargs):
    assert not pretrained, "no pretrained weights are bundled"
    return _factory(1024, 48, patch_size, stride, kwargs)
