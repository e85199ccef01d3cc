"""ChannelVim -- the UN-POOLED channel-wise-tokenization baseline with a middle class token: mirror of
models/channel_wise_tokenization/models_channel_mamba.py (``PatchEmbedPerChannel`` :22-190, ``Block`` :224-309,
``create_block`` :312-350, ``VisionMamba`` :400-619, factory :622-644; the model cell_imaging/config/ChannelVimS.yaml
trains).  Same constructor kwargs, attribute names, ``state_dict`` keys (``cls_token``, ``pos_embed`` without a class
slot, ``patch_embed.channel_embed``) and factory name.

Every block is the plain Vim block on the un-pooled Vim mixer (fastvim_amd/vim.py, fastvim_amd/mamba_simple.py): the
sequence -- ``num_patches * channels + 1`` tokens, Channel-First -- goes through the same fused HIP kernels as the Vim
baseline, so the paper's FastChannelVim-vs-ChannelVim comparison (JUMP-CP shapes) runs on one code base, like
FastVim-vs-Vim does for classification (``bench.py`` ``other_configs``).
"""
from functools import partial

import torch
import torch.nn as nn

from .fastvim import DropPath, _init_weights, trunc_normal_
from .layernorm import RMSNorm, layer_norm_fn
from .mamba_simple_faster import half_io, linear_module
from .models_channel_mamba_faster import PatchEmbedPerChannel as _PatchEmbedPerChannel, segm_init_weights
from .vim import create_block


PatchEmbedPerChannel = _PatchEmbedPerChannel      # (models_channel_mamba.py:22-190: same parameters and token order; overlapping
                                                  #  patches -- stride != patch_size -- are not built)


class VisionMamba(nn.Module):
    def __init__(self, img_size=224, patch_size=16, stride=16, depth=24, embed_dim=192, channels=3, num_classes=1000,
                 ssm_cfg=None, drop_rate=0.0, drop_path_rate=0.1, norm_epsilon: float = 1e-5, rms_norm: bool = False,
                 initializer_cfg=None, fused_add_norm=False, residual_in_fp32=False, device=None, dtype=None,
                 final_pool_type="none", if_abs_pos_embed=False, if_cls_token=False, init_layer_scale=None,
                 scan_order="Channel-First", hcs=True, sort_channels=True, use_norm_after_ssm=True, **kwargs):
        factory_kwargs = {"device": device, "dtype": dtype}
        kwargs.update(factory_kwargs)
        super().__init__()
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.final_pool_type = final_pool_type
        self.if_abs_pos_embed = if_abs_pos_embed
        self.if_cls_token = if_cls_token
        self.channels = channels
        self.num_classes = num_classes
        self.d_model = self.num_features = self.embed_dim = embed_dim
        self.scan_order = scan_order
        self.patch_size = patch_size
        self.patch_embed = _PatchEmbedPerChannel(img_size=img_size, patch_size=patch_size, stride=stride, in_chans=channels,
                                                 embed_dim=embed_dim, hcs=hcs, scan_order=scan_order,
                                                 sort_channels=sort_channels)
        num_patches = self.patch_embed.num_patches
        if if_cls_token:
            self.cls_token = nn.Parameter(torch.zeros(1, 1, self.embed_dim))
        if if_abs_pos_embed:
            self.pos_embed = nn.Parameter(torch.zeros(1, num_patches, self.embed_dim))
            self.pos_drop = nn.Dropout(p=drop_rate)
        if if_cls_token:          # (the reference builds the head only with a class token, :470-475)
            self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]      # stochastic depth decay rule
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
        return {"pos_embed", "pos_embed_obj", "cls_token", "dist_token", "cls_token_head", "cls_token_tail"}

    def forward_features(self, x, inference_params=None):
        """:553-611.  The position embedding (one row per PATCH, shared by the patch's channel tokens) is added in the
        patch embedding's epilogue; the class token sits in the middle of the sequence and is what comes back."""
        if self.if_abs_pos_embed:
            x, tokens_per_patch, h, w, _ = self.patch_embed(x, pos_embed=self.pos_embed)
            x = self.pos_drop(x)
        else:
            x, tokens_per_patch, h, w, _ = self.patch_embed(x)
        B, M, _ = x.shape
        if not self.if_cls_token:
            raise NotImplementedError("ChannelVim returns its class token (reference :609-611)")
        token_position = M // 2           # "in channel mamba it will be middle channel middle token"
        cls_token = self.cls_token.expand(B, -1, -1).to(x.dtype)
        x = torch.cat((x[:, :token_position, :], cls_token, x[:, token_position:, :]), dim=1)
        residual = None
        hidden_states = x
        if self.training:
            DropPath.predraw([l.drop_path for l in self.layers] + [self.drop_path], B, x.device)
        for layer in self.layers:
            hidden_states, residual = layer(hidden_states, residual, inference_params=inference_params)
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
        return hidden_states[:, token_position, :]

    def forward(self, x, return_features=False, inference_params=None):
        half = half_io(x)
        x = self.forward_features(x, inference_params)
        if not return_features:
            x = linear_module(self.head, x)
        return x.to(torch.float16) if half else x


def channelvim_small_patch16_224_final_pool_mean_abs_pos_embed_with_midclstok_div2(
        pretrained=False, patch_size=16, stride=16, if_abs_pos_embed=True, **kwargs):
    """ChannelVim-S/16 (models_channel_mamba.py:622-644)."""
    if pretrained:
        raise RuntimeError("no pretrained ChannelVim weights are published (reference url is 'to.do')")
    model = VisionMamba(patch_size=patch_size, stride=stride, if_abs_pos_embed=if_abs_pos_embed, embed_dim=384, depth=24,
                        rms_norm=True, residual_in_fp32=True, fused_add_norm=True, final_pool_type="mean",
                        if_cls_token=True, **kwargs)
    model.default_cfg = {}
    return model
