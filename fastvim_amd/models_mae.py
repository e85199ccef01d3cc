"""FastVim masked autoencoder (MAE pre-training) -- drop-in for ``MaskedAutoencoderViM`` and its factories of
models/mae/models_mamba_faster_mae_vimdecoder.py:515-950.

Same constructor kwargs, ``state_dict`` keys and initialisation as the reference.  The encoder runs only the kept
tokens through ``Mamba_masked`` mixers (fastvim_amd/mamba_simple_masked_faster.py); the decoder is the un-pooled Vim
mixer over the full token grid (fastvim_amd/mamba_simple.py); both on the fused HIP kernels.  Host-side index
bookkeeping (argsort of the masking noise, the odd-layer re-ordering of the kept tokens) stays in torch, as in the
reference.

One extension over the reference API: ``random_masking`` / ``forward`` / ``forward_encoder`` accept ``noise``
(N, L) -- the per-token uniform scores whose argsort decides which tokens are kept -- so that a run can be
reproduced across devices (torch's CPU and GPU generators differ); ``noise=None`` draws it exactly like the
reference (``torch.rand(N, L, device=x.device)``, :750).
"""
import math
from functools import partial
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

from .fastvim import PatchEmbed, _compute_dtype, _init_weights, trunc_normal_
from .layernorm import RMSNorm, layer_norm_fn
from .mamba_simple_faster import linear_module
from .mamba_simple_masked_faster import Mamba_masked
from .vim import create_block


def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    """(M,) positions -> (M, D) [sin | cos] features (models_mamba_faster_mae_vimdecoder.py:53-71)."""
    assert embed_dim % 2 == 0
    omega = np.arange(embed_dim // 2, dtype=np.float32)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size):
    """Fixed 2-D sin-cos position embedding, (grid_size**2, embed_dim); the w coordinate feeds the first half
    (models_mamba_faster_mae_vimdecoder.py:25-50)."""
    assert embed_dim % 2 == 0
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape([2, 1, grid_size, grid_size])
    emb_h = get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0])
    emb_w = get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])
    return np.concatenate([emb_h, emb_w], axis=1)


class Block_masked(nn.Module):
    """Add -> (RMS/Layer)Norm -> masked mixer on the kept tokens (models_mamba_faster_mae_vimdecoder.py:279-401).
    Odd layers see the transposed grid: the kept ids are mapped to the transposed numbering, the kept tokens are
    re-sorted into that scan order for the mixer and restored afterwards (:372-394)."""

    def __init__(self, dim, mixer_cls, norm_cls=nn.LayerNorm, fused_add_norm=False, residual_in_fp32=False,
                 rotate_every_block=True, layer_idx=None, token_size=None):
        super().__init__()
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.mixer = mixer_cls(dim)
        self.norm = norm_cls(dim)
        self.rotate_every_block = rotate_every_block
        self.layer_idx = layer_idx
        self.token_size = token_size
        if self.fused_add_norm:
            assert isinstance(self.norm, (nn.LayerNorm, RMSNorm)), \
                "Only LayerNorm and RMSNorm are supported for fused_add_norm"
        self.rotate_indices = self.compute_rotate_indices(*token_size)

    @staticmethod
    def compute_rotate_indices(H, W):
        indices = torch.arange(H * W, dtype=torch.long)
        i, j = indices.div(W, rounding_mode="floor"), indices % W
        return j * H + i

    def forward(self, hidden_states: Tensor, residual: Optional[Tensor] = None, ids_keep=None, inference_params=None):
        cdt = _compute_dtype(hidden_states)
        is_rms = isinstance(self.norm, RMSNorm)
        if self.fused_add_norm:
            hidden_states, residual = layer_norm_fn(
                hidden_states, self.norm.weight, self.norm.bias, residual=residual, eps=self.norm.eps,
                prenorm=True, residual_in_fp32=self.residual_in_fp32, is_rms_norm=is_rms, out_dtype=cdt)
        else:
            residual = hidden_states if residual is None else residual + hidden_states
            hidden_states = layer_norm_fn(residual.to(self.norm.weight.dtype), self.norm.weight, self.norm.bias,
                                          eps=self.norm.eps, is_rms_norm=is_rms, out_dtype=cdt)
            if self.residual_in_fp32:
                residual = residual.to(torch.float32)
        rot = self.rotate_every_block is True and self.layer_idx % 2 != 0
        if rot:
            if self.rotate_indices.device != hidden_states.device:
                self.rotate_indices = self.rotate_indices.to(hidden_states.device)
            ids_keep = self.rotate_indices[ids_keep]
            ids_keep, order = torch.sort(ids_keep, dim=1)        # kept tokens in the transposed grid's scan order
            inverse = torch.argsort(order, 1)
            d = hidden_states.shape[-1]
            hidden_states = torch.gather(hidden_states, 1, order.unsqueeze(-1).expand(-1, -1, d))
        hidden_states = self.mixer(hidden_states, ids_keep, inference_params=inference_params)
        if rot:
            hidden_states = torch.gather(hidden_states, 1, inverse.unsqueeze(-1).expand(-1, -1, d))
        return hidden_states, residual

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        raise NotImplementedError("FastVim mixers have no inference cache")


def create_block_masked(d_model, ssm_cfg=None, norm_epsilon=1e-5, rms_norm=False, residual_in_fp32=False,
                        fused_add_norm=False, layer_idx=None, device=None, dtype=None, init_layer_scale=None,
                        scanpath_type="rowwise", use_norm_after_ssm=True, rotate_every_block=True,
                        collapse_method="mean", token_size=None):
    """models_mamba_faster_mae_vimdecoder.py:404-465: odd layers get the swapped token_size."""
    if ssm_cfg is None:
        ssm_cfg = {}
    factory_kwargs = {"device": device, "dtype": dtype}
    rot = rotate_every_block is True and layer_idx % 2 != 0
    mixer_cls = partial(Mamba_masked, layer_idx=layer_idx, init_layer_scale=init_layer_scale,
                        scanpath_type=scanpath_type, use_norm_after_ssm=use_norm_after_ssm,
                        token_size=[token_size[1], token_size[0]] if rot else list(token_size),
                        collapse_method=collapse_method, **ssm_cfg, **factory_kwargs)
    norm_cls = partial(nn.LayerNorm if not rms_norm else RMSNorm, eps=norm_epsilon, **factory_kwargs)
    block = Block_masked(d_model, mixer_cls, norm_cls=norm_cls, fused_add_norm=fused_add_norm,
                         residual_in_fp32=residual_in_fp32, rotate_every_block=rotate_every_block,
                         layer_idx=layer_idx, token_size=token_size)
    block.layer_idx = layer_idx
    return block


class MaskedAutoencoderViM(nn.Module):
    def __init__(self, img_size=224, patch_size=16, stride=16, depth=24, embed_dim=192, decoder_embed_dim=512,
                 decoder_depth=2, norm_pix_loss=True, channels=3, ssm_cfg=None, drop_rate=0.0,
                 norm_epsilon: float = 1e-5, rms_norm: bool = False, initializer_cfg=None, fused_add_norm=False,
                 residual_in_fp32=False, device=None, dtype=None, init_layer_scale=None, use_norm_after_ssm=True,
                 embed_layer=PatchEmbed, scanpath_type="rowwise", rotate_every_block=True, collapse_method="mean",
                 **kwargs):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.rotate_every_block = rotate_every_block
        self.d_model = self.num_features = self.embed_dim = embed_dim
        self.patch_size = patch_size
        in_chans = channels
        # ---- encoder
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      strict_img_size=False, dynamic_img_pad=True, scanpath_type=scanpath_type)
        num_patches = self.patch_embed.num_patches
        self.num_patches = num_patches
        self.token_size = self.patch_embed.grid_size
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches, embed_dim), requires_grad=False)   # fixed sin-cos
        self.layers = nn.ModuleList([
            create_block_masked(embed_dim, ssm_cfg=ssm_cfg, norm_epsilon=norm_epsilon, rms_norm=rms_norm,
                                residual_in_fp32=residual_in_fp32, fused_add_norm=fused_add_norm, layer_idx=i,
                                init_layer_scale=init_layer_scale, scanpath_type=scanpath_type,
                                use_norm_after_ssm=use_norm_after_ssm, rotate_every_block=rotate_every_block,
                                collapse_method=collapse_method, token_size=self.token_size, **factory_kwargs)
            for i in range(depth)])
        self.norm_f = (nn.LayerNorm if not rms_norm else RMSNorm)(embed_dim, eps=norm_epsilon, **factory_kwargs)
        # ---- decoder
        self.decoder_embed = nn.Linear(embed_dim, decoder_embed_dim, bias=True)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, num_patches, decoder_embed_dim), requires_grad=False)
        self.decoder_blocks = nn.ModuleList([
            create_block(decoder_embed_dim, ssm_cfg=ssm_cfg, norm_epsilon=norm_epsilon, rms_norm=rms_norm,
                         residual_in_fp32=residual_in_fp32, fused_add_norm=fused_add_norm, layer_idx=i,
                         use_norm_after_ssm=use_norm_after_ssm, init_layer_scale=init_layer_scale, **factory_kwargs)
            for i in range(decoder_depth)])
        self.decoder_norm = (nn.LayerNorm if not rms_norm else RMSNorm)(decoder_embed_dim, eps=norm_epsilon,
                                                                        **factory_kwargs)
        self.decoder_pred = nn.Linear(decoder_embed_dim, patch_size ** 2 * in_chans, bias=True)
        self.norm_pix_loss = norm_pix_loss
        # ---- initialisation, in the reference's order (:650-682)
        g = int(num_patches ** 0.5)
        self.pos_embed.data.copy_(torch.from_numpy(get_2d_sincos_pos_embed(embed_dim, g)).float().unsqueeze(0))
        self.decoder_pos_embed.data.copy_(
            torch.from_numpy(get_2d_sincos_pos_embed(decoder_embed_dim, g)).float().unsqueeze(0))
        w = self.patch_embed.proj.weight.data
        torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))          # like nn.Linear, not nn.Conv2d
        trunc_normal_(self.mask_token, std=0.02)
        self.decoder_embed.apply(self._init_weights_decoder)
        self.decoder_norm.apply(self._init_weights_decoder)
        self.decoder_pred.apply(self._init_weights_decoder)
        icfg = initializer_cfg if initializer_cfg is not None else {}
        self.decoder_blocks.apply(partial(_init_weights, n_layer=decoder_depth, **icfg))
        self.apply(partial(_init_weights, n_layer=depth, **icfg))

    def _init_weights_decoder(self, m):
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "dist_token"}

    def patchify(self, imgs):
        """(N, 3, H, W) -> (N, L, patch_size**2 * 3)"""
        p = self.patch_embed.patch_size[0]
        assert imgs.shape[2] == imgs.shape[3] and imgs.shape[2] % p == 0
        h = w = imgs.shape[2] // p
        x = imgs.reshape(imgs.shape[0], 3, h, p, w, p)
        x = x.permute(0, 2, 4, 3, 5, 1)                 # "nchpwq->nhwpqc": a pure permutation, no contraction
        return x.reshape(imgs.shape[0], h * w, p ** 2 * 3)

    def unpatchify(self, x):
        """(N, L, patch_size**2 * 3) -> (N, 3, H, W)"""
        p = self.patch_embed.patch_size[0]
        h = w = int(x.shape[1] ** 0.5)
        assert h * w == x.shape[1]
        x = x.reshape(x.shape[0], h, w, p, p, 3)
        x = x.permute(0, 5, 1, 3, 2, 4)                 # "nhwpqc->nchpwq"
        return x.reshape(x.shape[0], 3, h * p, h * p)

    def random_masking(self, x, mask_ratio, noise=None):
        """Per-sample random masking by argsort of uniform noise; the kept ids are sorted ascending because the
        mixers are sequential (models_mamba_faster_mae_vimdecoder.py:738-772)."""
        N, L, D = x.shape
        len_keep = int(L * (1 - mask_ratio))
        if noise is None:
            noise = torch.rand(N, L, device=x.device)
        ids_shuffle = torch.argsort(noise, dim=1)          # ascend: small is keep, large is remove
        ids_shuffle[:, :len_keep] = ids_shuffle[:, :len_keep].sort().values
        ids_shuffle = ids_shuffle.contiguous()
        ids_restore = torch.argsort(ids_shuffle, dim=1)
        ids_keep = ids_shuffle[:, :len_keep]
        x_masked = torch.gather(x, dim=1, index=ids_keep.unsqueeze(-1).repeat(1, 1, D))
        mask = torch.ones([N, L], device=x.device)
        mask[:, :len_keep] = 0
        mask = torch.gather(mask, dim=1, index=ids_restore)   # 0 is keep, 1 is remove
        return x_masked, mask, ids_restore, ids_keep

    def _final_norm(self, norm, hidden_states, residual):
        is_rms = isinstance(norm, RMSNorm)
        if not self.fused_add_norm:
            residual = hidden_states if residual is None else residual + hidden_states
            return layer_norm_fn(residual.to(norm.weight.dtype), norm.weight, norm.bias, eps=norm.eps,
                                 is_rms_norm=is_rms)
        return layer_norm_fn(hidden_states, norm.weight, norm.bias, eps=norm.eps, residual=residual, prenorm=False,
                             residual_in_fp32=self.residual_in_fp32, is_rms_norm=is_rms)

    def forward_encoder(self, x, mask_ratio, inference_params=None, noise=None):
        x = self.patch_embed(x, self.pos_embed)
        x, mask, ids_restore, ids_keep = self.random_masking(x, mask_ratio, noise)
        residual = None
        hidden_states = x
        for layer in self.layers:
            hidden_states, residual = layer(hidden_states, residual, ids_keep.clone(), inference_params=inference_params)
        return self._final_norm(self.norm_f, hidden_states, residual), mask, ids_restore

    def forward_decoder(self, x, ids_restore, inference_params=None):
        x = linear_module(self.decoder_embed, x)
        mask_tokens = self.mask_token.repeat(x.shape[0], ids_restore.shape[1] - x.shape[1], 1)
        x = torch.cat([x, mask_tokens.to(x.dtype)], dim=1)
        x = torch.gather(x, dim=1, index=ids_restore.unsqueeze(-1).repeat(1, 1, x.shape[2]))    # unshuffle
        x = x + self.decoder_pos_embed
        residual = None
        for layer in self.decoder_blocks:
            x, residual = layer(x, residual, inference_params=inference_params)
        x = self._final_norm(self.decoder_norm, x, residual)
        return linear_module(self.decoder_pred, x)

    def forward_loss(self, imgs, pred, mask):
        """imgs (N, 3, H, W); pred (N, L, p*p*3); mask (N, L), 1 = removed: mean squared error on removed patches."""
        target = self.patchify(imgs)
        if self.norm_pix_loss:
            mean = target.mean(dim=-1, keepdim=True)
            var = target.var(dim=-1, keepdim=True)
            target = (target - mean) / (var + 1.0e-6) ** 0.5
        loss = ((pred - target) ** 2).mean(dim=-1)
        return (loss * mask).sum() / mask.sum()

    def forward(self, imgs, mask_ratio=0.75, inference_params=None, noise=None):
        latent, mask, ids_restore = self.forward_encoder(imgs, mask_ratio, inference_params, noise=noise)
        pred = self.forward_decoder(latent, ids_restore, inference_params)
        loss = self.forward_loss(imgs, pred, mask)
        return loss, pred, mask


def _mae(embed_dim, depth, patch_size, stride, kwargs):
    return MaskedAutoencoderViM(patch_size=patch_size, stride=stride, embed_dim=embed_dim, depth=depth,
                                decoder_embed_dim=512, decoder_depth=2, rms_norm=True, residual_in_fp32=True,
                                fused_add_norm=True, **kwargs)


def mae_FastVim_base_dec512d2b(patch_size=16, stride=16, **kwargs):
    return _mae(768, 24, patch_size, stride, kwargs)


def mae_FastVim_large_dec512d2b(patch_size=16, stride=16, **kwargs):
    return _mae(1024, 48, patch_size, stride, kwargs)


def mae_FastVim_huge_dec512d2b(patch_size=14, stride=14, **kwargs):
    return _mae(1280, 64, patch_size, stride, kwargs)
