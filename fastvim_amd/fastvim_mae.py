"""Vim masked autoencoder -- the UN-POOLED MAE baseline with a middle class token: mirror of models/mae/fastvim_mae.py
(``MaskedAutoencoderViM`` :309-709, factories ``mae_vim_{base,large,huge}_dec512d2b`` :714-767; the model
mae/config/pretrain_VimB.yaml pre-trains).  Same constructor kwargs, ``state_dict`` keys (``cls_token``, ``pos_embed`` /
``decoder_pos_embed`` with a leading class slot) and initialisation order as the reference.

Encoder and decoder are plain Vim blocks on the un-pooled Vim mixer (fastvim_amd/vim.py, fastvim_amd/mamba_simple.py) --
the encoder over the kept tokens with the class token in their middle, the decoder over the full grid with the class
token appended -- on the same fused HIP kernels as the FastVim MAE (fastvim_amd/models_mae.py), whose host-side helpers
(patchify, masking with a ``noise`` hook, loss, final norm) it inherits.
"""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from .fastvim import _init_weights, trunc_normal_
from .layernorm import RMSNorm
from .mamba_simple_faster import linear_module
from .models_mae import MaskedAutoencoderViM as _FastVimMAE, get_2d_sincos_pos_embed as _sincos_grid
from .vim import PatchEmbed, create_block


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    """(grid_size**2 (+ 1), embed_dim); ``cls_token``: a zero row in front (fastvim_mae.py:25-40)."""
    pe = _sincos_grid(embed_dim, grid_size)
    if cls_token:
        pe = np.concatenate([np.zeros([1, embed_dim]), pe], axis=0)
    return pe


class MaskedAutoencoderViM(_FastVimMAE):
    def __init__(self, img_size=224, patch_size=16, stride=16, depth=24, embed_dim=192, decoder_embed_dim=512,
                 decoder_depth=8, norm_pix_loss=True, channels=3, ssm_cfg=None, drop_rate=0.0,
                 norm_epsilon: float = 1e-5, rms_norm: bool = False, initializer_cfg=None, fused_add_norm=False,
                 residual_in_fp32=False, device=None, dtype=None, init_layer_scale=None, use_norm_after_ssm=True,
                 embed_layer=PatchEmbed, **kwargs):
        factory_kwargs = {"device": device, "dtype": dtype}
        nn.Module.__init__(self)          # (the FastVim MAE's constructor builds pooled mixers: not wanted here)
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.d_model = self.num_features = self.embed_dim = embed_dim
        self.patch_size = patch_size
        in_chans = channels
        # ---- encoder (:349-388)
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      strict_img_size=False, dynamic_img_pad=True)
        num_patches = self.patch_embed.num_patches
        self.num_patches = num_patches
        self.token_size = self.patch_embed.grid_size
        self.cls_token = nn.Parameter(torch.zeros(1, 1, self.embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim), requires_grad=False)   # fixed sin-cos
        blk = partial(create_block, ssm_cfg=ssm_cfg, norm_epsilon=norm_epsilon, rms_norm=rms_norm,
                      residual_in_fp32=residual_in_fp32, fused_add_norm=fused_add_norm,
                      use_norm_after_ssm=use_norm_after_ssm, init_layer_scale=init_layer_scale, **factory_kwargs)
        self.layers = nn.ModuleList([blk(embed_dim, layer_idx=i) for i in range(depth)])
        self.norm_f = (nn.LayerNorm if not rms_norm else RMSNorm)(embed_dim, eps=norm_epsilon, **factory_kwargs)
        # ---- decoder (:392-426)
        self.decoder_embed = nn.Linear(embed_dim, decoder_embed_dim, bias=True)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, decoder_embed_dim), requires_grad=False)
        self.decoder_blocks = nn.ModuleList([blk(decoder_embed_dim, layer_idx=i) for i in range(decoder_depth)])
        self.decoder_norm = (nn.LayerNorm if not rms_norm else RMSNorm)(decoder_embed_dim, eps=norm_epsilon,
                                                                        **factory_kwargs)
        self.decoder_pred = nn.Linear(decoder_embed_dim, patch_size ** 2 * in_chans, bias=True)
        self.norm_pix_loss = norm_pix_loss
        # ---- initialisation, in the reference's order (:431-472)
        g = int(num_patches ** 0.5)
        self.pos_embed.data.copy_(torch.from_numpy(get_2d_sincos_pos_embed(embed_dim, g, cls_token=True)).float().unsqueeze(0))
        self.decoder_pos_embed.data.copy_(
            torch.from_numpy(get_2d_sincos_pos_embed(decoder_embed_dim, g, cls_token=True)).float().unsqueeze(0))
        w = self.patch_embed.proj.weight.data
        torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))          # like nn.Linear, not nn.Conv2d
        trunc_normal_(self.cls_token, std=0.02)
        trunc_normal_(self.mask_token, std=0.02)
        self.decoder_embed.apply(self._init_weights_decoder)
        self.decoder_norm.apply(self._init_weights_decoder)
        self.decoder_pred.apply(self._init_weights_decoder)
        icfg = initializer_cfg if initializer_cfg is not None else {}
        self.decoder_blocks.apply(partial(_init_weights, n_layer=decoder_depth, **icfg))
        self.apply(partial(_init_weights, n_layer=depth, **icfg))

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "cls_token", "dist_token"}

    def forward_encoder(self, x, mask_ratio, inference_params=None, noise=None):
        """:583-631: patch embedding + position (the class slot of ``pos_embed`` goes to the class token), masking, the
        class token in the MIDDLE of the kept tokens, Vim blocks, final norm."""
        x = self.patch_embed(x, self.pos_embed[:, 1:, :])
        x, mask, ids_restore, _ = self.random_masking(x, mask_ratio, noise)
        M = x.shape[1]
        cls_tokens = (self.cls_token + self.pos_embed[:, :1, :]).expand(x.shape[0], -1, -1).to(x.dtype)
        token_position = M // 2
        x = torch.cat((x[:, :token_position, :], cls_tokens, x[:, token_position:, :]), dim=1)
        residual = None
        hidden_states = x
        for layer in self.layers:
            hidden_states, residual = layer(hidden_states, residual, inference_params=inference_params)
        return self._final_norm(self.norm_f, hidden_states, residual), mask, ids_restore

    def forward_decoder(self, x, ids_restore, inference_params=None):
        """:633-691: mask tokens appended and unshuffled WITHOUT the class token, which is re-attached at the END of the
        sequence, carried through the decoder blocks and dropped from the prediction."""
        token_position = (x.shape[1] - 1) // 2
        x = linear_module(self.decoder_embed, x)
        mask_tokens = self.mask_token.repeat(x.shape[0], ids_restore.shape[1] + 1 - x.shape[1], 1).to(x.dtype)
        x_ = torch.cat([x[:, :token_position, :], x[:, token_position + 1:, :], mask_tokens], dim=1)      # no cls token
        x_ = torch.gather(x_, dim=1, index=ids_restore.unsqueeze(-1).repeat(1, 1, x.shape[2]))            # unshuffle
        x_ = x_ + self.decoder_pos_embed[:, 1:]
        cls_tokens = x[:, token_position:token_position + 1, :] + self.decoder_pos_embed[:, :1, :]
        n = x_.shape[1]
        x = torch.cat([x_, cls_tokens.to(x_.dtype)], dim=1)
        residual = None
        for layer in self.decoder_blocks:
            x, residual = layer(x, residual, inference_params=inference_params)
        x = self._final_norm(self.decoder_norm, x, residual)
        x = linear_module(self.decoder_pred, x)
        return x[:, :n, :]                                           # remove cls token


def _mae_vim(embed_dim, depth, patch_size, stride, kwargs):
    model = MaskedAutoencoderViM(patch_size=patch_size, stride=stride, embed_dim=embed_dim, depth=depth,
                                 decoder_embed_dim=512, decoder_depth=2, rms_norm=True, residual_in_fp32=True,
                                 fused_add_norm=True, **kwargs)
    model.default_cfg = {}
    return model


def mae_vim_base_dec512d2b(patch_size=16, stride=16, **kwargs):
    return _mae_vim(768, 24, patch_size, stride, kwargs)


def mae_vim_large_dec512d2b(patch_size=16, stride=16, **kwargs):
    return _mae_vim(1024, 48, patch_size, stride, kwargs)


def mae_vim_huge_dec512d2b(patch_size=16, stride=16, **kwargs):
    return _mae_vim(1280, 64, patch_size, stride, kwargs)
