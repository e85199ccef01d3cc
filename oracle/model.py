"""FastVim Block / backbone oracle (test infrastructure; see oracle/__init__.py).

Functional restatement over a reference-keyed ``state_dict`` of
``PatchEmbed.forward`` (models/fastvim.py:72-103), ``Block.forward`` (:146-212),
``VisionMamba.forward_features`` / ``forward`` (:484-557).
"""
import math
import zlib

import torch
import torch.nn.functional as F

from .mixer import fastvim_mixer_oracle
from .norm import fused_add_norm_oracle

MIXER_KEYS = (
    "A_log", "D", "A_b_log", "D_b", "in_proj.weight", "layernorm.weight", "layernorm.bias",
    "conv1d.weight", "conv1d.bias", "x_proj.weight", "dt_proj.weight", "dt_proj.bias",
    "conv1d_b.weight", "conv1d_b.bias", "x_proj_b.weight", "dt_proj_b.weight", "dt_proj_b.bias",
    "out_proj.weight",
)


def state_dict_shapes(embed_dim=192, depth=24, img_size=224, patch_size=16, channels=3,
                      num_classes=1000, d_state=16, d_conv=4, expand=2):
    """Key -> shape table of the reference FastVim ``state_dict``
    (checkpoint contract; models/fastvim.py:391-451,
    mamba_simple_faster.py:78-177)."""
    d, d_in = embed_dim, expand * embed_dim
    R = math.ceil(d / 16)
    ih, iw = (img_size, img_size) if isinstance(img_size, int) else img_size
    gh, gw = ih // patch_size, iw // patch_size
    shapes = {
        "pos_embed": (1, gh * gw, d),
        "patch_embed.proj.weight": (d, channels, patch_size, patch_size),
        "patch_embed.proj.bias": (d,),
        "head.weight": (num_classes, d),
        "head.bias": (num_classes,),
        "norm_f.weight": (d,),
    }
    for i in range(depth):
        pre = f"layers.{i}."
        shapes[pre + "norm.weight"] = (d,)
        m = pre + "mixer."
        shapes[m + "in_proj.weight"] = (2 * d_in, d)
        shapes[m + "out_proj.weight"] = (d, d_in)
        shapes[m + "layernorm.weight"] = (d_in,)
        shapes[m + "layernorm.bias"] = (d_in,)
        for sfx in ("", "_b"):
            shapes[m + f"A{sfx}_log"] = (d_in, d_state)
            shapes[m + f"D{sfx}"] = (d_in,)
            shapes[m + f"conv1d{sfx}.weight"] = (d_in, 1, d_conv)
            shapes[m + f"conv1d{sfx}.bias"] = (d_in,)
            shapes[m + f"x_proj{sfx}.weight"] = (R + 2 * d_state, d_in)
            shapes[m + f"dt_proj{sfx}.weight"] = (d_in, R)
            shapes[m + f"dt_proj{sfx}.bias"] = (d_in,)
    return shapes


def make_state_dict(seed=0, shapes=None, **cfg):
    """Deterministic synthetic parameters, re-derivable anywhere from (seed, key):
    each tensor is drawn from ``Generator().manual_seed(crc32(key) ^ seed)`` with a
    per-kind scale that keeps activations O(1) through 24 blocks.  Lets the 28 MB
    FastVim-T state be rebuilt on the GPU box instead of committed."""
    sd = {}
    for key, shape in (shapes if shapes is not None else state_dict_shapes(**cfg)).items():
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
        r = torch.randn(shape, generator=g, dtype=torch.float32)
        leaf = key.split("mixer.")[-1] if "mixer." in key else key
        if leaf.startswith("A") and leaf.endswith("_log"):
            n = shape[1]
            t = torch.log(torch.arange(1, n + 1, dtype=torch.float32))[None, :] + 0.05 * r
        elif leaf in ("D", "D_b"):
            t = 1.0 + 0.1 * r
        elif leaf.endswith("norm.weight") or leaf == "norm_f.weight":
            t = 1.0 + 0.1 * r
        elif leaf == "layernorm.bias":
            t = 0.05 * r
        elif leaf.startswith("dt_proj") and leaf.endswith(".bias"):
            u = torch.rand(shape, generator=g)
            dt = torch.exp(u * (math.log(0.1) - math.log(0.001)) + math.log(0.001)).clamp(min=1e-4)
            t = dt + torch.log(-torch.expm1(-dt))
        elif leaf.startswith("dt_proj") and leaf.endswith(".weight"):
            t = r * shape[1] ** -0.5
        elif leaf.startswith("conv1d") and leaf.endswith(".weight"):
            t = 0.5 * r
        elif leaf.endswith(".bias"):
            t = 0.02 * r
        elif leaf == "pos_embed":
            t = 0.02 * r
        elif leaf == "out_proj.weight":
            t = r * (shape[1] ** -0.5) * 0.5
        else:  # linear / conv weights: fan-in scaling
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = r * fan_in ** -0.5
        sd[key] = t.contiguous()
    return sd


def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def patch_embed_oracle(sd, x, patch_size, cd):
    """models/fastvim.py:72-103 with dynamic_img_pad=True, rowwise scan path:
    Conv2d(k=s=patch) restated as patch-unfold + matmul."""
    Bsz, C, H, W = x.shape
    ps = patch_size
    pad_h, pad_w = (ps - H % ps) % ps, (ps - W % ps) % ps
    xf = F.pad(x.to(cd), (0, pad_w, 0, pad_h))
    gh, gw = xf.shape[2] // ps, xf.shape[3] // ps
    patches = xf.reshape(Bsz, C, gh, ps, gw, ps).permute(0, 2, 4, 1, 3, 5).reshape(Bsz, gh * gw, C * ps * ps)
    Wp = sd["patch_embed.proj.weight"].to(cd).reshape(-1, C * ps * ps)
    return patches @ Wp.t() + sd["patch_embed.proj.bias"].to(cd), (gh, gw)


def fastvim_block_oracle(sd_layer, hidden, residual, layer_idx, token_size, *, norm_eps=1e-5,
                         rotate_every_block=True, row_scale=None, compute_dtype=torch.float64,
                         mixer_kwargs=None, storage_dtype=None):
    """models/fastvim.py:146-212 with fused_add_norm=True, rms_norm=True,
    residual_in_fp32=True.  ``row_scale`` is the DropPath per-sample scale applied to
    ``hidden`` before the add (:182-190).  Returns (hidden_out, residual_out).
    ``storage_dtype``: the autocast mode of the HIP path -- the normalised rows and the mixer's tensors take a round trip
    through it where the HIP path stores them (see ``fastvim_mixer_oracle``); the residual stream stays in ``compute_dtype``."""
    cd = compute_dtype
    h, res = fused_add_norm_oracle(hidden, sd_layer["norm.weight"], None, residual, norm_eps,
                                   prenorm=True, residual_in_fp32=True, is_rms_norm=True,
                                   row_scale=row_scale if residual is not None else None,
                                   compute_dtype=cd)
    if storage_dtype is not None:
        h = h.to(storage_dtype).to(cd)
        mixer_kwargs = dict(mixer_kwargs or {}, storage_dtype=storage_dtype)
    T0, T1 = token_size
    Bsz, M, d = h.shape
    rot = rotate_every_block and layer_idx % 2 != 0
    ts = token_size
    if rot:                                                               # :192-200
        h = h.reshape(Bsz, T0, T1, d).transpose(1, 2).reshape(Bsz, M, d)
        ts = (T1, T0)                                                     # create_block :244-260
    h = fastvim_mixer_oracle(_sub(sd_layer, "mixer."), h, ts, compute_dtype=cd, **(mixer_kwargs or {}))
    if rot:                                                               # :204-210
        h = h.reshape(Bsz, T1, T0, d).transpose(1, 2).reshape(Bsz, M, d)
    return h, res


def fastvim_forward_oracle(sd, x, *, patch_size=16, depth=24, norm_eps=1e-5, rotate_every_block=True,
                           final_pool_type="mean", row_scales=None, compute_dtype=torch.float64,
                           return_features=False, return_hidden=False, mixer_kwargs=None,
                           scanpath_type="rowwise", storage_dtype=None):
    """models/fastvim.py:484-557 (if_abs_pos_embed=True, fused_add_norm, rms_norm,
    residual_in_fp32).  ``row_scales``: optional list of depth+1 per-sample DropPath scales
    ((B,) tensors or None); entry i is applied inside block i, entry ``depth`` before norm_f.
    ``scanpath_type="colwise"`` (Pool_row): the patch grid is transposed before it is flattened and
    the token grid is (gw, gh) (models/fastvim.py:45-51, 97-98).
    ``storage_dtype`` (e.g. torch.bfloat16; rowwise path): same math in ``compute_dtype`` with a round trip through it at
    every tensor the HIP path stores in it under autocast -- the unfolded patches and the projection weight, the patch
    projection's product (bias and position table are added in fp32 after it), every block's normalised rows and mixer
    tensors, the final norm's rows, the pooled feature, the head's weight and its logits.  For bf16 parity tests at a few
    ulps instead of percent-level bounds."""
    cd = compute_dtype
    rq = (lambda t: t) if storage_dtype is None else (lambda t: t.to(storage_dtype).to(cd))
    if storage_dtype is None:
        h, token_size = patch_embed_oracle(sd, x, patch_size, cd)
    else:
        Bsz, C, H, W = x.shape
        ps = patch_size
        gh, gw = H // ps, W // ps
        patches = rq(x.to(cd)).reshape(Bsz, C, gh, ps, gw, ps).permute(0, 2, 4, 1, 3, 5).reshape(Bsz, gh * gw, C * ps * ps)
        h = rq(patches @ rq(sd["patch_embed.proj.weight"].to(cd).reshape(-1, C * ps * ps)).t()) + sd["patch_embed.proj.bias"].to(cd)
        token_size = (gh, gw)
    if scanpath_type == "colwise":
        gh, gw = token_size
        h = h.reshape(h.shape[0], gh, gw, -1).transpose(1, 2).reshape(h.shape[0], gh * gw, -1)
        token_size = (gw, gh)
    h = h + sd["pos_embed"].to(cd)                                        # :500
    h = h.to(cd if cd == torch.float64 else torch.float32)
    residual = None
    hiddens = []
    for i in range(depth):
        rs = row_scales[i] if row_scales is not None else None
        h, residual = fastvim_block_oracle(_sub(sd, f"layers.{i}."), h, residual, i, token_size,
                                           norm_eps=norm_eps, rotate_every_block=rotate_every_block,
                                           row_scale=rs, compute_dtype=cd, mixer_kwargs=mixer_kwargs,
                                           storage_dtype=storage_dtype)
        hiddens.append(h)
    rs = row_scales[depth] if row_scales is not None else None
    h = fused_add_norm_oracle(h, sd["norm_f.weight"], None, residual, norm_eps, prenorm=False,
                              residual_in_fp32=True, is_rms_norm=True, row_scale=rs, compute_dtype=cd)
    if storage_dtype is not None and final_pool_type == "mean":
        feat = rq(rq(h.to(cd)).mean(1))
        if return_features:
            return (feat, hiddens) if return_hidden else feat
        logits = rq(feat @ rq(sd["head.weight"].to(cd)).t() + sd["head.bias"].to(cd))
        return (logits, hiddens) if return_hidden else logits
    if final_pool_type == "mean":                                         # :541-542
        feat = h.to(cd).mean(1)
    elif final_pool_type == "none":
        feat = h[:, -1, :].to(cd)
    else:
        feat = h.to(cd)
    if return_features:
        return (feat, hiddens) if return_hidden else feat
    logits = feat @ sd["head.weight"].to(cd).t() + sd["head.bias"].to(cd)  # :554
    if final_pool_type == "max":
        logits = logits.max(1)[0]
    return (logits, hiddens) if return_hidden else logits


# ------------------------------------------------------------------------------------------------
# FastChannelVim (channel-wise tokenization, Channel-First scan order)
# ------------------------------------------------------------------------------------------------
def channel_state_dict_shapes(embed_dim=384, depth=24, img_size=224, patch_size=16, channels=8, **kw):
    """Reference ``state_dict`` of models_channel_mamba_faster.VisionMamba (:458-589): the FastVim table
    with a Conv3d(1, d, (1, p, p)) patch projection and a ``channel_embed`` lookup table."""
    shapes = state_dict_shapes(embed_dim=embed_dim, depth=depth, img_size=img_size, patch_size=patch_size,
                               channels=1, **kw)
    shapes["patch_embed.proj.weight"] = (embed_dim, 1, 1, patch_size, patch_size)
    shapes["patch_embed.channel_embed.weight"] = (channels, embed_dim)
    return shapes


def make_channel_state_dict(seed=0, **cfg):
    """``make_state_dict`` for the channel model (same per-key generators and scales)."""
    return make_state_dict(seed, shapes=channel_state_dict_shapes(**cfg))


def channel_patch_embed_oracle(sd, x, patch_size, cd, channels=None):
    """PatchEmbedPerChannel.forward (models_channel_mamba_faster.py:130-203), rowwise scan path,
    Channel-First order: the shared Conv3d(1, d, (1, p, p)) restated as per-channel patch-unfold +
    matmul, plus the channel embedding; tokens come out ordered (row, col, channel).
    ``channels``: the HCS subset (sorted list) or None for all."""
    Bsz, C, H, W = x.shape
    ps = patch_size
    idx = list(range(C)) if channels is None else list(channels)
    xf = x[:, idx].to(cd)
    gh, gw = H // ps, W // ps
    k = len(idx)
    patches = xf.reshape(Bsz, k, gh, ps, gw, ps).permute(0, 2, 4, 1, 3, 5).reshape(Bsz, gh * gw * k, ps * ps)
    Wp = sd["patch_embed.proj.weight"].to(cd).reshape(-1, ps * ps)
    tok = patches @ Wp.t() + sd["patch_embed.proj.bias"].to(cd)                       # :188
    ce = sd["patch_embed.channel_embed.weight"].to(cd)[idx]                            # :159-181
    tok = tok.reshape(Bsz, gh * gw, k, -1) + ce[None, None]                            # :191
    return tok.reshape(Bsz, gh * gw * k, -1), (gh, gw), k


def channel_block_oracle(sd_layer, hidden, residual, layer_idx, token_size, tokens_per_patch, *,
                         norm_eps=1e-5, rotate_every_block=True, row_scale=None,
                         compute_dtype=torch.float64, mixer_kwargs=None, scan_order="Channel-First", compress2d=False):
    """Block.forward of the channel model (models_channel_mamba_faster.py:249-331).
    ``scan_order="Spatial-First"``: tokens ordered (channel, row, col), pooling groups (channel, row)
    (mamba_simple_channel_faster.py:226-241) = the FastVim mixer on a (t*rows, cols) grid; rotated layers transpose
    each channel's grid (:300-305).  ``compress2d``: the 2-D compress variant
    (models_channel_mamba_faster_2dcompress.py:265-300, mamba_simple_channel_faster_2dcompress.py:226-249): layers cycle
    row scan / column scan (cells transposed) / channel scan; the row and column scans pool cols AND channels, the
    channel scan pools every cell."""
    cd = compute_dtype
    h, res = fused_add_norm_oracle(hidden, sd_layer["norm.weight"], None, residual, norm_eps,
                                   prenorm=True, residual_in_fp32=True, is_rms_norm=True,
                                   row_scale=row_scale if residual is not None else None,
                                   compute_dtype=cd)
    T0, T1 = token_size
    Bsz, M, d = h.shape
    t = tokens_per_patch
    rot = rotate_every_block and ((layer_idx + 2) % 3 == 0 if compress2d else layer_idx % 2 != 0)
    ts = token_size
    spatial = scan_order == "Spatial-First"
    if rot:                                                               # :307-311
        if spatial:
            h = h.reshape(Bsz, t, T0, T1, d).transpose(2, 3).reshape(Bsz, M, d)
        else:
            h = h.reshape(Bsz, T0, T1, t, d).transpose(1, 2).reshape(Bsz, M, d)
        ts = (T1, T0)                                                     # create_block :363-374
    if compress2d and (layer_idx + 1) % 3 == 0:       # channel-wise scan: one pooling group per channel slot
        grid, tpp = (1, ts[0] * ts[1]), t
    elif compress2d:                                  # row / column scan over cells AND channels
        grid, tpp = (ts[0], ts[1] * t), 1
    elif spatial:
        grid, tpp = (t * ts[0], ts[1]), 1
    else:
        grid, tpp = ts, t
    h = fastvim_mixer_oracle(_sub(sd_layer, "mixer."), h, grid, tokens_per_patch=tpp, compute_dtype=cd,
                             **(mixer_kwargs or {}))
    if rot:                                                               # :325-329
        if spatial:
            h = h.reshape(Bsz, t, T1, T0, d).transpose(2, 3).reshape(Bsz, M, d)
        else:
            h = h.reshape(Bsz, T1, T0, t, d).transpose(1, 2).reshape(Bsz, M, d)
    return h, res


def channel_forward_oracle(sd, x, *, patch_size=16, depth=24, norm_eps=1e-5, rotate_every_block=True,
                           final_pool_type="mean", channels=None, row_scales=None,
                           compute_dtype=torch.float64, return_features=False, mixer_kwargs=None,
                           scan_order="Channel-First", compress2d=False):
    """VisionMamba.forward_features / forward of the channel model
    (models_channel_mamba_faster.py:614-682), Channel-First, fused_add_norm + rms_norm + fp32 residual."""
    cd = compute_dtype
    h, token_size, t = channel_patch_embed_oracle(sd, x, patch_size, cd, channels)
    if "pos_embed" in sd:                                                  # if_abs_pos_embed (:617-627)
        h = h + torch.repeat_interleave(sd["pos_embed"].to(cd), t, 1)      # :626-627
    if scan_order == "Spatial-First":                                      # (row, col, channel) -> (channel, row, col)
        Bsz = h.shape[0]
        h = h.reshape(Bsz, token_size[0] * token_size[1], t, -1).transpose(1, 2).reshape(Bsz, -1, h.shape[-1])
    h = h.to(cd if cd == torch.float64 else torch.float32)
    residual = None
    for i in range(depth):
        rs = row_scales[i] if row_scales is not None else None
        h, residual = channel_block_oracle(_sub(sd, f"layers.{i}."), h, residual, i, token_size, t,
                                           norm_eps=norm_eps, rotate_every_block=rotate_every_block,
                                           row_scale=rs, compute_dtype=cd, mixer_kwargs=mixer_kwargs,
                                           scan_order=scan_order, compress2d=compress2d)
    rs = row_scales[depth] if row_scales is not None else None
    h = fused_add_norm_oracle(h, sd["norm_f.weight"], None, residual, norm_eps, prenorm=False,
                              residual_in_fp32=True, is_rms_norm=True, row_scale=rs, compute_dtype=cd)
    if final_pool_type == "mean":
        feat = h.to(cd).mean(1)
    elif final_pool_type == "none":
        feat = h[:, -1, :].to(cd)
    else:
        feat = h.to(cd)
    if return_features:
        return feat
    logits = feat @ sd["head.weight"].to(cd).t() + sd["head.bias"].to(cd)
    if final_pool_type == "max":
        logits = logits.max(1)[0]
    return logits


# ------------------------------------------------------------------------------------------------
# Vim baseline (models/vim.py, mamba_simple.py bidirectional v2): un-pooled mixer, middle class token
# ------------------------------------------------------------------------------------------------
def vim_mixer_oracle(p, hidden, **kw):
    """mamba_simple.Mamba.forward with LayerNorm after the SSM (:293-400): the FastVim mixer on an
    L x 1 grid -- pooling over one column and its expansion are identities."""
    return fastvim_mixer_oracle(p, hidden, (hidden.shape[1], 1), **kw)


def vim_forward_oracle(sd, x, *, patch_size=16, depth=24, norm_eps=1e-5, use_middle_cls_token=True,
                       compute_dtype=torch.float64, return_features=False):
    """VisionMamba.forward of models/vim.py (:410-508) with if_cls_token, if_abs_pos_embed, fused_add_norm,
    rms_norm, residual_in_fp32: the class token is inserted at M // 2 (or 0) and read back after norm_f."""
    cd = compute_dtype
    h, _ = patch_embed_oracle(sd, x, patch_size, cd)
    Bsz, M, d = h.shape
    cls = sd["cls_token"].to(cd).expand(Bsz, -1, -1)
    pos = M // 2 if use_middle_cls_token else 0
    h = torch.cat((h[:, :pos], cls, h[:, pos:]), dim=1)
    h = h + sd["pos_embed"].to(cd)
    h = h.to(cd if cd == torch.float64 else torch.float32)
    residual = None
    for i in range(depth):
        sdl = _sub(sd, f"layers.{i}.")
        hn, residual = fused_add_norm_oracle(h, sdl["norm.weight"], None, residual, norm_eps, prenorm=True,
                                             residual_in_fp32=True, is_rms_norm=True, compute_dtype=cd)
        h = vim_mixer_oracle(_sub(sdl, "mixer."), hn, compute_dtype=cd)
    h = fused_add_norm_oracle(h, sd["norm_f.weight"], None, residual, norm_eps, prenorm=False,
                              residual_in_fp32=True, is_rms_norm=True, compute_dtype=cd)
    feat = h[:, pos, :].to(cd)
    if return_features:
        return feat
    return feat @ sd["head.weight"].to(cd).t() + sd["head.bias"].to(cd)


# ------------------------------------------------------------------------------------------------
# FastVim MAE pre-training model (models/mae/models_mamba_faster_mae_vimdecoder.py)
# ------------------------------------------------------------------------------------------------
def mae_random_masking_oracle(L, mask_ratio, noise):
    """random_masking (:738-772) from given uniform noise (N, L): ids_keep (sorted), ids_restore, mask."""
    len_keep = int(L * (1 - mask_ratio))
    ids_shuffle = torch.argsort(noise, dim=1)
    ids_shuffle[:, :len_keep] = ids_shuffle[:, :len_keep].sort().values
    ids_restore = torch.argsort(ids_shuffle, dim=1)
    mask = torch.ones(noise.shape[0], L, dtype=torch.float64)
    mask[:, :len_keep] = 0
    return ids_shuffle[:, :len_keep].contiguous(), ids_restore, torch.gather(mask, 1, ids_restore)


def mae_forward_oracle(sd, imgs, noise, *, mask_ratio=0.75, patch_size=16, depth=24, decoder_depth=2,
                       norm_eps=1e-5, norm_pix_loss=True, rotate_every_block=True, compute_dtype=torch.float64):
    """MaskedAutoencoderViM.forward (:882-893) with fused_add_norm, rms_norm, residual_in_fp32 (the
    mae_FastVim_* factories, :897-950).  ``noise`` (N, L) replaces the torch.rand draw of random_masking.
    Returns (loss, pred (N, L, p*p*3), mask (N, L))."""
    from .mixer import masked_mixer_oracle
    cd = compute_dtype
    h, (gh, gw) = patch_embed_oracle(sd, imgs, patch_size, cd)
    h = h + sd["pos_embed"].to(cd)                                        # :780
    Bsz, L, d = h.shape
    ids_keep, ids_restore, mask = mae_random_masking_oracle(L, mask_ratio, noise)
    h = torch.gather(h, 1, ids_keep[..., None].expand(-1, -1, d))
    bidx = torch.arange(Bsz)[:, None]
    rotate_indices = (torch.arange(L) % gw) * gh + torch.arange(L) // gw  # Block_masked.compute_rotate_indices (:320-323)
    residual = None
    for i in range(depth):                                                # Block_masked.forward (:325-396)
        sdl = _sub(sd, f"layers.{i}.")
        hn, residual = fused_add_norm_oracle(h, sdl["norm.weight"], None, residual, norm_eps, prenorm=True,
                                             residual_in_fp32=True, is_rms_norm=True, compute_dtype=cd)
        rot = rotate_every_block and i % 2 != 0
        ids, ts = ids_keep, (gh, gw)
        if rot:
            ids = rotate_indices[ids_keep]
            order = torch.argsort(ids, dim=1)
            inverse = torch.argsort(order, 1)
            ids = ids[bidx, order]
            hn = hn[bidx, order]
            ts = (gw, gh)
        h = masked_mixer_oracle(_sub(sdl, "mixer."), hn, ids, ts, compute_dtype=cd)
        if rot:
            h = h[bidx, inverse]
    h = fused_add_norm_oracle(h, sd["norm_f.weight"], None, residual, norm_eps, prenorm=False,
                              residual_in_fp32=True, is_rms_norm=True, compute_dtype=cd)
    # decoder (:819-862)
    x = h.to(cd) @ sd["decoder_embed.weight"].to(cd).t() + sd["decoder_embed.bias"].to(cd)
    dd = x.shape[-1]
    x = torch.cat([x, sd["mask_token"].to(cd).expand(Bsz, L - x.shape[1], dd)], dim=1)
    x = torch.gather(x, 1, ids_restore[..., None].expand(-1, -1, dd))
    x = x + sd["decoder_pos_embed"].to(cd)
    residual = None
    for i in range(decoder_depth):
        sdl = _sub(sd, f"decoder_blocks.{i}.")
        xn, residual = fused_add_norm_oracle(x, sdl["norm.weight"], None, residual, norm_eps, prenorm=True,
                                             residual_in_fp32=True, is_rms_norm=True, compute_dtype=cd)
        x = vim_mixer_oracle(_sub(sdl, "mixer."), xn, compute_dtype=cd)
    x = fused_add_norm_oracle(x, sd["decoder_norm.weight"], None, residual, norm_eps, prenorm=False,
                              residual_in_fp32=True, is_rms_norm=True, compute_dtype=cd)
    pred = x.to(cd) @ sd["decoder_pred.weight"].to(cd).t() + sd["decoder_pred.bias"].to(cd)
    # loss (:864-880)
    p = patch_size
    C = imgs.shape[1]
    target = imgs.to(cd).reshape(Bsz, C, gh, p, gw, p)
    target = torch.einsum("nchpwq->nhwpqc", target).reshape(Bsz, gh * gw, p * p * C)
    if norm_pix_loss:
        target = (target - target.mean(-1, keepdim=True)) / (target.var(-1, keepdim=True) + 1.0e-6) ** 0.5
    loss = ((pred - target) ** 2).mean(-1)
    loss = (loss * mask.to(cd)).sum() / mask.to(cd).sum()
    return loss, pred, mask
