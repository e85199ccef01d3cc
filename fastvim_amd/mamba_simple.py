"""Vim (un-pooled, bidirectional "v2") mixer: mirror of ``Mamba`` in
mamba-1p1p1/mamba_ssm/modules/mamba_simple.py:42-408 -- same constructor kwargs, parameter
names/shapes/initialisation, ``forward(hidden_states (B, L, D)) -> (B, L, D)``.

The Vim mixer is the FastVim mixer with nothing pooled: conv + SiLU both ways, x_proj / dt_proj, a scan over
all L tokens per direction, + D x, average, LayerNorm, * SiLU(z), out_proj (mamba_simple.py:226-268 fused,
:293-400 reference path).  It therefore runs on the same fused HIP kernels as a ``rows x 1 x t`` token grid
with one patch column (the channel-path geometry, include/fastvim_hip.h): every token is its own pooling
group, so pooling and expansion are identities, and the scan length is L.  The baseline the paper compares
FastVim against (README.md:15) is thus measured on the same code path.
"""
import torch.nn.functional as F

from .mamba_simple_faster import FastVimMixerFn, Mamba as _FastVimMamba, _compute_dtype, mixer_apply


def _split_rows_exact(L):
    """rows * t == L with t >= 3 tokens per row (conv halo) and as many rows as possible (parallelism): the MAE masked
    mixer's kept-token sequence (49 = 7 x 7 at mask ratio 0.75), which is not padded."""
    best = 1
    for r in range(1, L + 1):
        if L % r == 0 and L // r >= 3:
            best = r
    return best


def _split_rows(L):
    """(rows, tokens per row, padded length).  The row walkers parallelise over rows, and their fastest un-pooled form
    is the 8-token cell: the sequence is cut into rows of 8 tokens and padded at its END to a whole number of rows
    (197 tokens of Vim-T at 224 px -> 25 rows, 3 pad tokens = +1.5 %) instead of being walked as ONE 197-token row per
    image (197 is prime: conv kernels 140 / 293 us per layer instead of ~14 / ~32).  Short sequences keep one row."""
    if L < 16:
        return 1, L, L
    rows = -(-L // 8)
    return rows, 8, rows * 8


class Mamba(_FastVimMamba):
    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False,
                 use_fast_path=True, layer_idx=None, device=None, dtype=None, init_layer_scale=None,
                 use_norm_after_ssm=True):
        super().__init__(d_model, d_state=d_state, d_conv=d_conv, expand=expand, dt_rank=dt_rank, dt_min=dt_min,
                         dt_max=dt_max, dt_init=dt_init, dt_scale=dt_scale, dt_init_floor=dt_init_floor,
                         conv_bias=conv_bias, bias=bias, use_fast_path=use_fast_path, layer_idx=layer_idx,
                         device=device, dtype=dtype, init_layer_scale=init_layer_scale, token_size=[1, 1],
                         use_norm_after_ssm=use_norm_after_ssm, collapse_method="mean", scaling_factor=1)
        del self.pre_x_shape, self.num_of_rows, self.num_of_col, self.scanpath_type, self.collapse_method
        del self.scaling_factor, self.use_our_selective_scan

    def forward(self, hidden_states, inference_params=None):
        """hidden_states: (B, L, D) -> (B, L, D).  With ``use_norm_after_ssm=False`` the gate follows the fused
        path of the reference (mamba_inner_fn_no_out_proj applies SiLU(z) inside, :270-292)."""
        if inference_params is not None:
            raise NotImplementedError("no inference cache / step() in this build (training hot path only)")
        if self.d_conv != 4 or self.d_state != 16:
            raise RuntimeError("fastvim_amd kernels are built for d_conv=4, d_state=16")
        L = hidden_states.shape[1]
        if L < 3:
            raise RuntimeError("Vim mixer: sequence length must be >= 3")
        rows, t, Lp = _split_rows(L)
        if Lp > L:          # zero tokens at the end: in_proj has no bias here, so their conv input is the conv's own zero padding
            if self.in_proj.bias is not None:
                rows, t, Lp = 1, L, L
            else:
                hidden_states = F.pad(hidden_states, (0, 0, 0, Lp - L))
        cdt = _compute_dtype(hidden_states)
        ln_w = self.layernorm.weight if self.use_norm_after_ssm else None
        ln_b = self.layernorm.bias if self.use_norm_after_ssm else None
        ln_eps = self.layernorm.eps if self.use_norm_after_ssm else 0.0
        out = mixer_apply(
            FastVimMixerFn, hidden_states, self.in_proj.weight, self.in_proj.bias,
            self.conv1d.weight, self.conv1d.bias, self.conv1d_b.weight, self.conv1d_b.bias,
            self.x_proj.weight, self.x_proj_b.weight,
            self.dt_proj.weight, self.dt_proj.bias, self.dt_proj_b.weight, self.dt_proj_b.bias,
            self.A_log, self.A_b_log, self.D, self.D_b, ln_w, ln_b,
            self.out_proj.weight, self.out_proj.bias,
            rows, 1, False, False, 1.0, float(ln_eps), cdt, self.__dict__.get("_fv"), t, L)
        if Lp > L:
            out = out[:, :L]
        if self.init_layer_scale is not None:
            out = out * self.gamma
        return out
