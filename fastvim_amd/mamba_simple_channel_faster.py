"""FastChannelVim mixer: mirror of ``Mamba`` in
mamba-1p1p1/mamba_ssm/modules/mamba_simple_channel_faster.py:25-481 -- same constructor kwargs
(adds ``scan_order``), same parameter names/shapes/initialisation, same
``forward(hidden_states (B, L, D), tokens_per_patch) -> (B, L, D)``.

Channel-First order (the default and what the FastChannelVim-S/16 entry point uses): the
sequence position of a token is ``(row*cols + col)*tokens_per_patch + channel`` and the pooling
group of a token is ``(row, channel)`` (:242-256), so the pooled scan runs over
``rows*tokens_per_patch`` steps.  The same fused HIP kernels as the FastVim mixer run it, with the
``tokens_per_patch`` argument of the C ABI (include/fastvim_hip.h).

Spatial-First order (:226-241, 259-274, 325-331, 376-382): position ``(channel*rows + row)*cols + col``, pooling
group ``(channel, row)`` -- exactly the FastVim mixer on a ``(tokens_per_patch*rows) x cols`` grid, which is how it
runs here (no ``tokens_per_patch`` in the kernels; the channel ``Block`` transposes rotated layers physically).
"""
from .mamba_simple_faster import FastVimMixerFn, Mamba as _FastVimMamba, _compute_dtype, mixer_apply


class Mamba(_FastVimMamba):
    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False,
                 use_fast_path=False, layer_idx=None, device=None, dtype=None, init_layer_scale=None,
                 scanpath_type="rowwise", token_size=None, use_norm_after_ssm=True,
                 use_our_selective_scan=False, scan_order="Channel-First", collapse_method="mean"):
        if scan_order not in ("Channel-First", "Spatial-First"):
            raise ValueError(scan_order)
        # the reference asserts even grids (:68-73)
        assert token_size[0] % 2 == 0, "num_of_rows needs to be even for this implementation since we do compress and expand"
        assert token_size[1] % 2 == 0, "num_of_col needs to be even for this implementation since we do compress and expand"
        super().__init__(d_model, d_state=d_state, d_conv=d_conv, expand=expand, dt_rank=dt_rank, dt_min=dt_min,
                         dt_max=dt_max, dt_init=dt_init, dt_scale=dt_scale, dt_init_floor=dt_init_floor,
                         conv_bias=conv_bias, bias=bias, use_fast_path=use_fast_path, layer_idx=layer_idx,
                         device=device, dtype=dtype, init_layer_scale=init_layer_scale,
                         scanpath_type=scanpath_type, token_size=token_size,
                         use_norm_after_ssm=use_norm_after_ssm, use_our_selective_scan=use_our_selective_scan,
                         collapse_method=collapse_method, scaling_factor=1)
        self.scan_order = scan_order
        del self.pre_x_shape

    def forward(self, hidden_states, tokens_per_patch, inference_params=None, transposed_grid=False):
        """hidden_states: (B, rows*cols*tokens_per_patch, D) in Channel-First order -> same shape.
        ``transposed_grid`` as in the FastVim mixer: the grid cells (not the channel tokens inside a
        cell) are stored transposed, which is what the channel ``Block`` holds on odd layers
        (models_channel_mamba_faster.py:307-311, 325-329)."""
        if inference_params is not None:
            raise NotImplementedError("FastVim mixers have no inference cache (reference: no step())")
        if self.d_conv != 4 or self.d_state != 16:
            raise RuntimeError("fastvim_amd kernels are built for d_conv=4, d_state=16 (the FastVim configs)")
        cdt = _compute_dtype(hidden_states)
        ln_w = self.layernorm.weight if self.use_norm_after_ssm else None
        ln_b = self.layernorm.bias if self.use_norm_after_ssm else None
        ln_eps = self.layernorm.eps if self.use_norm_after_ssm else 0.0
        rows, cols, tpp = self._geometry(int(tokens_per_patch))
        if self.scan_order == "Spatial-First" and transposed_grid:
            raise RuntimeError("Spatial-First mixers take physically transposed tokens (the channel Block does that)")
        out = mixer_apply(
            FastVimMixerFn, hidden_states, self.in_proj.weight, self.in_proj.bias,
            self.conv1d.weight, self.conv1d.bias, self.conv1d_b.weight, self.conv1d_b.bias,
            self.x_proj.weight, self.x_proj_b.weight,
            self.dt_proj.weight, self.dt_proj.bias, self.dt_proj_b.weight, self.dt_proj_b.bias,
            self.A_log, self.A_b_log, self.D, self.D_b, ln_w, ln_b,
            self.out_proj.weight, self.out_proj.bias,
            rows, cols, bool(transposed_grid), self.collapse_method == "max",
            1.0, float(ln_eps), cdt, self.__dict__.get("_fv"), tpp)
        if self.init_layer_scale is not None:
            out = out * self.gamma
        return out

    def _geometry(self, tokens_per_patch):
        """(rows, cols, tokens per cell) of the pooling grid the kernels see."""
        if self.scan_order == "Spatial-First":
            return tokens_per_patch * self.num_of_rows, self.num_of_col, 1
        return self.num_of_rows, self.num_of_col, tokens_per_patch
