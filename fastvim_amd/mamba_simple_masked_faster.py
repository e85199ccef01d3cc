"""MAE masked FastVim mixer: drop-in for ``Mamba_masked`` of
mamba-1p1p1/mamba_ssm/modules/mamba_simple_masked_faster.py:18-325.

``forward(hidden_states, ids_keep)``: ``hidden_states`` (B, Lk, d_model) holds only the kept tokens of an MAE
pre-training step, ``ids_keep`` (B, Lk) their positions in the full rows x cols grid.  The kept sequence goes
through the same fused HIP kernels as the dense mixer, in their un-pooled geometry (conv + SiLU per kept token, one
skip tensor, LayerNorm + gate per token); what is specific to the masked path are two small deterministic kernels
between the conv and the scan (``fv_rows_segment_sum`` / ``fv_rows_gather``, csrc/masked.hip):

  * pooled[dir][b][r] = sum over the kept tokens of row r of conv_dir / cols -- the reference's
    ``compute_row_means_constantdivide`` (:376-416; constant divide, empty rows stay zero);
  * the scan output of a row is gathered back to its kept tokens (:281-283, 311-314).

Reference quirk reproduced on purpose (the golden vectors in tests/golden/masked.pt pin it): the backward branch
pools and gathers the conv of the FLIPPED sequence with the UN-flipped ``ids_keep`` and scans it in ascending row
order (:236-239, 296-314), so in original token order its row index is the mirrored one, ``r(Lk-1-t)``.  Here that
is one more index array; the rows of the backward direction are stored in reverse so that the scan kernel's
descending walk of direction 1 visits them in ascending order.
"""
import torch

from . import _lib as L
from . import mixer_ops as M
from .mamba_simple import _split_rows_exact as _split_rows
from .gemm import gemm_any_bnn, gemm_any_btn
from .mamba_simple_faster import (Mamba as _FastVimMamba, _compute_dtype, _shadow, mixer_apply, linear_dgrad, linear_fwd,
                                  linear_wgrad)


class MaskedFastVimMixerFn(torch.autograd.Function):
    """hidden (B, Lk, d), ids_keep (B, Lk) -> out (B, Lk, d)."""

    @staticmethod
    def forward(ctx, hidden, ids_keep, W_in, b_in, cw, cb, cw_b, cb_b, Wx, Wx_b, Wdt, bdt, Wdt_b, bdt_b, A_log, A_b_log,
                D, D_b, ln_w, ln_b, W_out, b_out, rows, cols, ln_eps, cdt):
        L.require_gpu(hidden)
        B, Lk, d = hidden.shape
        if ids_keep.shape != (B, Lk):
            raise RuntimeError(f"Mamba_masked: ids_keep {tuple(ids_keep.shape)} does not match hidden {tuple(hidden.shape)}")
        if Lk < 3:
            raise RuntimeError("Mamba_masked: needs at least 3 kept tokens (conv halo)")
        d_in = W_in.shape[0] // 2
        srows = _split_rows(Lk)                 # un-pooled kernel geometry: srows x 1 x t, contiguous sequence
        t = Lk // srows
        with torch.autocast("cuda", enabled=False):
            r = torch.div(ids_keep, cols, rounding_mode="floor").to(torch.int32)          # row of kept token t
            idx = torch.stack([r, (rows - 1) - r.flip(1)]).contiguous()                   # (2, B, Lk)
            h_c = hidden.to(cdt).contiguous()
            W_in_c, W_out_c = _shadow(W_in, cdt), _shadow(W_out, cdt)
            xz = linear_fwd(h_c.view(B * Lk, d), W_in_c, b_in).view(B, Lk, 2 * d_in)
            cw2, cwb2 = cw.reshape(d_in, -1), cw_b.reshape(d_in, -1)
            xc_tok, skip = M.conv_pool_fwd(xz, cw2, cb, cwb2, cb_b, srows, 1, False, False, 1.0, t, D=D, D_b=D_b)
            xcomp = M.rows_segment_sum(xc_tok, idx, rows, 1.0 / cols)                     # (2, B, rows, d_in)
            Wx2 = torch.stack([Wx, Wx_b])
            x_dbl = M.xproj_fwd(xcomp, Wx2.to(cdt).contiguous())
            yc = M.scan_fwd(xcomp, x_dbl, Wdt, bdt, A_log, Wdt_b, bdt_b, A_b_log)        # (2, B, rows, d_in) fp32
            yct = M.rows_gather(yc, idx)                                                  # (2, B, Lk, d_in) fp32
            g, mean, rstd = M.combine_fwd(xz, skip, yct, ln_w, ln_b, ln_eps, srows, 1, False, tpp=t)
            out = linear_fwd(g.view(B * Lk, d_in), W_out_c, b_out).view(B, Lk, d)
        ctx.save_for_backward(h_c, idx, W_in, cw, cb, cw_b, cb_b, Wx2, Wdt, bdt, Wdt_b, bdt_b, A_log, A_b_log, D, D_b,
                              ln_w, ln_b, W_out, xz, xcomp, x_dbl, g, skip, yct, mean, rstd)
        ctx.geo = (rows, cols, srows, t)
        ctx.has_bias = (b_in is not None, b_out is not None)
        ctx.cdt = cdt
        ctx.in_dtype = hidden.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        (h_c, idx, W_in, cw, cb, cw_b, cb_b, Wx2, Wdt, bdt, Wdt_b, bdt_b, A_log, A_b_log, D, D_b, ln_w, ln_b, W_out,
         xz, xcomp, x_dbl, g, skip, yct, mean, rstd) = ctx.saved_tensors
        rows, cols, srows, t = ctx.geo
        cdt = ctx.cdt
        B, Lk, d = h_c.shape
        d_in = W_in.shape[0] // 2
        with torch.autocast("cuda", enabled=False):
            dout = dout.to(cdt).contiguous()
            do2 = dout.view(B * Lk, d)
            dg = linear_dgrad(do2, _shadow(W_out, cdt))
            dW_out = linear_wgrad(do2, g.view(B * Lk, d_in), W_out)      # None when queued / accumulated into the flat gradient
            db_out = do2.float().sum(0) if ctx.has_bias[1] else None
            cw2, cwb2 = cw.reshape(d_in, -1), cw_b.reshape(d_in, -1)
            dxz = torch.empty_like(xz)
            d_o, dyct, p1 = M.combine_bwd(dg, xz, skip, yct, ln_w, ln_b, mean, rstd, dxz, srows, 1, False, tpp=t)
            dyc = M.rows_segment_sum(dyct.view(B, Lk, d_in), idx, rows)                   # (2, B, rows, d_in) fp32
            dxcomp, dx_dbl, ps = M.scan_bwd(xcomp, x_dbl, Wdt, bdt, A_log, Wdt_b, bdt_b, A_b_log, dyc,
                                            dyc_per_direction=True)
            # x_proj adjoint (selective_scan_interface.py:726-734), both directions
            xc2 = xcomp.view(2, B * rows, d_in)
            dWx2 = gemm_any_btn(dx_dbl, xc2)                       # (2, W, d_in) fp32 = dx_dbl^T xc, fp32-MFMA GEMM
            dxcomp = (dxcomp.view(2, B * rows, d_in) + gemm_any_bnn(dx_dbl, Wx2, out_dtype=torch.float32)).view(2, B, rows, d_in)
            dxc_tok = M.rows_gather(dxcomp, idx, 1.0 / cols)                              # (2, B, Lk, d_in) fp32
            p2 = M.conv_pool_bwd(xz, d_o, dxc_tok, cw2, cb, cwb2, cb_b, D, D_b, dxz, srows, 1, False, False, 1.0, tpp=t)
            dxz2 = dxz.view(B * Lk, 2 * d_in)
            dhidden = linear_dgrad(dxz2, _shadow(W_in, cdt)).view(B, Lk, d).to(ctx.in_dtype)
            dW_in = linear_wgrad(dxz2, h_c.view(B * Lk, d), W_in)
            db_in = dxz2.float().sum(0) if ctx.has_bias[0] else None
        n4 = 4 * d_in
        N_, R_ = A_log.shape[1], Wdt.shape[1]
        g_cw, g_cwb = p2[0:n4].view(cw.shape), p2[n4:2 * n4].view(cw_b.shape)
        g_cb = p2[2 * n4:2 * n4 + d_in] if cb is not None else None
        g_cbb = p2[2 * n4 + d_in:2 * n4 + 2 * d_in] if cb_b is not None else None
        g_D, g_Db = p2[2 * n4 + 2 * d_in:2 * n4 + 3 * d_in], p2[2 * n4 + 3 * d_in:2 * n4 + 4 * d_in]
        g_A = [ps[k, :d_in * N_].view(d_in, N_) for k in range(2)]
        g_Wdt = [ps[k, d_in * N_:d_in * (N_ + R_)].view(d_in, R_) for k in range(2)]
        g_bdt = [ps[k, d_in * (N_ + R_):] for k in range(2)]
        g_lw = p1[0] if ln_w is not None else None
        g_lb = p1[1] if ln_w is not None else None
        return (dhidden, None, dW_in, db_in, g_cw, g_cb, g_cwb, g_cbb, dWx2[0], dWx2[1], g_Wdt[0], g_bdt[0], g_Wdt[1],
                g_bdt[1], g_A[0], g_A[1], g_D, g_Db, g_lw, g_lb, dW_out, db_out, None, None, None, None)


class Mamba_masked(_FastVimMamba):
    """Same parameters and ``state_dict`` keys as the dense FastVim mixer (mamba_simple_masked_faster.py:19-165)."""

    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False,
                 use_fast_path=False, layer_idx=None, device=None, dtype=None, init_layer_scale=None,
                 scanpath_type="rowwise", token_size=None, use_norm_after_ssm=True, collapse_method="mean"):
        if collapse_method != "mean":
            raise NotImplementedError("Mamba_masked: the reference only defines collapse_method='mean' (:234-239)")
        super().__init__(d_model, d_state=d_state, d_conv=d_conv, expand=expand, dt_rank=dt_rank, dt_min=dt_min,
                         dt_max=dt_max, dt_init=dt_init, dt_scale=dt_scale, dt_init_floor=dt_init_floor,
                         conv_bias=conv_bias, bias=bias, use_fast_path=use_fast_path, layer_idx=layer_idx,
                         device=device, dtype=dtype, init_layer_scale=init_layer_scale, scanpath_type=scanpath_type,
                         token_size=token_size, use_norm_after_ssm=use_norm_after_ssm, collapse_method="mean",
                         scaling_factor=1)
        del self.scaling_factor, self.use_our_selective_scan

    def forward(self, hidden_states, ids_keep, inference_params=None):
        """hidden_states: (B, Lk, D) kept tokens; ids_keep: (B, Lk) integer positions in the rows x cols grid.
        Returns (B, Lk, D)."""
        if inference_params is not None:
            raise NotImplementedError("FastVim mixers have no inference cache (reference: no step())")
        if self.d_conv != 4 or self.d_state != 16:
            raise RuntimeError("fastvim_amd kernels are built for d_conv=4, d_state=16 (the FastVim configs)")
        cdt = _compute_dtype(hidden_states)
        ln_w = self.layernorm.weight if self.use_norm_after_ssm else None
        ln_b = self.layernorm.bias if self.use_norm_after_ssm else None
        ln_eps = self.layernorm.eps if self.use_norm_after_ssm else 0.0
        out = mixer_apply(
            MaskedFastVimMixerFn, hidden_states, ids_keep, self.in_proj.weight, self.in_proj.bias,
            self.conv1d.weight, self.conv1d.bias, self.conv1d_b.weight, self.conv1d_b.bias,
            self.x_proj.weight, self.x_proj_b.weight,
            self.dt_proj.weight, self.dt_proj.bias, self.dt_proj_b.weight, self.dt_proj_b.bias,
            self.A_log, self.A_b_log, self.D, self.D_b, ln_w, ln_b,
            self.out_proj.weight, self.out_proj.bias,
            self.num_of_rows, self.num_of_col, float(ln_eps), cdt)
        if self.init_layer_scale is not None:
            out = out * self.gamma
        return out
