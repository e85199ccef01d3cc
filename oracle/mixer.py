"""FastVim bidirectional pooled Mamba mixer oracle (test infrastructure; see oracle/__init__.py).

Follows ``Mamba.forward`` of the FastVim mixer, non-fused default path
(mamba-1p1p1/mamba_ssm/modules/mamba_simple_faster.py:181-457), and with
``tokens_per_patch > 1`` the Channel-First channel mixer
(mamba_simple_channel_faster.py:181-408).  Written flip-free: the reference's
``x.flip(-1)`` + causal conv + scan + ``out_b.flip(-1)`` is restated as an
anti-causal conv on the original order, row means in original order and a scan
over pooled positions in descending order.
"""
import torch
import torch.nn.functional as F

from .conv import causal_conv1d_oracle
from .scan import selective_scan_oracle


class _RoundGrad(torch.autograd.Function):
    """Identity whose GRADIENT takes a round trip through ``dt``: marks a tensor whose gradient the HIP path stores in
    the compute dtype (the output of a bf16 GEMM, a bf16 tensor written by a backward kernel)."""

    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).to(g.dtype), None


class _XProjStorage(torch.autograd.Function):
    """x_dbl = pooled @ round(Wx)^T as the HIP path differentiates it under autocast: the data gradient multiplies the
    fp32 gradient rows by the fp32 MASTER weight (fv_mixer_xproj_bwd2 reads the parameter, not its bf16 shadow), the
    weight gradient multiplies the bf16-ROUNDED gradient rows by the stored pooled rows (it joins the grouped bf16 GEMM,
    as the reference's autocast backward does)."""

    @staticmethod
    def forward(ctx, pooled2, Wx, dt):
        ctx.save_for_backward(pooled2, Wx)
        ctx.dt = dt
        return pooled2 @ Wx.to(dt).to(Wx.dtype).t()

    @staticmethod
    def backward(ctx, g):
        pooled2, Wx = ctx.saved_tensors
        return g @ Wx, g.to(ctx.dt).to(g.dtype).t() @ pooled2, None


def _direction(x_bdl, p, sfx, rows, cols, tpp, collapse, scaling, reverse, cd, rq=None, rgrad=None):
    """One scan direction.  x_bdl: (B, d_in, L) in compute dtype.
    Returns out (B, d_in, L) = expand(scan(pool(conv(x)))) + D * conv(x); with ``rq`` (a storage-rounding function,
    see fastvim_mixer_oracle) the pair (expand(scan(...)), D * conv(x)) with the pooled tensor and x_dbl rounded where
    the HIP path stores them."""
    Bsz, d_in, L = x_bdl.shape
    w = p[f"conv1d{sfx}.weight"].to(cd).reshape(d_in, -1)
    b = p.get(f"conv1d{sfx}.bias")
    # mamba_simple_faster.py:274-285 (conv + SiLU on x / on flipped x)
    xc = causal_conv1d_oracle(x_bdl, w, b, "silu", anticausal=reverse, compute_dtype=cd, out_dtype=cd)
    # :287-305 pooling across `cols` (channel variant: mamba_simple_channel_faster.py:242-256)
    grid = xc.reshape(Bsz, d_in, rows, cols, tpp)
    if collapse == "mean":
        pooled = grid.mean(3)
        if scaling != 1:
            pooled = pooled * scaling
    elif collapse == "max":
        pooled = grid.max(3).values
    else:
        raise NotImplementedError(collapse)
    pooled = pooled.reshape(Bsz, d_in, rows * tpp)                       # (B, d_in, Lc)
    if rq is not None:
        pooled = rq(pooled)                                              # xc is stored in the compute dtype
    Lc = rows * tpp
    # :321-337 x_proj -> (dt, B, C); dt_proj without bias (bias goes into the scan)
    Wx = p[f"x_proj{sfx}.weight"].to(cd)
    Wdt = p[f"dt_proj{sfx}.weight"].to(cd)
    R = Wdt.shape[1]
    N = (Wx.shape[0] - R) // 2
    if rgrad is not None:                                                # storage-rounded forward AND backward
        x_dbl = _XProjStorage.apply(pooled.permute(0, 2, 1).reshape(Bsz * Lc, d_in), Wx, rgrad)
    else:
        if rq is not None:
            Wx = rq(Wx)                                                  # bf16 shadow weight
        x_dbl = pooled.permute(0, 2, 1).reshape(Bsz * Lc, d_in) @ Wx.t()    # (B*Lc, R+2N)
    if rq is not None:
        x_dbl = rq(x_dbl)                                                # stored in the compute dtype
    dt = (x_dbl[:, :R] @ Wdt.t()).reshape(Bsz, Lc, d_in).permute(0, 2, 1)
    Bm = x_dbl[:, R:R + N].reshape(Bsz, Lc, N).permute(0, 2, 1)
    Cm = x_dbl[:, R + N:].reshape(Bsz, Lc, N).permute(0, 2, 1)
    A = -torch.exp(p[f"A{sfx}_log"].float()).to(cd)                      # :197-198
    # :343-354 selective scan, D=None, z=None, softplus, delta_bias = dt_proj.bias
    y = selective_scan_oracle(pooled, dt, A, Bm, Cm, None, None, p[f"dt_proj{sfx}.bias"].float(),
                              True, False, compute_dtype=cd, out_dtype=cd, reverse=reverse)
    # :356-358 repeat_interleave(cols) + D * conv_out
    y = y.reshape(Bsz, d_in, rows, 1, tpp).expand(Bsz, d_in, rows, cols, tpp).reshape(Bsz, d_in, L)
    if rq is not None:
        return y, p[f"D{sfx}"].float().to(cd)[None, :, None] * xc
    return y + p[f"D{sfx}"].float().to(cd)[None, :, None] * xc


def fastvim_mixer_oracle(p, hidden, token_size, tokens_per_patch=1, collapse_method="mean",
                         scaling_factor=1, use_norm_after_ssm=True, ln_eps=1e-5,
                         compute_dtype=torch.float64, out_dtype=None, storage_dtype=None, round_grads=False):
    """p: dict of tensors keyed like the reference mixer's state_dict
    (``in_proj.weight``, ``conv1d.weight`` (d_in,1,W), ``conv1d.bias``, ``x_proj.weight``,
    ``dt_proj.weight``, ``dt_proj.bias``, ``A_log``, ``D``, the same with ``_b``,
    ``layernorm.weight/bias``, ``out_proj.weight``, optional ``gamma``).
    hidden: (B, L, d_model) with L = rows*cols*tokens_per_patch.  Returns (B, L, d_model).
    ``storage_dtype`` (e.g. torch.bfloat16): emulate the autocast mode of the HIP path -- same math in
    ``compute_dtype``, with a round trip through ``storage_dtype`` at every tensor the HIP path STORES in it: the
    input, the projection weights (bf16 shadows), xz, the pooled conv output xc, x_dbl, the skip term
    D*conv_f + D_b*conv_b (rounded once), the gated output g and the result.  Lets a bf16 parity test use a
    tolerance of a few bf16 ulps instead of percent-level bounds.
    ``round_grads`` (with ``storage_dtype``): the same for the BACKWARD pass -- autograd through this function then rounds
    the gradient wherever the HIP backward stores it in the compute dtype: d g (out_proj's data gradient), d xz (both
    halves, written by combine_bwd / conv_pool_bwd), the gradient of the skip term (d_o; the pooled gradient dyc is
    accumulated in fp32 BEFORE that rounding, csrc/combine_wave.hip), and x_proj as ``_XProjStorage`` describes.  The
    caller rounds d hidden (the in_proj data gradient is a bf16 GEMM output)."""
    cd = compute_dtype
    out_dtype = hidden.dtype if out_dtype is None else out_dtype
    rows, cols = token_size
    tpp = tokens_per_patch
    Bsz, L, d = hidden.shape
    assert L == rows * cols * tpp
    rq = None if storage_dtype is None else (lambda t: t.to(storage_dtype).to(cd))
    rgrad = storage_dtype if (round_grads and storage_dtype is not None) else None
    rg = (lambda t: _RoundGrad.apply(t, storage_dtype)) if rgrad is not None else (lambda t: t)
    W_in = p["in_proj.weight"].to(cd)
    W_out = p["out_proj.weight"].to(cd)
    hid = hidden.to(cd)
    if rq is not None:
        W_in, W_out, hid = rq(W_in), rq(W_out), rq(hid)
    d_in = W_in.shape[0] // 2
    xz = hid @ W_in.t()                                                  # :189-193
    if p.get("in_proj.bias") is not None:
        xz = xz + p["in_proj.bias"].to(cd)
    if rq is not None:
        xz = rg(rq(xz))
    x = xz[..., :d_in].permute(0, 2, 1)                                  # (B, d_in, L)
    z = xz[..., d_in:]                                                   # (B, L, d_in)
    out_f = _direction(x, p, "", rows, cols, tpp, collapse_method, scaling_factor, False, cd, rq, rgrad)
    out_b = _direction(x, p, "_b", rows, cols, tpp, collapse_method, scaling_factor, True, cd, rq, rgrad)
    if rq is not None:                                                   # (y_f + y_b + round(D conv_f + D_b conv_b)) / 2
        o = ((out_f[0] + out_b[0] + rg(rq(out_f[1] + out_b[1]))) / 2).permute(0, 2, 1)
    else:
        o = ((out_f + out_b) / 2).permute(0, 2, 1)                       # :434-444
    if use_norm_after_ssm:
        o = F.layer_norm(o, (d_in,), p["layernorm.weight"].to(cd), p["layernorm.bias"].to(cd), ln_eps)
    g = o * F.silu(z)
    if rq is not None:
        g = rg(rq(g))
    y = g @ W_out.t()
    if p.get("out_proj.bias") is not None:
        y = y + p["out_proj.bias"].to(cd)
    if p.get("gamma") is not None:                                       # :455-456
        y = y * p["gamma"].to(cd)
    return y.to(out_dtype)


def _pooled_direction(pooled, p, sfx, cd):
    """x_proj -> dt_proj -> selective scan (ascending) over the pooled rows.  pooled: (B, d_in, rows)."""
    Bsz, d_in, Lc = pooled.shape
    Wx = p[f"x_proj{sfx}.weight"].to(cd)
    Wdt = p[f"dt_proj{sfx}.weight"].to(cd)
    R = Wdt.shape[1]
    N = (Wx.shape[0] - R) // 2
    x_dbl = pooled.permute(0, 2, 1).reshape(Bsz * Lc, d_in) @ Wx.t()
    dt = (x_dbl[:, :R] @ Wdt.t()).reshape(Bsz, Lc, d_in).permute(0, 2, 1)
    Bm = x_dbl[:, R:R + N].reshape(Bsz, Lc, N).permute(0, 2, 1)
    Cm = x_dbl[:, R + N:].reshape(Bsz, Lc, N).permute(0, 2, 1)
    A = -torch.exp(p[f"A{sfx}_log"].float()).to(cd)
    return selective_scan_oracle(pooled, dt, A, Bm, Cm, None, None, p[f"dt_proj{sfx}.bias"].float(),
                                 True, False, compute_dtype=cd, out_dtype=cd)


def masked_mixer_oracle(p, hidden, ids_keep, token_size, use_norm_after_ssm=True, ln_eps=1e-5,
                        compute_dtype=torch.float64, out_dtype=None):
    """MAE masked FastVim mixer: ``Mamba_masked.forward(hidden_states, ids_keep)``
    (mamba-1p1p1/mamba_ssm/modules/mamba_simple_masked_faster.py:167-325) with the constant-divide row means
    (``compute_row_means_constantdivide``, :376-416).  hidden: (B, Lk, d_model) kept tokens only; ids_keep:
    (B, Lk) their positions in the rows x cols grid.  Written flip-free; position t of the kept sequence has row
    ``r(t) = ids_keep[t] // cols`` and mirrored row ``m(t) = r(Lk-1-t)``:

      * forward branch : causal conv, pooled_f[r] = sum_{r(t)=r} conv_f[t] / cols, ascending scan,
        out_f[t] = scan_f[r(t)] + D * conv_f[t];
      * backward branch: the reference pools the conv of the FLIPPED sequence with the UN-flipped ids (:236-239),
        scans it ascending as well, gathers with the un-flipped ids (:311-314) and flips back, i.e. in original
        order pooled_b[r] = sum_{m(t)=r} conv_b[t] / cols and out_b[t] = scan_b[m(t)] + D_b * conv_b[t]
        with conv_b the anti-causal conv."""
    cd = compute_dtype
    out_dtype = hidden.dtype if out_dtype is None else out_dtype
    rows, cols = token_size
    Bsz, Lk, d = hidden.shape
    W_in = p["in_proj.weight"].to(cd)
    d_in = W_in.shape[0] // 2
    xz = hidden.to(cd) @ W_in.t()
    if p.get("in_proj.bias") is not None:
        xz = xz + p["in_proj.bias"].to(cd)
    x = xz[..., :d_in].permute(0, 2, 1)                                  # (B, d_in, Lk)
    z = xz[..., d_in:]
    r = (ids_keep // cols).long()                                        # (B, Lk)
    m = r.flip(1)
    outs = []
    for sfx, anti, idx in (("", False, r), ("_b", True, m)):
        w = p[f"conv1d{sfx}.weight"].to(cd).reshape(d_in, -1)
        conv = causal_conv1d_oracle(x, w, p.get(f"conv1d{sfx}.bias"), "silu", anticausal=anti,
                                    compute_dtype=cd, out_dtype=cd)      # (B, d_in, Lk)
        pooled = torch.zeros(Bsz, d_in, rows, dtype=cd)
        pooled = pooled.scatter_add(2, idx[:, None, :].expand(Bsz, d_in, Lk), conv) / cols
        y = _pooled_direction(pooled, p, sfx, cd)                        # (B, d_in, rows)
        y = torch.gather(y, 2, idx[:, None, :].expand(Bsz, d_in, Lk))
        outs.append(y + p[f"D{sfx}"].float().to(cd)[None, :, None] * conv)
    o = ((outs[0] + outs[1]) / 2).permute(0, 2, 1)
    if use_norm_after_ssm:
        o = F.layer_norm(o, (d_in,), p["layernorm.weight"].to(cd), p["layernorm.bias"].to(cd), ln_eps)
    g = o * F.silu(z)
    y = g @ p["out_proj.weight"].to(cd).t()
    if p.get("out_proj.bias") is not None:
        y = y + p["out_proj.bias"].to(cd)
    if p.get("gamma") is not None:
        y = y * p["gamma"].to(cd)
    return y.to(out_dtype)
