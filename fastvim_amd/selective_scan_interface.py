"""Operator API of the scan: mirror of
mamba-1p1p1/mamba_ssm/ops/selective_scan_interface.py:12-123 (``SelectiveScanFn``,
``selective_scan_fn``) on top of the HIP kernels (csrc/scan_bdl.hip).

Same signature, argument meaning, dtype rules and error behaviour (RuntimeError);
real ``A`` only -- FastVim never builds a complex ``A``
(mamba_simple_faster.py:197).
"""
import ctypes

import torch

from . import _lib as L


def _scan_fwd(u, delta, A, B, C, D, z, delta_bias, delta_softplus, want_last_state):
    batch, dim, seqlen = u.shape
    dstate = A.shape[1]
    B_var, C_var = B.dim() >= 3, C.dim() >= 3
    n_groups = B.shape[1] if B_var else (C.shape[1] if C_var else 1)
    out = torch.empty_like(u)
    last = torch.empty(batch, dim, dstate, device=u.device, dtype=torch.float32) if want_last_state else None
    with torch.cuda.device(u.device):
        rc = L.lib().fv_selective_scan_fwd(
            L.ptr(u), L.ptr(delta), L.ptr(A), L.ptr(B), L.ptr(C), L.ptr(D), L.ptr(z), L.ptr(delta_bias),
            L.ptr(out), L.ptr(last), L.i32(batch), L.i32(dim), L.i32(seqlen), L.i32(dstate),
            L.i32(n_groups), L.i32(B_var), L.i32(C_var), L.i32(delta_softplus),
            L.i32(L.dtype_code(u.dtype)), L.stream_of(u))
    L.check(rc, "selective_scan_fwd")
    return out, last


def _scan_bwd(u, delta, A, B, C, D, z, delta_bias, dout, delta_softplus):
    """The backward kernel launch: (du, ddelta, dA, dB, dC, dD, dz, dbias); dA, dB, dC, dD, dbias in fp32
    (variable B / C as (batch, n_groups, dstate, seqlen))."""
    dout = dout.contiguous()
    batch, dim, seqlen = u.shape
    dstate = A.shape[1]
    B_var, C_var = B.dim() == 4, C.dim() == 4
    n_groups = B.shape[1] if B_var else (C.shape[1] if C_var else 1)
    dev = u.device
    f32 = dict(device=dev, dtype=torch.float32)
    du, ddelta = torch.empty_like(u), torch.empty_like(delta)
    dz = torch.empty_like(z) if z is not None else None
    dA = torch.empty(dim, dstate, **f32)
    dB = torch.empty(B.shape, **f32)
    dC = torch.empty(C.shape, **f32)
    dD = torch.empty(dim, **f32)
    dbias = torch.empty(dim, **f32)
    lib = L.lib()
    ws_bytes = lib.fv_selective_scan_bwd_workspace(L.i32(batch), L.i32(dim), L.i32(seqlen), L.i32(dstate),
                                                   L.i32(n_groups), L.i32(B_var), L.i32(C_var))
    ws = torch.empty(max(int(ws_bytes), 4), device=dev, dtype=torch.uint8)
    with torch.cuda.device(dev):
        rc = lib.fv_selective_scan_bwd(
            L.ptr(u), L.ptr(delta), L.ptr(A), L.ptr(B), L.ptr(C), L.ptr(D), L.ptr(z), L.ptr(delta_bias),
            L.ptr(dout), L.ptr(du), L.ptr(ddelta), L.ptr(dA), L.ptr(dB), L.ptr(dC), L.ptr(dD), L.ptr(dz),
            L.ptr(dbias), L.ptr(ws), L.i32(batch), L.i32(dim), L.i32(seqlen), L.i32(dstate),
            L.i32(n_groups), L.i32(B_var), L.i32(C_var), L.i32(delta_softplus),
            L.i32(L.dtype_code(u.dtype)), L.stream_of(u))
    L.check(rc, "selective_scan_bwd")
    return du, ddelta, dA, dB, dC, dD, dz, dbias


def _validate(u, delta, A, B, C, D, z, delta_bias):
    """Shape/dtype checks of selective_scan.cpp:233-305."""
    L.require_gpu(u, delta, A, B, C, D, z, delta_bias)
    if A.is_complex():
        raise RuntimeError("selective_scan_fn: complex A is not supported by the MI355X build (FastVim uses real A)")
    if u.dim() != 3 or delta.shape != u.shape:
        raise RuntimeError("selective_scan_fn: u and delta must both be (batch, dim, seqlen)")
    if u.dtype not in (torch.float32, torch.float16, torch.bfloat16) or delta.dtype != u.dtype:
        raise RuntimeError("selective_scan_fn: u/delta must share a dtype in {fp32, fp16, bf16}")
    batch, dim, seqlen = u.shape
    if A.dtype != torch.float32 or A.dim() != 2 or A.shape[0] != dim:
        raise RuntimeError("selective_scan_fn: A must be fp32 (dim, dstate)")
    dstate = A.shape[1]
    if dstate > 256:
        raise RuntimeError("selective_scan only supports state dimension <= 256")
    for name, M in (("B", B), ("C", C)):
        if M.dim() == 2:
            if M.shape != (dim, dstate) or M.dtype != torch.float32:
                raise RuntimeError(f"selective_scan_fn: constant {name} must be fp32 (dim, dstate)")
        elif M.dim() == 4:
            if M.shape[0] != batch or M.shape[2] != dstate or M.shape[3] != seqlen or dim % M.shape[1]:
                raise RuntimeError(f"selective_scan_fn: variable {name} must be (batch, n_groups, dstate, seqlen)")
            if M.dtype != u.dtype:
                raise RuntimeError(f"selective_scan_fn: variable {name} must have u's dtype")
        else:
            raise RuntimeError(f"selective_scan_fn: bad {name} rank")
    if B.dim() == 4 and C.dim() == 4 and B.shape[1] != C.shape[1]:
        raise RuntimeError("selective_scan_fn: B and C must have the same number of groups")
    for name, v in (("D", D), ("delta_bias", delta_bias)):
        if v is not None and (v.dtype != torch.float32 or v.shape != (dim,)):
            raise RuntimeError(f"selective_scan_fn: {name} must be fp32 (dim,)")
    if z is not None and (z.shape != u.shape or z.dtype != u.dtype):
        raise RuntimeError("selective_scan_fn: z must match u")


class SelectiveScanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                return_last_state=False):
        # contiguity fixes of selective_scan_interface.py:27-44 (we need fully contiguous rows)
        u, delta = u.contiguous(), delta.contiguous()
        A = A.contiguous()
        B, C = B.contiguous(), C.contiguous()
        D = D.contiguous() if D is not None else None
        z = z.contiguous() if z is not None else None
        delta_bias = delta_bias.contiguous() if delta_bias is not None else None
        ctx.squeeze_B = ctx.squeeze_C = False
        if B.dim() == 3:
            B = B.unsqueeze(1)
            ctx.squeeze_B = True
        if C.dim() == 3:
            C = C.unsqueeze(1)
            ctx.squeeze_C = True
        _validate(u, delta, A, B, C, D, z, delta_bias)
        out, last = _scan_fwd(u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state)
        ctx.delta_softplus = delta_softplus
        ctx.has_D, ctx.has_z, ctx.has_bias = D is not None, z is not None, delta_bias is not None
        ctx.save_for_backward(u, delta, A, B, C, D, z, delta_bias)
        if return_last_state:
            ctx.mark_non_differentiable(last)
            return out, last
        return out

    @staticmethod
    def backward(ctx, dout, *args):
        u, delta, A, B, C, D, z, delta_bias = ctx.saved_tensors
        du, ddelta, dA, dB, dC, dD, dz, dbias = _scan_bwd(u, delta, A, B, C, D, z, delta_bias, dout, ctx.delta_softplus)
        # variable B/C grads are produced in fp32 and cast back (selective_scan.cpp:461-462,488)
        if B.dim() == 4:
            dB = dB.to(B.dtype)
        if C.dim() == 4:
            dC = dC.to(C.dtype)
        if ctx.squeeze_B:
            dB = dB.squeeze(1)
        if ctx.squeeze_C:
            dC = dC.squeeze(1)
        return (du, ddelta, dA, dB, dC, dD if ctx.has_D else None, dz,
                dbias if ctx.has_bias else None, None, None)


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False):
    """if return_last_state is True, returns (out, last_state); last_state is (batch, dim, dstate)
    fp32 and carries no gradient (selective_scan_interface.py:105-123)."""
    return SelectiveScanFn.apply(u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state)


# ------------------------------------------------------------------------------------------------
# Fused op surface of the reference (optional path of the mixer, use_fast_path=True)
# ------------------------------------------------------------------------------------------------
class _InnerFnNoOutProjWithoutZ(torch.autograd.Function):
    """conv1d + SiLU -> [mean over the grid columns] -> x_proj -> dt_proj -> selective scan -> [repeat over the columns
    + D * conv_out] as ONE autograd node with a hand-written backward, in the reference op layout (batch, dim, seqlen):
    the structure of ``FastVim_MambaInnerFnNoOutProj_withoutZ`` (selective_scan_interface.py:452-776; pooled, D outside
    the scan) and ``MambaInnerFnNoOutProj_withoutZ`` (:779-1016; un-pooled, D inside the scan).  Like the reference at
    its default ``checkpoint_lvl=1`` (:600-603, :818-820) the node keeps x, x_dbl and the output-side operands only and
    re-derives conv_out, the pooled rows and delta in backward (:636-666).  Kernels: csrc/conv_bdl.hip, csrc/scan_bdl.hip,
    the MFMA GEMMs (bf16 form where its alignment rules hold, fp32-MFMA form otherwise) with deterministic split-K
    weight gradients -- no library GEMM, no atomics."""

    @staticmethod
    def _front(x, w32, b32, pre_x_shape, scaling_factor):
        """conv_out (B, d, L), the scanned rows (B, d, Lc) -- pooled over the grid columns, or conv_out itself -- and the
        same rows as the (B Lc, d) GEMM operand."""
        from .causal_conv1d import _conv_fwd
        conv_out = _conv_fwd(x, w32, b32, True)
        if pre_x_shape is not None:
            pooled = conv_out.reshape(pre_x_shape).mean(dim=3)
            if scaling_factor != 1:
                pooled = pooled * scaling_factor
        else:
            pooled = conv_out
        Bsz, dim, Lc = pooled.shape
        pm = pooled.transpose(1, 2).reshape(Bsz * Lc, dim).contiguous()
        return conv_out, pooled.contiguous(), pm

    @staticmethod
    def _delta(x_dbl, Wdt, R, Bsz, Lc):
        from .mamba_simple_faster import linear_fwd
        dlow = x_dbl[:, :R].contiguous()
        dm = linear_fwd(dlow, Wdt)                                        # (B Lc, dim)
        return dlow, dm.view(Bsz, Lc, -1).transpose(1, 2).contiguous()    # delta (B, dim, Lc)

    @staticmethod
    def forward(ctx, x, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C, D, delta_bias,
                B_proj_bias, C_proj_bias, delta_softplus, num_of_col, scaling_factor, pre_x_shape):
        from .mamba_simple_faster import linear_fwd
        L.require_gpu(x)
        cdt = x.dtype
        if x.stride(-1) != 1 or not x.is_contiguous():
            x = x.contiguous()
        w32 = conv1d_weight.detach().reshape(conv1d_weight.shape[0], -1).float().contiguous()
        b32 = conv1d_bias.detach().float().contiguous() if conv1d_bias is not None else None
        Wx = x_proj_weight.detach().to(cdt).contiguous()
        Wdt = delta_proj_weight.detach().to(cdt).contiguous()
        R, N = Wdt.shape[1], A.shape[-1]
        A = A.contiguous()
        D32 = D.detach().float().contiguous() if D is not None else None
        bias32 = delta_bias.detach().float().contiguous() if delta_bias is not None else None
        conv_out, pooled, pm = _InnerFnNoOutProjWithoutZ._front(x, w32, b32, pre_x_shape, scaling_factor)
        Bsz, dim, Lc = pooled.shape
        x_dbl = linear_fwd(pm, Wx)                                        # (B Lc, R + 2N)
        _, delta = _InnerFnNoOutProjWithoutZ._delta(x_dbl, Wdt, R, Bsz, Lc)
        var_B, var_C = B is None, C is None
        Bs, Cs = _InnerFnNoOutProjWithoutZ._bc(x_dbl, B, C, B_proj_bias, C_proj_bias, R, N, Bsz, Lc)
        _validate(pooled, delta, A, Bs, Cs, None if pre_x_shape is not None else D32, None, bias32)
        if pre_x_shape is not None:
            yc, _ = _scan_fwd(pooled, delta, A, Bs, Cs, None, None, bias32, delta_softplus, False)
            out = torch.empty_like(conv_out)
            rc = L.lib().fv_scan_expand_skip_fwd(L.ptr(yc), L.ptr(conv_out), L.ptr(D32), L.ptr(out), L.i32(Bsz), L.i32(dim),
                                                 L.i32(conv_out.shape[2]), L.i32(Lc), L.i32(L.dtype_code(cdt)),
                                                 L.stream_of(x))
            L.check(rc, "scan_expand_skip_fwd")
        else:
            out, _ = _scan_fwd(pooled, delta, A, Bs, Cs, D32, None, bias32, delta_softplus, False)
        ctx.delta_softplus, ctx.num_of_col, ctx.scaling_factor, ctx.pre_x_shape = (delta_softplus, num_of_col,
                                                                                   scaling_factor, pre_x_shape)
        ctx.var_B, ctx.var_C = var_B, var_C
        ctx.has = (conv1d_bias is not None, D is not None, delta_bias is not None, B_proj_bias is not None,
                   C_proj_bias is not None)
        ctx.dts = (conv1d_weight.dtype, None if conv1d_bias is None else conv1d_bias.dtype, x_proj_weight.dtype,
                   delta_proj_weight.dtype, None if D is None else D.dtype, None if delta_bias is None else delta_bias.dtype,
                   None if B_proj_bias is None else B_proj_bias.dtype, None if C_proj_bias is None else C_proj_bias.dtype)
        ctx.conv_w_shape = conv1d_weight.shape
        e = x.new_empty(0)
        ctx.save_for_backward(x, w32, b32 if b32 is not None else e, x_dbl, Wx, Wdt, A, e if var_B else Bs, e if var_C else Cs,
                              D32 if D32 is not None else e, bias32 if bias32 is not None else e,
                              B_proj_bias if B_proj_bias is not None else e, C_proj_bias if C_proj_bias is not None else e)
        return out

    @staticmethod
    def _bc(x_dbl, B, C, B_proj_bias, C_proj_bias, R, N, Bsz, Lc):
        """B, C as the scan kernel takes them: (batch, 1, dstate, Lc) when they come out of x_dbl (:516-548), else as given."""
        if B is None:
            Bv = x_dbl[:, R:R + N]
            if B_proj_bias is not None:
                Bv = Bv + B_proj_bias.to(Bv.dtype)
            Bs = Bv.reshape(Bsz, Lc, N).transpose(1, 2).unsqueeze(1).contiguous()
        else:
            Bs = B.contiguous()
        if C is None:
            Cv = x_dbl[:, -N:]
            if C_proj_bias is not None:
                Cv = Cv + C_proj_bias.to(Cv.dtype)
            Cs = Cv.reshape(Bsz, Lc, N).transpose(1, 2).unsqueeze(1).contiguous()
        else:
            Cs = C.contiguous()
        return Bs, Cs

    @staticmethod
    def backward(ctx, dout):
        from .causal_conv1d import _conv_bwd
        from .mamba_simple_faster import linear_dgrad, linear_wgrad
        x, w32, b32, x_dbl, Wx, Wdt, A, Bsv, Csv, D32, bias32, Bpb, Cpb = ctx.saved_tensors
        has_cb, has_D, has_bias, has_Bpb, has_Cpb = ctx.has
        b32 = b32 if has_cb else None
        D32 = D32 if has_D else None
        bias32 = bias32 if has_bias else None
        cdt = x.dtype
        R, N = Wdt.shape[1], A.shape[-1]
        pooled_op = ctx.pre_x_shape is not None
        # re-derive what forward dropped (checkpoint_lvl 1 of the reference)
        conv_out, pooled, pm = _InnerFnNoOutProjWithoutZ._front(x, w32, b32, ctx.pre_x_shape, ctx.scaling_factor)
        Bsz, dim, Lc = pooled.shape
        dlow, delta = _InnerFnNoOutProjWithoutZ._delta(x_dbl, Wdt, R, Bsz, Lc)
        Bs, Cs = _InnerFnNoOutProjWithoutZ._bc(x_dbl, None if ctx.var_B else Bsv, None if ctx.var_C else Csv,
                                                Bpb if has_Bpb else None, Cpb if has_Cpb else None, R, N, Bsz, Lc)
        dout = dout.to(cdt).contiguous()
        Lfull = conv_out.shape[2]
        if pooled_op:
            cols = Lfull // Lc
            dyc = dout.reshape(Bsz, dim, Lc, cols).float().sum(-1).to(cdt)
            du, ddelta, dA, dB, dC, _, _, dbias = _scan_bwd(pooled, delta, A, Bs, Cs, None, None, bias32, dyc,
                                                            ctx.delta_softplus)
            dD = (dout.float() * conv_out.float()).sum((0, 2)) if has_D else None
        else:
            du, ddelta, dA, dB, dC, dD, _, dbias = _scan_bwd(pooled, delta, A, Bs, Cs, D32, None, bias32, dout,
                                                             ctx.delta_softplus)
        # dt_proj and x_proj adjoints
        ddm = ddelta.transpose(1, 2).reshape(Bsz * Lc, dim).contiguous()
        # B / C handed in: their x_dbl columns feed nothing, so their gradient is zero
        dx_dbl = torch.empty_like(x_dbl) if ctx.var_B and ctx.var_C else torch.zeros_like(x_dbl)
        dx_dbl[:, :R] = linear_dgrad(ddm, Wdt)
        dWdt = linear_wgrad(ddm, dlow)                                   # (dim, R) fp32
        dB_in = dC_in = dBpb = dCpb = None
        if ctx.var_B:
            dBm = dB.squeeze(1).transpose(1, 2).reshape(Bsz * Lc, N)
            dx_dbl[:, R:R + N] = dBm
            dBpb = dBm.sum(0) if has_Bpb else None
        else:
            dB_in = dB.to(Bsv.dtype) if Bsv.dim() == 4 else dB
        if ctx.var_C:
            dCm = dC.squeeze(1).transpose(1, 2).reshape(Bsz * Lc, N)
            dx_dbl[:, -N:] = dCm
            dCpb = dCm.sum(0) if has_Cpb else None
        else:
            dC_in = dC.to(Csv.dtype) if Csv.dim() == 4 else dC
        dWx = linear_wgrad(dx_dbl, pm)                                   # (R + 2N, dim) fp32
        dpooled = linear_dgrad(dx_dbl, Wx).view(Bsz, Lc, dim).transpose(1, 2).float() + du.float()
        if pooled_op:
            dconv = (dpooled * (ctx.scaling_factor / cols)).unsqueeze(-1).expand(Bsz, dim, Lc, cols)
            if has_D:
                dconv = dconv + (dout.float() * D32[None, :, None]).view(Bsz, dim, Lc, cols)
            dconv = dconv.reshape(Bsz, dim, Lfull)
        else:
            dconv = dpooled
        dx, dw, db = _conv_bwd(x, w32, b32, dconv.to(cdt).contiguous(), True)
        wdt_, cbdt, xdt, ddt, Ddt, bdt_, Bpdt, Cpdt = ctx.dts
        return (dx, dw.reshape(ctx.conv_w_shape).to(wdt_), db.to(cbdt) if has_cb else None, dWx.to(xdt), dWdt.to(ddt), dA,
                dB_in, dC_in, dD.to(Ddt) if has_D else None, dbias.to(bdt_) if has_bias else None,
                dBpb.to(Bpdt) if dBpb is not None else None, dCpb.to(Cpdt) if dCpb is not None else None,
                None, None, None, None)


def _inner_apply(x, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C, D, delta_bias, B_proj_bias,
                 C_proj_bias, delta_softplus, num_of_col, scaling_factor, pre_x_shape):
    if A.is_complex():
        raise NotImplementedError("complex A is not supported")
    if torch.is_autocast_enabled():          # custom_fwd of the reference: projections in the autocast dtype
        adt = torch.get_autocast_dtype("cuda")
        x_proj_weight, delta_proj_weight = x_proj_weight.to(adt), delta_proj_weight.to(adt)
        x = x.to(adt)
    with torch.autocast("cuda", enabled=False):
        return _InnerFnNoOutProjWithoutZ.apply(x, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C, D,
                                               delta_bias, B_proj_bias, C_proj_bias, delta_softplus, num_of_col,
                                               scaling_factor, pre_x_shape)


def FastVim_mamba_inner_fn_no_out_proj_withoutZ(
        x, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B=None, C=None, D=None, delta_bias=None,
        B_proj_bias=None, C_proj_bias=None, delta_softplus=True, num_of_col=14, collapse_method="mean",
        scaling_factor=1, pre_x_shape=None):
    """One direction of the FastVim mixer in the reference op layout: x (batch, dim, seqlen) ->
    conv1d+SiLU -> mean over the grid columns -> x_proj / dt_proj -> selective scan over the pooled
    rows -> repeat over columns + D * conv_out; returns (batch, dim, seqlen)
    (selective_scan_interface.py:452-606 forward, 609-776 backward, 1716-1753 wrapper; same argument list).

    One autograd node with a hand-written backward that re-derives conv_out and delta, as the reference's
    (``_InnerFnNoOutProjWithoutZ``).  The training hot path does not go through here: fastvim_amd.mamba_simple_faster
    drives the channel-last kernels for both directions at once."""
    if collapse_method != "mean":
        raise NotImplementedError("FastVim_mamba_inner_fn_no_out_proj_withoutZ: collapse_method='mean' only "
                                  "(the reference leaves other methods undefined here)")
    if pre_x_shape is None:
        raise ValueError("FastVim_mamba_inner_fn_no_out_proj_withoutZ: pre_x_shape (batch, dim, rows, cols) is required")
    return _inner_apply(x, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C, D, delta_bias,
                        B_proj_bias, C_proj_bias, delta_softplus, num_of_col, scaling_factor, tuple(pre_x_shape))


def mamba_inner_fn_no_out_proj_withoutZ(x, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B=None,
                                        C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None,
                                        delta_softplus=True):
    """One direction of the Vim baseline mixer in the reference op layout (the op ``mamba_simple.Mamba`` calls with
    ``use_norm_after_ssm=True``, mamba_simple.py:226-255): x (batch, dim, seqlen) -> conv1d + SiLU -> x_proj / dt_proj ->
    selective scan over ALL seqlen steps with the D skip inside the scan, no gate; returns (batch, dim, seqlen)
    (selective_scan_interface.py:779-1016 ``MambaInnerFnNoOutProj_withoutZ``, wrapper :1684-1713; same argument list).

    One autograd node like its FastVim sibling above; the ``fastvim_amd.mamba_simple.Mamba`` module runs both directions
    on the channel-last kernels instead."""
    return _inner_apply(x, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C, D, delta_bias,
                        B_proj_bias, C_proj_bias, delta_softplus, 1, 1, None)


# ------------------------------------------------------------------------------------------------
# "Compressed scan" of the FastVim kernel fork (experimental in the reference; forward only there)
# ------------------------------------------------------------------------------------------------
class CompressedSelectiveScanFn(torch.autograd.Function):
    """out (B, D, L) = repeat_interleave(scan(u_compressed (B, D, Lc), delta, A, B, C), L / Lc) + D * u
    (fastvim_kernel/mamba-1p1p1/faster_mamba_ssm/ops/selective_scan_interface.py:9-116, native
    `faster_selective_scan_cuda.fwd`).  The reference's backward for this op is broken (SURVEY.md
    section 9); here it is the exact adjoint: the scan backward kernel on the cf-summed output gradient."""

    @staticmethod
    def forward(ctx, u, u_compressed, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                return_last_state=False):
        if z is not None:
            raise ValueError("compressed selective scan: z is not supported (reference: same)")
        L.require_gpu(u)
        u, u_c, delta = u.contiguous(), u_compressed.contiguous(), delta.contiguous()
        A, B, C = A.contiguous(), B.contiguous(), C.contiguous()
        ctx.squeeze_B, ctx.squeeze_C = B.dim() == 3, C.dim() == 3
        if B.dim() == 3:
            B = B.unsqueeze(1)
        if C.dim() == 3:
            C = C.unsqueeze(1)
        D = D.contiguous() if D is not None else None
        delta_bias = delta_bias.contiguous() if delta_bias is not None else None
        _validate(u_c, delta, A, B, C, D, None, delta_bias)
        if u.shape[:2] != u_c.shape[:2] or u.shape[2] % u_c.shape[2] or u.dtype != u_c.dtype:
            raise RuntimeError("compressed selective scan: u (B, D, L) must be a whole multiple of u_compressed (B, D, Lc)")
        yc, last = _scan_fwd(u_c, delta, A, B, C, None, None, delta_bias, delta_softplus, return_last_state)
        out = torch.empty_like(u)
        rc = L.lib().fv_scan_expand_skip_fwd(L.ptr(yc), L.ptr(u), L.ptr(D), L.ptr(out), L.i32(u.shape[0]),
                                             L.i32(u.shape[1]), L.i32(u.shape[2]), L.i32(u_c.shape[2]),
                                             L.i32(L.dtype_code(u.dtype)), L.stream_of(u))
        L.check(rc, "scan_expand_skip_fwd")
        ctx.delta_softplus = delta_softplus
        ctx.has_D, ctx.has_bias = D is not None, delta_bias is not None
        ctx.save_for_backward(u, u_c, delta, A, B, C, D, delta_bias)
        if return_last_state:
            ctx.mark_non_differentiable(last)
            return out, last
        return out

    @staticmethod
    def backward(ctx, dout, *args):
        u, u_c, delta, A, B, C, D, delta_bias = ctx.saved_tensors
        Bsz, dim, Lf = u.shape
        Lc = u_c.shape[2]
        dyc = dout.reshape(Bsz, dim, Lc, Lf // Lc).float().sum(-1).to(u_c.dtype)
        with torch.enable_grad():
            leaves = [t.detach().requires_grad_() for t in (u_c, delta, A, B, C)]
            db = delta_bias.detach().requires_grad_() if ctx.has_bias else None
            y = SelectiveScanFn.apply(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], None, None, db,
                                      ctx.delta_softplus, False)
        grads = torch.autograd.grad(y, leaves + ([db] if db is not None else []), dyc)
        du_c, ddelta, dA, dB, dC = grads[:5]
        dbias = grads[5] if db is not None else None
        du = dD = None
        if ctx.has_D:
            du = (dout.float() * D[None, :, None]).to(u.dtype)
            dD = (dout.float() * u.float()).sum((0, 2))
        if ctx.squeeze_B:
            dB = dB.squeeze(1)
        if ctx.squeeze_C:
            dC = dC.squeeze(1)
        return du, du_c, ddelta, dA, dB, dC, dD, None, dbias, None, None


def compressed_selective_scan_fn(u, u_compressed, delta, A, B, C, D=None, z=None, delta_bias=None,
                                 delta_softplus=False, return_last_state=False):
    """``faster_mamba_ssm.ops.selective_scan_interface.selective_scan_fn`` of the FastVim kernel fork."""
    return CompressedSelectiveScanFn.apply(u, u_compressed, delta, A, B, C, D, z, delta_bias, delta_softplus,
                                           return_last_state)
