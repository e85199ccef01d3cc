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
                L.i32(n_groups), L.i32(B_var), L.i32(C_var), L.i32(ctx.delta_softplus),
                L.i32(L.dtype_code(u.dtype)), L.stream_of(u))
        L.check(rc, "selective_scan_bwd")
        # variable B/C grads are produced in fp32 and cast back (selective_scan.cpp:461-462,488)
        if B_var:
            dB = dB.to(B.dtype)
        if C_var:
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
def FastVim_mamba_inner_fn_no_out_proj_withoutZ(
        x, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B=None, C=None, D=None, delta_bias=None,
        B_proj_bias=None, C_proj_bias=None, delta_softplus=True, num_of_col=14, collapse_method="mean",
        scaling_factor=1, pre_x_shape=None):
    """One direction of the FastVim mixer in the reference op layout: x (batch, dim, seqlen) ->
    conv1d+SiLU -> mean over the grid columns -> x_proj / dt_proj -> selective scan over the pooled
    rows -> repeat over columns + D * conv_out; returns (batch, dim, seqlen)
    (selective_scan_interface.py:452-606 forward, 1716-1753 wrapper; same argument list).

    The training hot path does not go through here (fastvim_amd.mamba_simple_faster drives the fused
    channel-last kernels for both directions at once); this keeps the op importable for code written
    against the reference, built from the HIP ops above -- causal_conv1d_fn and selective_scan_fn --
    with autograd composing their hand-written backwards."""
    import torch.nn.functional as F
    from .causal_conv1d import causal_conv1d_fn
    if collapse_method != "mean":
        raise NotImplementedError("FastVim_mamba_inner_fn_no_out_proj_withoutZ: collapse_method='mean' only "
                                  "(the reference leaves other methods undefined here)")
    if A.is_complex():
        raise NotImplementedError("complex A is not supported")
    delta_rank = delta_proj_weight.shape[1]
    d_state = A.shape[-1]
    if torch.is_autocast_enabled():
        adt = torch.get_autocast_dtype("cuda")
        x_proj_weight, delta_proj_weight = x_proj_weight.to(adt), delta_proj_weight.to(adt)
        x = x.to(adt)
    with torch.autocast("cuda", enabled=False):
        conv_out = causal_conv1d_fn(x, conv1d_weight.reshape(conv1d_weight.shape[0], -1), conv1d_bias,
                                    activation="silu")
        pooled = conv_out.reshape(pre_x_shape).mean(dim=3)                          # (B, d, Lc)
        if scaling_factor != 1:
            pooled = pooled * scaling_factor
        Bsz, dim, Lc = pooled.shape
        # x_proj / dt_proj through the build's own MFMA GEMMs (LinearFn: bf16 kernel where its alignment rules hold, the
        # fp32-MFMA kernel otherwise) with their deterministic split-K weight gradients -- no library GEMM
        from .mamba_simple_faster import LinearFn
        x_dbl = LinearFn.apply(pooled.transpose(1, 2).reshape(Bsz * Lc, dim), x_proj_weight.to(pooled.dtype), pooled.dtype)
        delta = LinearFn.apply(x_dbl[:, :delta_rank], delta_proj_weight.to(pooled.dtype), pooled.dtype)      # (B Lc, dim)
        delta = delta.t().reshape(dim, Bsz, Lc).transpose(0, 1)
        if B is None:
            B = x_dbl[:, delta_rank:delta_rank + d_state]
            if B_proj_bias is not None:
                B = B + B_proj_bias.to(B.dtype)
            B = B.view(Bsz, Lc, d_state).transpose(1, 2).unsqueeze(1)
        if C is None:
            C = x_dbl[:, -d_state:]
            if C_proj_bias is not None:
                C = C + C_proj_bias.to(C.dtype)
            C = C.view(Bsz, Lc, d_state).transpose(1, 2).unsqueeze(1)
        out = selective_scan_fn(pooled, delta, A, B, C, None, None, delta_bias, delta_softplus)
        out = out.repeat_interleave(num_of_col, dim=2)
        if D is not None:
            out = out + D.to(out.dtype).unsqueeze(-1) * conv_out
    return out


def mamba_inner_fn_no_out_proj_withoutZ(x, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B=None,
                                        C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None,
                                        delta_softplus=True):
    """One direction of the Vim baseline mixer in the reference op layout (the op ``mamba_simple.Mamba`` calls with
    ``use_norm_after_ssm=True``, mamba_simple.py:226-255): x (batch, dim, seqlen) -> conv1d + SiLU -> x_proj / dt_proj ->
    selective scan over ALL seqlen steps with the D skip inside the scan, no gate; returns (batch, dim, seqlen)
    (selective_scan_interface.py:779-1016 ``MambaInnerFnNoOutProj_withoutZ``, wrapper :1684-1713; same argument list).

    Like its FastVim sibling above it is the compatibility surface, composed of the HIP ops ``causal_conv1d_fn`` and
    ``selective_scan_fn`` (autograd chains their hand-written backwards); the ``fastvim_amd.mamba_simple.Mamba`` module
    runs both directions on the fused channel-last kernels instead."""
    import torch.nn.functional as F
    from .causal_conv1d import causal_conv1d_fn
    if A.is_complex():
        raise NotImplementedError("complex A is not supported")
    delta_rank = delta_proj_weight.shape[1]
    d_state = A.shape[-1]
    if torch.is_autocast_enabled():
        adt = torch.get_autocast_dtype("cuda")
        x_proj_weight, delta_proj_weight = x_proj_weight.to(adt), delta_proj_weight.to(adt)
        x = x.to(adt)
    with torch.autocast("cuda", enabled=False):
        conv_out = causal_conv1d_fn(x, conv1d_weight.reshape(conv1d_weight.shape[0], -1), conv1d_bias, activation="silu")
        Bsz, dim, L = conv_out.shape
        from .mamba_simple_faster import LinearFn
        x_dbl = LinearFn.apply(conv_out.transpose(1, 2).reshape(Bsz * L, dim), x_proj_weight.to(conv_out.dtype),
                               conv_out.dtype)                                                                   # (b l) d
        delta = LinearFn.apply(x_dbl[:, :delta_rank], delta_proj_weight.to(conv_out.dtype), conv_out.dtype)
        delta = delta.t().reshape(dim, Bsz, L).transpose(0, 1)
        if B is None:                      # variable B (:826-838)
            B = x_dbl[:, delta_rank:delta_rank + d_state]
            if B_proj_bias is not None:
                B = B + B_proj_bias.to(B.dtype)
            B = B.view(Bsz, L, d_state).transpose(1, 2).unsqueeze(1)
        if C is None:                      # variable C (:842-854)
            C = x_dbl[:, -d_state:]
            if C_proj_bias is not None:
                C = C + C_proj_bias.to(C.dtype)
            C = C.view(Bsz, L, d_state).transpose(1, 2).unsqueeze(1)
        return selective_scan_fn(conv_out, delta, A, B, C, D, None, delta_bias, delta_softplus)


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
