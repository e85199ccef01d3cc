"""Selective-scan oracle (test infrastructure; see oracle/__init__.py).

Semantic spec followed: ``selective_scan_ref``
(mamba-1p1p1/mamba_ssm/ops/selective_scan_interface.py:126-206):

    delta = softplus(delta + delta_bias)            (lines 154-157)
    dA_t  = exp(delta_t * A)                        (line 175)
    dBu_t = delta_t * B_t * u_t                     (lines 176-183)
    x_t   = dA_t * x_{t-1} + dBu_t                  (line 188)
    y_t   = <x_t, C_t>                              (lines 189-195)
    out   = (y + D*u) * silu(z)                     (lines 202-204)

Real ``A`` only (FastVim never uses complex A: ``A = -exp(A_log)``,
mamba_simple_faster.py:197).
"""
import torch
import torch.nn.functional as F


def _expand_groups(M, dim):
    """(B,G,N,L) -> (B,dim,N,L) by repeating each group over dim//G channels
    (selective_scan_interface.py:182,185)."""
    Bsz, G, N, L = M.shape
    return M[:, :, None].expand(Bsz, G, dim // G, N, L).reshape(Bsz, dim, N, L)


def selective_scan_oracle(u, delta, A, B, C, D=None, z=None, delta_bias=None,
                          delta_softplus=False, return_last_state=False,
                          compute_dtype=torch.float64, out_dtype=None, reverse=False):
    """u, delta: (B,D,L); A: (D,N); B, C: (D,N) | (B,N,L) | (B,G,N,L); D, delta_bias: (D);
    z: (B,D,L).  Math in ``compute_dtype`` (fp64 by default); autograd-differentiable.
    ``reverse=True`` runs the recurrence from t=L-1 down to 0 (the FastVim
    backward direction without materialising flips, mamba_simple_faster.py:272,438)."""
    out_dtype = u.dtype if out_dtype is None else out_dtype
    cd = compute_dtype
    Bsz, dim, L = u.shape
    N = A.shape[1]
    uf, df, Af = u.to(cd), delta.to(cd), A.to(cd)
    if delta_bias is not None:
        df = df + delta_bias.to(cd)[None, :, None]
    if delta_softplus:
        df = F.softplus(df)  # threshold 20, same as the reference's F.softplus default
    Bf, Cf = B.to(cd), C.to(cd)
    if Bf.dim() == 4:
        Bf = _expand_groups(Bf, dim)
    if Cf.dim() == 4:
        Cf = _expand_groups(Cf, dim)

    def at(M, t):  # -> broadcastable to (B, D, N)
        if M.dim() == 2:
            return M[None]
        if M.dim() == 3:
            return M[:, None, :, t]
        return M[:, :, :, t]

    x = torch.zeros(Bsz, dim, N, dtype=cd, device=u.device)
    ys = [None] * L
    order = range(L - 1, -1, -1) if reverse else range(L)
    for t in order:
        dt = df[:, :, t, None]
        x = torch.exp(dt * Af[None]) * x + dt * at(Bf, t) * uf[:, :, t, None]
        ys[t] = (x * at(Cf, t)).sum(-1)
    y = torch.stack(ys, dim=2) if L > 0 else uf.new_zeros(Bsz, dim, 0)
    if D is not None:
        y = y + uf * D.to(cd)[None, :, None]
    if z is not None:
        y = y * F.silu(z.to(cd))
    y = y.to(out_dtype)
    return (y, x) if return_last_state else y


def compressed_scan_oracle(u_full, u_c, delta, A, B, C, D=None, delta_bias=None,
                           delta_softplus=False, return_last_state=False,
                           compute_dtype=torch.float64, out_dtype=None):
    """FastVim 'compressed scan' semantics: scan at length Lc, repeat each output
    cf = L/Lc times, add D*u_full at full length
    (fastvim_kernel/mamba-1p1p1/faster_mamba_ssm/ops/selective_scan_interface.py:188-252)."""
    out_dtype = u_c.dtype if out_dtype is None else out_dtype
    L, Lc = u_full.shape[2], u_c.shape[2]
    assert L % Lc == 0
    cf = L // Lc
    res = selective_scan_oracle(u_c, delta, A, B, C, None, None, delta_bias, delta_softplus,
                                return_last_state, compute_dtype, compute_dtype)
    y, last = res if return_last_state else (res, None)
    y = y.repeat_interleave(cf, dim=2)
    if D is not None:
        y = y + u_full.to(compute_dtype) * D.to(compute_dtype)[None, :, None]
    y = y.to(out_dtype)
    return (y, last) if return_last_state else y


def selective_scan_ref_port(u, delta, A, B, C, D=None, z=None, delta_bias=None,
                            delta_softplus=False, return_last_state=False):
    """fp32 port of the reference's pure-PyTorch scan, same operation order
    (precomputed (B,D,L,N) discretisation tensors, Python loop over L) so it is a
    fair stand-in for ``selective_scan_ref`` as the timed CPU baseline
    (selective_scan_interface.py:151-206).  Real A; B/C (B,N,L) or (D,N)."""
    in_dtype = u.dtype
    u = u.float()
    delta = delta.float()
    if delta_bias is not None:
        delta = delta + delta_bias.float()[:, None]
    if delta_softplus:
        delta = F.softplus(delta)
    Bsz, dim, L = u.shape
    N = A.shape[1]
    B = B.float()
    C = C.float()
    dA = torch.exp(delta[..., None] * A[None, :, None, :])            # (B,D,L,N)
    if B.dim() == 2:
        dBu = (delta * u)[..., None] * B[None, :, None, :]
    elif B.dim() == 3:
        dBu = (delta * u)[..., None] * B.transpose(1, 2)[:, None]     # (B,1,L,N)
    else:
        dBu = (delta * u)[..., None] * _expand_groups(B, dim).permute(0, 1, 3, 2)
    if C.dim() == 4:
        C = _expand_groups(C, dim)
    x = A.new_zeros(Bsz, dim, N)
    ys = []
    for t in range(L):
        x = dA[:, :, t] * x + dBu[:, :, t]
        if C.dim() == 2:
            ys.append((x * C[None]).sum(-1))
        elif C.dim() == 3:
            ys.append((x * C[:, None, :, t]).sum(-1))
        else:
            ys.append((x * C[:, :, :, t]).sum(-1))
    y = torch.stack(ys, dim=2)
    if D is not None:
        y = y + u * D[:, None]
    if z is not None:
        y = y * F.silu(z.float())
    y = y.to(in_dtype)
    return (y, x) if return_last_state else y
