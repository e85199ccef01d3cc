"""Launch each fused-mixer kernel a few times at the benchmark shape (for rocprofv3 --pmc runs)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from fastvim_amd import mixer_ops as M
from fastvim_amd.layernorm import layer_norm_fn
from fastvim_amd.gemm import gemm_nn, gemm_nt, gemm_tn
B, rows, cols, d = 128, 14, 14, 192
dtype = torch.bfloat16
dev = "cuda"
d_in, L, R_, N = 2 * d, rows * cols, 12, 16
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, dt=dtype: torch.randn(*s, device=dev, generator=g).to(dt)
xz = rn(B, L, 2 * d_in)
cw, cwb = rn(d_in, 4, dt=torch.float32) * 0.5, rn(d_in, 4, dt=torch.float32) * 0.5
cb, cbb = rn(d_in, dt=torch.float32) * 0.1, rn(d_in, dt=torch.float32) * 0.1
D, Db = torch.ones(d_in, device=dev), torch.ones(d_in, device=dev)
lnw, lnb = torch.ones(d_in, device=dev), torch.zeros(d_in, device=dev)
Wdt = rn(d_in, R_, dt=torch.float32) * R_ ** -0.5
bdt = torch.full((d_in,), -4.0, device=dev)
A_log = torch.log(torch.arange(1, N + 1, device=dev, dtype=torch.float32)).repeat(d_in, 1).contiguous()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(n):
    xc, skip = M.conv_pool_fwd(xz, cw, cb, cwb, cbb, rows, cols, False, 0, 1.0, D=D, D_b=Db)
    x_dbl = rn(2, B * rows, R_ + 2 * N)
    yc = M.scan_fwd(xc, x_dbl, Wdt, bdt, A_log, Wdt, bdt, A_log)
    gout, mean, rstd = M.combine_fwd(xz, skip, yc, lnw, lnb, 1e-5, rows, cols, False)
    dg = rn(B, L, d_in)
    dxz = torch.empty_like(xz)
    d_o, dyc, _ = M.combine_bwd(dg, xz, skip, yc, lnw, lnb, mean, rstd, dxz, rows, cols, False)
    dxc, dxd, _ = M.scan_bwd(xc, x_dbl, Wdt, bdt, A_log, Wdt, bdt, A_log, dyc, keep_chunks=True)
    M.conv_pool_bwd(xz, d_o, dxc, cw, cb, cwb, cbb, D, Db, dxz, rows, cols, False, 0, 1.0)
    # round 5, as the FastVim-T step issues them: the scan backward with the x_proj adjoint folded in, the conv + pool
    # adjoint on the two-addend pooled gradient, the rows of the x_proj weight gradient
    Wx32 = rn(2, R_ + 2 * N, d_in, dt=torch.float32) * d_in ** -0.5
    if M.scan_bwd_xproj_ok(xc, Wdt, False, rows, cols, 1):
        dxc_a, dxc_b, dxd2, _ = M.scan_bwd_xproj(xc, x_dbl, Wdt, bdt, A_log, Wdt, bdt, A_log, dyc, Wx32[0], Wx32[1])
        M.conv_pool_bwd(xz, d_o, dxc_a, cw, cb, cwb, cbb, D, Db, dxz, rows, cols, False, 0, 1.0, dxc2=dxc_b)
        M.chunk_rows_bf16([(dxd2, torch.empty(2, B * rows, (R_ + 2 * N + 7) // 8 * 8, device=dev, dtype=torch.bfloat16))])
    hid, res = rn(B, L, d), torch.randn(B, L, d, device=dev, generator=g)
    nw = torch.ones(d, device=dev, requires_grad=True)
    hid.requires_grad_(); res.requires_grad_()
    y, ro = layer_norm_fn(hid, nw, None, residual=res, eps=1e-5, prenorm=True, residual_in_fp32=True, is_rms_norm=True)
    torch.autograd.backward((y, ro), (torch.randn_like(y), torch.randn_like(ro)))
    # the six projection GEMMs of a block (in_proj / out_proj: forward, data gradient, weight gradient)
    Mtok = B * L
    h2, g2 = rn(Mtok, d), rn(Mtok, d_in)
    W_in, W_out = rn(2 * d_in, d), rn(d, d_in)
    xz2, do2 = rn(Mtok, 2 * d_in), rn(Mtok, d)
    gemm_nt(h2, W_in); gemm_nt(g2, W_out)
    gemm_nn(do2, W_out); gemm_nn(xz2, W_in)
    gemm_tn(xz2, h2, splits=28); gemm_tn(do2, g2, splits=28)
    M.xproj_bwd(dxd, xc, rn(R_ + 2 * N, d_in, dt=torch.float32), rn(R_ + 2 * N, d_in, dt=torch.float32), dxc)
    # the two fused projection kernels of a chained block (out_proj + add + RMSNorm; in_proj data gradient + norm adjoint)
    import ctypes
    from fastvim_amd import _lib as L_
    lib = L_.lib()
    resid, rstd_ = torch.randn(Mtok, d, device=dev, generator=g), torch.rand(Mtok, device=dev, generator=g) + 0.5
    nw_, sc_ = torch.ones(d, device=dev), torch.ones(B, device=dev)
    y_, ro_, rs_ = torch.empty(Mtok, d, device=dev, dtype=dtype), torch.empty(Mtok, d, device=dev), torch.empty(Mtok, device=dev)
    gg_ = torch.randn(Mtok, d, device=dev, generator=g)
    pw_ = torch.empty(lib.fv_gemm_bf16_dgrad_addnorm_blocks(L_.i32(Mtok)), d, device=dev)
    L_.check(lib.fv_gemm_bf16_addnorm(L_.ptr(g2), L_.ptr(W_out), L_.ptr(resid), L_.ptr(nw_), L_.ptr(sc_), L_.i32(L), L_.ptr(y_),
                                      L_.ptr(ro_), L_.ptr(rs_), L_.i32(Mtok), L_.i32(d), L_.i32(d_in), ctypes.c_long(d_in),
                                      ctypes.c_long(d_in), ctypes.c_float(1e-5), L_.stream_of(g2)), "addnorm")
    # round 5: combine as the A-tile producer of that launch
    if M.combine_out_proj_addnorm_ok(xz, rows, cols, 1, d):
        M.combine_out_proj_addnorm(xz, skip, yc, lnw, lnb, 1e-5, rows, cols, False, M.combine_buffers(xz, lnw), W_out, resid, nw_, sc_, L, 1e-5)
    dg_ = torch.empty(Mtok, d_in, device=dev, dtype=dtype)
    L_.check(lib.fv_gemm_bf16_dgrad_addnorm_bwd2(L_.ptr(xz2), L_.ptr(W_in), L_.ptr(gg_), L_.ptr(resid), L_.ptr(rstd_), L_.ptr(nw_),
                                                 L_.ptr(sc_), L_.i32(L), L_.ptr(y_), L_.ptr(ro_), L_.ptr(pw_), L_.i32(Mtok), L_.i32(d),
                                                 L_.i32(2 * d_in), ctypes.c_long(2 * d_in), ctypes.c_long(d), L_.ptr(W_out), L_.ptr(dg_),
                                                 L_.i32(d_in), ctypes.c_long(d_in), L_.stream_of(xz2)), "dgrad_addnorm_bwd")
    # round 6: the conv + pool adjoint as the A-tile producer of that launch
    if M.conv_pool_bwd_dgrad_ok(xz, rows, cols, 1, d, False) and M.scan_bwd_xproj_ok(xc, Wdt, False, rows, cols, 1):
        M.conv_pool_bwd_dgrad(xz, d_o, dxc_a, dxc_b, cw, cb, cwb, cbb, D, Db, dxz, rows, cols, False, 1.0, W_in.t().contiguous(),
                              gg_, resid, rstd_, nw_, sc_, L, W2=W_out)
torch.cuda.synchronize()
