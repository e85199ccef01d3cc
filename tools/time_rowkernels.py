"""Time the fused-mixer row kernels standalone at a given shape (HIP-graph of back-to-back launches,
HIP events): python tools/time_rowkernels.py [B rows cols d tpp]   -> one line of us per kernel."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from bench import time_kernel
from fastvim_amd import mixer_ops as M

a = [int(v) for v in sys.argv[1:6]] + [128, 14, 14, 192, 1][len(sys.argv) - 1:]
B, rows, cols, d, tpp = a
dtype, dev = torch.bfloat16, "cuda"
d_in, L, R_, N = 2 * d, rows * cols * tpp, (d + 15) // 16, 16
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
xc, skip = M.conv_pool_fwd(xz, cw, cb, cwb, cbb, rows, cols, False, 0, 1.0, tpp, D=D, D_b=Db)
x_dbl = rn(2, B * rows * tpp, R_ + 2 * N)
yc = M.scan_fwd(xc, x_dbl, Wdt, bdt, A_log, Wdt, bdt, A_log)
gout, mean, rstd = M.combine_fwd(xz, skip, yc, lnw, lnb, 1e-5, rows, cols, False, tpp=tpp)
dg = rn(B, L, d_in)
dxz = torch.empty_like(xz)
d_o, dyc, _ = M.combine_bwd(dg, xz, skip, yc, lnw, lnb, mean, rstd, dxz, rows, cols, False, tpp=tpp)
dxc, dxd, _ = M.scan_bwd(xc, x_dbl, Wdt, bdt, A_log, Wdt, bdt, A_log, dyc)
out = {}
out["conv_pool_fwd"] = time_kernel(lambda: M.conv_pool_fwd(xz, cw, cb, cwb, cbb, rows, cols, False, 0, 1.0, tpp, D=D, D_b=Db))
out["scan_fwd"] = time_kernel(lambda: M.scan_fwd(xc, x_dbl, Wdt, bdt, A_log, Wdt, bdt, A_log))
Wx2 = (torch.randn(2, R_ + 2 * N, d_in, device=dev, generator=g) * d_in ** -0.5).to(dtype)
out["xproj_fwd"] = time_kernel(lambda: M.xproj_fwd(xc, Wx2))
if M.xproj_scan_fwd(xc, Wx2, Wdt, bdt, A_log, Wdt, bdt, A_log) is not None:
    out["xproj_scan_fwd"] = time_kernel(lambda: M.xproj_scan_fwd(xc, Wx2, Wdt, bdt, A_log, Wdt, bdt, A_log))
out["combine_fwd"] = time_kernel(lambda: M.combine_fwd(xz, skip, yc, lnw, lnb, 1e-5, rows, cols, False, tpp=tpp))
out["combine_bwd"] = time_kernel(lambda: M.combine_bwd(dg, xz, skip, yc, lnw, lnb, mean, rstd, dxz, rows, cols, False, tpp=tpp))
out["scan_bwd"] = time_kernel(lambda: M.scan_bwd(xc, x_dbl, Wdt, bdt, A_log, Wdt, bdt, A_log, dyc))
out["conv_pool_bwd"] = time_kernel(lambda: M.conv_pool_bwd(xz, d_o, dxc, cw, cb, cwb, cbb, D, Db, dxz, rows, cols, False, 0, 1.0, tpp=tpp))
W_ = R_ + 2 * N
if W_ in M.XPROJ_WIDTHS:
    Wx = rn(W_, d_in, dt=torch.float32)
    dxc2, dxd_chunks, _ = M.scan_bwd(xc, x_dbl, Wdt, bdt, A_log, Wdt, bdt, A_log, dyc, keep_chunks=True)
    gW = torch.zeros(2, W_, d_in, device=dev)
    real_reduce = M.reduce_partials
    M.reduce_partials = lambda part, n, out=None, **kw: out if out is not None else part[0]     # kernel only
    out["xproj_bwd"] = time_kernel(lambda: M.xproj_bwd(dxd_chunks, xc, Wx, Wx, dxc2, grad_out=gW))
    M.reduce_partials = real_reduce
    out["xproj_bwd_nodw"] = time_kernel(lambda: M.xproj_bwd(dxd_chunks, xc, Wx, Wx, dxc2, dw=False))
print(os.environ.get("FASTVIM_DBG", ""), " ".join(f"{k}={v * 1e6:.1f}" for k, v in out.items()), flush=True)
