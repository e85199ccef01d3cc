"""Block shape of the long-row conv + pool kernels (tuning build): times conv_pool_fwd / conv_pool_bwd at the cfg4
(FastVim-B 2048 px, batch 8) and cfg5 (FastChannelVim-S, batch 64) shapes, HBM-cold (rotating operand sets).
usage: FASTVIM_BWD_CHAN_GROUPS=g FASTVIM_BWD_CHAN_RG=r FASTVIM_FWD_CHAN_GROUPS=g python tools/probe/r05_chan_shape.py cfg4|cfg5"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from fastvim_amd import mixer_ops as M
which = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
B, rows, cols, tpp, d_in = {"cfg4": (8, 128, 128, 1, 1536), "cfg5": (64, 14, 14, 8, 768), "T": (128, 14, 14, 1, 384), "B": (128, 14, 14, 1, 1536), "S": (128, 14, 14, 1, 768), "T16": (128, 16, 16, 1, 384)}[which]
dev, dt = "cuda", (torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "fp32" else torch.bfloat16)
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, d=dt: torch.randn(*s, device=dev, generator=g).to(d)
L = rows * cols * tpp
cw, cwb = rn(d_in, 4, d=torch.float32) * 0.5, rn(d_in, 4, d=torch.float32) * 0.5
cb, cbb = rn(d_in, d=torch.float32) * 0.1, rn(d_in, d=torch.float32) * 0.1
D, Db = torch.ones(d_in, device=dev), torch.ones(d_in, device=dev)
nbytes = B * L * 2 * d_in * 2
SETS = max(2, min(24, int(3e9 // (4 * nbytes))))
sets = [dict(xz=rn(B, L, 2 * d_in), d_o=rn(B, L, d_in), dxc=rn(2, B, rows * tpp, d_in, d=torch.float32), dxz=torch.empty(B, L, 2 * d_in, device=dev, dtype=dt),
             dxc2=rn(2, B, rows * tpp, d_in) if which == "T" else None)
        for _ in range(SETS)]
def fwd(s): return M.conv_pool_fwd(s["xz"], cw, cb, cwb, cbb, rows, cols, False, 0, 1.0, tpp, D=D, D_b=Db)
def bwd(s): return M.conv_pool_bwd(s["xz"], s["d_o"], s["dxc"], cw, cb, cwb, cbb, D, Db, s["dxz"], rows, cols, False, 0, 1.0, tpp=tpp, dxc2=s["dxc2"])
real = M.reduce_partials
def time(fn, n=5):
    for s in sets: fn(s)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n * SETS)]
    k = 0
    for _ in range(n):
        for s in sets:
            ev[k][0].record(); fn(s); ev[k][1].record(); k += 1
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2]
tf = time(fwd)
M.reduce_partials = lambda part, n, out=None, **kw: part[0]
tb = time(bwd)
M.reduce_partials = real
env = {k: v for k, v in os.environ.items() if k.startswith("FASTVIM_")}
fb = nbytes + B * L * d_in * 2           # xz read + skip write (xc negligible)
bb = nbytes * 2 + B * L * d_in * 2       # xz + d_o read, dxz x-half write... (dxz x half = nbytes / 2)
print(f"{which} {str(dt)[6:]} {env}: conv_pool_fwd {tf:8.1f} us ({fb / tf / 1e3:6.0f} GB/s)   conv_pool_bwd {tb:8.1f} us   sets {SETS}")
