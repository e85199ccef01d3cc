"""conv_pool_bwd time vs rows per wave (B chosen so that every wave of the 256 x 4-row-group grid gets 1, 2, 3, 4 rows)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from bench import time_kernel
from fastvim_amd import mixer_ops as M
rows, cols, d = 14, 14, 192
dtype, dev = torch.bfloat16, "cuda"
d_in, Ltok = 2 * d, rows * cols
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, dt=dtype: torch.randn(*s, device=dev, generator=g).to(dt)
cw, cwb = rn(d_in, 4, dt=torch.float32) * 0.5, rn(d_in, 4, dt=torch.float32) * 0.5
cb, cbb = rn(d_in, dt=torch.float32) * 0.1, rn(d_in, dt=torch.float32) * 0.1
D, Db = torch.ones(d_in, device=dev), torch.ones(d_in, device=dev)
out = []
for B in [int(a) for a in sys.argv[1:]] or [73, 128, 146, 219, 292]:
    xz = rn(B, Ltok, 2 * d_in); d_o = rn(B, Ltok, d_in); dxc = rn(2, B, rows, d_in, dt=torch.float32); dxz = torch.empty_like(xz)
    t = time_kernel(lambda: M.conv_pool_bwd(xz, d_o, dxc, cw, cb, cwb, cbb, D, Db, dxz, rows, cols, False, 0, 1.0))
    out.append(f"B={B}:{t * 1e6:.1f}")
print(" ".join(out))
