"""Launch the pooled-scan backward a few times at a mixer shape (target of rocprofv3 --pmc passes).
usage: python tools/run_scan_bwd.py [n] [B rows d_model]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from fastvim_amd import mixer_ops as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B, rows, d = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (128, 14, 192)
dev, dtype = "cuda", torch.bfloat16
d_in, R_, N = 2 * d, max(1, (d + 15) // 16), 16
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, dt=dtype: torch.randn(*s, device=dev, generator=g).to(dt)
xc = rn(2, B, rows, d_in)
x_dbl = rn(2, B * rows, R_ + 2 * N)
Wdt = rn(d_in, R_, dt=torch.float32) * R_ ** -0.5
bdt = torch.full((d_in,), -4.0, device=dev)
A_log = torch.log(torch.arange(1, N + 1, device=dev, dtype=torch.float32)).repeat(d_in, 1).contiguous()
dyc = rn(B, rows, d_in, dt=torch.float32)
for _ in range(n):
    M.scan_bwd(xc, x_dbl, Wdt, bdt, A_log, Wdt, bdt, A_log, dyc, keep_chunks=True)
torch.cuda.synchronize()
