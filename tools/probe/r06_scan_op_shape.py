"""The reference-layout op selective_scan_fn (forward + backward) at one BASELINE configuration's (B, D, L, N), n times: target
of the rocprofv3 --pmc passes of tools/probe/r06_scan_op_pmc.sh.  usage: python tools/probe/r06_scan_op_shape.py cfg2|cfg3|cfg4|cfg5 [n]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from fastvim_amd.selective_scan_interface import selective_scan_fn
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B, D, L, N = {"cfg2": (128, 384, 14, 16), "cfg3": (128, 1536, 14, 16), "cfg4": (8, 1536, 128, 16), "cfg5": (64, 768, 112, 16)}[cfg]
g = torch.Generator().manual_seed(0)
u, dl = torch.randn(B, D, L, generator=g), 0.5 * torch.rand(B, D, L, generator=g)
A = (-0.5 * torch.rand(D, N, generator=g)).cuda().requires_grad_()
Bm, Cm = torch.randn(B, N, L, generator=g), torch.randn(B, N, L, generator=g)
db = (0.5 * torch.rand(D, generator=g)).cuda().requires_grad_()
q = [t.cuda().bfloat16().requires_grad_() for t in (u, dl, Bm, Cm)]
go = torch.randn(B, D, L).cuda().bfloat16()
for _ in range(n):
    y = selective_scan_fn(q[0], q[1], A, q[2], q[3], None, None, db, True)
    torch.autograd.grad(y, q + [A, db], go)
torch.cuda.synchronize()
e = 2
print(f"{cfg}: algorithmic MB fwd {(e * (3 * B * D * L + 2 * B * N * L) + 4 * (D * N + D)) / 1e6:.2f} "
      f"bwd {(e * (5 * B * D * L + 2 * B * N * L) + 4 * (2 * B * N * L + 2 * D * N + 2 * D)) / 1e6:.2f}")
