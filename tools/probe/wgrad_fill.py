"""Weight-gradient groups of nb blocks at FastVim-T, one tile class at a time (in_proj problems; out_proj + two x_proj
problems), own operands: launch + partial sums timed per split-K factor.  How small groups (one backward segment of the
data-parallel step) should be sliced.  usage: python tools/probe/wgrad_fill.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from fastvim_amd.gemm import gemm_tn_grouped
from bench import time_kernel
Mt, d, d_in, dev = 128 * 196, 192, 384, "cuda"
Mx = 128 * 14
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
def make_in(nb, s):
    return [(rn(Mt, 2 * d_in), rn(Mt, d), torch.zeros(2 * d_in * d, device=dev), s) for _ in range(nb)]
def make_out(nb, s, sx):
    group = []
    for i in range(nb):
        group += [(rn(Mt, d), rn(Mt, d_in), torch.zeros(d * d_in, device=dev), s),
                  (rn(Mx, 48)[:, :44], rn(Mx, d_in), torch.zeros(44 * d_in, device=dev), sx),
                  (rn(Mx, 48)[:, :44], rn(Mx, d_in), torch.zeros(44 * d_in, device=dev), sx)]
    return group
for nb in (24, 12, 8, 6, 4, 3, 2):
    for s in (2, 4, 7, 8, 14, 28):
        grp = make_in(nb, s)
        t = time_kernel(lambda: gemm_tn_grouped(grp), iters=8)
        print(f"in  blocks {nb:2d} splits {s:2d}      ({nb * 6 * s:4d} wgs): {t*1e6:7.1f} us  ({t*1e6/nb:5.1f} per block)", flush=True)
        del grp
    for s in (2, 4, 7, 8, 14, 28):
        for sx in (2, 4, 7, 14):
            grp = make_out(nb, s, sx)
            t = time_kernel(lambda: gemm_tn_grouped(grp), iters=8)
            print(f"out blocks {nb:2d} splits {s:2d} x {sx:2d} ({nb * (4 * s + 4 * sx):4d} wgs): {t*1e6:7.1f} us  ({t*1e6/nb:5.1f} per block)", flush=True)
            del grp
