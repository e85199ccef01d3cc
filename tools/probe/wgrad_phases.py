"""Grouped weight-gradient launch (own operands per problem) with phases switched off (tuning build, FASTVIM_GEMM_DBG)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from bench import time_kernel
from fastvim_amd.gemm import gemm_tn_grouped, grouped_splits
from fastvim_amd import mixer_ops as M
Mt, d, d_in, dev = 128 * 196, 192, 384, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
sp = grouped_splits(Mt)
group = []
for i in range(8):
    group += [(rn(Mt, 2 * d_in), rn(Mt, d), torch.zeros(2 * d_in * d, device=dev), sp),
              (rn(Mt, d), rn(Mt, d_in), torch.zeros(d * d_in, device=dev), sp)]
M.reduce_partials = lambda part, n, out=None, **kw: out          # GEMM launch only
t = time_kernel(lambda: gemm_tn_grouped(group), iters=10)
print(f"dbg={os.environ.get('FASTVIM_GEMM_DBG', '0')}: {t * 1e6:.1f} us")
