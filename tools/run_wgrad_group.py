"""One grouped weight-gradient launch as the step issues it: 8 blocks' in_proj / out_proj problems at the FastVim-T shape,
every problem with its own operands (target of rocprofv3 --pmc passes).  usage: python tools/run_wgrad_group.py [n]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from fastvim_amd.gemm import gemm_tn_grouped, grouped_splits
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
Mt, d, d_in, dev = 128 * 196, 192, 384, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
sp = grouped_splits(Mt, M=192, N=384)
group = []
for i in range(8):
    group += [(rn(Mt, 2 * d_in), rn(Mt, d), torch.zeros(2 * d_in * d, device=dev), sp),
              (rn(Mt, d), rn(Mt, d_in), torch.zeros(d * d_in, device=dev), sp)]
for _ in range(n):
    gemm_tn_grouped(group)
torch.cuda.synchronize()
