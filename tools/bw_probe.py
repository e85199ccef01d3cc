"""Achievable HBM write / copy rates on this GPU for tensors of the projection-output size (probe)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from bench import time_kernel
for mb in (9.6, 19.3, 38.5, 77, 308):
    n = int(mb * 1e6 / 2)
    x = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    y = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    tf = time_kernel(lambda: x.fill_(1.0))
    tc = time_kernel(lambda: y.copy_(x))
    print(f"{mb:6.1f} MB: fill {tf*1e6:6.1f} us = {mb/tf/1e6:5.2f} TB/s   copy {tc*1e6:6.1f} us = {2*mb/tc/1e6:5.2f} TB/s (r+w)")
# read-only streams (a reduction over the buffer): what a kernel that only reads -- a weight-gradient GEMM -- can get
for mb in (77, 308, 616, 1850):
    n = int(mb * 1e6 / 4)
    x = torch.ones(n, device="cuda", dtype=torch.float32)
    tr = time_kernel(lambda: x.sum(), iters=10)
    print(f"{mb:6.1f} MB: read (sum) {tr*1e6:7.1f} us = {mb/tr/1e6:5.2f} TB/s")
