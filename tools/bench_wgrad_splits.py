"""Split-K sweep of the weight-gradient GEMM at one projection shape: python tools/bench_wgrad_splits.py d [M]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from fastvim_amd.gemm import gemm_tn, auto_splits
from bench import time_kernel
d = int(sys.argv[1]) if len(sys.argv) > 1 else 768
M = int(sys.argv[2]) if len(sys.argv) > 2 else 25088
d_in = 2 * d
for (I, J, tag) in [(2 * d_in, d, "in_proj wgrad"), (d, d_in, "out_proj wgrad")]:
    x = torch.randn(M, I, device="cuda").bfloat16(); y = torch.randn(M, J, device="cuda").bfloat16()
    fl = 2.0 * M * I * J
    print(tag, "auto =", auto_splits(M, I, J))
    for s in (1, 2, 4, 7, 8, 14, 16, 28, 49, 56, 98):
        if (M // 64) % s:
            continue
        t = time_kernel(lambda: gemm_tn(x, y, splits=s), iters=10)
        print(f"  splits {s:3d} {t*1e6:8.1f} us {fl/t/1e12:7.1f} TFLOP/s", flush=True)
