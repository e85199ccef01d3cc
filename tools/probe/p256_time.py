"""Timing of the forward GEMM at a few FastVim-B / S shapes under the current FASTVIM_GEMM_* environment."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from fastvim_amd.gemm import gemm_nt
from bench import time_kernel
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K) in [(131072, 3072, 768), (131072, 768, 1536), (25088, 3072, 768), (25088, 768, 1536), (100352, 1536, 384), (100352, 384, 768)]:
    if N % 256: continue
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = torch.randn(N, K, device="cuda", generator=g).bfloat16()
    t = time_kernel(lambda: gemm_nt(a, w), iters=10)
    print(f"  {M}x{N}x{K}: {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:7.1f} TFLOP/s", flush=True)
