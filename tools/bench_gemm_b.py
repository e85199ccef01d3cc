"""MFMA GEMM vs torch (hipBLASLt) at the FastVim-S/B projection shapes."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from fastvim_amd.gemm import gemm_nt, gemm_nn, gemm_tn
from bench import time_kernel
M = 25088
def run(name, fn, flops):
    t = time_kernel(fn, iters=10)
    print(f"{name:44s} {t*1e6:8.1f} us  {flops/t/1e12:7.1f} TFLOP/s", flush=True)
for d in (384, 768):
    d_in = 2 * d
    for (N, K, tag) in [(2 * d_in, d, "in_proj fwd"), (d, d_in, "out_proj fwd")]:
        a = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
        fl = 2.0 * M * N * K
        run(f"d={d} mfma  NT {tag} N{N} K{K}", lambda: gemm_nt(a, w), fl)
        run(f"d={d} torch NT {tag}", lambda: torch.nn.functional.linear(a, w), fl)
    for (K, N, tag) in [(2 * d_in, d, "in_proj dgrad"), (d, d_in, "out_proj dgrad")]:
        g = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(K, N, device="cuda").bfloat16()
        fl = 2.0 * M * N * K
        run(f"d={d} mfma  NN {tag} N{N} K{K}", lambda: gemm_nn(g, w), fl)
        run(f"d={d} torch NN {tag}", lambda: g @ w, fl)
    for (I, J, tag) in [(2 * d_in, d, "in_proj wgrad"), (d, d_in, "out_proj wgrad")]:
        x = torch.randn(M, I, device="cuda").bfloat16(); y = torch.randn(M, J, device="cuda").bfloat16()
        fl = 2.0 * M * I * J
        run(f"d={d} mfma  TN {tag} (auto split)", lambda: gemm_tn(x, y, splits=None), fl)
        run(f"d={d} torch TN {tag} (x.t() @ y, fp32 out)", lambda: torch.mm(x.t(), y, out_dtype=torch.float32), fl)
