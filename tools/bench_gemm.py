import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from fastvim_amd.gemm import gemm_nt, gemm_nn, gemm_tn
from bench import time_kernel
M = 25088
def run(name, fn, flops, bytes_):
    t = time_kernel(fn)
    print(f"{name:40s} {t*1e6:8.1f} us  {flops/t/1e12:7.1f} TFLOP/s  {bytes_/t/1e9:8.1f} GB/s", flush=True)
for (N, K, tag) in [(768, 192, "in_proj fwd"), (192, 384, "out_proj fwd"), (192, 768, "patch/dh"), (384, 192, "dg")]:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
    fl = 2.0 * M * N * K; by = 2.0 * (M * K + N * K + M * N)
    run(f"mfma  NT {tag} M{M} N{N} K{K}", lambda: gemm_nt(a, w), fl, by)
    run(f"torch NT {tag}", lambda: torch.nn.functional.linear(a, w), fl, by)
    b = w.t().contiguous()
    run(f"mfma  NN {tag}", lambda: gemm_nn(a, b), fl, by)
    run(f"torch NN {tag}", lambda: a @ b, fl, by)
for (I, J, tag) in [(768, 192, "dW_in"), (192, 384, "dW_out"), (192, 768, "dW_patch")]:
    x = torch.randn(M, I, device="cuda").bfloat16(); y = torch.randn(M, J, device="cuda").bfloat16()
    fl = 2.0 * M * I * J; by = 2.0 * M * (I + J)
    for s in (8, 16, 28):
        run(f"mfma  TN {tag} splits {s}", lambda: gemm_tn(x, y, splits=s), fl, by)
    run(f"torch TN {tag} (bmm16+reduce)", lambda: __import__('fastvim_amd.mamba_simple_faster', fromlist=['x'])._wgrad(x, y), fl, by)
