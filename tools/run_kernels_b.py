"""The large-shape kernels of round 2 as rocprofv3 --pmc targets: the phased 256 x 256 GEMM in its forward and data-gradient
forms at the FastVim-B 2048 px shapes and the chunked scan kernels at the FastChannelVim-S shape, n launches each.
usage: python tools/run_kernels_b.py [n]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from fastvim_amd.gemm import gemm_nn, gemm_nt
from fastvim_amd import mixer_ops as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = "cuda"
bf = lambda *s: (torch.rand(*s, device=dev) - 0.5).bfloat16()
a, w = bf(131072, 768), bf(3072, 768)
g, wk = bf(131072, 3072), bf(3072, 768)
for _ in range(n):
    gemm_nt(a, w)
for _ in range(n):
    gemm_nn(g, wk)
B, Lc, d_in, Rk, N = 64, 112, 768, 24, 16
xc, x_dbl = bf(2, B, Lc, d_in), bf(2, B * Lc, Rk + 2 * N)
Wdt = [(torch.rand(d_in, Rk, device=dev) - 0.5) * 0.2 for _ in range(2)]
bdt = [torch.rand(d_in, device=dev) - 3.0 for _ in range(2)]
Al = [torch.log(torch.arange(1, N + 1, device=dev, dtype=torch.float32)).repeat(d_in, 1) for _ in range(2)]
dyc = torch.rand(B, Lc, d_in, device=dev) - 0.5
for _ in range(n):
    yc, ck = M.scan_fwd(xc, x_dbl, Wdt[0], bdt[0], Al[0], Wdt[1], bdt[1], Al[1], want_ckpt=True)
    M.scan_bwd(xc, x_dbl, Wdt[0], bdt[0], Al[0], Wdt[1], bdt[1], Al[1], dyc, ckpt=ck, keep_chunks=True)
torch.cuda.synchronize()
