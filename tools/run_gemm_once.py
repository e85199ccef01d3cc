import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from fastvim_amd.gemm import gemm_nt, gemm_nn, gemm_tn
M = 25088
a = torch.randn(M, 192, device="cuda").bfloat16(); w = torch.randn(768, 192, device="cuda").bfloat16()
for _ in range(3): gemm_nt(a, w)
a2 = torch.randn(M, 384, device="cuda").bfloat16(); w2 = torch.randn(192, 384, device="cuda").bfloat16()
for _ in range(3): gemm_nt(a2, w2)
x = torch.randn(M, 768, device="cuda").bfloat16(); y = torch.randn(M, 192, device="cuda").bfloat16()
for _ in range(3): gemm_tn(x, y, splits=28)
torch.cuda.synchronize()
