"""x_proj forward: the hand-written kernel against torch.bmm (hipBLASLt) at the FastVim shapes."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from bench import time_kernel
from fastvim_amd import mixer_ops as M
for name, Mrows, d_in, W in (("T", 1792, 384, 44), ("S", 1792, 768, 56), ("B", 1792, 1536, 80), ("B2048", 1024, 1536, 80), ("C", 7168, 768, 56)):
    xc = torch.randn(2, 1, Mrows, d_in, device="cuda").bfloat16()
    Wx = (torch.randn(2, W, d_in, device="cuda") * d_in ** -0.5).bfloat16()
    t1 = time_kernel(lambda: M.xproj_fwd(xc, Wx))
    t2 = time_kernel(lambda: torch.bmm(xc.view(2, Mrows, d_in), Wx.transpose(1, 2)))
    print(f"{name}: kernel {t1*1e6:.1f} us, torch.bmm {t2*1e6:.1f} us")
