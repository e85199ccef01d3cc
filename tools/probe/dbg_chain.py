import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from fastvim_amd.layernorm import layer_norm_fn
from fastvim_amd.mamba_simple_faster import LinearFn, OutProjAddNormFn
dev = "cuda"
for (B, Ltok, d_in) in [(2, 196, 128), (2, 196, 384), (2, 196, 64), (2, 196, 192), (2, 64, 128), (1, 64, 128)]:
    d = 192
    gen = torch.Generator(device="cuda").manual_seed(11)
    rn = lambda *s: torch.randn(*s, device=dev, generator=gen)
    g = rn(B, Ltok, d_in).bfloat16(); res = rn(B, Ltok, d)
    W = torch.sin(torch.arange(d * d_in, device=dev).float()).view(d, d_in) * d_in ** -0.5
    nw = 1 + 0.1 * torch.cos(torch.arange(d, device=dev).float())
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y, ro = OutProjAddNormFn.apply(g, W, res, nw, 1e-5, None, torch.bfloat16)
        h = LinearFn.apply(g, W, torch.bfloat16)
        y2, ro2 = layer_norm_fn(h, nw, None, residual=res, eps=1e-5, prenorm=True, residual_in_fp32=True, is_rms_norm=True, out_dtype=torch.bfloat16)
    dr = (ro - ro2).abs().view(-1, d); dy = (y.float() - y2.float()).abs().view(-1, d)
    bad_r = (dr.max(1).values > 0).nonzero().flatten(); bad_y = (dy.max(1).values > 0).nonzero().flatten()
    print((B, Ltok, d_in), "res_out rows differing:", bad_r.numel(), bad_r[:10].tolist(), " y rows differing:", bad_y.numel(), bad_y[:10].tolist(),
          "cols:", (dy.max(0).values > 0).nonzero().flatten()[:12].tolist())
