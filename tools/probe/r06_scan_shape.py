"""The chunked pooled-scan kernels (scan_cl_fwd_chunked / scan_cl_bwd_chunked) at a configuration's mixer shape, as the
training step launches them (the forward leaves its per-chunk checkpoints for the backward): target of the rocprofv3 --pmc
passes of tools/probe/r06_scan_pmc.sh.  usage: python tools/probe/r06_scan_shape.py cfg5|vim|cfg4 [n]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from fastvim_amd import _lib as L_
if os.environ.get("PROBE_LIB"):
    L_.LIB_PATH = os.environ["PROBE_LIB"]
from fastvim_amd import mixer_ops as M
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B, Lc, d_in, Rk = {"cfg5": (64, 112, 768, 24), "vim": (128, 200, 384, 12), "cfg4": (8, 128, 1536, 48)}[cfg]
N, dev = 16, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
bf = lambda *s: (torch.rand(*s, device=dev, generator=g) - 0.5).bfloat16()
xc, x_dbl = bf(2, B, Lc, d_in), bf(2, B * Lc, Rk + 2 * N)
Wdt = [(torch.rand(d_in, Rk, device=dev, generator=g) - 0.5) * 0.2 for _ in range(2)]
bdt = [torch.rand(d_in, device=dev, generator=g) - 3.0 for _ in range(2)]
Al = [torch.log(torch.arange(1, N + 1, device=dev, dtype=torch.float32)).repeat(d_in, 1) for _ in range(2)]
dyc = torch.rand(B, Lc, d_in, device=dev, generator=g) - 0.5
for _ in range(n):
    yc, ck = M.scan_fwd(xc, x_dbl, Wdt[0], bdt[0], Al[0], Wdt[1], bdt[1], Al[1], want_ckpt=True)
    M.scan_bwd(xc, x_dbl, Wdt[0], bdt[0], Al[0], Wdt[1], bdt[1], Al[1], dyc, ckpt=ck, keep_chunks=True)
torch.cuda.synchronize()
# algorithmic bytes of the two launches: inputs, outputs, checkpoints written / read
e = 2
small = B * Lc * d_in
nchunk = (Lc + 15) // 16
fwd = 2 * (small * e + B * Lc * (Rk + 2 * N) * e + small * 4) + 2 * B * nchunk * d_in * N * 4
bwd = 2 * (small * e + B * Lc * (Rk + 2 * N) * (e + 4) + small * 4) + small * 4 + 2 * B * nchunk * d_in * N * 4
print(f"{cfg}: algorithmic MB fwd {fwd / 1e6:.1f} bwd {bwd / 1e6:.1f}")
if "--time" in sys.argv:          # the two launches timed alone (events around 20 back-to-back launches, best of 3)
    def t(fn):
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1000 / 20)
        return best
    yc, ck = M.scan_fwd(xc, x_dbl, Wdt[0], bdt[0], Al[0], Wdt[1], bdt[1], Al[1], want_ckpt=True)
    tf = t(lambda: M.scan_fwd(xc, x_dbl, Wdt[0], bdt[0], Al[0], Wdt[1], bdt[1], Al[1], want_ckpt=True))
    tb = t(lambda: M.scan_bwd(xc, x_dbl, Wdt[0], bdt[0], Al[0], Wdt[1], bdt[1], Al[1], dyc, ckpt=ck, keep_chunks=True))
    print(f"{cfg}: scan_fwd {tf:.1f} us  scan_bwd {tb:.1f} us (incl. host launch path)")
