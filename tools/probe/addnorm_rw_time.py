"""fv_gemm_bf16_addnorm_rw (persistent workgroups, weight in registers) against fv_gemm_bf16_addnorm (64-row tiles) at the
FastVim-T shape, HBM-cold (operand sets rotated past the Infinity Cache) and cache-warm.  usage: python tools/probe/addnorm_rw_time.py"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from bench import time_kernel, rotating
from fastvim_amd import _lib as L_
lib = L_.lib()
M, d, K = 25088, 192, 384
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
W = (rn(d, K) * K ** -0.5).bfloat16()
nw, sc = torch.ones(d, device="cuda"), torch.ones(128, device="cuda")
base = dict(a=rn(M, K).bfloat16(), resid=rn(M, d), y=torch.empty(M, d, device="cuda", dtype=torch.bfloat16),
            ro=torch.empty(M, d, device="cuda"), rs=torch.empty(M, device="cuda"))
def tiled(s):
    L_.check(lib.fv_gemm_bf16_addnorm(L_.ptr(s["a"]), L_.ptr(W), L_.ptr(s["resid"]), L_.ptr(nw), L_.ptr(sc), L_.i32(196), L_.ptr(s["y"]),
                                      L_.ptr(s["ro"]), L_.ptr(s["rs"]), L_.i32(M), L_.i32(d), L_.i32(K), ctypes.c_long(K), ctypes.c_long(K),
                                      ctypes.c_float(1e-5), L_.stream_of(s["a"])), "addnorm")
def rw(s):
    L_.check(lib.fv_gemm_bf16_addnorm_rw(L_.ptr(s["a"]), L_.ptr(W), L_.ptr(s["resid"]), L_.ptr(nw), L_.ptr(sc), L_.i32(196), L_.ptr(s["y"]),
                                         L_.ptr(s["ro"]), L_.ptr(s["rs"]), L_.i32(M), L_.i32(d), L_.i32(K), ctypes.c_long(K), ctypes.c_long(K),
                                         ctypes.c_float(1e-5), L_.stream_of(s["a"])), "addnorm_rw")
nbytes = M * (K * 2 + d * (4 + 4 + 2) + 4)
for name, fn in (("tiled 64-row kernel", tiled), ("register-weight kernel", rw), ("tiled 64-row kernel", tiled), ("register-weight kernel", rw)):
    fns = rotating(fn, base, ("a", "resid", "y", "ro", "rs"), nbytes)
    t, tw = time_kernel(fns), time_kernel(fns[0])
    print(f"{name:24s} cold {t * 1e6:6.2f} us ({nbytes / t / 1e12:.2f} TB/s)   warm {tw * 1e6:6.2f} us", flush=True)
