"""fv_gemm_bf16_addnorm2 (out_proj + add + RMSNorm + in_proj in one launch) against the three-kernel path."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from bench import time_kernel
from fastvim_amd import _lib as L
from fastvim_amd.gemm import gemm_nt
M, d, d_in, dev = (int(sys.argv[1]) if len(sys.argv) > 1 else 128 * 196), 192, 384, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
SETS = 6
W_out = (rn(d, d_in) * d_in ** -0.5).bfloat16(); W_in = (rn(2 * d_in, d) * d ** -0.5).bfloat16()
nw = 1 + 0.1 * rn(d)
sets = [dict(g=rn(M, d_in).bfloat16(), res=rn(M, d)) for _ in range(SETS)]
lib = L.lib()
y, ro, rs = torch.empty(M, d, device=dev, dtype=torch.bfloat16), torch.empty(M, d, device=dev), torch.empty(M, device=dev)
xz = torch.empty(M, 2 * d_in, device=dev, dtype=torch.bfloat16)
i = [0]
def call(second):
    s = sets[i[0] % SETS]; i[0] += 1
    rc = lib.fv_gemm_bf16_addnorm2(L.ptr(s["g"]), L.ptr(W_out), L.ptr(s["res"]), L.ptr(nw), L.ptr(None), L.i32(1), L.ptr(y), L.ptr(ro), L.ptr(rs),
                                   L.i32(M), L.i32(d), L.i32(d_in), ctypes.c_long(d_in), ctypes.c_long(d_in), ctypes.c_float(1e-5),
                                   L.ptr(W_in if second else None), L.ptr(xz if second else None), L.i32(2 * d_in if second else 0),
                                   ctypes.c_long(d), L.stream_of(y))
    L.check(rc, "addnorm2")
def fused2(): call(True)
def fused_then_inproj():
    call(False); return gemm_nt(y, W_in)
i[0] = 0; fused2(); torch.cuda.synchronize()
ref = gemm_nt(y, W_in)
print("second phase xz equal:", torch.equal(xz, ref), " max|diff|:", (xz.float() - ref.float()).abs().max().item())
t2 = time_kernel(fused2, iters=24); t3 = time_kernel(fused_then_inproj, iters=24)
print(f"fused with in_proj {t2 * 1e6:.1f} us   fused + in_proj launch {t3 * 1e6:.1f} us")
# y / residual_out / rstd with and without the second phase, repeated
i[0] = 0; call(False); torch.cuda.synchronize()
y0, ro0, rs0 = y.clone(), ro.clone(), rs.clone()
bad = 0
for rep in range(30):
    i[0] = 0; y.zero_(); ro.zero_(); rs.zero_(); xz.zero_()
    call(True); torch.cuda.synchronize()
    ok = torch.equal(y, y0) and torch.equal(ro, ro0) and torch.equal(rs, rs0) and torch.equal(xz, gemm_nt(y0, W_in))
    bad += not ok
print("runs with a difference:", bad, "of 30")
