"""fv_gemm_bf16_dgrad_addnorm_bwd against gemm_nn + fv_add_norm_bwd: values and time (rotating operand sets)."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from bench import time_kernel
from fastvim_amd import _lib as L
from fastvim_amd.gemm import gemm_nn
M, d, d_in, dev = 128 * 196, 192, 384, "cuda"
K = 2 * d_in
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
SETS = 6
W = (rn(K, d) * K ** -0.5).bfloat16()
nw = 1 + 0.1 * rn(d)
scale = ((torch.rand(128, device=dev, generator=g) > 0.2).float() / 0.8).contiguous()
sets = [dict(dxz=rn(M, K).bfloat16(), gg=rn(M, d), r=rn(M, d), rstd=torch.rand(M, device=dev, generator=g) + 0.5) for _ in range(SETS)]
lib = L.lib()
nb_f = lib.fv_gemm_bf16_dgrad_addnorm_blocks(L.i32(M)); nb_u = lib.fv_add_norm_blocks(L.i32(M))
out = dict(dx=torch.empty(M, d, device=dev, dtype=torch.bfloat16), dri=torch.empty(M, d, device=dev), pw=torch.empty(nb_f, d, device=dev))
out2 = dict(dx=torch.empty(M, d, device=dev, dtype=torch.bfloat16), dri=torch.empty(M, d, device=dev), pw=torch.empty(nb_u, d, device=dev))
i = [0]
def fused():
    s = sets[i[0] % SETS]; i[0] += 1
    rc = lib.fv_gemm_bf16_dgrad_addnorm_bwd(L.ptr(s["dxz"]), L.ptr(W), L.ptr(s["gg"]), L.ptr(s["r"]), L.ptr(s["rstd"]), L.ptr(nw), L.ptr(scale),
                                            L.i32(196), L.ptr(out["dx"]), L.ptr(out["dri"]), L.ptr(out["pw"]), L.i32(M), L.i32(d), L.i32(K),
                                            ctypes.c_long(K), ctypes.c_long(d), L.stream_of(W))
    L.check(rc, "fused")
def unfused():
    s = sets[i[0] % SETS]; i[0] += 1
    dy = gemm_nn(s["dxz"], W)
    rc = lib.fv_add_norm_bwd(L.ptr(dy), L.i32(L.FV_BF16), L.ptr(s["gg"]), L.i32(L.FV_F32), L.ptr(s["r"]), L.i32(L.FV_F32), L.ptr(nw), L.ptr(None),
                             L.ptr(s["rstd"]), L.ptr(scale), L.i32(196), L.ptr(out2["dx"]), L.i32(L.FV_BF16), L.ptr(out2["dri"]), L.i32(L.FV_F32),
                             L.ptr(out2["pw"]), L.ptr(None), L.i32(M), L.i32(d), L.i32(1), L.stream_of(W))
    L.check(rc, "unfused")
W_out = (rn(d, d_in) * d ** -0.5).bfloat16()          # out_proj.weight (192, 384): d g = d x @ W_out
dg = torch.empty(M, d_in, device=dev, dtype=torch.bfloat16)
def fused2():
    s = sets[i[0] % SETS]; i[0] += 1
    rc = lib.fv_gemm_bf16_dgrad_addnorm_bwd2(L.ptr(s["dxz"]), L.ptr(W), L.ptr(s["gg"]), L.ptr(s["r"]), L.ptr(s["rstd"]), L.ptr(nw), L.ptr(scale),
                                             L.i32(196), L.ptr(out["dx"]), L.ptr(out["dri"]), L.ptr(out["pw"]), L.i32(M), L.i32(d), L.i32(K),
                                             ctypes.c_long(K), ctypes.c_long(d), L.ptr(W_out), L.ptr(dg), L.i32(d_in), ctypes.c_long(d_in),
                                             L.stream_of(W))
    L.check(rc, "fused2")
def fused_then_dgrad():
    fused(); return gemm_nn(out["dx"], W_out)
i[0] = 0; fused2(); torch.cuda.synchronize()
ref_dg = gemm_nn(out["dx"], W_out)
print("second phase d g equal:", torch.equal(dg, ref_dg), " max|diff|:", (dg.float() - ref_dg.float()).abs().max().item())
i[0] = 0; fused(); i[0] = 0; unfused(); torch.cuda.synchronize()
print("dx equal:", torch.equal(out["dx"], out2["dx"]), " dres_in equal:", torch.equal(out["dri"], out2["dri"]),
      " max|diff| dres_in:", (out["dri"] - out2["dri"]).abs().max().item(),
      " dw rel diff:", ((out["pw"].sum(0) - out2["pw"].sum(0)).abs().max() / out2["pw"].sum(0).abs().max()).item())
tf = time_kernel(fused, iters=24); tu = time_kernel(unfused, iters=24)
t2 = time_kernel(fused2, iters=24); t3 = time_kernel(fused_then_dgrad, iters=24)
print(f"fused {tf * 1e6:.1f} us   gemm_nn + add_norm_bwd {tu * 1e6:.1f} us   fused with second phase {t2 * 1e6:.1f} us   fused + out_proj dgrad launch {t3 * 1e6:.1f} us")
