"""Where a workgroup of the fused projection + norm kernels spends its life (tuning build: phase stamps, FV_STAMP in
csrc/gemm_mfma.hip).  HBM-cold: the launches rotate through operand sets; the stamps of the LAST launch are read.
usage: python -m fastvim_amd.build --tuning && python tools/probe/fused_stamps.py [bwd|fwd]"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from fastvim_amd import _lib as L
which = sys.argv[1] if len(sys.argv) > 1 else "bwd"
M, d, d_in, dev = 128 * 196, 192, 384, "cuda"
K = 2 * d_in
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
SETS = 5
lib = L.lib()
nb = lib.fv_gemm_bf16_dgrad_addnorm_blocks(L.i32(M))
W_in = (rn(K, d) * K ** -0.5).bfloat16()
W_out = (rn(d, d_in) * d ** -0.5).bfloat16()
nw = 1 + 0.1 * rn(d)
scale = torch.ones(128, device=dev)
sets = [dict(dxz=rn(M, K).bfloat16(), g2=rn(M, d_in).bfloat16(), gg=rn(M, d), r=rn(M, d), rstd=torch.rand(M, device=dev, generator=g) + 0.5,
             dx=torch.empty(M, d, device=dev, dtype=torch.bfloat16), dri=torch.empty(M, d, device=dev), pw=torch.empty(nb, d, device=dev),
             dg=torch.empty(M, d_in, device=dev, dtype=torch.bfloat16), rs=torch.empty(M, device=dev)) for _ in range(SETS)]
stamps = torch.zeros(nb, 8, device=dev, dtype=torch.int64)
lib._cdll.fv_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))

def launch(s):
    if which == "bwd":
        rc = lib.fv_gemm_bf16_dgrad_addnorm_bwd2(L.ptr(s["dxz"]), L.ptr(W_in), L.ptr(s["gg"]), L.ptr(s["r"]), L.ptr(s["rstd"]), L.ptr(nw),
                                                 L.ptr(scale), L.i32(196), L.ptr(s["dx"]), L.ptr(s["dri"]), L.ptr(s["pw"]), L.i32(M), L.i32(d),
                                                 L.i32(K), ctypes.c_long(K), ctypes.c_long(d), L.ptr(W_out), L.ptr(s["dg"]), L.i32(d_in),
                                                 ctypes.c_long(d_in), L.stream_of(W_in))
    else:
        rc = lib.fv_gemm_bf16_addnorm(L.ptr(s["g2"]), L.ptr(W_out), L.ptr(s["r"]), L.ptr(nw), L.ptr(scale), L.i32(196), L.ptr(s["dx"]),
                                      L.ptr(s["dri"]), L.ptr(s["rs"]), L.i32(M), L.i32(d), L.i32(d_in), ctypes.c_long(d_in),
                                      ctypes.c_long(d_in), ctypes.c_float(1e-5), L.stream_of(W_in))
    L.check(rc, which)

for i in range(2 * SETS):
    launch(sets[i % SETS])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(SETS - 1):
    launch(sets[i])
e0.record(); launch(sets[SETS - 1]); e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
raw = stamps.cpu()
torch.save(raw, os.path.join(R, "gpurun_out", f"r03_stamps_{which}.pt")) if os.path.isdir(os.path.join(R, "gpurun_out")) else None
slots_used = [0, 1, 2, 3, 4, 5, 6] if which == "bwd" else [0, 1, 2, 3, 4, 6]
ok = (raw[:, slots_used] != 0).all(1)
print(f"workgroups with every stamp written: {int(ok.sum())} of {nb}")
t = raw[ok].double()
slots = [0, 1, 2, 3, 4, 5, 6] if which == "bwd" else [0, 1, 2, 3, 4, 6]
names = {0: "start", 1: "first stage landed", 2: "K loop done", 3: "product tile in LDS", 4: "norm rows done (stores issued)",
         5: "partial d weight written", 6: "second phase / end"}
life = t[:, 6] - t[:, 0]
print(f"{which}: launch {us:.1f} us (events, incl. launch overhead); workgroup life median {life.median().item():.0f} ticks "
      f"(min {life.min().item():.0f}, max {life.max().item():.0f}; s_memtime ticks, one counter per XCD: only differences inside a workgroup mean anything)")
prev = 0
for s_ in slots[1:]:
    dlt = t[:, s_] - t[:, prev]
    print(f"  {names[prev]:34s} -> {names[s_]:34s}: median {dlt.median().item():8.0f} ticks ({100 * dlt.median().item() / life.median().item():5.1f} % of life)"
          f"  p10 {dlt.quantile(0.1).item():8.0f}  p90 {dlt.quantile(0.9).item():8.0f}")
    prev = s_
