"""Round 6: fv_mixer_conv_pool_bwd_dgrad against fv_mixer_conv_pool_bwd2 + fv_gemm_bf16_dgrad_addnorm_bwd2 at the FastVim-T
shape, HBM-cold (operand sets rotated past the Infinity Cache), timed with events around graphs of 24 launches.
usage: python tools/probe/r06_convdgrad_time.py [sets]      (PROBE_LIB=<path>: load that build of the library instead)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fastvim_amd import _lib as L_
if os.environ.get("PROBE_LIB"):
    L_.LIB_PATH = os.environ["PROBE_LIB"]
from fastvim_amd import mixer_ops as M

B, rows, cols, d_in, d = 128, 14, 14, 384, 192
Mrows, rps = B * rows * cols, rows * cols
nset = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
sets = []
for _ in range(nset):
    sets.append(dict(xz=rn(B, rps, 2 * d_in).bfloat16(), d_o=rn(B, rps, d_in).bfloat16(), dxc=rn(2, B, rows, d_in),
                     dxc2=rn(2, B, rows, d_in).bfloat16(), dxz=rn(B, rps, 2 * d_in).bfloat16(), gg=rn(Mrows, d), r=rn(Mrows, d),
                     rstd=0.5 + torch.rand(Mrows, device=dev, generator=g)))
cw, cb, cwb, cbb = 0.5 * rn(d_in, 4), 0.1 * rn(d_in), 0.5 * rn(d_in, 4), 0.1 * rn(d_in)
D, Db = 1 + 0.1 * rn(d_in), 1 + 0.1 * rn(d_in)
W_in = (rn(2 * d_in, d) * d ** -0.5).bfloat16()
W_in_t = W_in.t().contiguous()
W_out = (rn(d, d_in) * d_in ** -0.5).bfloat16()
nw = 1 + 0.1 * rn(d)
sc = torch.ones(B, device=dev)
lib = L_.lib()
nb0 = lib.fv_gemm_bf16_dgrad_addnorm_blocks(L_.i32(Mrows))
for s in sets:
    s["dx"] = torch.empty(Mrows, d, device=dev, dtype=torch.bfloat16); s["dri"] = torch.empty(Mrows, d, device=dev)
    s["pw"] = torch.empty(nb0, d, device=dev); s["dg"] = torch.empty(Mrows, d_in, device=dev, dtype=torch.bfloat16)


def fused(s, tr):
    M.conv_pool_bwd_dgrad(s["xz"], s["d_o"], s["dxc"], s["dxc2"], cw, cb, cwb, cbb, D, Db, s["dxz"], rows, cols, tr, 1.0, W_in_t,
                          s["gg"], s["r"], s["rstd"], nw, sc, rps, W2=W_out)


def conv(s, tr):
    M.conv_pool_bwd(s["xz"], s["d_o"], s["dxc"], cw, cb, cwb, cbb, D, Db, s["dxz"], rows, cols, tr, False, 1.0, dxc2=s["dxc2"])


def dgrad(s, tr):
    L_.check(lib.fv_gemm_bf16_dgrad_addnorm_bwd2(
        L_.ptr(s["dxz"]), L_.ptr(W_in), L_.ptr(s["gg"]), L_.ptr(s["r"]), L_.ptr(s["rstd"]), L_.ptr(nw), L_.ptr(sc), L_.i32(rps),
        L_.ptr(s["dx"]), L_.ptr(s["dri"]), L_.ptr(s["pw"]), L_.i32(Mrows), L_.i32(d), L_.i32(2 * d_in), ctypes.c_long(2 * d_in),
        ctypes.c_long(d), L_.ptr(W_out), L_.ptr(s["dg"]), L_.i32(d_in), ctypes.c_long(d_in), L_.stream_of(s["dxz"])), "dgrad")


def timeit(fn, tr, reps=4):
    n = 24
    gr = torch.cuda.CUDAGraph()
    fn(sets[0], tr)
    torch.cuda.synchronize()
    with torch.cuda.graph(gr):
        for i in range(n):
            fn(sets[i % nset], tr)
    gr.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / n)
    return best


only_fused = os.environ.get("PROBE_ONLY_FUSED") == "1"
for tr in (False, True):
    if only_fused:
        print(f"transposed={tr}: fused {timeit(fused, tr):.2f} us", flush=True)
    else:
        print(f"transposed={tr}: fused {timeit(fused, tr):.2f} us   conv_pool_bwd {timeit(conv, tr):.2f} us   dgrad+norm adjoint {timeit(dgrad, tr):.2f} us", flush=True)
