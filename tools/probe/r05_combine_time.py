"""Round 5: fv_mixer_combine_out_proj_addnorm against fv_mixer_combine_fwd + fv_gemm_bf16_addnorm at the FastVim-T shape,
HBM-cold (operand sets rotated past the Infinity Cache), timed with events around graphs of 24 launches.
usage: python tools/probe/r05_combine_time.py [sets]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fastvim_amd import _lib as L_
if os.environ.get("PROBE_LIB"):      # a scratch build of the library (tools/probe/r05_combine_phases.sh)
    L_.LIB_PATH = os.environ["PROBE_LIB"]
from fastvim_amd import mixer_ops as M

B, rows, cols, d_in, d = 128, 14, 14, 384, 192
Mrows = B * rows * cols
nset = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
sets = []
for _ in range(nset):
    sets.append(dict(xz=rn(B, rows * cols, 2 * d_in).bfloat16(), skip=rn(B, rows * cols, d_in).bfloat16(), yc=rn(2, B, rows, d_in),
                     res=rn(Mrows, d), out=None))
lw, lb = 1 + 0.1 * rn(d_in), 0.1 * rn(d_in)
W = (rn(d, d_in) * d_in ** -0.5).bfloat16()
nw = 1 + 0.1 * rn(d)
sc = torch.ones(B, device=dev)
for s in sets:
    s["out"] = M.combine_buffers(s["xz"], lw)
    s["y"] = torch.empty(Mrows, d, device=dev, dtype=torch.bfloat16); s["ro"] = torch.empty(Mrows, d, device=dev); s["rs"] = torch.empty(Mrows, device=dev)
lib = L_.lib()


def fused(s, transposed):
    M.combine_out_proj_addnorm(s["xz"], s["skip"], s["yc"], lw, lb, 1e-5, rows, cols, transposed, s["out"], W, s["res"], nw, sc, rows * cols, 1e-5)


def comb(s, transposed):
    M.combine_fwd(s["xz"], s["skip"], s["yc"], lw, lb, 1e-5, rows, cols, transposed, out=s["out"])


def gemm(s, transposed):
    L_.check(lib.fv_gemm_bf16_addnorm(L_.ptr(s["out"][0]), L_.ptr(W), L_.ptr(s["res"]), L_.ptr(nw), L_.ptr(sc), L_.i32(rows * cols),
                                      L_.ptr(s["y"]), L_.ptr(s["ro"]), L_.ptr(s["rs"]), L_.i32(Mrows), L_.i32(d), L_.i32(d_in),
                                      ctypes.c_long(d_in), ctypes.c_long(d_in), ctypes.c_float(1e-5), L_.stream_of(W)), "gemm")


def timeit(fn, transposed, reps=4):
    n = 24
    gr = torch.cuda.CUDAGraph()
    fn(sets[0], transposed)
    torch.cuda.synchronize()
    with torch.cuda.graph(gr):
        for i in range(n):
            fn(sets[i % nset], transposed)
    gr.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / n)
    return best


for tr in (False, True):
    print(f"transposed={tr}: fused {timeit(fused, tr):.2f} us   combine {timeit(comb, tr):.2f} us   out_proj+add+norm {timeit(gemm, tr):.2f} us", flush=True)
