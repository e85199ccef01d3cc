"""Time fv_reduce_partials_multi on the gradient-partial mix of FastVim-T layers (16 jobs per launch, as in the step)."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from fastvim_amd import _lib as L
from bench import time_kernel
layer = [(28, 768 * 192), (28, 192 * 384), (448, 12 * 384), (128, 2 * 384 * 29), (448, 2 * 384), (112, 2 * 44 * 384)]
jobs = (layer * 3)[:16]
parts = [torch.randn(S, n, device="cuda") for S, n in jobs]
outs = [torch.zeros(n, device="cuda") for S, n in jobs]
k = len(jobs)
ins = (ctypes.c_void_p * k)(*[p.data_ptr() for p in parts])
os_ = (ctypes.c_void_p * k)(*[o.data_ptr() for o in outs])
Ss = (ctypes.c_int * k)(*[S for S, n in jobs])
ns = (ctypes.c_size_t * k)(*[n for S, n in jobs])
def run():
    rc = L.lib().fv_reduce_partials_multi(ins, os_, Ss, ns, L.i32(k), L.i32(0), L.stream_of(parts[0]))
    assert rc == 0
run()
torch.cuda.synchronize()
for o, p in zip(outs, parts):
    assert torch.allclose(o, p.sum(0), rtol=1e-4, atol=1e-3)
t = time_kernel(run, iters=20)
by = sum(S * n * 4 + n * 4 for S, n in jobs)
print(f"reduce_multi 16 jobs: {t*1e6:.1f} us, {by/1e6:.1f} MB, {by/t/1e9:.0f} GB/s")
