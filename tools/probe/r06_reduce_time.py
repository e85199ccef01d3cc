"""The deferred fixed-order reductions of one FastVim-T step (job list of tools/probe/r06_reduce_jobs.py), timed stand-alone:
three fv_reduce_partials_multi launches, HBM-cold (550 MB of partials are read once per replay).  PROBE_LIB selects a library.
usage: python tools/probe/r06_reduce_time.py"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from fastvim_amd import _lib as L_
if os.environ.get("PROBE_LIB"):
    L_.LIB_PATH = os.environ["PROBE_LIB"]
lib = L_.lib()
launches = [
    [(512, 4608)] * 23 + [(64, 22272)] * 24 + [(448, 768)] * 24 + [(512, 192)] * 23 + [(224, 4608), (2048, 192)],
    [(6, 147456)] * 24 + [(6, 73728)] * 20 + [(4, 16896)] * 48 + [(2048, 192), (2, 192000), (1, 37632), (196, 192)],
    [(6, 73728)] * 4 + [(6, 147456)],
]
if "--classes" in sys.argv:          # one launch per job class instead
    launches = [[(512, 4608)] * 23, [(64, 22272)] * 24, [(448, 768)] * 24, [(512, 192)] * 23, [(6, 147456)] * 24, [(4, 16896)] * 48]
dev = "cuda"
data = []
for jobs in launches:
    parts = [torch.randn(S, n, device=dev) for S, n in jobs]
    outs = [torch.zeros(n, device=dev) for _, n in jobs]
    data.append((parts, outs))


def launch(parts, outs):
    k = len(parts)
    ins = (ctypes.c_void_p * k)(*[p.data_ptr() for p in parts])
    os_ = (ctypes.c_void_p * k)(*[o.data_ptr() for o in outs])
    Ss = (ctypes.c_int * k)(*[p.shape[0] for p in parts])
    ns = (ctypes.c_size_t * k)(*[p.shape[1] for p in parts])
    L_.check(lib.fv_reduce_partials_multi(ins, os_, Ss, ns, L_.i32(k), L_.i32(1), L_.stream_of(parts[0])), "reduce")


for i, (parts, outs) in enumerate(data):
    launch(parts, outs)
torch.cuda.synchronize()
for i, (parts, outs) in enumerate(data):
    mb = sum(p.numel() for p in parts) * 4 / 1e6
    best = 1e9
    for _ in range(6):
        for j, (p2, o2) in enumerate(data):          # the other launches' partials push this one's out of the caches
            if j != i:
                launch(p2, o2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); launch(parts, outs); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000)
    print(f"launch {i}: {len(parts)} jobs, {mb:.1f} MB read: {best:.1f} us = {mb / best * 1e6 / 1e6:.2f} TB/s", flush=True)
chk = sum(float(o.double().sum()) for _, outs in data for o in outs)
print(f"checksum {chk:.6e}")
