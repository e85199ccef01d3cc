"""bench.py with the split-K factor of the FastVim-T projection weight gradients overridden (round-5 re-sweep of round 3's
choice of six slices).  usage: python tools/probe/r05_wgrad_splits.py <in_proj slices> <out_proj slices> [bench.py args]"""
import os, runpy, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import fastvim_amd.gemm as g
si, so = int(sys.argv[1]), int(sys.argv[2])
orig = g.grouped_splits
def patched(Kd, target=None, M=None, N=None):
    if target is None and M is not None and N is not None and Kd % 64 == 0:
        kt = Kd // 64
        want = si if (M, N) == (768, 192) else so if (M, N) == (192, 384) else 0
        if want:
            per = -(-kt // want)
            if -(-kt // per) == want:
                return want
    return orig(Kd, target, M, N)
g.grouped_splits = patched
import fastvim_amd.mamba_simple_faster as msf
if hasattr(msf, "grouped_splits"): msf.grouped_splits = patched
sys.argv = ["bench.py", "--no-other-configs", "--no-cpu-baseline", "--no-kernels", "--no-scan-op"] + sys.argv[3:]
import io, json, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path(os.path.join(R, "bench.py"), run_name="__main__")
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
print(f"in_proj {si} / out_proj {so} slices: {d['ms_per_step']} ms  loss {d['config'].get('final_loss')}")
