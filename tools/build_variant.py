"""Build a VARIANT of the kernel library into a scratch directory (the in-tree library and objects are never touched):
    python tools/build_variant.py OUT.so [extra hipcc flags ...]      e.g.  ab/tuning.so -DFASTVIM_TUNING_HOOKS
Load it with PROBE_LIB=OUT.so (bench.py, tools/probe/*.py).  Used for same-box A/Bs through the dispatchers' tuning hooks."""
import os, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from fastvim_amd import build as fb

out, extra = os.path.abspath(sys.argv[1]), sys.argv[2:]
tmp = tempfile.mkdtemp(prefix="fvvar_")


def comp(src):
    obj = os.path.join(tmp, os.path.basename(src) + ".o")
    cmd = [fb.HIPCC, *fb.FLAGS, *extra, *fb._file_flags(src), "-x", "hip", "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr)
    return obj


with ThreadPoolExecutor(max_workers=6) as ex:
    objs = list(ex.map(comp, fb._sources()))
r = subprocess.run([fb.HIPCC, "-shared", "-fPIC", f"--offload-arch={fb.ARCH}", "-o", out, *objs], capture_output=True, text=True)
if r.returncode:
    raise RuntimeError(r.stderr)
print(out)
