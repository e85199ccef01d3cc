"""Build libfastvim_hip.so (gfx950) in-tree with hipcc.  No torch headers involved:
the library is a plain C-ABI shared object (include/fastvim_hip.h)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "csrc", "_obj")
LIB = os.path.join(PKG, "libfastvim_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = os.environ.get("FASTVIM_ARCH", "gfx950")      # build-time A/B knob: e.g. gfx950:xnack-
# Per-source flags come from a source's first line (``// hipcc-flags: ...``).  -fgpu-flush-denormals-to-zero is one of
# them: fp32 denormals are flushed in the scan / conv / mixer kernels only -- the code that replaces the reference's own
# CUDA extension, which is built with nvcc --use_fast_math (implies --ftz=true: mamba-1p1p1/setup.py:102-153); the expansions
# of exp / log / rcp / rsqrt lose their denormal-range scaling code (FastVim-T step 5.736 -> 5.708 ms on one box,
# profiles/r04_ab_flush_denormals.log).  The optimizer, the loss, the norms, the GEMMs and the glue kernels stand in for
# stock PyTorch / Triton code that does not flush, and are built without it (round 5, advisor).
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffast-math", "-fno-finite-math-only",
         "-Wno-unused-result", "-DNDEBUG"]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _deps_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(PKG, "..", "include", "fastvim_hip.h"))
    hs.append(os.path.abspath(__file__))
    return max(os.path.getmtime(h) for h in hs)


def _file_flags(src):
    """Extra flags a source asks for on its first line: ``// hipcc-flags: -fno-slp-vectorize``."""
    with open(src) as f:
        first = f.readline()
    return first.split("hipcc-flags:", 1)[1].split() if "hipcc-flags:" in first else []


def _compile(src, obj, verbose):
    cmd = [HIPCC, *FLAGS, *os.environ.get("FASTVIM_EXTRA_FLAGS", "").split(), *_file_flags(src), "-x", "hip", "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr)
    return obj


def build(force=False, verbose=False, tuning=False):
    """``tuning=True`` (``python -m fastvim_amd.build --tuning``): compile the kernel dispatchers' A/B hooks in
    (FASTVIM_* environment variables, tools/README.md); the default library reads no environment variable."""
    os.makedirs(OBJ, exist_ok=True)
    stamp = os.path.join(OBJ, ".tuning")
    was = os.path.exists(stamp)
    if was != tuning:
        force = True
    if tuning and "-DFASTVIM_TUNING_HOOKS" not in FLAGS:
        FLAGS.append("-DFASTVIM_TUNING_HOOKS")
    if not tuning and "-DFASTVIM_TUNING_HOOKS" in FLAGS:
        FLAGS.remove("-DFASTVIM_TUNING_HOOKS")
    extra = os.environ.get("FASTVIM_EXTRA_FLAGS", "").split()      # build-time A/B knobs (e.g. -DFV_BUF_STORE_AUX=16)
    # what the objects were compiled with: a change of target, flags or A/B knobs -- in either direction -- rebuilds all
    sig = " ".join([ARCH, *FLAGS, "|", *extra])
    fstamp = os.path.join(OBJ, ".flags")
    if not os.path.exists(fstamp) or open(fstamp).read() != sig:
        force = True
    if force:
        # the stamps describe what the OBJECTS were built with: they are withdrawn now and written again only after every
        # compile job and the link have succeeded -- an interrupted or failed rebuild leaves no stamp, so the next run
        # rebuilds everything instead of linking objects of two flag sets into one library (round 5, advisor)
        for st_ in (fstamp, stamp):
            if os.path.exists(st_):
                os.remove(st_)
    hdr_m = _deps_mtime()
    jobs, objs = [], []
    for src in _sources():
        obj = os.path.join(OBJ, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m):
            jobs.append((src, obj))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(lambda j: _compile(j[0], j[1], verbose), jobs))
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if not os.path.exists(fstamp):
        with open(fstamp, "w") as f:
            f.write(sig)
    if tuning and not os.path.exists(stamp):
        open(stamp, "w").close()
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, tuning="--tuning" in sys.argv))
