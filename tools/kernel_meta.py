"""Per-kernel code-object metadata of libfastvim_hip.so (registers, spills, scratch, LDS) -- no GPU needed.

    python tools/kernel_meta.py                 # every kernel that spills or uses scratch
    python tools/kernel_meta.py --all           # every kernel
    python tools/kernel_meta.py scan_cl_bwd     # kernels whose demangled name contains the substring

The library is copied to a scratch directory first: `llvm-objdump --offloading` writes the bundles it extracts next to
its input.  tests/test_no_spills.py asserts on `kernels()` for the instantiations the BASELINE configurations dispatch.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "fastvim_amd", "libfastvim_hip.so")
LLVM = os.environ.get("FASTVIM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")

_FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
           "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size")


def _demangle(names):
    filt = shutil.which("c++filt") or os.path.join(LLVM, "llvm-cxxfilt")
    r = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True)
    out = r.stdout.split("\n")[:len(names)]
    return [o.replace("(anonymous namespace)::", "") for o in out]


def kernels(lib=LIB):
    """[{name, mangled, vgpr_count, vgpr_spill_count, private_segment_fixed_size, ...}] for every kernel of the library."""
    tmp = tempfile.mkdtemp(prefix="fvmeta_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], capture_output=True, check=True, cwd=tmp)
        rows = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)],
                                   capture_output=True, text=True, check=True).stdout
            cur = None
            for line in notes.split("\n"):
                m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)\s*$", line)
                if not m:
                    continue
                key, val = m.group(1), m.group(2)
                if line.lstrip().startswith("- .") and line.startswith("  - "):
                    cur = {}
                    rows.append(cur)
                if cur is None:
                    continue
                if key == "name":
                    cur["mangled"] = val
                elif key in _FIELDS:
                    cur[key] = int(val)
        rows = [r for r in rows if "mangled" in r]
        for r, n in zip(rows, _demangle([r["mangled"] for r in rows])):
            r["name"] = n
        return rows
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def spills(row):
    return row.get("vgpr_spill_count", 0) > 0 or row.get("private_segment_fixed_size", 0) > 0


def short(name):
    """Kernel name without its parameter list: `scan_cl_bwd_short_kernel<bf16, 3, 14, true, true>`."""
    depth = 0
    for i, c in enumerate(name):
        if c == "<":
            depth += 1
        elif c == ">":
            depth -= 1
        elif c == "(" and depth == 0:
            return name[:i].replace("void ", "").replace("__hip_bfloat16", "bf16")
    return name.replace("void ", "").replace("__hip_bfloat16", "bf16")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = kernels()
    sel = [r for r in rows if ("--all" in sys.argv or args or spills(r)) and all(a in r["name"] for a in args)]
    for r in sorted(sel, key=lambda r: r["name"]):
        print(f"{r.get('vgpr_count', 0):4d} vgpr {r.get('agpr_count', 0):3d} agpr {r.get('vgpr_spill_count', 0):4d} vspill "
              f"{r.get('sgpr_spill_count', 0):3d} sspill {r.get('private_segment_fixed_size', 0):5d} scratch "
              f"{r.get('group_segment_fixed_size', 0):6d} lds  {short(r['name'])}")
    print(f"{len(rows)} kernels, {sum(spills(r) for r in rows)} with spills / scratch", file=sys.stderr)
