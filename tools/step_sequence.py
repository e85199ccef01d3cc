"""Kernel sequence of ONE graph-replayed training step from a rocprofv3 rocpd database (the dispatches between two
consecutive adamw_flat_kernel launches).  Library kernels (torch / copies) are printed with their neighbours so that
the host op issuing each can be identified.  usage: python tools/step_sequence.py <results.db> [all]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, stream_id from kernels order by start").fetchall() if "stream_id" in [r[1] for r in db.execute("pragma table_info(kernels)")] \
    else [(*r, 0) for r in db.execute("select name, start, end from kernels order by start").fetchall()]
marks = [i for i, r in enumerate(rows) if "adamw_flat_kernel" in r[0]]
lo, hi = marks[-2] + 1, marks[-1] + 1
step = rows[lo:hi]
ours = ("fvi::", "anonymous namespace", "gemm_", "scan_", "conv_pool", "combine_", "add_norm", "xproj", "reduce_partials", "adamw", "soft_ce", "bump_step")
def short(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "")[:70]
lib_us = 0.0
print(f"{len(step)} dispatches, {(step[-1][2] - step[0][1]) / 1e3:.1f} us wall")
for i, (n, s, e, st) in enumerate(step):
    mine = any(o in n for o in ours)
    if not mine: lib_us += (e - s) / 1e3
    if len(sys.argv) > 2 or not mine:
        prev = short(step[i - 1][0]) if i else ""
        gap = (s - step[i - 1][2]) / 1e3 if i else 0.0
        print(f"{i:4d} {(e - s) / 1e3:7.1f} us (gap {gap:5.1f}) st{st} {short(n)}   <- after [{prev[:40]}]")
print(f"library kernels: {lib_us:.1f} us per step")
