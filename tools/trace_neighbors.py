"""For every dispatch whose name contains <pattern> in a rocprofv3 rocpd database: the kernels launched just before
and after it (to find which host op issues it).  usage: python tools/trace_neighbors.py <results.db> <pattern> [n]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("columns:", cols)
extra = [c for c in ("grid_size", "grid_x", "workgroup_size", "grid_size_x") if c in cols]
rows = db.execute(f"select name, start, end {''.join(', ' + c for c in extra)} from kernels order by start").fetchall()
seen = collections.Counter()
for i, r in enumerate(rows):
    if pat in r[0]:
        prev = rows[i - 1][0][:60] if i else ""
        nxt = rows[i + 1][0][:60] if i + 1 < len(rows) else ""
        seen[(prev, nxt, r[3:] )] += 1
for (prev, nxt, ex), c in seen.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print(f"{c:5d}  {ex}  after [{prev}]  before [{nxt}]")
