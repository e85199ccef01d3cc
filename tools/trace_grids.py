"""Per (kernel name pattern, grid size) dispatch count and average duration from a rocprofv3 rocpd database.
usage: python tools/trace_grids.py <results.db> <pattern>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select grid_x, workgroup_x, count(*), avg(end - start), min(end - start), max(end - start) from kernels "
                  "where name like ? group by grid_x, workgroup_x order by 4 desc", ("%" + sys.argv[2] + "%",)).fetchall()
for g, w, c, a, mn, mx in rows:
    print(f"grid {g:>9} wg {w:>4} blocks {g // max(w, 1):>7}  calls {c:>5}  avg {a / 1e3:8.1f} us  min {mn / 1e3:8.1f}  max {mx / 1e3:8.1f}")
