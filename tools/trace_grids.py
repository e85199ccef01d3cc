"""Per (kernel name pattern, grid size) dispatch count and average duration from a rocprofv3 rocpd database.
usage: python tools/trace_grids.py <results.db> <pattern>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select substr(name, 1, 46), grid_x * grid_y * grid_z, workgroup_x, lds_size, vgpr_count + accum_vgpr_count, count(*), avg(end - start), min(end - start), max(end - start) from kernels "
                  "where name like ? group by name, grid_x, grid_y, grid_z, workgroup_x order by count(*) * avg(end - start) desc", ("%" + sys.argv[2] + "%",)).fetchall()
for nm, g, w, lds, regs, c, a, mn, mx in rows:
    print(f"{nm:46s} blocks {g // max(w, 1):>7} x {w:>4}  lds {lds:>6}  regs {regs:>3}  calls {c:>5}  avg {a / 1e3:7.1f} us  min {mn / 1e3:7.1f}  max {mx / 1e3:7.1f}")
