"""Summarise a rocprofv3 ``*_results.db`` (rocpd SQLite output of ``--kernel-trace --stats``) into the
per-kernel CSV the older ``--output-format csv`` produced: name, calls, total/avg/min/max ns, share.
usage: python tools/rocpd_stats.py <results.db> <out.csv> [skip_first_n_dispatches_per_kernel]"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 160 else name[:157] + "..."


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                      "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for n, c, t, a, mn, mx in rows:
            w.writerow([short(n), c, t, round(a, 1), mn, mx, round(100.0 * t / total, 3)])
    print(f"{len(rows)} kernels, {total / 1e6:.2f} ms of kernel time -> {out_path}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
