"""usage: as tools/probe/find_copy_waits.py (assembly of every csrc/*.hip in /tmp/isa/all_<name>.s).
For every kernel, walk the instruction stream with an in-order model of the vector-memory counter and report the LOADS
that a `s_waitcnt vmcnt(N)` retires fewer than DIST instructions after their issue while sitting inside a loop body --
candidates for "requested and waited for in the same breath".  Straight-line order only (branches are not followed), so
a hit is a place to read, not a verdict."""
import re, glob, subprocess, sys
DIST = int(sys.argv[1]) if len(sys.argv) > 1 else 40
def demangle(n):
    return subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()[:120]
rows = []
for f in sorted(glob.glob('/tmp/isa/all_*.s')):
    kern = None; q = []; n = 0; inloop = False; hits = 0; total = 0
    def flush():
        if kern and hits >= 4: rows.append((hits, total, f.split('all_')[1], kern))
    for l in open(f):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            flush(); kern = m.group(1); q = []; n = 0; inloop = False; hits = 0; total = 0; continue
        if 'Loop Header' in l: inloop = True
        if 's_endpgm' in l: inloop = False
        t = l.strip()
        if not t or t.startswith(('.', ';')): continue
        n += 1
        if re.match(r'(global|buffer|flat|scratch)_(load|store|atomic)', t):
            q.append((n, 'load' in t.split()[0], inloop))
        m = re.match(r's_waitcnt.*vmcnt\((\d+)\)', t)
        if m:
            k = int(m.group(1))
            while len(q) > k:
                n0, isload, lp = q.pop(0)
                if isload and lp:
                    total += 1
                    if n - n0 < DIST: hits += 1
    flush()
for hits, total, fn, kern in sorted(rows, reverse=True)[:60]:
    print(f"{hits:4d} of {total:4d} in-loop loads retired < {DIST} instructions after issue  {fn:22s} {demangle(kern)}")
