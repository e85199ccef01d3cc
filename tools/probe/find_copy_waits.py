"""usage: compile every csrc/*.hip with `hipcc ... -S --cuda-device-only -o /tmp/isa/all_<name>.s` (the flags of
fastvim_amd/build.py), then `python tools/probe/find_copy_waits.py`.
Heuristic: inside a loop, a `s_waitcnt vmcnt(N)` directly followed (within 3 lines) by a plain `v_mov_b32 vD, vS` whose
source vS was the destination of a global / buffer load issued within the previous 120 lines: the load went to a
temporary and the copy into the loop-carried register waits for it."""
import re, sys, glob, subprocess
def demangle(n):
    try: return subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()[:110]
    except Exception: return n
for f in sorted(glob.glob('/tmp/isa/all_*.s')):
    lines=open(f).read().split('\n')
    kern=None; hits={}
    loads=[]  # (lineno, set(regs))
    for i,l in enumerate(lines):
        m=re.match(r'^(_Z\w+):', l)
        if m: kern=m.group(1); loads=[]
        m=re.match(r'\s*(global_load|buffer_load)_\w+\s+v(\[(\d+):(\d+)\]|(\d+))', l)
        if m:
            if m.group(3): regs=set(range(int(m.group(3)), int(m.group(4))+1))
            else: regs={int(m.group(5))}
            loads.append((i,regs))
            loads=[x for x in loads if i-x[0]<150]
        if 's_waitcnt vmcnt' in l and kern:
            for k in range(i+1, min(i+4,len(lines))):
                mm=re.match(r'\s*v_mov_b32_e32\s+v(\d+),\s*v(\d+)\s*$', lines[k])
                if mm:
                    src=int(mm.group(2))
                    if any(src in r for (ln,r) in loads if i-ln<150):
                        hits[kern]=hits.get(kern,0)+1
                    break
    for k,v in sorted(hits.items(), key=lambda kv:-kv[1]):
        if v>=3: print(f.split('all_')[1], v, demangle(k))
