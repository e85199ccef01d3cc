"""What the deferred fixed-order reductions of one FastVim-T step read: every fv_reduce_partials_multi launch's job list
(partials x length), printed once.  usage: python tools/probe/r06_reduce_jobs.py [MODEL IMG BATCH]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import bench
from fastvim_amd import mixer_ops as M
a = sys.argv[1:]
model, img, batch = (a[0], int(a[1]), int(a[2])) if len(a) >= 3 else ("T", 224, 128)
seen = []
orig = M._Deferred._launch.__func__
def launch(cls, jobs):
    seen.append([(j[2], j[0].numel() // j[2]) for j in jobs])
    return orig(cls, jobs)
M._Deferred._launch = classmethod(launch)
torch.cuda.set_device(0)
bench.run_training_steps(model, img, batch, 8, "bf16", 2, 1, 0, 1, torch.device("cuda", 0))
per_step = len(seen) // 2 if len(seen) > 3 else len(seen)
tot = 0
for k, jobs in enumerate(seen[:per_step]):
    b = sum(s * n * 4 for s, n in jobs)
    tot += b
    from collections import Counter
    c = Counter(jobs)
    print(f"launch {k}: {len(jobs)} jobs, {b / 1e6:.1f} MB read, {sum(n for _, n in jobs) * 4 / 1e6:.1f} MB written")
    for (s, n), cnt in sorted(c.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[1]):
        print(f"    {cnt:3d} x  S={s:5d}  n={n:8d}  ({cnt * s * n * 4 / 1e6:7.2f} MB)")
print(f"total {tot / 1e6:.1f} MB per step over {per_step} launches (of {len(seen)} seen)")
