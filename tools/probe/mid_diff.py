"""Where does the fused forward launch differ from the three launches?  Per output: mismatching elements, max difference,
and which pooling rows / halves they fall in.  usage: python tools/probe/mid_diff.py [B] [rows] [transposed]"""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import torch
from test_mixer_mid_gpu import _inputs, _three, _one, NAMES
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 14
tr = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
t = _inputs(B, rows, 384, 12, seed=1)
ref = _three(t, rows, tr)
for rep in range(3):
    got = _one(t, rows, tr)
    torch.cuda.synchronize()
    for n, a, b in zip(NAMES, got, ref):
        d = (a.float() - b.float())
        bad = d != 0
        msg = f"rep {rep} {n:6s} mismatches {int(bad.sum()):8d} / {bad.numel()}  max |diff| {d.abs().max().item():.3e}"
        if n in ("xc", "yc") and bad.any():
            per_row = bad.sum(dim=(0, 1, 3)).tolist()
            msg += f"  per row {per_row}"
        if n in ("skip", "g") and bad.any():
            msg += f"  per image (first 8) {bad.view(B, -1).sum(1)[:8].tolist()}"
        print(msg, flush=True)
from fastvim_amd import mixer_ops as M
print("error words", M.mixer_mid_errors(), "flags nonzero", [int(f.abs().sum()) for f in M._MID_FLAGS.values()])
