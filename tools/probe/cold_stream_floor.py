"""What a plain elementwise kernel gets on HBM-cold operands of the FastVim-T tensor sizes (the floor the row kernels are
priced against): c = a + b over bf16 tensors of U = 19.3 MB (2 reads + 1 write = combine_fwd's 3U), a 5-stream form
(combine_bwd's 5U) and a copy (2U), operand sets rotated past the Infinity Cache exactly as bench.py's kernel table does."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from bench import time_kernel, rotating
dev = "cuda"
n = 128 * 196 * 384
U = n * 2
def mk(k):
    return {f"t{i}": torch.randn(n, device=dev).bfloat16() for i in range(k)}
for name, k, fn, streams in (
    ("copy      (1 read, 1 write)", 2, lambda s: s["t1"].copy_(s["t0"]), 2),
    ("add       (2 reads, 1 write)", 3, lambda s: torch.add(s["t0"], s["t1"], out=s["t2"]), 3),
    ("addcmul x2 (3 reads, 2 writes)", 5, lambda s: (torch.addcmul(s["t0"], s["t1"], s["t2"], out=s["t3"]), s["t4"].copy_(s["t0"])), 6),
):
    base = mk(k)
    fns = rotating(fn, base, tuple(base.keys()), U * k)
    t, tw = time_kernel(fns), time_kernel(fns[0])
    print(f"{name:32s} cold {t*1e6:6.1f} us = {streams*U/t/1e12:4.2f} TB/s   warm {tw*1e6:6.1f} us = {streams*U/tw/1e12:4.2f} TB/s  ({len(fns)} sets)")
