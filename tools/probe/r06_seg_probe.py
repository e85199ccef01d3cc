"""Serial (quad-sharing) vs time-segmented op forward on few rows: where does the segmented kernel still pay?
usage (tuning build): FASTVIM_SCAN_SHORT_SEG=0/1 python tools/probe/r06_seg_probe.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
import bench
from fastvim_amd.selective_scan_interface import selective_scan_fn
for (B, D, L) in ((1, 1536, 128), (2, 1536, 128), (4, 1536, 128), (8, 1536, 128), (16, 768, 112), (4, 768, 64), (8, 384, 128)):
    g = torch.Generator().manual_seed(0)
    q = [t.cuda().bfloat16() for t in (torch.randn(B, D, L, generator=g), 0.5 * torch.rand(B, D, L, generator=g),
                                       torch.randn(B, 16, L, generator=g), torch.randn(B, 16, L, generator=g))]
    A, db = (-0.5 * torch.rand(D, 16, generator=g)).cuda(), (0.5 * torch.rand(D, generator=g)).cuda()
    with torch.no_grad():
        t = bench.time_kernel(lambda: selective_scan_fn(q[0], q[1], A, q[2], q[3], None, None, db, True), iters=20)
    print(f"({B}, {D}, {L}): {t * 1e6:.1f} us  blocks {B * D // 64}", flush=True)
