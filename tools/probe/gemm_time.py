"""One forward / data-gradient GEMM shape, HBM-cold (rotating operand sets), median of event times.
usage: python tools/probe/gemm_time.py nt|nn M N K   (tuning build: FASTVIM_GEMM_* hooks select the kernel)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from fastvim_amd.gemm import gemm_nn, gemm_nt
kind, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
nb = 2 * (M * K + M * N + N * K)
SETS = max(2, min(16, int(1.5e9 // nb)))
sets = [(torch.randn(M, K, device="cuda").bfloat16(), (torch.randn(N, K, device="cuda") if kind == "nt" else torch.randn(K, N, device="cuda")).bfloat16())
        for _ in range(SETS)]
fn = gemm_nt if kind == "nt" else gemm_nn
for a, w in sets: fn(a, w)
torch.cuda.synchronize()
ev = []
for _ in range(6):
    for a, w in sets:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(a, w); e1.record(); ev.append((e0, e1))
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)[len(ev) // 2]
env = {k: v for k, v in os.environ.items() if k.startswith("FASTVIM_")}
print(f"{kind} M={M} N={N} K={K} {env}: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TFLOP/s  ({SETS} operand sets)")
