"""Where a workgroup of the fused forward launch (csrc/mixer_mid_fwd.hip) spends its life (tuning build: MID_STAMP), next to
the three launches it replaces, HBM-cold (launches rotate through operand sets; the stamps of the LAST launch are read).
usage: python -m fastvim_amd.build --tuning && python tools/probe/mid_stamps.py"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from fastvim_amd import _lib as L, mixer_ops as M
from test_mixer_mid_gpu import _inputs, _three, _one
B, rows, SETS = 128, 14, 10
sets = [_inputs(B, rows, 384, 12, seed=k) for k in range(SETS)]
lib = L.lib()
stamps = torch.zeros(2 * B, 16, device="cuda", dtype=torch.int64)
def timed(fn, tr):
    for s in sets: fn(s, rows, tr)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(SETS + 1)]
    e[0].record()
    for k, s in enumerate(sets):
        fn(s, rows, tr); e[k + 1].record()
    torch.cuda.synchronize()
    return sorted(e[k].elapsed_time(e[k + 1]) * 1e3 for k in range(SETS))[SETS // 2]
for tr in (False, True):
    t3 = timed(_three, tr)
    lib._cdll.fv_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
    t1 = timed(_one, tr)
    lib._cdll.fv_debug_set_stamps(ctypes.c_void_p(0))
    t = stamps.cpu().double()
    names = ["P1 conv + pool + skip (2 rounds of 4 rows)", "X1 publish xc, wait for the partner", "stage partner rows + barrier",
             "x_proj MFMA (K split over 12 waves) + barrier", "x_dbl: 12-wave sum, bf16, store + barrier", "scan chunk 0 (dt_proj, softplus, 14 steps, y)",
             "scan chunk 1", "X2 publish yc, wait for the partner", "partner's yc rows + barrier", "P3 combine (49 token pairs over 12 waves)"]
    life = t[:, 10] - t[:, 0]
    tpu = 2100.0      # s_memtime counts shader cycles (one counter per XCD: only differences inside a workgroup mean anything); ~2.1 GHz
    print(f"transposed={tr}: three launches {t3:.1f} us, fused launch {t1:.1f} us (events, median of {SETS}, cold operands, eager: "
          f"host launch gaps included); workgroup life median {life.median():.0f} ticks = {life.median() / tpu:.1f} us at 2.1 GHz")
    for k, nm in enumerate(names):
        d = t[:, k + 1] - t[:, k]
        print(f"  {nm:52s} median {d.median() / tpu:6.2f} us   p90 {d.quantile(0.9) / tpu:6.2f}   max {d.max() / tpu:6.2f}   {100 * d.median() / life.median():5.1f} %")
    print(f"  half 0 vs half 1 life {life[0::2].median() / tpu:.2f} / {life[1::2].median() / tpu:.2f} us")
