"""One configuration's step time with Python-side knobs set from the command line (C-side knobs: FASTVIM_* environment
variables of a tuning build).  usage: python tools/probe/ab_step.py MODEL IMG BATCH STEPS [--presum N] [--no-direct-acc]"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
import bench
from fastvim_amd import mixer_ops
a = sys.argv[1:]
model, img, batch, steps = a[0], int(a[1]), int(a[2]), int(a[3])
if "--no-direct-acc" in a:
    from fastvim_amd import gemm as _g
    _g.DIRECT_ACC = False
if "--wgrad-splits" in a:       # --wgrad-splits IN,OUT: split-K factors of the in_proj / out_proj weight gradients (uneven slices allowed)
    from fastvim_amd import gemm as _g2
    _si, _so = [int(v) for v in a[a.index("--wgrad-splits") + 1].split(",")]
    _orig = _g2.grouped_splits
    def _gs(Kd, target=None, M=None, N=None):
        if M is not None and N is not None and M * N >= 192 * 192 and Kd >= 4096:
            return _si if M > N else _so
        return _orig(Kd, target, M, N)
    _g2.grouped_splits = _gs
    import fastvim_amd.mamba_simple_faster as _msf
if "--max-jobs" in a:           # jobs per deferred-reduction launch (96 until round 6, 144 since)
    mixer_ops._Deferred.max_jobs = int(a[a.index("--max-jobs") + 1])
if "--presum" in a:
    mixer_ops._XPROJ_PRESUM = int(a[a.index("--presum") + 1])
torch.cuda.set_device(0)
el, lv, ex = bench.run_training_steps(model, img, batch, 8, "bf16", steps, 3, 0, 1, torch.device("cuda", 0))
print(json.dumps({"ms_per_step": round(el / steps * 1e3, 3), "loss": round(lv, 4)}))
