"""The segmented training step (fastvim_amd/pipeline.py: forward graph | K backward graphs | optimizer graph, gradient
buckets exchanged between them) against the single-graph step: with one rank it must produce the same parameters bit
for bit; with two ranks sharing the GPU over gloo it must run end to end and report its exchange."""
import copy
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n_seg,graph,batch,fill", [(3, True, 16, True), (6, True, 16, True), (2, False, 16, True),
                                                    (3, True, 128, False), (3, True, 128, True)])
def test_segmented_step_equals_single_step(n_seg, graph, batch, fill, monkeypatch):
    """Bit for bit when every weight gradient is cut into the same K slices in both steps: batch 16 (49 K tiles: one
    slice whatever the group) and batch 128 with ``gemm.FILL`` off.  With it on, the small groups of a segment are cut
    finer (fastvim_amd.gemm.fill_splits) -- same sums in another order: losses and weights agree to rounding."""
    import fastvim_amd.gemm as gemm_mod
    monkeypatch.setattr(gemm_mod, "FILL", fill)
    exact = batch == 16 or not fill
    from fastvim_amd.fastvim import VisionMamba
    from fastvim_amd.flat import FlatAdamW, FlatTrainingState
    from fastvim_amd.losses import SoftTargetCrossEntropy
    from fastvim_amd.pipeline import SegmentedTrainStep

    def make():
        torch.manual_seed(0)
        m = VisionMamba(img_size=224, depth=6, embed_dim=192, num_classes=100, rms_norm=True, residual_in_fp32=True,
                        fused_add_norm=True, final_pool_type="mean", if_abs_pos_embed=True, drop_path_rate=0.1).cuda().train()
        flat = FlatTrainingState(m)
        nd = {n for n, p in m.named_parameters() if p.ndim <= 1 or n.endswith(".bias") or n in m.no_weight_decay()
              or getattr(p, "_no_weight_decay", False)}
        return m, flat, FlatAdamW(flat, m, lr=1e-3, weight_decay=0.05, no_decay=nd, ema_decay=0.999)

    x = torch.randn(batch, 3, 224, 224, device="cuda")
    tgt = torch.softmax(torch.randn(batch, 100, device="cuda"), -1)
    crit = SoftTargetCrossEntropy()
    m1, f1, o1 = make()
    torch.manual_seed(7)
    ref = []
    for _ in range(5):
        f1.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = crit(m1(x), tgt)
        loss.backward()
        o1.step()
        ref.append(loss.item())
    m2, f2, o2 = make()
    torch.manual_seed(7)
    seg = SegmentedTrainStep(m2, f2, o2, crit, x, tgt, n_segments=n_seg, use_graph=graph, warmup=2)
    assert seg.K == n_seg and sorted(sum(([a, b] for a, b in seg.runs), [])) == sorted(
        [0, 6] + 2 * [r[0] for r in seg.runs if r[0] != 0])
    # the warm-up steps of the capture are undone (parameters, optimizer state, EMA, step count, RNG streams are
    # snapshotted and restored): the first replay is step 0 of the trajectory
    got = []
    for _ in range(5):
        got.append(seg.step().item())
    torch.cuda.synchronize()
    assert o2.step_t.item() == 5.0
    if exact:
        assert got == ref, (got, ref)
        assert torch.equal(f1.param_flat, f2.param_flat)
        assert torch.equal(o1.ema, o2.ema)
    else:
        assert got != ref or not torch.equal(f1.param_flat, f2.param_flat)    # the finer cut was taken
        torch.testing.assert_close(torch.tensor(got), torch.tensor(ref), rtol=0, atol=2e-3)
        # AdamW turns a rounding-level difference of a near-zero gradient into a step of up to lr: 5 steps of 1e-3
        assert (f1.param_flat - f2.param_flat).abs().max().item() <= 5.5e-3
        assert (f1.param_flat - f2.param_flat).abs().mean().item() <= 2e-5
    f1.close(); f2.close()


@pytest.mark.parametrize("model,batch,total_mb,wire,extra", [
    ("T", 16, 28.7, ["float32"] * 3, []),                            # FastVim-T: 7.17 M fp32 gradients, 12 / 9 / 7 MB buckets
    ("B", 8, 390.7, ["float32"] * 3, []),                            # FastVim-B: 97.7 M, 163 / 131 / 98 MB; fp32 wire by default
    ("B", 8, 390.7, ["bfloat16", "bfloat16", "float32"], ["--comm-dtype", "auto"]),      # opt-in: bf16 from 100 MB on
    ("C", 4, 102.6, ["float32"] * 3, ["--channels", "8"]),           # FastChannelVim-S/16 (BASELINE configs[4]): 25.6 M
])
def test_two_rank_segmented_bench_line(model, batch, total_mb, wire, extra):
    """PLAIN ``python bench.py --gpus 2`` (no launcher: bench.py starts its own torch.distributed.run child before it touches
    the GPU), two ranks on GPU 0 over gloo (FASTVIM_BENCH_ONE_GPU=1): the N > 1 bench path = segmented step with bucketed,
    asynchronously launched all-reduces, sums scaled by 1 / world inside the optimizer kernel; the line reports the
    exchange (backend, bucket sizes, wire format per bucket, exposed time).  FastVim-T / -B widths and the channel model."""
    env = dict(os.environ, FASTVIM_BENCH_ONE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", model, "--steps", "3", "--warmup", "1",
           "--batch", str(batch), "--buckets", "3", "--no-cpu-baseline", "--no-kernels", "--no-scan-op"] + extra
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ddp"]["overlapped"] is True and out["ddp"]["buckets"] == 3 and out["ddp"]["ranks"] == 2
    assert out["ddp"]["backend"] == "gloo"                                   # "nccl" (= RCCL) without the one-GPU test hook
    assert abs(sum(out["ddp"]["bucket_MB"]) - total_mb) < 0.5
    assert out["ddp"]["wire_dtype"] == wire
    assert out["ddp"]["allreduce_exposed_ms"] is not None and out["config"]["final_loss"] == out["config"]["final_loss"]
    assert out["config"]["global_batch"] == 2 * batch and out["value"] > 0
    ms = out["ddp"]["bucket_allreduce_ms"]                                   # launch -> completion of each bucket's all-reduce
    assert len(ms) == 3 and all(m is not None and m > 0 for m in ms)
    assert out["ddp"]["launcher_env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and out["ddp"]["launcher_env"]["MASTER_ADDR"] == "127.0.0.1"


@pytest.mark.parametrize("buckets", [0, 1, 3, 6])
def test_eight_rank_bench_line_on_one_gpu(buckets):
    """``python bench.py --gpus 8`` as the driver's scaling run issues it, eight ranks on GPU 0 over gloo
    (FASTVIM_BENCH_ONE_GPU=1), batch 2 per rank: everything that depends on the rank COUNT -- eight children on one port,
    per-rank seeds, bucket bounds, rank 0's stdout relay, the barrier-bracketed timing -- and every legal --buckets value
    (0: one all-reduce after a single-graph backward; 1 / 3 / 6: the chain of graphs).  No kernel table under N > 1 by
    default: the line stays short and no rank idles beside a busy one."""
    env = dict(os.environ, FASTVIM_BENCH_ONE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--batch", "2", "--steps", "2", "--warmup", "1",
           "--buckets", str(buckets), "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ddp"]["ranks"] == 8 and out["config"]["global_batch"] == 16 and out["scaling"] == "weak"
    assert out["ddp"]["overlapped"] is (buckets > 0) and out["ddp"]["buckets"] == max(buckets, 1)
    assert "kernels" not in out and "other_configs" not in out
    assert out["config"]["final_loss"] == out["config"]["final_loss"] and out["value"] > 0


def test_self_launch_passes_the_childs_failure_on():
    """``python bench.py --gpus 2`` relays the launcher child's exit code: a failing rank (test hook
    FASTVIM_BENCH_FAIL_RANK) must not turn into rc 0, a JSON line or a hang."""
    env = dict(os.environ, FASTVIM_BENCH_ONE_GPU="1", FASTVIM_BENCH_FAIL_RANK="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0", "--batch", "2", "--no-cpu-baseline", "--no-kernels", "--no-scan-op"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


_RCCL_SCRIPT = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch, torch.distributed as dist
from fastvim_amd.fastvim import VisionMamba
from fastvim_amd.flat import FlatAdamW, FlatTrainingState
from fastvim_amd.losses import SoftTargetCrossEntropy
from fastvim_amd.pipeline import SegmentedTrainStep
from fastvim_amd import ddp
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=0, world_size=1)
# one rank: make the exchange believe there are two, so every bucket really goes through RCCL (a one-rank all-reduce is
# the identity) and the mean halves the gradient
ddp.GradExchange.world_size = property(lambda self: 2)
def run(pretend):
    torch.manual_seed(0)
    m = VisionMamba(img_size=224, depth=6, embed_dim=192, num_classes=100, rms_norm=True, residual_in_fp32=True,
                    fused_add_norm=True, final_pool_type="mean", if_abs_pos_embed=True).cuda().train()
    flat = FlatTrainingState(m)
    opt = FlatAdamW(flat, m, lr=1e-3, weight_decay=0.0)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(16, 3, 224, 224, device="cuda", generator=g)
    tgt = torch.softmax(torch.randn(16, 100, device="cuda", generator=g), -1)
    seg = SegmentedTrainStep(m, flat, opt, SoftTargetCrossEntropy(), x, tgt, n_segments=3, use_graph=True, warmup=1)
    if not pretend:
        seg.exchange.launch = lambda k: None
        seg.exchange.finish = lambda mean=True: None      # (the 1 / 2 is the optimizer kernel's gradient scale either way)
    losses = [seg.step(time_exposed=pretend).item() for _ in range(4)]
    torch.cuda.synchronize()
    gsum = flat.grad_flat.double().abs().sum().item()
    exp = seg.exposed_ms() if pretend else None
    pend = len(seg.exchange._pending)
    flat.close()
    return losses, gsum, exp, pend
a = run(True)
b = run(False)
dist.destroy_process_group()
print(json.dumps({"rccl": a, "plain": b}))
'''


def test_rccl_buckets_between_backward_graphs(tmp_path):
    """The exchange of the segmented step over the real RCCL backend (one rank; the exchange is told there are two so
    that every bucket is launched): asynchronous all-reduces on the process group's stream between the replays of the
    backward graphs, joined before the optimizer graph.  A one-rank all-reduce is the identity, so losses and gradients
    must equal, bit for bit, those of the same step with the exchange replaced by the bare division."""
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(_RCCL_SCRIPT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script), ROOT, "29547"], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    (la, ga, exp, pend), (lb, gb, _, _) = out["rccl"], out["plain"]
    assert la == lb and ga == gb and all(v == v for v in la), out
    assert pend == 0 and exp is not None and exp >= 0.0


def test_graph_capture_beside_a_busy_process_group_watchdog():
    """A live process group's watchdog THREAD polls the events of finished collectives; under torch's default "global"
    capture mode an event query from any thread while another captures is an error that takes the process down
    (tools/probe/capture_vs_watchdog.py global: hipErrorStreamCaptureInvalidated / abort -- what intermittently killed
    ``test_rccl_buckets_between_backward_graphs`` in round 3).  ``SegmentedTrainStep`` and ``bench.py --gpus N`` capture in
    "thread_local" mode when a process group is alive: 40 captures, each right behind a burst of 20 async all-reduces."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe", "capture_vs_watchdog.py"), "thread_local"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "40 captures beside a busy watchdog, no error" in r.stdout and "y[0] = 8000.0" in r.stdout
