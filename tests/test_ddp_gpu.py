"""The multi-rank benchmark path on a one-GPU box: two ranks share GPU 0 and exchange gradients through gloo
(FASTVIM_BENCH_ONE_GPU=1), so graph replay -> flat all-reduce -> fused optimizer runs end to end exactly as the
driver launches it for N > 1 (there RCCL over xGMI).  The sharding / all-reduce arithmetic itself is covered on
CPU by tests/test_ddp_cpu.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("kernel_rows", [False, True])
def test_two_rank_bench_line(kernel_rows):
    """kernel_rows: ``--kernels`` -- rank 0 times its kernel table (graph captures of its own) while the process group is
    alive and the other rank waits in the closing barrier.  Without the flag an N > 1 line carries no kernel table (round
    5: the scaling run stays short and no rank idles beside a busy one)."""
    env = dict(os.environ, FASTVIM_BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29538" if kernel_rows else "29537", os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--steps", "3", "--warmup", "1", "--batch", "16", "--no-cpu-baseline", "--no-scan-op"]
    if kernel_rows:
        cmd.append("--kernels")
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                 # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 32 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and out["config"]["final_loss"] == out["config"]["final_loss"]      # finite
    assert ("kernels" in out) == kernel_rows
    if kernel_rows:
        assert out["roofline"]["frac"] > 0 and out["roofline"]["elementwise_floor_same_size_cold"]["add_2r1w"]["us"] > 0
