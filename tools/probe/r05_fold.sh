#!/bin/bash
# round 5: the x_proj adjoint inside the short scan backward -- parity tests, then the FastVim-T step with / without it,
# alternating, same box.  usage (GPU box): bash tools/probe/r05_fold.sh
mkdir -p gpurun_out/r05
python -m pytest tests/test_xproj_fold_gpu.py -x -q 2>&1 | tail -15 | tee gpurun_out/r05/fold_tests.log
for i in 1 2 3; do
  echo -n "fold:    "; python tools/probe/bench_ms.py --steps 40 --warmup 10
  echo -n "no fold: "; python tools/probe/bench_ms.py --steps 40 --warmup 10 --no-xproj-fold
done 2>&1 | tee gpurun_out/r05/ab_xproj_fold.log
