#!/bin/bash
# round 5: combine as the A-tile producer of the out_proj + add + norm launch -- parity tests, then the FastVim-T step
# with / without it, alternating, same box.  usage (GPU box): bash tools/probe/r05_combine.sh
mkdir -p gpurun_out/r05
python -m pytest tests/test_combine_gemm_gpu.py -x -q 2>&1 | tail -15 | tee gpurun_out/r05/combine_tests.log
for i in 1 2 3; do
  echo -n "fused:    "; python tools/probe/bench_ms.py --steps 40 --warmup 10
  echo -n "separate: "; python tools/probe/bench_ms.py --steps 40 --warmup 10 --no-combine-fusion
done 2>&1 | tee gpurun_out/r05/ab_combine_fusion.log
