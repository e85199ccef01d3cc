#!/bin/bash
# round 5, final tree: a few dispatch hooks re-swept on the FastVim-T step (tuning build; one box, two passes)
mkdir -p gpurun_out/r05
python -m fastvim_amd.build --tuning > /dev/null
out=gpurun_out/r05/hook_sweep.log; : > $out
run() { echo -n "$1: " >> $out; env $1 python tools/probe/bench_ms.py --steps 20 --warmup 5 >> $out 2>&1; }
for pass in 1 2; do
  run "X=0"
  for v in 192 224 320 384 512; do run "FASTVIM_BWD_GRID=$v"; done
  for v in 56 48; do run "FASTVIM_FUSED_RPT=$v"; done
  for v in 1 4; do run "FASTVIM_SCAN_NBB=$v"; done
  run "FASTVIM_GEMM_STREAM=0"
  run "FASTVIM_GEMM_N96=0"
done
python -m fastvim_amd.build > /dev/null
cat $out
