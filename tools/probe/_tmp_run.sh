mkdir -p gpurun_out/r05
out=gpurun_out/r05/ab_fwd_short.log; : > $out
REPS=3 bash tools/ab.sh tools/probe/bench_ms.py --steps 20 --warmup 5 >> $out 2>&1
cat $out
python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r05/gpu_suite_3.log
cat gpurun_out/r05/gpu_suite_3.log
