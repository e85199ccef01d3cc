mkdir -p gpurun_out/r05
out=gpurun_out/r05/ab_scan_step.log; : > $out
python -m pytest tests/test_scan_gpu.py tests/test_channel_gpu.py tests/test_xproj_fold_gpu.py -m gpu -q -x 2>&1 | tail -2 >> $out
for cfg in "--steps 20 --warmup 5" "--model B --steps 10 --warmup 3" "--model C --batch 64 --steps 8 --warmup 3" "--model B --batch 8 --img 2048 --steps 5 --warmup 2" "--model S --steps 10 --warmup 3" "--img 256 --steps 20 --warmup 5"; do
  echo "## bench.py $cfg" >> $out
  REPS=2 bash tools/ab.sh tools/probe/bench_ms.py $cfg >> $out 2>&1
done
cat $out
