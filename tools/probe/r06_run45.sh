mkdir -p gpurun_out/r06_t
python -m pytest tests/test_ops_gpu.py tests/test_flat_gpu.py tests/test_gemm_gpu.py tests/test_scan_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
  for j in 96 144 208; do
    echo -n "T max_jobs=$j: "; python tools/probe/ab_step.py T 224 128 30 --max-jobs $j 2>/dev/null | tail -1
  done
done | tee gpurun_out/r06_t/ab_reduce_one_launch.log
