mkdir -p gpurun_out/r06_a
python -m pytest tests/test_channel_gpu.py tests/test_config34_gpu.py tests/test_flat_gpu.py tests/test_no_spills.py -x -q 2>&1 | tail -5 > gpurun_out/r06_a/tests.log
AB=1 bash tools/probe/ab_long_scan.sh > gpurun_out/r06_a/ab_long.log 2>&1
cat gpurun_out/r06_a/tests.log gpurun_out/r06_a/ab_long.log
