mkdir -p gpurun_out/r06_c
python -m pytest tests/test_pipeline_gpu.py tests/test_convpool_dgrad_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/r06_c/t.log; cat gpurun_out/r06_c/t.log
python tools/probe/r06_convdgrad_time.py 2>&1 | grep -v amdgpu > gpurun_out/r06_c/time.log; cat gpurun_out/r06_c/time.log
bash tools/probe/r06_convdgrad_phases.sh > /dev/null 2>&1; cp gpurun_out/r06/convdgrad_phases.log gpurun_out/r06_c/; cat gpurun_out/r06_c/convdgrad_phases.log
bash tools/probe/r05_trace.sh r06_v1 > gpurun_out/r06_c/trace.log 2>&1; tail -3 gpurun_out/r06_c/trace.log
head -30 gpurun_out/prof/r06_v1_graph_step_kernel_stats.csv | cut -c1-150
