mkdir -p gpurun_out/r06_j
python -m pytest tests/test_scan_gpu.py tests/test_ops_gpu.py -x -q 2>&1 | tail -4 > gpurun_out/r06_j/t.log; cat gpurun_out/r06_j/t.log
export PROBE_LIB=$GRAFT_REPO_ROOT/ab/tuning.so
for v in 0 1 0 1; do echo "FASTVIM_SCAN_SHORT_TAB=$v"; FASTVIM_SCAN_SHORT_TAB=$v python tools/probe/scan_op_time.py 2>/dev/null; done > gpurun_out/r06_j/scan_op_ab.log 2>&1
cat gpurun_out/r06_j/scan_op_ab.log
