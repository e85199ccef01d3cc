mkdir -p gpurun_out/r06_o
python -m pytest tests/test_scan_gpu.py tests/test_ops_gpu.py -x -q 2>&1 | tail -3 > gpurun_out/r06_o/t.log; cat gpurun_out/r06_o/t.log
for i in 1 2; do
  echo "round-5 library (ab/base.so)"; PROBE_LIB=$GRAFT_REPO_ROOT/ab/base.so python tools/probe/scan_op_time.py 2>/dev/null
  echo "this tree"; python tools/probe/scan_op_time.py 2>/dev/null
done > gpurun_out/r06_o/scan_op_ab.log 2>&1
cat gpurun_out/r06_o/scan_op_ab.log
