mkdir -p gpurun_out/r06_n
python -m pytest tests/test_scan_gpu.py tests/test_ops_gpu.py tests/test_xproj_fold_gpu.py tests/test_mixer_gpu.py -x -q 2>&1 | tail -4 > gpurun_out/r06_n/t.log; cat gpurun_out/r06_n/t.log
export PROBE_LIB=$GRAFT_REPO_ROOT/ab/tuning.so
for v in 0 1 0 1; do echo "FASTVIM_SCAN_SHORT_BWD_REG=$v"; FASTVIM_SCAN_SHORT_BWD_REG=$v python tools/probe/scan_op_time.py 2>/dev/null | head -2; done > gpurun_out/r06_n/scan_op_bwd_ab.log 2>&1
cat gpurun_out/r06_n/scan_op_bwd_ab.log
unset PROBE_LIB
bash tools/probe/r06_run13.sh
