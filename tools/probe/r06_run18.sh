mkdir -p gpurun_out/r06_r
export PROBE_LIB=$GRAFT_REPO_ROOT/ab/tuning.so
for v in 1 0 1 0; do echo "FASTVIM_SCAN_SHORT_SEG=$v"; FASTVIM_SCAN_SHORT_SEG=$v python tools/probe/scan_op_time.py 2>/dev/null | sed -n 3,3p; done > gpurun_out/r06_r/seg.log 2>&1
cat gpurun_out/r06_r/seg.log
