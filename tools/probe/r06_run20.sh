mkdir -p gpurun_out/r06_s
python -m pytest tests/test_config34_gpu.py tests/test_mixer_gpu.py -x -q 2>&1 | tail -3 > gpurun_out/r06_s/t.log; cat gpurun_out/r06_s/t.log
for i in 1 2; do
  for v in old new; do echo -n "$v: "; PROBE_LIB=$GRAFT_REPO_ROOT/ab/$v.so python tools/probe/ab_step.py B 224 128 6 2>/dev/null | tail -1; done
done > gpurun_out/r06_s/ab_scan_fwd_quad.log 2>&1
cat gpurun_out/r06_s/ab_scan_fwd_quad.log
