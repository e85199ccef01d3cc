mkdir -p gpurun_out/r06_t
python -m pytest tests/test_mixer_gpu.py tests/test_config34_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
  for v in new5 f4; do
    echo -n "$v T: "; PROBE_LIB=$GRAFT_REPO_ROOT/ab/$v.so python tools/probe/bench_ms.py --steps 30 --warmup 5 2>/dev/null | tail -1
  done
done | tee gpurun_out/r06_t/ab_fwd_short_lds_prefetch.log
for i in 1 2; do
  for v in new5 f4 f6; do
    echo -n "$v B224: "; PROBE_LIB=$GRAFT_REPO_ROOT/ab/$v.so python tools/probe/ab_step.py B 224 128 6 2>/dev/null | tail -1
  done
done | tee gpurun_out/r06_t/ab_scan_cl_fwd_groups.log
