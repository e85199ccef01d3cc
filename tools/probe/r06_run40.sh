mkdir -p gpurun_out/r06_t
python -m pytest tests/test_mixer_gpu.py tests/test_config34_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
  for v in s0 s1; do
    echo -n "$v B224: "; PROBE_LIB=$GRAFT_REPO_ROOT/ab/$v.so python tools/probe/bench_ms.py --model B --batch 128 --steps 6 --warmup 2 2>/dev/null | tail -1
  done
done | tee gpurun_out/r06_t/ab_scan_fwd_staging.log
