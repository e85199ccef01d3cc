mkdir -p gpurun_out/r06_t
python -m pytest tests/test_xproj_bwd_gpu.py tests/test_channel_gpu.py tests/test_config34_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
  for v in y0 y1; do
    echo -n "$v C: "; PROBE_LIB=$GRAFT_REPO_ROOT/ab/$v.so python tools/probe/bench_ms.py --model C --batch 64 --steps 6 --warmup 2 2>/dev/null | tail -1
  done
done | tee gpurun_out/r06_t/ab_xproj_bwd_staging_general.log
