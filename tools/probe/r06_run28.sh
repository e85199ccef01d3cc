mkdir -p gpurun_out/r06_t
python -m pytest tests/test_config34_gpu.py tests/test_mixer_gpu.py tests/test_vim_gpu.py tests/test_channel_gpu.py tests/test_baselines_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
  for v in new3 new4; do
    for c in cfg5 vim cfg4; do echo -n "$v "; PROBE_LIB=$GRAFT_REPO_ROOT/ab/$v.so python tools/probe/r06_scan_shape.py $c 3 --time 2>/dev/null | tail -1; done
  done
done | tee gpurun_out/r06_t/ab_chunked_scan_kernels2.log
for i in 1 2; do
  for v in new3 new4; do
    for cfg in "--model C --batch 64 --steps 6 --warmup 2" "--model V --batch 128 --steps 8 --warmup 2" "--model B --batch 8 --img 2048 --steps 4 --warmup 2"; do
      echo -n "$v $cfg: "; PROBE_LIB=$GRAFT_REPO_ROOT/ab/$v.so python tools/probe/bench_ms.py $cfg 2>/dev/null | tail -1
    done
  done
done | tee gpurun_out/r06_t/ab_fwd_chunked_raw_prefetch.log
