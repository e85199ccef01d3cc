mkdir -p gpurun_out/r06_t
python -m pytest tests/test_config34_gpu.py tests/test_mixer_gpu.py tests/test_channel_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
  for v in new5 c1 c2; do
    for cfg in "--model C --batch 64 --steps 6 --warmup 2" "--model B --batch 8 --img 2048 --steps 4 --warmup 2"; do
      echo -n "$v $cfg: "; PROBE_LIB=$GRAFT_REPO_ROOT/ab/$v.so python tools/probe/bench_ms.py $cfg 2>/dev/null | tail -1
    done
  done
done | tee gpurun_out/r06_t/ab_chan_conv_alternating_buffers.log
