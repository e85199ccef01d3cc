mkdir -p gpurun_out/r06_t
python -m pytest tests/test_ops_gpu.py tests/test_config34_gpu.py tests/test_channel_gpu.py tests/test_mixer_gpu.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2 3; do
  for m in 1 2; do
    for cfg in "--model B --batch 128 --steps 6 --warmup 2" "--model C --batch 64 --steps 6 --warmup 2"; do
      echo -n "FASTVIM_XPROJ_FWD_MT=$m $cfg: "; FASTVIM_XPROJ_FWD_MT=$m PROBE_LIB=$GRAFT_REPO_ROOT/ab/tuning.so python tools/probe/bench_ms.py $cfg 2>/dev/null | tail -1
    done
  done
done | tee gpurun_out/r06_t/ab_xproj_fwd_two_tiles.log
