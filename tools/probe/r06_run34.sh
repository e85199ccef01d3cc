mkdir -p gpurun_out/r06_t
python -m pytest tests/test_vim_gpu.py tests/test_mixer_gpu.py tests/test_masked_gpu.py tests/test_mae_gpu.py tests/test_channel_gpu.py tests/test_baselines_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
  for v in g0 g1; do
    echo -n "$v Vim-T: "; PROBE_LIB=$GRAFT_REPO_ROOT/ab/$v.so python tools/probe/bench_ms.py --model V --batch 128 --steps 8 --warmup 2 2>/dev/null | tail -1
  done
done | tee gpurun_out/r06_t/ab_generic_conv_bwd_pipelined.log
