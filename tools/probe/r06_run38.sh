mkdir -p gpurun_out/r06_t
python -m pytest tests/test_xproj_bwd_gpu.py tests/test_flat_gpu.py tests/test_pipeline_gpu.py tests/test_config34_gpu.py tests/test_channel_gpu.py tests/test_model_gpu.py tests/test_chain_gpu.py tests/test_mixer_gpu.py tests/test_ddp_gpu.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do
  for m in 0 1; do
    for cfg in "--model B --batch 128 --steps 6 --warmup 2" "--model C --batch 64 --steps 6 --warmup 2" "--model B --batch 8 --img 2048 --steps 4 --warmup 2"; do
      echo -n "FASTVIM_XPROJ_BWD_MMB=$m $cfg: "; FASTVIM_XPROJ_BWD_MMB=$m PROBE_LIB=$GRAFT_REPO_ROOT/ab/tuning.so python tools/probe/bench_ms.py $cfg 2>/dev/null | tail -1
    done
  done
  echo -n "B224 --no-xproj-two-addends: "; python tools/probe/bench_ms.py --model B --batch 128 --steps 6 --warmup 2 --no-xproj-two-addends 2>/dev/null | tail -1
  echo -n "B224 (two addends): "; python tools/probe/bench_ms.py --model B --batch 128 --steps 6 --warmup 2 2>/dev/null | tail -1
done | tee gpurun_out/r06_t/ab_xproj_bwd_bf16_mm.log
