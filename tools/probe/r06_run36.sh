mkdir -p gpurun_out/r06_t
python -m pytest tests/test_xproj_bwd_gpu.py tests/test_flat_gpu.py tests/test_pipeline_gpu.py tests/test_config34_gpu.py tests/test_mixer_gpu.py tests/test_model_gpu.py tests/test_chain_gpu.py -m gpu -x -q 2>&1 | tail -5
for i in 1 2 3; do
  for f in "--no-xproj-two-addends" ""; do
    echo -n "B224 [$f]: "; python tools/probe/bench_ms.py --model B --batch 128 --steps 6 --warmup 2 $f 2>/dev/null | tail -1
  done
done | tee gpurun_out/r06_t/ab_xproj_two_addends.log
