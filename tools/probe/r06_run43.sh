mkdir -p gpurun_out/r06_t
python -m pytest tests/test_xproj_bwd_gpu.py tests/test_channel_gpu.py tests/test_config34_gpu.py -m gpu -x -q 2>&1 | tail -5
for i in 1 2 3; do
  for f in "--no-xproj-two-addends" ""; do
    echo -n "C [$f]: "; python tools/probe/bench_ms.py --model C --batch 64 --steps 6 --warmup 2 $f 2>/dev/null | tail -1
  done
done | tee gpurun_out/r06_t/ab_two_addends_chan.log
for i in 1 2; do
  for f in "--no-xproj-two-addends" ""; do
    echo -n "B2048 [$f]: "; python tools/probe/bench_ms.py --model B --batch 8 --img 2048 --steps 4 --warmup 2 $f 2>/dev/null | tail -1
  done
done | tee -a gpurun_out/r06_t/ab_two_addends_chan.log
