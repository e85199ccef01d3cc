mkdir -p gpurun_out/r06_e
python -m pytest tests/test_xproj_bwd_gpu.py tests/test_xproj_fold_gpu.py -x -q 2>&1 | tail -6 > gpurun_out/r06_e/t.log; cat gpurun_out/r06_e/t.log
python -m pytest tests/test_mixer_gpu.py tests/test_config34_gpu.py tests/test_vim_gpu.py -x -q 2>&1 | tail -6 > gpurun_out/r06_e/t2.log; cat gpurun_out/r06_e/t2.log
# A/B against the round-5 base library (ab/base.so) and the presum threshold with the new kernel
for cfg in "B 224 128 6" "V 224 128 8" "C 224 64 6"; do
  echo "== $cfg"
  python tools/probe/ab_step.py $cfg 2>/dev/null | tail -1
  python tools/probe/ab_step.py $cfg --presum 1000 2>/dev/null | tail -1
  python tools/probe/ab_step.py $cfg 2>/dev/null | tail -1
  python tools/probe/ab_step.py $cfg --presum 1000 2>/dev/null | tail -1
done > gpurun_out/r06_e/presum.log 2>&1
cat gpurun_out/r06_e/presum.log
