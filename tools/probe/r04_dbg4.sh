mkdir -p gpurun_out/r04e
python tools/probe/mid_diff.py 128 14 0 2>&1 | grep -v amdgpu | grep "rep 0\|error" > gpurun_out/r04e/diff.log; python tools/probe/mid_diff.py 5 16 1 2>&1 | grep "rep 0\|error" >> gpurun_out/r04e/diff.log
cat gpurun_out/r04e/diff.log
timeout 900 python -m pytest tests/test_mixer_mid_gpu.py -q > gpurun_out/r04e/mid.log 2>&1; echo rc=$?; tail -8 gpurun_out/r04e/mid.log
for i in 1 2; do
echo "== bench fused"; timeout 600 python bench.py --steps 20 --warmup 5 --no-other-configs --no-scan-op --no-cpu-baseline --no-kernels > gpurun_out/r04e/bench_fused$i.json 2> gpurun_out/r04e/bench.err; echo rc=$?
echo "== bench three"; timeout 600 python bench.py --steps 20 --warmup 5 --no-other-configs --no-scan-op --no-cpu-baseline --no-kernels --no-mid-fusion > gpurun_out/r04e/bench_three$i.json 2>> gpurun_out/r04e/bench.err; echo rc=$?
done
for f in bench_fused1 bench_three1 bench_fused2 bench_three2; do grep -o '"ms_per_step": [^,]*' gpurun_out/r04e/$f.json; grep -o '"final_loss_hex": "[^"]*"' gpurun_out/r04e/$f.json; done
python -m fastvim_amd.build --tuning > gpurun_out/r04e/build.log 2>&1; tail -1 gpurun_out/r04e/build.log
python tools/probe/mid_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04e/mid_stamps.log
