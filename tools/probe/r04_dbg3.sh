mkdir -p gpurun_out/r04c
python tools/probe/mid_diff.py 128 14 0 > gpurun_out/r04c/diff.log 2>&1; python tools/probe/mid_diff.py 5 16 1 >> gpurun_out/r04c/diff.log 2>&1
cat gpurun_out/r04c/diff.log | grep -v amdgpu
timeout 900 python -m pytest tests/test_mixer_mid_gpu.py -q > gpurun_out/r04c/mid.log 2>&1; echo rc=$?; tail -8 gpurun_out/r04c/mid.log
export FASTVIM_BENCH_ONE_GPU=1 FASTVIM_BENCH_HANG_DUMP=200
A="--gpus 2 --model T --steps 3 --warmup 1 --batch 16 --buckets 3 --no-cpu-baseline --no-kernels --no-scan-op"
for i in 1 2; do echo "== self launch $i"; timeout 400 python bench.py $A > gpurun_out/r04c/self$i.out 2> gpurun_out/r04c/self$i.err; echo rc=$?; grep -c final_loss gpurun_out/r04c/self$i.out; done
echo "== self launch, three forward launches"; timeout 400 python bench.py $A --no-mid-fusion > gpurun_out/r04c/self3.out 2> gpurun_out/r04c/self3.err; echo rc=$?
unset FASTVIM_BENCH_ONE_GPU FASTVIM_BENCH_HANG_DUMP
for f in self1 self2 self3; do grep -o '"final_loss": [^,]*' gpurun_out/r04c/$f.out; grep -i "non-finite\|Error" gpurun_out/r04c/$f.err | head -3; done
echo "== bench fused"; timeout 600 python bench.py --steps 20 --warmup 5 --no-other-configs --no-scan-op --no-cpu-baseline --no-kernels > gpurun_out/r04c/bench_fused.json 2> gpurun_out/r04c/bench.err; echo rc=$?
echo "== bench three"; timeout 600 python bench.py --steps 20 --warmup 5 --no-other-configs --no-scan-op --no-cpu-baseline --no-kernels --no-mid-fusion > gpurun_out/r04c/bench_three.json 2>> gpurun_out/r04c/bench.err; echo rc=$?
echo "== bench fused"; timeout 600 python bench.py --steps 20 --warmup 5 --no-other-configs --no-scan-op --no-cpu-baseline --no-kernels > gpurun_out/r04c/bench_fused2.json 2>> gpurun_out/r04c/bench.err; echo rc=$?
for f in bench_fused bench_three bench_fused2; do grep -o '"value": [^,]*, "unit": "images/sec", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": [^,]*' gpurun_out/r04c/$f.json; grep -o '"final_loss_hex": "[^"]*"' gpurun_out/r04c/$f.json; done
