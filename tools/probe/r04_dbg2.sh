mkdir -p gpurun_out/r04b
export FASTVIM_BENCH_ONE_GPU=1 FASTVIM_BENCH_HANG_DUMP=150
A="--gpus 2 --model T --steps 3 --warmup 1 --batch 16 --buckets 3 --no-cpu-baseline --no-kernels --no-scan-op"
echo "== self launch"; timeout 400 python bench.py $A > gpurun_out/r04b/self.out 2> gpurun_out/r04b/self.err; echo rc=$?
echo "== torchrun"; timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 bench.py $A > gpurun_out/r04b/tr.out 2> gpurun_out/r04b/tr.err; echo rc=$?
unset FASTVIM_BENCH_ONE_GPU FASTVIM_BENCH_HANG_DUMP
echo "== mid tests"; timeout 900 python -m pytest tests/test_mixer_mid_gpu.py -x -q > gpurun_out/r04b/mid.log 2>&1; echo rc=$?; tail -15 gpurun_out/r04b/mid.log
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 --no-other-configs --no-scan-op --no-cpu-baseline > gpurun_out/r04b/bench.json 2> gpurun_out/r04b/bench.err; echo rc=$?
tail -c 1500 gpurun_out/r04b/self.err; echo; tail -c 600 gpurun_out/r04b/tr.err; head -c 600 gpurun_out/r04b/bench.json
