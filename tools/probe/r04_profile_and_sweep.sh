# end-of-round evidence: kernel trace + PMC passes of the FastVim-T step (profiles/r04_v1_*), and the per-GPU batch sweep
mkdir -p gpurun_out/r04i
bash tools/profile_step.sh r04_v1 > gpurun_out/r04i/profile.log 2>&1; tail -3 gpurun_out/r04i/profile.log
for b in 32 64 128 256 512; do
  python bench.py --batch $b --steps 10 --warmup 3 --no-kernels --no-cpu-baseline --no-other-configs --no-scan-op 2>/dev/null | grep -o '"value": [^,]*, "unit": "images/sec", "n_gpus": 1, "steps": 10, "warmup": 3, "ms_per_step": [^,]*' | sed "s/^/batch $b: /"
done | tee gpurun_out/r04i/batch_sweep.log
