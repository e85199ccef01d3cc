mkdir -p gpurun_out/r04k
python -m pytest tests/test_chain_gpu.py -q > gpurun_out/r04k/chain.log 2>&1; echo rc=$?; tail -6 gpurun_out/r04k/chain.log
python tools/probe/addnorm_rw_time.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04k/rw_time.log
for i in 1 2; do
for f in "" "--no-rw"; do echo -n "bench $f: "; python bench.py --steps 20 --warmup 5 --no-other-configs --no-scan-op --no-cpu-baseline --no-kernels $f 2>/dev/null | grep -o '"ms_per_step": [^,]*\|"final_loss_hex": "[^"]*"' | tr '\n' ' '; echo; done
done | tee gpurun_out/r04k/ab.log
