mkdir -p gpurun_out/r06_t
python -m pytest tests/test_ops_gpu.py tests/test_gemm_gpu.py tests/test_flat_gpu.py -m gpu -x -q 2>&1 | tail -3
python tools/probe/r06_reduce_jobs.py 2>&1 | grep -v amdgpu | tail -40 > gpurun_out/r06_t/reduce_jobs.log
for i in 1 2; do
echo "round-5 library (ab/base.so)"; PROBE_LIB=ab/base.so python tools/probe/r06_reduce_time.py 2>&1 | grep -v amdgpu
echo "this tree"; python tools/probe/r06_reduce_time.py 2>&1 | grep -v amdgpu
done | tee gpurun_out/r06_t/ab_reduce_time.log
for i in 1 2; do
PROBE_LIB=ab/new.so python tools/probe/bench_ms.py --steps 30 --warmup 5
python tools/probe/bench_ms.py --steps 30 --warmup 5
done | tee gpurun_out/r06_t/ab_reduce_step.log
