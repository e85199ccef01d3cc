mkdir -p gpurun_out/r06_b
python -m pytest tests/test_convpool_dgrad_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r06_b/t1.log
cat gpurun_out/r06_b/t1.log
python -m pytest tests/test_chain_gpu.py tests/test_model_gpu.py tests/test_flat_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r06_b/t2.log
cat gpurun_out/r06_b/t2.log
for i in 1 2; do
python tools/probe/bench_ms.py --steps 20 --warmup 5
python tools/probe/bench_ms.py --steps 20 --warmup 5 --no-conv-dgrad
done > gpurun_out/r06_b/ab.log 2>&1
cat gpurun_out/r06_b/ab.log
