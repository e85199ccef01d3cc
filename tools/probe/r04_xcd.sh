mkdir -p gpurun_out/r04n
python -m pytest tests/test_mixer_gpu.py tests/test_chain_gpu.py tests/test_model_gpu.py tests/test_config34_gpu.py -x -q -k "not 4104 and not 1000" > gpurun_out/r04n/tests.log 2>&1; echo rc=$?; tail -4 gpurun_out/r04n/tests.log
REPS=3 bash tools/ab.sh tools/probe/bench_ms.py --steps 20 --warmup 5 2>&1 | tee gpurun_out/r04n/ab_T.log
REPS=1 bash tools/ab.sh tools/probe/bench_ms.py --steps 6 --warmup 2 --model B 2>&1 | tee gpurun_out/r04n/ab_B.log
REPS=1 bash tools/ab.sh tools/probe/bench_ms.py --steps 6 --warmup 2 --model S 2>&1 | tee gpurun_out/r04n/ab_S.log
