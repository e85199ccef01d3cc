mkdir -p gpurun_out/r05
python -m pytest tests/test_scan_gpu.py tests/test_xproj_fold_gpu.py -m gpu -q -x 2>&1 | tail -2 > gpurun_out/r05/adj.log
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-kernels --no-scan-op 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('T', d['value'], d['ms_per_step'], d['config'].get('final_loss_hex'))" >> gpurun_out/r05/adj.log; done
python -m fastvim_amd.build --tuning > /dev/null
python tools/probe/scan_stamps.py 192 xproj 2>&1 | grep -v amdgpu >> gpurun_out/r05/adj.log
python -m fastvim_amd.build > /dev/null
cat gpurun_out/r05/adj.log
