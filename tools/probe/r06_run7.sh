mkdir -p gpurun_out/r06_g
python -m pytest tests/test_xproj_bwd_gpu.py tests/test_xproj_fold_gpu.py -x -q 2>&1 | tail -4 > gpurun_out/r06_g/t.log; cat gpurun_out/r06_g/t.log
bash tools/probe/r06_run6.sh
cp gpurun_out/r06_f/ab_xproj_bwd_mm.log gpurun_out/r06_g/
