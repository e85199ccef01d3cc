mkdir -p gpurun_out/r06_d
python -m pytest tests/test_convpool_dgrad_gpu.py tests/test_chain_gpu.py -x -q 2>&1 | tail -4 > gpurun_out/r06_d/t.log; cat gpurun_out/r06_d/t.log
STEP=1 bash tools/probe/r06_convdgrad_variants.sh "-DCD_PDEPTH=6" "-DCD_W2_PIPE=0" "-DCD_PDEPTH=6 -DCD_W2_PIPE=0" > /dev/null 2>&1
cp gpurun_out/r06/convdgrad_variants.log gpurun_out/r06_d/; cat gpurun_out/r06_d/convdgrad_variants.log
