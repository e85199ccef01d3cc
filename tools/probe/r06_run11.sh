mkdir -p gpurun_out/r06_k
STEP=1 bash tools/probe/r06_convdgrad_variants.sh "-DCD_SB_EVERY=2" "-DCD_SB_EVERY=4" "-DCD_SB_EVERY=17" > /dev/null 2>&1
cp gpurun_out/r06/convdgrad_variants.log gpurun_out/r06_k/sb_variants.log; cat gpurun_out/r06_k/sb_variants.log
