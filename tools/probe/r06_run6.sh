mkdir -p gpurun_out/r06_f
export PROBE_LIB=$GRAFT_REPO_ROOT/ab/tuning.so
for cfg in "B 224 128 6" "V 224 128 8" "C 224 64 6" "B 2048 8 4"; do
  echo "== $cfg   (FASTVIM_XPROJ_BWD_MM = 0 / 1, alternating)"
  for i in 1 2; do
    FASTVIM_XPROJ_BWD_MM=0 python tools/probe/ab_step.py $cfg 2>/dev/null | tail -1
    FASTVIM_XPROJ_BWD_MM=1 python tools/probe/ab_step.py $cfg 2>/dev/null | tail -1
  done
done > gpurun_out/r06_f/ab_xproj_bwd_mm.log 2>&1
cat gpurun_out/r06_f/ab_xproj_bwd_mm.log
