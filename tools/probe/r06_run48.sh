mkdir -p gpurun_out/r06_t
for i in 1 2 3; do
  for ps in 8 16; do
    echo -n "B224 presum=$ps: "; python tools/probe/ab_step.py B 224 128 6 --presum $ps 2>/dev/null | tail -1
  done
done | tee gpurun_out/r06_t/ab_xproj_presum_after_mmb.log
