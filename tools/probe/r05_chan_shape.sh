#!/bin/bash
# A/B of the long-row conv + pool kernels' block shapes at cfg4 / cfg5 (tuning build)
mkdir -p gpurun_out/r05
python -m fastvim_amd.build --tuning > /dev/null
out=gpurun_out/r05/chan_shape.log; : > $out
for cfg in cfg5 cfg4; do
  python tools/probe/r05_chan_shape.py $cfg >> $out 2>&1
  for g in 1 2 3 6; do for r in 1 2 4; do
    FASTVIM_BWD_CHAN_GROUPS=$g FASTVIM_BWD_CHAN_RG=$r FASTVIM_FWD_CHAN_GROUPS=$g python tools/probe/r05_chan_shape.py $cfg >> $out 2>&1
  done; done
done
python -m fastvim_amd.build > /dev/null
cat $out
