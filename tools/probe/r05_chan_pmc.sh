#!/bin/bash
# round 5: PMC passes over the cell-walking conv + pool kernels at the cfg4 / cfg5 shapes (tools/probe/r05_chan_shape.py),
# folded per kernel by tools/pmc_summary.py -> gpurun_out/prof/r05_chan_<cfg>_pmc.json
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in cfg4 cfg5; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/c1 -- python3 $R/tools/probe/r05_chan_shape.py $cfg > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c2 -- python3 $R/tools/probe/r05_chan_shape.py $cfg > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/c3 -- python3 $R/tools/probe/r05_chan_shape.py $cfg > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $(find $OUT/c1 -name "*counter_collection.csv" | head -1) $(find $OUT/c2 -name "*counter_collection.csv" | head -1) $(find $OUT/c3 -name "*counter_collection.csv" | head -1) $OUT/r05_chan_${cfg}_pmc.json
  rm -rf $OUT/c1 $OUT/c2 $OUT/c3
done
cat $OUT/r05_chan_cfg4_pmc.json $OUT/r05_chan_cfg5_pmc.json
