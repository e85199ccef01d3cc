#!/bin/bash
# round 6: three PMC passes over the chunked scan kernels at the cfg5 (FastChannelVim-S) and Vim-T mixer shapes
# (tools/probe/r06_scan_shape.py), folded per kernel by tools/pmc_summary.py -> gpurun_out/prof/r06_scan_<cfg>_pmc.json
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in cfg5 vim; do
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/s1 -- python3 $R/tools/probe/r06_scan_shape.py $cfg > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/s2 -- python3 $R/tools/probe/r06_scan_shape.py $cfg > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/s3 -- python3 $R/tools/probe/r06_scan_shape.py $cfg > $OUT/r06_scan_${cfg}_bytes.txt 2>/dev/null
  python3 $R/tools/pmc_summary.py $(find $OUT/s1 -name "*counter_collection.csv" | head -1) $(find $OUT/s2 -name "*counter_collection.csv" | head -1) $(find $OUT/s3 -name "*counter_collection.csv" | head -1) $OUT/r06_scan_${cfg}_pmc.json
  rm -rf $OUT/s1 $OUT/s2 $OUT/s3
done
cat $OUT/r06_scan_cfg5_pmc.json $OUT/r06_scan_vim_pmc.json $OUT/r06_scan_*_bytes.txt
