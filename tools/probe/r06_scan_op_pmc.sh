#!/bin/bash
# round 6 (verdict item 6): a --pmc traffic row per shape for the reference-layout op selective_scan_fn, forward and backward
# kernels: three passes over tools/probe/r06_scan_op_shape.py per configuration, folded by tools/pmc_summary.py
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in cfg2 cfg3 cfg4 cfg5; do
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/o1 -- python3 $R/tools/probe/r06_scan_op_shape.py $cfg > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/o2 -- python3 $R/tools/probe/r06_scan_op_shape.py $cfg > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/o3 -- python3 $R/tools/probe/r06_scan_op_shape.py $cfg > $OUT/r06_scan_op_${cfg}_bytes.txt 2>/dev/null
  python3 $R/tools/pmc_summary.py $(find $OUT/o1 -name "*counter_collection.csv" | head -1) $(find $OUT/o2 -name "*counter_collection.csv" | head -1) $(find $OUT/o3 -name "*counter_collection.csv" | head -1) $OUT/r06_scan_op_${cfg}_pmc.json
  rm -rf $OUT/o1 $OUT/o2 $OUT/o3
done
for cfg in cfg2 cfg3 cfg4 cfg5; do cat $OUT/r06_scan_op_${cfg}_bytes.txt; python3 -c "
import json
d=json.load(open('$OUT/r06_scan_op_${cfg}_pmc.json'))['kernels']
for k,v in d.items(): print(' ', k, v.get('traffic_bytes'), v.get('SQ_INSTS_VALU_per_wave'), v.get('SQ_ACTIVE_INST_VALU_frac'), v.get('SQ_WAIT_ANY_frac'))
"; done
