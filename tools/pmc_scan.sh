#!/bin/bash
# Two rocprofv3 --pmc passes over tools/run_scan_bwd.py (the pooled-scan backward); CSVs land in gpurun_out/pmc_scan/.
# usage (on the GPU box): bash tools/pmc_scan.sh
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_scan
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/tools/run_scan_bwd.py 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/tools/run_scan_bwd.py 3 > /dev/null 2>&1
find $OUT -name "*counter_collection.csv"
