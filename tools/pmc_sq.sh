#!/bin/bash
# Issue-side rocprofv3 --pmc passes over tools/run_kernels.py (every fused-mixer kernel at the benchmark shape);
# CSVs land in gpurun_out/pmc_sq/{a,b}.  usage (on the GPU box): bash tools/pmc_sq.sh
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/tools/run_kernels.py 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/tools/run_kernels.py 3 > /dev/null 2>&1
find $OUT -name "*counter_collection.csv"
