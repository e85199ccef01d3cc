#!/bin/bash
# rocprofv3 --pmc passes over tools/run_kernels_b.py (phased GEMM, chunked scan); CSVs land in gpurun_out/pmc_b.
# usage (on the GPU box): bash tools/pmc_b.sh <tag>
TAG=${1:-r02_v8}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_b
mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/tools/run_kernels_b.py 3 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/tools/run_kernels_b.py 3 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/c -- python3 $GRAFT_REPO_ROOT/tools/run_kernels_b.py 3 > /dev/null 2>&1
for d in a b c; do f=$(find $OUT/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_b_pmc_$d.csv; done
rm -rf $OUT/a $OUT/b $OUT/c
ls -la $OUT
