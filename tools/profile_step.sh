#!/bin/bash
# End-of-round evidence run on the GPU box: kernel trace of the graph-replayed training step (folded to a CSV)
# and the three PMC passes over the hand-written kernels.  Outputs under gpurun_out/prof/.
# usage: bash tools/profile_step.sh <tag> [trace-only] [extra bench.py args ...]      (e.g. r02_v1; r02_vim trace-only --model V)
TAG=${1:-run}
shift
TRACE_ONLY=0
if [ "$1" = "trace-only" ]; then TRACE_ONLY=1; shift; fi
EXTRA="$@"
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels --no-other-configs $EXTRA > $OUT/${TAG}_bench_under_trace.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB $OUT/${TAG}_graph_step_kernel_stats.csv > $OUT/rocpd.log 2>&1
if [ $TRACE_ONLY = 1 ]; then rm -rf $OUT/trace; ls -la $OUT; exit 0; fi
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -- python3 $R/tools/run_kernels.py 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2 -- python3 $R/tools/run_kernels.py 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc3 -- python3 $R/tools/run_kernels.py 3 > /dev/null 2>&1
for i in 1 2 3; do cp $(find $OUT/pmc$i -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc${i}_counter_collection.csv; done
python3 $R/tools/pmc_summary.py $OUT/${TAG}_pmc1_counter_collection.csv $OUT/${TAG}_pmc2_counter_collection.csv $OUT/${TAG}_pmc3_counter_collection.csv $OUT/${TAG}_pmc_traffic.json
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3
ls -la $OUT
