#!/bin/bash
# round 5: kernel trace of the replayed FastVim-T step: per-kernel stats + the full dispatch sequence of one step with gaps
TAG=${1:-r05_v1}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels --no-other-configs --no-scan-op > $OUT/${TAG}_bench_under_trace.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB $OUT/${TAG}_graph_step_kernel_stats.csv > $OUT/rocpd.log 2>&1
python3 $R/tools/step_sequence.py $DB all > $OUT/${TAG}_step_sequence.txt 2>&1
rm -rf $OUT/trace
head -3 $OUT/${TAG}_step_sequence.txt; tail -2 $OUT/${TAG}_step_sequence.txt
