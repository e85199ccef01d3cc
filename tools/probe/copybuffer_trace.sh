#!/bin/bash
# which host ops issue the __amd_rocclr_copyBuffer nodes of the replayed FastVim-T step (27 per step)?
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace_seq -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernels --no-other-configs --no-scan-op > /dev/null 2> $OUT/trace_seq.err
DB=$(find $OUT/trace_seq -name "*.db" | head -1)
python3 $R/tools/step_sequence.py $DB > $OUT/r03_step_sequence.txt 2>&1
rm -rf $OUT/trace_seq
head -70 $OUT/r03_step_sequence.txt
