#!/bin/bash
# weight gradients as throttled background launches during backward (FlatTrainingState(wgrad_overlap=(chunk, workgroups)))
# against the launch after the last block: FastVim-T headline configuration (and FastVim-B with B=1)
r() { echo -n "$1: "; shift; "$@" 2>/dev/null | tail -1; }
M=${M:-T}; ST=${ST:-40}
for i in 1 2; do
r "end of backward      " python tools/probe/ab_step.py $M 224 128 $ST
r "bg chunk 16 wgs 32   " python tools/probe/ab_step.py $M 224 128 $ST --bg 16 32
r "bg chunk 16 wgs 64   " python tools/probe/ab_step.py $M 224 128 $ST --bg 16 64
r "bg chunk 32 wgs 48   " python tools/probe/ab_step.py $M 224 128 $ST --bg 32 48
r "bg chunk 8  wgs 24   " python tools/probe/ab_step.py $M 224 128 $ST --bg 8 24
r "bg chunk 16 wgs 128  " python tools/probe/ab_step.py $M 224 128 $ST --bg 16 128
r "bg chunk 16 full grid" python tools/probe/ab_step.py $M 224 128 $ST --bg 16 0
done
