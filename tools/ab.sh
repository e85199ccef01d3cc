#!/bin/bash
# In-box A/B of two builds of the library: runs "$@" (a script path relative to the repo, plus arguments) once against
# ab/base.so (a copy of the tree under /tmp/ab_base with that library) and once against the in-tree build, alternating
# REPS times.  usage (on the GPU box): bash tools/ab.sh tools/time_rowkernels.py [args]
REPS=${REPS:-2}
B=/tmp/ab_base
rm -rf $B && mkdir -p $B && cp -r fastvim_amd tools oracle profiles bench.py BASELINE.json $B/ 2>/dev/null
cp ab/base.so $B/fastvim_amd/libfastvim_hip.so
for i in $(seq $REPS); do
  echo -n "base: "; (cd $B && python "$@" 2>/dev/null | tail -1)
  echo -n "new:  "; python "$@" 2>/dev/null | tail -1
done
