#!/bin/bash
# configs 4 (FastVim-B 2048 px bs 8) and 5 (FastChannelVim-S/16 bs 64): persistent grid sizes and chunked-scan workgroup size
python -m fastvim_amd.build --tuning > /dev/null || exit 1
trap 'python -m fastvim_amd.build > /dev/null' EXIT
r() { echo -n "$1: "; shift; env "$@" 2>/dev/null | tail -1; }
for cfg in "B 2048 8 4" "C 224 64 6"; do
echo "== $cfg"
for i in 1 2; do
r "base             " python tools/probe/ab_step.py $cfg
r "NORM_WAVES=4096  " FASTVIM_NORM_WAVES=4096 python tools/probe/ab_step.py $cfg
r "NORM_WAVES=16384 " FASTVIM_NORM_WAVES=16384 python tools/probe/ab_step.py $cfg
r "BWD_GRID=512     " FASTVIM_BWD_GRID=512 python tools/probe/ab_step.py $cfg
r "COMBINE_GRID=1024" FASTVIM_COMBINE_GRID=1024 python tools/probe/ab_step.py $cfg
r "COMBINE_GRID=2048" FASTVIM_COMBINE_GRID=2048 python tools/probe/ab_step.py $cfg
r "SCAN_CK_WAVES=12 " FASTVIM_SCAN_CK_WAVES=12 python tools/probe/ab_step.py $cfg
r "SCAN_CK_WAVES=4  " FASTVIM_SCAN_CK_WAVES=4 python tools/probe/ab_step.py $cfg
done
done
