#!/bin/bash
# FastVim-B 224 px bs 128 (config 3): batch elements per workgroup of the short pooled-scan backward, persistent grid sizes
python -m fastvim_amd.build --tuning > /dev/null || exit 1
trap 'python -m fastvim_amd.build > /dev/null' EXIT
r() { echo -n "$1: "; shift; env "$@" 2>/dev/null | tail -1; }
for i in 1 2; do
r "base         " python tools/probe/ab_step.py B 224 128 8
r "SCAN_NBB=2   " FASTVIM_SCAN_NBB=2 python tools/probe/ab_step.py B 224 128 8
r "SCAN_NBB=8   " FASTVIM_SCAN_NBB=8 python tools/probe/ab_step.py B 224 128 8
r "BWD_GRID=512 " FASTVIM_BWD_GRID=512 python tools/probe/ab_step.py B 224 128 8
r "COMBINE_GRID=1024" FASTVIM_COMBINE_GRID=1024 python tools/probe/ab_step.py B 224 128 8
r "NORM_WAVES=8192" FASTVIM_NORM_WAVES=8192 python tools/probe/ab_step.py B 224 128 8
done
