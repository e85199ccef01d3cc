#!/bin/bash
# Step-time sweep over the dispatchers' A/B hooks.  The hooks exist only in a TUNING build (-DFASTVIM_TUNING_HOOKS): the
# shipped library reads no environment variable and every row below would be the same configuration, so the tuning build
# is made here first (and the default build restored at the end).
python -m fastvim_amd.build --tuning > /dev/null || exit 1
trap 'python -m fastvim_amd.build > /dev/null' EXIT
run() { python bench.py --no-other-configs --no-cpu-baseline --no-kernels --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
run base
FASTVIM_BWD_GRID=224 run BWD_GRID=224
FASTVIM_BWD_GRID=448 run BWD_GRID=448
FASTVIM_COMBINE_GRID=256 run COMBINE_GRID=256
FASTVIM_COMBINE_GRID=448 run COMBINE_GRID=448
FASTVIM_NORM_WAVES=2048 run NORM_WAVES=2048
FASTVIM_NORM_WAVES=8192 run NORM_WAVES=8192
FASTVIM_GEMM_STREAM=0 run GEMM_STREAM=0
FASTVIM_GEMM_N96=0 run GEMM_N96=0
FASTVIM_XPROJ_ROWS=8 run XPROJ_ROWS=8
FASTVIM_XPROJ_ROWS=32 run XPROJ_ROWS=32
FASTVIM_FWD_NP=2 run FWD_NP=2
run base
