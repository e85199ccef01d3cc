#!/bin/bash
# A/B of the fused projection + norm kernels' rows per workgroup (tuning build: FASTVIM_FUSED_RPT): step time of the
# FastVim-T headline configuration, alternating.
python -m fastvim_amd.build --tuning > /dev/null || exit 1
run() { python bench.py --no-other-configs --no-cpu-baseline --no-kernels --no-scan-op --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['config']['final_loss'])"; }
for i in 1 2; do
FASTVIM_FUSED_RPT=64 run rpt64
FASTVIM_FUSED_RPT=0 run auto
FASTVIM_FUSED_RPT=56 run rpt56
FASTVIM_FUSED_RPT=33 run rpt33
done
python -m fastvim_amd.build > /dev/null
