#!/bin/bash
# FastVim-B 224 px bs 128: A/B of the forward scan kernel for short pooled lengths and of the x_proj adjoint's pre-sum.
python -m fastvim_amd.build --tuning > /dev/null || exit 1
r() { echo -n "$1: "; shift; env "$@" 2>/dev/null | tail -1; }
for i in 1 2; do
r "base            " python tools/probe/ab_step.py B 224 128 8
r "fwd_short_ck    " FASTVIM_SCAN_FWD_SHORT_CK=1 python tools/probe/ab_step.py B 224 128 8
r "presum 8        " python tools/probe/ab_step.py B 224 128 8 --presum 8
r "both            " FASTVIM_SCAN_FWD_SHORT_CK=1 python tools/probe/ab_step.py B 224 128 8 --presum 8
done
python -m fastvim_amd.build > /dev/null
