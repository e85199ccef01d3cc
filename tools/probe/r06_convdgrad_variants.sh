#!/bin/bash
# build-time variants of the fused conv adjoint + data gradient kernel, each linked into a SCRATCH library (the in-tree one is
# never touched): stand-alone HBM-cold time (tools/probe/r06_convdgrad_time.py) and, with STEP=1, the FastVim-T step.
# usage: bash tools/probe/r06_convdgrad_variants.sh "-DCD_PDEPTH=6" "-DCD_W2_PIPE=0" ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06; S=/tmp/cdvar; rm -rf $S; mkdir -p $S
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffast-math -fno-finite-math-only -Wno-unused-result -DNDEBUG -fno-slp-vectorize -fgpu-flush-denormals-to-zero"
OBJS=$(ls fastvim_amd/csrc/_obj/*.o | grep -v convpool_dgrad)
i=0
for v in "" "$@"; do
  /opt/rocm/bin/hipcc $FL $v -x hip -c fastvim_amd/csrc/convpool_dgrad.hip -o $S/cd.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $S/lib$i.so $OBJS $S/cd.o
  echo -n "[$v] "; PROBE_ONLY_FUSED=1 PROBE_LIB=$S/lib$i.so python tools/probe/r06_convdgrad_time.py 2>&1 | grep -v amdgpu | tr '\n' ' '
  if [ "$STEP" = 1 ]; then echo -n " step: "; PROBE_LIB=$S/lib$i.so python tools/probe/bench_ms.py --steps 20 --warmup 5 | tr '\n' ' '; PROBE_LIB=$S/lib$i.so python tools/probe/bench_ms.py --steps 20 --warmup 5 | tr '\n' ' '; fi
  echo; i=$((i+1))
done | tee gpurun_out/r06/convdgrad_variants.log
