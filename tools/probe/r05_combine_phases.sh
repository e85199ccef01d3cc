#!/bin/bash
# phase probes of the fused combine + out_proj + add + norm kernel: ONE object rebuilt with -DCG_DBG=n into a SCRATCH
# directory and linked with the tree's other objects into a scratch library that the timing script loads (PROBE_LIB) -- the
# in-tree library is never touched (the first version relinked it in place and left the CG_DBG=4 build behind).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05; S=/tmp/cgprobe; rm -rf $S; mkdir -p $S
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffast-math -fno-finite-math-only -Wno-unused-result -DNDEBUG -fno-slp-vectorize -fgpu-flush-denormals-to-zero"
OBJS=$(ls fastvim_amd/csrc/_obj/*.o | grep -v combine_gemm)
for d in 0 1 2 3 4; do
  /opt/rocm/bin/hipcc $FL -DCG_DBG=$d -x hip -c fastvim_amd/csrc/combine_gemm.hip -o $S/cg.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $S/lib$d.so $OBJS $S/cg.o
  echo -n "CG_DBG=$d: "; PROBE_LIB=$S/lib$d.so python tools/probe/r05_combine_time.py 2>&1 | grep -v amdgpu | head -1
done | tee gpurun_out/r05/combine_phases.log
