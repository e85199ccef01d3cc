#!/bin/bash
# phase probes of the fused combine + out_proj + add + norm kernel: rebuild ONE object with -DCG_DBG=n, relink, time
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffast-math -fno-finite-math-only -Wno-unused-result -DNDEBUG -fno-slp-vectorize -fgpu-flush-denormals-to-zero"
for d in 0 1 2 3 4; do
  /opt/rocm/bin/hipcc $FL -DCG_DBG=$d -x hip -c fastvim_amd/csrc/combine_gemm.hip -o fastvim_amd/csrc/_obj/combine_gemm.hip.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o fastvim_amd/libfastvim_hip.so fastvim_amd/csrc/_obj/*.o
  echo -n "CG_DBG=$d: "; python tools/probe/r05_combine_time.py 2>&1 | grep -v amdgpu | head -1
done | tee gpurun_out/r05/combine_phases.log
