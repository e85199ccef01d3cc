#!/bin/bash
# phase probes of the fused conv adjoint + data gradient kernel: ONE object rebuilt with -DCD_DBG=n into a SCRATCH directory,
# linked with the tree's other objects into a scratch library that the timing script loads (PROBE_LIB): the in-tree library is
# never touched.  CD_DBG: 1 no conv arithmetic, 2 no K loop, 3 no epilogue (returns behind the product tile), 4 no second phase
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06; S=/tmp/cdprobe; rm -rf $S; mkdir -p $S
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffast-math -fno-finite-math-only -Wno-unused-result -DNDEBUG -fno-slp-vectorize -fgpu-flush-denormals-to-zero"
OBJS=$(ls fastvim_amd/csrc/_obj/*.o | grep -v convpool_dgrad)
for d in 0 1 2 3 4; do
  /opt/rocm/bin/hipcc $FL -DCD_DBG=$d -x hip -c fastvim_amd/csrc/convpool_dgrad.hip -o $S/cd.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $S/lib$d.so $OBJS $S/cd.o
  echo -n "CD_DBG=$d: "; PROBE_ONLY_FUSED=1 PROBE_LIB=$S/lib$d.so python tools/probe/r06_convdgrad_time.py 2>&1 | grep -v amdgpu | tr '\n' ' '; echo
done | tee gpurun_out/r06/convdgrad_phases.log
