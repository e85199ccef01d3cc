# target-feature variants of the code object: gfx950 (xnack any) vs gfx950:xnack- (no replay-safe sequences needed)
mkdir -p gpurun_out/r04x
for rep in 1 2; do
for a in gfx950 gfx950:xnack- gfx950:sramecc+:xnack-; do
  FASTVIM_ARCH="$a" python -m fastvim_amd.build --force > gpurun_out/r04x/build.log 2>&1 || tail -3 gpurun_out/r04x/build.log
  echo -n "arch [$a]: "; python tools/probe/bench_ms.py --steps 20 --warmup 5
done; done | tee gpurun_out/r04x/xnack.log
