# does flushing fp32 denormals (-fgpu-flush-denormals-to-zero: shorter expansions of exp / log / rcp / rsqrt) move the step?
mkdir -p gpurun_out/r04v
for rep in 1 2; do
for fl in "" "-fgpu-flush-denormals-to-zero"; do
  FASTVIM_EXTRA_FLAGS="$fl" python -m fastvim_amd.build --force > gpurun_out/r04v/build.log 2>&1 || tail -3 gpurun_out/r04v/build.log
  echo -n "flags [$fl]: "; python tools/probe/bench_ms.py --steps 20 --warmup 5
done; done | tee gpurun_out/r04v/ftz.log
