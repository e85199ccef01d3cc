mkdir -p gpurun_out/r04d
python -m fastvim_amd.build --tuning > gpurun_out/r04d/build.log 2>&1; tail -2 gpurun_out/r04d/build.log
python tools/probe/mid_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04d/mid_stamps.log
