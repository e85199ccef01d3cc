mkdir -p gpurun_out/r06_t
python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "reduc" 2>&1 | tail -15
for m in 0 1 2 3; do
echo "FASTVIM_REDUCE_VEC=$m"; FASTVIM_REDUCE_VEC=$m PROBE_LIB=ab/tuning.so python tools/probe/r06_reduce_time.py 2>&1 | grep -v amdgpu
done | tee gpurun_out/r06_t/ab_reduce_time.log
for m in 0 1; do
echo "FASTVIM_REDUCE_VEC=$m"; FASTVIM_REDUCE_VEC=$m PROBE_LIB=ab/tuning.so python tools/probe/r06_reduce_time.py --classes 2>&1 | grep -v amdgpu
done | tee gpurun_out/r06_t/ab_reduce_time_classes.log
