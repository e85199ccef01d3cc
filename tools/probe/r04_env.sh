mkdir -p gpurun_out/r04y
for rep in 1 2; do
echo -n "default: "; python tools/probe/bench_ms.py --steps 20 --warmup 5
echo -n "HIP_FORCE_DEV_KERNARG=0: "; HIP_FORCE_DEV_KERNARG=0 python tools/probe/bench_ms.py --steps 20 --warmup 5
echo -n "HIP_FORCE_DEV_KERNARG=1: "; HIP_FORCE_DEV_KERNARG=1 python tools/probe/bench_ms.py --steps 20 --warmup 5
echo -n "GPU_MAX_HW_QUEUES=1: "; GPU_MAX_HW_QUEUES=1 python tools/probe/bench_ms.py --steps 20 --warmup 5
echo -n "HSA_ENABLE_INTERRUPT=0: "; HSA_ENABLE_INTERRUPT=0 python tools/probe/bench_ms.py --steps 20 --warmup 5
done | tee gpurun_out/r04y/env.log
