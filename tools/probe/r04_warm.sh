mkdir -p gpurun_out/r04u
for i in 1 2; do for w in 5 25 100; do echo -n "steps 20 warmup $w: "; python tools/probe/bench_ms.py --steps 20 --warmup $w; done; echo -n "steps 100 warmup 5: "; python tools/probe/bench_ms.py --steps 100 --warmup 5; done | tee gpurun_out/r04u/warm.log
