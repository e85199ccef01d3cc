# HIP-graph packet capture on / off for the FastVim-T step (the switch bench.py and fastvim_amd default to 0)
mkdir -p gpurun_out/r04t
for i in 1 2 3; do for v in 0 1; do echo -n "DEBUG_CLR_GRAPH_PACKET_CAPTURE=$v: "; DEBUG_CLR_GRAPH_PACKET_CAPTURE=$v python tools/probe/bench_ms.py --steps 40 --warmup 5; done; done | tee gpurun_out/r04t/packet.log
