#!/bin/bash
# which captured steps replay wrongly with the runtime's graph "packet capture", and does
# serialising the kernels (AMD_SERIALIZE_KERNEL=3: every dispatch waits for the previous one) change it?  The build's
# own second stream (_SideStream) is OFF in every run (FlatTrainingState default), so a failure here cannot be a
# missing cross-stream dependency of the build's launches.
run() {      # name, env...
  name=$1; shift
  out=$(env "$@" python bench.py --model M --batch 128 --steps 12 --warmup 3 --no-kernels --no-cpu-baseline --no-other-configs --no-scan-op 2>&1 | tail -1)
  echo "$name :: $(echo "$out" | python -c "import sys,json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('finite loss', d['config']['final_loss'], 'ms', d['ms_per_step'])
except Exception:
    print('FAILED:', l[-120:])")"
}
run "packet_capture=0                      " DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run "packet_capture=1                      " DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run "packet_capture=1 serialize_kernel=3   " DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 AMD_SERIALIZE_KERNEL=3
run "packet_capture=1 serialize_copy=3     " DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 AMD_SERIALIZE_COPY=3
run "packet_capture=1 no-graph (eager)     " DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 FASTVIM_EAGER=1
