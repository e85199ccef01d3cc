#!/bin/bash
# do the captured steps that replayed wrongly in rounds 1-2 under the runtime's graph "packet capture"
# (MAE step at bs >= 64, FastVim-T 512 px bs 32) still do?  Each configuration with the switch on and off, 60 replays;
# the build's own second stream (_SideStream) is OFF in every run (FlatTrainingState default).
run() {      # name, env var=value, bench args...
  name=$1; ev=$2; shift; shift
  out=$(env $ev python bench.py "$@" --steps 60 --warmup 5 --no-kernels --no-cpu-baseline --no-other-configs --no-scan-op 2>&1 | tail -1)
  echo "$name $ev :: $(echo "$out" | python -c "import sys,json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('finite, loss', d['config']['final_loss'], 'ms', d['ms_per_step'])
except Exception:
    print('FAILED:', l[-160:])")"
}
for pc in 0 1; do
run "MAE FastVim-B bs128   " DEBUG_CLR_GRAPH_PACKET_CAPTURE=$pc --model M --batch 128
run "MAE FastVim-B bs64    " DEBUG_CLR_GRAPH_PACKET_CAPTURE=$pc --model M --batch 64
run "FastVim-T 512px bs32  " DEBUG_CLR_GRAPH_PACKET_CAPTURE=$pc --model T --img 512 --batch 32
run "FastVim-T 224px bs128 " DEBUG_CLR_GRAPH_PACKET_CAPTURE=$pc --model T
done
