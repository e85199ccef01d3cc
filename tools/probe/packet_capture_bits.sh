#!/bin/bash
# Is the MAE step bitwise reproducible run to run, and does the runtime's graph packet capture change its bits?
# (final loss of 20 replays as a hex float; two runs per setting)
for pc in 0 0 1 1; do
  for cfg in "--model M --batch 64" "--model T --batch 128"; do
    out=$(DEBUG_CLR_GRAPH_PACKET_CAPTURE=$pc python bench.py $cfg --steps 20 --warmup 3 --no-kernels --no-cpu-baseline --no-other-configs --no-scan-op 2>/dev/null | tail -1)
    echo "pc=$pc $cfg :: $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['final_loss_hex'], d['config']['final_loss'])")"
  done
done
echo "eager (no graph), MAE:"
for i in 1 2; do
  out=$(python bench.py --model M --batch 64 --no-graph --steps 20 --warmup 3 --no-kernels --no-cpu-baseline --no-other-configs --no-scan-op 2>/dev/null | tail -1)
  echo "eager :: $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['final_loss_hex'], d['config']['final_loss'])")"
done
