# step time of the configurations with long pooled lengths (config 5, config 4, unpooled Vim-T); with ab/base.so present
# and AB=1 the same against the base library (tools/ab.sh)
for cfg in "--model C --batch 64 --steps 6 --warmup 2" "--model B --batch 8 --img 2048 --steps 4 --warmup 2" "--model V --batch 128 --steps 8 --warmup 2"; do
  echo "== $cfg"
  if [ "$AB" = 1 ]; then REPS=2 bash tools/ab.sh tools/probe/bench_ms.py $cfg; else python tools/probe/bench_ms.py $cfg; python tools/probe/bench_ms.py $cfg; fi
done
