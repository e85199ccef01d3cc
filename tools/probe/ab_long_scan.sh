for cfg in "--model C --batch 64 --steps 6 --warmup 2" "--model B --batch 8 --img 2048 --steps 4 --warmup 2" "--model V --batch 128 --steps 8 --warmup 2"; do
  echo "== $cfg"
  REPS=2 bash tools/ab.sh tools/probe/bench_ms.py $cfg
done
