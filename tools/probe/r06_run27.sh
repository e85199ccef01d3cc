mkdir -p gpurun_out/r06_t
for i in 1 2; do
  for w in 12 4; do
    for cfg in "--model C --batch 64 --steps 6 --warmup 2" "--model V --batch 128 --steps 8 --warmup 2"; do
      echo -n "FASTVIM_SCAN_CK_WAVES=$w $cfg: "; FASTVIM_SCAN_CK_WAVES=$w PROBE_LIB=$GRAFT_REPO_ROOT/ab/tuning.so python tools/probe/bench_ms.py $cfg 2>/dev/null | tail -1
    done
  done
done | tee gpurun_out/r06_t/ab_ck_waves.log
