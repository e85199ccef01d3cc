R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/seg
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in seg one; do
  if [ $mode = seg ]; then A="--segmented --buckets 3"; else A=""; fi
  rm -rf /tmp/tr
  rocprofv3 --kernel-trace -d /tmp/tr -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernels --no-other-configs --no-scan-op $A > $OUT/$mode.json 2> $OUT/$mode.err
  DB=$(find /tmp/tr -name "*.db" | head -1)
  python3 $R/tools/step_sequence.py $DB all > $OUT/$mode.seq
done
