mkdir -p gpurun_out/r06_t
python -m pytest tests/test_xproj_bwd_gpu.py -m gpu -x -q 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/tx -- python3 $GRAFT_REPO_ROOT/bench.py --model B --batch 128 --steps 6 --warmup 2 --no-cpu-baseline --no-kernels --no-other-configs --no-scan-op > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof/tx -name "*.db" | head -1)
python3 tools/rocpd_stats.py $DB gpurun_out/prof/tx_stats.csv > /dev/null 2>&1
grep -i "xproj\|conv_pool_bwd" gpurun_out/prof/tx_stats.csv | cut -c1-160
rm -rf gpurun_out/prof/tx
