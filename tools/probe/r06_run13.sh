mkdir -p gpurun_out/r06_m; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06_m/tr -- python3 $GRAFT_REPO_ROOT/tools/probe/r06_scan_op_shape.py cfg3 20 > /dev/null 2>&1
DB=$(find $GRAFT_REPO_ROOT/gpurun_out/r06_m/tr -name "*.db" | head -1)
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB $GRAFT_REPO_ROOT/gpurun_out/r06_m/scan_op_cfg3_kernel_stats.csv > /dev/null 2>&1
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r06_m/tr
head -14 $GRAFT_REPO_ROOT/gpurun_out/r06_m/scan_op_cfg3_kernel_stats.csv | cut -c1-160
