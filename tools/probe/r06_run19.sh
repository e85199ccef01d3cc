export PROBE_LIB=$GRAFT_REPO_ROOT/ab/tuning.so
for v in 1 0; do echo "FASTVIM_SCAN_SHORT_SEG=$v"; FASTVIM_SCAN_SHORT_SEG=$v python tools/probe/r06_seg_probe.py 2>/dev/null; done
