mkdir -p gpurun_out/r06_h
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r06_h/full_tests.log; cat gpurun_out/r06_h/full_tests.log
bash tools/profile_step.sh r06_v2 > gpurun_out/r06_h/profile.log 2>&1; tail -3 gpurun_out/r06_h/profile.log
bash tools/probe/r06_scan_pmc.sh > gpurun_out/r06_h/scan_pmc.log 2>&1; tail -12 gpurun_out/r06_h/scan_pmc.log
