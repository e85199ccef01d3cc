mkdir -p gpurun_out/r06_t
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r06_t/full_tests.log; cat gpurun_out/r06_t/full_tests.log
python bench.py --no-other-configs --no-cpu-baseline --no-kernels --no-scan-op 2>/dev/null | tail -1 | cut -c1-200
