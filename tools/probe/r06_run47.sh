mkdir -p gpurun_out/r06_t
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r06_t/full_tests.log; cat gpurun_out/r06_t/full_tests.log
bash tools/profile_step.sh r06_v4 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r06_t/r06_v6_bench_builder_run.json 2> gpurun_out/r06_t/bench.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r06_t/r06_v6_bench_builder_run.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('floor_ratio'), d['roofline']['committed_profile']['step_trace'])
for k,v in d['other_configs'].items():
    if isinstance(v,dict) and 'ms_per_step' in v: print(k, v['ms_per_step'])
"
