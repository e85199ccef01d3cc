mkdir -p gpurun_out/r06_t
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r06_t/full_tests.log; cat gpurun_out/r06_t/full_tests.log
bash tools/profile_step.sh r06_cfg3 trace-only --model B --steps 10 --warmup 3 > /dev/null 2>&1
bash tools/profile_step.sh r06_cfg5 trace-only --model C --batch 64 --steps 6 --warmup 2 > /dev/null 2>&1
bash tools/profile_step.sh r06_cfg4 trace-only --model B --batch 8 --img 2048 --steps 4 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r06_t/r06_v5_bench_builder_run.json 2> gpurun_out/r06_t/bench.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r06_t/r06_v5_bench_builder_run.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('floor_ratio'), d['roofline']['committed_profile']['step_trace'])
for k,v in d['other_configs'].items():
    if isinstance(v,dict) and 'ms_per_step' in v: print(k, v['ms_per_step'])
"
