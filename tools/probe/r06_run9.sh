mkdir -p gpurun_out/r06_i
bash tools/profile_step.sh r06_cfg3 trace-only --model B --steps 10 --warmup 3 > /dev/null 2>&1
bash tools/profile_step.sh r06_cfg4 trace-only --model B --batch 8 --img 2048 --steps 6 --warmup 2 > /dev/null 2>&1
bash tools/profile_step.sh r06_cfg5 trace-only --model C --batch 64 --steps 10 --warmup 3 > /dev/null 2>&1
bash tools/profile_step.sh r06_vim trace-only --model V --steps 6 --warmup 2 > /dev/null 2>&1
ls -la gpurun_out/prof | grep r06_
python bench.py > gpurun_out/r06_i/r06_v1_bench_builder_run.json 2> gpurun_out/r06_i/bench.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r06_i/r06_v1_bench_builder_run.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('floor_ratio'), d['roofline'].get('traffic'))
print(d['roofline_step'])
for k,v in d['other_configs'].items():
    if isinstance(v,dict) and 'ms_per_step' in v: print(k, v['ms_per_step'])
print({k:(v['us'],v.get('us_in_step_trace'),v['launches_per_step'],v.get('floor_ratio')) for k,v in d['kernels'].items()})
"
