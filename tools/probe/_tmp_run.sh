mkdir -p gpurun_out/prof gpurun_out/r05
bash tools/profile_step.sh r05_v3 > gpurun_out/r05/profile_step_v3.log 2>&1
bash tools/probe/r05_trace.sh r05_v3s > gpurun_out/r05/trace_v3s.log 2>&1
bash tools/profile_step.sh r05_cfg3 trace-only --model B --steps 10 --warmup 3 > /dev/null 2>&1
bash tools/profile_step.sh r05_cfg4 trace-only --model B --batch 8 --img 2048 --steps 5 --warmup 2 > /dev/null 2>&1
bash tools/profile_step.sh r05_cfg5 trace-only --model C --batch 64 --steps 8 --warmup 3 > /dev/null 2>&1
bash tools/profile_step.sh r05_vim trace-only --model V --steps 5 --warmup 2 > /dev/null 2>&1
ls -la gpurun_out/prof | tail -30
