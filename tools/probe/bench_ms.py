"""bench.py with the given arguments, reduced to 'ms_per_step' (for tools/ab.sh)."""
import json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--no-other-configs", "--no-cpu-baseline", "--no-kernels",
                      "--no-scan-op"] + sys.argv[1:], capture_output=True, text=True, cwd=R).stdout
d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
print(d["ms_per_step"], d["config"].get("final_loss"))
