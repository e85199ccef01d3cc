"""The op-level selective_scan_fn at the BASELINE config shapes (bench.py's scan_op table without the CPU column).
usage: python tools/probe/scan_op_time.py"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
from bench import scan_op_table
for k, v in scan_op_table(cpu=False).items():
    print(k, json.dumps({a: v[a] for a in ("fwd_us", "fwd_us_warm", "fwd_GBps", "fwd_hbm_frac", "fwd_bwd_us", "bwd_GBps")}), flush=True)
