#!/bin/bash
# WRITE_SIZE (and FETCH_SIZE-free) calibration: what does the counter report for 64 MiB written with 4 / 8 / 16-byte stores?
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/probe/write_size.hip -o /tmp/write_size || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ws_pmc
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/ws_pmc -- /tmp/write_size > /dev/null 2>&1
F=$(find /tmp/ws_pmc -name "*counter_collection.csv" | head -1)
python3 - "$F" <<'PY' | tee $OUT/write_size_calibration.log
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if r.get("Counter_Name") == "WRITE_SIZE":
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
print("WRITE_SIZE calibration: 64 MiB = 65536 KiB stored per kernel")
for k, v in acc.items():
    m = sum(v) / len(v)
    print(f"  {k:28s} WRITE_SIZE {m:10.0f} KiB  = {m / 65536:5.2f} x the bytes stored   ({len(v)} launches)")
PY
