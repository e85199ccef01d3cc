#!/bin/bash
# Cache-side rocprofv3 --pmc passes over one grouped weight-gradient launch with its own operands per problem
# (tools/run_wgrad_group.py); every pass under its own timeout.  usage (on the GPU box): bash tools/pmc_wgrad.sh
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_wgrad
rm -rf $OUT; mkdir -p $OUT
run() {   # name counters...
  local name=$1; shift
  timeout 100 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/tools/run_wgrad_group.py 2 > $OUT/$name.log 2>&1
  echo "== $name rc=$? ($*)"
  f=$(find $OUT/$name -name "*counter_collection.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "grouped" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"   {k:40s} {sum(v)/len(v):16.0f}   (n={len(v)})")
PY
}
run c SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
run c2 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16
run a TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run a2 TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
run b TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum
