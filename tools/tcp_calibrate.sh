#!/bin/bash
# What one TCP_TOTAL_ACCESSES count is: the gather microbenchmark (known lane-loads per launch) under the counter
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tcpcal
timeout -k 5 120 rocprofv3 --pmc TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace -d /tmp/tcpcal -o c --output-format csv -- $GRAFT_REPO_ROOT/tools/microbench/gather_rate > /tmp/tcpcal.log 2>&1
tail -12 /tmp/tcpcal.log
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/tcpcal/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
acc = {}
for r in rows:
    k = (int(r["Dispatch_Id"]), r["Kernel_Name"][:60])
    acc.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(acc): print(k, acc[k])
PY
