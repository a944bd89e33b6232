#!/bin/bash
# kernel trace of single-chunk runs of configs[1] and of a 250k-doc (1 GiB) slice of the mixed corpus: bash tools/r02_trace.sh <tag>
set -e
tag=$1
root=$(pwd)
out=$root/gpurun_out/kt_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/english -o kt -- python3 $root/bench.py --workload cfg2 --no-cpu-baseline --no-verify --no-subrecords --steps 10 > $out/english.json 2> $out/english.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/mixed -o kt -- python3 $root/bench.py --workload cfg3 --docs 250000 --no-cpu-baseline --no-verify --no-subrecords --steps 5 > $out/mixed.json 2> $out/mixed.log
for w in english mixed; do echo "== $w"; python3 - <<PY
import csv, json
rows = list(csv.DictReader(open("$out/$w/kt_kernel_stats.csv")))
d = json.load(open("$out/$w.json"))
print("value %.0f MB/s, %.3f ms/step" % (d["value"], d["ms_per_step"]))
for r in rows[:9]:
    print("%-28s calls %3s avg %9.1f us  min %9.1f" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-28:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
