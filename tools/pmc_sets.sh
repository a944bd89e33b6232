#!/bin/bash
# A few counter passes (one rocprofv3 --pmc run each, no trace domains) over a short bench run, per-kernel averages printed:
#   bash tools/pmc_sets.sh <tag> "<bench args>" "SET1 counters..." "SET2 counters..." ...
set +e
tag=$1; shift
bargs=$1; shift
root=$(pwd)
out=$root/gpurun_out/pmcs_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $set -d $out/p$i -o c -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-subrecords --no-verify $bargs > $out/p$i.log 2>&1
  echo "pass $i done ($?): $set"
done
cd $root
python3 tools/pmc_summary.py $out $out/summary.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/summary.csv")))
for r in rows:
    if r["kernel"] in ("k_mark_docs","k_doc_offsets","k_tile_scan","k_validate_utf8"): continue
    print(r["kernel"])
    for k,v in r.items():
        if k not in ("kernel","launches") and v not in ("",None): print("   %-40s %s" % (k,v))
PY
