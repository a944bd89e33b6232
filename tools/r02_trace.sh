#!/bin/bash
# kernel trace of serial (inflight 1) english and mixed runs: bash tools/r02_trace.sh <tag>
set -e
tag=$1
root=$(pwd)
out=$root/gpurun_out/kt_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/english -o kt -- python3 $root/bench.py --no-cpu-baseline --no-verify --inflight 1 --steps 10 > $out/english.json 2> $out/english.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/mixed -o kt -- python3 $root/bench.py --no-cpu-baseline --no-verify --inflight 1 --steps 10 --workload mixed --docs-per-gpu 25000 > $out/mixed.json 2> $out/mixed.log
for w in english mixed; do echo "== $w"; f=$(find $out/$w -name "*kernel_stats.csv" | head -1); cut -d, -f1-4,6,7 $f | sed 's/(anonymous namespace):://' | head -14; done
