#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench command (run through gpurun from the repo root):
#   bash tools/kernel_trace.sh <tag> [bench args...]   -> gpurun_out/kt_<tag>/
set -e
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/kt_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python3 $root/bench.py --no-cpu-baseline "$@" > $out/bench.json 2> $out/rocprof.log
cat $out/bench.json
