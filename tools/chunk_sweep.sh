#!/bin/bash
# Headline (configs[2]) against chunk size and chunks in flight:  bash tools/chunk_sweep.sh <tag>
tag=$1
out=gpurun_out/sweep_$tag
mkdir -p $out
for cfg in "1024 2" "1024 3" "1024 4" "512 4" "2048 2" "2048 3" "4096 1"; do
  set -- $cfg
  timeout -k 10 170 python bench.py --no-subrecords --no-cpu-baseline --no-verify --steps 4 --warmup 1 --chunk-mb $1 --in-flight $2 > $out/c$1_f$2.json 2> $out/c$1_f$2.err
  python - <<PY
import json
try:
    j = json.loads(open("$out/c$1_f$2.json").read().strip().splitlines()[-1])
    print("chunk $1 MiB, $2 in flight: %.1f MB/s  %.3f ms/step" % (j["value"], j["ms_per_step"]))
except Exception as e:
    print("chunk $1 MiB, $2 in flight: failed", e)
PY
done
