#!/bin/bash
# Where a kernel's time goes: libraries built with -DJTK_EXP=n stop a kernel after one of its phases (wrong results, right
# timing).  Build here (hipcc), run on the GPU box:   bash tools/r02_phases.sh build | run
# 1, 2: pack (loads only / no copy-out); 3, 4: piece_resolve (up to the piece list / up to the queue claim).  The merge's
# phase numbers in DESIGN.md 5.3 were taken with hooks 5-7 of commit 8ec8051 (tiny only / + bin 0 / + bins 1, 2).
set -e
if [ "$1" = build ]; then
  mkdir -p tools/exp
  for n in 1 2 3 4; do make -s -C jtokkit_amd/csrc OUT=../../tools/exp/libjtk_exp$n.so EXTRA=-DJTK_EXP=$n; done
  exit 0
fi
root=$(pwd)
mkdir -p gpurun_out/phases
cd /tmp && export TMPDIR=/tmp
for n in 0 1 2 3 4; do
  if [ $n = 0 ]; then unset JTOKKIT_AMD_LIB; else export JTOKKIT_AMD_LIB=$root/tools/exp/libjtk_exp$n.so; fi
  for wl in "cfg2" "cfg3 --docs 250000"; do
    tag=$(echo $wl | cut -d' ' -f1)
    rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/phases/e${n}_$tag -o kt -- python3 $root/bench.py --workload $wl --no-cpu-baseline --no-verify --no-subrecords --steps 5 > $root/gpurun_out/phases/e${n}_$tag.json 2> $root/gpurun_out/phases/e${n}_$tag.log || true
    python3 - <<PY
import csv
try:
    rows = list(csv.DictReader(open("$root/gpurun_out/phases/e${n}_$tag/kt_kernel_stats.csv")))
    d = {r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]: float(r["AverageNs"]) / 1e3 for r in rows}
    print("exp $n $tag: " + "  ".join("%s %.1f" % (k, d.get(k, 0)) for k in ("k_pretok_split", "k_piece_resolve", "k_bpe_merge", "k_pack_tokens")), flush=True)
except Exception as ex:
    print("exp $n $tag: failed", ex, flush=True)
PY
  done
done
