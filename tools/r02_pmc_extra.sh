#!/bin/bash
# extra SQ counters (instruction fetch, scalar, LDS pipeline) per kernel:  bash tools/r02_pmc_extra.sh <tag> [bench args...]
set -e
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/pmcx_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
BENCH_ARGS="$*"
pass() {
  name=$1; shift
  rocprofv3 --pmc "$@" -d $out/$name -o c --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-subrecords --serial --gen-workers 1 $BENCH_ARGS > $out/$name.log 2>&1
  echo "pass $name done"
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM
pass b SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
pass c SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM
cd $root
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$out/*/c_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
for k in sorted(agg):
    if not k.startswith("k_") and "k_pretok" not in k: continue
    print(k)
    for c in sorted(agg[k]):
        print("   %-28s %12.4g per launch" % (c, agg[k][c] / max(1, cnt[k][c]) * 1.0))
PY
