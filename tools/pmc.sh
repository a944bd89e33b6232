#!/bin/bash
# PMC passes (one counter group per run; --pmc is never combined with trace domains) over a serial bench run:
#   bash tools/pmc.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>/summary.csv
set +e
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -k 5 ${PMC_PASS_TIMEOUT:-240} rocprofv3 --pmc "$@" -d $out/$name -o c --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-subrecords --serial --gen-workers 1 $BENCH_ARGS > $out/$name.log 2>&1
  echo "pass $name done"
}
BENCH_ARGS="$*"
if [ -z "$PMC_SQ_ONLY" ]; then   # PMC_SQ_ONLY=1: instruction / wait counters only
pass fetch FETCH_SIZE
pass write WRITE_SIZE
fi
if [ -z "$PMC_HBM_ONLY" ]; then   # PMC_HBM_ONLY=1: the two HBM passes alone
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
fi
if [ -n "$PMC_CACHES" ]; then   # PMC_CACHES=1: also the cache counters (slow on large batches)
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum
fi
cd $root
python3 tools/pmc_summary.py $out $out/summary.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/summary.csv")))
keys=["kernel","hbm_read_MB_corrected","hbm_write_MB","SQ_BUSY_CYCLES","SQ_WAVE_CYCLES","SQ_INSTS_VALU","SQ_INSTS_LDS","SQ_INSTS_VMEM_RD","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_LDS_BANK_CONFLICT","SQ_WAIT_INST_LDS","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","TCP_TCC_READ_REQ_sum","TCP_TOTAL_CACHE_ACCESSES_sum","TCC_HIT_sum","TCC_MISS_sum"]
for r in rows:
    print(" | ".join("%s=%s"%(k.replace("SQ_","").replace("_sum",""),r.get(k,"")) for k in keys))
PY
