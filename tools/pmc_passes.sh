#!/bin/bash
# Counter passes over bench.py on the GPU box (run through gpurun from the repo root):
#   bash tools/pmc_passes.sh <tag> [bench args...]
# One rocprofv3 --pmc run per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass), no trace
# domains combined with --pmc.  Summaries: python tools/pmc_summary.py gpurun_out/pmc_<tag>
set -e
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  rocprofv3 --pmc "$@" -d $out/$name -o c -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS > $out/$name.log 2>&1
  echo "pass $name done"
}
BENCH_ARGS="$*"
if [ -n "$PMC_QUICK" ]; then   # PMC_QUICK=1: two passes only
  pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
  pass fetch FETCH_SIZE
  exit 0
fi
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum
