#!/bin/bash
# L1 accesses of k_piece_resolve by phase: variants built with -DJTK_RES_PHASE=1 (staging + piece list only) and =2 (+ probes, no
# write-out) against the full kernel; one TCP counter pass each on a 1 GiB chunk of the mixed corpus
for v in res1 res2 full; do
  if [ $v != full ]; then export JTOKKIT_AMD_LIB=$PWD/tools/variants/$v.so; else unset JTOKKIT_AMD_LIB; fi
  bash tools/pmc_sets.sh resphase_$v "--workload cfg3 --docs 250000 --serial --gen-workers 1" "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_PENDING_STALL_CYCLES_sum" 2>&1 | grep -A 5 "k_piece_resolve" | sed "s/^/$v: /"
done
