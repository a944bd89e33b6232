#!/bin/bash
# stage times of library variants (tools/variants/*.so) on configs[1] and a 1 GiB slice of the mixed corpus, without the
# after-the-clock check (experiment builds may give wrong tokens)
for lib in default $(ls tools/variants/*.so 2>/dev/null); do
  if [ "$lib" != default ]; then export JTOKKIT_AMD_LIB=$PWD/$lib; fi
  for wl in "cfg2" "cfg3 --docs 250000" "vocab"; do
    timeout -k 5 150 python bench.py --workload $wl --steps 6 --warmup 2 --no-subrecords --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib | $wl |', d['value'], 'MB/s', d['ms_per_step'], 'ms', {k:round(v,3) for k,v in d['kernel_ms'].items() if v>0.05})
"
  done
done
