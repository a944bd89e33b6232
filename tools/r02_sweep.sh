#!/bin/bash
# chunk size / chunks in flight sweep on a 200k-doc slice of the mixed corpus
mkdir -p gpurun_out/sweep
for wl in "cfg3 --docs 200000"; do
for opt in "--serial" "--in-flight 2" "--in-flight 3" "--in-flight 3 --chunk-mb 128" "--in-flight 2 --chunk-mb 512" "--in-flight 4 --chunk-mb 128" "--chunk-mb 2048"; do
  python bench.py --workload $wl --steps 5 --warmup 2 --no-subrecords --no-cpu-baseline --no-verify $opt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl | $opt |', d['value'], 'MB/s', d['ms_per_step'], 'ms', {k:round(v,2) for k,v in d['kernel_ms'].items() if v>0.05})
"
done; done
python bench.py --workload cfg2 --steps 10 --warmup 2 --no-subrecords --no-cpu-baseline 2>/dev/null | cut -c1-900
