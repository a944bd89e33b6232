#!/bin/bash
# blocking callers against the number of service workers (configs[1] documents): bash tools/percall_workers.sh <tag>
tag=$1
make -C tools/percall -s
python - <<PY
import numpy as np, sys
sys.path.insert(0, ".")
from jtokkit_amd import corpus
t, o = corpus.english(20000)
with open("/tmp/percall_cfg2.bin", "wb") as f:
    f.write(np.int64(len(o) - 1).tobytes()); f.write(o.tobytes()); f.write(t.tobytes())
PY
export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
for T in 16 64 128; do for wk in 1 2 3 4 6 8; do
  JTK_SERVICE_TRACE=1 timeout -k 5 60 ./tools/percall/percall_bench jtokkit_amd/libjtokkit_amd.so - jtokkit_amd/data/cl100k_base.tiktoken /tmp/percall_cfg2.bin $T 1 1 $wk 2> /tmp/pw.err | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('threads',d['threads'],'workers',d['service_workers'],'blocking %.0f (%.1f/batch)'%(d['service_blocking']['docs_per_s'],d['service_blocking']['docs_per_device_batch']), flush=True)
"
  grep service /tmp/pw.err | head -1
done; done | tee gpurun_out/percall_workers_$tag.txt
