#!/bin/bash
# async per-call throughput by producer threads x documents in flight per thread: bash tools/r02_percall_depth.sh
make -C tools/percall -s
python - <<PY
import numpy as np, sys
sys.path.insert(0, ".")
from jtokkit_amd import corpus
for name, (t, o) in (("cfg1", corpus.sentences(1000)), ("cfg2", corpus.english(20000))):
    with open("/tmp/percall_%s.bin" % name, "wb") as f:
        f.write(np.int64(len(o) - 1).tobytes()); f.write(o.tobytes()); f.write(t.tobytes())
PY
export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
for c in cfg2 cfg1; do for tk in 2:8192 4:1024 4:4096 8:4096 16:1024 64:1; do
  T=${tk%%:*}; K=${tk##*:}
  ./tools/percall/percall_bench jtokkit_amd/libjtokkit_amd.so oracle/libjtk_oracle.so jtokkit_amd/data/cl100k_base.tiktoken /tmp/percall_$c.bin $T $K 1 2 | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$c threads',d['threads'],'K',d['in_flight_per_thread'],'blocking %.0f'%d['service_blocking']['docs_per_s'],'async %.0f (%.1f/batch)'%(d['service_async']['docs_per_s'],d['service_async']['docs_per_device_batch']),'oracle %.0f'%d['oracle_per_call']['docs_per_s'], flush=True)
"
done; done
