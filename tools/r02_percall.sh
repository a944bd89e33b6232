#!/bin/bash
# per-call shapes on the GPU box: writes gpurun_out/percall_<tag>.json     usage: tools/r02_percall.sh <tag> ["T:K ..."]
tag=$1
combos=${2:-"1:64 16:64 16:1024 64:1"}
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
for c in cfg1 cfg2; do for tk in $combos; do
  T=${tk%%:*}; K=${tk##*:}
  ./tools/percall/percall_bench jtokkit_amd/libjtokkit_amd.so oracle/libjtk_oracle.so jtokkit_amd/data/cl100k_base.tiktoken /tmp/percall_$c.bin $T $K 2 | sed "s/^{/{\"corpus\": \"$c\", /"
done; done | tee gpurun_out/percall_$tag.json
