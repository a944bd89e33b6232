"""Per-call shapes with the service's trace on (JTK_SERVICE_TRACE=1): where a worker's time goes per device batch.
usage: python tools/percall_trace.py [threads ...]   (blocking callers, configs[1] documents)"""
import os, subprocess, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jtokkit_amd import corpus
exe = os.path.join(ROOT, "tools", "percall", "percall_bench")
subprocess.check_call(["make", "-C", os.path.dirname(exe), "-s"])
t, o = corpus.english(20000)
path = "/tmp/jtk_percall_trace_%d.bin" % os.getpid()
with open(path, "wb") as f:
    f.write(np.int64(len(o) - 1).tobytes()); f.write(o.tobytes()); f.write(t.tobytes())
env = dict(os.environ, JTK_SERVICE_TRACE="1", LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
for threads in [int(a) for a in sys.argv[1:]] or [16, 64]:
    p = subprocess.run([exe, os.path.join(ROOT, "jtokkit_amd", "libjtokkit_amd.so"), os.path.join(ROOT, "oracle", "libjtk_oracle.so"),
                        os.path.join(ROOT, "jtokkit_amd", "data", "cl100k_base.tiktoken"), path, str(threads), "1", "1"],
                       capture_output=True, text=True, env=env, timeout=120)
    r = json.loads(p.stdout)
    print(threads, "threads:", {k: r[k]["docs_per_s"] for k in ("direct", "service_blocking", "service_async", "oracle_per_call")})
    print(p.stderr[-1200:])
os.unlink(path)
