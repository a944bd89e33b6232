"""End to end on the headline corpus (pinned host text in -> ids in pinned host memory) against host chunk size and chunks in
flight.  usage: python tools/e2e_sweep.py [docs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, jtokkit_amd
from jtokkit_amd import _native as N
docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
text, doc_off = bench.make_corpus("mixed", docs, 3, min(16, len(os.sched_getaffinity(0))))
print("corpus: %d docs, %.2f GB" % (docs, len(text) / 1e9), flush=True)
enc = jtokkit_amd.get_encoding("cl100k_base")
hb = jtokkit_amd.HostBuffer(len(text))
hb.array[:] = text
for mb, fl in ((64, 3), (32, 3), (32, 4), (64, 4), (128, 3), (128, 4), (256, 3), (64, 2)):
    b = enc.new_batch()
    b.set_option(N.JTK_OPT_HOST_CHUNK_BYTES, mb << 20)
    b.set_option(N.JTK_OPT_CHUNKS_IN_FLIGHT, fl)
    b.encode_host(hb.array, doc_off, ordinary=False, to_host=True)
    t0 = time.perf_counter()
    for _ in range(2):
        b.encode_host(hb.array, doc_off, ordinary=False, to_host=True)
    dt = (time.perf_counter() - t0) / 2
    nt = int(b.host_result().tok_off[-1])
    print("host chunks of %3d MiB, %d in flight: %.1f ms/step, %.1f GB/s of input, ids up at %.1f GB/s" % (
        mb, fl, dt * 1e3, len(text) / dt / 1e9, 4 * nt / dt / 1e9), flush=True)
    b.close()
