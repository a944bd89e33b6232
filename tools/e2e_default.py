"""Host buffers in, ids to pinned host memory, with the library's DEFAULT options (250k mixed documents): python tools/e2e_default.py"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import bench, jtokkit_amd
text, doc_off = bench.make_corpus("mixed", 250000, 3, 16)
enc = jtokkit_amd.get_encoding("cl100k_base")
hb = jtokkit_amd.HostBuffer(len(text)); hb.array[:] = text
b = enc.new_batch()
b.encode_host(hb.array, doc_off, ordinary=False, to_host=True)
t0 = time.perf_counter()
for _ in range(3): b.encode_host(hb.array, doc_off, ordinary=False, to_host=True)
dt = (time.perf_counter() - t0) / 3
print("defaults: %.1f ms for %.2f GB -> %.1f GB/s" % (dt * 1e3, len(text) / 1e9, len(text) / dt / 1e9))
