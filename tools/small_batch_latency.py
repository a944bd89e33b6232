"""Latency of one small device batch (27 documents of 1 KB, what 64 blocking per-call threads produce): python tools/small_batch_latency.py"""
import sys, time, numpy as np
sys.path.insert(0, ".")
import torch, jtokkit_amd
from jtokkit_amd import corpus
enc = jtokkit_amd.get_encoding("cl100k_base", device=0)
text, off = corpus.english(27)
print("bytes", len(text), "docs", len(off) - 1)
b = enc.new_batch()
hb_text = jtokkit_amd.encoding.HostBuffer(len(text) + 64); hb_off = jtokkit_amd.encoding.HostBuffer((len(off)) * 8)
d_text = torch.from_numpy(text).cuda(); d_off = torch.from_numpy(off).cuda()
def t(f, n=2000):
    for _ in range(50): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print("device-resident encode + sync: %.1f us" % t(lambda: b.encode_device(d_text.data_ptr(), d_off.data_ptr(), len(off) - 1, len(text), ordinary=False, sync=True)))
b.set_profiling(True)
b.encode_device(d_text.data_ptr(), d_off.data_ptr(), len(off) - 1, len(text), ordinary=False, sync=True)
print("kernel times (events):", {k: round(v * 1e3, 1) for k, v in b.kernel_times().items()}, "us")
b.set_profiling(False)
hb_text.array[:len(text)] = text
pin_off = np.frombuffer(hb_off.array, dtype=np.int64)[:len(off)]
pin_off[:] = off
def svc():
    b.encode_host(hb_text.array[:len(text)], pin_off, to_host=True)
    b.host_result()
print("pinned in, TO_HOST, host_result (what a service worker does per device batch): %.1f us" % t(svc))
