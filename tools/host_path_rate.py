#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (jtk_batch_encode + jtk_batch_fetch) on the
bench workload: pageable numpy buffers in, numpy buffers out.  Reported in DESIGN.md, never as bench `value`."""
import json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jtokkit_amd
from jtokkit_amd import corpus

def main():
    n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    text, doc_off = corpus.english(n_docs, seed=2)
    enc = jtokkit_amd.get_encoding("cl100k_base", device=0)
    b = enc.new_batch()
    for _ in range(2):
        b.encode_host(text, doc_off, ordinary=True); r = b.fetch()
    t_enc = t_fetch = 0.0
    steps = 5
    for _ in range(steps):
        t0 = time.perf_counter(); b.encode_host(text, doc_off, ordinary=True); t1 = time.perf_counter()
        r = b.fetch(); t2 = time.perf_counter()
        t_enc += t1 - t0; t_fetch += t2 - t1
    mb = doc_off[-1] / 1e6
    print(json.dumps({"workload": "cl100k_base, %d English docs, %.1f MB, %d tokens" % (n_docs, mb, len(r.tokens)),
                      "encode_host_ms": round(t_enc / steps * 1e3, 2), "fetch_ms": round(t_fetch / steps * 1e3, 2),
                      "MBps_encode_only": round(mb / (t_enc / steps), 1),
                      "MBps_encode_plus_fetch": round(mb / ((t_enc + t_fetch) / steps), 1)}))

if __name__ == "__main__":
    main()
