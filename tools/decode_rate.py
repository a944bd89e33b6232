#!/usr/bin/env python3
"""Device batch decode rate on the bench workload (ids and offsets resident in HBM): the encode result of 100k English
documents is decoded back.  usage: python tools/decode_rate.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jtokkit_amd
from jtokkit_amd import corpus

def main():
    text, doc_off = corpus.english(100000, seed=2)
    enc = jtokkit_amd.get_encoding("cl100k_base")
    b = enc.new_batch()
    b.encode_host(text, doc_off, ordinary=True)
    tok_ptr, off_ptr, _ = b.device_result()
    nt, nd, _ = b.result()
    for _ in range(3):
        b.decode_device(tok_ptr, off_ptr, nd, nt)
    t0 = time.perf_counter()
    steps = 10
    for _ in range(steps):
        nb = b.decode_device(tok_ptr, off_ptr, nd, nt)
    dt = (time.perf_counter() - t0) / steps
    out, byte_off, status = b.decode_fetch()
    ok = bool(np.array_equal(out, text) and np.array_equal(byte_off, doc_off))
    print(json.dumps({"tokens": int(nt), "bytes": int(nb), "ms_per_decode": round(dt * 1e3, 3), "output_GBps": round(nb / dt / 1e9, 1),
                      "tokens_per_s_G": round(nt / dt / 1e9, 2), "round_trip_ok": ok}))

if __name__ == "__main__":
    main()
