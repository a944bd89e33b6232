#!/usr/bin/env python3
"""One-off check of a batch beyond 2^32 bytes (cfg-3-shaped mixed UTF-8): encode on the device, decode the tokens on
the device, compare with the input; a document sample against the oracle.  Not part of the test suite (minutes of host
time to build the corpus).  usage: python tools/big_batch_check.py [n_docs]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import jtokkit_amd
from jtokkit_amd import corpus
import oracle_lib

def main():
    n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1100000
    t0 = time.time()
    text, doc_off = corpus.mixed(n_docs, seed=41)
    print("corpus: %d docs, %.3f GB in %.0f s" % (n_docs, len(text) / 1e9, time.time() - t0), flush=True)
    enc = jtokkit_amd.get_encoding("cl100k_base")
    o = oracle_lib.get("cl100k_base")
    b = enc.new_batch()
    t0 = time.time()
    nt = b.encode_host(text, doc_off, ordinary=True)
    t_enc = time.time() - t0
    res = b.fetch()
    ok = bool((res.status == 0).all() and nt == len(res.tokens) and (np.diff(res.tok_off) >= 0).all() and res.tok_off[-1] == nt)
    nb = b.decode_host(res.tokens, res.tok_off)
    out, byte_off, status = b.decode_fetch()
    ok = ok and nb == len(text) and bool((status == 0).all()) and bool(np.array_equal(byte_off, doc_off)) and bool(np.array_equal(out, text))
    rng = np.random.default_rng(3)
    bad = 0
    for d in rng.choice(n_docs, 500, replace=False).tolist() + [0, n_docs - 1]:
        if res.doc(d).tolist() != o.encode_ordinary(text[doc_off[d]:doc_off[d + 1]].tobytes()):
            bad += 1
    print(json.dumps({"n_docs": n_docs, "bytes": int(len(text)), "tokens": int(nt), "beyond_2^32_bytes": bool(len(text) > 2**32),
                      "encode_host_s": round(t_enc, 3), "round_trip_and_offsets_ok": ok, "oracle_sample_mismatches": bad}))
    sys.exit(0 if ok and bad == 0 else 1)

if __name__ == "__main__":
    main()
