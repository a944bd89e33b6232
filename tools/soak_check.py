#!/usr/bin/env python3
"""Randomized soak run on the GPU box (not part of the suite): ~100k documents per run -- random mixed-script text,
long runs of one character, CRLF mixes, the mixed corpus -- through all four encodings on REUSED encodings and batch
objects, every batch compared with the oracle.  usage: python tools/soak_check.py [base_seed]"""
import sys, os, random, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import jtokkit_amd, oracle_lib, regex_crosscheck as rc
from jtokkit_amd import corpus
base = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
t0 = time.time()
total_docs = 0
for name in ("cl100k_base", "r50k_base", "p50k_base", "p50k_edit"):
    enc = jtokkit_amd.get_encoding(name); o = oracle_lib.get(name)
    for seed in range(6):
        rng = random.Random(base + seed)
        texts = [rc.random_text(rng, rng.randint(0, 400)) for _ in range(3000)]
        # long whitespace / digit / letter / CJK runs and CRLF mixes
        for _ in range(200):
            k = rng.randint(1, 3000)
            texts.append(rng.choice([" ", "\n", "\r\n", "a", "9", "日", "é", "'", "\t ", "🍕"]) * k + rng.choice(["", "x", " 1", "\n"]))
        bs = [t.encode("utf-8") for t in texts]
        doc_off = np.zeros(len(bs) + 1, dtype=np.int64); np.cumsum([len(b) for b in bs], out=doc_off[1:])
        text = np.frombuffer(b"".join(bs), dtype=np.uint8)
        res = enc.encode_batch_packed(text, doc_off, ordinary=True)
        exp_tok, exp_off = o.encode_batch(text, doc_off, threads=16)
        assert np.array_equal(res.tok_off, exp_off) and np.array_equal(res.tokens, exp_tok), (name, seed)
        total_docs += len(bs)
    for wl_seed in (base + 21, base + 22):
        text, doc_off = corpus.mixed(3000, seed=wl_seed)
        res = enc.encode_batch_packed(text, doc_off, ordinary=True)
        exp_tok, exp_off = o.encode_batch(text, doc_off, threads=16)
        assert np.array_equal(res.tok_off, exp_off) and np.array_equal(res.tokens, exp_tok), (name, wl_seed)
        total_docs += len(doc_off) - 1
print("soak ok: %d documents, 4 encodings, %.0f s" % (total_docs, time.time() - t0))
