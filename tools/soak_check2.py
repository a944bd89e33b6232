#!/usr/bin/env python3
"""Second randomized soak (GPU box, not part of the suite): the API surface around the plain batch encode, on reused
objects, against the oracle: encode() with special-token literals sprinkled in (per-document status), maxTokens
truncation, batch decode round trips, very long single pieces (wave / workgroup kernels), a small trained rank table.
usage: python tools/soak_check2.py [seed0 [n_rounds]]"""
import sys, os, random, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import jtokkit_amd, oracle_lib, regex_crosscheck as rc
from jtokkit_amd import _native as N

def pack(bs):
    off = np.zeros(len(bs) + 1, dtype=np.int64)
    if bs: np.cumsum([len(b) for b in bs], out=off[1:])
    return (np.frombuffer(b"".join(bs), dtype=np.uint8) if off[-1] else np.zeros(0, dtype=np.uint8)), off

def main():
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    t0 = time.time(); checked = 0
    specials = ["<|endoftext|>", "<|fim_prefix|>", "<|endofprompt|>", "<|", "<|endoftext", "|>"]
    for name in ("cl100k_base", "p50k_edit", "r50k_base"):
        enc = jtokkit_amd.get_encoding(name); o = oracle_lib.get(name)
        b = enc.new_batch()
        for r in range(rounds):
            rng = random.Random(seed0 * 1000 + r)
            n = rng.choice([1, 7, 300, 2500])
            texts = []
            for _ in range(n):
                t = rc.random_text(rng, rng.randint(0, 200))
                if rng.random() < 0.15:
                    p = rng.randint(0, len(t)); t = t[:p] + rng.choice(specials) + t[p:]
                if rng.random() < 0.03:
                    t += rng.choice(["a", " ", "\n", "0", "é", "日", "x y "]) * rng.randint(300, 20000)
                texts.append(t)
            if rng.random() < 0.5:
                texts.append(rng.choice(["a", "\n", "9"]) * rng.randint(9000, 150000))     # a giant piece
            bs = [t.encode("utf-8") for t in texts]
            text, off = pack(bs)
            # encode(): per-document status for special tokens
            b.encode_host(text, off, ordinary=False)
            res = b.fetch()
            for d, t in enumerate(texts):
                try:
                    exp = o.encode(t)
                    assert res.status[d] == 0 and res.doc(d).tolist() == exp, (name, r, d, "encode")
                except oracle_lib.OracleError as e:
                    assert res.status[d] == N.JTK_ERR_UNSUPPORTED_SPECIAL, (name, r, d, "special", e.code)
            # encodeOrdinary + truncation + decode on the same batch object
            b.encode_host(text, off, ordinary=True)
            res = b.fetch()
            mx = rng.choice([0, 1, 2, 5, 17, 64, 1000])
            kept, flag = b.truncate(mx)
            for d, t in enumerate(texts):
                exp_toks, exp_tr = o.encode_ordinary(t, mx)
                assert res.doc(d)[:kept[d]].tolist() == exp_toks and bool(flag[d]) == exp_tr, (name, r, d, "truncate", mx)
            nb = b.decode_host(res.tokens, res.tok_off)
            out, boff, st = b.decode_fetch()
            assert nb == len(text) and np.array_equal(out, text) and np.array_equal(boff, off) and (st == 0).all(), (name, r, "decode")
            checked += 3 * len(texts)
        b.close()
    print("soak2 ok: %d document checks in %.0f s (seed0 %d)" % (checked, time.time() - t0, seed0))

if __name__ == "__main__":
    main()
