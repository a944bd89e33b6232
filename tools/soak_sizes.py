#!/usr/bin/env python3
"""Size sweep on one reused batch object (GPU box, not part of the suite): prefixes of one mixed-script text whose byte
lengths straddle every internal boundary (64-byte blocks, 2 KiB tiles, the split kernel's 31,744-byte span and its halves), interleaved
with longer batches, each compared with the oracle.  usage: python tools/soak_sizes.py"""
import sys, random, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import jtokkit_amd, oracle_lib, regex_crosscheck as rc

def main():
    t0 = time.time()
    rng = random.Random(77)
    s = "".join(rc.random_text(rng, 200) + rng.choice(["\n", " ", "\r\n", ""]) for _ in range(600))
    raw = s.encode("utf-8")
    checked = 0
    for name in ("cl100k_base", "r50k_base"):
        enc = jtokkit_amd.get_encoding(name); o = oracle_lib.get(name)
        b = enc.new_batch()
        targets = set()
        for base in (64, 2048, 4096, 15872, 15872 * 2, 15872 * 3, 31744 * 2, 32768, 65536):
            for d in range(-130, 131, 1 if base >= 2048 else 7):
                if 0 < base + d <= len(raw): targets.add(base + d)
        for n in sorted(targets, key=lambda x: (x * 2654435761) % 1000003):       # shuffled: long and short alternate
            m = n
            while m > 0 and (raw[m] & 0xC0) == 0x80 if m < len(raw) else False: m -= 1   # cut at a character boundary
            doc = raw[:m]
            text = np.frombuffer(doc, dtype=np.uint8)
            b.encode_host(text, np.array([0, m], dtype=np.int64), ordinary=True)
            res = b.fetch()
            assert res.tokens.tolist() == o.encode_ordinary(doc), (name, n, m)
            checked += 1
        b.close()
    print("soak sizes ok: %d lengths in %.0f s" % (checked, time.time() - t0))

if __name__ == "__main__":
    main()
