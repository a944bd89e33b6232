import sys, random
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import jtokkit_amd
from jtokkit_amd import _native as N
enc = jtokkit_amd.get_encoding("cl100k_base")
rng = random.Random(5)
frag = [b"a", b" ", b"\xc3\xa9", b"\xe6\x97\xa5", b"\xf0\x9f\x8d\x95", b"\x80", b"\xc3", b"\xe6\x97", b"\xf0\x9f\x8d", b"\xed\xa0\x80", b"\xc0\xaf", b"\xf4\x90\x80\x80", b"\xff", b"\xe0\x80\x80", b"\xef\xbf\xbd", b"\xf4\x8f\xbf\xbf"]
docs = []
for _ in range(20000):
    docs.append(b"".join(rng.choice(frag) for _ in range(rng.randint(0, 12))))
res = enc.encode_batch(docs, ordinary=True, validate=True)
bad = 0
for d, b in enumerate(docs):
    try:
        b.decode("utf-8"); ok = True
    except UnicodeDecodeError:
        ok = False
    if (res.status[d] == 0) != ok:
        bad += 1
        if bad < 5: print("mismatch", d, b, res.status[d], ok)
print("soak3 validate: mismatches", bad, "of", len(docs), "invalid docs:", int((res.status != 0).sum()))
