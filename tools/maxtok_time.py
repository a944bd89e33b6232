"""Time jtk_batch_encode_max_tokens on the mixed corpus (JTK_MAXTOK_TRACE=1 prints the rounds)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jtokkit_amd
from jtokkit_amd import corpus
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
mx = int(sys.argv[2]) if len(sys.argv) > 2 else 10
text, off = corpus.mixed(nd)
enc = jtokkit_amd.get_encoding("cl100k_base")
b = enc.new_batch()
for i in range(3):
    t0 = time.perf_counter()
    r = b.encode_max_tokens(text, off, mx, ordinary=True)
    print("call %d: %.1f ms" % (i, (time.perf_counter() - t0) * 1e3), file=sys.stderr)
