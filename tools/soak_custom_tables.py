import sys, random, base64, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import jtokkit_amd, oracle_lib, regex_crosscheck as rc
from jtokkit_amd import corpus
from test_gpu_parity import _train_tiny_bpe
t0 = time.time()
for merges in (0, 40, 600, 3000):
    ranks = _train_tiny_bpe(corpus.english(60, seed=5)[0].tobytes() + " 日本語 の テキスト 한국어 ".encode() * 30, merges, seed=merges)
    data = b"\n".join(base64.b64encode(k) + b" " + str(v).encode() for k, v in sorted(ranks.items(), key=lambda kv: kv[1])) + b"\n"
    for kind in (0, 1):
        enc = jtokkit_amd.new_custom_encoding("tiny%d_%d" % (merges, kind), kind, ranks, {})
        o = oracle_lib.OracleEncoding("tiny%d_%d" % (merges, kind), kind, data, {})
        for seed in (1, 2):
            text, doc_off = corpus.mixed(1500, seed=seed + merges)
            res = enc.encode_batch_packed(text, doc_off, ordinary=True)
            exp_tok, exp_off = o.encode_batch(text, doc_off, threads=16)
            assert np.array_equal(res.tok_off, exp_off) and np.array_equal(res.tokens, exp_tok), (merges, kind, seed, "mixed")
            rng = random.Random(seed)
            texts = [rc.random_text(rng, rng.randint(0, 300)) for _ in range(1500)] + [rng.choice(["a", " ", "ab", "日", "the ", "\n"]) * rng.randint(1, 5000) for _ in range(60)]
            r2 = enc.encode_batch(texts, ordinary=True)
            for d, t in enumerate(texts):
                assert r2.doc(d).tolist() == o.encode_ordinary(t), (merges, kind, seed, d)
        enc.close()
    print("vocab of %d tokens ok" % len(ranks), flush=True)
print("soak4 ok in %.0f s" % (time.time() - t0))
