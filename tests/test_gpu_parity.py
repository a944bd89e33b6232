"""GPU parity tests: the HIP path, called through the C ABI (include/jtokkit_amd.h), against
  (1) the reference's golden CSV fixtures (reference/Cl100kBaseTestTest.java:21-111 and siblings),
  (2) the CPU oracle (oracle/jtk_oracle.cpp) on seeded synthetic inputs, bit-exact token ids.
Every test here needs a real MI355X (`-m gpu`).
"""
import os
import random

import numpy as np
import pytest

import golden_util
import oracle_lib
import regex_crosscheck as rc

pytestmark = pytest.mark.gpu

NAMES = golden_util.ENCODING_NAMES
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def jt():
    import jtokkit_amd
    return jtokkit_amd


def _assert_batch_equals_oracle(enc, oenc, texts, ordinary=True):
    res = enc.encode_batch(texts, ordinary=ordinary)
    assert len(res) == len(texts)
    assert res.tok_off[0] == 0
    bad = []
    for d, t in enumerate(texts):
        exp = oenc.encode_ordinary(t)
        got = res.doc(d).tolist()
        if got != exp:
            bad.append((d, t, got, exp))
    assert not bad, "first mismatch: %r" % (bad[0],)
    assert res.tok_off[-1] == len(res.tokens)
    return res


@pytest.mark.parametrize("name", NAMES)
def test_golden_rows_batch(jt, name):
    """All 423 rows of the reference's fixture as ONE batch: encode == expected (CsvFileSource tests)."""
    enc = jt.get_encoding(name)
    rows = golden_util.load_rows(name)
    res = enc.encode_batch([r[0] for r in rows])
    assert (res.status == 0).all()
    for d, (inp, expected, _) in enumerate(rows):
        assert res.doc(d).tolist() == expected, inp
    res_o = enc.encode_batch([r[0] for r in rows], ordinary=True)
    assert np.array_equal(res_o.tokens, res.tokens) and np.array_equal(res_o.tok_off, res.tok_off)


@pytest.mark.parametrize("name", NAMES)
def test_golden_rows_single_calls(jt, name):
    """Per-call Encoding methods, as reference/Cl100kBaseTestTest.java:21-103 exercises them."""
    enc = jt.get_encoding(name)
    rows = golden_util.load_rows(name)
    for inp, expected, expected10 in rows[::7] + rows[-2:]:
        assert enc.encode(inp) == expected                                      # :21-29
        assert enc.decode(enc.encode(inp)) == inp                               # :33-37
        r = enc.encode(inp, 10)                                                 # :41-52
        assert r.get_tokens() == expected10
        assert r.is_truncated() == (len(expected) > len(expected10))
        assert inp.startswith(enc.decode(r.get_tokens()))                       # :56-60
        assert enc.encode_ordinary(inp) == expected                             # :64-72
        r = enc.encode_ordinary(inp, 10)                                        # :76-88
        assert r.get_tokens() == expected10 and r.is_truncated() == (len(expected) > len(expected10))
        assert enc.count_tokens(inp) == len(expected)


def test_known_answer_literals(jt):
    enc = jt.get_encoding("cl100k_base")
    assert enc.encode("hello world") == [15339, 1917]
    assert enc.encode_ordinary("hello <|endoftext|> world") == [15339, 83739, 8862, 728, 428, 91, 29, 1917]
    assert list(enc.decode_bytes([15339, 1917])) == [104, 101, 108, 108, 111, 32, 119, 111, 114, 108, 100]
    assert enc.encode("This is a sample sentence.") == [2028, 374, 264, 6205, 11914, 13]
    r = enc.encode("This is a sample sentence.", 3)
    assert r.get_tokens() == [2028, 374, 264] and r.is_truncated()
    r = enc.encode("I love \U0001f355", 4)
    assert r.get_tokens() == [40, 3021] and r.is_truncated()
    assert enc.get_name() == "cl100k_base"


def test_special_tokens_and_errors(jt):
    enc = jt.get_encoding("cl100k_base")
    with pytest.raises(jt.UnsupportedOperationError):                           # GptBytePairEncoding.java:52-56
        enc.encode("hello <|endoftext|> world")
    with pytest.raises(jt.UnsupportedOperationError):
        enc.count_tokens("x<|fim_middle|>")
    s = "Hello<|endoftext|>, <|fim_prefix|> <|fim_middle|> world <|fim_suffix|> ! <|endofprompt|>"
    for name in NAMES:                                                          # Cl100kBaseTestTest.java:105-111
        e = jt.get_encoding(name)
        assert e.decode(e.encode_ordinary(s)) == s
    # batch: per-document status, other documents unaffected; literal split over two docs is not a hit
    res = enc.encode_batch(["a <|endoftext|> b", "plain text", "<|endof", "text|>", "<|endofprompt|>"])
    assert res.status.tolist() == [-2, 0, 0, 0, -2]
    assert res.doc(1).tolist() == enc.encode("plain text")
    # r50k knows only <|endoftext|>
    r5 = jt.get_encoding("r50k_base")
    assert r5.encode("a<|fim_prefix|>b") == oracle_lib.get("r50k_base").encode("a<|fim_prefix|>b")
    with pytest.raises(ValueError):                                             # :313 IllegalArgumentException
        enc.decode([100261])
    assert enc.encode(None) == [] and enc.encode("") == []
    r = enc.encode("", 10)
    assert r.get_tokens() == [] and not r.is_truncated()
    r = enc.encode("abc", 0)
    assert r.get_tokens() == [] and r.is_truncated()


def test_empty_and_ragged_batches(jt):
    enc = jt.get_encoding("cl100k_base")
    o = oracle_lib.get("cl100k_base")
    res = enc.encode_batch([])
    assert len(res) == 0 and len(res.tokens) == 0 and res.tok_off.tolist() == [0]
    res = enc.encode_batch(["", "", ""])
    assert res.tok_off.tolist() == [0, 0, 0, 0]
    texts = ["", "a", "", " ", "hello world", "", "\n", "", "x" * 70, "", ""]
    _assert_batch_equals_oracle(enc, o, texts)


@pytest.mark.parametrize("name,kind", [("cl100k_base", 1), ("r50k_base", 0), ("p50k_base", 0)])
def test_fuzz_vs_oracle(jt, name, kind):
    enc = jt.get_encoding(name)
    o = oracle_lib.get(name)
    rng = random.Random(4242 + kind)
    for rnd in range(6):
        texts = [rc.random_text(rng, rng.choice([4, 30, 200])) for _ in range(700)]
        _assert_batch_equals_oracle(enc, o, texts)


@pytest.mark.parametrize("name", ["cl100k_base", "r50k_base"])
def test_long_runs_and_long_pieces(jt, name):
    """Runs and pieces that cross LDS windows and tiles: digit / space / newline / letter / symbol runs
    of 63..9000 characters, multi-byte runs, at varied alignments (the reference has no fixture > 146 B)."""
    enc = jt.get_encoding(name)
    o = oracle_lib.get(name)
    texts = []
    units = ["7", " ", "\n", "a", "=", "\r\n", "　", "é", "中", "٣", " \n", "ab ", "\U0001f355", "x1", "'s"]
    for u in units:
        for n in (63, 64, 65, 130, 257, 700, 4095, 4097, 4500, 8000):
            k = max(1, n // len(u.encode()))
            for pre in ("", "ab", "!", "x " * 2029):
                for post in ("", "z", " z", "\n"):
                    if n > 700 and (pre in ("ab", "!") or post in ("z", "\n")):
                        continue
                    texts.append(pre + u * k + post)
    _assert_batch_equals_oracle(enc, o, texts)


def test_utf8_validation_flag(jt):
    """JTK_ENCODE_VALIDATE_UTF8: malformed documents are flagged per document; well-formed ones are not,
    and their tokens are unaffected.  Without the flag malformed bytes still round-trip as bytes."""
    enc = jt.get_encoding("cl100k_base")
    good = ["plain", "caf\u00e9 \u4e2d\u6587 \U0001f355", "", "\u00e9", "\U0001f355"]
    bad = [b"\xff", b"ab\x80", b"\xc3", b"\xe4\xb8", b"\xc0\xaf", b"\xed\xa0\x80", b"\xf4\x90\x80\x80", b"\xe0\x9f\xbf",
           b"x\xf0\x9f\x8d", b"\xf8\x88\x80\x80\x80"]
    docs = []
    for i in range(max(len(good), len(bad))):
        if i < len(good):
            docs.append(good[i].encode("utf-8"))
        if i < len(bad):
            docs.append(bad[i])
    exp = [(-6 if d in bad else 0) for d in docs]
    res = enc.encode_batch(docs, ordinary=True, validate=True)
    assert res.status.tolist() == exp
    for i, d in enumerate(docs):
        if d not in bad:
            assert res.doc(i).tolist() == enc.encode_ordinary(d.decode("utf-8"))
    # a sequence cut by the document boundary is malformed in both documents
    res = enc.encode_batch([b"\xe4\xb8", b"\xad"], ordinary=True, validate=True)
    assert res.status.tolist() == [-6, -6]
    res = enc.encode_batch(docs, ordinary=True)
    assert (res.status == 0).all() and enc.decode_bytes(res.tokens) == b"".join(docs)


def test_giant_pieces_and_limit(jt):
    """Pieces above the LDS kernels' 8 KiB (second phase, parts in global scratch) still equal the oracle;
    only a single unsplittable piece above 1 MiB is refused, per document."""
    enc = jt.get_encoding("cl100k_base")
    o = oracle_lib.get("cl100k_base")
    texts = ["ok text", "a" * 9000, "more ok", " " * 20000 + "x", "=" * 12345, "ab" * 6000, "\n" * 10000]
    res = _assert_batch_equals_oracle(enc, o, texts)
    assert (res.status == 0).all()
    res = enc.encode_batch(["fine", "a" * ((1 << 20) + 5), "also fine"])
    assert res.status.tolist() == [0, -10, 0]
    assert res.doc(0).tolist() == enc.encode("fine") and res.doc(2).tolist() == enc.encode("also fine")


@pytest.mark.parametrize("name", ["cl100k_base", "p50k_edit"])
def test_synthetic_corpora_vs_oracle(jt, name):
    """Seeded cfg-2-like (English) and cfg-3-like (mixed UTF-8) corpora, every document compared."""
    from jtokkit_amd import corpus
    enc = jt.get_encoding(name)
    o = oracle_lib.get(name)
    for text, doc_off in (corpus.sentences(1000), corpus.english(3000), corpus.mixed(600)):
        res = enc.encode_batch_packed(text, doc_off, ordinary=True)
        exp_tok, exp_off = o.encode_batch(text, doc_off, threads=8)
        assert np.array_equal(res.tok_off, exp_off)
        assert np.array_equal(res.tokens, exp_tok)
        assert (res.status == 0).all()


def test_full_size_properties_cfg2(jt):
    """BASELINE config 2 at full size (100k docs x ~1 KB): size-independent properties -- decode(tokens)
    reproduces the input bytes exactly, offsets are monotone, a 1 % document sample equals the oracle."""
    from jtokkit_amd import corpus
    enc = jt.get_encoding("cl100k_base")
    o = oracle_lib.get("cl100k_base")
    text, doc_off = corpus.english(100000)
    res = enc.encode_batch_packed(text, doc_off, ordinary=True)
    assert (res.status == 0).all()
    assert (np.diff(res.tok_off) >= 0).all() and res.tok_off[-1] == len(res.tokens)
    assert enc.decode_bytes(res.tokens) == text.tobytes()
    rng = np.random.default_rng(0)
    for d in rng.choice(len(doc_off) - 1, 1000, replace=False):
        doc = text[doc_off[d]:doc_off[d + 1]].tobytes()
        assert res.doc(d).tolist() == o.encode_ordinary(doc)


def test_device_entry_point_and_profiling(jt):
    """jtk_batch_encode_device with torch-owned HBM buffers, as bench.py drives it."""
    import torch
    from jtokkit_amd import corpus
    enc = jt.get_encoding("cl100k_base")
    o = oracle_lib.get("cl100k_base")
    text, doc_off = corpus.english(2000)
    dev = torch.device("cuda:0")
    d_text = torch.from_numpy(text).to(dev)
    d_off = torch.from_numpy(doc_off).to(dev)
    torch.cuda.synchronize()
    b = enc.new_batch()
    b.set_profiling(True)
    nt = b.encode_device(d_text.data_ptr(), d_off.data_ptr(), len(doc_off) - 1, len(text), ordinary=True)
    res = b.fetch()
    exp_tok, exp_off = o.encode_batch(text, doc_off, threads=8)
    assert nt == len(exp_tok) and np.array_equal(res.tokens, exp_tok) and np.array_equal(res.tok_off, exp_off)
    times = b.kernel_times()
    assert set(times) >= {"pretok_split", "bpe_merge", "pack"} and all(v >= 0 for v in times.values())
    # results also readable in place
    tp, op, sp = b.device_result()
    assert tp and op and sp
    b.close()


def test_table_swap_back_to_back(jt):
    """BASELINE config 5: r50k_base then p50k_base on the same corpus, tables cached per encoding."""
    from jtokkit_amd import corpus
    text, doc_off = corpus.mixed(300)
    for name in ("r50k_base", "p50k_base", "r50k_base"):
        enc = jt.get_encoding(name)
        res = enc.encode_batch_packed(text, doc_off, ordinary=True)
        exp_tok, exp_off = oracle_lib.get(name).encode_batch(text, doc_off, threads=8)
        assert np.array_equal(res.tokens, exp_tok) and np.array_equal(res.tok_off, exp_off)


@pytest.mark.parametrize("name", NAMES)
def test_batch_decode_golden_rows(jt, name):
    """Device batch decode == the oracle's decodeBytes on the reference's expected token lists, and == the input
    text (the round trip Cl100kBaseTestTest.java:33-37 checks), incl. empty lists."""
    enc = jt.get_encoding(name)
    o = oracle_lib.get(name)
    rows = golden_util.load_rows(name)
    lists = [r[1] for r in rows] + [[], rows[0][1], []]
    got = enc.decode_batch(lists)
    assert len(got) == len(lists)
    for q, toks in enumerate(lists):
        assert got[q] == o.decode_bytes(toks), q
    for q, (inp, _, _) in enumerate(rows):
        assert got[q] == inp.encode("utf-8")
    assert enc.decode_batch([]) == []


def test_batch_decode_unknown_and_special_ids(jt):
    """Unknown id -> JTK_ERR_UNKNOWN_TOKEN for that list only (GptBytePairEncoding.java:313); special-token ids
    decode to their literals (:308-311); p50k_base's hole at 50256 is the endoftext special."""
    enc = jt.get_encoding("cl100k_base")
    from jtokkit_amd import _native as N
    b = enc.new_batch()
    ids = np.array([15339, 1917, 15339, 999999, 1917, 100257, 15339, -5], dtype=np.int32)
    seq_off = np.array([0, 2, 5, 6, 6, 8], dtype=np.int64)
    nb = b.decode_host(ids, seq_off)
    out, byte_off, status = b.decode_fetch()
    assert status.tolist() == [0, N.JTK_ERR_UNKNOWN_TOKEN, 0, 0, N.JTK_ERR_UNKNOWN_TOKEN]
    raw = out.tobytes()
    assert raw[byte_off[0]:byte_off[1]] == b"hello world"
    assert raw[byte_off[2]:byte_off[3]] == b"<|endoftext|>"
    assert byte_off[3] == byte_off[4] and byte_off[-1] == nb == len(raw)
    with pytest.raises(jt.EncodingError):
        enc.decode_batch([[15339], [123456789]])
    b.close()
    p50k = jt.get_encoding("p50k_base")
    assert p50k.decode_batch([[50256]]) == [b"<|endoftext|>"]
    assert p50k.decode_batch([[50256]]) == [p50k.decode_bytes([50256])]


def test_batch_decode_full_size_round_trip(jt):
    """cfg-2-sized corpus: device decode of the device encode, document by document == the input bytes; mixed-script
    corpus likewise (long tokens, tiles whose bytes exceed the LDS stage)."""
    from jtokkit_amd import corpus
    enc = jt.get_encoding("cl100k_base")
    for text, doc_off in (corpus.english(100000), corpus.mixed(2000)):
        b = enc.new_batch()
        b.encode_host(text, doc_off, ordinary=True)
        res = b.fetch()
        nb = b.decode_host(res.tokens, res.tok_off)
        out, byte_off, status = b.decode_fetch()
        assert nb == len(text) and (status == 0).all()
        assert np.array_equal(byte_off, doc_off)
        assert np.array_equal(out, text)
        b.close()


@pytest.mark.parametrize("name", NAMES)
def test_batch_max_tokens_golden_rows(jt, name):
    """encode(text, 10) for all fixture rows in one device call == the fixture's third column (tokens) and the
    oracle's truncated flag (Cl100kBaseTestTest.java:40-67 semantics)."""
    enc = jt.get_encoding(name)
    o = oracle_lib.get(name)
    rows = golden_util.load_rows(name)
    got = enc.encode_batch_max_tokens([r[0] for r in rows], 10)
    for (inp, _, expected10), r in zip(rows, got):
        exp_toks, exp_tr = o.encode(inp, 10)
        assert exp_toks == expected10
        assert r.get_tokens() == expected10 and r.is_truncated() == exp_tr, inp


def test_batch_max_tokens_fuzz(jt):
    """Random mixed-script texts (multi-byte characters cut by the limit, U+FFFD in the text, empty documents),
    several limits incl. 0: device truncation == oracle encode(text, max) == the per-call jtk_encode path."""
    enc = jt.get_encoding("cl100k_base")
    o = oracle_lib.get("cl100k_base")
    rng = random.Random(7)
    texts = [rc.random_text(rng, rng.randint(0, 60)) for _ in range(300)]
    texts += ["", "�", "a��b", "🍕🍕🍕", "I love 🍕", "�" * 7, "é" * 9, "한국어 텍스트 " * 5]
    for mx in (0, 1, 2, 3, 5, 8, 13, 40, 10000, 2**31 - 1):
        got = enc.encode_batch_max_tokens(texts, mx, ordinary=True)
        for t, r in zip(texts, got):
            exp_toks, exp_tr = o.encode_ordinary(t, mx)
            assert r.get_tokens() == exp_toks and r.is_truncated() == exp_tr, (mx, t)
    for t in texts[::29] + texts[-8:]:
        for mx in (1, 3, 7):
            r = enc.encode_ordinary(t, mx)
            exp_toks, exp_tr = o.encode_ordinary(t, mx)
            assert r.get_tokens() == exp_toks and r.is_truncated() == exp_tr


def test_batch_max_tokens_long_documents_early_exit(jt):
    """Documents far longer than the limit needs (the early exit encodes their leading bytes only): prose, mixed scripts,
    code, white-space runs with line breaks placed around every prefix length the early exit tries (8 * max + 64, then
    x4), digits, contractions, specials.  == oracle encode(text, max) on the whole document, every encoding."""
    from jtokkit_amd import corpus
    rng = random.Random(11)
    docs = []
    for text, off in (corpus.english(40, seed=3), corpus.mixed(60, seed=4)):
        docs += [bytes(text[off[i]:off[i + 1]]).decode("utf-8") for i in range(len(off) - 1)]
    ws = [" ", "  ", "\n", " \n", "\n\n ", "\t", "\r\n", "\u00a0", "\u2003", "\u3000", " \n \n  \n ", "   "]
    for mx in (1, 4, 10, 50):
        for grow in (1, 4, 16):
            cut = (8 * mx + 64) * grow
            for _ in range(6):
                head = rc.random_text(rng, 400)[:cut - rng.randint(0, 40)]
                run = "".join(rng.choice(ws) for _ in range(rng.randint(1, 30)))
                docs.append(head + run + rc.random_text(rng, 200) + "they'll we've 1234567 " * 20)
    docs += [" " * 5000 + "x", "\n" * 3000 + "end", "a" * 9000, "1234567890" * 700, ("it's " * 40 + "\n") * 30,
             "word " * 2000, "한국어 텍스트 " * 600, "x" + " \n" * 2500 + "y" * 50, "hello <|endoftext|> " * 300]
    for name in NAMES:
        enc = jt.get_encoding(name)
        o = oracle_lib.get(name)
        for mx in (0, 1, 4, 10, 50, 700):
            got = enc.encode_batch_max_tokens(docs, mx, ordinary=True)
            for t, r in zip(docs, got):
                exp_toks, exp_tr = o.encode_ordinary(t, mx)
                assert r.get_tokens() == exp_toks and r.is_truncated() == exp_tr, (name, mx, t[:80])
    enc = jt.get_encoding("cl100k_base")
    with pytest.raises(Exception, match="special"):
        enc.encode_batch_max_tokens(["x" * 5000 + "<|endoftext|>"], 3)
    assert enc.encode_batch_max_tokens(["x" * 5000 + "<|endoftext|>"], 3, ordinary=True)[0].get_tokens() == \
        oracle_lib.get("cl100k_base").encode_ordinary("x" * 5000 + "<|endoftext|>", 3)[0]


def _train_tiny_bpe(corpus_bytes, n_merges, seed=0):
    """A small byte-pair table trained the textbook way: all 256 bytes (ranks shuffled, as in the real tables where
    rank != byte value), then n_merges merges of the most frequent adjacent pair inside whitespace-split words."""
    rng = random.Random(seed)
    order = list(range(256))
    rng.shuffle(order)
    ranks = {bytes([b]): r for r, b in enumerate(order)}
    words = {}
    for wd in corpus_bytes.split(b" "):
        if wd:
            key = tuple(bytes([c]) for c in b" " + wd)
            words[key] = words.get(key, 0) + 1
    for _ in range(n_merges):
        pairs = {}
        for wd, cnt in words.items():
            for a, b2 in zip(wd, wd[1:]):
                pairs[(a, b2)] = pairs.get((a, b2), 0) + cnt
        if not pairs:
            break
        (a, b2), _ = max(pairs.items(), key=lambda kv: (kv[1], kv[0]))
        if a + b2 in ranks:
            break
        ranks[a + b2] = len(ranks)
        new_words = {}
        for wd, cnt in words.items():
            out, i = [], 0
            while i < len(wd):
                if i + 1 < len(wd) and wd[i] == a and wd[i + 1] == b2:
                    out.append(a + b2); i += 2
                else:
                    out.append(wd[i]); i += 1
            new_words[tuple(out)] = new_words.get(tuple(out), 0) + cnt
        words = new_words
    return ranks


@pytest.mark.parametrize("kind", [0, 1])
def test_custom_encoding(jt, kind):
    """a custom rank table through the same kernels (registerGptBytePairEncoding /
    GptBytePairEncodingParams in the reference), both shipped patterns; GPU == oracle built from the same table."""
    import base64
    from jtokkit_amd import corpus
    text, doc_off = corpus.mixed(60, seed=11)
    ranks = _train_tiny_bpe(corpus.english(40, seed=5)[0].tobytes() + " 日本語 の テキスト 한국어 ".encode() * 20, 600)
    specials = {"<|endoftext|>": len(ranks) + 10}
    enc = jt.new_custom_encoding("tiny_%d" % kind, kind, ranks, specials)
    data = b"\n".join(base64.b64encode(k) + b" " + str(v).encode() for k, v in sorted(ranks.items(), key=lambda kv: kv[1])) + b"\n"
    o = oracle_lib.OracleEncoding("tiny_%d" % kind, kind, data, specials)
    assert enc.vocab_size() == len(ranks)
    res = enc.encode_batch_packed(text, doc_off, ordinary=True)
    exp_tok, exp_off = o.encode_batch(text, doc_off, threads=4)
    assert np.array_equal(res.tokens, exp_tok) and np.array_equal(res.tok_off, exp_off)
    rng = random.Random(3)
    texts = [rc.random_text(rng, rng.randint(0, 80)) for _ in range(200)]
    _assert_batch_equals_oracle(enc, o, texts)
    assert enc.decode_batch([res.doc(d).tolist() for d in range(5)]) == [text[doc_off[d]:doc_off[d + 1]].tobytes() for d in range(5)]
    with pytest.raises(jt.UnsupportedOperationError):
        enc.encode("a <|endoftext|> b")
    enc.close()


def test_large_batch_1gb(jt):
    """A 1 GB batch (cfg-3-shaped mixed UTF-8, 250k docs x ~4 KB): 64-bit positions everywhere, > 2^31 scratch bytes.
    Size-independent properties: device decode of the tokens reproduces the input exactly with the documents' byte
    offsets, token offsets are monotone; a document sample equals the oracle."""
    from jtokkit_amd import corpus
    enc = jt.get_encoding("cl100k_base")
    o = oracle_lib.get("cl100k_base")
    text, doc_off = corpus.mixed(250000, seed=31)
    assert len(text) > 900e6
    b = enc.new_batch()
    nt = b.encode_host(text, doc_off, ordinary=True)
    res = b.fetch()
    assert (res.status == 0).all() and nt == len(res.tokens)
    assert (np.diff(res.tok_off) >= 0).all() and res.tok_off[-1] == nt
    nb = b.decode_host(res.tokens, res.tok_off)
    out, byte_off, status = b.decode_fetch()
    assert nb == len(text) and (status == 0).all() and np.array_equal(byte_off, doc_off)
    assert np.array_equal(out, text)
    rng = np.random.default_rng(1)
    for d in rng.choice(len(doc_off) - 1, 300, replace=False).tolist() + [0, len(doc_off) - 2]:
        doc = text[doc_off[d]:doc_off[d + 1]].tobytes()
        assert res.doc(d).tolist() == o.encode_ordinary(doc), d
    b.close()


def test_headline_corpus_at_full_size(jt):
    """BASELINE config 3 at its full size -- the very corpus bench.py times (1M mixed UTF-8 documents, 4.1 GB, four 1 GiB
    chunks on two scratch sets), device-resident, encode() with the special-token check.  Size-independent properties over
    ALL documents: status 0, token offsets monotone and closed, device decode of the 1.77 G tokens reproduces the 4.1 GB byte
    for byte with the documents' offsets; 3000 sampled documents (and the first and last of every chunk) equal the oracle."""
    import sys
    import torch
    sys.path.insert(0, ROOT)
    import bench
    enc = jt.get_encoding("cl100k_base")
    o = oracle_lib.get("cl100k_base")
    text, doc_off = bench.make_corpus("mixed", 1000000, 3, min(16, len(os.sched_getaffinity(0))))
    assert len(doc_off) == 1000001 and len(text) > 4.0e9
    dev = torch.device("cuda:0")
    d_text, d_off = torch.from_numpy(text).to(dev), torch.from_numpy(doc_off).to(dev)
    b = enc.new_batch()
    nt = b.encode_device(d_text.data_ptr(), d_off.data_ptr(), len(doc_off) - 1, len(text), ordinary=False)
    res = b.fetch()
    assert (res.status == 0).all() and nt == len(res.tokens) and nt > 1.5e9
    assert res.tok_off[0] == 0 and res.tok_off[-1] == nt and (np.diff(res.tok_off) >= 0).all()
    del d_text
    nb = b.decode_host(res.tokens, res.tok_off)
    out, byte_off, status = b.decode_fetch()
    assert nb == len(text) and (status == 0).all() and np.array_equal(byte_off, doc_off)
    assert np.array_equal(out, text)
    del out
    rng = np.random.default_rng(2)
    edges = [d for c in range(1, 4) for d in (np.searchsorted(doc_off, c << 30) - 1, np.searchsorted(doc_off, c << 30))]
    for d in rng.choice(len(doc_off) - 1, 3000, replace=False).tolist() + [0, len(doc_off) - 2] + [int(e) for e in edges]:
        doc = text[doc_off[d]:doc_off[d + 1]].tobytes()
        assert res.doc(d).tolist() == o.encode(doc.decode("utf-8")), d
    b.close()


def test_ids_streamed_to_host_when_the_guess_is_too_small(jt):
    """JTK_ENCODE_TO_HOST sizes the pinned ids buffer by a guess (one token per two bytes) and fills it chunk by chunk while
    later chunks are encoded; text with more tokens than that -- emoji, digits with separators, control bytes -- makes the buffer
    grow in mid-job, with copies of earlier chunks still in flight.  3 MB in 256 KiB host chunks: == oracle."""
    enc = jt.get_encoding("cl100k_base")
    o = oracle_lib.get("cl100k_base")
    rng = random.Random(21)
    dense = ["😀", "🤖", "🧪", "\x01", "\x7f", "1,", "²", "\u0601", "🀄", "𝔘"]
    docs = ["".join(rng.choice(dense) for _ in range(rng.randint(1, 400))).encode("utf-8") for _ in range(6000)]
    doc_off = np.zeros(len(docs) + 1, dtype=np.int64)
    np.cumsum([len(d) for d in docs], out=doc_off[1:])
    text = np.frombuffer(b"".join(docs), dtype=np.uint8)
    assert len(text) > 2.5e6
    exp_tok, exp_off = o.encode_batch(text, doc_off, threads=8, ordinary=True)
    assert len(exp_tok) * 2 > len(text)                    # more than one token per two bytes: the guess is too small
    b = enc.new_batch()
    b.set_option(jt._native.JTK_OPT_HOST_CHUNK_BYTES, 256 << 10)
    for _ in range(2):
        nt = b.encode_host(text, doc_off, ordinary=True, to_host=True)
        res = b.host_result()
        assert nt == len(exp_tok) and np.array_equal(res.tok_off, exp_off) and np.array_equal(res.tokens[:nt], exp_tok)
    # and a sparse text right after on the same batch (the grown buffer is kept; no overflow this time)
    from jtokkit_amd import corpus
    t2, o2 = corpus.english(3000)
    e2, eo2 = o.encode_batch(t2, o2, threads=8, ordinary=True)
    nt = b.encode_host(t2, o2, ordinary=True, to_host=True)
    res = b.host_result()
    assert nt == len(e2) and np.array_equal(res.tok_off, eo2) and np.array_equal(res.tokens[:nt], e2)
    b.close()


def test_many_tiny_and_empty_documents(jt):
    """200k documents of 0..5 bytes (every byte position a document start somewhere, empty documents in runs, multi-byte
    characters alone in a document): every document equals the oracle, offsets are exact; also through device decode."""
    enc = jt.get_encoding("cl100k_base")
    o = oracle_lib.get("cl100k_base")
    rng = random.Random(11)
    alphabet = ["a", " ", "é", "日", "🍕", "1", "\n", "'s", "<|", "Z", "\r\n", "."]
    docs = []
    for _ in range(200000):
        k = rng.choice((0, 0, 1, 1, 2, 3))
        docs.append("".join(rng.choice(alphabet) for _ in range(k)).encode("utf-8"))
    res = enc.encode_batch(docs, ordinary=True)
    assert (res.status == 0).all()
    doc_off = np.zeros(len(docs) + 1, dtype=np.int64)
    np.cumsum([len(d) for d in docs], out=doc_off[1:])
    text = np.frombuffer(b"".join(docs), dtype=np.uint8)
    exp_tok, exp_off = o.encode_batch(text, doc_off, threads=8)
    assert np.array_equal(res.tok_off, exp_off) and np.array_equal(res.tokens, exp_tok)
    got = enc.decode_batch([res.doc(d).tolist() for d in range(0, 2000)])
    assert got == docs[:2000]


def test_batch_reuse_shorter_batch_after_longer(jt):
    """One batch object, a long batch and then shorter ones whose length lands just below a multiple of the split
    kernel's span (31,744 bytes per workgroup, 3,968 per wave; 15,872 per workgroup in round 1: the mask words after the end are then not rewritten): nothing of the earlier batch may
    leak into the later results.  Found by the randomized soak run (tools/soak_check.py)."""
    enc = jt.get_encoding("r50k_base")
    o = oracle_lib.get("r50k_base")
    b = enc.new_batch()
    long_text = np.frombuffer(("a b\r\nc 1 " * 20000).encode(), dtype=np.uint8)
    b.encode_host(long_text, np.array([0, len(long_text)], dtype=np.int64), ordinary=True)
    for n in (15872 * 2 - 2, 15872 * 2 - 1, 15872 * 3 - 64, 15872 - 130, 15872 * 4 - 127, 31744 * 2 - 1, 31744 * 3 - 64, 3968 * 5 - 3,
              2047, 2048, 4095):
        body = ("日" * (n // 3))[: n // 3]
        doc = (body.encode() + b"\n" * n)[:n]
        text = np.frombuffer(doc, dtype=np.uint8)
        b.encode_host(text, np.array([0, n], dtype=np.int64), ordinary=True)
        res = b.fetch()
        assert res.tokens.tolist() == o.encode_ordinary(doc), n
        # refill the scratch with a longer batch again
        b.encode_host(long_text, np.array([0, len(long_text)], dtype=np.int64), ordinary=True)
    b.close()


def test_count_tokens_batch(jt):
    """JTK_ENCODE_COUNT_ONLY: Encoding.countTokens()/countTokensOrdinary() for a whole batch == len(encode()) of the
    oracle (GptBytePairEncoding.java:122-129), incl. multi-token pieces, long pieces and empty documents; no ids exist."""
    from jtokkit_amd import corpus
    enc = jt.get_encoding("cl100k_base")
    o = oracle_lib.get("cl100k_base")
    rows = golden_util.load_rows("cl100k_base")
    texts = [r[0] for r in rows] + ["", "a" * 5000, "日本語" * 700, " \n" * 300]
    assert enc.count_tokens_batch(texts) == [len(o.encode(t)) for t in texts]
    text, doc_off = corpus.mixed(400, seed=9)
    b = enc.new_batch()
    b.encode_host(text, doc_off, ordinary=True, count_only=True)
    counts, status = b.fetch_counts()
    exp_tok, exp_off = o.encode_batch(text, doc_off, threads=8)
    assert (status == 0).all() and np.array_equal(counts, np.diff(exp_off))
    with pytest.raises(jt.EncodingError):
        b.fetch()
    tp, op, sp = b.device_result()
    assert not tp and op and sp
    with pytest.raises(jt.UnsupportedOperationError):
        enc.count_tokens_batch(["a <|endoftext|> b"])
    assert enc.count_tokens_batch(["a <|endoftext|> b"], ordinary=True) == [len(o.encode_ordinary("a <|endoftext|> b"))]
    b.close()


def test_two_batches_in_flight(jt):
    """What bench.py does by default: two batch objects of one encoding, each on its own stream, encodes enqueued back to
    back without waiting (sync=False) on different inputs; both results must be exact.  Repeated so that the two
    pipelines overlap in different phases."""
    import torch
    from jtokkit_amd import corpus
    enc = jt.get_encoding("cl100k_base")
    o = oracle_lib.get("cl100k_base")
    dev = torch.device("cuda:0")
    inputs = [corpus.english(3000, seed=71), corpus.mixed(900, seed=72)]
    expected = [o.encode_batch(t, off, threads=8) for t, off in inputs]
    d_in = [(torch.from_numpy(t).to(dev), torch.from_numpy(off).to(dev)) for t, off in inputs]
    torch.cuda.synchronize()
    batches = [enc.new_batch(), enc.new_batch()]
    for rep in range(6):
        order = (0, 1) if rep % 2 == 0 else (1, 0)
        for i in order:
            t, off = inputs[i]
            batches[i].encode_device(d_in[i][0].data_ptr(), d_in[i][1].data_ptr(), len(off) - 1, len(t), ordinary=True, sync=False)
        for i in (0, 1):
            res = batches[i].fetch()
            assert np.array_equal(res.tokens, expected[i][0]) and np.array_equal(res.tok_off, expected[i][1]), (rep, i)
    for b in batches:
        b.close()


def test_async_encode_with_giant_piece_is_complete(jt):
    """A non-synchronising device encode (n_tokens == NULL) followed only by a stream sync and reads through
    jtk_batch_device_result: pieces above 8 KiB (merged by a whole workgroup in the last phase of the merge kernel)
    must already be in the result -- no host-driven second phase."""
    import torch
    enc = jt.get_encoding("cl100k_base")
    o = oracle_lib.get("cl100k_base")
    docs = [b"before ", b"a" * 10240, b" between", b"=" * 9001 + b" x", b"after"]
    text = np.frombuffer(b"".join(docs), dtype=np.uint8)
    doc_off = np.zeros(len(docs) + 1, dtype=np.int64)
    np.cumsum([len(d) for d in docs], out=doc_off[1:])
    dev = torch.device("cuda:0")
    d_text, d_off = torch.from_numpy(text.copy()).to(dev), torch.from_numpy(doc_off).to(dev)
    torch.cuda.synchronize()
    b = enc.new_batch()
    st = torch.cuda.ExternalStream(b.stream(), device=dev)
    b.encode_device(d_text.data_ptr(), d_off.data_ptr(), len(docs), len(text), ordinary=True, sync=False)
    st.synchronize()
    tp, op, sp = b.device_result()

    class _Dev:
        def __init__(self, ptr, n, typestr):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}
    off = torch.as_tensor(_Dev(op, len(docs) + 1, "<i8"), device=dev).cpu().numpy()
    status = torch.as_tensor(_Dev(sp, len(docs), "<i4"), device=dev).cpu().numpy()
    toks = torch.as_tensor(_Dev(tp, int(off[-1]), "<i4"), device=dev).cpu().numpy()
    assert (status == 0).all()
    for d, doc in enumerate(docs):
        assert toks[off[d]:off[d + 1]].tolist() == o.encode_ordinary(doc), d
    b.close()


def test_device_doc_offsets_are_validated(jt):
    """Offsets handed over in device memory are caller memory: decreasing or out-of-range ones are reported as the
    batch's worst status (JTK_ERR_INVALID_ARGUMENT), not followed."""
    import torch
    enc = jt.get_encoding("cl100k_base")
    dev = torch.device("cuda:0")
    text = np.frombuffer(b"hello world, hello again and again", dtype=np.uint8)
    d_text = torch.from_numpy(text.copy()).to(dev)
    b = enc.new_batch()
    for bad in ([0, 20, 10, len(text)], [0, 5, 10**9, len(text)], [0, -3, 9, len(text)], [1, 5, 9, len(text)]):
        d_off = torch.tensor(bad, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        b.encode_device(d_text.data_ptr(), d_off.data_ptr(), 3, len(text), ordinary=True)
        assert b.result()[2] == -1, bad
    d_off = torch.tensor([0, 5, 9, len(text)], dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    b.encode_device(d_text.data_ptr(), d_off.data_ptr(), 3, len(text), ordinary=True)
    assert b.result()[2] == 0
    b.close()


def test_create_destroy_loop_does_not_leak(jt):
    """jtk_encoding_create/destroy and jtk_batch_create/destroy give back all device memory (decode tables included)."""
    import torch
    from jtokkit_amd import registry
    enc0 = jt.get_encoding("cl100k_base")          # keeps the HIP context and the cached encoding alive
    enc0.encode("warm up")
    torch.cuda.synchronize()

    def one():
        e = registry.new_encoding("r50k_base", device=0)
        bt = e.new_batch()
        bt.encode_host(np.frombuffer(b"some text to encode " * 500, dtype=np.uint8), np.array([0, 10000], dtype=np.int64))
        assert e.decode_batch([[31373, 995]]) == [b"hello world"]
        bt.close()
        e.close()
    one()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for _ in range(8):
        one()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    assert free0 - free1 < (8 << 20), "device memory shrank by %d bytes over 8 create/destroy rounds" % (free0 - free1)


@pytest.mark.parametrize("in_flight", [1, 2, 3])
def test_chunked_batches_equal_oracle(jt, in_flight):
    """A batch larger than one chunk is cut into runs of whole documents that flow through the batch's scratch sets on their own
    streams (device and host entry points, results on the device or streamed to pinned host memory).  With a tiny chunk size a
    small corpus makes dozens of chunks: ragged documents, empty documents at chunk boundaries, documents larger than a chunk,
    special-token status, every encoding's split pattern -- all bit-exact vs the oracle."""
    import torch
    from jtokkit_amd import corpus, _native as N
    dev = torch.device("cuda:0")
    for name, (text, doc_off) in (("cl100k_base", corpus.mixed(700, seed=31)), ("r50k_base", corpus.english(4000, seed=32))):
        enc = jt.get_encoding(name)
        o = oracle_lib.get(name)
        # extra documents: empty ones, one much larger than a chunk, one with a special-token literal
        docs = [text[doc_off[d]:doc_off[d + 1]].tobytes() for d in range(len(doc_off) - 1)]
        big = b" ".join(docs[:300])[:700000]
        while big and (big[-1] & 0xC0) == 0x80:
            big = big[:-1]
        if big and big[-1] >= 0xC0:
            big = big[:-1]
        docs = docs[:100] + [b"", b""] + [big] + [b""] + docs[100:400] + [b"x <|endoftext|> y"] + docs[400:]
        text2 = np.frombuffer(b"".join(docs), dtype=np.uint8).copy()
        off2 = np.zeros(len(docs) + 1, dtype=np.int64)
        np.cumsum([len(d) for d in docs], out=off2[1:])
        exp_tok, exp_off = o.encode_batch(text2, off2, threads=8)
        sp = docs.index(b"x <|endoftext|> y")
        b = enc.new_batch()
        b.set_option(N.JTK_OPT_CHUNK_BYTES, 64 * 1024)
        b.set_option(N.JTK_OPT_HOST_CHUNK_BYTES, 96 * 1024)
        b.set_option(N.JTK_OPT_CHUNKS_IN_FLIGHT, in_flight)
        # device entry point
        d_text, d_off = torch.from_numpy(text2).to(dev), torch.from_numpy(off2).to(dev)
        torch.cuda.synchronize()
        for rep in range(2):
            b.encode_device(d_text.data_ptr(), d_off.data_ptr(), len(docs), len(text2), ordinary=False, sync=(rep == 0))
            res = b.fetch()
            assert np.array_equal(res.tok_off, exp_off) and np.array_equal(res.tokens, exp_tok), (name, "device", rep)
            assert res.status[sp] == -2 and (np.delete(res.status, sp) == 0).all()
            assert b.result()[2] == -2
        # host entry point, result on the device
        b.encode_host(text2, off2, ordinary=True)
        res = b.fetch()
        assert np.array_equal(res.tok_off, exp_off) and np.array_equal(res.tokens, exp_tok), (name, "host")
        assert (res.status == 0).all()
        # host entry point from pinned memory, result streamed to pinned host memory
        hb = jt.HostBuffer(len(text2))
        hb.array[:] = text2
        nt = b.encode_host(hb.array, off2, ordinary=True, to_host=True)
        res = b.host_result()
        assert nt == len(exp_tok) and np.array_equal(res.tok_off, exp_off) and np.array_equal(res.tokens, exp_tok), (name, "to_host")
        assert (res.status == 0).all()
        # maxTokens for the whole (chunked) batch
        kept, flag = b.truncate(7)
        for d in (0, 50, 102, 103, len(docs) - 1):
            e, tr = o.encode_ordinary(docs[d], 7)
            assert res.tokens[res.tok_off[d]:res.tok_off[d] + kept[d]].tolist() == e and bool(flag[d]) == tr, (name, d)
        hb.close()
        b.close()


def test_comm_stitch_one_rank_rehearsal(jt):
    """The N > 1 path behind the C ABI, rehearsed with a one-rank RCCL communicator on the one GPU of this box:
    jtk_comm_unique_id / jtk_comm_create (ncclCommInitRank), then after a non-synchronising encode the stitch on the batch's
    stream (ncclAllGather of the shard total, base, global offsets)."""
    import torch
    from jtokkit_amd import corpus, sharding
    enc = jt.get_encoding("cl100k_base")
    o = oracle_lib.get("cl100k_base")
    text, doc_off = corpus.english(3000, seed=41)
    bounds = sharding.shard_plan(doc_off, 2)
    my_text, my_off, first = sharding.local_shard(text, doc_off, 1, 2)          # the second of two shards, as rank 1 would hold it
    assert first == bounds[1]
    dev = torch.device("cuda:0")
    d_text, d_off = torch.from_numpy(np.ascontiguousarray(my_text)).to(dev), torch.from_numpy(np.ascontiguousarray(my_off)).to(dev)
    torch.cuda.synchronize()
    comm = sharding.Comm(sharding.Comm.unique_id(), 1, 0, 0)
    b = enc.new_batch()
    n_docs = len(my_off) - 1
    g_off = torch.empty(n_docs + 1, dtype=torch.int64, device=dev)
    b.encode_device(d_text.data_ptr(), d_off.data_ptr(), n_docs, len(my_text), ordinary=True, sync=False)
    _, off_ptr, _ = b.device_result()
    comm.stitch(off_ptr, n_docs, g_off.data_ptr(), b.stream())
    totals, base = comm.fetch(b.stream())
    exp_tok, exp_off = o.encode_batch(np.ascontiguousarray(my_text), np.ascontiguousarray(my_off), threads=8)
    assert totals.tolist() == [len(exp_tok)] and base == 0
    assert np.array_equal(g_off.cpu().numpy(), exp_off)
    comm.close()
    b.close()


def test_service_submit_wait_pipelines(jt):
    """jtk_service_submit / jtk_service_wait from several producer threads, each keeping a few hundred documents in flight
    (the service's queue is sharded by producer thread; a device batch then holds documents of all of them): every ticket
    comes back with exactly its own document's tokens."""
    import ctypes as C
    import threading
    from jtokkit_amd import _native as N, corpus
    enc = jt.get_encoding("cl100k_base")
    o = oracle_lib.get("cl100k_base")
    text, doc_off = corpus.sentences(2400, seed=61)
    t2, o2 = corpus.mixed(400, mean_bytes=900, lo=64, hi=6000, seed=62)
    docs = [text[doc_off[d]:doc_off[d + 1]].tobytes() for d in range(len(doc_off) - 1)]
    docs += [t2[o2[d]:o2[d + 1]].tobytes() for d in range(len(o2) - 1)] + [b"", b"x"]
    want = [o.encode_ordinary(d) for d in docs]
    svc = enc._service()
    lib = N.lib()
    errors = []
    n_threads, window = 6, 300

    def producer(t):
        try:
            mine = list(range(t, len(docs), n_threads))
            outs = [np.empty(len(docs[i]) + 1, dtype=np.int32) for i in mine]
            tickets = [C.c_void_p() for _ in mine]

            def finish(k):
                nt, tr = C.c_int64(0), C.c_int(0)
                rc = lib.jtk_service_wait(svc, tickets[k], C.byref(nt), C.byref(tr))
                assert rc == 0, (rc, mine[k])
                assert outs[k][:nt.value].tolist() == want[mine[k]], mine[k]

            for k, i in enumerate(mine):
                rc = lib.jtk_service_submit(svc, docs[i], len(docs[i]), N.JTK_ENCODE_ORDINARY, -1, outs[k].ctypes.data, len(outs[k]),
                                            C.byref(tickets[k]))
                assert rc == 0, (rc, i)
                if k >= window:
                    finish(k - window)
            for k in range(max(0, len(mine) - window), len(mine)):
                finish(k)
        except BaseException as ex:          # noqa: BLE001 -- reported to the main thread
            errors.append((t, repr(ex)))

    th = [threading.Thread(target=producer, args=(t,)) for t in range(n_threads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:3]


def test_per_call_methods_from_many_threads(jt):
    """The reference's calling shape: Encoding.encode(String) per document from a pool of threads
    (benchmark/.../AbstractMultiThreadedBenchmark.java:35-45).  The per-call methods are thread-safe and coalesced into device
    batches by the encoding's jtk_service; every caller gets exactly its own document's tokens (incl. maxTokens, the
    special-token exception and empty strings)."""
    import threading
    from jtokkit_amd import corpus
    enc = jt.get_encoding("cl100k_base")
    o = oracle_lib.get("cl100k_base")
    text, doc_off = corpus.sentences(1200, seed=51)
    t2, o2 = corpus.mixed(300, mean_bytes=700, lo=64, hi=4096, seed=52)
    docs = [text[doc_off[d]:doc_off[d + 1]].tobytes().decode() for d in range(len(doc_off) - 1)]
    docs += [t2[o2[d]:o2[d + 1]].tobytes().decode() for d in range(len(o2) - 1)]
    docs += ["", "x", "hello <|endoftext|> world"]
    errors = []
    n_threads = 12

    def worker(t):
        try:
            for i in range(t, len(docs), n_threads):
                d = docs[i]
                if "<|endoftext|>" in d:
                    with pytest.raises(jt.UnsupportedOperationError):
                        enc.encode(d)
                    assert enc.encode_ordinary(d) == o.encode_ordinary(d)
                    continue
                assert enc.encode(d) == o.encode(d), i
                if i % 5 == 0:
                    r = enc.encode(d, 9)
                    e, tr = o.encode(d, 9)
                    assert r.get_tokens() == e and r.is_truncated() == tr, i
                if i % 7 == 0:
                    assert enc.count_tokens(d) == len(o.encode(d))
        except BaseException as ex:          # noqa: BLE001 -- reported to the main thread
            errors.append((t, repr(ex)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[0]


def test_custom_pattern_pieces_path(jt):
    """Custom split patterns (api/GptBytePairEncodingParams.java:36-46): the caller matches on the host, the device does the
    whole-piece shortcut, bytePairMerge and packing for the matches (jtk_batch_encode_pieces).  Patterns here leave gaps
    (unmatched text is skipped as matcher.find() does), make pieces of hundreds of bytes, and match at document edges;
    expected = the oracle's encodeOrdinaryInternal loop over the same matches."""
    import regex
    from jtokkit_amd import corpus, _native as N
    enc = jt.get_encoding("cl100k_base")
    o = oracle_lib.get("cl100k_base")
    text, doc_off = corpus.mixed(300, mean_bytes=900, lo=64, hi=8192, seed=61)
    docs = [text[doc_off[d]:doc_off[d + 1]].tobytes() for d in range(len(doc_off) - 1)] + [b"", b"   ", b"x", b"  lead and trail  "]
    text2 = np.frombuffer(b"".join(docs), dtype=np.uint8).copy()
    off2 = np.zeros(len(docs) + 1, dtype=np.int64)
    np.cumsum([len(d) for d in docs], out=off2[1:])
    for pat in (r"\w+|[^\w\s]+",                  # drops all whitespace: gaps everywhere
                r"[^\n]+",                        # whole lines: pieces of hundreds of bytes
                r"\p{L}{1,5}|\p{N}|\s+"):         # short letter chunks, single digits; punctuation is skipped
        cp = regex.compile(pat)
        henc = jt.HipEncoding.__new__(jt.HipEncoding)
        henc._host_pattern = cp
        pb, pe = jt.HipEncoding._match_on_host(henc, text2, off2)
        b = enc.new_batch()
        b.set_option(N.JTK_OPT_HOST_CHUNK_BYTES, 128 * 1024)          # several chunks
        nt = b.encode_pieces(text2, off2, pb, pe)
        res = b.fetch()
        assert (res.status == 0).all() and nt == len(res.tokens) == res.tok_off[-1]
        k = 0
        for d, doc in enumerate(docs):
            lo, hi = off2[d], off2[d + 1]
            k0 = k
            while k < len(pb) and pb[k] < hi:
                k += 1
            exp = o.encode_pieces(doc, pb[k0:k] - lo, pe[k0:k] - lo)
            assert res.doc(d).tolist() == exp, (pat, d)
        b.close()
    # encode() (not encodeOrdinary): the special-token check still applies
    b = enc.new_batch()
    t = np.frombuffer(b"ok doc|has <|endoftext|> inside", dtype=np.uint8)
    b.encode_pieces(t, np.array([0, 7, len(t)]), np.array([0, 3, 7, 11]), np.array([2, 6, 10, 24]), ordinary=False)
    assert b.fetch().status.tolist() == [0, -2]
    # malformed piece lists are refused
    with pytest.raises(jt.EncodingError):
        b.encode_pieces(t, np.array([0, 7, len(t)]), np.array([0, 5]), np.array([6, 9]))      # overlapping
    with pytest.raises(jt.EncodingError):
        b.encode_pieces(t, np.array([0, 7, len(t)]), np.array([5]), np.array([9]))            # crosses a document boundary
    b.close()


@pytest.mark.parametrize("kind", [0, 1])
def test_hand_made_rank_table_with_unreproducible_entries(jt, kind):
    """The reference accepts ANY rank map (api/GptBytePairEncodingParams.java:36-46, EncodingFactory.java:117-119) and looks every
    regex piece up whole before merging (GptBytePairEncoding.java:81-83).  A hand-made table may hold entries that merging
    their bytes cannot produce -- short ones (<= 16 bytes), long ones (more than 16, up to hundreds), and ones whose pieces
    would otherwise be cut inside.  Special tokens with arbitrary first bytes ride along.  GPU == oracle built from the same
    table."""
    import base64
    from jtokkit_amd import corpus
    ranks = _train_tiny_bpe(corpus.english(40, seed=5)[0].tobytes(), 400)
    extra = [b"zzzzqqqq", b"qzqzqz", b"abcdefghijklmnopqrstuvwxyz", b" internationalisation", b"x" * 40, b"0123456789" * 9,
             " мультибайтовыйтокен".encode(), b"ab" * 150, b"\n\n\n\n\n\n\n\n\n\n\n\n\n\n\n\n\n\n\n\n"]
    for e in extra:
        assert e not in ranks
        ranks[e] = len(ranks) + 7          # also leaves holes in the rank range
    specials = {"[[stop]]": 100000, "~end~": 100001, "<|x|>": 100002}
    enc = jt.new_custom_encoding("handmade_%d" % kind, kind, ranks, specials)
    data = b"\n".join(base64.b64encode(k) + b" " + str(v).encode() for k, v in sorted(ranks.items(), key=lambda kv: kv[1])) + b"\n"
    o = oracle_lib.OracleEncoding("handmade_%d" % kind, kind, data, specials)
    rng = random.Random(kind)
    texts = []
    for e in extra:
        s = e.decode("utf-8")
        texts += [s, s + s, "a " + s + " b", s + "s", "(" + s + ")", s[:-1], s[1:], " " + s, s + "\n" + s]
    texts += [rc.random_text(rng, rng.randint(0, 120)) for _ in range(300)]
    texts += [" ".join(rng.choice([e.decode("utf-8") for e in extra] + ["the", "of", "x", "zzzz", "qqqq"]) for _ in range(rng.randint(1, 40))) for _ in range(200)]
    _assert_batch_equals_oracle(enc, o, texts)
    # the whole entries really come out as ONE token where the pattern keeps them in one piece
    assert enc.encode_ordinary("abcdefghijklmnopqrstuvwxyz") == [ranks[b"abcdefghijklmnopqrstuvwxyz"]]
    assert enc.encode_ordinary("zzzzqqqq") == [ranks[b"zzzzqqqq"]]
    assert enc.encode_ordinary("ab" * 150) == [ranks[b"ab" * 150]]
    # special tokens that do not start with "<|"
    for t in ("say [[stop]] now", "the ~end~", "<|x|>"):
        with pytest.raises(jt.UnsupportedOperationError):
            enc.encode(t)
        assert enc.encode_ordinary(t) == o.encode_ordinary(t)
    assert enc.encode("[stop]] ~end <|x") == o.encode("[stop]] ~end <|x")
    assert enc.decode([100000, 100001]) == "[[stop]]~end~"
    enc.close()


def test_rank_map_without_all_single_bytes(jt):
    """The reference takes any rank map (api/GptBytePairEncodingParams.java:36-46); with single bytes missing it throws on the
    documents that need one (TokenEncoder.java:66-68: a piece whose merge leaves such a byte alone) and encodes the others.
    A trained table with eleven single bytes removed ('q', 'z', digits 7..9, newline, two UTF-8 lead bytes ...): per document the
    device gives JTK_ERR_UNENCODABLE exactly where the oracle built from the same map fails, and the oracle's tokens elsewhere;
    batch, count-only, maxTokens and per-call paths."""
    import base64
    from jtokkit_amd import corpus
    ranks = _train_tiny_bpe(corpus.english(40, seed=8)[0].tobytes() + "mañana 日本語 q z 789 qu iz".encode() * 20, 500)
    for b in (b"q", b"z", b"7", b"8", b"9", b"\n", b"\xc3", b"\xe6", b"Q", b"~", b"\x00"):
        ranks.pop(b, None)
    enc = jt.new_custom_encoding("partial_bytes", 1, ranks, {})
    data = b"\n".join(base64.b64encode(k) + b" " + str(v).encode() for k, v in sorted(ranks.items(), key=lambda kv: kv[1])) + b"\n"
    o = oracle_lib.OracleEncoding("partial_bytes", 1, data, {})
    rng = random.Random(9)
    words = ["the", "quick", "quiz", "zebra", "a", "of", "mañana", "日本語", "789", "1", "q", "z", "~", "Queen", "size", "\n", " ", "  ", "x", "iz", "qu", "."]
    texts = [" ".join(rng.choice(words) for _ in range(rng.randint(0, 30))) for _ in range(600)]
    texts += [rc.random_text(rng, rng.randint(0, 80)) for _ in range(300)] + ["", "q", "the quiz", "no bad letters here", "line\nbreak"]
    exp = []
    for t in texts:
        try:
            exp.append(o.encode_ordinary(t))
        except oracle_lib.OracleError:
            exp.append(None)
    assert sum(e is None for e in exp) > 100 and sum(e is not None for e in exp) > 100
    bs = [t.encode("utf-8") for t in texts]
    doc_off = np.zeros(len(bs) + 1, dtype=np.int64)
    np.cumsum([len(x) for x in bs], out=doc_off[1:])
    text = np.frombuffer(b"".join(bs), dtype=np.uint8)
    b = enc.new_batch()
    b.encode_host(text, doc_off, ordinary=True)
    res = b.fetch()
    for d, e in enumerate(exp):
        if e is None:
            assert res.status[d] == jt._native.JTK_ERR_UNENCODABLE, (d, texts[d])
        else:
            assert res.status[d] == 0 and res.doc(d).tolist() == e, (d, texts[d])
    b.encode_host(text, doc_off, ordinary=True, count_only=True)
    counts, status = b.fetch_counts()
    for d, e in enumerate(exp):
        assert (status[d] == jt._native.JTK_ERR_UNENCODABLE) == (e is None) and (e is None or counts[d] == len(e))
    toks, kept, flag, status = b.encode_max_tokens(text, doc_off, 1000, ordinary=True)
    for d, e in enumerate(exp):
        assert (status[d] == jt._native.JTK_ERR_UNENCODABLE) == (e is None) and (e is None or toks[d, :kept[d]].tolist() == e)
    b.close()
    for t, e in list(zip(texts, exp))[:60] + list(zip(texts, exp))[-5:]:
        if e is None:
            with pytest.raises(ValueError, match="Unknown token for encoding"):
                enc.encode_ordinary(t)
        else:
            assert enc.encode_ordinary(t) == e
    enc.close()


def test_many_and_long_special_tokens(jt):
    """GptBytePairEncodingParams accepts any special-token map (api/GptBytePairEncodingParams.java:36-46): 60 literals here,
    some far longer than a tile edge matters for (up to 300 bytes), with arbitrary first bytes.  encode() refuses exactly the
    documents that contain one (GptBytePairEncoding.java:52-56), in a batch, per call and with a token limit; the ids decode."""
    import base64
    from jtokkit_amd import corpus
    ranks = _train_tiny_bpe(corpus.english(30, seed=6)[0].tobytes(), 300)
    rng = random.Random(5)
    specials = {}
    for i in range(60):
        n = rng.choice([1, 2, 5, 9, 31, 32, 33, 64, 100, 300])
        lit = "".join(rng.choice("<|>[]~#@abcXYZ_0189é中") for _ in range(n)) + "#%d;" % i
        specials[lit] = 200000 + i
    enc = jt.new_custom_encoding("manyspecials", 1, ranks, specials)
    data = b"\n".join(base64.b64encode(k) + b" " + str(v).encode() for k, v in sorted(ranks.items(), key=lambda kv: kv[1])) + b"\n"
    o = oracle_lib.OracleEncoding("manyspecials", 1, data, specials)
    lits = list(specials)
    texts, has = [], []
    for k in range(400):
        body = rc.random_text(rng, rng.randint(0, 200))
        if k % 3 == 0:
            lit = rng.choice(lits)
            cut = rng.randint(0, len(body))
            texts.append(body[:cut] + lit + body[cut:]); has.append(True)
        elif k % 3 == 1:
            lit = rng.choice(lits)
            texts.append(body + lit[:-1]); has.append(any(l in body + lit[:-1] for l in lits))     # all but the last character
        else:
            texts.append(body); has.append(any(l in body for l in lits))
    b = enc.new_batch()
    text = np.frombuffer("".join(texts).encode("utf-8"), dtype=np.uint8)
    doc_off = np.zeros(len(texts) + 1, dtype=np.int64)
    np.cumsum([len(t.encode("utf-8")) for t in texts], out=doc_off[1:])
    b.encode_host(text, doc_off, ordinary=False)
    res = b.fetch()
    for d, (t, h) in enumerate(zip(texts, has)):
        assert (res.status[d] == jt._native.JTK_ERR_UNSUPPORTED_SPECIAL) == h, (d, t[:60])
        if not h:
            assert res.tokens[res.tok_off[d]:res.tok_off[d + 1]].tolist() == o.encode(t)
    toks, kept, flag, status = b.encode_max_tokens(text, doc_off, 5, ordinary=False)
    assert [(s == jt._native.JTK_ERR_UNSUPPORTED_SPECIAL) for s in status] == has
    b.close()
    for t, h in list(zip(texts, has))[:40]:
        if h:
            with pytest.raises(jt.UnsupportedOperationError):
                enc.encode(t)
        else:
            assert enc.encode(t) == o.encode(t)
        assert enc.encode_ordinary(t) == o.encode_ordinary(t)
    assert enc.decode([200000, 200059]) == lits[0] + lits[59]
    enc.close()


def test_service_tickets_polled_by_spinning(tmp_path):
    """The completion hand-off of the per-call service (jtk_service.cpp: one state word per ticket, published with an exchange):
    16 native threads, 200k documents, every ticket polled with jtk_service_done and collected -- and so freed -- the moment it
    reads done; every result against the oracle (tests/cpp/service_spin.cpp)."""
    import subprocess
    exe = os.path.join(str(tmp_path), "service_spin")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "cpp", "service_spin.cpp"), "-ldl"])
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    p = subprocess.run([exe, os.path.join(ROOT, "jtokkit_amd", "libjtokkit_amd.so"), os.path.join(ROOT, "oracle", "libjtk_oracle.so"),
                        os.path.join(ROOT, "jtokkit_amd", "data", "cl100k_base.tiktoken"), "16", "200000", "64"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0 and "spin ok: 200000 documents" in p.stdout, (p.returncode, p.stdout[-300:], p.stderr[-300:])


def test_bench_two_rank_rehearsal():
    """bench.py's N > 1 path (configs[3]: the corpus sharded by contiguous document ranges, one rank per process, the offset
    stitch after every step, max-over-ranks timing, per-rank verification against the oracle) launched the way the driver
    launches it -- two ranks by torch.distributed.run -- but with both ranks on this box's one GPU (JTK_BENCH_REHEARSAL=1: the
    totals travel over gloo instead of RCCL)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, JTK_BENCH_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--docs", "200000",
           "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert p.returncode == 0, (p.returncode, p.stdout[-500:], p.stderr[-1500:])
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["value"] > 0 and r["config"]["docs"] == 200000
    assert len(r["per_rank"]["MBps"]) == 2 and sum(r["per_rank"]["bytes"]) == r["config"]["bytes"]
    assert "== CPU oracle" in r["verified"]
