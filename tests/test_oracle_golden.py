"""Pins the CPU oracle (oracle/jtk_oracle.cpp) to the reference's own fixtures.

Mirrors reference lib/src/test/java/com/knuddels/jtokkit/reference/Cl100kBaseTestTest.java:21-111
(and R50kBaseTest / P50kBaseTest / P50kEditTest, identical modulo names) on the four golden CSVs,
plus the known-answer literals in api/Encoding.java:18-19,73-74,170-171, README.md:83-84 and
docs/docs/getting-started/usage.md:89-100.  CPU only.
"""
import random

import pytest

import golden_util
import oracle_lib
import regex_crosscheck as rc

NAMES = golden_util.ENCODING_NAMES


@pytest.fixture(scope="module", params=NAMES)
def enc_rows(request):
    return oracle_lib.get(request.param), golden_util.load_rows(request.param)


def test_fixture_shape():
    for name in NAMES:
        assert len(golden_util.load_rows(name)) == 423


def test_encodes_correctly(enc_rows):                       # Cl100kBaseTestTest.java:21-29
    enc, rows = enc_rows
    for inp, expected, _ in rows:
        assert enc.encode(inp) == expected, inp


def test_encodes_stable(enc_rows):                          # :33-37
    enc, rows = enc_rows
    for inp, _, _ in rows:
        assert enc.decode(enc.encode(inp)) == inp


def test_encodes_correctly_with_max_tokens(enc_rows):       # :41-52
    enc, rows = enc_rows
    for inp, expected, expected10 in rows:
        toks, truncated = enc.encode(inp, 10)
        assert toks == expected10, inp
        assert truncated == (len(expected) > len(expected10)), inp


def test_encodes_stable_with_max_tokens(enc_rows):          # :56-60
    enc, rows = enc_rows
    for inp, _, _ in rows:
        toks, _ = enc.encode(inp, 10)
        assert inp.startswith(enc.decode(toks))


def test_encode_ordinary_correct(enc_rows):                 # :64-88
    enc, rows = enc_rows
    for inp, expected, expected10 in rows:
        assert enc.encode_ordinary(inp) == expected
        toks, truncated = enc.encode_ordinary(inp, 10)
        assert toks == expected10
        assert truncated == (len(expected) > len(expected10))


def test_encode_ordinary_special_tokens_roundtrip():        # :105-111
    s = "Hello<|endoftext|>, <|fim_prefix|> <|fim_middle|> world <|fim_suffix|> ! <|endofprompt|>"
    for name in NAMES:
        enc = oracle_lib.get(name)
        assert enc.decode(enc.encode_ordinary(s)) == s


def test_known_answer_literals():
    enc = oracle_lib.get("cl100k_base")
    assert enc.encode("hello world") == [15339, 1917]                                   # api/Encoding.java:18-19
    assert enc.encode_ordinary("hello <|endoftext|> world") == [15339, 83739, 8862, 728, 428, 91, 29, 1917]
    assert list(enc.decode_bytes([15339, 1917])) == [104, 101, 108, 108, 111, 32, 119, 111, 114, 108, 100]
    assert enc.encode("This is a sample sentence.") == [2028, 374, 264, 6205, 11914, 13]  # README.md:83-84
    assert enc.encode("This is a sample sentence.", 3) == ([2028, 374, 264], True)      # usage.md:89-100
    assert enc.encode("I love \U0001f355", 4) == ([40, 3021], True)


def test_error_behaviour():
    enc = oracle_lib.get("cl100k_base")
    with pytest.raises(oracle_lib.OracleError) as e:                                    # GptBytePairEncoding.java:52-56
        enc.encode("hello <|endoftext|> world")
    assert e.value.code == oracle_lib.ERR_UNSUPPORTED_SPECIAL
    with pytest.raises(oracle_lib.OracleError) as e:                                    # :313
        enc.decode_bytes([100261])
    assert e.value.code == oracle_lib.ERR_UNKNOWN_TOKEN
    assert enc.encode(None) == []                                                       # :48-50
    assert enc.encode("") == []
    assert enc.encode("", 10) == ([], False)
    assert enc.encode("abc", 0) == ([], True)                                           # maxTokens <= 0


def test_shortcut_is_pure_optimisation_for_shipped_tables():
    """merge(T) == [rank(T)] for every table token: checked on a sample per table."""
    import base64
    import os
    for name in ("cl100k_base", "r50k_base", "p50k_base"):
        enc = oracle_lib.get(name)
        path = os.path.join(oracle_lib.DATA_DIR, oracle_lib.ENCODINGS[name]["file"])
        lines = open(path, "rb").read().split(b"\n")
        rng = random.Random(7)
        for line in rng.sample([l for l in lines if l], 3000):
            tok, rank = line.split()
            assert enc.merge_piece(base64.b64decode(tok)) == [int(rank)]


@pytest.mark.parametrize("name,kind", [("cl100k_base", 1), ("r50k_base", 0)])
def test_scanner_matches_general_regex_engine(name, kind):
    """Branches no reference fixture reaches (CR/LF, digit runs > 3, upper-case contractions, long s,
    multi-space runs, non-ASCII whitespace): oracle scanner vs the Python `regex` engine running the
    reference's pattern text.  Provisional -- `regex` is not the JVM."""
    enc = oracle_lib.get(name)
    rng = random.Random(20240 + kind)
    for _ in range(8000):
        s = rc.random_text(rng)
        assert enc.split(s) == rc.split(kind, s), repr(s)
    for inp, _, _ in golden_util.load_rows(name):
        assert enc.split(inp) == rc.split(kind, inp)


def test_split_examples_from_survey():
    cl = oracle_lib.get("cl100k_base")
    r5 = oracle_lib.get("r50k_base")
    assert cl.split("1234567 89") == [b"123", b"456", b"7", b" ", b"89"]
    assert r5.split("1234567 89") == [b"1234567", b" 89"]
    assert cl.split("x \n\n  y") == [b"x", b" \n\n", b" ", b" y"]
    assert cl.split("!!!\n\nx") == [b"!!!\n\n", b"x"]
    assert cl.split("foo.bar()") == [b"foo", b".bar", b"()"]
    assert cl.split("DON'T") == [b"DON", b"'T"]
    assert r5.split("DON'T") == [b"DON", b"'", b"T"]
    assert cl.split("a\tb") == [b"a", b"\tb"]
    assert r5.split("a\tb") == [b"a", b"\t", b"b"]
