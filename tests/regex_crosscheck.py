"""Independent cross-check of the split patterns with a general backtracking regex engine.

The reference's patterns (EncodingFactory.java:63,105) are handed to the Python `regex` module with
\\p{L}, \\p{N}, \\s spelled out as explicit code-point classes taken from the same Unicode 13.0 data
as the build (tools/gen_unicode_tables.py), so the comparison is about matching structure
(leftmost-first alternation, greedy backtracking, look-ahead), not Unicode-version drift.
`regex` is NOT the JVM: vectors it confirms beyond the reference's fixtures are "provisional".
"""
import unicodedata

import regex


def _classify(cp):
    if 0x9 <= cp <= 0xD or cp == 0x85:
        return "W"
    cat = unicodedata.category(chr(cp))
    if cat in ("Zs", "Zl", "Zp"):
        return "W"
    if cat[0] == "L":
        return "L"
    if cat[0] == "N":
        return "N"
    return "O"


def _class_body(want):
    parts = []
    start = None
    for cp in range(0x110000 + 1):
        c = _classify(cp) if cp < 0x110000 and not (0xD800 <= cp <= 0xDFFF) else None
        if c == want:
            if start is None:
                start = cp
        elif start is not None:
            parts.append("\\U%08X-\\U%08X" % (start, cp - 1))
            start = None
    return "".join(parts)


_L = _class_body("L")
_N = _class_body("N")
_W = _class_body("W")

R50K = ("'s|'t|'re|'ve|'m|'ll|'d| ?[" + _L + "]+| ?[" + _N + "]+| ?[^" + _W + _L + _N + "]+|[" + _W
        + "]+(?![^" + _W + "])|[" + _W + "]+")
CL100K = ("(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\\r\\n" + _L + _N + "]?[" + _L + "]+|[" + _N + "]{1,3}| ?[^"
          + _W + _L + _N + "]+[\\r\\n]*|[" + _W + "]*[\\r\\n]+|[" + _W + "]+(?![^" + _W + "])|[" + _W + "]+")

_compiled = {}


def pattern(kind):
    if kind not in _compiled:
        _compiled[kind] = regex.compile(CL100K if kind == 1 else R50K)
    return _compiled[kind]


def split(kind, text):
    return [m.group().encode("utf-8") for m in pattern(kind).finditer(text)]


_CPS = [
    # letters: ASCII (contraction letters both cases), Latin-1/ext, long s, dotted/dotless i, Kelvin sign,
    # CJK, Hangul, Cyrillic, Greek, Arabic, Devanagari, astral letter, titlecase, modifier letter
    *map(ord, "abcdstrevmlSTREVMLD"), 0xE9, 0x17F, 0xDF, 0x130, 0x131, 0x212A, 0x4E2D, 0x6587, 0xD55C, 0x416,
    0x44F, 0x3B1, 0x627, 0x915, 0x1D49C, 0x1C5, 0x2B0,
    # numbers: Nd (ASCII, Arabic-Indic, Tamil, astral), No (superscript two, one half, circled), Nl (roman)
    *map(ord, "0123456789"), 0x663, 0xB2, 0xBD, 0x2167, 0xBE7, 0x1D7D8, 0x3280,
    # whitespace under UNICODE_CHARACTER_CLASS
    0x20, 0x20, 0x20, 0x09, 0x0A, 0x0D, 0xA0, 0x3000, 0x2028, 0x2029, 0x85, 0x0B, 0x0C, 0x1680, 0x2003, 0x202F, 0x205F,
    # other: punctuation, symbols, emoji, VS16, ZWJ, ZWSP, controls that are NOT \\s (1C, 1F, 00, 7F), BOM, U+180E
    *map(ord, "''!.,-_()\"=<|>"), 0x3002, 0x3001, 0x2603, 0xFE0F, 0x1F355, 0x1F469, 0x200D, 0x200B, 0x1C, 0x1F, 0x00,
    0xFEFF, 0x180E, 0x20AC, 0xA9, 0x300, 0xFFFD, 0x7F,
]
ALPHABET = [chr(c) for c in _CPS]
SNIPPETS = ["'s", "'t", "'re", "'ve", "'m", "'ll", "'d", "'S", "'T", "'RE", "'Ve", "'LL", "'\u017f", "'rE", " '",
            "\r\n", "\n\n", "  ", "   ", " \n", "\n ", "12345", "1 2", "a1", "1a", "!!", "!a", " !", "\t!", "don't",
            "I'm", "<|endoftext|>", "foo.bar()", "x \n\n  y", "\u3000\u3000a"]


def random_text(rng, max_len=24):
    n = rng.randint(0, max_len)
    out = []
    for _ in range(n):
        if rng.random() < 0.25:
            out.append(rng.choice(SNIPPETS))
        else:
            out.append(rng.choice(ALPHABET))
    return "".join(out)
