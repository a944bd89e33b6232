"""ctypes binding of the CPU oracle (oracle/libjtk_oracle.so) -- test infrastructure only.

The oracle restates GptBytePairEncoding.encode() (reference GptBytePairEncoding.java:71-103,
200-300); this module is what the tests use as the checker.  The product never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
DATA_DIR = os.path.join(ROOT, "jtokkit_amd", "data")

ERR_UNSUPPORTED_SPECIAL = -2
ERR_UNKNOWN_TOKEN = -3

# EncodingFactory.java:24-53 (special tokens), :63,77,91,105 (patterns), :64,78,92,106 (rank files)
ENCODINGS = {
    "r50k_base": dict(kind=0, file="r50k_base.tiktoken", specials={"<|endoftext|>": 50256}),
    "p50k_base": dict(kind=0, file="p50k_base.tiktoken", specials={"<|endoftext|>": 50256}),
    "p50k_edit": dict(kind=0, file="p50k_base.tiktoken",
                      specials={"<|endoftext|>": 50256, "<|fim_prefix|>": 50281,
                                "<|fim_middle|>": 50282, "<|fim_suffix|>": 50283}),
    "cl100k_base": dict(kind=1, file="cl100k_base.tiktoken",
                        specials={"<|endoftext|>": 100257, "<|fim_prefix|>": 100258,
                                  "<|fim_middle|>": 100259, "<|fim_suffix|>": 100260,
                                  "<|endofprompt|>": 100276}),
}

_lib = None


def build():
    so = os.path.join(ORACLE_DIR, "libjtk_oracle.so")
    src = os.path.join(ORACLE_DIR, "jtk_oracle.cpp")
    if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(so) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.jtko_create.restype = C.c_void_p
        L.jtko_create.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_size_t, C.c_char_p,
                                  C.POINTER(C.c_int), C.c_int]
        L.jtko_destroy.argtypes = [C.c_void_p]
        L.jtko_vocab_size.restype = C.c_long
        L.jtko_vocab_size.argtypes = [C.c_void_p]
        L.jtko_encode.restype = C.c_long
        L.jtko_encode.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_long,
                                  C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
        L.jtko_split.restype = C.c_long
        L.jtko_split.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.jtko_merge_piece.restype = C.c_long
        L.jtko_merge_piece.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.jtko_encode_pieces.restype = C.c_long
        L.jtko_encode_pieces.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_size_t]
        L.jtko_decode.restype = C.c_long
        L.jtko_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.jtko_encode_batch.restype = C.c_long
        L.jtko_encode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
        L.jtko_unicode_version.restype = C.c_char_p
        _lib = L
    return _lib


class OracleError(Exception):
    def __init__(self, code):
        super().__init__("oracle error %d" % code)
        self.code = code


class OracleEncoding:
    """Mirrors the reference's Encoding interface (api/Encoding.java) on the CPU oracle."""

    def __init__(self, name, kind=None, data=None, specials=None):
        """A predefined encoding by name, or (custom) any rank file bytes with one of the two patterns."""
        self.name = name
        if data is None:
            cfg = ENCODINGS[name]
            kind, specials = cfg["kind"], cfg["specials"]
            with open(os.path.join(DATA_DIR, cfg["file"]), "rb") as f:
                data = f.read()
        sp = b"".join(k.encode() + b"\0" for k in specials)
        ids = (C.c_int * max(len(specials), 1))(*specials.values())
        self._h = lib().jtko_create(name.encode(), kind, data, len(data), sp, ids, len(specials))
        if not self._h:
            raise RuntimeError("oracle: could not load the rank table of " + name)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().jtko_destroy(self._h)
            self._h = None

    def _encode(self, text, ordinary, max_tokens):
        if text is None:
            return [], False
        b = text if isinstance(text, bytes) else text.encode("utf-8")
        cap = len(b) + 1
        out = np.empty(cap, dtype=np.int32)
        trunc = C.c_int(0)
        n = lib().jtko_encode(self._h, b, len(b), 1 if ordinary else 0,
                              -1 if max_tokens is None else max_tokens,
                              out.ctypes.data, cap, C.byref(trunc))
        if n < 0:
            raise OracleError(n)
        return out[:n].tolist(), bool(trunc.value)

    def encode(self, text, max_tokens=None):
        toks, tr = self._encode(text, False, max_tokens)
        return toks if max_tokens is None else (toks, tr)

    def encode_ordinary(self, text, max_tokens=None):
        toks, tr = self._encode(text, True, max_tokens)
        return toks if max_tokens is None else (toks, tr)

    def count_tokens(self, text):
        return len(self.encode(text))

    def split(self, text):
        b = text if isinstance(text, bytes) else text.encode("utf-8")
        ends = np.empty(len(b) + 1, dtype=np.int64)
        n = lib().jtko_split(self._h, b, len(b), ends.ctypes.data, len(ends))
        if n < 0:
            raise OracleError(n)
        e = [0] + ends[:n].tolist()
        return [b[e[i]:e[i + 1]] for i in range(n)]

    def merge_piece(self, piece):
        out = np.empty(len(piece) + 1, dtype=np.int32)
        n = lib().jtko_merge_piece(self._h, piece, len(piece), out.ctypes.data, len(out))
        if n < 0:
            raise OracleError(n)
        return out[:n].tolist()

    def encode_pieces(self, text, begins, ends):
        """The matches [begins[i], ends[i]) of a caller-supplied pattern over `text` (bytes) -> token list."""
        b = np.ascontiguousarray(begins, dtype=np.int64)
        e = np.ascontiguousarray(ends, dtype=np.int64)
        out = np.empty(len(text) + 1, dtype=np.int32)
        n = lib().jtko_encode_pieces(self._h, text, b.ctypes.data, e.ctypes.data, len(b), out.ctypes.data, len(out))
        if n < 0:
            raise OracleError(n)
        return out[:n].tolist()

    def decode_bytes(self, tokens):
        ids = np.asarray(tokens, dtype=np.int32)
        n = lib().jtko_decode(self._h, ids.ctypes.data, len(ids), None, 0)
        if n < 0:
            raise OracleError(n)
        out = np.empty(max(n, 1), dtype=np.uint8)
        lib().jtko_decode(self._h, ids.ctypes.data, len(ids), out.ctypes.data, n)
        return out[:n].tobytes()

    def decode(self, tokens):
        return self.decode_bytes(tokens).decode("utf-8", errors="replace")

    def encode_batch(self, text_u8, doc_off, threads=1, ordinary=True, want_tokens=True):
        """text_u8: np.uint8[n_bytes]; doc_off: np.int64[n_docs+1] -> (tokens int32[], tok_off int64[n+1])."""
        text_u8 = np.ascontiguousarray(text_u8, dtype=np.uint8)
        doc_off = np.ascontiguousarray(doc_off, dtype=np.int64)
        n_docs = len(doc_off) - 1
        counts = np.zeros(max(n_docs, 1), dtype=np.int32)
        total = lib().jtko_encode_batch(self._h, text_u8.ctypes.data, doc_off.ctypes.data, n_docs,
                                        1 if ordinary else 0, threads, counts.ctypes.data, None, None)
        if total < 0:
            raise OracleError(total)
        tok_off = np.zeros(n_docs + 1, dtype=np.int64)
        np.cumsum(counts[:n_docs], out=tok_off[1:])
        if not want_tokens:
            return None, tok_off
        tokens = np.empty(max(int(total), 1), dtype=np.int32)
        lib().jtko_encode_batch(self._h, text_u8.ctypes.data, doc_off.ctypes.data, n_docs,
                                1 if ordinary else 0, threads, None, tokens.ctypes.data, tok_off.ctypes.data)
        return tokens[:total], tok_off


_cache = {}


def get(name):
    if name not in _cache:
        _cache[name] = OracleEncoding(name)
    return _cache[name]
