"""Host-side mirror of the reference's Encoding interface on top of the C ABI.

Mirrors `com.knuddels.jtokkit.api.Encoding` (reference api/Encoding.java:29-189) method for method
so that parity tests read like the reference's own (reference/Cl100kBaseTestTest.java), plus the
batch entry points that are the reason this library exists.  The JVM is absent from the build image,
so this Python class (and the C++ header jtokkit_amd/csrc/jtk_encoding.hpp) stand where the Java
`HipEncoding implements Encoding` of INTEGRATION.md would.
"""
import ctypes as C
import threading

import numpy as np

from . import _native as N


class UnsupportedOperationError(Exception):
    """java.lang.UnsupportedOperationException (GptBytePairEncoding.java:54)."""


class EncodingError(Exception):
    def __init__(self, code, msg):
        super().__init__("jtokkit_amd error %d: %s" % (code, msg))
        self.code = code


def _check(rc):
    if rc == N.JTK_OK:
        return
    msg = N.last_error()
    if rc == N.JTK_ERR_UNSUPPORTED_SPECIAL:
        raise UnsupportedOperationError("Encoding special tokens is not supported yet.")
    if rc == N.JTK_ERR_UNKNOWN_TOKEN:
        raise ValueError("Unknown token for decoding: " + msg)     # IllegalArgumentException
    if rc == N.JTK_ERR_UNENCODABLE:
        raise ValueError("Unknown token for encoding: " + msg)     # IllegalArgumentException (TokenEncoder.java:66-68)
    if rc == N.JTK_ERR_BAD_RANK_FILE:
        raise RuntimeError(msg)                                    # IllegalStateException
    raise EncodingError(rc, msg)


class EncodingResult:
    """api/EncodingResult.java"""

    def __init__(self, tokens, truncated):
        self.tokens = tokens
        self.truncated = truncated

    def get_tokens(self):
        return self.tokens

    def is_truncated(self):
        return self.truncated

    getTokens = get_tokens
    isTruncated = is_truncated

    def __repr__(self):
        return "EncodingResult{tokens=%r, truncated=%s}" % (self.tokens, str(self.truncated).lower())


class BatchResult:
    """Packed result of a batch encode: tokens[int32], tok_off[int64, n_docs+1], status[int32, n_docs]."""

    def __init__(self, tokens, tok_off, status):
        self.tokens = tokens
        self.tok_off = tok_off
        self.status = status

    def doc(self, d):
        return self.tokens[self.tok_off[d]:self.tok_off[d + 1]]

    def __len__(self):
        return len(self.tok_off) - 1


class HostBuffer:
    """Page-locked host memory (jtk_host_alloc) as a numpy array: input buffers the device reads by DMA."""

    def __init__(self, n_bytes):
        p = C.c_void_p()
        _check(N.lib().jtk_host_alloc(int(n_bytes), C.byref(p)))
        self._p = p
        self.array = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(int(n_bytes), 1),))[:int(n_bytes)]

    def close(self):
        if getattr(self, "_p", None):
            self.array = None
            N.lib().jtk_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:          # (interpreter shutdown: the module's globals may be gone)
            pass


class Batch:
    """One caller thread's stream + device scratch (jtk_batch)."""

    def __init__(self, encoding):
        self.encoding = encoding
        h = C.c_void_p()
        _check(N.lib().jtk_batch_create(encoding._h, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            N.lib().jtk_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:          # (interpreter shutdown: the module's globals may be gone)
            pass

    def set_option(self, option, value):
        """jtk_batch_set_option: N.JTK_OPT_CHUNK_BYTES, N.JTK_OPT_CHUNKS_IN_FLIGHT."""
        _check(N.lib().jtk_batch_set_option(self._h, int(option), int(value)))

    def encode_host(self, text_u8, doc_off, ordinary=False, validate=False, count_only=False, to_host=False):
        """Host buffers in (numpy arrays, or anything with .ctypes.data such as a pinned HostBuffer view).  to_host: the result
        is streamed to the batch's pinned host memory while later chunks are encoded (read it with host_result())."""
        text_u8 = np.ascontiguousarray(text_u8, dtype=np.uint8)
        doc_off = np.ascontiguousarray(doc_off, dtype=np.int64)
        nt = C.c_int64(0)
        flags = ((N.JTK_ENCODE_ORDINARY if ordinary else 0) | (N.JTK_ENCODE_VALIDATE_UTF8 if validate else 0)
                 | (N.JTK_ENCODE_COUNT_ONLY if count_only else 0) | (N.JTK_ENCODE_TO_HOST if to_host else 0))
        self._count_only = count_only
        _check(N.lib().jtk_batch_encode(self._h, text_u8.ctypes.data, doc_off.ctypes.data, len(doc_off) - 1,
                                        flags, C.byref(nt)))
        return nt.value

    def encode_pieces(self, text_u8, doc_off, piece_begin, piece_end, ordinary=True, to_host=False):
        """jtk_batch_encode_pieces: the caller's own pattern has been matched on the host; piece i is
        text[piece_begin[i]:piece_end[i]] (positions in the whole batch).  Returns the token total."""
        text_u8 = np.ascontiguousarray(text_u8, dtype=np.uint8)
        doc_off = np.ascontiguousarray(doc_off, dtype=np.int64)
        pb = np.ascontiguousarray(piece_begin, dtype=np.int64)
        pe = np.ascontiguousarray(piece_end, dtype=np.int64)
        nt = C.c_int64(0)
        flags = (N.JTK_ENCODE_ORDINARY if ordinary else 0) | (N.JTK_ENCODE_TO_HOST if to_host else 0)
        self._count_only = False
        _check(N.lib().jtk_batch_encode_pieces(self._h, text_u8.ctypes.data, doc_off.ctypes.data, len(doc_off) - 1,
                                               pb.ctypes.data, pe.ctypes.data, len(pb), flags, C.byref(nt)))
        return nt.value

    def host_result(self):
        """After encode_host(to_host=True): zero-copy numpy views of the batch's pinned result buffers (valid until the
        next encode on this batch)."""
        nt, nd, _ = self.result()
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(N.lib().jtk_batch_host_result(self._h, C.byref(a), C.byref(b), C.byref(c)))

        def view(ptr, n, ctype, dtype):
            if not ptr or n == 0:
                return np.zeros(0, dtype=dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,))
        tokens = view(a.value, nt, C.c_int32, np.int32)
        return BatchResult(tokens, view(b.value, nd + 1, C.c_int64, np.int64), view(c.value, nd, C.c_int32, np.int32))

    def encode_device(self, d_text_ptr, d_doc_off_ptr, n_docs, n_bytes, ordinary=False, stream=None, sync=True):
        nt = C.c_int64(0)
        _check(N.lib().jtk_batch_encode_device(self._h, d_text_ptr, d_doc_off_ptr, n_docs, n_bytes,
                                               N.JTK_ENCODE_ORDINARY if ordinary else 0, stream,
                                               C.byref(nt) if sync else None))
        return nt.value if sync else None

    def stream(self):
        """The batch's own HIP stream handle (int)."""
        return N.lib().jtk_batch_stream(self._h)

    def result(self):
        nt, nd, ws = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        _check(N.lib().jtk_batch_result(self._h, C.byref(nt), C.byref(nd), C.byref(ws)))
        return nt.value, nd.value, ws.value

    def fetch_counts(self):
        """Token count per document and status (works after any encode; the only fetch after a count-only one)."""
        _, nd, _ = self.result()
        tok_off = np.empty(nd + 1, dtype=np.int64)
        status = np.zeros(max(nd, 1), dtype=np.int32)
        _check(N.lib().jtk_batch_fetch(self._h, None, 0, tok_off.ctypes.data, status.ctypes.data))
        return np.diff(tok_off), status[:nd]

    def fetch(self):
        nt, nd, _ = self.result()
        tokens = np.empty(max(nt, 1), dtype=np.int32)
        tok_off = np.empty(nd + 1, dtype=np.int64)
        status = np.zeros(max(nd, 1), dtype=np.int32)
        _check(N.lib().jtk_batch_fetch(self._h, tokens.ctypes.data, nt, tok_off.ctypes.data, status.ctypes.data))
        return BatchResult(tokens[:nt], tok_off, status[:nd])

    def device_result(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(N.lib().jtk_batch_device_result(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # ---- maxTokens for the whole batch (device) ---------------------------------------------------------
    def encode_max_tokens(self, text_u8, doc_off, max_tokens, ordinary=False):
        """jtk_batch_encode_max_tokens: (tokens [n_docs, max_tokens], kept [n_docs], truncated [n_docs], status [n_docs])."""
        text_u8 = np.ascontiguousarray(text_u8, dtype=np.uint8)
        doc_off = np.ascontiguousarray(doc_off, dtype=np.int64)
        nd, mt = len(doc_off) - 1, max(0, int(max_tokens))
        toks = np.zeros((nd, mt), dtype=np.int32)
        kept = np.zeros(nd, dtype=np.int64)
        flag = np.zeros(nd, dtype=np.uint8)
        status = np.zeros(nd, dtype=np.int32)
        _check(N.lib().jtk_batch_encode_max_tokens(self._h, text_u8.ctypes.data, doc_off.ctypes.data, nd,
                                                   N.JTK_ENCODE_ORDINARY if ordinary else 0, mt, toks.ctypes.data,
                                                   kept.ctypes.data, flag.ctypes.data, status.ctypes.data))
        return toks, kept, flag, status

    def truncate(self, max_tokens):
        """Encoding.encode(text, maxTokens) for every document of the last encode -> (kept int64[n], truncated bool[n])."""
        _check(N.lib().jtk_batch_truncate(self._h, int(max_tokens)))
        _, nd, _ = self.result()
        kept = np.zeros(max(nd, 1), dtype=np.int64)
        flag = np.zeros(max(nd, 1), dtype=np.uint8)
        _check(N.lib().jtk_batch_fetch_truncated(self._h, kept.ctypes.data, flag.ctypes.data))
        return kept[:nd], flag[:nd].astype(bool)

    # ---- batch decode (device) -------------------------------------------------------------------------
    def decode_host(self, ids, seq_off):
        """ids int32[n], seq_off int64[n_seqs+1] -> total byte count (result stays on the device)."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)
        nb = C.c_int64(0)
        _check(N.lib().jtk_batch_decode(self._h, ids.ctypes.data, seq_off.ctypes.data, len(seq_off) - 1, C.byref(nb)))
        self._dec_shape = (nb.value, len(seq_off) - 1)
        return nb.value

    def decode_device(self, d_ids_ptr, d_seq_off_ptr, n_seqs, n_ids, stream=None):
        nb = C.c_int64(0)
        _check(N.lib().jtk_batch_decode_device(self._h, d_ids_ptr, d_seq_off_ptr, n_seqs, n_ids, stream, C.byref(nb)))
        self._dec_shape = (nb.value, n_seqs)
        return nb.value

    def decode_fetch(self):
        nb, ns = self._dec_shape
        out = np.empty(max(nb, 1), dtype=np.uint8)
        byte_off = np.empty(ns + 1, dtype=np.int64)
        status = np.zeros(max(ns, 1), dtype=np.int32)
        _check(N.lib().jtk_batch_decode_fetch(self._h, out.ctypes.data, nb, byte_off.ctypes.data, status.ctypes.data))
        return out[:nb], byte_off, status[:ns]

    def set_profiling(self, on=True):
        _check(N.lib().jtk_batch_set_profiling(self._h, 1 if on else 0))

    def kernel_times(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        n = C.c_int(0)
        _check(N.lib().jtk_batch_kernel_times(self._h, names, ms, 16, C.byref(n)))
        return {names[i].decode(): float(ms[i]) for i in range(n.value)}

    def encode_one(self, text, ordinary, max_tokens):
        if text is None:
            return [], False
        b = text if isinstance(text, (bytes, bytearray)) else text.encode("utf-8")
        cap = len(b) + 1
        out = np.empty(cap, dtype=np.int32)
        nt = C.c_int64(0)
        tr = C.c_int(0)
        _check(N.lib().jtk_encode(self._h, bytes(b), len(b), N.JTK_ENCODE_ORDINARY if ordinary else 0,
                                  -1 if max_tokens is None else max(0, int(max_tokens)), out.ctypes.data, cap,
                                  C.byref(nt), C.byref(tr)))
        return out[:nt.value].tolist(), bool(tr.value)


class HipEncoding:
    """GPU-backed `Encoding` (reference GptBytePairEncoding.java:18 is the class this replaces)."""

    def __init__(self, name, pattern_kind, tiktoken_bytes, special_tokens, device=0, host_pattern=None):
        """host_pattern: a compiled pattern object with finditer() over str (e.g. the `regex` module with the Java
        pattern's text) for encodings whose split pattern is neither of the two the device evaluates; the batch methods
        then match on the host and encode the matches through jtk_batch_encode_pieces."""
        self._host_pattern = host_pattern
        lits = [k.encode("utf-8") for k in special_tokens]
        arr = (C.c_char_p * max(len(lits), 1))(*lits)
        ids = (C.c_int32 * max(len(lits), 1))(*special_tokens.values())
        h = C.c_void_p()
        _check(N.lib().jtk_encoding_create(name.encode("utf-8"), pattern_kind, tiktoken_bytes, len(tiktoken_bytes),
                                           arr, ids, len(lits), device, C.byref(h)))
        self._h = h
        self._name = name
        self._batch = None
        self._svc = None
        self._svc_lock = threading.Lock()
        # close() against per-call methods in progress on other threads: calls are counted, close() waits for them to leave
        # and later ones raise instead of touching freed handles
        self._life = threading.Condition()
        self._calls = 0
        self._closed = False

    def close(self):
        life = getattr(self, "_life", None)
        if life is not None:
            with life:
                self._closed = True
                while self._calls:
                    life.wait()
        if self._batch is not None:
            self._batch.close()
            self._batch = None
        if getattr(self, "_svc", None):
            N.lib().jtk_service_destroy(self._svc)
            self._svc = None
        if getattr(self, "_h", None):
            N.lib().jtk_encoding_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def new_batch(self):
        return Batch(self)

    def _b(self):
        if self._batch is None:
            self._batch = Batch(self)
        return self._batch

    # ---- api/Encoding.java ---------------------------------------------------------------------------
    # "The encoding must be thread-safe" (api/EncodingRegistry.java:51,61): the per-call methods go through the
    # encoding's jtk_service, which coalesces concurrent callers into device batches.
    def _service(self):
        if self._svc is None:
            with self._svc_lock:
                if self._svc is None:
                    h = C.c_void_p()
                    _check(N.lib().jtk_service_create(self._h, 2, C.byref(h)))
                    self._svc = h
        return self._svc

    def _encode_one(self, text, ordinary, max_tokens):
        if text is None:
            return [], False
        b = text if isinstance(text, (bytes, bytearray)) else text.encode("utf-8")
        cap = len(b) + 1
        out = np.empty(cap, dtype=np.int32)
        nt = C.c_int64(0)
        tr = C.c_int(0)
        with self._life:
            if self._closed:
                raise RuntimeError("encoding is closed")
            self._calls += 1
        try:
            _check(N.lib().jtk_service_encode(self._service(), bytes(b), len(b), N.JTK_ENCODE_ORDINARY if ordinary else 0,
                                              -1 if max_tokens is None else max(0, int(max_tokens)), out.ctypes.data, cap,
                                              C.byref(nt), C.byref(tr)))
        finally:
            with self._life:
                self._calls -= 1
                if not self._calls:
                    self._life.notify_all()
        return out[:nt.value].tolist(), bool(tr.value)

    def encode(self, text, max_tokens=None):                       # Encoding.java:29,61
        toks, tr = self._encode_one(text, False, max_tokens)
        return toks if max_tokens is None else EncodingResult(toks, tr)

    def encode_ordinary(self, text, max_tokens=None):              # :80,107
        toks, tr = self._encode_one(text, True, max_tokens)
        return toks if max_tokens is None else EncodingResult(toks, tr)

    def count_tokens(self, text):                                  # :127
        return len(self.encode(text))

    def count_tokens_ordinary(self, text):                         # :147
        return len(self.encode_ordinary(text))

    def decode_bytes(self, tokens):                                # :181
        ids = np.ascontiguousarray(tokens, dtype=np.int32)
        n = C.c_int64(0)
        _check(N.lib().jtk_decode(self._h, ids.ctypes.data, len(ids), None, 0, C.byref(n)))
        out = np.empty(max(n.value, 1), dtype=np.uint8)
        _check(N.lib().jtk_decode(self._h, ids.ctypes.data, len(ids), out.ctypes.data, n.value, C.byref(n)))
        return out[:n.value].tobytes()

    def decode(self, tokens):                                      # :164  new String(bytes, UTF_8)
        return self.decode_bytes(tokens).decode("utf-8", errors="replace")

    def get_name(self):                                            # :189
        return self._name

    encodeOrdinary = encode_ordinary
    countTokens = count_tokens
    countTokensOrdinary = count_tokens_ordinary
    decodeBytes = decode_bytes
    getName = get_name

    # ---- batch -----------------------------------------------------------------------------------------
    def encode_batch(self, texts, ordinary=False, validate=False):
        """List of str/bytes -> BatchResult (one jtk_batch_encode call)."""
        bs = [t if isinstance(t, (bytes, bytearray)) else t.encode("utf-8") for t in texts]
        doc_off = np.zeros(len(bs) + 1, dtype=np.int64)
        if bs:
            np.cumsum([len(b) for b in bs], out=doc_off[1:])
        text = np.frombuffer(b"".join(bs), dtype=np.uint8) if doc_off[-1] else np.zeros(0, dtype=np.uint8)
        return self.encode_batch_packed(text, doc_off, ordinary, validate)

    def encode_batch_packed(self, text_u8, doc_off, ordinary=False, validate=False):
        b = self._b()
        if self._host_pattern is not None:
            pb, pe = self._match_on_host(text_u8, doc_off)
            b.encode_pieces(text_u8, doc_off, pb, pe, ordinary)
        else:
            b.encode_host(text_u8, doc_off, ordinary, validate)
        return b.fetch()

    def _match_on_host(self, text_u8, doc_off):
        """while (matcher.find()) over every document (GptBytePairEncoding.java:77-80) -> byte ranges of the matches."""
        raw = np.ascontiguousarray(text_u8, dtype=np.uint8).tobytes()
        pb, pe = [], []
        for d in range(len(doc_off) - 1):
            lo, hi = int(doc_off[d]), int(doc_off[d + 1])
            doc = raw[lo:hi].decode("utf-8")
            # character index -> byte offset
            pos = 0
            ci = 0
            for m in self._host_pattern.finditer(doc):
                s, e = m.span()
                if e == s:
                    continue
                pos += len(doc[ci:s].encode("utf-8"))
                nb = len(doc[s:e].encode("utf-8"))
                pb.append(lo + pos)
                pe.append(lo + pos + nb)
                pos += nb
                ci = e
        return np.array(pb, dtype=np.int64), np.array(pe, dtype=np.int64)

    def encode_batch_max_tokens(self, texts, max_tokens, ordinary=False):
        """List of str/bytes -> list of EncodingResult, as Encoding.encode(text, maxTokens) gives for each.  Only the leading
        bytes of each document are encoded (jtk_batch_encode_max_tokens), as the reference stops matching at maxTokens."""
        bs = [t if isinstance(t, (bytes, bytearray)) else t.encode("utf-8") for t in texts]
        doc_off = np.zeros(len(bs) + 1, dtype=np.int64)
        if bs:
            np.cumsum([len(x) for x in bs], out=doc_off[1:])
        text = np.frombuffer(b"".join(bs), dtype=np.uint8) if doc_off[-1] else np.zeros(0, dtype=np.uint8)
        if self._host_pattern is not None or len(bs) * max(0, int(max_tokens)) > (1 << 28):
            # a custom pattern (matched on the host), or a limit so large that n_docs x max_tokens ids are no sensible array:
            # encode whole, cut on the device
            res = self.encode_batch_packed(text, doc_off, ordinary)
            b = self._b()
            if len(res.status) and res.status.min() < 0:
                _check(int(res.status.min()))
            kept, flag = b.truncate(max(0, int(max_tokens)))
            return [EncodingResult(res.tokens[res.tok_off[d]:res.tok_off[d] + kept[d]].tolist(), bool(flag[d])) for d in range(len(bs))]
        toks, kept, flag, status = self._b().encode_max_tokens(text, doc_off, max_tokens, ordinary)
        if len(status) and status.min() < 0:
            _check(int(status.min()))
        return [EncodingResult(toks[d, :kept[d]].tolist(), bool(flag[d])) for d in range(len(bs))]

    def count_tokens_batch(self, texts, ordinary=False):
        """Encoding.countTokens / countTokensOrdinary for every text, one device call, no token ids written."""
        bs = [t if isinstance(t, (bytes, bytearray)) else t.encode("utf-8") for t in texts]
        doc_off = np.zeros(len(bs) + 1, dtype=np.int64)
        if bs:
            np.cumsum([len(x) for x in bs], out=doc_off[1:])
        text = np.frombuffer(b"".join(bs), dtype=np.uint8) if doc_off[-1] else np.zeros(0, dtype=np.uint8)
        b = self._b()
        b.encode_host(text, doc_off, ordinary, count_only=True)
        counts, status = b.fetch_counts()
        if len(status) and status.min() < 0:
            _check(int(status.min()))
        return counts.tolist()

    def decode_batch(self, token_lists, strict=True):
        """List of token-id lists -> list of bytes (Encoding.decodeBytes for each), one device call.
        strict: raise for a list with an unknown id, as the reference does (GptBytePairEncoding.java:313)."""
        seq_off = np.zeros(len(token_lists) + 1, dtype=np.int64)
        if token_lists:
            np.cumsum([len(t) for t in token_lists], out=seq_off[1:])
        ids = np.fromiter((i for t in token_lists for i in t), dtype=np.int32, count=int(seq_off[-1]))
        b = self._b()
        b.decode_host(ids, seq_off)
        out, byte_off, status = b.decode_fetch()
        if strict and len(status) and status.min() < 0:
            q = int(np.argmin(status))
            raise EncodingError(int(status[q]), "Unknown token for decoding (list %d)" % q)
        raw = out.tobytes()
        return [raw[byte_off[q]:byte_off[q + 1]] for q in range(len(token_lists))]

    def vocab_size(self):
        return N.lib().jtk_encoding_vocab_size(self._h)

    def pair_count(self):
        return N.lib().jtk_encoding_pair_count(self._h)
