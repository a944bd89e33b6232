"""The four predefined encodings (reference EncodingFactory.java:60-109) behind the reference's names.

Only the parameters travel here: pattern kind, rank file and special tokens.  The reference's registry
classes (lookup plumbing) are not rebuilt; the Java side plugs in through
EncodingRegistry.registerCustomEncoding.
"""
import os
import threading

from . import _native as N
from .encoding import HipEncoding

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

ENDOFTEXT, FIM_PREFIX, FIM_MIDDLE, FIM_SUFFIX, ENDOFPROMPT = (
    "<|endoftext|>", "<|fim_prefix|>", "<|fim_middle|>", "<|fim_suffix|>", "<|endofprompt|>")

# EncodingFactory.java:24-53, :60-109
ENCODING_PARAMS = {
    "r50k_base": (N.JTK_PATTERN_R50K, "r50k_base.tiktoken", {ENDOFTEXT: 50256}),
    "p50k_base": (N.JTK_PATTERN_R50K, "p50k_base.tiktoken", {ENDOFTEXT: 50256}),
    "p50k_edit": (N.JTK_PATTERN_R50K, "p50k_base.tiktoken",
                  {ENDOFTEXT: 50256, FIM_PREFIX: 50281, FIM_MIDDLE: 50282, FIM_SUFFIX: 50283}),
    "cl100k_base": (N.JTK_PATTERN_CL100K, "cl100k_base.tiktoken",
                    {ENDOFTEXT: 100257, FIM_PREFIX: 100258, FIM_MIDDLE: 100259, FIM_SUFFIX: 100260,
                     ENDOFPROMPT: 100276}),
}

_lock = threading.Lock()
_cache = {}


def new_encoding(name, device=0):
    kind, fname, specials = ENCODING_PARAMS[name]
    path = os.path.join(DATA_DIR, fname)
    if not os.path.exists(path):
        raise RuntimeError("Could not find " + path)              # IllegalStateException, EncodingFactory.java:142
    with open(path, "rb") as f:
        data = f.read()
    return HipEncoding(name, kind, data, specials, device)


def get_encoding(name, device=0):
    """Lazy, cached per (name, device) -- the rank-table swap of BASELINE config 5 is a cache hit."""
    key = (name, device)
    with _lock:
        enc = _cache.get(key)
        if enc is None:
            enc = _cache[key] = new_encoding(name, device)
        return enc


def new_custom_encoding(name, pattern_kind, mergeable_ranks, special_tokens=None, device=0, host_pattern=None):
    """A custom byte-pair encoding on the device: what EncodingRegistry.registerGptBytePairEncoding(
    GptBytePairEncodingParams(name, pattern, mergeableRanks, specialTokens)) builds in the reference
    (api/GptBytePairEncodingParams.java:36-46, AbstractEncodingRegistry.java:64-66, EncodingFactory.java:117-119).

    mergeable_ranks: {bytes: rank}.  pattern_kind: JTK_PATTERN_R50K or JTK_PATTERN_CL100K are evaluated on the device;
    any other pattern: pass host_pattern (an object with finditer(), e.g. regex.compile(java_pattern_text)) -- the batch
    methods then match on the host and encode the matches on the device (jtk_batch_encode_pieces).  A rank table need not
    contain all 256 single bytes (text that needs a missing one raises ValueError, as TokenEncoder.java:66-68 throws); entries
    that bytePairMerge of their own bytes does
    not reproduce are honoured through the whole-piece lookup at any length (GptBytePairEncoding.java:81-83)."""
    import base64
    lines = [base64.b64encode(k) + b" " + str(int(v)).encode() for k, v in sorted(mergeable_ranks.items(), key=lambda kv: kv[1])]
    data = b"\n".join(lines) + b"\n"
    return HipEncoding(name, pattern_kind, data, dict(special_tokens or {}), device, host_pattern=host_pattern)
