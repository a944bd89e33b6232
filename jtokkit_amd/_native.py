"""ctypes binding of libjtokkit_amd.so (the C ABI in include/jtokkit_amd.h).

The shared library is built in-tree by `make -C jtokkit_amd/csrc` (or __graft_entry__.build()).
There is no fallback: if the library is missing, loading fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JTOKKIT_AMD_LIB") or os.path.join(_HERE, "libjtokkit_amd.so")   # (override: kernel experiments)

JTK_OK = 0
JTK_ERR_INVALID_ARGUMENT = -1
JTK_ERR_UNSUPPORTED_SPECIAL = -2
JTK_ERR_UNKNOWN_TOKEN = -3
JTK_ERR_CAPACITY = -4
JTK_ERR_BAD_RANK_FILE = -5
JTK_ERR_BAD_UTF8 = -6
JTK_ERR_NO_DEVICE = -7
JTK_ERR_HIP = -8
JTK_ERR_UNSUPPORTED_TABLE = -9
JTK_ERR_PIECE_TOO_LONG = -10
JTK_ERR_OUT_OF_MEMORY = -11
JTK_ERR_UNENCODABLE = -12

JTK_PATTERN_R50K = 0
JTK_PATTERN_CL100K = 1
JTK_ENCODE_ORDINARY = 1
JTK_ENCODE_VALIDATE_UTF8 = 2
JTK_ENCODE_COUNT_ONLY = 4
JTK_ENCODE_TO_HOST = 8
JTK_OPT_CHUNK_BYTES = 1
JTK_OPT_CHUNKS_IN_FLIGHT = 2
JTK_OPT_HOST_CHUNK_BYTES = 3
JTK_OPT_REUSE_CHUNK_PLAN = 4

# every symbol include/jtokkit_amd.h declares: (restype, argtypes)
_p = C.c_void_p
_i64 = C.c_int64
SIGNATURES = {
    "jtk_version": (C.c_char_p, []),
    "jtk_last_error": (C.c_char_p, []),
    "jtk_device_count": (C.c_int, []),
    "jtk_encoding_create": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p),
                                      C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(_p)]),
    "jtk_encoding_destroy": (None, [_p]),
    "jtk_encoding_name": (C.c_char_p, [_p]),
    "jtk_encoding_device": (C.c_int, [_p]),
    "jtk_encoding_vocab_size": (_i64, [_p]),
    "jtk_encoding_pair_count": (_i64, [_p]),
    "jtk_batch_create": (C.c_int, [_p, C.POINTER(_p)]),
    "jtk_batch_destroy": (None, [_p]),
    "jtk_batch_set_option": (C.c_int, [_p, C.c_int, _i64]),
    "jtk_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(_p)]),
    "jtk_host_free": (None, [_p]),
    "jtk_batch_host_result": (C.c_int, [_p, C.POINTER(_p), C.POINTER(_p), C.POINTER(_p)]),
    "jtk_batch_encode": (C.c_int, [_p, _p, _p, _i64, C.c_uint32, C.POINTER(_i64)]),
    "jtk_batch_encode_pieces": (C.c_int, [_p, _p, _p, _i64, _p, _p, _i64, C.c_uint32, C.POINTER(_i64)]),
    "jtk_batch_encode_device": (C.c_int, [_p, _p, _p, _i64, _i64, C.c_uint32, _p, C.POINTER(_i64)]),
    "jtk_batch_stream": (_p, [_p]),
    "jtk_batch_result": (C.c_int, [_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(C.c_int32)]),
    "jtk_batch_fetch": (C.c_int, [_p, _p, _i64, _p, _p]),
    "jtk_batch_device_result": (C.c_int, [_p, C.POINTER(_p), C.POINTER(_p), C.POINTER(_p)]),
    "jtk_batch_set_profiling": (C.c_int, [_p, C.c_int]),
    "jtk_batch_kernel_times": (C.c_int, [_p, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]),
    "jtk_batch_truncate": (C.c_int, [_p, _i64]),
    "jtk_batch_encode_max_tokens": (C.c_int, [_p, _p, _p, _i64, C.c_uint32, _i64, _p, _p, _p, _p]),
    "jtk_batch_fetch_truncated": (C.c_int, [_p, _p, _p]),
    "jtk_batch_device_truncated": (C.c_int, [_p, C.POINTER(_p), C.POINTER(_p)]),
    "jtk_batch_decode": (C.c_int, [_p, _p, _p, _i64, C.POINTER(_i64)]),
    "jtk_batch_decode_device": (C.c_int, [_p, _p, _p, _i64, _i64, _p, C.POINTER(_i64)]),
    "jtk_batch_decode_fetch": (C.c_int, [_p, _p, _i64, _p, _p]),
    "jtk_batch_decode_device_result": (C.c_int, [_p, C.POINTER(_p), C.POINTER(_p), C.POINTER(_p)]),
    "jtk_service_create": (C.c_int, [_p, C.c_int, C.POINTER(_p)]),
    "jtk_service_destroy": (None, [_p]),
    "jtk_service_encode": (C.c_int, [_p, _p, _i64, C.c_uint32, _i64, _p, _i64, C.POINTER(_i64), C.POINTER(C.c_int)]),
    "jtk_service_submit": (C.c_int, [_p, _p, _i64, C.c_uint32, _i64, _p, _i64, C.POINTER(_p)]),
    "jtk_service_wait": (C.c_int, [_p, _p, C.POINTER(_i64), C.POINTER(C.c_int)]),
    "jtk_service_done": (C.c_int, [_p]),
    "jtk_service_stats": (C.c_int, [_p, C.POINTER(_i64), C.POINTER(_i64)]),
    "jtk_service_set_limits": (C.c_int, [_p, _i64, _i64]),
    "jtk_shard_plan": (C.c_int, [_p, _i64, C.c_int, _p]),
    "jtk_comm_unique_id": (C.c_int, [_p]),
    "jtk_comm_create": (C.c_int, [_p, C.c_int, C.c_int, C.c_int, C.POINTER(_p)]),
    "jtk_comm_destroy": (None, [_p]),
    "jtk_comm_world": (C.c_int, [_p]),
    "jtk_comm_rank": (C.c_int, [_p]),
    "jtk_comm_stitch": (C.c_int, [_p, _p, _i64, _p, _p, C.POINTER(_p), C.POINTER(_p)]),
    "jtk_comm_fetch": (C.c_int, [_p, _p, _p, C.POINTER(_i64)]),
    "jtk_encode": (C.c_int, [_p, _p, _i64, C.c_uint32, _i64, _p, _i64, C.POINTER(_i64), C.POINTER(C.c_int)]),
    "jtk_decode": (C.c_int, [_p, _p, _i64, _p, _i64, C.POINTER(_i64)]),
}

_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same soname
    as /opt/rocm's); if this library pulled in the system copy first, a later `import torch` would load
    a second runtime and fail with "No HIP GPUs are available".  So when torch is installed, its copy
    is loaded first and libjtokkit_amd.so binds to it by soname, whatever the import order."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "jtokkit_amd: %s is missing -- build it with `make -C jtokkit_amd/csrc` "
                "(there is no CPU fallback)" % LIB_PATH)
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the library does not export the symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    return lib().jtk_last_error().decode("utf-8", "replace")
