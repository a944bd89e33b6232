"""CPU-only tests of the product's host side: the C-ABI library loads and exports every symbol
include/jtokkit_amd.h declares, fails loudly without a GPU, and the host/device-shared logic (split
rules, rank-table build, lane merge) agrees with the oracle when run on the CPU through the test shim
tests/hostsim (no compute entry point of the product is called here)."""
import ctypes as C
import os
import random
import re
import subprocess

import numpy as np
import pytest

import golden_util
import oracle_lib
import regex_crosscheck as rc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    so = os.path.join(ROOT, "jtokkit_amd", "libjtokkit_amd.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "jtokkit_amd", "csrc")])
    from jtokkit_amd import _native
    return _native


def test_header_symbols_are_exported(native):
    """Every function declared in include/jtokkit_amd.h is exported by the shared library and bound."""
    hdr = open(os.path.join(ROOT, "include", "jtokkit_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(jtk_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = native.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), "library does not export " + name
        assert name in native.SIGNATURES, "python binding misses " + name
    assert set(native.SIGNATURES) == declared


def test_status_codes_match_header(native):
    hdr = open(os.path.join(ROOT, "include", "jtokkit_amd.h")).read()
    for name, val in re.findall(r"(JTK_[A-Z_0-9]+)\s*=\s*(-?\d+)u?", hdr):
        if hasattr(native, name):
            assert getattr(native, name) == int(val), name


def test_fails_loudly_without_gpu(native):
    """No CPU fallback: creating an encoding without a HIP device is an error, not a slow path."""
    import jtokkit_amd
    if native.lib().jtk_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(jtokkit_amd.EncodingError) as e:
        jtokkit_amd.new_encoding("cl100k_base")
    assert e.value.code == native.JTK_ERR_NO_DEVICE


def test_bad_rank_file_is_rejected(native):
    """EncodingFactory.java:150-152: a line without two fields -> IllegalStateException."""
    lib = native.lib()
    h = C.c_void_p()
    bad = b"IQ== 0\nnot-a-valid-line\n"
    rc_ = lib.jtk_encoding_create(b"x", 1, bad, len(bad), None, None, 0, 0, C.byref(h))
    assert rc_ == native.JTK_ERR_BAD_RANK_FILE
    # a map that lacks single bytes is accepted as the reference accepts it (the failure comes at encode time, per document:
    # test_rank_map_without_all_single_bytes) -- unless its ids leave no room for the missing bytes' pseudo ids
    no_room = b"IQ== 131070\n"
    rc_ = lib.jtk_encoding_create(b"x", 1, no_room, len(no_room), None, None, 0, 0, C.byref(h))
    assert rc_ == native.JTK_ERR_UNSUPPORTED_TABLE
    missing_bytes = b"IQ== 0\nIg== 1\n"           # parses, lacks most single bytes: table building passes, then no device here
    rc_ = lib.jtk_encoding_create(b"x", 1, missing_bytes, len(missing_bytes), None, None, 0, 0, C.byref(h))
    assert rc_ != native.JTK_ERR_UNSUPPORTED_TABLE and rc_ != native.JTK_ERR_BAD_RANK_FILE
    if rc_ == native.JTK_OK:
        lib.jtk_encoding_destroy(h)
    assert lib.jtk_encoding_create(b"x", 7, bad, len(bad), None, None, 0, 0, C.byref(h)) == native.JTK_ERR_INVALID_ARGUMENT


# ---- host/device-shared logic on the CPU (tests/hostsim) -------------------------------------------------

@pytest.fixture(scope="module")
def sim():
    d = os.path.join(ROOT, "tests", "hostsim")
    subprocess.check_call(["make", "-C", d, "-s"])
    L = C.CDLL(os.path.join(d, "libjtk_hostsim.so"))
    L.sim_split.argtypes = [C.c_int, C.c_char_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    L.sim_tables_create.restype = C.c_void_p
    L.sim_tables_create.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
    L.sim_tables_destroy.argtypes = [C.c_void_p]
    L.sim_tables_pairs.restype = C.c_int64
    L.sim_tables_pairs.argtypes = [C.c_void_p]
    L.sim_merge_piece.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p]
    L.sim_tok8_lookup.restype = C.c_int64
    L.sim_tok8_lookup.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.sim_tok8_count.restype = C.c_int64
    L.sim_tok8_count.argtypes = [C.c_void_p]
    L.sim_split_masks.argtypes = [C.c_int, C.c_char_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.sim_block_classify_check.restype = C.c_int64
    L.sim_block_classify_check.argtypes = [C.c_int, C.c_char_p, C.c_int64]
    return L


def _sim_pieces(sim, kind, docs):
    bs = [d.encode("utf-8") for d in docs]
    text = b"".join(bs)
    off = np.zeros(len(bs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(b) for b in bs])
    ms = np.zeros(len(text) + 1, dtype=np.uint8)
    sim.sim_split(kind, text, len(text), off.ctypes.data, len(bs), ms.ctypes.data)
    starts = np.nonzero(ms)[0].tolist()
    return [text[starts[i]:starts[i + 1]] for i in range(len(starts) - 1)]


@pytest.mark.parametrize("name,kind", [("cl100k_base", 1), ("r50k_base", 0)])
def test_split_rules_match_oracle(sim, name, kind):
    """jtk_split_rules.h (the per-byte form the pretok_split kernel evaluates) vs the oracle's sequential
    backtracking matcher, on multi-document batches."""
    enc = oracle_lib.get(name)
    rng = random.Random(99 + kind)
    for _ in range(4000):
        docs = [rc.random_text(rng, 30) for _ in range(rng.randint(1, 4))]
        exp = []
        for d in docs:
            exp += enc.split(d)
        assert _sim_pieces(sim, kind, docs) == exp, docs
    docs = [r[0] for r in golden_util.load_rows(name)]
    exp = []
    for d in docs:
        exp += enc.split(d)
    assert _sim_pieces(sim, kind, docs) == exp


@pytest.mark.parametrize("kind", [1, 0])
def test_mask_algebra_matches_per_byte_rules(sim, kind):
    """jtk_split_masks.h (the rules for 64 bytes at a time, with carries between blocks and the per-position fallbacks) vs
    jtk_split_rules.h position by position, for waves of 62, 3 and 1 blocks; also bounds how often the fallback runs."""
    from jtokkit_amd import corpus
    rng = random.Random(17 + kind)
    batches = []
    for _ in range(300):
        docs = [rc.random_text(rng, rng.choice((5, 30, 200))) for _ in range(rng.randint(1, 6))]
        batches.append([d.encode("utf-8") for d in docs])
    t, o = corpus.mixed(40, mean_bytes=2048, lo=256, hi=8192)
    batches.append([t[o[i]:o[i + 1]].tobytes() for i in range(len(o) - 1)])
    ws = " \t\n\r\u3000\u00a0"
    for _ in range(200):                                   # whitespace / newline / digit runs across block edges
        parts = []
        for _ in range(rng.randint(2, 12)):
            k = rng.random()
            if k < 0.4: parts.append("".join(rng.choice(ws) for _ in range(rng.randint(1, 90))))
            elif k < 0.6: parts.append("".join(rng.choice("0123456789\u0663\uff15") for _ in range(rng.randint(1, 150))))
            else: parts.append(rng.choice(["x", ";", "word", "'s", "\u4e2d", ")", "a1"]))
        batches.append(["".join(parts).encode("utf-8")])
    total = slow_total = 0
    for bs in batches:
        text = b"".join(bs)
        off = np.zeros(len(bs) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(b) for b in bs])
        ref = np.zeros(len(text) + 1, dtype=np.uint8)
        sim.sim_split(kind, text, len(text), off.ctypes.data, len(bs), ref.ctypes.data)
        for wave_blocks in (62, 3, 1):
            ms = np.zeros(len(text) + 1, dtype=np.uint8)
            n_slow = C.c_int64(0)
            sim.sim_split_masks(kind, text, len(text), off.ctypes.data, len(bs), wave_blocks, ms.ctypes.data, C.byref(n_slow))
            assert np.array_equal(ms, ref), (wave_blocks, bs)
            if wave_blocks == 62:
                total += len(text)
                slow_total += n_slow.value
    assert slow_total < 0.005 * total, (slow_total, total)


def test_lead_bytes_whose_characters_are_all_letters(sim):
    """jtk_lead_all_letters (the split kernel skips the decode of such characters): checked against Python's Unicode data
    for every lead byte -- CJK ideographs U+5000-8FFF, Hangul U+B000-CFFF and basic Cyrillic qualify; U+4000-4FFF (hexagram
    symbols at U+4DC0) and U+9000-9FFF (unassigned tail) do not."""
    import unicodedata
    if not unicodedata.unidata_version.startswith("13."):
        pytest.skip("class tables are Unicode 13.0")

    def rng(b):
        if 0xC2 <= b <= 0xDF: return ((b & 0x1F) << 6, ((b & 0x1F) << 6) + 63)
        if b == 0xE0: return (0x800, 0xFFF)
        if b == 0xED: return (0xD000, 0xD7FF)
        if 0xE1 <= b <= 0xEF: return ((b & 0xF) << 12, ((b & 0xF) << 12) + 0xFFF)
        if b == 0xF0: return (0x10000, 0x3FFFF)
        if 0xF1 <= b <= 0xF3: return ((b & 7) << 18, ((b & 7) << 18) + 0x3FFFF)
        if b == 0xF4: return (0x100000, 0x10FFFF)
        return None
    got = [b for b in range(256) if sim.sim_lead_all_letters(b)]
    exp = []
    for b in range(256):
        r = rng(b)
        if r and all(unicodedata.category(chr(c)).startswith("L") for c in range(r[0], r[1] + 1)):
            exp.append(b)
    assert got == exp
    assert {0xD0, 0xD1, 0xE5, 0xE6, 0xE7, 0xE8, 0xEB, 0xEC} <= set(got) and 0xE4 not in got and 0xE9 not in got


@pytest.mark.parametrize("kind", [1, 0])
def test_block_classification_matches_per_byte_rules(sim, kind):
    """jtk_block_classify.h (one lane classifies 64 bytes: flag table + shift-or accumulation, 'all letters' lead bytes,
    per-character decode of the rest) vs the per-byte class rule, on fuzz text, the mixed corpus and every code point."""
    from jtokkit_amd import corpus
    rng = random.Random(5 + kind)
    texts = ["".join(rc.random_text(rng, 40) for _ in range(3000))]
    texts.append(corpus.mixed(60, mean_bytes=2048, lo=256, hi=8192)[0].tobytes().decode("utf-8"))
    every = "".join(chr(c) for c in range(1, 0x110000) if not 0xD800 <= c < 0xE000)
    texts += [every, "a" + every, "ab" + every, "abc" + every]      # every alignment of every character in its block
    for t in texts:
        b = t.encode("utf-8")
        assert sim.sim_block_classify_check(kind, b, len(b)) == 0


@pytest.mark.parametrize("name,pairs", [("cl100k_base", 233378), ("r50k_base", 108299), ("p50k_base", 108599)])
def test_pair_table_and_lane_merge_match_oracle(sim, name, pairs):
    """jtk_tables.cpp builds the (left id, right id) -> rank table (SURVEY appendix B counts); the lane
    merge of jtk_merge_core.h on it equals the oracle's byte-string bytePairMerge."""
    cfg = oracle_lib.ENCODINGS[name]
    data = open(os.path.join(oracle_lib.DATA_DIR, cfg["file"]), "rb").read()
    st = C.c_int(0)
    h = sim.sim_tables_create(name.encode(), cfg["kind"], data, len(data), C.byref(st))
    assert st.value == 0 and h
    assert sim.sim_tables_pairs(h) == pairs
    enc = oracle_lib.get(name)
    out = np.zeros(64, dtype=np.int32)
    rng = random.Random(5)
    pieces = []
    for inp, _, _ in golden_util.load_rows(name)[::3]:
        pieces += [p for p in enc.split(inp) if len(p) <= 64]
    alphabet = b"abcdefghijklmnopqrstuvwxyz ETAOIN.,!0123456789\xe4\xb8\xad\xe6\x96\x87\xf0\x9f\x8d\x95\n"
    for _ in range(4000):
        pieces.append(bytes(rng.choice(alphabet) for _ in range(rng.randint(1, 64))))
    for p in pieces:
        k = sim.sim_merge_piece(h, p, len(p), out.ctypes.data)
        assert k >= 0 and out[:k].tolist() == enc.merge_piece(p), p
    # whole-piece table: a hit iff the piece (<= 8 bytes) is a table entry, and then it is that entry's rank
    import base64
    table = {}
    for line in data.split(b"\n"):
        if line:
            tok, rank = line.split()
            table[base64.b64decode(tok)] = int(rank)
    assert sim.sim_tok8_count(h) == sum(1 for t in table if len(t) <= 8)
    for p in pieces + [t for t in list(table)[::17] if len(t) <= 8]:
        if len(p) <= 8:
            assert sim.sim_tok8_lookup(h, p, len(p)) == table.get(p, -1), p
    sim.sim_tables_destroy(h)


def test_corpus_generators_are_deterministic_and_valid():
    from jtokkit_amd import corpus
    a = corpus.english(300)
    b = corpus.english(300)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    text, off = corpus.mixed(200)
    assert off[0] == 0 and off[-1] == len(text) and (np.diff(off) > 0).all()
    for d in range(len(off) - 1):
        text[off[d]:off[d + 1]].tobytes().decode("utf-8")          # every document is well-formed UTF-8
    t1, o1 = corpus.sentences(1000)
    assert len(o1) == 1001 and t1.max() < 128
    bounds = corpus.shard_by_bytes(off, 4)
    assert bounds[0] == 0 and bounds[-1] == len(off) - 1 and bounds == sorted(bounds)


def test_shard_plan_matches_python_reference():
    """jtk_shard_plan (host-only entry point of the C ABI): contiguous document ranges balanced by bytes."""
    from jtokkit_amd import corpus, sharding
    text, doc_off = corpus.mixed(500, mean_bytes=900, lo=64, hi=8192, seed=17)
    doc_off = np.concatenate([[0, 0], doc_off[1:], [doc_off[-1]]]).astype(np.int64)     # empty documents at both ends
    for world in (1, 2, 3, 8):
        b = sharding.shard_plan(doc_off, world)
        assert b == sharding.shard_by_bytes(doc_off, world)
        assert b[0] == 0 and b[-1] == len(doc_off) - 1 and all(x <= y for x, y in zip(b, b[1:]))
        sizes = [int(doc_off[b[r + 1]] - doc_off[b[r]]) for r in range(world)]
        assert max(sizes) - min(sizes) <= 2 * 8192


def test_rank_map_without_all_single_bytes(sim):
    """A rank map need not hold all 256 single bytes (api/GptBytePairEncodingParams.java:36-46 takes any map); the reference
    then fails on a piece whose merge leaves such a byte alone (TokenEncoder.java:66-68) and encodes every other piece.  The
    table builder gives the missing bytes pseudo ids that take part in the merge like any id: pieces whose result holds one are
    exactly those the oracle (built from the same map) refuses, and all other pieces merge to the oracle's tokens."""
    import base64
    import random
    sim.sim_tables_pseudo_base.restype = C.c_int64
    sim.sim_tables_pseudo_base.argtypes = [C.c_void_p]
    rng = random.Random(3)
    alphabet = b"abcdeqxz .\n"
    ranks = {}
    for ch in b"abcde .":                        # 'q', 'x', 'z' and '\n' are no tokens on their own
        ranks[bytes([ch])] = len(ranks)
    for tok in (b"ab", b"qa", b"abq", b"xz", b"qab", b"de", b" a", b"zq", b"qq", b"ez", b"cde", b"\n\n", b"abqa"):
        ranks[tok] = len(ranks) + 3             # (holes in the id range)
    data = b"\n".join(base64.b64encode(k) + b" " + str(v).encode() for k, v in sorted(ranks.items(), key=lambda kv: kv[1])) + b"\n"
    st = C.c_int(0)
    h = sim.sim_tables_create(b"partial", 1, data, len(data), C.byref(st))
    assert h and st.value == 0
    base = sim.sim_tables_pseudo_base(h)
    assert base == max(ranks.values()) + 1
    o = oracle_lib.OracleEncoding("partial", 1, data, {})
    out = np.zeros(64, dtype=np.int32)
    n_err = n_ok = 0
    pieces = [bytes(rng.choice(alphabet) for _ in range(rng.randint(1, 12))) for _ in range(4000)] + [b"q", b"qa", b"qab", b"abqa", b"zq", b"x"]
    for p in pieces:
        whole = sim.sim_tok8_lookup(h, p, len(p))
        if whole >= 0:
            got = [whole]
        else:
            k = sim.sim_merge_piece(h, p, len(p), out.ctypes.data)
            assert k > 0
            got = out[:k].tolist()
        if p in ranks:
            exp = [ranks[p]]                       # GptBytePairEncoding.java:81-83
        else:
            try:
                exp = o.merge_piece(p)
            except oracle_lib.OracleError:
                exp = None                         # TokenEncoder.java:66-68
        if exp is None:
            assert max(got) >= base, p
            n_err += 1
        else:
            assert got == exp and max(got) < base, p
            n_ok += 1
    assert n_err > 100 and n_ok > 100
    sim.sim_tables_destroy(h)


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_service_host_logic_under_sanitizers(tmp_path, sanitizer):
    """jtk_service.cpp (queue sharded by producer thread, one-word ticket hand-off where the waiter frees the ticket the moment
    it reads done, blocking / submit-wait / polled callers mixed, destroy with tickets still queued) compiled for the CPU with
    ThreadSanitizer and with AddressSanitizer against a stubbed batch (tests/cpp/service_tsan.cpp): 8 threads, 3 rounds,
    72,000 documents + 18,000 left queued at shutdown, every result checked, no report from the sanitizer."""
    exe = os.path.join(str(tmp_path), "service_san")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=" + sanitizer, "-pthread", "-D__HIP_PLATFORM_AMD__",
                           "-I/opt/rocm/include", "-o", exe, os.path.join(ROOT, "tests", "cpp", "service_tsan.cpp"),
                           os.path.join(ROOT, "jtokkit_amd", "csrc", "jtk_service.cpp")])
    p = subprocess.run([exe, "8", "3000", "32"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "service ok: 72000 documents" in p.stdout, (p.returncode, p.stdout[-300:], p.stderr[-2000:])
    assert "WARNING: ThreadSanitizer" not in p.stderr and "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-2000:]


def _build_mirror_smoke(tmp_path):
    exe = os.path.join(str(tmp_path), "mirror_smoke")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "mirror_smoke.cpp"),
                           "-L" + os.path.join(ROOT, "jtokkit_amd"), "-ljtokkit_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "jtokkit_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_mirror_compiles_and_fails_loudly_without_a_device(tmp_path):
    """jtokkit_amd/csrc/jtk_encoding.hpp (the C++ face of the boundary) builds against the C ABI; without a HIP device its
    constructor throws the mapped exception -- there is no CPU path behind it."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present: covered by the gpu tier")
    exe = _build_mirror_smoke(tmp_path)
    p = subprocess.run([exe, os.path.join(ROOT, "jtokkit_amd", "data", "cl100k_base.tiktoken")], capture_output=True, text=True)
    assert p.returncode == 3 and "no HIP device" in p.stdout, (p.returncode, p.stdout, p.stderr)


@pytest.mark.gpu
def test_cpp_mirror_encodes_on_the_device(tmp_path):
    exe = _build_mirror_smoke(tmp_path)
    p = subprocess.run([exe, os.path.join(ROOT, "jtokkit_amd", "data", "cl100k_base.tiktoken")], capture_output=True, text=True)
    assert p.returncode == 0 and "mirror ok" in p.stdout, (p.returncode, p.stdout, p.stderr)

# ---- the Java shim and its JNI glue are source only (no JDK here): at least their seams must agree ------------------------
def _java_natives():
    import re
    src = open(os.path.join(ROOT, "jtokkit_amd", "java", "com", "knuddels", "jtokkit", "hip", "HipEncoding.java")).read()
    out = {}
    for m in re.finditer(r"private static native\s+([\w\[\].]+)\s+(\w+)\s*\(([^)]*)\)\s*;", src, re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        types = [a.split()[-2] if len(a.split()) >= 2 else a for a in (x.strip() for x in args.split(",")) if a]
        out[name] = (ret, types)
    return out


def _jni_functions():
    import re
    src = open(os.path.join(ROOT, "jtokkit_amd", "java", "jtk_jni.c")).read()
    out = {}
    for m in re.finditer(r"JNIEXPORT\s+(\w+)\s+JNICALL\s+FN\((\w+)\)\s*\(([^)]*)\)", src, re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        types = [a.split()[0] for a in (x.strip() for x in args.split(",")) if a]
        assert types[:2] == ["JNIEnv*", "jclass"], (name, types)
        out[name] = (ret, types[2:])
    return out


_JNI_TYPE = {"long": "jlong", "int": "jint", "void": "void", "boolean": "jboolean", "byte[]": "jbyteArray", "int[]": "jintArray",
             "long[]": "jlongArray", "boolean[]": "jbooleanArray", "String": "jstring", "String[]": "jobjectArray",
             "byte[][]": "jobjectArray", "ByteBuffer": "jobject", "BatchResult": "jobject"}


def test_java_native_declarations_match_the_jni_glue():
    """Every `native` method of HipEncoding.java has a Java_..._<name> function in jtk_jni.c with the same argument and result
    types (and the other way round): the two files are never compiled here, so this is the only check of that seam."""
    java, glue = _java_natives(), _jni_functions()
    assert len(java) >= 16 and set(java) == set(glue), (sorted(set(java) ^ set(glue)))
    for name, (ret, types) in java.items():
        assert (_JNI_TYPE[ret], [_JNI_TYPE[t] for t in types]) == glue[name], (name, java[name], glue[name])


def test_missing_rccl_is_a_status_code_with_a_message():
    """jtk_comm_* bind RCCL at run time; a library that cannot be loaded must come back as JTK_ERR_HIP with a message, not as a
    crash (dlerror() clears the message it returns: calling it twice handed NULL to std::string)."""
    code = ("import ctypes as C\n"
            "L = C.CDLL(%r)\n"
            "L.jtk_last_error.restype = C.c_char_p\n"
            "buf = (C.c_uint8 * 128)()\n"
            "rc = L.jtk_comm_unique_id(buf)\n"
            "print(rc, L.jtk_last_error().decode())\n") % os.path.join(ROOT, "jtokkit_amd", "libjtokkit_amd.so")
    env = dict(os.environ, JTK_RCCL_LIB="/nonexistent/librccl.so")
    p = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert p.returncode == 0, (p.returncode, p.stderr[-300:])
    rc, msg = p.stdout.split(" ", 1)
    assert int(rc) == -8 and "RCCL not found" in msg and "nonexistent" in msg, p.stdout
