"""Where k_strip_encode's wave time goes, by kind of text -- needs a library built with -DJTK_ENC_STAMP
(tools/build_variants.sh stamp:"-DJTK_ENC_STAMP"; JTOKKIT_AMD_LIB=tools/variants/stamp.so python tools/enc_stamps.py)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=64)
    args = ap.parse_args()
    import torch
    import jtokkit_amd
    from jtokkit_amd import corpus, _native
    lib = _native.lib()
    lib.jtk_debug_stamps.restype = C.c_int
    lib.jtk_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
    lib.jtk_debug_stamps_expand.restype = C.c_int
    lib.jtk_debug_stamps_expand.argtypes = [C.POINTER(C.c_ulonglong)]
    enc = jtokkit_amd.get_encoding("cl100k_base", device=0)
    n_docs = args.mb * 1000000 // 4096
    buf = (C.c_ulonglong * 16)()
    for name in ("_english_stream", "_cjk_stream", "_multiscript_stream", "_emoji_stream", "_code_stream"):
        rng = np.random.default_rng(7)
        stream = getattr(corpus, name)(rng, int(n_docs * 4096 * 1.05) + 4 * 32768)
        text, off = corpus._assemble(rng, [stream], [n_docs], 4096, 256, 32768)
        d_text = torch.from_numpy(text).cuda()
        d_off = torch.from_numpy(off).cuda()
        b = enc.new_batch()
        b.set_profiling(True)
        b.encode_device(d_text.data_ptr(), d_off.data_ptr(), len(off) - 1, len(text), ordinary=False, sync=True)
        lib.jtk_debug_stamps(buf)
        lib.jtk_debug_stamps_expand(buf)
        b.encode_device(d_text.data_ptr(), d_off.data_ptr(), len(off) - 1, len(text), ordinary=False, sync=True)
        lib.jtk_debug_stamps(buf)
        v = [int(x) for x in buf]
        lib.jtk_debug_stamps_expand(buf)
        x = [int(t) for t in buf]
        print("   expand: per strip %.1f kcycles (prologue %.1f, record waits %.1f, steps to their stores %.1f), %.1f steps" % (
            x[0] / max(1, x[5]) / 1e3, x[1] / max(1, x[5]) / 1e3, x[2] / max(1, x[5]) / 1e3, x[3] / max(1, x[5]) / 1e3, x[4] / max(1, x[5])))
        tot, main, mrg, hol = v[0], v[1], v[2], v[3]
        strips, pieces = max(1, v[8]), v[9]
        print("%-20s strip_encode %.3f ms expand %.3f ms | wave time: main %.0f%% hole batches %.0f%% (merge %.0f%%) other %.0f%% | per strip: %.0f pieces, %.0f holes (%.0f%%), "
              "%.1f chunks, %.2f batches (%.0f%% of lanes), %.2f rounds (%.0f%% of lanes), %.0f memo hits, %.0f kcycles" % (
                  name, b.kernel_times()["strip_encode"], b.kernel_times()["strip_expand"], 100.0 * main / tot, 100.0 * hol / tot, 100.0 * mrg / tot,
                  100.0 * (tot - main - mrg - hol) / tot, pieces / strips, v[12] / strips, 100.0 * v[12] / max(1, pieces), v[4] / strips,
                  v[10] / strips, 100.0 * v[11] / max(1, 64 * v[10]), v[5] / strips, 100.0 * v[6] / max(1, 64 * v[5]), v[13] / strips, (main + mrg + hol) / strips / 1e3), flush=True)
        del b, d_text, d_off


if __name__ == "__main__":
    main()
