"""Per-kernel times by kind of text: each of the mixed corpus's five streams alone (about `--mb` MB each, 4 KB documents),
one batch encode on the device, three runs, the last one's kernel times.   python tools/stream_times.py [--mb 128]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=128)
    ap.add_argument("--encoding", default="cl100k_base")
    ap.add_argument("--only", default=None, help="one stream: english, cjk, multiscript, emoji, code")
    args = ap.parse_args()
    import torch
    import jtokkit_amd
    from jtokkit_amd import corpus
    enc = jtokkit_amd.get_encoding(args.encoding, device=0)
    n_docs = args.mb * 1000000 // 4096
    for name in ("_english_stream", "_cjk_stream", "_multiscript_stream", "_emoji_stream", "_code_stream"):
        if args.only and name != "_%s_stream" % args.only:
            continue
        rng = np.random.default_rng(7)
        stream = getattr(corpus, name)(rng, int(n_docs * 4096 * 1.05) + 4 * 32768)
        text, off = corpus._assemble(rng, [stream], [n_docs], 4096, 256, 32768)
        d_text = torch.from_numpy(text).cuda()
        d_off = torch.from_numpy(off).cuda()
        b = enc.new_batch()
        b.set_profiling(True)
        for _ in range(3):
            b.encode_device(d_text.data_ptr(), d_off.data_ptr(), len(off) - 1, len(text), ordinary=False, sync=True)
        kt = b.kernel_times()
        nt = b.result()[0]
        tot = sum(kt.values())
        print("%-20s %6.1f MB %9d tokens (%.2f B/token)  total %.3f ms = %.1f GB/s | " % (name, len(text) / 1e6, nt, len(text) / nt, tot, len(text) / tot / 1e6)
              + "  ".join("%s %.3f" % (k, v) for k, v in kt.items()), flush=True)
        del b, d_text, d_off


if __name__ == "__main__":
    main()
