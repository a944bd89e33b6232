"""The N > 1 path on CPU: two gloo ranks shard one batch by bytes, encode their shards (with the CPU
oracle standing in for the GPU encode -- this tier has no GPU), all-gather the shard token totals and
stitch global offsets.  The stitched result must equal the single-process encoding of the whole batch."""
import os
import socket
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jtokkit_amd import corpus, sharding
    import oracle_lib
    text, doc_off = corpus.mixed(240, mean_bytes=600, lo=64, hi=4096, seed=11)
    # a few empty documents, also at shard edges
    doc_off = np.concatenate([[0, 0], doc_off[1:120], [doc_off[120]] * 3, doc_off[121:], [doc_off[-1]]]).astype(np.int64)
    enc = oracle_lib.get("cl100k_base")
    my_text, my_off, first_doc = sharding.local_shard(text, doc_off, rank, world)
    toks, tok_off = enc.encode_batch(np.ascontiguousarray(my_text), np.ascontiguousarray(my_off), threads=2)
    totals, base = sharding.gather_shard_totals(len(toks))
    g_off = sharding.stitch_offsets(tok_off, base)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), toks=toks, g_off=g_off, first_doc=first_doc,
             totals=totals.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_stitch(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from jtokkit_amd import corpus
    import oracle_lib
    text, doc_off = corpus.mixed(240, mean_bytes=600, lo=64, hi=4096, seed=11)
    doc_off = np.concatenate([[0, 0], doc_off[1:120], [doc_off[120]] * 3, doc_off[121:], [doc_off[-1]]]).astype(np.int64)
    exp_tok, exp_off = oracle_lib.get("cl100k_base").encode_batch(text, doc_off, threads=4)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert parts[0]["first_doc"] == 0 and parts[1]["first_doc"] > 0
    assert np.array_equal(np.concatenate([p["toks"] for p in parts]), exp_tok)
    g = np.concatenate([parts[0]["g_off"][:-1], parts[1]["g_off"]])
    assert np.array_equal(g, exp_off)
    for p in parts:
        assert p["totals"].sum() == len(exp_tok)
    # shards are balanced by bytes, not by document count
    b = __import__("jtokkit_amd.sharding", fromlist=["x"]).shard_by_bytes(doc_off, 2)
    half = doc_off[b[1]]
    assert abs(int(half) - int(doc_off[-1]) // 2) <= 4096
