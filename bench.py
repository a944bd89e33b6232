#!/usr/bin/env python3
"""bench.py -- input MB/s of the cl100k_base batch encode path on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path over one batch of synthetic documents that is already resident in HBM: ONE
jtk_batch_encode_device call (inside it the batch is cut into chunks of whole documents that flow through the library's
scratch sets on their own streams: mark_docs, pretok_split, piece_resolve, bpe_merge, tile_scan, pack, doc_offsets per
chunk), ending with every document's token ids and offsets in HBM.

N = 1 (default): BASELINE.json configs[2] -- cl100k_base, 1M mixed UTF-8 docs (emoji + CJK, ~4 KB each, ~4.1 GB), the largest
single-GPU configuration -- is the headline; the same JSON line carries sub-records for configs[1] (100k English docs),
configs[4] (r50k_base + p50k_base back to back on the 1M-doc corpus), one shard of configs[3] (1.25M English docs), an
end-to-end figure (host buffers in, token ids on the host) and a vocabulary-stress corpus.
N > 1: configs[3] -- 10M English docs (~10 GB) in contiguous document shards, one per rank (strong scaling: the corpus is
fixed, 10M / N docs per GPU), one RCCL all-gather of the per-shard token totals per step for the offset stitch.
Launch: `python bench.py` or `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)
SHARD_DOCS = {"mixed": 25000, "english": 100000}     # documents per generator call (one seed each)


class _DevArray:
    """Zero-copy view of device memory for torch.as_tensor (__cuda_array_interface__)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


# ---- corpora ------------------------------------------------------------------------------------------------------
def _gen_shard(job):
    kind, n_docs, seed = job
    from jtokkit_amd import corpus
    text, doc_off = (corpus.mixed if kind == "mixed" else corpus.english)(n_docs, seed=seed)
    return text, doc_off


def make_corpus(kind, n_docs, seed0, workers):
    """n_docs documents of corpus.mixed / corpus.english, generated in shards of SHARD_DOCS[kind] documents (shard i: seed
    seed0 + i) by a process pool and concatenated in shard order."""
    per = SHARD_DOCS[kind]
    jobs = []
    left, i = n_docs, 0
    while left > 0:
        jobs.append((kind, min(per, left), seed0 + i))
        left -= per
        i += 1
    if len(jobs) == 1 or workers <= 1:
        parts = [_gen_shard(j) for j in jobs]
    else:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:
            parts = []
            for k, part in enumerate(pool.imap(_gen_shard, jobs, chunksize=1)):
                parts.append(part)
                if len(jobs) > 20 and (k + 1) % 10 == 0:
                    print("[bench] corpus shards %d / %d" % (k + 1, len(jobs)), file=sys.stderr, flush=True)
    total = sum(len(t) for t, _ in parts)
    text = np.empty(total, dtype=np.uint8)
    doc_off = np.empty(n_docs + 1, dtype=np.int64)
    doc_off[0] = 0
    pos, d = 0, 0
    for t, off in parts:
        text[pos:pos + len(t)] = t
        doc_off[d + 1:d + len(off)] = off[1:] + pos
        pos += len(t)
        d += len(off) - 1
    return text, doc_off


def vocab_stress_corpus(n_docs, mean_tokens=256, seed=10):
    """Documents made by decoding uniformly random cl100k_base token ids (every rank-table entry equally likely, ~100k
    distinct 'words'): the whole-piece and pair tables are hit all over, not in a small hot set.  Invalid UTF-8 is fine for
    encodeOrdinary on bytes (the oracle and the device treat the bytes alike); ids whose bytes are not valid UTF-8 on their
    own are kept only if the concatenation decodes -- simplest: keep ASCII/valid-UTF-8 entries only."""
    import base64
    path = os.path.join(ROOT, "jtokkit_amd", "data", "cl100k_base.tiktoken")
    toks = []
    for line in open(path, "rb"):
        a, _ = line.split()
        t = base64.b64decode(a)
        try:
            t.decode("utf-8")
        except UnicodeDecodeError:
            continue
        toks.append(t)
    lens = np.array([len(t) for t in toks], dtype=np.int64)
    offs = np.zeros(len(toks) + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    flat = np.frombuffer(b"".join(toks), dtype=np.uint8)
    rng = np.random.default_rng(seed)
    n_tok = n_docs * mean_tokens
    ids = rng.integers(0, len(toks), n_tok)
    l = lens[ids]
    ends = np.cumsum(l)
    total = int(ends[-1])
    idx = np.arange(total, dtype=np.int64)
    idx -= np.repeat(ends - l, l)
    idx += np.repeat(offs[ids], l)
    text = flat[idx]
    doc_off = np.concatenate([[0], ends[mean_tokens - 1::mean_tokens]]).astype(np.int64)
    return np.ascontiguousarray(text), doc_off


# ---- CPU baseline ---------------------------------------------------------------------------------------------------
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(encoding, text, doc_off, max_threads=16, ordinary=False, budget_s=20.0):
    """The oracle (CPU restatement of GptBytePairEncoding.encode, kind "port") on the host cores, the way the reference's JMH
    harness measures (AbstractMultiThreadedBenchmark.java:35-45: every document one task on a fixed pool, wall time for the
    whole corpus; benchmark/build.gradle.kts:20-25: 1 warm-up + 5 measured single-shot iterations), on a bounded prefix of the
    same workload, at 1 thread and at all allotted cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    enc = oracle_lib.get(encoding)
    # the GPU box allots 16 host cores per GPU (the machine itself shows far more)
    cores = min(len(os.sched_getaffinity(0)), max_threads)
    n_docs = len(doc_off) - 1

    def run(nd, threads):
        t0 = time.perf_counter()
        enc.encode_batch(text, doc_off[:nd + 1], threads=threads, ordinary=ordinary, want_tokens=False)
        return time.perf_counter() - t0

    probe = min(n_docs, 200)
    run(probe, cores)                                   # page in the table
    rate = doc_off[probe] / run(probe, cores)
    per_pass = budget_s / 2 / 6                         # 6 passes per thread count, half the budget each
    nd_all = int(min(n_docs, max(probe, np.searchsorted(doc_off, rate * per_pass))))
    nd_one = int(min(n_docs, max(probe, np.searchsorted(doc_off, rate / cores * per_pass))))

    def measure(nd, threads):
        run(nd, threads)                                # 1 warm-up iteration
        ts = sorted(run(nd, threads) for _ in range(5)) # 5 measured iterations
        return doc_off[nd] / 1e6 / ts[2], doc_off[nd] / 1e6 / ts[0], doc_off[nd] / 1e6 / ts[4]
    med_all, best_all, worst_all = measure(nd_all, cores)
    med_one, _, _ = measure(nd_one, 1)
    return {
        "value": round(med_all, 2), "unit": "MB/s", "cores": cores, "kind": "port",
        "cpu_model": cpu_model(), "one_thread_MBps": round(med_one, 2),
        "method": "1 warm-up + 5 timed passes, median (range %.1f-%.1f MB/s); one task per document on a fixed pool" % (worst_all, best_all),
        "sample": "first %d docs (%.1f MB) of the headline corpus on %d threads; first %d docs (%.1f MB) on 1 thread" % (
            nd_all, doc_off[nd_all] / 1e6, cores, nd_one, doc_off[nd_one] / 1e6),
    }


# ---- timing helpers ---------------------------------------------------------------------------------------------------
STAGE_KERNEL = {"bpe_merge": "k_bpe_merge", "piece_resolve": "k_piece_resolve", "pretok_split": "k_pretok_split<1>",
                "pack": "k_pack_tokens", "doc_offsets": "k_doc_offsets", "tile_scan": "k_tile_scan"}


def time_encode(torch, batches, d_text, d_off, n_docs, n_bytes, steps, warmup, ordinary, world=1, dist=None, after_step=None):
    """`warmup` untimed steps, then `steps` timed ones bracketed by barrier + synchronize.  Steps rotate over `batches` (each
    has its own streams and scratch) without waiting for one another; everything has finished when the clock stops.
    Returns (seconds, per-stage ms summed over the chunks of a step, averaged over steps, tokens)."""
    streams = [torch.cuda.ExternalStream(b.stream()) for b in batches]
    acc = {}
    nb = len(batches)

    def run(k, collect):
        pending = []
        for i in range(k):
            b, st = batches[i % nb], streams[i % nb]
            b.encode_device(d_text.data_ptr(), d_off.data_ptr(), n_docs, n_bytes, ordinary=ordinary, sync=False)
            if after_step is not None:
                with torch.cuda.stream(st):
                    after_step(b)
            pending.append(b)
            if len(pending) == nb:                              # the oldest step's events: waits for that step only
                bb = pending.pop(0)
                if collect:
                    for name, v in bb.kernel_times().items():
                        acc[name] = acc.get(name, 0.0) + v
        for bb in pending:
            if collect:
                for name, v in bb.kernel_times().items():
                    acc[name] = acc.get(name, 0.0) + v

    torch.cuda.synchronize()
    run(warmup, False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nt = batches[0].result()[0]
    return dt, {k: v / steps for k, v in acc.items()}, nt


def verify_sample(enc_name, batch, text, doc_off, n_sample, ordinary, seed=0):
    """After the clock: a seeded document sample of the batch's last result against the CPU oracle, bit-exact."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    o = oracle_lib.get(enc_name)
    n_docs = len(doc_off) - 1
    res = batch.fetch()
    assert int(res.tok_off[-1]) == len(res.tokens) and (np.diff(res.tok_off) >= 0).all()
    bad = np.nonzero(res.status != 0)[0]
    if len(bad):
        raise SystemExit("bench: document %d has status %d" % (int(bad[0]), int(res.status[bad[0]])))
    rng = np.random.default_rng(seed)
    sample = np.sort(rng.choice(n_docs, min(n_docs, n_sample), replace=False))
    lens = doc_off[sample + 1] - doc_off[sample]
    s_off = np.zeros(len(sample) + 1, dtype=np.int64)
    np.cumsum(lens, out=s_off[1:])
    idx = np.arange(int(s_off[-1]), dtype=np.int64)
    idx -= np.repeat(s_off[:-1], lens)
    idx += np.repeat(doc_off[sample], lens)
    s_text = text[idx]
    exp_tok, exp_off = o.encode_batch(s_text, s_off, threads=min(16, len(os.sched_getaffinity(0))), ordinary=True)
    for k, d in enumerate(sample):
        got = res.tokens[res.tok_off[d]:res.tok_off[d + 1]]
        if not np.array_equal(got, exp_tok[exp_off[k]:exp_off[k + 1]]):
            raise SystemExit("bench: document %d differs from the oracle" % int(d))
    return len(sample)


def verify_host_sample(enc_name, res, text, doc_off, n_sample, ordinary, seed=3):
    """The same check on a result that already lives in host memory (the end-to-end path)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    o = oracle_lib.get(enc_name)
    n_docs = len(doc_off) - 1
    assert int(res.tok_off[-1]) == len(res.tokens) and (np.diff(res.tok_off) >= 0).all() and (res.status == 0).all()
    rng = np.random.default_rng(seed)
    sample = np.sort(rng.choice(n_docs, min(n_docs, n_sample), replace=False))
    for d in sample:
        doc = text[doc_off[d]:doc_off[d + 1]].tobytes()
        exp = o.encode_ordinary(doc) if ordinary else o.encode(doc)
        if res.tokens[res.tok_off[d]:res.tok_off[d + 1]].tolist() != exp:
            raise SystemExit("bench: document %d of the end-to-end result differs from the oracle" % int(d))
    return len(sample)


def traffic_for(workload_key, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this workload (profiles/pmc_traffic.json), collected
    and corrected as MI355X_MICROARCH.md prescribes (separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x 2 on gfx950)."""
    try:
        pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        k = pt["workloads"][workload_key]["kernels"][kernel]
        return int((k["hbm_read_MB"] + k["hbm_write_MB"]) * 1e6), pt.get("build", "?")
    except (OSError, ValueError, KeyError):
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["cfg3", "cfg2", "cfg4", "vocab"], default=None,
                    help="headline workload (default: cfg3 at N = 1, cfg4 at N > 1)")
    ap.add_argument("--docs", type=int, default=None, help="documents in the headline corpus (default: the config's stated size)")
    ap.add_argument("--encoding", default="cl100k_base")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--ordinary", action="store_true",
                    help="time encodeOrdinary() instead of encode() (skips the special-token check of GptBytePairEncoding.java:52-56)")
    ap.add_argument("--no-verify", action="store_true", help="skip the after-the-clock check of a document sample against the oracle")
    ap.add_argument("--no-subrecords", action="store_true", help="headline only")
    ap.add_argument("--no-cfg4", action="store_true", help="skip the cfg4_full sub-record (10M documents on one GPU: about two minutes)")
    ap.add_argument("--chunk-mb", type=int, default=None, help="JTK_OPT_CHUNK_BYTES in MiB (library default: 1024)")
    ap.add_argument("--in-flight", type=int, default=None, help="JTK_OPT_CHUNKS_IN_FLIGHT (library default: 2)")
    ap.add_argument("--serial", action="store_true", help="one chunk at a time (clean per-kernel times, no overlap)")
    ap.add_argument("--gen-workers", type=int, default=None)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    rehearsal = os.environ.get("JTK_BENCH_REHEARSAL") == "1"     # all ranks on GPU 0 over gloo: exercises the N > 1 code on a one-GPU box
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    import jtokkit_amd
    from jtokkit_amd import _native as N, sharding
    workers = args.gen_workers or max(1, min(16, len(os.sched_getaffinity(0))) // max(1, min(world, 8 if not rehearsal else world)))

    def new_batch(enc):
        b = enc.new_batch()
        if args.chunk_mb:
            b.set_option(N.JTK_OPT_CHUNK_BYTES, args.chunk_mb << 20)
        if args.serial:
            b.set_option(N.JTK_OPT_CHUNKS_IN_FLIGHT, 1)
        elif args.in_flight:
            b.set_option(N.JTK_OPT_CHUNKS_IN_FLIGHT, args.in_flight)
        b.set_option(N.JTK_OPT_REUSE_CHUNK_PLAN, 1)               # the step loop hands the same offsets array every step
        b.set_profiling(True)
        return b

    # ---- headline workload: this rank's documents, resident in HBM ----------------------------------------------------
    wl_name = args.workload or ("cfg3" if world == 1 else "cfg4")
    if wl_name == "cfg3":
        total_docs = args.docs or 1000000
        text, doc_off = make_corpus("mixed", total_docs, 3, workers)
        wl = "%s, %s mixed UTF-8 docs (emoji + CJK, ~4 KB each)" % (args.encoding, "1M" if total_docs == 1000000 else str(total_docs))
        wl_key, scaling = "cfg3", "strong"
    elif wl_name == "cfg2":
        total_docs = args.docs or 100000
        text, doc_off = make_corpus("english", total_docs, 2, workers)
        wl = "%s, %dk synthetic English docs (~1 KB each)" % (args.encoding, total_docs // 1000)
        wl_key, scaling = "cfg2", "strong"
    elif wl_name == "vocab":
        total_docs = args.docs or 100000
        text, doc_off = vocab_stress_corpus(total_docs)
        wl = "%s, %dk vocabulary-stress docs (uniformly random rank-table entries)" % (args.encoding, total_docs // 1000)
        wl_key, scaling = "vocab", "strong"
    else:
        total_docs = args.docs or 10000000
        # contiguous document ranges of the 10M-doc corpus (corpus shards of 100k docs, seed 4 + shard index), one per rank
        n_shards = (total_docs + SHARD_DOCS["english"] - 1) // SHARD_DOCS["english"]
        s0, s1 = rank * n_shards // world, (rank + 1) * n_shards // world
        my_docs = min(total_docs, s1 * SHARD_DOCS["english"]) - s0 * SHARD_DOCS["english"]
        text, doc_off = make_corpus("english", my_docs, 4 + s0, workers)
        wl = "%s, %s English docs (~1 KB each) sharded over %d GPU(s) by contiguous document ranges" % (
            args.encoding, "10M" if total_docs == 10000000 else str(total_docs), world)
        wl_key, scaling = "cfg2", "strong"
    n_docs, n_bytes = len(doc_off) - 1, int(doc_off[-1])
    log("corpus ready: %d docs, %.1f MB" % (n_docs, n_bytes / 1e6))
    d_text = torch.from_numpy(text).to(dev)
    d_off = torch.from_numpy(doc_off).to(dev)
    enc = jtokkit_amd.get_encoding(args.encoding, device=local_rank)
    batch = new_batch(enc)

    after = None
    comm = None
    comm_error = None
    if world > 1 and not rehearsal:
        # the library's own RCCL communicator (jtk_comm_*); the unique id travels over the process group that launched us.
        # Every rank must take the same route: a failure anywhere (no librccl for dlopen, ncclCommInitRank refused) is agreed on
        # with an all-reduce, and the stitch then runs through torch.distributed (RCCL all the same) -- the line says which.
        flag = torch.ones(1, device=dev)
        uid = None
        if rank == 0:
            try:
                uid = sharding.Comm.unique_id()
            except Exception as e:                                   # noqa: BLE001 -- reported in the line
                comm_error, flag[0] = "jtk_comm_unique_id: %s" % e, 0.0
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() > 0:
            idt = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                idt = torch.frombuffer(bytearray(uid), dtype=torch.uint8).to(dev)
            dist.broadcast(idt, 0)
            try:
                comm = sharding.Comm(bytes(idt.cpu().numpy().tobytes()), world, rank, local_rank)
            except Exception as e:                                   # noqa: BLE001
                comm_error, flag[0] = "jtk_comm_create: %s" % e, 0.0
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if flag.item() == 0 and comm is not None:
                comm.close()
                comm = None
        if comm is None:
            log("jtk_comm unavailable (%s): the stitch runs through torch.distributed" % (comm_error or "another rank failed"))
    if comm is not None:
        d_global_off = torch.empty(n_docs + 1, dtype=torch.int64, device=dev)

        def after(b):
            # on the batch's stream, right behind the encode: ncclAllGather of the shard token totals (1 x int64 per rank),
            # base = exclusive prefix, global offsets = local + base -- nothing waits for the host
            _, off_ptr, _ = b.device_result()
            comm.stitch(off_ptr, n_docs, d_global_off.data_ptr(), b.stream())
    elif world > 1:
        def after(b):
            # rehearsal on one GPU (gloo), or jtk_comm unavailable: the same step through torch.distributed
            _, off_ptr, _ = b.device_result()
            g_off = torch.as_tensor(_DevArray(off_ptr, n_docs + 1, "<i8"), device=dev)
            _, base = sharding.gather_shard_totals(g_off[-1:])
            if getattr(b, "_global_off", None) is None:
                b._global_off = torch.empty_like(g_off)
            torch.add(g_off, base, out=b._global_off)           # the stitch: global offsets of this shard's documents

    dt, stage_ms, nt = time_encode(torch, [batch], d_text, d_off, n_docs, n_bytes, args.steps, args.warmup, args.ordinary,
                                   world, dist, after)

    log("timed region done: %.3f s for %d steps" % (dt, args.steps))
    # max over ranks, sums of bytes
    stats = torch.tensor([dt, float(n_bytes), float(nt), float(n_docs)], dtype=torch.float64,
                         device=torch.device("cpu") if rehearsal else dev)
    if world > 1:
        allst = [torch.zeros_like(stats) for _ in range(world)]
        dist.all_gather(allst, stats)
        allst = torch.stack(allst).cpu().numpy()
    else:
        allst = stats.cpu().numpy()[None, :]
    t_max = float(allst[:, 0].max())
    total_bytes = float(allst[:, 1].sum())

    # after the clock: the stitch's result (every rank sees every total; bases are their exclusive prefix)
    if comm is not None:
        totals, base = comm.fetch(batch.stream())
        assert int(totals[rank]) == nt and int(base) == int(totals[:rank].sum())
        g = d_global_off.cpu().numpy()
        assert g[0] == base and g[-1] == base + nt
    # after the clock: this rank's result against the oracle (1 % of the documents, at most 10,000)
    verified = None
    if not args.no_verify:
        ns = verify_sample(args.encoding, batch, text, doc_off, min(10000, max(500, n_docs // 100)), args.ordinary)
        verified = "%d sampled documents per rank == CPU oracle, bit-exact; offsets monotone; all status 0" % ns
        if world > 1:
            ok = torch.tensor([1.0], device=torch.device("cpu") if rehearsal else dev)
            dist.all_reduce(ok)
            assert int(ok.item()) == world

    out = None
    if rank == 0:
        steps = args.steps
        dom = max((k for k in stage_ms if k in STAGE_KERNEL), key=stage_ms.get)     # stages that are one kernel (bpe_merge: its lean kernel dominates)
        dom_kernel = STAGE_KERNEL[dom]
        # algorithmic bytes of one batch: input once + int32 tokens once + both offset arrays
        bytes_alg = n_bytes + 4 * nt + 16 * (n_docs + 1)
        n_launch = max(1, len(batch_chunks(batch, n_bytes, args)))
        avg_launch_ms = stage_ms[dom] / n_launch
        achieved = bytes_alg / n_launch / (avg_launch_ms * 1e-3) / 1e9
        traffic, traffic_build = traffic_for(wl_key, dom_kernel)
        out = {
            "metric": "input MB/s encoded (cl100k_base), HBM-resident input and output, bit-exact vs CPU oracle",
            "value": round(total_bytes * steps / t_max / 1e6, 1),
            "unit": "MB/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(t_max / steps * 1e3, 4),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": wl, "docs": int(allst[:, 3].sum()), "bytes": int(total_bytes), "tokens": int(allst[:, 2].sum()),
                       "docs_per_gpu": n_docs, "bytes_per_gpu": n_bytes,
                       "encode": "encodeOrdinary()" if args.ordinary else "encode() incl. the special-token check",
                       "chunking": "%d chunks of whole documents per step (~%d MiB each), %s in flight on their own streams" % (
                           n_launch, (args.chunk_mb or 1024), "1" if args.serial else str(args.in_flight or 2)),
                       "sharding": ("contiguous document ranges, one per GPU; RCCL all-gather of the shard token totals per step"
                                    if world > 1 else "single GPU")},
            "roofline": {"bound": "hbm", "kernel": dom_kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_source": ("profiles/pmc_traffic.json [%s], build %s, per launch (one chunk)" % (wl_key, traffic_build)) if traffic else None,
                         "algorithmic_bytes_per_launch": int(bytes_alg / n_launch),
                         "avg_launch_ms": round(avg_launch_ms, 4), "launches_per_step": n_launch,
                         "whole_step_frac": round(bytes_alg / (t_max / steps) / 1e9 / HBM_PEAK_GBS, 5)},
            "kernel_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "verified": verified,
        }
        if world > 1:
            # what every rank did (max over ranks is the clock): bytes and seconds per rank, the communicator the stitch ran on
            out["per_rank"] = {"MBps": [round(float(r[1]) * steps / float(r[0]) / 1e6, 1) for r in allst],
                               "bytes": [int(r[1]) for r in allst], "tokens": [int(r[2]) for r in allst]}
            out["rccl_world"] = comm.comm_world() if comm is not None else None
            out["stitch"] = ("jtk_comm_stitch: ncclAllGather of the shard token totals (1 x int64 per rank) on the batch's stream, every step"
                             if comm is not None else "rehearsal on one GPU: torch.distributed (gloo) all_gather of the totals" if rehearsal
                             else "torch.distributed all_gather_into_tensor (RCCL) of the totals on the batch's stream; jtk_comm failed: %s" % comm_error)
        else:
            out["scaling_note"] = ("N = 1 times configs[2]; the N > 1 lines time configs[3] (10M English docs) sharded over N GPUs: the "
                                   "one-GPU point of that curve is the cfg4_full sub-record of this line")
    if world == 1 and not args.no_subrecords:
        sub = {}
        batch.close()                                                   # (its scratch: ~74 GB for 1 GiB chunks)
        dev_arrays = {"text": d_text, "off": d_off}
        del d_text, d_off
        subrecords(sub, torch, dev, args, jtokkit_amd, new_batch, text, doc_off, dev_arrays, workers, wl_name)
        out.update(sub)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            log("CPU baseline")
            out["cpu_baseline"] = cpu_baseline(args.encoding, text, doc_off, max_threads=args.cpu_threads, ordinary=args.ordinary)
        print(json.dumps(out), flush=True)
    batch.close()
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def per_call_record(args):
    """tools/percall/percall_bench (built by __graft_entry__.build()): jtk_encode per document (direct), jtk_service_encode
    (blocking callers coalesced into device batches), jtk_service_submit/wait (many documents in flight per thread) and, as
    the CPU baseline of this shape, the oracle's per-call encode from the same number of threads."""
    from jtokkit_amd import corpus
    exe = os.path.join(ROOT, "tools", "percall", "percall_bench")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "-s"])
    rec = {"workload": "configs[0] corpus (1k short ASCII sentences) and 20k docs of configs[1], one call per document",
           "unit": "documents/s (MB/s of input)", "runs": []}
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    for name, (t, o) in (("cfg1", corpus.sentences(1000)), ("cfg2", corpus.english(20000))):
        path = "/tmp/jtk_percall_%s_%d.bin" % (name, os.getpid())
        with open(path, "wb") as f:
            f.write(np.int64(len(o) - 1).tobytes())
            f.write(o.tobytes())
            f.write(t.tobytes())
        for threads, in_flight in ((16, 1), (64, 1), (4, 1024), (8, 4096), (2, 8192)):
            p = subprocess.run([exe, os.path.join(ROOT, "jtokkit_amd", "libjtokkit_amd.so"),
                                "-" if args.no_cpu_baseline else os.path.join(ROOT, "oracle", "libjtk_oracle.so"),
                                os.path.join(ROOT, "jtokkit_amd", "data", "cl100k_base.tiktoken"), path, str(threads), str(in_flight), "1"],
                               capture_output=True, text=True, env=env, timeout=120)
            if p.returncode != 0:
                rec["runs"].append({"corpus": name, "threads": threads, "error": p.stderr[-300:]})
                continue
            r = json.loads(p.stdout)
            r["corpus"] = name
            rec["runs"].append(r)
        os.unlink(path)
    return rec


def batch_chunks(batch, n_bytes, args):
    """The chunk count the library used for a batch of n_bytes (same rule as jtk_batch_encode_device)."""
    cb = (args.chunk_mb or 1024) << 20
    if n_bytes <= cb + cb // 4:
        return [0]
    return list(range((n_bytes + cb - 1) // cb))


def log(msg):
    """Progress on stderr (a long run must not look hung; stdout carries the one JSON line)."""
    print("[bench %6.1f s] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def pcie_rates(torch, dev, mb=1024):
    """Pinned host <-> device copy rates of this box (one direction at a time), for the end-to-end record."""
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    d = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    out = {}
    for name, (dst, src) in (("h2d", (d, h)), ("d2h", (h, d))):
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        out[name] = (mb << 20) * 3 / (time.perf_counter() - t0) / 1e9
    return out


def subrecords(out, torch, dev, args, jtokkit_amd, new_batch, text, doc_off, dev_arrays, workers, wl_name):
    """N = 1: the other BASELINE.json configs and the measurements next to the headline, each checked against the oracle
    after its clock."""
    n_docs, n_bytes = len(doc_off) - 1, int(doc_off[-1])
    d_text, d_off = dev_arrays["text"], dev_arrays["off"]
    enc = jtokkit_amd.get_encoding("cl100k_base", device=0)

    def rate(nb, dt, steps):
        return round(nb * steps / dt / 1e6, 1)

    # ---- configs[1]: 100k English docs, HBM-resident; one batch alone and two batches in flight (round 1's headline mode)
    log("sub-records: configs[1]")
    if wl_name != "cfg2":
        t2, o2 = make_corpus("english", 100000, 2, workers)
        d_t2, d_o2 = torch.from_numpy(t2).to(dev), torch.from_numpy(o2).to(dev)
        b1, b2 = new_batch(enc), new_batch(enc)
        dt1, st1, nt2 = time_encode(torch, [b1], d_t2, d_o2, len(o2) - 1, len(t2), 20, 3, args.ordinary)
        ns = None if args.no_verify else verify_sample("cl100k_base", b1, t2, o2, 1000, args.ordinary)
        dt2, _, _ = time_encode(torch, [b1, b2], d_t2, d_o2, len(o2) - 1, len(t2), 20, 4, args.ordinary)
        alg = len(t2) + 4 * nt2 + 16 * len(o2)
        dom = max((k for k in st1 if k in STAGE_KERNEL), key=st1.get)
        n_launch = len(batch_chunks(b1, len(t2), args))
        out["cfg2"] = {"workload": "cl100k_base, 100k synthetic English docs (~1 KB each), 1xMI355X", "bytes": len(t2), "tokens": int(nt2),
                       "value": rate(len(t2), dt1, 20), "unit": "MB/s", "ms_per_step": round(dt1 / 20 * 1e3, 4),
                       "two_batches_in_flight": {"value": rate(len(t2), dt2, 20), "ms_per_step": round(dt2 / 20 * 1e3, 4)},
                       "kernel_ms": {k: round(v, 4) for k, v in st1.items()},
                       "roofline": {"kernel": STAGE_KERNEL[dom], "achieved": round(alg / (st1[dom] * 1e-3) / 1e9, 2),
                                    "frac": round(alg / (st1[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                    "traffic": traffic_for("cfg2", STAGE_KERNEL[dom])[0], "launches_per_step": n_launch,
                                    "whole_step_frac": round(alg / (dt1 / 20) / 1e9 / HBM_PEAK_GBS, 5)},
                       "verified": None if ns is None else "%d sampled documents == CPU oracle" % ns}
        # ---- end to end: host buffers in (pinned), token ids on the host (pinned), PCIe both ways inside the clock
        hb = jtokkit_amd.HostBuffer(len(t2))
        hb.array[:] = t2
        be = new_batch(enc)
        be.set_option(jtokkit_amd._native.JTK_OPT_HOST_CHUNK_BYTES, 16 << 20)
        be.set_option(jtokkit_amd._native.JTK_OPT_CHUNKS_IN_FLIGHT, 3)
        for _ in range(2):
            be.encode_host(hb.array, o2, ordinary=args.ordinary, to_host=True)
        t0 = time.perf_counter()
        for _ in range(5):
            be.encode_host(hb.array, o2, ordinary=args.ordinary, to_host=True)
        dte = (time.perf_counter() - t0) / 5
        res = be.host_result()
        if not args.no_verify:
            r1 = b1.fetch()
            assert np.array_equal(res.tokens, r1.tokens) and np.array_equal(res.tok_off, r1.tok_off)
        for _ in range(1):
            be.encode_host(t2, o2, ordinary=args.ordinary)
        t0 = time.perf_counter()
        for _ in range(3):
            be.encode_host(t2, o2, ordinary=args.ordinary)
            be.fetch()
        dtp = (time.perf_counter() - t0) / 3
        out["end_to_end"] = {"workload": "configs[1] corpus: host buffers in -> token ids, offsets and status in host memory, every step",
                             "pinned": {"value": round(len(t2) / dte / 1e6, 1), "unit": "MB/s of input", "ms_per_step": round(dte * 1e3, 3),
                                        "path": "jtk_host_alloc input, JTK_ENCODE_TO_HOST (16 MiB chunks, 3 in flight: H2D, kernels and D2H overlap), result read in place",
                                        "bytes_over_pcie": int(len(t2) + 4 * nt2 + 20 * len(o2))},
                             "pageable": {"value": round(len(t2) / dtp / 1e6, 1), "unit": "MB/s of input", "ms_per_step": round(dtp * 1e3, 3),
                                          "path": "numpy arrays in, jtk_batch_encode + jtk_batch_fetch into numpy arrays"},
                             "verified": None if args.no_verify else "identical to the HBM-resident result"}
        hb.close()
        for b in (b1, b2, be):
            b.close()
        del d_t2, d_o2

    # ---- configs[4]: r50k_base then p50k_base back to back on the 1M-doc corpus (rank-table swap)
    log("sub-records: configs[4]")
    if wl_name == "cfg3":
        encs = {n: jtokkit_amd.get_encoding(n, device=0) for n in ("r50k_base", "p50k_base")}
        bs = {n: new_batch(e) for n, e in encs.items()}
        alone = {}
        for n in encs:
            dt_n, _, nt_n = time_encode(torch, [bs[n]], d_text, d_off, n_docs, n_bytes, 2, 1, args.ordinary)
            alone[n] = (dt_n / 2, nt_n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            for n in ("r50k_base", "p50k_base"):
                bs[n].encode_device(d_text.data_ptr(), d_off.data_ptr(), n_docs, n_bytes, ordinary=args.ordinary, sync=False)
                bs[n].result()                                   # strictly back to back: the next table's pass starts after this one's end
        torch.cuda.synchronize()
        dt_bb = (time.perf_counter() - t0) / reps
        ver = None
        if not args.no_verify:
            ver = {n: verify_sample(n, bs[n], text, doc_off, 2000, args.ordinary, seed=5) for n in encs}
        out["cfg5"] = {"workload": "r50k_base + p50k_base back to back on the 1M-doc mixed corpus (two passes, one per rank table)",
                       "value": round(2 * n_bytes / dt_bb / 1e6, 1), "unit": "MB/s (bytes of both passes / time of both)",
                       "ms_per_pair": round(dt_bb * 1e3, 3),
                       "alone_ms": {n: round(v[0] * 1e3, 3) for n, v in alone.items()}, "tokens": {n: int(v[1]) for n, v in alone.items()},
                       "swap_ms": round((dt_bb - sum(v[0] for v in alone.values())) * 1e3, 3),
                       "swap": "both encodings' device tables stay resident (a few MB each); switching is passing another table pointer to "
                               "the same kernels, so the swap cost is the back-to-back pair minus the two passes alone (cold L2 for the new tables)",
                       "verified": None if ver is None else "%d sampled documents per encoding == CPU oracle" % min(ver.values())}
        for b in bs.values():
            b.close()

    # ---- end to end on the headline workload: pinned host text in -> token ids, offsets and status in pinned host memory; H2D, kernels
    # and D2H overlap chunk by chunk.  It moves 1.7 x as many bytes up as down, so the D2H link bounds it.
    log("sub-records: end to end on the headline corpus")
    if wl_name == "cfg3":
        link = pcie_rates(torch, dev)
        hb = jtokkit_amd.HostBuffer(n_bytes)
        hb.array[:] = text
        be = new_batch(enc)
        be.set_option(jtokkit_amd._native.JTK_OPT_HOST_CHUNK_BYTES, 64 << 20)
        be.set_option(jtokkit_amd._native.JTK_OPT_CHUNKS_IN_FLIGHT, 3)
        be.encode_host(hb.array, doc_off, ordinary=args.ordinary, to_host=True)
        t0 = time.perf_counter()
        for _ in range(2):
            be.encode_host(hb.array, doc_off, ordinary=args.ordinary, to_host=True)
        dte = (time.perf_counter() - t0) / 2
        res = be.host_result()
        nt_e = int(res.tok_off[-1])
        ver = None
        if not args.no_verify:
            ver = verify_host_sample("cl100k_base", res, text, doc_off, 3000, args.ordinary)
        up = 4 * nt_e + 12 * (n_docs + 1)
        out["end_to_end_cfg3"] = {
            "workload": "the headline corpus: pinned host buffers in -> token ids, offsets and status in pinned host memory, every step",
            "value": round(n_bytes / dte / 1e6, 1), "unit": "MB/s of input", "ms_per_step": round(dte * 1e3, 2),
            "bytes_host_to_device": int(n_bytes + 8 * (n_docs + 1)), "bytes_device_to_host": int(up),
            "pcie_GBps_measured": {k: round(v, 1) for k, v in link.items()},
            "d2h_link_fraction": round(up / dte / 1e9 / link["d2h"], 3), "h2d_link_fraction": round(n_bytes / dte / 1e9 / link["h2d"], 3),
            "path": "jtk_host_alloc input, JTK_ENCODE_TO_HOST (64 MiB chunks, 3 in flight), result read in place",
            "verified": None if ver is None else "%d sampled documents == CPU oracle" % ver}
        hb.close()
        be.close()

        # ---- Encoding.encode(text, maxTokens) over the first 200k documents of the headline corpus: the early exit (leading bytes
        # only) beside encoding every document whole and truncating afterwards.  Host buffers in, host arrays out, both ways.
        log("sub-records: maxTokens early exit")
        nd_m, mx = min(n_docs, 200000), 10
        off_m = doc_off[:nd_m + 1]
        text_m = text[:int(off_m[-1])]
        bm = new_batch(enc)
        bm.encode_max_tokens(text_m, off_m, mx, ordinary=True)             # (buffers grow on the first call)
        t0 = time.perf_counter()
        tk_m, kept_m, flag_m, st_m = bm.encode_max_tokens(text_m, off_m, mx, ordinary=True)
        dt_early = time.perf_counter() - t0
        bm.encode_host(text_m, off_m, ordinary=True)
        t0 = time.perf_counter()
        bm.encode_host(text_m, off_m, ordinary=True)
        kept_w, flag_w = bm.truncate(mx)
        res_w = bm.fetch()
        dt_whole = time.perf_counter() - t0
        assert np.array_equal(kept_m, kept_w) and np.array_equal(flag_m, flag_w) and (st_m == 0).all()
        pick = np.random.default_rng(3).choice(nd_m, 5000, replace=False)
        for d in pick:
            assert np.array_equal(tk_m[d, :kept_m[d]], res_w.tokens[res_w.tok_off[d]:res_w.tok_off[d] + kept_w[d]])
        out["max_tokens"] = {
            "workload": "encodeOrdinary(text, %d) for the first %d documents of the headline corpus (%.0f MB)" % (mx, nd_m, len(text_m) / 1e6),
            "early_exit_s": round(dt_early, 4), "whole_then_truncate_s": round(dt_whole, 4), "speedup": round(dt_whole / dt_early, 1),
            "docs_per_s": round(nd_m / dt_early), "unit": "documents/s (early exit)",
            "verified": "kept counts and truncated flags of all documents, ids of 5000 sampled documents == encoding whole then truncating"}
        del res_w
        bm.close()

    # ---- configs[3] on ONE GPU: the N = 1 point of the 10M-doc strong-scaling curve (the N > 1 lines of this bench time the same
    # corpus sharded over N GPUs by contiguous document ranges), chunked through the same scratch sets
    if wl_name == "cfg3" and not args.no_cfg4:
        dev_arrays.clear()                                            # (the headline corpus leaves the device: 10 GB of text follow)
        del d_text, d_off
        torch.cuda.empty_cache()
        log("sub-records: configs[3] on one GPU (generating 10M documents)")
        t0 = time.perf_counter()
        t4, o4 = make_corpus("english", 10000000, 4, workers)
        gen_s = time.perf_counter() - t0
        log("10M documents generated in %.0f s" % gen_s)
        d_t4, d_o4 = torch.from_numpy(t4).to(dev), torch.from_numpy(o4).to(dev)
        b4 = new_batch(enc)
        dt4, st4, nt4 = time_encode(torch, [b4], d_t4, d_o4, len(o4) - 1, len(t4), 3, 1, args.ordinary)
        ns = None if args.no_verify else verify_sample("cl100k_base", b4, t4, o4, 10000, args.ordinary, seed=7)
        out["cfg4_full"] = {"workload": "configs[3] on one GPU: cl100k_base, 10M English docs (~1 KB each), %d chunks of ~%d MiB" % (
                                len(batch_chunks(b4, len(t4), args)), args.chunk_mb or 1024),
                            "bytes": len(t4), "tokens": int(nt4), "value": rate(len(t4), dt4, 3), "unit": "MB/s",
                            "ms_per_step": round(dt4 / 3 * 1e3, 3), "kernel_ms": {k: round(v, 3) for k, v in st4.items()},
                            "corpus_generation_s": round(gen_s, 1),
                            "verified": None if ns is None else "%d sampled documents == CPU oracle" % ns}
        b4.close()
        del d_t4, d_o4, t4, o4
        torch.cuda.empty_cache()

    # ---- the reference's per-call shape (one Encoding.encode call per document from a pool of threads,
    # benchmark/.../AbstractMultiThreadedBenchmark.java:35-45) through the C ABI, driven by native threads
    if wl_name == "cfg3":
        log("sub-records: per-call shapes")
        out["per_call"] = per_call_record(args)

    # ---- vocabulary stress: documents of uniformly random rank-table entries (the lookups miss the caches realistically)
    if wl_name != "vocab":
        log("sub-records: vocabulary stress")
        tv, ov = vocab_stress_corpus(100000)
        d_tv, d_ov = torch.from_numpy(tv).to(dev), torch.from_numpy(ov).to(dev)
        bv = new_batch(enc)
        dtv, stv, ntv = time_encode(torch, [bv], d_tv, d_ov, len(ov) - 1, len(tv), 10, 2, True)
        ns = None if args.no_verify else verify_sample("cl100k_base", bv, tv, ov, 500, True, seed=9)
        out["vocab_stress"] = {"workload": "100k docs of 256 uniformly random cl100k_base rank-table entries each (~100k distinct words)",
                               "bytes": len(tv), "tokens": int(ntv), "value": rate(len(tv), dtv, 10), "unit": "MB/s",
                               "ms_per_step": round(dtv / 10 * 1e3, 4), "kernel_ms": {k: round(v, 4) for k, v in stv.items()},
                               "verified": None if ns is None else "%d sampled documents == CPU oracle" % ns}
        bv.close()


if __name__ == "__main__":
    main()
