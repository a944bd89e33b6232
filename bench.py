#!/usr/bin/env python3
"""bench.py -- input MB/s of the cl100k_base batch encode path on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path (mark_docs, pretok_split, piece_resolve, bpe_merge, pack -- every
kernel of jtk_batch_encode_device) over one batch of synthetic documents that is already resident in
HBM, ending with the total token count on the host; for N > 1 ranks each step also all-gathers the
per-shard token totals over RCCL and stitches the shard's token offsets into global ones.

N = 1 workload: BASELINE.json configs[1] -- cl100k_base, 100k synthetic English docs (~1 KB each).
N > 1: weak scaling, every rank encodes its own 100k-doc shard (own seed); value = all ranks' input
bytes / max-over-ranks time.  Launch: `python bench.py` or
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)


class _DevArray:
    """Zero-copy view of device memory for torch.as_tensor (__cuda_array_interface__)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def cpu_baseline(text, doc_off, budget_s=12.0, max_threads=16, ordinary=False):
    """Oracle (CPU restatement of GptBytePairEncoding.encode, kind "port") on the host cores, on a
    bounded prefix of the same workload: one task per document on a fixed pool, as the reference's JMH
    harness does (AbstractMultiThreadedBenchmark.java:35-45)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    enc = oracle_lib.get("cl100k_base")
    # the GPU box allots 16 host cores per GPU (the machine itself shows far more)
    cores = min(len(os.sched_getaffinity(0)), max_threads)
    n_docs = len(doc_off) - 1

    def run(nd, threads):
        t0 = time.perf_counter()
        enc.encode_batch(text, doc_off[:nd + 1], threads=threads, ordinary=ordinary, want_tokens=False)
        return time.perf_counter() - t0

    probe = min(n_docs, 400)
    run(probe, cores)                                   # warm-up (page in the table)
    dt = run(probe, cores)
    rate = doc_off[probe] / dt
    nd = int(min(n_docs, max(probe, np.searchsorted(doc_off, rate * budget_s))))
    dt = run(nd, cores)
    one = min(nd, max(probe, nd // cores))
    dt1 = run(one, 1)
    return {
        "value": round(doc_off[nd] / 1e6 / dt, 2), "unit": "MB/s", "cores": cores, "kind": "port",
        "sample": "first %d docs (%.1f MB) of the same corpus, %d threads, one task per doc; "
                  "1 thread on %d docs: %.2f MB/s" % (nd, doc_off[nd] / 1e6, cores, one, doc_off[one] / 1e6 / dt1),
    }


# stage (HIP-event bracket in jtk_batch_kernel_times) -> the kernel it launches, as rocprofv3 names it
STAGE_KERNEL = {"bpe_merge": "k_bpe_merge_all", "piece_resolve": "k_piece_resolve", "pretok_split": "k_pretok_split<1>",
                "pack": "k_pack_tokens", "doc_offsets": "k_doc_offsets"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs-per-gpu", type=int, default=100000)
    ap.add_argument("--workload", choices=["english", "mixed"], default="english")
    ap.add_argument("--encoding", default="cl100k_base")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--ordinary", action="store_true",
                    help="time encodeOrdinary() instead of encode() (skips the special-token check of GptBytePairEncoding.java:52-56)")
    ap.add_argument("--no-verify", action="store_true", help="skip the after-the-clock check of a document sample against the oracle")
    ap.add_argument("--serial-pass", action="store_true",
                    help="after the timed region, 5 more steps strictly one after the other: per-kernel times without overlap")
    ap.add_argument("--inflight", type=int, default=2,
                    help="batches in flight, each on its own HIP stream with its own scratch (1 = strictly one after the other)")
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
    from jtokkit_amd import corpus, sharding

    # ---- workload: this rank's shard, resident in HBM ----------------------------------------------
    if args.workload == "english":
        text, doc_off = corpus.english(args.docs_per_gpu, seed=2 + rank)
        wl = "cl100k_base, %dk synthetic English docs (~1 KB each) per GPU" % (args.docs_per_gpu // 1000)
    else:
        text, doc_off = corpus.mixed(args.docs_per_gpu, seed=3 + rank)
        wl = "%s, %dk mixed UTF-8 docs (emoji + CJK, ~4 KB each) per GPU" % (args.encoding, args.docs_per_gpu // 1000)
    if args.encoding != "cl100k_base":
        wl = wl.replace("cl100k_base", args.encoding)
    n_docs, n_bytes = len(doc_off) - 1, int(doc_off[-1])
    d_text = torch.from_numpy(text).to(dev)
    d_off = torch.from_numpy(doc_off).to(dev)
    enc = jtokkit_amd.get_encoding(args.encoding, device=local_rank)
    # `inflight` batches, each with its own HIP stream and scratch: the kernels of step i + 1 fill the CUs that step i's
    # tail leaves idle (every step still does all of its work; all of them have finished when the clock stops).
    # Nothing in a step waits for the host.
    n_fl = max(1, args.inflight)
    batches = [enc.new_batch() for _ in range(n_fl)]
    streams = [torch.cuda.ExternalStream(b.stream(), device=dev) for b in batches]     # the batches' own streams, seen by torch
    for b in batches:
        b.set_profiling(True)

    def step(i):
        b, st = batches[i % n_fl], streams[i % n_fl]
        with torch.cuda.stream(st):
            b.encode_device(d_text.data_ptr(), d_off.data_ptr(), n_docs, n_bytes, ordinary=args.ordinary, stream=st.cuda_stream, sync=False)
            if world > 1:
                # shard token totals -> every rank; exclusive prefix = this shard's global token offset (same stream)
                _, off_ptr, _ = b.device_result()
                g_off = torch.as_tensor(_DevArray(off_ptr, n_docs + 1, "<i8"), device=dev)
                _, base = sharding.gather_shard_totals(g_off[-1:])
                sharding.stitch_offsets(g_off, base)
        return b

    def run(k, acc):
        pending = []
        for i in range(k):
            pending.append(step(i))
            if len(pending) == n_fl:                          # the oldest step's events: waits for that step only
                b = pending.pop(0)
                if acc is not None:
                    for name, v in b.kernel_times().items():
                        acc[name] = acc.get(name, 0.0) + v
        for b in pending:
            if acc is not None:
                for name, v in b.kernel_times().items():
                    acc[name] = acc.get(name, 0.0) + v

    torch.cuda.synchronize()
    run(args.warmup, None)
    stage_ms = {}
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps, stage_ms)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nt = batches[0].result()[0]

    # --serial-pass, after the clock: a few steps strictly one after the other, for per-kernel times without overlap
    serial_ms = {}
    if n_fl > 1 and args.serial_pass:
        for _ in range(5):
            batches[0].encode_device(d_text.data_ptr(), d_off.data_ptr(), n_docs, n_bytes, ordinary=args.ordinary,
                                     stream=streams[0].cuda_stream, sync=True)
            for name, v in batches[0].kernel_times().items():
                serial_ms[name] = serial_ms.get(name, 0.0) + v / 5

    # max over ranks, sum of bytes
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

    if rank == 0:
        steps = args.steps
        for k in stage_ms:
            stage_ms[k] /= steps
        dom = max((k for k in stage_ms if k in STAGE_KERNEL), key=stage_ms.get)     # stages that are exactly one kernel
        dom_kernel = STAGE_KERNEL.get(dom, dom)
        # HBM bytes of one launch of that kernel from the committed PMC passes (same workload only)
        traffic = None
        try:
            pt = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")))
            if pt.get("workload") == wl and dom_kernel in pt["kernels"]:
                k = pt["kernels"][dom_kernel]
                traffic = int((k["hbm_read_MB"] + k["hbm_write_MB"]) * 1e6)
        except (OSError, ValueError, KeyError):
            traffic = None
        # algorithmic bytes of one batch: input once + int32 tokens once + both offset arrays
        bytes_alg = n_bytes + 4 * nt + 16 * (n_docs + 1)
        achieved = bytes_alg / (stage_ms[dom] * 1e-3) / 1e9
        out = {
            "metric": "input MB/s encoded (cl100k_base), bit-exact vs CPU oracle",
            "value": round(total_bytes * steps / t_max / 1e6, 1),
            "unit": "MB/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(t_max / steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": wl, "docs_per_gpu": n_docs, "bytes_per_gpu": n_bytes, "tokens_per_gpu": int(nt),
                       "sharding": "contiguous doc shards, one per GPU" + ("; RCCL all-gather of shard token totals" if world > 1 else ""),
                       "batches_in_flight": n_fl},
            "roofline": {"bound": "hbm", "kernel": dom_kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(bytes_alg),
                         "avg_launch_ms": round(stage_ms[dom], 4)},
            "kernel_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "kernel_ms_serial": {k: round(v, 4) for k, v in serial_ms.items()} or None,
        }
        if serial_ms:
            # the same kernel alone on the GPU (steps one after the other, after the timed region): its duration without
            # the time-sharing that batches in flight bring
            a = bytes_alg / (serial_ms[dom] * 1e-3) / 1e9
            out["roofline_serial"] = {"kernel": dom_kernel, "achieved": round(a, 2), "frac": round(a / HBM_PEAK_GBS, 5),
                                      "avg_launch_ms": round(serial_ms[dom], 4)}
        if world == 1 and not args.no_verify:
            # after the clock (N = 1; with N > 1 the shard's offsets are global): the last result of every batch in flight, a document sample against the CPU oracle
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib
            o = oracle_lib.get(args.encoding)
            rng = np.random.default_rng(0)
            sample = rng.choice(n_docs, min(n_docs, 500), replace=False)
            for b in batches[:min(n_fl, args.steps + args.warmup)]:
                res = b.fetch()
                assert (res.status == 0).all() and int(res.tok_off[-1]) == len(res.tokens) == nt
                for d in sample:
                    doc = text[doc_off[d]:doc_off[d + 1]].tobytes()
                    exp = o.encode(doc) if not args.ordinary else o.encode_ordinary(doc)
                    if res.doc(int(d)).tolist() != exp:
                        raise SystemExit("bench: document %d differs from the oracle" % int(d))
            out["verified"] = "%d sampled documents of each of the %d batches in flight == CPU oracle" % (len(sample), n_fl)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(text, doc_off, max_threads=args.cpu_threads, ordinary=args.ordinary)
        print(json.dumps(out), flush=True)
    for b in batches:
        b.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
