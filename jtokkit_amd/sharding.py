"""Multi-GPU sharding of a document batch.

Documents are independent (every Encoding.encode call is a pure function of one string,
reference GptBytePairEncoding.java:71-103), so a batch shards as contiguous document ranges balanced by
bytes, one range per rank / GPU, each rank holding its own copy of the rank tables.  The only exchange
is one all-gather of the per-shard token totals (RCCL over xGMI when the backend is "nccl"; gloo in
the CPU tests); the exclusive prefix of the totals is the shard's global token offset.
"""
import ctypes as C

import numpy as np


def shard_plan(doc_off, world_size):
    """jtk_shard_plan (C ABI): the same byte-balanced contiguous ranges as shard_by_bytes, computed by the library."""
    from . import _native as N
    doc_off = np.ascontiguousarray(doc_off, dtype=np.int64)
    bounds = np.zeros(world_size + 1, dtype=np.int64)
    rc = N.lib().jtk_shard_plan(doc_off.ctypes.data, len(doc_off) - 1, world_size, bounds.ctypes.data)
    if rc != 0:
        raise RuntimeError("jtk_shard_plan: " + N.last_error())
    return [int(b) for b in bounds]


class Comm:
    """jtk_comm: an RCCL communicator (one rank per process / GPU) for the offset stitch of a sharded batch."""

    @staticmethod
    def unique_id():
        from . import _native as N
        buf = (C.c_uint8 * 128)()
        rc = N.lib().jtk_comm_unique_id(buf)
        if rc != 0:
            raise RuntimeError("jtk_comm_unique_id: " + N.last_error())
        return bytes(buf)

    def __init__(self, unique_id, world, rank, device):
        from . import _native as N
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        rc = N.lib().jtk_comm_create(buf, world, rank, device, C.byref(h))
        if rc != 0:
            raise RuntimeError("jtk_comm_create: " + N.last_error())
        self._h = h
        self.world, self.rank = world, rank

    def comm_world(self):
        """jtk_comm_world: the number of ranks of the RCCL communicator itself."""
        from . import _native as N
        return int(N.lib().jtk_comm_world(self._h))

    def stitch(self, d_tok_off_ptr, n_docs, d_global_off_ptr, stream):
        """Queued on `stream`: all-gather of the shard totals, base, global offsets.  Returns device pointers (totals, base)."""
        from . import _native as N
        a, b = C.c_void_p(), C.c_void_p()
        rc = N.lib().jtk_comm_stitch(self._h, d_tok_off_ptr, n_docs, d_global_off_ptr, stream, C.byref(a), C.byref(b))
        if rc != 0:
            raise RuntimeError("jtk_comm_stitch: " + N.last_error())
        return a.value, b.value

    def fetch(self, stream):
        from . import _native as N
        totals = np.zeros(self.world, dtype=np.int64)
        base = C.c_int64(0)
        rc = N.lib().jtk_comm_fetch(self._h, stream, totals.ctypes.data, C.byref(base))
        if rc != 0:
            raise RuntimeError("jtk_comm_fetch: " + N.last_error())
        return totals, base.value

    def close(self):
        if getattr(self, "_h", None):
            from . import _native as N
            N.lib().jtk_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:          # (interpreter shutdown: the module's globals may be gone)
            pass


def shard_by_bytes(doc_off, world_size):
    """Contiguous document ranges balanced by bytes: returns world_size + 1 document indices."""
    doc_off = np.asarray(doc_off)
    total = int(doc_off[-1])
    n_docs = len(doc_off) - 1
    bounds = [0]
    for r in range(1, world_size):
        bounds.append(int(np.searchsorted(doc_off, total * r // world_size, side="left")))
    bounds.append(n_docs)
    bounds = np.maximum.accumulate(np.clip(bounds, 0, n_docs))
    return [int(b) for b in bounds]


def local_shard(text, doc_off, rank, world_size):
    """This rank's slice of a packed batch: (text_slice, doc_off rebased to 0, first_doc)."""
    b = shard_by_bytes(doc_off, world_size)
    d0, d1 = b[rank], b[rank + 1]
    lo, hi = int(doc_off[d0]), int(doc_off[d1])
    return text[lo:hi], np.asarray(doc_off[d0:d1 + 1]) - lo, d0


def gather_shard_totals(n_tokens_local, group=None, device=None):
    """All-gather of one int64 per rank; returns (totals[world], base offset of this rank).
    n_tokens_local: an int, or a 1-element device tensor (e.g. the last entry of the shard's token offsets)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if isinstance(n_tokens_local, torch.Tensor):          # already on the device: no host round trip
        mine = n_tokens_local.reshape(1).to(dtype=torch.int64).clone()
        device = mine.device
    else:
        mine = torch.tensor([int(n_tokens_local)], dtype=torch.int64, device=device)
    if dist.get_backend(group) == "gloo":                 # CPU rehearsal of the multi-GPU path (tests, one-GPU boxes)
        parts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(parts, mine.cpu(), group=group)
        totals = torch.cat(parts).to(mine.device)
    else:
        totals = torch.zeros(world, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(totals, mine, group=group)
    base = totals[:rank].sum()
    return totals, base


def stitch_offsets(local_tok_off, base):
    """Global token offsets of this shard's documents = local offsets + the shard's base (in place for
    torch tensors, a new array for numpy)."""
    if isinstance(local_tok_off, np.ndarray):
        return local_tok_off + int(base)
    return local_tok_off.add_(base)
