"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards by reads: every rank filters its own contiguous slice of the read stream against the
same index, with no data-path collective.  The only exchange is at start-up: the rank that built the
packed index broadcasts its four device buffers (genome bytes, dir, hs, f2) over xGMI; the other ranks
receive them in place into buffers the library allocated (include/linear_amd.h, lnr_index_alloc /
lnr_index_blob / lnr_index_adopt).  SURVEY.md 8(e).

The helpers are written against a small duck-typed "index owner" so the exchange logic is covered by
world_size-2 gloo tests on CPU (tests/test_dist_cpu.py) without a GPU.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, order-preserving split of n_items over world ranks (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class DevBlob:
    """Wraps a raw device pointer so torch can view it without a copy (__cuda_array_interface__)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def blob_tensor(ptr: int, nbytes: int, device) -> torch.Tensor:
    return torch.as_tensor(DevBlob(ptr, nbytes), device=device)


def broadcast_meta(meta: np.ndarray | None, src: int, device) -> np.ndarray:
    """Broadcast a small int64 vector whose length the receivers do not know."""
    n = torch.zeros(1, dtype=torch.int64, device=device)
    if dist.get_rank() == src:
        n[0] = meta.size
    dist.broadcast(n, src)
    t = torch.zeros(int(n.item()), dtype=torch.int64, device=device)
    if dist.get_rank() == src:
        t.copy_(torch.from_numpy(np.ascontiguousarray(meta, dtype=np.int64)))
    dist.broadcast(t, src)
    return t.cpu().numpy()


def broadcast_blobs(tensors: list[torch.Tensor], src: int, chunk_bytes: int = 1 << 30) -> None:
    """In-place broadcast of byte tensors, in chunks of at most chunk_bytes (large, few collectives: the
    xGMI links are point-to-point, so fewer/larger transfers are what keeps them busy)."""
    for t in tensors:
        flat = t.view(-1)
        for s in range(0, flat.numel(), chunk_bytes):
            dist.broadcast(flat[s:s + chunk_bytes], src)


def broadcast_index(flt, src: int, device) -> dict:
    """flt: linear_amd.Filter (or a test double with index_info_vec/seq_len/index_alloc_from/index_blobs/index_adopt).
    After the call every rank holds the identical index.  Returns {'bytes': total, 'seconds': wall}."""
    import time
    rank = dist.get_rank()
    meta = None
    if rank == src:
        meta = np.concatenate([flt.index_info_vec(), np.asarray(flt.seq_len(), dtype=np.int64)])
    meta = broadcast_meta(meta, src, device)
    if rank != src:
        nseq = int(meta[0])
        flt.index_alloc_from(meta[:8], meta[8:8 + nseq])
    blobs = [blob_tensor(p, b, device) if isinstance(p, int) else p for p, b in flt.index_blobs()]
    if device != "cpu" and torch.cuda.is_available():
        torch.cuda.synchronize()
    dist.barrier()
    t0 = time.time()
    broadcast_blobs(blobs, src)
    if device != "cpu" and torch.cuda.is_available():
        torch.cuda.synchronize()
    dist.barrier()
    dt = time.time() - t0
    if rank != src:
        flt.index_adopt()
    return {"bytes": int(sum(b for _, b in flt.index_blobs())), "seconds": dt}
