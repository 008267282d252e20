"""CPU, world_size 2, gloo: the multi-GPU plumbing of linear_amd.dist -- read sharding and the start-up index
broadcast protocol (meta vector, receiver-side allocation, in-place blob broadcast, adopt) -- exercised with a
test double that keeps its "device buffers" in CPU tensors.  The HIP library itself needs a GPU; what runs here is
exactly the host logic bench.py and a multi-GPU front-end use."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from linear_amd import dist as ldist


def test_shard_range_partitions_in_order():
    for n in (0, 1, 7, 8, 100_003):
        for w in (1, 2, 3, 8):
            r = [ldist.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


class FakeIndexOwner:
    """Duck-types linear_amd.Filter for broadcast_index: four byte blobs + an 8-int info vector + sequence lengths."""

    def __init__(self):
        self.blobs = None
        self.info = None
        self._seq_len = None
        self.adopted = False

    def build(self, seed):
        rng = np.random.default_rng(seed)
        self._seq_len = np.array([1000, 2345], dtype=np.int64)
        sizes = [4096, 1 << 16, 777 * 8, 300 * 16]
        self.blobs = [torch.from_numpy(rng.integers(0, 255, size=s, dtype=np.uint8)) for s in sizes]
        self.info = np.array([2, 3, sizes[0], sizes[1] // 4, sizes[2] // 8, sizes[3] // 16, 99, 0], dtype=np.int64)

    def index_info_vec(self):
        return self.info

    def seq_len(self):
        return self._seq_len

    def index_alloc_from(self, vec8, seq_len):
        self.info = np.asarray(vec8, dtype=np.int64).copy()
        self._seq_len = np.asarray(seq_len, dtype=np.int64).copy()
        sizes = [int(vec8[2]), int(vec8[3]) * 4, int(vec8[4]) * 8, int(vec8[5]) * 16]
        self.blobs = [torch.zeros(s, dtype=torch.uint8) for s in sizes]

    def index_blobs(self):
        return [(b, b.numel()) for b in self.blobs]

    def index_adopt(self):
        self.adopted = True


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f = FakeIndexOwner()
    if rank == 0:
        f.build(1234)
    res = ldist.broadcast_index(f, 0, "cpu")
    ref = FakeIndexOwner()
    ref.build(1234)
    ok = all(torch.equal(a, b) for a, b in zip(f.blobs, ref.blobs)) and np.array_equal(f.info, ref.info) and np.array_equal(f.seq_len(), ref.seq_len())
    ok = ok and (rank == 0 or f.adopted) and res["bytes"] == sum(b.numel() for b in ref.blobs)
    # read sharding: every rank takes its slice, the concatenation restores file order
    n = 1003
    lo, hi = ldist.shard_range(n, rank, world)
    mine = torch.arange(lo, hi, dtype=torch.int64)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([hi - lo]))
    pad = torch.zeros(max(int(s.item()) for s in sizes), dtype=torch.int64)
    pad[: hi - lo] = mine
    parts = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    cat = torch.cat([p[: int(s.item())] for p, s in zip(parts, sizes)])
    ok = ok and torch.equal(cat, torch.arange(n, dtype=torch.int64))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_index_broadcast_and_sharding_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
