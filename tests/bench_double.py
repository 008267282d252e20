"""TEST DOUBLE (never shipped, only loaded when LNR_BENCH_DOUBLE is set): stands in for linear_amd.Filter so that bench.py's
argument path, rank start-up (`--gpus N` -> N child ranks), index broadcast protocol, per-rank batches, max-over-ranks timing
and the one-JSON-line contract run on CPU with gloo (tests/test_bench_cli_cpu.py).  Its "device buffers" are CPU tensors and
its "filter" only counts reads -- no result of it is ever reported as a measurement."""
from __future__ import annotations

import time
from types import SimpleNamespace

import numpy as np
import torch


def make_genome():
    rng = np.random.default_rng(5)
    return [rng.integers(0, 4, size=n, dtype=np.uint8) for n in (5000, 3000)]


def sample_reads(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 4, (n * 50,), generator=g, dtype=torch.uint8), torch.arange(n + 1, dtype=torch.int64) * 50


class FilterDouble:
    def __init__(self, device=-1, scratch_budget=0, index_type=1, gap_len=0, dup=0):
        self.blobs, self._seq_len, self.info, self.adopted, self.n = None, None, None, False, 0
        self.seen = []

    def build_index(self, seqs, T=1):
        self._seq_len = np.array([s.size for s in seqs], dtype=np.int64)
        cat = np.concatenate(seqs)
        sizes = [cat.size, 1 << 12, 64 * 8, 32 * 16]
        self.blobs = [torch.from_numpy(cat.copy())] + [torch.full((s,), k + 1, dtype=torch.uint8) for k, s in enumerate(sizes[1:])]
        self.info = np.array([len(seqs), T, sizes[0], sizes[1] // 4, sizes[2] // 8, sizes[3] // 16, 77, 0], dtype=np.int64)
        return self.index_info()

    def index_info(self):
        i = self.info
        return SimpleNamespace(nseq=int(i[0]), layout_threads=int(i[1]), genome_bytes=int(i[2]), dir_len=int(i[3]), hs_len=int(i[4]), f2_len=int(i[5]),
                               n_samples=int(i[6]), build_ms=0.0)

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
        assert int(self.blobs[1][0]) == 1 and int(self.blobs[3][-1]) == 3, "broadcast did not deliver the owner's blobs"

    def filter_batch_dev(self, reads_ptr, off_ptr, n):
        self.n = n
        self.seen.append(reads_ptr)
        time.sleep(0.002)

    seed_lookup_batch_dev = filter_batch_dev

    # the host entry point of the timed region (bench.py: lnr_filter_submit / lnr_filter_wait with two batches in flight)
    def host_alloc(self, nbytes):
        return np.zeros(nbytes, np.uint8)

    def filter_submit(self, reads, off):
        self.inflight = getattr(self, "inflight", 0) + 1
        assert self.inflight <= 3
        self.n = off.size - 1

    def filter_wait(self, copy=True):
        assert self.inflight >= 1
        self.inflight -= 1
        time.sleep(0.002)
        return self.n, 0

    def gap_stream(self, set_to=-1):
        return 0

    def set_gap(self, gap_len, dup=0):
        pass

    def stats(self):
        d = {k: 0 for k in ("reads", "bases", "jobs", "samples", "lookups", "bucket_entries", "anchors", "remap_reads", "cords", "seed_bytes", "prep_ms",
                            "seed_count_ms", "seed_gather_ms", "job_ms", "tail_ms", "total_ms", "seed_count_launches", "seed_gather_launches", "job_launches")}
        d.update(reads=self.n, seed_count_launches=1, seed_count_ms=1.0, seed_bytes=1000)
        return d

    def close(self):
        pass
