"""Seeded synthetic genomes and long reads (SURVEY.md §8d inputs).

Bases are SeqAn ``Dna5`` ordinals (A,C,G,T,N = 0..4), one byte per base -- the
layout of ``String<Dna5>`` that the reference hands to its hot path
(reference include/base.h:111-122) and therefore what the C ABI takes.

Everything is a pure function of its seed (numpy ``PCG64``), so tests, the
golden-vector generator and ``bench.py`` regenerate identical inputs anywhere.
"""
from __future__ import annotations

import numpy as np

CPL = np.array([3, 2, 1, 0, 4], dtype=np.uint8)


def revcomp(seq: np.ndarray) -> np.ndarray:
    return CPL[seq[::-1]]


def random_ref(length: int, seed: int) -> np.ndarray:
    """Uniform iid ACGT (config C1 style reference)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 4, size=length, dtype=np.uint8)


def repeat_ref(length: int, seed: int, n_families: int = 12, divergence: float = 0.02,
               frac_repeat: float = 0.45, frac_tandem: float = 0.10) -> np.ndarray:
    """Repeat-rich reference: random backbone + diverged copies of repeat families
    + tandem repeats (the stress set of BASELINE.md §2)."""
    rng = np.random.default_rng(seed)
    ref = rng.integers(0, 4, size=length, dtype=np.uint8)
    fams = [rng.integers(0, 4, size=int(rng.integers(300, 6000)), dtype=np.uint8) for _ in range(n_families)]
    filled = 0
    while filled < frac_repeat * length:
        f = fams[int(rng.integers(0, n_families))]
        cp = f.copy()
        mut = rng.random(cp.size) < divergence
        cp[mut] = (cp[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
        if rng.random() < 0.5:
            cp = revcomp(cp)
        pos = int(rng.integers(0, length - cp.size))
        ref[pos:pos + cp.size] = cp
        filled += cp.size
    filled = 0
    while filled < frac_tandem * length:
        unit = rng.integers(0, 4, size=int(rng.integers(2, 60)), dtype=np.uint8)
        n = int(rng.integers(200, 3000))
        t = np.tile(unit, n // unit.size + 1)[:n]
        pos = int(rng.integers(0, length - n))
        ref[pos:pos + n] = t
        filled += n
    return ref


def add_n_runs(ref: np.ndarray, seed: int, n_runs: int = 3, max_run: int = 2000, lead: int = 0, trail: int = 0) -> np.ndarray:
    """Plant N runs (value 4), optionally telomere-like leading/trailing runs."""
    rng = np.random.default_rng(seed)
    ref = ref.copy()
    for _ in range(n_runs):
        n = int(rng.integers(1, max_run))
        pos = int(rng.integers(0, ref.size - n))
        ref[pos:pos + n] = 4
    if lead:
        ref[:lead] = 4
    if trail:
        ref[ref.size - trail:] = 4
    return ref


def mutate(seg: np.ndarray, err: float, rng: np.random.Generator, split=(0.4, 0.3, 0.3)) -> np.ndarray:
    """iid per-base errors: substitution / deletion / insertion with the given split."""
    if err <= 0:
        return seg.copy()
    n = seg.size
    u = rng.random(n)
    sub = u < err * split[0]
    dele = (u >= err * split[0]) & (u < err * (split[0] + split[1]))
    ins = (u >= err * (split[0] + split[1])) & (u < err)
    out = seg.copy()
    ok = out < 4
    m = sub & ok
    out[m] = (out[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
    counts = np.ones(n, dtype=np.int64)
    counts[dele] = 0
    counts[ins] = 2
    res = np.repeat(out, counts)
    # second copy of an "ins" base becomes a random base
    idx_end = np.cumsum(counts)[ins] - 1
    res[idx_end] = rng.integers(0, 4, size=idx_end.size, dtype=np.uint8)
    return res


def sample_reads(refs: list[np.ndarray], n_reads: int, read_len: int, err: float, seed: int,
                 revcomp_mode: str = "random", len_jitter: float = 0.0) -> tuple[np.ndarray, np.ndarray, list[tuple[int, int, int]]]:
    """Sample reads uniformly from the references.

    revcomp_mode: "random" (50 %), "odd" (odd-numbered reads reverse-complemented, config C1) or "none".
    Returns (concatenated bases, offsets[n+1], truth[(seq_id, pos, strand)]).
    """
    rng = np.random.default_rng(seed)
    sizes = np.array([r.size for r in refs], dtype=np.float64)
    chunks, truth = [], []
    for i in range(n_reads):
        L = read_len if len_jitter <= 0 else max(250, int(read_len * (1 + len_jitter * (rng.random() * 2 - 1))))
        sid = int(rng.choice(len(refs), p=sizes / sizes.sum()))
        ref = refs[sid]
        span = min(ref.size, int(L * (1 + err)) + 64)
        pos = int(rng.integers(0, ref.size - span + 1))
        seg = mutate(ref[pos:pos + span], err, rng)
        seg = seg[:L]
        if revcomp_mode == "random":
            rc = bool(rng.random() < 0.5)
        elif revcomp_mode == "odd":
            rc = bool(i & 1)
        else:
            rc = False
        if rc:
            seg = revcomp(seg)
        chunks.append(seg)
        truth.append((sid, pos, int(rc)))
    off = np.zeros(n_reads + 1, dtype=np.uint64)
    off[1:] = np.cumsum([c.size for c in chunks])
    return np.concatenate(chunks) if chunks else np.zeros(0, np.uint8), off, truth


def pack_reads(read_list: list[np.ndarray]) -> tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(read_list) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([r.size for r in read_list])
    cat = np.concatenate(read_list) if read_list else np.zeros(0, np.uint8)
    return np.ascontiguousarray(cat, dtype=np.uint8), off


def edge_reads(refs: list[np.ndarray], seed: int) -> list[np.ndarray]:
    """Edge cases the reference handles on this path: too-short reads (<=200, mapper.cpp:440),
    just-long-enough reads, junk (unmappable -> remap loop pmpfinder.cpp:2749), reads with N,
    chimeric reads (two loci / inversion -> chainBlocksCords both strands), reads at sequence ends."""
    rng = np.random.default_rng(seed)
    ref = refs[0]
    out = []
    out.append(ref[1000:1100].copy())                       # 100 bp: skipped
    out.append(ref[1000:1200].copy())                       # 200 bp: skipped (needs > 200)
    out.append(ref[1000:1201].copy())                       # 201 bp
    out.append(ref[5000:5300].copy())                       # 300 bp
    out.append(rng.integers(0, 4, size=6000, dtype=np.uint8))  # junk
    out.append(rng.integers(0, 4, size=1500, dtype=np.uint8))  # short junk
    r = mutate(ref[20000:28000], 0.08, rng)
    r[1000:1010] = 4                                        # N run inside a read
    r[4000] = 4
    out.append(r)
    r = ref[30000:36000].copy()
    r[:30] = 4                                              # N at the very start (hashInit skip)
    out.append(r)
    a = mutate(ref[40000:45000], 0.05, rng)
    b = revcomp(mutate(ref[60000:65000], 0.05, rng))        # inversion-like chimera
    out.append(np.concatenate([a, b]))
    a = mutate(ref[100000:104000], 0.03, rng)
    b = mutate(ref[300000:304000], 0.03, rng)               # translocation-like chimera
    out.append(np.concatenate([a, b]))
    a = mutate(ref[200000:204000], 0.03, rng)
    b = mutate(ref[206000:210000], 0.03, rng)               # 2 kb deletion
    out.append(np.concatenate([a, b]))
    a = mutate(ref[220000:223000], 0.03, rng)
    b = mutate(ref[221000:225000], 0.03, rng)               # duplication
    out.append(np.concatenate([a, b]))
    out.append(ref[:3000].copy())                           # sequence start
    out.append(ref[ref.size - 3000:].copy())                # sequence end
    out.append(revcomp(ref[ref.size - 4000:]))              # sequence end, reverse
    out.append(np.full(1000, 4, dtype=np.uint8))            # all N
    out.append(np.zeros(2000, dtype=np.uint8))              # poly-A
    if len(refs) > 1:
        out.append(mutate(refs[1][1000:9000], 0.1, rng))
        out.append(revcomp(mutate(refs[-1][2000:7000], 0.1, rng)))
    return out


def chr22_like(seed: int = 2022, length: int = 50_818_468, lead_n: int = 10_510_000) -> np.ndarray:
    """Stand-in for GRCh38 chr22 (no network / no genome file on the box): same length, the ~10.5 Mb of
    leading N of the p-arm, and a human-like repeat spectrum on the rest -- interspersed repeat families
    with 2-25 % divergence over ~45 % of the sequence, tandem repeats over ~3 %, a few large segmental
    duplications -- so minimizer buckets, anchor counts and the chaining load are neither the repeat-free
    nor the worst-case regime of BASELINE.md section 2."""
    rng = np.random.default_rng(seed)
    ref = rng.integers(0, 4, size=length, dtype=np.uint8)
    body0 = lead_n
    fams = [(rng.integers(0, 4, size=int(rng.integers(280, 6500)), dtype=np.uint8), float(rng.choice([0.02, 0.05, 0.10, 0.15, 0.20, 0.25], p=[0.05, 0.10, 0.20, 0.25, 0.25, 0.15])))
            for _ in range(40)]
    target = 0.45 * (length - body0)
    filled = 0
    while filled < target:
        f, div = fams[int(rng.integers(0, len(fams)))]
        a = int(rng.integers(0, max(1, f.size - 300)))
        b = int(rng.integers(a + 280, f.size + 1))
        cp = f[a:b].copy()
        mut = rng.random(cp.size) < div
        cp[mut] = (cp[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
        if rng.random() < 0.5:
            cp = revcomp(cp)
        pos = int(rng.integers(body0, length - cp.size))
        ref[pos:pos + cp.size] = cp
        filled += cp.size
    filled = 0
    while filled < 0.03 * (length - body0):
        unit = rng.integers(0, 4, size=int(rng.integers(1, 70)), dtype=np.uint8)
        n = int(rng.integers(100, 4000))
        t = np.tile(unit, n // unit.size + 1)[:n]
        pos = int(rng.integers(body0, length - n))
        ref[pos:pos + n] = t
        filled += n
    for _ in range(6):   # segmental duplications, 1-2 % divergence
        n = int(rng.integers(20_000, 120_000))
        src = int(rng.integers(body0, length - n))
        dst = int(rng.integers(body0, length - n))
        cp = ref[src:src + n].copy()
        mut = rng.random(n) < 0.015
        cp[mut] = (cp[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
        ref[dst:dst + n] = cp
    ref[:lead_n] = 4
    for _ in range(8):   # internal assembly gaps
        n = int(rng.integers(1000, 50_000))
        pos = int(rng.integers(body0, length - n))
        ref[pos:pos + n] = 4
    return ref


def plant_repeats(ref: np.ndarray, rng: np.random.Generator, fams, frac: float, lo: int = 0, hi: int | None = None) -> None:
    """Overwrite ~frac of ref[lo:hi) with diverged copies of the (consensus, divergence) families, in place."""
    hi = ref.size if hi is None else hi
    target, filled = frac * (hi - lo), 0
    while filled < target:
        f, div = fams[int(rng.integers(0, len(fams)))]
        a = int(rng.integers(0, max(1, f.size - 300)))
        b = int(rng.integers(min(a + 280, f.size), f.size + 1))
        cp = f[a:b].copy()
        mut = rng.random(cp.size) < div
        cp[mut] = (cp[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
        if rng.random() < 0.5:
            cp = revcomp(cp)
        pos = int(rng.integers(lo, hi - cp.size))
        ref[pos:pos + cp.size] = cp
        filled += cp.size


def scale_refs(seed: int = 38, big: int = 262_500_000, n_small: int = 23) -> list[np.ndarray]:
    """Scale-pinning reference set: 24 sequences, the first longer than any human chromosome
    (262.5 Mb > chr1's 248 Mb: x + 2^20 crosses 2^28, the binning histogram has 8 800 bins), the others 0.4-6 Mb
    (ids up to 23), repeat families shared between the sequences, a telomere-like leading N run and
    assembly gaps in the big one.  ~320 Mb in all: small enough for the reference itself to index here."""
    rng = np.random.default_rng(seed)
    fams = [(rng.integers(0, 4, size=int(rng.integers(280, 6500)), dtype=np.uint8), float(rng.choice([0.02, 0.05, 0.10, 0.15, 0.20])))
            for _ in range(60)]
    lens = [big] + [int(rng.integers(400_000, 6_000_000)) for _ in range(n_small)]
    refs = []
    for k, L in enumerate(lens):
        r = rng.integers(0, 4, size=L, dtype=np.uint8)
        if k == 0:
            # repeats only where the reads come from (both ends and the middle): planting 45 % of 262 Mb would take minutes
            for lo, hi in ((10_000, 6_000_000), (128_000_000, 132_000_000), (L - 8_000_000, L)):
                plant_repeats(r, rng, fams, 0.35, lo, hi)
            r[:10_000] = 4
            for _ in range(5):
                n = int(rng.integers(1000, 60_000))
                p = int(rng.integers(20_000_000, L - 20_000_000))
                r[p:p + n] = 4
        else:
            plant_repeats(r, rng, fams, 0.35)
        refs.append(r)
    return refs


def scale_reads(refs: list[np.ndarray], seed: int = 39, n_per: int = 40, read_len: int = 8000, err: float = 0.10):
    """Reads for scale_refs: from the start, the middle and the last megabases of the big sequence (x beyond 2^28),
    from the highest sequence ids, plus chimeras that join the far end of the big sequence to sequence 23."""
    rng = np.random.default_rng(seed)
    big = refs[0]
    out = []

    def take(ref, lo, hi, L):
        span = int(L * (1 + err)) + 64
        p = int(rng.integers(lo, hi - span))
        s = mutate(ref[p:p + span], err, rng)[:L]
        return revcomp(s) if rng.random() < 0.5 else s
    for lo, hi in ((10_000, 6_000_000), (128_000_000, 132_000_000), (big.size - 8_000_000, big.size)):
        for _ in range(n_per):
            out.append(take(big, lo, hi, int(read_len * (0.5 + rng.random()))))
    for sid in (1, 12, 22, 23):
        for _ in range(n_per // 4):
            out.append(take(refs[sid], 0, refs[sid].size, read_len))
    for _ in range(6):
        out.append(np.concatenate([take(big, big.size - 3_000_000, big.size, 4000), take(refs[23], 0, refs[23].size, 4000)]))
    out.append(big[big.size - 5000:].copy())                 # the very end of the big sequence
    out.append(rng.integers(0, 4, size=7000, dtype=np.uint8))  # junk -> re-map round against the full table
    return pack_reads(out)
