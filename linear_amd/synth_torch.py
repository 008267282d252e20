"""Device-side synthetic read generator (torch): ONT-profile reads sampled from a reference that already
sits in HBM.  Used by bench.py so that 10^5-10^6 reads are produced in seconds and are resident in HBM when
the timed region starts.  Same error model as synth.mutate (iid per base: 40 % sub / 30 % del / 30 % ins)."""
from __future__ import annotations

import torch

_CPL = None


@torch.no_grad()
def sample_reads_cuda(ref: torch.Tensor, n_reads: int, read_len: int, err: float, seed: int, non_n_start: int = 0,
                      chunk: int = 8192) -> tuple[torch.Tensor, torch.Tensor]:
    """ref: uint8 cuda tensor of Dna5 ordinals.  Returns (bases uint8 [n_reads*read_len], off int64 [n_reads+1]) on the device.
    Reads are drawn from [non_n_start, len) and redrawn (by shifting) if they start inside an N run."""
    dev = ref.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    span = int(read_len * (1 + err)) + 64
    out = torch.empty((n_reads, read_len), dtype=torch.uint8, device=dev)
    cpl = torch.tensor([3, 2, 1, 0, 4], dtype=torch.uint8, device=dev)
    ar = torch.arange(span, device=dev)
    for s in range(0, n_reads, chunk):
        m = min(chunk, n_reads - s)
        pos = torch.randint(non_n_start, ref.numel() - span, (m,), generator=g, device=dev)
        seg = ref[pos[:, None] + ar[None, :]]                                  # [m, span]
        u = torch.rand((m, span), generator=g, device=dev)
        sub = u < err * 0.4
        dele = (u >= err * 0.4) & (u < err * 0.7)
        ins = (u >= err * 0.7) & (u < err)
        rnd = torch.randint(1, 4, (m, span), generator=g, device=dev, dtype=torch.uint8)
        base = torch.where(sub & (seg < 4), (seg + rnd) & 3, seg)
        counts = torch.ones((m, span), dtype=torch.int32, device=dev)
        counts[dele] = 0
        counts[ins] = 2
        start = torch.cumsum(counts, dim=1) - counts                            # output position of each source base
        buf = torch.zeros((m, read_len + 2), dtype=torch.uint8, device=dev)
        rows = torch.arange(m, device=dev)[:, None].expand(m, span)
        keep = (counts > 0) & (start < read_len)
        buf[rows[keep], start[keep]] = base[keep]
        keep2 = ins & (start + 1 < read_len)
        rnd2 = torch.randint(0, 4, (m, span), generator=g, device=dev, dtype=torch.uint8)
        buf[rows[keep2], (start + 1)[keep2]] = rnd2[keep2]
        rd = buf[:, :read_len]
        rc = torch.rand((m,), generator=g, device=dev) < 0.5
        rev = cpl[rd.flip(1).long()]
        out[s:s + m] = torch.where(rc[:, None], rev, rd)
    off = torch.arange(n_reads + 1, device=dev, dtype=torch.int64) * read_len
    return out.reshape(-1), off


# GRCh38 primary assembly: chr1..22, X, Y (bases)
GRCH38_LENS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
               135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
               46709983, 50818468, 156040895, 57227415]
_ACRO = {12: 16_000_000, 13: 16_000_000, 14: 17_000_000, 20: 5_000_000, 21: 10_500_000}   # leading N of the acrocentric p-arms


@torch.no_grad()
def grch38_like_cuda(device, seed: int = 38, scale: float = 1.0, n_families: int = 1500):
    """Stand-in for the GRCh38 primary assembly, generated in HBM (no genome file or network on the box): 24 sequences with
    the human chromosome lengths (x `scale`), telomere / centromere / acrocentric N runs, interspersed repeat families with a
    skewed copy-number spectrum and 2-25 % divergence over ~45 % of the sequence, tandem repeats over ~3 %, intra- and
    inter-chromosomal segmental duplications.  Returns (one uint8 tensor holding all sequences back to back, offsets[25])."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lens = [max(200_000, int(L * scale)) for L in GRCH38_LENS]
    offs = [0]
    for L in lens:
        offs.append(offs[-1] + L)
    total = offs[-1]
    gen = torch.randint(0, 4, (total,), generator=g, device=device, dtype=torch.uint8)
    # repeat families: consensus sequences back to back
    flen = torch.randint(280, 6500, (n_families,), generator=g, device=device)
    fstart = torch.cumsum(flen, 0) - flen
    cons = torch.randint(0, 4, (int(flen.sum().item()),), generator=g, device=device, dtype=torch.uint8)
    divs = torch.tensor([0.02, 0.05, 0.10, 0.15, 0.20, 0.25], device=device)
    fdiv = divs[torch.multinomial(torch.tensor([0.05, 0.10, 0.20, 0.25, 0.25, 0.15], device=device), n_families, replacement=True, generator=g)]
    fw = 1.0 / (torch.arange(n_families, device=device, dtype=torch.float32) + 10.0)   # skewed copy numbers
    SLOT = 8192

    def plant(lo: int, hi: int, p: float, tandem: bool):
        """one round: at most one copy per SLOT-sized slot of [lo, hi), so the copies of a round never overlap (deterministic)"""
        ns = (hi - lo) // SLOT
        if ns < 1:
            return
        sel = torch.nonzero(torch.rand(ns, generator=g, device=device) < p).flatten()
        k = sel.numel()
        if k == 0:
            return
        if tandem:
            ulen = torch.randint(1, 70, (k,), generator=g, device=device)
            clen = torch.randint(100, 4000, (k,), generator=g, device=device)
            units = torch.randint(0, 4, (k, 70), generator=g, device=device, dtype=torch.uint8)
        else:
            fam = torch.multinomial(fw, k, replacement=True, generator=g)
            fl = flen[fam]
            a = (torch.rand(k, generator=g, device=device) * (fl - 300).clamp(min=1)).long()
            b = a + 280 + (torch.rand(k, generator=g, device=device) * (fl - a - 280 + 1).clamp(min=1)).long()
            b = torch.minimum(b, fl)
            clen = (b - a).clamp(min=1)
            rc = torch.rand(k, generator=g, device=device) < 0.5
        off = (torch.rand(k, generator=g, device=device) * (SLOT - clen).clamp(min=1)).long()
        cstart = torch.cumsum(clen, 0) - clen
        tot = int(clen.sum().item())
        cid = torch.repeat_interleave(torch.arange(k, device=device), clen)
        pos = torch.arange(tot, device=device) - cstart[cid]
        if tandem:
            base = units[cid, pos % ulen[cid]]
        else:
            src = torch.where(rc[cid], clen[cid] - 1 - pos, pos) + a[cid] + fstart[fam][cid]
            base = cons[src]
            base = torch.where(rc[cid], 3 - base, base)
            mut = torch.rand(tot, generator=g, device=device) < fdiv[fam][cid]
            base = torch.where(mut, (base + torch.randint(1, 4, (tot,), generator=g, device=device, dtype=torch.uint8)) & 3, base)
        gen[lo + sel[cid] * SLOT + off[cid] + pos] = base

    for c in range(24):
        lo, hi = offs[c], offs[c + 1]
        for rnd in range(5):
            plant(lo + 1234 * rnd, hi, 0.42, False)
        plant(lo + 777, hi, 0.12, True)
    # segmental duplications (1-2 % divergence), within and between chromosomes
    cpu = torch.Generator()
    cpu.manual_seed(seed + 1)
    n_sd = max(4, int(160 * scale))
    for _ in range(n_sd):
        n = int(torch.randint(20_000, 120_000, (1,), generator=cpu).item())
        c1, c2 = (int(v) for v in torch.randint(0, 24, (2,), generator=cpu))
        if torch.rand(1, generator=cpu).item() < 0.6:
            c2 = c1
        if lens[c1] <= n + 2 or lens[c2] <= n + 2:
            continue
        s = offs[c1] + int(torch.randint(0, lens[c1] - n, (1,), generator=cpu).item())
        d = offs[c2] + int(torch.randint(0, lens[c2] - n, (1,), generator=cpu).item())
        cp = gen[s:s + n].clone()
        mut = torch.rand(n, generator=g, device=device) < 0.015
        cp = torch.where(mut, (cp + torch.randint(1, 4, (n,), generator=g, device=device, dtype=torch.uint8)) & 3, cp)
        gen[d:d + n] = cp
    # N runs: telomeres, acrocentric p-arms, a centromere-like gap, a few assembly gaps
    for c in range(24):
        lo, L = offs[c], lens[c]
        lead = int(_ACRO.get(c, 10_000) * scale) if c in _ACRO else min(10_000, L // 50)
        gen[lo:lo + lead] = 4
        gen[lo + L - min(10_000, L // 50):lo + L] = 4
        cen = int(L * 0.4)
        gen[lo + cen:lo + cen + int(L * 0.012)] = 4
        for _ in range(3):
            n = int(torch.randint(1000, 50_000, (1,), generator=cpu).item())
            if L > 4 * n:
                p = int(torch.randint(L // 10, L - n - L // 10, (1,), generator=cpu).item())
                gen[lo + p:lo + p + n] = 4
    return gen, offs


@torch.no_grad()
def sample_reads_multi_cuda(gen: torch.Tensor, offs, n_reads: int, read_len: int, err: float, seed: int,
                            chunk: int = 8192, ends=None) -> tuple[torch.Tensor, torch.Tensor]:
    """As sample_reads_cuda, for a genome of several sequences held in `gen`: sequence i occupies [offs[i], ends[i])
    (ends defaults to offs[1:], i.e. back to back; the library's padded genome blob passes both).  Windows that straddle
    two sequences or touch an N run at their start, middle or end are redrawn."""
    dev = gen.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    span = int(read_len * (1 + err)) + 64
    starts = torch.as_tensor(list(offs[:-1]) if ends is None else list(offs), device=dev, dtype=torch.int64)
    stops = torch.as_tensor(list(offs[1:]) if ends is None else list(ends), device=dev, dtype=torch.int64)
    lo, hi = int(starts[0].item()), int(stops[-1].item())
    out = torch.empty((n_reads, read_len), dtype=torch.uint8, device=dev)
    cpl = torch.tensor([3, 2, 1, 0, 4], dtype=torch.uint8, device=dev)
    ar = torch.arange(span, device=dev)
    probe = torch.tensor([0, span // 4, span // 2, 3 * span // 4, span - 1], device=dev)
    s = 0
    while s < n_reads:
        m = min(chunk, n_reads - s)
        cand = torch.randint(lo, hi - span, (2 * m + 64,), generator=g, device=dev)
        sid = torch.searchsorted(starts, cand, right=True) - 1
        inside = cand + span <= stops[sid]
        clean = (gen[cand[:, None] + probe[None, :]] < 4).all(dim=1)
        pos = cand[inside & clean][:m]
        m = pos.numel()
        if m == 0:
            continue
        seg = gen[pos[:, None] + ar[None, :]]
        u = torch.rand((m, span), generator=g, device=dev)
        sub = u < err * 0.4
        dele = (u >= err * 0.4) & (u < err * 0.7)
        ins = (u >= err * 0.7) & (u < err)
        rnd = torch.randint(1, 4, (m, span), generator=g, device=dev, dtype=torch.uint8)
        base = torch.where(sub & (seg < 4), (seg + rnd) & 3, seg)
        counts = torch.ones((m, span), dtype=torch.int32, device=dev)
        counts[dele] = 0
        counts[ins] = 2
        start = torch.cumsum(counts, dim=1) - counts
        buf = torch.zeros((m, read_len + 2), dtype=torch.uint8, device=dev)
        rows = torch.arange(m, device=dev)[:, None].expand(m, span)
        keep = (counts > 0) & (start < read_len)
        buf[rows[keep], start[keep]] = base[keep]
        keep2 = ins & (start + 1 < read_len)
        rnd2 = torch.randint(0, 4, (m, span), generator=g, device=dev, dtype=torch.uint8)
        buf[rows[keep2], (start + 1)[keep2]] = rnd2[keep2]
        rd = buf[:, :read_len]
        rc = torch.rand((m,), generator=g, device=dev) < 0.5
        out[s:s + m] = torch.where(rc[:, None], cpl[rd.flip(1).long()], rd)
        s += m
    off = torch.arange(n_reads + 1, device=dev, dtype=torch.int64) * read_len
    return out.reshape(-1), off


@torch.no_grad()
def plant_svs_cuda(reads: torch.Tensor, n: int, src_len: int, out_len: int, frac: float, seed: int, chunk: int = 4096) -> tuple[torch.Tensor, torch.Tensor, int]:
    """BASELINE configs[4]'s read profile: a fraction `frac` of the reads gets ONE planted structural variant of 50 bp - 5 kb in its middle --
    deletion, insertion of random bases, tandem duplication or inversion, a quarter each -- applied to reads sampled `src_len` long (the slack
    a deletion eats); every read comes out `out_len` long.  Returns (bases, offsets, number of reads with an SV)."""
    dev = reads.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    src = reads.view(n, src_len)
    out = src[:, :out_len].clone()
    sel = torch.nonzero(torch.rand(n, generator=g, device=dev) < frac).flatten()
    cpl = torch.tensor([3, 2, 1, 0, 4], dtype=torch.uint8, device=dev)
    p = torch.arange(out_len, device=dev)[None, :]
    for s0 in range(0, sel.numel(), chunk):
        rows = sel[s0:s0 + chunk]
        k = rows.numel()
        kind = torch.randint(0, 4, (k, 1), generator=g, device=dev)
        m = (50 + (torch.rand(k, 1, generator=g, device=dev) ** 2) * 4950).long()
        cut = torch.randint(6000, out_len - 6000, (k, 1), generator=g, device=dev)
        a, b = p >= cut, p >= cut + m
        idx_del = p + m * a
        idx_ins = torch.where(b, p - m, p)
        idx_dup = torch.where(a, p - m, p)
        idx_inv = torch.where(a & ~b, 2 * cut + m - 1 - p, p)
        idx = torch.where(kind == 0, idx_del, torch.where(kind == 1, idx_ins, torch.where(kind == 2, idx_dup, idx_inv))).clamp_(0, src_len - 1)
        v = torch.gather(src[rows], 1, idx)
        rnd = torch.randint(0, 4, (k, out_len), generator=g, device=dev, dtype=torch.uint8)
        v = torch.where((kind == 1) & a & ~b, rnd, v)
        v = torch.where((kind == 3) & a & ~b, cpl[v.long()], v)
        out[rows] = v
    off = torch.arange(n + 1, device=dev, dtype=torch.int64) * out_len
    return out.reshape(-1), off, int(sel.numel())
