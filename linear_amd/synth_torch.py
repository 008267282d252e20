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
