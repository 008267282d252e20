#!/usr/bin/env python3
"""GPU probe: GRCh38-like stand-in at a given scale -> index build -> a few batches through the device entry point.
Prints index sizes, per-read counters and stage times (exploration tool; bench.py is the contract)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from linear_amd import Filter, api
if os.environ.get("LNR_PROBE_LIB"):
    api.SO = os.path.abspath(os.environ["LNR_PROBE_LIB"])   # A/B runs of several builds on one box
from linear_amd.synth_torch import grch38_like_cuda, sample_reads_multi_cuda

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--reads", type=int, default=100_000)
ap.add_argument("--read-len", type=int, default=10_000)
ap.add_argument("--batches", type=int, default=3)
ap.add_argument("--T", type=int, default=16)
ap.add_argument("--seed-only", action="store_true")
ap.add_argument("--chr22", action="store_true", help="the round-1 workload (configs[1]) instead of the GRCh38 stand-in")
ap.add_argument("--index-type", type=int, default=1, help="the reference's -i: 1 DIndex, 2 HIndex")
ap.add_argument("--check", type=int, default=0, help="compare the first N reads of batch 0 with the oracle (builds the oracle's own index)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
if a.chr22:
    from linear_amd import synth
    ref = synth.chr22_like()
    gen = torch.from_numpy(ref).to(dev); offs = [0, ref.size]
t0 = time.time()
if not a.chr22:
    gen, offs = grch38_like_cuda(dev, scale=a.scale); torch.cuda.synchronize()
print(f"genome {gen.numel() / 1e6:.0f} Mb in {time.time() - t0:.1f}s, N fraction {(gen == 4).float().mean().item():.3f}", flush=True)
f = Filter(device=0, index_type=a.index_type)
t0 = time.time()
info = f.build_index_ptrs([gen.data_ptr() + o for o in offs[:-1]], [offs[i + 1] - offs[i] for i in range(len(offs) - 1)], a.T)
print(f"index {time.time() - t0:.2f}s wall, {info.build_ms:.0f} ms device: hs {info.hs_len} samples {info.n_samples} f2 {info.f2_len}", flush=True)
print("mem GB", torch.cuda.mem_get_info()[0] / 1e9, "free of", torch.cuda.mem_get_info()[1] / 1e9, flush=True)
batches = []
for b in range(a.batches):
    t0 = time.time(); r, o = sample_reads_multi_cuda(gen, [10_510_000, offs[1]] if a.chr22 else offs, a.reads, a.read_len, 0.10, 777 + b); torch.cuda.synchronize()
    batches.append((r, o)); print(f"batch {b} generated in {time.time() - t0:.1f}s", flush=True)
for rep in range(2):
    for b, (r, o) in enumerate(batches):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if a.seed_only: f.seed_lookup_batch_dev(r.data_ptr(), o.data_ptr(), a.reads)
        else: f.filter_batch_dev(r.data_ptr(), o.data_ptr(), a.reads)
        dt = time.perf_counter() - t0
        st = f.stats()
        print(f"rep {rep} batch {b}: {dt * 1e3:.1f} ms  {a.reads / dt:.0f} reads/s | per read: lookups {st['lookups'] / a.reads:.0f} entries {st['bucket_entries'] / a.reads:.0f} anchors {st['anchors'] / a.reads:.0f} "
              f"cords {st['cords'] / a.reads:.0f} remap {st['remap_reads']} | ms prep {st['prep_ms']:.2f} seed {st['seed_count_ms']:.2f} ({st['seed_count_launches']} launches, {st['seed_bytes'] / 1e9:.2f} GB) job {st['job_ms']:.1f} tail {st['tail_ms']:.2f} total {st['total_ms']:.1f} job_launches {st['job_launches']}", flush=True)
if a.check:
    from oracle import pyorc
    pyorc.build(ref=False)
    h = gen.cpu().numpy()
    t0 = time.time(); orc = pyorc.Checker("oracle", [h[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)], a.T, a.index_type); print(f"oracle index {time.time() - t0:.1f}s", flush=True)
    r, o = batches[0]
    hr = r[: a.check * a.read_len].cpu().numpy(); ho = o[: a.check + 1].cpu().numpy().astype(np.uint64)
    t0 = time.time(); ooff, ocs, oce, ost = orc.map_batch(hr, ho, threads=os.cpu_count()); tc = time.time() - t0
    coff, cs, ce = f.filter_batch(hr, ho)
    print(f"oracle {a.check / tc:.0f} reads/s on {os.cpu_count()} threads; parity {np.array_equal(coff, ooff) and np.array_equal(cs, ocs) and np.array_equal(ce, oce)}", flush=True)
    dir_, hs, _, _ = f.index_export()
    if a.index_type == 2: print("index parity (ysa)", np.array_equal(hs, orc.ysa()), flush=True)
    else: print("index parity", np.array_equal(dir_, orc.dir()), np.array_equal(hs, orc.hs()), flush=True)
