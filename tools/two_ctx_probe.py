#!/usr/bin/env python3
"""GPU probe: do two contexts on ONE GPU, driven by two host threads, fill each other's idle time (re-map round, host gaps)?
Aggregate reads/s of N steps split over the contexts against one context doing all N."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from linear_amd import Filter
from linear_amd.synth_torch import grch38_like_cuda, sample_reads_multi_cuda
dev = torch.device("cuda", 0)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2
gen, offs = grch38_like_cuda(dev, scale=scale); torch.cuda.synchronize()
fl = []
for k in range(nctx):
    f = Filter(device=0)
    f.build_index_ptrs([gen.data_ptr() + o for o in offs[:-1]], [offs[i + 1] - offs[i] for i in range(24)], 16)
    fl.append(f)
n = 100_000
batches = [sample_reads_multi_cuda(gen, offs, n, 10_000, 0.10, 777 + b) for b in range(8)]
torch.cuda.synchronize()
def run(f, idx, reps):
    for _ in range(reps):
        for b in idx:
            r, o = batches[b]
            f.filter_batch_dev(r.data_ptr(), o.data_ptr(), n)
for f in fl: run(f, range(8), 1)          # warm-up: every context sees every batch
torch.cuda.synchronize(); t0 = time.perf_counter(); run(fl[0], range(8), 2); dt1 = time.perf_counter() - t0
print(f"one context : {16 * n / dt1:.0f} reads/s ({dt1 / 16 * 1e3:.1f} ms per step)", flush=True)
th = [threading.Thread(target=run, args=(fl[k], range(k, 8, nctx), 2)) for k in range(nctx)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
dt2 = time.perf_counter() - t0
print(f"{nctx} contexts  : {16 * n / dt2:.0f} reads/s ({dt2 / 16 * 1e3:.1f} ms per step)", flush=True)
