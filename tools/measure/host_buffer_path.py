# Times lnr_filter_batch (host buffers in and out, PCIe included) on the bench batch -- the number DESIGN.md quotes beside `value`.
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from linear_amd import api, synth
from linear_amd.synth_torch import sample_reads_cuda
f = api.Filter(device=0)
ref = synth.chr22_like(); f.build_index([ref], 1)
d_ref = torch.from_numpy(ref).cuda()
n = 100000
d_reads, d_off = sample_reads_cuda(d_ref, n, 10000, 0.10, 777, non_n_start=10_510_000)
reads = d_reads.cpu().numpy(); off = d_off.cpu().numpy().astype(np.uint64)
for it in range(4):
    t0 = time.time(); coff, cs, ce = f.filter_batch(reads, off); t1 = time.time()
    print("host-buffer path: %.1f ms, %.3f M reads/s, cords %d" % ((t1 - t0) * 1e3, n / (t1 - t0) / 1e6, cs.size), flush=True)
