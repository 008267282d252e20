#!/usr/bin/env python3
"""Diagnostic: builds the library with -DLNR_PROF (in-kernel cycle stamps, see lnr_hd.h LNR_TICK) into
gpurun_out/liblinear_amd_prof.so and prints the share of k_job's lane-0 time spent per phase.  Not a timing build."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from linear_amd import build as lb, api, synth
from linear_amd.synth_torch import sample_reads_cuda
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
so = os.path.join(ROOT, "gpurun_out", "liblinear_amd_prof.so")
subprocess.check_call([lb.hipcc_path()] + lb.FLAGS + ["-DLNR_PROF", "-o", so, os.path.join(lb.CSRC, "lnr_api.hip")])
api.SO = so
f = api.Filter(device=0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ref = synth.chr22_like()
f.build_index([ref], 1)
d_ref = torch.from_numpy(ref).cuda()
d_reads, d_off = sample_reads_cuda(d_ref, n, 10000, 0.10, 777, non_n_start=10_510_000)
f.filter_batch_dev(d_reads.data_ptr(), d_off.data_ptr(), n)
out = (C.c_ulonglong * 32)()
f.lib.lnr_prof_read.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
assert f.lib.lnr_prof_read(f.h, out) == 0
names = ["carve+setup", "binning", "radix sort", "phase1: filter list + introsort + xs/ys", "chain DP (wave)", "traceback", "gather_blocks + prefilter_chains2",
         "chain_blocks + filter_blocks_hits", "filter_hits (window dist)", "path_dst_2 (extension)"]
tot = sum(out[:10])
print("stats", f.stats())
for i, nm in enumerate(names):
    print(f"{nm:45s} {out[i] / 1e6:12.1f} Mcycles  {100.0 * out[i] / max(tot, 1):5.1f} %   max single job-phase {out[16 + i] / 1e6:9.2f} Mcycles")
print("cycles per job (lane 0 sum):", tot / max(f.stats()['jobs'], 1))
print(f"k_dp_big: slowest block {out[12] / 1e6:.1f} Mcycles, its m = {out[13]}, its sum of windows = {out[14]} (avg window {out[14] / max(out[13], 1):.0f}), n after binning = {out[15]}; sum of windows over all heavy jobs = {out[11]}")
