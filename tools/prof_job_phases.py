#!/usr/bin/env python3
"""Diagnostic: builds the library with -DLNR_PROF (in-kernel cycle stamps, see lnr_hd.h LNR_TICK) as the variant
tools/_variants/jobprof.so (linear_amd.build, the same helper tools/build_variant.sh uses) and prints the share of k_job's lane-0
time spent per phase.  Not a timing build."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from linear_amd import build as lb, api, synth
from linear_amd.synth_torch import sample_reads_cuda
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
os.makedirs(os.path.join(ROOT, "tools", "_variants"), exist_ok=True)
so = os.path.join(ROOT, "tools", "_variants", "jobprof.so")
if not os.path.exists(so) or os.environ.get("LNR_PROF_REBUILD"):
    lb.build(force=False, defines=["-DLNR_PROF"], out=so)
api.SO = so
f = api.Filter(device=0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
if len(sys.argv) > 2 and sys.argv[2] == "grch38":
    from linear_amd.synth_torch import grch38_like_cuda, sample_reads_multi_cuda
    gen, offs = grch38_like_cuda(torch.device("cuda", 0), seed=38, scale=float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)
    f.build_index_ptrs([gen.data_ptr() + o for o in offs[:-1]], [offs[i + 1] - offs[i] for i in range(24)], 16)
    d_reads, d_off = sample_reads_multi_cuda(gen, offs, n, 10000, 0.10, 777)
else:
    ref = synth.chr22_like()
    f.build_index([ref], 1)
    d_ref = torch.from_numpy(ref).cuda()
    d_reads, d_off = sample_reads_cuda(d_ref, n, 10000, 0.10, 777, non_n_start=10_510_000)
f.filter_batch_dev(d_reads.data_ptr(), d_off.data_ptr(), n)
out = (C.c_ulonglong * 192)()
f.lib.lnr_prof_read.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
assert f.lib.lnr_prof_read(f.h, out) == 0
names = ["carve+setup", "binning", "radix sort", "phase1: filter list + introsort + xs/ys", "chain DP", "traceback", "gather_blocks + prefilter_chains2",
         "chain_blocks + filter_blocks_hits", "filter_hits (window dist)", "path_dst_2 (extension)"]
print("stats", f.stats())
for cls, cname in enumerate(["k_job round 0", "k_job round 1 (re-map)", "k_job_heavy round 0", "k_job_heavy round 1"]):
    o = out[32 * cls: 32 * cls + 32]
    tot = sum(o[:10]); jobs = max(o[10], 1)
    print(f"== {cname}: {o[10]} jobs, anchors in {o[13] / jobs:.0f}/job, anchors in DP {o[11] / jobs:.0f}/job, pairs {o[12] / jobs:.0f}/job (window {o[12] / max(o[11], 1):.1f}); lane-0 cycles/job {tot / jobs:.0f}")
    for i, nm in enumerate(names):
        print(f"   {nm:42s} {o[i] / 1e6:10.1f} Mcyc {100.0 * o[i] / max(tot, 1):5.1f} %  per job {o[i] / jobs:9.0f}  max {o[16 + i] / 1e6:8.2f} Mcyc")
    print(f"   block DP {o[28] / jobs:.0f} cyc/job: scores {o[26] / jobs:.0f}, reduction {o[29] / jobs:.0f}, record + sync {o[27] / jobs:.0f}")
    mp = out[128 + 16 * cls: 128 + 16 * cls + 16]
    print(f"   biggest job of the class: {mp[12]} anchors in, {mp[10]} in the DP, {mp[11]} pairs; Mcyc per phase: " + " ".join(f"{mp[i] / 1e6:.2f}" for i in range(10)) + f"  (sum {sum(mp[:10]) / 1e6:.1f}); list filter {mp[14] / 1e6:.2f}, introsort {mp[15] / 1e6:.2f} (phase 3 column = carve + x/y fill)")

# ---- per-workgroup timeline (100 MHz ticks): who runs when, and what the tail consists of
f.lib.lnr_prof_timeline.restype = C.c_longlong
f.lib.lnr_prof_timeline.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_ulonglong), C.c_ulonglong, C.POINTER(C.c_uint)]
for rnd in range(4):
    cap = 1 << 20
    buf = np.zeros(cap * 4, dtype=np.uint64)
    nh = C.c_uint(0)
    k = f.lib.lnr_prof_timeline(f.h, rnd, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), cap, C.byref(nh))
    if k <= 0:
        continue
    t = buf[: 4 * k].reshape(k, 4)
    np.save(os.path.join(ROOT, "gpurun_out", f"timeline_r{rnd}.npy"), t)
    ok = t[:, 1] > 0
    t0 = t[ok, 0].min()
    st = (t[:, 0].astype(np.int64) - int(t0)) / 100.0      # us
    en = (t[:, 1].astype(np.int64) - int(t0)) / 100.0
    m = (t[:, 3] >> np.uint64(32)).astype(np.int64); pairs = (t[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
    H = nh.value
    print(f"== round {rnd}: {k} workgroups, heavy prefix {H}; span {en[ok].max() / 1000:.1f} ms")
    for name, sl in (("heavy", slice(0, H)), ("normal", slice(H, k))):
        s_, e_, m_, p_ = st[sl], en[sl], m[sl], pairs[sl]
        if len(s_) == 0:
            continue
        d = e_ - s_
        print(f"  {name}: first start {s_.min() / 1000:.2f} ms, last start {s_.max() / 1000:.2f} ms, last end {e_.max() / 1000:.2f} ms; sum of durations {d.sum() / 1000:.0f} ms; "
              f"mean {d.mean():.0f} us, p50 {np.percentile(d, 50):.0f}, p99 {np.percentile(d, 99):.0f}, max {d.max():.0f} us")
        top = np.argsort(-e_)[:8]
        for i in top:
            print(f"     pos {i + (H if name == 'normal' else 0):6d}: start {s_[i] / 1000:7.2f} ms  dur {d[i] / 1000:7.2f} ms  end {e_[i] / 1000:7.2f} ms  anchors-in-DP {m_[i]:6d} pairs {p_[i]:10d}")
        grid = np.linspace(0, e_.max(), 21)[1:]
        res = [(int(((s_ <= g) & (e_ > g)).sum())) for g in grid]
        print(f"     resident workgroups at 5% steps of its span: {res}")
