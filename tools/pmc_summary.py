#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes of bench.py into profiles/<round>/pmc_seed_fused.json.

Usage (after the GPU runs below have been merged back into gpurun_out/):
    tools/pmc_summary.py gpurun_out/final/pmc_fetch gpurun_out/final/pmc_write gpurun_out/final/calib profiles/r01 \
        --reads 100000 --read-len 10000 --err 0.1 --layout-threads 1 --steps 2 --warmup 1

The passes are separate because FETCH_SIZE (3 TCC slots) and WRITE_SIZE (2) do not fit one pass
(/opt/skills/guides/MI355X_MICROARCH.md, PMC slot table):
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/final/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/final/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/final/calib     -- tools/calib_fetch

Corrections, as that guide's HBM section prescribes: both counters are in KiB; WRITE_SIZE is exact; FETCH_SIZE reads
half the bytes of a wide coalesced streaming load and is "uncalibrated" for other widths, so the factor for this
kernel's pattern (independent 4/8-byte random reads) is taken from tools/calib_fetch.hip run under the same counter.
"""
import argparse
import collections
import csv
import glob
import json
import os


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            out[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir"); ap.add_argument("write_dir"); ap.add_argument("calib_dir"); ap.add_argument("out_dir")
    ap.add_argument("--reads", type=int, default=100000); ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--err", type=float, default=0.10); ap.add_argument("--layout-threads", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2); ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seed-only", action="store_true"); ap.add_argument("--small", action="store_true")
    a = ap.parse_args()

    fe, wr, ca = per_kernel(a.fetch_dir, "FETCH_SIZE"), per_kernel(a.write_dir, "WRITE_SIZE"), per_kernel(a.calib_dir, "FETCH_SIZE")
    n_acc = 1 << 26
    calib = {}
    for k, width in (("k_rand8", 8), ("k_rand4", 4), ("k_stream16", 16)):
        v = ca[k][-1] * 1024.0                       # second launch of each (first warms nothing: the table is 4 GiB)
        calib[k] = {"counted_bytes_per_access": v / n_acc, "algorithmic_bytes_per_access": width}
    # The streaming kernel reproduces the guide's x2 (8 of 16 bytes counted).  The random kernels count 64 B per access, and
    # their COUNTED rate tops out where the streaming kernel's does (3.2-3.4 TB/s counted; the streaming one is known to move
    # twice that): an L2 miss fills a whole 128-byte line, tallied at 64 B like any other request.  So x2 for this kernel too.
    stream_factor = 16.0 / calib["k_stream16"]["counted_bytes_per_access"]
    fetch_factor = 2.0

    name = "lnr::k_seed_fused"
    steps = a.steps + a.warmup
    f_all, w_all = fe[name], wr[name]
    lps = len(f_all) / steps
    # the timed steps only (drop the warm-up's launches)
    f_t, w_t = f_all[int(a.warmup * lps):], w_all[int(a.warmup * lps):]
    fetch_b = sum(f_t) * 1024.0 / len(f_t)
    write_b = sum(w_t) * 1024.0 / len(w_t)
    rec = {
        "workload": {"reads": a.reads, "read_len": a.read_len, "err": a.err, "layout_threads": a.layout_threads,
                     "small": a.small, "seed_only": a.seed_only},
        "launches_per_step": lps,
        "fetch_bytes_per_launch_raw": fetch_b,
        "write_bytes_per_launch": write_b,
        "fetch_factor": fetch_factor,
        "traffic_bytes_per_launch": fetch_factor * fetch_b + write_b,
        "calibration": calib,
        "stream16_factor_measured": stream_factor,
        "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, KiB x 1024) of `bench.py --steps %d --warmup %d --no-cpu-baseline`, "
                  "mean over the timed launches of k_seed_fused; FETCH_SIZE x2 (gfx950 correction): tools/calib_fetch.hip counts %.1f B per independent 8-byte random read, "
                  "%.1f B per 4-byte one and %.2f of 16 B for the streaming read, all three saturating at the same counted rate (profiles/r01/calib_fetch_size.json), "
                  "i.e. every request is a 128-byte line tallied at 64 B"
                  % (a.steps, a.warmup, calib["k_rand8"]["counted_bytes_per_access"], calib["k_rand4"]["counted_bytes_per_access"], calib["k_stream16"]["counted_bytes_per_access"]),
        "per_kernel_fetch_KiB": {k: v for k, v in fe.items() if k.startswith("lnr::")},
        "per_kernel_write_KiB": {k: v for k, v in wr.items() if k.startswith("lnr::")},
    }
    os.makedirs(a.out_dir, exist_ok=True)
    json.dump(rec, open(os.path.join(a.out_dir, "pmc_seed_fused.json"), "w"), indent=1)
    print(json.dumps({k: rec[k] for k in ("launches_per_step", "fetch_bytes_per_launch_raw", "write_bytes_per_launch", "calibration", "stream16_factor_measured")}, indent=1))


if __name__ == "__main__":
    main()
