#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes of bench.py (GRCh38 stand-in) into profiles/r02/pmc_seed.json -- the `traffic` figure of
bench.py's roofline object.

    tools/pmc_seed_r02.py <fetch_dir> <write_dir> <rdreq_dir> profiles/r02 --steps 3 --warmup 1 --layout-threads 16

The passes are separate rocprofv3 runs of the same command (`python3 bench.py --steps S --warmup W --no-cpu-baseline`), one
counter set each (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2: /opt/skills/guides/MI355X_MICROARCH.md, PMC slots).
Corrections as that guide's HBM section prescribes: both counters are in KiB; WRITE_SIZE is exact; FETCH_SIZE tallies every
128-byte request at 64 B on gfx950, so it is doubled -- cross-checked here against TCC_EA0_RDREQ_sum x 128 B from the third pass."""
import argparse
import collections
import csv
import glob
import json
import os


def per_kernel(d, counter):
    f = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True))[-1]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            out[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir"); ap.add_argument("write_dir"); ap.add_argument("rdreq_dir"); ap.add_argument("out_dir")
    ap.add_argument("--workload", default="grch38"); ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--reads", type=int, default=100000); ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--err", type=float, default=0.10); ap.add_argument("--layout-threads", type=int, default=16)
    ap.add_argument("--steps", type=int, default=3); ap.add_argument("--warmup", type=int, default=1)
    a = ap.parse_args()
    name = "lnr::k_seed_fused"
    fe, wr, rq = per_kernel(a.fetch_dir, "FETCH_SIZE")[name], per_kernel(a.write_dir, "WRITE_SIZE")[name], per_kernel(a.rdreq_dir, "TCC_EA0_RDREQ_sum")[name]
    # the timed steps' launches: the last 2 * steps (two per step: round 0 and the re-map round); earlier ones are warm-up and the
    # first batch's repeat after the anchor buffer grew
    k = 2 * a.steps
    f_t, w_t, r_t = fe[-k:], wr[-k:], rq[-k:]
    fetch_b = sum(f_t) * 1024.0 / len(f_t)
    write_b = sum(w_t) * 1024.0 / len(w_t)
    rdreq_b = sum(r_t) * 128.0 / len(r_t)
    rec = {
        "workload": {"workload": a.workload, "scale": a.scale, "reads": a.reads, "read_len": a.read_len, "err": a.err, "layout_threads": a.layout_threads, "seed_only": False},
        "launches_per_step": 2.0,
        "fetch_bytes_per_launch_raw": fetch_b,
        "write_bytes_per_launch": write_b,
        "fetch_factor": 2.0,
        "rdreq_x_128B_per_launch": rdreq_b,
        "traffic_bytes_per_launch": 2.0 * fetch_b + write_b,
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_EA0_RDREQ_sum (three separate passes, KiB x 1024) of `python3 bench.py --steps %d --warmup %d "
                  "--no-cpu-baseline --no-cli --no-gap50`, mean over the timed launches of lnr::k_seed_fused (two per step); FETCH_SIZE x 2 (gfx950: 128-byte requests tallied at 64 B, "
                  "MI355X_MICROARCH.md HBM section; the request counter x 128 B of the third pass gives %.3f of that), WRITE_SIZE exact" % (a.steps, a.warmup, rdreq_b / (2.0 * fetch_b)),
        "launches_seen": {"fetch": len(fe), "write": len(wr), "rdreq": len(rq)},
        "per_launch_KiB_fetch": f_t, "per_launch_KiB_write": w_t,
    }
    os.makedirs(a.out_dir, exist_ok=True)
    json.dump(rec, open(os.path.join(a.out_dir, "pmc_seed.json"), "w"), indent=1)
    print(json.dumps({k2: rec[k2] for k2 in ("fetch_bytes_per_launch_raw", "write_bytes_per_launch", "rdreq_x_128B_per_launch", "traffic_bytes_per_launch")}, indent=1))


if __name__ == "__main__":
    main()
