#!/usr/bin/env python3
"""bench.py -- reads/sec of the `linear filter` hot path on MI355X (BASELINE.json metric).

Workload (N = 1): BASELINE.json configs[1] -- 100 000 synthetic 10 kb ONT-error-profile reads (10 % errors,
40/30/30 sub/del/ins, 50 % reverse-complemented) against chr22, `-f 2 -i 1`, `-g 0` (apxMap only; the gap
re-mapper is next tier).  No genome file or network exists on the box, so chr22 is the seeded stand-in
`synth.chr22_like` (same length, leading N arm, human-like repeat spectrum); `config.workload` says so.

One step = one pass of the whole hot path (read prep + features + seed lookup + filter/chain/extend + block
chaining -> cords) over one batch of reads that is already resident in HBM.  Index build (and, for N > 1, its RCCL
broadcast) is done once before the timed region and reported in `config`.

N > 1: one process per GPU (torch.distributed.run); rank 0 builds the packed index and broadcasts it over
RCCL/xGMI; every rank then filters its own batch (weak scaling: per-GPU batch fixed, no data-path collective).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def recorded_traffic(args, launches_per_step):
    """HBM-side bytes per k_seed_fused launch from the committed rocprofv3 PMC passes (profiles/r01/pmc_seed_fused.json,
    made by tools/pmc_summary.py from separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this same command).  Counters cannot
    be read from inside the process, so the figure is only reported when the workload is the one the passes were taken on."""
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01", "pmc_seed_fused.json")
    if not os.path.exists(p):
        return None
    rec = json.load(open(p))
    key = {"reads": args.reads, "read_len": args.read_len, "err": args.err, "layout_threads": args.layout_threads,
           "small": bool(args.small), "seed_only": bool(args.seed_only)}
    if rec.get("workload") != key or abs(rec.get("launches_per_step", 0) - launches_per_step) > 1e-9:
        return None
    return {"bytes_per_launch": rec["traffic_bytes_per_launch"], "source": rec["source"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=100_000, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=10_000)
    ap.add_argument("--err", type=float, default=0.10)
    ap.add_argument("--layout-threads", type=int, default=1, help="reference -t whose index layout is reproduced")
    ap.add_argument("--cpu-sample", type=int, default=16_000, help="reads of the same workload timed on the host cores with the oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed-only", action="store_true", help="time only stage a7 (seed lookup) -- used for the roofline profile")
    ap.add_argument("--small", action="store_true", help="tiny reference/batch (plumbing check)")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: everything else (RCCL prints a version banner to stdout at init) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or bool(os.environ.get("LNR_BENCH_DIST_AT_1"))   # the env switch rehearses the N>1 plumbing on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if world == 1:
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from linear_amd import build as lb
    if local_rank == 0:
        lb.build()
    if use_dist:
        dist.barrier()
    from linear_amd import Filter, synth
    from linear_amd import dist as ldist
    from linear_amd.synth_torch import sample_reads_cuda

    # ---- reference + index (outside the timed region)
    t0 = time.time()
    if args.small:
        ref = synth.repeat_ref(2_000_000, 99)
        non_n = 0
        ref_name = "synthetic 2 Mb repeat-rich reference (--small)"
    else:
        ref = synth.chr22_like()
        non_n = 10_510_000
        ref_name = "chr22 stand-in synth.chr22_like(seed 2022): 50 818 468 bp, 10.5 Mb leading N, human-like repeat spectrum"
    t_ref = time.time() - t0
    flt = Filter(device=local_rank)
    index_s, bcast = 0.0, None
    if rank == 0:
        t0 = time.time()
        info = flt.build_index([ref], args.layout_threads)
        index_s = time.time() - t0
        log(f"[bench] reference generated in {t_ref:.1f}s; index built in {index_s:.2f}s wall ({info.build_ms:.1f} ms device): hs {info.hs_len}, samples {info.n_samples}, f2 {info.f2_len}")
    if use_dist:
        bcast = ldist.broadcast_index(flt, 0, dev)
        if rank == 0:
            log(f"[bench] index broadcast: {bcast['bytes'] / 1e9:.2f} GB in {bcast['seconds'] * 1e3:.1f} ms ({bcast['bytes'] / 1e9 / max(bcast['seconds'], 1e-9):.1f} GB/s)")

    # ---- reads of this rank, generated in HBM
    d_ref = torch.from_numpy(ref).to(dev)
    t0 = time.time()
    d_reads, d_off = sample_reads_cuda(d_ref, args.reads, args.read_len, args.err, 777 + rank, non_n_start=non_n)
    torch.cuda.synchronize()
    log(f"[bench] rank {rank}: {args.reads} reads x {args.read_len} bp generated on device in {time.time() - t0:.1f}s")
    del d_ref

    def step():
        if args.seed_only:
            flt.seed_lookup_batch_dev(d_reads.data_ptr(), d_off.data_ptr(), args.reads)
        else:
            flt.filter_batch_dev(d_reads.data_ptr(), d_off.data_ptr(), args.reads)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    acc = {}
    for _ in range(args.steps):
        step()
        st = flt.stats()
        for k, v in st.items():
            acc[k] = acc.get(k, 0) + v
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    st = flt.stats()
    K = max(args.steps, 1)
    # k_seed_fused: one launch per round (round 0 = whole reads, round 1 = the re-mapped gaps); timed by the library with
    # HIP events recorded on the stream the kernel is launched on (lnr_api.hip run_jobs, Timer t_sc)
    launches = max(int(acc["seed_count_launches"]), 1)
    seed_ms = acc["seed_count_ms"] / launches
    seed_bytes = acc["seed_bytes"] / launches
    achieved = seed_bytes / (seed_ms * 1e-3) / 1e9 if seed_ms > 0 else 0.0
    traffic = recorded_traffic(args, launches / K)

    out = {
        "metric": "reads/sec (whole node) + HBM GB/s on seed lookup, 10 kb reads vs GRCh38",
        "value": args.reads * world * args.steps / dt,
        "unit": "reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": f"{args.reads} synthetic {args.read_len} bp ONT-profile reads per GPU per step ({args.err:.0%} errors 40/30/30 sub/del/ins, 50% revcomp) vs {ref_name}; "
                        f"linear filter -f 2 -i 1 -g 0 -p 1, index layout -t {args.layout_threads}" + ("; SEED LOOKUP STAGE ONLY" if args.seed_only else ""),
            "reads_per_gpu_per_step": args.reads,
            "read_len": args.read_len,
            "parallelism": f"read-sharded x{world}, index built on rank 0" + (" + RCCL broadcast" if world > 1 else ""),
            "index_build_s": round(index_s, 3),
            "index_broadcast_s": round(bcast["seconds"], 4) if bcast else None,
            "index_bytes": bcast["bytes"] if bcast else None,
            "per_read": {"samples": st["samples"] / args.reads, "lookups": st["lookups"] / args.reads, "bucket_entries": st["bucket_entries"] / args.reads,
                         "anchors": st["anchors"] / args.reads, "cords": st["cords"] / args.reads, "remap_reads": st["remap_reads"]},
            "stage_ms_per_step": {"prep": acc["prep_ms"] / K, "seed": acc["seed_count_ms"] / K,
                                  "job": acc["job_ms"] / K, "tail": acc["tail_ms"] / K, "total_device": acc["total_ms"] / K},
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "k_seed_fused (minimizers of the 2-bit packed read -> bucket bitmap -> dir -> hs -> Y filter -> anchors), "
                      f"{launches / K:g} launches per step, averages per launch",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": seed_bytes,
            "launch_ms": seed_ms,
            "traffic": traffic["bytes_per_launch"] if traffic else None,
            "traffic_source": traffic["source"] if traffic else None,
        },
    }

    # ---- CPU baseline + parity of the sample (rank 0, N = 1 only)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.seed_only:
        from oracle import pyorc
        pyorc.build(ref=False)
        ns = min(args.cpu_sample, args.reads)
        cores = min(os.cpu_count() or 1, 16)
        h_reads = d_reads[: ns * args.read_len].cpu().numpy()
        h_off = d_off[: ns + 1].cpu().numpy().astype(np.uint64)
        t0 = time.time()
        orc = pyorc.Checker("oracle", [ref], args.layout_threads)
        t_oidx = time.time() - t0
        orc.map_batch(h_reads[: 64 * args.read_len], h_off[:65], threads=cores)   # warm the thread pool
        t0 = time.time()
        ooff, ocs, oce, ost = orc.map_batch(h_reads, h_off, threads=cores)
        t_cpu = time.time() - t0
        coff, cs, ce = flt.filter_batch(h_reads, h_off)
        same = bool(np.array_equal(coff, ooff) and np.array_equal(cs, ocs) and np.array_equal(ce, oce))
        out["cpu_baseline"] = {"value": ns / t_cpu, "unit": "reads/s", "cores": cores, "kind": "port",
                               "sample": f"first {ns} reads of the same batch, oracle/lnr_oracle.cpp (bit-exact restatement of the reference) with {cores} OpenMP threads; "
                                         f"{t_cpu:.2f} s wall; its index build ({t_oidx:.1f} s, 1 thread) not included",
                               "pair_evals_per_read": float(ost[4]) / ns}
        out["parity"] = {"checked_reads": ns, "bit_exact_vs_oracle": same}
        log(f"[bench] cpu baseline {ns / t_cpu:.0f} reads/s on {cores} cores; GPU/oracle parity on the sample: {same}")
        if not same:
            log("[bench] PARITY FAILURE on the bench sample")
        # PCIe-inclusive rate of the host-buffer entry point, for DESIGN.md (never `value`)
        t0 = time.time()
        flt.filter_batch(h_reads, h_off)
        log(f"[bench] host-buffer entry point (PCIe in/out) on the {ns}-read sample: {ns / (time.time() - t0):.0f} reads/s")

    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    flt.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
