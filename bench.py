#!/usr/bin/env python3
"""bench.py -- reads/sec of the `linear filter` hot path on MI355X (BASELINE.json metric: "reads/sec (whole node) + HBM GB/s
on seed lookup, 10 kb reads vs GRCh38").

Workload (N = 1, the default): BASELINE.json configs[2] -- synthetic 10 kb ONT-error-profile reads (10 % errors, 40/30/30
sub/del/ins, 50 % reverse-complemented) against the full GRCh38 primary assembly, `-f 2 -i 1`, `-g 0`.  No genome file or
network exists on the box, so GRCh38 is the seeded stand-in `synth_torch.grch38_like_cuda` (24 sequences with the human
chromosome lengths, 3.09 Gb, N runs, repeat families, tandem repeats, segmental duplications) generated in HBM;
`config.workload` says so.  Every step filters a DIFFERENT batch of 100 000 reads (steps + warmup distinct batches, at most 16
resident; 1 M reads = 10 steps).  `--workload chr22` is configs[1] (the round-1 line), `--workload small` a plumbing check.

One step = one pass of the whole hot path (read prep + features + seed lookup + filter / chain / extend + block chaining ->
cords) over one batch.  Two timed regions run over the same batches: the batch already resident in HBM
(`config.device_resident_reads_per_s`), and -- the line's `value`, SURVEY 8(d)'s metric -- host read blocks in -> host cords out
through `lnr_filter_submit` / `lnr_filter_wait` with three batches in flight (reads up, cords down inside the clock).  The
default line also measures the reference's DEFAULT mode on the same batches and context (`config.gap50`: `-g 1` = gaps of 50 and
more re-mapped, with its own reference baseline and parity check; `--no-gap50` skips it) and the front-end binary end to end on
FASTA files (`config.cli_end_to_end_reads_per_s`; `--no-cli`).  `--gap 50 [--dup 1]` makes the gap path the line itself;
`--workload ccs_sv` is BASELINE configs[4] on one GPU (15 kb CCS-profile reads, SVs planted on the device, `-g 50 -dup 1`).
Index build (and, for N > 1, its RCCL broadcast) happens once before the timed region and is reported in `config`.

N > 1: one process per GPU.  Launched either by the driver through torch.distributed.run, or by `python bench.py --gpus N`
itself: with no WORLD_SIZE in the environment this script starts the N ranks as a child torch.distributed.run BEFORE any GPU
call and relays rank 0's JSON line.  Rank 0 builds the packed index and broadcasts it over RCCL/xGMI; every rank then filters
its own batches (weak scaling: per-GPU batch fixed, no data-path collective).

CPU baseline (rank 0, N = 1): the REAL reference (oracle/_ref/libref_linear.so, the reference's own translation units compiled
by oracle/Makefile; `kind: "reference"`) running its calculator loop on all host cores the process may use; where that library
is absent the bit-exact restatement oracle/lnr_oracle.cpp (`kind: "port"`).  Test infrastructure, used here as the checker and
as the timed CPU leg only.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_INT_PEAK_TOPS = 78.6    # same guide: 157.3 TFLOP/s fp32 vector = 2 x 78.6 T lane-operations/s (256 CU x 4 SIMD x 32 lanes x 2.4 GHz)
DP_OPS_PER_PAIR = 31         # VALU instructions of one chaining-DP predecessor step without a candidate


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def host_cores() -> tuple[int, str]:
    """Cores this process may use: the affinity mask, cut by a cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = f"affinity mask {n} logical CPUs"
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            lim = max(1, int(int(q) / int(p)))
            if lim < n:
                n, note = lim, f"cgroup quota {lim} CPUs"
    except Exception:
        pass
    return n, note


def spawn_ranks(args) -> int:
    """`bench.py --gpus N` without a launcher: start the N ranks (fresh processes, nothing here has touched the GPU) and relay
    the one JSON line rank 0 prints."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE)
    line = None
    for raw in p.stdout:
        t = raw.decode(errors="replace").strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        elif t:
            log(t)
    rc = p.wait()
    if line:
        print(line, flush=True)
    return rc if rc else (0 if line else 1)


def recorded_traffic(workload_key: dict, launches_per_step: float):
    """HBM-side bytes per seed-lookup launch from the committed rocprofv3 PMC passes (profiles/rNN/pmc_seed.json of the latest round that has
    one, made by tools/pmc_seed_r02.py from separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this same command).  Counters cannot be
    read from inside the process, so the figure is only reported when the workload is the one the passes were taken on."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", "pmc_seed.json")))
    if not found:
        return None
    rec = json.load(open(found[-1]))
    if rec.get("workload") != workload_key or abs(rec.get("launches_per_step", 0) - launches_per_step) > 1e-9:
        print(f"[bench] roofline.traffic: the counter passes in {os.path.relpath(found[-1], ROOT)} were taken on another workload ({rec.get('workload')}): not reported", file=sys.stderr, flush=True)
        return None
    return {"bytes_per_launch": rec["traffic_bytes_per_launch"], "source": rec["source"] + " [" + os.path.relpath(found[-1], ROOT) + "]"}


def padded_layout(seq_len):
    """Start offsets of the sequences inside the library's genome blob (lnr_api.hip set_index_layout: each sequence is
    followed by at least 64 zero bytes and starts 64-byte aligned)."""
    starts, o = [], 0
    for L in seq_len:
        starts.append(o)
        o += (int(L) + 64 + 63) // 64 * 64
    return starts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["grch38", "chr22", "small", "ccs_sv"], default="grch38",
                    help="grch38 = configs[2] (default); chr22 = configs[1]; ccs_sv = configs[4] at one GPU: 15 kb CCS-profile reads (0.5 %% errors) with a planted SV in 5 %% of them vs the GRCh38 stand-in, -g 50 -dup 1")
    ap.add_argument("--scale", type=float, default=1.0, help="grch38 only: scale of the chromosome lengths")
    ap.add_argument("--reads", type=int, default=100_000, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=10_000)
    ap.add_argument("--err", type=float, default=0.10)
    ap.add_argument("--layout-threads", type=int, default=0, help="reference -t whose index layout is reproduced (0 = the host cores the CPU baseline runs on)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the timed CPU sample")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = every core this process may use)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cli", action="store_true", help="skip the front-end's end-to-end measurement (FASTA files in -> SAM out)")
    ap.add_argument("--no-gap50", action="store_true", help="skip the -g 50 leg (counter passes: only the headline path's kernels run)")
    ap.add_argument("--seed-only", action="store_true", help="time only stage a7 (seed lookup) -- used for the roofline profile")
    ap.add_argument("--gap", type=int, default=0, help="the reference's -g: 0 = apxMap only (the headline configuration); > 0 = cords go through the gap re-mapper (SURVEY 8 f1)")
    ap.add_argument("--dup", type=int, default=0, help="the reference's -dup (with --gap)")
    ap.add_argument("--small", action="store_true", help="alias of --workload small")
    args = ap.parse_args()
    if args.small:
        args.workload = "small"
    explicit = {a.split("=")[0] for a in sys.argv[1:] if a.startswith("--")}
    if args.workload == "ccs_sv":                      # BASELINE configs[4]: the flags and the read profile it names (unless given on the command line)
        if "--read-len" not in explicit: args.read_len = 15_000
        if "--err" not in explicit: args.err = 0.005
        if "--gap" not in explicit: args.gap = 50
        if "--dup" not in explicit: args.dup = 1
        if "--reads" not in explicit: args.reads = 66_000
    if args.workload == "small":
        args.reads = min(args.reads, 2000)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))

    # stdout carries exactly one JSON line: everything else (RCCL's banner, the reference's progress lines) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    double = bool(os.environ.get("LNR_BENCH_DOUBLE"))   # CPU rehearsal of the N > 1 plumbing (tests/test_bench_cli_cpu.py): gloo + a test double

    import numpy as np
    import torch
    import torch.distributed as dist
    if double:
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs an MI355X"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)

    def sync():
        if not double:
            torch.cuda.synchronize()

    use_dist = world > 1 or bool(os.environ.get("LNR_BENCH_DIST_AT_1"))   # the env switch rehearses the N>1 plumbing on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if world == 1:
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if double:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    cores, cores_note = host_cores()
    if args.cpu_threads:
        cores = args.cpu_threads
    T = args.layout_threads or cores

    from linear_amd import dist as ldist
    if double:
        from tests.bench_double import FilterDouble as Filter, make_genome, sample_reads
    else:
        from linear_amd import build as lb
        if local_rank == 0:
            lb.build()
        if use_dist:
            dist.barrier()
        from linear_amd import Filter
        from linear_amd.synth_torch import grch38_like_cuda, sample_reads_multi_cuda

    # ---- reference + index (outside the timed region); rank 0 owns the build
    flt = Filter(device=local_rank, gap_len=args.gap, dup=args.dup)
    index_s, bcast, t_ref, ref_name, info = 0.0, None, 0.0, "", None
    host_genome = None     # [numpy per sequence] for the CPU baseline (rank 0, N = 1)
    if args.workload in ("grch38", "ccs_sv"):
        ref_name = (f"GRCh38 stand-in synth_torch.grch38_like_cuda(seed 38, scale {args.scale:g}): 24 sequences with the human chromosome lengths, "
                    "N runs (telomeres, centromeres, acrocentric arms), 1500 repeat families over ~45 %, tandem repeats, segmental duplications")
    elif args.workload == "chr22":
        ref_name = "chr22 stand-in synth.chr22_like(seed 2022): 50 818 468 bp, 10.5 Mb leading N, human-like repeat spectrum"
    else:
        ref_name = "synthetic 2 Mb repeat-rich reference (--workload small)"
    if rank == 0:
        t0 = time.time()
        if double:
            seqs = make_genome()
            t_ref = time.time() - t0
            t0 = time.time()
            info = flt.build_index(seqs, T)
        elif args.workload in ("grch38", "ccs_sv"):
            gen, offs = grch38_like_cuda(dev, seed=38, scale=args.scale)
            sync()
            t_ref = time.time() - t0
            t0 = time.time()
            info = flt.build_index_ptrs([gen.data_ptr() + o for o in offs[:-1]], [offs[i + 1] - offs[i] for i in range(len(offs) - 1)], T)
            if world == 1 and not (args.no_cpu_baseline and args.no_cli) and not args.seed_only:
                h = gen.cpu().numpy()
                host_genome = [h[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
            del gen
        else:
            from linear_amd import synth
            seqs = [synth.chr22_like()] if args.workload == "chr22" else [synth.repeat_ref(2_000_000, 99)]
            t_ref = time.time() - t0
            t0 = time.time()
            info = flt.build_index(seqs, T)
            host_genome = seqs
        index_s = time.time() - t0
        log(f"[bench] reference generated in {t_ref:.1f}s; index built in {index_s:.2f}s wall ({info.build_ms:.1f} ms device): "
            f"{info.nseq} sequences, hs {info.hs_len}, samples {info.n_samples}, f2 {info.f2_len}, layout -t {T}")
    if use_dist:
        bcast = ldist.broadcast_index(flt, 0, "cpu" if double else dev)
        if rank == 0:
            log(f"[bench] index broadcast to {dist.get_world_size()} ranks: {bcast['bytes'] / 1e9:.2f} GB in {bcast['seconds'] * 1e3:.1f} ms "
                f"({bcast['bytes'] / 1e9 / max(bcast['seconds'], 1e-9):.1f} GB/s)")
    info = flt.index_info()
    index_bytes = int(sum(b for _, b in flt.index_blobs()))

    # ---- this rank's read batches, generated in HBM from the library's own copy of the genome (every rank has it after the broadcast)
    nb = max(1, min(args.steps + args.warmup, 16))
    seq_len = [int(v) for v in flt.seq_len()]
    t0 = time.time()
    sv_note = ""
    if double:
        batches = [sample_reads(args.reads, 1000 * rank + b) for b in range(nb)]
    else:
        starts = padded_layout(seq_len)
        gp, gb = flt.index_blobs()[0]
        gview = ldist.blob_tensor(gp, gb, dev)
        non_n = [starts[0] + 10_510_000] + starts[1:] if args.workload == "chr22" else starts
        ends = [s + L for s, L in zip(starts, seq_len)]
        if args.workload == "ccs_sv":
            from linear_amd.synth_torch import plant_svs_cuda
            batches, n_sv = [], 0
            for b in range(nb):
                src_len = args.read_len + 5200
                r0, _ = sample_reads_multi_cuda(gview, non_n, args.reads, src_len, args.err, 777 + 1000 * rank + b, ends=ends)
                r1, o1, k = plant_svs_cuda(r0, args.reads, src_len, args.read_len, 0.05, 4242 + 1000 * rank + b)
                batches.append((r1, o1)); n_sv += k
                del r0
            sv_note = f", one planted SV (deletion / insertion / tandem duplication / inversion, 50 bp - 5 kb) in {n_sv / (nb * args.reads):.1%} of the reads"
        else:
            batches = [sample_reads_multi_cuda(gview, non_n, args.reads, args.read_len, args.err, 777 + 1000 * rank + b, ends=ends) for b in range(nb)]
            sv_note = ""
        sync()
    log(f"[bench] rank {rank}: {nb} distinct batches of {args.reads} reads x {args.read_len} bp generated on device in {time.time() - t0:.1f}s")

    def step(k):
        r, o = batches[k % nb]
        if args.seed_only:
            flt.seed_lookup_batch_dev(r.data_ptr(), o.data_ptr(), args.reads)
        else:
            flt.filter_batch_dev(r.data_ptr(), o.data_ptr(), args.reads)

    def timed(fn_step, pre=None, post=None):
        """W untimed + exactly K timed steps between barrier + synchronize on both sides; the MAX over the ranks"""
        for k in range(args.warmup):
            fn_step(k)
        if pre:
            pre()
        sync()
        if use_dist:
            dist.barrier()
        t0 = time.perf_counter()
        acc_ = {}
        for k in range(args.steps):
            fn_step(args.warmup + k)
            for key, v in flt.stats().items():
                acc_[key] = acc_.get(key, 0) + v
        sync()
        if use_dist:
            dist.barrier()
        dt_ = time.perf_counter() - t0
        if post:
            post()
        if use_dist:
            t = torch.tensor([dt_], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_, acc_

    # (1) the batch resident in HBM (config.device_resident_reads_per_s; the roofline's seed launches are timed here)
    dt_dev, acc = timed(step)
    # (2) THE METRIC (first read batch submitted -> last cords batch returned): host read blocks in, host cords out, through
    # lnr_filter_submit / lnr_filter_wait with three batches in flight (step k: submit(k + 2); wait(k) -- the wait downloads batch k's cords
    # while it computes batch k + 1, the upload of k + 2 runs under those kernels), in steady state: the pipeline is already full when the clock
    # starts, as it is for every later step.  Inside the clock: K uploads, K computes, K result downloads.  The two batches still in flight
    # after the last timed step are drained behind the clock.
    host_path = not args.seed_only
    dt = dt_dev
    if host_path:
        nhb = max(3, min(4 if world > 1 else 8, nb))
        hb = []
        for k in range(nhb):
            buf = flt.host_alloc(int(batches[k % nb][0].numel()))
            buf[:] = batches[k % nb][0].cpu().numpy()
            hb.append((buf, batches[k % nb][1].cpu().numpy().astype(np.uint64)))
        flt.filter_submit(*hb[0])
        flt.filter_submit(*hb[1 % nhb])

        def host_step(k):
            flt.filter_submit(*hb[(k + 2) % nhb])
            flt.filter_wait(copy=False)
        dt, acc_h = timed(host_step, post=lambda: (flt.filter_wait(copy=False), flt.filter_wait(copy=False)))
    if args.gap:
        flt.gap_stream(1)                 # (whatever the drained batch left: the timed batches above all ran in the extended state)

    K = max(args.steps, 1)
    total_reads = args.reads * K
    # seed lookup: one launch per round (round 0 = whole reads, round 1 = the re-mapped gaps); timed by the library with HIP
    # events recorded on the stream the kernels are launched on (lnr_api.hip seed_jobs, Timer t_seed)
    launches = max(int(acc["seed_count_launches"]), 1)
    seed_ms = acc["seed_count_ms"] / launches
    seed_bytes = acc["seed_bytes"] / launches
    achieved = seed_bytes / (seed_ms * 1e-3) / 1e9 if seed_ms > 0 else 0.0
    wl_key = {"workload": args.workload, "scale": args.scale, "reads": args.reads, "read_len": args.read_len, "err": args.err,
              "layout_threads": T, "seed_only": bool(args.seed_only)}
    if args.gap:
        wl_key["gap"] = args.gap
    dup_flag = " -dup 1" if args.dup else ""
    traffic = recorded_traffic(wl_key, launches / K)
    dev_rate = args.reads * world * args.steps / dt_dev
    rate = args.reads * world * args.steps / dt

    out = {
        "metric": "reads/sec (whole node) + HBM GB/s on seed lookup, 10 kb reads vs GRCh38",
        "value": rate,
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
            "workload": f"{args.reads} synthetic {args.read_len} bp {'PacBio-CCS' if args.err < 0.02 else 'ONT'}-profile reads per GPU per step, a different batch every step ({args.err:.1%} errors 40/30/30 sub/del/ins, "
                        f"50% revcomp{sv_note}) vs {ref_name}; linear filter -f 2 -i 1 -g {args.gap}{dup_flag} -p 1, index layout -t {T}" + ("; SEED LOOKUP STAGE ONLY" if args.seed_only else ""),
            "baseline_config": {"grch38": "configs[2] (10 kb ONT reads vs full GRCh38; 1 M reads = 10 steps of 100 k)", "chr22": "configs[1]", "small": "plumbing",
                                "ccs_sv": "configs[4] at ONE GPU (15 kb CCS-profile reads with planted SVs vs GRCh38, -g 50 -dup 1; 66 k reads per step)"}[args.workload],
            "value_is": ("host read blocks in -> host cords out (lnr_filter_submit / lnr_filter_wait, pinned blocks, 3 batches in flight, steady state; reads H2D + cords D2H inside the clock)"
                         if host_path else "seed stage only, batch resident in HBM"),
            "reads_per_gpu_per_step": args.reads,
            "distinct_batches": nb,
            "read_len": args.read_len,
            "parallelism": f"read-sharded x{world}, index built on rank 0" + (" + RCCL broadcast" if world > 1 else ""),
            "ranks_reported_by_backend": dist.get_world_size() if use_dist else 1,
            "reference_generation_s": round(t_ref, 2),
            "index_build_s": round(index_s, 3),
            "index_build_device_ms": round(float(info.build_ms), 1),
            "index_broadcast_s": round(bcast["seconds"], 4) if bcast else None,
            "index_bytes": index_bytes,
            "index": {"nseq": int(info.nseq), "genome_bytes": int(info.genome_bytes), "hs_len": int(info.hs_len), "dir_len": int(info.dir_len),
                      "f2_len": int(info.f2_len), "samples": int(info.n_samples), "layout_threads": int(T)},
            "per_read": {"samples": acc["samples"] / total_reads, "lookups": acc["lookups"] / total_reads, "bucket_entries": acc["bucket_entries"] / total_reads,
                         "anchors": acc["anchors"] / total_reads, "cords": acc["cords"] / total_reads, "remap_reads_per_step": acc["remap_reads"] / K},
            "stage_ms_per_step": {"prep": acc["prep_ms"] / K, "seed": acc["seed_count_ms"] / K,
                                  "job": acc["job_ms"] / K, "tail": acc["tail_ms"] / K, "gap": acc.get("gap_ms", 0.0) / K, "total_device": acc["total_ms"] / K},
            "device_resident_reads_per_s": dev_rate,
            "gap_second_pass_per_step": acc.get("gap_second_pass", 0) / K,
            "gap_path_note": "this line is -g " + str(args.gap) + "; the reference's default mode (-g 1 = gaps of 50 and more re-mapped) is measured in this same run: config.gap50_*",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "seed lookup stage a3/a4/a7 (minimizers of the 2-bit packed read -> bucket lookup -> Y filter -> anchors), "
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

    # ---- the reference's DEFAULT mode in the same line: -g 1 (gaps of 50 and more re-mapped, mapper.cpp:207-231), same batches, same context
    gap50 = None
    if args.gap == 0 and not args.no_gap50 and not args.seed_only and not double and world == 1 and rank == 0 and args.workload in ("grch38", "chr22"):
        flt.set_gap(1, 0)
        step(0)                                   # (also takes the stream through its first extension: the timed steps run in the steady state)
        sync()
        gk = min(3, args.steps)
        t0 = time.perf_counter()
        gacc = {}
        for k in range(gk):
            step(1 + k)
            for key, v in flt.stats().items():
                gacc[key] = gacc.get(key, 0) + v
        sync()
        gdt = time.perf_counter() - t0
        gap50 = {"reads_per_s": args.reads * gk / gdt, "ms_per_step": gdt / gk * 1e3, "steps": gk, "gap_stage_ms_per_step": gacc.get("gap_ms", 0.0) / gk,
                 "second_pass_reads_per_step": gacc.get("gap_second_pass", 0) / gk, "mode": "batch resident in HBM, linear filter -g 1 (= 50)"}
        out["config"]["gap50_reads_per_s"] = gap50["reads_per_s"]
        out["config"]["gap50"] = gap50
        log(f"[bench] -g 50 (the reference's default mode): {gap50['reads_per_s']:.0f} reads/s, {gap50['ms_per_step']:.1f} ms per step (gap stage {gap50['gap_stage_ms_per_step']:.1f} ms)")
        flt.set_gap(0, 0)

    parity_ok = True
    out["config"]["host_path_reads_per_s"] = rate if host_path else None
    if rank == 0:
        log(f"[bench] device-resident {dev_rate:.0f} reads/s; host entry point (pinned reads in, cords out, PCIe both ways, three batches in flight) {rate:.0f} reads/s = {rate / dev_rate:.0%}")

    if rank == 0 and world == 1 and not double and not args.no_cpu_baseline and not args.seed_only and host_genome is not None:
        # ---- CPU baseline + parity on a bounded sample of batch 0
        from oracle import pyorc
        pyorc.build(ref=False)
        kind = "ref" if pyorc.have_ref() else "oracle"
        t0 = time.time()
        chk = pyorc.Checker(kind, host_genome, T)
        t_cidx = time.time() - t0
        h_reads = batches[0][0].cpu().numpy()
        h_off = batches[0][1].cpu().numpy().astype(np.uint64)

        def cpu_run(n0, n1):
            rr = h_reads[int(h_off[n0]):int(h_off[n1])]
            oo = h_off[n0:n1 + 1] - h_off[n0]
            t0 = time.time()
            res = chk.map_batch(rr, oo, threads=cores, gap_len=args.gap, dup=args.dup)
            return time.time() - t0, res, rr, oo
        pilot = min(args.reads, max(4 * cores, 256))
        tp, _, _, _ = cpu_run(0, pilot)
        ns = int(min(args.reads, max(pilot, pilot * args.cpu_seconds / max(tp, 1e-3))))
        t_cpu, (ooff, ocs, oce, ost), rr, oo = cpu_run(0, ns)
        if args.gap:
            flt.gap_stream(0)        # the sample is a read stream of its own, as the CPU run took it (lnr_gap_stream)
        coff, cs, ce = flt.filter_batch(rr, oo)
        parity_ok = bool(np.array_equal(coff, ooff) and np.array_equal(cs, ocs) and np.array_equal(ce, oce))
        label = ("the reference itself: oracle/_ref/libref_linear.so = the reference's own translation units (base, cords, shape_extend, index_util, "
                 "cluster_util, pmpfinder, gap, gap_util) compiled by oracle/Makefile, calculator loop of Mapper::p_calRecords with -g " + str(args.gap) + (" -dup 1" if args.dup else "")) if kind == "ref" else \
                "oracle/lnr_oracle.cpp (bit-exact restatement of the reference)"
        out["cpu_baseline"] = {"value": ns / t_cpu, "unit": "reads/s", "cores": cores, "kind": "reference" if kind == "ref" else "port",
                               "sample": f"first {ns} reads of batch 0, {label}, {cores} OpenMP threads ({cores_note}); {t_cpu:.2f} s wall; "
                                         f"its index + genome features build ({t_cidx:.1f} s, -t {T}) not included"}
        out["parity"] = {"checked_reads": ns, "bit_exact": parity_ok, "against": out["cpu_baseline"]["kind"]}
        log(f"[bench] cpu baseline ({kind}) {ns / t_cpu:.0f} reads/s on {cores} threads; GPU parity on the sample: {parity_ok}")
        if gap50 is not None:
            def cpu_run_g(n1):
                rr2 = h_reads[: int(h_off[n1])]
                oo2 = h_off[: n1 + 1] - h_off[0]
                t0 = time.time()
                res = chk.map_batch(rr2, oo2, threads=cores, gap_len=1, dup=0)
                return time.time() - t0, res, rr2, oo2
            tg, _, _, _ = cpu_run_g(pilot)
            ng = int(min(args.reads, max(pilot, pilot * 0.6 * args.cpu_seconds / max(tg, 1e-3))))
            t_g, (goff, gcs, gce, _), rr2, oo2 = cpu_run_g(ng)
            flt.set_gap(1, 0)                      # a new stream, as the CPU run took the sample
            coff2, cs2, ce2 = flt.filter_batch(rr2, oo2)
            flt.set_gap(0, 0)
            gpar = bool(np.array_equal(coff2, goff) and np.array_equal(cs2, gcs) and np.array_equal(ce2, gce))
            out["config"]["gap50"]["cpu_baseline"] = {"value": ng / t_g, "unit": "reads/s", "cores": cores, "kind": "reference" if kind == "ref" else "port",
                                                      "sample": f"first {ng} reads of batch 0 with -g 1, {cores} OpenMP threads, {t_g:.2f} s wall"}
            out["config"]["gap50"]["parity"] = {"checked_reads": ng, "bit_exact": gpar}
            out["config"]["gap50"]["x_cpu"] = gap50["reads_per_s"] / (ng / t_g)
            log(f"[bench] -g 50 cpu baseline ({kind}) {ng / t_g:.0f} reads/s on {cores} threads -> {gap50['reads_per_s'] / (ng / t_g):.1f} x; GPU parity on the sample: {gpar}")
            parity_ok = parity_ok and gpar
        # secondary roofline: chaining-DP predecessor pairs per second next to the VALU integer peak.  Pair counts
        # come from the restatement's counter on a slice of the sample (the reference has no counter).
        if kind == "ref":
            orc = pyorc.Checker("oracle", host_genome, T)
            n2 = min(ns, 2000)
            _, _, _, ost = orc.map_batch(rr[: int(oo[n2])], oo[: n2 + 1], threads=cores)
            pairs_per_read = float(ost[4]) / n2
            orc.close()
        else:
            pairs_per_read = float(ost[4]) / ns
        job_s = acc["job_ms"] / K * 1e-3
        pair_rate = pairs_per_read * args.reads / job_s if job_s > 0 else 0.0
        peak_pairs = VALU_INT_PEAK_TOPS * 1e12 / DP_OPS_PER_PAIR
        out["roofline_secondary"] = {"bound": "valu-int", "kernel": "per-read job kernels (a8-a16; the chaining DP is ~40 % of their VALU work)",
                                     "achieved": pair_rate / 1e9, "peak": peak_pairs / 1e9, "unit": "G predecessor pairs/s", "frac": pair_rate / peak_pairs,
                                     "pairs_per_read": pairs_per_read,
                                     "note": f"peak = {VALU_INT_PEAK_TOPS} T lane-ops/s / {DP_OPS_PER_PAIR} VALU instructions per pair; achieved = pairs of one step / job-kernel time of one step"}
        if not parity_ok:
            log("[bench] PARITY FAILURE on the bench sample")
        chk.close()

    if rank == 0 and world == 1 and not double and not args.seed_only and not args.no_cli and host_genome is not None and args.workload in ("grch38", "chr22", "small"):
        # ---- the front-end binary end to end: FASTA files in -> .sam out through reader -> GPU -> writer, batch 0 as a read file
        import re
        import shutil
        import tempfile
        td = tempfile.mkdtemp(prefix="lnr_cli_")
        try:
            abc = np.frombuffer(b"ACGTN", np.uint8)
            t0 = time.time()
            with open(os.path.join(td, "ref.fa"), "wb") as f:
                for k, sq in enumerate(host_genome):
                    f.write(b">chr%d\n" % (k + 1))
                    f.write(abc[np.minimum(sq, 4)].tobytes())
                    f.write(b"\n")
            hr = batches[0][0].cpu().numpy().reshape(args.reads, args.read_len)
            txt = np.empty((args.reads, args.read_len + 1), np.uint8)
            txt[:, :-1] = abc[hr]
            txt[:, -1] = 10
            with open(os.path.join(td, "reads.fa"), "wb") as f:
                rows = txt.tobytes()
                step_b = args.read_len + 1
                for i in range(args.reads):
                    f.write(b">read_%d\n" % i)
                    f.write(rows[i * step_b:(i + 1) * step_b])
            t_write = time.time() - t0
            from linear_amd import build as lb2
            cmd = [lb2.CLI, "filter", os.path.join(td, "reads.fa"), os.path.join(td, "ref.fa"), "-t", str(T), "-g", str(args.gap), "-o", os.path.join(td, "out"), "--block-reads", "20000"] + (["-dup", "1"] if args.dup else [])
            t0 = time.time()
            pc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
            t_cli = time.time() - t0
            m = re.search(rb"output files out: ([0-9.]+) s = ([0-9.]+) reads/s", pc.stderr)
            if pc.returncode == 0 and m:
                out["config"]["cli_end_to_end_reads_per_s"] = float(m.group(2))
                out["config"]["cli_end_to_end"] = {"reads": args.reads, "read_phase_s": float(m.group(1)), "whole_run_s": round(t_cli, 2), "sam_bytes": os.path.getsize(os.path.join(td, "out.sam")),
                                                   "what": "linear_amd/linear_filter filter reads.fa ref.fa (plain FASTA, %d reads x %d bp; parallel mapped reader -> lnr_filter_submit / _wait -> writer on -t host threads -> .sam); "
                                                           "read_phase = first read block fetched .. last text written; whole_run adds genome load + index" % (args.reads, args.read_len)}
                mb = re.search(rb"Stage busy time\[s\]: ([^\n]*)", pc.stderr)
                out["config"]["cli_end_to_end"]["stage_busy_s"] = mb.group(1).decode() if mb else None
                log(f"[bench] front-end end to end: {float(m.group(2)):.0f} reads/s in the read phase ({float(m.group(1)):.2f} s), whole run {t_cli:.1f} s (files written in {t_write:.1f} s); busy: {mb.group(1).decode() if mb else '?'}")
            else:
                log("[bench] front-end run failed: " + pc.stderr.decode(errors="replace")[-500:])
        finally:
            shutil.rmtree(td, ignore_errors=True)

    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    flt.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if not parity_ok:
        sys.exit(1)


if __name__ == "__main__":
    main()
