#!/usr/bin/env python3
"""Randomised parity stress on the GPU box: many small configurations (reference type, index layout, read length, error
rate, N runs, several sequences), HIP path through the C ABI against the CPU oracle, bit-exact.  Not part of the test
suites (it takes minutes); run as  python tools/stress_parity.py [n_configs] [seed].
LNR_STRESS_ONLY=3,17 re-runs just those configurations of the sequence; LNR_STRESS_LIBS=a.so,b.so checks each of them with
several builds of the library (bisecting a mismatch)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from linear_amd import Filter, synth  # noqa: E402
from linear_amd import build as lb  # noqa: E402
from oracle import pyorc  # noqa: E402


OPTION_SETS = [{}, {}, {"LNR_MID_CAP": "64"}, {"LNR_MID_CAP": "64"}, {"LNR_HEAVY_CAP": "64"}, {"LNR_HEAVY_CAP": "64"}, {"LNR_DP_SPLIT_CAP": "64"}, {"LNR_SPLIT_CAP": "200"},
               {"LNR_SPLIT_CAP": "200", "LNR_LANE_ORDER": "heavy"}, {"LNR_MID_CAP": "300", "LNR_HEAVY_CAP": "900"}, {"LNR_JOB_LDS_KB": "2"},
               {"LNR_SEED_BM": "0"}, {"LNR_SEED_BM": "0", "LNR_MID_CAP": "64"}, {"LNR_POST_SPLIT": "1"}, {"LNR_POST_SPLIT": "1", "LNR_SEED_BM": "0"}, {"LNR_JOB_LDS_KB": "1"}, {"LNR_MID_CAP": "64", "LNR_MID_WAVES": "2"}, {"LNR_MID_CAP": "300", "LNR_MID_WAVES": "2", "LNR_SEED_BM": "0"}]
ALL_KEYS = sorted({k for o in OPTION_SETS for k in o})


def main():
    ncfg = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    lb.build()
    pyorc.build(ref=False)
    rng = np.random.default_rng(seed0)
    bad = 0
    only = {int(x) for x in os.environ.get("LNR_STRESS_ONLY", "").split(",") if x}
    libs = [x for x in os.environ.get("LNR_STRESS_LIBS", "").split(",") if x]
    itype = int(os.environ.get("LNR_STRESS_INDEX_TYPE", "1"))   # the reference's -i: 1 DIndex, 2 HIndex
    for k in range(ncfg):
        s = int(rng.integers(1, 1 << 30))
        kind = int(rng.integers(0, 3))
        T = int(rng.choice([1, 2, 3, 4, 8]))
        L = int(rng.choice([201, 260, 700, 3000, 9000, 20000, 40000, 120000]))
        err = float(rng.choice([0.0, 0.03, 0.1, 0.15]))
        nreads = int(rng.choice([1, 7, 64, 300, 1500]))
        if kind == 0:
            refs = [synth.random_ref(int(rng.integers(200_000, 900_000)), s)]
        elif kind == 1:
            refs = [synth.repeat_ref(int(rng.integers(300_000, 1_500_000)), s, n_families=int(rng.integers(3, 20)))]
        else:
            refs = [synth.repeat_ref(400_000, s), synth.add_n_runs(synth.random_ref(250_000, s + 1), s + 2, lead=int(rng.integers(0, 5000))),
                    synth.repeat_ref(150_000, s + 3, n_families=4)]
        jitter = float(rng.choice([0.0, 0.5]))
        # library options (read at lnr_create): exercise the other size classes and orchestration modes now and then
        opt = OPTION_SETS[int(rng.integers(0, len(OPTION_SETS)))]
        if only and k not in only:
            continue
        reads, off, _ = synth.sample_reads(refs, nreads, L, err, s + 7, "random", len_jitter=jitter)
        for kv in ALL_KEYS:
            os.environ.pop(kv, None)
        os.environ.update(opt)
        t0 = time.time()
        o = pyorc.Checker("oracle", refs, T, itype)
        ooff, ocs, oce, _ = o.map_batch(reads, off, threads=8)
        t1 = time.time()
        for lib in libs:
            from linear_amd import api
            api.SO = os.path.abspath(lib)
            f = Filter(device=0, index_type=itype)
            f.build_index(refs, T)
            coff, cs, ce = f.filter_batch(reads, off)
            f.close()
            same = bool(np.array_equal(coff, ooff) and np.array_equal(cs, ocs) and np.array_equal(ce, oce))
            nd = -1
            if not same and np.array_equal(coff, ooff):
                d = np.nonzero((cs != ocs) | (ce != oce))[0]
                rd = np.unique(np.searchsorted(ooff, d, side="right") - 1)
                nd = len(rd)
            print(f"[stress] cfg {k} lib {lib}: {'ok' if same else 'MISMATCH'} (reads differing: {nd})", flush=True)
        f = Filter(device=0, index_type=itype)
        f.build_index(refs, T)
        coff, cs, ce = f.filter_batch(reads, off)
        f.close()
        same = bool(np.array_equal(coff, ooff) and np.array_equal(cs, ocs) and np.array_equal(ce, oce))
        bad += 0 if same else 1
        print(f"[stress] cfg {k}: kind {kind} T {T} L {L} err {err} reads {nreads} opts {opt} cords {cs.size}: {'ok' if same else 'MISMATCH'} (oracle {t1 - t0:.1f}s)", flush=True)
    nrun = len(only) if only else ncfg
    print(f"[stress] {nrun - bad}/{nrun} configurations bit-exact (index type {itype})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
