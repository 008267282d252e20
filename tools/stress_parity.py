#!/usr/bin/env python3
"""Randomised parity stress on the GPU box: many small configurations (reference type, index layout, read length, error
rate, N runs, several sequences), HIP path through the C ABI against the CPU oracle, bit-exact.  Not part of the test
suites (it takes minutes); run as  python tools/stress_parity.py [n_configs] [seed].
LNR_STRESS_ONLY=3,17 re-runs just those configurations of the sequence; LNR_STRESS_LIBS=a.so,b.so checks each of them with
several builds of the library (bisecting a mismatch).  LNR_STRESS_GAP=1: the gap re-mapper is on (a random -g of 1, 5, 50 or 200, -dup at
random) and a third of the reads carry a planted insertion, deletion, duplication, inversion or foreign insert."""
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


def plant_svs(reads, off, refs, rng):
    """every third read gets one structural change in its middle: deletion, random insertion, tandem duplication, inversion or a
    stretch of another place of the reference (the cases the gap re-mapper exists for)"""
    cpl = np.array([3, 2, 1, 0, 4], np.uint8)
    out = []
    for i in range(off.size - 1):
        r = reads[int(off[i]):int(off[i + 1])]
        if i % 3 == 0 and r.size > 1200:
            cut = int(rng.integers(400, r.size - 600)); m = int(rng.integers(60, min(1500, r.size - cut - 100)))
            kind = int(rng.integers(0, 5))
            if kind == 0: r = np.concatenate([r[:cut], r[cut + m:]])
            elif kind == 1: r = np.concatenate([r[:cut], rng.integers(0, 4, m, dtype=np.uint8), r[cut:]])
            elif kind == 2: r = np.concatenate([r[:cut], r[max(cut - m, 0):cut], r[cut:]])
            elif kind == 3: r = np.concatenate([r[:cut], cpl[r[cut:cut + m][::-1]], r[cut + m:]])
            else:
                g = refs[int(rng.integers(0, len(refs)))]; x = int(rng.integers(0, max(g.size - m, 1)))
                r = np.concatenate([r[:cut], g[x:x + m], r[cut:]])
        out.append(np.ascontiguousarray(r))
    o2 = np.zeros(len(out) + 1, np.uint64)
    o2[1:] = np.cumsum([x.size for x in out])
    return (np.concatenate(out) if out else np.zeros(0, np.uint8)), o2


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
    with_gap = os.environ.get("LNR_STRESS_GAP", "0") == "1"
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
        gap_len, dup = 0, 0
        if with_gap:                                  # (drawn before a configuration is skipped: LNR_STRESS_ONLY re-runs exactly the configuration of the full run)
            gap_len, dup = int(rng.choice([1, 5, 50, 200])), int(rng.integers(0, 2))
        if only and k not in only:
            continue
        reads, off, _ = synth.sample_reads(refs, nreads, L, err, s + 7, "random", len_jitter=jitter)
        if with_gap:
            reads, off = plant_svs(reads, off, refs, np.random.default_rng(s + 11))
        for kv in ALL_KEYS:
            os.environ.pop(kv, None)
        os.environ.update(opt)
        t0 = time.time()
        o = pyorc.Checker("oracle", refs, T, itype)
        ooff, ocs, oce, _ = o.map_batch(reads, off, threads=8, gap_len=gap_len, dup=dup)
        t1 = time.time()
        for lib in libs:
            from linear_amd import api
            lib, *envs = lib.split("@")                   # lib.so@KEY=VAL@KEY=VAL: library knobs for this one run
            api.SO = os.path.abspath(lib)
            saved = {e.split("=")[0]: os.environ.get(e.split("=")[0]) for e in envs}
            os.environ.update(dict(e.split("=", 1) for e in envs))
            f = Filter(device=0, index_type=itype, gap_len=gap_len, dup=dup)
            f.build_index(refs, T)
            coff, cs, ce = f.filter_batch(reads, off)
            lib = lib + " " + " ".join(envs) + f" (second-pass reads {f.stats()['gap_second_pass']})"
            f.close()
            for kk, vv in saved.items():
                os.environ.pop(kk, None) if vv is None else os.environ.__setitem__(kk, vv)
            same = bool(np.array_equal(coff, ooff) and np.array_equal(cs, ocs) and np.array_equal(ce, oce))
            nd = -1
            if not same and np.array_equal(coff, ooff):
                d = np.nonzero((cs != ocs) | (ce != oce))[0]
                rd = np.unique(np.searchsorted(ooff, d, side="right") - 1)
                nd = len(rd)
            print(f"[stress] cfg {k} lib {lib}: {'ok' if same else 'MISMATCH'} (reads differing: {nd})", flush=True)
        f = Filter(device=0, index_type=itype, gap_len=gap_len, dup=dup)
        f.build_index(refs, T)
        coff, cs, ce = f.filter_batch(reads, off)
        second = f.stats()["gap_second_pass"]
        f.close()
        same = bool(np.array_equal(coff, ooff) and np.array_equal(cs, ocs) and np.array_equal(ce, oce))
        bad += 0 if same else 1
        if not same:
            nr = off.size - 1
            diff = [i for i in range(nr) if not (int(coff[i + 1] - coff[i]) == int(ooff[i + 1] - ooff[i]) and np.array_equal(cs[int(coff[i]):int(coff[i + 1])], ocs[int(ooff[i]):int(ooff[i + 1])])
                                                 and np.array_equal(ce[int(coff[i]):int(coff[i + 1])], oce[int(ooff[i]):int(ooff[i + 1])]))]
            print(f"[stress] cfg {k}: reads that differ: {diff[:20]} ({len(diff)} of {nr})", flush=True)
            if os.environ.get("LNR_STRESS_SAVE"):
                np.savez_compressed(os.environ["LNR_STRESS_SAVE"], reads=reads, off=off, T=T, gap_len=gap_len, dup=dup, itype=itype, nrefs=len(refs), **{f"ref{q}": r for q, r in enumerate(refs)})
        print(f"[stress] cfg {k}: kind {kind} T {T} L {L} err {err} reads {nreads} opts {opt} -g {gap_len} -dup {dup} (second-pass reads {second}) cords {cs.size}: {'ok' if same else 'MISMATCH'} (oracle {t1 - t0:.1f}s)", flush=True)
    nrun = len(only) if only else ncfg
    print(f"[stress] {nrun - bad}/{nrun} configurations bit-exact (index type {itype})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
