"""Seeded parity cases shared by tools/make_golden.py (which runs the REAL reference
through oracle/_ref and stores its outputs under tests/golden/) and by the tests
(which regenerate the same inputs from the seeds and compare).

A case = (reference sequences, reads, index layout T).  ``T`` is the reference's
``-t``: the DIndex content depends on it.
"""
from __future__ import annotations

import hashlib

import numpy as np

from linear_amd import synth


def _reads_list(reads, off):
    return [reads[int(off[i]):int(off[i + 1])] for i in range(off.size - 1)]


def case_c1():
    """BASELINE config 1 (scaled): error-free 5 kb reads vs 1 Mb random reference, odd reads reverse-complemented."""
    ref = synth.random_ref(1_000_000, 12345)
    reads, off, _ = synth.sample_reads([ref], 250, 5000, 0.0, 777, "odd")
    return [ref], reads, off


def case_ont():
    """ONT-profile 10 kb reads (10 % errors, 40/30/30 sub/del/ins) vs 2 Mb random reference."""
    ref = synth.random_ref(2_000_000, 4242)
    reads, off, _ = synth.sample_reads([ref], 120, 10000, 0.10, 778, "random")
    return [ref], reads, off


def case_rep():
    """Repeat-rich reference: exercises bucket omission, tie-sensitive sorts, both traceback algorithms."""
    rep = synth.repeat_ref(1_000_000, 99)
    reads, off, _ = synth.sample_reads([rep], 80, 10000, 0.10, 779, "random")
    return [rep], reads, off


def case_edge():
    """Three reference sequences with N runs + edge-case reads (short, junk -> remap loop, N, chimeras, sequence ends)."""
    r0 = synth.add_n_runs(synth.random_ref(600_000, 5), 6, n_runs=4, max_run=3000, lead=5000, trail=3000)
    r1 = synth.add_n_runs(synth.repeat_ref(400_000, 7), 8, n_runs=2, max_run=500)
    r2 = synth.random_ref(30_000, 9)
    refs = [r0, r1, r2]
    er = synth.edge_reads(refs, 11)
    reads2, off2, _ = synth.sample_reads(refs, 60, 8000, 0.08, 780, "random", len_jitter=0.5)
    reads, off = synth.pack_reads(er + _reads_list(reads2, off2))
    return refs, reads, off


def case_ccs_sv():
    """BASELINE configs[4]'s workload in small: PacBio-CCS-profile reads (15 kb, 0.5 % errors) with planted structural variants
    (insertion, deletion, tandem duplication, inversion, foreign insert; 50 bp - 5 kb) in two reads out of three, on a repeat-rich
    reference and one with N runs -- what `-g 50 -dup 1` exists for (gap-path goldens only: CASES_G50)."""
    refs = [synth.repeat_ref(1_500_000, 515), synth.add_n_runs(synth.random_ref(700_000, 516), 517, n_runs=3, max_run=800)]
    rng = np.random.default_rng(518)
    cpl = np.array([3, 2, 1, 0, 4], np.uint8)
    out = []
    for k in range(72):
        ref = refs[k % 2]
        x0 = int(rng.integers(1000, ref.size - 30000))
        seg = ref[x0:x0 + 16000].copy()
        cut = int(rng.integers(3000, 11000)); m = int(rng.choice([50, 120, 400, 1200, 3000, 5000]))
        kind = k % 6
        if kind == 1: seg = np.concatenate([seg[:cut], seg[cut + m:]])
        elif kind == 2: seg = np.concatenate([seg[:cut], rng.integers(0, 4, m, dtype=np.uint8), seg[cut:]])
        elif kind == 3: seg = np.concatenate([seg[:cut], seg[max(cut - m, 0):cut], seg[cut:]])
        elif kind == 4: seg = np.concatenate([seg[:cut], cpl[seg[cut:cut + m][::-1]], seg[cut + m:]])
        elif kind == 5: seg = np.concatenate([seg[:cut], ref[x0 + 20000:x0 + 20000 + m], seg[cut:]])
        r, o_, _ = synth.sample_reads([seg], 1, min(seg.size - 50, 15000), 0.005, 600 + k, "random")
        out.append(np.ascontiguousarray(r[: int(o_[1])]))
    reads, off = synth.pack_reads(out)
    return refs, reads, off


def case_chim():
    """Chimeric reads: 3 - 5 segments of 2.5 - 4 kb from unrelated places / strands / sequences glued together, 3 % errors -> 3 - 5 SAM
    lines per read, each with an SA:Z listing the others (the reference's NM cache of createSAZTagCigarOneChimeric, flag 2048 chains).
    CLI goldens only (CASES_CLI)."""
    refs = [synth.random_ref(900_000, 8101), synth.repeat_ref(600_000, 8102), synth.random_ref(120_000, 8103)]
    rng = np.random.default_rng(8104)
    cpl = np.array([3, 2, 1, 0, 4], np.uint8)
    out = []
    for k in range(36):
        segs = []
        for _ in range(int(rng.integers(3, 6)) if k % 6 else 1):
            ref = refs[int(rng.integers(0, 3))]
            m = int(rng.integers(2500, 4000))
            x0 = int(rng.integers(100, ref.size - m - 100))
            sg = ref[x0:x0 + m]
            segs.append(cpl[sg[::-1]] if rng.random() < 0.5 else sg)
        seg = np.concatenate(segs)
        r, o_, _ = synth.sample_reads([seg], 1, seg.size - 40, 0.03, 8200 + k, "random")
        out.append(np.ascontiguousarray(r[: int(o_[1])]))
    reads, off = synth.pack_reads(out)
    return refs, reads, off


def case_scale():
    """Scale pin: 24 sequences, the first 262.5 Mb (longer than chr1: x + 2^20 beyond 2^28, 8 800 binning
    bins, 37 M index entries), ids up to 23, reads from both ends of the big sequence, chimeras big-end + sequence 23."""
    refs = synth.scale_refs()
    reads, off = synth.scale_reads(refs)
    return refs, reads, off


def case_hbig():
    """HIndex (-i 2) with blocks of 1024 entries and more: a 480 kb tandem repeat (period 40, 1 % substitutions) between random
    flanks; every sampled X of the repeat collects thousands of entries, so lookups meet the virtual-head / (Y, X) nodes and the
    `ptr >= 64` rule reads the word in front of a body."""
    rng = np.random.default_rng(5)
    unit = rng.integers(0, 4, 40, dtype=np.uint8)
    tand = np.tile(unit, 12000)
    mut = rng.random(tand.size) < 0.01
    tand[mut] = rng.integers(0, 4, int(mut.sum()), dtype=np.uint8)
    ref = np.concatenate([synth.random_ref(300_000, 31), tand, synth.random_ref(200_000, 32)])
    reads, off, _ = synth.sample_reads([ref], 30, 6000, 0.05, 91, "random")
    return [ref], reads, off


# HIndex (-i 2) goldens: name -> (builder, [T layouts]); files <name>_i2_T<T>.npz
CASES_I2 = {
    "c1": (case_c1, [1]),
    "edge": (case_edge, [1, 3]),
    "hbig": (case_hbig, [1, 4]),
}

# gap-path goldens (-g 50 and -g 50 -dup 1, made by the reference): name -> (builder, T); files <name>_g50_T<T>.npz
CASES_G50 = {"ont": (case_ont, 1), "edge": (case_edge, 1), "ccs_sv": (case_ccs_sv, 1), "rep": (case_rep, 1), "chim": (case_chim, 1)}

# goldens made by the REAL `linear filter` program (oracle/_ref/linear, tools/make_cli_golden.py): name -> builder; files cli_<name>.npz hold
# the program's .sam and .apf bytes at -t 1 for -g 0, -g 50 and -g 50 -dup 1
CASES_CLI = {"edge": case_edge, "ccs_sv": case_ccs_sv, "rep": case_rep, "chim": case_chim}
CLI_MODES = {"g0": ["-g", "0"], "g50": ["-g", "50"], "g50dup1": ["-g", "50", "-dup", "1"], "gdef": []}

CASES = {
    # name: (builder, [T layouts])
    "c1": (case_c1, [1]),
    "ont": (case_ont, [1, 4]),
    "rep": (case_rep, [1, 8]),
    "edge": (case_edge, [1, 3]),
    "scale": (case_scale, [4]),
}

# reads whose per-stage outputs are stored (default: the first N_STAGE_READS mappable reads)
STAGE_IDX = {"scale": [0, 45, 85, 100, 119, 125, 131, 160, 163, 166, 167]}

N_STAGE_READS = 6   # reads per case whose per-stage outputs are stored


def input_digest(refs, reads, off) -> str:
    h = hashlib.sha256()
    for r in refs:
        h.update(np.ascontiguousarray(r).tobytes())
    h.update(np.ascontiguousarray(reads).tobytes())
    h.update(np.ascontiguousarray(off).tobytes())
    return h.hexdigest()


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def text_ids(n_reads: int, n_refs: int):
    """Read / reference names used for the SAM and APF goldens (read ids keep blanks: the reference prints the whole header line)."""
    return [f"read_{i} len extra={i * 3}" for i in range(n_reads)], [f"chr{k + 1}" for k in range(n_refs)]


CMD_LINE = "linear filter reads.fa ref.fa -g 0"


def write_fasta_case(dirpath, refs, reads, off, width=80):
    """The case as the files a user would hand to `linear filter`: ref.fa (ids chr1.. + a description the reference cuts off) and reads.fa
    (ids with blanks, kept whole).  Returns (reads path, genome path, read ids, genome ids)."""
    import os
    n = off.size - 1
    rid, gid = text_ids(n, len(refs))
    abc = np.frombuffer(b"ACGTN", np.uint8)
    gp, rp = os.path.join(str(dirpath), "ref.fa"), os.path.join(str(dirpath), "reads.fa")
    with open(gp, "wb") as f:
        for k, r in enumerate(refs):
            f.write(b">" + gid[k].encode() + b" some description\n")
            t = abc[r].tobytes()
            f.write(b"\n".join(t[i:i + width] for i in range(0, len(t), width)) + b"\n")
    with open(rp, "wb") as f:
        for i in range(n):
            f.write(b">" + rid[i].encode() + b"\n" + abc[reads[int(off[i]):int(off[i + 1])]].tobytes() + b"\n")
    return rp, gp, rid, gid
