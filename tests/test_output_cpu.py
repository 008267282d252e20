"""CPU: output shaping (linear_amd/csrc/lnr_output.cpp behind lnr_writer_*) against the reference's own writer
functions -- cords2BamLink + fillBamRecords + printAlignSamBam and print_cords_apf, run through oracle/_ref by
tools/make_golden.py and stored in tests/golden/*.npz (`sam`, `apf`).  Byte-identical text for the same cords: SAM header and
records (flags, CIGAR with '=' / 'X' / 'I' / 'D' / 'S', SA:Z, MAPQ 255) and APF (blank lines per the reference's block rule, the
block being the call)."""
import os

import numpy as np
import pytest

from tests import cases

GOLD = os.path.join(os.path.dirname(__file__), "golden")
PARAMS = [(n, T) for n, (_, Ts) in cases.CASES.items() for T in Ts]


def first_diff(a: bytes, b: bytes) -> str:
    la, lb = a.split(b"\n"), b.split(b"\n")
    for i, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            return f"line {i}: want {x[:200]!r} got {y[:200]!r}"
    return f"{len(la)} vs {len(lb)} lines"


@pytest.mark.parametrize("name,T", PARAMS)
def test_writer_equals_reference_text(case_inputs, name, T):
    from linear_amd import build as lb
    lb.build()
    from linear_amd.api import Writer
    refs, reads, off = case_inputs(name)
    g = np.load(os.path.join(GOLD, f"{name}_T{T}.npz"))
    n = off.size - 1
    rid, gid = cases.text_ids(n, len(refs))
    w = Writer(gid, [r.size for r in refs])
    rl = np.diff(off.astype(np.int64)).astype(np.uint64)
    want_sam, want_apf = g["sam"].tobytes(), g["apf"].tobytes()
    for threads in (1, 3):
        sam = w.sam_header(cases.CMD_LINE) + w.format(g["cord_off"], g["cords_str"], g["cords_end"], rl, rid, "sam", threads)
        apf = w.format(g["cord_off"], g["cords_str"], g["cords_end"], rl, rid, "apf", threads)
        assert sam == want_sam, first_diff(want_sam, sam)
        assert apf == want_apf, first_diff(want_apf, apf)
    w.close()


@pytest.mark.skipif(not os.path.exists("/root/reference/src/pmpfinder.cpp"), reason="reference tree not present")
def test_writer_equals_live_reference_on_fresh_reads(oracle_lib):
    """Fresh seed without a stored golden: chimeric, inverted and repeat-rich reads -> several records per read, 'X' gaps, large
    shifts split by thd_DI / thd_X."""
    from linear_amd import synth
    from linear_amd.api import Writer
    refs = [synth.repeat_ref(500_000, 77), synth.random_ref(300_000, 78)]
    rng = np.random.default_rng(79)
    lst = []
    for k in range(40):
        a = synth.mutate(refs[0][20_000 + 9000 * k: 24_000 + 9000 * k], 0.06, rng)
        b = synth.mutate(refs[k % 2][100_000 + 2000 * k: 103_000 + 2000 * k + 150 * (k % 7)], 0.06, rng)
        lst.append(np.concatenate([a, synth.revcomp(b) if k % 3 == 0 else b]))
    reads, off = synth.pack_reads(lst)
    r = oracle_lib.Checker("ref", refs, 2)
    n = off.size - 1
    rid, gid = cases.text_ids(n, len(refs))
    want_sam, want_apf = r.format(reads, off, rid, gid, "x")
    cs_l, ce_l, coff = [], [], np.zeros(n + 1, np.uint64)
    for i in range(n):
        cs, ce = r.map_read(reads[int(off[i]):int(off[i + 1])])
        cs_l.append(cs); ce_l.append(ce); coff[i + 1] = coff[i] + cs.size
    w = Writer(gid, [x.size for x in refs])
    rl = np.diff(off.astype(np.int64)).astype(np.uint64)
    sam = w.sam_header("x") + w.format(coff, np.concatenate(cs_l), np.concatenate(ce_l), rl, rid, "sam")
    apf = w.format(coff, np.concatenate(cs_l), np.concatenate(ce_l), rl, rid, "apf")
    assert sam == want_sam, first_diff(want_sam, sam)
    assert apf == want_apf, first_diff(want_apf, apf)
    assert b"X" in want_sam.split(b"\n", 5)[-1] and want_sam.count(b"\t2048\t") + want_sam.count(b"\t2064\t") >= 5
