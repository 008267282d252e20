"""CPU: the parity pin to THE PROGRAM USERS RUN.  tests/golden/cli_<case>.npz hold the bytes of the .sam and .apf the real `linear filter`
binary wrote (oracle/_ref/linear = the reference's own translation units incl. linear.cpp / mapper.cpp / parallel_io.cpp / args_parser.cpp,
compiled in place by oracle/Makefile; made by tools/make_cli_golden.py at `-t 1`) for -g 0, -g 50, -g 50 -dup 1 and no -g at all.

Here: the CPU restatement (oracle, file-order stream semantics of the gap re-mapper's per-thread GapParms) + the product's writer
(lnr_writer, incl. the SA:Z NM cache of createSAZTagCigarOneChimeric) reproduce those bytes.  The GPU twin of this test
(tests/test_gpu_parity.py::test_gpu_linear_filter_cli_equals_the_real_program) runs the product's own `linear_filter` binary on the FASTA files.

Known and excluded: read_11 of the `edge` case at -g > 0 (5 827 N of 6 987 bases).  The program's result for that read depends on which reads
were processed before it in ways no parameter carries ([6, 11] changes it, [0..11] does not; the same reference code called on
zeroed slack memory -- oracle/_ref/libref_linear.so -- gives the oracle's answer): reads past the end of SeqAn strings, i.e. heap contents."""
import os

import numpy as np
import pytest

from tests import cases

GOLD = os.path.join(os.path.dirname(__file__), "golden")
UB_READS = {("edge", "g50"): {b"read_11 len extra=33"}, ("edge", "g50dup1"): {b"read_11 len extra=33"}, ("edge", "gdef"): {b"read_11 len extra=33"}}
MODE_OPTS = {"g0": (0, 0), "g50": (50, 0), "g50dup1": (50, 1), "gdef": (1, 0)}     # gdef: no -g on the command line = Options::gap_len 1 -> 50


def sam_by_read(text: bytes):
    head, recs = [], {}
    for l in text.split(b"\n"):
        if l.startswith(b"@"):
            head.append(l)
        elif l:
            recs.setdefault(l.split(b"\t")[0], []).append(l)
    return head, recs


def apf_by_read(text: bytes):
    recs, cur = {}, None
    for l in text.split(b"\n"):
        if not l:
            continue                       # blank lines depend on the reference's adaptive block size
        if l.startswith(b"@ "):
            cur = l[2:].rsplit(b" ", 8)[0]
        recs.setdefault(cur, []).append(l)
    return recs


@pytest.mark.parametrize("name", list(cases.CASES_CLI))
def test_oracle_plus_writer_reproduce_the_real_program(oracle_lib, case_inputs, name):
    from linear_amd import build as lb
    lb.build()
    from linear_amd.api import Writer
    refs, reads, off = cases.CASES_CLI[name]() if name not in cases.CASES and name not in cases.CASES_G50 else case_inputs(name)
    g = np.load(os.path.join(GOLD, f"cli_{name}.npz"))
    assert cases.input_digest(refs, reads, off) == str(g["digest"])
    n = off.size - 1
    rid, gid = cases.text_ids(n, len(refs))
    w = Writer(gid, [r.size for r in refs])
    rl = np.diff(off.astype(np.int64)).astype(np.uint64)
    o = oracle_lib.Checker("oracle", refs, 1)
    multi = 0
    for mode, (gl, dup) in MODE_OPTS.items():
        coff, cs, ce, _ = o.map_batch(reads, off, threads=4, gap_len=gl, dup=dup)
        skip = UB_READS.get((name, mode), set())
        head, recs = sam_by_read(w.sam_header("") + w.format(coff, cs, ce, rl, rid, "sam"))
        whead, wrecs = sam_by_read(g[f"sam_{mode}"].tobytes())
        assert head == whead                                   # incl. `@PG ... CL:` -- the program prints an empty command line
        assert list(recs) == list(wrecs)                       # the same reads have records, in the same order
        for k in wrecs:
            if k not in skip:
                assert recs[k] == wrecs[k], (mode, k)
            multi += len(wrecs[k]) >= 3
        apf, wapf = apf_by_read(w.format(coff, cs, ce, rl, rid, "apf")), apf_by_read(g[f"apf_{mode}"].tobytes())
        assert list(apf) == list(wapf)
        for k in wapf:
            if k not in skip:
                assert apf[k] == wapf[k], (mode, k)
    if name == "rep":
        assert multi >= 20, "reads of >= 3 SAM lines (SA:Z lists of several records: the NM cache) are what this case is for"
    o.close(); w.close()


@pytest.mark.skipif(not os.path.exists("/root/reference/src/gap_util.cpp"), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("name,dup", [("chim", 0), ("edge", 1)])
def test_every_gap_call_of_the_oracle_replays_through_the_reference(oracle_lib, case_inputs, name, dup):
    """Finer than whole reads: every mapExtend / mapExtends / mapGeneric call mapGap_ makes on a case (arguments as the restatement passes them,
    incl. the swapped chain metrics and the stream state) is replayed through the REFERENCE's own function and must return the same tiles --
    this is what found the sequence a call works on (the id of ITS OWN gap_str: gap_util.cpp:4055,4097,4508) for gaps between cords of
    two reference sequences."""
    import ctypes as C
    u8p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
    refs, reads, off = case_inputs(name)
    r, o = oracle_lib.Checker("ref", refs, 1), oracle_lib.Checker("oracle", refs, 1)
    p = lambda a, t: a.ctypes.data_as(t)
    f = o.lib.orc_map_read_g_trace
    f.restype = C.c_uint64
    f.argtypes = [C.c_void_p, u8p, C.c_uint64, C.c_uint32, C.c_int, C.c_int, u64p, C.c_uint64]
    gm = r.lib.ref_gap_map
    gm.restype = C.c_uint64
    gm.argtypes = [C.c_void_p, u8p, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, u64p, u64p, u64p, C.c_uint64]
    buf = np.zeros(1 << 22, np.uint64)
    cap = 1 << 14
    sb, eb = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
    ncalls, kinds, ext = 0, set(), 0
    for idx in range(off.size - 1):
        rd = np.ascontiguousarray(reads[int(off[idx]):int(off[idx + 1])])
        if (name, idx) == ("edge", 11):
            continue
        n = int(f(o.h, p(rd, u8p), rd.size, 50, dup, ext, p(buf, u64p), buf.size))
        assert n <= buf.size
        i = 0
        while i < n:
            which, gs1, ge1, gs2, ge2, d, alt, n1, n2 = (int(x) for x in buf[i:i + 9])
            d = d - (1 << 64) if d >> 63 else d
            tl = buf[i + 9:i + 9 + 2 * (n1 + n2)].reshape(-1, 2)
            i += 9 + 2 * (n1 + n2)
            nb2 = C.c_uint64()
            nb = gm(r.h, p(rd, u8p), rd.size, which, gs1, ge1, gs2, ge2, d, alt, p(sb, u64p), p(eb, u64p), C.byref(nb2), cap)
            m1 = nb & 0xffffffff
            tot = m1 + nb2.value
            assert m1 == n1 and nb2.value == n2 and np.array_equal(sb[:tot], tl[:, 0]) and np.array_equal(eb[:tot], tl[:, 1]), (idx, which, hex(gs1), hex(ge1), d, alt)
            ncalls += 1
            kinds.add((which, alt))
            ext = ext or which != 1
    assert ncalls > 60 and {w for w, _ in kinds} >= ({1, 2, 3} if name == "chim" else {1, 3}) and any(a & 2 for _, a in kinds), (ncalls, kinds)
