"""CPU: the bottom layer of the gap path restatement (oracle/lnr_gap.inc, SURVEY 8 f1 -- work in progress) against the
reference's own functions, called one by one through oracle/_ref (ref_gap_* hooks).  Only runs where the reference tree is."""
import ctypes as C
import os

import numpy as np
import pytest

from linear_amd import synth

pytestmark = pytest.mark.skipif(not os.path.exists("/root/reference/src/gap_util.cpp"), reason="reference tree not present (GPU box)")
HERE = os.path.dirname(os.path.abspath(__file__))
u8p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)


def libs():
    from oracle import pyorc
    pyorc.build(ref=True)
    o = C.CDLL(os.path.join(HERE, "..", "oracle", "liblnr_oracle.so"))
    r = C.CDLL(os.path.join(HERE, "..", "oracle", "_ref", "libref_linear.so"))
    for lib, pfx in ((o, "orc_"), (r, "ref_")):
        f = getattr(lib, pfx + "gap_anchors")
        f.restype = C.c_uint64
        f.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_uint64, u64p, C.c_uint64]
        f = getattr(lib, pfx + "gap_anchor_pair")
        f.restype = C.c_uint64
        f.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, u64p, u64p, u64p, C.c_uint64]
        f = getattr(lib, pfx + "gap_canchors")
        f.restype = C.c_uint64
        f.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, u64p, C.c_uint64]
        f = getattr(lib, pfx + "gap_score")
        f.restype = C.c_int
        f.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]
        f = getattr(lib, pfx + "gap_xdrop")
        f.restype = C.c_uint64
        f.argtypes = [u64p, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_int)]
    return o, r


def cord(idx, x, y, strand=0):
    return (idx << 50) | (x << 20) | y | (strand << 61)


def make_pair(seed, glen=6000, rlen=5000, err=0.08, with_n=False):
    rng = np.random.default_rng(seed)
    g = synth.random_ref(glen, seed)
    if with_n:
        g = synth.add_n_runs(g, seed + 1, n_runs=2, max_run=40)
    x0 = int(rng.integers(0, glen - rlen))
    reads, off, _ = synth.sample_reads([g[x0:x0 + rlen + 200]], 1, rlen, err, seed + 2, "none")
    return np.ascontiguousarray(g), np.ascontiguousarray(reads[: int(off[1])]), x0


def p(a, t):
    return a.ctypes.data_as(t)


def test_gap_kmer_anchors_match_reference():
    o, r = libs()
    cap = 1 << 20
    for seed in range(12):
        g, rd, x0 = make_pair(100 + seed, with_n=seed % 3 == 0)
        rng = np.random.default_rng(seed)
        for shape_len, s1, s2 in ((9, 5, 1), (5, 3, 1), (13, 4, 2)):
            xs, ys = x0 + int(rng.integers(0, 300)), int(rng.integers(0, 300))
            xe, ye = min(xs + int(rng.integers(500, 3000)), g.size - 1), min(ys + int(rng.integers(500, 3000)), rd.size - 1)
            for strand in (0, 1):
                gs, ge = cord(0, xs, ys, strand), cord(0, xe, ye, strand)
                for direction, lo, hi in ((0, xs - ys - 150, xs - ys + 150), (1, 0, 0), (-1, 0, 0)):
                    a, b = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
                    na = o.orc_gap_anchors(p(g, u8p), g.size, p(rd, u8p), rd.size, gs, ge, shape_len, s1, s2, direction, lo, hi, rd.size - 1, p(a, u64p), cap)
                    nb = r.ref_gap_anchors(p(g, u8p), g.size, p(rd, u8p), rd.size, gs, ge, shape_len, s1, s2, direction, lo, hi, rd.size - 1, p(b, u64p), cap)
                    assert na == nb and np.array_equal(a[:na], b[:nb]), (seed, shape_len, strand, direction, na, nb)
            gs1, ge1, gs2, ge2 = cord(0, xs, ys), cord(0, xs + 800, ys + 800), cord(0, xe - 800, ye - 800), cord(0, xe, ye)
            a1, a2, b1, b2 = (np.zeros(cap, np.uint64) for _ in range(4))
            n1a, n1b = C.c_uint64(), C.c_uint64()
            n2a = o.orc_gap_anchor_pair(p(g, u8p), g.size, p(rd, u8p), rd.size, gs1, ge2, shape_len, s1, s2, rd.size - 1, gs1, ge1, gs2, ge2, p(a1, u64p), C.byref(n1a), p(a2, u64p), cap)
            n2b = r.ref_gap_anchor_pair(p(g, u8p), g.size, p(rd, u8p), rd.size, gs1, ge2, shape_len, s1, s2, rd.size - 1, gs1, ge1, gs2, ge2, p(b1, u64p), C.byref(n1b), p(b2, u64p), cap)
            assert n1a.value == n1b.value and n2a == n2b and np.array_equal(a1[: n1a.value], b1[: n1b.value]) and np.array_equal(a2[:n2a], b2[:n2b])
        for shape_len, step in ((4, 1), (8, 2), (3, 1)):
            a, b = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
            args = (p(g, u8p), g.size, p(rd, u8p), rd.size, x0 + 100, x0 + 900, 100, 900, step, 1, shape_len, x0 - 60, x0 + 60)
            na, nb = o.orc_gap_canchors(*args, p(a, u64p), cap), r.ref_gap_canchors(*args, p(b, u64p), cap)
            assert na == nb and np.array_equal(a[:na], b[:nb]), ("c", seed, shape_len)


def test_gap_chain_scores_and_xdrop_match_reference():
    o, r = libs()
    rng = np.random.default_rng(5)

    def anchor(x, y, s):
        return (s << 50) | (((x - y + (1 << 20)) & ((1 << 30) - 1)) << 20) | y

    for _ in range(20000):
        x1, y1 = int(rng.integers(2000, 60000)), int(rng.integers(0, 9000))
        dx, dy = int(rng.integers(-400, 1500)), int(rng.integers(-400, 1500))
        if rng.random() < 0.3:
            dx = dy + int(rng.integers(-20, 20))
        a1, a2 = anchor(x1 + dx, min(max(y1 + dy, 0), (1 << 20) - 1), int(rng.integers(0, 2))), anchor(x1, y1, int(rng.integers(0, 2)))
        for w in (1, 2):
            assert o.orc_gap_score(w, a1, a2, 0, 0, 0, 0) == r.ref_gap_score(w, a1, a2, 0, 0, 0, 0)
        s1, s2 = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        c11 = cord(0, x1, y1, s1); c12 = cord(0, x1 + 96, y1 + 96, s1)
        c21 = cord(0, max(x1 + dx, 0), max(y1 + dy, 0), s2); c22 = cord(0, max(x1 + dx, 0) + 96, max(y1 + dy, 0) + 96, s2)
        for w in (3, 4):
            for cs in (0, 1):
                assert o.orc_gap_score(w, c11, c12, c21, c22, 10000, cs) == r.ref_gap_score(w, c11, c12, c21, c22, 10000, cs)
    for _ in range(300):
        n = int(rng.integers(2, 60))
        xs = np.cumsum(rng.integers(1, 90, n)) + 5000
        ys = np.cumsum(rng.integers(1, 90, n)) + 100
        if rng.random() < 0.5:
            k = int(rng.integers(1, n)); xs[k:] += int(rng.integers(100, 600))
        ch = np.array([anchor(int(x), int(y), 0) for x, y in zip(xs, ys)], np.uint64)
        for direction in (1, -1):
            for er in (0, 1):
                a, b = ch.copy(), ch.copy()
                ra, rb = C.c_int(), C.c_int()
                na = o.orc_gap_xdrop(p(a, u64p), n, direction, er, C.byref(ra)); nb = r.ref_gap_xdrop(p(b, u64p), n, direction, er, C.byref(rb))
                assert na == nb and ra.value == rb.value and np.array_equal(a[:na], b[:nb])


def test_gap_chains_from_anchors_match_reference():
    """anchors -> chains (chainAnchorsBase with the gap scores) -> tiles -> block chaining of the tiles (chainTiles), with both sets
    of chain metrics, and the choice of the chain that continues the gap's end (getClosestExtensionChain_)."""
    o, r = libs()
    for lib, pfx in ((o, "orc_"), (r, "ref_")):
        f = getattr(lib, pfx + "gap_chains")
        f.restype = C.c_uint64
        f.argtypes = [u64p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int, u64p, C.c_uint64, C.POINTER(C.c_int)]
    cap = 1 << 20
    n_nonempty = 0
    for seed in range(16):
        g, rd, x0 = make_pair(300 + seed, err=0.05 + 0.01 * (seed % 8))
        if seed % 4 == 1:   # a duplicated stretch: two chains compete
            g = np.ascontiguousarray(np.concatenate([g[:3000], g[2200:3000], g[3000:]]))
        rng = np.random.default_rng(seed)
        xs, ys = x0 + 50, 50
        xe, ye = min(xs + 2500, g.size - 1), min(ys + 2500, rd.size - 1)
        gs, ge = cord(0, xs, ys), cord(0, xe, ye)
        for shape_len, s1, s2, direction in ((9, 5, 1, 0), (5, 3, 1, 1), (9, 5, 1, -1)):
            a = np.zeros(cap, np.uint64)
            lo, hi = xs - ys - 200, xs - ys + 200
            na = o.orc_gap_anchors(p(g, u8p), g.size, p(rd, u8p), rd.size, gs, ge, shape_len, s1, s2, direction, lo, hi, rd.size - 1, p(a, u64p), cap)
            anc = np.ascontiguousarray(a[:na])
            for alt in (0, 1):
                for closest in (0, 1, 2):
                    ta, tb = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
                    pa, pb = (C.c_int * 2)(), (C.c_int * 2)()
                    d = direction if direction else 1
                    n1 = o.orc_gap_chains(p(anc, u64p), na, rd.size, alt, d, gs, ge, closest, p(ta, u64p), cap, pa)
                    n2 = r.ref_gap_chains(p(anc, u64p), na, rd.size, alt, d, gs, ge, closest, p(tb, u64p), cap, pb)
                    assert n1 == n2 and np.array_equal(ta[:n1], tb[:n2]) and (not closest or (pa[0], pa[1]) == (pb[0], pb[1])), (seed, shape_len, direction, alt, closest, n1, n2)
                    n_nonempty += n1 > 0
    assert n_nonempty > 50


def test_gap_map_generic_and_extend_match_reference():
    """mapGeneric / mapExtend / mapExtends (k-mer join, chaining, re-extension with the small pattern, clipping, tiles along the chains
    with the window features, reform_tiles) of reads with planted insertions and deletions, against the reference."""
    from oracle import pyorc
    o, r = libs()
    for lib, pfx in ((o, "orc_"), (r, "ref_")):
        f = getattr(lib, pfx + "gap_map")
        f.restype = C.c_uint64
        f.argtypes = [C.c_void_p, u8p, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, u64p, u64p, u64p, C.c_uint64]
    rng = np.random.default_rng(11)
    g = synth.random_ref(60000, 77)
    co, cr = pyorc.Checker("oracle", [g], 1), pyorc.Checker("ref", [g], 1)
    cap = 4096
    nontrivial = 0
    for k in range(40):
        x0 = int(rng.integers(1000, 40000))
        L = 6000
        seg = g[x0:x0 + L + 600].copy()
        kind = k % 4
        cut = int(rng.integers(2000, 3500))
        if kind == 1:   # deletion in the read
            seg = np.concatenate([seg[:cut], seg[cut + int(rng.integers(150, 500)):]])
        elif kind == 2:  # insertion in the read
            seg = np.concatenate([seg[:cut], rng.integers(0, 4, int(rng.integers(150, 500)), dtype=np.uint8), seg[cut:]])
        elif kind == 3:  # tandem duplication
            d = int(rng.integers(150, 400)); seg = np.concatenate([seg[:cut], seg[cut - d:cut], seg[cut:]])
        reads, off, _ = synth.sample_reads([seg], 1, L, 0.06, 500 + k, "none")
        rd = np.ascontiguousarray(reads[: int(off[1])])
        ys, ye = cut - int(rng.integers(300, 900)), min(cut + int(rng.integers(600, 1400)), rd.size - 200)
        xs, xe = x0 + ys, x0 + ye + int(rng.integers(-200, 200))
        gs, ge = cord(0, xs, ys), cord(0, xe, ye)
        cases_ = [(1, gs, ge, 0, 0, 0, 0), (1, gs, ge, 0, 0, 0, 1), (2, gs, cord(0, xs + 1500, ys + 1500), 0, 0, 1, 0), (2, cord(0, xe - 1500, ye - 1500), ge, 0, 0, -1, 0),
                  (3, gs, cord(0, xs + 1200, ys + 1200), cord(0, xe - 1200, ye - 1200), ge, 0, 1)]
        for which, a, b, c2, d2, direction, alt in cases_:
            sa, ea, sb, eb = (np.zeros(cap, np.uint64) for _ in range(4))
            na2, nb2 = C.c_uint64(), C.c_uint64()
            na = o.orc_gap_map(co.h, p(rd, u8p), rd.size, which, a, b, c2, d2, direction, alt, p(sa, u64p), p(ea, u64p), C.byref(na2), cap)
            nb = r.ref_gap_map(cr.h, p(rd, u8p), rd.size, which, a, b, c2, d2, direction, alt, p(sb, u64p), p(eb, u64p), C.byref(nb2), cap)
            tot = (na & 0xffffffff) + na2.value
            assert na == nb and na2.value == nb2.value and np.array_equal(sa[:tot], sb[:tot]) and np.array_equal(ea[:tot], eb[:tot]), (k, kind, which, direction, alt, na, nb)
            nontrivial += tot > 3
    assert nontrivial > 60
