"""CPU: the product's gap path (linear_amd/csrc/lnr_gap_hd.h, SURVEY 8 f1 -- work in progress), compiled for the host by the test
shim, against the oracle's restatement (oracle/lnr_gap.inc), layer by layer through matching hooks."""
import ctypes as C
import os

import numpy as np
import pytest

from linear_amd import synth
from tests import shimlib

HERE = os.path.dirname(os.path.abspath(__file__))
u8p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
SIGS = {
    "gap_anchors": (C.c_uint64, [u8p, C.c_uint64, u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_uint64, u64p, C.c_uint64]),
    "gap_anchor_pair": (C.c_uint64, [u8p, C.c_uint64, u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, u64p, u64p, u64p,
                                     C.c_uint64]),
    "gap_canchors": (C.c_uint64, [u8p, C.c_uint64, u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, u64p, C.c_uint64]),
    "gap_score": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]),
}


def libs():
    from oracle import pyorc
    pyorc.build(ref=False)
    shimlib.build()
    o = C.CDLL(os.path.join(HERE, "..", "oracle", "liblnr_oracle.so"))
    s = C.CDLL(shimlib.SO)
    for lib, pfx in ((o, "orc_"), (s, "hs_")):
        for name, (res, args) in SIGS.items():
            f = getattr(lib, pfx + name, None)
            if f is not None:
                f.restype, f.argtypes = res, args
    return o, s


def cord(idx, x, y, strand=0):
    return (idx << 50) | (x << 20) | y | (strand << 61)


def p(a, t):
    return a.ctypes.data_as(t)


def make_pair(seed, glen=6000, rlen=5000, err=0.08, with_n=False):
    rng = np.random.default_rng(seed)
    g = synth.random_ref(glen, seed)
    if with_n:
        g = synth.add_n_runs(g, seed + 1, n_runs=2, max_run=40)
    x0 = int(rng.integers(0, glen - rlen))
    reads, off, _ = synth.sample_reads([g[x0:x0 + rlen + 200]], 1, rlen, err, seed + 2, "none")
    return np.ascontiguousarray(g), np.ascontiguousarray(reads[: int(off[1])]), x0


def test_product_gap_anchors_and_scores_match_oracle():
    o, s = libs()
    cap = 1 << 20
    for seed in range(10):
        g, rd, x0 = make_pair(900 + seed, with_n=seed % 3 == 0)
        rng = np.random.default_rng(seed)
        for shape_len, s1, s2 in ((9, 5, 1), (5, 3, 1), (13, 4, 2)):
            xs, ys = x0 + int(rng.integers(0, 300)), int(rng.integers(0, 300))
            xe, ye = min(xs + int(rng.integers(500, 3000)), g.size - 1), min(ys + int(rng.integers(500, 3000)), rd.size - 1)
            for strand in (0, 1):
                gs, ge = cord(0, xs, ys, strand), cord(0, xe, ye, strand)
                for direction, lo, hi in ((0, xs - ys - 150, xs - ys + 150), (1, 0, 0), (-1, 0, 0)):
                    a, b = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
                    args = (p(g, u8p), g.size, p(rd, u8p), rd.size, gs, ge, shape_len, s1, s2, direction, lo, hi, rd.size - 1)
                    na, nb = o.orc_gap_anchors(*args, p(a, u64p), cap), s.hs_gap_anchors(*args, p(b, u64p), cap)
                    assert na == nb and np.array_equal(a[:na], b[:nb]), (seed, shape_len, strand, direction, na, nb)
            gs1, ge1, gs2, ge2 = cord(0, xs, ys), cord(0, xs + 800, ys + 800), cord(0, xe - 800, ye - 800), cord(0, xe, ye)
            a1, a2, b1, b2 = (np.zeros(cap, np.uint64) for _ in range(4))
            n1a, n1b = C.c_uint64(), C.c_uint64()
            args = (p(g, u8p), g.size, p(rd, u8p), rd.size, gs1, ge2, shape_len, s1, s2, rd.size - 1, gs1, ge1, gs2, ge2)
            n2a = o.orc_gap_anchor_pair(*args, p(a1, u64p), C.byref(n1a), p(a2, u64p), cap)
            n2b = s.hs_gap_anchor_pair(*args, p(b1, u64p), C.byref(n1b), p(b2, u64p), cap)
            assert n1a.value == n1b.value and n2a == n2b and np.array_equal(a1[: n1a.value], b1[: n1b.value]) and np.array_equal(a2[:n2a], b2[:n2b])
        for shape_len, step in ((4, 1), (8, 2), (3, 1)):
            a, b = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
            args = (p(g, u8p), g.size, p(rd, u8p), rd.size, x0 + 100, x0 + 900, 100, 900, step, 1, shape_len, x0 - 60, x0 + 60)
            na, nb = o.orc_gap_canchors(*args, p(a, u64p), cap), s.hs_gap_canchors(*args, p(b, u64p), cap)
            assert na == nb and np.array_equal(a[:na], b[:nb]), ("c", seed, shape_len)
    rng = np.random.default_rng(5)

    def anchor(x, y, st):
        return (st << 50) | (((x - y + (1 << 20)) & ((1 << 30) - 1)) << 20) | y

    for _ in range(20000):
        x1, y1 = int(rng.integers(2000, 60000)), int(rng.integers(0, 9000))
        dx, dy = int(rng.integers(-400, 1500)), int(rng.integers(-400, 1500))
        if rng.random() < 0.3:
            dx = dy + int(rng.integers(-20, 20))
        a1, a2 = anchor(x1 + dx, min(max(y1 + dy, 0), (1 << 20) - 1), int(rng.integers(0, 2))), anchor(x1, y1, int(rng.integers(0, 2)))
        for w in (1, 2):
            assert o.orc_gap_score(w, a1, a2, 0, 0, 0, 0) == s.hs_gap_score(w, a1, a2, 0, 0, 0, 0)
        s1, s2 = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        c11 = cord(0, x1, y1, s1); c12 = cord(0, x1 + 96, y1 + 96, s1)
        c21 = cord(0, max(x1 + dx, 0), max(y1 + dy, 0), s2); c22 = cord(0, max(x1 + dx, 0) + 96, max(y1 + dy, 0) + 96, s2)
        for w in (3, 4):
            for cs in (0, 1):
                assert o.orc_gap_score(w, c11, c12, c21, c22, 10000, cs) == s.hs_gap_score(w, c11, c12, c21, c22, 10000, cs)


def test_product_gap_chains_match_oracle():
    o, s = libs()
    for lib, pfx in ((o, "orc_"), (s, "hs_")):
        f = getattr(lib, pfx + "gap_chains")
        f.restype = C.c_uint64
        f.argtypes = [u64p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int, u64p, C.c_uint64, C.POINTER(C.c_int)]
    cap = 1 << 20
    n_nonempty = 0
    for seed in range(16):
        g, rd, x0 = make_pair(300 + seed, err=0.05 + 0.01 * (seed % 8))
        if seed % 4 == 1:
            g = np.ascontiguousarray(np.concatenate([g[:3000], g[2200:3000], g[3000:]]))
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
                    n2 = s.hs_gap_chains(p(anc, u64p), na, rd.size, alt, d, gs, ge, closest, p(tb, u64p), cap, pb)
                    assert n1 == n2 and np.array_equal(ta[:n1], tb[:n2]) and (not closest or (pa[0], pa[1]) == (pb[0], pb[1])), (seed, shape_len, direction, alt, closest, n1, n2)
                    n_nonempty += n1 > 0
    assert n_nonempty > 50


def sv_reads(refs, n, seed, L=8000):
    """reads with planted insertions, deletions, duplications, inversions and translocated stretches"""
    rng = np.random.default_rng(seed)
    cpl = np.array([3, 2, 1, 0, 4], np.uint8)
    out = []
    for k in range(n):
        ref = refs[k % len(refs)]
        x0 = int(rng.integers(1000, ref.size - 12000))
        seg = ref[x0:x0 + L + 1000].copy()
        cut = int(rng.integers(2000, 6000)); m = int(rng.integers(60, 1500))
        kind = k % 6
        if kind == 1: seg = np.concatenate([seg[:cut], seg[cut + m:]])
        elif kind == 2: seg = np.concatenate([seg[:cut], rng.integers(0, 4, m, dtype=np.uint8), seg[cut:]])
        elif kind == 3: seg = np.concatenate([seg[:cut], seg[max(cut - m, 0):cut], seg[cut:]])
        elif kind == 4: seg = np.concatenate([seg[:cut], cpl[seg[cut:cut + m][::-1]], seg[cut + m:]])
        elif kind == 5: seg = np.concatenate([seg[:cut], ref[x0 + 20000 - min(20000, x0):][:m], seg[cut:]])
        r, o_, _ = synth.sample_reads([seg], 1, min(seg.size - 50, L), float(rng.choice([0.0, 0.03, 0.1])), seed + 700 + k, "random")
        out.append(np.ascontiguousarray(r[: int(o_[1])]))
    return out


def test_product_gap_map_units_match_oracle():
    """mapGeneric / mapExtend / mapExtends of the product header against the oracle on planted indels (the same cases as
    tests/test_gap_units_cpu.py pins the oracle to the reference with)."""
    from oracle import pyorc
    o, s = libs()
    sig = (C.c_uint64, [C.c_void_p, u8p, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, u64p, u64p, u64p, C.c_uint64])
    o.orc_gap_map.restype, o.orc_gap_map.argtypes = sig
    s.hs_gap_map.restype, s.hs_gap_map.argtypes = sig
    rng = np.random.default_rng(11)
    g = synth.random_ref(60000, 77)
    co, cs_ = pyorc.Checker("oracle", [g], 1), shimlib.Shim([g], 1)
    cap = 4096
    nontrivial = 0
    for k in range(40):
        x0 = int(rng.integers(1000, 40000))
        L = 6000
        seg = g[x0:x0 + L + 600].copy()
        kind = k % 4
        cut = int(rng.integers(2000, 3500))
        if kind == 1:
            seg = np.concatenate([seg[:cut], seg[cut + int(rng.integers(150, 500)):]])
        elif kind == 2:
            seg = np.concatenate([seg[:cut], rng.integers(0, 4, int(rng.integers(150, 500)), dtype=np.uint8), seg[cut:]])
        elif kind == 3:
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
            nb = s.hs_gap_map(cs_.h, p(rd, u8p), rd.size, which, a, b, c2, d2, direction, alt, p(sb, u64p), p(eb, u64p), C.byref(nb2), cap)
            tot = (na & 0xffffffff) + na2.value
            assert na == nb and na2.value == nb2.value and np.array_equal(sa[:tot], sb[:tot]) and np.array_equal(ea[:tot], eb[:tot]), (k, kind, which, direction, alt, na, nb)
            nontrivial += tot > 3
    assert nontrivial > 60
    co.close()


@pytest.mark.parametrize("name", ["ont", "edge", "ccs_sv", "rep", "chim"])
def test_product_gap_path_matches_reference_golden(case_inputs, name):
    """apxMap + mapGaps + reformCords of the product's host build against cords the real reference produced with -g 50 [-dup 1] on the case
    as one read stream in file order (`-t 1`): the stream state (thd_cts_major_limit 1 -> 3 at the first extension) is carried read to read"""
    from tests import cases
    refs, reads, off = case_inputs(name)
    g = np.load(os.path.join(HERE, "golden", f"{name}_g50_T1.npz"))
    assert cases.input_digest(refs, reads, off) == str(g["digest"])
    sh = shimlib.Shim(refs, 1)
    for dup in (0, 1):
        co = g[f"cord_off_dup{dup}"]
        ext = 0
        for i in range(off.size - 1):
            cs, ce = sh.map_read_gap(reads[int(off[i]):int(off[i + 1])], 50, dup, ext)
            ext = sh.ext_out
            assert np.array_equal(cs, g[f"cords_str_dup{dup}"][int(co[i]):int(co[i + 1])]), f"dup {dup} read {i}"
            assert np.array_equal(ce, g[f"cords_end_dup{dup}"][int(co[i]):int(co[i + 1])]), f"dup {dup} read {i}"


def test_product_gap_path_matches_oracle_on_planted_svs():
    from oracle import pyorc
    pyorc.build(ref=False)
    refs = [synth.repeat_ref(300_000, 61), synth.add_n_runs(synth.random_ref(200_000, 62), 63, n_runs=2, max_run=600)]
    reads_l = sv_reads(refs, 48, 2027)
    for T in (1, 3):
        o = pyorc.Checker("oracle", refs, T)
        sh = shimlib.Shim(refs, T)
        for i, rd in enumerate(reads_l):
            for gap_len, dup in ((50, 0), (50, 1), (1, 0), (5, 1), (200, 0)):
                for ext in (0, 1):
                    a, b = o.map_read_gap(rd, gap_len, dup, ext), sh.map_read_gap(rd, gap_len, dup, ext)
                    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and o.ext_out == sh.ext_out, (T, i, gap_len, dup, ext)
        o.close()


def test_product_gap_path_survives_small_arenas_and_work_budget():
    """What a k_gap worker does when a read outgrows its arena or its work budget: a clean refusal (the read is then redone with a
    larger arena, or by a whole wave), never a different answer -- whatever the point at which the memory runs out."""
    from oracle import pyorc
    pyorc.build(ref=False)
    refs = [synth.repeat_ref(300_000, 61), synth.add_n_runs(synth.random_ref(200_000, 62), 63, n_runs=2, max_run=600)]
    reads_l = sv_reads(refs, 12, 2029)
    nrun = synth.random_ref(7000, 5).copy(); nrun[600:6400] = 4   # a read of N: 10^5 anchors with thousands of predecessors each
    o = pyorc.Checker("oracle", refs, 1)
    sh = shimlib.Shim(refs, 1)
    fn = sh.lib.hs_map_read_g_lim
    fn.restype = C.c_int64
    fn.argtypes = [C.c_void_p, u8p, C.c_uint64, C.c_uint32, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64]
    ok = refused = 0
    for i, rd in enumerate(reads_l):
        want = o.map_read_gap(rd, 50, 1)
        for arena in (256, 3000, 20_000, 50_000, 90_000, 150_000, 300_000, 1 << 20):
            for keep in (2000, 30_000, 1 << 20):
                n = fn(sh.h, p(rd, u8p), rd.size, 50, 1, arena, keep, 1 << 62)
                assert n >= 0 or n == -11, (i, arena, keep, n)
                if n >= 0:
                    cs, ce = np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.uint64)
                    sh.lib.hs_get_cords(sh.h, p(cs, u64p), p(ce, u64p))
                    assert np.array_equal(cs[:n], want[0]) and np.array_equal(ce[:n], want[1]), (i, arena, keep)
                    ok += 1
                else:
                    refused += 1
    assert ok > 20 and refused > 20, (ok, refused)
    a = o.map_read_gap(nrun, 50, 0)
    assert fn(sh.h, p(nrun, u8p), nrun.size, 50, 0, 64 << 20, 1 << 20, 3_000_000) in (-12, len(a[0]))
    o.close()


def test_product_positive_only_chain_scores_agree_with_the_literal_ones():
    """gap_anchor_score{1,2}_pos (what the chain DP evaluates) = the literal scores wherever those are positive, non-positive elsewhere"""
    o, s = libs()
    rng = np.random.default_rng(77)

    def anchor(x, y, st):
        return (st << 50) | (((x - y + (1 << 20)) & ((1 << 30) - 1)) << 20) | y

    npos = 0
    for it in range(120000):
        y2 = int(rng.integers(0, 900000)); x2 = y2 + int(rng.integers(3000, 200000))
        if it % 3 == 0:     # around the positive region: small dy, small distance from the diagonal
            dy = int(rng.integers(-3, 260)); dx = dy + int(rng.integers(-70, 71))
        elif it % 3 == 1:
            dy = int(rng.integers(-50, 3000)); dx = int(rng.integers(-50, 3000))
        else:
            dy = int(rng.integers(0, 120)); dx = int(rng.integers(-10, 200))
        y1 = min(max(y2 + dy, 0), (1 << 20) - 1)
        if it % 29 == 0:    # far from the diagonal: the literal form's int conversion wraps, and so must the other
            dx = int(rng.integers(70000, 400000))
        a1, a2 = anchor(x2 + dx, y1, int(rng.integers(0, 2)) if it % 17 == 0 else 0), anchor(x2, y2, 0)
        for lit, pos in ((1, 6), (2, 7)):
            want, got = o.orc_gap_score(lit, a1, a2, 0, 0, 0, 0), s.hs_gap_score(pos, a1, a2, 0, 0, 0, 0)
            assert (got == want) if want > 0 else (got <= 0), (lit, dx, dy, want, got)
            npos += want > 0
    assert npos > 10000


def test_column_dp_delta_scores_equal_the_anchor_scores():
    """the team's column DP (lnr_gap_hd.h gap_dp_columns) scores a predecessor from (dx, dy) behind a box test; over every (dx, dy) of a rectangle
    around the boxes, on equal and on opposite strands: positive exactly where the anchor forms (what the serial DP and the oracle-pinned host path use)
    are positive, and the same value there."""
    sh = shimlib.Shim([synth.random_ref(5000, 1)], 1)
    f = sh.lib.hs_gap_delta_check
    f.restype = C.c_uint64
    f.argtypes = [C.c_int, C.c_uint32, C.c_int32, C.c_int32]
    for fn, dxm in ((1, 400), (2, 4300), (5, 200)):
        assert f(fn, dxm, -6, 320) == 0, fn
