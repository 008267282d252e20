"""CPU: the oracle (our restatement) must reproduce the golden vectors that the REAL
reference produced (tools/make_golden.py via oracle/_ref).  Bit-exact: every value on this
path is an integer word."""
import os

import numpy as np
import pytest

from tests import cases

GOLD = os.path.join(os.path.dirname(__file__), "golden")
PARAMS = [(n, T) for n, (_, Ts) in cases.CASES.items() for T in Ts]


def load(name, T):
    return np.load(os.path.join(GOLD, f"{name}_T{T}.npz"))


@pytest.mark.parametrize("name,T", PARAMS)
def test_oracle_matches_reference_golden(oracle_lib, case_inputs, name, T):
    refs, reads, off = case_inputs(name)
    g = load(name, T)
    assert cases.input_digest(refs, reads, off) == str(g["digest"]), "synthetic generator drifted from the golden inputs"
    o = oracle_lib.Checker("oracle", refs, T)
    # index
    dir_, hs = o.dir(), o.hs()
    assert o.lib.orc_fill_mismatch(o.h) == 0
    assert hs.size == int(g["hs_len"])
    assert np.array_equal(hs[:4096], g["hs_head"])
    assert cases.sha(dir_) == str(g["dir_sha"])
    assert cases.sha(hs) == str(g["hs_sha"])
    # genome features (last element excluded)
    for k in range(len(refs)):
        f2 = o.f2(k)
        assert f2.shape[0] == int(g["f2_len"][k])
        assert cases.sha(f2[:-1]) == str(g["f2_sha"][k])
    # stages
    for k, i in enumerate(g["stage_reads"]):
        rd = reads[int(off[i]):int(off[i + 1])]
        for s, nm in enumerate(("raw", "filt", "xsort", "hits")):
            assert np.array_equal(o.stage(rd, s), g[f"st{k}_{nm}"]), f"stage {nm} read {i}"
        assert np.array_equal(o.read_features(rd, 0), g[f"st{k}_f1fwd"])
        assert np.array_equal(o.read_features(rd, 1), g[f"st{k}_f1rev"])
        a7, _ = o.seed_lookup(rd, 100, rd.size - 50, 7)
        assert np.array_equal(a7, g[f"st{k}_raw7"])
    # final cords of every read
    coff, cs, ce, st = o.map_batch(reads, off, threads=4)
    assert np.array_equal(coff, g["cord_off"])
    assert np.array_equal(cs, g["cords_str"])
    assert np.array_equal(ce, g["cords_end"])
    # serial path == threaded path
    i = int(g["stage_reads"][0])
    c1 = o.map_read(reads[int(off[i]):int(off[i + 1])])
    assert np.array_equal(c1[0], cs[int(coff[i]):int(coff[i + 1])])
    o.close()


@pytest.mark.skipif(not os.path.exists("/root/reference/src/pmpfinder.cpp"), reason="reference tree not present (GPU box)")
def test_oracle_matches_live_reference(oracle_lib):
    """Where the reference is buildable, also compare on a fresh seed that has no stored golden."""
    from linear_amd import synth
    ref = synth.repeat_ref(300_000, 2024)
    reads, off, _ = synth.sample_reads([ref], 25, 6000, 0.12, 31, "random")
    o = oracle_lib.Checker("oracle", [ref], 2)
    r = oracle_lib.Checker("ref", [ref], 2)
    assert np.array_equal(o.dir(), r.dir()) and np.array_equal(o.hs(), r.hs())
    for i in range(off.size - 1):
        rd = reads[int(off[i]):int(off[i + 1])]
        a, b = o.map_read(rd), r.map_read(rd)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ---- HIndex (-i 2): ysa is the byte-exact surface (the open-addressed table's layout is not observable)
PARAMS_I2 = [(n, T) for n, (_, Ts) in cases.CASES_I2.items() for T in Ts]


@pytest.mark.parametrize("name,T", PARAMS_I2)
def test_oracle_hindex_matches_reference_golden(oracle_lib, case_inputs, name, T):
    refs, reads, off = case_inputs(name)
    g = np.load(os.path.join(GOLD, f"{name}_i2_T{T}.npz"))
    assert cases.input_digest(refs, reads, off) == str(g["digest"]), "synthetic generator drifted from the golden inputs"
    o = oracle_lib.Checker("oracle", refs, T, index_type=2)
    ysa = o.ysa()
    assert ysa.size == int(g["ysa_len"]) and o.empty_dir() == int(g["empty_dir"])
    assert np.array_equal(ysa[:4096], g["ysa_head"])
    assert cases.sha(ysa) == str(g["ysa_sha"])
    for k, i in enumerate(g["stage_reads"]):
        rd = reads[int(off[i]):int(off[i + 1])]
        assert np.array_equal(o.seed_lookup(rd)[0], g[f"st{k}_raw"]), f"raw anchors read {i}"
        assert np.array_equal(o.seed_lookup(rd, 100, rd.size - 50, 7)[0], g[f"st{k}_raw7"]), f"raw anchors (alpha 7) read {i}"
    coff = g["cord_off"]
    for i in range(off.size - 1):
        cs, ce = o.map_read(reads[int(off[i]):int(off[i + 1])])
        assert np.array_equal(cs, g["cords_str"][int(coff[i]):int(coff[i + 1])]), f"cords read {i}"
        assert np.array_equal(ce, g["cords_end"][int(coff[i]):int(coff[i + 1])])
    o.close()


@pytest.mark.skipif(not os.path.exists("/root/reference/src/pmpfinder.cpp"), reason="reference tree not present (GPU box)")
def test_oracle_hindex_matches_live_reference(oracle_lib):
    from linear_amd import synth
    ref = synth.add_n_runs(synth.repeat_ref(400_000, 77), 3, n_runs=3, max_run=800)
    reads, off, _ = synth.sample_reads([ref], 20, 7000, 0.08, 13, "random")
    for T in (1, 5):
        o = oracle_lib.Checker("oracle", [ref], T, index_type=2)
        r = oracle_lib.Checker("ref", [ref], T, index_type=2)
        assert np.array_equal(o.ysa(), r.ysa()) and o.empty_dir() == r.empty_dir()
        for i in range(off.size - 1):
            rd = reads[int(off[i]):int(off[i + 1])]
            assert np.array_equal(o.seed_lookup(rd)[0], r.seed_lookup(rd)[0])
            a, b = o.map_read(rd), r.map_read(rd)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ---- gap path (-g 50 [-dup 1]): the oracle's restatement (oracle/lnr_gap.inc) against the reference's goldens
@pytest.mark.parametrize("name", ["ont", "edge", "ccs_sv", "rep", "chim"])
def test_oracle_gap_path_matches_reference_golden(oracle_lib, case_inputs, name):
    """the case as ONE read stream in file order (`linear filter -t 1`): the reference keeps one GapParms per thread for the run and the first
    mapExtend / mapExtends leaves thd_cts_major_limit = 3 behind for every later read (read by read here; the batch entry point -- serial until
    the state flips, parallel behind -- must give the same)"""
    refs, reads, off = case_inputs(name)
    g = np.load(os.path.join(GOLD, f"{name}_g50_T1.npz"))
    assert cases.input_digest(refs, reads, off) == str(g["digest"])
    o = oracle_lib.Checker("oracle", refs, 1)
    for dup in (0, 1):
        co = g[f"cord_off_dup{dup}"]
        ext = 0
        for i in range(off.size - 1):
            cs, ce = o.map_read_gap(reads[int(off[i]):int(off[i + 1])], 50, dup, ext)
            ext = o.ext_out
            assert np.array_equal(cs, g[f"cords_str_dup{dup}"][int(co[i]):int(co[i + 1])]), f"dup {dup} read {i}"
            assert np.array_equal(ce, g[f"cords_end_dup{dup}"][int(co[i]):int(co[i + 1])]), f"dup {dup} read {i}"
        assert ext == int(g[f"ext_out_dup{dup}"])
        coff, cs, ce, _ = o.map_batch(reads, off, threads=4, gap_len=50, dup=dup)
        assert np.array_equal(coff, co) and np.array_equal(cs, g[f"cords_str_dup{dup}"]) and np.array_equal(ce, g[f"cords_end_dup{dup}"]) and o.ext_out == ext
    o.close()


@pytest.mark.skipif(not os.path.exists("/root/reference/src/gap_util.cpp"), reason="reference tree not present (GPU box)")
def test_oracle_gap_path_matches_live_reference(oracle_lib):
    """Fresh inputs: reads with planted insertions, deletions, duplications and inversions (the gap re-mapper's cases), several -g
    values, with and without -dup, repeat-rich and multi-sequence references, -t 1 and 3."""
    from linear_amd import synth
    rng = np.random.default_rng(2026)
    refs = [synth.repeat_ref(300_000, 61), synth.add_n_runs(synth.random_ref(200_000, 62), 63, n_runs=2, max_run=600)]
    reads_l = []
    cpl = np.array([3, 2, 1, 0, 4], np.uint8)
    for k in range(60):
        ref = refs[k % 2]
        x0 = int(rng.integers(1000, ref.size - 12000))
        seg = ref[x0:x0 + 9000].copy()
        cut = int(rng.integers(2000, 6000)); n = int(rng.integers(60, 1500))
        kind = k % 6
        if kind == 1: seg = np.concatenate([seg[:cut], seg[cut + n:]])
        elif kind == 2: seg = np.concatenate([seg[:cut], rng.integers(0, 4, n, dtype=np.uint8), seg[cut:]])
        elif kind == 3: seg = np.concatenate([seg[:cut], seg[max(cut - n, 0):cut], seg[cut:]])
        elif kind == 4: seg = np.concatenate([seg[:cut], cpl[seg[cut:cut + n][::-1]], seg[cut + n:]])
        elif kind == 5: seg = np.concatenate([seg[:cut], ref[x0 + 20000 - min(20000, x0):][:n], seg[cut:]])
        r, o_, _ = synth.sample_reads([seg], 1, min(seg.size - 50, 8000), float(rng.choice([0.0, 0.03, 0.1])), 700 + k, "random")
        reads_l.append(np.ascontiguousarray(r[: int(o_[1])]))
    for T in (1, 3):
        o = oracle_lib.Checker("oracle", refs, T)
        r = oracle_lib.Checker("ref", refs, T)
        for i, rd in enumerate(reads_l):
            for gap_len, dup in ((50, 0), (50, 1), (1, 0), (5, 1), (200, 0)):
                for ext in (0, 1):
                    a, b = o.map_read_gap(rd, gap_len, dup, ext), r.map_read_gap(rd, gap_len, dup, ext)
                    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and o.ext_out == r.ext_out, (T, i, gap_len, dup, ext)
