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
