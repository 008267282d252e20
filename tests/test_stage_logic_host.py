"""CPU: the product's stage logic (linear_amd/csrc/lnr_hd.h + ref_sort.h -- the code the HIP
kernels execute) compiled for the host, against the oracle and the reference's goldens.
Bit-exact (integer words)."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import cases, shimlib

GOLD = os.path.join(os.path.dirname(__file__), "golden")
PARAMS = [(n, T) for n, (_, Ts) in cases.CASES.items() for T in Ts]


def test_ref_sort_equals_libstdcxx_sort_with_ties():
    """ref_sort must reproduce std::sort's permutation for tied keys."""
    shimlib.build()
    lib = C.CDLL(shimlib.SO)
    p = C.POINTER(C.c_uint64)
    rng = np.random.default_rng(7)
    for it in range(400):
        n = int(rng.integers(0, 60)) if it % 3 == 0 else int(rng.integers(0, 5000))
        if it % 97 == 0:
            n = 150_000
        keys = int(rng.integers(1, 12)) if it % 2 else int(rng.integers(1, 2000))
        a = (rng.integers(0, keys, size=n).astype(np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
        if it % 4 == 1:
            a.sort()
        if it % 4 == 2:
            a = np.sort(a)[::-1].copy()
        for desc in (0, 1):
            b, c = a.copy(), a.copy()
            lib.hs_ref_sort_hi32(b.ctypes.data_as(p), C.c_uint64(n), desc)
            lib.hs_std_sort_hi32(c.ctypes.data_as(p), C.c_uint64(n), desc)
            assert np.array_equal(b, c), f"n={n} keys={keys} desc={desc}"


def test_branch_free_chain_scores_equal_the_literal_ones():
    """The lane-parallel DP scores pairs with chain_score_bl / chain_score0_bl (float quotient estimate + exact
    correction); they must agree with the literal getApxChainScore restatements on every input."""
    shimlib.build()
    lib = C.CDLL(shimlib.SO)
    lib.hs_chain_score_fuzz.restype = C.c_uint64
    lib.hs_chain_score_fuzz.argtypes = [C.c_uint64, C.c_uint64]
    for seed in range(4):
        assert lib.hs_chain_score_fuzz(seed, 5_000_000) == 0


def test_block_score2_in_32_bits_equals_the_literal_one():
    """The block DP scores with a 32-bit getApxChainScore2 (float quotient estimate + exact correction); it must agree with the
    literal 64-bit restatement on every pair of cords."""
    shimlib.build()
    lib = C.CDLL(shimlib.SO)
    lib.hs_block_score2_fuzz.restype = C.c_uint64
    lib.hs_block_score2_fuzz.argtypes = [C.c_uint64, C.c_uint64]
    for seed in range(4):
        assert lib.hs_block_score2_fuzz(seed, 5_000_000) == 0


@pytest.mark.parametrize("name,T", PARAMS)
def test_stage_logic_matches_golden(case_inputs, name, T):
    refs, reads, off = case_inputs(name)
    g = np.load(os.path.join(GOLD, f"{name}_T{T}.npz"))
    s = shimlib.Shim(refs, T)
    # closed-form minimizer sampling + "even position in run" rule == the reference's rolling two-pass build
    assert cases.sha(s.dir()) == str(g["dir_sha"])
    assert cases.sha(s.hs()) == str(g["hs_sha"])
    for k in range(len(refs)):
        f2 = s.f2(k)
        assert f2.shape[0] == int(g["f2_len"][k]) and cases.sha(f2[:-1]) == str(g["f2_sha"][k])
    for k, i in enumerate(g["stage_reads"]):
        rd = reads[int(off[i]):int(off[i + 1])]
        a, _ = s.seed_lookup(rd)
        assert np.array_equal(a, g[f"st{k}_raw"])
        a7, _ = s.seed_lookup(rd, 100, rd.size - 50, 7)
        assert np.array_equal(a7, g[f"st{k}_raw7"])
        assert np.array_equal(s.read_features(rd, 0), g[f"st{k}_f1fwd"])
        assert np.array_equal(s.read_features(rd, 1), g[f"st{k}_f1rev"])
        s.map_read(rd, dbg=True)
        for st, nm in ((1, "filt"), (2, "xsort"), (3, "hits")):
            assert np.array_equal(s.stage(st), g[f"st{k}_{nm}"]), f"stage {nm} read {i}"
    coff = g["cord_off"]
    for i in range(off.size - 1):
        cs, ce = s.map_read(reads[int(off[i]):int(off[i + 1])])
        assert np.array_equal(cs, g["cords_str"][int(coff[i]):int(coff[i + 1])]), f"read {i}"
        assert np.array_equal(ce, g["cords_end"][int(coff[i]):int(coff[i + 1])]), f"read {i}"
    s.close()


def test_packed_minimizer_equals_byte_form(case_inputs):
    """2-bit packed read path of the seed kernel == byte path (which is pinned to the reference above)."""
    shimlib.build()
    lib = C.CDLL(shimlib.SO)
    lib.hs_packed_vs_bytes.restype = C.c_uint64
    lib.hs_packed_vs_bytes.argtypes = [C.POINTER(C.c_uint8), C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    tot = fbs = 0
    for name in ("ont", "edge"):
        refs, reads, off = case_inputs(name)
        for i in range(off.size - 1):
            rd = np.ascontiguousarray(reads[int(off[i]):int(off[i + 1])])
            if rd.size < 300:
                continue
            for (rs, re, al) in ((0, rd.size, 15), (100, rd.size - 37, 7)):
                fb, ns = C.c_uint64(), C.c_uint64()
                bad = lib.hs_packed_vs_bytes(rd.ctypes.data_as(C.POINTER(C.c_uint8)), rd.size, rs, re, al, C.byref(fb), C.byref(ns))
                assert bad == 0, f"{name} read {i}"
                tot += ns.value
                fbs += fb.value
    assert tot > 50_000 and fbs < tot * 0.05


def test_packed_cell_features_equal_byte_features(case_inputs):
    """Window features from the 2-bit packed strands (per-cell counts summed three at a time) == byte-wise features."""
    shimlib.build()
    lib = C.CDLL(shimlib.SO)
    lib.hs_read_features_packed.restype = C.c_uint64
    lib.hs_read_features_packed.argtypes = [C.POINTER(C.c_uint8), C.c_uint64, C.c_int, C.POINTER(C.c_int32), C.c_uint64]
    lib.hs_read_features.restype = C.c_uint64
    lib.hs_read_features.argtypes = [C.POINTER(C.c_uint8), C.c_uint64, C.c_int, C.POINTER(C.c_int32), C.c_uint64]
    for name in ("ont", "edge"):
        refs, reads, off = case_inputs(name)
        for i in range(0, off.size - 1, 3):
            rd = np.ascontiguousarray(reads[int(off[i]):int(off[i + 1])])
            if rd.size <= 200:
                continue
            cap = rd.size // 16 + 8
            for strand in (0, 1):
                a = np.zeros((cap, 3), np.int32)
                b = np.zeros((cap, 3), np.int32)
                na = lib.hs_read_features(rd.ctypes.data_as(C.POINTER(C.c_uint8)), rd.size, strand, a.ctypes.data_as(C.POINTER(C.c_int32)), cap)
                nb = lib.hs_read_features_packed(rd.ctypes.data_as(C.POINTER(C.c_uint8)), rd.size, strand, b.ctypes.data_as(C.POINTER(C.c_int32)), cap)
                assert na == nb and np.array_equal(a[:na], b[:nb]), f"{name} read {i} strand {strand}"
