"""GPU (-m gpu): the HIP path, called through the C ABI, against the oracle and the reference's golden
vectors.  Bit-exact: every value on this path is an integer word."""
import os

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
PARAMS = [(n, T) for n, (_, Ts) in cases.CASES.items() for T in Ts]


@pytest.fixture(scope="module")
def flt():
    from linear_amd import build as lb
    lb.build()
    from linear_amd import Filter
    f = Filter(device=0)
    yield f
    f.close()


@pytest.mark.parametrize("name,T", PARAMS)
def test_gpu_matches_golden(flt, case_inputs, name, T):
    refs, reads, off = case_inputs(name)
    g = np.load(os.path.join(GOLD, f"{name}_T{T}.npz"))
    assert cases.input_digest(refs, reads, off) == str(g["digest"])
    info = flt.build_index(refs, T)
    assert info.hs_len == int(g["hs_len"])
    dir_, hs, f2, f2_off = flt.index_export()
    assert cases.sha(dir_) == str(g["dir_sha"]), "dir differs from the reference"
    assert np.array_equal(hs[:4096], g["hs_head"])
    assert cases.sha(hs) == str(g["hs_sha"]), "hs differs from the reference"
    for k in range(len(refs)):
        f = f2[int(f2_off[k]):int(f2_off[k + 1])]
        assert f.shape[0] == int(g["f2_len"][k])
        assert cases.sha(np.ascontiguousarray(f[:-1])) == str(g["f2_sha"][k]), f"f2 of sequence {k}"
    # stage a7: raw anchors
    aoff, anc = flt.seed_lookup_batch(reads, off)
    for k, i in enumerate(g["stage_reads"]):
        got = anc[int(aoff[i]):int(aoff[i + 1])]
        assert np.array_equal(got, g[f"st{k}_raw"]), f"raw anchors of read {i}"
    # the whole path
    coff, cs, ce = flt.filter_batch(reads, off)
    assert np.array_equal(coff, g["cord_off"])
    assert np.array_equal(cs, g["cords_str"])
    assert np.array_equal(ce, g["cords_end"])


def test_gpu_matches_oracle_fresh_seed(flt, oracle_lib):
    """Fresh inputs without a stored golden: oracle as the checker, incl. the seed-byte counters of SURVEY 8(d)."""
    from linear_amd import synth
    refs = [synth.repeat_ref(500_000, 321), synth.add_n_runs(synth.random_ref(200_000, 322), 323, lead=3000)]
    reads, off, _ = synth.sample_reads(refs, 200, 7000, 0.1, 324, "random", len_jitter=0.6)
    o = oracle_lib.Checker("oracle", refs, 3)
    flt.build_index(refs, 3)
    dir_, hs, _, _ = flt.index_export()
    assert np.array_equal(dir_, o.dir()) and np.array_equal(hs, o.hs())
    ooff, ocs, oce, ost = o.map_batch(reads, off, threads=8)
    coff, cs, ce = flt.filter_batch(reads, off)
    assert np.array_equal(coff, ooff) and np.array_equal(cs, ocs) and np.array_equal(ce, oce)
    st = flt.stats()
    assert (st["samples"], st["lookups"], st["bucket_entries"], st["anchors"]) == tuple(int(x) for x in ost[:4])


def test_gpu_edge_batches(flt, oracle_lib):
    """Empty batch, batch of only too-short reads, single read, ragged lengths."""
    from linear_amd import synth
    ref = synth.random_ref(300_000, 77)
    flt.build_index([ref], 1)
    o = oracle_lib.Checker("oracle", [ref], 1)
    coff, cs, ce = flt.filter_batch(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert coff.tolist() == [0] and cs.size == 0
    short, soff = synth.pack_reads([ref[10:60], ref[100:300], np.zeros(0, np.uint8)])
    coff, cs, ce = flt.filter_batch(short, soff)
    assert coff.tolist() == [0, 0, 0, 0]
    lst = [ref[1000:1000 + L].copy() for L in (201, 250, 999, 4097, 12000, 30000)]
    rd, ro = synth.pack_reads(lst)
    coff, cs, ce = flt.filter_batch(rd, ro)
    ooff, ocs, oce, _ = o.map_batch(rd, ro, threads=2)
    assert np.array_equal(coff, ooff) and np.array_equal(cs, ocs) and np.array_equal(ce, oce)
    # offsets that do not start at zero (a window into a larger buffer)
    coff2, cs2, ce2 = flt.filter_batch(rd, ro[2:])
    assert np.array_equal(cs2, ocs[int(ooff[2]):]) and np.array_equal(coff2, ooff[2:] - ooff[2])


def test_gpu_scratch_slicing(oracle_lib):
    """A tiny scratch budget forces the job kernel to run in several slices; results must not change."""
    from linear_amd import Filter, synth
    ref = synth.repeat_ref(400_000, 55)
    reads, off, _ = synth.sample_reads([ref], 120, 5000, 0.08, 56, "random")
    f = Filter(device=0, scratch_budget=4 << 20)
    f.build_index([ref], 1)
    coff, cs, ce = f.filter_batch(reads, off)
    assert f.stats()["job_launches"] > 1
    o = oracle_lib.Checker("oracle", [ref], 1)
    ooff, ocs, oce, _ = o.map_batch(reads, off, threads=4)
    assert np.array_equal(coff, ooff) and np.array_equal(cs, ocs) and np.array_equal(ce, oce)
    f.close()


@pytest.mark.parametrize("var,val", [("LNR_MID_CAP", "64"), ("LNR_DP_SPLIT_CAP", "64"), ("LNR_SPLIT_CAP", "300"), ("LNR_POST_SPLIT", "1")])
def test_gpu_other_size_class_paths(case_inputs, monkeypatch, var, val):
    """Force the reads through the 4-wave kernel, the split path (pre -> 16-wave DP kernel -> post) and the two-lane
    orchestration, and the k_post split (serial stages with one lane per read): same cords as the reference."""
    from linear_amd import Filter
    monkeypatch.setenv(var, val)
    f = Filter(device=0)
    for name, T in (("rep", 1), ("edge", 3), ("ont", 4)):
        refs, reads, off = case_inputs(name)
        g = np.load(os.path.join(GOLD, f"{name}_T{T}.npz"))
        f.build_index(refs, T)
        coff, cs, ce = f.filter_batch(reads, off)
        assert np.array_equal(coff, g["cord_off"]) and np.array_equal(cs, g["cords_str"]) and np.array_equal(ce, g["cords_end"])
    f.close()


def test_gpu_two_wave_middle_class(case_inputs, monkeypatch):
    """Force every job through the two-wave form of the middle class (the round-0 default at human scale): same cords.
    (Found by bench.py's parity check against the reference: the workgroup radix sort assumed at least 256 threads.)"""
    from linear_amd import Filter
    monkeypatch.setenv("LNR_MID_CAP", "64")
    monkeypatch.setenv("LNR_MID_WAVES", "2")
    f = Filter(device=0)
    for name, T in (("rep", 1), ("edge", 3), ("ont", 4)):
        refs, reads, off = case_inputs(name)
        g = np.load(os.path.join(GOLD, f"{name}_T{T}.npz"))
        f.build_index(refs, T)
        coff, cs, ce = f.filter_batch(reads, off)
        assert np.array_equal(coff, g["cord_off"]) and np.array_equal(cs, g["cords_str"]) and np.array_equal(ce, g["cords_end"])
    f.close()


def test_gpu_heavy_path(case_inputs, monkeypatch):
    """Force every job through the 16-wave kernel: same cords."""
    from linear_amd import Filter
    monkeypatch.setenv("LNR_HEAVY_CAP", "64")
    f = Filter(device=0)
    for name, T in (("rep", 1), ("edge", 3)):
        refs, reads, off = case_inputs(name)
        g = np.load(os.path.join(GOLD, f"{name}_T{T}.npz"))
        f.build_index(refs, T)
        coff, cs, ce = f.filter_batch(reads, off)
        assert np.array_equal(coff, g["cord_off"]) and np.array_equal(cs, g["cords_str"]) and np.array_equal(ce, g["cords_end"])
    f.close()


def test_gpu_index_receiver_path(case_inputs):
    """Multi-GPU receiver path on one device: a second context allocates the index from the owner's metadata
    (lnr_index_alloc), receives the four device blobs in place (what the RCCL broadcast does; here a device-to-device
    copy through the same zero-copy tensor views linear_amd.dist uses), adopts it (lnr_index_adopt rebuilds the derived
    bucket bitmap) and must then produce the reference's cords."""
    import torch
    from linear_amd import Filter
    from linear_amd.dist import blob_tensor
    for name, T in (("ont", 4), ("edge", 3)):
        refs, reads, off = case_inputs(name)
        g = np.load(os.path.join(GOLD, f"{name}_T{T}.npz"))
        owner = Filter(device=0)
        owner.build_index(refs, T)
        recv = Filter(device=0)
        recv.index_alloc_from(owner.index_info_vec(), owner.seq_len())
        src, dst = owner.index_blobs(), recv.index_blobs()
        assert len(src) == len(dst) == 4
        for (ps, bs), (pd, bd) in zip(src, dst):
            assert bs == bd
            if bs:
                blob_tensor(pd, bd, "cuda:0").copy_(blob_tensor(ps, bs, "cuda:0"))
        torch.cuda.synchronize()
        recv.index_adopt()
        owner.close()
        coff, cs, ce = recv.filter_batch(reads, off)
        assert np.array_equal(coff, g["cord_off"]) and np.array_equal(cs, g["cords_str"]) and np.array_equal(ce, g["cords_end"])
        recv.close()


def test_gpu_every_read_in_the_remap_round(oracle_lib):
    """A small batch in which every read goes through the re-map round (found by tools/stress_parity.py: the early tail-B
    launch then has nothing to do and must be skipped, not launched with an empty grid)."""
    from linear_amd import Filter, synth
    refs = [synth.repeat_ref(400_000, 2024), synth.add_n_runs(synth.random_ref(250_000, 2025), 2026, lead=1000), synth.repeat_ref(150_000, 2027, n_families=4)]
    f = Filter(device=0)
    f.build_index(refs, 2)
    o = oracle_lib.Checker("oracle", refs, 2)
    for nreads, L, err, seed in ((12, 700, 0.15, 5), (64, 700, 0.1, 6), (8, 3000, 0.15, 7)):
        reads, off, _ = synth.sample_reads(refs, nreads, L, err, seed, "random")
        coff, cs, ce = f.filter_batch(reads, off)
        ooff, ocs, oce, _ = o.map_batch(reads, off, threads=2)
        assert np.array_equal(coff, ooff) and np.array_equal(cs, ocs) and np.array_equal(ce, oce)
    f.close()


def test_gpu_traceback_of_long_score_arrays_with_rejected_chains(oracle_lib):
    """Repeat-rich reference, 1500 reads: reads with more than 1024 anchors in the chaining DP take the chunk-bound search
    of the anchor traceback, and many of their walks are rejected and put scores back (found by tools/stress_parity.py,
    seed 4242 configuration 110: bounds tightened while a walk's elements were provisionally deleted went stale)."""
    from linear_amd import Filter, synth
    s = 229311090
    refs = [synth.repeat_ref(843667, s, n_families=6)]
    reads, off, _ = synth.sample_reads(refs, 1500, 9000, 0.03, s + 7, "random", len_jitter=0.0)
    f = Filter(device=0)
    f.build_index(refs, 3)
    coff, cs, ce = f.filter_batch(reads, off)
    f.close()
    o = oracle_lib.Checker("oracle", refs, 3)
    ooff, ocs, oce, _ = o.map_batch(reads, off, threads=8)
    assert np.array_equal(coff, ooff) and np.array_equal(cs, ocs) and np.array_equal(ce, oce)


def test_gpu_reads_that_begin_inside_an_n_run(oracle_lib):
    """hashInit's N-skip at the read start is found by a whole wave from the N bits (k_prep): reads whose first bases are N
    for 1 .. 4000 positions -- around the 21-base window, the 16-position groups and the 992-position steps of the search --
    reads that are N throughout, N runs further inside, and a short read that takes the literal walk."""
    from linear_amd import Filter, synth
    ref = synth.random_ref(300_000, 4711)
    refs = [ref]
    L = 6000
    lead = [0, 1, 5, 15, 16, 17, 20, 21, 22, 31, 32, 33, 63, 64, 65, 500, 991, 992, 993, 1008, 1500, 1983, 1984, 1985, 4000, 5979, 5980, 5999, L]
    rng = np.random.default_rng(99)
    parts, off = [], [0]
    for k, n_lead in enumerate(lead):
        p = int(rng.integers(1000, 290_000 - L))
        r = ref[p:p + L].copy()
        r[:n_lead] = 4
        if k % 3 == 1 and n_lead + 40 < L:          # a second run a little further in: the first window of 21 clean bases may lie between them
            r[n_lead + 25:n_lead + 40] = 4
        if k % 3 == 2 and n_lead + 20 < L:          # ... or only 20 clean bases between two runs
            r[n_lead + 20:n_lead + 30] = 4
        parts.append(r); off.append(off[-1] + L)
    short = ref[5000:5300].copy(); short[:37] = 4   # 2 * packed groups < 64: the literal walk
    parts.append(short); off.append(off[-1] + short.size)
    reads = np.concatenate(parts); off = np.asarray(off, dtype=np.uint64)
    f = Filter(device=0)
    f.build_index(refs, 1)
    coff, cs, ce = f.filter_batch(reads, off)
    f.close()
    o = oracle_lib.Checker("oracle", refs, 1)
    ooff, ocs, oce, _ = o.map_batch(reads, off, threads=2)
    assert np.array_equal(coff, ooff) and np.array_equal(cs, ocs) and np.array_equal(ce, oce)
    assert cs.size > 0


def test_gpu_submit_wait_pipeline_and_pinned_input(case_inputs):
    """lnr_filter_submit / lnr_filter_wait with three batches in flight (a wait hands out batch k while it computes batch k + 1 and the upload of
    k + 2 runs), from pinned memory (lnr_host_alloc: direct DMA) and from a pageable numpy array (staged): the cords of each batch equal the
    reference's, in submission order; a fourth submit is refused; the gap stream cannot be set while batches are in flight."""
    from linear_amd import Filter, LnrError
    refs, reads, off = case_inputs("ont")
    g = np.load(os.path.join(GOLD, "ont_T4.npz"))
    f = Filter(device=0)
    f.build_index(refs, 4)
    n = off.size - 1
    h = n // 2
    pin = f.host_alloc(int(off[h]))
    pin[:] = reads[: int(off[h])]
    o1 = np.ascontiguousarray(off[: h + 1])
    r2 = np.ascontiguousarray(reads[int(off[h]):])
    o2 = np.ascontiguousarray(off[h:] - off[h])
    for _ in range(2):
        f.filter_submit(pin, o1)
        f.filter_submit(r2, o2)
        f.filter_submit(pin, o1)
        with pytest.raises(LnrError):
            f.filter_submit(r2, o2)          # a fourth batch in flight is refused
        with pytest.raises(LnrError):
            f.gap_stream(0)                  # (the next batch may have been computed already)
        c1 = f.filter_wait()
        f.filter_submit(r2, o2)              # the pipeline stays three deep: 1 handed out, 2 computed ahead, 3 + 4 uploading
        c2 = f.filter_wait()
        c3 = f.filter_wait()
        c4 = f.filter_wait()
        with pytest.raises(LnrError):
            f.filter_wait()                  # nothing in flight any more
        coff, cs, ce = g["cord_off"], g["cords_str"], g["cords_end"]
        k = int(coff[h])
        for c in (c1, c3):
            assert np.array_equal(c[0], coff[: h + 1]) and np.array_equal(c[1], cs[:k]) and np.array_equal(c[2], ce[:k])
        for c in (c2, c4):
            assert np.array_equal(c[0], coff[h:] - coff[h]) and np.array_equal(c[1], cs[k:]) and np.array_equal(c[2], ce[k:])
    with pytest.raises(LnrError):
        f.filter_wait()
    bad = np.array([0, 10, 5], dtype=np.uint64)
    with pytest.raises(LnrError) as e:
        f.filter_batch(reads[:10], bad)
    assert e.value.status == -1               # offsets not monotone -> LNR_ERR_ARG, not an allocation failure
    f.close()


def test_gpu_base_values_above_4_read_as_n(oracle_lib):
    """Bytes above 4 in the reference or in a read are taken as N (header contract), never fed raw into the hash arithmetic."""
    from linear_amd import Filter, synth
    ref = synth.random_ref(400_000, 31)
    dirty = ref.copy()
    rng = np.random.default_rng(32)
    pos = rng.integers(1000, 399_000, size=40)
    dirty[pos] = rng.integers(5, 256, size=40).astype(np.uint8)
    clean = dirty.copy(); clean[clean > 4] = 4
    reads, off, _ = synth.sample_reads([clean], 40, 5000, 0.05, 33, "random")
    rd = reads.copy(); rd[::997] = 200
    rc = rd.copy(); rc[rc > 4] = 4
    f = Filter(device=0)
    f.build_index([dirty], 2)
    o = oracle_lib.Checker("oracle", [clean], 2)
    dir_, hs, _, _ = f.index_export()
    assert np.array_equal(dir_, o.dir()) and np.array_equal(hs, o.hs())
    coff, cs, ce = f.filter_batch(rd, off)
    ooff, ocs, oce, _ = o.map_batch(rc, off, threads=4)
    assert np.array_equal(coff, ooff) and np.array_equal(cs, ocs) and np.array_equal(ce, oce)
    f.close()


@pytest.mark.parametrize("name,T", [("edge", 3), ("rep", 8), ("scale", 4)])
def test_gpu_cords_to_sam_and_apf_text(flt, case_inputs, name, T):
    """north_star's parity surface end to end: reads -> HIP path -> cords -> lnr_writer -> the SAM and APF bytes the reference's
    own writer functions produced for its own cords (tests/golden, made by tools/make_golden.py through oracle/_ref)."""
    from linear_amd.api import Writer
    refs, reads, off = case_inputs(name)
    g = np.load(os.path.join(GOLD, f"{name}_T{T}.npz"))
    flt.build_index(refs, T)
    coff, cs, ce = flt.filter_batch(reads, off)
    n = off.size - 1
    rid, gid = cases.text_ids(n, len(refs))
    w = Writer(gid, [r.size for r in refs])
    rl = np.diff(off.astype(np.int64)).astype(np.uint64)
    assert w.sam_header(cases.CMD_LINE) + w.format(coff, cs, ce, rl, rid, "sam") == g["sam"].tobytes()
    assert w.format(coff, cs, ce, rl, rid, "apf") == g["apf"].tobytes()
    w.close()


def test_gpu_linear_filter_cli_end_to_end(case_inputs, tmp_path):
    """The front-end binary (linear_amd/linear_filter: reader -> submit / wait -> writer, all through the C ABI) on FASTA files,
    two reads blocks in flight: its .sam and .apf equal what the reference's writer functions printed for the reference's cords
    (the @PG line carries the command line and is compared apart; APF blank lines depend on the block size)."""
    import subprocess
    from linear_amd import build as lb
    lb.build()
    name, T = "edge", 3
    refs, reads, off = case_inputs(name)
    g = np.load(os.path.join(GOLD, f"{name}_T{T}.npz"))
    n = off.size - 1
    rid, gid = cases.text_ids(n, len(refs))
    abc = np.frombuffer(b"ACGTN", np.uint8)
    with open(tmp_path / "ref.fa", "wb") as f:
        for k, r in enumerate(refs):
            f.write(b">" + gid[k].encode() + b" some description\n")
            t = abc[r].tobytes()
            f.write(b"\n".join(t[i:i + 80] for i in range(0, len(t), 80)) + b"\n")
    with open(tmp_path / "reads.fa", "wb") as f:
        for i in range(n):
            f.write(b">" + rid[i].encode() + b"\n" + abc[reads[int(off[i]):int(off[i + 1])]].tobytes() + b"\n")
    for block in (1000, 17):
        p = subprocess.run([lb.CLI, "filter", str(tmp_path / "reads.fa"), str(tmp_path / "ref.fa"), "-t", str(T), "-g", "0", "-o", str(tmp_path / "out"), "-ot", "3", "--block-reads", str(block)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-1000:]
        sam = open(tmp_path / "out.sam", "rb").read().split(b"\n")
        want = g["sam"].tobytes().split(b"\n")
        assert [l for l in sam if not l.startswith(b"@PG")] == [l for l in want if not l.startswith(b"@PG")]
        assert [l for l in sam if l.startswith(b"@PG")][0] == b"@PG\tID:M1-3\tPN:Linear\tCL:"     # (empty in the real program: base.cpp:64-72)
        apf = [l for l in open(tmp_path / "out.apf", "rb").read().split(b"\n") if l]
        assert apf == [l for l in g["apf"].tobytes().split(b"\n") if l]


def test_gpu_apx_gaps_export(oracle_lib):
    """apxMap's second output (the uncovered stretches the reference's gap re-mapper starts from) through lnr_last_gaps: the
    oracle's apx_gaps, read by read -- junk and chimeric reads have gaps, clean reads none."""
    from linear_amd import Filter, synth
    refs = [synth.repeat_ref(500_000, 91), synth.random_ref(300_000, 92)]
    rng = np.random.default_rng(93)
    lst = []
    for k in range(24):
        a = synth.mutate(refs[0][30_000 + 7000 * k: 34_000 + 7000 * k], 0.08, rng)
        junk = rng.integers(0, 4, size=1500 + 200 * (k % 5), dtype=np.uint8)
        b = synth.mutate(refs[1][50_000 + 3000 * k: 53_500 + 3000 * k], 0.08, rng)
        lst.append(np.concatenate([a, junk, b]) if k % 2 else a)
    lst.append(rng.integers(0, 4, size=6000, dtype=np.uint8))
    reads, off = synth.pack_reads(lst)
    f = Filter(device=0)
    f.build_index(refs, 2)
    f.filter_batch(reads, off)
    goff, gaps = f.last_gaps()
    o = oracle_lib.Checker("oracle", refs, 2)
    tot = 0
    for i in range(off.size - 1):
        o.map_read(reads[int(off[i]):int(off[i + 1])])
        want = o.gaps()
        got = gaps[int(goff[i]):int(goff[i + 1])]
        assert np.array_equal(got, want), f"read {i}"
        tot += want.shape[0]
    assert tot >= 10
    f.close()


# ---- HIndex (-i 2)
PARAMS_I2 = [(n, T) for n, (_, Ts) in cases.CASES_I2.items() for T in Ts]


@pytest.mark.parametrize("name,T", PARAMS_I2)
def test_gpu_hindex_matches_golden(case_inputs, name, T):
    """-i 2 through the C ABI against the reference's own output: ysa byte for byte, raw anchors of the stage reads (both
    sampling steps) and the cords of every read."""
    from linear_amd import Filter
    refs, reads, off = case_inputs(name)
    g = np.load(os.path.join(GOLD, f"{name}_i2_T{T}.npz"))
    assert cases.input_digest(refs, reads, off) == str(g["digest"])
    f = Filter(device=0, index_type=2)
    info = f.build_index(refs, T)
    assert info.hs_len == int(g["ysa_len"])
    _, ysa, _, _ = f.index_export()
    assert np.array_equal(ysa[:4096], g["ysa_head"])
    assert cases.sha(ysa) == str(g["ysa_sha"]), "ysa differs from the reference"
    aoff, anc = f.seed_lookup_batch(reads, off)
    for k, i in enumerate(g["stage_reads"]):
        assert np.array_equal(anc[int(aoff[i]):int(aoff[i + 1])], g[f"st{k}_raw"]), f"raw anchors of read {i}"
    coff, cs, ce = f.filter_batch(reads, off)
    assert np.array_equal(coff, g["cord_off"])
    assert np.array_equal(cs, g["cords_str"])
    assert np.array_equal(ce, g["cords_end"])
    f.close()


def test_gpu_hindex_matches_oracle_fresh_seed(oracle_lib):
    """-i 2 on inputs without a golden (N runs, three sequences, reads that go through the re-map round), the oracle as the checker;
    and the receiver path: a second context adopts the broadcast blobs and derives its lookup tables from ysa."""
    import torch
    from linear_amd import Filter, synth
    from linear_amd.dist import blob_tensor
    refs = [synth.add_n_runs(synth.repeat_ref(500_000, 41), 42, n_runs=3, max_run=900, lead=2500), synth.random_ref(150_000, 43), synth.repeat_ref(60_000, 44)]
    reads, off, _ = synth.sample_reads(refs, 120, 7000, 0.1, 45, "random", len_jitter=0.6)
    o = oracle_lib.Checker("oracle", refs, 3, index_type=2)
    f = Filter(device=0, index_type=2)
    f.build_index(refs, 3)
    _, ysa, _, _ = f.index_export()
    assert np.array_equal(ysa, o.ysa())
    want = [o.map_read(reads[int(off[i]):int(off[i + 1])]) for i in range(off.size - 1)]
    for flt in (f, None):
        if flt is None:
            flt = Filter(device=0, index_type=2)
            flt.index_alloc_from(f.index_info_vec(), f.seq_len())
            for (ps, bs), (pd, bd) in zip(f.index_blobs(), flt.index_blobs()):
                assert bs == bd
                if bs:
                    blob_tensor(pd, bd, "cuda:0").copy_(blob_tensor(ps, bs, "cuda:0"))
            torch.cuda.synchronize()
            flt.index_adopt()
        coff, cs, ce = flt.filter_batch(reads, off)
        for i in range(off.size - 1):
            a, b = int(coff[i]), int(coff[i + 1])
            assert np.array_equal(cs[a:b], want[i][0]) and np.array_equal(ce[a:b], want[i][1]), f"read {i}"
        if flt is not f:
            flt.close()
    f.close()


def test_gpu_batch_rerun_on_per_read_overflow(case_inputs, monkeypatch):
    """A read that outgrows its cord capacity does not cost the batch: the batch is run again with 4x, then 16x the
    per-read capacities.  LNR_CAP_SHRINK makes the first attempt(s) overflow; the result must still be the reference's, and with
    capacities that stay too small the call fails loudly instead of returning short lists."""
    from linear_amd import Filter
    from linear_amd.api import LnrError
    refs, reads, off = case_inputs("ont")
    g = np.load(os.path.join(GOLD, "ont_T1.npz"))
    monkeypatch.setenv("LNR_CAP_SHRINK", "64")
    f = Filter(device=0)
    f.build_index(refs, 1)
    coff, cs, ce = f.filter_batch(reads, off)
    assert np.array_equal(coff, g["cord_off"]) and np.array_equal(cs, g["cords_str"]) and np.array_equal(ce, g["cords_end"])
    f.close()
    monkeypatch.setenv("LNR_CAP_SHRINK", "4096")
    f = Filter(device=0)
    f.build_index(refs, 1)
    with pytest.raises(LnrError) as e:
        f.filter_batch(reads, off)
    assert e.value.status == -8 and "overflow" in str(e.value)
    f.close()


@pytest.mark.parametrize("gap_len,dup", [(0, 0), (50, 1)])
def test_gpu_seqan_side_binding_on_the_gpu(case_inputs, tmp_path, gap_len, dup):
    """integration/gpu_filter.h -- the file INTEGRATION.md tells a maintainer of the reference to add: SeqAn StringSet<String<Dna5>> in, StringSet<String<
    uint64_t>> out -- executed on the GPU: tests/_build/gpufilter_driver is compiled against the reference's vendored SeqAn headers by
    __graft_entry__.build() where the reference tree is (here it is only run).  Two filterBlock calls on one binding; cords equal the goldens."""
    import struct
    import subprocess
    exe = os.path.join(os.path.dirname(__file__), "_build", "gpufilter_driver")
    if not os.path.exists(exe):
        if os.path.exists("/root/reference/seqan/include/seqan/sequence.h"):
            import __graft_entry__ as ge
            ge.build()
        else:
            pytest.fail("tests/_build/gpufilter_driver is missing: __graft_entry__.build() makes it where /root/reference is present")
    refs, reads, off = case_inputs("edge")
    n = off.size - 1
    with open(tmp_path / "case.bin", "wb") as f:
        f.write(struct.pack("<I", len(refs)))
        for r in refs:
            f.write(struct.pack("<Q", r.size)); f.write(np.ascontiguousarray(r, np.uint8).tobytes())
        f.write(struct.pack("<I", n))
        for i in range(n):
            rd = reads[int(off[i]):int(off[i + 1])]
            f.write(struct.pack("<Q", rd.size)); f.write(np.ascontiguousarray(rd, np.uint8).tobytes())
    T = 1 if gap_len else 3
    p = subprocess.run([exe, str(tmp_path / "case.bin"), str(tmp_path / "out.bin"), str(T), str(gap_len), str(dup)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, (p.returncode, p.stderr.decode()[-500:])
    g = np.load(os.path.join(GOLD, "edge_g50_T1.npz" if gap_len else "edge_T3.npz"))
    sfx = f"_dup{dup}" if gap_len else ""
    coff, cs, ce = g["cord_off" + sfx], g["cords_str" + sfx], g["cords_end" + sfx]
    raw = open(tmp_path / "out.bin", "rb").read()
    assert struct.unpack_from("<I", raw, 0)[0] == n
    pos = 4
    for i in range(n):
        k = struct.unpack_from("<Q", raw, pos)[0]; pos += 8
        a = np.frombuffer(raw, np.uint64, k, pos); pos += 8 * k
        b = np.frombuffer(raw, np.uint64, k, pos); pos += 8 * k
        assert k == int(coff[i + 1] - coff[i]) and np.array_equal(a, cs[int(coff[i]):int(coff[i + 1])]) and np.array_equal(b, ce[int(coff[i]):int(coff[i + 1])]), i


def test_gpu_two_pageable_submits_back_to_back(flt, case_inputs):
    """lnr_filter_submit twice in a row from PAGEABLE memory (the documented two-in-flight pattern): both batches go through the context's two
    pinned staging buffers; the second submit must not overwrite a chunk the first one's DMA has not read yet (the "recorded"
    state of the staging events was call-local).  Batches of different content; both results must equal the one-by-one results."""
    refs, reads, off = case_inputs("ont")
    flt.build_index(refs, 4)
    n = off.size - 1
    h = n // 2
    b0 = (np.ascontiguousarray(reads[: int(off[h])]), np.ascontiguousarray(off[: h + 1]))
    b1 = (np.ascontiguousarray(reads[int(off[h]):]), np.ascontiguousarray((off[h:] - off[h]).astype(np.uint64)))
    want0, want1 = flt.filter_batch(*b0), flt.filter_batch(*b1)
    for _ in range(3):
        flt.filter_submit(*b0)
        flt.filter_submit(*b1)
        got0 = flt.filter_wait()
        got1 = flt.filter_wait()
        for w, g_ in ((want0, got0), (want1, got1)):
            assert all(np.array_equal(a, b) for a, b in zip(w, g_))


# ---- the gap re-mapper (-g > 0): mapGaps + reformCords on the GPU (k_gap)
@pytest.mark.parametrize("name", ["ont", "edge", "ccs_sv", "rep", "chim"])
def test_gpu_gap_path_matches_golden(case_inputs, name):
    """lnr_opts.gap_len = 50 [dup = 1] through the C ABI against the cords the real reference produced with -g 50 [-dup 1] on the case as one
    read stream in file order (`-t 1`): the first read that goes through mapExtend / mapExtends changes what every later read sees
    (thd_cts_major_limit, lnr_gap_stream).  Whole case in one batch, and as three batches on one context (the state is carried)."""
    from linear_amd import Filter
    refs, reads, off = case_inputs(name)
    g = np.load(os.path.join(GOLD, f"{name}_g50_T1.npz"))
    assert cases.input_digest(refs, reads, off) == str(g["digest"])
    for dup in (0, 1):
        f = Filter(device=0, gap_len=50, dup=dup)
        f.build_index(refs, 1)
        coff, cs, ce = f.filter_batch(reads, off)
        assert f.stats()["gap_ms"] > 0
        assert f.gap_stream() == int(g[f"ext_out_dup{dup}"])
        assert np.array_equal(coff, g[f"cord_off_dup{dup}"]), f"dup {dup}"
        assert np.array_equal(cs, g[f"cords_str_dup{dup}"]), f"dup {dup}"
        assert np.array_equal(ce, g[f"cords_end_dup{dup}"]), f"dup {dup}"
        if dup == 0:
            n = off.size - 1
            cuts = [0, 2, n // 3, n]                      # a new stream, three batches: the state crosses the batch borders
            assert f.gap_stream(0) == 0
            parts = []
            for a, b in zip(cuts[:-1], cuts[1:]):
                o2 = (off[a:b + 1] - off[a]).astype(np.uint64)
                parts.append(f.filter_batch(reads[int(off[a]):int(off[b])], o2))
            assert np.array_equal(np.concatenate([p[1] for p in parts]), cs) and np.array_equal(np.concatenate([p[2] for p in parts]), ce)
        f.close()


def test_gpu_gap_path_matches_oracle_on_planted_svs(oracle_lib):
    """reads with planted insertions, deletions, duplications, inversions and translocated stretches on a repeat-rich and an N-run
    reference, several -g values with and without -dup, index layouts -t 1 and 3; the oracle (pinned to the reference on the same kind
    of input by tests/test_oracle_golden.py) as the checker."""
    from linear_amd import Filter, synth
    from tests.test_gap_shim_cpu import sv_reads
    refs = [synth.repeat_ref(300_000, 61), synth.add_n_runs(synth.random_ref(200_000, 62), 63, n_runs=2, max_run=600)]
    rl = sv_reads(refs, 96, 2028)
    off = np.zeros(len(rl) + 1, np.uint64)
    off[1:] = np.cumsum([r.size for r in rl])
    reads = np.concatenate(rl)
    for T in (1, 3):
        o = oracle_lib.Checker("oracle", refs, T)
        for gap_len, dup in ((50, 0), (50, 1), (1, 0), (5, 1), (200, 0)):
            f = Filter(device=0, gap_len=gap_len, dup=dup)
            f.build_index(refs, T)
            coff, cs, ce = f.filter_batch(reads, off)
            f.close()
            ooff, ocs, oce, _ = o.map_batch(reads, off, threads=8, gap_len=gap_len, dup=dup)        # (file-order stream semantics)
            assert np.array_equal(coff, ooff), (T, gap_len, dup)
            bad = [i for i in range(len(rl)) if not (np.array_equal(cs[int(coff[i]):int(coff[i + 1])], ocs[int(ooff[i]):int(ooff[i + 1])]) and
                                                      np.array_equal(ce[int(coff[i]):int(coff[i + 1])], oce[int(ooff[i]):int(ooff[i + 1])]))]
            assert not bad, (T, gap_len, dup, bad[:8])
            poff, pcs, pce, _ = o.map_batch(reads, off, threads=8)
            changed = sum(not np.array_equal(cs[int(coff[i]):int(coff[i + 1])], pcs[int(poff[i]):int(poff[i + 1])]) for i in range(len(rl)))
            assert changed > len(rl) // 2, "the gap path should change the cords of most of these reads"
        o.close()


@pytest.mark.parametrize("name", list(cases.CASES_CLI))
def test_gpu_linear_filter_cli_equals_the_real_program(name, tmp_path):
    """The product's `linear_filter` binary (reader -> HIP path -> writer, all through the C ABI) on the FASTA files of a case against the
    bytes the REAL `linear filter` program wrote for the same files at -t 1 (tests/golden/cli_<case>.npz, tools/make_cli_golden.py:
    oracle/_ref/linear): -g 0, -g 50, -g 50 -dup 1 and no -g at all (= the reference's default, gaps of 50).  .sam byte for byte incl. the
    header; .apf line for line (blank lines follow the reference's adaptive block size).  One read is excluded: read_11 of
    `edge` at -g > 0, whose result in the reference depends on heap contents (tests/test_cli_golden_cpu.py)."""
    import subprocess
    from linear_amd import build as lb
    from tests.test_cli_golden_cpu import UB_READS, sam_by_read, apf_by_read
    lb.build()
    refs, reads, off = cases.CASES_CLI[name]()
    g = np.load(os.path.join(GOLD, f"cli_{name}.npz"))
    rp, gp, _, _ = cases.write_fasta_case(tmp_path, refs, reads, off)
    for mode, flags in cases.CLI_MODES.items():
        for block in (["--block-reads", "23", "--gpus", "2", "--devices", "0,0"] if mode == "g50dup1" else []), :      # (two contexts on the one GPU: two calculators,
                                                                                                                             # the index moved by lnr_index_broadcast, the gap stream across them)
            p = subprocess.run([lb.CLI, "filter", rp, gp, "-t", "1", "-ot", "3", "-o", str(tmp_path / "out")] + flags + list(block), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
            assert p.returncode == 0, p.stderr.decode()[-1000:]
            sam, want = open(tmp_path / "out.sam", "rb").read(), g[f"sam_{mode}"].tobytes()
            skip = UB_READS.get((name, mode), set())
            if not skip:
                assert sam == want, mode
            head, recs = sam_by_read(sam)
            whead, wrecs = sam_by_read(want)
            assert head == whead and list(recs) == list(wrecs)
            for k in wrecs:
                assert k in skip or recs[k] == wrecs[k], (mode, k)
            apf, wapf = apf_by_read(open(tmp_path / "out.apf", "rb").read()), apf_by_read(g[f"apf_{mode}"].tobytes())
            assert list(apf) == list(wapf)
            for k in wapf:
                assert k in skip or apf[k] == wapf[k], (mode, k)


def test_gpu_linear_filter_cli_with_gap_path(case_inputs, tmp_path):
    """`linear_filter filter ... -g 50 -dup 1` end to end (the reference's default mode is -g on): the .sam / .apf are the writer's text of
    the cords the real reference produced with -g 50 -dup 1 (tests/golden/edge_g50_T1.npz) -- i.e. the options reach the kernels and the
    gap path's cords go through the same output surface; and without -g the front-end runs with the reference's default (-g 1 = 50)."""
    import subprocess
    from linear_amd import build as lb
    from linear_amd.api import Writer
    lb.build()
    refs, reads, off = case_inputs("edge")
    g = np.load(os.path.join(GOLD, "edge_g50_T1.npz"))
    n = off.size - 1
    rid, gid = cases.text_ids(n, len(refs))
    abc = np.frombuffer(b"ACGTN", np.uint8)
    with open(tmp_path / "ref.fa", "wb") as f:
        for k, r in enumerate(refs):
            t = abc[r].tobytes()
            f.write(b">" + gid[k].encode() + b"\n" + b"\n".join(t[i:i + 70] for i in range(0, len(t), 70)) + b"\n")
    with open(tmp_path / "reads.fa", "wb") as f:
        for i in range(n):
            f.write(b">" + rid[i].encode() + b"\n" + abc[reads[int(off[i]):int(off[i + 1])]].tobytes() + b"\n")
    w = Writer(gid, [r.size for r in refs])
    rl = np.diff(off.astype(np.int64)).astype(np.uint64)
    for flags, dup in ((["-g", "50", "-dup", "1"], 1), ([], 0)):
        p = subprocess.run([lb.CLI, "filter", str(tmp_path / "reads.fa"), str(tmp_path / "ref.fa"), "-t", "1", "-o", str(tmp_path / "out"), "-ot", "3"] + flags,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-1000:]
        coff, cs, ce = g[f"cord_off_dup{dup}"], g[f"cords_str_dup{dup}"], g[f"cords_end_dup{dup}"]
        sam = [l for l in open(tmp_path / "out.sam", "rb").read().split(b"\n") if not l.startswith(b"@PG")]
        assert sam == [l for l in (w.sam_header("x") + w.format(coff, cs, ce, rl, rid, "sam")).split(b"\n") if not l.startswith(b"@PG")], flags
        apf = [l for l in open(tmp_path / "out.apf", "rb").read().split(b"\n") if l]
        assert apf == [l for l in w.format(coff, cs, ce, rl, rid, "apf").split(b"\n") if l], flags
    w.close()
