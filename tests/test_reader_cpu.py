"""CPU: the input-side reader (linear_amd/csrc/lnr_reader.cpp behind lnr_reader_*) against the reference's own
reader -- SeqAn readRecords, run through oracle/_ref by tools/make_golden.py and stored in tests/golden/reader.npz; compared
live as well where /root/reference is present.  Same records, same Dna5 ordinals, same ids, whatever the block size."""
import os

import numpy as np
import pytest

from tests import reader_cases

GOLD = os.path.join(os.path.dirname(__file__), "golden", "reader.npz")


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    from linear_amd import build as lb
    lb.build()
    return reader_cases.write_cases(str(tmp_path_factory.mktemp("reader")))


def read_all(path, dst_cap, max_reads):
    from linear_amd.api import Reader
    r = Reader(path)
    dst = np.zeros(dst_cap, np.uint8)
    bases, lens, ids, blocks = [], [], [], 0
    while True:
        n, off, i = r.next(dst, max_reads)
        if n == 0:
            break
        blocks += 1
        assert n <= max_reads and int(off[n]) <= dst_cap
        bases.append(dst[: int(off[n])].copy())
        lens += np.diff(off.astype(np.int64)).tolist()
        ids += i
    r.close()
    off = np.zeros(len(lens) + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    return (np.concatenate(bases) if bases else np.zeros(0, np.uint8)), off, ids, blocks


@pytest.mark.parametrize("dst_cap,max_reads", [(1 << 20, 1000), (12000, 3), (20000, 1), (13000, 1000)])
def test_reader_equals_reference_reader(files, dst_cap, max_reads):
    paths, seqs = files
    g = np.load(GOLD)
    for name, path in paths.items():
        b, off, ids, blocks = read_all(path, dst_cap, max_reads)
        assert np.array_equal(off, g[name + ":off"]), name
        assert np.array_equal(b, g[name + ":bases"]), name
        assert ids == [str(x) for x in g[name + ":ids"]], name
        assert off.size - 1 == len(seqs)
        if dst_cap < 30000:
            assert blocks > 1            # block boundaries (incl. a record carried over to the next block) do not change the records


@pytest.mark.skipif(not os.path.exists("/root/reference/src/pmpfinder.cpp"), reason="reference tree not present")
def test_reader_equals_live_reference_reader(files, oracle_lib):
    paths, _ = files
    for name, path in paths.items():
        rb, ro, rids = oracle_lib.ref_read_file(path)
        b, off, ids, _ = read_all(path, 1 << 20, 64)
        assert np.array_equal(b, rb) and np.array_equal(off, ro) and ids == rids, name


def test_reader_limits_and_superset(tmp_path):
    """A record longer than the block is LNR_ERR_LIMIT; characters outside ACGTN (where the reference's reader throws and the
    Mapper drops the block, mapper.cpp:917-933) decode by SeqAn's char -> Dna5 table: U/u = T, everything else N."""
    from linear_amd import LnrError
    from linear_amd.api import Reader
    p = tmp_path / "x.fa"
    p.write_text(">a\nACGUacguRYKM-*.\n>b\n" + "A" * 500 + "\n")
    r = Reader(str(p))
    dst = np.zeros(100, np.uint8)
    n, off, ids = r.next(dst, 10)
    assert n == 1 and ids == ["a"] and dst[:15].tolist() == [0, 1, 2, 3, 0, 1, 2, 3, 4, 4, 4, 4, 4, 4, 4]
    with pytest.raises(LnrError) as e:
        r.next(dst, 10)
    assert e.value.status == -6
    r.close()
    with pytest.raises(LnrError):
        Reader(str(tmp_path / "missing.fa"))


@pytest.mark.parametrize("fmt", ["fasta", "fastq", "fastq_multiline"])
def test_parallel_and_serial_parsers_agree(tmp_path, fmt, monkeypatch):
    """plain files go through the mapped, multi-threaded parser; the byte-wise serial parser (what gzip input uses, pinned to SeqAn's reader above)
    must give the same records, ordinals and ids for any block size and thread count -- multi-line records, CRLF, blanks, lower case, IUPAC codes,
    empty records, '>' and '@' inside header and quality lines; multi-line FASTQ makes the parallel path hand over to the serial one mid-file."""
    from linear_amd import build as lb
    lb.build()
    rng = np.random.default_rng(17)
    abc = np.frombuffer(b"ACGTacgtNnRYKMU", np.uint8)
    path = tmp_path / ("x." + fmt)
    with open(path, "wb") as f:
        for i in range(1500):
            L = int(rng.integers(0, 900)) if i % 97 else 0
            s = abc[rng.integers(0, abc.size, L)].tobytes()
            eol = b"\r\n" if i % 5 == 0 else b"\n"
            if fmt == "fasta":
                w = int(rng.integers(20, 200))
                body = eol.join(s[k:k + w] for k in range(0, max(L, 1), w)) if i % 3 else s
                if i % 11 == 0:
                    body = body.replace(b"A", b"A ", 1)
                f.write(b">rd%d > x @ y" % i + eol + body + eol + (eol if i % 7 == 0 else b""))
            else:
                q = bytes(rng.integers(33, 74, L, dtype=np.uint8).tolist())        # '@' (64) and '>' (62) occur in qualities
                if fmt == "fastq_multiline" and i > 700 and L > 50:
                    f.write(b"@rd%d" % i + eol + s[:30] + eol + s[30:] + eol + b"+" + eol + q[:40] + eol + q[40:] + eol)
                else:
                    f.write(b"@rd%d desc" % i + eol + s + eol + b"+" + (b"rd%d" % i if i % 2 else b"") + eol + q + eol)
    monkeypatch.setenv("LNR_READER_SERIAL", "1")
    want = read_all(str(path), 1 << 22, 100000)
    monkeypatch.delenv("LNR_READER_SERIAL")
    assert want[1].size - 1 == 1500
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("LNR_READER_THREADS", threads)
        for cap, mr in ((1 << 22, 100000), (5000, 7), (1000, 1), (40000, 64)):
            got = read_all(str(path), cap, mr)
            assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0]) and got[2] == want[2], (threads, cap, mr)
