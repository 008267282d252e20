"""ctypes bindings for the two CPU checkers (TEST INFRASTRUCTURE ONLY).

``Checker("oracle")`` loads oracle/liblnr_oracle.so (our restatement),
``Checker("ref")`` loads oracle/_ref/libref_linear.so (the real reference,
only present where /root/reference was available to build it).  Both expose
the same calls, so parity tests are written once.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_u8p = C.POINTER(C.c_uint8)
_u64p = C.POINTER(C.c_uint64)
_i32p = C.POINTER(C.c_int32)


def build(ref: bool = True) -> None:
    subprocess.check_call(["make", "-s", "-C", HERE, "liblnr_oracle.so"] + (["ref"] if ref else []))


def have_ref() -> bool:
    return os.path.exists(os.path.join(HERE, "_ref", "libref_linear.so"))


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(t)


class Checker:
    def __init__(self, kind: str, seqs: list[np.ndarray], T: int = 1, index_type: int = 1):
        assert kind in ("oracle", "ref")
        path = os.path.join(HERE, "liblnr_oracle.so") if kind == "oracle" else os.path.join(HERE, "_ref", "libref_linear.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.kind = kind
        self.pfx = "orc_" if kind == "oracle" else "ref_"
        self.lib = C.CDLL(path)
        L = self.lib
        f = self._f
        f("create2").restype = C.c_void_p
        f("create2").argtypes = [C.POINTER(_u8p), _u64p, C.c_uint32, C.c_uint32, C.c_int]
        for n in ("ysa_len", "empty_dir"):
            f(n).restype = C.c_uint64
            f(n).argtypes = [C.c_void_p]
        f("ysa").restype = _u64p
        f("ysa").argtypes = [C.c_void_p]
        f("destroy").argtypes = [C.c_void_p]
        for n in ("dir_len", "hs_len"):
            f(n).restype = C.c_uint64
            f(n).argtypes = [C.c_void_p]
        f("dir").restype = _i32p
        f("dir").argtypes = [C.c_void_p]
        f("hs").restype = _u64p
        f("hs").argtypes = [C.c_void_p]
        f("f2_len").restype = C.c_uint64
        f("f2_len").argtypes = [C.c_void_p, C.c_uint32]
        f("f2").restype = _i32p
        f("f2").argtypes = [C.c_void_p, C.c_uint32]
        f("read_features").restype = C.c_uint64
        f("read_features").argtypes = [_u8p, C.c_uint64, C.c_int, _i32p, C.c_uint64]
        f("seed_lookup").restype = C.c_uint64
        f("seed_lookup").argtypes = [C.c_void_p, _u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, _u64p, C.c_uint64, _u64p]
        f("map_read").restype = C.c_uint64
        f("map_read").argtypes = [C.c_void_p, _u8p, C.c_uint64]
        f("get_cords").argtypes = [C.c_void_p, _u64p, _u64p]
        if kind == "oracle":
            L.orc_map_batch.restype = C.c_uint64
            L.orc_map_batch.argtypes = [C.c_void_p, _u8p, _u64p, C.c_uint32, C.c_int, _u64p, _u64p, _u64p, C.c_uint64, _u64p]
            L.orc_debug.argtypes = [C.c_void_p, C.c_int]
            L.orc_debug_get.restype = C.c_uint64
            L.orc_debug_get.argtypes = [C.c_void_p, C.c_int, _u64p, C.c_uint64]
            L.orc_fill_mismatch.restype = C.c_uint64
            L.orc_fill_mismatch.argtypes = [C.c_void_p]
            L.orc_reset_stats.argtypes = [C.c_void_p]
            L.orc_get_stats.argtypes = [C.c_void_p, _u64p]
        else:
            L.ref_map_batch.restype = C.c_uint64
            L.ref_map_batch.argtypes = [C.c_void_p, _u8p, _u64p, C.c_uint32, C.c_int, _u64p, _u64p, _u64p, C.c_uint64]
            L.ref_stage.restype = C.c_uint64
            L.ref_stage.argtypes = [C.c_void_p, _u8p, C.c_uint64, C.c_int, _u64p, C.c_uint64]
        self._seqs = [np.ascontiguousarray(s, dtype=np.uint8) for s in seqs]
        ptrs = (_u8p * len(seqs))(*[_p(s, _u8p) for s in self._seqs])
        lens = np.array([s.size for s in self._seqs], dtype=np.uint64)
        self.index_type = index_type   # the reference's -i: 1 DIndex, 2 HIndex
        self.h = C.c_void_p(f("create2")(ptrs, _p(lens, _u64p), len(seqs), T, index_type))
        self.nseq = len(seqs)

    def _f(self, name):
        return getattr(self.lib, self.pfx + name)

    def close(self):
        if self.h:
            self._f("destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- index / features
    def ysa(self) -> np.ndarray:
        """HIndex (-i 2): the sorted block array the lookups walk (parity surface like dir / hs for -i 1)."""
        n = int(self._f("ysa_len")(self.h))
        return np.ctypeslib.as_array(self._f("ysa")(self.h), shape=(n,)).copy() if n else np.zeros(0, dtype=np.uint64)

    def empty_dir(self) -> int:
        return int(self._f("empty_dir")(self.h))

    def dir(self) -> np.ndarray:
        n = self._f("dir_len")(self.h)
        return np.ctypeslib.as_array(self._f("dir")(self.h), shape=(n,)).copy()

    def hs(self) -> np.ndarray:
        n = self._f("hs_len")(self.h)
        if n == 0:
            return np.zeros(0, np.uint64)
        return np.ctypeslib.as_array(self._f("hs")(self.h), shape=(n,)).copy()

    def f2(self, sid: int) -> np.ndarray:
        n = self._f("f2_len")(self.h, sid)
        if n == 0:
            return np.zeros((0, 3), np.int32)
        return np.ctypeslib.as_array(self._f("f2")(self.h, sid), shape=(n, 3)).copy()

    def read_features(self, read: np.ndarray, strand: int) -> np.ndarray:
        read = np.ascontiguousarray(read, dtype=np.uint8)
        cap = read.size // 16 + 8
        out = np.zeros((cap, 3), np.int32)
        n = self._f("read_features")(_p(read, _u8p), read.size, strand, _p(out, _i32p), cap)
        return out[:n]

    # ---- per read
    def seed_lookup(self, read: np.ndarray, read_str: int = 0, read_end: int | None = None, alpha: int = 15):
        read = np.ascontiguousarray(read, dtype=np.uint8)
        if read_end is None:
            read_end = read.size
        st = np.zeros(4, np.uint64)
        n = self._f("seed_lookup")(self.h, _p(read, _u8p), read.size, read_str, read_end, alpha, None, 0, _p(st, _u64p))
        out = np.zeros(max(int(n), 1), np.uint64)
        self._f("seed_lookup")(self.h, _p(read, _u8p), read.size, read_str, read_end, alpha, _p(out, _u64p), out.size, _p(st, _u64p))
        return out[:n], st

    def map_read(self, read: np.ndarray):
        read = np.ascontiguousarray(read, dtype=np.uint8)
        n = self._f("map_read")(self.h, _p(read, _u8p), read.size)
        cs = np.zeros(max(int(n), 1), np.uint64)
        ce = np.zeros(max(int(n), 1), np.uint64)
        self._f("get_cords")(self.h, _p(cs, _u64p), _p(ce, _u64p))
        return cs[:n], ce[:n]

    def map_read_gap(self, read: np.ndarray, gap_len: int = 50, dup: int = 0, ext: int = 0):
        """apxMap + mapGaps + reformCords as `linear filter -g gap_len [-dup 1]` runs them.  `ext`: the stream state the
        read starts from (0 = nothing extended yet in this thread's stream, 1 = extended: thd_cts_major_limit 3); the state it leaves
        behind is kept in self.ext_out."""
        fn = self._f("map_read_g2")
        fn.restype = C.c_uint64
        fn.argtypes = [C.c_void_p, _u8p, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_int)]
        read = np.ascontiguousarray(read, dtype=np.uint8)
        st = C.c_int(ext)
        n = int(fn(self.h, _p(read, _u8p), read.size, gap_len, dup, C.byref(st)))
        self.ext_out = st.value
        cs, ce = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        if n:
            self._f("get_cords")(self.h, _p(cs, _u64p), _p(ce, _u64p))
        return cs, ce

    def gaps(self) -> np.ndarray:
        """apx_gaps of the last map_read: [k, 2] cord words (start, end)."""
        fn = self._f("get_gaps")
        fn.restype = C.c_uint64
        fn.argtypes = [C.c_void_p, _u64p, C.c_uint64]
        out = np.zeros(2 * 4096, np.uint64)
        n = fn(self.h, _p(out, _u64p), 4096)
        return out[: 2 * n].reshape(-1, 2).copy()

    def map_batch(self, reads: np.ndarray, off: np.ndarray, threads: int = 1, gap_len: int = 0, dup: int = 0, ext: int | None = 0):
        """CSR cords for a batch on `threads` host threads; returns (cord_off, cords_str, cords_end, stats5).  The reference
        (kind "ref") runs its own per-thread scratch as Mapper::p_calRecords does and has no counters (stats5 = zeros).
        gap_len > 0: the gap re-mapper behind apxMap (-g gap_len [-dup dup]).  `ext`: the stream state the batch starts from -- the result
        is the program's `-t 1` result in file order whatever `threads` is (the state after the batch is kept in self.ext_out);
        ext=None (reference only): one GapParms per thread and a dynamic schedule, as the program runs with -t threads."""
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = off.size - 1
        cap = int(off[-1]) // (16 if gap_len else 32) + 64 * n + 1024
        coff = np.zeros(n + 1, np.uint64)
        cs = np.zeros(cap, np.uint64)
        ce = np.zeros(cap, np.uint64)
        st = np.zeros(5, np.uint64)
        if self.kind == "ref":
            assert int(off[0]) == 0
            self.lib.ref_map_batch_g2.restype = C.c_uint64
            self.lib.ref_map_batch_g2.argtypes = [C.c_void_p, _u8p, _u64p, C.c_uint32, C.c_int, _u64p, _u64p, _u64p, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_int)]
            es = C.c_int(ext or 0)
            tot = self.lib.ref_map_batch_g2(self.h, _p(reads, _u8p), _p(off, _u64p), n, threads, _p(coff, _u64p), _p(cs, _u64p), _p(ce, _u64p), cap, gap_len, dup,
                                            None if ext is None else C.byref(es))
            self.ext_out = es.value
            assert tot <= cap, "cord capacity"
            return coff, cs[:tot], ce[:tot], st
        self.lib.orc_map_batch_g2.restype = C.c_uint64
        self.lib.orc_map_batch_g2.argtypes = [C.c_void_p, _u8p, _u64p, C.c_uint32, C.c_int, _u64p, _u64p, _u64p, C.c_uint64, _u64p, C.c_uint32, C.c_int, C.POINTER(C.c_int)]
        es = C.c_int(ext or 0)
        tot = self.lib.orc_map_batch_g2(self.h, _p(reads, _u8p), _p(off, _u64p), n, threads, _p(coff, _u64p), _p(cs, _u64p), _p(ce, _u64p), cap, _p(st, _u64p), gap_len, dup, C.byref(es))
        self.ext_out = es.value
        assert tot <= cap, "cord capacity"
        return coff, cs[:tot], ce[:tot], st

    def format(self, reads: np.ndarray, off: np.ndarray, read_ids: list[str], genome_ids: list[str], cmd_line: str = "") -> tuple[bytes, bytes]:
        """reference only: (SAM records, APF text) of a block, produced by the reference's own cords2BamLink / fillBamRecords /
        printAlignSamBam / print_cords_apf."""
        assert self.kind == "ref"
        import tempfile
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        self.lib.ref_format.argtypes = [C.c_void_p, _u8p, _u64p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]
        with tempfile.TemporaryDirectory() as td:
            ps, pa = os.path.join(td, "o.sam"), os.path.join(td, "o.apf")
            rc = self.lib.ref_format(self.h, _p(reads, _u8p), _p(off, _u64p), off.size - 1, "\n".join(read_ids).encode(), "\n".join(genome_ids).encode(),
                                     cmd_line.encode(), ps.encode(), pa.encode())
            assert rc == 0
            return open(ps, "rb").read(), open(pa, "rb").read()

    def stage(self, read: np.ndarray, stage: int) -> np.ndarray:
        """Stage dump of the first apxMap_ pass: 0 raw anchors, 1 filtered, 2 x-desc sorted, 3 hits after anchor chaining
        (oracle additionally: 4 hits after block chaining, 5 after window filter, 6 cords after path_dst)."""
        read = np.ascontiguousarray(read, dtype=np.uint8)
        if self.kind == "ref":
            n = self.lib.ref_stage(self.h, _p(read, _u8p), read.size, stage, None, 0)
            out = np.zeros(max(int(n), 1), np.uint64)
            self.lib.ref_stage(self.h, _p(read, _u8p), read.size, stage, _p(out, _u64p), out.size)
            return out[:n]
        self.lib.orc_debug(self.h, 1)
        self._f("map_read")(self.h, _p(read, _u8p), read.size)
        n = self.lib.orc_debug_get(self.h, stage, None, 0)
        out = np.zeros(max(int(n), 1), np.uint64)
        self.lib.orc_debug_get(self.h, stage, _p(out, _u64p), out.size)
        self.lib.orc_debug(self.h, 0)
        return out[:n]


def ref_read_file(path: str):
    """The reference's own reader (SeqAn readRecords) on a FASTA / FASTQ(.gz) file -> (bases, off, ids).  TEST-ONLY."""
    lib = C.CDLL(os.path.join(HERE, "_ref", "libref_linear.so"))
    lib.ref_read_file.restype = C.c_uint64
    lib.ref_read_file.argtypes = [C.c_char_p, _u8p, C.c_uint64, _u64p, C.c_uint64, C.c_char_p, C.c_uint64]
    cap = max(os.path.getsize(path) * 12, 1 << 16)
    bases = np.zeros(cap, np.uint8)
    off = np.zeros(1 << 16, np.uint64)
    ids = C.create_string_buffer(1 << 22)
    n = lib.ref_read_file(os.fsencode(path), _p(bases, _u8p), cap, _p(off, _u64p), off.size - 1, ids, len(ids))
    assert n < (1 << 60), f"ref_read_file failed ({n:#x})"
    return bases[: int(off[n])].copy(), off[: n + 1].copy(), ids.value.decode(errors="replace").split("\n")[:n]
