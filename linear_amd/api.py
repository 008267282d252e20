"""Python binding (ctypes) of the C ABI in include/linear_amd.h.

Thin plumbing only: numpy / torch buffers in, the reference's cord words out.  There is no
Python or CPU implementation of the path here; if liblinear_amd.so is missing or no GPU is
usable, construction raises."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.environ.get("LNR_LIB") or os.path.join(HERE, "liblinear_amd.so")   # LNR_LIB: a variant build (tools/build_variant.sh) for A/B measurements

_u8p, _u64p, _i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)


class LnrOpts(C.Structure):
    _fields_ = [("device", C.c_int32), ("index_type", C.c_uint32), ("feature_type", C.c_uint32), ("preset", C.c_uint32),
                ("gap_len", C.c_uint32), ("dup", C.c_uint32), ("scratch_budget", C.c_uint64)]


class LnrIndexInfo(C.Structure):
    _fields_ = [("nseq", C.c_uint32), ("layout_threads", C.c_uint32), ("genome_bytes", C.c_uint64), ("dir_len", C.c_uint64),
                ("hs_len", C.c_uint64), ("f2_len", C.c_uint64), ("n_samples", C.c_uint64), ("build_ms", C.c_double)]


class LnrCords(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("n_cords", C.c_uint64), ("cord_off", _u64p), ("cords_str", _u64p), ("cords_end", _u64p)]


class LnrCordsDev(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("n_cords", C.c_uint64), ("d_cord_off", C.c_void_p), ("d_cords_str", C.c_void_p), ("d_cords_end", C.c_void_p)]


class LnrGaps(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("n_gaps", C.c_uint64), ("gap_off", _u64p), ("gaps", _u64p)]


class LnrAnchors(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("n_anchors", C.c_uint64), ("anchor_off", _u64p), ("anchors", _u64p)]


class LnrStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("reads", "bases", "jobs", "samples", "lookups", "bucket_entries", "anchors", "remap_reads", "cords", "seed_bytes")] + \
               [(k, C.c_double) for k in ("prep_ms", "seed_count_ms", "seed_gather_ms", "job_ms", "tail_ms", "total_ms")] + \
               [(k, C.c_uint32) for k in ("seed_count_launches", "seed_gather_launches", "job_launches", "gap_second_pass")] + [("gap_ms", C.c_double)]


class LnrError(RuntimeError):
    def __init__(self, status: int, msg: str, detail: str = ""):
        super().__init__(f"linear_amd: {msg} (status {status}){': ' + detail if detail else ''}")
        self.status = status


EXPORTS = ["lnr_opts_default", "lnr_create", "lnr_destroy", "lnr_strerror", "lnr_last_error", "lnr_index_build", "lnr_index_info_get",
           "lnr_index_export", "lnr_index_alloc", "lnr_index_blob", "lnr_index_adopt", "lnr_filter_batch", "lnr_filter_batch_dev",
           "lnr_cords_to_host", "lnr_seed_lookup_batch", "lnr_seed_lookup_batch_dev", "lnr_last_stats", "lnr_filter_submit", "lnr_filter_wait",
           "lnr_host_alloc", "lnr_host_free", "lnr_reader_open", "lnr_reader_next", "lnr_reader_ids", "lnr_reader_error", "lnr_reader_close",
           "lnr_writer_create", "lnr_writer_format", "lnr_writer_sam_header", "lnr_writer_destroy", "lnr_last_gaps", "lnr_gap_stream", "lnr_set_gap", "lnr_index_broadcast", "lnr_writer_set_preset", "lnr_writer_set_read_group"]


def load_library() -> C.CDLL:
    if not os.path.exists(SO):
        raise LnrError(-2, "liblinear_amd.so is not built (run linear_amd.build.build()); there is no fallback path")
    lib = C.CDLL(SO)
    lib.lnr_strerror.restype = C.c_char_p
    lib.lnr_strerror.argtypes = [C.c_int]
    lib.lnr_last_error.restype = C.c_char_p
    lib.lnr_last_error.argtypes = [C.c_void_p]
    lib.lnr_opts_default.argtypes = [C.POINTER(LnrOpts)]
    lib.lnr_create.argtypes = [C.POINTER(LnrOpts), C.POINTER(C.c_void_p)]
    lib.lnr_destroy.argtypes = [C.c_void_p]
    lib.lnr_index_build.argtypes = [C.c_void_p, C.POINTER(_u8p), _u64p, C.c_uint32, C.c_uint32]
    lib.lnr_index_info_get.argtypes = [C.c_void_p, C.POINTER(LnrIndexInfo)]
    lib.lnr_index_export.argtypes = [C.c_void_p, _i32p, _u64p, _i32p, _u64p]
    lib.lnr_index_alloc.argtypes = [C.c_void_p, C.POINTER(LnrIndexInfo), _u64p]
    lib.lnr_index_blob.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), _u64p]
    lib.lnr_index_adopt.argtypes = [C.c_void_p]
    lib.lnr_filter_batch.argtypes = [C.c_void_p, _u8p, _u64p, C.c_uint32, C.POINTER(LnrCords)]
    lib.lnr_filter_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(LnrCordsDev)]
    lib.lnr_cords_to_host.argtypes = [C.c_void_p, C.POINTER(LnrCords)]
    lib.lnr_seed_lookup_batch.argtypes = [C.c_void_p, _u8p, _u64p, C.c_uint32, C.POINTER(LnrAnchors)]
    lib.lnr_seed_lookup_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    lib.lnr_last_stats.argtypes = [C.c_void_p, C.POINTER(LnrStats)]
    lib.lnr_gap_stream.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    lib.lnr_set_gap.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    lib.lnr_filter_submit.argtypes = [C.c_void_p, C.c_void_p, _u64p, C.c_uint32]
    lib.lnr_filter_wait.argtypes = [C.c_void_p, C.POINTER(LnrCords)]
    lib.lnr_host_alloc.restype = C.c_void_p
    lib.lnr_host_alloc.argtypes = [C.c_size_t]
    lib.lnr_host_free.argtypes = [C.c_void_p]
    lib.lnr_reader_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    lib.lnr_reader_next.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, _u64p, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.lnr_reader_ids.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(_u64p)]
    lib.lnr_reader_error.restype = C.c_char_p
    lib.lnr_reader_error.argtypes = [C.c_void_p]
    lib.lnr_reader_close.argtypes = [C.c_void_p]
    lib.lnr_writer_create.argtypes = [C.POINTER(C.c_char_p), _u64p, C.c_uint32, C.POINTER(C.c_void_p)]
    lib.lnr_writer_format.argtypes = [C.c_void_p, C.POINTER(LnrCords), _u64p, C.c_char_p, _u64p, C.c_int, C.c_uint32, C.POINTER(C.c_void_p), _u64p]
    lib.lnr_writer_sam_header.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), _u64p]
    lib.lnr_writer_destroy.argtypes = [C.c_void_p]
    lib.lnr_last_gaps.argtypes = [C.c_void_p, C.POINTER(LnrGaps)]
    return lib


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(t)


class Filter:
    """One context = one GPU.  Mirrors the Mapper's compute interface: build the index once, then filter read blocks."""

    def __init__(self, device: int = -1, scratch_budget: int = 0, index_type: int = 1, gap_len: int = 0, dup: int = 0):
        self.lib = load_library()
        o = LnrOpts()
        self.lib.lnr_opts_default(C.byref(o))
        o.device = device
        o.scratch_budget = scratch_budget
        o.index_type = index_type          # the reference's -i: 1 DIndex, 2 HIndex
        o.gap_len = gap_len                # the reference's -g: 0 = apxMap only, > 0 = + gap re-mapper (mapGaps, reformCords)
        o.dup = dup                        # the reference's -dup
        h = C.c_void_p()
        st = self.lib.lnr_create(C.byref(o), C.byref(h))
        if st != 0:
            raise LnrError(st, self.lib.lnr_strerror(st).decode())
        self.h = h
        self._seq_len = None

    def _ck(self, st: int):
        if st != 0:
            raise LnrError(st, self.lib.lnr_strerror(st).decode(), self.lib.lnr_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.lnr_destroy(self.h)
            self.h = None
            for p in getattr(self, "_pinned", []):
                self.lib.lnr_host_free(p)
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------- index
    def build_index(self, seqs: list[np.ndarray], layout_threads: int = 1) -> "LnrIndexInfo":
        seqs = [np.ascontiguousarray(s, dtype=np.uint8) for s in seqs]
        ptrs = (_u8p * len(seqs))(*[_p(s, _u8p) for s in seqs])
        lens = np.array([s.size for s in seqs], dtype=np.uint64)
        self._ck(self.lib.lnr_index_build(self.h, ptrs, _p(lens, _u64p), len(seqs), layout_threads))
        self._seq_len = lens
        return self.index_info()

    def build_index_ptrs(self, ptrs: list[int], lens: list[int], layout_threads: int = 1) -> "LnrIndexInfo":
        """Sequences given as raw addresses (host or device memory, e.g. slices of a torch tensor in HBM)."""
        arr = (_u8p * len(ptrs))(*[C.cast(C.c_void_p(int(p)), _u8p) for p in ptrs])
        lens = np.array(lens, dtype=np.uint64)
        self._ck(self.lib.lnr_index_build(self.h, arr, _p(lens, _u64p), len(ptrs), layout_threads))
        self._seq_len = lens
        return self.index_info()

    def index_info(self) -> "LnrIndexInfo":
        info = LnrIndexInfo()
        self._ck(self.lib.lnr_index_info_get(self.h, C.byref(info)))
        return info

    def index_export(self):
        info = self.index_info()
        dir_ = np.zeros(info.dir_len, np.int32)
        hs = np.zeros(max(info.hs_len, 1), np.uint64)
        f2 = np.zeros((max(info.f2_len, 1), 3), np.int32)
        f2_off = np.zeros(info.nseq + 1, np.uint64)
        self._ck(self.lib.lnr_index_export(self.h, _p(dir_, _i32p), _p(hs, _u64p), _p(f2, _i32p), _p(f2_off, _u64p)))
        return dir_, hs[:info.hs_len], f2[:info.f2_len], f2_off

    def index_blobs(self):
        """(device pointer, bytes) of the four index buffers (genome, dir, hs, f2) for an in-place RCCL broadcast."""
        out = []
        for k in range(4):
            p, b = C.c_void_p(), C.c_uint64()
            self._ck(self.lib.lnr_index_blob(self.h, k, C.byref(p), C.byref(b)))
            out.append((p.value, b.value))
        return out

    def index_alloc(self, info: "LnrIndexInfo", seq_len: np.ndarray):
        seq_len = np.ascontiguousarray(seq_len, dtype=np.uint64)
        self._ck(self.lib.lnr_index_alloc(self.h, C.byref(info), _p(seq_len, _u64p)))
        self._seq_len = seq_len

    def index_adopt(self):
        self._ck(self.lib.lnr_index_adopt(self.h))

    # small-vector forms used by linear_amd.dist.broadcast_index
    def index_info_vec(self) -> np.ndarray:
        i = self.index_info()
        return np.array([i.nseq, i.layout_threads, i.genome_bytes, i.dir_len, i.hs_len, i.f2_len, i.n_samples, 0], dtype=np.int64)

    def seq_len(self) -> np.ndarray:
        return np.asarray(self._seq_len, dtype=np.int64)

    def index_alloc_from(self, vec8: np.ndarray, seq_len: np.ndarray):
        info = LnrIndexInfo()
        info.nseq, info.layout_threads, info.genome_bytes, info.dir_len, info.hs_len, info.f2_len, info.n_samples = [int(v) for v in vec8[:7]]
        self.index_alloc(info, np.asarray(seq_len, dtype=np.uint64))

    # --------------------------------------------------------------- batches
    def filter_batch(self, reads: np.ndarray, off: np.ndarray):
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = off.size - 1
        out = LnrCords()
        self._ck(self.lib.lnr_filter_batch(self.h, _p(reads, _u8p), _p(off, _u64p), n, C.byref(out)))
        return self._cords_np(out)

    def host_alloc(self, nbytes: int) -> np.ndarray:
        """uint8 array over pinned host memory from lnr_host_alloc (freed with host_free or at close)."""
        p = self.lib.lnr_host_alloc(nbytes)
        if not p:
            raise LnrError(-4, "pinned host allocation failed")
        a = np.ctypeslib.as_array(C.cast(C.c_void_p(p), _u8p), shape=(nbytes,))
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        return a

    def filter_submit(self, reads: np.ndarray, off: np.ndarray):
        """Starts the upload of a batch (at most three in flight); reads/off must stay alive and unchanged until the filter_wait that returns it."""
        assert reads.dtype == np.uint8 and reads.flags.c_contiguous and off.dtype == np.uint64 and off.flags.c_contiguous
        self._inflight = getattr(self, "_inflight", [])
        self._ck(self.lib.lnr_filter_submit(self.h, C.c_void_p(reads.ctypes.data), _p(off, _u64p), off.size - 1))
        self._inflight.append((reads, off))

    def filter_wait(self, copy: bool = True):
        out = LnrCords()
        self._ck(self.lib.lnr_filter_wait(self.h, C.byref(out)))
        if getattr(self, "_inflight", None):
            self._inflight.pop(0)
        return self._cords_np(out) if copy else (out.n_reads, out.n_cords)

    @staticmethod
    def _cords_np(out: "LnrCords"):
        n = out.n_reads
        coff = np.ctypeslib.as_array(out.cord_off, shape=(n + 1,)).copy()
        if out.n_cords:
            cs = np.ctypeslib.as_array(out.cords_str, shape=(out.n_cords,)).copy()
            ce = np.ctypeslib.as_array(out.cords_end, shape=(out.n_cords,)).copy()
        else:
            cs, ce = np.zeros(0, np.uint64), np.zeros(0, np.uint64)
        return coff, cs, ce

    def filter_batch_dev(self, d_reads_ptr: int, d_off_ptr: int, n: int) -> "LnrCordsDev":
        out = LnrCordsDev()
        self._ck(self.lib.lnr_filter_batch_dev(self.h, C.c_void_p(d_reads_ptr), C.c_void_p(d_off_ptr), n, C.byref(out)))
        return out

    def last_gaps(self):
        """apx_gaps of the last filter call: (gap_off[n+1], gaps[k, 2])."""
        out = LnrGaps()
        self._ck(self.lib.lnr_last_gaps(self.h, C.byref(out)))
        goff = np.ctypeslib.as_array(out.gap_off, shape=(out.n_reads + 1,)).copy()
        g = np.ctypeslib.as_array(out.gaps, shape=(2 * out.n_gaps,)).copy().reshape(-1, 2) if out.n_gaps else np.zeros((0, 2), np.uint64)
        return goff, g

    def cords_to_host(self):
        out = LnrCords()
        self._ck(self.lib.lnr_cords_to_host(self.h, C.byref(out)))
        return self._cords_np(out)

    def seed_lookup_batch(self, reads: np.ndarray, off: np.ndarray):
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = off.size - 1
        out = LnrAnchors()
        self._ck(self.lib.lnr_seed_lookup_batch(self.h, _p(reads, _u8p), _p(off, _u64p), n, C.byref(out)))
        aoff = np.ctypeslib.as_array(out.anchor_off, shape=(n + 1,)).copy()
        a = np.ctypeslib.as_array(out.anchors, shape=(out.n_anchors,)).copy() if out.n_anchors else np.zeros(0, np.uint64)
        return aoff, a

    def seed_lookup_batch_dev(self, d_reads_ptr: int, d_off_ptr: int, n: int):
        self._ck(self.lib.lnr_seed_lookup_batch_dev(self.h, C.c_void_p(d_reads_ptr), C.c_void_p(d_off_ptr), n))

    def set_gap(self, gap_len: int, dup: int = 0) -> None:
        """-g / -dup of this context from now on (lnr_set_gap); a new read stream starts."""
        self._ck(self.lib.lnr_set_gap(self.h, gap_len, dup))

    def gap_stream(self, set_to: int = -1) -> int:
        """The read stream's state of the gap re-mapper (lnr_gap_stream): -1 query, 0 start a new stream, 1 mark it extended; returns the state."""
        st = C.c_int(0)
        self._ck(self.lib.lnr_gap_stream(self.h, set_to, C.byref(st)))
        return st.value

    def stats(self) -> dict:
        st = LnrStats()
        self._ck(self.lib.lnr_last_stats(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in LnrStats._fields_}


class Reader:
    """FASTA / FASTQ (plain or gzip) -> blocks in the ABI's layout (host code; no GPU needed)."""

    def __init__(self, path: str):
        self.lib = load_library()
        h = C.c_void_p()
        st = self.lib.lnr_reader_open(os.fsencode(path), C.byref(h))
        if st != 0:
            raise LnrError(st, self.lib.lnr_strerror(st).decode(), path)
        self.h = h

    def next(self, dst: np.ndarray, max_reads: int):
        """Fills dst (uint8, e.g. Filter.host_alloc) -> (n, off[n+1], ids[n]); n == 0 at end of file."""
        off = np.zeros(max_reads + 1, np.uint64)
        n = C.c_uint32()
        st = self.lib.lnr_reader_next(self.h, C.c_void_p(dst.ctypes.data), dst.size, _p(off, _u64p), max_reads, C.byref(n))
        if st != 0:
            raise LnrError(st, self.lib.lnr_strerror(st).decode(), self.lib.lnr_reader_error(self.h).decode())
        ids_p, ido_p = C.c_char_p(), _u64p()
        self.lib.lnr_reader_ids(self.h, C.byref(ids_p), C.byref(ido_p))
        ido = np.ctypeslib.as_array(ido_p, shape=(n.value + 1,)) if n.value else np.zeros(1, np.uint64)
        raw = C.string_at(C.cast(ids_p, C.c_void_p), int(ido[n.value])) if n.value else b""
        ids = [raw[int(ido[k]):int(ido[k + 1]) - 1].decode(errors="replace") for k in range(n.value)]
        return n.value, off[: n.value + 1].copy(), ids

    def close(self):
        if getattr(self, "h", None):
            self.lib.lnr_reader_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Writer:
    """Cords of a batch -> SAM records / APF text (host code; no GPU needed)."""

    def __init__(self, genome_ids: list[str], genome_len: list[int]):
        self.lib = load_library()
        arr = (C.c_char_p * len(genome_ids))(*[g.encode() for g in genome_ids])
        lens = np.array(genome_len, dtype=np.uint64)
        h = C.c_void_p()
        st = self.lib.lnr_writer_create(arr, _p(lens, _u64p), len(genome_ids), C.byref(h))
        if st != 0:
            raise LnrError(st, self.lib.lnr_strerror(st).decode())
        self.h = h

    def format(self, cord_off: np.ndarray, cords_str: np.ndarray, cords_end: np.ndarray, read_len: np.ndarray, read_ids: list[str], what: str, threads: int = 4) -> bytes:
        cord_off = np.ascontiguousarray(cord_off, dtype=np.uint64)
        cs = np.ascontiguousarray(cords_str, dtype=np.uint64)
        ce = np.ascontiguousarray(cords_end, dtype=np.uint64)
        rl = np.ascontiguousarray(read_len, dtype=np.uint64)
        c = LnrCords()
        c.n_reads, c.n_cords = cord_off.size - 1, cs.size
        c.cord_off, c.cords_str, c.cords_end = _p(cord_off, _u64p), _p(cs, _u64p), _p(ce, _u64p)
        blob = b"".join(i.encode() + b"\0" for i in read_ids)
        ido = np.zeros(len(read_ids) + 1, np.uint64)
        ido[1:] = np.cumsum([len(i.encode()) + 1 for i in read_ids])
        text, size = C.c_void_p(), C.c_uint64()
        st = self.lib.lnr_writer_format(self.h, C.byref(c), _p(rl, _u64p), blob, _p(ido, _u64p), {"sam": 1, "apf": 2}[what], threads, C.byref(text), C.byref(size))
        if st != 0:
            raise LnrError(st, self.lib.lnr_strerror(st).decode())
        return C.string_at(text, size.value)

    def sam_header(self, command_line: str) -> bytes:
        text, size = C.c_void_p(), C.c_uint64()
        self.lib.lnr_writer_sam_header(self.h, command_line.encode(), C.byref(text), C.byref(size))
        return C.string_at(text, size.value)

    def close(self):
        if getattr(self, "h", None):
            self.lib.lnr_writer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
