"""ctypes binding of tests/_build/libhost_shim.so: the product's stage logic
(linear_amd/csrc/lnr_hd.h) compiled for the host, TEST-ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libhost_shim.so")
_u8p, _u64p, _i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)


def build():
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    srcs = [os.path.join(HERE, "host_shim.cpp"), os.path.join(HERE, "..", "linear_amd", "csrc", "lnr_hd.h"),
            os.path.join(HERE, "..", "linear_amd", "csrc", "ref_sort.h"), os.path.join(HERE, "..", "linear_amd", "csrc", "lnr_gap_hd.h")]
    if os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(s) for s in srcs):
        return
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", SO, srcs[0]])


def _p(a, t):
    return a.ctypes.data_as(t)


class Shim:
    def __init__(self, seqs, T=1):
        build()
        L = self.lib = C.CDLL(SO)
        L.hs_create.restype = C.c_void_p
        L.hs_create.argtypes = [C.POINTER(_u8p), _u64p, C.c_uint32, C.c_uint32]
        L.hs_destroy.argtypes = [C.c_void_p]
        for n in ("hs_dir_len", "hs_hs_len"):
            getattr(L, n).restype = C.c_uint64
            getattr(L, n).argtypes = [C.c_void_p]
        L.hs_dir.restype = _i32p
        L.hs_dir.argtypes = [C.c_void_p]
        L.hs_hs.restype = _u64p
        L.hs_hs.argtypes = [C.c_void_p]
        L.hs_f2_len.restype = C.c_uint64
        L.hs_f2_len.argtypes = [C.c_void_p, C.c_uint32]
        L.hs_f2.argtypes = [C.c_void_p, C.c_uint32, _i32p]
        L.hs_read_features.restype = C.c_uint64
        L.hs_read_features.argtypes = [_u8p, C.c_uint64, C.c_int, _i32p, C.c_uint64]
        L.hs_seed_lookup.restype = C.c_uint64
        L.hs_seed_lookup.argtypes = [C.c_void_p, _u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, _u64p, C.c_uint64, _u64p]
        L.hs_map_read.restype = C.c_long
        L.hs_map_read.argtypes = [C.c_void_p, _u8p, C.c_uint64, C.c_int]
        L.hs_get_cords.argtypes = [C.c_void_p, _u64p, _u64p]
        L.hs_debug_get.restype = C.c_uint64
        L.hs_debug_get.argtypes = [C.c_void_p, C.c_int, _u64p, C.c_uint64]
        L.hs_get_stats.argtypes = [C.c_void_p, _u64p]
        self._seqs = [np.ascontiguousarray(s, dtype=np.uint8) for s in seqs]
        ptrs = (_u8p * len(seqs))(*[_p(s, _u8p) for s in self._seqs])
        lens = np.array([s.size for s in self._seqs], dtype=np.uint64)
        self.h = C.c_void_p(L.hs_create(ptrs, _p(lens, _u64p), len(seqs), T))

    def close(self):
        if self.h:
            self.lib.hs_destroy(self.h)
            self.h = None

    def dir(self):
        n = self.lib.hs_dir_len(self.h)
        return np.ctypeslib.as_array(self.lib.hs_dir(self.h), shape=(n,)).copy()

    def hs(self):
        n = self.lib.hs_hs_len(self.h)
        return np.ctypeslib.as_array(self.lib.hs_hs(self.h), shape=(n,)).copy() if n else np.zeros(0, np.uint64)

    def f2(self, sid):
        n = self.lib.hs_f2_len(self.h, sid)
        out = np.zeros((n, 3), np.int32)
        self.lib.hs_f2(self.h, sid, _p(out, _i32p))
        return out

    def read_features(self, read, strand):
        read = np.ascontiguousarray(read, dtype=np.uint8)
        cap = read.size // 16 + 8
        out = np.zeros((cap, 3), np.int32)
        n = self.lib.hs_read_features(_p(read, _u8p), read.size, strand, _p(out, _i32p), cap)
        return out[:n]

    def seed_lookup(self, read, read_str=0, read_end=None, alpha=15):
        read = np.ascontiguousarray(read, dtype=np.uint8)
        if read_end is None:
            read_end = read.size
        st = np.zeros(4, np.uint64)
        n = self.lib.hs_seed_lookup(self.h, _p(read, _u8p), read.size, read_str, read_end, alpha, None, 0, _p(st, _u64p))
        out = np.zeros(max(int(n), 1), np.uint64)
        self.lib.hs_seed_lookup(self.h, _p(read, _u8p), read.size, read_str, read_end, alpha, _p(out, _u64p), out.size, _p(st, _u64p))
        return out[:n], st

    def map_read(self, read, dbg=False):
        read = np.ascontiguousarray(read, dtype=np.uint8)
        n = self.lib.hs_map_read(self.h, _p(read, _u8p), read.size, int(dbg))
        assert n >= 0, f"shim error {n}"
        cs = np.zeros(max(int(n), 1), np.uint64)
        ce = np.zeros(max(int(n), 1), np.uint64)
        self.lib.hs_get_cords(self.h, _p(cs, _u64p), _p(ce, _u64p))
        return cs[:n], ce[:n]

    def map_read_gap(self, read, gap_len=50, dup=0, ext=0):
        """apxMap + the product's gap path (lnr_gap_hd.h: mapGaps + reformCords) on the host.  `ext`: the stream state the read starts
        from (oracle/pyorc.py map_read_gap); the state it leaves is kept in self.ext_out."""
        read = np.ascontiguousarray(read, dtype=np.uint8)
        self.lib.hs_map_read_g2.restype = C.c_int64
        self.lib.hs_map_read_g2.argtypes = [C.c_void_p, _u8p, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_int)]
        st = C.c_int(ext)
        n = self.lib.hs_map_read_g2(self.h, _p(read, _u8p), read.size, gap_len, dup, C.byref(st))
        self.ext_out = st.value
        assert n >= 0, f"shim error {n}"
        cs = np.zeros(max(int(n), 1), np.uint64)
        ce = np.zeros(max(int(n), 1), np.uint64)
        self.lib.hs_get_cords(self.h, _p(cs, _u64p), _p(ce, _u64p))
        return cs[:n], ce[:n]

    def stage(self, stage):
        n = self.lib.hs_debug_get(self.h, stage, None, 0)
        out = np.zeros(max(int(n), 1), np.uint64)
        self.lib.hs_debug_get(self.h, stage, _p(out, _u64p), out.size)
        return out[:n]
