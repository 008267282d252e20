#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (oracle/_ref, built from
/root/reference by oracle/Makefile) on the seeded cases of tests/cases.py.

Only runs where /root/reference exists.  The stored vectors are data (inputs are
regenerated from seeds; outputs are the reference's words), never reference source.

Per (case, T):
  digest            sha256 of the regenerated inputs (guards against generator drift)
  dir_sha/hs_sha    sha256 of the DIndex arrays, hs_len, dir_nonempty
  hs_head           first 4096 hs entries (spot check with readable diffs)
  f2_sha[k]/f2_len  genome window features per sequence (last element excluded: SURVEY App. C.5)
  cord_off, cords_str, cords_end   CSR of the final cords of every read (the parity surface)
  stage reads: raw anchors, filtered anchors, x-sorted anchors, chained hits, read features
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pyorc  # noqa: E402
from tests import cases  # noqa: E402


def main():
    pyorc.build(ref=True)
    assert pyorc.have_ref(), "oracle/_ref not built (no /root/reference?)"
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    only = sys.argv[1:]
    if not only or "reader" in only:
        # input side: the reference's own SeqAn reader on the seeded FASTA / FASTQ fixtures
        import tempfile
        from tests import reader_cases
        with tempfile.TemporaryDirectory() as td:
            files, _ = reader_cases.write_cases(td)
            d = {}
            for name, path in files.items():
                b, o, ids = pyorc.ref_read_file(path)
                d[name + ":bases"], d[name + ":off"], d[name + ":ids"] = b, o, np.array(ids)
            np.savez_compressed(os.path.join(outdir, "reader.npz"), **d)
            print("reader.npz:", {k: int(v.size) for k, v in d.items() if k.endswith(":off")})
    # gap path (-g 50 [-dup 1]): the reference's cords after mapGaps + reformCords, the case taken as one read stream (-t 1)
    for name, (builder, T) in cases.CASES_G50.items():
        if only and "g50" not in only and (name + "_g50") not in only:
            continue
        refs, reads, off = builder()
        n = off.size - 1
        r = pyorc.Checker("ref", refs, T)
        d = {"digest": cases.input_digest(refs, reads, off), "T": T, "n_reads": n}
        for dup in (0, 1):
            # the whole case as ONE read stream in file order through one GapParms (`linear filter -t 1`: mapper.cpp:233-237,447 keeps one
            # per thread for the run and mapExtend / mapExtends leave thd_cts_major_limit = 3 behind; DESIGN 5c "stream state")
            coff, cs, ce, _ = r.map_batch(reads, off, threads=1, gap_len=50, dup=dup, ext=0)
            d[f"cord_off_dup{dup}"], d[f"cords_str_dup{dup}"], d[f"cords_end_dup{dup}"] = coff, cs, ce
            d[f"ext_out_dup{dup}"] = r.ext_out
        path = os.path.join(outdir, f"{name}_g50_T{T}.npz")
        np.savez_compressed(path, **d)
        print(f"{path}: reads {n} cords {int(d['cord_off_dup0'][-1])} / {int(d['cord_off_dup1'][-1])} size {os.path.getsize(path) / 1024:.0f} kB")
        r.close()
    # HIndex (-i 2): ysa digest, raw anchors of a few reads, cords of every read
    for name, (builder, layouts) in cases.CASES_I2.items():
        if only and (name + "_i2") not in only and "i2" not in only:
            continue
        refs, reads, off = builder()
        n = off.size - 1
        for T in layouts:
            r = pyorc.Checker("ref", refs, T, index_type=2)
            ysa = r.ysa()
            d = {"digest": cases.input_digest(refs, reads, off), "T": T, "n_reads": n, "ysa_sha": cases.sha(ysa), "ysa_len": ysa.size,
                 "ysa_head": ysa[:4096], "empty_dir": r.empty_dir()}
            coff = np.zeros(n + 1, np.uint64)
            cs_l, ce_l = [], []
            for i in range(n):
                cs, ce = r.map_read(reads[int(off[i]):int(off[i + 1])])
                cs_l.append(cs); ce_l.append(ce)
                coff[i + 1] = coff[i] + cs.size
            d["cord_off"], d["cords_str"], d["cords_end"] = coff, np.concatenate(cs_l), np.concatenate(ce_l)
            idx = [i for i in range(n) if int(off[i + 1] - off[i]) > 200][: cases.N_STAGE_READS]
            d["stage_reads"] = np.array(idx)
            for k, i in enumerate(idx):
                rd = reads[int(off[i]):int(off[i + 1])]
                d[f"st{k}_raw"], _ = r.seed_lookup(rd)
                d[f"st{k}_raw7"], _ = r.seed_lookup(rd, 100, rd.size - 50, 7)
            path = os.path.join(outdir, f"{name}_i2_T{T}.npz")
            np.savez_compressed(path, **d)
            print(f"{path}: reads {n} cords {int(coff[-1])} ysa {ysa.size} size {os.path.getsize(path) / 1024:.0f} kB")
            r.close()
    for name, (builder, layouts) in cases.CASES.items():
        if only and name not in only:
            continue
        refs, reads, off = builder()
        n = off.size - 1
        for T in layouts:
            r = pyorc.Checker("ref", refs, T)
            d = {"digest": cases.input_digest(refs, reads, off), "T": T, "n_reads": n}
            dir_, hs = r.dir(), r.hs()
            d["dir_sha"], d["hs_sha"], d["hs_len"] = cases.sha(dir_), cases.sha(hs), hs.size
            d["dir_nonempty"] = int((np.diff(dir_.astype(np.int64)) > 0).sum())
            d["hs_head"] = hs[:4096]
            d["f2_len"] = np.array([r.f2(k).shape[0] for k in range(len(refs))])
            d["f2_sha"] = np.array([cases.sha(r.f2(k)[:-1]) for k in range(len(refs))])
            coff = np.zeros(n + 1, np.uint64)
            cs_l, ce_l = [], []
            for i in range(n):
                rd = reads[int(off[i]):int(off[i + 1])]
                cs, ce = r.map_read(rd)
                cs_l.append(cs)
                ce_l.append(ce)
                coff[i + 1] = coff[i] + cs.size
            d["cord_off"] = coff
            d["cords_str"] = np.concatenate(cs_l) if cs_l else np.zeros(0, np.uint64)
            d["cords_end"] = np.concatenate(ce_l) if ce_l else np.zeros(0, np.uint64)
            # stage dumps on reads long enough to be mapped
            idx = [i for i in range(n) if int(off[i + 1] - off[i]) > 200][: cases.N_STAGE_READS]
            # for the edge case take a spread of the edge reads as well
            if name == "edge":
                idx = [i for i in range(n) if int(off[i + 1] - off[i]) > 200][:16]
            if name in cases.STAGE_IDX:
                idx = cases.STAGE_IDX[name]
            d["stage_reads"] = np.array(idx)
            for k, i in enumerate(idx):
                rd = reads[int(off[i]):int(off[i + 1])]
                for s, nm in enumerate(("raw", "filt", "xsort", "hits")):
                    d[f"st{k}_{nm}"] = r.stage(rd, s)
                d[f"st{k}_f1fwd"] = r.read_features(rd, 0)
                d[f"st{k}_f1rev"] = r.read_features(rd, 1)
                a7, _ = r.seed_lookup(rd, 100, rd.size - 50, 7)
                d[f"st{k}_raw7"] = a7
            # output shaping: SAM (header + records) and APF text of the whole block, by the reference's own writer functions
            rid, gid = cases.text_ids(n, len(refs))
            sam, apf = r.format(reads, off, rid, gid, cases.CMD_LINE)
            d["sam"], d["apf"] = np.frombuffer(sam, np.uint8), np.frombuffer(apf, np.uint8)
            path = os.path.join(outdir, f"{name}_T{T}.npz")
            np.savez_compressed(path, **d)
            print(f"{path}: reads {n} cords {int(coff[-1])} hs {hs.size} size {os.path.getsize(path) / 1024:.0f} kB")
            r.close()


if __name__ == "__main__":
    main()
