#!/usr/bin/env python3
"""tests/golden/cli_<case>.npz: the output of the program users run.

Runs the REAL `linear filter` binary (oracle/_ref/linear: the reference's own translation units compiled in place by oracle/Makefile
with plain g++) on FASTA dumps of the seeded cases of tests/cases.py (CASES_CLI) at `-t 1` for every mode of cases.CLI_MODES and stores
the bytes of its .sam and .apf.  `-t 1` because with `-g > 0` the program's result depends on the order reads meet a thread (one
GapParms per thread for the whole run, mapper.cpp:233-237,447) -- only one thread is reproducible in the reference itself.

Only runs where /root/reference exists.  The stored vectors are data (the program's output text), never reference source."""
from __future__ import annotations

import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pyorc  # noqa: E402
from tests import cases  # noqa: E402

BIN = os.path.join(ROOT, "oracle", "_ref", "linear")


def run_cli(rp, gp, flags, td, threads=1, index_type=1):
    pre = os.path.join(td, "out")
    for ext in (".sam", ".apf"):
        if os.path.exists(pre + ext):
            os.remove(pre + ext)
    cmd = [BIN, "filter", rp, gp, "-t", str(threads), "-ot", "3", "-o", pre] + (["-i", str(index_type)] if index_type != 1 else []) + flags
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=td)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return open(pre + ".sam", "rb").read(), open(pre + ".apf", "rb").read()


def main():
    pyorc.build(ref=True)
    assert os.path.exists(BIN), "oracle/_ref/linear not built (no /root/reference?)"
    outdir = os.path.join(ROOT, "tests", "golden")
    only = sys.argv[1:]
    for name, builder in cases.CASES_CLI.items():
        if only and name not in only:
            continue
        refs, reads, off = builder()
        d = {"digest": cases.input_digest(refs, reads, off), "n_reads": off.size - 1}
        with tempfile.TemporaryDirectory() as td:
            rp, gp, _, _ = cases.write_fasta_case(td, refs, reads, off)
            for mode, flags in cases.CLI_MODES.items():
                sam, apf = run_cli(rp, gp, flags, td)
                d[f"sam_{mode}"], d[f"apf_{mode}"] = np.frombuffer(sam, np.uint8), np.frombuffer(apf, np.uint8)
                lines = [l for l in sam.split(b"\n") if l and not l.startswith(b"@")]
                per = {}
                for l in lines:
                    q = l.split(b"\t")[0]
                    per[q] = per.get(q, 0) + 1
                hist = np.bincount(list(per.values()))
                print(f"{name} {mode}: sam {len(sam)} B, apf {len(apf)} B, records {len(lines)}, lines-per-read histogram {hist.tolist()}")
        path = os.path.join(outdir, f"cli_{name}.npz")
        np.savez_compressed(path, **d)
        print(f"{path}: {os.path.getsize(path) / 1024:.0f} kB")


if __name__ == "__main__":
    main()
