"""Seeded FASTA / FASTQ fixtures for the input-side reader (shared by tools/make_golden.py, which decodes them with the
reference's own SeqAn reader through oracle/_ref, and tests/test_reader_cpu.py)."""
import gzip
import os

import numpy as np


def write_cases(d):
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(20260)

    def mk(alpha, lens):
        return ["".join(rng.choice(list(alpha), size=int(n))) for n in lens]
    lens = [10, 130, 5000, 61, 1, 0, 977, 12000, 60, 120]
    seqs = mk("ACGTNacgtn", lens)
    out = {}

    def fa(name, crlf=False, blank=False, width=60, opener=open):
        p = os.path.join(d, name)
        with opener(p, "wt", newline="") as f:
            for i, s in enumerate(seqs):
                f.write(f">read{i} some description {i * 7}\n")
                for k in range(0, len(s), width):
                    f.write(s[k:k + width] + ("\r\n" if crlf else "\n"))
                if blank:
                    f.write("\n")
        out[name] = p

    def fq(name, multiline=False, opener=open):
        p = os.path.join(d, name)
        with opener(p, "wt", newline="") as f:
            for i, s in enumerate(seqs):
                q = "".join(chr(33 + (k * 7 + i) % 40) for k in range(len(s)))
                if multiline and len(s) > 100:
                    f.write(f"@q{i}/1 len={len(s)}\n" + "\n".join(s[k:k + 80] for k in range(0, len(s), 80)) + "\n+\n" + "\n".join(q[k:k + 80] for k in range(0, len(q), 80)) + "\n")
                else:
                    f.write(f"@q{i}/1 len={len(s)}\n{s}\n+\n{q}\n")
        out[name] = p
    fa("plain.fa"); fa("crlf_blank.fa", crlf=True, blank=True, width=70); fa("wide.fa.gz", width=100000, opener=gzip.open)
    fq("reads.fq"); fq("reads.fq.gz", opener=gzip.open); fq("multiline.fq", multiline=True)
    return out, seqs
