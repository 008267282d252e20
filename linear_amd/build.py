"""Builds linear_amd/liblinear_amd.so (the HIP library behind include/linear_amd.h) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this also runs in the authoring container; the resulting .so
travels to the GPU box with the snapshot."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "liblinear_amd.so")
CLI = os.path.join(HERE, "linear_filter")
SOURCES = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp", ".inc")))   # every source of csrc/: a new header cannot be forgotten
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def hipcc_path() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the MI355X library cannot be built")


def needs_build() -> bool:
    if not os.path.exists(SO) or not os.path.exists(CLI):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "..", "include", "linear_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False) -> str:
    if force or needs_build():
        # host-only translation units (FASTA / FASTQ reader, SAM / APF writer) by the host compiler, device code + C ABI by hipcc, one library
        objs = []
        for src in ("lnr_reader.cpp", "lnr_output.cpp"):
            obj = os.path.join(HERE, src.replace(".cpp", ".o"))
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-c", os.path.join(CSRC, src), "-o", obj])
            objs.append("-Wl," + obj)                         # (-Wl: hipcc would compile a bare .o as HIP source)
        cmd = [hipcc_path()] + FLAGS + ["-o", SO, os.path.join(CSRC, "lnr_api.hip")] + objs + ["-lz", "-lpthread"]
        subprocess.check_call(cmd)
        # the `linear filter` front-end over the ABI (plain C++, links the library)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", os.path.join(CSRC, "linear_filter_main.cpp"), "-o", CLI, SO,
                               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"])
    return SO


if __name__ == "__main__":
    print(build(force=True))
