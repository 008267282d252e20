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


DEVICE_TUS = ["lnr_api.hip", "lnr_gap_kernels.hip"]      # compiled side by side (the gap re-mapper's kernels are half of the compile time)
# sources a translation unit does NOT see: an edit there leaves its object alone (everything else in csrc/ + the public header is a dependency)
NOT_A_DEP = {"lnr_api.hip": {"lnr_gap_kernels.hip", "lnr_gap_hd.h", "lnr_reader.cpp", "lnr_output.cpp", "linear_filter_main.cpp"},
             "lnr_gap_kernels.hip": {"lnr_api.hip", "lnr_kernels.hip", "lnr_reader.cpp", "lnr_output.cpp", "linear_filter_main.cpp"},
             "lnr_reader.cpp": set(SOURCES) - {"lnr_reader.cpp"}, "lnr_output.cpp": set(SOURCES) - {"lnr_output.cpp"}}


def stale(obj: str, src: str) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [os.path.join(CSRC, f) for f in SOURCES if f not in NOT_A_DEP[src]] + [os.path.join(HERE, "..", "include", "linear_amd.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, defines: tuple = (), out: str | None = None) -> str:
    """defines / out: variant builds for A/B and diagnostic runs (tools/build_variant.sh)"""
    so = out or SO
    if force or out or needs_build():
        from concurrent.futures import ThreadPoolExecutor
        tag = "" if not out else "." + os.path.basename(out).replace(".so", "")
        jobs = []
        # host-only translation units (FASTA / FASTQ reader, SAM / APF writer) by the host compiler, device code + C ABI by hipcc, one library
        for src in ("lnr_reader.cpp", "lnr_output.cpp"):
            obj = os.path.join(HERE, src.replace(".cpp", ".o"))
            jobs.append((obj, None if not (force or stale(obj, src)) else ["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-c", os.path.join(CSRC, src), "-o", obj]))
        for src in DEVICE_TUS:
            obj = os.path.join(HERE, src.replace(".hip", tag + ".o"))
            jobs.append((obj, None if not (force or out or stale(obj, src)) else
                         [hipcc_path()] + [f for f in FLAGS if f != "-shared"] + list(defines) + ["-c", os.path.join(CSRC, src), "-o", obj]))
        with ThreadPoolExecutor(len(jobs)) as ex:
            for rc, (obj, cmd) in zip(ex.map(lambda j: subprocess.call(j[1]) if j[1] else 0, jobs), jobs):
                if rc:
                    raise RuntimeError("compile failed: " + " ".join(cmd))
        objs = ["-Wl," + o for o, _ in jobs]                      # (-Wl: hipcc would compile a bare .o as HIP source)
        subprocess.check_call([hipcc_path(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", so] + objs + ["-lz", "-lpthread"])
        if not out:
            # the `linear filter` front-end over the ABI (plain C++, links the library)
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", os.path.join(CSRC, "linear_filter_main.cpp"), "-o", CLI, SO,
                                   "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lpthread"])
    return so


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1:         # python -m linear_amd.build <variant name> [-DFLAG ...]  ->  tools/_variants/<name>.so
        vdir = os.path.join(HERE, "..", "tools", "_variants")
        os.makedirs(vdir, exist_ok=True)
        print(build(defines=tuple(sys.argv[2:]), out=os.path.join(vdir, sys.argv[1] + ".so")))
    else:
        print(build(force=True))
