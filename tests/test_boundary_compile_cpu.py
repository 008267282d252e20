"""CPU: the drop-in boundary as a C / C++ consumer sees it.
 1. include/linear_amd.h is plain C: a C11 translation unit includes it and static-asserts the size and field offsets of
    every struct against the ctypes mirrors in linear_amd/api.py (what the GPU tests marshal through).
 2. integration/gpu_filter.h -- the adaptor INTEGRATION.md tells a maintainer to add -- compiles against the reference's own
    vendored SeqAn headers, links with liblinear_amd.so and runs: without a GPU the context stays null and the caller keeps
    its CPU path (skipped where /root/reference is absent, i.e. on the GPU box)."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
SEQAN = "/root/reference/seqan/include"


def test_header_is_plain_c_and_layouts_match_ctypes(tmp_path):
    from linear_amd import api
    pairs = [("lnr_opts", api.LnrOpts), ("lnr_index_info", api.LnrIndexInfo), ("lnr_cords", api.LnrCords), ("lnr_cords_dev", api.LnrCordsDev),
             ("lnr_anchors", api.LnrAnchors), ("lnr_stats", api.LnrStats), ("lnr_gaps", api.LnrGaps)]
    lines = ['#include <stddef.h>', '#include "linear_amd.h"']
    for cname, st in pairs:
        lines.append(f'_Static_assert(sizeof({cname}) == {C.sizeof(st)}, "sizeof {cname}");')
        for fname, _ in st._fields_:
            lines.append(f'_Static_assert(offsetof({cname}, {fname}) == {getattr(st, fname).offset}, "offsetof {cname}.{fname}");')
    lines.append("_Static_assert(LNR_OK == 0 && LNR_ERR_NO_DEVICE == -2 && LNR_ERR_INTERNAL == -8, \"status codes\");")
    lines.append("int main(void) { lnr_opts o; lnr_status (*f)(const lnr_opts *, lnr_ctx **) = lnr_create; (void)f; (void)o; return 0; }")
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines) + "\n")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", f"-I{INC}", str(src)])


@pytest.mark.skipif(not os.path.exists(os.path.join(SEQAN, "seqan", "sequence.h")), reason="reference tree (SeqAn headers) not present")
def test_seqan_side_adaptor_compiles_links_and_runs(tmp_path):
    from linear_amd import build as lb
    so = lb.build()
    src = tmp_path / "adaptor.cpp"
    src.write_text('''
#include "gpu_filter.h"
#include <cstdio>
using namespace seqan;
int main() {
    GpuFilter gpu(-1);
    StringSet<String<Dna5> > genomes, reads;
    StringSet<String<uint64_t> > cs, ce;
    String<Dna5> g = "ACGTNACGTACGTTTGACCA";
    appendValue(genomes, g);
    appendValue(reads, g);
    // the byte view the ABI takes is the String's own storage: ordinals 0..4
    const uint8_t *b = reinterpret_cast<const uint8_t *>(&genomes[0][0]);
    if (!(b[0] == 0 && b[1] == 1 && b[2] == 2 && b[3] == 3 && b[4] == 4 && sizeof(Dna5) == 1)) return 3;
    if (!gpu.ok()) { std::printf("no device: %s\\n", lnr_strerror(LNR_ERR_NO_DEVICE)); return 0; }
    int rc = gpu.buildIndex(genomes, 1);
    if (rc != LNR_OK) return 4;
    rc = gpu.filterBlock(reads, cs, ce);
    std::printf("filterBlock rc %d, reads %u\\n", rc, (unsigned)length(cs));
    return rc == LNR_OK && length(cs) == 1 ? 0 : 5;
}
''')
    exe = tmp_path / "adaptor"
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-w", f"-I{INC}", f"-I{os.path.join(ROOT, 'integration')}", f"-I{SEQAN}", "-DSEQAN_ENABLE_DEBUG=0",
                           str(src), "-o", str(exe), so, f"-Wl,-rpath,{os.path.dirname(so)}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"])
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout.decode(), out.stderr.decode()[-500:])
