// tests/gpufilter_driver.cpp -- runs integration/gpu_filter.h (the reference-side binding a maintainer of xp3i4/linear adds: SeqAn StringSets in,
// SeqAn StringSets out) on a real GPU.  Needs the reference's vendored SeqAn headers to COMPILE, so __graft_entry__.build() builds it where
// /root/reference exists (tests/_build/gpufilter_driver, an untracked artefact that travels to the GPU box like the library itself);
// tests/test_gpu_parity.py::test_gpu_seqan_side_binding_on_the_gpu runs it there.
//   usage: gpufilter_driver <case.bin> <out.bin> <T> <gap_len> <dup>
//   case.bin: u32 nseq, {u64 len, bytes}*, u32 nreads, {u64 len, bytes}*   (Dna5 ordinals)
//   out.bin : u32 nreads, {u64 ncords, cords_str words, cords_end words}*
#include "gpu_filter.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace seqan;

static bool read_set(FILE *f, StringSet<String<Dna5> > &set) {
    uint32_t n = 0;
    if (fread(&n, 4, 1, f) != 1) return false;
    for (uint32_t i = 0; i < n; i++) {
        uint64_t len = 0;
        if (fread(&len, 8, 1, f) != 1) return false;
        std::vector<uint8_t> b(len);
        if (len && fread(b.data(), 1, len, f) != len) return false;
        String<Dna5> s;
        resize(s, len);
        for (uint64_t k = 0; k < len; k++) s[k] = Dna5(b[k] > 4 ? 4 : b[k]);
        appendValue(set, s);
    }
    return true;
}

int main(int argc, char **argv) {
    if (argc < 6) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    StringSet<String<Dna5> > genomes, reads;
    if (!read_set(f, genomes) || !read_set(f, reads)) return 2;
    fclose(f);
    GpuFilter gpu(0, 1, (unsigned)atoi(argv[4]), (unsigned)atoi(argv[5]));
    if (!gpu.ok()) { fprintf(stderr, "no usable GPU\n"); return 3; }
    if (gpu.buildIndex(genomes, (unsigned)atoi(argv[3])) != LNR_OK) { fprintf(stderr, "index: %s\n", lnr_last_error(gpu.ctx)); return 4; }
    StringSet<String<uint64_t> > cs, ce;
    // two blocks through the same binding: the read stream (and with -g > 0 its state) continues across calls, as across the reference's blocks
    StringSet<String<Dna5> > b1, b2;
    for (unsigned j = 0; j < length(reads); j++) appendValue(j < length(reads) / 3 ? b1 : b2, reads[j]);
    StringSet<String<uint64_t> > cs1, ce1, cs2, ce2;
    if (gpu.filterBlock(b1, cs1, ce1) != LNR_OK || gpu.filterBlock(b2, cs2, ce2) != LNR_OK) { fprintf(stderr, "filter: %s\n", lnr_last_error(gpu.ctx)); return 5; }
    FILE *o = fopen(argv[2], "wb");
    if (!o) return 2;
    uint32_t n = (uint32_t)length(reads);
    fwrite(&n, 4, 1, o);
    for (unsigned j = 0; j < n; j++) {
        String<uint64_t> &s = j < length(b1) ? cs1[j] : cs2[j - length(b1)], &e = j < length(b1) ? ce1[j] : ce2[j - length(b1)];
        uint64_t k = length(s);
        fwrite(&k, 8, 1, o);
        if (k) { fwrite(&s[0], 8, k, o); fwrite(&e[0], 8, k, o); }
    }
    fclose(o);
    return 0;
}
