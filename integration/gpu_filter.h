// gpu_filter.h -- reference-side binding of include/linear_amd.h: the one file a maintainer of xp3i4/linear adds
// (as include/gpu_filter.h) to route Mapper's compute path through the MI355X library.  SeqAn types stay on this side of
// the boundary; only pointers and sizes cross it.  Compiles with the reference's own flags (C++11/14); link with
// -llinear_amd.  tests/test_boundary_compile_cpu.py builds it against the reference's vendored SeqAn headers.
//
//   buildIndex   replaces createFeatures(genomes, f2, type, threads) + Mapper::createIndex -> createIndexDynamic
//                (src/linear.cpp:76-77, include/pmpfinder.h:196-197, include/index_util.h:297-302)
//   filterBlock  replaces the `for j` loop of Mapper::p_calRecords (src/mapper.cpp:438-462, -g 0) for one block of reads
#ifndef LINEAR_GPU_FILTER_H
#define LINEAR_GPU_FILTER_H

#include <cstdint>
#include <vector>

#include <seqan/sequence.h>

#include "linear_amd.h"

struct GpuFilter {
    lnr_ctx *ctx;
    explicit GpuFilter(int device = -1, unsigned index_type = 1 /* options.index_t: 1 DIndex, 2 HIndex (mapper.cpp:200) */,
                       unsigned gap_len = 1 /* options.gap_len (mapper.cpp:209-231) */, unsigned f_dup = 0 /* options.f_dup (mapper.cpp:208) */) : ctx(nullptr) {
        lnr_opts o;
        lnr_opts_default(&o);
        o.device = device;
        o.index_type = index_type;
        o.gap_len = gap_len;
        o.dup = f_dup;
        if (lnr_create(&o, &ctx) != LNR_OK) ctx = nullptr;   // no GPU -> the caller keeps the CPU path
    }
    ~GpuFilter() { lnr_destroy(ctx); }
    bool ok() const { return ctx != nullptr; }

    int buildIndex(seqan::StringSet<seqan::String<seqan::Dna5> > &genomes, unsigned threads) {
        std::vector<const uint8_t *> p;
        std::vector<uint64_t> n;
        for (unsigned i = 0; i < length(genomes); i++) {
            // String<Dna5> stores one ordinal byte per base contiguously (seqan/sequence/string_alloc.h:66-71)
            p.push_back(reinterpret_cast<const uint8_t *>(&genomes[i][0]));
            n.push_back(length(genomes[i]));
        }
        return lnr_index_build(ctx, p.data(), n.data(), (uint32_t)p.size(), threads);   // threads = -t (index layout)
    }

    int filterBlock(seqan::StringSet<seqan::String<seqan::Dna5> > &reads,
                    seqan::StringSet<seqan::String<uint64_t> > &cords_str,
                    seqan::StringSet<seqan::String<uint64_t> > &cords_end) {
        std::vector<uint8_t> cat;
        std::vector<uint64_t> off(1, 0);
        for (unsigned j = 0; j < length(reads); j++) {
            const uint8_t *b = reinterpret_cast<const uint8_t *>(&reads[j][0]);
            cat.insert(cat.end(), b, b + length(reads[j]));
            off.push_back(cat.size());
        }
        lnr_cords out;
        int rc = lnr_filter_batch(ctx, cat.data(), off.data(), (uint32_t)length(reads), &out);
        if (rc != LNR_OK) return rc;
        resize(cords_str, length(reads));
        resize(cords_end, length(reads));
        for (unsigned j = 0; j < length(reads); j++) {
            uint64_t a = out.cord_off[j], e = out.cord_off[j + 1];
            resize(cords_str[j], e - a);
            resize(cords_end[j], e - a);
            for (uint64_t k = a; k < e; k++) { cords_str[j][k - a] = out.cords_str[k]; cords_end[j][k - a] = out.cords_end[k]; }
        }
        return LNR_OK;
    }
};

#endif
