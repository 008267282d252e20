// linear_filter_main.cpp -- `linear filter` front-end over the C ABI (include/linear_amd.h): the reference's command line
// (src/args_parser.cpp:31,150-270: `linear filter <reads> <genome> [-o prefix] [-t N] [-g len] [-dup 0|1] [-ot 1|2|3]`) driving
//     lnr_reader_*  (FASTA / FASTQ(.gz) -> pinned blocks)  ->  lnr_filter_submit / lnr_filter_wait (HIP hot path)  ->
//     lnr_writer_*  (cords -> <prefix>.sam / <prefix>.apf, mapper.cpp:360,627)
// with two read blocks in flight: block k + 1 is decoded and uploaded while block k is on the GPU, and block k - 1's text is
// written.  Plain C++ host code: everything it does goes through the ABI, so it doubles as the integration example.
// -g as in the reference: 1 (the default) = gaps of 50 and more are re-mapped, 0 = off (base.cpp:34, mapper.cpp:209-231).  Out of
// scope here as in the library: alignment (-a).  Output order = input order.
#include "../../include/linear_amd.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Block {
    uint8_t *bases = nullptr; uint64_t cap = 0;
    std::vector<uint64_t> off, len, id_off;
    std::vector<char> ids;
    uint32_t n = 0;
};

int main(int argc, char **argv) {
    if (argc < 4 || strcmp(argv[1], "filter") != 0) {
        fprintf(stderr, "usage: %s filter <reads.fa|fq[.gz]> <genome.fa[.gz]> [-o prefix] [-t threads] [-i 1|2] [-g gap_len] [-dup 0|1] [-ot 1|2|3] [-b reads_per_block]\n", argv[0]);
        return 2;
    }
    std::string reads_path = argv[2], genome_path = argv[3], prefix = "out";
    unsigned threads = 1, ot = 3, gap = 1, dup = 0, index_type = 1;   // (Options::Options base.cpp:28-45)
    uint32_t block_reads = 65536;
    for (int i = 4; i + 1 < argc; i += 2) {
        std::string k = argv[i];
        if (k == "-o") prefix = argv[i + 1];
        else if (k == "-t") threads = (unsigned)atoi(argv[i + 1]);
        else if (k == "-g") gap = (unsigned)atoi(argv[i + 1]);
        else if (k == "-dup") dup = (unsigned)atoi(argv[i + 1]) ? 1u : 0u;
        else if (k == "-i") index_type = (unsigned)atoi(argv[i + 1]);   // 1 DIndex, 2 HIndex (args_parser.cpp:221)
        else if (k == "-ot") ot = (unsigned)atoi(argv[i + 1]);
        else if (k == "-b") block_reads = (uint32_t)atoi(argv[i + 1]);
        else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    if (threads < 1) threads = 1;
    double t0 = now();
    // ---- genome
    lnr_reader *gr = nullptr;
    if (lnr_reader_open(genome_path.c_str(), &gr) != LNR_OK) { fprintf(stderr, "E[10]: can't open genome file %s\n", genome_path.c_str()); return 1; }
    std::vector<std::vector<uint8_t> > genome;
    std::vector<std::string> gid;
    {
        std::vector<uint8_t> buf((size_t)1 << 30);
        std::vector<uint64_t> off(2);
        for (;;) {
            uint32_t n = 0;
            lnr_status s = lnr_reader_next(gr, buf.data(), buf.size(), off.data(), 1, &n);
            if (s != LNR_OK) { fprintf(stderr, "E: genome: %s\n", lnr_reader_error(gr)); return 1; }
            if (!n) break;
            genome.emplace_back(buf.begin(), buf.begin() + (long)off[1]);
            const char *ids; const uint64_t *io;
            lnr_reader_ids(gr, &ids, &io);
            std::string id(ids);
            gid.push_back(id.substr(0, id.find(' ')));            // genome ids are cut at the first blank (base.cpp:188-195)
        }
        lnr_reader_close(gr);
    }
    if (genome.empty() || genome.size() >= 1024) { fprintf(stderr, "E: %zu reference sequences (1 .. 1023 supported, linear.cpp:107)\n", genome.size()); return 1; }
    lnr_ctx *ctx = nullptr;
    lnr_opts opts;
    lnr_opts_default(&opts);
    opts.index_type = index_type;
    opts.gap_len = gap;
    opts.dup = dup;
    lnr_status s = lnr_create(&opts, &ctx);
    if (s != LNR_OK) { fprintf(stderr, "E: %s\n", lnr_strerror(s)); return 1; }
    std::vector<const uint8_t *> gp; std::vector<uint64_t> gl; std::vector<const char *> gn;
    for (size_t i = 0; i < genome.size(); i++) { gp.push_back(genome[i].data()); gl.push_back(genome[i].size()); gn.push_back(gid[i].c_str()); }
    if ((s = lnr_index_build(ctx, gp.data(), gl.data(), (uint32_t)gp.size(), threads)) != LNR_OK) { fprintf(stderr, "E: index: %s (%s)\n", lnr_strerror(s), lnr_last_error(ctx)); return 1; }
    fprintf(stderr, "  End creating index Elapsed time[s] %.2f\n", now() - t0);
    lnr_writer *wr = nullptr;
    lnr_writer_create(gn.data(), gl.data(), (uint32_t)gn.size(), &wr);
    FILE *fsam = (ot & 2) ? fopen((prefix + ".sam").c_str(), "wb") : nullptr;
    FILE *fapf = (ot & 1) ? fopen((prefix + ".apf").c_str(), "wb") : nullptr;
    const char *text; uint64_t size;
    // `@PG ... CL:` stays empty: the reference's Options constructor fills cmd_line only `if (length(argv) < 1)` (base.cpp:64-72), i.e. never
    if (fsam) { lnr_writer_sam_header(wr, "", &text, &size); fwrite(text, 1, size, fsam); }
    // ---- reads: two pinned blocks, one being decoded / uploaded while the other is on the GPU
    lnr_reader *rr = nullptr;
    if (lnr_reader_open(reads_path.c_str(), &rr) != LNR_OK) { fprintf(stderr, "E[10]: can't open read file %s\n", reads_path.c_str()); return 1; }
    Block blk[2];
    for (auto &b : blk) { b.cap = (uint64_t)block_reads * 12000 + (1u << 20); b.bases = (uint8_t *)lnr_host_alloc(b.cap); b.off.resize((size_t)block_reads + 1); if (!b.bases) { fprintf(stderr, "E: pinned allocation\n"); return 1; } }
    auto fetch = [&](Block &b) -> bool {
        if (lnr_reader_next(rr, b.bases, b.cap, b.off.data(), block_reads, &b.n) != LNR_OK) { fprintf(stderr, "E: reads: %s\n", lnr_reader_error(rr)); exit(1); }
        if (!b.n) return false;
        const char *ids; const uint64_t *io;
        lnr_reader_ids(rr, &ids, &io);
        b.id_off.assign(io, io + b.n + 1);
        b.ids.assign(ids, ids + io[b.n]);
        b.len.resize(b.n);
        for (uint32_t i = 0; i < b.n; i++) b.len[i] = b.off[i + 1] - b.off[i];
        return true;
    };
    uint64_t total_reads = 0;
    int cur = 0;
    bool have = fetch(blk[0]);
    if (have && lnr_filter_submit(ctx, blk[0].bases, blk[0].off.data(), blk[0].n) != LNR_OK) { fprintf(stderr, "E: %s\n", lnr_last_error(ctx)); return 1; }
    while (have) {
        Block &b = blk[cur], &nx = blk[cur ^ 1];
        bool more = fetch(nx);
        if (more && lnr_filter_submit(ctx, nx.bases, nx.off.data(), nx.n) != LNR_OK) { fprintf(stderr, "E: %s\n", lnr_last_error(ctx)); return 1; }
        lnr_cords c;
        if ((s = lnr_filter_wait(ctx, &c)) != LNR_OK) { fprintf(stderr, "E: filter: %s (%s)\n", lnr_strerror(s), lnr_last_error(ctx)); return 1; }
        if (fsam) { lnr_writer_format(wr, &c, b.len.data(), b.ids.data(), b.id_off.data(), 1, threads, &text, &size); fwrite(text, 1, size, fsam); }
        if (fapf) { lnr_writer_format(wr, &c, b.len.data(), b.ids.data(), b.id_off.data(), 2, threads, &text, &size); fwrite(text, 1, size, fapf); }
        total_reads += b.n;
        have = more; cur ^= 1;
    }
    if (fsam) fclose(fsam);
    if (fapf) fclose(fapf);
    lnr_reader_close(rr);
    for (auto &b : blk) lnr_host_free(b.bases);
    lnr_writer_destroy(wr);
    lnr_destroy(ctx);
    double dt = now() - t0;
    fprintf(stderr, "  Processed: %llu reads in %.2f s = %.2f reads/s\n", (unsigned long long)total_reads, dt, total_reads / (dt > 0 ? dt : 1));
    return 0;
}
