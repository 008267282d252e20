// linear_filter_main.cpp -- the `linear filter` front-end over the C ABI (include/linear_amd.h).  Plain C++ host code: everything it does goes
// through the ABI, so it doubles as the integration example.  It mirrors the reference's program around the hot path:
//   command line   src/args_parser.cpp:14-343 -- `filter` word optional, bare -g / -r / -ss mean 1 (:42-71), every option of the table at :150-270 by
//                  its short and long name, several read files and the `x` separator (:297-319), E[01] / E[02] / E[05] / E[06] (mapper.cpp:143-160)
//   defaults       Options::Options src/base.cpp:26-50 -- -t 16, -ot 2 (.sam only), -g 1 (= gaps of 50), -p 1, -i 1, -f 2
//   output naming  Mapper::p_printResults src/mapper.cpp:478-509 -- without -o one output per read file, named by the file's name up to its first '.';
//                  with -o one output for all read files
//   pipeline       process3 / p_ThreadProcess src/linear.cpp:68-91, src/parallel_io.cpp:372-608 -- ONE fetcher, calculators, ONE printer, output in
//                  input order -- here: a reader thread (FASTA / FASTQ(.gz) -> pinned blocks), one calculator thread + one lnr_ctx PER GPU (blocks dealt in
//                  order, three blocks in flight per GPU: upload, kernels and download of consecutive blocks overlap), a writer thread that restores
//                  the file order and formats on -t host threads.
//   several GPUs   --gpus N (extension): the index is built once on the first GPU and moved to the others with RCCL (lnr_index_broadcast, north_star;
//                  --index-mode build = every GPU builds its own instead; both times are printed).  Reads shard, nothing else is exchanged.
//   gap stream     with -g > 0 the reference's result depends on the order reads meet a thread's GapParms (lnr_gap_stream in the header): the blocks
//                  are taken strictly in file order until a block reports the stream "extended"; from then on every GPU runs freely with that state.
//                  The result is the reference's `-t 1` result whatever --gpus is.
// Not built (exit 1 with a message, never a silently different result): BAM output (-ot 4 / 8), -ss 1 (SEQ printing), -c 0, -f 1, -r 1, -p 0, -b 0
// (the reference's -b 0 path writes a header-only SAM: SURVEY App. C.7).
#include "../../include/linear_amd.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Options {
    std::vector<std::string> r_paths, g_paths;
    std::string oPath, read_group, sample_name;
    unsigned gap_len = 1, apx_chain_flag = 1, reform_ccs = 0, bal_flag = 1, f_output_type = 2, f_dup = 0, sensitivity = 1, thread = 16;
    int index_t = 1, feature_t = 2, sequence_sam = 0;
    // extensions of this front-end
    unsigned gpus = 1, block_reads = 65536, index_build_each = 0;
    std::vector<int> devices;
};

static bool is_number(const std::string &s) { if (s.empty()) return false; for (char c : s) if (c < '0' || c > '9') return false; return true; }

static void usage() {
    fprintf(stderr,
            "linear filter - options and arguments.\n\nSYNOPSIS\n    linear filter [OPTIONS] read.fa/fastq(.gz) genome.fa(.gz)\n    linear filter [OPTIONS] reads_1 reads_2 ... x genome_1 genome_2 ...\n\n"
            "Basic options\n    -o,  --output STR          prefix of the output (default: the read file's name up to its first '.')\n"
            "    -ot, --output_type INT     1 .apf, 2 .sam {DEFAULT}, 3 both (4 / 8: BAM, not built here)\n    -t,  --thread INT          threads: the index layout of the reference's -t and the host threads of the writer {16}\n"
            "    -g,  --gap_len INT         minimal length of gaps to re-map; -g 0 off; bare -g or 1 = 50 {DEFAULT}\n    -rg, --read_group STR      @RG ID\n    -sn, --sample_name STR     @RG SM\n"
            "    -ss, --sequence_sam INT    0 {DEFAULT} (1 not built here)\nMore options\n    -dup, --duplication INT    0 {DEFAULT} | 1\n    -b,  --bal_flag INT        1 {DEFAULT}\n"
            "    -p,  --preset INT          1 {DEFAULT} | 2   (0 not built here)\n    -i,  --index_type INT      1 {DEFAULT} | 2\n    -c,  --apx_c_flag INT      1 {DEFAULT}\n    -f,  --feature_type INT    2 {DEFAULT}\n"
            "    -r,  --reform_ccs_cigar_flag INT   0 {DEFAULT}\nMI355X front-end\n    --gpus INT                 GPUs to use {1}\n    --devices LIST             their HIP ordinals, e.g. 0,1,2,3\n"
            "    --block-reads INT          reads per block {65536}\n    --index-mode bcast|build   several GPUs: RCCL broadcast of the index {DEFAULT} or every GPU builds its own\n");
}

// returns 0 ok, 1 error, 2 help shown
static int parse_command_line(int argc, char **argv, Options &o) {
    std::vector<std::string> a;
    for (int i = 0; i < argc; i++) {
        if (i == 1 && strcmp(argv[1], "filter") == 0) continue;                                  // args_parser.cpp:31-40
        a.push_back(argv[i]);
        std::string s = argv[i];
        if ((s == "-a" || s == "-g" || s == "-os" || s == "-oa" || s == "-r" || s == "-ss") && (i + 1 >= argc || !is_number(argv[i + 1]))) a.push_back("1");   // :42-71
    }
    if (a.size() < 3) { usage(); return 2; }                                                      // :73-77 (-h appended)
    struct Opt { const char *sh, *lg; int kind; };   // kind 0 string, 1 integer
    static const Opt table[] = {{"o", "output", 0}, {"ot", "output_type", 1}, {"t", "thread", 1}, {"g", "gap_len", 1}, {"rg", "read_group", 0}, {"sn", "sample_name", 0},
                                {"ss", "sequence_sam", 1}, {"dup", "duplication", 1}, {"b", "bal_flag", 1}, {"p", "preset", 1}, {"i", "index_type", 1}, {"c", "apx_c_flag", 1},
                                {"f", "feature_type", 1}, {"r", "reform_ccs_cigar_flag", 1}, {"gpus", "gpus", 1}, {"devices", "devices", 0}, {"blk", "block-reads", 1}, {"ix", "index-mode", 0}};
    std::vector<std::string> pos;
    for (size_t i = 1; i < a.size(); i++) {
        const std::string &s = a[i];
        if (s == "-h" || s == "--help") { usage(); return 2; }
        if (s == "--version") { fprintf(stderr, "linear filter (MI355X path) 1.8.2\n"); return 2; }
        if (s.size() < 2 || s[0] != '-' || is_number(s.substr(1))) { pos.push_back(s); continue; }
        std::string name = s.substr(s[1] == '-' ? 2 : 1), val;
        bool has_val = false;
        size_t eq = name.find('=');
        if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); has_val = true; }
        const Opt *op = nullptr;
        for (const Opt &t : table) if (name == t.sh || name == t.lg) op = &t;
        if (!op) { fprintf(stderr, "linear filter: illegal option -- %s\n", name.c_str()); return 1; }
        if (!has_val) { if (i + 1 >= a.size()) { fprintf(stderr, "linear filter: option requires an argument -- %s\n", name.c_str()); return 1; } val = a[++i]; }
        if (op->kind == 1 && !is_number(val)) { fprintf(stderr, "linear filter: the given value '%s' cannot be casted to integer\n", val.c_str()); return 1; }
        unsigned v = op->kind == 1 ? (unsigned)strtoul(val.c_str(), nullptr, 10) : 0;
        std::string k = op->lg;
        if (k == "output") o.oPath = val; else if (k == "output_type") o.f_output_type = v; else if (k == "thread") o.thread = v; else if (k == "gap_len") o.gap_len = v;
        else if (k == "read_group") o.read_group = val; else if (k == "sample_name") o.sample_name = val; else if (k == "sequence_sam") o.sequence_sam = (int)v;
        else if (k == "duplication") o.f_dup = v; else if (k == "bal_flag") o.bal_flag = v; else if (k == "preset") o.sensitivity = v; else if (k == "index_type") o.index_t = (int)v;
        else if (k == "apx_c_flag") o.apx_chain_flag = v; else if (k == "feature_type") o.feature_t = (int)v; else if (k == "reform_ccs_cigar_flag") o.reform_ccs = v;
        else if (k == "gpus") o.gpus = v; else if (k == "block-reads") o.block_reads = v;
        else if (k == "devices") { size_t p = 0; while (p <= val.size()) { size_t q = val.find(',', p); if (q == std::string::npos) q = val.size(); if (q > p) o.devices.push_back(atoi(val.substr(p, q - p).c_str())); p = q + 1; } }
        else if (k == "index-mode") { if (val == "build") o.index_build_each = 1; else if (val != "bcast") { fprintf(stderr, "linear filter: --index-mode bcast|build\n"); return 1; } }
    }
    if (pos.size() < 2) { fprintf(stderr, "\033[1;31mE[01]:\033[0m: Please specify the files of reads and genomes\n"); return 1; }   // :291-296
    if (pos.size() == 2) { o.r_paths.push_back(pos[0]); o.g_paths.push_back(pos[1]); }
    else {                                                                                                                        // :297-319
        bool cart = false;
        for (const std::string &p : pos) { if (p == "x") cart = true; else (cart ? o.g_paths : o.r_paths).push_back(p); }
        if (!cart) { fprintf(stderr, "\033[1;31mE[02]:\033[0mPlease add '\033[1;31mx\033[0m' between files of reads and genomes.\n"); return 1; }
    }
    return 0;
}

static std::string output_prefix_of(const std::string &path) {        // getFileName(path, "/", ~0) then getFileName(.., ".", 0) (mapper.cpp:481-482)
    size_t s = path.rfind('/');
    std::string base = s == std::string::npos ? path : path.substr(s + 1);
    size_t d = base.find('.');
    return d == std::string::npos ? base : base.substr(0, d);
}

// ---- pipeline plumbing
struct Block {
    uint8_t *bases = nullptr; uint64_t cap = 0;
    std::vector<uint64_t> off, len, id_off;
    std::vector<char> ids;
    uint32_t n = 0;
    uint64_t seq = 0;          // position in the read stream
    int file = 0;              // index of the read file it came from
    lnr_cords cords{};         // host arrays of the context's result slot (valid until the worker's second next result)
    int worker = -1;
};
template <class T> struct Queue {
    std::mutex m; std::condition_variable cv; std::deque<T> q; bool closed = false;
    void push(T v) { { std::lock_guard<std::mutex> l(m); q.push_back(v); } cv.notify_one(); }
    bool pop(T &v) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !q.empty() || closed; }); if (q.empty()) return false; v = q.front(); q.pop_front(); return true; }
    bool try_pop(T &v) { std::lock_guard<std::mutex> l(m); if (q.empty()) return false; v = q.front(); q.pop_front(); return true; }
    void close() { { std::lock_guard<std::mutex> l(m); closed = true; } cv.notify_all(); }
};
struct Shared {
    Queue<Block *> free_blocks, ready;
    std::mutex m; std::condition_variable cv;
    std::map<uint64_t, Block *> done;        // finished blocks waiting for their turn at the writer
    bool workers_done = false;
    // the gap stream (see the header of this file)
    int ext = 0; uint64_t turn = 0;
    // result-slot hand-back: per worker, the sequence numbers of the blocks whose text has been written
    std::vector<uint64_t> written_upto;      // per worker: number of its blocks the writer is through with
    std::atomic<int> failed{0};
    std::string err;
    void fail(const std::string &e) { std::lock_guard<std::mutex> l(m); if (!failed) { err = e; failed = 1; } cv.notify_all(); }
};

int main(int argc, char **argv) {
    double t_start = now();
    Options o;
    int pr = parse_command_line(argc, argv, o);
    if (pr) return pr == 2 ? 0 : 1;
    fprintf(stderr, "Linear: Extensible Long-read Algorithms Framework (MI355X filter path)\n");
    for (const std::string &p : o.r_paths) if (access(p.c_str(), F_OK) == -1) { fprintf(stderr, "\033[1;31mE[05]:\033[0mCan't open file %s\n", p.c_str()); return 1; }
    for (const std::string &p : o.g_paths) if (access(p.c_str(), F_OK) == -1) { fprintf(stderr, "\033[1;31mE[06]:\033[0mCan't open file %s\n", p.c_str()); return 1; }
    // what the MI355X path does not build: say so instead of writing something else
    const char *nb = nullptr;
    if (o.f_output_type & 12) nb = "-ot 4 / 8 (BAM output)"; else if (!(o.f_output_type & 3)) nb = "-ot without 1 (.apf) or 2 (.sam)";
    else if (o.sequence_sam) nb = "-ss 1 (read sequences in the SAM)"; else if (!o.apx_chain_flag) nb = "-c 0"; else if (o.feature_t != 2) nb = "-f other than 2";
    else if (o.reform_ccs) nb = "-r 1"; else if (o.sensitivity != 1 && o.sensitivity != 2) nb = "-p other than 1 or 2"; else if (!o.bal_flag) nb = "-b 0 (the reference's -b 0 path writes a header-only SAM)";
    else if (o.index_t != 1 && o.index_t != 2) nb = "-i other than 1 or 2";
    if (nb) { fprintf(stderr, "\033[1;31mE[m02G]:\033[0m %s is not built in the MI355X filter path\n", nb); return 1; }
    if (o.thread < 1) o.thread = 1;
    if (o.gpus < 1) o.gpus = 1;
    if (o.block_reads < 1) o.block_reads = 1;
    if (o.devices.empty()) for (unsigned g = 0; g < o.gpus; g++) o.devices.push_back((int)g);
    o.gpus = (unsigned)o.devices.size();

    // ---- genomes (loadRecords over every genome file, ids cut at the first blank: base.cpp:188-195)
    std::vector<std::vector<uint8_t> > genome;
    std::vector<std::string> gid;
    {
        std::vector<uint8_t> buf((size_t)1 << 30);
        std::vector<uint64_t> off(2);
        for (const std::string &gpath : o.g_paths) {
            lnr_reader *gr = nullptr;
            if (lnr_reader_open(gpath.c_str(), &gr) != LNR_OK) { fprintf(stderr, "\033[1;31mE[06]:\033[0mCan't open file %s\n", gpath.c_str()); return 1; }
            for (;;) {
                uint32_t n = 0;
                if (lnr_reader_next(gr, buf.data(), buf.size(), off.data(), 1, &n) != LNR_OK) { fprintf(stderr, "E: genome %s: %s\n", gpath.c_str(), lnr_reader_error(gr)); return 1; }
                if (!n) break;
                genome.emplace_back(buf.begin(), buf.begin() + (long)off[1]);
                const char *ids; const uint64_t *io;
                lnr_reader_ids(gr, &ids, &io);
                std::string id(ids);
                gid.push_back(id.substr(0, id.find(' ')));
            }
            lnr_reader_close(gr);
        }
    }
    if (genome.size() >= 1024) { fprintf(stderr, "\033[1;31mE[m01G]:\033[0m Too many reference genoemes <=1024\n"); return 1; }   // linear.cpp:107-113
    if (genome.empty()) { fprintf(stderr, "E: no reference sequence in the genome files\n"); return 1; }

    // ---- contexts (one per GPU) + index
    const unsigned G = o.gpus;
    std::vector<lnr_ctx *> ctx(G, nullptr);
    for (unsigned g = 0; g < G; g++) {
        lnr_opts lo;
        lnr_opts_default(&lo);
        lo.device = o.devices[g]; lo.index_type = (uint32_t)o.index_t; lo.preset = o.sensitivity; lo.gap_len = o.gap_len; lo.dup = o.f_dup ? 1 : 0;
        lnr_status s = lnr_create(&lo, &ctx[g]);
        if (s != LNR_OK) { fprintf(stderr, "E: GPU %d: %s\n", o.devices[g], lnr_strerror(s)); return 1; }
    }
    std::vector<const uint8_t *> gp; std::vector<uint64_t> gl; std::vector<const char *> gn;
    for (size_t i = 0; i < genome.size(); i++) { gp.push_back(genome[i].data()); gl.push_back(genome[i].size()); gn.push_back(gid[i].c_str()); }
    {
        double t0 = now();
        lnr_status s = lnr_index_build(ctx[0], gp.data(), gl.data(), (uint32_t)gp.size(), o.thread);
        if (s != LNR_OK) { fprintf(stderr, "E: index: %s (%s)\n", lnr_strerror(s), lnr_last_error(ctx[0])); return 1; }
        double t_build = now() - t0;
        if (G > 1) {
            t0 = now();
            if (o.index_build_each) {
                std::vector<std::thread> th; std::vector<lnr_status> st(G, LNR_OK);
                for (unsigned g = 1; g < G; g++) th.emplace_back([&, g] { st[g] = lnr_index_build(ctx[g], gp.data(), gl.data(), (uint32_t)gp.size(), o.thread); });
                for (auto &t : th) t.join();
                for (unsigned g = 1; g < G; g++) if (st[g] != LNR_OK) { fprintf(stderr, "E: index on GPU %d: %s (%s)\n", o.devices[g], lnr_strerror(st[g]), lnr_last_error(ctx[g])); return 1; }
                fprintf(stderr, "  Index on %u more GPUs: every GPU built its own in %.3f s (the first took %.3f s)\n", G - 1, now() - t0, t_build);
            } else {
                double sec = 0;
                lnr_status sb = lnr_index_broadcast(ctx.data(), G, 0, &sec);
                if (sb != LNR_OK) { fprintf(stderr, "E: index broadcast: %s (%s)\n", lnr_strerror(sb), lnr_last_error(ctx[0])); return 1; }
                fprintf(stderr, "  Index on %u more GPUs: RCCL broadcast + derived tables in %.3f s (building it took %.3f s; --index-mode build lets every GPU build its own)\n", G - 1, sec, t_build);
            }
        }
        fprintf(stderr, "  End creating index Elapsed time[s] %.2f\n", now() - t_start);
    }
    lnr_writer *wr = nullptr;
    if (lnr_writer_create(gn.data(), gl.data(), (uint32_t)gn.size(), &wr) != LNR_OK) { fprintf(stderr, "E: writer\n"); return 1; }
    lnr_writer_set_preset(wr, o.sensitivity);
    lnr_writer_set_read_group(wr, o.read_group.c_str(), o.sample_name.c_str());

    // ---- the pipeline
    Shared sh;
    sh.written_upto.assign(G, 0);
    const unsigned NB = 4 * G + 2;
    std::vector<Block> blocks(NB);
    for (auto &b : blocks) {
        b.cap = (uint64_t)o.block_reads * 12000 + (1u << 20);
        if (b.cap > ((uint64_t)3 << 30)) b.cap = (uint64_t)3 << 30;
        b.bases = (uint8_t *)lnr_host_alloc(b.cap);
        b.off.resize((size_t)o.block_reads + 1);
        if (!b.bases) { fprintf(stderr, "E: pinned host allocation of %llu bytes failed\n", (unsigned long long)b.cap); return 1; }
        sh.free_blocks.push(&b);
    }
    std::atomic<uint64_t> total_reads{0};
    std::atomic<uint64_t> us_reader{0}, us_gpu{0}, us_writer{0};        // busy time of the three stages (microseconds), printed at the end
    const double t_reads0 = now();
    // reader: the one fetcher (parallel_io.cpp:433-485)
    std::thread reader([&] {
        uint64_t seq = 0;
        for (size_t f = 0; f < o.r_paths.size() && !sh.failed; f++) {
            lnr_reader *rr = nullptr;
            if (lnr_reader_open(o.r_paths[f].c_str(), &rr) != LNR_OK) { sh.fail("can't open read file " + o.r_paths[f]); break; }
            for (;;) {
                Block *b = nullptr;
                if (!sh.free_blocks.pop(b) || sh.failed) break;
                double tr0 = now();
                lnr_status rs_ = lnr_reader_next(rr, b->bases, b->cap, b->off.data(), o.block_reads, &b->n);
                us_reader += (uint64_t)((now() - tr0) * 1e6);
                if (rs_ != LNR_OK) { sh.fail(std::string("reads: ") + lnr_reader_error(rr)); sh.free_blocks.push(b); break; }
                if (!b->n) { sh.free_blocks.push(b); break; }
                const char *ids; const uint64_t *io;
                lnr_reader_ids(rr, &ids, &io);
                b->id_off.assign(io, io + b->n + 1);
                b->ids.assign(ids, ids + io[b->n]);
                b->len.resize(b->n);
                for (uint32_t i = 0; i < b->n; i++) b->len[i] = b->off[i + 1] - b->off[i];
                b->seq = seq++; b->file = (int)f;
                sh.ready.push(b);
            }
            lnr_reader_close(rr);
        }
        sh.ready.close();
    });
    // calculators: one per GPU.  Free-running (no gap re-mapper, or the read stream has "extended"): three blocks in flight -- lnr_filter_wait hands
    // out block k while it computes k + 1 and the upload of k + 2 runs.  Before that (-g > 0, stream not extended yet): one block at a time, strictly
    // in file order across all GPUs, each starting from the state the block before left.
    std::vector<std::thread> workers;
    for (unsigned g = 0; g < G; g++) workers.emplace_back([&, g] {
        std::deque<Block *> fl;                 // submitted, not handed out yet
        uint64_t count = 0;                     // results this context has handed out
        bool eof = false, free_run = o.gap_len == 0;
        while (!sh.failed) {
            size_t depth = free_run ? 3 : 1;
            while (!eof && fl.size() < depth) {
                Block *b = nullptr;
                if (fl.empty()) { if (!sh.ready.pop(b)) { eof = true; break; } }
                else if (!sh.ready.try_pop(b)) break;
                if (!free_run) {                 // (the context is idle here: depth 1)
                    std::unique_lock<std::mutex> l(sh.m);
                    sh.cv.wait(l, [&] { return sh.failed || sh.ext || sh.turn == b->seq; });
                    if (sh.failed) return;
                    if (sh.ext) free_run = true;
                    l.unlock();
                    if (lnr_gap_stream(ctx[g], free_run ? 1 : 0, nullptr) != LNR_OK) { sh.fail(std::string("gap stream: ") + lnr_last_error(ctx[g])); return; }
                }
                if (lnr_filter_submit(ctx[g], b->bases, b->off.data(), b->n) != LNR_OK) { sh.fail(std::string("submit: ") + lnr_last_error(ctx[g])); return; }
                fl.push_back(b);
            }
            if (fl.empty()) break;
            Block *b = fl.front();
            fl.pop_front();
            // the result slot this wait fills was handed out two results ago: the writer must be through with that block
            { std::unique_lock<std::mutex> l(sh.m); sh.cv.wait(l, [&] { return sh.failed || count < 2 || sh.written_upto[g] + 2 > count; }); if (sh.failed) return; }
            double tg0 = now();
            lnr_status s = lnr_filter_wait(ctx[g], &b->cords);
            us_gpu += (uint64_t)((now() - tg0) * 1e6);
            if (s != LNR_OK) { sh.fail(std::string("filter: ") + lnr_strerror(s) + " (" + lnr_last_error(ctx[g]) + ")"); return; }
            count++;
            if (!free_run) {
                int st = 0;
                lnr_gap_stream(ctx[g], -1, &st);
                std::lock_guard<std::mutex> l(sh.m);
                if (st) { sh.ext = 1; free_run = true; }
                sh.turn = b->seq + 1;
            }
            b->worker = (int)g;
            { std::lock_guard<std::mutex> l(sh.m); sh.done[b->seq] = b; }
            sh.cv.notify_all();
        }
    });
    // writer: the one printer, in file order (parallel_io.cpp:522-569)
    std::thread writer([&] {
        FILE *fsam = nullptr, *fapf = nullptr;
        std::string cur_prefix; bool any_open = false; int cur_file = -1;
        uint64_t want = 0;
        const char *text; uint64_t size;
        for (;;) {
            Block *b = nullptr;
            {
                std::unique_lock<std::mutex> l(sh.m);
                sh.cv.wait(l, [&] { return sh.failed || sh.done.count(want) || (sh.workers_done && sh.done.empty()); });
                if (sh.failed) break;
                auto it = sh.done.find(want);
                if (it == sh.done.end()) break;
                b = it->second; sh.done.erase(it);
            }
            if (b->file != cur_file) {                                   // p_printResults: a new output when the prefix changes (or once with -o)
                cur_file = b->file;
                std::string prefix = o.oPath.empty() ? output_prefix_of(o.r_paths[(size_t)b->file]) : o.oPath;
                bool fresh = !any_open || (o.oPath.empty() && prefix != cur_prefix);
                if (fresh) {
                    if (fsam) fclose(fsam);
                    if (fapf) fclose(fapf);
                    fsam = (o.f_output_type & 2) ? fopen((prefix + ".sam").c_str(), "wb") : nullptr;
                    fapf = (o.f_output_type & 1) ? fopen((prefix + ".apf").c_str(), "wb") : nullptr;
                    if (((o.f_output_type & 2) && !fsam) || ((o.f_output_type & 1) && !fapf)) { sh.fail("can't write output files with prefix " + prefix); break; }
                    // `@PG ... CL:` stays empty: the reference's Options constructor fills cmd_line only `if (length(argv) < 1)` (base.cpp:64-72), i.e. never
                    if (fsam) { lnr_writer_sam_header(wr, "", &text, &size); fwrite(text, 1, size, fsam); }
                    cur_prefix = prefix; any_open = true;
                }
            }
            double tw0 = now();
            if (fsam) { lnr_writer_format(wr, &b->cords, b->len.data(), b->ids.data(), b->id_off.data(), 1, o.thread, &text, &size); if (fwrite(text, 1, size, fsam) != size) { sh.fail("write error (.sam)"); break; } }
            if (fapf) { lnr_writer_format(wr, &b->cords, b->len.data(), b->ids.data(), b->id_off.data(), 2, o.thread, &text, &size); if (fwrite(text, 1, size, fapf) != size) { sh.fail("write error (.apf)"); break; } }
            us_writer += (uint64_t)((now() - tw0) * 1e6);
            total_reads += b->n;
            { std::lock_guard<std::mutex> l(sh.m); sh.written_upto[(size_t)b->worker]++; }
            sh.cv.notify_all();
            sh.free_blocks.push(b);
            want++;
        }
        if (fsam) fclose(fsam);
        if (fapf) fclose(fapf);
    });
    for (auto &t : workers) t.join();
    { std::lock_guard<std::mutex> l(sh.m); sh.workers_done = true; }
    sh.cv.notify_all();
    if (sh.failed) { sh.free_blocks.close(); sh.ready.close(); }
    writer.join();
    sh.free_blocks.close();
    reader.join();
    for (auto &b : blocks) lnr_host_free(b.bases);
    lnr_writer_destroy(wr);
    for (auto *c : ctx) lnr_destroy(c);
    if (sh.failed) { fprintf(stderr, "\033[1;31mE:\033[0m %s\n", sh.err.c_str()); return 1; }
    double dt = now() - t_start;
    const double t_reads = now() - t_reads0;
    fprintf(stderr, "  Processed: %llu reads on %u GPU%s; read files in -> output files out: %.3f s = %.0f reads/s\n", (unsigned long long)total_reads.load(), G, G > 1 ? "s" : "", t_reads,
            total_reads.load() / (t_reads > 0 ? t_reads : 1));
    fprintf(stderr, "  Stage busy time[s]: reader %.3f, GPU (upload wait + kernels + download, all calculators) %.3f, writer %.3f\n", us_reader.load() / 1e6, us_gpu.load() / 1e6, us_writer.load() / 1e6);
    fprintf(stderr, "Time in sum[s] %.2f      \n", dt);
    return 0;
}
