// lnr_reader.cpp -- input side of the hot path: FASTA / FASTQ records, plain or gzip, decoded straight into the
// byte layout the C ABI takes (SeqAn Dna5 ordinals, reads back to back + offsets), so that a front-end fills a pinned block
// (lnr_host_alloc) and hands it to lnr_filter_submit without another copy.  Host code, no GPU involved.
//
// Replaces, for the filter path, what the reference's fetcher does per block (src/parallel_io.cpp:433-485: SeqAn
// readRecords(ids, reads, SeqFileIn, n) into StringSet<String<Dna5>>):
//   * format by the first record character: '>' FASTA, '@' FASTQ (seqan/seq_io/fasta_fastq.h); gzip detected by zlib;
//   * the id is the whole header line without its marker (reads keep it whole, mapper.cpp / base.cpp:188-195 cut genome ids at
//     the first blank -- lnr_reader_next_ids returns the whole line, the caller cuts);
//   * sequence characters convert as SeqAn's char -> Dna5 table does (basic/alphabet_residue_tabs.h:107-140): A/a 0, C/c 1,
//     G/g 2, T/t/U/u 3, everything else N = 4; blanks and line ends inside a record are skipped; multi-line FASTA and
//     multi-line FASTQ (quality length = sequence length) are accepted.
#include "../../include/linear_amd.h"

// Plain (not gzip) files take a PARALLEL path (f4's reason to exist: feed a GPU that filters 2 M reads/s): the file is mapped, one pass of
// memchr finds the record boundaries of the next block (a FASTA record ends before the next '>' at a line start; a FASTQ record is taken as four
// lines -- anything else, e.g. multi-line FASTQ, hands the rest of the file to the serial parser below), then `threads` host threads count and
// convert the bases of their share of the records straight into the caller's (pinned) block.  Same records, same ordinals, same ids as the serial
// parser (tests/test_reader_cpu.py runs both on every fixture).  A gzip file is one inflate stream and stays serial.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

struct lnr_reader {
    gzFile f = nullptr;
    std::vector<unsigned char> buf;
    size_t pos = 0, end = 0;
    bool eof = false;
    int format = 0;              // 0 unknown, 1 FASTA, 2 FASTQ
    std::string err;
    std::vector<char> ids;       // ids of the last block, '\0' separated
    std::vector<uint8_t> spill;  // a record that did not fit the last block any more (a gzip stream cannot be rewound): first record of the next
    std::vector<char> spill_id;
    bool have_spill = false;
    std::vector<uint64_t> id_off;
    uint64_t records = 0, bases = 0;
    unsigned char tab[256];
    // the mapped file of the parallel path (plain files only)
    const unsigned char *map = nullptr; size_t map_len = 0, mpos = 0; bool use_map = false; unsigned threads = 8;

    bool fill() {
        if (eof) return false;
        if (pos < end) return true;
        int n = gzread(f, buf.data(), (unsigned)buf.size());
        if (n < 0) { int e; err = gzerror(f, &e); eof = true; return false; }
        if (n == 0) { eof = true; return false; }
        pos = 0; end = (size_t)n;
        return true;
    }
    int peek() { return fill() ? buf[pos] : -1; }
    int get() { return fill() ? buf[pos++] : -1; }
    // appends the rest of the current line (without CR / LF) to s (or drops it when s is null); returns false at end of file with nothing read
    bool line(std::vector<char> *s) {
        bool any = false;
        while (fill()) {
            any = true;
            unsigned char *p = buf.data() + pos, *e = buf.data() + end;
            unsigned char *nl = (unsigned char *)memchr(p, '\n', (size_t)(e - p));
            unsigned char *stop = nl ? nl : e;
            if (s) s->insert(s->end(), (char *)p, (char *)stop);
            pos = (size_t)(stop - buf.data());
            if (nl) { pos++; break; }
        }
        if (s) while (!s->empty() && s->back() == '\r') s->pop_back();
        return any;
    }
};

extern "C" {

lnr_status lnr_reader_open(const char *path, lnr_reader **out) {
    if (!path || !out) return LNR_ERR_ARG;
    *out = nullptr;
    lnr_reader *r = new (std::nothrow) lnr_reader();
    if (!r) return LNR_ERR_NOMEM;
    r->f = gzopen(path, "rb");
    if (!r->f) { delete r; return LNR_ERR_ARG; }
    gzbuffer(r->f, 1u << 20);
    r->buf.resize(4u << 20);
    if (!getenv("LNR_READER_SERIAL")) {
        int fd = open(path, O_RDONLY);
        struct stat st;
        unsigned char magic[2] = {0, 0};
        if (fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 2 && pread(fd, magic, 2, 0) == 2 && !(magic[0] == 0x1f && magic[1] == 0x8b)) {
            void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) { r->map = (const unsigned char *)m; r->map_len = (size_t)st.st_size; r->use_map = true; (void)madvise(m, r->map_len, MADV_SEQUENTIAL); }
        }
        if (fd >= 0) close(fd);
        unsigned hw = std::thread::hardware_concurrency();
        r->threads = hw ? (hw < 16 ? hw : 16) : 4;
        if (const char *e = getenv("LNR_READER_THREADS")) { int v = atoi(e); if (v >= 1 && v <= 256) r->threads = (unsigned)v; }
    }
    memset(r->tab, 4, sizeof r->tab);
    r->tab['A'] = r->tab['a'] = 0; r->tab['C'] = r->tab['c'] = 1; r->tab['G'] = r->tab['g'] = 2;
    r->tab['T'] = r->tab['t'] = r->tab['U'] = r->tab['u'] = 3;
    *out = r;
    return LNR_OK;
}

void lnr_reader_close(lnr_reader *r) {
    if (!r) return;
    if (r->f) gzclose(r->f);
    if (r->map) munmap((void *)r->map, r->map_len);
    delete r;
}

const char *lnr_reader_error(const lnr_reader *r) { return r ? r->err.c_str() : "null reader"; }

// Next block of records: at most max_reads records and never more than dst_cap bases (a record that does not fit any more is left
// for the next call; one that could never fit is LNR_ERR_LIMIT).  off[0] = 0 .. off[*n_out] written.  *n_out == 0 at end of file.
lnr_status lnr_reader_next(lnr_reader *r, uint8_t *dst, uint64_t dst_cap, uint64_t *off, uint32_t max_reads, uint32_t *n_out) {
    if (!r || !dst || !off || !n_out) return LNR_ERR_ARG;
    *n_out = 0;
    off[0] = 0;
    r->ids.clear(); r->id_off.assign(1, 0);
    uint64_t used = 0;
    uint32_t n = 0;
    if (r->have_spill) {
        if (r->spill.size() > dst_cap) { r->err = "a record is longer than the block"; return LNR_ERR_LIMIT; }
        memcpy(dst, r->spill.data(), r->spill.size());
        used = r->spill.size();
        r->ids.insert(r->ids.end(), r->spill_id.begin(), r->spill_id.end());
        n = 1; off[1] = used; r->id_off.push_back(r->ids.size());
        r->have_spill = false; r->spill.clear(); r->spill_id.clear();
    }
    if (r->use_map && !r->have_spill) {
        struct Rec { size_t hdr, hdr_end, seq, seq_end; uint64_t nb; };
        std::vector<Rec> recs;
        const unsigned char *M = r->map;
        const size_t ML = r->map_len;
        auto is_ws = [](unsigned char c) { return c == '\n' || c == '\r' || c == ' ' || c == '\t'; };
        size_t pos = r->mpos, fallback_at = (size_t)-1;
        uint64_t cum = 0;
        while (n + recs.size() < max_reads) {
            while (pos < ML && is_ws(M[pos])) pos++;
            if (pos >= ML) break;
            if (r->format == 0) r->format = M[pos] == '>' ? 1 : (M[pos] == '@' ? 2 : -1);
            if (r->format < 0 || M[pos] != (r->format == 1 ? '>' : '@')) { fallback_at = pos; break; }     // (the serial parser reports it)
            Rec q;
            q.hdr = pos + 1;
            const unsigned char *nl = (const unsigned char *)memchr(M + q.hdr, '\n', ML - q.hdr);
            q.hdr_end = nl ? (size_t)(nl - M) : ML;
            q.seq = q.hdr_end < ML ? q.hdr_end + 1 : ML;
            size_t next;
            if (r->format == 1) {
                size_t p = q.seq;
                for (;;) {
                    const unsigned char *g = p < ML ? (const unsigned char *)memchr(M + p, '>', ML - p) : nullptr;
                    if (!g) { q.seq_end = ML; break; }
                    if ((size_t)(g - M) == q.seq || g[-1] == '\n') { q.seq_end = (size_t)(g - M); break; }
                    p = (size_t)(g - M) + 1;
                }
                next = q.seq_end;
            } else {
                const unsigned char *e2 = q.seq < ML ? (const unsigned char *)memchr(M + q.seq, '\n', ML - q.seq) : nullptr;
                if (!e2 || (size_t)(e2 - M) + 1 >= ML || e2[1] != '+') { fallback_at = pos; break; }             // not the four-line form
                const unsigned char *e3 = (const unsigned char *)memchr(e2 + 1, '\n', ML - (size_t)(e2 + 1 - M));
                if (!e3) { fallback_at = pos; break; }
                const unsigned char *qs = e3 + 1;
                const unsigned char *e4 = (size_t)(qs - M) < ML ? (const unsigned char *)memchr(qs, '\n', ML - (size_t)(qs - M)) : nullptr;
                const unsigned char *qe = e4 ? e4 : M + ML;
                size_t sl = (size_t)(e2 - (M + q.seq)), ql = (size_t)(qe - qs);
                while (sl && M[q.seq + sl - 1] == '\r') sl--;
                while (ql && qs[ql - 1] == '\r') ql--;
                bool clean = sl == ql;
                for (size_t i = 0; clean && i < sl; i++) clean = !is_ws(M[q.seq + i]);
                for (size_t i = 0; clean && i < ql; i++) clean = !is_ws(qs[i]);
                if (!clean) { fallback_at = pos; break; }
                q.seq_end = q.seq + sl;
                next = (size_t)(qe - M);
            }
            uint64_t span = q.seq_end - q.seq;
            if (used + cum + span > dst_cap) {                       // the block may be full: decide by the record's exact number of bases
                uint64_t nb = 0;
                for (size_t i = q.seq; i < q.seq_end; i++) nb += !is_ws(M[i]);
                if (used + cum + nb > dst_cap) {
                    if (n + recs.size() == 0) { r->err = "a record is longer than the block"; return LNR_ERR_LIMIT; }
                    break;
                }
            }
            cum += span;
            recs.push_back(q);
            pos = next;
        }
        const size_t R = recs.size();
        if (R) {
            unsigned T = r->threads < R ? r->threads : (unsigned)R;
            if (cum < (1u << 20)) T = 1;
            auto share = [&](unsigned t, size_t &a, size_t &b) { a = R * t / T; b = R * (t + 1) / T; };
            auto run = [&](auto fn) {
                std::vector<std::thread> th;
                for (unsigned t = 1; t < T; t++) th.emplace_back(fn, t);
                fn(0u);
                for (auto &x : th) x.join();
            };
            run([&](unsigned t) {                                     // pass 1: bases per record
                size_t a, b; share(t, a, b);
                for (size_t k = a; k < b; k++) {
                    uint64_t nb = 0;
                    const unsigned char *p = M + recs[k].seq, *e = M + recs[k].seq_end;
                    if (r->format == 2) nb = (uint64_t)(e - p);       // (a clean single line)
                    else for (; p < e; p++) nb += !is_ws(*p);
                    recs[k].nb = nb;
                }
            });
            for (size_t k = 0; k < R; k++) { off[n + k + 1] = off[n + k] + recs[k].nb; }
            if (off[n + R] > dst_cap) { r->err = "internal: block accounting"; return LNR_ERR_INTERNAL; }
            const unsigned char *tab = r->tab;
            run([&](unsigned t) {                                     // pass 2: convert into the caller's block
                size_t a, b; share(t, a, b);
                for (size_t k = a; k < b; k++) {
                    uint8_t *d = dst + off[n + k];
                    const unsigned char *p = M + recs[k].seq, *e = M + recs[k].seq_end;
                    if (r->format == 2) { for (; p < e; p++) *d++ = tab[*p]; }
                    else for (; p < e; p++) { unsigned char ch = *p; if (!is_ws(ch)) *d++ = tab[ch]; }
                }
            });
            for (size_t k = 0; k < R; k++) {
                size_t he = recs[k].hdr_end;
                while (he > recs[k].hdr && M[he - 1] == '\r') he--;
                r->ids.insert(r->ids.end(), (const char *)M + recs[k].hdr, (const char *)M + he);
                r->ids.push_back('\0');
                r->id_off.push_back(r->ids.size());
                r->bases += recs[k].nb;
            }
            r->records += R;
            n += (uint32_t)R;
            used = off[n];
        }
        r->mpos = pos;
        if (fallback_at != (size_t)-1) {                              // the serial parser takes over from this record on
            r->use_map = false;
            gzseek(r->f, (z_off_t)fallback_at, SEEK_SET);
            r->pos = r->end = 0; r->eof = false;
        } else { *n_out = n; return LNR_OK; }
    }
    while (n < max_reads) {
        int c;
        while ((c = r->peek()) == '\n' || c == '\r' || c == ' ' || c == '\t') r->pos++;     // blank lines between records
        if (c < 0) break;
        if (r->format == 0) r->format = c == '>' ? 1 : (c == '@' ? 2 : -1);
        if (r->format < 0 || c != (r->format == 1 ? '>' : '@')) { r->err = "record does not start with '>' / '@'"; return LNR_ERR_ARG; }
        r->pos++;
        size_t id_start = r->ids.size();
        r->line(&r->ids);
        r->ids.push_back('\0');
        uint64_t start = used, len = 0;
        bool spilling = false;
        // bases of one line (or what is left of it) -> dst, or -> the spill buffer once the block is full
        auto seq_line = [&]() {
            while (r->fill()) {
                unsigned char *p = r->buf.data() + r->pos, *e = r->buf.data() + r->end;
                bool eol = false;
                for (; p < e; p++) {
                    unsigned char ch = *p;
                    if (ch == '\n') { eol = true; p++; break; }
                    if (ch == '\r' || ch == ' ' || ch == '\t') continue;
                    if (!spilling && used < dst_cap) dst[used++] = r->tab[ch];
                    else {
                        if (!spilling) { r->spill.assign(dst + start, dst + used); used = start; spilling = true; }
                        r->spill.push_back(r->tab[ch]);
                    }
                    len++;
                }
                r->pos = (size_t)(p - r->buf.data());
                if (eol) break;
            }
        };
        if (r->format == 1) {
            while ((c = r->peek()) >= 0 && c != '>') seq_line();
        } else {
            while ((c = r->peek()) >= 0 && c != '+') seq_line();   // sequence lines up to the '+' line ...
            if (c == '+') r->line(nullptr);
            uint64_t got = 0;                                      // ... then as many quality characters as there were bases
            while (got < len && r->fill()) {
                unsigned char ch = r->buf[r->pos++];
                if (ch == '\n' || ch == '\r' || ch == ' ' || ch == '\t') continue;
                got++;
            }
            if (got < len) { r->err = "FASTQ record with fewer qualities than bases"; return LNR_ERR_ARG; }
            if (len) r->line(nullptr);                             // rest of the last quality line
        }
        r->records++;
        r->bases += len;
        if (spilling) {                                            // keep it for the next block
            r->spill_id.assign(r->ids.begin() + (long)id_start, r->ids.end());
            r->ids.resize(id_start);
            r->have_spill = true;
            if (n == 0 && r->spill.size() > dst_cap) { r->err = "a record is longer than the block"; return LNR_ERR_LIMIT; }
            break;
        }
        n++;
        off[n] = used;
        r->id_off.push_back(r->ids.size());
    }
    *n_out = n;
    return LNR_OK;
}

// ids of the last block: *ids = '\0'-separated header lines, id_off[k] = start of id k (n + 1 entries)
lnr_status lnr_reader_ids(const lnr_reader *r, const char **ids, const uint64_t **id_off) {
    if (!r || !ids || !id_off) return LNR_ERR_ARG;
    *ids = r->ids.data();
    *id_off = r->id_off.data();
    return LNR_OK;
}

}  // extern "C"
