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

#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
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
    memset(r->tab, 4, sizeof r->tab);
    r->tab['A'] = r->tab['a'] = 0; r->tab['C'] = r->tab['c'] = 1; r->tab['G'] = r->tab['g'] = 2;
    r->tab['T'] = r->tab['t'] = r->tab['U'] = r->tab['u'] = 3;
    *out = r;
    return LNR_OK;
}

void lnr_reader_close(lnr_reader *r) {
    if (!r) return;
    if (r->f) gzclose(r->f);
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
