// lnr_output.cpp -- output shaping of the hot path: the cords of a batch -> SAM records and APF text, as the
// reference's calculator tail and printer produce them, on host threads.  Host code, no GPU involved.
//
// Restates (reference paths relative to the reference root):
//   cords2BamLink            src/f_io.cpp:899-1011   one BAM-link record per run of cords that ifCreateNew_ (f_io.cpp:674-692) keeps
//                                                     together: leading soft clip = first y, cord2cigar_ per cord, trailing soft clip
//   cord2cigar_              src/f_io.cpp:758-875    '=' / 'I' / 'D' rectangles of a cord, 'X' between non-overlapping cords, split of
//                                                     large diagonal shifts (thd_DI 80, thd_X 200: preset 1, mapper.cpp:185-186)
//   insertNewBamRecord       src/align_util.cpp:301-343   flag 16 for the reverse strand, 2048 for every record after a run ended
//   createSAZTagCigar & co.  src/align_util.cpp:452-744   SA:Z = the other records of the read as rname,pos,strand,xSyMz[ID]0S,mapq,nm;
//   writeSam                 src/f_io.cpp:313-412    the text line; MAPQ is SeqAn's default 255, RNEXT '*', PNEXT 0, TLEN 0, SEQ / QUAL '*'
//   print_cords_apf          src/f_io.cpp:100-207    '@' header per cord block + one '|' line per cord
// Parity: byte-identical to the reference's own functions on the goldens (tests/test_output_cpu.py; APF blank lines follow the
// reference's rule for a block = one call).
#include "../../include/linear_amd.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

typedef uint64_t u64;
typedef int64_t i64;

inline u64 cx(u64 v) { return (v >> 20) & ((1ULL << 30) - 1); }     // get_cord_x cords.cpp:159
inline u64 cy(u64 v) { return v & 0xfffffULL; }                     // get_cord_y
inline u64 cid(u64 v) { return (v >> 50) & 1023ULL; }               // get_cord_id
inline u64 cstrand(u64 v) { return (v >> 61) & 1ULL; }
inline bool cend(u64 v) { return (v >> 60) & 1ULL; }                // is_cord_block_end
inline u64 shift_cord(u64 v, i64 x, i64 y) { return (u64)((i64)v + (x << 20) + y); }   // cords.cpp:135-141 (packed domain)

struct Cig { char op; uint32_t n; };
struct Rec { int rid; i64 pos; unsigned flag; std::vector<Cig> cigar; };

inline void push_shrink(std::vector<Cig> &c, char op, uint32_t n) {   // appendCigarShrink f_io.cpp:659-669
    if (!c.empty() && c.back().op == op) c.back().n += n;
    else c.push_back({op, n});
}
inline void rect(u64 a, u64 b, int f_m, Cig &c1, Cig &c2) {           // createRectangleCigarPair f_io.cpp:697-718
    u64 dx = cx(b) - cx(a), dy = cy(b) - cy(a);
    c1.op = f_m ? 'X' : '=';
    if (dx >= dy) { c2.op = 'D'; c1.n = (uint32_t)dy; c2.n = (uint32_t)(dx - dy); }
    else { c2.op = 'I'; c1.n = (uint32_t)dx; c2.n = (uint32_t)(dy - dx); }
}
inline void put_pair(std::vector<Cig> &c, const Cig &c1, const Cig &c2) {
    if (c1.n) push_shrink(c, c1.op, c1.n);
    if (c2.n) push_shrink(c, c2.op, c2.n);
}

int if_create_new(u64 c1s, u64 c1e, u64 c2s, u64 thd_large_X) {      // ifCreateNew_ f_io.cpp:674-692
    u64 x11 = cx(c1s), y11 = cy(c1s), x12 = cx(c1e), y12 = cy(c1e), x21 = cx(c2s), y21 = cy(c2s);
    return cend(c1s) || x11 > x21 || y11 > y21 || ((i64)(x21 - x12) > (i64)thd_large_X && (i64)(y21 - y12) > (i64)thd_large_X) || cstrand(c1s ^ c2s);
}

u64 cord2cigar(u64 cigar_str, u64 c1s, u64 c1e, u64 c2s, std::vector<Cig> &cigar, i64 thd_DI, i64 thd_X) {   // cord2cigar_ f_io.cpp:758-875
    Cig g1, g2;
    u64 x0 = cx(cigar_str), y0 = cy(cigar_str), x11 = cx(c1s), y11 = cy(c1s), x12 = cx(c1e), y12 = cy(c1e), x21 = cx(c2s), y21 = cy(c2s);
    if (x0 - y0 != x11 - y11) return ~0ULL;
    if (x12 >= x21 && y12 >= y21) { rect(c1s, c2s, 0, g1, g2); put_pair(cigar, g1, g2); }
    else if (x12 < x21 && y12 < y21) {
        rect(c1s, c1e, 0, g1, g2); put_pair(cigar, g1, g2);
        i64 DI = (i64)(x21 - x12 - y21 + y12);
        i64 X = (i64)std::min(x21 - x12, y21 - y12);
        if (std::llabs(DI) > thd_DI && X > thd_X) {
            i64 split_n = std::min((i64)std::ceil((float)std::llabs(DI) / (float)thd_DI), X);
            i64 split_DI = thd_DI, split_X = X / split_n;
            u64 s = c1e;
            for (i64 i = 0; i < split_n - 1; i++) {
                u64 e = DI < 0 ? shift_cord(s, split_X, split_X + split_DI) : shift_cord(s, split_X + split_DI, split_X);
                rect(s, e, 0, g1, g2); put_pair(cigar, g1, g2);
                s = e;
            }
            rect(s, c2s, 1, g1, g2); put_pair(cigar, g1, g2);
        } else { rect(c1e, c2s, 1, g1, g2); put_pair(cigar, g1, g2); }
    } else { rect(c1s, c2s, 0, g1, g2); put_pair(cigar, g1, g2); }   // the two remaining cases share one body in the reference
    return c2s;
}

// cords2BamLink (single read) f_io.cpp:899-1011
void cords_to_records(const u64 *cs, const u64 *ce, u64 n, u64 L, std::vector<Rec> &recs, u64 thd_large_X, i64 thd_DI, i64 thd_X) {
    recs.clear();
    u64 cigar_str = 0;
    int f_new = 1;
    unsigned flag = 0;
    std::vector<size_t> rec_ptr, end_ptr;
    for (u64 i = 1; i < n; i++) {
        if (f_new) {
            if (i != 1) { rec_ptr.push_back(recs.size() - 1); end_ptr.push_back(i - 1); }
            f_new = 0;
            Rec r; r.rid = (int)cid(cs[i]); r.pos = (i64)cx(cs[i]); r.flag = flag | (cstrand(cs[i]) ? 16u : 0u);
            if (cy(cs[i]) != 0) r.cigar.push_back({'S', (uint32_t)cy(cs[i])});     // insertNewBamRecord align_util.cpp:325-333
            recs.push_back(std::move(r));
            cigar_str = cs[i];
            flag = 0;
        }
        u64 c1s = cs[i], c1e = ce[i], c2s;
        if (i == n - 1 || if_create_new(cs[i], ce[i], cs[i + 1], thd_large_X)) { c2s = ce[i]; f_new = 1; flag = 2048; }
        else c2s = cs[i + 1];
        cigar_str = cord2cigar(cigar_str, c1s, c1e, c2s, recs.back().cigar, thd_DI, thd_X);
        if (cigar_str == ~0ULL) break;
        if (i == n - 1) { rec_ptr.push_back(recs.size() - 1); end_ptr.push_back(n - 1); }
    }
    for (size_t k = 0; k < end_ptr.size(); k++) {
        i64 clipped = (i64)(int)(L - cy(ce[end_ptr[k]]));
        if (clipped > 0) recs[rec_ptr[k]].cigar.push_back({'S', (uint32_t)clipped});
    }
}

void put_u(std::string &s, unsigned long long v) { char b[24]; int n = snprintf(b, sizeof b, "%llu", v); s.append(b, (size_t)n); }
void put_i(std::string &s, long long v) { char b[24]; int n = snprintf(b, sizeof b, "%lld", v); s.append(b, (size_t)n); }

// SA:Z entry of one record (createSAZTagCigar / createSAZTagOneChimeric, align_util.cpp:452-520,682-714): xS yM z[I|D] 0S, zeros kept
// `first_visit`: createSAZTagCigarOneChimeric (align_util.cpp:642-678) sums NM only while the record's saz_cigar is still empty, i.e. the
// first time any line of the read lists this record; every later listing prints 0 (the cached saz_cigar is merged, nm_i_sum stays at its
// sentinel).  Line 0 therefore carries the real NM of every other record, line 1 the real NM of record 0 only, all else 0.
void saz_entry(const Rec &r, const char *gname, bool first_visit, std::string &out) {
    unsigned long long s0 = 0, cm = 0, nm = 0;
    long long ci = 0;
    for (size_t i = 0; i < r.cigar.size(); i++) {
        const Cig &c = r.cigar[i];
        if (i == 0 && c.op == 'S') s0 = c.n;
        else if (c.op == '=') cm += c.n;
        else if (c.op == 'X') { cm += c.n; nm += c.n; }
        else if (c.op == 'I') { ci -= c.n; nm += c.n; }
        else if (c.op == 'D') { ci += c.n; nm += c.n; }
        // (the reference's branch for a trailing 'S' tests `i == length(cigar[i]) - 1`, i.e. i == 0: never taken, the last count stays 0)
    }
    out += gname; out += ',';
    put_i(out, r.pos + 1); out += ',';
    out += (r.flag & 16) ? '-' : '+'; out += ',';
    put_u(out, s0); out += 'S';
    put_u(out, (unsigned)cm); out += 'M';
    put_u(out, (unsigned)std::llabs(ci)); out += ci < 0 ? 'I' : 'D';
    out += "0S,255,";
    put_i(out, first_visit ? (int)nm : 0); out += ';';
}

struct Writer {
    std::vector<std::string> gid;
    std::vector<u64> glen;
    std::string text;
    u64 thd_large_X = 8000; i64 thd_DI = 80, thd_X = 200;      // mapper.cpp:465,185-186 (preset 1)
    std::string rg, sn;                                        // -rg / -sn
};

void sam_read(const Writer &w, const u64 *cs, const u64 *ce, u64 n, u64 L, const char *qname, std::string &out, std::vector<Rec> &recs) {
    cords_to_records(cs, ce, n, L, recs, w.thd_large_X, w.thd_DI, w.thd_X);
    std::vector<char> saz_done(recs.size(), 0);       // "saz_cigar not empty" per record (fillBamRecordLinkRecords walks the heads in record order)
    for (size_t it = 0; it < recs.size(); it++) {
        const Rec &r = recs[it];
        const char *g = (size_t)r.rid < w.gid.size() ? w.gid[(size_t)r.rid].c_str() : "*";
        out += qname; out += '\t';
        put_u(out, r.flag); out += '\t';
        out += g; out += '\t';
        put_i(out, r.pos + 1); out += "\t255\t";
        if (r.cigar.empty()) out += '*';
        for (const Cig &c : r.cigar) { put_u(out, c.n); out += c.op; }
        out += "\t*\t0\t0\t*\t*";
        if (recs.size() > 1) {                         // SA:Z: every other line of the read, in record order (createSAZTagOneLine)
            out += "\tSA:Z:";
            for (size_t j = 0; j < recs.size(); j++)
                if (j != it) {
                    saz_entry(recs[j], (size_t)recs[j].rid < w.gid.size() ? w.gid[(size_t)recs[j].rid].c_str() : "*", !saz_done[j], out);
                    saz_done[j] = 1;
                }
        }
        out += '\n';
    }
}

void apf_read(const Writer &w, const u64 *c, u64 n, u64 L, const char *rid, bool blank_before, std::string &out) {   // print_cords_apf f_io.cpp:100-207
    if (n == 0) return;
    int fflag = 0;
    for (u64 j = 1; j < n; j++) {
        if (cend(c[j - 1])) {
            u64 m = j; int main_cnt = 0, block_len = 0;
            while (m < n && !cend(c[m])) { if (cstrand(c[m])) main_cnt++; block_len++; m++; }
            char main_icon = main_cnt > block_len / 2 ? '-' : (main_cnt == block_len / 2 ? (cstrand(c[j]) ? '-' : '+') : '+');
            u64 r_end = 0, s_end = 0;
            for (u64 i = j;; i++)
                if (cend(c[i]) || i == n - 1) { r_end = cy(c[i]) + 96; s_end = cx(c[i]) + 96; break; }
            if (blank_before) out += '\n';
            u64 g = cid(c[j]);
            out += "@ "; out += rid; out += ' ';
            put_u(out, L); out += ' ';
            put_u(out, cy(c[j])); out += ' ';
            put_u(out, std::min(r_end, L)); out += ' ';
            out += main_icon; out += ' ';
            out += g < w.gid.size() ? w.gid[g].c_str() : "*"; out += ' ';
            put_u(out, g < w.glen.size() ? w.glen[g] : 0); out += ' ';
            put_u(out, cx(c[j])); out += ' ';
            put_u(out, s_end); out += '\n';
            fflag = 1;
        }
        i64 d1 = 0, d2 = 0;
        if (!fflag) { d1 = (i64)(cx(c[j]) - cx(c[j - 1])); d2 = (i64)(cy(c[j]) - cy(c[j - 1])); }
        out += "| ";
        put_u(out, cy(c[j])); out += ' ';
        put_u(out, cx(c[j])); out += ' ';
        put_i(out, d2); out += ' ';
        put_i(out, d1); out += ' ';
        out += cstrand(c[j]) ? '-' : '+';
        out += '\n';
        fflag = 0;
    }
}

}  // namespace

struct lnr_writer { Writer w; };

extern "C" {

lnr_status lnr_writer_create(const char *const *genome_ids, const uint64_t *genome_len, uint32_t nseq, lnr_writer **out) {
    if (!genome_ids || !genome_len || !out) return LNR_ERR_ARG;
    lnr_writer *p = new (std::nothrow) lnr_writer();
    if (!p) return LNR_ERR_NOMEM;
    for (uint32_t i = 0; i < nseq; i++) { p->w.gid.emplace_back(genome_ids[i] ? genome_ids[i] : "*"); p->w.glen.push_back(genome_len[i]); }
    *out = p;
    return LNR_OK;
}
void lnr_writer_destroy(lnr_writer *w) { delete w; }

// what: 1 = SAM records, 2 = APF.  Reads are formatted on `threads` host threads and concatenated in read order.
lnr_status lnr_writer_format(lnr_writer *wr, const lnr_cords *cords, const uint64_t *read_len, const char *read_ids, const uint64_t *id_off,
                             int what, uint32_t threads, const char **text, uint64_t *size) {
    if (!wr || !cords || !read_len || !read_ids || !id_off || !text || !size || (what != 1 && what != 2)) return LNR_ERR_ARG;
    const Writer &w = wr->w;
    uint32_t n = cords->n_reads;
    if (threads < 1) threads = 1;
    if (threads > n) threads = n ? n : 1;
    std::vector<std::string> part(threads);
    auto work = [&](uint32_t t) {
        uint32_t lo = (uint32_t)((u64)n * t / threads), hi = (uint32_t)((u64)n * (t + 1) / threads);
        std::vector<Rec> recs;
        std::string &o = part[t];
        for (uint32_t k = lo; k < hi; k++) {
            u64 a = cords->cord_off[k], e = cords->cord_off[k + 1];
            const char *id = read_ids + id_off[k];
            if (what == 1) sam_read(w, cords->cords_str + a, cords->cords_end + a, e - a, read_len[k], id, o, recs);
            else apf_read(w, cords->cords_str + a, e - a, read_len[k], id, k > 0, o);
        }
    };
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < threads; t++) th.emplace_back(work, t);
    work(0);
    for (auto &t : th) t.join();
    wr->w.text.clear();
    for (auto &p : part) wr->w.text += p;
    *text = wr->w.text.data();
    *size = wr->w.text.size();
    return LNR_OK;
}

// SAM header as `linear filter` writes it: @SQ per reference sequence, then @RG and @PG (no @HD)
lnr_status lnr_writer_sam_header(lnr_writer *wr, const char *command_line, const char **text, uint64_t *size) {
    if (!wr || !text || !size) return LNR_ERR_ARG;
    std::string &o = wr->w.text;
    o.clear();
    for (size_t i = 0; i < wr->w.gid.size(); i++) { o += "@SQ\tSN:"; o += wr->w.gid[i]; o += "\tLN:"; put_u(o, wr->w.glen[i]); o += '\n'; }
    o += "@RG\tID:"; o += wr->w.rg; o += "\tSM:"; o += wr->w.sn; o += "\n@PG\tID:M1-3\tPN:Linear\tCL:";
    o += command_line ? command_line : "";
    o += '\n';
    *text = o.data(); *size = o.size();
    return LNR_OK;
}

lnr_status lnr_writer_set_preset(lnr_writer *wr, uint32_t preset) {
    if (!wr || preset > 2) return LNR_ERR_ARG;
    if (preset == 1) { wr->w.thd_DI = 80; wr->w.thd_X = 200; }                     // mapper.cpp:181-186
    else { wr->w.thd_DI = ((i64)1 << 60) - 1; wr->w.thd_X = ((i64)1 << 60) - 1; }  // FIOParms::FIOParms f_io.cpp:14-22
    return LNR_OK;
}
lnr_status lnr_writer_set_read_group(lnr_writer *wr, const char *read_group, const char *sample_name) {
    if (!wr) return LNR_ERR_ARG;
    wr->w.rg = read_group ? read_group : ""; wr->w.sn = sample_name ? sample_name : "";
    return LNR_OK;
}

}  // extern "C"
