// ref_harness.cpp -- thin C-API shim over the *real* reference (xp3i4/linear).
//
// TEST INFRASTRUCTURE.  Compiled by oracle/Makefile together with the
// reference's own translation units, in place under /root/reference, into
// oracle/_ref/libref_linear.so (git-ignored; never committed, never shipped
// with the product).  It mirrors the C API of oracle/lnr_oracle.cpp so the
// tests and tools/make_golden.py can run the reference and the restatement
// side by side on identical inputs.  Own code: it only *calls* the reference
// (createIndexDynamic index_util.h:297, createFeatures pmpfinder.h:196,
// getDIndexMatchAll pmpfinder.cpp:1856, apxMap pmpfinder.h:213).
#include "base.h"
#include "cords.h"
#include "shape_extend.h"
#include "index_util.h"
#include "cluster_util.h"
#include "pmpfinder.h"
#include "gap.h"
#include <seqan/seq_io.h>
#include "f_io.h"
#include <fstream>
#include <sstream>
#include <omp.h>
#include <vector>
#include <cstring>

using namespace seqan;

// external-linkage functions of pmpfinder.cpp that have no header declaration
unsigned getDIndexMatchAll(DIndex &index, String<Dna5> &read, String<uint64_t> &set, uint64_t read_str, uint64_t read_end, PMPParms &pm_pmp);
uint64_t filterAnchors(Anchors &anchors, uint64_t shape_len, uint64_t thd_anchor_accept_density, uint64_t thd_anchor_accept_min,
                       unsigned thd_anchor_err_bit, uint64_t thd_max_anchors_num, uint64_t thd_anchor_accept_err, int alg_type);
int chainAnchorsHits(String<uint64_t> &anchors, String<uint64_t> &hits, String<int> &hits_chains_score, PMPParms &pm_pmp);

unsigned getHIndexMatchAll(LIndex &index, String<Dna5> &read, String<uint64_t> &set, uint64_t map_str, uint64_t map_end, PMPParms &pm_pmp);   // pmpfinder.cpp:1918

// external-linkage functions of gap_util.cpp that have no header declaration (unit hooks for the gap path restatement)
int g_stream_(String<Dna5> &seq1, String<Dna5> &seq2, String<uint64_t> &g_hs, uint64_t gap_str, uint64_t gap_end, unsigned shape_len, int step1, int step2, GapParms &gap_parms);
int g_create_anchors_(String<uint64_t> &g_hs, String<uint64_t> &g_hs_anchor, int shape_len, int direction, int64_t anchor_lower, int64_t anchor_upper, uint64_t rvcp_const, uint64_t gap_str,
                      uint64_t gap_end, GapParms &gap_parms);
int g_CreateExtendAnchorsPair_(String<uint64_t> &g_hs, String<uint64_t> &g_hs_anchor1, String<uint64_t> &g_hs_anchor2, int shape_len, uint64_t rvcp_const, uint64_t gap_str1, uint64_t gap_end1,
                               uint64_t gap_str2, uint64_t gap_end2, GapParms &gap_parms);
int c_createAnchors2(String<uint64_t> &g_hs, String<uint64_t> &g_anchors, int g_hs_end, int64_t anchor_lower, int64_t anchor_upper);
int dropChainGapX(String<uint64_t> &chains, uint64_t (*getX)(uint64_t), uint64_t (*getY)(uint64_t), int direction, bool f_erase, GapParms &gap_parms);
uint64_t g_hs_anchor_getX(uint64_t val);
uint64_t g_hs_anchor_getY(uint64_t val);
int g_CreateChainsFromAnchors_(String<uint64_t> &anchors, String<uint64_t> &tiles, uint64_t &gap_str, uint64_t &gap_end, uint64_t read_len, GapParms &gap_parms);
std::pair<int, int> getClosestExtensionChain_(String<uint64_t> &tmp_tiles, uint64_t gap_str, uint64_t gap_end, bool f_erase_tiles, GapParms &gap_parms);

namespace {
const size_t PAD = 64;

// Build a String<Dna5> whose allocator slack behind end() is zero ('A'), so the
// reference's reads past the end (shape_extend.cpp:294, pmpfinder.cpp:637-647)
// are deterministic and equal to the pin documented in lnr_oracle.cpp.
void assign_padded(String<Dna5> &s, const uint8_t *p, uint64_t n) {
    resize(s, n + PAD, Dna5(0));
    for (uint64_t i = 0; i < n; i++) s[i] = Dna5(p[i]);
    for (uint64_t i = n; i < n + PAD; i++) s[i] = Dna5(0);
    resize(s, n);   // shrinking keeps the storage (and the zeroed slack)
}

struct RefCtx {
    StringSet<String<Dna5> > g;
    IndexDynamic *idx;
    StringSet<FeaturesDynamic> f2;
    unsigned T;
    // per-read scratch (mapper.cpp:423-433)
    Anchors anchors;
    String<uint64_t> hit;
    String<UPair> gaps;
    String<Dna5> com;
    StringSet<FeaturesDynamic> f1;
    PMPParms pm;
    GlobalParms pg;
    String<uint64_t> cs, ce;
};
}  // namespace

extern "C" {

static void *ref_create_i(const uint8_t *const *seqs, const uint64_t *lens, uint32_t nseq, uint32_t T, int index_type);
void *ref_create(const uint8_t *const *seqs, const uint64_t *lens, uint32_t nseq, uint32_t T) { return ref_create_i(seqs, lens, nseq, T, 1); }
// index_type as on the command line: 1 = DIndex (default), 2 = HIndex (mapper.cpp:200, index_util.cpp:2436-2448)
void *ref_create2(const uint8_t *const *seqs, const uint64_t *lens, uint32_t nseq, uint32_t T, int index_type) { return ref_create_i(seqs, lens, nseq, T, index_type); }
static void *ref_create_i(const uint8_t *const *seqs, const uint64_t *lens, uint32_t nseq, uint32_t T, int index_type) {
    RefCtx *c = new RefCtx();
    c->T = T ? T : 1;
    omp_set_num_threads(c->T);
    resize(c->g, nseq);
    for (uint32_t i = 0; i < nseq; i++) assign_padded(c->g[i], seqs[i], lens[i]);
    createFeatures(c->g, c->f2, 2, c->T);                     // linear.cpp:76 (process3)
    c->idx = new IndexDynamic(c->g);
    c->idx->setIndexType(index_type);                         // -i 1 -> DIndex, -i 2 -> HIndex
    createIndexDynamic(c->g, *c->idx, 0, length(c->g), c->T, false);
    c->pm.pm_cah.thd_stop_chain_len_ratio = 0;                // default preset -p 1 (mapper.cpp:181-185)
    resize(c->f1, 2);
    c->f1[0].init(2);
    c->f1[1].init(2);
    return c;
}
void ref_destroy(void *h) { RefCtx *c = (RefCtx *)h; delete c->idx; delete c; }
uint64_t ref_dir_len(void *h) { return length(((RefCtx *)h)->idx->dindex.getDir()); }
uint64_t ref_hs_len(void *h) { return length(((RefCtx *)h)->idx->dindex.getHs()); }
const int32_t *ref_dir(void *h) { return (const int32_t *)&(((RefCtx *)h)->idx->dindex.getDir()[0]); }
const uint64_t *ref_hs(void *h) { RefCtx *c = (RefCtx *)h; return length(c->idx->dindex.getHs()) ? &(c->idx->dindex.getHs()[0]) : nullptr; }
uint64_t ref_ysa_len(void *h) { return length(((RefCtx *)h)->idx->hindex.ysa); }
const uint64_t *ref_ysa(void *h) { RefCtx *c = (RefCtx *)h; return length(c->idx->hindex.ysa) ? &(c->idx->hindex.ysa[0]) : nullptr; }
uint64_t ref_empty_dir(void *h) { return ((RefCtx *)h)->idx->hindex.emptyDir; }
uint64_t ref_f2_len(void *h, uint32_t id) { return length(((RefCtx *)h)->f2[id].fs2_48); }
const int32_t *ref_f2(void *h, uint32_t id) { return (const int32_t *)&(((RefCtx *)h)->f2[id].fs2_48[0]); }

uint64_t ref_read_features(const uint8_t *read, uint64_t len, int strand, int32_t *out, uint64_t cap) {
    String<Dna5> r, com;
    assign_padded(r, read, len);
    FeaturesDynamic f(2);
    if (strand) {
        _compltRvseStr(r, com);
        createFeatures(begin(com), end(com), f);
    } else createFeatures(begin(r), end(r), f);
    uint64_t n = length(f.fs2_48);
    if (n && cap) memcpy(out, &f.fs2_48[0], (n < cap ? n : cap) * 12);
    return n;
}

uint64_t ref_seed_lookup(void *h, const uint8_t *read, uint64_t len, uint64_t read_str, uint64_t read_end, int alpha,
                         uint64_t *out, uint64_t cap, uint64_t *stats4) {
    RefCtx *c = (RefCtx *)h;
    String<Dna5> r;
    assign_padded(r, read, len);
    String<uint64_t> set;
    appendValue(set, 0);
    PMPParms pm;
    pm.pm_gdima.thd_alpha = alpha;
    if (c->idx->isHIndex()) { pm.pm_ghima.thd_alpha = alpha; getHIndexMatchAll(c->idx->hindex, r, set, read_str, create_cord(MAX_CORD_ID, MAX_CORD_X, read_end, 0), pm); }
    else getDIndexMatchAll(c->idx->dindex, r, set, read_str, read_end, pm);
    uint64_t n = length(set);
    if (out && n) memcpy(out, &set[0], (n < cap ? n : cap) * 8);
    if (stats4) { stats4[0] = stats4[1] = stats4[2] = 0; stats4[3] = n - 1; }
    return n;
}

uint64_t ref_map_read(void *h, const uint8_t *read, uint64_t len) {
    RefCtx *c = (RefCtx *)h;
    String<Dna5> r;
    assign_padded(r, read, len);
    clear(c->cs);
    clear(c->ce);
    if (len <= 200) return 0;                                 // mapper.cpp:430,440
    String<CordInfo> ci;
    _compltRvseStr(r, c->com);
    // keep the slack behind the reverse complement deterministic as well
    { uint64_t n = length(c->com); resize(c->com, n + PAD, Dna5(0)); resize(c->com, n); }
    createFeatures(begin(r), end(r), c->f1[0]);
    createFeatures(begin(c->com), end(c->com), c->f1[1]);
    apxMap(*c->idx, r, c->anchors, c->hit, c->f1, c->f2, c->gaps, c->cs, c->ce, ci, 1, c->pg, c->pm);
    return length(c->cs);
}
// apxMap + the gap re-mapper as Mapper::p_calRecords runs them for `-g gap_len [-dup f_dup]` (mapper.cpp:207-231,438-453): SURVEY 8 f1,
// not built on the GPU yet -- this entry point makes its goldens.
uint64_t ref_map_read_g2(void *h, const uint8_t *read, uint64_t len, uint32_t gap_len, int f_dup, int *ext_state);
uint64_t ref_map_read_g(void *h, const uint8_t *read, uint64_t len, uint32_t gap_len, int f_dup) { return ref_map_read_g2(h, read, len, gap_len, f_dup, nullptr); }
// ext_state (optional, in / out): the stream state the read starts from and leaves behind (see ref_map_batch_g2)
uint64_t ref_map_read_g2(void *h, const uint8_t *read, uint64_t len, uint32_t gap_len, int f_dup, int *ext_state) {
    uint64_t n0 = ref_map_read(h, read, len);
    RefCtx *c = (RefCtx *)h;
    if (len <= 200 || gap_len == 0) return n0;
    String<Dna5> r;
    assign_padded(r, read, len);
    GapParms gp(0.2);
    gp.f_dup = f_dup;
    gp.thd_gap_len_min = gap_len == 1 ? 50 : (gap_len < 10 ? 10 : gap_len);
    gp.read_id = "read";
    if (ext_state && *ext_state) gp.thd_cts_major_limit = 3;
    String<uint64_t> clips;
    mapGaps(c->g, r, c->com, c->cs, c->ce, clips, c->gaps, c->f1, c->f2, gp);
    CordsParms cp;
    reformCords(c->cs, c->ce, &reformCordsDxDy1, cp);
    if (ext_state) *ext_state = gp.thd_cts_major_limit == 3;
    return length(c->cs);
}

// ---- unit hooks for the restatement of the gap path (oracle/lnr_gap.inc): single reference functions on raw arrays
static uint64_t out_u64(String<uint64_t> &v, uint64_t *out, uint64_t cap) { uint64_t n = length(v); for (uint64_t i = 0; i < n && i < cap; i++) out[i] = v[i]; return n; }
uint64_t ref_gap_anchors(const uint8_t *g, uint64_t glen, const uint8_t *r, uint64_t rlen, uint64_t gap_str, uint64_t gap_end, int shape_len, int step1, int step2, int direction,
                         int64_t anchor_lower, int64_t anchor_upper, uint64_t rvcp_const, uint64_t *out, uint64_t cap) {
    String<Dna5> s1, s2; assign_padded(s1, g, glen); assign_padded(s2, r, rlen);
    GapParms gp(0.2);
    String<uint64_t> g_hs, anc;
    g_stream_(s1, s2, g_hs, gap_str, gap_end, (unsigned)shape_len, step1, step2, gp);
    g_create_anchors_(g_hs, anc, shape_len, direction, anchor_lower, anchor_upper, rvcp_const, gap_str, gap_end, gp);
    return out_u64(anc, out, cap);
}
uint64_t ref_gap_anchor_pair(const uint8_t *g, uint64_t glen, const uint8_t *r, uint64_t rlen, uint64_t gs, uint64_t ge, int shape_len, int step1, int step2, uint64_t rvcp_const,
                             uint64_t gap_str1, uint64_t gap_end1, uint64_t gap_str2, uint64_t gap_end2, uint64_t *out1, uint64_t *n1, uint64_t *out2, uint64_t cap) {
    String<Dna5> s1, s2; assign_padded(s1, g, glen); assign_padded(s2, r, rlen);
    GapParms gp(0.2);
    String<uint64_t> g_hs, a1, a2;
    g_stream_(s1, s2, g_hs, gs, ge, (unsigned)shape_len, step1, step2, gp);
    g_CreateExtendAnchorsPair_(g_hs, a1, a2, shape_len, rvcp_const, gap_str1, gap_end1, gap_str2, gap_end2, gp);
    *n1 = out_u64(a1, out1, cap);
    return out_u64(a2, out2, cap);
}
uint64_t ref_gap_canchors(const uint8_t *g, uint64_t glen, const uint8_t *r, uint64_t rlen, uint64_t s1s, uint64_t s1e, uint64_t s2s, uint64_t s2e, int step1, int step2, int shape_len,
                          int64_t anchor_lower, int64_t anchor_upper, uint64_t *out, uint64_t cap) {
    String<Dna5> s1, s2; assign_padded(s1, g, glen); assign_padded(s2, r, rlen);
    String<uint64_t> g_hs, anc;
    c_stream_(s1, g_hs, s1s, s1e, step1, shape_len, 0);
    c_stream_(s2, g_hs, s2s, s2e, step2, shape_len, 1);
    c_createAnchors2(g_hs, anc, (int)length(g_hs), anchor_lower, anchor_upper);
    return out_u64(anc, out, cap);
}

// alt != 0: the chain metrics as mapGap_ sets them for its indel branch (gap.cpp:123-130)
static void gp_alt(GapParms &gp, int alt) {          // bit 0: mapGap_'s swapped chain metrics (gap.cpp:124-130); bit 1: thd_cts_major_limit 3 (stream state "extended")
    if (alt & 2) gp.thd_cts_major_limit = 3;
    if (!(alt & 1)) return;
    gp.chn_score1.thd_min_chain_len = 1; gp.chn_score1.thd_abort_score = 0; gp.chn_score1.getScore = &getGapAnchorsChainScore2;
    gp.chn_score2.thd_abort_score = 0; gp.chn_score2.getScore2 = &getGapBlocksChainScore3;
}
uint64_t ref_gap_chains(const uint64_t *anchors, uint64_t n, uint64_t read_len, int alt, int direction, uint64_t gap_str, uint64_t gap_end, int closest, uint64_t *out, uint64_t cap, int *pr) {
    String<uint64_t> a, tiles; resize(a, n); for (uint64_t i = 0; i < n; i++) a[i] = anchors[i];
    GapParms gp(0.2); gp_alt(gp, alt); gp.direction = direction;
    g_CreateChainsFromAnchors_(a, tiles, gap_str, gap_end, read_len, gp);
    if (closest) { std::pair<int, int> r = getClosestExtensionChain_(tiles, gap_str, gap_end, closest == 2, gp); pr[0] = r.first; pr[1] = r.second; }
    return out_u64(tiles, out, cap);
}

// mapGeneric (which 1), mapExtend (2, direction), mapExtends (3) of one read against the context's genome; tiles_str then tiles_end
uint64_t ref_gap_map(void *h, const uint8_t *read, uint64_t len, int which, uint64_t gs1, uint64_t ge1, uint64_t gs2, uint64_t ge2, int direction, int alt, uint64_t *out_str, uint64_t *out_end,
                     uint64_t *n2, uint64_t cap) {
    RefCtx *c = (RefCtx *)h;
    String<Dna5> r; assign_padded(r, read, len);
    _compltRvseStr(r, c->com);
    { uint64_t n = length(c->com); resize(c->com, n + PAD, Dna5(0)); resize(c->com, n); }
    createFeatures(begin(r), end(r), c->f1[0]);
    createFeatures(begin(c->com), end(c->com), c->f1[1]);
    GapParms gp(0.2); gp_alt(gp, alt);
    gp.read_len = len; gp.ref_len = length(c->g[get_cord_id(gs1)]);
    String<uint64_t> ts1, te1, ts2, te2;
    if (which == 1) mapGeneric(c->g, r, c->com, c->f1, c->f2, ts1, te1, gs1, ge1, gp);
    else if (which == 2) mapExtend(c->g, r, c->com, c->f1, c->f2, ts1, te1, gs1, ge1, direction, gp);
    else mapExtends(c->g, r, c->com, c->f1, c->f2, ts1, te1, ts2, te2, gs1, ge1, gs2, ge2, 0, gp);
    uint64_t n1 = length(ts1);
    for (uint64_t i = 0; i < n1 && i < cap; i++) { out_str[i] = ts1[i]; out_end[i] = i < length(te1) ? te1[i] : 0; }
    *n2 = length(ts2);
    for (uint64_t i = 0; i < length(ts2) && n1 + i < cap; i++) { out_str[n1 + i] = ts2[i]; out_end[n1 + i] = i < length(te2) ? te2[i] : 0; }
    return n1 | ((uint64_t)length(te1) << 32);
}
int ref_gap_score(int which, uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t read_len, int strand) {
    ChainScoreParms p; p.chn_block_strand = strand;
    switch (which) {
        case 1: return getGapAnchorsChainScore(a, b, p);
        case 2: return getGapAnchorsChainScore2(a, b, p);
        case 3: return getGapBlocksChainScore2(a, b, c, d, read_len, p);
        default: return getGapBlocksChainScore3(a, b, c, d, read_len, p);
    }
}
uint64_t ref_gap_xdrop(uint64_t *chain, uint64_t n, int direction, int f_erase, int *ret) {
    String<uint64_t> c; resize(c, n); for (uint64_t i = 0; i < n; i++) c[i] = chain[i];
    GapParms gp(0.2);
    *ret = dropChainGapX(c, &g_hs_anchor_getX, &g_hs_anchor_getY, direction, f_erase != 0, gp);
    for (uint64_t i = 0; i < length(c); i++) chain[i] = c[i];
    return length(c);
}
uint64_t ref_get_gaps(void *h, uint64_t *out_pairs, uint64_t cap_pairs) {   // apx_gaps of the last ref_map_read
    RefCtx *c = (RefCtx *)h;
    for (uint64_t i = 0; i < length(c->gaps) && i < cap_pairs; i++) { out_pairs[2 * i] = c->gaps[i].first; out_pairs[2 * i + 1] = c->gaps[i].second; }
    return length(c->gaps);
}
void ref_get_cords(void *h, uint64_t *cords_str, uint64_t *cords_end) {
    RefCtx *c = (RefCtx *)h;
    uint64_t n = length(c->cs);
    if (n) { memcpy(cords_str, &c->cs[0], n * 8); memcpy(cords_end, &c->ce[0], n * 8); }
}

// The reference's calculator loop over a block of reads (Mapper::p_calRecords, mapper.cpp:404-473 with -g 0): an OpenMP team,
// per-thread scratch exactly as mapper.cpp:423-433 declares it (anchors, crhit, f1, apx_gaps, comStr and a private parameter
// copy, because toggle() mutates it: mapper.cpp:233-237,447).  Used as the CPU baseline of bench.py (`kind: "reference"`).
// Returns the total number of cords; cord_off[n+1] always, cords only while they fit `cap`.
uint64_t ref_map_batch_g(void *h, const uint8_t *reads, const uint64_t *off, uint32_t n, int threads, uint64_t *cord_off,
                         uint64_t *cords_str, uint64_t *cords_end, uint64_t cap, uint32_t gap_len, int f_dup);
uint64_t ref_map_batch(void *h, const uint8_t *reads, const uint64_t *off, uint32_t n, int threads, uint64_t *cord_off,
                       uint64_t *cords_str, uint64_t *cords_end, uint64_t cap) {
    return ref_map_batch_g(h, reads, off, n, threads, cord_off, cords_str, cords_end, cap, 0, 0);
}
// the calculator loop with the gap re-mapper (-g gap_len [-dup f_dup]) behind apxMap, as Mapper::p_calRecords runs it (mapper.cpp:207-231).
// GapParms: ONE per thread for the whole call, as the Mapper keeps one per thread for the whole run (mapper.cpp:233-237, used :447) --
// mapExtend / mapExtends leave it modified (gap_util.cpp:4046-4071,4088-4119), so with more than one thread the result of a read can
// depend on which reads its thread met before (in the reference itself).
//   ext_state == NULL : exactly that, `threads` threads, dynamic schedule (the CPU baseline of bench.py).
//   ext_state != NULL : the program's `-t 1` result (file order through one GapParms) computed on `threads` threads: reads are taken one by
//                       one through ONE GapParms until its thd_cts_major_limit flips to 3 (the only leaked field that is read before it is
//                       written again; see oracle/lnr_oracle.cpp orc_map_read_g2), the rest in parallel on copies of that object.
//                       *ext_state in (0 fresh stream, 1 already extended) / out.
static GapParms harness_gap_parms(uint32_t gap_len, int f_dup) {
    GapParms gp(0.2);
    gp.f_dup = f_dup;
    gp.thd_gap_len_min = gap_len == 1 ? 50 : (gap_len < 10 ? 10 : gap_len);     // mapper.cpp:207-231
    gp.read_id = "read";
    return gp;
}
uint64_t ref_map_batch_g2(void *h, const uint8_t *reads, const uint64_t *off, uint32_t n, int threads, uint64_t *cord_off,
                          uint64_t *cords_str, uint64_t *cords_end, uint64_t cap, uint32_t gap_len, int f_dup, int *ext_state) {
    RefCtx *c = (RefCtx *)h;
    std::vector<String<uint64_t> > CS(n), CE(n);
    if (threads < 1) threads = 1;
    GapParms gp0 = harness_gap_parms(gap_len, f_dup);
    if (ext_state && *ext_state) gp0.thd_cts_major_limit = 3;
    uint32_t i0 = 0;
    struct Scratch {
        Anchors anchors; String<uint64_t> hit; String<UPair> gaps; String<Dna5> com, r; StringSet<FeaturesDynamic> f1; PMPParms pm; GlobalParms pg;
        Scratch(RefCtx *c) : pm(c->pm), pg(c->pg) { resize(f1, 2); f1[0].init(2); f1[1].init(2); }
    };
    auto one = [&](Scratch &S, uint32_t i, GapParms &gp) {
        uint64_t len = off[i + 1] - off[i];
        if (len <= 200) return;                         // mapper.cpp:430,440
        assign_padded(S.r, reads + off[i], len);
        String<CordInfo> ci;
        _compltRvseStr(S.r, S.com);
        { uint64_t m = length(S.com); resize(S.com, m + PAD, Dna5(0)); resize(S.com, m); }
        createFeatures(begin(S.r), end(S.r), S.f1[0]);
        createFeatures(begin(S.com), end(S.com), S.f1[1]);
        apxMap(*c->idx, S.r, S.anchors, S.hit, S.f1, c->f2, S.gaps, CS[i], CE[i], ci, 1, S.pg, S.pm);
        if (gap_len) {
            String<uint64_t> clips;
            mapGaps(c->g, S.r, S.com, CS[i], CE[i], clips, S.gaps, S.f1, c->f2, gp);
            CordsParms cp;
            reformCords(CS[i], CE[i], &reformCordsDxDy1, cp);
        }
    };
    if (ext_state && gap_len) {
        Scratch S(c);
        while (i0 < n && gp0.thd_cts_major_limit != 3) one(S, i0++, gp0);
    }
#pragma omp parallel num_threads(threads)
    {
        Scratch S(c);
        GapParms gp = gp0;
#pragma omp for schedule(dynamic, 16)
        for (uint32_t i = i0; i < n; i++) one(S, i, gp);
    }
    if (ext_state) *ext_state = gp0.thd_cts_major_limit == 3;
    uint64_t tot = 0;
    cord_off[0] = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint64_t k = length(CS[i]);
        if (k && tot + k <= cap) { memcpy(cords_str + tot, &CS[i][0], k * 8); memcpy(cords_end + tot, &CE[i][0], k * 8); }
        tot += k;
        cord_off[i + 1] = tot;
    }
    return tot;
}
uint64_t ref_map_batch_g(void *h, const uint8_t *reads, const uint64_t *off, uint32_t n, int threads, uint64_t *cord_off,
                         uint64_t *cords_str, uint64_t *cords_end, uint64_t cap, uint32_t gap_len, int f_dup) {
    return ref_map_batch_g2(h, reads, off, n, threads, cord_off, cords_str, cords_end, cap, gap_len, f_dup, nullptr);
}

// The reference's reader: SeqAn readRecords on a SeqFileIn, the call of its fetcher (src/parallel_io.cpp:433-485), into
// StringSet<String<Dna5>>.  Pins the product's own FASTA / FASTQ reader (linear_amd/csrc/lnr_reader.cpp): same records, same
// ordinals, same ids.  Returns the number of records; bases back to back, off[n+1]; ids '\n'-separated.
uint64_t ref_read_file(const char *path, uint8_t *bases, uint64_t cap, uint64_t *off, uint64_t max_n, char *ids, uint64_t ids_cap) {
    SeqFileIn fin;
    if (!open(fin, path)) return ~0ULL;
    StringSet<CharString> idset;
    StringSet<String<Dna5> > seqs;
    uint64_t n = 0, used = 0, iu = 0;
    off[0] = 0;
    try {
        while (!atEnd(fin)) {
            clear(idset); clear(seqs);
            readRecords(idset, seqs, fin, 100);
            for (unsigned i = 0; i < length(seqs) && n < max_n; i++) {
                uint64_t L = length(seqs[i]);
                if (used + L > cap) return ~1ULL;
                for (uint64_t k = 0; k < L; k++) bases[used + k] = (uint8_t)ordValue(seqs[i][k]);
                used += L;
                off[++n] = used;
                uint64_t il = length(idset[i]);
                if (iu + il + 1 > ids_cap) return ~2ULL;
                memcpy(ids + iu, toCString(idset[i]), il); iu += il; ids[iu++] = '\n';
            }
        }
    } catch (...) { return ~3ULL; }
    if (iu < ids_cap) ids[iu] = 0;
    return n;
}

// Output shaping by the reference's own functions: the calculator's tail (mapper.cpp:463-470: cords2BamLink with
// thd_large_X 8000 and the preset-1 thd_DI 80 / thd_X 200, mapper.cpp:185-186, then fillBamRecords) and its printers
// (printAlignSamBam -> writeSam, f_io.cpp:313-412,540-648; print_cords_apf, f_io.cpp:100-207).  SAM (header records built as
// mapper.cpp:292-320 builds them, written by the reference's printAlignSamHeader) goes to sam_path, APF to apf_path.
int ref_format(void *h, const uint8_t *reads, const uint64_t *off, uint32_t n, const char *read_ids_nl, const char *genome_ids_nl,
               const char *cmd_line, const char *sam_path, const char *apf_path) {
    RefCtx *c = (RefCtx *)h;
    StringSet<String<Dna5> > rds;
    StringSet<CharString> rids, gids;
    StringSet<String<uint64_t> > cs, ce;
    StringSet<String<CordInfo> > cinfo;
    { std::stringstream ss(read_ids_nl); std::string l; while (std::getline(ss, l)) appendValue(rids, CharString(l.c_str())); }
    { std::stringstream ss(genome_ids_nl); std::string l; while (std::getline(ss, l)) appendValue(gids, CharString(l.c_str())); }
    if (length(rids) != n || length(gids) != length(c->g)) return -1;
    resize(rds, n); resize(cs, n); resize(ce, n); resize(cinfo, n);
    for (uint32_t i = 0; i < n; i++) {
        uint64_t len = off[i + 1] - off[i];
        assign_padded(rds[i], reads + off[i], len);
        if (len <= 200) continue;                                 // mapper.cpp:430,440
        _compltRvseStr(rds[i], c->com);
        { uint64_t m = length(c->com); resize(c->com, m + PAD, Dna5(0)); resize(c->com, m); }
        createFeatures(begin(rds[i]), end(rds[i]), c->f1[0]);
        createFeatures(begin(c->com), end(c->com), c->f1[1]);
        apxMap(*c->idx, rds[i], c->anchors, c->hit, c->f1, c->f2, c->gaps, cs[i], ce[i], cinfo[i], 1, c->pg, c->pm);
    }
    FIOParms fio;
    fio.thd_DI = 80; fio.thd_X = 200;
    fio.f_output_type = 0;
    fp_handler_.setPrintSam(fio.f_output_type);
    StringSet<String<BamAlignmentRecordLink> > bam;
    cords2BamLink(cs, ce, cinfo, bam, rds, 96, 8000, fio.thd_DI, fio.thd_X);
    fillBamRecords(c->g, rds, gids, rids, bam, fio);
    {   // header records as Mapper's constructor assembles them (mapper.cpp:292-320; read group / sample name default to "")
        BamHeaderRecord hr;
        for (unsigned i = 0; i < length(c->g); i++) {
            clear(hr); hr.type = seqan::BAM_HEADER_REFERENCE;
            setTagValue("SN", gids[i], hr); setTagValue("LN", std::to_string(length(c->g[i])), hr);
            appendValue(fio.bam_header, hr);
        }
        clear(hr); hr.type = seqan::BAM_HEADER_READ_GROUP; setTagValue("ID", "", hr); setTagValue("SM", "", hr); appendValue(fio.bam_header, hr);
        clear(hr); hr.type = seqan::BAM_HEADER_PROGRAM; setTagValue("ID", "M1-3", hr); setTagValue("PN", "Linear", hr); setTagValue("CL", cmd_line, hr);
        appendValue(fio.bam_header, hr);
    }
    { std::ofstream of(sam_path); printAlignSamBam(c->g, rds, gids, rids, bam, of, 1, fio); }
    { std::ofstream of(apf_path); print_cords_apf(cs, c->g, rds, gids, rids, of); }
    return 0;
}

// stage dumps reproduced by calling the reference's own stage functions in apxMap_'s order
// (pmpfinder.cpp:2646-2652, 2520-2526): 0 raw anchors, 1 filtered anchors, 2 x-desc sorted anchors, 3 hits after anchor chaining
uint64_t ref_stage(void *h, const uint8_t *read, uint64_t len, int stage, uint64_t *out, uint64_t cap) {
    RefCtx *c = (RefCtx *)h;
    String<Dna5> r;
    assign_padded(r, read, len);
    Anchors anchors;
    anchors.init(1);
    PMPParms pm;
    pm.pm_cah.thd_stop_chain_len_ratio = 0;
    getDIndexMatchAll(c->idx->dindex, r, anchors.set, 0, len, pm);
    String<uint64_t> *res = &anchors.set;
    String<uint64_t> hits;
    String<int> hits_score;
    if (stage >= 1) filterAnchors(anchors, c->pg.shape_len, 1, 2, 2, 5, 2500, 2);
    if (stage >= 2) {
        initHits(hits);
        initHitsScore(hits_score);
        chainAnchorsHits(anchors.set, hits, hits_score, pm);   // sorts anchors.set in place
        if (stage == 3) res = &hits;
    }
    uint64_t n = length(*res);
    if (out && n) memcpy(out, &(*res)[0], (n < cap ? n : cap) * 8);
    return n;
}
}  // extern "C"
