// host_shim.cpp -- TEST-ONLY host build of the product's stage logic (linear_amd/csrc/lnr_hd.h,
// ref_sort.h) so that the device code can be checked against the oracle on a machine
// without a GPU.  It is compiled by tests/ into tests/_build/libhost_shim.so and is never
// part of the shipped library: the product path runs these same functions inside HIP
// kernels only (linear_amd/csrc/lnr_kernels.hip).
#include <memory>
#include <vector>
#include <algorithm>
#include <cstring>
#include "../linear_amd/csrc/lnr_hd.h"
#include "../linear_amd/csrc/lnr_gap_hd.h"

using namespace lnr;

namespace {
const size_t PAD = 64;
struct Shim {
    std::vector<std::vector<u8>> seqs;
    std::vector<u64> lens;
    u32 T;
    std::vector<i32> dir;
    std::vector<u64> hs;
    std::vector<F96> f2;
    std::vector<u64> f2_off;
    std::vector<u64> cs, ce;
    u64 stats[5] = {0, 0, 0, 0, 0};
    std::vector<u64> dbg[4];
};

void features_closed(const u8 *s, u64 n_entries, F96 *out) {
    for (u64 m = 0; m < n_entries; m++) {
        i32 w0 = 0, w1 = 0, w2 = 0;
        for (u64 j = 16 * m; j < 16 * m + 48; j++) add2mer(w0, w1, w2, s[j], s[j + 1]);
        out[m].v0 = w0; out[m].v1 = w1; out[m].v2 = w2; out[m].pad = 0;
    }
}

void build_index(Shim &c) {
    const size_t full = ((size_t)1 << 26) + 1;
    std::vector<u32> Xs;
    std::vector<u64> vals;
    for (size_t i = 0; i < c.seqs.size(); i++) {
        const u8 *s = c.seqs[i].data();
        for (u32 t = 0; t < c.T; t++) {
            i64 t_str, t_end;
            chunk_bounds(c.lens[i], c.T, t, t_str, t_end);
            if (t_str >= t_end) continue;
            u64 ns = chunk_num_samples(t_str, t_end);
            if (!ns) continue;
            int ks = shape_init_skip(s + t_str);
            int C = shape_const(s, (u64)t_str, ks, (u64)t_str);
            u64 run_start = 0;
            u32 prevX = 0;
            for (u64 m = 0; m < ns; m++) {
                u64 j = (u64)t_str + 8 + 9 * m;
                SeedOut o = seed_sample(s, j, (u64)t_str, (u64)t_str, ks, C);
                if (m == 0 || o.X != prevX) run_start = m;
                prevX = o.X;
                bool rec = ((m - run_start) & 1) == 0;   // recorded iff even position inside its run of equal minimizers
                if (rec) { Xs.push_back(o.X); vals.push_back(create_cord(i, j + ANCHOR_ZERO, o.Y, o.strand)); }
            }
        }
    }
    c.dir.assign(full, 0);
    for (u32 x : Xs) c.dir[x]++;
    i64 sum = 0;
    for (size_t i = 0; i < full; i++) {
        if (c.dir[i] > 400) c.dir[i] = 0;
        sum += c.dir[i];
        c.dir[i] = (i32)(sum - c.dir[i]);
    }
    c.hs.assign((size_t)sum, 0);
    std::vector<i32> fill(full, 0);
    for (size_t k = 0; k < Xs.size(); k++) {
        u32 x = Xs[k];
        if (c.dir[x + 1] - c.dir[x]) c.hs[c.dir[x] + fill[x]++] = vals[k];
    }
    for (size_t i = 0; i + 1 < full; i++)
        if (c.dir[i + 1] > c.dir[i]) std::sort(c.hs.begin() + c.dir[i], c.hs.begin() + c.dir[i + 1]);
}

void seed_lookup(Shim &c, const u8 *read, u64 L, u64 read_str, u64 read_end, int alpha, std::vector<u64> &a) {
    int ks = shape_init_skip(read);
    u64 k0 = read_str + 21;
    int C = shape_const(read, 0, ks, k0);
    u32 ns = seed_num_samples(read_str, read_end, (u32)alpha);
    u32 xprev = 0;
    for (u32 s = 0; s < ns; s++) {
        u64 k = k0 + alpha - 1 + (u64)alpha * s;
        SeedOut o = seed_sample(read, k, k0, 0, ks, C);
        c.stats[0]++;
        if (o.X != xprev) {
            c.stats[1]++;
            c.stats[2] += (u64)(c.dir[o.X + 1] - c.dir[o.X]);
            for (i32 i = c.dir[o.X]; i < c.dir[o.X + 1]; i++)
                if (y_match(cord_y(c.hs[i]), o.Y)) { a.push_back(val2anchor(c.hs[i], k, L, o.strand)); c.stats[3]++; }
        }
        xprev = o.X;
    }
}

int run_job(Shim &c, const u8 *read, u64 L, u64 read_str, u64 read_end, int mode, const FeatView f1[2], Vec<u64> &cords, bool dbg) {
    std::vector<u64> a;
    a.push_back(0);
    seed_lookup(c, read, L, read_str, read_end, job_parm(mode).alpha, a);
    if (dbg) c.dbg[0] = a;
    u32 n = (u32)a.size();
    u32 cap = n + 2;
    a.resize(cap);
    u64 maxlen = 0;
    for (u64 l : c.lens) maxlen = std::max(maxlen, l);
    u32 nbins = (u32)((maxlen + (2ULL << 20)) / 30000 + 2);
    std::vector<u16> bins(nbins, 0);
    n = binning_filter_serial(a.data(), n, bins.data(), nbins);
    if (n > 1) { a[0] = 0; std::sort(a.begin(), a.begin() + n); }
    std::vector<char> scratch(job_scratch_bytes(cap));
    Arena ar; ar.init(scratch.data(), scratch.size());
    JobCtx jc;
    jc.L = L; jc.read_str = read_str; jc.read_end = read_end; jc.mode = mode;
    jc.f1[0] = f1[0]; jc.f1[1] = f1[1];
    jc.g.base = c.f2.data(); jc.g.off = c.f2_off.data(); jc.g.nseq = (u32)c.seqs.size();
    jc.bins = bins.data(); jc.nbins = nbins; jc.pair_evals = &c.stats[4]; jc.prof = nullptr; jc.traceback_done = 0;
    JobDebug jd; memset(&jd, 0, sizeof(jd));
    std::vector<u64> d1(cap), d2(cap), d3(cap + 2);
    u32 n1 = 0, n2 = 0, n3 = 0;
    if (dbg) { jd.filt = d1.data(); jd.nfilt = &n1; jd.xsort = d2.data(); jd.nxsort = &n2; jd.hits_chain = d3.data(); jd.nhits_chain = &n3; }
    int ovf = 0;
    JobScratch S;
    LeaderScratch ls;
    u32 m = job_phase1(a.data(), n, dbg ? &jd : nullptr, ls);
    if (!job_carve(ar, m, S, &ovf)) return 1;
    job_fill_xy(a.data(), m, S, 0, 1);
    if (m >= 2) best_chains_serial(S.xs, S.ys, m, S.rec, job_parm(mode).score_type, jc.pair_evals);
    u64 *H = nullptr; u32 nH = 0;
    int rc = job_phase3a(a.data(), m, S, jc, dbg ? &jd : nullptr, H, nH, ls);
    if (!rc && nH >= 2) {
        filter_hits_flags(H, nH, jc.f1, jc.g, S.cnt, 0, 1);
        nH = filter_hits_apply(H, nH, S.cnt);
        path_dst_2(H, nH, jc.f1, jc.g, cords, read_str, read_end, L);
    }
    if (ovf || *cords.ovf) rc = 1;
    if (dbg) { c.dbg[1].assign(d1.begin(), d1.begin() + n1); c.dbg[2].assign(d2.begin(), d2.begin() + n2); c.dbg[3].assign(d3.begin(), d3.begin() + n3); }
    return rc;
}

int map_read(Shim &c, const u8 *read_in, u64 L, bool dbg) {
    c.cs.clear(); c.ce.clear();
    for (auto &d : c.dbg) d.clear();
    if (L <= 200) return 0;
    std::vector<u8> rd(L + PAD, 0), rc(L + PAD, 0);
    memcpy(rd.data(), read_in, L);
    static const u8 cpl[5] = {3, 2, 1, 0, 4};
    for (u64 k = 0; k < L; k++) rc[k] = cpl[rd[L - 1 - k]];
    u32 nf = read_feature_count(L);
    std::vector<F96> f1a(nf), f1b(nf);
    features_closed(rd.data(), nf, f1a.data());
    features_closed(rc.data(), nf, f1b.data());
    FeatView f1[2];
    f1[0].p = f1a.data(); f1[0].n = nf; f1[1].p = f1b.data(); f1[1].n = nf;
    u32 cap_c = (u32)(16 * (L / 64) + 256);
    std::vector<u64> cords(cap_c);
    int ovf = 0;
    Vec<u64> cv; cv.init(cords.data(), cap_c, &ovf);
    if (run_job(c, rd.data(), L, 0, L, 0, f1, cv, dbg)) return -1;
    std::vector<char> scratch(tail_scratch_bytes(cap_c));
    Arena ar; ar.init(scratch.data(), scratch.size());
    std::vector<UP> gaps(cap_c);
    u32 ngaps = 0, remap = 0, nc = cv.n;
    LeaderScratch ls;
    if (tail_a(cords.data(), nc, L, ar, gaps.data(), cap_c, ngaps, remap, ls)) return -2;
    cv.n = nc;
    if (remap) {
        for (u32 i = 0; i < ngaps; i++) {
            UP y = forward_y(gaps[i], L);
            if (run_job(c, rd.data(), L, y.first, y.second, 1, f1, cv, false)) return -3;
        }
    }
    ar.init(scratch.data(), scratch.size());
    std::vector<u64> os(cap_c), oe(cap_c);
    u32 nout = 0;
    if (tail_b(cords.data(), cv.n, L, ar, os.data(), oe.data(), cap_c, nout, ls)) return -4;
    c.cs.assign(os.begin(), os.begin() + nout);
    c.ce.assign(oe.begin(), oe.begin() + nout);
    return 0;
}
}  // namespace

extern "C" {
void *hs_create(const u8 *const *seqs, const u64 *lens, u32 nseq, u32 T) {
    Shim *c = new Shim();
    c->T = T ? T : 1;
    c->f2_off.push_back(0);
    for (u32 i = 0; i < nseq; i++) {
        c->seqs.emplace_back(lens[i] + PAD, 0);
        memcpy(c->seqs.back().data(), seqs[i], lens[i]);
        c->lens.push_back(lens[i]);
        c->f2_off.push_back(c->f2_off.back() + genome_feature_count(lens[i]));
    }
    c->f2.resize(c->f2_off.back());
    for (u32 i = 0; i < nseq; i++) features_closed(c->seqs[i].data(), c->f2_off[i + 1] - c->f2_off[i], c->f2.data() + c->f2_off[i]);
    build_index(*c);
    return c;
}
void hs_destroy(void *h) { delete (Shim *)h; }
u64 hs_dir_len(void *h) { return ((Shim *)h)->dir.size(); }
u64 hs_hs_len(void *h) { return ((Shim *)h)->hs.size(); }
const i32 *hs_dir(void *h) { return ((Shim *)h)->dir.data(); }
const u64 *hs_hs(void *h) { return ((Shim *)h)->hs.data(); }
u64 hs_f2_len(void *h, u32 id) { Shim *c = (Shim *)h; return c->f2_off[id + 1] - c->f2_off[id]; }
void hs_f2(void *h, u32 id, i32 *out) {
    Shim *c = (Shim *)h;
    for (u64 k = c->f2_off[id]; k < c->f2_off[id + 1]; k++) { *out++ = c->f2[k].v0; *out++ = c->f2[k].v1; *out++ = c->f2[k].v2; }
}
u64 hs_read_features(const u8 *read, u64 L, int strand, i32 *out, u64 cap) {
    std::vector<u8> s(L + PAD, 0);
    static const u8 cpl[5] = {3, 2, 1, 0, 4};
    if (strand) for (u64 k = 0; k < L; k++) s[k] = cpl[read[L - 1 - k]];
    else memcpy(s.data(), read, L);
    u32 nf = read_feature_count(L);
    std::vector<F96> f(nf);
    features_closed(s.data(), nf, f.data());
    for (u32 k = 0; k < nf && k < cap; k++) { out[3 * k] = f[k].v0; out[3 * k + 1] = f[k].v1; out[3 * k + 2] = f[k].v2; }
    return nf;
}
u64 hs_seed_lookup(void *h, const u8 *read, u64 L, u64 read_str, u64 read_end, int alpha, u64 *out, u64 cap, u64 *stats4) {
    Shim *c = (Shim *)h;
    std::vector<u8> s(L + PAD, 0);
    memcpy(s.data(), read, L);
    std::vector<u64> a;
    a.push_back(0);
    memset(c->stats, 0, sizeof(c->stats));
    seed_lookup(*c, s.data(), L, read_str, read_end, alpha, a);
    if (out) memcpy(out, a.data(), std::min<u64>(a.size(), cap) * 8);
    if (stats4) for (int i = 0; i < 4; i++) stats4[i] = c->stats[i];
    return a.size();
}
long hs_map_read(void *h, const u8 *read, u64 L, int dbg) {
    Shim *c = (Shim *)h;
    int rc = map_read(*c, read, L, dbg != 0);
    if (rc) return rc;
    return (long)c->cs.size();
}
void hs_get_cords(void *h, u64 *cs, u64 *ce) {
    Shim *c = (Shim *)h;
    if (!c->cs.empty()) { memcpy(cs, c->cs.data(), c->cs.size() * 8); memcpy(ce, c->ce.data(), c->ce.size() * 8); }
}
u64 hs_debug_get(void *h, int stage, u64 *out, u64 cap) {
    Shim *c = (Shim *)h;
    if (stage < 0 || stage > 3) return 0;
    if (out) memcpy(out, c->dbg[stage].data(), std::min<u64>(c->dbg[stage].size(), cap) * 8);
    return c->dbg[stage].size();
}
void hs_get_stats(void *h, u64 *out5) { memcpy(out5, ((Shim *)h)->stats, 40); }


// ---- the product's gap path (lnr_gap_hd.h) on the host, hook for hook like the oracle's orc_gap_* (tests/test_gap_shim_cpu.py)
namespace {
struct GapHost {
    static const size_t M1 = (size_t)64 << 20;
    std::unique_ptr<char[]> mem; GArena ar; LeaderScratch ls; std::vector<u8> g, r, c; GapCtx X;
    GapHost(const u8 *gp_, u64 glen, const u8 *rp, u64 rlen) : mem(new char[M1]), g(glen + PAD, 0), r(rlen + PAD, 0), c(rlen + PAD, 0) {
        memcpy(g.data(), gp_, glen); memcpy(r.data(), rp, rlen);
        static const u8 cpl[5] = {3, 2, 1, 0, 4};
        for (u64 k = 0; k < rlen; k++) c[k] = cpl[r[rlen - k - 1] > 4 ? 4 : r[rlen - k - 1]];
        ar.init(mem.get(), M1);
        X.ar = &ar; X.ls = &ls; X.read.p = r.data(); X.read.len = rlen; X.com.p = c.data(); X.com.len = rlen;
        X.g = g.data(); so = 0; sl = glen; X.seq_off = &so; X.seq_len = &sl;
    }
    u64 so, sl;
};
u64 out64(const GVec<u64> &v, u64 *out, u64 cap) { for (u32 i = 0; i < v.n && i < cap; i++) out[i] = v[i]; return v.n; }
}
u64 hs_gap_anchors(const u8 *g, u64 glen, const u8 *r, u64 rlen, u64 gap_str, u64 gap_end, int shape_len, int step1, int step2, int direction, i64 lower, i64 upper, u64 rvcp, u64 *out, u64 cap) {
    GapHost H(g, glen, r, rlen);
    GVec<u64> g_hs, anc; g_hs.init(&H.ar); anc.init(&H.ar);
    g_stream(H.X.ref(0), H.X.read, g_hs, gap_str, gap_end, (u32)shape_len, step1, step2);
    g_create_anchors(g_hs, anc, shape_len, direction, lower, upper, rvcp, gap_str, gap_end, H.X);
    return H.ar.ovf ? ~0ULL : out64(anc, out, cap);
}
u64 hs_gap_anchor_pair(const u8 *g, u64 glen, const u8 *r, u64 rlen, u64 gs, u64 ge, int shape_len, int step1, int step2, u64 rvcp, u64 gs1, u64 ge1, u64 gs2, u64 ge2, u64 *out1, u64 *n1, u64 *out2,
                       u64 cap) {
    GapHost H(g, glen, r, rlen);
    GVec<u64> g_hs, a1, a2; g_hs.init(&H.ar); a1.init(&H.ar); a2.init(&H.ar);
    g_stream(H.X.ref(0), H.X.read, g_hs, gs, ge, (u32)shape_len, step1, step2);
    g_create_anchor_pair(g_hs, a1, a2, shape_len, rvcp, gs1, ge1, gs2, ge2, H.X);
    *n1 = out64(a1, out1, cap);
    return out64(a2, out2, cap);
}
u64 hs_gap_canchors(const u8 *g, u64 glen, const u8 *r, u64 rlen, u64 s1s, u64 s1e, u64 s2s, u64 s2e, int step1, int step2, int shape_len, i64 lower, i64 upper, u64 *out, u64 cap) {
    GapHost H(g, glen, r, rlen);
    GVec<u64> g_hs, anc; g_hs.init(&H.ar); anc.init(&H.ar);
    c_stream(H.X.ref(0), g_hs, s1s, s1e, step1, shape_len, 0);
    c_stream(H.X.read, g_hs, s2s, s2e, step2, shape_len, 1);
    c_create_anchors2(g_hs, anc, lower, upper, H.ls.st);
    return out64(anc, out, cap);
}
u64 hs_gap_chains(const u64 *anchors, u64 n, u64 read_len, int alt, int direction, u64 gap_str, u64 gap_end, int closest, u64 *out, u64 cap, int *pr) {
    u8 z = 0;
    GapHost H(&z, 1, &z, 1);
    if (alt) { H.X.gp.chn1_min_len = 1; H.X.gp.chn1_abort = 0; H.X.gp.chn1_fn = 2; H.X.gp.chn2_abort = 0; H.X.gp.chn2_fn = 3; }
    H.X.gp.direction = direction;
    GVec<u64> a, tiles; a.init(&H.ar, (u32)n + 16); tiles.init(&H.ar, (u32)n + 16);
    for (u64 i = 0; i < n; i++) a.push(anchors[i]);
    g_chains_from_anchors(a, tiles, read_len, H.X);
    if (closest) { IPair r = closest_extension_chain(tiles, gap_str, gap_end, closest == 2, H.X.gp); pr[0] = r.first; pr[1] = r.second; }
    return H.ar.ovf ? ~0ULL : out64(tiles, out, cap);
}
// ---- the whole gap layer on a Shim context (genome + f2 from hs_create): hs_gap_map = orc_gap_map, hs_map_read_g = orc_map_read_g
namespace {
struct GapRead {
    size_t M1, M2;
    std::unique_ptr<char[]> mem, mem2; GArena ar, keep; LeaderScratch ls; std::vector<u8> g, r, c; std::vector<u64> so, sl; std::vector<F96> f1a, f1b; GapCtx X;
    GapRead(Shim &S, const u8 *rp, u64 L, size_t m1 = (size_t)256 << 20, size_t m2 = (size_t)64 << 20) : M1(m1), M2(m2), mem(new char[m1]), mem2(new char[m2]), r(L + PAD, 0), c(L + PAD, 0) {
        for (size_t i = 0; i < S.seqs.size(); i++) { so.push_back(g.size()); sl.push_back(S.lens[i]); g.insert(g.end(), S.seqs[i].begin(), S.seqs[i].begin() + S.lens[i]); g.insert(g.end(), PAD, 0); }
        memcpy(r.data(), rp, L);
        static const u8 cpl[5] = {3, 2, 1, 0, 4};
        for (u64 k = 0; k < L; k++) c[k] = cpl[r[L - k - 1]];
        u32 nf = read_feature_count(L);
        f1a.resize(nf); f1b.resize(nf);
        features_closed(r.data(), nf, f1a.data());
        features_closed(c.data(), nf, f1b.data());
        ar.init(mem.get(), M1); keep.init(mem2.get(), M2);
        X.ar = &ar; X.ls = &ls; X.read.p = r.data(); X.read.len = L; X.com.p = c.data(); X.com.len = L;
        X.g = g.data(); X.seq_off = so.data(); X.seq_len = sl.data();
        X.f1[0].p = f1a.data(); X.f1[0].n = nf; X.f1[1].p = f1b.data(); X.f1[1].n = nf;
        X.gf.base = S.f2.data(); X.gf.off = S.f2_off.data(); X.gf.nseq = (u32)S.seqs.size();
    }
};
}
u64 hs_gap_map(void *h, const u8 *read, u64 len, int which, u64 gs1, u64 ge1, u64 gs2, u64 ge2, int direction, int alt, u64 *out_str, u64 *out_end, u64 *n2, u64 cap) {
    Shim &S = *(Shim *)h;
    GapRead H(S, read, len);
    GapParms &gp = H.X.gp;
    if (alt & 1) { gp.chn1_min_len = 1; gp.chn1_abort = 0; gp.chn1_fn = 2; gp.chn2_abort = 0; gp.chn2_fn = 3; }
    if (alt & 2) gp.thd_cts_major_limit = 3;          // the stream state "extended" (oracle/lnr_oracle.cpp orc_map_read_g2)
    gp.read_len = len; gp.ref_len = S.lens[cord_id(gs1)];
    GSeq ref = H.X.ref(cord_id(gs1));
    GVec<u64> ts1, te1, ts2, te2; ts1.init(&H.keep, 64); te1.init(&H.keep, 64); ts2.init(&H.keep, 64); te2.init(&H.keep, 64);
    if (which == 1) gap_map_generic(ref, ts1, te1, gs1, ge1, H.X);
    else if (which == 2) gap_map_extend(ref, ts1, te1, gs1, ge1, direction, H.X);
    else gap_map_extends(ref, ts1, te1, ts2, te2, gs1, ge1, gs2, ge2, H.X);
    if (H.ar.ovf || H.keep.ovf) return ~0ULL;
    u64 n1 = ts1.n;
    for (u64 i = 0; i < n1 && i < cap; i++) { out_str[i] = ts1[(u32)i]; out_end[i] = i < te1.n ? te1[(u32)i] : 0; }
    *n2 = ts2.n;
    for (u64 i = 0; i < ts2.n && n1 + i < cap; i++) { out_str[n1 + i] = ts2[(u32)i]; out_end[n1 + i] = i < te2.n ? te2[(u32)i] : 0; }
    return n1 | ((u64)te1.n << 32);
}
// apxMap + mapGaps + reformCords (Mapper::p_calRecords with -g gap_len [-dup], mapper.cpp:207-231,438-453); cords through hs_get_cords
static i64 map_read_g_lim(void *h, const u8 *read, u64 len, u32 gap_len, int f_dup, u64 arena_bytes, u64 keep_bytes, u64 work_cap, int *ext_state = nullptr);
i64 hs_map_read_g(void *h, const u8 *read, u64 len, u32 gap_len, int f_dup) { return map_read_g_lim(h, read, len, gap_len, f_dup, (u64)256 << 20, (u64)64 << 20, ~0ULL); }
// *ext_state in / out: the stream state of the reference's per-thread GapParms (0 nothing extended yet, 1 extended: thd_cts_major_limit 3)
i64 hs_map_read_g2(void *h, const u8 *read, u64 len, u32 gap_len, int f_dup, int *ext_state) { return map_read_g_lim(h, read, len, gap_len, f_dup, (u64)256 << 20, (u64)64 << 20, ~0ULL, ext_state); }
// the same with the arena sizes and the work budget of a k_gap worker: -11 = out of arena, -12 = over the work budget (the cords of apxMap stay)
i64 hs_map_read_g_lim(void *h, const u8 *read, u64 len, u32 gap_len, int f_dup, u64 arena_bytes, u64 keep_bytes, u64 work_cap) { return map_read_g_lim(h, read, len, gap_len, f_dup, arena_bytes, keep_bytes, work_cap); }
static i64 map_read_g_lim(void *h, const u8 *read, u64 len, u32 gap_len, int f_dup, u64 arena_bytes, u64 keep_bytes, u64 work_cap, int *ext_state) {
    Shim &S = *(Shim *)h;
    int rc = map_read(S, read, len, false);
    if (rc) return rc;
    if (len <= 200 || gap_len == 0) return (i64)S.cs.size();
    GapRead H(S, read, len, arena_bytes, keep_bytes);
    H.X.work_cap = work_cap;
    H.X.gp.f_dup = f_dup;
    H.X.gp.thd_gap_len_min = gap_len == 1 ? 50 : (gap_len < 10 ? 10 : gap_len);
    if (ext_state && *ext_state) H.X.gp.thd_cts_major_limit = 3;
    GVec<u64> cs, ce; cs.init(&H.keep, (u32)S.cs.size() * 2 + 64); ce.init(&H.keep, (u32)S.cs.size() * 2 + 64);
    for (u64 v : S.cs) cs.push(v);
    for (u64 v : S.ce) ce.push(v);
    int gr = gap_map_gaps(cs, ce, H.keep, H.X);
    gap_reform_cords(cs, ce);
    if (H.ar.ovf == 2) return -12;
    if (gr || H.ar.ovf || H.keep.ovf || cs.n != ce.n) return -11;
    if (ext_state) *ext_state = H.X.gp.thd_cts_major_limit == 3;
    S.stats[0] = H.ar.hw; S.stats[1] = H.keep.hw;   // (arena high-water marks, read back through hs_get_stats)
    S.cs.assign(cs.p, cs.p + cs.n);
    S.ce.assign(ce.p, ce.p + ce.n);
    return (i64)S.cs.size();
}
// the column DP's (dx, dy) forms against the anchor forms: over x2 = base, x1 = base + dx, y2 = ybase, y1 = ybase + dy for every (dx, dy) of the
// rectangle, same strand and opposite strands.  Returns the number of pairs where "positive?" or the positive value differ.
u64 hs_gap_delta_check(int fn, u32 dx_max, i32 dy_lo, i32 dy_hi) {
    auto anchor = [](u64 x, u64 y, u64 strand) { return (strand << 50) | ((((x - y + G_ANCHOR_ZERO) & ((1ULL << 30) - 1))) << 20) | y; };
    u64 bad = 0;
    const u64 xb = 50000, yb = 3000;
    for (u32 dx = 0; dx <= dx_max; dx++)
        for (i32 dy = dy_lo; dy <= dy_hi; dy++)
            for (int st = 0; st < 4; st++) {
                u64 a1 = anchor(xb + dx, (u64)((i64)yb + dy), (u64)(st & 1)), a2 = anchor(xb, yb, (u64)(st >> 1));
                int lit = gap_dp_score(fn, a1, a2);
                u32 ys1 = (u32)ganc_y(a1) | ((u32)ganc_strand(a1) << 24), ys2 = (u32)ganc_y(a2) | ((u32)ganc_strand(a2) << 24);
                u32 ddx = (u32)ganc_x(a1) - (u32)ganc_x(a2);
                i32 ddy = (i32)(ys1 - ys2);
                int alt = gap_score_box(fn, ddx, ddy) ? gap_score_delta(fn, ddx, (u32)ddy) : -1;
                if ((lit > 0) != (alt > 0) || (lit > 0 && lit != alt)) bad++;
            }
    return bad;
}
int hs_gap_score(int which, u64 a, u64 b, u64 c, u64 d, u64 read_len, int strand) {
    switch (which) {
        case 1: return gap_anchor_score1(a, b);
        case 2: return gap_anchor_score2(a, b);
        case 3: return gap_block_score2(a, b, c, d, read_len, strand);
        case 5: return gap_clip_score(a, b);
        case 6: return gap_anchor_score1_pos(a, b);
        case 7: return gap_anchor_score2_pos(a, b);
        default: return gap_block_score3(a, b, c, d, read_len, strand);
    }
}

// packed-read form of the minimizer sample against the byte form: returns the number of samples that disagree
// (samples where the packed form declines are counted separately in *n_fallback)
u64 hs_packed_vs_bytes(const u8 *read, u64 L, u64 read_str, u64 read_end, int alpha, u64 *n_fallback, u64 *n_samples) {
    std::vector<u8> s(L + PAD, 0);
    memcpy(s.data(), read, L);
    u64 nw = packed_words(L);
    std::vector<u64> pk(nw, 0);
    std::vector<u32> nm(nw, 0);
    for (u64 i = 0; i < L; i++) {
        u8 b = s[i];
        if (b > 3) nm[i >> 5] |= 1u << (i & 31);
        else pk[i >> 5] |= (u64)b << (2 * (i & 31));
    }
    const u8 *sb = s.data();
    int ks = shape_init_skip(sb);
    u64 k0 = read_str + 21;
    int C = shape_const(sb, 0, ks, k0);
    u32 ns = seed_num_samples(read_str, read_end, (u32)alpha);
    u64 bad = 0, fb = 0;
    // the kernels' view of the read (PackedSeq): element access, hashInit skip, shape constant, byte-path samples
    PackedSeq ps; ps.pk = pk.data(); ps.nm = nm.data(); ps.L = L;
    for (u64 i = 0; i < L + 40; i++) if (ps[i] != s[i]) bad++;
    if (shape_init_skip(ps) != ks) bad++;
    if (shape_const(ps, 0, ks, k0) != C) bad++;
    for (u64 kk = 0; kk + 64 < L; kk += 97) if (shape_const(ps, 0, ks, kk) != shape_const(sb, 0, ks, kk)) bad++;
    for (u32 q = 0; q < ns; q++) {
        u64 k = k0 + alpha - 1 + (u64)alpha * q;
        SeedOut a = seed_sample(sb, k, k0, 0, ks, C), b;
        if (q < 4 || q % 13 == 0) { SeedOut c = seed_sample(ps, k, k0, 0, ks, C); if (a.X != c.X || a.Y != c.Y || a.strand != c.strand) bad++; }
        if (!seed_sample_packed(pk.data(), nm.data(), k, k0, C, b)) { fb++; SeedOut c = seed_sample(ps, k, k0, 0, ks, C); if (a.X != c.X || a.Y != c.Y || a.strand != c.strand) bad++; continue; }
        if (a.X != b.X || a.Y != b.Y || a.strand != b.strand) bad++;
    }
    *n_fallback = fb; *n_samples = ns;
    return bad;
}

// read features computed from the packed strand via per-cell counts, written as 3 ints per entry
u64 hs_read_features_packed(const u8 *read, u64 L, int strand, i32 *out, u64 cap) {
    std::vector<u8> s(L + PAD, 0);
    static const u8 cpl[5] = {3, 2, 1, 0, 4};
    if (strand) for (u64 k = 0; k < L; k++) s[k] = cpl[read[L - 1 - k]];
    else memcpy(s.data(), read, L);
    u64 nw = packed_words(L) + 2;
    std::vector<u64> pk(nw, 0);
    std::vector<u32> nm(nw, 0);
    for (u64 i = 0; i < L; i++) { u8 b = s[i]; if (b > 3) nm[i >> 5] |= 1u << (i & 31); else pk[i >> 5] |= (u64)b << (2 * (i & 31)); }
    u32 nf = read_feature_count(L);
    std::vector<i32> c0(nf + 2), c1(nf + 2), c2(nf + 2);
    for (u32 c = 0; c < nf + 2; c++) cell_2mers_packed(pk.data(), nm.data(), 16ULL * c, c0[c], c1[c], c2[c]);
    for (u32 m = 0; m < nf && m < cap; m++) { out[3 * m] = c0[m] + c0[m + 1] + c0[m + 2]; out[3 * m + 1] = c1[m] + c1[m + 1] + c1[m + 2]; out[3 * m + 2] = c2[m] + c2[m + 1] + c2[m + 2]; }
    return nf;
}

// fuzz hook for ref_sort: sorts keys (compare on the high 32 bits only, descending when desc != 0)
void hs_ref_sort_hi32(u64 *a, u64 n, int desc) {
    SortStack st;
    if (desc) ref_sort(a, (long)n, [](const u64 &x, const u64 &y) { return (x >> 32) > (y >> 32); }, st);
    else ref_sort(a, (long)n, [](const u64 &x, const u64 &y) { return (x >> 32) < (y >> 32); }, st);
}
void hs_std_sort_hi32(u64 *a, u64 n, int desc) {
    if (desc) std::sort(a, a + n, [](const u64 &x, const u64 &y) { return (x >> 32) > (y >> 32); });
    else std::sort(a, a + n, [](const u64 &x, const u64 &y) { return (x >> 32) < (y >> 32); });
}
// fuzz hook: the 32-bit block_score2 against the literal 64-bit getApxChainScore2 (cluster_util.cpp:586-631); returns mismatches
static int block_score2_literal(u64 c11, u64 c22) {
    i64 dy = (i64)(cord_y(c11) - cord_y(c22));
    i64 dx = (i64)(cord_x(c11) - cord_x(c22));
    if (dx < 0 || dy < 0 || cord_strand(c11 ^ c22) || dx > 20000 || dy > 20000) return INT_MIN;
    i64 da = labs64(dx - dy);
    i64 derr = (100 * da) / max64(max64(labs64(dy), 100), labs64(dx));
    if (da > 100 || derr > 50) {
        if (dx < dy) return (int)(100 - 30 - dy / 1000 - dx / 100);
        return (int)(100 - 30 - dy / 100 - dx / 1000);
    }
    return (int)(100 - dy / 95);
}
u64 hs_block_score2_fuzz(u64 seed, u64 n) {
    u64 bad = 0, st = seed * 0x9E3779B97F4A7C15ULL + 1;
    auto rnd = [&st]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (u64 it = 0; it < n; it++) {
        u64 a = rnd(), b = rnd(), c = rnd();
        u64 x2 = a & 0x3fffffff, y2 = (a >> 32) & 0xfffff, id = (a >> 52) & 3, x1, y1;
        switch (c & 7) {
        case 0: x1 = b & 0x3fffffff; y1 = (b >> 32) & 0xfffff; break;
        case 1: x1 = x2 + b % 20002; y1 = y2 + (b >> 32) % 20002; break;                                  // the whole accepted range
        case 2: { u64 d = b % 20001; x1 = x2 + d; y1 = y2 + d + (b >> 40) % 241 - 120; break; }            // around da = 100
        case 3: x1 = x2 + b % 300; y1 = y2 + (b >> 32) % 300; break;
        case 4: { u64 d = b % 20001; x1 = x2 + d; y1 = y2 + d / 2 + (b >> 40) % 7 - 3; break; }            // around derr = 50
        case 5: { u64 d = b % 20001; y1 = y2 + d; x1 = x2 + d / 2 + (b >> 40) % 7 - 3; break; }
        case 6: x1 = x2 + b % 20002 - 1; y1 = y2 + (b >> 32) % 3 - 1; break;
        default: x1 = x2 + (b >> 32) % 3 - 1; y1 = y2 + b % 20002 - 1; break;
        }
        x1 &= 0x3fffffff; y1 &= 0xfffff;
        u64 c11 = mk_cord((id << 30) + x1, y1, (c >> 8) & 1), c22 = mk_cord((id << 30) + x2, y2, (c >> 9) & 1 & ((c >> 10) & 1));
        if (block_score2(c11, c22) != block_score2_literal(c11, c22)) bad++;
    }
    return bad;
}
// fuzz hook: branch-free chain scores (what the lane-parallel DP evaluates) against the literal ones; returns mismatches
u64 hs_chain_score_fuzz(u64 seed, u64 n) {
    u64 bad = 0, st = seed * 0x9E3779B97F4A7C15ULL + 1;
    auto rnd = [&st]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (u64 it = 0; it < n; it++) {
        u64 a = rnd(), b = rnd(), c = rnd();
        u32 x2 = (u32)(a & 0x3fffffff), y2 = (u32)((a >> 32) & 0xfffff);
        u32 x1, y1;
        switch (c & 7) {
        case 0: x1 = (u32)(b & 0x3fffffff); y1 = (u32)((b >> 32) & 0xfffff); break;                    // anything
        case 1: x1 = x2 + (u32)(b % 400); y1 = y2 + (u32)((b >> 32) % 400); break;                       // the usual window
        case 2: x1 = x2 + (u32)(b % 5000); y1 = y2 + (u32)((b >> 32) % 5000); break;
        case 3: x1 = x2 + (u32)(b % 400) - 40; y1 = y2 + (u32)((b >> 32) % 64) - 8; break;               // around the cut-offs
        case 4: x1 = x2 + (u32)(b & 0x3fffff); y1 = y2 + (u32)((b >> 32) & 0xfffff); break;              // long gaps
        case 5: { u32 d = (u32)(b % 200000); x1 = x2 + d; y1 = y2 + d + (u32)((b >> 40) % 41) - 20; break; }   // near-diagonal, long
        case 6: { u32 d = (u32)(b % 3000); x1 = x2 + d; y1 = y2 + d * (u32)((b >> 40) % 7) / 3; break; }
        default: x1 = x2 + (u32)(b % 60); y1 = y2 + (u32)((b >> 32) % 60); break;
        }
        x1 &= 0x3fffffff; y1 &= 0xfffff;
        if (chain_score(x1, y1, x2, y2) != chain_score_bl(x1, y1, x2, y2)) bad++;
        if (chain_score0(x1, y1, x2, y2) != chain_score0_bl(x1, y1, x2, y2)) bad++;
        // the two-stage form the DP kernel evaluates: positive literal score <=> candidate with the same positive score
        {
            DpPair p0, p1;
            bool c0 = dp_pair_cand<0>(x1, y1, x2, y2, p0), c1 = dp_pair_cand<1>(x1, y1, x2, y2, p1);
            int l0 = chain_score(x1, y1, x2, y2), l1 = chain_score0(x1, y1, x2, y2);
            int s0 = c0 ? dp_pair_score<0>(p0) : 0, s1 = c1 ? dp_pair_score<1>(p1) : 0;
            if (l0 > 0 ? !(c0 && s0 == l0) : (c0 && s0 > 0)) bad++;
            if (l1 > 0 ? !(c1 && s1 == l1) : (c1 && s1 > 0)) bad++;
            if (x1 >= x2) {   // the form for x-sorted predecessors (dx >= 0) must agree with the general one
                DpPair q0, q1;
                bool d0 = dp_pair_cand<0, true>(x1, y1, x2, y2, q0), d1 = dp_pair_cand<1, true>(x1, y1, x2, y2, q1);
                // (da of the sorted form is only meaningful for dy >= 0: every use sits behind a test that dy is positive)
                if (d0 != c0 || d1 != c1 || q0.M != p0.M || q0.dy != p0.dy || q1.M != p1.M || (p0.dy >= 0 && (q0.da != p0.da || q1.da != p1.da))) bad++;
            }
        }
    }
    return bad;
}
}
