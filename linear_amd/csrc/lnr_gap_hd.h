// lnr_gap_hd.h -- the gap re-mapper (mapGaps / reformCords, gap.cpp / gap_util.cpp / cords.cpp:504-687) as host + device
// functions of the product, in the idiom of lnr_hd.h: plain arrays from a per-read arena, the tie-sensitive sorts through
// ref_sort.h, the chain traceback and the block chaining through the forms lnr_hd.h already has.  One read is one serial walk over
// its gaps (the tiles of a gap are inserted into the cord list before the next gap is looked at).  WORK IN PROGRESS (round 2): the
// layers below are checked on the host against the oracle (tests/test_gap_shim_cpu.py through tests/host_shim.cpp); the kernel that
// runs them (one wave per read) and the ABI switch (-g) are the next step.  Every function cites the reference lines it follows.
#pragma once
#include "lnr_hd.h"

namespace lnr {

// ---- arena + growable array of the gap path: temporaries of one gap are released together (mark / release)
struct GArena {
    char *base; u64 off, cap; int ovf;
    LNR_HD void init(void *b, u64 c) { base = (char *)b; off = 0; cap = c; ovf = 0; }
    LNR_HD void *get(u64 bytes) {
        bytes = (bytes + 15) & ~15ULL;
        if (off + bytes > cap) { ovf = 1; return (void *)base; }   // (callers check ovf at the end: the result is then discarded)
        void *r = base + off; off += bytes; return r;
    }
    LNR_HD u64 mark() const { return off; }
    LNR_HD void release(u64 m) { off = m; }
};
template <class T> struct GVec {
    T *p; u32 n, cap; GArena *ar;
    LNR_HD void init(GArena *a, u32 c0 = 16) { ar = a; n = 0; cap = c0; p = (T *)a->get((u64)c0 * sizeof(T)); }
    LNR_HD void reserve(u32 c) {
        if (c <= cap) return;
        u32 nc = cap * 2 > c ? cap * 2 : c;
        T *q = (T *)ar->get((u64)nc * sizeof(T));
        if (ar->ovf) return;
        for (u32 i = 0; i < n; i++) q[i] = p[i];
        p = q; cap = nc;
    }
    LNR_HD void push(const T &v) { reserve(n + 1); if (n < cap) p[n++] = v; }
    LNR_HD T &operator[](u32 i) { return p[i]; }
    LNR_HD const T &operator[](u32 i) const { return p[i]; }
    LNR_HD T &back() { return p[n - 1]; }
    LNR_HD bool empty() const { return n == 0; }
    LNR_HD void clear() { n = 0; }
    LNR_HD void resize(u32 m, const T &fill = T()) { reserve(m); if (m > cap) return; for (u32 i = n; i < m; i++) p[i] = fill; n = m; }
    LNR_HD void insert(u32 pos, const T *src, u32 m) {       // src must not alias this vector
        if (!m) return;
        reserve(n + m);
        if (n + m > cap) return;
        for (u32 i = n; i > pos; i--) p[i - 1 + m] = p[i - 1];
        for (u32 i = 0; i < m; i++) p[pos + i] = src[i];
        n += m;
    }
    LNR_HD void erase(u32 a, u32 b) { if (b <= a) return; for (u32 i = b; i < n; i++) p[a + i - b] = p[i]; n -= b - a; }
    LNR_HD void append(const GVec<T> &o) { insert(n, o.p, o.n); }
};
template <class T> LNR_HD inline T gmin3(T a, T b, T c) { T m = a < b ? a : b; return m < c ? m : c; }
LNR_HD inline i64 gabs(i64 v) { return v < 0 ? -v : v; }

// ---- formats (gap_util.cpp:261-336 tile signs, :480-584 gap anchors / g_hs words)
static const u64 TILE_STR = 1ULL << 62, TILE_END = 1ULL << 63, G_ANCHOR_ZERO = 1ULL << 20;
LNR_HD inline u64 is_tile_end(u64 v) { return v & TILE_END; }
LNR_HD inline bool is_tile_start(u64 v) { return (v & TILE_STR) != 0; }
LNR_HD inline void set_tile_end(u64 &v) { v |= TILE_END; }
LNR_HD inline void set_tile_start(u64 &v) { v |= TILE_STR; }
LNR_HD inline void remove_tile_sgn(u64 &v) { v &= ~(TILE_STR | TILE_END); }
LNR_HD inline void copy_tile_sgn(u64 t1, u64 &t2) { t2 = (t1 & (TILE_STR | TILE_END)) | (t2 & ~(TILE_STR | TILE_END)); }
LNR_HD inline u64 tile_strand(u64 v) { return (v >> 61) & 1; }
LNR_HD inline u64 g_hs_make(u64 xval, u64 type, u64 strand, u64 coord) { return (xval << 33) + (type << 31) + (strand << 30) + coord; }
LNR_HD inline u64 g_hs_xt(u64 v) { return (v >> 31) & 0xffffffffULL; }
LNR_HD inline u64 ganc_y(u64 a) { return a & 0xfffffULL; }
LNR_HD inline u64 ganc_x(u64 a) { return ((a >> 20) & ((1ULL << 30) - 1)) - G_ANCHOR_ZERO + ganc_y(a); }
LNR_HD inline u64 ganc_stranchor(u64 a) { return ((a >> 20) & ((1ULL << 31) - 1)) - G_ANCHOR_ZERO; }
LNR_HD inline u64 ganc_strand(u64 a) { return (a >> 50) & 1ULL; }
LNR_HD inline u64 cord2stranchor(u64 c) { return cord_x(c) - cord_y(c) + (cord_strand(c) << 30); }
LNR_HD inline u64 ganc_make(u64 hs1, u64 hs2, u64 revscomp_const) {                       // g_hs_setAnchor_ :548-557
    u64 strand = ((hs1 ^ hs2) >> 30) & 1;
    u64 x = revscomp_const * strand - ((strand << 1) - 1) * (hs2 & ((1ULL << 30) - 1));
    return (((hs1 + G_ANCHOR_ZERO - x) & ((1ULL << 30) - 1)) << 20) + x + (strand << 50);
}
LNR_HD inline u64 canc_make(u64 hs1, u64 hs2) { u64 x = hs2 & ((1ULL << 30) - 1); return (((hs1 - x + G_ANCHOR_ZERO) & ((1ULL << 30) - 1)) << 20) + x; }   // c_2Anchor_ :558
LNR_HD inline u64 ganc_tile(u64 a) {                                                       // g_hs_anchor2Tile :574-584
    u64 strand = (a >> 50) & 1, y = ganc_y(a);
    return (((a - (G_ANCHOR_ZERO << 20) + ((a & 0xfffffULL) << 20)) & ~(1ULL << 50)) & ~0xfffffULL) + y + (strand << 61);
}

// ---- parameters (GapParms gap_util.h:97-196 with the defaults of gap_util.cpp:27-90) and the read's context
struct GapParms {
    float thd_err = 0.2f, thd_gmsa_d_anchor_rate = 0.1f;
    int direction = 0, int_precision = 10000, thd_tile_size = 96;
    u32 thd_accept_score = 32, thd_ctfcs_pattern_in_window = 1;
    u64 thd_cts_major_limit = 1, ref_len = 0, read_len = 0;
    i64 thd_ctfas2_connect_danchor = 50, thd_ctfas2_connect_dy_dx = 150, thd_me_reject_gap = 200, thd_smcn_danchor = 12;
    int thd_eis_shape_len = 9, thd_eis_step1 = 5, thd_eis_step2 = 1, thd_etfas_shape_len = 5, thd_etfas_step1 = 3, thd_etfas_step2 = 1;
    int thd_dcgx_window_size = 5, thd_dcgx_Xdrop_peak = 125, thd_dcgx_Xdrop_sum = 300;
    int thd_tts_overlap_size = 81, thd_tts_gap_size = 100;
    u64 thd_dcomx_err_dx = 25, thd_dcomx_err_dy = 25, thd_eicos_clip_dxy = 30;
    int thd_eicos_f_as_ins = 1;
    int thd_ccps_window_size = 5, thd_ccps_clip1_upper = 80000, thd_ccps_clip2_lower = 120000;
    i64 thd_mg1_danc_indel = 80, thd_max_extend2 = 5000, f_dup = 0, thd_gap_len_min = 0;
    int f_rfts_clip = 1;
    int chn1_min_len = 1, chn1_abort = 50, chn1_fn = 1;   // anchors: getGapAnchorsChainScore (1) / ...Score2 (2)
    int chn2_min_len = 1, chn2_abort = 0, chn2_fn = 2;    // blocks: getGapBlocksChainScore2 (2) / ...Score3 (3)
};
struct GSeq { const u8 *p; u64 len; };                    // bases with >= 64 zero bytes behind the end
struct GapCtx {                                           // one read
    GArena *ar; LeaderScratch *ls;
    GSeq read, com;                                       // the read and its reverse complement (both padded copies)
    const u8 *g; const u64 *seq_off; const u64 *seq_len;  // genome
    FeatView f1[2]; GenomeFeat gf;
    GapParms gp;
    LNR_HD GSeq ref(u64 id) const { GSeq s; s.p = g + seq_off[id]; s.len = seq_len[id]; return s; }
};

// ---- k-mer streams (shape_extend.cpp:86-116 hashInit, :231-243 hashNextV, :122-131 / :213-219 the single-strand pair)
struct GShape { u64 h, crh; int x, left; u32 span; };
LNR_HD inline u64 gshape_init(GShape &me, const u8 *it) {
    me.left = 0; me.h = 0; me.crh = 0; me.x = -3;
    u64 k = 0, count = 0;
    while (count < me.span) { if (it[k + count] == 4) { k += count + 1; count = 0; } else count++; }
    u32 bit = 2;
    for (u32 i = 0; i < me.span - 1; ++i) { u64 v = it[k + i]; me.x += ((int)v << 1) - 3; me.h = (me.h << 2) + v; me.crh += (3ULL - v) << bit; bit += 2; }
    return k;
}
LNR_HD inline void g_kmer_stream(const GSeq &seq, GVec<u64> &g_hs, u64 str, u64 end, int shape_len, int step, u64 type) {   // g_mapHs_kmer_ gap_util.cpp:632-662
    if (seq.len < (u64)shape_len) return;
    GShape sh; sh.span = (u32)shape_len;
    gshape_init(sh, seq.p + str);
    u64 mask = (1ULL << (2 * sh.span - 2)) - 1;
    int count = 0;
    u64 lim = end < seq.len - (u64)shape_len ? end : seq.len - (u64)shape_len;
    for (u64 k = str; k < lim; k++) {
        const u8 *it = seq.p + k;
        int v2 = it[sh.span - 1];
        sh.h = ((sh.h & mask) << 2) + (u64)v2;
        sh.crh = ((sh.crh >> 2) & mask) + ((3ULL - (u64)(i64)v2) << (2 * sh.span - 2));
        sh.x += (v2 - sh.left) << 1;
        sh.left = it[0];
        u64 strand = sh.x < 0 ? 1 : 0;
        if (++count == step) { g_hs.push(g_hs_make(strand ? sh.crh : sh.h, type, strand, k)); count = 0; }
    }
}
LNR_HD inline void g_stream(const GSeq &ref, const GSeq &read, GVec<u64> &g_hs, u64 gap_str, u64 gap_end, u32 shape_len, int step1, int step2) {   // g_stream_ :1663-1688
    u64 gs_str = cord_x(gap_str), gs_end = cord_x(gap_end), gr_str = cord_y(gap_str), gr_end = cord_y(gap_end);
    if (cord_strand(gap_str)) { u64 a = read.len - gr_str - 1, b = read.len - gr_end - 1; gr_str = b; gr_end = a; }
    g_kmer_stream(ref, g_hs, gs_str, gs_end, (int)shape_len, step1, 0);
    g_kmer_stream(read, g_hs, gr_str, gr_end, (int)shape_len, step2, 1);
}
LNR_HD inline void c_stream(const GSeq &seq, GVec<u64> &g_hs, u64 sq_str, u64 sq_end, int step, int shape_len, u64 type) {   // c_stream_ :1694-1716
    if (seq.len < (u64)shape_len) return;
    u32 span = (u32)shape_len;
    u64 h = 0, mask = (1ULL << (2 * span - 2)) - 1;
    for (u32 i = 0; i < span - 1; ++i) h = (h << 2) + seq.p[sq_str + i];
    int count = 0;
    u64 lim = sq_end < seq.len - (u64)shape_len ? sq_end : seq.len - (u64)shape_len;
    for (u64 k = sq_str; k < lim; k++) {
        h = ((h & mask) << 2) + seq.p[k + span - 1];
        if (++count == step) { g_hs.push(g_hs_make(h, type, 0, k)); count = 0; }
    }
}

// ---- anchors from the sorted k-mer list (gap_util.cpp:669-752, 1596-1661, 1818-1853)
LNR_HD inline void g_set_anchors(const GVec<u64> &g_hs, GVec<u64> &out, int p1, int p2, int k, u64 rvcp, i64 lower, i64 upper, u64 gap_str, u64 gap_end, int direction, const GapParms &gp) {
    if (direction == 0) {
        for (int i = p1; i < p2; i++) for (int j = p2; j < k; j++) {
            u64 a = ganc_make(g_hs[(u32)i], g_hs[(u32)j], rvcp);
            i64 t = (i64)ganc_stranchor(a);
            if (t < upper && t >= lower) out.push(a);
        }
        return;
    }
    i64 y_ref = direction < 0 ? (i64)cord_y(gap_end) : (i64)cord_y(gap_str);
    i64 base = (i64)cord2stranchor(direction < 0 ? gap_end : gap_str);
    i64 d_anchor = (i64)((1LL << 7) * gp.thd_gmsa_d_anchor_rate);
    for (int i = p1; i < p2; i++) for (int j = p2; j < k; j++) {
        u64 a = ganc_make(g_hs[(u32)i], g_hs[(u32)j], rvcp);
        i64 t = (i64)ganc_stranchor(a);
        i64 dy = direction < 0 ? y_ref - (i64)ganc_y(a) : (i64)ganc_y(a) - y_ref;
        if (dy < 0 || (ganc_strand(a) ^ cord_strand(gap_str))) continue;
        i64 acc = (dy >> 7) * d_anchor; if (acc < 50) acc = 50;
        i64 lo = base - acc; if (lo < 0) lo = 0;
        if (t < base + acc && t >= lo) out.push(a);
    }
}
template <class F> LNR_HD inline void g_hs_blocks(GVec<u64> &g_hs, int shape_len, SortStack &st, F &&emit) {
    u64 mask = (1ULL << (2 * shape_len + 33)) - 1;
    ref_sort(g_hs.p, (long)g_hs.n, [mask](const u64 &a, const u64 &b) { return (a & mask) < (b & mask); }, st);
    int p1 = 0, p2 = 0;
    for (int k = 1; k < (int)g_hs.n; k++) {
        u64 t = g_hs_xt((g_hs[(u32)k] ^ g_hs[(u32)k - 1]) & mask);
        if (t == 0) continue;
        if (t == 1) { p2 = k; continue; }
        emit(p1, p2, k);
        p1 = k; p2 = k;
    }
}
LNR_HD inline void g_create_anchors(GVec<u64> &g_hs, GVec<u64> &anchors, int shape_len, int direction, i64 lower, i64 upper, u64 rvcp, u64 gap_str, u64 gap_end, GapCtx &X) {
    g_hs_blocks(g_hs, shape_len, X.ls->st, [&](int p1, int p2, int k) { g_set_anchors(g_hs, anchors, p1, p2, k, rvcp, lower, upper, gap_str, gap_end, direction, X.gp); });
}
LNR_HD inline void g_create_anchor_pair(GVec<u64> &g_hs, GVec<u64> &a1, GVec<u64> &a2, int shape_len, u64 rvcp, u64 gs1, u64 ge1, u64 gs2, u64 ge2, GapCtx &X) {
    g_hs_blocks(g_hs, shape_len, X.ls->st, [&](int p1, int p2, int k) {
        g_set_anchors(g_hs, a1, p1, p2, k, rvcp, 0, 0, gs1, ge1, 1, X.gp);
        g_set_anchors(g_hs, a2, p1, p2, k, rvcp, 0, 0, gs2, ge2, -1, X.gp);
    });
}
LNR_HD inline void c_create_anchors2(GVec<u64> &g_hs, GVec<u64> &out, i64 lower, i64 upper, SortStack &st) {
    int p1 = 0, p2 = 0;
    ref_sort(g_hs.p, (long)g_hs.n, [](const u64 &a, const u64 &b) { return a < b; }, st);
    for (int k = 1; k < (int)g_hs.n; k++) {
        u64 t = g_hs_xt(g_hs[(u32)k] ^ g_hs[(u32)k - 1]);
        if (t == 0) continue;
        if (t == 1) { p2 = k; continue; }
        for (int i = p1; i < p2; i++) {
            i64 x = (i64)(g_hs[(u32)i] & ((1ULL << 30) - 1));
            for (int j = p2; j < k; j++) {
                i64 y = (i64)(g_hs[(u32)j] & ((1ULL << 30) - 1));
                if (lower <= x - y && x - y < upper) out.push(canc_make(g_hs[(u32)i], g_hs[(u32)j]));
            }
        }
        p1 = k; p2 = k;
    }
}

// ---- chain scores (gap_util.cpp:966-1175, 2126-2162)
LNR_HD inline int gap_anchor_score1(u64 a1, u64 a2) {
    i64 dy = (i64)ganc_y(a1) - (i64)ganc_y(a2), dx = (i64)ganc_x(a1) - (i64)ganc_x(a2);
    if (dy < 0 || ganc_strand(a1 ^ a2) || (gabs(dx) < 8 && dx != dy)) return -10000;
    i64 da = gabs((i64)(ganc_stranchor(a2) - ganc_stranchor(a1)));
    i64 derr = (100 * da) / (dy > 50 ? dy : 50);
    int s_derr = derr < 10 ? 0 : (derr < 15 ? (int)(10 + 2 * derr) : (int)(derr * derr / 10 + 40));
    int s_dy = dy < 100 ? (int)(dy / 4) : (dy < 200 ? (int)(dy / 3 - 9) : (int)(dy - 145));
    return 100 - s_dy - s_derr;
}
LNR_HD inline int gap_anchor_score2(u64 a1, u64 a2) {
    i64 dy = (i64)ganc_y(a1) - (i64)ganc_y(a2), dx = (i64)ganc_x(a1) - (i64)ganc_x(a2);
    if (dy < 0 || ganc_strand(a1 ^ a2) || ((gabs(dx) < 8 || gabs(dy) < 8) && dx != dy)) return -10000;
    i64 da = gabs((i64)(ganc_stranchor(a2) - ganc_stranchor(a1)));
    i64 m = dx > dy ? dx : dy; if (m < 50) m = 50;
    i64 derr = (100 * da) / m;
    int s_derr = derr < 5 ? (int)(4 * derr) : (derr < 10 ? (int)(6 * derr - 10) : (int)(derr * derr - 5 * derr));
    return 100 - (int)(dy * (dy + 300) / 300) - s_derr;
}
LNR_HD inline int gap_clip_score(u64 a1, u64 a2) {                                        // getExtendClipScore
    i64 dy = (i64)ganc_y(a1) - (i64)ganc_y(a2), dx = (i64)ganc_x(a1) - (i64)ganc_x(a2);
    if (dy <= 0 || ganc_strand(a1 ^ a2) || ((gabs(dx) < 3 || gabs(dy) < 3) && dx != dy)) return -10000;
    i64 da = gabs((i64)(ganc_stranchor(a2) - ganc_stranchor(a1)));
    int s_da = da < 2 ? (int)(30 + 5 * da) : (da < 5 ? (int)(36 + 2 * da) : (int)(41 + da));
    return 100 - (int)(dy * (12 * dy + 650) / 450) - s_da;
}
LNR_HD inline int chain_block_dxdy(u64 c11, u64 c12, u64 c21, u64 c22, u64 L, int strand, i64 &dx, i64 &dy) {   // getChainBlockDxDy cluster_util.cpp:774-808
    if (cord_strand(c11) != (u64)strand) {
        if (cord_strand(c22) != (u64)strand) { dy = (i64)(cord_y(c21) - cord_y(c12)); dx = (i64)(cord_x(c21) - cord_x(c12)); }
        else { dy = (i64)(L - cord_y(c12) - 1 - cord_y(c22)); dx = (i64)(cord_x(c11) - cord_x(c22)); }
    } else {
        if (cord_strand(c22) != (u64)strand) { dy = (i64)(cord_y(c11) - L + 1 + cord_y(c21)); dx = (i64)(cord_x(c11) - cord_x(c22)); }
        else { dy = (i64)(cord_y(c11) - cord_y(c22)); dx = (i64)(cord_x(c11) - cord_x(c22)); }
    }
    return (int)cord_strand(c11 ^ c22);
}
LNR_HD inline int gap_block_score2(u64 c11, u64 c12, u64 c21, u64 c22, u64 L, int strand) {
    i64 dx, dy;
    int f_type = chain_block_dxdy(c11, c12, c21, c22, L, strand, dx, dy);
    i64 dx_ = gabs(dx), dy_ = gabs(dy), da = dx - dy;
    if (dx < -40 || dy < -40) return (int)0x80000000;
    i64 s_dy = dy_ > 300 ? dy_ / 4 - 25 : dy_ / 6, s_dx = dx_ > 300 ? dx_ / 4 - 25 : dx_ / 6;
    if (f_type == 1) return (int)(80 - s_dy);
    i64 m1 = dx_ / 4 > 50 ? dx_ / 4 : 50, m2 = dy / 4 > 50 ? dy / 4 : 50;
    if (da < -m1) return dx > -50 ? (int)(80 - s_dx) : (int)(40 - s_dy);
    if (da > m2) return (int)(80 - s_dy);
    return (int)(100 - s_dy);
}
LNR_HD inline int gap_block_score3(u64 c11, u64 c12, u64 c21, u64 c22, u64 L, int strand) {
    i64 dx, dy;
    int f_type = chain_block_dxdy(c11, c12, c21, c22, L, strand, dx, dy);
    i64 dx_ = gabs(dx), dy_ = gabs(dy), da = dx - dy;
    if (dx < 0 || dy < 0) return (int)0x80000000;
    i64 s_dy = dy_ > 300 ? dy_ / 4 - 25 : dy_ / 6;
    if (f_type == 1) return (int)(20 - s_dy);
    i64 m = dx_ > dy_ ? dx_ : dy_; i64 m100 = m > 100 ? m : 100;
    i64 r = 100 * gabs(da) / m100;
    i64 s_da = da < 15 ? r * (r + 20) / 40 : (da < 30 ? r * (r + 50) / 45 : r * (r + 100) / 45);
    return (int)(100 - s_da - m * (m + 450) / 2000);
}

// ---- X-drop on a chain by gap lengths (dropChainGapX gap_util.cpp:757-803), on tiles
LNR_HD inline void drop_chain_gap_x(GVec<u64> &ch, int direction, const GapParms &gp) {
    int n = (int)ch.n;
    if (direction == 1) {
        for (int i = 1; i < n; i++) {
            u32 di = i + 1 >= gp.thd_dcgx_window_size ? (u32)gp.thd_dcgx_window_size : 1u;
            if ((i64)(cord_x(ch[(u32)i]) - cord_x(ch[(u32)i - 1])) > gp.thd_dcgx_Xdrop_peak || (i64)(cord_x(ch[(u32)i]) - cord_x(ch[(u32)i + 1 - di])) > gp.thd_dcgx_Xdrop_sum ||
                (i64)(cord_y(ch[(u32)i]) - cord_y(ch[(u32)i - 1])) > gp.thd_dcgx_Xdrop_peak || (i64)(cord_y(ch[(u32)i]) - cord_y(ch[(u32)i + 1 - di])) > gp.thd_dcgx_Xdrop_sum) { ch.n = (u32)i; return; }
        }
    } else if (direction == -1) {
        for (int i = n - 2; i > 0; i--) {
            u32 di = n - i >= gp.thd_dcgx_window_size ? (u32)gp.thd_dcgx_window_size : 1u;
            if ((i64)(cord_x(ch[(u32)i + 1]) - cord_x(ch[(u32)i])) > gp.thd_dcgx_Xdrop_peak || (i64)(cord_x(ch[(u32)i + di - 1]) - cord_x(ch[(u32)i])) > gp.thd_dcgx_Xdrop_sum ||
                (i64)(cord_y(ch[(u32)i + 1]) - cord_y(ch[(u32)i])) > gp.thd_dcgx_Xdrop_peak || (i64)(cord_y(ch[(u32)i + di - 1]) - cord_y(ch[(u32)i])) > gp.thd_dcgx_Xdrop_sum) { ch.erase(0, (u32)i + 1); return; }
        }
    }
}

}  // namespace lnr
