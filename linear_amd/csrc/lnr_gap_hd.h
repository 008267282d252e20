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

// =================================================================== chains and tiles ====
// chainAnchorsBase (cluster_util.cpp:445-462) = getBestChains (:53-111) + the traceback of lnr_hd.h; every chain becomes tiles, the
// last tile of a chain carries the end sign (g_CreateChainsFromAnchors_ gap_util.cpp:1207-1216)
struct TileSink {
    const u64 *anchors; GVec<u64> *tiles; u32 first_len, nchains; bool to_tiles;
    LNR_HD void emit(const i32 *idx, const i32 *sc, u32 n) {
        (void)sc;
        for (u32 k = 0; k < n; k++) tiles->push(to_tiles ? ganc_tile(anchors[idx[k]]) : anchors[idx[k]]);
        if (to_tiles) set_tile_end(tiles->back());
        if (nchains == 0) first_len = n;
        nchains++;
    }
};
template <class Score>
LNR_HD inline void gap_chain_anchors(const u64 *anchors, u32 n, GVec<u64> &out, bool to_tiles, u32 depth, u64 dx_depth, int bestn, int min_len, int abort_score, Score score, GapCtx &X, bool first_only = false) {
    if (n < 2) return;
    u64 m0 = X.ar->mark();
    Rec r;
    i32 *blk = (i32 *)X.ar->get((u64)n * 9 * sizeof(i32));
    r.score = blk; r.score2 = blk + n; r.len = blk + 2 * n; r.p2 = blk + 3 * n; r.root = blk + 4 * n; r.leaf = blk + 5 * n;
    i32 *chain = blk + 6 * n, *chain_sc = blk + 7 * n, *cnt = blk + 8 * n;
    if (X.ar->ovf) return;
    // (the reference sizes the records without clearing them and sets only record 0's score, length and predecessor: root and leaf of
    // record 0 are whatever the allocator left -- 0 in practice, as here)
    for (u32 i = 0; i < n; i++) { r.score[i] = 0; r.score2[i] = 0; r.len[i] = 0; r.p2[i] = 0; r.root[i] = 0; r.leaf[i] = 0; }
    r.score[0] = 0; r.len[0] = 1; r.p2[0] = -1;
    for (int i = 0; i < (int)n; i++) {
        int j_str = i - (int)depth < 0 ? 0 : i - (int)depth, max_j = i, best = -1;
        for (int j = i - 1; j >= 0 && (j >= j_str || ganc_x(anchors[j]) - ganc_x(anchors[i]) < dx_depth); j--) {
            int sc = score(anchors[j], anchors[i]);
            if (sc > 0 && sc + r.score[j] >= best) { max_j = j; best = sc + r.score[j]; }
        }
        if (best > 0) { r.p2[i] = max_j; r.score[i] = best; r.len[i] = r.len[max_j] + 1; r.score2[i] = best; r.root[i] = r.root[max_j]; r.leaf[i] = 1; r.leaf[max_j] = 0; }
        else { r.p2[i] = -1; r.score[i] = 0; r.len[i] = 1; r.score2[i] = 0; r.root[i] = i; r.leaf[i] = 1; }
    }
    // the output may grow while the records live above it in the arena: collect into a vector allocated before them
    TileSink sink; sink.anchors = anchors; sink.tiles = &out; sink.first_len = 0; sink.nchains = 0; sink.to_tiles = to_tiles;
    out.reserve(out.n + n);
    traceback(r, n, sink, chain, chain_sc, cnt, min_len, abort_score, bestn, 0.7f, *X.ls);
    (void)first_only;
    (void)m0;   // (not released: `out` may have been re-allocated above the mark)
}
// gather_blocks_ (pmpfinder.cpp:1484-1530) on tiles: a block ends at the tile-end sign (chainTiles gap_util.cpp:1177-1189, f_set_end 0)
LNR_HD inline void gap_gather_tile_blocks(const u64 *t, u32 n, GVec<UP> &sep, u64 L, u64 large_gap) {
    if (n < 2) return;
    u32 p_str = 0;
    for (u32 i = 1; i < n; i++)
        if (is_tile_end(t[i - 1]) || !consecutive(t[i - 1], t[i], large_gap)) { UP q; q.first = p_str; q.second = i; sep.push(q); p_str = i; }
    UP q; q.first = p_str; q.second = n; sep.push(q);
    (void)L;
}
LNR_HD inline void gap_best_chains2(const u64 *rec_, const UP *sep, const i32 *sep_score, u32 nb, Rec r, u64 L, int fn, int strand) {   // getBestChains2 cluster_util.cpp:469-526
    for (u32 i = 0; i < nb; i++) {
        int j_str = (int)i - 20 < 0 ? 0 : (int)i - 20, max_j = (int)i, best = -1;
        for (u32 j = (u32)j_str; j < i; j++) {
            int sc = fn == 3 ? gap_block_score3(rec_[sep[j].first], rec_[sep[j].second - 1], rec_[sep[i].first], rec_[sep[i].second - 1], L, strand)
                             : gap_block_score2(rec_[sep[j].first], rec_[sep[j].second - 1], rec_[sep[i].first], rec_[sep[i].second - 1], L, strand);
            if (sc > 0 && sc + r.score[j] + sep_score[i] >= best) { max_j = (int)j; best = sc + r.score[j] + sep_score[i]; }
        }
        if (best > 0) { r.p2[i] = max_j; r.score[i] = best; r.len[i] = (i32)(sep[i].second - sep[i].first) + r.len[max_j]; r.score2[i] = best; r.root[i] = r.root[max_j]; r.leaf[i] = 1; r.leaf[max_j] = 0; }
        else { r.p2[i] = -1; r.score[i] = sep_score[i]; r.len[i] = (i32)(sep[i].second - sep[i].first); r.score2[i] = r.score[i]; r.root[i] = (i32)i; r.leaf[i] = 1; }
    }
}
// chainBlocksCords (cluster_util.cpp:1068-1102) for tiles: both strands, the better one, its major chains (_filterBlocksCords :865-931 without
// header, the tile-end sign as the block end)
LNR_HD inline void gap_chain_blocks_cords(GVec<u64> &tiles, GVec<UP> &sep, u64 L, u32 init_score, u64 major_limit, GapCtx &X) {
    u32 nb = sep.n;
    UP *sp[2]; BlockSink cc[2];
    for (int strand = 0; strand < 2; strand++) {
        sp[strand] = (UP *)X.ar->get((u64)(nb + 1) * sizeof(UP));
        i32 *ib = (i32 *)X.ar->get((u64)(nb + 1) * 12 * sizeof(i32));
        UP *el = (UP *)X.ar->get((u64)(nb + 2) * 2 * sizeof(UP));
        if (X.ar->ovf) return;
        for (u32 i = 0; i < nb; i++) sp[strand][i] = sep[i];
        UP *s = sp[strand];
        const u64 *cords = tiles.p;
        if (strand)
            ref_sort(s, (long)nb, [cords, L](const UP &a, const UP &b) {
                u64 y1 = !cord_strand(cords[a.first]) ? L - 1 - cord_y(cords[a.second - 1]) : cord_y(cords[a.first]);
                u64 y2 = !cord_strand(cords[b.first]) ? L - 1 - cord_y(cords[b.second - 1]) : cord_y(cords[b.first]);
                return y1 > y2;
            }, X.ls->st);
        else
            ref_sort(s, (long)nb, [cords, L](const UP &a, const UP &b) {
                u64 y1 = cord_strand(cords[a.first]) ? L - 1 - cord_y(cords[a.second - 1]) : cord_y(cords[a.first]);
                u64 y2 = cord_strand(cords[b.first]) ? L - 1 - cord_y(cords[b.second - 1]) : cord_y(cords[b.first]);
                return y1 > y2;
            }, X.ls->st);
        i32 *sc = ib;
        for (u32 i = 0; i < nb; i++) sc[i] = (i32)((s[i].second - s[i].first) * init_score);
        BlockSink &k = cc[strand];
        k.el = el; k.off = ib + (nb + 1); k.nchains = 0; k.nel = 0; k.cap = (nb + 2) * 2; k.ovf = &X.ar->ovf; k.first_len = 0; k.off[0] = 0; k.elements = s;
        if (nb >= 2) {                                                                      // chainBlocksBase :533-577 (f_sort 0)
            Rec r; i32 *q = ib + 2 * (nb + 1);
            r.score = q; r.score2 = q + (nb + 1); r.len = q + 2 * (nb + 1); r.p2 = q + 3 * (nb + 1); r.root = q + 4 * (nb + 1); r.leaf = q + 5 * (nb + 1);
            i32 *chain = q + 6 * (nb + 1), *chain_sc = q + 7 * (nb + 1), *cnt = q + 8 * (nb + 1);
            for (u32 i = 0; i < nb; i++) { r.score[i] = 0; r.score2[i] = 0; r.len[i] = 0; r.p2[i] = 0; r.root[i] = 0; r.leaf[i] = 0; }
            gap_best_chains2(cords, s, sc, nb, r, L, X.gp.chn2_fn, strand);
            traceback(r, nb, k, chain, chain_sc, cnt, X.gp.chn2_min_len, X.gp.chn2_abort, 3, 0.7f, *X.ls);
        }
    }
    int best = best_strand(cc[0], cc[1]);
    for (u32 i = 0; i < nb; i++) sep[i] = sp[best][i];
    BlockSink &ch = cc[best];
    revert_chain_block_strand(ch, tiles.p, best);
    if (ch.nchains == 0) return;
    u64 *out = (u64 *)X.ar->get((u64)(tiles.n + 1) * 8);
    if (X.ar->ovf) return;
    u32 n = 0; u64 len_current = 0;
    for (i32 i = ch.off[0]; i < ch.off[1]; i++) {
        for (u64 j = ch.el[i].first; j < ch.el[i].second; j++) out[n++] = tiles[(u32)j] & ~TILE_END;
        len_current += ch.el[i].second - ch.el[i].first;
    }
    out[n - 1] |= TILE_END;
    float bound = 0.8 * len_current;
    u32 major_n = 1;
    for (u32 c = 1; c < ch.nchains && major_n < major_limit; c++) {
        len_current = 0;
        for (i32 j = ch.off[c]; j < ch.off[c + 1]; j++) len_current += ch.el[j].second - ch.el[j].first;
        if ((float)len_current > bound) {
            ++major_n;
            for (i32 j = ch.off[c]; j < ch.off[c + 1]; j++) for (u64 k2 = ch.el[j].first; k2 < ch.el[j].second; k2++) out[n++] = tiles[(u32)k2] & ~TILE_END;
            out[n - 1] |= TILE_END;
        }
    }
    for (u32 i = 0; i < n; i++) tiles[i] = out[i];
    tiles.n = n;
}
LNR_HD inline void gap_chain_tiles(GVec<u64> &tiles, u64 L, u64 gap_size, GapCtx &X) {       // chainTiles gap_util.cpp:1177-1189
    GVec<UP> sep; sep.init(X.ar);
    gap_gather_tile_blocks(tiles.p, tiles.n, sep, L, gap_size);
    gap_chain_blocks_cords(tiles, sep, L, 64, X.gp.thd_cts_major_limit, X);
}
LNR_HD inline void g_chains_from_anchors(GVec<u64> &anchors, GVec<u64> &tiles, u64 L, GapCtx &X) {   // g_CreateChainsFromAnchors_ gap_util.cpp:1191-1222
    ref_sort(anchors.p, (long)anchors.n, [](const u64 &a, const u64 &b) { return ganc_x(a) > ganc_x(b); }, X.ls->st);
    int fn = X.gp.chn1_fn;
    gap_chain_anchors(anchors.p, anchors.n, tiles, true, 20, 80, 20, X.gp.chn1_min_len, X.gp.chn1_abort, [fn](u64 a, u64 b) { return fn == 2 ? gap_anchor_score2(a, b) : gap_anchor_score1(a, b); }, X);
    gap_chain_tiles(tiles, L, 100, X);
}
struct IPair { int first, second; };
LNR_HD inline IPair closest_extension_chain(GVec<u64> &t, u64 gap_str, u64 gap_end, bool f_erase, const GapParms &gp) {   // getClosestExtensionChain_ gap_util.cpp:1227-1270
    int pre_i = 0;
    IPair z; z.first = 0; z.second = 0;
    for (int i = 0; i < (int)t.n; i++) {
        if (!is_tile_end(t[(u32)i])) continue;
        i64 danchor = 0, dx = 0, dy = 0;
        if (gp.direction < 0) { dy = (i64)cord_y(gap_end) - (i64)cord_y(t[(u32)i]); dx = (i64)cord_x(gap_end) - (i64)cord_x(t[(u32)i]); danchor = dx - dy; }
        else if (gp.direction > 0) { dy = (i64)cord_y(t[(u32)pre_i]) - (i64)cord_y(gap_str); dx = (i64)cord_x(t[(u32)pre_i]) - (i64)cord_x(gap_str); danchor = dx - dy; }
        i64 m = gabs(dy) > gabs(dx) ? gabs(dy) : gabs(dx);
        if (gabs(danchor) < gp.thd_ctfas2_connect_danchor && m < gp.thd_ctfas2_connect_dy_dx) {
            if (f_erase) { t.erase(0, (u32)pre_i); t.n = (u32)(i + 1 - pre_i); z.second = (int)t.n; return z; }
            z.first = pre_i; z.second = i + 1; return z;
        }
        pre_i = i + 1;
    }
    if (f_erase) t.clear();
    return z;
}

}  // namespace lnr
