// lnr_gap_hd.h -- the gap re-mapper (mapGaps / reformCords, gap.cpp / gap_util.cpp / cords.cpp:504-687) as host + device
// functions of the product, in the idiom of lnr_hd.h: arrays from a per-read arena (mark / release per gap), the tie-sensitive sorts
// through ref_sort.h, the chain traceback and the block chaining through the forms lnr_hd.h already has.  One read is one serial walk
// over its gaps (the tiles of a gap are inserted into the cord list before the next gap is looked at).  k_gap (lnr_kernels.hip) runs
// it with one wave per read: every lane executes this code on the same data, and the loops marked `coop` deal their iterations over
// the 64 lanes (chain DP, k-mer join).  The same source compiled by g++ (tests/host_shim.cpp) is what the CPU tests check function
// by function against the oracle (tests/test_gap_shim_cpu.py).  Every function cites the reference lines it follows.  DESIGN.md 5c.
#pragma once
#include "lnr_hd.h"
#if defined(__HIPCC__)
#include "lnr_wave.h"
#endif

namespace lnr {

// ---- arena + growable array of the gap path: temporaries of one gap are released together (mark / release)
struct GArena {
    char *base; u64 off, cap, hw, want = 0; int ovf;   // want: the request that did not fit (what the next launch orders the flagged reads by)
    // the first 64 bytes are the dump: what a vector that could not be allocated points at (capacity 0), so that a stray element
    // access after an overflow stays inside the arena.  ovf: 1 = out of memory, 2 = over the work budget (both: the read is redone
    // with a larger arena / by the cooperative form, or reported)
    LNR_HD void init(void *b, u64 c) { base = (char *)b; off = 64; cap = c; ovf = c < 64 ? 1 : 0; hw = 0; }
    LNR_HD void *get(u64 bytes) {
        bytes = (bytes + 15) & ~15ULL;
        if (ovf || off + bytes > cap) { if (!ovf) { ovf = 1; want = bytes; } return (void *)base; }   // (raw users check ovf before touching the block; GVec falls back to capacity 0)
        void *r = base + off; off += bytes; if (off > hw) hw = off; return r;
    }
    LNR_HD u64 mark() const { return off; }
    LNR_HD void release(u64 m) { off = m; }
};
template <class T> struct GVec {
    T *p; u32 n, cap; GArena *ar;
    LNR_HD void init(GArena *a, u32 c0 = 16) { ar = a; n = 0; cap = c0; p = (T *)a->get((u64)c0 * sizeof(T)); if (a->ovf) cap = 0; }
    LNR_HD void reserve(u32 c) {
        if (c <= cap) return;
        u32 nc = cap * 2 > c ? cap * 2 : c;
        T *q = (T *)ar->get((u64)nc * sizeof(T));
        if (ar->ovf) return;
        for (u32 i = 0; i < n; i++) q[i] = p[i];
        p = q; cap = nc;
    }
    LNR_HD void push(const T &v) { reserve(n + 1); if (n < cap) p[n++] = v; }
    // (element access is clamped: after an overflow the vectors are short or empty and the code that follows may index past them
    //  before it reaches the next overflow check; results of such a read are discarded)
    LNR_HD T &operator[](u32 i) { return p[i < cap ? i : 0]; }
    LNR_HD const T &operator[](u32 i) const { return p[i < cap ? i : 0]; }
    LNR_HD T &back() { return p[n ? n - 1 : 0]; }
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
struct GapTeam;
struct GSeq { const u8 *p; u64 len; };                    // bases with >= 64 zero bytes behind the end
struct GapCtx {                                           // one read
    GArena *ar; LeaderScratch *ls;
    GSeq read, com;                                       // the read and its reverse complement (both padded copies)
    const u8 *g; const u64 *seq_off; const u64 *seq_len;  // genome
    FeatView f1[2]; GenomeFeat gf;
    GapParms gp;
    u64 work = 0, work_cap = ~0ULL;                       // pair evaluations of the chain DPs so far / the budget (over it: ar->ovf = 2)
    u64 deadline = 0;                                     // device, first launch: wall_clock64() after which the read is given up and left to the team launch (ovf = 2); 0 = none
    int coop = 0;                                         // device: all 64 lanes of the wave run this read together (k_gap, second launch)
    int team = 0; struct GapTeam *tm = nullptr;           // device: helper waves of the workgroup for the long rows of the chain DP (k_gap_team)
    int hand = 0;                                         // device: a single wave of the first stage -- a later stage (teams, larger arenas) redoes what it gives up
#ifdef LNR_GAP_DEVPROF
    unsigned long long prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // ticks per phase (diagnostic build, tools/measure/gap_prof.sh)
    unsigned long long dp_t = 0, dp_n = 0, dp_mode = 0, dp_fn = 0;          // the read's longest chain DP: ticks, anchors, 1 = by columns, score function
#endif
    LNR_HD GSeq ref(u64 id) const { GSeq s; s.p = g + seq_off[id]; s.len = seq_len[id]; return s; }
};

// The first launch runs one wave per read and ends when its slowest read does: a read that is still busy after the deadline is abandoned there
// (like one that outgrew its arena: ovf = 2, its apxMap cords stay) and redone by the team launch, which has idle CUs to spare while the handful
// of truly heavy reads set its duration.  Which launch a read ends up in does not change its result.
LNR_HD inline bool gap_late(GapCtx &X) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (X.deadline && !X.ar->ovf && wall_clock64() > X.deadline) X.ar->ovf = 2;
#endif
    return X.ar->ovf != 0;
}
#if defined(LNR_GAP_DEVPROF) && defined(__HIP_DEVICE_COMPILE__)
struct GpScope { GapCtx &X; int i; unsigned long long t0; __device__ GpScope(GapCtx &x, int k) : X(x), i(k), t0(wall_clock64()) {} __device__ ~GpScope() { X.prof[i] += wall_clock64() - t0; } };
#define GP(X, k) GpScope _gp_scope(X, k)
#define GP2(X, k) GpScope _gp_scope2(X, k)
#else
#define GP(X, k) do {} while (0)
#define GP2(X, k) do {} while (0)
#endif

// the comparators of the gap path's sorts as ONE type (the team form of the sort hands it to the helper waves through LDS)
struct GapCmp {
    int kind; u64 mask;      // 1 stranchor ascending (filterGapAnchors), 2 x descending (chains), 3 masked word ascending (k-mer list), 4 word ascending
    LNR_HD bool operator()(const u64 &a, const u64 &b) const {
        switch (kind) {
            case 1: return ganc_stranchor(a) < ganc_stranchor(b);
            case 2: return ganc_x(a) > ganc_x(b);
            case 3: return (a & mask) < (b & mask);
            default: return a < b;
        }
    }
};
// std::sort of the gap path.  Host and lane-per-read form: ref_sort (libstdc++'s introsort, serial).  Wave-per-read form: the same
// algorithm with the partitions of the large ranges done by all 64 lanes and the small ranges finished one per lane
// (gap_sort_wave, lnr_kernels.hip: the list formulation of ref_sort.h) -- same permutation, ties included.
#if defined(__HIPCC__)
template <class T, class Comp> __device__ void gap_sort_wave(T *a, u32 n, Comp comp, GapCtx &X);
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// Short arrays in the wave-per-read form: every lane ranks the (up to two) elements it holds by counting -- the elements before it in the
// order plus the equivalent ones with a smaller index, i.e. the STABLE order.  That is what std::sort produces up to 16 elements (a plain
// insertion sort there) for any comparator, and for any length when equivalent elements are identical words (no tie to break: every sort
// gives the same array).  The serial emulation these arrays went through before pays a memory round trip per comparison.
template <class Comp> __device__ inline void gap_rank_sort(u64 *a, u32 n, Comp comp) {       // n <= 128
    const u32 lane = threadIdx.x & 63;
    u64 e0 = lane < n ? a[lane] : 0, e1 = lane + 64 < n ? a[lane + 64] : 0;
    u32 r0 = 0, r1 = 0;
    const u32 n0 = n < 64 ? n : 64;
    for (u32 t = 0; t < n0; t++) {
        u64 v = __shfl(e0, (int)t);
        r0 += (comp(v, e0) || (!comp(e0, v) && t < lane)) ? 1u : 0u;
        r1 += (comp(v, e1) || !comp(e1, v)) ? 1u : 0u;                 // (t < lane + 64 always)
    }
    for (u32 t = 64; t < n; t++) {
        u64 v = __shfl(e1, (int)(t - 64));
        r0 += comp(v, e0) ? 1u : 0u;                                   // (t > lane always)
        r1 += (comp(v, e1) || (!comp(e1, v) && t - 64 < lane)) ? 1u : 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    if (lane < n) a[r0] = e0;
    if (lane + 64 < n) a[r1] = e1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
#endif
template <class T, class Comp> LNR_HD inline void gap_sort(T *a, long n, Comp comp, GapCtx &X) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (X.coop && n > 96) { gap_sort_wave(a, (u32)n, comp, X); return; }
#endif
    ref_sort(a, n, comp, X.ls->st);
}
inline LNR_HD void gap_sort(u64 *a, long n, GapCmp comp, GapCtx &X) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (X.coop && n >= 2 && (n <= 16 || (comp.kind == 4 && n <= 128))) { gap_rank_sort(a, (u32)n, comp); return; }
    if (X.coop && n > 96) { gap_sort_wave(a, (u32)n, comp, X); return; }
#endif
    ref_sort(a, n, comp, X.ls->st);
}

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
// Closed form of the rolling state (device, wave-per-read form: one lane per pushed k-mer).  With v = the bases of the window at k
// (values 0..4, N = 4), s = span:   hValue  = ((sum_{i>=1} v[k+s-1-i] 4^i) mod 4^s) + v[k+s-1]      (the newest base is added after the mask: an N carries)
//                                    crhValue = sum_i ((3 - v[k+i]) & 3) << 2i, all ones above bit 2s when the newest base is N ((3 - 4) << (2s - 2))
//                                    x        = C + 2 * sum(v), C = x_init - 2 * sum(v[str .. str+s-2])  (= -3s when hashInit skipped nothing)
// -- the recurrences of hashNextV (shape_extend.cpp:231-243) unrolled; valid from the first iteration on when hashInit found its N-free window
// at the stream's first base (skip 0: the init digits ARE the window's older digits); a stream that starts inside an N run keeps the serial loop.
LNR_HD inline u64 kmer_h_closed(const u8 *w, u32 span) {
    u64 S = 0;
    for (u32 i = 0; i < span; i++) S = S * 4 + w[i];
    u64 nw = w[span - 1];
    return ((S - nw) & ((1ULL << (2 * span)) - 1)) + nw;
}
LNR_HD inline void g_kmer_stream(const GSeq &seq, GVec<u64> &g_hs, u64 str, u64 end, int shape_len, int step, u64 type, int coop = 0) {   // g_mapHs_kmer_ gap_util.cpp:632-662
    if (seq.len < (u64)shape_len) return;
    GShape sh; sh.span = (u32)shape_len;
    u64 skip = gshape_init(sh, seq.p + str);
    u64 mask = (1ULL << (2 * sh.span - 2)) - 1;
    int count = 0;
    u64 lim = end < seq.len - (u64)shape_len ? end : seq.len - (u64)shape_len;
#if defined(__HIP_DEVICE_COMPILE__)
    if (coop && skip == 0 && lim > str) {
        const u32 span = sh.span, lane = threadIdx.x & 63;
        u32 M = (u32)((lim - str) / (u64)step), n0 = g_hs.n;
        if (!M) return;
        g_hs.reserve(n0 + M);
        if (n0 + M > g_hs.cap) return;
        const int C = -3 * (int)span;
        for (u32 m = lane; m < M; m += 64) {
            u64 k = str + (u64)(m + 1) * (u64)step - 1;
            const u8 *w = seq.p + k;
            u64 S = 0, crh = 0; int ws = 0;
            for (u32 i = 0; i < span; i++) { u64 v = w[i]; S = S * 4 + v; crh |= ((3 - v) & 3) << (2 * i); ws += (int)v; }
            u64 nw = w[span - 1];
            u64 h = ((S - nw) & ((1ULL << (2 * span)) - 1)) + nw;
            if (nw == 4) crh |= ~0ULL << (2 * span);
            u64 strand = (C + 2 * ws) < 0 ? 1 : 0;
            g_hs.p[n0 + m] = g_hs_make(strand ? crh : h, type, strand, k);
        }
        g_hs.n = n0 + M;
        return;
    }
#endif
    (void)coop; (void)skip;
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
LNR_HD inline void g_stream(const GSeq &ref, const GSeq &read, GVec<u64> &g_hs, u64 gap_str, u64 gap_end, u32 shape_len, int step1, int step2, int coop = 0) {   // g_stream_ :1663-1688
    u64 gs_str = cord_x(gap_str), gs_end = cord_x(gap_end), gr_str = cord_y(gap_str), gr_end = cord_y(gap_end);
    if (cord_strand(gap_str)) { u64 a = read.len - gr_str - 1, b = read.len - gr_end - 1; gr_str = b; gr_end = a; }
    g_kmer_stream(ref, g_hs, gs_str, gs_end, (int)shape_len, step1, 0, coop);
    g_kmer_stream(read, g_hs, gr_str, gr_end, (int)shape_len, step2, 1, coop);
}
LNR_HD inline void c_stream(const GSeq &seq, GVec<u64> &g_hs, u64 sq_str, u64 sq_end, int step, int shape_len, u64 type, int coop = 0) {   // c_stream_ :1694-1716
    if (seq.len < (u64)shape_len) return;
    u32 span = (u32)shape_len;
    u64 lim = sq_end < seq.len - (u64)shape_len ? sq_end : seq.len - (u64)shape_len;
#if defined(__HIP_DEVICE_COMPILE__)
    if (coop && lim > sq_str) {                              // (no hashInit here: the closed form holds from the first iteration on)
        const u32 lane = threadIdx.x & 63;
        u32 M = (u32)((lim - sq_str) / (u64)step), n0 = g_hs.n;
        if (!M) return;
        g_hs.reserve(n0 + M);
        if (n0 + M > g_hs.cap) return;
        for (u32 m = lane; m < M; m += 64) {
            u64 k = sq_str + (u64)(m + 1) * (u64)step - 1;
            g_hs.p[n0 + m] = g_hs_make(kmer_h_closed(seq.p + k, span), type, 0, k);
        }
        g_hs.n = n0 + M;
        return;
    }
#endif
    (void)coop;
    u64 h = 0, mask = (1ULL << (2 * span - 2)) - 1;
    for (u32 i = 0; i < span - 1; ++i) h = (h << 2) + seq.p[sq_str + i];
    int count = 0;
    for (u64 k = sq_str; k < lim; k++) {
        h = ((h & mask) << 2) + seq.p[k + span - 1];
        if (++count == step) { g_hs.push(g_hs_make(h, type, 0, k)); count = 0; }
    }
}

// ---- anchors from the sorted k-mer list (gap_util.cpp:669-752, 1596-1661, 1818-1853)
// keep(a): is the anchor of a k-mer pair inside the diagonal band of the gap (direction 0: [lower, upper); else the band that widens
// with the distance from the end the extension starts at)
struct GAncBand {
    int direction; i64 lower, upper, y_ref, base, d_anchor; u64 strand;
    LNR_HD bool keep(u64 a) const {
        i64 t = (i64)ganc_stranchor(a);
        if (direction == 0) return t < upper && t >= lower;
        i64 dy = direction < 0 ? y_ref - (i64)ganc_y(a) : (i64)ganc_y(a) - y_ref;
        if (dy < 0 || (ganc_strand(a) ^ strand)) return false;
        i64 acc = (dy >> 7) * d_anchor; if (acc < 50) acc = 50;
        i64 lo = base - acc; if (lo < 0) lo = 0;
        return t < base + acc && t >= lo;
    }
};
#if defined(__HIPCC__)
__device__ inline bool gap_join_team(GapCtx &X, const GVec<u64> &g_hs, GVec<u64> &out, int p1, int p2, int k, int kind, u64 rvcp, i64 lower, i64 upper, const GAncBand *B);
#endif
LNR_HD inline void g_set_anchors(const GVec<u64> &g_hs, GVec<u64> &out, int p1, int p2, int k, u64 rvcp, i64 lower, i64 upper, u64 gap_str, u64 gap_end, int direction, const GapParms &gp, int coop = 0, GapCtx *Xp = nullptr) {
    if (out.ar->ovf) return;
    GAncBand B;
    B.direction = direction; B.lower = lower; B.upper = upper; B.strand = cord_strand(gap_str);
    B.y_ref = direction < 0 ? (i64)cord_y(gap_end) : (i64)cord_y(gap_str);
    B.base = (i64)cord2stranchor(direction < 0 ? gap_end : gap_str);
    B.d_anchor = (i64)((1LL << 7) * gp.thd_gmsa_d_anchor_rate);
#if defined(__HIP_DEVICE_COMPILE__)
    if (coop) {                                              // the block's (reference k-mer, read k-mer) pairs, 64 at a time in pair order (i major)
        if (Xp && gap_join_team(*Xp, g_hs, out, p1, p2, k, 0, rvcp, 0, 0, &B)) return;   // (a big block on a team: all waves)
        const int lane = (int)(threadIdx.x & 63);
        const u32 ni = (u32)(p2 - p1), nj = (u32)(k - p2), np = ni * nj;
        for (u32 pb = 0; pb < np; pb += 64) {
            u32 q = pb + (u32)lane;
            u32 qi = q / nj, qj = q - qi * nj;
            u64 a = q < np ? ganc_make(g_hs.p[(u32)p1 + qi], g_hs.p[(u32)p2 + qj], rvcp) : 0;
            bool kp = q < np && B.keep(a);
            u64 m = __ballot(kp);
            u32 cnt = (u32)__popcll(m);
            if (!cnt) continue;
            out.reserve(out.n + cnt);
            if (out.n + cnt > out.cap) return;
            if (kp) out.p[out.n + (u32)__popcll(m & ((1ULL << lane) - 1))] = a;
            out.n += cnt;
        }
        return;
    }
#endif
    (void)coop;
    for (int i = p1; i < p2; i++) for (int j = p2; j < k; j++) {
        u64 a = ganc_make(g_hs[(u32)i], g_hs[(u32)j], rvcp);
        if (B.keep(a)) out.push(a);
    }
}
template <class F> LNR_HD inline void g_hs_blocks(GVec<u64> &g_hs, int shape_len, GapCtx &X, F &&emit) {
    u64 mask = (1ULL << (2 * shape_len + 33)) - 1;
    { GP(X, 0); gap_sort(g_hs.p, (long)g_hs.n, GapCmp{3, mask}, X); }
    GP(X, 1);
    int p1 = 0, p2 = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    if (X.coop) {
        // 64 neighbour comparisons per load; the events (t == 1: the block's read k-mers begin; t > 1: the block ends) are then taken in
        // order -- a block is handed on only when it holds pairs (emit with an empty side does nothing)
        const u32 n = g_hs.n, lane = threadIdx.x & 63;
        for (u32 base = 1; base < n; base += 64) {
            u32 k = base + lane;
            u64 t = k < n ? g_hs_xt((g_hs.p[k] ^ g_hs.p[k - 1]) & mask) : 0;
            u64 m1 = __ballot(t == 1), ev = __ballot(t >= 1);
            while (ev) {
                int b = __builtin_ctzll(ev);
                ev &= ev - 1;
                int kk = (int)base + b;
                if ((m1 >> b) & 1) p2 = kk;
                else { if (p2 > p1 && kk > p2) emit(p1, p2, kk); p1 = kk; p2 = kk; }
            }
        }
        return;
    }
#endif
    for (int k = 1; k < (int)g_hs.n; k++) {
        u64 t = g_hs_xt((g_hs[(u32)k] ^ g_hs[(u32)k - 1]) & mask);
        if (t == 0) continue;
        if (t == 1) { p2 = k; continue; }
        emit(p1, p2, k);
        p1 = k; p2 = k;
    }
}
LNR_HD inline void g_create_anchors(GVec<u64> &g_hs, GVec<u64> &anchors, int shape_len, int direction, i64 lower, i64 upper, u64 rvcp, u64 gap_str, u64 gap_end, GapCtx &X) {
    g_hs_blocks(g_hs, shape_len, X, [&](int p1, int p2, int k) { g_set_anchors(g_hs, anchors, p1, p2, k, rvcp, lower, upper, gap_str, gap_end, direction, X.gp, X.coop, &X); });
}
LNR_HD inline void g_create_anchor_pair(GVec<u64> &g_hs, GVec<u64> &a1, GVec<u64> &a2, int shape_len, u64 rvcp, u64 gs1, u64 ge1, u64 gs2, u64 ge2, GapCtx &X) {
    g_hs_blocks(g_hs, shape_len, X, [&](int p1, int p2, int k) {
        g_set_anchors(g_hs, a1, p1, p2, k, rvcp, 0, 0, gs1, ge1, 1, X.gp, X.coop, &X);
        g_set_anchors(g_hs, a2, p1, p2, k, rvcp, 0, 0, gs2, ge2, -1, X.gp, X.coop, &X);
    });
}
LNR_HD inline void c_create_anchors2(GVec<u64> &g_hs, GVec<u64> &out, i64 lower, i64 upper, SortStack &st, GapCtx *Xp = nullptr) {
    int p1 = 0, p2 = 0;
    if (Xp) gap_sort(g_hs.p, (long)g_hs.n, GapCmp{4, 0}, *Xp);
    else ref_sort(g_hs.p, (long)g_hs.n, [](const u64 &a, const u64 &b) { return a < b; }, st);
#if defined(__HIP_DEVICE_COMPILE__)
    if (Xp && Xp->coop) {
        // the wave-per-read form: 64 neighbour comparisons per load, the blocks in order, a block's (reference k-mer, read k-mer) pairs
        // dealt over the lanes and the kept ones appended in pair order (i major, as the loops below)
        const u32 n = g_hs.n, lane = threadIdx.x & 63;
        for (u32 base = 1; base < n; base += 64) {
            u32 kq = base + lane;
            u64 t = kq < n ? g_hs_xt(g_hs.p[kq] ^ g_hs.p[kq - 1]) : 0;
            u64 m1 = __ballot(t == 1), ev = __ballot(t >= 1);
            while (ev) {
                int bb = __builtin_ctzll(ev);
                ev &= ev - 1;
                int k = (int)base + bb;
                if ((m1 >> bb) & 1) { p2 = k; continue; }
                if (out.ar->ovf) return;
                const u32 ni = (u32)(p2 - p1), nj = (u32)(k - p2);
                u32 np = p2 > p1 && k > p2 ? ni * nj : 0;
                if (np && gap_join_team(*Xp, g_hs, out, p1, p2, k, 1, 0, lower, upper, nullptr)) np = 0;   // (a big block on a team)
                for (u32 pb = 0; pb < np; pb += 64) {
                    u32 q = pb + lane;
                    u32 qi = q / nj, qj = q - qi * nj;
                    u64 hi_ = q < np ? g_hs.p[(u32)p1 + qi] : 0, hj_ = q < np ? g_hs.p[(u32)p2 + qj] : 0;
                    i64 d = (i64)(hi_ & ((1ULL << 30) - 1)) - (i64)(hj_ & ((1ULL << 30) - 1));
                    bool kp = q < np && lower <= d && d < upper;
                    u64 m = __ballot(kp);
                    u32 cnt = (u32)__popcll(m);
                    if (!cnt) continue;
                    out.reserve(out.n + cnt);
                    if (out.n + cnt > out.cap) return;
                    if (kp) out.p[out.n + (u32)__popcll(m & ((1ULL << lane) - 1))] = canc_make(hi_, hj_);
                    out.n += cnt;
                }
                p1 = k; p2 = k;
            }
        }
        return;
    }
#endif
    for (int k = 1; k < (int)g_hs.n; k++) {
        u64 t = g_hs_xt(g_hs[(u32)k] ^ g_hs[(u32)k - 1]);
        if (t == 0) continue;
        if (t == 1) { p2 = k; continue; }
        if (out.ar->ovf) return;
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
// The chain DP only asks whether a score is positive and, if so, for its value.  These forms return the score of the functions above
// whenever that is positive and some non-positive number otherwise, in 32-bit arithmetic behind two range tests (the 64-bit divide
// of the literal form is ~150 instructions on the GPU).  Score 1 is positive only for dy < 245 (s_dy < 100) and derr <= 24, i.e.
// 100 da < 25 max(dy, 50): da <= 60 (dy < 245 + 32 768 then: no wrap in s_dy either).  Score 2 only for dy <= 79 and derr <= 12; with da = |dx - dy| that bounds dx by 147 and da by 19.
// tests/test_gap_shim_cpu.py compares them with the literal forms over the whole positive region and its surroundings.
LNR_HD inline int gap_anchor_score1_pos(u64 a1, u64 a2) {
    i32 dy = (i32)ganc_y(a1) - (i32)ganc_y(a2);
    i64 dx = (i64)ganc_x(a1) - (i64)ganc_x(a2);
    if (dy < 0 || ganc_strand(a1 ^ a2) || (gabs(dx) < 8 && dx != dy)) return -1;
    u64 da64 = (u64)gabs((i64)(ganc_stranchor(a2) - ganc_stranchor(a1)));
    if (da64 >= 65536) return gap_anchor_score1(a1, a2);   // (the literal form's int conversion of derr^2 / 10 wraps from da ~ 73 000 on: reproduced, not reasoned about)
    if (da64 >= 64 || dy >= 245) return -1;
    u32 da = (u32)da64, m = (u32)(dy > 50 ? dy : 50), derr = (100u * da) / m;
    i32 s_derr = derr < 10 ? 0 : (derr < 15 ? (i32)(10 + 2 * derr) : (i32)(derr * derr / 10 + 40));
    i32 s_dy = dy < 100 ? dy / 4 : (dy < 200 ? dy / 3 - 9 : dy - 145);
    return 100 - s_dy - s_derr;
}
LNR_HD inline int gap_anchor_score2_pos(u64 a1, u64 a2) {
    i32 dy = (i32)ganc_y(a1) - (i32)ganc_y(a2);
    i64 dx = (i64)ganc_x(a1) - (i64)ganc_x(a2);
    if (dy < 0 || dy >= 128 || ganc_strand(a1 ^ a2) || ((gabs(dx) < 8 || dy < 8) && dx != dy)) return -1;
    u64 da64 = (u64)gabs((i64)(ganc_stranchor(a2) - ganc_stranchor(a1)));
    if (da64 >= 64 || gabs(dx) >= 4096) return -1;
    u32 da = (u32)da64;
    i32 m = (i32)dx > dy ? (i32)dx : dy; if (m < 50) m = 50;
    u32 derr = (100u * da) / (u32)m;
    i32 s_derr = derr < 5 ? (i32)(4 * derr) : (derr < 10 ? (i32)(6 * derr - 10) : (i32)(derr * derr - 5 * derr));
    return 100 - dy * (dy + 300) / 300 - s_derr;
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

// ---- the three DP scores as functions of (dx, dy) of a same-strand pair that passed the cheap tests of the column DP (gap_dp_columns): what
// gap_anchor_score1_pos / gap_anchor_score2_pos / gap_clip_score compute behind their rejections (da = |dx - dy| because the strands are equal).
// fn 1: dx >= 0, 0 <= dy < 245, da < 64;  fn 2: 0 <= dy < 128, da < 64, dx < 4096;  fn 5: 1 <= dy < 31, da < 59.  tests/test_gap_shim_cpu.py
// compares them with the anchor forms over those boxes.
// floor(100 da / m) for da < 64, 50 <= m < 4096: a float estimate with an exact correction instead of the integer division sequence
LNR_HD inline u32 gap_derr_q(u32 da, u32 m) {
#if defined(__HIP_DEVICE_COMPILE__)
    float inv = __builtin_amdgcn_rcpf((float)m);
#else
    float inv = 1.0f / (float)m;
#endif
    u32 nn = 100u * da;
    u32 q = (u32)((float)nn * inv);
    i32 r = (i32)(nn - q * m);                      // the true remainder lies in [-m, 2m)
    return r < 0 ? q - 1 : ((u32)r >= m ? q + 1 : q);
}
LNR_HD inline int gap_score_delta(int fn, u32 dx, u32 dy) {
    u32 da = dx > dy ? dx - dy : dy - dx;
    if (fn == 1) {
        u32 m = dy > 50 ? dy : 50, derr = gap_derr_q(da, m);
        i32 s_derr = derr < 10 ? 0 : (derr < 15 ? (i32)(10 + 2 * derr) : (i32)(derr * derr / 10 + 40));
        i32 s_dy = dy < 100 ? (i32)(dy / 4) : (dy < 200 ? (i32)(dy / 3) - 9 : (i32)dy - 145);
        return 100 - s_dy - s_derr;
    }
    if (fn == 2) {
        u32 m = dx > dy ? dx : dy; if (m < 50) m = 50;
        u32 derr = gap_derr_q(da, m);
        i32 s_derr = derr < 5 ? (i32)(4 * derr) : (derr < 10 ? (i32)(6 * derr) - 10 : (i32)(derr * derr - 5 * derr));
        return 100 - (i32)(dy * (dy + 300) / 300) - s_derr;
    }
    i32 s_da = da < 2 ? (i32)(30 + 5 * da) : (da < 5 ? (i32)(36 + 2 * da) : (i32)(41 + da));
    return 100 - (i32)(dy * (12 * dy + 650) / 450) - s_da;
}
// the cheap tests: is (dx, dy) inside the box outside of which the score cannot be positive, and not one of the rejected near pairs
LNR_HD inline bool gap_score_box(int fn, u32 dx, i32 dy) {
    u32 da = (i32)dx > dy ? dx - (u32)dy : (u32)dy - dx;
    if (fn == 1) return (u32)dy < 245u && da < 64u && !(dx < 8 && (i32)dx != dy);
    if (fn == 2) return (u32)dy < 128u && da < 64u && dx < 4096u && !((dx < 8 || dy < 8) && (i32)dx != dy);
    return (u32)(dy - 1) < 30u && da < 59u && !((dx < 3 || dy < 3) && (i32)dx != dy);
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

// ---- the long rows of the chain DP by several waves (k_gap_team): the main wave (wave 0) runs the read; when a row goes far beyond
// the register window it posts the row in LDS, every wave of the workgroup scans its share of the predecessors (blocks of 256,
// dealt round-robin) and posts its best key, the main wave takes the maximum.  Helper waves do nothing else: they wait at the
// workgroup barrier for the next row or for the exit command.
LNR_HD inline int gap_dp_score(int fn, u64 a, u64 b) { return fn == 2 ? gap_anchor_score2_pos(a, b) : (fn == 5 ? gap_clip_score(a, b) : gap_anchor_score1_pos(a, b)); }
#ifndef K_GAP_TEAM_ROW
#define K_GAP_TEAM_ROW 2048   // predecessors of the previous row from which a row is dealt over the team (two barriers per row)
#endif
struct GStage { u32 x, y, z, w; };
#ifndef K_GAP_SINGLE_MAX
#define K_GAP_SINGLE_MAX 8192    // elements of one sort / anchors of one chain DP from which a single wave hands its read to a team (4096: too many hand-offs, 577 ms; 16384: 332 ms; 8192: 328 ms)
#endif
#ifndef K_GAP_YB_MAX
#define K_GAP_YB_MAX 1024     // y buckets (64 wide, both strands) the column DP keeps cursors for in LDS
#endif
#ifndef K_GAP_YB_MIN
#define K_GAP_YB_MIN 4096     // anchors from which the column DP takes its predecessors from y buckets instead of scanning the x window
#endif
struct GapTeam { const u64 *anchors; const i32 *score; u64 ai, dx_depth; int i, j_str, fn, cmd; u64 part[16]; Rec rec; u32 n, depth; int dup; u32 ncols; u32 *xs, *ys;
                 u64 *s_a; u32 *s_L, *s_R; u64 *s_tasks; const u64 *s_queue; u32 s_nq, s_next; GapCmp s_cmp;     // cmd 3: the ranges of a big sort dealt over the waves
                 const u64 *j_hs; u64 *j_out; u32 j_p1, j_p2, j_k; int j_kind, j_pass; u64 j_rvcp; i64 j_lower, j_upper; GAncBand j_band;   // cmd 4: the pairs of a big k-mer block
                 GStage stage[16][128];
                 unsigned long long *tw; u32 tw_mask;
                 GStage *yb_list; u32 yb_ymin, yb_ymax, yb_start[K_GAP_YB_MAX + 1], yb_head[K_GAP_YB_MAX], yb_tail[K_GAP_YB_MAX]; };   // cmd 2 on many anchors: the rows of the columns done so far, listed per y bucket (gap_dp_columns)   // cmd 2: open-addressed set of the anchor words (the check for identical anchors)   // cmd: 0 exit, 1 one long row, 2 a whole DP by columns; xs / ys: x and y | strand << 24 of the anchors; stage: per wave, the pairs that passed the cheap tests
#ifndef K_GAP_COL_MEAN
#define K_GAP_COL_MEAN 12     // mean anchors per column from which the column form pays (one workgroup barrier per column)
#endif
#ifndef K_GAP_COL_MIN
#define K_GAP_COL_MIN 1024    // anchors from which a chain DP is run column by column on the whole team
#endif
#if defined(__HIPCC__)
// the predecessors [0, i - 65] of row i in blocks of 256 from the top, block b for wave b mod nw; returns this wave's best key
struct GapDpFn { int fn; __device__ int operator()(u64 a, u64 b) const { return gap_dp_score(fn, a, b); } };
template <class Score>
__device__ inline u64 gap_dp_row_share(const u64 *anchors, const i32 *rscore, u64 ai, int i, int j_str, u64 dx_depth, Score score, int wave, int nw, u32 *nblk = nullptr) {
    const int lane = (int)(threadIdx.x & 63);
    u64 xi = ganc_x(ai), key = 0;
    u32 blocks = 0;
    for (int jb = i - 65 - 256 * wave; jb >= 0; jb -= 256 * nw) {
        blocks++;
        u64 av[4]; i32 sv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            int j2 = jb - 64 * u - lane;
            av[u] = j2 >= 0 ? anchors[j2] : 0;
            sv[u] = j2 >= 0 ? rscore[j2] : 0;
        }
        bool stop = false;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            int j2 = jb - 64 * u - lane;
            bool ok = j2 >= 0 && (j2 >= j_str || ganc_x(av[u]) - xi < dx_depth);
            if (ok) {
                int sc = score(av[u], ai);
                if (sc > 0) { u64 k = ((u64)(u32)(sc + sv[u]) << 32) | (u64)(0xffffffffu - (u32)j2); key = k > key ? k : key; }
            }
            stop = stop || !ok;
        }
        if (__any(stop)) break;                                  // (x-descending: once a predecessor is out of range all earlier ones are)
    }
    for (int m = 32; m; m >>= 1) { u64 o = __shfl_xor(key, m); key = o > key ? o : key; }
    if (nblk) *nblk = blocks;
    return key;
}
// ---- a whole chain DP on the team, COLUMN BY COLUMN.  The anchors are x-descending; a column = the run of anchors with one x.  A pair of
// anchors with dx = 0 never scores (all three score functions reject |dx| < 8 (3) unless dx == dy, and dx == dy == 0 would be the same k-mer
// pair twice), so the rows of a column depend on earlier columns only: they are dealt over the waves (row c0 + w, c0 + w + nw, ...), each wave
// scans its row's whole window from memory (the predecessors before the column: the same blocks of 256 as gap_dp_row_share) and writes the
// row's record; one workgroup barrier per column.  Same records as the serial scan: the window of row i is [first j with x_j - x_i < dx_depth
// or i - depth, i - 1], of which [c0, i - 1] cannot score; ties keep the smallest j (the key's low word).  Identical anchor words inside a
// column (never produced by the joins; checked all the same) set *dup and the caller falls back to the single-wave form.
template <class Score>
__device__ inline void gap_dp_columns(const u64 *anchors, u32 n, Rec r, u32 depth, u64 dx_depth, Score score, int fn, int w, int nw, GapTeam *tm) {
    const int lane = (int)(threadIdx.x & 63);
    u32 *XS = tm->xs, *YS = tm->ys;
    GStage *stage = tm->stage[w];
    // x and y | strand << 24 of every anchor, once: the scan below then costs a dozen 32-bit instructions per predecessor, and only the pairs
    // inside the box of the score function (a few per cent) are staged in LDS and scored, 64 at a time
    // Identical anchor words (they would share a column) are looked for ONCE, with an open-addressed set of the words: the check per row
    // against the rows before it in its column is quadratic in the column, and a satellite repeat has columns of thousands of anchors (one
    // reference k-mer against every copy in the read) -- that check, not the scoring, was most of such a DP's time.
    unsigned long long *TW = tm->tw;
    const u32 twm = tm->tw_mask;
    const unsigned long long TW_EMPTY = ~0ULL;                   // (no anchor word has all bits set)
    if (TW) {
        for (u32 i = (u32)w * 64 + (u32)lane; i <= twm; i += (u32)nw * 64) TW[i] = TW_EMPTY;
        __syncthreads();
    }
    u32 myc = 0, my_lo = 0xffffffffu, my_hi = 0;
    bool twin_any = false;
    GStage *YL = tm->yb_list;
    for (u32 i = (u32)w * 64 + (u32)lane; i < n; i += (u32)nw * 64) {
        u64 a = anchors[i];
        u32 x = (u32)ganc_x(a), y = (u32)ganc_y(a);
        XS[i] = x; YS[i] = y | ((u32)ganc_strand(a) << 24);
        myc += (i == 0 || (u32)ganc_x(anchors[i - 1]) != x) ? 1u : 0u;
        my_lo = y < my_lo ? y : my_lo; my_hi = y > my_hi ? y : my_hi;
        if (TW) {
            u32 h = (u32)((a * 0x9E3779B97F4A7C15ULL) >> 32) & twm;
            for (;;) {
                unsigned long long old = atomicCAS(&TW[h], TW_EMPTY, (unsigned long long)a);
                if (old == TW_EMPTY) break;
                if (old == (unsigned long long)a) { twin_any = true; break; }
                h = (h + 1) & twm;
            }
        }
    }
    myc = wave_sum(myc);
    if (lane == 0 && myc) atomicAdd(&tm->ncols, myc);
    if (__any(twin_any) && lane == 0) tm->dup = 1;
    if (YL) {
        my_lo = wave_min_u32(my_lo); my_hi = wave_max_u32(my_hi);
        if (lane == 0) { atomicMin(&tm->yb_ymin, my_lo); atomicMax(&tm->yb_ymax, my_hi); }
    }
    __syncthreads();
    if (tm->dup == 1) return;                                    // (uniform: the caller redoes the DP in the single-wave form)
    // columns of a few anchors each would be one workgroup barrier per few rows: the single-wave form (the last 64 records in registers) is the
    // better one there -- every wave sees the same count and leaves; wave 0 then runs that form (dup = 2: not a duplicate, just "not by columns")
    if ((u64)tm->ncols * K_GAP_COL_MEAN > (u64)n) { if (w == 0 && lane == 0) tm->dup = 2; __syncthreads(); return; }
    const u32 dxd = (u32)dx_depth;
    // ---- Many anchors.  In a satellite repeat one reference k-mer matches hundreds of read positions: the x window of a row (dx < 80) then
    // holds ten thousand predecessors, the scan of it is the DP's time (VALU-bound), and only those with 0 <= dy < H (245 / 128 / 31) can
    // score.  So the rows of the columns done so far are kept in one list per (strand, y >> 6), appended when their column is finished WITH
    // what a later row needs of them -- x, y | strand, final score, index: 16 bytes, read back coalesced.  A list is in column order, i.e.
    // x-descending: what has left the x window leaves at its front (head cursor, advanced by whoever sees it).  A row looks at the <= 5
    // buckets its box covers; the same pairs pass the same tests and reach the same scoring as in the scan below.
    const u32 ymin = tm->yb_ymin;
    const u32 nby = YL ? ((tm->yb_ymax - ymin) >> 6) + 1 : 0;
    const bool by_y = YL != nullptr && 2 * nby <= K_GAP_YB_MAX;
    const u32 boxh = fn == 1 ? 245u : (fn == 2 ? 128u : 31u);
    if (by_y) {
        const u32 nbk = 2 * nby;
        for (u32 b = (u32)w * 64 + (u32)lane; b < nbk; b += (u32)nw * 64) tm->yb_tail[b] = 0;
        __syncthreads();
        for (u32 i = (u32)w * 64 + (u32)lane; i < n; i += (u32)nw * 64) { u32 ys = YS[i]; atomicAdd(&tm->yb_tail[(((ys & 0xffffffu) - ymin) >> 6) + (ys >> 24) * nby], 1u); }
        __syncthreads();
        if (w == 0) {
            u32 run = 0;
            for (u32 b0 = 0; b0 < nbk; b0 += 64) {
                u32 b = b0 + (u32)lane;
                u32 c = b < nbk ? tm->yb_tail[b] : 0u;
                u32 inc = wave_incl_scan(c);
                if (b < nbk) tm->yb_start[b] = run + inc - c;
                run += (u32)__shfl((int)inc, 63);
            }
            if (lane == 0) tm->yb_start[nbk] = run;
        }
        __syncthreads();
        for (u32 b = (u32)w * 64 + (u32)lane; b < nbk; b += (u32)nw * 64) { tm->yb_tail[b] = 0; tm->yb_head[b] = 0; }
        for (u32 i = (u32)w * 64 + (u32)lane; i < n; i += (u32)nw * 64) { GStage e_; e_.x = 0; e_.y = 0; e_.z = 0; e_.w = 0xffffffffu; YL[i] = e_; }   // (a slot reads as "no row" until its store has landed)
        __syncthreads();
    }
#ifdef LNR_GAP_DEVPROF
    unsigned long long tp_[6] = {0, 0, 0, 0, 0, 0}, tq_ = wall_clock64(), tr_; u32 ncol_ = 0; unsigned long long ncand_ = 0;
#define TP_(k) do { tr_ = wall_clock64(); tp_[k] += tr_ - tq_; tq_ = tr_; } while (0)
#else
#define TP_(k) do {} while (0)
#endif
    u32 c0 = 0;
    while (c0 < n) {
        const u32 x0 = XS[c0];
        u32 c1 = c0 + 1;
        for (;;) {
            u32 idx = c1 + (u32)lane;
            bool same = idx < n && XS[idx] == x0;
            u64 m = __ballot(!same);
            if (m) { c1 += (u32)__builtin_ctzll(m); break; }
            c1 += 64;
        }
        TP_(0);
#ifdef LNR_GAP_DEVPROF
        ncol_++;
#endif
        for (u32 i = c0 + (u32)w; i < c1; i += (u32)nw) {
            const u64 ai = anchors[i];
            const u32 ysi = YS[i];
            const int j_str = (int)i - (int)depth < 0 ? 0 : (int)i - (int)depth;
            if (!TW) {                                           // (no room for the set: the check per row)
                bool twin = false;
                for (u32 jb = c0; jb < i; jb += 64) { u32 j = jb + (u32)lane; twin = twin || (j < i && anchors[j] == ai); }
                if (__any(twin)) { if (lane == 0) tm->dup = 1; }
            }
            u64 key = 0;
            {   // the last `depth` predecessors whatever their distance (those of this column cannot score): the anchor form of the score
                int j = (int)c0 - 1 - lane;
                if (j >= j_str && j >= 0) {
                    int sc = score(anchors[j], ai);
                    if (sc > 0) key = ((u64)(u32)(sc + r.score[j]) << 32) | (u64)(0xffffffffu - (u32)j);
                }
            }
            TP_(1);
            u32 nst = 0;
            auto flush = [&](u32 cnt) {                        // the first cnt (<= 64) staged pairs are scored
                WLDS();
                GStage e = stage[lane];
                if ((u32)lane < cnt) {
                    int sc = gap_score_delta(fn, e.x, e.y);
                    if (sc > 0) { u64 k = ((u64)(u32)(sc + (i32)e.z) << 32) | (u64)(0xffffffffu - e.w); key = k > key ? k : key; }
                }
                if (nst > cnt) { GStage t = stage[64 + lane]; WLDS(); if ((u32)lane < nst - cnt) stage[lane] = t; }
                nst -= cnt;
                WLDS();
            };
            if (by_y) {
                const u32 yi = ysi & 0xffffffu, sb = (ysi >> 24) * nby;
                const u32 b_lo = (yi - ymin) >> 6;
                u32 b_hi = (yi + boxh - 1 - ymin) >> 6;
                b_hi = b_hi < nby ? b_hi : nby - 1;
                const u32 nq = b_hi - b_lo + 1;                       // <= 5 buckets
                u32 q_start = 0, q_head = 0, q_len = 0;
                if ((u32)lane < nq) {
                    u32 b = sb + b_lo + (u32)lane;
                    q_start = tm->yb_start[b]; q_head = tm->yb_head[b];
                    q_len = tm->yb_tail[b] - q_head;
                }
                u32 pre_incl = wave_incl_scan(q_len);
                const u32 total = (u32)__shfl((int)pre_incl, 63);
#ifdef LNR_GAP_DEVPROF
                ncand_ += total;
#endif
                u32 pre[6], qb[5], qh[5];                              // first candidate index of every bucket, where its live part starts in the list, its head
                pre[0] = 0;
#pragma unroll
                for (int q = 0; q < 5; q++) { pre[q + 1] = (u32)__shfl((int)pre_incl, q); qb[q] = (u32)__shfl((int)(q_start + q_head), q); qh[q] = (u32)__shfl((int)q_head, q); }
                // bucket by bucket (the bucket is then wave-uniform: no per-lane bookkeeping), 256 entries per step, four loads in flight per lane
                for (u32 t = 0; t < nq; t++) {
                    u32 base_ = 0, len_ = 0, head_ = 0;
#pragma unroll
                    for (int k = 0; k < 5; k++) if ((u32)k == t) { base_ = qb[k]; len_ = pre[k + 1] - pre[k]; head_ = qh[k]; }
                    u32 gone = 0; bool lead = true;                    // leading entries found outside the x window
                    for (u32 o = 0; o < len_; o += 256) {
                        GStage cv[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            u32 g = o + 64u * (u32)u + (u32)lane;
                            GStage c_; c_.x = 0; c_.y = 0; c_.z = 0; c_.w = 0xffffffffu;
                            if (g < len_) c_ = YL[base_ + g];
                            cv[u] = c_;
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            if (o + 64u * (u32)u >= len_) break;
                            const GStage c_ = cv[u];
                            bool in = c_.w < c0;                       // (0xffffffff: past the end, or a slot a wave of THIS column is filling: not a predecessor)
                            u32 dx = c_.x - x0;
                            bool live = in && dx < dxd;
                            if (lead) {                                // the front of the list that has left the window
                                u64 mq = __ballot(in), ml = __ballot(live);
                                if (ml) { gone += (u32)__popcll(mq & ((1ULL << __builtin_ctzll(ml)) - 1)); lead = false; }
                                else gone += (u32)__popcll(mq);
                            }
                            i32 dy = (i32)(c_.y - ysi);
                            bool pass = live && gap_score_box(fn, dx, dy);
                            // (entries that came through a y bucket pass the box often: scored in place, not compacted through LDS)
                            if (__any(pass)) {
                                int sc = pass ? gap_score_delta(fn, dx, (u32)dy) : 0;
                                if (sc > 0) { u64 k = ((u64)(u32)(sc + (i32)c_.z) << 32) | (u64)(0xffffffffu - c_.w); key = k > key ? k : key; }
                            }
                        }
                    }
                    if (gone && lane == 0) atomicMax(&tm->yb_head[sb + b_lo + t], head_ + gone);
                }
            } else
            for (int jb = (int)c0 - 1; jb >= 0; jb -= 256) {
                u32 xv[4], yv[4]; i32 sv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    int j2 = jb - 64 * u - lane;
                    xv[u] = j2 >= 0 ? XS[j2] : 0u; yv[u] = j2 >= 0 ? YS[j2] : 0u; sv[u] = j2 >= 0 ? r.score[j2] : 0;
                }
                bool stop = false;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    int j2 = jb - 64 * u - lane;
                    u32 dx = xv[u] - x0;
                    bool ok = j2 >= 0 && dx < dxd;
                    i32 dy = (i32)(yv[u] - ysi);                 // (another strand: bit 24 differs, far outside every box)
                    bool pass = ok && gap_score_box(fn, dx, dy);
                    u64 m = __ballot(pass);
                    if (m) {
                        if (pass) { GStage e_; e_.x = dx; e_.y = (u32)dy; e_.z = (u32)sv[u]; e_.w = (u32)j2; stage[nst + (u32)__popcll(m & ((1ULL << lane) - 1))] = e_; }
                        nst += (u32)__popcll(m);
                        if (nst >= 64) flush(64);
                    }
                    stop = stop || !ok;
                }
                if (__any(stop)) break;
            }
            if (nst) flush(nst);
            TP_(2);
            key = (u64)wave_max_i64((i64)key);                 // (keys are below 2^63: a DPP reduction instead of six bpermute round trips)
            if (lane == 0) {
                if (key) {
                    int best = (int)(key >> 32), max_j = (int)(0xffffffffu - (u32)key);
                    r.p2[i] = max_j; r.score[i] = best; r.len[i] = r.len[max_j] + 1; r.score2[i] = best; r.root[i] = r.root[max_j]; r.leaf[i] = 1; r.leaf[max_j] = 0;
                } else { r.p2[i] = -1; r.score[i] = 0; r.len[i] = 1; r.score2[i] = 0; r.root[i] = (i32)i; r.leaf[i] = 1; }
                if (by_y) {                                      // the row joins its bucket's list (rows of one column in any order: they share one x)
                    u32 b = (((ysi & 0xffffffu) - ymin) >> 6) + (ysi >> 24) * nby;
                    GStage e_; e_.x = x0; e_.y = ysi; e_.z = key ? (u32)(key >> 32) : 0u; e_.w = i;
                    YL[tm->yb_start[b] + atomicAdd(&tm->yb_tail[b], 1u)] = e_;
                }
            }
            TP_(3);
        }
        __syncthreads();                                         // the column's records are written: the next column reads them
        TP_(4);
        c0 = c1;
    }
#ifdef LNR_GAP_DEVPROF
    if (n > 50000 && w == 0 && lane == 0)
        printf("[gap prof] column DP of %u anchors, fn %d, by_y %d: %u columns; wave 0 (x10 ns): column start %llu, depth clause %llu, window %llu (candidates %llu), reduce + record %llu, barrier %llu\n",
               n, fn, (int)by_y, ncol_, tp_[0], tp_[1], tp_[2], ncand_, tp_[3], tp_[4]);
#endif
}
// ---- the (reference k-mer, read k-mer) pairs of one big block on the team (cmd 4): wave w takes the w-th contiguous share of the pair indices
// (i major, as the serial loops), pass 1 counts the pairs it keeps, pass 2 writes them behind the shares before it -- the output is in pair order
#ifndef K_GAP_JOIN_TEAM_MIN
#define K_GAP_JOIN_TEAM_MIN 65536
#endif
__device__ inline void gap_join_team_share(GapTeam *tm, int w, int nw, int pass) {
    const u32 lane = threadIdx.x & 63;
    const u64 ni = tm->j_p2 - tm->j_p1, nj = tm->j_k - tm->j_p2, np = ni * nj;
    const u64 lo = np * (u64)w / (u64)nw, hi = np * (u64)(w + 1) / (u64)nw;
    const u64 *hs = tm->j_hs;
    u64 base = 0;
    if (pass == 2) for (int v = 0; v < w; v++) base += tm->part[v];
    u64 run = 0;
    for (u64 pb = lo; pb < hi; pb += 64) {
        u64 q = pb + lane;
        bool in = q < hi;
        u64 qi = q / nj, qj = q - qi * nj;
        u64 hi_ = in ? hs[tm->j_p1 + qi] : 0, hj_ = in ? hs[tm->j_p2 + qj] : 0, a;
        bool kp;
        if (tm->j_kind == 0) { a = ganc_make(hi_, hj_, tm->j_rvcp); kp = in && tm->j_band.keep(a); }
        else { i64 d = (i64)(hi_ & ((1ULL << 30) - 1)) - (i64)(hj_ & ((1ULL << 30) - 1)); kp = in && tm->j_lower <= d && d < tm->j_upper; a = canc_make(hi_, hj_); }
        u64 m = __ballot(kp);
        if (pass == 2 && kp) tm->j_out[base + run + (u64)__popcll(m & ((1ULL << lane) - 1))] = a;
        run += (u64)__popcll(m);
    }
    if (pass == 1 && lane == 0) tm->part[w] = run;
}
// wave 0's side of cmd 4; returns false when the team is not used (the caller runs the single-wave loop)
__device__ inline bool gap_join_team(GapCtx &X, const GVec<u64> &g_hs, GVec<u64> &out, int p1, int p2, int k, int kind, u64 rvcp, i64 lower, i64 upper, const GAncBand *B) {
    if (X.team <= 1 || (u64)(p2 - p1) * (u64)(k - p2) < K_GAP_JOIN_TEAM_MIN) return false;
    GapTeam *tm = X.tm;
    if ((threadIdx.x & 63) == 0) {
        tm->j_hs = g_hs.p; tm->j_p1 = (u32)p1; tm->j_p2 = (u32)p2; tm->j_k = (u32)k; tm->j_kind = kind; tm->j_rvcp = rvcp; tm->j_lower = lower; tm->j_upper = upper;
        if (B) tm->j_band = *B;
        tm->j_pass = 1; tm->cmd = 4;
    }
    __syncthreads();                                             // (A)
    gap_join_team_share(tm, 0, X.team, 1);
    __syncthreads();                                             // (B1) the counts are posted
    u64 total = 0;
    for (int v = 0; v < X.team; v++) total += tm->part[v];
    out.reserve(out.n + (u32)total);
    bool fits = (u64)out.n + total <= out.cap;
    if ((threadIdx.x & 63) == 0) { tm->j_out = out.p + out.n; tm->j_pass = fits ? 2 : 0; }
    __syncthreads();                                             // (B2)
    if (fits) gap_join_team_share(tm, 0, X.team, 2);
    __syncthreads();                                             // (B3)
    if (fits) out.n += (u32)total;
    return true;
}
__device__ void gap_sort_team_share(GapTeam *tm);             // (lnr_gap_kernels.hip: the queued ranges of a big sort, one at a time per wave)
__device__ inline void gap_team_helper_loop(GapTeam *tm, int wave, int nw) {
    for (;;) {
        __syncthreads();                                         // (A) a command is posted
        if (tm->cmd == 0) break;
        if (tm->cmd == 3) { gap_sort_team_share(tm); __syncthreads(); continue; }
        if (tm->cmd == 4) {
            gap_join_team_share(tm, wave, nw, 1);
            __syncthreads();                                     // (B1)
            __syncthreads();                                     // (B2) wave 0 has reserved the output
            if (tm->j_pass == 2) gap_join_team_share(tm, wave, nw, 2);
            __syncthreads();                                     // (B3)
            continue;
        }
        if (tm->cmd == 2) { gap_dp_columns(tm->anchors, tm->n, tm->rec, tm->depth, tm->dx_depth, GapDpFn{tm->fn}, tm->fn, wave, nw, tm); continue; }
        u64 key = gap_dp_row_share(tm->anchors, tm->score, tm->ai, tm->i, tm->j_str, tm->dx_depth, GapDpFn{tm->fn}, wave, nw);
        if ((threadIdx.x & 63) == 0) tm->part[wave] = key;
        __syncthreads();                                         // (B) the shares are posted
    }
}
#endif

// ---- traceBackChains (cluster_util.cpp:306-335) in the wave-per-read form: the scans over the records (distinct roots, the search for the
// best score of traceBackChains0 and the best score before it, the table of traceBackChains1) run on all 64 lanes; the walks along a chain and
// the emission stay the uniform code of lnr_hd.h (every lane, same data).  Same results as traceback(): the scan of traceBackChains0 returns
// the FIRST index of the maximal score > -1 and, as its second best, the maximum of the scores before that index.
#if defined(__HIPCC__)
__device__ inline i64 gtb_range_best(const i32 *score, u32 lo, u32 hi) {
    const u32 lane = threadIdx.x & 63;
    i64 best = -1;
    for (u32 j0 = lo; j0 < hi; j0 += 256) {
        i32 sc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { u32 j = j0 + 64 * u + lane; sc[u] = j < hi ? score[j] : -1; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            u32 j = j0 + 64 * u + lane;
            if (sc[u] > -1) { i64 k = ((i64)sc[u] << 32) | (i64)(u32)(0x7fffffff - (int)j); best = k > best ? k : best; }
        }
    }
    return wave_max_i64(best);
}
__device__ inline int gtb_prefix_max(const i32 *score, u32 end) {
    const u32 lane = threadIdx.x & 63;
    i64 m = -1;
    for (u32 j0 = 0; j0 < end; j0 += 256) {
        i32 sc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { u32 j = j0 + 64 * u + lane; sc[u] = j < end ? score[j] : -1; }
#pragma unroll
        for (int u = 0; u < 4; u++) { i64 v = sc[u]; m = v > m ? v : m; }
    }
    return (int)wave_max_i64(m);
}
// traceback1_table (lnr_hd.h) with the lanes over the records: trees are numbered by their first leaf (per 64 records the roots not yet listed
// are appended lowest lane first), the best leaf of a tree is the maximum of (score, earliest j) -- an atomic max on a 64-bit key in ls.ranks
__device__ inline int gtb_table_wave(const Rec &r, u32 n, LeaderScratch &ls) {
    const u32 lane = threadIdx.x & 63;
    int nl = 0;
    unsigned long long *key = (unsigned long long *)ls.ranks;
    key[lane] = 0;
    WSYNC();
    for (u32 base = 0; base < n; base += 64) {
        u32 j = base + lane;
        bool lf = j < n && r.leaf[j] != 0;
        i32 root = lf ? r.root[j] : -1;
        int k = -1;
        if (lf) for (int q = 0; q < nl; q++) if (ls.l_root[q] == root) { k = q; break; }
        u64 pend = __ballot(lf && k < 0);
        while (pend) {
            int src = (int)__builtin_ctzll(pend);
            i32 rt = __shfl(root, src);
            if (nl < 64) { if (lane == 0) ls.l_root[nl] = rt; }
            if (lf && k < 0 && root == rt) k = nl < 64 ? nl : -2;      // -2: table full, the leaf is ignored like the serial form does
            if (nl < 64) nl++;
            WSYNC();
            pend = __ballot(lf && k == -1);
        }
        if (lf && k >= 0) {
            unsigned long long kv = ((unsigned long long)(u32)(r.score[j] + 0x40000000) << 32) | (unsigned long long)(0xffffffffu - j);
            atomicMax(&key[k], kv);
        }
    }
    WSYNC();
    if ((int)lane < nl) {
        unsigned long long kv = key[lane];
        u32 j = 0xffffffffu - (u32)(kv & 0xffffffffu);
        ls.l_score[lane] = (i32)(u32)(kv >> 32) - 0x40000000; ls.l_len[lane] = r.len[j]; ls.l_leaf[lane] = (i32)j;
    }
    WSYNC();
    return nl;
}
template <class Sink>
__device__ inline void gap_traceback_wave(Rec r, u32 n, Sink &sink, i32 *chain, i32 *chain_sc, i32 *cnt, int min_len, int abort_score, int bestn, float stop_ratio, LeaderScratch &ls) {
    const u32 lane = threadIdx.x & 63;
    for (u32 i = lane; i < n; i += 64) cnt[i] = 0;
    WSYNC();
    for (u32 i = lane; i < n; i += 64) cnt[r.root[i]] = 1;
    WSYNC();
    u32 c = 0;
    for (u32 i = lane; i < n; i += 64) c += (u32)cnt[i];
    u32 root_num = wave_sum(c);
    if (root_num > 50) {
        int search_times = bestn < 50 ? bestn : 50;
        for (int it = 0; it < search_times; it++) {
            i64 best = gtb_range_best(r.score, 0, n);
            Tb0Scan sc; sc.max_score = -1; sc.max_2nd = -1; sc.max_str = -1; sc.max_len = 0;
            if (best >= 0) {
                sc.max_score = (int)(best >> 32);
                sc.max_str = 0x7fffffff - (int)(u32)(best & 0xffffffff);
                sc.max_len = r.len[sc.max_str];
                sc.max_2nd = gtb_prefix_max(r.score, (u32)sc.max_str);
            }
            bool more = tb0_step(r, sc, sink, chain, chain_sc, min_len, abort_score, stop_ratio);
            WSYNC();
            if (!more) break;
        }
    } else {
        int nl = gtb_table_wave(r, n, ls);
        traceback1_emit(r, nl, sink, chain, chain_sc, min_len, abort_score, bestn, stop_ratio, ls);
    }
}
#endif
template <class Sink>
LNR_HD inline void gap_traceback(Rec r, u32 n, Sink &sink, i32 *chain, i32 *chain_sc, i32 *cnt, int min_len, int abort_score, int bestn, float stop_ratio, GapCtx &X) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (X.coop && n > 256) { gap_traceback_wave(r, n, sink, chain, chain_sc, cnt, min_len, abort_score, bestn, stop_ratio, *X.ls); return; }
#endif
    traceback(r, n, sink, chain, chain_sc, cnt, min_len, abort_score, bestn, stop_ratio, *X.ls);
}

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
LNR_HD inline void gap_chain_anchors(const u64 *anchors, u32 n, GVec<u64> &out, bool to_tiles, u32 depth, u64 dx_depth, int bestn, int min_len, int abort_score, Score score, GapCtx &X, int fn_id = 0) {
    if (n < 2 || X.ar->ovf) return;
#if defined(__HIP_DEVICE_COMPILE__)
    if (X.hand && n >= K_GAP_SINGLE_MAX) { X.ar->ovf = 2; return; }   // (a chain DP this long: the read goes to a team, see gap_sort_wave)
#endif
    u64 m0 = X.ar->mark();
    Rec r;
    i32 *blk = (i32 *)X.ar->get((u64)n * 9 * sizeof(i32));
    r.score = blk; r.score2 = blk + n; r.len = blk + 2 * n; r.p2 = blk + 3 * n; r.root = blk + 4 * n; r.leaf = blk + 5 * n;
    i32 *chain = blk + 6 * n, *chain_sc = blk + 7 * n, *cnt = blk + 8 * n;
    if (X.ar->ovf) return;
    // (the reference sizes the records without clearing them and sets only record 0's score, length and predecessor: root and leaf of
    // record 0 are whatever the allocator left -- 0 in practice, as here)
#if defined(__HIP_DEVICE_COMPILE__)
    if (X.coop) { for (u32 i = threadIdx.x & 63; i < n; i += 64) { r.score[i] = 0; r.score2[i] = 0; r.len[i] = 0; r.p2[i] = 0; r.root[i] = 0; r.leaf[i] = 0; } WSYNC(); }
    else
#endif
    for (u32 i = 0; i < n; i++) { r.score[i] = 0; r.score2[i] = 0; r.len[i] = 0; r.p2[i] = 0; r.root[i] = 0; r.leaf[i] = 0; }
    r.score[0] = 0; r.len[0] = 1; r.p2[0] = -1;
    {
    GP(X, 4);
#if defined(LNR_GAP_DEVPROF) && defined(__HIP_DEVICE_COMPILE__)
    struct DpT { GapCtx &X; u32 n; int fn; bool *bc; unsigned long long t0; __device__ ~DpT() { unsigned long long t = wall_clock64() - t0; if (t > X.dp_t) { X.dp_t = t; X.dp_n = n; X.dp_mode = *bc; X.dp_fn = (unsigned long long)fn; } } };
    bool by_columns_ = false;
    DpT dpt_{X, n, fn_id, &by_columns_, (unsigned long long)wall_clock64()};
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    bool by_columns = false;
    u32 *xs_ = nullptr;
    if (X.coop && X.team > 1 && fn_id && n >= K_GAP_COL_MIN) { xs_ = (u32 *)X.ar->get((u64)n * 8); if (X.ar->ovf) return; }
    unsigned long long *tw_ = nullptr; u32 tw_cap = 64;
    if (xs_) {
        while (tw_cap < 2 * n) tw_cap <<= 1;
        if (X.ar->off + (u64)tw_cap * 8 + 64 <= X.ar->cap) tw_ = (unsigned long long *)X.ar->get((u64)tw_cap * 8);   // (only when it fits)
    }
    GStage *yl_ = nullptr;
    if (xs_ && n >= K_GAP_YB_MIN && X.ar->off + (u64)n * 16 + 64 <= X.ar->cap) yl_ = (GStage *)X.ar->get((u64)n * 16);   // (only when it fits: the scan form needs no list)
    if (xs_) {
        GapTeam *tm = X.tm;
        if ((threadIdx.x & 63) == 0) { tm->yb_list = yl_; tm->yb_ymin = 0xffffffffu; tm->yb_ymax = 0; tm->tw = tw_; tm->tw_mask = tw_cap - 1; tm->anchors = anchors; tm->rec = r; tm->n = n; tm->depth = depth; tm->dx_depth = dx_depth; tm->fn = fn_id; tm->dup = 0; tm->ncols = 0; tm->xs = xs_; tm->ys = xs_ + n; tm->cmd = 2; }
        __syncthreads();                                         // (A)
        gap_dp_columns(anchors, n, r, depth, dx_depth, GapDpFn{fn_id}, fn_id, 0, X.team, tm);
        by_columns = tm->dup == 0;                               // (a column held the same anchor twice: the single-wave form below redoes the DP)
        if (!by_columns) { for (u32 i = threadIdx.x & 63; i < n; i += 64) { r.score[i] = 0; r.score2[i] = 0; r.len[i] = 0; r.p2[i] = 0; r.root[i] = 0; r.leaf[i] = 0; } WSYNC(); r.score[0] = 0; r.len[0] = 1; r.p2[0] = -1; }
    }
#if defined(LNR_GAP_DEVPROF) && defined(__HIP_DEVICE_COMPILE__)
    by_columns_ = by_columns;
#endif
    if (by_columns) {}
    else if (X.coop) {
        // every lane runs the read's code with the same data; here the predecessors of anchor i are dealt over the lanes, 64 at a
        // time.  The scan ends at the first j below j_str whose x is dx_depth away: the anchors are x-descending, so that is a
        // threshold in j.  Among equal sums the serial scan (j descending, >=) keeps the smallest j: the key's low word.
        // The last 64 records stay in registers (lane l: record i - 1 - l; anchor, score, length, root), shifted by one lane per
        // anchor: a load of what the previous iteration stored would wait for that store to reach L2 (microseconds per anchor).
        // Memory is read only for the predecessors beyond the window and for the anchors themselves, 64 per load.
        const int lane = (int)(threadIdx.x & 63);
        u64 wa = 0, blk = 0;
        i32 ws = 0, wl = 0, wr = 0;
        u32 far_prev = 0;
        for (int i = 0; i < (int)n; i++) {
            if ((i & 1023) == 0 && gap_late(X)) return;
            if ((i & 63) == 0) blk = (u32)(i + lane) < n ? anchors[i + lane] : 0;
            int j_str = i - (int)depth < 0 ? 0 : i - (int)depth;
            u64 ai = __shfl(blk, i & 63), xi = ganc_x(ai), key = 0;
            {
                int j = i - 1 - lane;
                bool ok = j >= 0 && (j >= j_str || ganc_x(wa) - xi < dx_depth);
                if (ok) {
                    int sc = score(wa, ai);
                    if (sc > 0) key = ((u64)(u32)(sc + ws) << 32) | (u64)(0xffffffffu - (u32)j);
                }
                if (!__any(!ok) && i >= 65) {
                    // beyond the window.  The row before tells how long this one is (the x window moves slowly): long rows go to the team
                    u64 far;
                    u32 nb = 0;
                    if (X.team > 1 && fn_id && far_prev >= K_GAP_TEAM_ROW) {
                        GapTeam *tm = X.tm;
                        if (lane == 0) { tm->anchors = anchors; tm->score = r.score; tm->ai = ai; tm->dx_depth = dx_depth; tm->i = i; tm->j_str = j_str; tm->fn = fn_id; tm->cmd = 1; }
                        __syncthreads();                                 // (A)
                        u64 mine = gap_dp_row_share(anchors, r.score, ai, i, j_str, dx_depth, GapDpFn{fn_id}, 0, X.team, &nb);
                        nb *= (u32)X.team;
                        if (lane == 0) tm->part[0] = mine;
                        __syncthreads();                                 // (B)
                        far = 0;
                        for (int w = 0; w < X.team; w++) { u64 o = tm->part[w]; far = o > far ? o : far; }
                    } else
                        far = gap_dp_row_share(anchors, r.score, ai, i, j_str, dx_depth, score, 0, 1, &nb);
                    far_prev = nb * 256;                                 // the row's length in predecessors (rounded up to blocks)
                    key = far > key ? far : key;
                } else far_prev = 0;
            }
            for (int m = 32; m; m >>= 1) { u64 o = __shfl_xor(key, m); key = o > key ? o : key; }
            int best = key ? (int)(key >> 32) : -1, max_j = key ? (int)(0xffffffffu - (u32)key) : i;
            i32 s_i, len_i, root_i;
            if (best > 0) {
                int d = i - 1 - max_j;
                i32 lm, rm;
                if (d < 64) { lm = __shfl(wl, d); rm = __shfl(wr, d); } else { lm = r.len[max_j]; rm = r.root[max_j]; }
                s_i = best; len_i = lm + 1; root_i = rm;
                r.p2[i] = max_j; r.score[i] = best; r.len[i] = len_i; r.score2[i] = best; r.root[i] = root_i; r.leaf[i] = 1; r.leaf[max_j] = 0;
            } else {
                s_i = 0; len_i = 1; root_i = i;
                r.p2[i] = -1; r.score[i] = 0; r.len[i] = 1; r.score2[i] = 0; r.root[i] = i; r.leaf[i] = 1;
            }
            u64 na = __shfl_up(wa, 1);
            i32 ns = __shfl_up(ws, 1), nl = __shfl_up(wl, 1), nr = __shfl_up(wr, 1);
            wa = lane ? na : ai; ws = lane ? ns : s_i; wl = lane ? nl : len_i; wr = lane ? nr : root_i;
        }
    } else
#endif
    for (int i = 0; i < (int)n; i++) {
        int j_str = i - (int)depth < 0 ? 0 : i - (int)depth, max_j = i, best = -1, j = i - 1;
        for (; j >= 0 && (j >= j_str || ganc_x(anchors[j]) - ganc_x(anchors[i]) < dx_depth); j--) {
            int sc = score(anchors[j], anchors[i]);
            if (sc > 0 && sc + r.score[j] >= best) { max_j = j; best = sc + r.score[j]; }
        }
        if (best > 0) { r.p2[i] = max_j; r.score[i] = best; r.len[i] = r.len[max_j] + 1; r.score2[i] = best; r.root[i] = r.root[max_j]; r.leaf[i] = 1; r.leaf[max_j] = 0; }
        else { r.p2[i] = -1; r.score[i] = 0; r.len[i] = 1; r.score2[i] = 0; r.root[i] = i; r.leaf[i] = 1; }
        X.work += (u64)(i - 1 - j);
        if (X.work > X.work_cap) { X.ar->ovf = 2; return; }
    }
    }
    GP(X, 5);
    // the output may grow while the records live above it in the arena: collect into a vector allocated before them
    TileSink sink; sink.anchors = anchors; sink.tiles = &out; sink.first_len = 0; sink.nchains = 0; sink.to_tiles = to_tiles;
    out.reserve(out.n + n);
    gap_traceback(r, n, sink, chain, chain_sc, cnt, min_len, abort_score, bestn, 0.7f, X);
    (void)fn_id;
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
    if (X.ar->ovf) return;
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
    GP(X, 6);
    GVec<UP> sep; sep.init(X.ar);
    gap_gather_tile_blocks(tiles.p, tiles.n, sep, L, gap_size);
    if (X.ar->ovf) return;
    gap_chain_blocks_cords(tiles, sep, L, 64, X.gp.thd_cts_major_limit, X);
}
LNR_HD inline void g_chains_from_anchors(GVec<u64> &anchors, GVec<u64> &tiles, u64 L, GapCtx &X) {   // g_CreateChainsFromAnchors_ gap_util.cpp:1191-1222
    if (X.ar->ovf) return;
    { GP(X, 3); gap_sort(anchors.p, (long)anchors.n, GapCmp{2, 0}, X); }
    int fn = X.gp.chn1_fn;
    gap_chain_anchors(anchors.p, anchors.n, tiles, true, 20, 80, 20, X.gp.chn1_min_len, X.gp.chn1_abort, [fn](u64 a, u64 b) { return fn == 2 ? gap_anchor_score2_pos(a, b) : gap_anchor_score1_pos(a, b); }, X, fn == 2 ? 2 : 1);
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

// ---- tiles along a chain, scored by the window features (gap_util.cpp:805-905, 1275-1470)
LNR_HD inline u32 tile_fscore(u64 tile, const GapCtx &X) {                                   // _get_tile_f_
    u64 n1 = tile_strand(tile), n2 = cord_id(tile);
    if (n2 < X.gf.nseq) return wdist_checked(X.f1[n1], f2_view(X.gf, n2), cord_y(tile) >> 4, cord_x(tile) >> 4);
    return ~0u;
}
LNR_HD inline u32 tile_fscore_tri(u64 &t, const GapCtx &X, u64 lower_x, u64 lower_y, u64 upper_x, u64 upper_y) {   // _get_tile_f_tri_
    u64 x = cord_x(t), y = cord_y(t);
    int ts4 = X.gp.thd_tile_size / 4;
    int shift = gmin3(ts4, int(x - lower_x), int(y - lower_y));
    u32 f1 = tile_fscore(t, X), mn = f1;
    u64 tl = shift_cord(t, -shift, -shift);
    u32 f2 = tile_fscore(tl, X);
    if (f2 < f1) { t = tl; mn = f2; }
    shift = gmin3(ts4, int(upper_x - x - 1), int(upper_y - y - 1));
    u64 tr = shift_cord(t, shift, shift);
    u32 f3 = tile_fscore(tr, X);
    if (f3 < mn) { t = tr; mn = f3; }
    return mn;
}
LNR_HD inline void tiles_from_chain(const GVec<u64> &ch, GVec<u64> &tiles, u64 gap_str, u64 gap_end, int it_str, int it_end, GapCtx &X) {   // g_CreateTilesFromChains_ :1275-1359 (chain = tiles)
    if (it_end - it_str == 0) return;
    GP(X, 8);
    u64 pre_chain = ch[(u32)it_str], pre_tile = 0;
    i64 tmp_shift = X.gp.thd_tile_size / 2;
    u64 step = (u64)(X.gp.thd_tile_size / 3);
    int kcount = 0, scan_str = it_str, scan_end = it_str;
    for (int i = it_str; i <= it_end; i++) {
        if (i == it_end || tile_strand(ch[(u32)i] ^ pre_chain) || cord_x(ch[(u32)i]) > cord_x(pre_chain) + step || cord_y(ch[(u32)i]) > cord_y(pre_chain) + step) {
            if (i == it_end) scan_end = it_end;
            for (int j = scan_end - 1; j >= scan_str; j--) {
                u64 c = ch[(u32)j];
                u64 nt = create_cord(cord_id(gap_str), cord_x(c) - (u64)tmp_shift, cord_y(c) - (u64)tmp_shift, tile_strand(c));
                u64 lower = tiles.empty() ? gap_str : tiles.back();
                u32 score = tile_fscore_tri(nt, X, cord_x(lower), cord_y(lower), cord_x(gap_end), cord_y(gap_end));
                if (kcount >= (int)X.gp.thd_ctfcs_pattern_in_window && score <= 32 && cord_y(nt) > cord_y(pre_tile)) {
                    if (tiles.empty() || is_tile_end(tiles.back())) set_tile_start(nt);
                    tiles.push(nt);
                    pre_tile = nt; kcount = i - j; pre_chain = c;
                    break;
                }
            }
            scan_str = i; scan_end = i + 1;
        } else { scan_end++; kcount++; }
    }
    if (!tiles.empty()) set_tile_end(tiles.back());
}
LNR_HD inline void tiles_from_chain2(const GVec<u64> &ch, GVec<u64> &tiles_str, GVec<u64> &tiles_end, u64 gap_str, u64 gap_end, int it_str, int it_end, GapCtx &X) {   // :1364-1470
    GVec<u64> ts, te; ts.init(X.ar); te.init(X.ar);
    tiles_from_chain(ch, ts, gap_str, gap_end, it_str, it_end, X);
    if (ts.empty()) return;
    i64 tile_size = X.gp.thd_tile_size;
    u64 c0 = ch[(u32)it_str], c1 = ch[(u32)it_end - 1];
    for (u32 i = 0; i < ts.n; i++) {
        i64 dx1 = (i64)cord_x(c0) - (i64)cord_x(ts[i]), dy1 = (i64)cord_y(c0) - (i64)cord_y(ts[i]);
        if (dx1 <= 0 && dy1 <= 0) {
            if (dx1 == 0 && dy1 == 0) break;
            u64 head = create_cord(cord_id(gap_str), cord_x(c0), cord_y(c0), tile_strand(c0));
            remove_tile_sgn(head);
            if (i == 0) ts.insert(0, &head, 1);
            else { ts[i - 1] = head; ts.erase(0, i - 1); }
            break;
        }
        if (i == ts.n - 1) { ts.clear(); ts.push(create_cord(cord_id(gap_str), cord_x(c0), cord_y(c0), tile_strand(c0))); }
    }
    te.resize(ts.n);
    for (u32 i = 0; i < ts.n; i++) te[i] = shift_cord(ts[i], tile_size, tile_size);
    for (int i = (int)te.n - 1; i >= 0; i--) {
        i64 dx1 = (i64)cord_x(c1) - (i64)cord_x(te[(u32)i]), dy1 = (i64)cord_y(c1) - (i64)cord_y(te[(u32)i]);
        if (dx1 >= 0 && dy1 >= 0) {
            if (dx1 == 0 && dy1 == 0) break;
            ts.n = (u32)i + 1; te.n = (u32)i + 1;
            u64 tail_end = create_cord(cord_id(gap_str), cord_x(c1), cord_y(c1), tile_strand(c1));
            u64 tail_str = shift_cord(tail_end, -tile_size, -tile_size);
            if (is_tile_end(ts[(u32)i])) { remove_tile_sgn(ts[(u32)i]); remove_tile_sgn(te[(u32)i]); set_tile_end(tail_str); set_tile_end(tail_end); }
            ts.push(tail_str); te.push(tail_end);
            break;
        }
        if (i == 0) { ts.n = 1; te.n = 1; te[0] = shift_cord(te[0], dx1, dy1); }
    }
    tiles_str.append(ts);
    tiles_end.append(te);
}
// window walks that also return the window's distance (pmpfinder.cpp:838-880, 995-1045) and extendPatch (:2881-2963)
LNR_HD inline u64 gap_next_window(FeatView f1, FeatView f2, u64 cord, float &score) {
    u64 gid = cord_id(cord), strand = cord_strand(cord), x_pre = cord_x(cord) >> 4, y_pre = cord_y(cord) >> 4, x_min = 0;
    if (y_pre + 12 > f1.n || x_pre + 12 > f2.n) return 0;
    u64 y = y_pre + 5;
    u32 mn = window_best3<false>(f1, f2, y, x_pre + 3, x_min);
    if (mn > 36) return 0;
    score += (float)mn;
    if (x_min - x_pre > 5) return mk_cord((gid << 30) + ((x_pre + 5) << 4), (x_pre + 5 - x_min + y) << 4, strand);
    return mk_cord((gid << 30) + (x_min << 4), y << 4, strand);
}
LNR_HD inline u64 gap_previous_window(FeatView f1, FeatView f2, u64 cord, float &score) {
    u64 gid = cord_id(cord), strand = cord_strand(cord), x_suf = cord_x(cord) >> 4, y_suf = cord_y(cord) >> 4, x_min = 0;
    if (y_suf < 5 || x_suf < 6) return 0;
    u64 y = y_suf - 5;
    u32 mn = window_best3<false>(f1, f2, y, x_suf - 6, x_min);
    if (mn > 36) return 0;
    score += (float)mn;
    if (x_suf - x_min > 5) return mk_cord((gid << 30) + ((x_suf - 5) << 4), (x_suf - x_min - 5 + y) << 4, strand);
    return mk_cord((gid << 30) + (x_min << 4), y << 4, strand);
}
LNR_HD inline int gap_extend_patch(GVec<u64> &cords, int kk, u64 cord1, u64 cord2, int overlap_size, int gap_size, u32 thd_accept_score, GapCtx &X) {
    float score = 0;
    {
        i64 x1 = (i64)cord_x40(cord1), y1 = (i64)cord_y(cord1), x2 = (i64)cord_x40(cord2), y2 = (i64)cord_y(cord2);
        if (gabs(x1 - x2) < overlap_size && gabs(y1 - y2) < overlap_size && !(cord_strand(cord1) ^ cord_strand(cord2))) return 0;
    }
    u64 strand1 = cord_strand(cord1), strand2 = cord_strand(cord2), gid1 = cord_id(cord1), gid2 = cord_id(cord2);
    int len = 0;
    u64 cord = cord1;
    GVec<u64> tmp; tmp.init(X.ar);
    u64 x_bound = cord_x(cord2), y_bound = cord_y(cord2);
    while ((i64)cord_x40(cord) + gap_size <= (i64)cord_x40(cord2)) {
        cord = gap_next_window(X.f1[strand1], f2_view(X.gf, gid1), cord, score);
        if (cord && cord_y(cord) < y_bound && cord_x(cord) < x_bound && score < (float)thd_accept_score) tmp.push(cord);
        else break;
    }
    u64 nw = cord1;
    if (!tmp.empty()) {
        len += (int)tmp.n; nw = tmp.back();
        cords.insert((u32)kk, tmp.p, tmp.n);
        x_bound = cord_x(tmp.back()); y_bound = cord_y(tmp.back());
        tmp.clear();
    } else { x_bound = cord_x(cord1); y_bound = cord_y(cord1); }
    cord = cord2;
    while ((i64)cord_x40(nw) + gap_size <= (i64)cord_x40(cord)) {
        cord = gap_previous_window(X.f1[strand2], f2_view(X.gf, gid2), cord, score);
        if (cord && cord_y(cord) > y_bound && cord_x(cord) > x_bound && score < (float)thd_accept_score) tmp.push(cord);
        else break;
    }
    if (!tmp.empty()) {
        for (u32 i = 0; i < tmp.n / 2; i++) rs_swap(tmp[i], tmp[tmp.n - 1 - i]);
        cords.insert((u32)(kk + len), tmp.p, tmp.n);
        len += (int)tmp.n;
    }
    return len;
}
LNR_HD inline void gap_trim_tiles(GVec<u64> &t, u64 gap_str, u64 gap_end, u64 rvcp, int direction, GapCtx &X) {   // trimTiles gap_util.cpp:1498-1594
    const GapParms &gp = X.gp;
    u64 ts = (u64)gp.thd_tile_size;
    i64 sx = (i64)(cord_x(gap_end) - cord_x(gap_str)), sy = (i64)(cord_y(gap_end) - cord_y(gap_str));
    if (sx > (i64)ts) sx = (i64)ts;
    if (sy > (i64)ts) sy = (i64)ts;
    u64 cord_end_ = shift_cord(gap_end, -sx, -sy);
    for (int i = 0; i < (int)t.n; i++) {
        if (is_tile_start(t[(u32)i]) && direction >= 0) {
            int nn = gap_extend_patch(t, i, gap_str, t[(u32)i], gp.thd_tts_overlap_size, gp.thd_tts_gap_size, gp.thd_accept_score, X);
            if (nn) { set_tile_start(t[(u32)i]); i += nn; t[(u32)i] &= ~TILE_STR; }
        }
        if (is_tile_end(t[(u32)i]) && direction <= 0) {
            int nn = gap_extend_patch(t, i + 1, t[(u32)i], cord_end_, gp.thd_tts_overlap_size, gp.thd_tts_gap_size, gp.thd_accept_score, X);
            if (nn) { t[(u32)i] &= ~TILE_END; i += nn; set_tile_end(t[(u32)i]); }
        }
        if (i >= 1 && !is_tile_end(t[(u32)i - 1]) && !is_tile_start(t[(u32)i])) i += gap_extend_patch(t, i, t[(u32)i - 1], t[(u32)i], gp.thd_tts_overlap_size, gp.thd_tts_gap_size, gp.thd_accept_score, X);
    }
    i64 x_str = (i64)cord_x(gap_str), y_str = (i64)cord_y(gap_str), x_end = (i64)cord_x(gap_end), y_end = (i64)cord_y(gap_end);
    int di = 0;
    for (int i = 0; i < (int)t.n; i++) {
        u64 v = t[(u32)i];
        i64 x_t = (i64)cord_x(v);
        i64 y_t = tile_strand(v ^ gap_str) ? (i64)(rvcp - 1 - cord_y(v) - ts) : (i64)cord_y(v);
        if (x_t < x_str || x_t + (i64)ts > x_end || y_t < y_str || y_t + (i64)ts > y_end) {
            if (is_tile_start(v) && is_tile_end(v)) {}
            else if (is_tile_start(v)) { if (i + 1 < (int)t.n) set_tile_start(t[(u32)i + 1]); }
            else if (is_tile_end(v)) { if (i - di - 1 > 0) set_tile_end(t[(u32)(i - di - 1)]); }
            di++;
        } else t[(u32)(i - di)] = v;
    }
    if (di) t.n -= (u32)di;
}

// ---- clipping (gap_util.cpp:2169-2330)
template <class GX> LNR_HD inline void gap_accumulate_score(const GVec<u64> &ch, GVec<int> &gs, int shape_len, GX getx, const GapParms &gp) {
    gs.resize(ch.n); for (u32 i = 0; i < ch.n; i++) gs[i] = 0;
    if (ch.empty()) return;
    u64 pre = getx(ch[0]);
    for (u32 i = 1; i < ch.n; i++) {
        u64 x = getx(ch[i]);
        int ng = int(x - pre) > shape_len ? (int)(x - pre - (u64)shape_len) : 0;
        gs[i] += gs[i - 1] + ng * gp.int_precision;
        pre = x;
    }
}
LNR_HD inline int gap_clip_chain_(GVec<u64> &ch, const GVec<int> &gsx, const GVec<int> &gsy, int direction, bool f_clip, const GapParms &gp) {
    if (ch.empty()) return -1;
    bool left = direction <= 0;
    int n = (int)ch.n, clip_i = left ? -1 : n - 1, w = gp.thd_ccps_window_size, best = (int)0x80000000, found = 0;
    for (int i = 1; i < n - 1; i++) {
        int i_str = i - w > 0 ? i - w : 0, i_end = i + w < n - 1 ? i + w : n - 1, d1 = i - i_str, d2 = i_end - i;
        int cx1 = (gsx[(u32)i] - gsx[(u32)i_str]) / d1, cx2 = (gsx[(u32)i_end] - gsx[(u32)i]) / d2, cy1 = (gsy[(u32)i] - gsy[(u32)i_str]) / d1, cy2 = (gsy[(u32)i_end] - gsy[(u32)i]) / d2;
        if (left) { rs_swap(cx1, cx2); rs_swap(cy1, cy2); }
        int d = cx2 - cx1 + cy2 - cy1;
        if (d > best && cx1 < gp.thd_ccps_clip1_upper && cy1 < gp.thd_ccps_clip1_upper && (cx2 > gp.thd_ccps_clip2_lower || cy2 > gp.thd_ccps_clip2_lower)) { best = d; clip_i = i; found = 1; }
    }
    if (f_clip && found) { if (left) ch.erase(0, (u32)clip_i + 1); else ch.n = (u32)clip_i + 1; }
    return clip_i + 1;
}
template <class GX, class GY> LNR_HD inline int gap_clip_chain(GVec<u64> &ch, int shape_len, int direction, bool f_clip, GX getx, GY gety, GapCtx &X) {
    GVec<int> gsx, gsy; gsx.init(X.ar); gsy.init(X.ar);
    gap_accumulate_score(ch, gsx, shape_len, getx, X.gp);
    gap_accumulate_score(ch, gsy, shape_len, gety, X.gp);
    return gap_clip_chain_(ch, gsx, gsy, direction, f_clip, X.gp);
}
LNR_HD inline void stick_main_chain(GVec<u64> &c1, const GVec<u64> &c2, i64 thd) {          // stickMainChain :2276-2330 (chain1: gap anchors, chain2: tiles)
    if (c1.empty() || c2.empty()) return;
    int di = 0, jj = (int)c2.n - 1;
    u64 x1, x2 = cord_x(c2[(u32)jj]);
    for (int i = 0; i < (int)c1.n; i++) {
        x1 = ganc_x(c1[(u32)i]);
        if (x1 < x2) for (int j = jj - 1; j >= 0; j--) { x2 = cord_x(c2[(u32)j]); if (x1 >= x2) { jj = j; break; } }
        if (x1 < x2) jj = 0;
        i64 a1 = (i64)(x1 - ganc_y(c1[(u32)i])), a2 = (i64)(cord_x(c2[(u32)jj]) - cord_y(c2[(u32)jj]));
        if (a1 >= a2 + thd || a1 < a2 - thd) di++;
        else c1[(u32)(i - di)] = c1[(u32)i];
    }
    c1.n -= (u32)di;
}

// =================================================================== tiles <-> cords, overlaps, extension ====
LNR_HD inline void gap_reform_tiles(GVec<u64> &ts, GVec<u64> &te, u64 gap_str, u64 gap_end, int direction, const GapParms &gp) {   // reform_tiles gap_util.cpp:3042-3128
    i64 x1 = (i64)cord_x(gap_str), x2 = (i64)cord_x(gap_end), y1 = (i64)cord_y(gap_str), y2 = (i64)cord_y(gap_end), d1, d2, T = gp.thd_tile_size;
    if (!ts.empty()) {
        d1 = gmin3((i64)cord_x(ts.back()) - x1, (i64)cord_y(ts.back()) - y1, T);
        d2 = gmin3(x2 - (i64)cord_x(ts.back()), y2 - (i64)cord_y(ts.back()), T);
    } else d1 = d2 = gmin3(x2 - x1, y2 - y1, T);
    u64 head_str = gap_str, tail_end = gap_end, head_end = shift_cord(head_str, d1, d1), tail_str = shift_cord(tail_end, -d2, -d2);
    remove_tile_sgn(head_str); remove_tile_sgn(tail_str); remove_tile_sgn(head_end);
    set_tile_end(tail_str); set_tile_end(tail_end);
    if (!ts.empty()) { copy_tile_sgn(ts.back(), tail_str); copy_tile_sgn(ts[0], head_str); remove_tile_sgn(ts.back()); remove_tile_sgn(ts[0]); }
    if (direction != -1) ts.insert(0, &head_str, 1);
    if (direction != 1) ts.push(tail_str);
    if (te.empty()) {
        te.resize(ts.n);
        for (u32 i = 0; i < ts.n; i++) { i64 d = gmin3(x2 - (i64)cord_x(ts[i]), y2 - (i64)cord_y(ts[i]), T); te[i] = shift_cord(ts[i], d, d); }
    } else {
        if (direction != -1) te.insert(0, &head_end, 1);
        if (direction != 1) te.push(tail_end);
    }
}
LNR_HD inline int gap_insert_tiles1(GVec<u64> &cords, u32 &pos, GVec<u64> &tiles, int direction, int max_segs) {   // insert_tiles2Cords_ :3148-3238
    if ((tiles.n < 2 && direction == 0) || tiles.empty()) return 1;
    int segs = 0;
    for (u32 i = 0; i < tiles.n; i++) if (is_tile_end(tiles[i])) { tiles[i] |= F_END; ++segs; }
    if (segs > max_segs) return segs | (1 << 30);
    u64 recd = cords[pos] & F_RECD;
    for (u32 i = 0; i < tiles.n; i++) { u64 &t = tiles[i]; remove_tile_sgn(t); t &= ~F_MAIN; if (recd) t |= F_RECD; else t &= ~F_RECD; }   // set_tiles_cords_sgns :619-627
    if (direction == -1) {
        if (is_end(cords[pos])) tiles.back() |= F_END; else tiles.back() &= ~F_END;
        cords[pos] = tiles.back(); if (tiles.n) tiles.n--;
        cords.insert(pos, tiles.p, tiles.n);
        pos += tiles.n;
    } else if (direction == 1) {
        u64 tmp = cords[pos];
        cords[pos] = tiles[0];
        cords.insert(pos + 1, tiles.p + 1, tiles.n - 1);
        pos += tiles.n - 1;
        if (is_end(tmp)) cords[pos] |= F_END; else cords[pos] &= ~F_END;
    } else {
        u64 tmp = cords[pos];
        cords[pos - 1] = tiles[0];
        cords[pos] = tiles.back();
        if (is_end(tmp)) cords[pos] |= F_END; else cords[pos] &= ~F_END;
        cords.insert(pos, tiles.p + 1, tiles.n - 2);
        pos += tiles.n - 2;
    }
    tiles.clear();
    return 0;
}
LNR_HD inline void gap_insert_tiles(GVec<u64> &cs, GVec<u64> &ce, u32 &pos, GVec<u64> &ts, GVec<u64> &te, int direction, int max_segs) {   // :3240-3268 (cords_end is never empty here)
    u32 p2 = pos;
    gap_insert_tiles1(cs, pos, ts, direction, max_segs);
    gap_insert_tiles1(ce, p2, te, direction, max_segs);
}
template <class GX, class GY>
LNR_HD inline IPair gap_chain_overlaps(const GVec<u64> &c1, const GVec<u64> &c2, GX getX, GY getY, const GapParms &gp) {   // getExtendsIntervalChainsOverlaps :3272-3315
    IPair r;
    if (c1.empty() || c2.empty()) { r.first = (int)c1.n; r.second = 0; return r; }
    u64 x2 = getX(c2[0]), y2 = getY(c2[0]);
    x2 = x2 > gp.thd_dcomx_err_dx ? x2 - gp.thd_dcomx_err_dx : 0;
    y2 = y2 > gp.thd_dcomx_err_dy ? y2 - gp.thd_dcomx_err_dy : 0;
    int i1 = 0;
    for (int i = (int)c1.n - 1; i >= 0; i--) if (getX(c1[(u32)i]) < x2 && getY(c1[(u32)i]) < y2) { i1 = i + 1; break; }
    u64 x1 = getX(c1[c1.n - 1]) + gp.thd_dcomx_err_dx, y1 = getY(c1[c1.n - 1]) + gp.thd_dcomx_err_dy;
    x1 = (gp.ref_len - x1 > gp.thd_dcomx_err_dx) ? x1 + gp.thd_dcomx_err_dx : gp.ref_len;
    y1 = (gp.read_len - y1 > gp.thd_dcomx_err_dy) ? y1 + gp.thd_dcomx_err_dy : gp.read_len;
    int i2 = (int)c2.n;
    for (int i = 0; i < (int)c2.n; i++) if (getX(c2[(u32)i]) > x1 && getY(c2[(u32)i]) > y1) { i2 = i; break; }
    r.first = i1; r.second = i2;
    return r;
}
// re-map with the small pattern along tiles[i_str, i_end) (mapAlongChain :3320-3377): the first resulting chain, as tiles
LNR_HD inline int gap_map_along_chain(const GSeq &ref, const GSeq &seq2, const GVec<u64> &ch, GVec<u64> &tiles, int i_str, int i_end, int shape_len, int step1, int step2, GapCtx &X) {
    if (ch.empty() || i_str < 0 || i_end > (int)ch.n || i_end <= i_str) return -1;
    GP(X, 7);
    GVec<u64> hs, anc; hs.init(X.ar, 1024); anc.init(X.ar, 1024);
    u64 a = ch[(u32)i_str], b = ch[(u32)i_end - 1];
    i64 as = (i64)(cord_x(a) - cord_y(a)), ae = (i64)(cord_x(b) - cord_y(b));
    {
        GP(X, 10);
        c_stream(ref, hs, cord_x(a), cord_x(b), step1, shape_len, 0, X.coop);
        c_stream(seq2, hs, cord_y(a), cord_y(b), step2, shape_len, 1, X.coop);
        c_create_anchors2(hs, anc, (as < ae ? as : ae) - 30, (as > ae ? as : ae) + 30, X.ls->st, &X);
        gap_sort(anc.p, (long)anc.n, GapCmp{2, 0}, X);
    }
    GP2(X, 11);
    stick_main_chain(anc, ch, X.gp.thd_smcn_danchor);
    GVec<u64> first; first.init(X.ar, anc.n + 16);
    // bestn 1: only the first chain is wanted; it is collected as anchors and turned into tiles below (chn_ext_clip_metric1: min length 1, abort 0)
    gap_chain_anchors(anc.p, anc.n, first, false, 15, 30, 1, 1, 0, [](u64 p, u64 q) { return gap_clip_score(p, q); }, X, 5);
    int f_strand = (int)tile_strand(ch[0]);
    for (u32 i = 0; i < first.n; i++) { u64 t = ganc_tile(first[i]); if (f_strand) t |= 1ULL << 61; tiles.push(t); }
    return 0;
}
template <class GX, class GY>
LNR_HD inline void gap_clip_overlaps_insdel2(GVec<u64> &c1, GVec<u64> &c2, int shape_len, GX getX, GY getY, GapCtx &X) {   // __extendsIntervalClipOverlapsInsDel_ :3382-3489
    if (c1.empty() || c2.empty()) return;
    const GapParms &gp = X.gp;
    GVec<int> g11, g12, g21, g22; g11.init(X.ar); g12.init(X.ar); g21.init(X.ar); g22.init(X.ar);
    gap_accumulate_score(c1, g11, shape_len, getX, gp); gap_accumulate_score(c1, g12, shape_len, getY, gp);
    gap_accumulate_score(c2, g21, shape_len, getX, gp); gap_accumulate_score(c2, g22, shape_len, getY, gp);
    gap_clip_chain_(c1, g11, g12, 1, true, gp);
    gap_clip_chain_(c2, g21, g22, -1, true, gp);
    int j1 = 0, j2 = 0, i_clip = 0, j_clip = -1, j1_pre = 0, j2_pre = 0, min_score = 0x7fffffff;
    u64 x21 = getX(c2[0]), x22 = getX(c2[0]);
    for (int i = 0; i < (int)c1.n; i++) {
        u64 x1 = getX(c1[(u32)i]), lo = x1, up = x1 + gp.thd_eicos_clip_dxy;
        for (int j = j1_pre; j < (int)c2.n && x21 < lo; j++) { x21 = getX(c2[(u32)j]); j1 = j; }
        if (x21 > up) continue;
        if (x21 < lo) break;
        for (int j = j2_pre; j < (int)c2.n && x22 <= up; j++) { x22 = getX(c2[(u32)j]); j2 = j; }
        if (x22 < lo) break;
        if (j1 > j_clip || j2_pre != j2) {
            int s11 = g11[(u32)i], s12 = g12[(u32)i];
            for (int j = (j1 > j2_pre ? j1 : j2_pre); j < j2; j++) {
                int s21 = g21[g21.n - 1] - g21[(u32)j], s22 = g22[g22.n - 1] - g22[(u32)j];
                int s_con = (i64)(getX(c2[(u32)j]) - getX(c1[(u32)i])) > shape_len ? (int)((getX(c2[(u32)j]) - getX(c1[(u32)i]) - (u64)shape_len) * (u64)gp.int_precision) : 0;
                int score = s11 + s12 + s21 + s22 + s_con;
                if (score < min_score) { min_score = score; i_clip = i; j_clip = j; }
            }
        }
        j1_pre = j1; j2_pre = j2;
    }
    c1.n = (u32)i_clip;
    c2.erase(0, (u32)(j_clip < 0 ? 0 : j_clip));
}
template <class GX, class GY>
LNR_HD inline void gap_clip_overlaps_insdel(GVec<u64> &c1, GVec<u64> &c2, int shape_len, GX getX, GY getY, GapCtx &X) {   // :3492-3522
    if (c1.empty() && c2.empty()) return;
    if (c1.empty()) gap_clip_chain(c2, shape_len, -1, true, getX, getY, X);
    else if (c2.empty()) gap_clip_chain(c1, shape_len, 1, true, getX, getY, X);
    else if (!X.gp.thd_eicos_f_as_ins) { gap_clip_chain(c1, shape_len, 1, true, getX, getY, X); gap_clip_chain(c2, shape_len, -1, true, getX, getY, X); }
    else gap_clip_overlaps_insdel2(c1, c2, shape_len, getX, getY, X);
}
LNR_HD inline u64 gtx(u64 v) { return cord_x(v); }
LNR_HD inline u64 gty(u64 v) { return cord_y(v); }
LNR_HD inline void gap_map_overlaps(const GSeq &ref, GVec<u64> &t1, GVec<u64> &t2, u64 gap_str1, u64 gap_end2, int shape_len, int step1, int step2, GapCtx &X) {   // extendsIntervalMapOverlaps_ :3577-3639
    drop_chain_gap_x(t1, 1, X.gp);
    drop_chain_gap_x(t2, -1, X.gp);
    GVec<u64> o1, o2; o1.init(X.ar); o2.init(X.ar);
    IPair ov = gap_chain_overlaps(t1, t2, gtx, gty, X.gp);
    if (!t1.empty()) gap_map_along_chain(ref, tile_strand(t1[0]) ? X.com : X.read, t1, o1, ov.first, (int)t1.n, shape_len, step1, step2, X);
    if (!t2.empty()) gap_map_along_chain(ref, tile_strand(t2[0]) ? X.com : X.read, t2, o2, 0, ov.second, shape_len, step1, step2, X);
    if (cord_x(gap_str1) - cord_y(gap_str1) > cord_x(gap_end2) - cord_y(gap_end2)) gap_clip_overlaps_insdel(o1, o2, shape_len, gtx, gty, X);
    else gap_clip_overlaps_insdel(o1, o2, shape_len, gty, gtx, X);
    t1.n = (u32)ov.first;
    t1.append(o1);
    t2.erase(0, (u32)ov.second);
    t2.insert(0, o2.p, o2.n);
}
LNR_HD inline void gap_remap_chain_one_end(const GSeq &ref, GVec<u64> &ch, int shape_len, int step1, int step2, int remap_num, int direction, GapCtx &X) {   // remapChainOneEnd :3761-3812
    if (!direction || ch.empty()) return;
    const GSeq &seq2 = tile_strand(ch[0]) ? X.com : X.read;
    GVec<u64> re; re.init(X.ar);
    int i_str, i_end;
    if (direction <= 0) { i_str = (int)ch.n - remap_num > 0 ? (int)ch.n - remap_num : 0; i_end = (int)ch.n; }
    else { i_str = 0; i_end = (int)ch.n < remap_num ? (int)ch.n : remap_num; }
    gap_map_along_chain(ref, seq2, ch, re, i_str, i_end, shape_len, step1, step2, X);
    gap_clip_chain(re, shape_len, direction, true, gtx, gty, X);
    if (direction <= 0) { ch.erase(0, (u32)i_end); ch.insert(0, re.p, re.n); }
    else if (!re.empty()) { ch.n = (u32)i_str; ch.append(re); }
}
LNR_HD inline int gap_reextend_chain_one_side(const GSeq &ref, GVec<u64> &ch, int ips, int ipe, int lower, int upper, int shape_len, int step1, int step2, int direction, GapCtx &X) {   // :3832-3916
    if (ch.empty() || ips < 0 || ipe < 0) return 0;
    int ii, i_str, i_end, len = (int)ch.n;
    GVec<u64> re; re.init(X.ar);
    if (direction <= 0) {
        i64 d = -gmin3((i64)cord_x(ch[(u32)ips]), (i64)cord_y(ch[(u32)ips]), (i64)lower);
        for (ii = ips; ii < ipe; ii++) if ((i64)(cord_x(ch[(u32)ii]) - cord_x(ch[(u32)ips])) >= upper) break;
        re.resize((u32)(ii - ips + 2));
        re[0] = shift_cord(ch[(u32)ips], d, d);
        for (int i = 0; i < ii - ips + 1; i++) re[(u32)i + 1] = ch[(u32)(ips + i)];
        i_str = ips; i_end = ii + 1;
    } else {
        int d = (int)gmin3((i64)(ref.len - cord_x(ch[(u32)ipe]) - 1), (i64)(X.read.len - cord_y(ch[(u32)ipe]) - 1), (i64)upper);
        for (ii = ipe; ii > ips; ii--) if ((i64)(cord_x(ch[(u32)ipe]) - cord_x(ch[(u32)ii])) >= lower) break;
        re.resize((u32)(ipe - ii + 2));
        for (int i = 0; i < ipe - ii + 1; i++) re[(u32)i] = ch[(u32)(ii + i)];
        re.back() = shift_cord(ch[(u32)ipe], d, d);
        i_str = ii; i_end = ipe + 1;
    }
    gap_remap_chain_one_end(ref, re, shape_len, step1, step2, (int)re.n, direction, X);
    ch.erase((u32)i_str, (u32)i_end);
    ch.insert((u32)i_str, re.p, re.n);
    return (int)ch.n - len;
}
LNR_HD inline void gap_extend_interval_one_side(const GSeq &ref, GVec<u64> &tiles, u64 gap_str, u64 gap_end, int direction, GapCtx &X) {   // extendIntervalOneSide :3953-3983 + extendTilesOneSide :3920-3950
    if (cord_strand(gap_str ^ gap_end)) return;
    GapParms &gp = X.gp;
    int od = gp.direction;
    gp.direction = direction;
    GVec<u64> g_hs, anc, chain; g_hs.init(X.ar, 2048); anc.init(X.ar, 2048); chain.init(X.ar, 256);
    { GP(X, 2); g_stream(ref, X.read, g_hs, gap_str, gap_end, (u32)gp.thd_eis_shape_len, gp.thd_eis_step1, gp.thd_eis_step2, X.coop); }
    g_create_anchors(g_hs, anc, gp.thd_eis_shape_len, direction, 0, 0, X.read.len - 1, gap_str, gap_end, X);
    g_chains_from_anchors(anc, chain, X.read.len, X);
    closest_extension_chain(chain, gap_str, gap_end, true, gp);
    gap_remap_chain_one_end(ref, chain, gp.thd_etfas_shape_len, gp.thd_etfas_step1, gp.thd_etfas_step2, 50, direction, X);
    tiles_from_chain(chain, tiles, gap_str, gap_end, 0, (int)chain.n, X);
    gap_trim_tiles(tiles, gap_str, gap_end, X.read.len - 1, direction, X);
    gp.direction = od;
}
LNR_HD inline void gap_extend_result_filter(GVec<u64> &ts, GVec<u64> &te, u64 gap_str, u64 gap_end, int direction, const GapParms &gp) {   // mapExtendResultFilter_ :3986-4031
    if (direction >= 0) {
        u64 pre = gap_str;
        for (int i = 0; i < (int)ts.n; i++) {
            i64 dy = (i64)(cord_y(ts[(u32)i]) - cord_y(pre)), dx = (i64)(cord_y(ts[(u32)i]) - cord_x(pre));   // (y - x: as in the reference)
            if (dy > gp.thd_me_reject_gap || dx > gp.thd_me_reject_gap) { ts.n = (u32)i; if (!te.empty()) te.erase((u32)i, te.n); break; }
            pre = ts[(u32)i];
        }
    }
    if (direction <= 0) {
        u64 pre = gap_end;
        for (int i = (int)ts.n - 1; i >= 0; i--) {
            i64 dy = (i64)(cord_y(pre) - cord_y(ts[(u32)i])), dx = (i64)(cord_y(pre) - cord_x(ts[(u32)i]));
            if (dy > gp.thd_me_reject_gap || dx > gp.thd_me_reject_gap) { ts.erase(0, (u32)i + 1); if (!te.empty()) te.erase(0, (u32)i + 1); break; }
            pre = ts[(u32)i];
        }
    }
}
LNR_HD inline void gap_map_extend(const GSeq &ref, GVec<u64> &ts, GVec<u64> &te, u64 gap_str, u64 gap_end, int direction, GapCtx &X) {   // mapExtend :4035-4069
    GapParms &gp = X.gp;
    float rate0 = gp.thd_gmsa_d_anchor_rate;
    gp.direction = direction; gp.thd_ctfas2_connect_danchor = 50; gp.thd_ctfas2_connect_dy_dx = 150; gp.thd_cts_major_limit = 3; gp.thd_gmsa_d_anchor_rate = 0.25f;
    gap_extend_interval_one_side(ref, ts, gap_str, gap_end, direction, X);
    gap_extend_result_filter(ts, te, gap_str, gap_end, direction, gp);
    if (!ts.empty() && direction >= 0) ts.back() &= ~TILE_END;
    gap_reform_tiles(ts, te, gap_str, gap_end, direction, gp);
    gp.thd_gmsa_d_anchor_rate = rate0;
}
LNR_HD inline void gap_map_extends(const GSeq &ref, GVec<u64> &ts1, GVec<u64> &te1, GVec<u64> &ts2, GVec<u64> &te2, u64 gs1, u64 ge1, u64 gs2, u64 ge2, GapCtx &X) {   // mapExtends :4073-4125 + extendsInterval :3696-3757
    GapParms &gp = X.gp;
    gp.thd_ctfas2_connect_danchor = 50; gp.thd_ctfas2_connect_dy_dx = 150; gp.thd_cts_major_limit = 3;
    int od = gp.direction, oclip = gp.f_rfts_clip;
    gp.f_rfts_clip = 0;
    if (!(cord_strand(gs1 ^ ge1) || cord_strand(gs2 ^ ge2) || cord_strand(gs1 ^ gs2))) {
        GVec<u64> g_hs, a1, a2, tmp1, tmp2; g_hs.init(X.ar, 2048); a1.init(X.ar, 2048); a2.init(X.ar, 2048); tmp1.init(X.ar, 256); tmp2.init(X.ar, 256);
        u64 id = cord_id(gs1), strand = cord_strand(gs1);
        u64 x1 = cord_x(gs1) < cord_x(gs2) ? cord_x(gs1) : cord_x(gs2), y1 = cord_y(gs1) < cord_y(gs2) ? cord_y(gs1) : cord_y(gs2);
        u64 x2 = cord_x(ge1), y2 = cord_y(ge1) > cord_y(ge2) ? cord_y(ge1) : cord_y(ge2);                   // (x of gap_end1 twice in the reference)
        { GP(X, 2); g_stream(ref, X.read, g_hs, create_cord(id, x1, y1, strand), create_cord(id, x2, y2, strand), (u32)gp.thd_eis_shape_len, gp.thd_eis_step1, gp.thd_eis_step2, X.coop); }
        g_create_anchor_pair(g_hs, a1, a2, gp.thd_eis_shape_len, X.read.len - 1, gs1, ge1, gs2, ge2, X);
        int od2 = gp.direction;                                                                             // extendsTilesFromAnchors :3643-3692
        gp.direction = 1;
        g_chains_from_anchors(a1, tmp1, X.read.len, X);
        closest_extension_chain(tmp1, gs1, ge1, true, gp);
        gp.direction = -1;
        g_chains_from_anchors(a2, tmp2, X.read.len, X);
        closest_extension_chain(tmp2, gs2, ge2, true, gp);
        gap_map_overlaps(ref, tmp1, tmp2, gs1, ge2, gp.thd_etfas_shape_len, gp.thd_etfas_step1, gp.thd_etfas_step2, X);
        tiles_from_chain2(tmp1, ts1, te1, gs1, ge1, 0, (int)tmp1.n, X);
        tiles_from_chain2(tmp2, ts2, te2, gs2, ge2, 0, (int)tmp2.n, X);
        gp.direction = od2;
    }
    gp.direction = 1;
    gap_extend_result_filter(ts1, te1, gs1, ge1, 1, gp);
    if (!ts1.empty()) ts1.back() &= ~TILE_END;
    gap_reform_tiles(ts1, te1, gs1, ge1, 1, gp);
    gp.direction = -1;
    gap_extend_result_filter(ts2, te2, gs2, ge2, -1, gp);
    gap_reform_tiles(ts2, te2, gs2, ge2, -1, gp);
    gp.direction = od;
    gp.f_rfts_clip = oclip;
}
LNR_HD inline int gap_reextend_clip_one_side(const GSeq &ref, GVec<u64> &ch, u64 lo_c, u64 up_c, int ips, int ipe, int direction, GapCtx &X) {   // reExtendClipOneSide :4129-4168
    if (ch.empty() || ips < 0 || ipe < 0) return 0;
    int lower = 60, upper = 60;
    if (direction <= 0) {
        int dx = (int)(cord_x(ch[(u32)ips]) - cord_x(lo_c));
        int dy = (tile_strand(ch[(u32)ips]) ^ tile_strand(lo_c)) ? (int)(cord_y(up_c) - X.read.len + cord_y(ch[(u32)ips])) : (int)(cord_y(ch[(u32)ips]) - cord_y(lo_c));
        lower = gmin3(dx, dy, lower);
    } else {
        int dx = (int)(cord_x(up_c) - 1 - cord_x(ch[(u32)ipe]));
        int dy = (tile_strand(ch[(u32)ipe]) ^ tile_strand(up_c)) ? (int)(X.read.len - 1 - cord_y(ch[(u32)ipe]) - cord_y(lo_c)) : (int)(cord_y(up_c) - cord_y(ch[(u32)ipe]));
        upper = gmin3(dx, dy, upper);
    }
    return gap_reextend_chain_one_side(ref, ch, ips, ipe, lower, upper, X.gp.thd_etfas_shape_len, X.gp.thd_etfas_step1, X.gp.thd_etfas_step2, direction, X);
}
LNR_HD inline void gap_tiles_from_anchors2(const GSeq &ref, GVec<u64> &anchors, GVec<u64> &ts, GVec<u64> &te, u64 gap_str, u64 gap_end, u64 read_len, GapCtx &X) {   // createTilesFromAnchors2_ :4171-4247
    GVec<u64> tmp; tmp.init(X.ar, 256);
    g_chains_from_anchors(anchors, tmp, read_len, X);
    int pre_i = 0;
    for (int i = 0; i < (int)tmp.n; i++) {
        bool blk_end = is_tile_end(tmp[(u32)i]) != 0;
        bool flip = !blk_end && i < int(tmp.n - 1) && tile_strand(tmp[(u32)i] ^ tmp[(u32)i + 1]);
        if (!blk_end && !flip) continue;
        u32 len0 = ts.n;
        u64 head = tmp[(u32)pre_i], tail = tmp[(u32)i];
        i += gap_reextend_clip_one_side(ref, tmp, gap_str, gap_end, pre_i, i, -1, X);
        i += gap_reextend_clip_one_side(ref, tmp, gap_str, gap_end, pre_i, i, 1, X);
        if (!(tmp.empty() || pre_i < 0 || i < 0)) {
            copy_tile_sgn(head, tmp[(u32)pre_i]);
            copy_tile_sgn(tail, tmp[(u32)i]);
            tiles_from_chain2(tmp, ts, te, gap_str, gap_end, pre_i, i + 1, X);
            if (flip && len0 != ts.n) { ts.back() &= ~TILE_END; te.back() &= ~TILE_END; }
        }
        pre_i = i + 1;
    }
}
LNR_HD inline void gap_filter_anchors(GVec<u64> &a, GapCtx &X) {                            // filterGapAnchors gap_util.cpp:4275-4441 (density 20, accept 20, err bit 0)
    GP(X, 9);
    GVec<UP> list; list.init(X.ar);
    if (a.n > 1) {
        a[0] = 0;
        gap_sort(a.p, (long)a.n, GapCmp{1, 0}, X);
#if defined(__HIP_DEVICE_COMPILE__)
        if (X.coop) {
            // the walk over the sorted anchors, 64 elements per step.  Inside a block that started at b the running "median" anchor the serial
            // loop compares element i with is a[(b + i - 1) >> 1] -- set by the step before, which was a continuation or the block's start -- so
            // the continuation test of every element of the block is known from (b, i) alone; the block ends at the first element that fails it.
            // Noise anchors make blocks of one or two elements: the walk then must not pay a memory round trip per block.  A step holds its 64
            // elements and the 64 before them in registers; the blocks inside the step are closed one after the other from those (the median
            // anchor of a short block lies in the window: a shuffle; of a long one: a load, which long blocks amortise).
            const u32 n = a.n, lane = threadIdx.x & 63;
            u64 b = 1, count = 1, min_y = ganc_y(a[1]), max_y = min_y;        // (the serial loop's first step: element 1 opens a block, nothing is listed)
            u32 i0 = 2;
            u64 w_lo = lane < 2 ? a.p[lane] : 0;                              // elements i0 - 64 + lane (only those >= 0 are ever asked for)
            {   // align: w_lo holds elements [i0 - 64, i0): for i0 = 2 that is lanes 62, 63 = elements 0, 1
                u64 t = __shfl(w_lo, (int)((lane + 2) & 63));
                w_lo = lane >= 62 ? t : 0;
            }
            bool done = false;
            while (i0 < n && !done) {
                const u32 nin = n - i0 < 64 ? n - i0 : 64;
                const u32 i = i0 + lane;
                const bool in = lane < nin;
                const u64 ai = in ? a.p[i] : 0;
                const u64 y = ganc_y(ai);
                u32 cur = 0;                                                   // lanes below cur are dealt with
                while (cur < nin) {
                    // the median anchor of every remaining element, for the block that starts at b
                    u64 k = (b + i - 1) >> 1;
                    u64 ak;
                    {
                        u64 from_hi = __shfl(ai, (int)((u32)(k - i0) & 63)), from_lo = __shfl(w_lo, (int)((u32)(k + 64 - i0) & 63));
                        bool far = in && lane >= cur && k + 64 < (u64)i0;
                        ak = k >= (u64)i0 ? from_hi : from_lo;
                        if (__any(far)) { if (far) ak = a.p[(u32)k]; }
                    }
                    u64 dy2 = (u64)gabs((i64)(y - ganc_y(ak)));
                    bool cont = in && lane >= cur && ganc_stranchor(ai) - ganc_stranchor(ak) < dy2;
                    u64 mfail = __ballot(in && lane >= cur && !cont);
                    u32 f = mfail ? (u32)__builtin_ctzll(mfail) : nin;          // elements [cur, f) continue the block
                    bool mine = lane >= cur && lane < f;
                    u64 ly = mine ? y : min_y, hy = mine ? y : max_y;
                    for (int m = 32; m; m >>= 1) { u64 o1 = __shfl_xor(ly, m), o2 = __shfl_xor(hy, m); ly = o1 < ly ? o1 : ly; hy = o2 > hy ? o2 : hy; }
                    if (ly < min_y) min_y = ly;
                    if (hy > max_y) max_y = hy;
                    count += f - cur;
                    bool last_cont = !mfail && i0 + nin == n;                    // the array ends inside the block: the serial loop closes it at i = n - 1
                    if (mfail || last_cont) {
                        u64 iend = mfail ? (u64)i0 + f : (u64)n - 1;
                        u64 acc = ((max_y - min_y) * 20) >> 10; if (acc < 20) acc = 20;
                        if (count > acc) { UP e; e.first = b; e.second = iend; list.push(e); }
                        if (!mfail) { done = true; break; }
                        b = iend; count = 1;
                        u64 yf = __shfl(y, (int)f);
                        min_y = yf; max_y = yf;
                        cur = f + 1;
                        if (i0 + cur >= n) { done = true; break; }             // (the failing element was the last one: it opened a block of its own, count 1)
                    } else cur = nin;
                }
                w_lo = ai;                                                     // (a full step: the next window's lower half; a partial one ends the walk)
                i0 += nin;
            }
        } else
#endif
        {
        u64 ak2 = a[1], block_str = 1, count = 0, min_y = ~0ULL, max_y = 0;
        for (u32 i = 1; i < a.n; i++) {
            u64 y = ganc_y(a[i]);
            u64 dy2 = (u64)gabs((i64)(y - ganc_y(ak2)));
            bool cont = ganc_stranchor(a[i]) - ganc_stranchor(ak2) < dy2;
            if (cont) { if (min_y > y) min_y = y; if (max_y < y) max_y = y; ak2 = a[(u32)((block_str + i) >> 1)]; ++count; }
            if (!cont || i == a.n - 1) {
                u64 acc = ((max_y - min_y) * 20) >> 10; if (acc < 20) acc = 20;
                if (count > acc) { UP e; e.first = block_str; e.second = i; list.push(e); }
                block_str = i; ak2 = a[i]; min_y = y; max_y = y; count = 1;
            }
        }
        }
    }
    if (!list.empty()) {
        ref_sort(list.p, (long)list.n, [](const UP &p, const UP &q) { return (u32)(p.second - p.first) > (u32)(q.second - q.first); }, X.ls->st);
        if (a.n > 1000 && list.n > 10) {
            u32 im = list.n / 2, lm = (u32)(list[im].second - list[im].first), lx = (u32)(list[0].second - list[0].first);
            if ((float)lx > (float)lm * 1.5f && lx > lm + 20) {
                u32 it = 0, ls_ = 0, li = 0;
                u64 brk = (u64)((float)lm * 1.5f);
                for (u32 i = 0; i < (list.n < 5 ? list.n : 5u); i++) { it++; li = (u32)(list[i].second - list[i].first); ls_ += li; if (li < brk || ls_ > 2000) break; }
                list.n = it;
            } else list.clear();
        }
    }
    u32 it = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    if (X.coop) {
        // the in-place copy of the kept blocks, 64 elements per step.  The serial loop reads a[j] after it wrote a[it]: inside one step that only
        // matters when the destination runs less than 64 elements ahead of the source (an element written by this very step would be read by it);
        // such a step is taken element by element.
        const u32 lane = threadIdx.x & 63;
        for (u32 i = 0; i < list.n; i++) {
            u64 j = list[i].first, je = list[i].second;
            while (j < je) {
                u32 m = je - j < 64 ? (u32)(je - j) : 64u;
                if ((u64)it > j && (u64)it - j < 64) { for (u32 t = 0; t < m; t++) a.p[it + t] = a.p[(u32)j + t]; }
                else { u64 v = lane < m ? a.p[(u32)j + lane] : 0; WSYNC(); if (lane < m) a.p[it + lane] = v; }
                WSYNC();
                it += m; j += m;
            }
        }
        a.n = it;
        return;
    }
#endif
    for (u32 i = 0; i < list.n; i++) for (u64 j = list[i].first; j < list[i].second; j++) a[it++] = a[(u32)j];
    a.n = it;
}
LNR_HD inline void gap_map_generic(const GSeq &ref, GVec<u64> &ts, GVec<u64> &te, u64 gap_str, u64 gap_end, GapCtx &X) {   // mapGeneric :4492-4515 over mapInterval :4444-4489
    int oclip = X.gp.f_rfts_clip;
    X.gp.f_rfts_clip = 0;
    if (!cord_strand(gap_str ^ gap_end)) {
        GVec<u64> g_hs, anc; g_hs.init(X.ar, 2048); anc.init(X.ar, 2048);
        { GP(X, 2); g_stream(ref, X.read, g_hs, gap_str, gap_end, 9, 5, 1, X.coop); }
        g_create_anchors(g_hs, anc, 9, 0, -(((i64)1 << 62) - 1), ((i64)1 << 62) - 1, X.read.len - 1, gap_str, gap_end, X);
        if (anc.n > 1000) gap_filter_anchors(anc, X);
        gap_tiles_from_anchors2(ref, anc, ts, te, gap_str, gap_end, X.read.len - 1, X);
    }
    gap_reform_tiles(ts, te, gap_str, gap_end, 0, X.gp);
    X.gp.f_rfts_clip = oclip;
}

// =================================================================== mapGap_, mapGaps, reformCords ====
LNR_HD inline void gvec_cat(GVec<u64> &d, const GVec<u64> &s) { d.insert(d.n, s.p, s.n); }
// the addon of mapGap_: mapGeneric between two tiles, its inner tiles spliced in before tiles_str[i] (gap.cpp:232-271, :303-362)
LNR_HD inline u32 gap_splice_generic(const GSeq &ref, GVec<u64> &ts, GVec<u64> &te, u32 i, u64 from, u64 to, bool dup_marks, GapCtx &X) {
    u64 m = X.ar->mark();
    GVec<u64> t1, e1; t1.init(X.ar, 64); e1.init(X.ar, 64);
    gap_map_generic(ref, t1, e1, from, to, X);
    u32 added = 0;
    if (!t1.empty()) {
        t1.erase(0, 1); e1.erase(0, 1);
        if (t1.n) t1.n--;
        if (e1.n) e1.n--;
        if (!t1.empty()) {
            remove_tile_sgn(t1.back()); remove_tile_sgn(e1.back());
            if (dup_marks) {
                if (cord_x(t1[0]) < cord_x(ts[i - 1])) { set_tile_end(ts[i - 1]); set_tile_end(te[i - 1]); }
                if (cord_x(t1.back()) > cord_x(ts[i])) { set_tile_end(t1.back()); set_tile_end(e1.back()); }
            }
            ts.insert(i, t1.p, t1.n);
            te.insert(i, e1.p, e1.n);
        }
        added = t1.n;
    }
    X.ar->release(m);
    return added;
}
// map the gap [gap_str, gap_end) (mapGap_ gap.cpp:16-395).  tiles_str / tiles_end live in an arena of their own (the caller's): the
// temporaries of this gap are released from X.ar on return.
LNR_HD inline int gap_map_gap(u64 gap_str, u64 gap_end, GVec<u64> &tiles_str, GVec<u64> &tiles_end, int direction, GapCtx &X) {
    GapParms &gp = X.gp;
    tiles_str.clear(); tiles_end.clear();
    gap_str &= ~F_END; gap_end &= ~F_END;
    remove_tile_sgn(gap_str); remove_tile_sgn(gap_end);
    GSeq ref = X.ref(cord_id(gap_str));
    i64 ref_len = (i64)ref.len, read_len = (i64)X.read.len;
    i64 x1 = (i64)cord_x(gap_str), x2 = (i64)cord_x(gap_end), y1 = (i64)cord_y(gap_str), y2 = (i64)cord_y(gap_end), shift_x, shift_y, T = gp.thd_tile_size;
    u64 m0 = X.ar->mark();
    if (x1 + T > ref_len - 1 || y1 + T > read_len - 1 || x2 > ref_len - 1 || y2 > read_len - 1 || x2 < T || y2 < T) return 0;
    else if (cord_strand(gap_str ^ gap_end)) {
        if (direction != 0) return -1;
        const i64 ext1 = 500, ext2 = 5000;
        GVec<u64> ts1, ts2, te1, te2; ts1.init(X.ar, 64); ts2.init(X.ar, 64); te1.init(X.ar, 64); te2.init(X.ar, 64);
        shift_x = (x2 - x1 > 0) ? gmin3(ext2, (i64)(ref.len - 1 - cord_x(gap_str)), x2 - x1) : ext1;
        shift_y = (i64)((float)(x2 - x1) * (1 + gp.thd_err));
        if (shift_y > (i64)(X.read.len - 1 - cord_y(gap_str))) shift_y = (i64)(X.read.len - 1 - cord_y(gap_str));
        if (shift_x < 0) shift_x = 0;
        if (shift_y < 0) shift_y = 0;
        gap_map_extend(ref, ts1, te1, gap_str, shift_cord(gap_str, shift_x, shift_y), 1, X);     // (mapExtend takes seqs[id of ITS gap_str], gap_util.cpp:4055)
        shift_x = (x2 - x1 > 0) ? gmin3(x2 - x1, (i64)cord_x(gap_end), ext2) : ext1;
        shift_y = (i64)((float)(x2 - x1) * (1 + gp.thd_err));
        if (shift_y > (i64)cord_y(gap_end)) shift_y = (i64)cord_y(gap_end);
        if (shift_x < 0) shift_x = 0;
        if (shift_y < 0) shift_y = 0;
        { u64 gs_ = shift_cord(gap_end, -shift_x, -shift_y); gap_map_extend(X.ref(cord_id(gs_)), ts2, te2, gs_, gap_end, -1, X); }   // the other sequence when the gap joins cords of two
        if (!ts1.empty()) { gvec_cat(tiles_str, ts1); gvec_cat(tiles_end, te1); }
        if (!ts2.empty()) { gvec_cat(tiles_str, ts2); gvec_cat(tiles_end, te2); }
    } else if (x1 + T > x2 || y1 + T > y2) return 0;
    else if (y1 < y2) {
        i64 danc = x1 - x2 - y1 + y2;
        if (gabs(danc) > gp.thd_mg1_danc_indel && direction == 0) {
            int f_extends = 1;
            int s1m = gp.chn1_min_len, s1a = gp.chn1_abort, s1f = gp.chn1_fn, s2m = gp.chn2_min_len, s2a = gp.chn2_abort, s2f = gp.chn2_fn;
            gp.chn1_min_len = 1; gp.chn1_abort = 0; gp.chn1_fn = 2; gp.chn2_abort = 0; gp.chn2_fn = 3;
            GVec<u64> ts1, ts2, te1, te2; ts1.init(X.ar, 64); ts2.init(X.ar, 64); te1.init(X.ar, 64); te2.init(X.ar, 64);
            u64 gs1 = 0, gs2 = 0, ge1 = 0, ge2 = 0;
            i64 dyy = y2 - y1 > 0 ? y2 - y1 : 0, dxx = x2 - x1 > 0 ? x2 - x1 : 0, E = gp.thd_max_extend2;
            if (danc > 0) {
                shift_y = gmin3(dyy, E, (i64)(X.read.len - y1 - 1));
                shift_x = gmin3((i64)((float)shift_y * (1 + gp.thd_err)), E, (i64)(ref.len - x1 - 1));
                gs1 = gap_str; ge1 = shift_cord(gap_str, shift_x, shift_y);
                shift_y = gmin3(dyy, E, y2);
                shift_x = gmin3((i64)((float)shift_y * (1 + gp.thd_err)), E, x2);
                gs2 = shift_cord(gap_end, -shift_x, -shift_y); ge2 = gap_end;
                f_extends = x1 < x2 ? 1 : 2;
            } else if (x1 < x2) {
                shift_x = gmin3(dxx, E, (i64)(ref.len - x1 - 1));
                shift_y = gmin3((i64)((float)shift_x * (1 + gp.thd_err)), E, (i64)(X.read.len - y1 - 1));
                gs1 = gap_str; ge1 = shift_cord(gap_str, shift_x, shift_y);
                shift_x = gmin3(dxx, E, x2);
                shift_y = gmin3((i64)((float)shift_x * (1 + gp.thd_err)), E, y2);
                gs2 = shift_cord(gap_end, -shift_x, -shift_y); ge2 = gap_end;
                f_extends = 1;
            } else f_extends = 0;
            if (f_extends) {
                if (f_extends == 1) gap_map_extends(ref, ts1, te1, ts2, te2, gs1, ge1, gs2, ge2, X);
                else { gap_map_extend(ref, ts1, te1, gs1, ge1, 1, X); gap_map_extend(X.ref(cord_id(gs2)), ts2, te2, gs2, ge2, -1, X); }
                if (!ts1.empty()) { gvec_cat(tiles_str, ts1); gvec_cat(tiles_end, te1); remove_tile_sgn(tiles_str.back()); remove_tile_sgn(tiles_end.back()); }
                if (!ts2.empty()) { remove_tile_sgn(ts2[0]); remove_tile_sgn(te2[0]); gvec_cat(tiles_str, ts2); gvec_cat(tiles_end, te2); }
            }
            gp.chn1_min_len = s1m; gp.chn1_abort = s1a; gp.chn1_fn = s1f; gp.chn2_min_len = s2m; gp.chn2_abort = s2a; gp.chn2_fn = s2f;
        }
    }
    X.ar->release(m0);
    u64 v = gap_str; tiles_str.insert(0, &v, 1);
    v = shift_cord(gap_str, 1, 1); tiles_end.insert(0, &v, 1);
    tiles_str.push(shift_cord(gap_end, -1, -1));
    tiles_end.push(gap_end);
    for (u32 i = 1; i < tiles_str.n; i++) {                                                     // addon 1: what is still open between consecutive tiles
        i64 dx = (i64)(cord_x(tiles_str[i]) - cord_x(tiles_end[i - 1])), dy = (i64)(cord_y(tiles_str[i]) - cord_y(tiles_end[i - 1]));
        if (tile_strand(tiles_str[i] ^ tiles_str[i - 1])) continue;
        if (dx > 90 && dy > 90) i += gap_splice_generic(X.ref(cord_id(tiles_str[i - 1])), tiles_str, tiles_end, i, tiles_str[i - 1], tiles_str[i], false, X);   // mapGeneric: seqs[id of its gap_str] (gap_util.cpp:4508)
    }
    if (gp.f_dup) {                                                                              // addon 2: duplications (-dup 1)
        const float rate = 0.1f;
        for (u32 i = 1; i < tiles_str.n; i++) {
            if (tile_strand(tiles_str[i] ^ tiles_str[i - 1]) || is_tile_end(tiles_str[i - 1])) continue;
            i64 xa = (i64)cord_x(tiles_end[i - 1]), ya = (i64)cord_y(tiles_end[i - 1]), xb = (i64)cord_x(tiles_str[i]), yb = (i64)cord_y(tiles_str[i]);
            i64 dx = xb - xa, dy = yb - ya;
            if (dy > 100 && dy - dx > gp.thd_mg1_danc_indel) {
                i64 w = (i64)((float)dy * (1 + rate));
                i64 e1 = -(w < xa ? w : xa), lim = (i64)(ref.len - (u64)xb - 1), e2 = w < lim ? w : lim;
                { u64 gs_ = shift_cord(tiles_end[i - 1], e1, 0); i += gap_splice_generic(X.ref(cord_id(gs_)), tiles_str, tiles_end, i, gs_, shift_cord(tiles_str[i], e2, 0), true, X); }
            }
        }
    }
    for (int i = 1; i < (int)tiles_str.n - 1; i++) { tiles_str[(u32)i - 1] = tiles_str[(u32)i]; tiles_end[(u32)i - 1] = tiles_end[(u32)i]; }
    tiles_str.n = tiles_str.n >= 2 ? tiles_str.n - 2 : 0; tiles_end.n = tiles_end.n >= 2 ? tiles_end.n - 2 : 0;
    return 0;
}
LNR_HD inline i64 gap_max_gapsy_overlap(const UP *gapsy, u32 n, u64 gap_str, u64 gap_end) {   // _getMaxGapsyOverlap gap_util.cpp:343-362
    i64 gs = (i64)cord_y(gap_str), ge = (i64)cord_y(gap_end);
    for (u32 i = 0; i < n; i++) {
        i64 ys = (i64)gapsy[i].first, ye = (i64)gapsy[i].second;
        if (gs >= ys && gs <= ye) return (ge < ye ? ge : ye) - gs;
        else if (ge >= ys && ge <= ye) return ge - (gs > ys ? gs : ys);
    }
    return 0;
}
// re-map the gaps of one read's cords, the ends of the read included (mapGaps gap.cpp:407-576).  cs / ce and the tile lists live in
// `keep`; X.ar is the scratch of one gap.
LNR_HD inline int gap_map_gaps(GVec<u64> &cs, GVec<u64> &ce, GArena &keep, GapCtx &X) {
    if (cs.n <= 1) return 0;
    GapParms &gp = X.gp;
    GVec<u64> tiles_str, tiles_end; tiles_str.init(&keep, 256); tiles_end.init(&keep, 256);
    const int max_segs = 1000;
    const u64 max_extend = 2000;
    const i64 max_gap = 3000, extend_xy = 3;
    i64 block_size = gp.thd_tile_size, cord_gap = gp.thd_gap_len_min + block_size;
    u64 L = X.read.len;
    int ovf = 0;
    u32 cap = cs.n + 2;
    Vec<UP> str_ends, sep, gaps;
    str_ends.init((UP *)keep.get((u64)cap * sizeof(UP)), cap, &ovf);
    sep.init((UP *)keep.get((u64)cap * sizeof(UP)), cap, &ovf);
    gaps.init((UP *)keep.get((u64)(cap + 2) * sizeof(UP)), cap + 2, &ovf);
    if (keep.ovf) return 1;
    gather_blocks(cs.p, cs.n, &str_ends, sep, 1, cs.n, L, (u64)cord_gap, (u64)block_size, 0);
    gather_gaps_y(str_ends.p, str_ends.n, gaps, L, (u64)cord_gap, *X.ls);
    for (u32 i = 1; i < cs.n; i++) {
        if (gap_late(X)) return 1;
        u64 slen = X.seq_len[cord_id(cs[i])];
        gp.read_len = L; gp.ref_len = slen;
        if (is_end(cs[i - 1])) {                                                                 // the block's first cord: towards the read's start
            i64 sx = (i64)(slen - 1 - cord_x(cs[i])), sy = (i64)(L - 1 - cord_y(cs[i]));
            if (sx > block_size) sx = block_size;
            if (sy > block_size) sy = block_size;
            u64 gap_end = shift_cord(cs[i], sx, sy);
            if ((i64)cord_y(gap_end) > cord_gap) {
                sx = (i64)umin64(max_extend, cord_x(gap_end)); sy = (i64)umin64(max_extend, cord_y(gap_end));
                if (sx > sy * extend_xy) sx = sy * extend_xy;
                u64 gap_str = shift_cord(gap_end, -sx, -sy);
                gap_str &= ~F_END; gap_end &= ~F_END; remove_tile_sgn(gap_str); remove_tile_sgn(gap_end);
                if (gap_max_gapsy_overlap(gaps.p, gaps.n, gap_str, gap_end) > cord_gap) {
                    gap_map_gap(gap_str, gap_end, tiles_str, tiles_end, -1, X);
                    gap_insert_tiles(cs, ce, i, tiles_str, tiles_end, -1, max_segs);
                }
            }
        } else if (!consecutive(cs[i - 1], cs[i], (u64)cord_gap)) {
            i64 sx = (i64)(slen - 1 - cord_x(cs[i])), sy = (i64)(L - 1 - cord_y(cs[i]));
            if (sx > block_size) sx = block_size;
            if (sy > block_size) sy = block_size;
            if (!is_end(cs[i]) && !cord_strand(cs[i] ^ cs[i + 1])) {
                i64 a = (i64)(cord_x(cs[i + 1]) - cord_x(cs[i])), b = (i64)(cord_y(cs[i + 1]) - cord_y(cs[i]));
                if (a < sx) sx = a;
                if (b < sy) sy = b;
            }
            u64 gap_str = cs[i - 1], gap_end = shift_cord(cs[i], sx, sy);
            if (gabs((i64)(cord_x(gap_end) - cord_x(gap_str))) < max_gap) {
                gap_str &= ~F_END; gap_end &= ~F_END; remove_tile_sgn(gap_str); remove_tile_sgn(gap_end);
                gap_map_gap(gap_str, gap_end, tiles_str, tiles_end, 0, X);
                gap_insert_tiles(cs, ce, i, tiles_str, tiles_end, 0, max_segs);
            }
        }
        if (is_end(cs[i])) {                                                                     // the block's last cord: towards the read's end
            u64 gap_str = cs[i];
            if ((i64)(L - 1 - cord_y(gap_str)) > cord_gap) {
                i64 sx = (i64)umin64(max_extend, slen - cord_x(gap_str) - 1), sy = (i64)umin64(max_extend, L - cord_y(gap_str) - 1);
                if (sx > sy * extend_xy) sx = sy * extend_xy;
                u64 gap_end = shift_cord(gap_str, sx, sy);
                gap_str &= ~F_END; gap_end &= ~F_END; remove_tile_sgn(gap_str); remove_tile_sgn(gap_end);
                if (gap_max_gapsy_overlap(gaps.p, gaps.n, gap_str, gap_end) > cord_gap) {
                    gap_map_gap(gap_str, gap_end, tiles_str, tiles_end, 1, X);
                    gap_insert_tiles(cs, ce, i, tiles_str, tiles_end, 1, max_segs);
                }
            }
        }
        if (X.ar->ovf || keep.ovf) return 1;
    }
    return ovf;
}
// ---- reformCords with reformCordsDxDy1 (cords.cpp:504-687)
LNR_HD inline int gap_scale_dxdy(i64 &dx, i64 &d1, i64 &dy, i64 &d2) {                        // :562-585
    if (dx * dy >= 0 && d1 * d2 >= 0 && dx * d1 >= 0 && (dx || dy || d1 || d2)) {
        i64 c1 = gabs(d1 * dy), c2 = gabs(d2 * dx);
        if (c1 > c2) { if (dx != 0) d2 = d1 * dy / dx; }
        else if (c1 < c2) { if (dy != 0) d1 = d2 * dx / dy; }
        return 0;
    }
    return 1;
}
LNR_HD inline void gap_scale_region(u64 &c_str, u64 &c_end, i64 d11, i64 d12, i64 d21, i64 d22) {   // :591-603
    i64 dx = (i64)(cord_x(c_end) - cord_x(c_str)), dy = (i64)(cord_y(c_end) - cord_y(c_str));
    gap_scale_dxdy(dx, d11, dy, d12);
    gap_scale_dxdy(dx, d21, dy, d22);
    u64 ns = shift_cord(c_str, d11, d12), ne = shift_cord(c_end, d21, d22);
    c_str = ns; c_end = ne;
}
LNR_HD inline void gap_reform_cords(GVec<u64> &cs, GVec<u64> &ce) {
    if (cs.n != ce.n) return;
    const i64 min_dx = -20, min_dy = -20;                                                        // CordsParms cords.h:45-46
    u32 it = 1;
    while (it < cs.n) {
        u32 i1 = it - 1, i2 = it;
        u64 c11 = cs[i1], c12 = ce[i1], c21 = cs[i2], c22 = ce[i2];
        i64 dx1 = (i64)(cord_x(c21) - cord_x(c11)), dy1 = (i64)(cord_y(c21) - cord_y(c11));
        if (cord_x(cs[it]) > cord_x(ce[it]) || cord_y(cs[it]) > cord_y(ce[it])) {
            if (is_end(cs[it])) { cs[it - 1] |= F_END; ce[it - 1] |= F_END; }
            cs.erase(it, it + 1);
            ce.erase(it, it + 1);
        } else if (cord_strand(c11 ^ c22) || is_end(c11)) ++it;
        else if ((dx1 < 0 && dx1 > min_dx) || (dy1 < 0 && dy1 > min_dy)) {
            u64 lower, upper;
            if (i1 == 0 || is_end(cs[i1 - 1])) lower = 0;
            else if (cord_strand(cs[i1] ^ cs[i1 - 1])) lower = cs[i1];
            else lower = cs[i1 - 1];
            if (i2 == cs.n - 1 || is_end(cs[i2])) upper = ce[i2];
            else if (cord_strand(cs[i2] ^ cs[i2 + 1])) upper = ce[i2];
            else upper = cs[i2 + 1];
            i64 sx = (dx1 - 1) / 2 < 0 ? (dx1 - 1) / 2 : 0, sy = (dy1 - 1) / 2 < 0 ? (dy1 - 1) / 2 : 0;
            gap_scale_region(c11, c12, sx, sy, 0, 0);
            gap_scale_region(c21, c22, -sx, -sy, 0, 0);
            u64 x11 = cord_x(c11), y11 = cord_y(c11), x21 = cord_x(c21), y21 = cord_y(c21);
            if (x11 <= cord_x(c12) && x11 > cord_x(lower) && y11 <= cord_y(c12) && y11 > cord_y(lower) && x21 <= cord_x(c22) && x21 < cord_x(upper) && y21 <= cord_y(c22) &&
                y21 < cord_y(upper)) {
                cs[i1] = c11; ce[i1] = c12; cs[i2] = c21; ce[i2] = c22;
            }
            ++it;
        } else ++it;
    }
}

}  // namespace lnr
