// lnr_hd.h -- stage logic of the filter hot path as __host__ __device__ code over raw arrays.
//
// This is PRODUCT code: the HIP kernels in lnr_kernels.hip call these functions on the
// device.  It is written host/device-neutral only so that tests/ can also compile it with
// g++ and check every stage against the oracle on a machine without a GPU
// (tests/host_shim.cpp); the shipped library contains no host execution path for it.
//
// Reference behaviour restated here (xp3i4/linear, paths relative to the reference root):
//   cords.cpp (bit words), shape_extend.cpp:86-348 (gapped minimizer, in closed form, see
//   seed_sample), pmpfinder.cpp:1979-2091 (anchor filter), cluster_util.cpp:53-462 (chain DP
//   + traceback), pmpfinder.cpp:1484-1530,2366-2446 + cluster_util.cpp:469-732 (hit blocks),
//   pmpfinder.cpp:680-722,883-945,1079-1178,1309-1445 (window extension),
//   pmpfinder.cpp:1537-1667 + cluster_util.cpp:774-1102 (cord blocks), pmpfinder.cpp:2709-2804.
#pragma once
#include <stdint.h>
#include <limits.h>
#include "ref_sort.h"

namespace lnr {

typedef uint64_t u64;
typedef int64_t i64;
typedef uint32_t u32;
typedef int32_t i32;
typedef uint16_t u16;
typedef uint8_t u8;

struct UP { u64 first, second; };   // UPair

// ------------------------------------------------------------------ cords ----
static const u64 ANCHOR_ZERO = 1ULL << 20;
static const u64 F_END = 1ULL << 60, F_STRAND = 1ULL << 61, F_RECD = 1ULL << 62, F_MAIN = 1ULL << 63;
static const u64 VALUE_MASK_DSTR = ((1ULL << 60) - 1) | F_STRAND;
static const u64 MAX_CORD_ID = (1ULL << 10) - 1, MAX_CORD_X = (1ULL << 30) - 1;

LNR_HD inline u64 cord_x(u64 v) { return (v >> 20) & ((1ULL << 30) - 1); }
LNR_HD inline u64 cord_y(u64 v) { return v & 0xfffffULL; }
LNR_HD inline u64 cord_strand(u64 v) { return (v >> 61) & 1ULL; }
LNR_HD inline u64 cord_id(u64 v) { return (v >> 50) & 1023ULL; }
LNR_HD inline u64 cord_x40(u64 v) { return (v >> 20) & 0xffffffffffULL; }
LNR_HD inline u64 mk_cord(u64 idx, u64 y, u64 s) { return (idx << 20) + y + (s << 61); }
LNR_HD inline u64 create_cord(u64 id, u64 x, u64 y, u64 s) { return mk_cord((id << 30) + x, y, s); }
LNR_HD inline u64 shift_cord(u64 v, i64 x, i64 y) { return x < 0 ? v - ((u64)(-x) << 20) + (u64)y : v + ((u64)x << 20) + (u64)y; }
LNR_HD inline bool is_end(u64 v) { return (v & F_END) != 0; }
LNR_HD inline u64 hit2cord(u64 a) { return ((a + ((a & 0xfffffULL) << 20) - (ANCHOR_ZERO << 20)) & VALUE_MASK_DSTR) & ~(1ULL << 62); }
LNR_HD inline u64 anchor_x(u64 a) { return cord_x(hit2cord(a)); }
LNR_HD inline i64 labs64(i64 v) { return v < 0 ? -v : v; }
LNR_HD inline i64 max64(i64 a, i64 b) { return a > b ? a : b; }
LNR_HD inline i64 min64(i64 a, i64 b) { return a < b ? a : b; }
LNR_HD inline u64 umin64(u64 a, u64 b) { return a < b ? a : b; }
LNR_HD inline u64 umax64(u64 a, u64 b) { return a > b ? a : b; }
LNR_HD inline int consecutive(u64 c1, u64 c2, u64 thd) {
    u64 x1 = cord_x(c1), x2 = cord_x(c2), y1 = cord_y(c1), y2 = cord_y(c2);
    return !cord_strand(c1 ^ c2) && x1 <= x2 && y1 <= y2 && x2 - x1 < thd && y2 - y1 < thd;
}
LNR_HD inline UP forward_y(UP se, u64 L) {
    UP r;
    if (cord_strand(se.first)) { r.first = L - cord_y(se.second) - 1; r.second = L - cord_y(se.first) - 1; }
    else { r.first = cord_y(se.first); r.second = cord_y(se.second); }
    return r;
}

LNR_HD inline bool lnr_is_leader() {
#if defined(__HIP_DEVICE_COMPILE__)
    return (threadIdx.x & 63) == 0;
#else
    return true;
#endif
}
// make the leader's stores visible to the other lanes of the wave (workgroup = one wave in k_job)
LNR_HD inline void lnr_wave_sync() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
}
// Two ways the device runs the per-read stages: COOP = all 64 lanes of a wave execute the code together for ONE read (the leader
// stores, the others track the counts; the job kernels), or one lane per read (k_post: every lane is its own leader and nothing
// is shared between lanes).  The stage functions that differ take the mode as a template argument.
template <bool COOP> LNR_HD inline bool lnr_leader() { return COOP ? lnr_is_leader() : true; }
template <bool COOP> LNR_HD inline void lnr_sync() { if (COOP) lnr_wave_sync(); }
// A value every lane of the wave holds identically, moved to scalar registers: what is computed from it afterwards runs on the scalar unit
// instead of occupying the vector ALU for all 64 lanes (the SIMT-uniform phases are bound by VALU issue, not by memory).  Host: identity.
LNR_HD inline u32 lnr_uni32(u32 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (u32)__builtin_amdgcn_readfirstlane((int)v);
#else
    return v;
#endif
}
LNR_HD inline u64 lnr_uni64(u64 v) { return ((u64)lnr_uni32((u32)(v >> 32)) << 32) | (u64)lnr_uni32((u32)v); }
template <bool COOP> LNR_HD inline u64 lnr_u64(u64 v) { return COOP ? lnr_uni64(v) : v; }
// bounded vector view over caller-provided storage; overflow is recorded, never written past cap
template <class T>
struct Vec {
    T *p; u32 n, cap; int *ovf;
    LNR_HD void init(T *p_, u32 cap_, int *ovf_) { p = p_; n = 0; cap = cap_; ovf = ovf_; }
    LNR_HD void push(const T &v) { if (n < cap) p[n++] = v; else *ovf = 1; }
    // SIMT-uniform form for code that all lanes of a wave execute together: every lane tracks n, lane 0 stores
    LNR_HD void push_u(const T &v) {
        if (n < cap) { if (lnr_is_leader()) p[n] = v; n++; }
        else if (lnr_is_leader()) *ovf = 1;
    }
    template <bool COOP> LNR_HD void push_m(const T &v) {
        if (n < cap) { if (lnr_leader<COOP>()) p[n] = v; n++; }
        else if (lnr_leader<COOP>()) *ovf = 1;
    }
    LNR_HD T &operator[](u32 i) { return p[i]; }
    LNR_HD T &back() { return p[n - 1]; }
};
// Bump allocator over caller-provided storage.  With `next` set it is two-level: requests that do not
// fit the fast region (LDS in the kernel) fall through to the next one (global scratch).
struct Arena {
    char *base; u64 off, cap; int ovf; Arena *next;
    LNR_HD void init(void *b, u64 c) { base = (char *)b; off = 0; cap = c; ovf = 0; next = nullptr; }
    template <class T> LNR_HD T *get(u64 n) {   // iterative on purpose: no recursion in device code
        u64 bytes = (n * sizeof(T) + 15) & ~15ULL;
        Arena *a = this;
        while (a->off + bytes > a->cap) {
            if (!a->next) { a->ovf = 1; ovf = 1; return (T *)base; }
            a = a->next;
        }
        T *r = (T *)(a->base + a->off);
        a->off += bytes;
        return r;
    }
};
// bytes of per-job scratch needed by job_process for a job with `cap` anchor slots
LNR_HD inline u64 job_scratch_bytes(u64 cap) { return (cap + 2) * 160 + 1024; }
LNR_HD inline u64 tail_scratch_bytes(u64 cap) { return cap * 240 + 8192; }

// --------------------------------------------------------- minimizer shape ----
// Closed form of hashInit/hashNexth/hashNextX (shape_extend.cpp:86-116,173-184,245-348)
// for span 21, weight 13.  `s` points at a zero-padded sequence; the rolling state of
// the reference at position k after starting the roll at k0 (hashInit having run at
// s+init_at with N-skip ks) is reproduced from the bases alone:
//   x_k   = C + 2*sum(s[k..k+20]),  C = -63 + 2*sum(s[init_at+ks .. +19]) - 2*sum(s[k0..k0+19])
//   h_k   = sum VW[p]*4^(20-p) mod 2^42 (N=4 carries), crh_k = sum ((3-VW[p])&3) << 2p
//   VW[p] = s[k+p], except while fewer than 21 bases have been rolled in (n=k-k0+1<21):
//           the first 21-n slots still hold the tail of the hashInit window.
LNR_HD inline int lnr_popc64(u64 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(v);
#else
    return __builtin_popcountll(v);
#endif
}
// A read as the kernels hold it: 2 bits per base + an N bitmap (k_prep).  Indexing returns the Dna5 ordinal; positions at
// or past the end read as 0 ('A'), the value the reference finds in the zeroed slack behind a sequence (pinned UB).
struct PackedSeq {
    const u64 *pk; const u32 *nm; u64 L;
    LNR_HD u8 operator[](u64 i) const {
        u64 j = i < L ? i : L;                       // slack words behind the read are zero (k_prep), index L is always inside
        u32 n = nm[j >> 5]; u64 w = pk[j >> 5];      // two independent loads, no branch
        u32 v = (u32)(w >> (2 * (j & 31))) & 3;
        return (u8)(((n >> (j & 31)) & 1) ? 4 : v);
    }
    // sum of the ordinals of the 20 bases starting at p (p + 20 <= L + 44: inside the slack)
    LNR_HD int sum20(u64 p) const {
        u64 w = p >> 5; u32 sh = (u32)(p & 31);
        u64 lo = pk[w], hi = pk[w + 1];
        u64 bits = sh ? (lo >> (2 * sh)) | (hi << (64 - 2 * sh)) : lo;
        bits &= (1ULL << 40) - 1;
        u64 nb = ((((u64)nm[w + 1] << 32) | nm[w]) >> sh) & ((1ULL << 20) - 1);
        return lnr_popc64(bits & 0x5555555555555555ULL) + 2 * lnr_popc64(bits & 0xAAAAAAAAAAAAAAAAULL) + 4 * lnr_popc64(nb);
    }
};
// The caller's byte buffer seen the same way (ordinals above 4 clamp to N, positions at or past the end read as 0).
struct ByteSeq {
    const u8 *p; u64 L;
    LNR_HD u8 operator[](u64 i) const { if (i >= L) return 0; u8 b = p[i]; return b > 4 ? (u8)4 : b; }
};
struct SeedOut { u32 X; u32 Y; u32 strand; };

// Shape span / weight: 21 / 13 for the DIndex (-i 1, index_util.cpp:2586), 17 / 9 for the HIndex (-i 2, index_util.cpp:2600-2604).
template <int SPAN, class Seq> LNR_HD inline int shape_init_skip_t(const Seq &s) {   // N-skip of hashInit (shape_extend.cpp:95-105)
    u64 k = 0, count = 0;
    while (count < (u64)SPAN) {
        if (s[k + count] == 4) { k += count + 1; count = 0; }
        else count++;
    }
    return (int)k;
}
template <class Seq> LNR_HD inline int shape_init_skip(const Seq &s) { return shape_init_skip_t<21>(s); }
// x (the strand selector) after the roll at k is C + 2 * (sum of the SPAN bases at k): hashInit leaves -3 SPAN + 2 a (a = its SPAN - 1
// bases), every roll adds 2 * (base in - base out), the first one with "base out" = 0
template <int SPAN, class Seq> LNR_HD inline int shape_const_t(const Seq &s, u64 init_at, int ks, u64 k0) {
    int a = 0, b = 0;
    for (int i = 0; i < SPAN - 1; i++) { a += s[init_at + ks + i]; b += s[k0 + i]; }
    return -3 * SPAN + 2 * a - 2 * b;
}
template <class Seq> LNR_HD inline int shape_const(const Seq &s, u64 init_at, int ks, u64 k0) { return shape_const_t<21>(s, init_at, ks, k0); }
LNR_HD inline int shape_const(const PackedSeq &s, u64 init_at, int ks, u64 k0) {   // same value from the packed words
    return -63 + 2 * s.sum20(init_at + (u64)ks) - 2 * s.sum20(k0);
}
// hashNexth + hashNextX at position k (shape_extend.cpp:173-184, 245-348) as a function of the bases: the rolling state after
// n = k - k0 + 1 rolls since hashInit (at init_at, N-skip ks) is the last SPAN bases pushed, the oldest of them hashInit's own.
template <int SPAN, int WEIGHT, class Seq> LNR_HD inline SeedOut seed_sample_t(const Seq &s, u64 k, u64 k0, u64 init_at, int ks, int C) {
    u64 h = 0, crh = 0;
    int W = 0;
    u64 n = k - k0 + 1;
    int stale = n < (u64)SPAN ? (int)(SPAN - n) : 0;
    for (int p = 0; p < SPAN; p++) {
        u64 real = s[k + p];
        W += (int)real;
        u64 v = p < stale ? (u64)s[init_at + ks + n - 1 + p] : real;
        h = (h << 2) + v;
        crh |= ((3 - v) & 3) << (2 * p);
    }
    int x = C + 2 * W;
    u64 v2 = x > 0 ? (h & ((1ULL << (2 * SPAN)) - 1)) : crh;
    u64 X = (1ULL << (2 * SPAN)) - 1, t = 0;
    for (unsigned kk = 64 - 2 * SPAN; kk <= 64 - 2 * WEIGHT; kk += 2) {
        u64 v1 = v2 << kk >> (64 - 2 * WEIGHT);
        if (X > v1) { X = v1; t = kk; }
    }
    u64 Y = 0;
    if (x > 0) {
        i64 d = (i64)(t >> 1) + SPAN + WEIGHT - 32;
        for (i64 i = d; i < d + 4; i++) { u64 val = s[(u64)((i64)k + i)]; Y = val > 3 ? (Y << 2) : (Y << 2) + val; }
    } else {
        i64 d = 31 - WEIGHT - (i64)(t >> 1);
        for (i64 i = d; i > d - 4; i--) { i64 val = 3 - (i64)s[(u64)((i64)k + i)]; Y = val < 0 ? (Y << 2) : (Y << 2) + (u64)val; }
    }
    SeedOut o; o.X = (u32)X; o.Y = (u32)Y; o.strand = x > 0 ? 0 : 1;
    return o;
}
template <class Seq> LNR_HD inline SeedOut seed_sample(const Seq &s, u64 k, u64 k0, u64 init_at, int ks, int C) { return seed_sample_t<21, 13>(s, k, k0, init_at, ks, C); }
// --- the same sample from a 2-bit packed read ---------------------------------------------------------------
// pk: bases packed LSB-first, 32 per u64 (N stored as 0); nm: one bit per base, set for N.  A sample at k needs the
// 29 bases [k-4, k+25): the 21-mer plus the four flanking bases either side that YValue may read
// (shape_extend.cpp:290-327).  With the window w (base p at bits 2p):
//   crh = ~w (complement, first base least significant),  h = w with its 21 base pairs reversed,
//   sum of bases = popc(w & 01..) + 2*popc(w & 10..).
// Returns false when the packed form does not apply (an N inside the span, or the first samples of a job whose
// rolling state still holds hashInit bases): the caller then uses seed_sample on the byte copy.
LNR_HD inline u64 lnr_brev64(u64 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(v);
#else
    v = ((v >> 1) & 0x5555555555555555ULL) | ((v & 0x5555555555555555ULL) << 1);
    v = ((v >> 2) & 0x3333333333333333ULL) | ((v & 0x3333333333333333ULL) << 2);
    v = ((v >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((v & 0x0F0F0F0F0F0F0F0FULL) << 4);
    v = ((v >> 8) & 0x00FF00FF00FF00FFULL) | ((v & 0x00FF00FF00FF00FFULL) << 8);
    v = ((v >> 16) & 0x0000FFFF0000FFFFULL) | ((v & 0x0000FFFF0000FFFFULL) << 16);
    return (v >> 32) | (v << 32);
#endif
}
LNR_HD inline bool seed_sample_packed(const u64 *pk, const u32 *nm, u64 k, u64 k0, int C, SeedOut &o) {
    if (k - k0 + 1 < 21 || k < 4) return false;
    u64 s0 = k - 4;
    u64 w = s0 >> 5;
    u32 sh = (u32)(s0 & 31);
    u64 nbits = (((u64)nm[w + 1] << 32) | nm[w]) >> sh;
    if (nbits & ((1ULL << 29) - 1)) return false;
    u64 lo = pk[w], hi = pk[w + 1];
    u64 span = sh ? (lo >> (2 * sh)) | (hi << (64 - 2 * sh)) : lo;     // 29 bases, base (k-4+q) at bits 2q
    const u64 M42 = (1ULL << 42) - 1;
    u64 win = (span >> 8) & M42;
    int W = lnr_popc64(win & 0x5555555555555555ULL) + 2 * lnr_popc64(win & 0xAAAAAAAAAAAAAAAAULL);
    int x = C + 2 * W;
    u64 v2;
    if (x > 0) {
        u64 r = lnr_brev64(win) >> 22;                                    // pair order reversed, bits inside pairs swapped
        v2 = ((r & 0x2AAAAAAAAAAULL) >> 1) | ((r & 0x15555555555ULL) << 1);
    } else v2 = (~win) & M42;
    // smallest of the nine 26-bit windows v2 << kk >> 38, kk = 22, 24 .. 38 (= bits 38 - kk .. 63 - kk of v2), the first one on a
    // tie: one 32-bit minimum over window << 4 | index (the 64-bit compare-and-select form was a quarter of this function)
    u32 v2lo = (u32)v2, v2hi = (u32)(v2 >> 32);
    u32 best = 0xffffffffu;
#pragma unroll
    for (u32 i = 0; i < 9; i++) {
        u32 p = 16 - 2 * i;                                               // window i = kk 22 + 2 i starts at bit p
        u32 f = p ? ((v2lo >> p) | (v2hi << (32 - p))) : v2lo;
        u32 key = ((f & 0x3ffffffu) << 4) | i;
        best = key < best ? key : best;
    }
    u64 X = best >> 4, t = 22 + 2 * (best & 15u);
    u32 Y;
    if (x > 0) {
        u32 q = (u32)(t >> 1) + 2 + 4;                                   // span index of the first flank base
        u32 b = (u32)(span >> (2 * q)) & 0xFF;
        Y = ((b & 3) << 6) | (((b >> 2) & 3) << 4) | (((b >> 4) & 3) << 2) | ((b >> 6) & 3);
    } else {
        u32 q = (u32)(18 - (int)(t >> 1) - 3 + 4);
        u32 b = (u32)(span >> (2 * q)) & 0xFF;
        Y = (~b) & 0xFF;
    }
    o.X = (u32)X; o.Y = Y; o.strand = x > 0 ? 0 : 1;
    return true;
}
// words of the packed form of a read of L bases (one u64 / u32 of slack so that w+1 is always readable)
LNR_HD inline u64 packed_words(u64 L) { return (L + 63) / 32 + 2; }

LNR_HD inline bool y_match(u64 hs_y, u64 Y) {   // pmpfinder.cpp:1893-1894, ctz(0) pinned to "match"
    u64 v = hs_y ^ Y;
    if (v == 0) return true;
#if defined(__HIP_DEVICE_COMPILE__)
    int tz = __ffsll((unsigned long long)v) - 1;
#else
    int tz = __builtin_ctzll(v);
#endif
    return (v >> tz) < 4;
}
LNR_HD inline u64 val2anchor(u64 e, u64 y, u64 L, u64 shape_strand) {   // index_util.cpp:1509-1520
    if (cord_strand(e) ^ shape_strand) {
        u64 cy = L - 1 - y;
        return (e - (cy << 20) + cy - cord_y(e)) | F_STRAND;
    }
    return (e - (y << 20) + y - cord_y(e)) & ~F_STRAND;
}
// number of samples getDIndexMatchAll takes on [read_str, read_end) with step alpha (pmpfinder.cpp:1874-1880)
LNR_HD inline u32 seed_num_samples(u64 read_str, u64 read_end, u32 alpha) {
    u64 k_first = read_str + 21 + alpha - 1;
    if (read_end < 21 || k_first >= read_end - 21) return 0;
    return (u32)((read_end - 21 - k_first + alpha - 1) / alpha);
}

// Genome chunking of createDIndex (index_util.cpp:1654-1666): sequence of length len split for T layout threads.
LNR_HD inline void chunk_bounds(u64 len, u32 T, u32 t, i64 &t_str, i64 &t_end) {
    i64 b0 = (i64)(len / T * t);
    i64 b1 = t + 1 < T ? (i64)(len / T * (t + 1)) : (i64)len - 21;
    t_str = b0 + 21;
    t_end = b1 - 21;
}
// samples of a chunk sit at j = t_str + 8 + 9m < t_end (count > thd_min_step = 8, index_util.cpp:1677)
LNR_HD inline u64 chunk_num_samples(i64 t_str, i64 t_end) {
    if (t_end - t_str <= 8) return 0;
    return (u64)((t_end - t_str - 8 + 8) / 9);
}
LNR_HD inline u64 genome_feature_count(u64 len) { return len < 48 ? 0 : ((len - 48) >> 4) + 1; }   // pmpfinder.cpp:596

// ---------------------------------------------------------------- features ----
struct F96 { i32 v0, v1, v2, pad; };   // int96 + pad: one 16-byte load per entry
struct FeatView { const F96 *p; u32 n; };

// 2-mer -> (word, bit) table of pmpfinder.cpp:543-548 folded into arithmetic: pairs (a,b) with
// a,b<4 and not TT map to bit 6*((4a+b)%5) of word (4a+b)/5; anything else adds nothing.
LNR_HD inline void add2mer(i32 &w0, i32 &w1, i32 &w2, u32 a, u32 b) {
    if (a > 3 || b > 3) return;
    u32 c = 4 * a + b;
    if (c == 15) return;
    i32 add = 1 << (6 * (c % 5));
    u32 w = c / 5;
    if (w == 0) w0 += add; else if (w == 1) w1 += add; else w2 += add;
}
// 2-mer counts of the 16 two-mers that START in bases [c0, c0+16) of a 2-bit packed sequence (pk LSB-first, nm = N
// bitmap).  A feature entry m is the sum of cells m, m+1, m+2 (48 two-mers, pmpfinder.cpp:556-588).
LNR_HD inline void cell_2mers_packed(const u64 *pk, const u32 *nm, u64 c0, i32 &w0, i32 &w1, i32 &w2) {
    u64 w = c0 >> 5;
    u32 sh = (u32)(c0 & 31);
    u64 lo = pk[w], hi = pk[w + 1];
    u64 span = sh ? (lo >> (2 * sh)) | (hi << (64 - 2 * sh)) : lo;          // base (c0+q) at bits 2q; 17 bases needed
    u64 nb = ((((u64)nm[w + 1] << 32) | nm[w]) >> sh) & 0x1ffffULL;         // N flags of those 17 bases
    w0 = 0; w1 = 0; w2 = 0;
    for (int q = 0; q < 16; q++) {
        if ((nb >> q) & 3) continue;                                         // either base is N: not counted
        u32 nib = (u32)(span >> (2 * q)) & 15;                               // b_q + 4*b_{q+1}
        u32 c = ((nib & 3) << 2) | (nib >> 2);                               // 4*b_q + b_{q+1}
        if (c == 15) continue;                                               // TT is not counted (pmpfinder.cpp:547)
        i32 add = 1 << (6 * (c % 5));
        u32 wi = c / 5;
        if (wi == 0) w0 += add; else if (wi == 1) w1 += add; else w2 += add;
    }
}

LNR_HD inline u32 read_feature_count(u64 L) {   // length of createFeatures2_48's result (pmpfinder.cpp:556-588)
    if (L < 50) return 0;
    return (u32)(1 + (L - 50) / 16);
}
LNR_HD inline i64 script_dist(i32 s1, i32 s2) {   // pmpfinder.cpp:497-506
    const i32 mxu31 = (31 << 24) + (31 << 18) + (31 << 12) + (31 << 6) + 31;
    i32 d = s1 + mxu31 - s2;
    i32 a0 = ((d >> 24) & 63) - 31, a1 = ((d >> 18) & 63) - 31, a2 = ((d >> 12) & 63) - 31, a3 = ((d >> 6) & 63) - 31, a4 = (d & 63) - 31;
    return (i64)((a0 < 0 ? -a0 : a0) + (a1 < 0 ? -a1 : a1) + (a2 < 0 ? -a2 : a2) + (a3 < 0 ? -a3 : a3) + (a4 < 0 ? -a4 : a4));
}
LNR_HD inline i64 window_dist(const F96 *a, const F96 *b) {   // _windowDist2_48 pmpfinder.cpp:523-533
    F96 a0 = a[0], a3 = a[3], b0 = b[0], b3 = b[3];
    return script_dist(a0.v0, b0.v0) + script_dist(a0.v1, b0.v1) + script_dist(a0.v2, b0.v2) +
           script_dist(a3.v0, b3.v0) + script_dist(a3.v1, b3.v1) + script_dist(a3.v2, b3.v2);
}
LNR_HD inline u32 wdist_checked(FeatView f1, FeatView f2, u64 x1, u64 x2) {   // _windowDist pmpfinder.cpp:680-695
    if (x1 + 4 < f1.n && x2 + 4 < f2.n) return (u32)window_dist(f1.p + x1, f2.p + x2);
    return 1000;
}
LNR_HD inline u32 wdist_raw(FeatView f1, FeatView f2, u64 x1, u64 x2) {   // __windowDist pmpfinder.cpp:655-663; out of range pinned to abort score
    if (x1 + 3 < f1.n && x2 + 3 < f2.n) return (u32)window_dist(f1.p + x1, f2.p + x2);
    return 1000;
}

// In-kernel phase stamps: compiled in only for the diagnostic build (-DLNR_PROF); the shipped
// kernels contain no stamp.
#if defined(LNR_PROF) && defined(__HIP_DEVICE_COMPILE__)
// per-job phase cycles of the leader lane (diagnostic build): reported for the job with the most anchors in the DP
static __device__ __attribute__((unused)) unsigned long long lnr_job_ph_dummy;
#define LNR_TICK(prof, idx, last)                                                        \
    do {                                                                                 \
        if (prof) { unsigned long long t_ = clock64(); atomicAdd(&(prof)[idx], t_ - (last)); atomicMax(&(prof)[16 + (idx)], t_ - (last)); lnr_job_ph[idx] += t_ - (last); (last) = t_; } \
    } while (0)
#define LNR_TICK0(prof, idx, last)                                                       \
    do {                                                                                 \
        if (prof) { unsigned long long t_ = clock64(); atomicAdd(&(prof)[idx], t_ - (last)); atomicMax(&(prof)[16 + (idx)], t_ - (last)); (last) = t_; } \
    } while (0)
#else
#define LNR_TICK(prof, idx, last) do { } while (0)
#define LNR_TICK0(prof, idx, last) do { } while (0)
#endif

// -------------------------------------------------------------- parameters ----
struct JobParm { i32 alpha; i32 score_type; };   // PMPParms::toggle (pmpfinder.cpp:21-29,1778-1784,2493-2503)
LNR_HD inline JobParm job_parm(int mode) { JobParm p; p.alpha = mode ? 7 : 15; p.score_type = mode ? 1 : 0; return p; }

// ------------------------------------------------------------ anchor filter ----
// binningFilter pmpfinder.cpp:1979-2012, serial form (the kernel has a wave-parallel twin).
// `bins` holds nbins zeroed u16 counters and is returned zeroed.  Counters saturate at 0xffff
// (only "> 10" is tested).  Quirk kept: if nothing survives, everything is kept (:2007-2010).
LNR_HD inline u32 binning_filter_serial(u64 *a, u32 n, u16 *bins, u32 nbins) {
    for (u32 i = 0; i < n; i++) { u32 b = (u32)(cord_x(a[i]) / 30000); if (b < nbins && bins[b] != 0xffff) bins[b]++; }
    u32 ii = 0;
    for (u32 i = 0; i < n; i++) { u32 b = (u32)(cord_x(a[i]) / 30000); if (b < nbins && bins[b] > 10) a[ii++] = a[i]; }
    for (u32 b = 0; b < nbins; b++) bins[b] = 0;
    return ii ? ii : n;
}
// filterAnchorsList scan (pmpfinder.cpp:2034-2066) over the ascending-sorted anchors with a[0]==0;
// accepted ranges are compacted in place; returns the new length (filterAnchors1 :2073-2091)
LNR_HD inline u32 filter_anchor_list(u64 *a, u32 n) {
    const u64 density = 1, accept_min = 2;
    const unsigned err_bit = 2;
    if (n <= 1) return n;
    u64 ak2 = a[1];
    u64 block_str = 1, count_anchors = 0;
    u64 min_y = ~0ULL, max_y = 0;
    u32 ii = 0;
    for (u32 i = 1; i < n; i++) {
        u64 anc_y = cord_y(a[i]);
        u64 dy2 = (u64)labs64((i64)(anc_y - cord_y(ak2)));
        int f_cont = cord_x40(a[i] - ak2) < (dy2 >> err_bit);
        if (f_cont) {
            if (min_y > anc_y) min_y = anc_y;
            if (max_y < anc_y) max_y = anc_y;
            ak2 = a[(block_str + i) >> 1];
            ++count_anchors;
        }
        if (!f_cont || i == n - 1) {
            u64 thd = umax64(((max_y - min_y) * density >> 10), accept_min);
            if (count_anchors > thd) {
                // ranges are emitted in increasing order and ii <= block_str, so in-place forward copy is safe;
                // ak2 for later blocks is re-read from a[i] (i >= block end), which is not yet overwritten
                for (u64 j = block_str; j < i; j++) a[ii++] = a[j];
            }
            block_str = i;
            ak2 = a[i];
            min_y = anc_y; max_y = anc_y;
            count_anchors = 1;
        }
    }
    return ii;
}

// ------------------------------------------------------------ chain scoring ----
// Pair scores of the chaining DP (getApxChainScore / getApxChainScore0, cluster_util.cpp:337-443) in 32-bit
// arithmetic: x < 2^30 and y < 2^20 (cords.cpp:13-15), so dx, dy and da = |dx - dy| fit an int32.  The reference's
// derr = floor(100*da / max(|dy|,|dx|,50)) is only evaluated where it matters:
//   da >= M  <=>  derr >= 100  (reject, -1000);   da < 10 (score) / da < 30 (score0): the result does not use derr.
// 100*da fits 32 bits when da < 2^25; beyond that the 64-bit quotient is taken (M > da >= 2^25 only for |dx| that large).
LNR_HD inline i32 chain_derr32(i32 da, i32 M) {
    if (da < (1 << 25)) return (i32)((u32)(100u * (u32)da) / (u32)M);
    return (i32)((100LL * (i64)da) / (i64)M);
}
LNR_HD inline int chain_score0(u32 x1, u32 y1, u32 x2, u32 y2) {   // getApxChainScore0 cluster_util.cpp:337-385 (effective values)
    i32 dy = (i32)y1 - (i32)y2;
    if (dy < 5) return -10000;
    i32 dx = (i32)x1 - (i32)x2;
    i32 t = dx - dy;
    i32 da = t < 0 ? -t : t;
    i32 adx = dx < 0 ? -dx : dx;
    i32 M = dy > adx ? dy : adx;
    if (M < 50) M = 50;
    if (da >= M) return -1000;          // derr >= 100
    if (da < 30) return 100 - dy;
    return 100 - dy - da;
}
LNR_HD inline int chain_score(u32 x1, u32 y1, u32 x2, u32 y2) {   // getApxChainScore cluster_util.cpp:387-443
    i32 dy = (i32)y1 - (i32)y2;
    if (dy < 10) return -10000;
    i32 dx = (i32)x1 - (i32)x2;
    i32 t = dx - dy;
    i32 da = t < 0 ? -t : t;
    i32 adx = dx < 0 ? -dx : dx;
    i32 M = dy > adx ? dy : adx;
    if (M < 50) M = 50;
    if (da >= M) return -1000;          // derr >= 100
    i32 dq = dy / 15, score_dy;
    if (dq < 150) score_dy = dq / 5;
    else if (dq < 10000) score_dy = dq * dq / 200 + 20;   // dq < 2^20/15: dq*dq < 2^33 cannot happen here (dq < 10000 -> < 1e8)
    else score_dy = 10000;
    if (da < 10) return 100 - score_dy;
    i32 derr = chain_derr32(da, M);
    i32 score_derr;
    if (derr < 5) score_derr = 4 * derr;
    else if (derr < 10) score_derr = 6 * derr - 10;
    else score_derr = derr * derr - 5 * derr;
    return 100 - score_dy - score_derr;
}

// Branch-free twins for the lane-parallel DP (every lane scores a different anchor, so a branch would almost never be
// skipped by the whole wave).  Same values as chain_score / chain_score0 for every input with x < 2^30, y < 2^20
// (tests/test_stage_logic_host.py fuzzes the pair).  derr = floor(100 * da / M) comes from a float estimate that is
// within +-1 of the quotient (da < M < 2^30, quotient < 100) and one exact integer correction step.
LNR_HD inline int chain_score0_bl(u32 x1, u32 y1, u32 x2, u32 y2) {
    i32 dy = (i32)y1 - (i32)y2;
    i32 dx = (i32)x1 - (i32)x2;
    i32 t = dx - dy;
    i32 da = t < 0 ? -t : t;
    i32 adx = dx < 0 ? -dx : dx;
    i32 M = dy > adx ? dy : adx;
    M = M < 50 ? 50 : M;
    i32 res = 100 - dy - (da < 30 ? 0 : da);
    res = da >= M ? -1000 : res;
    res = dy < 5 ? -10000 : res;
    return res;
}
LNR_HD inline int chain_score_bl(u32 x1, u32 y1, u32 x2, u32 y2) {
    i32 dy = (i32)y1 - (i32)y2;
    i32 dx = (i32)x1 - (i32)x2;
    i32 t = dx - dy;
    u32 da = (u32)(t < 0 ? -t : t);
    u32 adx = (u32)(dx < 0 ? -dx : dx);
    u32 udy = (u32)(dy < 0 ? 0 : dy);
    u32 M = udy > adx ? udy : adx;
    M = M < 50 ? 50 : M;
    u32 dq = udy / 15;
    u32 dqc = dq < 9999 ? dq : 9999;
    u32 s_hi = dqc * dqc / 200 + 20;
    u32 score_dy = dq < 150 ? dq / 5 : (dq < 10000 ? s_hi : 10000);
    u32 dac = da < M ? da : 0;
#if defined(__HIP_DEVICE_COMPILE__)
    float inv = __builtin_amdgcn_rcpf((float)M);
#else
    float inv = 1.0f / (float)M;
#endif
    u32 q = (u32)((float)dac * 100.0f * inv);
    i32 r = (i32)(100u * dac - q * M);        // exact modulo 2^32; the true remainder lies in [-M, 2M)
    q = r < 0 ? q - 1 : ((u32)r >= M ? q + 1 : q);
    u32 score_derr = q < 5 ? 4 * q : (q < 10 ? 6 * q - 10 : q * q - 5 * q);
    score_derr = da < 10 ? 0 : score_derr;
    i32 res = 100 - (i32)score_dy - (i32)score_derr;
    res = da >= M ? -1000 : res;
    res = dy < 10 ? -10000 : res;
    return res;
}

// What the lane-parallel DP evaluates per pair, in two stages (lnr_kernels.hip dp_eval; fuzzed against the literal scores in
// tests/test_stage_logic_host.py: wherever the literal score is positive the pair is a candidate with the same score, and no
// non-positive pair is ever reported positive):
//   dp_pair_cand : a cheap NECESSARY condition for a positive score (the rest is skipped when no lane of a wave passes)
//   dp_pair_score: the score under that condition (select-style arithmetic, no branch)
// getApxChainScore : positive needs score_dy < 100, i.e. dy / 15 < 150 (beyond that score_dy >= 132), and then
//                    score_dy = (dy / 15) / 5 = dy / 75;  da >= 10 and 7 da >= M  =>  derr >= 14  =>  score_derr >= 126
// getApxChainScore0: score = 100 - dy - (da < 30 ? 0 : da) > 0 needs 5 <= dy < 100 and da < 100 (da >= M gives -1000)
struct DpPair { i32 dy; u32 da, M; };
// SORTED = the caller guarantees px >= xi (the anchor DP: predecessors come earlier in the x-descending order), which
// spares the |dx| and makes M one three-way maximum.
template <int ST, bool SORTED = false>
LNR_HD inline bool dp_pair_cand(u32 px, u32 py, u32 xi, u32 yi, DpPair &p) {
    i32 dy = (i32)py - (i32)yi, dx = (i32)px - (i32)xi;
    u32 da, M;
    i32 m3;                                              // max(dy, dx) when dx >= 0
    if (SORTED) {
        // dx >= 0; every use of da and M below sits behind a test that dy is positive, and for non-negative dx, dy the
        // distance from the diagonal is one unsigned absolute difference (v_sad_u32); a negative dy never wins the maximum
        u32 udx = (u32)dx, udy = (u32)dy;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("v_sad_u32 %0, %1, %2, 0" : "=v"(da) : "v"(udx), "v"(udy));   // the compiler emits min / max / sub for the select form here
#else
        da = udx > udy ? udx - udy : udy - udx;
#endif
        m3 = dy > dx ? dy : dx;
        M = (u32)(m3 < 50 ? 50 : m3);
    } else {
        i32 t = dx - dy;
        da = (u32)(t < 0 ? -t : t);
        u32 adx = (u32)(dx < 0 ? -dx : dx);
        M = (u32)(dy < 0 ? 0 : dy); M = M > adx ? M : adx; M = M < 50 ? 50 : M;
        m3 = (i32)M;
    }
    p.dy = dy; p.da = da; p.M = M;
    if (ST) return dy >= 5 && dy < 100 && da < 100 && da < M;
    // dy in [10, 2250) && da < M && (da < 10 || 7 da < M), without the 64-bit product and without a branch.  7 da < M forces
    // da < 375 -- M = max(dy, 50) < 2250 gives da < 322, M = |dx| > dy needs dx > 0 (else da = dy + |dx| > M) and
    // 7 (dx - dy) < dx, i.e. dx < 2625 -- so da >= 512 is rejected outright and 7 da only has to be right for da < 512 (a
    // 24-bit multiply on the GPU).  The case da < 10 needs no test of its own against max(M, 70): 7 da <= 63 < 70 then, and for
    // da >= 10 (7 da >= 70) the larger bound changes nothing when M < 70.
    u32 M70 = (u32)(m3 < 70 ? 70 : m3);
#if defined(__HIP_DEVICE_COMPILE__)
    u32 e7 = (u32)__umul24(da, 7u);
#else
    u32 e7 = (da & 0xffffffu) * 7u;
#endif
    return (u32)(dy - 10) < 2240u && da < 512u && e7 < M70;
}
// score_derr of getApxChainScore for a candidate pair (0 when da < 10): floor(100 da / M) by float estimate + exact correction
LNR_HD inline u32 dp_pair_sderr(const DpPair &p) {
    u32 da = p.da, M = p.M;
#if defined(__HIP_DEVICE_COMPILE__)
    float inv = __builtin_amdgcn_rcpf((float)M);
#else
    float inv = 1.0f / (float)M;
#endif
    u32 dac = da < M ? da : 0;
    u32 q = (u32)((float)dac * 100.0f * inv);
    i32 r = (i32)(100u * dac - q * M);              // exact modulo 2^32; the true remainder lies in [-M, 2M)
    u32 qm = q - 1, qp = q + 1;
    q = r < 0 ? qm : q;
    q = (r >= 0 && (u32)r >= M) ? qp : q;
    u32 e1 = 4 * q, e2 = 6 * q - 10, e3 = q * q - 5 * q;
    u32 sd = e3;
    sd = q < 10 ? e2 : sd;
    sd = q < 5 ? e1 : sd;
    return da < 10 ? 0 : sd;
}
template <int ST>
LNR_HD inline i32 dp_pair_score(const DpPair &p) {
    if (ST) return 100 - p.dy - (p.da < 30 ? 0 : (i32)p.da);
    return 100 - (i32)((u32)p.dy / 75u) - (i32)dp_pair_sderr(p);
}

struct Rec { i32 *score, *score2, *len, *p2, *root, *leaf; };   // ChainsRecord as SoA

// Small arrays that only the leader lane touches (introsort stack, tree table of traceBackChains1).  In the
// kernel one instance lives in LDS per wave; keeping them out of per-lane private memory is worth a factor in
// occupancy.
struct LeaderScratch {
    SortStack st;
    i32 l_root[64], l_score[64], l_len[64], l_leaf[64];
    u64 ranks[64];
};

// getBestChains cluster_util.cpp:53-111, serial form.  xs/ys = getAnchorX / y of the x-descending anchors.
LNR_HD inline void best_chains_serial(const u32 *xs, const u32 *ys, u32 n, Rec r, int score_type, u64 *pair_evals) {
    u32 p300 = 0;   // smallest j with xs[j]-xs[i] < 300 (non-decreasing in i)
    for (u32 i = 0; i < n; i++) {
        int j_str = (int)i - 20 < 0 ? 0 : (int)i - 20;
        while (p300 < i && xs[p300] - xs[i] >= 300) p300++;
        int j_lo = (int)p300 < j_str ? (int)p300 : j_str;
        int max_j = (int)i, best = -1;
        for (int j = (int)i - 1; j >= j_lo; j--) {
            int sc = score_type ? chain_score0(xs[j], ys[j], xs[i], ys[i]) : chain_score(xs[j], ys[j], xs[i], ys[i]);
            if (sc > 0 && sc + r.score[j] >= best) { max_j = j; best = sc + r.score[j]; }
        }
        if (pair_evals) *pair_evals += (u64)((int)i - j_lo);
        if (best > 0) {
            r.p2[i] = max_j; r.score[i] = best; r.len[i] = r.len[max_j] + 1; r.score2[i] = best;
            r.root[i] = r.root[max_j]; r.leaf[i] = 1; r.leaf[max_j] = 0;
        } else {
            r.p2[i] = -1; r.score[i] = 0; r.len[i] = 1; r.score2[i] = 0; r.root[i] = (i32)i; r.leaf[i] = 1;
        }
    }
}

// Chain sinks: traceback hands over finished chains element by element.
struct AnchorSink {      // chainAnchorsHits pmpfinder.cpp:2472-2479: chains -> hits (+ scores), block end after each
    const u64 *anchors; Vec<u64> *hits; Vec<i32> *hscore; u32 first_len, nchains;
    LNR_HD void emit(const i32 *idx, const i32 *sc, u32 n) {
        for (u32 k = 0; k < n; k++) { hits->push(hit2cord(anchors[idx[k]])); hscore->push(sc[k]); }
        hits->back() |= F_END;
        if (nchains == 0) first_len = n;
        nchains++;
    }
};
struct BlockSink {       // chains of blocks (StringSet<String<UPair>>) flattened: el[off[c]..off[c+1])
    const UP *elements; UP *el; i32 *off; u32 nchains, nel, cap; int *ovf; u32 first_len;
    LNR_HD void emit(const i32 *idx, const i32 *sc, u32 n) {
        (void)sc;
        if (nel + n > cap) { *ovf = 1; return; }
        for (u32 k = 0; k < n; k++) el[nel + k] = elements[idx[k]];
        nel += n;
        if (nchains == 0) first_len = n;
        nchains++;
        off[nchains] = (i32)nel;
    }
};

// traceBackChains0 cluster_util.cpp:121-205.  One iteration = (a) scan for the first maximal score and the
// running maximum seen before it, (b) everything else.  The scan is separate so that the kernel can run it
// with all lanes (tb0_scan_serial here, tb0_scan_wave there).
struct Tb0Scan { int max_score, max_2nd, max_str, max_len; };
LNR_HD inline Tb0Scan tb0_scan_serial(const Rec &r, u32 n) {
    Tb0Scan s; s.max_score = -1; s.max_2nd = -1; s.max_str = -1; s.max_len = 0;
    for (u32 j = 0; j < n; j++)
        if (r.score[j] > s.max_score) { s.max_2nd = s.max_score; s.max_str = (int)j; s.max_score = r.score[j]; s.max_len = r.len[j]; }
    return s;
}
// returns false when the search is over
template <class Sink>
LNR_HD inline bool tb0_step(Rec r, const Tb0Scan &sc, Sink &sink, i32 *chain, i32 *chain_sc, int min_len, int abort_score, float stop_ratio) {
    const int delete_score = -1000;
    bool f_done = sc.max_str == -1;
    int max_2nd = sc.max_2nd, max_score = sc.max_score, max_str = sc.max_str, max_len = sc.max_len;
    if (sink.nchains) { if ((float)max_len > (float)sink.first_len * stop_ratio) f_done = false; }
    if (f_done || max_score == 0) return false;
    if (max_len > min_len && max_score / (max_len - 1) > abort_score) {
        u32 cn = 0;
        for (int j = max_str; j != -1; j = r.p2[j]) {
            if (r.score[j] != delete_score) { chain[cn] = j; chain_sc[cn] = r.score2[j]; cn++; r.score[j] = delete_score; }
            else {
                int infix = r.score2[j];
                if (max_score - infix < max_2nd) {
                    for (int k = max_str; k != j; k = r.p2[k]) r.score[k] = r.score2[k] - infix;
                    cn = 0;
                }
                break;
            }
        }
        if (cn) sink.emit(chain, chain_sc, cn);
    }
    if (max_str != -1) r.score[max_str] = delete_score;
    return true;
}
template <class Sink>
LNR_HD inline void traceback0(Rec r, u32 n, Sink &sink, i32 *chain, i32 *chain_sc, int min_len, int abort_score, int bestn, float stop_ratio) {
    int search_times = bestn < 50 ? bestn : 50;
    for (int it = 0; it < search_times; it++) {
        Tb0Scan sc = tb0_scan_serial(r, n);
        if (!tb0_step(r, sc, sink, chain, chain_sc, min_len, abort_score, stop_ratio)) break;
    }
}
// traceBackChains1 cluster_util.cpp:213-304 (at most 50 trees reach this function), in two halves:
//   table: per tree (in order of its first leaf) the best leaf -- highest score, earliest on ties;
//   emit : trees by score (std::sort order), chains walked from the leaf.
LNR_HD inline int traceback1_table(const Rec &r, u32 n, LeaderScratch &ls) {
    i32 *l_root = ls.l_root, *l_score = ls.l_score, *l_len = ls.l_len, *l_leaf = ls.l_leaf;
    int nl = 0;
    for (u32 j = 0; j < n; j++) {
        if (r.leaf[j]) {
            int f_new = 1;
            for (int k = 0; k < nl; k++) {
                if (l_root[k] == r.root[j]) {
                    if (r.score[j] > l_score[k]) { l_score[k] = r.score[j]; l_len[k] = r.len[j]; l_leaf[k] = (i32)j; }
                    f_new = 0;
                }
            }
            if (f_new && nl < 64) { l_root[nl] = r.root[j]; l_score[nl] = r.score[j]; l_len[nl] = r.len[j]; l_leaf[nl] = (i32)j; nl++; }
        }
    }
    return nl;
}
template <class Sink>
LNR_HD inline void traceback1_emit(Rec r, int nl, Sink &sink, i32 *chain, i32 *chain_sc, int min_len, int abort_score, int bestn, float stop_ratio, LeaderScratch &ls) {
    i32 *l_score = ls.l_score, *l_len = ls.l_len, *l_leaf = ls.l_leaf;
    u64 *ranks = ls.ranks;   // (tree index, score) pairs; std::sort by score desc with ties (cluster_util.cpp:269)
    for (int i = 0; i < nl; i++) ranks[i] = ((u64)(u32)l_score[i] << 32) | (u32)i;
    ref_sort(ranks, (long)nl, [](const u64 &a, const u64 &b) { return (i32)(a >> 32) > (i32)(b >> 32); }, ls.st);
    int lim = bestn < nl ? bestn : nl;
    for (int i = 0; i < lim; i++) {
        int t = (int)(u32)ranks[i];
        int max_score = l_score[t], max_len = l_len[t], max_str = l_leaf[t];
        int mean = max_len > 1 ? max_score / (max_len - 1) : abort_score + 1;
        if (max_len > min_len && mean > abort_score) {
            u32 cn = 0;
            for (int j = max_str; j != -1; j = r.p2[j]) { chain[cn] = j; chain_sc[cn] = r.score2[j]; cn++; }
            if (cn) {
                if (sink.nchains && (float)cn / (float)sink.first_len < stop_ratio) break;   // f_stop: nothing is emitted afterwards
                sink.emit(chain, chain_sc, cn);
            }
        }
    }
}
template <class Sink>
LNR_HD inline void traceback1(Rec r, u32 n, Sink &sink, i32 *chain, i32 *chain_sc, int min_len, int abort_score, int bestn, float stop_ratio, LeaderScratch &ls) {
    int nl = traceback1_table(r, n, ls);
    traceback1_emit(r, nl, sink, chain, chain_sc, min_len, abort_score, bestn, stop_ratio, ls);
}
// traceBackChains cluster_util.cpp:306-335; cnt = n zeroed ints of scratch
template <class Sink>
LNR_HD inline void traceback(Rec r, u32 n, Sink &sink, i32 *chain, i32 *chain_sc, i32 *cnt, int min_len, int abort_score, int bestn, float stop_ratio, LeaderScratch &ls) {
    u32 root_num = 0;
    for (u32 i = 0; i < n; i++) cnt[i] = 0;
    for (u32 i = 0; i < n; i++) { if (cnt[r.root[i]] == 0) root_num++; cnt[r.root[i]] = 1; }
    if (root_num > 50) traceback0(r, n, sink, chain, chain_sc, min_len, abort_score, bestn, stop_ratio);
    else traceback1(r, n, sink, chain, chain_sc, min_len, abort_score, bestn, stop_ratio, ls);
}

// ------------------------------------------------------------------ blocks ----
// gather_blocks_ pmpfinder.cpp:1484-1530.  str_ends may be null (not needed by the caller).
LNR_HD inline void gather_blocks(u64 *cords, u32 ncords, Vec<UP> *str_ends, Vec<UP> &sep, u64 str_, u64 end_, u64 L, u64 large_gap, u64 cord_size, int f_set_end) {
    if (str_ends) str_ends->n = 0;
    if (ncords < 2) return;
    u64 dmax = cord_size / 2, d;
    u32 p_str = (u32)str_;
    for (u32 i = (u32)str_ + 1; i < end_; i++) {
        if (is_end(cords[i - 1]) || !consecutive(cords[i - 1], cords[i], large_gap)) {
            d = umin64(L - cord_y(cords[p_str]) - 1, dmax);
            u64 b_str = shift_cord(cords[p_str], (i64)d, (i64)d);
            d = umin64(L - cord_y(cords[i - 1]) - 1, dmax);
            u64 b_end = shift_cord(cords[i - 1], (i64)d, (i64)d);
            if (str_ends) { UP u; u.first = b_str; u.second = b_end; str_ends->push(u); }
            UP q; q.first = p_str; q.second = i; sep.push(q);
            if (f_set_end) cords[i - 1] |= F_END;
            p_str = i;
        }
    }
    d = umin64(L - cord_y(cords[ncords - 1]) - 1, dmax);
    u64 b_str = shift_cord(cords[p_str], (i64)d, (i64)d);
    u64 b_end = shift_cord(cords[ncords - 1], (i64)d, (i64)d);
    if (str_ends) { UP u; u.first = b_str; u.second = b_end; str_ends->push(u); }
    UP q; q.first = p_str; q.second = ncords; sep.push(q);
}

// preFilterChains2 pmpfinder.cpp:2366-2446 with getCordXY = get_cord_y.  sep is replaced.
LNR_HD inline void prefilter_chains2(u64 *hits, u32 nhits, Vec<UP> &sep, u64 *cuts, u64 *xy_strs, Vec<UP> &tmp, LeaderScratch &ls) {
    const u64 mask = 1ULL << 62;
    u32 nb = sep.n;
    for (u32 i = 0; i < nb; i++) { cuts[2 * i] = sep[i].first; cuts[2 * i + 1] = (sep[i].second - 1) | mask; xy_strs[i] = sep[i].first; }
    const u64 *h = hits;
    ref_sort(cuts, (long)(2 * nb), [h, mask](const u64 &a, const u64 &b) { return cord_y(h[a & ~mask]) < cord_y(h[b & ~mask]); }, ls.st);
    tmp.n = 0;
    for (u32 i = 0; i < 2 * nb; i++) {
        u64 cuty = cord_y(hits[cuts[i] & ~mask]);
        for (u32 j = 0; j < nb && xy_strs[j] < nhits; j++) {
            if (cuty < cord_y(hits[xy_strs[j]])) continue;
            for (u64 k = xy_strs[j]; k < sep[j].second; k++) {
                u64 ky = cord_y(hits[k]);
                u64 upper;
                bool cut;
                if (cuts[i] & mask) {
                    if (ky == cuty) { upper = k + 1; cut = true; }
                    else if (ky > cuty) { upper = k; cut = true; }
                    else cut = false;
                } else {
                    if (ky >= cuty) { upper = k; cut = true; }
                    else cut = false;
                }
                if (cut) {
                    u64 lower = xy_strs[j];
                    if (lower != upper) { UP u; u.first = lower; u.second = upper; tmp.push(u); xy_strs[j] = upper; }
                    break;
                }
            }
        }
    }
    sep.n = 0;
    for (u32 i = 0; i < tmp.n; i++) sep.push(tmp[i]);
    ref_sort(sep.p, (long)sep.n, [](const UP &a, const UP &b) { return a.second < b.second; }, ls.st);
    for (u32 i = 0; i < sep.n; i++) hits[sep[i].second - 1] |= F_END;
}

LNR_HD inline int block_score2(u64 c11, u64 c22) {   // getApxChainScore2 cluster_util.cpp:586-631
    // 32-bit throughout: y has 20 bits, x 30, and past the range test dx, dy lie in [0, 20000], so every quotient of the reference's
    // 64-bit arithmetic is a small unsigned one; floor(100 da / D) comes from a float estimate with an exact correction (as dp_pair_sderr)
    i32 dy = (i32)((u32)c11 & 0xfffffu) - (i32)((u32)c22 & 0xfffffu);
    i32 dx = (i32)((u32)(c11 >> 20) & 0x3fffffffu) - (i32)((u32)(c22 >> 20) & 0x3fffffffu);
    if (dx < 0 || dy < 0 || cord_strand(c11 ^ c22) || dx > 20000 || dy > 20000) return INT_MIN;
    u32 ux = (u32)dx, uy = (u32)dy;
    u32 da = ux > uy ? ux - uy : uy - ux;
    u32 D = uy > 100u ? uy : 100u; D = D > ux ? D : ux;
#if defined(__HIP_DEVICE_COMPILE__)
    float inv = __builtin_amdgcn_rcpf((float)D);
#else
    float inv = 1.0f / (float)D;
#endif
    u32 n = 100u * da;                               // <= 2 000 000: exact in a float
    u32 q = (u32)((float)n * inv);
    i32 r = (i32)(n - q * D);                        // the true remainder lies in [-D, 2D)
    q = r < 0 ? q - 1 : ((u32)r >= D ? q + 1 : q);
    if (da > 100u || q > 50u) {
        if (ux < uy) return (int)(100 - 30) - (int)(uy / 1000u) - (int)(ux / 100u);
        return (int)(100 - 30) - (int)(uy / 100u) - (int)(ux / 1000u);
    }
    return 100 - (int)(uy / 95u);
}
LNR_HD inline int block_score3(u64 c11, u64 c12, u64 c21, u64 c22, u64 L, int strand) {   // getApxChainScore3 + getChainBlockDxDy cluster_util.cpp:774-863
    i64 dx, dy;
    if (cord_strand(c11) != (u64)strand) {
        if (cord_strand(c22) != (u64)strand) { dy = (i64)(cord_y(c21) - cord_y(c12)); dx = (i64)(cord_x(c21) - cord_x(c12)); }
        else { dy = (i64)(L - cord_y(c12) - 1 - cord_y(c22)); dx = (i64)(cord_x(c11) - cord_x(c22)); }
    } else {
        if (cord_strand(c22) != (u64)strand) { dy = (i64)(cord_y(c11) - L + 1 + cord_y(c21)); dx = (i64)(cord_x(c11) - cord_x(c22)); }
        else { dy = (i64)(cord_y(c11) - cord_y(c22)); dx = (i64)(cord_x(c11) - cord_x(c22)); }
    }
    int f_type = (int)cord_strand(c11 ^ c22);
    i64 thd_min_dx = -(i64)L;
    i64 thd_max_dy = (i64)((float)L * 1.0f);
    i64 dx_ = labs64(dx), dy_ = labs64(dy), da = dx - dy;
    int score = 0;
    if (dy < -80 || dy > thd_max_dy || dx < thd_min_dx || dx_ > 15000) score = INT_MIN;
    else {
        i64 sdy = dy_ > 2000 ? min64(dy_ / 25 - 50, 70) : dy_ / 40;
        i64 sdx = dx_ > 2000 ? min64(dx_ / 25 - 50, 70) : dx_ / 40;
        if (f_type == 1) { if (dx > thd_min_dx) score = (int)(75 - sdy); }
        else if (da < -max64(dx_ / 4, 50)) { if (dx > -50) score = (int)(80 - sdx); else score = (int)(80 - sdy); }
        else if (da > max64(dy / 4, 50)) score = (int)(80 - sdy);
        else score = (int)(100 - sdy);
    }
    return score;
}
// getBestChains2 cluster_util.cpp:469-526.  which: 2 -> getApxChainScore2, 3 -> getApxChainScore3(strand)
LNR_HD inline void best_chains2(const u64 *hits, const UP *sep, const i32 *sep_score, u32 nb, Rec r, u64 L, int which, int strand) {
    for (u32 i = 0; i < nb; i++) {
        int j_str = (int)i - 20 < 0 ? 0 : (int)i - 20;
        int max_j = (int)i, best = -1;
        for (u32 j = (u32)j_str; j < i; j++) {
            int sc = which == 2 ? block_score2(hits[sep[j].first], hits[sep[i].second - 1])
                                : block_score3(hits[sep[j].first], hits[sep[j].second - 1], hits[sep[i].first], hits[sep[i].second - 1], L, strand);
            if (sc > 0 && sc + r.score[j] + sep_score[i] >= best) { max_j = (int)j; best = sc + r.score[j] + sep_score[i]; }
        }
        if (best > 0) {
            r.p2[i] = max_j; r.score[i] = best; r.len[i] = (i32)(sep[i].second - sep[i].first) + r.len[max_j]; r.score2[i] = best;
            r.root[i] = r.root[max_j]; r.leaf[i] = 1; r.leaf[max_j] = 0;
        } else {
            r.p2[i] = -1; r.score[i] = sep_score[i]; r.len[i] = (i32)(sep[i].second - sep[i].first); r.score2[i] = r.score[i];
            r.root[i] = (i32)i; r.leaf[i] = 1;
        }
    }
}
struct BlockScratch { u32 *ptr; UP *sep_tmp; i32 *score_tmp; Rec rec; i32 *chain, *chain_sc, *cnt; LeaderScratch *ls; };
// chainBlocksBase cluster_util.cpp:533-577, in three steps so that the kernel can run the middle one with all lanes:
//   prepare (tie-sensitive sort of the blocks by first-x + gather), DP (getBestChains2), traceback.
LNR_HD inline void chain_blocks_prepare(const u64 *records, const UP *sep, const i32 *sep_score, u32 nb, int f_sort, BlockScratch s) {
    if (f_sort) {
        // The reference sorts block indices with a comparator that dereferences twice (first-x of the block's first record).
        // The same comparisons on a packed key (first-x << 24 | index; sep_tmp doubles as the key array) give the same
        // permutation -- the sort only sees comparator results -- without a chain of dependent loads per comparison.
        u64 *e = (u64 *)s.sep_tmp;
        for (u32 i = 0; i < nb; i++) e[i] = (cord_x40(records[sep[i].first]) << 24) | (u64)i;
        ref_sort(e, (long)nb, [](const u64 &a, const u64 &b) { return (a >> 24) > (b >> 24); }, s.ls->st);
        for (u32 i = 0; i < nb; i++) s.ptr[i] = (u32)(e[i] & 0xffffffu);
    } else {
        for (u32 i = 0; i < nb; i++) s.ptr[i] = i;
    }
    for (u32 i = 0; i < nb; i++) { s.sep_tmp[i] = sep[s.ptr[i]]; s.score_tmp[i] = sep_score[s.ptr[i]]; }
}
LNR_HD inline void chain_blocks_trace(BlockSink &sink, u32 nb, BlockScratch s) {
    sink.elements = s.sep_tmp;
    traceback(s.rec, nb, sink, s.chain, s.chain_sc, s.cnt, 1, 0, 3, 0.7f, *s.ls);
}
LNR_HD inline void chain_blocks_base(BlockSink &sink, const u64 *records, const UP *sep, const i32 *sep_score, u32 nb, u64 L, int which, int strand,
                                     int f_sort, BlockScratch s) {
    if (nb < 2) return;
    chain_blocks_prepare(records, sep, sep_score, nb, f_sort, s);
    best_chains2(records, s.sep_tmp, s.score_tmp, nb, s.rec, L, which, strand);
    chain_blocks_trace(sink, nb, s);
}
// _filterBlocksHits cluster_util.cpp:633-719: rewrites hits from the chained blocks (note: the dummy hits[0] is dropped)
LNR_HD inline u32 filter_blocks_hits(const BlockSink &ch, const u64 *hits, u64 *out) {
    if (ch.nchains == 0) return 0xffffffffu;   // untouched
    u32 n = 0;
    u64 len_current = 0;
    for (i32 i = ch.off[0]; i < ch.off[1]; i++) {
        for (u64 j = ch.el[i].first; j < ch.el[i].second; j++) out[n++] = hits[j] & ~F_END;
        len_current += ch.el[i].second - ch.el[i].first;
    }
    out[n - 1] |= F_END;
    float bound = 0.8 * len_current;
    u32 major_n = 1;
    for (u32 c = 1; c < ch.nchains; c++) {
        len_current = 0;
        for (i32 j = ch.off[c]; j < ch.off[c + 1]; j++) len_current += ch.el[j].second - ch.el[j].first;
        bool f_append = false;
        if (major_n < 5 && (float)len_current > bound) { f_append = true; ++major_n; }
        // the reference's third branch needs a chain of zero hits and is unreachable
        if (f_append) {
            for (i32 j = ch.off[c]; j < ch.off[c + 1]; j++)
                for (u64 k = ch.el[j].first; k < ch.el[j].second; k++) out[n++] = hits[k] & ~F_END;
            out[n - 1] |= F_END;
        }
        out[n - 1] |= F_END;
    }
    return n;
}

// ----------------------------------------------------------------- windows ----
// best of the three candidate x-cells [x0, x0+3) against read cell y: minimal distance, first minimum wins.
// Device: called by all 64 lanes with identical arguments; lanes 0..2 evaluate one candidate each, the result is
// made uniform with shuffles (one memory round trip instead of twelve dependent ones).  Host: plain loop.
template <bool COOP = true>
LNR_HD inline u32 window_best3(FeatView f1, FeatView f2, u64 y, u64 x0, u64 &x_min) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (!COOP) {
        u32 mn1 = ~0u;
        for (u64 x = x0; x < x0 + 3; x++) { u32 t = wdist_raw(f1, f2, y, x); if (t < mn1) { mn1 = t; x_min = x; } }
        return mn1;
    }
    int lane = (int)(threadIdx.x & 63);
    u32 t = lane < 3 ? wdist_raw(f1, f2, y, x0 + (u64)lane) : 0xffffffffu;
    u32 t0 = __shfl(t, 0), t1 = __shfl(t, 1), t2 = __shfl(t, 2);
    u32 mn = t0; x_min = x0;
    if (t1 < mn) { mn = t1; x_min = x0 + 1; }
    if (t2 < mn) { mn = t2; x_min = x0 + 2; }
    return mn;
#else
    u32 mn = ~0u;
    for (u64 x = x0; x < x0 + 3; x++) { u32 t = wdist_raw(f1, f2, y, x); if (t < mn) { mn = t; x_min = x; } }
    return mn;
#endif
}
template <bool COOP = true>
LNR_HD inline u64 previous_window(FeatView f1, FeatView f2, u64 cord) {   // pmpfinder.cpp:883-945
    u64 gid = cord_id(cord), strand = cord_strand(cord);
    u64 x_suf = cord_x(cord) >> 4, y_suf = cord_y(cord) >> 4, x_min = 0;
    if (y_suf < 5 || x_suf < 6) return 0;
    u64 y = y_suf - 5;
    u32 mn = window_best3<COOP>(f1, f2, y, x_suf - 6, x_min);
    if (mn > 36) return 0;
    if (x_suf - x_min > 5) return mk_cord((gid << 30) + ((x_suf - 5) << 4), (x_suf - x_min - 5 + y) << 4, strand);
    return mk_cord((gid << 30) + (x_min << 4), y << 4, strand);
}
template <bool COOP = true>
LNR_HD inline u64 next_window(FeatView f1, FeatView f2, u64 cord) {   // pmpfinder.cpp:1079-1150
    u64 gid = cord_id(cord), strand = cord_strand(cord);
    u64 x_pre = cord_x(cord) >> 4, y_pre = cord_y(cord) >> 4, x_min = 0;
    if (y_pre + 12 > f1.n || x_pre + 12 > f2.n) return 0;
    u64 y = y_pre + 5;
    u32 mn = window_best3<COOP>(f1, f2, y, x_pre + 3, x_min);
    if (mn > 36) return 0;
    if (x_min - x_pre > 5) return mk_cord((gid << 30) + ((x_pre + 5) << 4), (x_pre + 5 - x_min + y) << 4, strand);
    return mk_cord((gid << 30) + (x_min << 4), y << 4, strand);
}
// SIMT-uniform: on the device every lane of the wave executes this with the same arguments (see window_best3);
// `tail` is the value of cords.back(), carried in a register so that no lane has to re-read the leader's store.
template <bool COOP = true>
LNR_HD inline bool extend_window_serial(FeatView f1, FeatView f2, Vec<u64> &cords, u64 &tail, u64 cordy_str, u64 cordy_end) {   // pmpfinder.cpp:1152-1178
    u32 p_str = cords.n - 1;
    u64 nc;
    while ((nc = previous_window<COOP>(f1, f2, tail)) && cord_y(nc) >= cordy_str) {
        if (cords.n >= cords.cap) { if (lnr_leader<COOP>()) *cords.ovf = 1; return false; }
        cords.template push_m<COOP>(nc); tail = nc;
    }
    u32 p_end = cords.n;
    if (p_end - p_str > 1) {
        lnr_sync<COOP>();
        if (lnr_leader<COOP>())
            for (u32 k = p_str; k < (p_str + p_end) / 2; k++) rs_swap(cords[k], cords[cords.n - k + p_str - 1]);
        lnr_sync<COOP>();
        tail = cords[cords.n - 1];
    }
    while ((nc = next_window<COOP>(f1, f2, tail)) && cord_y(nc) + 96 < cordy_end) {
        if (cords.n >= cords.cap) { if (lnr_leader<COOP>()) *cords.ovf = 1; return false; }
        cords.template push_m<COOP>(nc); tail = nc;
    }
    return true;
}
#if defined(__HIP_DEVICE_COMPILE__)
// The same walk with the window distances of several steps evaluated at once.  A step only chooses among three
// neighbouring windows, so the windows reachable within the next few steps form a small frontier: its distances are
// independent loads (one per lane), and the walk through them is register work.  One memory round trip per frontier instead
// of one per step; the choices, stop conditions and emitted cords are those of previous_window / next_window.
//   forward : step s looks at y0 + 5s and x in [x0 + 3s, x0 + 5s]                        -> 7 steps, 63 windows
//   backward: after d steps the state is (x0 - 5d + c, y0 - 5d + a), a + c <= d; step d looks at y0 - 5d + a (a < d) and
//             x0 - 5d + u (-1 <= u <= d)                                                   -> 4 steps, 50 windows
#if defined(LNR_PROF)
#define PD_CNT(k, v) do { lnr_pd_cnt[k] += (v); } while (0)
#define PD_T0() unsigned long long pd_t0_ = clock64()
#define PD_T(k) do { unsigned long long t_ = clock64(); lnr_pd_cnt[k] += t_ - pd_t0_; pd_t0_ = t_; } while (0)
static __device__ __attribute__((unused)) int lnr_pd_dummy;
#else
#define PD_CNT(k, v) do {} while (0)
#define PD_T0() do {} while (0)
#define PD_T(k) do {} while (0)
#endif
LNR_HD inline bool extend_window(FeatView f1, FeatView f2, Vec<u64> &cords, u64 &tail, u64 cordy_str, u64 cordy_end, unsigned long long *lnr_pd_cnt = nullptr) {
    const int lane = (int)(threadIdx.x & 63);
    (void)lnr_pd_cnt;
    PD_T0();
    // What every lane holds identically goes to scalar registers: the kernel runs at its VGPR cap, and a uniform pointer kept in (spilled)
    // vector registers cost a scratch reload and a wait for ALL outstanding memory traffic before every store of a cord -- a memory round
    // trip per cord pushed.  Scalar registers spill to VGPR lanes, which is a v_readlane.
    u64 *const cp = (u64 *)lnr_uni64((u64)cords.p);
    const u32 ccap = lnr_uni32(cords.cap);
    u32 cn = lnr_uni32(cords.n);
    f1.p = (const F96 *)lnr_uni64((u64)f1.p); f1.n = lnr_uni32(f1.n);
    f2.p = (const F96 *)lnr_uni64((u64)f2.p); f2.n = lnr_uni32(f2.n);
    const u32 p_str = cn - 1;
    tail = lnr_uni64(tail); cordy_str = lnr_uni64(cordy_str); cordy_end = lnr_uni64(cordy_end);
    // this lane's window inside the two frontiers
    const int d = lane < 3 ? 1 : (lane < 11 ? 2 : (lane < 26 ? 3 : 4));
    const int pre = d == 1 ? 0 : (d == 2 ? 3 : (d == 3 ? 11 : 26));
    const int la = (lane - pre) / (d + 2), lu = (lane - pre) % (d + 2) - 1;
    int s_ = 1;
    while ((s_ + 1) * (s_ + 1) - 1 <= lane) s_++;          // lane -> step: s^2 - 1 <= lane < (s + 1)^2 - 1
    const int off = lane - (s_ * s_ - 1);
    // Both walks start at the cord just pushed (after the backward cords are reversed the tail is that cord again): the first frontier of the
    // forward walk is loaded together with the backward one -- one memory round trip for the two.
    u32 dvf0;
    {
        u64 x0 = cord_x(tail) >> 4, y0 = cord_y(tail) >> 4;
        dvf0 = lane < 63 ? wdist_raw(f1, f2, y0 + 5 * (u64)s_, x0 + 3 * (u64)s_ + (u64)off) : 0xffffffffu;
    }
    // ---- backward
    {
        bool stop = false;
        while (!stop) {
            const u64 gid = cord_id(tail), strand = cord_strand(tail);
            const i64 x0 = (i64)(cord_x(tail) >> 4), y0 = (i64)(cord_y(tail) >> 4);
            i64 Y = y0 - 5 * d + la, X = x0 - 5 * d + lu;
            u32 dv = (lane < 50 && Y >= 0 && X >= 0) ? wdist_raw(f1, f2, (u64)Y, (u64)X) : 0xffffffffu;
            PD_CNT(1, 1);
            int a = 0, c = 0;
            for (int k = 0; k < 4; k++) {
                i64 x = x0 - 5 * k + c, y = y0 - 5 * k + a;       // state before step k + 1
                if (y < 5 || x < 6) { stop = true; break; }
                int dd = k + 1;
                int bl = (dd == 1 ? 0 : (dd == 2 ? 3 : (dd == 3 ? 11 : 26))) + a * (dd + 2) + c;
                u32 t0 = (u32)__builtin_amdgcn_readlane((int)dv, bl), t1 = (u32)__builtin_amdgcn_readlane((int)dv, bl + 1), t2 = (u32)__builtin_amdgcn_readlane((int)dv, bl + 2);
                u32 mn = t0; int j = 0;
                if (t1 < mn) { mn = t1; j = 1; }
                if (t2 < mn) { mn = t2; j = 2; }
                if (mn > 36) { stop = true; break; }
                u64 nc;
                if (j == 0) { nc = mk_cord((gid << 30) + ((u64)(x - 5) << 4), (u64)(y - 4) << 4, strand); a++; }
                else { nc = mk_cord((gid << 30) + ((u64)(x - 6 + j) << 4), (u64)(y - 5) << 4, strand); if (j == 2) c++; }
                if (!(cord_y(nc) >= cordy_str)) { stop = true; break; }
                if (cn >= ccap) { if (lnr_is_leader()) *cords.ovf = 1; cords.n = cn; return false; }
                if (lnr_is_leader()) cp[cn] = nc;
                cn++; tail = nc;
            }
        }
    }
    u32 p_end = cn;
    PD_T(6);
    PD_CNT(3, p_end - p_str - 1);
    if (p_end - p_str > 1) {
        PD_CNT(5, 1);
        lnr_wave_sync();
        if (lnr_is_leader())
            for (u32 k = p_str; k < (p_str + p_end) / 2; k++) rs_swap(cp[k], cp[cn - k + p_str - 1]);
        lnr_wave_sync();
        tail = lnr_uni64(cp[cn - 1]);
    }
    PD_T(7);
    // ---- forward
    {
        bool stop = false, have = true;
        while (!stop) {
            const u64 gid = cord_id(tail), strand = cord_strand(tail);
            const u64 x0 = cord_x(tail) >> 4, y0 = cord_y(tail) >> 4;
            u32 dv = have ? dvf0 : (lane < 63 ? wdist_raw(f1, f2, y0 + 5 * (u64)s_, x0 + 3 * (u64)s_ + (u64)off) : 0xffffffffu);
            have = false;
            PD_CNT(2, 1);
            u64 xc = x0, yc = y0;
            for (int k = 1; k <= 7; k++) {
                if (yc + 12 > f1.n || xc + 12 > f2.n) { stop = true; break; }
                int bl = k * k - 1 + (int)(xc - x0) - 3 * (k - 1);
                u32 t0 = (u32)__builtin_amdgcn_readlane((int)dv, bl), t1 = (u32)__builtin_amdgcn_readlane((int)dv, bl + 1), t2 = (u32)__builtin_amdgcn_readlane((int)dv, bl + 2);
                u32 mn = t0; u64 x_min = xc + 3;
                if (t1 < mn) { mn = t1; x_min = xc + 4; }
                if (t2 < mn) { mn = t2; x_min = xc + 5; }
                if (mn > 36) { stop = true; break; }
                u64 nc = mk_cord((gid << 30) + (x_min << 4), (yc + 5) << 4, strand);   // x_min - xc <= 5: the reference's other branch cannot be taken
                if (!(cord_y(nc) + 96 < cordy_end)) { stop = true; break; }
                if (cn >= ccap) { if (lnr_is_leader()) *cords.ovf = 1; cords.n = cn; return false; }
                if (lnr_is_leader()) cp[cn] = nc;
                cn++; tail = nc;
                xc = x_min; yc += 5;
                PD_CNT(4, 1);
            }
        }
    }
    PD_T(8);
    cords.n = cn;
    return true;
}
#else
LNR_HD inline bool extend_window(FeatView f1, FeatView f2, Vec<u64> &cords, u64 &tail, u64 cordy_str, u64 cordy_end) {
    return extend_window_serial(f1, f2, cords, tail, cordy_str, cordy_end);
}
#endif
struct GenomeFeat { const F96 *base; const u64 *off; u32 nseq; };   // f2 of all sequences, off[nseq+1] in entries
LNR_HD inline FeatView f2_view(GenomeFeat g, u64 id) {
    FeatView v;
    if (id >= g.nseq) { v.p = g.base; v.n = 0; return v; }
    v.p = g.base + g.off[id]; v.n = (u32)(g.off[id + 1] - g.off[id]);
    return v;
}
// _filterHits pmpfinder.cpp:1417-1445 in two steps: the window distances (elementwise, one hit per lane in the
// kernel), then the order-dependent compaction with block-end propagation.
LNR_HD inline void filter_hits_flags(const u64 *hits, u32 nhits, const FeatView f1[2], GenomeFeat g, i32 *keep, u32 first, u32 step) {
    for (u32 it = 1 + first; it < nhits; it += step) {
        u32 dist = wdist_checked(f1[cord_strand(hits[it])], f2_view(g, cord_id(hits[it])), cord_y(hits[it]) >> 4, cord_x(hits[it]) >> 4);
        keep[it] = dist < 50 ? 1 : 0;
    }
}
LNR_HD inline u32 filter_hits_apply(u64 *hits, u32 nhits, const i32 *keep) {
    u32 mv = 0;
    for (u32 it = 1; it < nhits; it++) {
        if (keep[it]) hits[it - mv] = hits[it];
        else mv++;
        if (is_end(hits[it])) hits[it - mv] |= F_END;
    }
    return nhits - mv;
}
// SIMT-uniform (all lanes execute it together on the device; hits are read-only, cords are written by the leader).
template <bool COOP = true>
LNR_HD inline void path_dst_2(const u64 *hits, u32 nhits, const FeatView f1_[2], GenomeFeat g, Vec<u64> &cords_, u64 read_str, u64 read_end, u64 L) {   // pmpfinder.cpp:1309-1410
    i64 hitBegin = 1, hitEnd = (i64)nhits;
    if (hitBegin >= hitEnd - 1) return;
    // (wave form: the uniform state in scalar registers, see extend_window; the count goes back to the caller's vector on every way out)
    Vec<u64> cords = cords_;
    struct NBack { Vec<u64> &dst; Vec<u64> &src; LNR_HD ~NBack() { dst.n = src.n; } } nback_{cords_, cords};
    FeatView f1[2] = {f1_[0], f1_[1]};
    if (COOP) {
        cords.p = (u64 *)lnr_uni64((u64)cords.p); cords.n = lnr_uni32(cords.n); cords.cap = lnr_uni32(cords.cap); cords.ovf = (int *)lnr_uni64((u64)cords.ovf);
        for (int k = 0; k < 2; k++) { f1[k].p = (const F96 *)lnr_uni64((u64)f1[k].p); f1[k].n = lnr_uni32(f1[k].n); }
        g.base = (const F96 *)lnr_uni64((u64)g.base); g.off = (const u64 *)lnr_uni64((u64)g.off); g.nseq = lnr_uni32(g.nseq);
        read_str = lnr_uni64(read_str); read_end = lnr_uni64(read_end); L = lnr_uni64(L);
        hitEnd = (i64)lnr_uni32(nhits);
    }
    u64 tail;
    if (cords.n == 0) { cords.template push_m<COOP>(F_END); tail = F_END; }   // initCords
    else tail = lnr_u64<COOP>(cords[cords.n - 1]);
    u64 ready_str, ready_end, cordy_str = 0, cordy_end = 0;
    bool f_sp_l, f_sp_r = false, f_block_end = false, f_append;
    i64 itt_next = hitBegin + 1, itt_first = hitBegin;
#if defined(LNR_PROF) && defined(__HIP_DEVICE_COMPILE__)
    unsigned long long pdc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pdt_ = clock64();
    struct PdPrint { unsigned long long *c; u32 nh; __device__ ~PdPrint() { if ((blockIdx.x % 9973) == 777 && (threadIdx.x & 63) == 0)
        printf("[pd2] block %u: hits %u, visited %llu, extensions %llu, backward frontiers %llu (cords %llu, reversals %llu), forward frontiers %llu (cords %llu); cycles: hit walk %llu, backward %llu, reversal %llu, forward %llu\n",
               blockIdx.x, nh, c[10], c[0], c[1], c[3], c[5], c[2], c[4], c[9], c[6], c[7], c[8]); } } pdp_{pdc_, nhits};
#endif
    // The hit walk reads every hit two or three times, each time as a dependent load (thousands of cycles under the load of the job kernels): on
    // the device the first 128 hits live in two registers per lane and are read with v_readlane (the index is wave-uniform); longer lists fall
    // back to memory beyond that.  The genome-feature view (two dependent loads of the offset table) is kept while the sequence id stays.
#if defined(__HIP_DEVICE_COMPILE__)
    u64 hreg0 = 0, hreg1 = 0;
    if (COOP) {
        const u32 ln_ = threadIdx.x & 63;
        hreg0 = ln_ < nhits ? hits[ln_] : 0; hreg1 = ln_ + 64 < nhits ? hits[ln_ + 64] : 0;
    }
    auto HIT = [&](i64 i) -> u64 {
        if (!COOP) return hits[i];
        if (i < 128) {
            const u64 r = i < 64 ? hreg0 : hreg1;
            const int l_ = (int)(i & 63);
            return ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(r >> 32), l_) << 32) | (u64)(u32)__builtin_amdgcn_readlane((int)(u32)r, l_);
        }
        return lnr_uni64(hits[i]);
    };
#else
    auto HIT = [&](i64 i) -> u64 { return hits[i]; };
#endif
    u64 view_id = ~0ULL; FeatView view_f2; view_f2.p = g.base; view_f2.n = 0;
    for (i64 itt = hitBegin; itt < hitEnd; itt = itt_next++) {
        const u64 hi = HIT(itt), hpv = HIT(itt - 1);
        bool first_i = is_end(hpv);
        ready_str = cord_strand(hi) ? L - read_end : read_str;
        ready_end = cord_strand(hi) ? L - read_str + 1 : read_end;
        i64 da_l = first_i ? 0 : labs64((i64)(cord_x(hi) - cord_x(hpv) - cord_y(hi) + cord_y(hpv)));
        f_sp_l = (da_l > 80) || cord_strand(hi ^ hpv);
        while (1) {
            if (itt_next >= hitEnd) { f_block_end = true; itt_first = itt_next; break; }
            const u64 hn = HIT(itt_next), hp = HIT(itt_next - 1);
            if (is_end(hp)) { f_block_end = true; itt_first = itt_next; break; }
            i64 da_r = labs64((i64)(cord_x(hn) - cord_x(hp) - cord_y(hn) + cord_y(hp)));
            f_sp_r = (da_r > 80) || cord_strand(hn ^ hp);
            if ((cord_y(hi) + 96 < cord_y(hn) && cord_x(hi) + 96 < cord_x(hn)) || f_sp_r) break;
            itt_next++;
        }
        f_append = false;
        if (!f_sp_r && !f_block_end) {
            cordy_str = f_sp_l ? hi : (first_i ? ready_str : cord_y(tail));
            cordy_end = cord_y(HIT(itt_next));
            if (cords.n >= cords.cap) { if (lnr_leader<COOP>()) *cords.ovf = 1; return; }
            cords.template push_m<COOP>(hi & ~F_END); tail = hi & ~F_END;
            f_append = true;
        } else {
            const u64 hl = HIT(itt_next - 1);
            if (!f_sp_l && cord_y(hl) >= 96 && cord_x(hl) >= 96) {
                u64 nc = shift_cord(hl, -96, -96);
                cordy_str = first_i ? read_str : cord_y(nc);
                cordy_end = cord_y(hl);
                if (cords.n >= cords.cap) { if (lnr_leader<COOP>()) *cords.ovf = 1; return; }
                cords.template push_m<COOP>(nc & ~F_END); tail = nc & ~F_END;
                f_append = true;
            }
        }
        if (is_end(hi) || f_block_end) { f_block_end = true; cordy_end = ready_end; }
        if (f_append && cord_id(hi) != view_id) { view_id = cord_id(hi); view_f2 = f2_view(g, view_id); }
#if defined(LNR_PROF) && defined(__HIP_DEVICE_COMPILE__)
        { unsigned long long t_ = clock64(); pdc_[9] += t_ - pdt_; pdt_ = t_; pdc_[0] += f_append ? 1 : 0; pdc_[10]++; }
        if (f_append && !(COOP ? extend_window(f1[cord_strand(hi)], view_f2, cords, tail, cordy_str, cordy_end, pdc_)
                               : extend_window_serial<false>(f1[cord_strand(hi)], view_f2, cords, tail, cordy_str, cordy_end))) return;
        pdt_ = clock64();
        if (0)
#endif
        if (f_append && !(COOP ? extend_window(f1[cord_strand(hi)], view_f2, cords, tail, cordy_str, cordy_end)
                               : extend_window_serial<false>(f1[cord_strand(hi)], view_f2, cords, tail, cordy_str, cordy_end))) return;
        if (f_block_end) { tail |= F_END; if (lnr_leader<COOP>()) cords[cords.n - 1] = tail; }
        itt_next = f_block_end ? itt_first : itt_next;
        f_sp_r = false; f_block_end = false;
    }
}

// -------------------------------------------------------------- cord blocks ----
LNR_HD inline u32 clean_blocks(u64 *cords, u32 n, u64 drop_len) {   // clean_blocks_ pmpfinder.cpp:1537-1581 (thd_map_error 50)
    if (n == 0) return 0;
    u64 ptr = 1, len = 0;
    for (u32 i = 1; i < n; i++) {
        len++;
        if (!is_end(cords[i - 1])) {
            i64 dx = (i64)(cord_x(cords[i]) - cord_x(cords[ptr - 1]));
            i64 dy = (i64)(cord_y(cords[i]) - cord_y(cords[ptr - 1]));
            if ((dx < 0 || dy < 0) && labs64(dx) < 50 && labs64(dy) < 50) { --len; --ptr; }
            else cords[ptr] = cords[i];
        } else cords[ptr] = cords[i];
        if (is_end(cords[i])) {
            ptr = len < drop_len ? ptr - len : ptr;
            len = 0;
            cords[ptr] |= F_END;
        }
        ptr++;
    }
    return (u32)ptr;
}
// gather_gaps_y_ pmpfinder.cpp:1592-1667; str_ends is sorted in place; gaps = forward-y intervals
LNR_HD inline int gather_gaps_y(UP *str_ends, u32 ns, Vec<UP> &gaps, u64 L, u64 gap_size, LeaderScratch &ls) {
    u64 cord_frt = 0, cord_end = L - 1;
    int sum = 0;
    UP u;
    if (ns == 0) { u.first = cord_frt; u.second = cord_end; gaps.push(u); sum += (int)(cord_y(cord_end) - cord_y(cord_frt)); return sum; }
    ref_sort(str_ends, (long)ns, [L](const UP &i, const UP &j) {
        u64 y1 = cord_strand(i.first) ? L - cord_y(i.second) - 1 : cord_y(i.first);
        u64 y2 = cord_strand(j.first) ? L - cord_y(j.second) - 1 : cord_y(j.first);
        return y1 < y2;
    }, ls.st);
    u64 f_cover = 0, cordy1 = 0, cordy2 = 0;
    UP y1 = forward_y(str_ends[0], L), y2 = y1;
    if (y1.first > gap_size) {
        cordy2 = cord_y(y1.first);
        u.first = cord_frt; u.second = cordy2; gaps.push(u);
        sum += (int)(cord_y(u.second) - cord_y(u.first));
    }
    for (u32 i = 1; i < ns; i++) {
        if (!f_cover) { y1 = forward_y(str_ends[i - 1], L); cordy1 = cord_y(y1.second); }
        y2 = forward_y(str_ends[i], L);
        cordy2 = cord_y(y2.first);
        if (y1.second > y2.second) f_cover = 1;
        else {
            if (y2.first > y1.second && y2.first - y1.second > gap_size) {
                u.first = cordy1; u.second = cordy2; gaps.push(u);
                sum += (int)(cord_y(u.second) - cord_y(u.first));
            }
            f_cover = 0;
        }
    }
    u64 max_y_end = f_cover ? y1.second : y2.second;
    if (L - max_y_end > gap_size) {
        u.first = max_y_end; u.second = cord_end; gaps.push(u);
        sum += (int)(cord_y(u.second) - cord_y(u.first));
    }
    return sum;
}

// chainBlocksSingleStrand cluster_util.cpp:936-975: sorts sep (in place), scores, chains
LNR_HD inline void chain_blocks_single_strand(const u64 *cords, UP *sep, u32 nb, i32 *sep_score, BlockSink &sink, int strand, u64 L, BlockScratch s) {
    if (strand)
        ref_sort(sep, (long)nb, [cords, L](const UP &a, const UP &b) {
            u64 y1 = !cord_strand(cords[a.first]) ? L - 1 - cord_y(cords[a.second - 1]) : cord_y(cords[a.first]);
            u64 y2 = !cord_strand(cords[b.first]) ? L - 1 - cord_y(cords[b.second - 1]) : cord_y(cords[b.first]);
            return y1 > y2;
        }, s.ls->st);
    else
        ref_sort(sep, (long)nb, [cords, L](const UP &a, const UP &b) {
            u64 y1 = cord_strand(cords[a.first]) ? L - 1 - cord_y(cords[a.second - 1]) : cord_y(cords[a.first]);
            u64 y2 = cord_strand(cords[b.first]) ? L - 1 - cord_y(cords[b.second - 1]) : cord_y(cords[b.first]);
            return y1 > y2;
        }, s.ls->st);
    for (u32 i = 0; i < nb; i++) sep_score[i] = (i32)((sep[i].second - sep[i].first) * 16);
    chain_blocks_base(sink, cords, sep, sep_score, nb, L, 3, strand, 0, s);
}
LNR_HD inline int best_strand(const BlockSink &c1, const BlockSink &c2) {   // getChainBlocksBestStrand cluster_util.cpp:979-1019
    u32 n = c1.nchains < c2.nchains ? c1.nchains : c2.nchains;
    int l1 = 0, l2 = 0;
    for (u32 i = 0; i < n; i++) {
        for (i32 j = c1.off[i]; j < c1.off[i + 1]; j++) l1 += (int)(c1.el[j].second - c1.el[j].first);
        for (i32 j = c2.off[i]; j < c2.off[i + 1]; j++) l2 += (int)(c2.el[j].second - c2.el[j].first);
        if (l1 < l2) return 1;
        else if (l1 > l2) return 0;
    }
    return 0;
}
LNR_HD inline void revert_chain_block_strand(BlockSink &cc, const u64 *cords, int strand) {   // cluster_util.cpp:1023-1063
    u64 f_strand = strand ? 1 : 0;
    for (u32 c = 0; c < cc.nchains; c++) {
        UP *el = cc.el + cc.off[c];
        u32 len = (u32)(cc.off[c + 1] - cc.off[c]) + 1;   // with the appended sentinel
        u64 swap_str = 0, pre = 0, cur = 0;
        for (u32 j = 0; j < len; j++) {
            if (j == len - 1 || cord_strand(cords[el[j].first]) == f_strand) cur = 0;
            else cur = 1;
            if (cur && !pre) swap_str = j;
            if (!cur && pre)
                for (u32 k = (u32)swap_str; k < (swap_str + j) / 2; k++) rs_swap(el[k], el[swap_str + j - 1 - k]);
            pre = cur;
        }
    }
}
LNR_HD inline u32 filter_blocks_cords(const BlockSink &ch, const u64 *hits, u64 *out, u64 major_limit) {   // _filterBlocksCords cluster_util.cpp:865-931 (f_header 1)
    if (ch.nchains == 0) return 0xffffffffu;
    u32 n = 0;
    u64 len_current = 0;
    out[n++] = hits[0];
    for (i32 i = ch.off[0]; i < ch.off[1]; i++) {
        for (u64 j = ch.el[i].first; j < ch.el[i].second; j++) out[n++] = hits[j] & ~F_END;
        len_current += ch.el[i].second - ch.el[i].first;
    }
    out[n - 1] |= F_END;
    float bound = 0.8 * len_current;
    u32 major_n = 1;
    for (u32 c = 1; c < ch.nchains && major_n < major_limit; c++) {
        len_current = 0;
        for (i32 j = ch.off[c]; j < ch.off[c + 1]; j++) len_current += ch.el[j].second - ch.el[j].first;
        if ((float)len_current > bound) {
            ++major_n;
            for (i32 j = ch.off[c]; j < ch.off[c + 1]; j++)
                for (u64 k = ch.el[j].first; k < ch.el[j].second; k++) out[n++] = hits[k] & ~F_END;
            out[n - 1] |= F_END;
        }
    }
    return n;
}

// ===================================================================== job ====
// One seeding job = apxMap_ on [read_str, read_end) of one read (pmpfinder.cpp:2632-2707)
// minus the seed lookup itself, which produced `a[0..n)` (a[0] is the dummy 0).
struct JobCtx {
    u64 L, read_str, read_end;
    int mode;
    FeatView f1[2];
    GenomeFeat g;
    u16 *bins; u32 nbins;
    u64 *pair_evals;
    unsigned long long *prof;   // diagnostic build only
    int traceback_done;         // kernel: the anchor traceback already ran (wave-parallel) and filled S.hits / S.hscore
};
// Per-job scratch, carved once from the job's arena (cap = anchor slots of the job).
struct JobScratch {
    Vec<u64> hits; Vec<i32> hscore; Vec<UP> sep, tmp;
    Rec rec;
    i32 *chain, *chain_sc, *cnt, *sep_score;
    u32 *xs, *ys;
    u64 *cuts, *xy_strs, *alt;
};
// cap = number of anchors that reach the chaining DP (m).  Arrays are requested hottest first so that a
// two-level arena keeps the DP / traceback state in its fast region.  S.alt (radix ping-pong buffer, sized
// by the raw anchor count) is carved by the caller.
LNR_HD inline bool job_carve(Arena &ar, u32 cap, JobScratch &S, int *ovf) {
    u32 c2 = cap + 2;
    S.xs = ar.get<u32>(c2); S.ys = ar.get<u32>(c2);
    S.rec.score = ar.get<i32>(c2); S.rec.score2 = ar.get<i32>(c2); S.rec.len = ar.get<i32>(c2);
    S.rec.p2 = ar.get<i32>(c2); S.rec.root = ar.get<i32>(c2); S.rec.leaf = ar.get<i32>(c2);
    // hits / hscore: always behind the fast region (the kernels reuse the whole fast region once the anchor traceback has filled them)
    Arena &far_ = ar.next ? *ar.next : ar;
    S.hits.init(far_.get<u64>(c2), c2, ovf);
    S.hscore.init(far_.get<i32>(c2), c2, ovf);
    if (far_.ovf) ar.ovf = 1;
    S.cnt = ar.get<i32>(c2); S.chain = ar.get<i32>(c2); S.chain_sc = ar.get<i32>(c2);
    S.sep.init(ar.get<UP>(c2), c2, ovf);
    S.sep_score = ar.get<i32>(c2);
    S.tmp.init(ar.get<UP>(c2), c2, ovf);
    S.cuts = ar.get<u64>(2 * (u64)c2); S.xy_strs = ar.get<u64>(c2);
    return !ar.ovf;
}
struct JobDebug { u64 *filt; u32 *nfilt; u64 *xsort; u32 *nxsort; u64 *hits_chain; u32 *nhits_chain; u64 *hits_blocks; u32 *nhits_blocks; };

// Phase 1 (serial): a = ascending-sorted anchors with a[0]==0.  filterAnchors1 compaction, the
// tie-sensitive x-descending sort of chainAnchorsHits (pmpfinder.cpp:2465), x/y extraction.
// Returns m, the number of anchors that enter the chaining DP.
LNR_HD inline u32 job_phase1(u64 *a, u32 n_sorted, JobDebug *dbg, LeaderScratch &ls) {
    u32 m = n_sorted > 1 ? filter_anchor_list(a, n_sorted) : n_sorted;   // filterAnchors1: length <= 1 -> unchanged
    if (dbg && dbg->filt) { for (u32 i = 0; i < m; i++) dbg->filt[i] = a[i]; *dbg->nfilt = m; }
    ref_sort(a, (long)m, [](const u64 &p, const u64 &q) { return anchor_x(p) > anchor_x(q); }, ls.st);
    if (dbg && dbg->xsort) { for (u32 i = 0; i < m; i++) dbg->xsort[i] = a[i]; *dbg->nxsort = m; }
    return m;
}
// x / y of the sorted anchors for the DP (elementwise; the kernel does this with all lanes)
LNR_HD inline void job_fill_xy(const u64 *a, u32 m, JobScratch &S, u32 first, u32 step) {
    for (u32 i = first; i < m; i += step) { S.xs[i] = (u32)anchor_x(a[i]); S.ys[i] = (u32)cord_y(a[i]); }
}
// (between the phases: the chaining DP over S.xs/S.ys into S.rec -- best_chains_serial here,
//  its wave-parallel twin in the kernel -- only when m >= 2, chainAnchorsBase cluster_util.cpp:450)

// Phase 3a: traceback -> hits, hit blocks, block chaining; leaves the surviving hits in (H, nH).  The caller
// continues with filter_hits_flags (elementwise), filter_hits_apply (leader) and path_dst_2 (SIMT-uniform) --
// path_dst alg 2, pmpfinder.cpp:1447-1469.  It comes in pieces so that the kernel can run the two quadratic
// middle steps (prefilter_chains2, getBestChains2) with all lanes; job_phase3a is the serial composition.
LNR_HD inline void job_blocks_gather(JobScratch &S, const JobCtx &c) {           // gather_blocks_ of pmpfinder.cpp:2535
    S.sep.n = 0; S.tmp.n = 0;
    gather_blocks(S.hits.p, S.hits.n, nullptr, S.sep, 1, S.hits.n, c.L, 600, 0, 0);
}
LNR_HD inline BlockScratch job_block_scratch(JobScratch &S, LeaderScratch &ls) {   // arrays of the anchor DP are dead and reused
    BlockScratch s; s.ptr = S.xs; s.sep_tmp = (UP *)S.cuts; s.score_tmp = (i32 *)S.ys; s.rec = S.rec; s.chain = S.chain_sc; s.chain_sc = (i32 *)S.xy_strs; s.cnt = S.cnt; s.ls = &ls;
    return s;
}
LNR_HD inline void job_blocks_scores(JobScratch &S, u32 first, u32 step) {       // pmpfinder.cpp:2540-2544
    for (u32 i = first; i < S.sep.n; i += step) S.sep_score[i] = S.hscore[(u32)S.sep[i].first] - S.hscore[(u32)S.sep[i].second - 1];
}
LNR_HD inline int job_blocks_finish(u64 *a, JobScratch &S, BlockSink &bs, JobDebug *dbg, u64 *&H, u32 &nH) {   // _filterBlocksHits
    Vec<u64> &hits = S.hits;
    u64 *hits2 = a;   // anchors are dead by now; hits never outnumber them
    u32 nh2 = filter_blocks_hits(bs, hits.p, hits2);
    if (nh2 == 0xffffffffu) { H = hits.p; nH = hits.n; } else { H = hits2; nH = nh2; }
    if (dbg && dbg->hits_blocks) { for (u32 i = 0; i < nH; i++) dbg->hits_blocks[i] = H[i]; *dbg->nhits_blocks = nH; }
    return *S.hits.ovf ? 1 : 0;
}
LNR_HD inline BlockSink job_block_sink(JobScratch &S) {
    BlockSink bs; bs.el = S.tmp.p; bs.off = S.chain; bs.nchains = 0; bs.nel = 0; bs.cap = S.tmp.cap; bs.ovf = S.hits.ovf; bs.first_len = 0; bs.off[0] = 0; bs.elements = nullptr;
    return bs;
}
LNR_HD inline int job_phase3a(u64 *a, u32 m, JobScratch &S, const JobCtx &c, JobDebug *dbg, u64 *&H, u32 &nH, LeaderScratch &ls) {
    Vec<u64> &hits = S.hits;
    unsigned long long tl_ = 0;
#if defined(LNR_PROF) && defined(__HIP_DEVICE_COMPILE__)
    tl_ = clock64();
#endif
    (void)tl_;
    if (!c.traceback_done) {
        hits.n = 0; S.hscore.n = 0;
        hits.push(F_END);        // initHits
        S.hscore.push(0);        // initHitsScore
        if (m >= 2) {
            AnchorSink sink; sink.anchors = a; sink.hits = &hits; sink.hscore = &S.hscore; sink.first_len = 0; sink.nchains = 0;
            traceback(S.rec, m, sink, S.chain, S.chain_sc, S.cnt, 1, 45, 50, 0.0f, ls);
        }
    }
    if (dbg && dbg->hits_chain) { for (u32 i = 0; i < hits.n; i++) dbg->hits_chain[i] = hits[i]; *dbg->nhits_chain = hits.n; }
    if (!c.traceback_done) LNR_TICK0(c.prof, 5, tl_);
    job_blocks_gather(S, c);
    prefilter_chains2(hits.p, hits.n, S.sep, S.cuts, S.xy_strs, S.tmp, ls);
    LNR_TICK0(c.prof, 6, tl_);
    job_blocks_scores(S, 0, 1);
    BlockSink bs = job_block_sink(S);
    BlockScratch s = job_block_scratch(S, ls);
    chain_blocks_base(bs, hits.p, S.sep.p, S.sep_score, S.sep.n, c.L, 2, 0, 1, s);   // chainBlocksHits cluster_util.cpp:721-732
    int rc = job_blocks_finish(a, S, bs, dbg, H, nH);
    LNR_TICK0(c.prof, 7, tl_);
    return rc;
}

// ==================================================================== tails ====
// Tail A = apxMap between the first apxMap_ and the remap loop (pmpfinder.cpp:2744-2749):
// clean, gather (sets block ends), gaps.  Returns the number of remap gaps written (0 = no remap).
LNR_HD inline u32 drop_len_of(u64 L) { i64 d = (i64)((double)L * 0.05 / 96); return (u32)(d < 2 ? d : 2); }

LNR_HD inline int tail_a(u64 *cords, u32 &ncords, u64 L, Arena &ar, UP *gaps_out, u32 gaps_cap, u32 &ngaps, u32 &remap, LeaderScratch &ls) {
    int ovf = 0;
    ncords = clean_blocks(cords, ncords, drop_len_of(L));
    u32 cap = ncords + 2;
    Vec<UP> str_ends; str_ends.init(ar.get<UP>(cap), cap, &ovf);
    Vec<UP> sep; sep.init(ar.get<UP>(cap), cap, &ovf);
    if (ar.ovf) return 1;
    gather_blocks(cords, ncords, &str_ends, sep, 1, ncords, L, 1000, 96, 1);
    Vec<UP> gaps; gaps.init(gaps_out, gaps_cap, &ovf);
    int sum = gather_gaps_y(str_ends.p, str_ends.n, gaps, L, 1000, ls);
    ngaps = gaps.n;
    remap = ((float)sum / (float)L >= 0.7f) ? 1 : 0;
    return ovf;
}
// Tail B = rest of apxMap (pmpfinder.cpp:2764-2801): re-gather, chain cord blocks on both strands, clean, flags.
// out_str/out_end receive the final cords; returns count through nout.
LNR_HD inline int tail_b(u64 *cords, u32 ncords, u64 L, Arena &ar, u64 *out_str, u64 *out_end, u32 out_cap, u32 &nout, LeaderScratch &ls) {
    int ovf = 0;
    u32 cap = ncords + 2;
    Vec<UP> sep; sep.init(ar.get<UP>(cap), cap, &ovf);
    gather_blocks(cords, ncords, nullptr, sep, 1, ncords, L, 1000, 96, 1);
    // chainBlocksCords cluster_util.cpp:1068-1102
    u32 nb = sep.n;
    UP *sep1 = ar.get<UP>(cap), *sep2 = ar.get<UP>(cap);
    i32 *score1 = ar.get<i32>(cap), *score2 = ar.get<i32>(cap);
    for (u32 i = 0; i < nb; i++) { sep1[i] = sep[i]; sep2[i] = sep[i]; }
    BlockSink c1, c2;
    c1.el = ar.get<UP>(cap + 1); c1.off = ar.get<i32>(cap + 2); c1.nchains = 0; c1.nel = 0; c1.cap = cap; c1.ovf = &ovf; c1.first_len = 0; c1.off[0] = 0;
    c2.el = ar.get<UP>(cap + 1); c2.off = ar.get<i32>(cap + 2); c2.nchains = 0; c2.nel = 0; c2.cap = cap; c2.ovf = &ovf; c2.first_len = 0; c2.off[0] = 0;
    BlockScratch s;
    s.ptr = ar.get<u32>(cap); s.sep_tmp = ar.get<UP>(cap); s.score_tmp = ar.get<i32>(cap);
    s.rec.score = ar.get<i32>(cap); s.rec.score2 = ar.get<i32>(cap); s.rec.len = ar.get<i32>(cap); s.rec.p2 = ar.get<i32>(cap); s.rec.root = ar.get<i32>(cap); s.rec.leaf = ar.get<i32>(cap);
    s.chain = ar.get<i32>(cap); s.chain_sc = ar.get<i32>(cap); s.cnt = ar.get<i32>(cap); s.ls = &ls;
    UP *sep_tmp2 = ar.get<UP>(cap);
    u64 *tmp_cords = ar.get<u64>(cap + 2);
    if (ar.ovf) return 1;
    chain_blocks_single_strand(cords, sep1, nb, score1, c1, 0, L, s);
    // chains of strand 0 reference s.sep_tmp through copies in c1.el, so the scratch can be reused
    BlockScratch s2 = s; s2.sep_tmp = sep_tmp2;
    chain_blocks_single_strand(cords, sep2, nb, score2, c2, 1, L, s2);
    int bst = best_strand(c1, c2);
    BlockSink &cc = bst == 0 ? c1 : c2;
    revert_chain_block_strand(cc, cords, bst);
    u32 n2 = filter_blocks_cords(cc, cords, tmp_cords, 2);
    u64 *C; u32 nC;
    if (n2 == 0xffffffffu) { C = cords; nC = ncords; } else { C = tmp_cords; nC = n2; }
    nC = clean_blocks(C, nC, drop_len_of(L));
    if (nC > out_cap) { nout = 0; return 1; }
    int seg = 0;
    const u64 d = (96ULL << 20) + 96ULL;
    for (u32 i = 0; i < nC; i++) {
        u64 v = C[i];
        if (seg) v |= F_RECD; else v &= ~F_RECD;
        v |= F_MAIN;
        if (is_end(v)) seg = 1 - seg;
        out_str[i] = v;
        out_end[i] = v + d;
    }
    nout = nC;
    return ovf;
}

}  // namespace lnr
