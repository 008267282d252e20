// lnr_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) of the filter hot path.
// Included by lnr_api.hip (single translation unit).  All per-read stage logic lives in
// lnr_hd.h; this file holds the data-parallel kernels and the wave-parallel twins of the
// serial stages (binning, radix sort, chaining DP).
//
// Kernel inventory (reference rows of SURVEY.md 8a in brackets):
//   index build  [a3-a5]: k_ix_chunk_const, k_ix_sample, k_ix_start_blk, k_max_top, k_ix_rec,
//                         k_ix_omit, k_scan_*, k_ix_scatter, k_ix_sort_small, k_ix_sort_big
//   features     [a6]   : k_f2 (genome), k_f1 (reads, both strands)
//   read prep    [a1,a2]: k_prep (2-bit packing of both strands + N bitmaps + hashInit N-skip)
//   seed lookup  [a3,a4,a7]: k_seed_fused (+ k_ix_bitmap at index time)
//   per-read job [a8-a16]: k_job (1 wave per read) / k_job_mid (4 waves) / k_job_heavy (16 waves per heavy read): binning,
//                         radix sort, filter, introsort, blocked chaining DP, traceback, blocks, windows
//   tails        [a17-a20]: k_tail_a, k_tail_b, k_gather_out
#pragma once
#include <hip/hip_runtime.h>
#include "lnr_hd.h"

namespace lnr {

}  // namespace lnr
#include "lnr_wave.h"
namespace lnr {

// small device -> pinned-host readbacks as a kernel's stores (see Readback in lnr_api.hip)
__global__ void __launch_bounds__(256) k_words_out(const u32 *src, u32 *dst, u64 n) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) dst[i] = src[i];
}

#if defined(K_JOB_WAVES) && K_JOB_WAVES > 4
#define JOB_INLINE __forceinline__
#else
#define JOB_INLINE
#endif
// ============================================================ generic scans ====
#define SCAN_TPB 256
#define SCAN_IPT 16
#define SCAN_BLK (SCAN_TPB * SCAN_IPT)

__global__ void __launch_bounds__(SCAN_TPB) k_scan_blk(const i32 *in, i32 *out, u64 n, i32 *blk_sums) {
    __shared__ i32 sh[SCAN_TPB];
    u64 base = (u64)blockIdx.x * SCAN_BLK + (u64)threadIdx.x * SCAN_IPT;
    i32 v[SCAN_IPT];
    i32 sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_IPT; k++) { u64 idx = base + k; v[k] = idx < n ? in[idx] : 0; sum += v[k]; }
    sh[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < SCAN_TPB; off <<= 1) {
        i32 t = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    i32 run = sh[threadIdx.x] - sum;
    if (threadIdx.x == SCAN_TPB - 1) blk_sums[blockIdx.x] = sh[SCAN_TPB - 1];
#pragma unroll
    for (int k = 0; k < SCAN_IPT; k++) { u64 idx = base + k; if (idx < n) out[idx] = run; run += v[k]; }
}
// exclusive scan of nblk block sums by one 1024-thread block
__global__ void __launch_bounds__(1024) k_scan_top(i32 *blk, u32 nblk) {
    __shared__ i32 sh[1024];
    __shared__ i32 carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (u32 base = 0; base < nblk; base += 1024) {
        u32 i = base + threadIdx.x;
        i32 v = i < nblk ? blk[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            i32 t = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        i32 c = carry;
        if (i < nblk) blk[i] = c + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + sh[1023];
        __syncthreads();
    }
}
__global__ void __launch_bounds__(SCAN_TPB) k_scan_add(i32 *out, u64 n, const i32 *blk) {
    u64 base = (u64)blockIdx.x * SCAN_BLK + (u64)threadIdx.x * SCAN_IPT;
    i32 add = blk[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_IPT; k++) { u64 idx = base + k; if (idx < n) out[idx] += add; }
}

// ============================================================== index build ====
// base ordinals above 4 are read as N (the reference's Dna5 has no such value; an unvalidated byte would leave the 26-bit
// minimizer range and corrupt the flag bits the build keeps beside it)
__global__ void __launch_bounds__(256) k_clamp_bases(u8 *g, u64 n16) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n16) return;
    uint4 v = ((uint4 *)g)[i];
    u32 w[4] = {v.x, v.y, v.z, v.w};
    bool ch = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        u32 x = w[k], hi = x & 0xF8F8F8F8u, c5 = ((x & 0x07070707u) + 0x03030303u) & 0x08080808u;   // byte >= 8, or 5..7
        u32 bad = ((hi | (hi >> 1) | (hi >> 2) | (hi >> 3) | (hi >> 4)) & 0x08080808u) | c5;
        if (bad) { u32 m = (bad >> 3) * 0xFFu; w[k] = (x & ~m) | (0x04040404u & m); ch = true; }
    }
    if (ch) ((uint4 *)g)[i] = make_uint4(w[0], w[1], w[2], w[3]);
}
struct ChunkDesc { u64 seq_off; i64 t_str; u64 samp_base; u32 nsamp; u32 seq_id; i32 ks; i32 C; };
static const u32 X_MASK = (1u << 26) - 1, X_REC = 1u << 30, X_FIRST = 1u << 31;

// hashInit's N-skip (shape_extend.cpp:95-105: the first position p >= t_str whose 21-mer holds no N) and the
// strand-selector constant of every chunk.  One 256-thread block per chunk: each thread brute-forces the 64
// start positions of its segment (20 bytes of look-ahead), block-min, next 16 KB tile until found -- GRCh38
// chromosomes open with megabases of N, which a single thread would walk for a second.
__global__ void __launch_bounds__(256) k_ix_chunk_const(const u8 *g, u64 gbytes, ChunkDesc *ch, u32 nch) {
    __shared__ unsigned long long s_best;
    u32 c = blockIdx.x;
    if (c >= nch) return;
    u64 s0 = ch[c].seq_off + (u64)ch[c].t_str;   // bytes past the buffer read as 0 (the zero padding continued)
    u64 found = ~0ULL;
    for (u64 tile = 0;; tile += 256 * 64) {
        if (threadIdx.x == 0) s_best = ~0ULL;
        __syncthreads();
        u64 seg = tile + (u64)threadIdx.x * 64;
        int run = 0;
        u64 mine = ~0ULL;
        for (int q = 0; q < 64 + 20; q++) {
            u64 idx = s0 + seg + (u64)q;
            if ((idx < gbytes ? g[idx] : (u8)0) == 4) run = 0; else run++;
            if (run >= 21 && q - 20 >= 0 && q - 20 < 64) { mine = seg + (u64)(q - 20); break; }
        }
        if (mine != ~0ULL) atomicMin(&s_best, (unsigned long long)mine);
        __syncthreads();
        found = s_best;
        __syncthreads();
        if (found != ~0ULL) break;
    }
    if (threadIdx.x == 0) {
        int ks = (int)found;
        ch[c].ks = ks;
        ch[c].C = shape_const(g + ch[c].seq_off, (u64)ch[c].t_str, ks, (u64)ch[c].t_str);
    }
}
// one thread per genome sample: minimizer X, Y, strand in closed form
__global__ void __launch_bounds__(256) k_ix_sample(const u8 *g, const ChunkDesc *ch, u32 nch, u64 nsamp, u32 *Xs, u64 *vals) {
    u64 m = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= nsamp) return;
    u32 lo = 0, hi = nch - 1;   // last chunk with samp_base <= m
    while (lo < hi) { u32 mid = (lo + hi + 1) >> 1; if (ch[mid].samp_base <= m) lo = mid; else hi = mid - 1; }
    ChunkDesc c = ch[lo];
    u64 ml = m - c.samp_base;
    u64 j = (u64)c.t_str + 8 + 9 * ml;
    SeedOut o = seed_sample(g + c.seq_off, j, (u64)c.t_str, (u64)c.t_str, c.ks, c.C);
    Xs[m] = o.X | (ml == 0 ? X_FIRST : 0);
    vals[m] = create_cord(c.seq_id, j + ANCHOR_ZERO, o.Y, o.strand);
}
// "recorded" rule of the two-pass build (index_util.cpp:1680-1685,1756-1767): a sample is kept iff it sits
// at an even position inside its run of equal minimizers (runs restart at chunk borders).  Needs, per
// sample, the index of the last run start at or before it: a max-scan, done in three kernels.
#define REC_TPB 256
#define REC_IPT 4
#define REC_BLK (REC_TPB * REC_IPT)
__device__ __forceinline__ bool ix_is_start(const u32 *Xs, u64 m) {
    u32 x = Xs[m];
    if (x & X_FIRST) return true;
    return (x & X_MASK) != (Xs[m - 1] & X_MASK);
}
__global__ void __launch_bounds__(REC_TPB) k_ix_start_blk(const u32 *Xs, u64 n, u32 *blk_max) {
    __shared__ u32 sh[REC_TPB];
    u64 base = (u64)blockIdx.x * REC_BLK + (u64)threadIdx.x * REC_IPT;
    u32 mx = 0;
    for (int k = 0; k < REC_IPT; k++) { u64 m = base + k; if (m < n && ix_is_start(Xs, m)) mx = (u32)m + 1; }
    sh[threadIdx.x] = mx;
    __syncthreads();
    for (int off = REC_TPB / 2; off > 0; off >>= 1) { if (threadIdx.x < (unsigned)off) sh[threadIdx.x] = max(sh[threadIdx.x], sh[threadIdx.x + off]); __syncthreads(); }
    if (threadIdx.x == 0) blk_max[blockIdx.x] = sh[0];
}
// exclusive running max over blocks
__global__ void __launch_bounds__(1024) k_max_top(u32 *blk, u32 nblk) {
    __shared__ u32 sh[1024];
    __shared__ u32 carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (u32 base = 0; base < nblk; base += 1024) {
        u32 i = base + threadIdx.x;
        u32 v = i < nblk ? blk[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            u32 t = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] = max(sh[threadIdx.x], t);
            __syncthreads();
        }
        u32 c = carry;
        u32 excl = threadIdx.x ? sh[threadIdx.x - 1] : 0;
        if (i < nblk) blk[i] = max(c, excl);
        __syncthreads();
        if (threadIdx.x == 1023) carry = max(c, sh[1023]);
        __syncthreads();
    }
}
__global__ void __launch_bounds__(REC_TPB) k_ix_rec(u32 *Xs, u64 n, const u32 *blk_prefix, i32 *cnt) {
    __shared__ u32 sh[REC_TPB];
    u64 base = (u64)blockIdx.x * REC_BLK + (u64)threadIdx.x * REC_IPT;
    u32 st[REC_IPT];
    u32 mx = 0;
    for (int k = 0; k < REC_IPT; k++) { u64 m = base + k; st[k] = (m < n && ix_is_start(Xs, m)) ? (u32)m + 1 : 0; mx = max(mx, st[k]); }
    sh[threadIdx.x] = mx;
    __syncthreads();
    for (int off = 1; off < REC_TPB; off <<= 1) {
        u32 t = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] = max(sh[threadIdx.x], t);
        __syncthreads();
    }
    u32 last = max(blk_prefix[blockIdx.x], threadIdx.x ? sh[threadIdx.x - 1] : 0u);
    __syncthreads();   // all reads of Xs[m-1] across thread borders are done (st[] computed) before flags are written
    for (int k = 0; k < REC_IPT; k++) {
        u64 m = base + k;
        if (m >= n) break;
        last = max(last, st[k]);
        if ((((u32)m + 1 - last) & 1) == 0) {
            u32 x = Xs[m];
            Xs[m] = x | X_REC;
            atomicAdd(&cnt[x & X_MASK], 1);
        }
    }
}
__global__ void k_ix_omit(i32 *cnt, u64 n) {   // buckets of more than 400 entries are emptied (index_util.cpp:1705-1708)
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && cnt[i] > 400) cnt[i] = 0;
}
__global__ void __launch_bounds__(256) k_ix_scatter(const u32 *Xs, const u64 *vals, u64 n, const i32 *dir, i32 *fill, u64 *hs) {
    u64 m = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n) return;
    u32 x = Xs[m];
    if (!(x & X_REC)) return;
    x &= X_MASK;
    i32 b = dir[x], e = dir[x + 1];
    if (e > b) { i32 p = atomicAdd(&fill[x], 1); hs[b + p] = vals[m]; }
}
// per-bucket ascending sort (index_util.cpp:1788-1796).  Entries are distinct, so the result is unique.
__global__ void __launch_bounds__(256) k_ix_sort_small(const i32 *dir, u64 nbuckets, u64 *hs, u32 *big, u32 *nbig) {
    u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbuckets) return;
    i32 s = dir[b], e = dir[b + 1];
    i32 len = e - s;
    if (len < 2) return;
    if (len > 32) { u32 i = atomicAdd(nbig, 1u); big[i] = (u32)b; return; }
    u64 *a = hs + s;
    for (int i = 1; i < len; i++) {
        u64 v = a[i];
        int j = i - 1;
        while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; j--; }
        a[j + 1] = v;
    }
}
__global__ void __launch_bounds__(64) k_ix_sort_big(const i32 *dir, u64 *hs, const u32 *big, u32 nbig) {
    __shared__ u64 sh[512];
    u32 bi = blockIdx.x;
    if (bi >= nbig) return;
    u32 b = big[bi];
    i32 s = dir[b], len = dir[b + 1] - s;   // 33..400
    for (int i = threadIdx.x; i < 512; i += 64) sh[i] = i < len ? hs[s + i] : ~0ULL;
    __syncthreads();
    for (int k = 2; k <= 512; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < 512; i += 64) {
                int l = i ^ j;
                if (l > i) {
                    u64 x = sh[i], y = sh[l];
                    bool up = (i & k) == 0;
                    if ((x > y) == up) { sh[i] = y; sh[l] = x; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < len; i += 64) hs[s + i] = sh[i];
}

// ================================================================= features ====
// genome window features f2 (createFeatures2_48 parallel form, pmpfinder.cpp:589-652): entry m of a
// sequence = 2-mer counts of bases [16m, 16m+48]; independent of the thread layout.
__global__ void __launch_bounds__(256) k_f2(const u8 *g, const u64 *seq_off, const u64 *f2_off, u32 nseq, u64 total, F96 *f2) {
    u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    u32 lo = 0, hi = nseq - 1;
    while (lo < hi) { u32 mid = (lo + hi + 1) >> 1; if (f2_off[mid] <= e) lo = mid; else hi = mid - 1; }
    const u8 *s = g + seq_off[lo] + 16 * (e - f2_off[lo]);
    i32 w0 = 0, w1 = 0, w2 = 0;
    u32 prev = s[0];
    for (int j = 1; j <= 48; j++) { u32 cur = s[j]; add2mer(w0, w1, w2, prev, cur); prev = cur; }
    F96 o; o.v0 = w0; o.v1 = w1; o.v2 = w2; o.pad = 0;
    f2[e] = o;
}

// ================================================================ read prep ====
// Per read: both strands 2-bit packed + N bitmaps (the reverse complement is _compltRvseStr, base.cpp:335-344) and the
// N-skip hashInit would take at the read start.  Packed layout per read: [forward words | reverse-complement words],
// packed_words(L) each (slack words zero).  A lane takes 16 strand positions: one 16-byte load (unaligned; for the reverse
// strand the window is read forwards and byte-reversed in registers), SWAR arithmetic folds the four dwords into 32 code
// bits + 16 N bits, and every lane stores its half word and its 16 bitmap bits -- consecutive lanes, consecutive addresses.
// Only the one window per strand that straddles the end of the read is assembled byte by byte.
__device__ __forceinline__ void prep_fold16(const u32 *x, bool rev, u32 vmask /* 0x01 per valid byte, same for the 4 dwords unless partial */,
                                            const u32 *vm4 /* per-dword masks or null */, u32 &codes, u32 &nbits) {
    codes = 0; nbits = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        u32 v = x[q];
        u32 vm = vm4 ? vm4[q] : vmask;
        u32 y = v & 0xFCFCFCFCu;                                           // ordinal > 3 -> N (values above 4 are clamped to N)
        u32 n01 = ((((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) >> 7) & 0x01010101u;
        u32 t = v & 0x03030303u;
        if (rev) t ^= vm * 3u;                                             // complement, not behind the end
        t &= ~(n01 * 255u);
        u32 c8 = (t | (t >> 6) | (t >> 12) | (t >> 18)) & 0xFFu;
        u32 n4 = (n01 | (n01 >> 7) | (n01 >> 14) | (n01 >> 21)) & 0xFu;
        codes |= c8 << (8 * q); nbits |= n4 << (4 * q);
    }
}
// Strand positions 16 g .. 16 g + 15 of strand `rev` (g may lie wholly behind the end: zeros).  Two halves so that a lane
// can have the loads of several groups in flight: the address computation has no branch (a group behind the end loads the
// first window of the read and is masked away; the window that straddles the end is moved back inside the read and shifted
// down in registers -- a byte loop there would be a chain of dependent loads).  Reads of at least 16 bases.
struct PrepGroup { u32 d[4]; u32 cnt; };
__device__ __forceinline__ void prep_group_load(const u8 *rd, u32 L, u32 g, bool rev, bool live, PrepGroup &p) {
    u32 i = 16 * g;
    bool in = live && i < L;
    u32 cnt = in ? (L - i < 16 ? L - i : 16) : 0;
    u32 sh = in ? 16 - cnt : 0;                          // bytes the window is moved by (0: the common case)
    u32 a = in ? (rev ? L - 16 - i + sh : i - sh) : 0;
    __builtin_memcpy(p.d, rd + a, 16);
    p.cnt = cnt;
}
__device__ __forceinline__ void prep_group_fold(const PrepGroup &p, bool rev, u32 &codes, u32 &nbits) {
    u32 x[4];
    if (rev) { x[0] = __builtin_bswap32(p.d[3]); x[1] = __builtin_bswap32(p.d[2]); x[2] = __builtin_bswap32(p.d[1]); x[3] = __builtin_bswap32(p.d[0]); }
    else { x[0] = p.d[0]; x[1] = p.d[1]; x[2] = p.d[2]; x[3] = p.d[3]; }
    if (p.cnt == 16) { prep_fold16(x, rev, 0x01010101u, nullptr, codes, nbits); return; }
    codes = 0; nbits = 0;
    if (p.cnt == 0) return;
    u64 lo = x[0] | ((u64)x[1] << 32), hi = x[2] | ((u64)x[3] << 32);
    u32 b = 8 * (16 - p.cnt);                            // 8 .. 120
    if (b >= 64) { lo = hi >> (b - 64); hi = 0; } else { lo = (lo >> b) | (hi << (64 - b)); hi >>= b; }
    x[0] = (u32)lo; x[1] = (u32)(lo >> 32); x[2] = (u32)hi; x[3] = (u32)(hi >> 32);
    u32 vm[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        u32 c = p.cnt > 4u * q ? p.cnt - 4u * q : 0; c = c > 4 ? 4 : c;
        vm[q] = 0x01010101u & (c == 4 ? 0xFFFFFFFFu : (1u << (8 * c)) - 1u);
        x[q] &= vm[q] * 255u;
    }
    prep_fold16(x, rev, 0, vm, codes, nbits);
}
// reads shorter than one window: byte by byte
__device__ __forceinline__ void prep_group_short(const u8 *rd, u32 L, u32 g, bool rev, u32 &codes, u32 &nbits) {
    codes = 0; nbits = 0;
    u32 i = 16 * g;
    if (i >= L) return;
    u32 x[4] = {0, 0, 0, 0}, vm[4] = {0, 0, 0, 0};
    for (u32 k = 0; i + k < L && k < 16; k++) {
        u32 b = rd[rev ? L - 1 - (i + k) : i + k];
        x[k >> 2] |= b << (8 * (k & 3)); vm[k >> 2] |= 1u << (8 * (k & 3));
    }
    prep_fold16(x, rev, 0, vm, codes, nbits);
}
// N-skip of hashInit at the read start (shape_init_skip: the first position that begins 21 consecutive non-N bases; positions
// behind the end count as non-N) by a whole wave: 62 x 16 start positions per step, each lane looking at its own 16 N bits and
// the 20 that follow.  A read that begins inside a long N run would otherwise have one lane walk it byte by byte -- 10^4
// dependent loads, milliseconds during which the kernel cannot end.  L >= 16, all 64 lanes active.
__device__ int prep_nskip_wave(const u8 *rd, u32 L) {
    int lane = lane_id();
    for (u32 base = 0;; base += 62 * 16) {
        PrepGroup pg;
        prep_group_load(rd, L, base / 16 + (u32)lane, false, true, pg);
        u32 c, nb;
        prep_group_fold(pg, false, c, nb);
        (void)c;
        u32 n1 = (u32)__shfl_down((int)nb, 1), n2 = (u32)__shfl_down((int)nb, 2);
        u64 z = ~((u64)nb | ((u64)n1 << 16) | ((u64)(n2 & 0xFu) << 32));             // 1 = not N, 36 positions
        u64 r2 = z & (z >> 1), r4 = r2 & (r2 >> 2), r8 = r4 & (r4 >> 4), r16 = r8 & (r8 >> 8);
        u32 m = (u32)(r16 & (r4 >> 16) & (z >> 20)) & 0xFFFFu;                        // bit s: positions s .. s+20 are all non-N
        u32 P = (lane < 62 && m) ? base + 16u * (u32)lane + (u32)__builtin_ctz(m) : 0xFFFFFFFFu;
        P = wave_min_u32(P);
        if (P != 0xFFFFFFFFu) return (int)P;
    }
}
#define PREP_UNROLL 4
__global__ void __launch_bounds__(256) k_prep(const u8 *__restrict__ src, const u64 *__restrict__ off, const u64 *__restrict__ pk_off, u32 n, u64 *__restrict__ pk, u32 *__restrict__ nm, i32 *__restrict__ read_ks) {
    for (u32 r = blockIdx.x; r < n; r += gridDim.x) {     // one read per workgroup by default (LNR_PREP_GRID); fewer workgroups loop
        u64 o = off[r];
        u32 L = (u32)(off[r + 1] - o);
        u32 nw = (u32)packed_words(L);
        const u8 *rd = src + o;
        u64 *pk64 = pk + pk_off[r];
        u32 *nm32 = nm + pk_off[r];
        u32 G = 2 * nw;                                   // groups of 16 positions per strand (slack included: zeros)
        for (u32 t0 = threadIdx.x; t0 < 2 * G; t0 += PREP_UNROLL * blockDim.x) {
            u32 codes[PREP_UNROLL], nbits[PREP_UNROLL];
            if (L >= 16) {
                PrepGroup pg[PREP_UNROLL];
#pragma unroll
                for (int u = 0; u < PREP_UNROLL; u++) {
                    u32 t = t0 + u * blockDim.x;
                    bool rev = t >= G;
                    prep_group_load(rd, L, rev ? t - G : t, rev, t < 2 * G, pg[u]);
                }
#pragma unroll
                for (int u = 0; u < PREP_UNROLL; u++) prep_group_fold(pg[u], t0 + u * blockDim.x >= G, codes[u], nbits[u]);
            } else {
#pragma unroll
                for (int u = 0; u < PREP_UNROLL; u++) {
                    u32 t = t0 + u * blockDim.x;
                    bool rev = t >= G;
                    codes[u] = 0; nbits[u] = 0;
                    if (t < 2 * G) prep_group_short(rd, L, rev ? t - G : t, rev, codes[u], nbits[u]);
                }
            }
            if (t0 < 64) {                                // wave 0, first pass: tasks 0 and 1 hold the N bits of the first 32 bases
                u32 nb0 = (u32)__builtin_amdgcn_readlane((int)nbits[0], 0), nb1 = (u32)__builtin_amdgcn_readlane((int)nbits[0], 1);
                // no N among the first 21 bases: hashInit skips nothing
                int ks = 0;
                if ((nb0 | (nb1 << 16)) & 0x1FFFFFu) {    // wave-uniform
                    if (L >= 16 && 2 * G >= 64) ks = prep_nskip_wave(rd, L);
                    else if (t0 == 0) { ByteSeq bs; bs.p = rd; bs.L = L; ks = shape_init_skip(bs); }   // short read: the literal walk
                }
                if (t0 == 0) read_ks[r] = ks;
            }
            // whole words leave the wave: the even lane of a pair stores the packed word and the bitmap word of both lanes (G and
            // the task index of an even lane are even, so a pair never straddles the strands or the end)
#pragma unroll
            for (int u = 0; u < PREP_UNROLL; u++) {
                u32 t = t0 + u * blockDim.x;
                u32 c_hi = DPP_MOV(0, codes[u], DPP_QUAD_XOR1, 0xf), n_hi = DPP_MOV(0, nbits[u], DPP_QUAD_XOR1, 0xf);
                if (t < 2 * G && !(t & 1)) {
                    pk64[t >> 1] = (u64)codes[u] | ((u64)c_hi << 32);
                    nm32[t >> 1] = nbits[u] | (n_hi << 16);
                }
            }
        }
    }
}
// read window features of both strands (createFeatures2_48 serial form, pmpfinder.cpp:556-588) from the packed strands:
// 16-base cells are counted once into LDS, an entry is the sum of three consecutive cells.
#define F1_TILE 1024
__global__ void __launch_bounds__(256) k_f1(const u64 *pk, const u32 *nm, const u64 *pk_off, const u32 *rlen, const u32 *nf, const u64 *f1_off, u32 n, F96 *f1) {
    __shared__ i32 c0[F1_TILE + 2], c1[F1_TILE + 2], c2[F1_TILE + 2];
    u32 r = blockIdx.x;
    if (r >= n) return;
    u32 cnt = nf[r];
    if (!cnt) return;
    u32 nw = (u32)packed_words(rlen[r]);
    F96 *out = f1 + f1_off[r];
    for (u32 strand = 0; strand < 2; strand++) {
        const u64 *p = pk + pk_off[r] + (strand ? nw : 0);
        const u32 *q = nm + pk_off[r] + (strand ? nw : 0);
        for (u32 m0 = 0; m0 < cnt; m0 += F1_TILE) {
            u32 me = cnt - m0 < F1_TILE ? cnt - m0 : F1_TILE;
            for (u32 c = threadIdx.x; c < me + 2; c += blockDim.x) cell_2mers_packed(p, q, 16ULL * (m0 + c), c0[c], c1[c], c2[c]);
            __syncthreads();
            for (u32 e = threadIdx.x; e < me; e += blockDim.x) {
                F96 o; o.v0 = c0[e] + c0[e + 1] + c0[e + 2]; o.v1 = c1[e] + c1[e + 1] + c1[e + 2]; o.v2 = c2[e] + c2[e + 1] + c2[e + 2]; o.pad = 0;
                out[strand * cnt + m0 + e] = o;
            }
            __syncthreads();
        }
    }
}

// ============================================================== seed lookup ====
struct JobArrays {
    const u32 *read, *str, *end, *mode;   // per job
};
struct ReadArrays {
    const u32 *len; const i32 *ks;
    const u64 *pk; const u32 *nm; const u64 *pk_off;   // 2-bit packed forward strand + N bitmap
};
struct SeedOutArrays {
    unsigned long long *cursor; u64 capacity; int *overflow;   // bump allocator over the anchor buffer (u64 slots)
    u64 *anchors; u64 *anc_off; u32 *job_cap; u32 *job_look; u32 *n_anchors;
};

// Bucket filter: 1 bit per 2^BM_GROUP_LOG2 consecutive minimizer buckets, set when any of them is non-empty.  More than
// half of the minimizers of an error-prone read fall into empty buckets; the filter answers those without touching `dir`.
// Measured: the exact bitmap (8 MB, group 1) beats a coarse 2 MB one (group 4, ~18 % false positives that each cost a
// `dir` line): 1.83 vs 1.99 ms per launch -- the exact bitmap's lines are served from L2 / Infinity Cache anyway.
#define BM_GROUP_LOG2 0
__global__ void __launch_bounds__(256) k_ix_bitmap(const i32 *dir, u64 nbuckets, u32 *bm) {
    u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 g0 = w * 32;                                   // first group of this word
    u64 ngroups = (nbuckets + (1u << BM_GROUP_LOG2) - 1) >> BM_GROUP_LOG2;
    if (g0 >= ngroups) return;
    u32 bits = 0;
    for (u32 q = 0; q < 32 && g0 + q < ngroups; q++) {
        u64 b0 = (g0 + q) << BM_GROUP_LOG2, b1 = b0 + (1u << BM_GROUP_LOG2);
        if (b1 > nbuckets) b1 = nbuckets;
        if (dir[b1] > dir[b0]) bits |= 1u << q;
    }
    bm[w] = bits;
}

// Bucket lines: the seed kernel's view of the DIndex.  Bucket X owns the 128-byte line X of `bl`: word 0 = first overflow line of
// the bucket in `ov` (low 32 bits) | bucket length (bits 32..47), words 1..15 = its first 15 entries (zero behind the end).  One
// HBM line then answers a lookup of a bucket of up to 15 entries completely -- dir line, hs line and the line fragments at both
// ends of the bucket's run in hs were three or more.  Longer buckets continue in `ov`: entries 15.. of every such bucket copied
// out of hs into whole 128-byte lines (zero padded), so that a run of m entries costs ceil(m / 16) lines and not the one more
// that a run at an arbitrary offset of hs touches at its ends.  Derived structures like the bitmap: built on every GPU from
// dir / hs (8.6 GB + ~2.6 GB of the 288 at GRCh38 scale), never broadcast.  dir and hs stay the parity surface.
#define BL_INLINE 15
__global__ void __launch_bounds__(256) k_ix_ovcount(const i32 *dir, u64 nbuckets, i32 *novl) {
    u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nbuckets) return;
    u32 dl = b < nbuckets ? (u32)(dir[b + 1] - dir[b]) : 0u;
    novl[b] = dl > BL_INLINE ? (i32)((dl - BL_INLINE + 15) >> 4) : 0;
}
__global__ void __launch_bounds__(256) k_ix_lines(const i32 *dir, const u64 *hs, const i32 *ovoff, u64 nbuckets, ulonglong2 *bl, u64 *ov, u64 *bh) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;     // one thread per 16-byte part of a line
    u64 b = t >> 3;
    if (b >= nbuckets) return;
    u32 part = (u32)t & 7;
    i32 ds = dir[b];
    u32 dl = (u32)(dir[b + 1] - ds);
    u32 o0 = (u32)ovoff[b];
    u64 w[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        u32 slot = 2 * part + (u32)k;
        if (slot == 0) w[k] = (u64)o0 | ((u64)dl << 32);
        else w[k] = slot - 1 < dl ? hs[(i64)ds + (i64)(slot - 1)] : 0ULL;
    }
    bl[t] = make_ulonglong2(w[0], w[1]);
    if (part == 0 && bh) bh[b] = w[0];                       // the header words on their own: what the seed kernel looks at one chunk ahead
    if (dl > BL_INLINE) {                                    // the eight threads of the bucket copy its overflow run
        u32 m = dl - BL_INLINE, mpad = ((m + 15) >> 4) << 4;
        for (u32 i = part; i < mpad; i += 8) ov[(u64)o0 * 16 + i] = i < m ? hs[(i64)ds + BL_INLINE + (i64)i] : 0ULL;
    }
}

// Seed lookup of one job per wave (getDIndexMatchAll, pmpfinder.cpp:1856-1913), one kernel, one pass over the read in chunks
// of 64 samples:
//   1. lane = sample: minimizer (2-bit packed path; byte path where an N or hashInit state intervenes), the "minimizer
//      changed" lookup rule (xpre == previous sample's X), optional bitmap test, then the header word of the bucket's line:
//      start and length of the bucket (the fetch also brings the line on chip for step 2);
//   2. lane = 128-byte line.  The lines a sample needs -- its bucket line, then the lines of `ov` that hold its entries from
//      the 16th on -- are numbered through the chunk in the reference's order; 64 lines per round go from HBM to LDS by
//      LDS-DMA (eight lanes per line), a lane takes one line (owner from one LDS scatter of the samples' first line numbers
//      + a ballot + a count-leading-zeros), runs the Y filter (pmpfinder.cpp:1890-1899) on its 16 words and keeps a 16-bit
//      mask of the entries that pass and belong to the bucket;
//   3. the survivors go, in lane order = the reference's order, into a ring in LDS; whenever 64 have gathered the wave turns
//      them into anchors (val2Anchor, index_util.cpp:1509-1520) and stores 512 contiguous bytes.
// The job's anchor segment is bump-allocated from an estimate (est_per_sample x samples, learned by the host from earlier
// batches) and moved to a segment of twice the size when it fills up -- the round-1 kernel counted all bucket lengths first
// and needed every bucket twice.
// History on the GRCh38 stand-in (100 k reads per launch, 663 lookups and 17.7 k bucket entries per read, 16.3 GB algorithmic):
//   round 1 kernel (lane = entry, one load in flight, per-lane binary search for the owner)    8.55 ms, 32 GB of HBM traffic
//   four loads in flight, owner by scatter + ballot, anchors from full groups of survivors     5.75 ms, 32 GB (5.6 TB/s: HBM bound)
//   bucket lines, lane = entry                                                                  7.1 ms, 24 GB (issue bound: 2.5 G VALU)
//   bucket lines, lane = line, LDS-DMA, headers one chunk ahead                                  5.35 ms, 31.5 GB
//   + overflow runs in whole lines (`ov`), Y filter of 6 instead of 10 instructions per word (this form)   4.95 ms
//   the same with the bucket lines prefetched into registers instead (every line fetched once: ~22 GB)     5.50 ms -- in-kernel
//     stamps: a wave waits 0.57 us per round for its lines and spends 4.5 us issuing: the kernel is bound by its instruction
//     count at 3 waves / SIMD (LDS: 13 KB per wave), not by traffic; the extra instructions of the register form cost more
//     than the saved traffic gave
#define SEED_RING 256
LNR_HD inline u32 y_match32(u32 hs_y, u32 Y) {   // y_match on the 20-bit y field (pmpfinder.cpp:1893-1894, ctz(0) pinned to "match")
    u32 v = hs_y ^ Y;
    u32 low = v & (0u - v);                        // lowest set bit (0 if none)
    return (v < 4u * low || v == 0) ? 1u : 0u;     // (v >> ctz(v)) < 4
}
// acc << 1 | "word does NOT pass the Y filter".  With v = y field ^ Y and low = lowest set bit of v: the word passes
// (y_match32) iff v < 4 low or v == 0, i.e. iff 4 low - v >= 0 (4 low == v is impossible for v != 0: low would not be v's lowest
// bit) -- the sign bit of 4 low - v is the answer and v_alignbit shifts it in.
__device__ inline u32 y_nomatch_push(u32 acc, u32 word, u32 Y) {
    u32 v = (word & 0xfffffu) ^ Y;
    u32 nv = 0u - v;
    u32 d = ((v & nv) << 2) + nv;
    return __builtin_amdgcn_alignbit(acc, d, 31);
}
__global__ void __launch_bounds__(64) k_seed_fused(JobArrays J, ReadArrays R, const ulonglong2 *bl, const u32 *bm /* null: table too dense to pay */, const u64 *ov, u32 njobs,
                                                   SeedOutArrays O, u32 est_per_sample_x16, const u64 *bh /* header words of the bucket lines, densely (null: read them from bl) */) {
    __shared__ uint4 s_rec[64];                // per sample: X, bucket start, bucket length, Y | strand << 8
    __shared__ u32 s_mk[64];                   // sample (lane + 1) whose first line has this number within the round
    __shared__ u64 s_src[64];                  // source address of the round's 64 lines (for the lanes that fetch them)
    __shared__ ulonglong2 s_line[64][8];       // the round's lines, landed by LDS-DMA: [line][16-byte part]
    __shared__ u64 s_sent[SEED_RING];          // ring of index entries that passed the Y filter ...
    __shared__ u32 s_sq[SEED_RING];            // ... and the sample (read-relative) | strand << 31 each belongs to
    u32 j = blockIdx.x;
    if (j >= njobs) return;
    int lane = lane_id();
    u32 r = J.read[j];
    const u64 *pk = R.pk + R.pk_off[r];
    const u32 *nm = R.nm + R.pk_off[r];
    u64 L = R.len[r];
    PackedSeq s; s.pk = pk; s.nm = nm; s.L = L;   // byte view for the rare samples the packed path declines (N nearby, hashInit state)
    u64 rs = J.str[j], re = J.end[j];
    u32 alpha = (u32)job_parm((int)J.mode[j]).alpha;
    int ks = R.ks[r];
    u64 k0 = rs + 21;
    int C = shape_const(s, 0, ks, k0);
    u32 ns = seed_num_samples(rs, re, alpha);
    // anchor segment: estimate now, grow when it fills up
    u64 seg_cap = ((((u64)ns * est_per_sample_x16) >> 4) + 192) & ~1ULL;
    unsigned long long off = 0;
    if (lane == 0) off = atomicAdd(O.cursor, (unsigned long long)seg_cap);
    off = __shfl((long long)off, 0);
    if (off + seg_cap > O.capacity) {            // the host re-runs with a larger buffer
        if (lane == 0) { *O.overflow = 1; O.n_anchors[j] = 0; O.job_cap[j] = 1; O.job_look[j] = 0; O.anc_off[j] = 0; }
        return;
    }
    u64 *out = O.anchors + off;
    if (lane == 0) out[0] = 0;   // the dummy the reference keeps at anchors[0] (base.cpp:272-277)
    u32 nout = 1;
    u32 cap = 0, looks = 0, carry = 0;
    u32 sh = 0, sc = 0;                            // ring head / fill (wave-uniform)
    bool dead = false;
    const u64 lane_le = (2ULL << lane) - 1ULL;     // lanes 0 .. lane
    auto emit = [&](u32 cnt) {                     // the first cnt (<= 64) entries of the ring become anchors
        if ((u64)nout + cnt > seg_cap) {           // segment full: move to one of twice the size (uniform branch)
            u64 ncap = (2 * seg_cap + 128) & ~1ULL;
            unsigned long long noff = 0;
            if (lane == 0) noff = atomicAdd(O.cursor, (unsigned long long)ncap);
            noff = __shfl((long long)noff, 0);
            if (noff + ncap > O.capacity) { if (lane == 0) *O.overflow = 1; dead = true; sc = 0; return; }
            u64 *nw = O.anchors + noff;
            WSYNC();                               // the wave's own earlier stores to the old segment have landed
            for (u32 i = (u32)lane; i < nout; i += 64) nw[i] = out[i];
            out = nw; off = noff; seg_cap = ncap;
        }
        WLDS();
        u32 p = (sh + (u32)lane) & (SEED_RING - 1);
        u64 ent = s_sent[p];
        u32 sq = s_sq[p];                          // sample | strand << 31
        if ((u32)lane < cnt) {
            u64 k = k0 + alpha - 1 + (u64)alpha * (sq & 0x7fffffffu);
            out[nout + (u32)lane] = val2anchor(ent, k, L, sq >> 31);
        }
        nout += cnt; sh = (sh + cnt) & (SEED_RING - 1); sc -= cnt;
        WLDS();
    };
    // ---- 1. minimizers of a chunk and the header words of its buckets: issued one chunk ahead, so the headers of chunk c + 1
    // travel while the lines of chunk c are fetched and filtered (a job is a chain of dependent memory round trips -- one for
    // the headers and three for the lines of every chunk -- and its latency, not the chip's bandwidth, set the kernel's time)
    auto sample_chunk = [&](u32 base, SeedOut &o, u64 &hdr) {
        u32 si = base + lane;
        bool valid = si < ns;
        o.X = 0; o.Y = 0; o.strand = 0;
        if (valid) {
            u64 k = k0 + alpha - 1 + (u64)alpha * si;
            if (!seed_sample_packed(pk, nm, k, k0, C, o)) o = seed_sample(s, k, k0, 0, ks, C);
        }
        u32 prev = __shfl_up(o.X, 1);
        if (lane == 0) prev = carry;
        carry = __shfl(o.X, 63);
        bool look = valid && o.X != prev;
        looks += (u32)__popcll(__ballot(look));
        u32 xg = o.X >> BM_GROUP_LOG2;
        bool fetch = look && (!bm || ((bm[xg >> 5] >> (xg & 31)) & 1));
        // the header comes from the dense table `bh` (8 bytes per bucket): read out of the bucket line itself it brought the whole 128-byte line
        // on chip one chunk early, and at GRCh38 scale that line was gone from L2 again when the LDS-DMA of step 2 asked for it -- every
        // bucket line crossed the fabric twice (round 2: 1.78 x the algorithmic bytes)
        hdr = fetch ? (bh ? bh[o.X] : bl[(u64)o.X * 8].x) : 0ULL;
    };
#ifndef SEED_PREFETCH
#define SEED_PREFETCH 1
#endif
    SeedOut o_n; u64 hdr_n = 0;
    if (SEED_PREFETCH && ns) sample_chunk(0, o_n, hdr_n);
    for (u32 base = 0; base < ns && !dead; base += 64) {
        SeedOut o = o_n; u64 hdr = hdr_n;
        if (!SEED_PREFETCH) sample_chunk(base, o, hdr);
        else if (base + 64 < ns) sample_chunk(base + 64, o_n, hdr_n);
        u32 ds = (u32)hdr, dl = (u32)(hdr >> 32) & 0xffffu;
        // lines of this sample: its bucket line + the aligned 16-entry lines of hs that hold entries 15 .. dl - 1
        u32 nl = dl ? 1u + ((dl + 15u - BL_INLINE) >> 4) : 0u;
        u32 lincl = wave_incl_scan(nl);
        u32 totalL = (u32)__builtin_amdgcn_readlane((int)lincl, 63);
        if (totalL == 0) continue;
        cap += wave_sum(dl);
        s_rec[lane] = make_uint4(o.X, ds, dl, (o.Y & 0xffu) | (o.strand << 8));
        u32 lexcl = lincl - nl;
        bool has = nl > 0;
        // ---- 2. the chunk's lines, 64 per round, one per lane
        for (u32 g0 = 0; g0 < totalL && !dead; g0 += 64) {
            s_mk[lane] = 0;
            WLDS();
            u32 rel = lexcl - g0;                 // (wraps for samples whose first line lies before the round)
            if (has && rel < 64) s_mk[rel] = (u32)lane + 1;
            WLDS();
            u32 m = s_mk[lane];
            u64 S = __ballot(m != 0);
            u64 cm = __ballot(has && lexcl < g0);                           // samples that start before the round: the last one runs into it
            int cq = cm ? 63 - __builtin_clzll(cm) : 0;
            u32 cexcl = (u32)__builtin_amdgcn_readlane((int)lexcl, cq);
            u64 below = S & lane_le;
            int pp = below ? 63 - __builtin_clzll(below) : 0;
            u32 mq = s_mk[pp];
            u32 q = below ? mq - 1 : (u32)cq;
            u32 li = below ? (u32)lane - (u32)pp : g0 + (u32)lane - cexcl;  // line of the sample: 0 = bucket line
            bool act = g0 + (u32)lane < totalL;
            uint4 rec = s_rec[q];
            const ulonglong2 *src = bl + (u64)rec.x * 8;
            u32 vm;                                                         // words of the line that are entries of this bucket
            {
                u32 n0 = rec.z < BL_INLINE ? rec.z : BL_INLINE;
                vm = ((1u << n0) - 1u) << 1;                                // bucket line: words 1 .. n0
                if (li) {
                    src = (const ulonglong2 *)(ov + ((u64)rec.y + li - 1) * 16);
                    u32 hi = rec.z - BL_INLINE - 16u * (li - 1);            // entries of the run from this line on
                    vm = hi >= 16 ? 0xffffu : (1u << hi) - 1u;
                }
                if (!act) { vm = 0; src = bl; }
            }
            // the 64 lines go from HBM to LDS without passing through registers: instruction t fetches lines 8 t .. 8 t + 7, eight
            // lanes per line (coalesced: a lane that loaded its own line made 64 requests per instruction and the kernel ran
            // at the rate of the address path)
            s_src[lane] = (u64)src;
            WLDS();
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const char *gp = (const char *)s_src[8 * t + (lane >> 3)] + 16 * (lane & 7);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gp, (__attribute__((address_space(3))) void *)&s_line[8 * t][0], 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            WLDS();
            u32 Y = rec.w & 0xffu;
            u32 nm16 = 0;
#pragma unroll
            for (int t = 7; t >= 0; t--) {
                u32 pt = ((u32)t + (u32)lane) & 7;
                ulonglong2 w = s_line[lane][pt];
                nm16 = y_nomatch_push(nm16, (u32)w.y, Y);
                nm16 = y_nomatch_push(nm16, (u32)w.x, Y);
            }
            u32 mr = ~nm16 & 0xffffu, rs16 = (2u * (u32)lane) & 15u;
            u32 mm = ((mr << rs16) | (mr >> (16u - rs16))) & 0xffffu;
            mm &= vm;
            u32 cnt = (u32)__popc(mm);
            u32 incl = wave_incl_scan(cnt);
            u32 tot = (u32)__builtin_amdgcn_readlane((int)incl, 63);
            if (tot == 0) continue;
            u32 pos = incl - cnt;
            u32 sqv = (base + q) | ((rec.w >> 8) << 31);
            // ---- 3. survivors -> ring (all lanes at once when they fit, else eight lanes at a time), anchors out.
            // A line has about one survivor: the k-th survivor of every lane is read from the staged line and placed at pos + k.
            // (A ring of 128 entries -- 14 instead of 12 waves per CU -- with sub-rounds of 32 / 4 lanes measured 1 % faster in a loop
            // over the seed stage alone and 1-2 % slower in the whole path: the four-lane fallback is taken often at sampling step 7.)
            const u64 *srcw = (const u64 *)&s_line[lane][0];
            u32 nq = sc + tot <= SEED_RING ? 1u : 8u;                       // (8 lanes hold at most 128 survivors; sc < 64 here)
            for (u32 qt = 0; qt < nq; qt++) {
                u32 qb = 0, qn = tot;
                if (nq > 1) {
                    qb = qt ? (u32)__builtin_amdgcn_readlane((int)incl, (int)(8 * qt - 1)) : 0u;
                    qn = (u32)__builtin_amdgcn_readlane((int)incl, (int)(8 * qt + 7)) - qb;
                }
                bool mine = nq == 1 || (u32)(lane >> 3) == qt;
                u32 pr = sh + sc + pos - qb;
                u32 mk = mine ? mm : 0u;
                while (__ballot(mk != 0)) {
                    if (mk) {
                        u32 kbit = (u32)__builtin_ctz(mk);
                        mk &= mk - 1;
                        u32 p = pr & (SEED_RING - 1);
                        s_sent[p] = srcw[kbit]; s_sq[p] = sqv; pr++;
                    }
                }
                sc += qn;
                while (sc >= 64 && !dead) emit(64);
            }
            WLDS();                                                         // the ring pushes read s_line: done before the next round lands
        }
    }
    while (sc && !dead) emit(sc < 64 ? sc : 64);
    if (lane == 0) { O.job_cap[j] = cap + 1; O.job_look[j] = looks; O.anc_off[j] = off; O.n_anchors[j] = dead ? 0 : nout; }
}

// ===================================================== HIndex (-i 2) build + lookup ====
// SURVEY 8 a21 / f3.  The parity surface is `ysa` (index_util.cpp:719-845 hash array, 430-560 block sort, 1294-1461 _createYSA): blocks
// [head = ptr << 40 | X][bodies = 1 << 63 | Y << 41 | reverse << 40 | id << 30 | pos, descending], X ascending, two zero words behind.
// The reference's open-addressed table (XString) is an exact dictionary as compiled, so this build keeps its own
// lookup tables, derived from ysa: hdir[X] = index of the block's head, and for the blocks of >= 1024 entries the sorted list of
// (X, Y20) -> first body of the Y run that the reference's table holds.
#define HX_SPAN 17
#define HX_WEIGHT 9
#define HX_STEP 8u
#define HX_BLOCKLIMIT 1024u
#define HX_XBITS (2 * HX_WEIGHT)
static const u64 HS_TYPEFLAG = 1ULL << 63, HS_MASK40 = (1ULL << 40) - 1, HS_PTRMASK = (1ULL << 23) - 1, HS_YMASK = (1ULL << 20) - 1, HS_CODEFLAG = 1ULL << 40;
LNR_HD inline u64 hs_head(u64 ptr, u64 x) { return ((ptr << 40) + x) & (HS_TYPEFLAG - 1); }
LNR_HD inline u64 hs_head_ptr(u64 v) { return (v >> 40) & HS_PTRMASK; }
LNR_HD inline u64 hs_body_y(u64 v) { return (v >> 41) & HS_YMASK; }
// A -t chunk of a sequence is cut into pieces of HX_PIECE positions, one thread each.  The reference's loop (__createHsArray,
// index_util.cpp:736-800) is a sequential walk with a rolling state, but the state is a function of the bases: after 17 rolls the
// hash words are the window itself, and the strand selector x is C + 2 * (sum of the window) with C = -51 from any hashInit whose
// window was clean -- only the chunk's very first hashInit can leave another C (an N among the chunk's first 16 bases), and that C
// holds until the first N enters a window (position kt0).  So a piece whose predecessor position has a clean window rebuilds the
// state there; one that starts inside an N cluster starts like the reference's re-initialisation does, at the first clean window.
// What needs the neighbours is settled on the host from four words per piece: the first sample of a piece is dropped when its X
// equals the X of the sample before it, and the chunk's last block is filed under the X of the chunk's last hashed position.
#define HX_PIECE 32768u
struct HxPiece { u64 seq_off; u64 start, chunk; u64 u, v; u64 out_base; u64 kt0; u64 kinit; u64 nc; u64 slen; u32 seq_id; u32 first; };
struct HxPieceOut { u32 cnt, firstX, lastX, endX, hashed, pad; };
// sequence bytes through an 8-byte window (one thread walks a chunk: a byte load per base would be a memory round trip per base)
struct HxBytes {
    const u8 *base; u64 cur; u64 word;
    __device__ u32 get(u64 i) {
        u64 w = i >> 3;
        if (w != cur) { word = *(const u64 *)(base + (w << 3)); cur = w; }
        return (u32)(word >> (8 * (i & 7))) & 0xffu;
    }
};
struct HxShape { u64 h, crh; int x, left; };
__device__ inline u64 hx_hash_init(HxShape &me, HxBytes &b, u64 p) {                // hashInit shape_extend.cpp:86-116
    me.left = 0; me.h = 0; me.crh = 0; me.x = -3;
    u64 k = 0, count = 0;
    while (count < HX_SPAN) {
        if (b.get(p + k + count) == 4) { k += count + 1; count = 0; }
        else count++;
    }
    unsigned bit = 2;
    for (unsigned i = 0; i < HX_SPAN - 1; ++i) {
        u64 val = b.get(p + k + i);
        me.x += ((int)val << 1) - 3;
        me.h = (me.h << 2) + val;
        me.crh += (3ULL - val) << bit;
        bit += 2;
    }
    return k;
}
__device__ inline void hx_roll(HxShape &me, u32 v_in, u32 v_left) {                 // state update of hashNext shape_extend.cpp:136-145
    const u64 mask = (1ULL << (2 * HX_SPAN - 2)) - 1;
    me.h = ((me.h & mask) << 2) + v_in;
    me.crh = ((me.crh >> 2) & mask) + ((3ULL - (u64)v_in) << (2 * HX_SPAN - 2));
    me.x += (int)(((u64)v_in - (u64)(i64)me.left) << 1);
    me.left = (int)v_left;
}
__device__ inline void hx_xy(const HxShape &me, u32 &X, u64 &Y, u32 &strand) {       // X / Y of hashNext shape_extend.cpp:146-167
    u64 v2; unsigned t = 0;
    if (me.x > 0) { v2 = me.h; strand = 0; } else { v2 = me.crh; strand = 1; }
    u64 xv = (1ULL << (2 * HX_SPAN)) - 1;
    for (unsigned k = 64 - 2 * HX_SPAN; k <= 64 - 2 * HX_WEIGHT; k += 2) {
        u64 v1 = v2 << k >> (64 - 2 * HX_WEIGHT);
        if (xv > v1) { xv = v1; t = k; }
    }
    X = (u32)xv;
    Y = (v2 >> (64 - t) << (64 - t - 2 * HX_WEIGHT)) + (v2 & ((1ULL << (64 - t - 2 * HX_WEIGHT)) - 1)) + ((u64)t << (2 * HX_SPAN - 2 * HX_WEIGHT - 1));
}
// first position p in [from, limit) whose 17 bases hold no N, or ~0 (reads bytes up to limit + 15)
__device__ inline u64 hx_find_clean(HxBytes &b, u64 from, u64 limit) {
    u32 run = 0;
    for (u64 q = from; q < limit + HX_SPAN - 1; q++) {
        if (b.get(q) == 4) run = 0; else run++;
        if (run >= HX_SPAN) return q - (HX_SPAN - 1);
    }
    return ~0ULL;
}
// per piece, bounded scans: first N among the bases that enter the piece's windows ([u + 16, v + 16)), first clean window in [u, v);
// the last piece of a sequence also looks behind the last window (positions v .. len: the padding is clean)
__global__ void __launch_bounds__(64) k_hx_pre(const u8 *g, const HxPiece *pc, u32 npc, u64 *firstN, u64 *firstClean, u64 *tailClean) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npc) return;
    HxPiece d = pc[i];
    HxBytes b; b.base = g + d.seq_off; b.cur = ~0ULL; b.word = 0;
    u64 hit = ~0ULL;
    for (u64 p = d.u + 16; p < d.v + 16; p++) if (b.get(p) == 4) { hit = p; break; }
    firstN[i] = hit;
    firstClean[i] = hx_find_clean(b, d.u, d.v);
    tailClean[i] = d.v + HX_SPAN - 1 >= d.slen ? hx_find_clean(b, d.v, d.slen + 1) : ~0ULL;
}
__global__ void __launch_bounds__(64) k_hx_piece(const u8 *g, const HxPiece *pc, u32 npc, u32 *fileX, u64 *body, HxPieceOut *po) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npc) return;
    HxPiece d = pc[i];
    HxBytes bi; bi.base = g + d.seq_off; bi.cur = ~0ULL; bi.word = 0;      // bytes entering the window
    HxBytes bl = bi;                                                           // bytes leaving it
    HxShape sh;
    const u64 start = d.start, chunk = d.chunk, v = d.v;
    const u64 bound = chunk - HX_SPAN + 1 + start;                              // jumps beyond it end the chunk (:771-774)
    const u64 k_end = chunk - (chunk + start) % HX_STEP + HX_STEP + start;
    u64 k = start;
    bool go = true;
    if (d.first) hx_hash_init(sh, bi, d.kinit);                                // = hashInit(start): kinit is the first clean window at or behind start
    else {
        u64 km = d.u - 1;
        bool clean = true;
        int W = 0;
        for (int q = 0; q < HX_SPAN; q++) { u32 c = bi.get(km + q); if (c == 4) clean = false; W += (int)c; }
        if (clean) {                                                            // the walk hashes u - 1: its state there, from the bases
            sh.h = 0; sh.crh = 0;
            for (int q = 0; q < HX_SPAN; q++) { u64 c = bi.get(km + q); sh.h = (sh.h << 2) + c; sh.crh |= (3ULL - c) << (2 * q); }
            int C = -3 * HX_SPAN;
            if (d.kinit != start && km < d.kt0) {                               // still under the chunk's first hashInit, which skipped Ns: C = -51 + 2 a - 2 b
                int a_ = 0, b_ = 0;
                for (int q = 0; q < HX_SPAN - 1; q++) { a_ += (int)bl.get(d.kinit + q); b_ += (int)bl.get(start + q); }
                C += 2 * a_ - 2 * b_;
            }
            sh.x = C + 2 * W;
            sh.left = (int)bi.get(km);
            k = d.u;
        } else {                                                                // the walk is jumping over an N cluster here: it lands on the first clean window
            u64 kc = hx_find_clean(bi, d.u, v);
            if (kc == ~0ULL || kc > bound) go = false;                          // (a landing beyond the bound belongs to the piece that held the trigger)
            else { hx_hash_init(sh, bi, kc); k = kc; }
        }
    }
    u32 preX = 0xffffffffu, firstX = 0, n = 0, hashed = 0;
    bool have_first = d.first != 0;                                             // the chunk's first sample is always kept (preX = ~0)
    u32 *fx = fileX + d.out_base; u64 *bd = body + d.out_base;
    if (go) for (; k < v; k++) {
        bool last = false;
        if (bi.get(k + HX_SPAN - 1) == 4) {
            u64 kc = hx_find_clean(bi, k, v);
            if (kc == ~0ULL) kc = d.nc;                                          // exact landing position behind this piece (host: suffix over the pieces)
            if (kc > bound) { hx_hash_init(sh, bi, kc); k = k_end; last = true; }
            else if (kc >= v) break;
            else { hx_hash_init(sh, bi, kc); k = kc; }
        }
        hx_roll(sh, bi.get(k + HX_SPAN - 1), bl.get(k));
        hashed = 1;
        if (k % HX_STEP == 0) {
            u32 X, strand; u64 Y;
            hx_xy(sh, X, Y, strand);
            if (!have_first || X != preX) {
                u64 w = (((Y << 41) | HS_TYPEFLAG) + ((u64)d.seq_id << 30) + k);
                if (strand) w |= HS_CODEFLAG;
                fx[n] = X; bd[n] = w; n++;
            }
            if (!have_first) { firstX = X; have_first = true; }
            preX = X;
        }
        if (last) break;
    }
    HxPieceOut o; o.cnt = n; o.firstX = firstX; o.lastX = preX; o.hashed = hashed; o.pad = 0; o.endX = 0;
    if (hashed) { u32 X, strand; u64 Y; hx_xy(sh, X, Y, strand); o.endX = X; }
    po[i] = o;
}
// staging -> dense arrays; src_skip drops the piece's first sample (its X equals the sample before it), patchX files the chunk's last block
struct HxCopy { u64 src, dst; u32 n; u32 patch; u32 patchX; u32 pad; };
__global__ void __launch_bounds__(256) k_hx_compact(const HxCopy *cp, u32 npc, const u32 *fileX, const u64 *body, u32 *Xs, u64 *bodies) {
    u32 c = blockIdx.x;
    if (c >= npc) return;
    HxCopy d = cp[c];
    for (u32 i = threadIdx.x; i < d.n; i += blockDim.x) {
        u32 x = fileX[d.src + i];
        if (d.patch && i == d.n - 1) x = d.patchX;
        Xs[d.dst + i] = x; bodies[d.dst + i] = body[d.src + i];
    }
}
// after the two sorts (bodies descending, then stable by X): run starts and per-X counts
__global__ void __launch_bounds__(256) k_hx_flags(const u32 *Xs, u64 n, i32 *flag, u32 *cntX) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 x = Xs[i];
    flag[i] = (i == 0 || Xs[i - 1] != x) ? 1 : 0;
    atomicAdd(&cntX[x], 1u);
}
// ysa[i + r] = body i (r = run starts up to and including i), the run's first body also writes the head; bodies of blocks under the
// block limit lose their Y field (_createYSA :1431-1434)
__global__ void __launch_bounds__(256) k_hx_assemble(const u32 *Xs, const u64 *bodies, u64 n, const i32 *flag_excl, const u32 *cntX, u64 *ysa, u64 ysa_len) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { ysa[ysa_len - 2] = 0; ysa[ysa_len - 1] = 0; }
    if (i >= n) return;
    u32 x = Xs[i];
    bool st = i == 0 || Xs[i - 1] != x;
    u64 r = (u64)flag_excl[i] + (st ? 1 : 0);
    u32 c = cntX[x];
    u64 w = bodies[i];
    if (c + 1 < HX_BLOCKLIMIT) w &= ~(HS_YMASK << 41);
    ysa[i + r] = w;
    if (st) ysa[i + r - 1] = hs_head((u64)c + 1, x);
}
// ---- lookup tables derived from ysa (at build and at adopt)
__global__ void __launch_bounds__(256) k_hx_derive(const u64 *ysa, u64 ysa_len, i32 *hdir, i32 *node_flag) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ysa_len) return;
    u64 w = ysa[i];
    node_flag[i] = 0;
    if (!(w & HS_TYPEFLAG)) { if (hs_head_ptr(w)) hdir[(u32)(w & ((1u << HX_XBITS) - 1))] = (i32)i; }
}
// nodes of the blocks of >= 1024 entries: a body whose Y field differs from the word in front of it (head included), _createYSA :1446-1452
__global__ void __launch_bounds__(256) k_hx_nodes_mark(const u64 *ysa, const i32 *hdir, i32 *node_flag) {
    u32 x = blockIdx.x;                                      // one workgroup per X
    i32 hd = hdir[x];
    if (hd < 0) return;
    u64 ptr = hs_head_ptr(ysa[hd]);
    if (ptr < HX_BLOCKLIMIT) return;
    for (u64 q = (u64)hd + 1 + threadIdx.x; q < (u64)hd + ptr; q += blockDim.x)
        if (hs_body_y(ysa[q] ^ ysa[q - 1])) node_flag[q] = 1;
}
__global__ void __launch_bounds__(256) k_hx_nodes_fill_blk(const u64 *ysa, const i32 *hdir, const i32 *node_flag, const i32 *node_excl, u64 *nkeys, u32 *nvals) {
    u32 x = blockIdx.x;
    i32 hd = hdir[x];
    if (hd < 0) return;
    u64 ptr = hs_head_ptr(ysa[hd]);
    if (ptr < HX_BLOCKLIMIT) return;
    for (u64 q = (u64)hd + 1 + threadIdx.x; q < (u64)hd + ptr; q += blockDim.x)
        if (node_flag[q]) { u32 slot = (u32)node_excl[q]; nkeys[slot] = ((u64)x << 20) | hs_body_y(ysa[q]); nvals[slot] = (u32)q; }
}
// getXDir (index_util.cpp:1071-1093) on the derived tables
__device__ inline u64 hx_get_xdir(u32 X, u32 Y, const u64 *ysa, u64 empty_dir, const i32 *hdir, const u64 *nkeys, const u32 *nvals, u32 nnodes) {
    i32 hd = hdir[X];
    if (hd < 0) return empty_dir;
    u64 ptr = hs_head_ptr(ysa[hd]);
    if (ptr < HX_BLOCKLIMIT) return (u64)hd + 1;
    if (Y > HS_YMASK || nnodes == 0) return empty_dir;
    u64 key = ((u64)X << 20) | Y;
    u32 lo = 0, hi = nnodes;                                 // first node with key >= key (the list is sorted, equal keys in ysa order)
    while (lo < hi) { u32 mid = (lo + hi) >> 1; if (nkeys[mid] < key) lo = mid + 1; else hi = mid; }
    return (lo < nnodes && nkeys[lo] == key) ? (u64)nvals[lo] : empty_dir;
}
// Seed lookup of one job per wave against the HIndex (getHIndexMatchAll, pmpfinder.cpp:1918-1974).  lane = sample for the
// minimizers of a chunk of 64; the lookups then run one after the other in sample order, the wave walking ysa 64 words at a time.
__global__ void __launch_bounds__(64) k_seed_hindex(JobArrays J, ReadArrays R, const u64 *ysa, u64 ysa_len, u64 empty_dir, const i32 *hdir, const u64 *nkeys, const u32 *nvals, u32 nnodes,
                                                    u32 njobs, SeedOutArrays O, u32 est_per_sample_x16) {
    u32 j = blockIdx.x;
    if (j >= njobs) return;
    int lane = lane_id();
    u32 r = J.read[j];
    PackedSeq s; s.pk = R.pk + R.pk_off[r]; s.nm = R.nm + R.pk_off[r]; s.L = R.len[r];
    u64 L = s.L;
    u64 rs = J.str[j], re = J.end[j];
    u32 alpha = (u32)job_parm((int)J.mode[j]).alpha;
    u64 k0 = rs;                                                                   // the loop rolls from read_str on (the DIndex form starts a span later)
    u32 ns = 0;
    if (re > HX_SPAN && rs + alpha - 1 < re - HX_SPAN) ns = (u32)((re - HX_SPAN - 1 - (rs + alpha - 1)) / alpha) + 1;   // samples k = rs + alpha - 1 + alpha i < re - span
    int ks = shape_init_skip_t<HX_SPAN>(s);
    int C = shape_const_t<HX_SPAN>(s, 0, ks, k0);
    u64 seg_cap = ((((u64)ns * est_per_sample_x16) >> 4) + 192) & ~1ULL;
    unsigned long long off = 0;
    if (lane == 0) off = atomicAdd(O.cursor, (unsigned long long)seg_cap);
    off = __shfl((long long)off, 0);
    if (off + seg_cap > O.capacity) {
        if (lane == 0) { *O.overflow = 1; O.n_anchors[j] = 0; O.job_cap[j] = 1; O.job_look[j] = 0; O.anc_off[j] = 0; }
        return;
    }
    u64 *out = O.anchors + off;
    if (lane == 0) out[0] = 0;
    u32 nout = 1, cap = 0, looks = 0, carry = 0;
    bool dead = false;
    for (u32 base = 0; base < ns && !dead; base += 64) {
        u32 si = base + (u32)lane;
        bool valid = si < ns;
        SeedOut o; o.X = 0; o.Y = 0; o.strand = 0;
        u64 k = k0 + alpha - 1 + (u64)alpha * si;
        if (valid) o = seed_sample_t<HX_SPAN, HX_WEIGHT>(s, k, k0, 0, ks, C);
        u32 prev = __shfl_up(o.X, 1);
        if (lane == 0) prev = carry;
        carry = __shfl(o.X, 63);
        bool look = valid && o.X != prev;
        u64 lm = __ballot(look);
        looks += (u32)__popcll(lm);
        while (lm && !dead) {
            int q = __builtin_ctzll(lm);
            lm &= lm - 1;
            u32 X = (u32)__builtin_amdgcn_readlane((int)o.X, q), Y = (u32)__builtin_amdgcn_readlane((int)o.Y, q), strand = (u32)__builtin_amdgcn_readlane((int)o.strand, q);
            u64 kq = k0 + alpha - 1 + (u64)alpha * (base + (u32)q);
            u64 pos = hx_get_xdir(X, Y, ysa, empty_dir, hdir, nkeys, nvals, nnodes);
            u64 ptr = hs_head_ptr(ysa[pos - 1]);
            if (pos == empty_dir || ptr >= 64) continue;
            for (;;) {                                                              // while (Y field == Y or == 0) emit; stop at the end of ysa
                u64 p = pos + (u64)lane;
                bool inr = p <= ysa_len - 1;
                u64 w = inr ? ysa[p] : 0;
                u32 wy = (u32)hs_body_y(w);
                bool ok = inr && (wy == Y || wy == 0);
                u64 bad = __ballot(!ok);
                u32 nv = bad ? (u32)__builtin_ctzll(bad) : 64u;
                if (nv) {
                    if ((u64)nout + nv > seg_cap) {                                 // segment full: move to one of twice the size
                        u64 ncap = (2 * seg_cap + 128) & ~1ULL;
                        unsigned long long noff = 0;
                        if (lane == 0) noff = atomicAdd(O.cursor, (unsigned long long)ncap);
                        noff = __shfl((long long)noff, 0);
                        if (noff + ncap > O.capacity) { if (lane == 0) *O.overflow = 1; dead = true; break; }
                        u64 *nw = O.anchors + noff;
                        WSYNC();
                        for (u32 i = (u32)lane; i < nout; i += 64) nw[i] = out[i];
                        out = nw; off = noff; seg_cap = ncap;
                    }
                    if ((u32)lane < nv) {
                        u64 idx = w & HS_MASK40;                                    // (idx_str = 0, idx_end = 2^40 - 1 for every job of this path: never out of range)
                        u64 id = (idx >> 30) & 1023ULL, x = idx & ((1ULL << 30) - 1);
                        u64 rev = ((w >> 40) & 1ULL) ^ strand;
                        u64 y = rev ? L - 1 - kq : kq;
                        out[nout + (u32)lane] = create_cord(id, x - y + ANCHOR_ZERO, y, rev);
                    }
                    nout += nv; cap += nv;
                }
                if (nv < 64) break;
                pos += 64;
                if (pos > ysa_len - 1) break;
            }
        }
    }
    if (lane == 0) { O.job_cap[j] = cap + 1; O.job_look[j] = looks; O.anc_off[j] = off; O.n_anchors[j] = dead ? 0 : nout; }
}

// =================================================================== job =====
struct JobArgs {
    const u32 *grp_order;   // groups in launch order (heaviest first: the long tail of repeat-rich reads starts early)
    const u32 *grp_beg;     // [ngroups+1] job ranges; all jobs of a group belong to one read and run in order
    JobArrays J;
    const u64 *anc_off; const u32 *job_cap; const u32 *n_anchors; const u64 *scr_off;
    u64 *anchors; char *scratch;
    const u32 *read_len; const u64 *f1_off; const u32 *nf; const F96 *f1;
    GenomeFeat g;
    u64 *cords; const u64 *cords_off; const u32 *cords_cap; u32 *ncords; i32 *read_err;
    u32 nbins; u32 grp_lo, grp_hi;
    u32 lds_bytes;          // dynamic LDS per block: binning histogram (swept in passes of lds_bytes bins) first, then the fast half of the job arena
    u32 arena_lds;          // bytes of that LDS the job arena may use
    u32 stop_after;         // diagnostic (LNR_STOP_AFTER, single-wave kernel only): leave the job after phase stop_after - 1; 0 = run everything
    u32 *jstate;            // split path: per job {anchors after binning, anchors after the list filter | ok << 31} handed from the pre to the DP / post kernel
    unsigned long long *prof;   // diagnostic build (-DLNR_PROF) only: per-phase cycle sums of lane 0
    unsigned long long *tl;     // diagnostic build only: per launch position {start, end (100 MHz ticks), hw id, anchors in the DP}
};

// wave-parallel twin of binning_filter_serial (binningFilter, pmpfinder.cpp:1979-2012): histogram of anchor x-field / 30000,
// anchors in bins of more than 10 are kept, order preserved (ballot compaction).
// The histogram lives in `hist_bytes` bytes of LDS, one saturating byte per bin.  A reference sequence of length G has
// (G + 2^21) / 30000 bins -- 8 350 for chr1, up to 35 800 at the format limit -- so the bin range is swept in passes of
// hist_bytes bins: the LDS a job workgroup needs does not depend on the reference (it did in round 1, and cut the
// residency of the bulk kernel at human scale).  With more than one pass the anchors of a kept bin are marked in place
// (bit 63, unused by the anchor format cords.cpp:319-322) and the compaction strips the mark.
#define BIN_SAT 100u   /* counts saturate here: only "more than 10" is asked.  At most 64 adds are in flight beyond it (one
                          wave instruction; its undo is issued before the next add), so a byte never carries into its neighbour */
__device__ u32 binning_exact_wave(u64 *a, u32 n, u32 *binw, u32 hist_bytes, u32 nbins) {
    int lane = lane_id();
    const u64 MARK = 1ULL << 63;
    u32 per = hist_bytes & ~3u;                         // bins per pass
    u32 npass = (nbins + per - 1) / per;
    for (u32 p = 0; p < npass; p++) {
        u32 b_lo = p * per, b_n = nbins - b_lo < per ? nbins - b_lo : per;
        for (u32 w = lane; w < (b_n + 3) / 4; w += 64) binw[w] = 0;
        WSYNC();
        for (u32 i0 = 0; i0 < n; i0 += 256) {   // four independent loads in flight per lane
            u64 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { u32 i = i0 + 64 * u + (u32)lane; v[u] = i < n ? a[i] : ~0ULL; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                u32 i = i0 + 64 * u + (u32)lane;
                u32 b = (u32)(cord_x(v[u]) / 30000) - b_lo;
                if (i < n && b < b_n) {
                    u32 sh = 8 * (b & 3), inc = 1u << sh;
                    u32 old = atomicAdd(&binw[b >> 2], inc);
                    if (((old >> sh) & 0xffu) >= BIN_SAT) atomicSub(&binw[b >> 2], inc);
                }
                WLDS();                         // this chunk's undo is issued before the next chunk's add
            }
        }
        WSYNC();
        if (npass == 1) break;
        for (u32 i0 = 0; i0 < n; i0 += 256) {   // mark the anchors of this pass's kept bins
            u64 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { u32 i = i0 + 64 * u + (u32)lane; v[u] = i < n ? a[i] : ~0ULL; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                u32 i = i0 + 64 * u + (u32)lane;
                u32 b = (u32)(cord_x(v[u]) / 30000) - b_lo;
                if (i < n && b < b_n && ((binw[b >> 2] >> (8 * (b & 3))) & 0xffu) > 10) a[i] = v[u] | MARK;
            }
        }
        WSYNC();
    }
    u32 ii = 0;
    // software pipeline: the next 256 anchors are in flight while this group of four chunks is compacted (in place: a
    // store lands at or below the chunk being compacted, i.e. below everything that is still to be read)
    u64 vn[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { u32 i = 64 * u + (u32)lane; vn[u] = i < n ? a[i] : 0; }
    for (u32 base = 0; base < n; base += 256) {
        u64 vc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) vc[u] = vn[u];
#pragma unroll
        for (int u = 0; u < 4; u++) { u32 i = base + 256 + 64 * u + (u32)lane; vn[u] = i < n ? a[i] : 0; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (base + 64 * u >= n) break;               // uniform
            u32 i = base + 64 * u + (u32)lane;
            u64 v = vc[u];
            bool keep = false;
            if (i < n) {
                if (npass == 1) {
                    u32 b = (u32)(cord_x(v) / 30000);
                    if (b < nbins) keep = ((binw[b >> 2] >> (8 * (b & 3))) & 0xffu) > 10;
                } else keep = (v & MARK) != 0;
            }
            u64 mask = __ballot(keep);
            if (keep) a[ii + __popcll(mask & lanemask_lt())] = v & ~MARK;
            ii += (u32)__popcll(mask);
        }
    }
    WSYNC();
    return ii;            // (0: nothing survives; the caller then keeps everything, pmpfinder.cpp:2007-2010)
}

// binningFilter at human scale.  A read carries ~1 700 anchors into this stage, all but a few hundred of them chance hits spread
// over the genome, and the exact histogram sweeps the whole list five times (two passes of bins at 6 KB of LDS: count, mark, count,
// mark, compact), and a reference at the format's length limit would need six passes of bins.  So first a conservative filter: one saturating byte per HASHED bin
// (bin mod 4096, 4 KB of LDS).  A hashed count is at least the bin's true count, so an anchor whose hashed count is <= 10 is
// dropped by the reference too; every anchor of a bin that the reference keeps survives.  The survivors (the true clusters plus
// the few chance hits that share a hashed bin with one) go to `tmp` in order, and the exact histogram runs on them alone: for a
// bin of more than 10 anchors all members survived, so its count among the survivors is its true count, and a smaller bin stays
// at or below its true count.  `a` is untouched until the result is known (nothing survives -> everything is kept).
// (Measured on the GRCh38 stand-in: no change of the kernel's time against the plain two-pass histogram -- the stage costs 1.3 ms of
// 31 when the kernel has the chip to itself; what made it look like 7 ms was the 4-wave kernel holding 14 of 16 wave slots per CU.)
#define BIN_HASH 4096u
__device__ JOB_INLINE u32 binning_wave(u64 *a, u32 n, u64 *tmp, u32 *binw, u32 hist_bytes, u32 nbins) {
    int lane = lane_id();
    if (n <= 256 || hist_bytes < BIN_HASH || !tmp) {           // short lists: the exact histogram directly (in place)
        u32 k = binning_exact_wave(a, n, binw, hist_bytes, nbins);
        return k ? k : n;
    }
    for (u32 w = lane; w < BIN_HASH / 4; w += 64) binw[w] = 0;
    WSYNC();
    for (u32 i0 = 0; i0 < n; i0 += 256) {
        u64 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { u32 i = i0 + 64 * u + (u32)lane; v[u] = i < n ? a[i] : ~0ULL; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            u32 i = i0 + 64 * u + (u32)lane;
            u32 b = (u32)(cord_x(v[u]) / 30000);
            if (i < n && b < nbins) {
                u32 hb = b & (BIN_HASH - 1), sh = 8 * (hb & 3), inc = 1u << sh;
                u32 old = atomicAdd(&binw[hb >> 2], inc);
                if (((old >> sh) & 0xffu) >= BIN_SAT) atomicSub(&binw[hb >> 2], inc);
            }
            WLDS();
        }
    }
    WSYNC();
    u32 s = 0;
    u64 vn[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { u32 i = 64 * u + (u32)lane; vn[u] = i < n ? a[i] : 0; }
    for (u32 base = 0; base < n; base += 256) {
        u64 vc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) vc[u] = vn[u];
#pragma unroll
        for (int u = 0; u < 4; u++) { u32 i = base + 256 + 64 * u + (u32)lane; vn[u] = i < n ? a[i] : 0; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (base + 64 * u >= n) break;               // uniform
            u32 i = base + 64 * u + (u32)lane;
            bool keep = false;
            if (i < n) {
                u32 b = (u32)(cord_x(vc[u]) / 30000);
                if (b < nbins) { u32 hb = b & (BIN_HASH - 1); keep = ((binw[hb >> 2] >> (8 * (hb & 3))) & 0xffu) > 10; }
            }
            u64 mask = __ballot(keep);
            if (keep) tmp[s + __popcll(mask & lanemask_lt())] = vc[u];
            s += (u32)__popcll(mask);
        }
    }
    WSYNC();
    if (s == 0) return n;
    u32 k = binning_exact_wave(tmp, s, binw, hist_bytes, nbins);
    if (k == 0) return n;
    for (u32 i = (u32)lane; i < k; i += 64) a[i] = tmp[i];
    WSYNC();
    return k;
}

// wave-level LSD radix sort, ascending u64 (the reference's ska_sort, base.cpp:570; result unique).
// Returns with the sorted keys in `a`.  hist = 256 LDS words.
__device__ void radix_sort_wave(u64 *a, u64 *alt, u32 n, u32 *hist) {
    int lane = lane_id();
    u64 *src = a, *dst = alt;
    for (int pass = 0; pass < 8; pass++) {
        int shift = pass * 8;
        for (int b = lane; b < 256; b += 64) hist[b] = 0;
        WSYNC();
        for (u32 i0 = 0; i0 < n; i0 += 256) {   // four independent loads in flight per lane
            u64 k[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { u32 i = i0 + 64 * u + lane; k[u] = i < n ? src[i] : 0; }
#pragma unroll
            for (int u = 0; u < 4; u++) if (i0 + 64 * u + lane < n) atomicAdd(&hist[(k[u] >> shift) & 255], 1u);
        }
        WSYNC();
        // exclusive prefix over the 256 bins: 4 bins per lane
        u32 c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
        bool uniform = (c0 == n) || (c1 == n) || (c2 == n) || (c3 == n);
        if (__ballot(uniform)) { WSYNC(); continue; }   // every key has the same digit: pass is the identity
        u32 s4 = c0 + c1 + c2 + c3;
        u32 ex = wave_incl_scan(s4) - s4;
        WSYNC();
        hist[4 * lane] = ex; hist[4 * lane + 1] = ex + c0; hist[4 * lane + 2] = ex + c0 + c1; hist[4 * lane + 3] = ex + c0 + c1 + c2;
        WSYNC();
        // software pipeline: the next 256 keys are in flight while this group of four chunks is placed
        u64 kn[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { u32 i = 64 * u + lane; kn[u] = i < n ? src[i] : 0; }
        for (u32 base = 0; base < n; base += 256) {
            u64 kc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) kc[u] = kn[u];
#pragma unroll
            for (int u = 0; u < 4; u++) { u32 i = base + 256 + 64 * u + lane; kn[u] = i < n ? src[i] : 0; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (base + 64 * u >= n) break;           // uniform
                u32 i = base + 64 * u + lane;
                bool valid = i < n;
                u64 key = kc[u];
                u32 d = (u32)(key >> shift) & 255;
                u64 m = __ballot(valid);
                for (int bit = 0; bit < 8; bit++) { u64 bm = __ballot((d >> bit) & 1); m &= ((d >> bit) & 1) ? bm : ~bm; }
                u32 pos = 0;
                if (valid) pos = hist[d] + (u32)__popcll(m & lanemask_lt());
                WLDS();                                  // cursor reads before cursor updates (LDS only; the stores need not land)
                if (valid) {
                    dst[pos] = key;
                    if ((m >> lane) >> 1 == 0) hist[d] += (u32)__popcll(m);   // highest lane of the digit group advances the cursor
                }
                WLDS();
            }
        }
        WSYNC();                                     // the pass's stores are visible before the next pass reads them
        u64 *t = src; src = dst; dst = t;
    }
    if (src != a) { for (u32 i = lane; i < n; i += 64) a[i] = src[i]; }
    WSYNC();
}

// The same sort by a multi-wave workgroup (k_job_mid / k_job_heavy): every wave counts and places one contiguous slice of
// the keys; per pass the per-wave digit counts are turned into per-wave cursors (bin b of wave w starts behind all smaller
// bins and behind bin b of the waves before it), so equal digits keep their order -- the result is the one of the wave form.
// rh = NW x 256 LDS words, tot = 256, flag = 1.  Every wave of the workgroup calls this; it ends with a workgroup barrier.
template <int NW>
__device__ void radix_sort_block(u64 *a, u64 *alt, u32 n, u32 (*rh)[256], u32 *tot, u32 *flag) {
    int lane = lane_id();
    int wave = (int)(threadIdx.x >> 6);
    u32 tid = threadIdx.x;
    u32 per = (((n + NW - 1) / NW) + 63u) & ~63u;            // slice length, whole chunks of 64
    u32 lo = (u32)wave * per; lo = lo < n ? lo : n;
    u32 hi = lo + per < n ? lo + per : n;
    u32 *hist = rh[wave];
    u64 *src = a, *dst = alt;
    if (tid == 0) *flag = 0;
    for (int pass = 0; pass < 8; pass++) {
        int shift = pass * 8;
        for (int b = lane; b < 256; b += 64) hist[b] = 0;
        WLDS();
        for (u32 i0 = lo; i0 < hi; i0 += 256) {              // four independent loads in flight per lane
            u64 k[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { u32 i = i0 + 64 * u + lane; k[u] = i < hi ? src[i] : 0; }
#pragma unroll
            for (int u = 0; u < 4; u++) if (i0 + 64 * u + lane < hi) atomicAdd(&hist[(k[u] >> shift) & 255], 1u);
        }
        __syncthreads();
        for (u32 b = tid; b < 256; b += NW * 64) {            // (a two-wave workgroup has 128 threads for the 256 bins)
            u32 t = 0;
            for (int w = 0; w < NW; w++) t += rh[w][b];
            tot[b] = t;
            if (t == n) *flag = (u32)pass + 1;                // every key has this digit: the pass is the identity
        }
        __syncthreads();
        if (*flag == (u32)pass + 1) continue;                 // uniform; the next write to flag lies behind the next barrier
        if (wave == 0) {                                      // exclusive prefix over the 256 bins: 4 bins per lane
            u32 c0 = tot[4 * lane], c1 = tot[4 * lane + 1], c2 = tot[4 * lane + 2], c3 = tot[4 * lane + 3];
            u32 s4 = c0 + c1 + c2 + c3;
            u32 ex = wave_incl_scan(s4) - s4;
            tot[4 * lane] = ex; tot[4 * lane + 1] = ex + c0; tot[4 * lane + 2] = ex + c0 + c1; tot[4 * lane + 3] = ex + c0 + c1 + c2;
        }
        __syncthreads();
        for (u32 b = tid; b < 256; b += NW * 64) {
            u32 run = tot[b];
            for (int w = 0; w < NW; w++) { u32 c = rh[w][b]; rh[w][b] = run; run += c; }
        }
        __syncthreads();
        {   // place this wave's slice (radix_sort_wave's loop with the wave's own cursors)
            u64 kn[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { u32 i = lo + 64 * u + lane; kn[u] = i < hi ? src[i] : 0; }
            for (u32 base = lo; base < hi; base += 256) {
                u64 kc[4];
#pragma unroll
                for (int u = 0; u < 4; u++) kc[u] = kn[u];
#pragma unroll
                for (int u = 0; u < 4; u++) { u32 i = base + 256 + 64 * u + lane; kn[u] = i < hi ? src[i] : 0; }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (base + 64 * u >= hi) break;           // uniform within the wave
                    u32 i = base + 64 * u + lane;
                    bool valid = i < hi;
                    u64 key = kc[u];
                    u32 d = (u32)(key >> shift) & 255;
                    u64 m = __ballot(valid);
                    for (int bit = 0; bit < 8; bit++) { u64 bm = __ballot((d >> bit) & 1); m &= ((d >> bit) & 1) ? bm : ~bm; }
                    u32 pos = 0;
                    if (valid) pos = hist[d] + (u32)__popcll(m & lanemask_lt());
                    WLDS();
                    if (valid) {
                        dst[pos] = key;
                        if ((m >> lane) >> 1 == 0) hist[d] += (u32)__popcll(m);
                    }
                    WLDS();
                }
            }
        }
        __syncthreads();                                      // the pass's stores are visible to the waves that read them next
        u64 *t = src; src = dst; dst = t;
    }
    if (src != a) { for (u32 i = tid; i < n; i += NW * 64) a[i] = src[i]; }
    __syncthreads();
}

// Wave-parallel, bit-exact std::sort(anchors, by getAnchorX descending) -- the tie-sensitive sort of
// chainAnchorsHits (pmpfinder.cpp:2465).  Algorithm = ref_sort.h's ref_sort_model: ranges above SORT_SMALL are
// partitioned by the whole wave with the list formulation of std::__unguarded_partition (ballot compaction of the
// "left scan stops here" / "right scan stops here" positions, pair count, parallel swaps); the remaining small
// ranges are independent and are finished one per lane (introsort loop + insertion sort).
#define SORT_SMALL 32
struct XDesc { LNR_HD bool operator()(const u64 &p, const u64 &q) const { return anchor_x(p) > anchor_x(q); } };

__device__ __forceinline__ u64 readlane_u64(u64 v, int l) {
    return ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), l) << 32) | (u64)(u32)__builtin_amdgcn_readlane((int)(u32)v, l);
}
__device__ JOB_INLINE void introsort_xdesc_wave(u64 *a, u32 n, u32 *Lbuf, u32 *Rbuf, u64 *tasks, LeaderScratch *ls /* LDS */, u64 *stage /* free LDS or null */, u32 stage_cap) {
    int lane = lane_id();
    XDesc comp;
    if (n <= SORT_SMALL) {
        if (lane == 0) ref_sort(a, (long)n, comp, ls->st);
        WSYNC();
        return;
    }
    // wave-uniform stack: kept in LDS (every lane writes the same value), not in per-lane private memory
    int *stk_first = ls->st.first, *stk_last = ls->st.last, *stk_depth = ls->st.depth;
    int sp = 0, lg = 0;
    for (u32 t = n; t > 1; t >>= 1) lg++;
    if (lane == 0) { stk_first[0] = 0; stk_last[0] = (int)n; stk_depth[0] = lg * 2; }
    sp = 1;
    u32 ntasks = 0;
    while (sp > 0) {
        --sp;
        WLDS();   // the stack is in LDS; the array itself was ordered by the WSYNC that ended the previous partition
        u32 first = (u32)stk_first[sp], last = (u32)stk_last[sp];
        int depth = stk_depth[sp];
        while (true) {
            if (last - first <= SORT_SMALL) {
                if (lane == 0) tasks[ntasks] = (u64)first | ((u64)last << 28) | ((u64)depth << 56);   // 28 + 28 + 8 bits
                ntasks++;
                break;
            }
            if (depth == 0) {   // depth limit: heap sort of the range (practically never reached).  Deferred to the task pass below:
                                // inlined here, its live ranges lifted the whole job kernel from 96 to 98 VGPRs (5 -> 4 waves per SIMD)
                if (lane == 0) tasks[ntasks] = (u64)first | ((u64)last << 28) | (1ULL << 63);
                ntasks++;
                break;
            }
            --depth;
            // __move_median_to_first(first, first+1, mid, last-1): the four elements are loaded by four lanes at once and the
            // scan below does not wait for the swap's stores -- it knows what a[pick] holds
            u32 iA = first + 1, iB = first + (last - first) / 2, iC = last - 1;
            u64 v4 = 0;
            if (lane < 4) v4 = a[lane == 0 ? iA : lane == 1 ? iB : lane == 2 ? iC : first];
            u64 va = readlane_u64(v4, 0), vb = readlane_u64(v4, 1), vc = readlane_u64(v4, 2), vf = readlane_u64(v4, 3);
            u32 pick;
            if (comp(va, vb)) pick = comp(vb, vc) ? iB : (comp(va, vc) ? iC : iA);
            else pick = comp(va, vc) ? iA : (comp(vb, vc) ? iC : iB);
            u64 vp = pick == iA ? va : (pick == iB ? vb : vc);
            if (lane == 0) { a[first] = vp; a[pick] = vf; }
            u32 xp = (u32)anchor_x(vp), xf = (u32)anchor_x(vf);
            u32 lo = first + 1, nL = 0, nR = 0;
            for (u32 base = lo; base < last; base += 256) {   // four chunks of loads in flight (global memory for long arrays)
                u32 xs4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { u32 i = base + 64 * u + lane; xs4[u] = i < last ? (u32)anchor_x(a[i]) : 0; }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (base + 64 * u >= last) break;       // uniform
                    u32 i = base + 64 * u + lane;
                    bool in = i < last;
                    u32 x = i == pick ? xf : xs4[u];
                    bool fL = in && !(x > xp);    // !comp(a[i], p): the left scan stops here
                    bool fR = in && !(xp > x);    // !comp(p, a[i]): the right scan stops here
                    u64 mL = __ballot(fL), mR = __ballot(fR);
                    if (fL) Lbuf[nL + __popcll(mL & lanemask_lt())] = i;
                    if (fR) Rbuf[nR + __popcll(mR & lanemask_lt())] = i;
                    nL += (u32)__popcll(mL); nR += (u32)__popcll(mR);
                }
            }
            WSYNC();
            u32 lim = nL < nR ? nL : nR, cnt = 0;
            for (u32 k = lane; k < lim; k += 64) cnt += Lbuf[k] < Rbuf[nR - 1 - k] ? 1u : 0u;
            u32 K = wave_sum(cnt);
            for (u32 k = lane; k < K; k += 64) { u32 i = Lbuf[k], j = Rbuf[nR - 1 - k]; u64 t = a[i]; a[i] = a[j]; a[j] = t; }
            u32 cut = last;
            if (K < nL) cut = Lbuf[K];
            if (K >= 1) { u32 r = Rbuf[nR - K]; cut = r < cut ? r : cut; }
            WSYNC();
            if (lane == 0) { stk_first[sp] = (int)cut; stk_last[sp] = (int)last; stk_depth[sp] = depth; }
            ++sp;
            last = cut;
        }
    }
    WSYNC();
    // The deferred ranges, one per lane.  They were emitted left to right (the loop above descends into the left part and
    // pops the adjacent right part next), so 64 consecutive tasks cover one contiguous span of at most 2048 elements: when
    // the array lives in global memory and LDS is free, the span is staged into LDS, sorted there (the insertion sorts are
    // chains of dependent accesses: ~100 instead of ~700 cycles each) and written back.
    for (u32 b = 0; b < ntasks; b += 64) {
        u32 t = b + (u32)lane;
        bool have = t < ntasks;
        u64 v = have ? tasks[t] : 0;
        u32 tl = b + 63 < ntasks ? b + 63 : ntasks - 1;
        u32 span_first = (u32)(tasks[b] & 0xfffffff), span_last = (u32)((tasks[tl] >> 28) & 0xfffffff);
        bool heap_any = __ballot(have && (v >> 63)) != 0;
        u32 span = span_last - span_first;
        if (stage && !heap_any && span <= stage_cap) {
            for (u32 i = lane; i < span; i += 64) stage[i] = a[span_first + i];
            WSYNC();
            if (have) rs_finish_range<16>(stage - span_first, (long)(v & 0xfffffff), (long)((v >> 28) & 0xfffffff), (int)((v >> 56) & 0x7f), comp);
            WSYNC();
            for (u32 i = lane; i < span; i += 64) a[span_first + i] = stage[i];
            WSYNC();
        } else if (have) {
            if (v >> 63) rs_heap_sort(a, (long)(v & 0xfffffff), (long)((v >> 28) & 0xfffffff), comp);
            else rs_finish_range<16>(a, (long)(v & 0xfffffff), (long)((v >> 28) & 0xfffffff), (int)((v >> 56) & 0x7f), comp);   // ranges of <= 32 elements
        }
    }
    WSYNC();
}

// ---- the same sort by a multi-wave workgroup --------------------------------------------------------------------
// What introsort does with a range depends on that range alone (its elements and its depth budget), so ranges can be
// finished in any order and by different waves.  Wave 0 partitions from the top and hands every range of at most n / NW
// elements to a queue in LDS instead of descending into it; after a barrier the waves take ranges from the queue and run
// the single-wave algorithm on them, each with its own stack, its own slice of the staging area and the parts of the
// position lists / task list that lie at the range's own indices (a range of s elements needs at most s entries of each).
#define SORT_DQ 96
struct SortDeferQ { u32 n, next; u32 first[SORT_DQ], last[SORT_DQ]; int depth[SORT_DQ]; };
// introsort of a[first0, last0) with depth budget depth0 by one wave; ranges of at most defer_thresh elements go to q (if any)
__device__ void introsort_xdesc_sub(u64 *a, u32 first0, u32 last0, int depth0, u32 *Lbuf, u32 *Rbuf, u64 *tasks, SortStack *st /* LDS */,
                                    u64 *stage /* free LDS or null */, u32 stage_cap, u32 defer_thresh, SortDeferQ *q) {
    int lane = lane_id();
    XDesc comp;
    if (last0 - first0 <= SORT_SMALL) {
        if (lane == 0) rs_finish_range<16>(a, (long)first0, (long)last0, depth0, comp);
        WSYNC();
        return;
    }
    int *stk_first = st->first, *stk_last = st->last, *stk_depth = st->depth;
    if (lane == 0) { stk_first[0] = (int)first0; stk_last[0] = (int)last0; stk_depth[0] = depth0; }
    int sp = 1;
    u32 ntasks = 0, qn = 0;
    while (sp > 0) {
        --sp;
        WLDS();
        u32 first = (u32)stk_first[sp], last = (u32)stk_last[sp];
        int depth = stk_depth[sp];
        while (true) {
            if (last - first <= SORT_SMALL) {
                if (lane == 0) tasks[ntasks] = (u64)first | ((u64)last << 28) | ((u64)depth << 56);
                ntasks++;
                break;
            }
            if (q && last - first <= defer_thresh && qn < SORT_DQ) {
                if (lane == 0) { q->first[qn] = first; q->last[qn] = last; q->depth[qn] = depth; }
                qn++;
                break;
            }
            if (depth == 0) {
                if (lane == 0) tasks[ntasks] = (u64)first | ((u64)last << 28) | (1ULL << 63);
                ntasks++;
                break;
            }
            --depth;
            u32 iA = first + 1, iB = first + (last - first) / 2, iC = last - 1;
            u64 v4 = 0;
            if (lane < 4) v4 = a[lane == 0 ? iA : lane == 1 ? iB : lane == 2 ? iC : first];
            u64 va = readlane_u64(v4, 0), vb = readlane_u64(v4, 1), vc = readlane_u64(v4, 2), vf = readlane_u64(v4, 3);
            u32 pick;
            if (comp(va, vb)) pick = comp(vb, vc) ? iB : (comp(va, vc) ? iC : iA);
            else pick = comp(va, vc) ? iA : (comp(vb, vc) ? iC : iB);
            u64 vp = pick == iA ? va : (pick == iB ? vb : vc);
            if (lane == 0) { a[first] = vp; a[pick] = vf; }
            u32 xp = (u32)anchor_x(vp), xf = (u32)anchor_x(vf);
            u32 lo = first + 1, nL = 0, nR = 0;
            for (u32 base = lo; base < last; base += 256) {
                u32 xs4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { u32 i = base + 64 * u + lane; xs4[u] = i < last ? (u32)anchor_x(a[i]) : 0; }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (base + 64 * u >= last) break;
                    u32 i = base + 64 * u + lane;
                    bool in = i < last;
                    u32 x = i == pick ? xf : xs4[u];
                    bool fL = in && !(x > xp);
                    bool fR = in && !(xp > x);
                    u64 mL = __ballot(fL), mR = __ballot(fR);
                    if (fL) Lbuf[nL + __popcll(mL & lanemask_lt())] = i;
                    if (fR) Rbuf[nR + __popcll(mR & lanemask_lt())] = i;
                    nL += (u32)__popcll(mL); nR += (u32)__popcll(mR);
                }
            }
            WSYNC();
            u32 lim = nL < nR ? nL : nR, cnt = 0;
            for (u32 k = lane; k < lim; k += 64) cnt += Lbuf[k] < Rbuf[nR - 1 - k] ? 1u : 0u;
            u32 K = wave_sum(cnt);
            for (u32 k = lane; k < K; k += 64) { u32 i = Lbuf[k], j = Rbuf[nR - 1 - k]; u64 t = a[i]; a[i] = a[j]; a[j] = t; }
            u32 cut = last;
            if (K < nL) cut = Lbuf[K];
            if (K >= 1) { u32 r = Rbuf[nR - K]; cut = r < cut ? r : cut; }
            WSYNC();
            if (lane == 0) { stk_first[sp] = (int)cut; stk_last[sp] = (int)last; stk_depth[sp] = depth; }
            ++sp;
            last = cut;
        }
    }
    if (q && lane == 0) q->n = qn;
    WSYNC();
    // the small ranges, one per lane; as many consecutive ones per pass as the staging slice holds
    for (u32 b = 0; b < ntasks;) {
        u32 t = b + (u32)lane;
        bool have = t < ntasks;
        u64 v = have ? tasks[t] : 0;
        u32 span_first = (u32)(tasks[b] & 0xfffffff);
        bool heap_any = __ballot(have && (v >> 63)) != 0;
        u64 fits = __ballot(have && (u32)((v >> 28) & 0xfffffff) - span_first <= stage_cap);   // ends increase with the lane: a prefix
        if (stage && !heap_any && fits) {
            u32 cntf = (u32)__popcll(fits);
            u32 span_last = (u32)((tasks[b + cntf - 1] >> 28) & 0xfffffff);
            u32 span = span_last - span_first;
            for (u32 i = lane; i < span; i += 64) stage[i] = a[span_first + i];
            WSYNC();
            if ((u32)lane < cntf) rs_finish_range<16>(stage - span_first, (long)(v & 0xfffffff), (long)((v >> 28) & 0xfffffff), (int)((v >> 56) & 0x7f), comp);
            WSYNC();
            for (u32 i = lane; i < span; i += 64) a[span_first + i] = stage[i];
            WSYNC();
            b += cntf;
        } else {
            if (have) {
                if (v >> 63) rs_heap_sort(a, (long)(v & 0xfffffff), (long)((v >> 28) & 0xfffffff), comp);
                else rs_finish_range<16>(a, (long)(v & 0xfffffff), (long)((v >> 28) & 0xfffffff), (int)((v >> 56) & 0x7f), comp);
            }
            b += 64;
        }
    }
    WSYNC();
}
// every wave of the workgroup calls this; ends with a workgroup barrier.  stacks = NW sort stacks in LDS.
template <int NW>
__device__ void introsort_xdesc_block(u64 *a, u32 n, u32 *Lbuf, u32 *Rbuf, u64 *tasks, SortStack *stacks, SortDeferQ *q, u64 *stage, u32 stage_cap) {
    int lane = lane_id();
    int wave = (int)(threadIdx.x >> 6);
    if (wave == 0) {
        if (lane == 0) { q->n = 0; q->next = 0; }
        int lg = 0;
        for (u32 t = n; t > 1; t >>= 1) lg++;
        u32 thresh = n / NW; thresh = thresh < 4 * SORT_SMALL ? 4 * SORT_SMALL : thresh;
        introsort_xdesc_sub(a, 0, n, lg * 2, Lbuf, Rbuf, tasks, &stacks[0], stage, stage_cap, thresh, q);
    }
    __syncthreads();
    u32 nq = q->n;
    u32 cap_w = stage_cap / NW;
    u64 *stg_w = (stage && cap_w >= 2 * SORT_SMALL) ? stage + (size_t)wave * cap_w : nullptr;
    for (;;) {
        u32 i = 0;
        if (lane == 0) i = atomicAdd(&q->next, 1u);
        i = (u32)__builtin_amdgcn_readfirstlane((int)i);
        if (i >= nq) break;
        u32 f = q->first[i], l = q->last[i];
        introsort_xdesc_sub(a, f, l, q->depth[i], Lbuf + f, Rbuf + f, tasks + f, &stacks[wave], stg_w, cap_w, 0, nullptr);
    }
    __syncthreads();
}

// wave-parallel twin of filter_anchor_list (filterAnchorsList, pmpfinder.cpp:2025-2071).  The serial loop compares anchor i
// with ak2 = a[(block_str + i - 1) >> 1]: inside a block that only depends on where the block started, so 64 anchors are
// tested at once and the first one that breaks the block is found with a ballot.  Block statistics (count, min / max y of
// the continuing anchors) are kept per lane and reduced when a block closes; accepted blocks are copied forward in place
// (ii <= block_str, chunks read before they are written).  Every lane returns the new count.
__device__ u32 filter_anchor_list_wave(u64 *a, u32 n) {
    int lane = lane_id();
    if (n <= 1) return n;
    u32 ii = 0, bs = 1, i0 = 2;
    u32 cnt = 1;
    u32 y1 = (u32)cord_y(a[1]);
    u32 bmn = y1, bmx = y1;                  // block min / max so far (uniform part: the block's first anchor)
    u32 pmn = 0xffffffffu, pmx = 0;          // per-lane part over the continuing anchors
    while (i0 < n) {
        u32 i = i0 + (u32)lane;
        bool valid = i < n;
        bool cont = false;
        u32 yi = 0;
        if (valid) {
            u64 ai = a[i], ak = a[(bs + i - 1) >> 1];
            yi = (u32)cord_y(ai);
            u32 yk = (u32)cord_y(ak);
            u64 dy2 = (u64)(yi > yk ? yi - yk : yk - yi);
            cont = cord_x40(ai - ak) < (dy2 >> 2);
        }
        u64 brk = __ballot(valid && !cont);
        u32 avail = n - i0 < 64 ? n - i0 : 64;
        u32 take = brk ? (u32)__builtin_ctzll(brk) : avail;      // continuing anchors at the head of this chunk
        if ((u32)lane < take) { pmn = yi < pmn ? yi : pmn; pmx = yi > pmx ? yi : pmx; }
        cnt += take;
        bool closes = brk != 0;
        bool at_end = !closes && i0 + take == n;                  // the last anchor continued: the loop's "i == n - 1" close
        if (closes || at_end) {
            u32 mn = wave_min_u32(pmn), mx = wave_max_u32(pmx);
            mn = mn < bmn ? mn : bmn; mx = mx > bmx ? mx : bmx;
            u32 thd = (mx - mn) >> 10; thd = thd < 2 ? 2 : thd;
            u32 iend = closes ? i0 + take : n - 1;                // the block is emitted as [bs, iend)
            if (cnt > thd) {
                // forward copy in place, no ordering point needed: a chunk's loads return before its stores issue (data
                // dependence), every store lands below the chunk it was read from (ii <= bs), and nothing read later -- the next
                // chunks, the next blocks' anchors and their ak2 -- lies below the current block start
                for (u32 j0 = bs; j0 < iend; j0 += 64) {
                    u32 j = j0 + (u32)lane;
                    u64 v = j < iend ? a[j] : 0;
                    if (j < iend) a[ii + (j - bs)] = v;
                }
                ii += iend - bs;
            }
            if (!closes) break;
            bs = iend;
            u32 yb = (u32)cord_y(a[bs]);
            bmn = yb; bmx = yb; pmn = 0xffffffffu; pmx = 0; cnt = 1;
            i0 = bs + 1;
        } else i0 += 64;
    }
    return ii;
}

// ---- tiled chaining DP (getBestChains, cluster_util.cpp:53-111) ---------------------------------------------
// Anchors are processed in tiles of 64 with ONE ANCHOR PER LANE; predecessors are the uniform operand:
//   before-tile : the predecessors [j_lo(t0), t0) -- their chain scores are final -- are loaded 64 at a time, one per lane
//                 (coalesced), and broadcast one by one out of registers (v_readlane -> SGPR operands); every lane scores
//                 the broadcast predecessor against its own anchor.  No memory access and no cross-lane reduction inside
//                 the 64 steps of a chunk, so the loop runs at VALU rate.  A multi-wave workgroup deals the chunks out
//                 over its waves and merges the per-wave candidates through LDS;
//   in-tile     : step l broadcasts anchor t0+l, whose score is final once steps < l are done, to the lanes k > l;
//   finish      : len / root follow the chosen predecessor: a gather for a before-tile predecessor, pointer jumping over
//                 lane registers for in-tile chains; one coalesced store of the tile's score / len / root / p2 / leaf.
// The reference scans j downwards and keeps the later (smaller) j on equal totals (`>= best`): the candidate key is
// (total << 32 | 0x7fffffff - j), maximised.  Window bounds j_lo(i) depend on x only (dp_window_bounds) and are
// non-decreasing, so j_lo(t0) bounds the whole tile.
#define DP_TILE 64
template <int NW>
struct DpTile {
    // multi-wave workgroups only (kept at one element otherwise: every byte of static LDS costs the single-wave kernel occupancy)
    long long wfar[NW >= 8 ? 2 : 1][NW >= 8 ? NW : 1][NW >= 8 ? DP_TILE : 1];   // per-wave candidates from the predecessors before the previous tile (pipelined form, double buffered)
    long long wnear[NW > 1 ? NW : 1][NW > 1 ? DP_TILE : 1];                     // ... from the previous tile itself (pipelined form) / from all predecessors (flat form)
    i32 tleaf[DP_TILE];
};
// j_lo(i) = min(first j with xs[j] - xs[i] < 300, max(0, i - 20)); xs is non-increasing -> binary search
__device__ __forceinline__ void dp_window_bounds(const u32 *xs, u32 m, i32 *jlo, int tid, int nthreads) {
    for (u32 i = tid; i < m; i += nthreads) {
        u32 lim = xs[i] + 300;
        u32 lo = 0, hi = i;
        while (lo < hi) { u32 mid = (lo + hi) >> 1; if (xs[mid] < lim) hi = mid; else lo = mid + 1; }
        int js = (int)i - 20 < 0 ? 0 : (int)i - 20;
        jlo[i] = (int)lo < js ? (int)lo : js;
    }
}
__device__ __forceinline__ i64 dp_key(int total, int j) { return ((i64)total << 32) | (i64)(u32)(0x7fffffff - j); }
// One predecessor (uniform) against the lane's anchor (lnr_hd.h dp_pair_cand / dp_pair_score).  Most pairs of a repeat-rich
// window are far off the diagonal and score <= 0: the score is only computed when some lane of the wave is a candidate.
// The running best is (total, j) in two registers.  GE = true for scans that visit j in descending order (before the tile:
// the reference keeps the later, smaller j on equal totals), false for the ascending in-tile steps (the earlier j stays).
template <int ST, bool GE>
__device__ __forceinline__ void dp_eval(i32 &btot, i32 &bj, u32 px, u32 py, i32 ps, int jj, u32 xi, u32 yi, bool act) {
    DpPair p;
    bool cand = dp_pair_cand<ST, true>(px, py, xi, yi, p) && act;   // predecessors have the larger (or equal) x
#ifndef LNR_DP_NOSKIP
    if (__ballot(cand) == 0) return;
#endif
    i32 sc;
    if (ST) sc = dp_pair_score<1>(p);
    else {
        // the division is only needed for da >= 10: neighbours on a true diagonal usually stay below that
        u32 sd = 0;
        if (__ballot(cand && p.da >= 10)) sd = dp_pair_sderr(p);
        sc = 100 - (i32)((u32)p.dy / 75u) - (i32)sd;
    }
    i32 tot = sc + ps;
    bool upd = cand && sc > 0 && (GE ? tot >= btot : tot > btot);
    btot = upd ? tot : btot;
    bj = upd ? jj : bj;
}
// before-tile candidates of this wave's share of the chunks; jl = the lane's j_lo (INT_MAX for an idle lane)
template <int ST>
__device__ __forceinline__ i64 dp_before_tile(const u32 *xs, const u32 *ys, const i32 *score, int t0, int lo, u32 xi, u32 yi, int jl, int wave, int nw) {
    int lane = lane_id();
    i32 btot = -1, bj = 0;
    for (int top = t0 - 64 * wave; top > lo; top -= 64 * nw) {   // chunk = predecessors top-1, top-2, ... (at most 64, not below lo)
        int jl_ = top - 1 - lane;
        u32 px = 0, py = 0; i32 ps = 0;
        if (jl_ >= lo) { px = xs[jl_]; py = ys[jl_]; ps = score[jl_]; }
        int cnt = __builtin_amdgcn_readfirstlane(top - lo < 64 ? top - lo : 64);   // scalar loop bound
        for (int s_ = 0; s_ < cnt; s_++) {
            u32 qx = (u32)__builtin_amdgcn_readlane((int)px, s_), qy = (u32)__builtin_amdgcn_readlane((int)py, s_);
            i32 qs = __builtin_amdgcn_readlane(ps, s_);
            int jj = top - 1 - s_;
            dp_eval<ST, true>(btot, bj, qx, qy, qs, jj, xi, yi, jj >= jl);
        }
    }
    return btot < 0 ? (i64)-1 : dp_key(btot, bj);
}
// in-tile steps, finish and store by ONE wave; `best` = merged before-tile candidate of the lane's anchor
template <int ST>
__device__ __forceinline__ void dp_in_tile_finish(i32 *tleaf, int t0, int tn, u32 xi, u32 yi, int jl, i64 best, Rec &r) {
    int lane = lane_id();
    i32 btot = best >= 0 ? (i32)(best >> 32) : -1, bj = best >= 0 ? 0x7fffffff - (i32)(u32)(best & 0xffffffff) : 0;
    for (int l = 0; l + 1 < tn; l++) {
        i32 mysc = btot > 0 ? btot : 0;                  // final for lane l at step l
        u32 qx = (u32)__builtin_amdgcn_readlane((int)xi, l), qy = (u32)__builtin_amdgcn_readlane((int)yi, l);
        i32 qs = __builtin_amdgcn_readlane(mysc, l);
        int jj = t0 + l;
        dp_eval<ST, false>(btot, bj, qx, qy, qs, jj, xi, yi, lane > l && jj >= jl);
    }
    best = btot < 0 ? (i64)-1 : dp_key(btot, bj);
    bool live = lane < tn;
    bool has = live && best >= 0;
    int mj = has ? 0x7fffffff - (int)(u32)(best & 0xffffffff) : -1;
    i32 sc = has ? (i32)(best >> 32) : 0;
    int par = (has && mj >= t0) ? mj - t0 : -1;
    i32 len = 1, root = t0 + lane;
    if (has && mj < t0) { len = r.len[mj] + 1; root = r.root[mj]; r.leaf[mj] = 0; }
    tleaf[lane] = 1;
    __builtin_amdgcn_wave_barrier();
    if (par >= 0) tleaf[par] = 0;
    // chains inside the tile: pointer jumping.  acc = anchors from this one up to (not including) par.
    i32 acc = 1;
    bool unres = par >= 0;
    while (__ballot(unres)) {
        int src = par < 0 ? lane : par;
        int p_par = __shfl(par, src); i32 p_len = __shfl(len, src), p_root = __shfl(root, src), p_acc = __shfl(acc, src);
        int p_unres = __shfl((int)unres, src);
        if (unres) {
            if (!p_unres) { len = acc + p_len; root = p_root; unres = false; }
            else { acc += p_acc; par = p_par; }
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (live) {
        u32 i = (u32)(t0 + lane);
        r.score[i] = sc; r.score2[i] = sc; r.len[i] = len; r.root[i] = root; r.p2[i] = mj; r.leaf[i] = tleaf[lane];
    }
}
template <int ST>
__device__ void best_chains_wave_t(const u32 *xs, const u32 *ys, u32 m, Rec r, i32 *jlo, i32 *tleaf) {
    int lane = lane_id();
    m = (u32)__builtin_amdgcn_readfirstlane((int)m);   // uniform: tile bounds and loop counts stay in scalar registers
    dp_window_bounds(xs, m, jlo, lane, 64);
    WSYNC();
    for (u32 t0 = 0; t0 < m; t0 += DP_TILE) {
        int tn = (int)(m - t0 < DP_TILE ? m - t0 : DP_TILE);
        u32 xi = 0, yi = 0; int jl = 0x7fffffff;
        if (lane < tn) { xi = xs[t0 + lane]; yi = ys[t0 + lane]; jl = jlo[t0 + lane]; }
        int lo = __builtin_amdgcn_readfirstlane(jl);
        i64 best = dp_before_tile<ST>(xs, ys, r.score, (int)t0, lo, xi, yi, jl, 0, 1);
        dp_in_tile_finish<ST>(tleaf, (int)t0, tn, xi, yi, jl, best, r);
        WSYNC();
    }
}
// one-wave driver (k_job)
__device__ void best_chains_wave(const u32 *xs, const u32 *ys, u32 m, Rec r, int score_type, i32 *jlo, i32 *tleaf) {
    if (score_type) best_chains_wave_t<1>(xs, ys, m, r, jlo, tleaf);
    else best_chains_wave_t<0>(xs, ys, m, r, jlo, tleaf);
}

// wave-parallel twin of tb0_scan_serial: first index of the maximal score (> -1), the running maximum seen before
// it (floor -1) and the chain length there.
// (score, first index) key of the maximal score > -1 in [lo, hi), -1 when there is none; every lane returns it
__device__ __forceinline__ i64 tb0_range_best(const Rec &r, u32 lo, u32 hi) {
    int lane = lane_id();
    i64 best = -1;
    for (u32 j0 = lo; j0 < hi; j0 += 256) {   // four independent loads in flight per lane (the scan is latency bound)
        int sc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { u32 j = j0 + 64 * u + (u32)lane; sc[u] = j < hi ? r.score[j] : -1; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            u32 j = j0 + 64 * u + (u32)lane;
            if (sc[u] > -1) { i64 key = ((i64)sc[u] << 32) | (i64)(u32)(0x7fffffff - (int)j); best = key > best ? key : best; }
        }
    }
    return wave_max_i64(best);
}
__device__ __forceinline__ Tb0Scan tb0_scan_result(const Rec &r, i64 best) {
    Tb0Scan s; s.max_score = -1; s.max_2nd = -1; s.max_str = -1; s.max_len = 0;
    if (best < 0) return s;
    s.max_score = (int)(best >> 32);
    s.max_str = 0x7fffffff - (int)(u32)(best & 0xffffffff);
    s.max_len = r.len[s.max_str];
    return s;
}
__device__ Tb0Scan tb0_scan_wave(const Rec &r, u32 n) {   // first index of the maximal score (> -1) and the chain length there; max_2nd is left open
    return tb0_scan_result(r, tb0_range_best(r, 0, n));
}
// maximum of the scores before index `end` (floor -1): the "second best" of traceBackChains0, needed only when a walk runs
// into an element an earlier chain already took
__device__ int tb0_prefix_max_wave(const Rec &r, u32 lo, u32 end) {
    int lane = lane_id();
    i64 m2 = -1;
    for (u32 j0 = lo; j0 < end; j0 += 256) {
        int sc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { u32 j = j0 + 64 * u + (u32)lane; sc[u] = j < end ? r.score[j] : -1; }
#pragma unroll
        for (int u = 0; u < 4; u++) { i64 v = sc[u]; m2 = v > m2 ? v : m2; }
    }
    return (int)wave_max_i64(m2);
}
// Long score arrays (repeat-rich reads: 10^4 anchors, 50 searches): the scores are cut into at most 64 chunks and lane c
// keeps an UPPER BOUND of chunk c's scores.  During traceBackChains0 a score only ever goes down -- it is set to
// delete_score, or to score2 - score2[hit], and a later walk through the same element meets its first deleted element no
// further down the chain than the earlier one did, where score2 is no smaller -- so a bound taken once stays valid and is
// tightened whenever its chunk is rescanned -- except by a rescan made while a walk's elements are provisionally deleted
// (the second-best scan below): those bounds are only kept when the chain is accepted.  A search looks at the first chunk
// with the largest bound; if that chunk's exact maximum equals the bound, no earlier chunk can hold the same score and no
// chunk a larger one.
#define TB0_CHUNKED_MIN 1024
__device__ __forceinline__ u32 tb0_chunk_len(u32 n) { return 256u * ((n + 16383u) / 16384u); }
__device__ i32 tb0_chunk_bounds(const Rec &r, u32 n, u32 C) {
    int lane = lane_id();
    i32 bound = -1;
    u32 c = 0;
    for (u32 lo = 0; lo < n; lo += C, c++) {
        u32 hi = lo + C < n ? lo + C : n;
        i32 mx = tb0_prefix_max_wave(r, lo, hi);
        if ((u32)lane == c) bound = mx;
    }
    return bound;
}
__device__ Tb0Scan tb0_scan_chunked(const Rec &r, u32 n, u32 C, i32 &bound) {
    int lane = lane_id();
    for (;;) {
        i64 top = wave_max_i64(bound > -1 ? (((i64)bound << 32) | (i64)(u32)(0x7fffffff - lane)) : (i64)-1);
        if (top < 0) return tb0_scan_result(r, -1);
        u32 c = (u32)(0x7fffffff - (int)(u32)(top & 0xffffffff));
        u32 lo = c * C, hi = lo + C < n ? lo + C : n;
        i64 best = tb0_range_best(r, lo, hi);
        i32 t = best < 0 ? -1 : (i32)(best >> 32);
        if ((u32)lane == c) bound = t;
        if (t == (i32)(top >> 32)) return tb0_scan_result(r, best);
    }
}
// exact maximum of the scores in [0, end) (floor -1) from the chunk bounds: the partial chunk is scanned, a whole chunk only
// while its bound still exceeds what has been found
__device__ int tb0_prefix_max_chunked(const Rec &r, u32 end, u32 C, i32 &bound) {
    int lane = lane_id();
    u32 cend = end / C;                                   // chunks [0, cend) lie wholly before `end`
    int m2 = tb0_prefix_max_wave(r, cend * C, end);
    for (;;) {
        i64 top = wave_max_i64(((u32)lane < cend && bound > m2) ? (((i64)bound << 32) | (i64)(u32)lane) : (i64)-1);
        if (top < 0) return m2;
        u32 c = (u32)(top & 0xffffffff);
        i32 t = tb0_prefix_max_wave(r, c * C, c * C + C);
        if ((u32)lane == c) bound = t;
        m2 = t > m2 ? t : m2;
    }
}
// wave-parallel twin of traceback1_table: lanes over the anchors, the (<= 50) trees in LDS.  Trees are numbered by their
// first leaf: per chunk of 64 anchors the not-yet-listed roots are appended lowest lane first.  The best leaf of a tree is
// the maximum of (score, earliest j) -- an LDS atomic max on a 64-bit key.
__device__ int traceback1_table_wave(const Rec &r, u32 n, LeaderScratch *ls) {
    int lane = lane_id();
    int nl = 0;
    unsigned long long *key = (unsigned long long *)ls->ranks;
    if (lane < 64) key[lane] = 0;
    WSYNC();
    for (u32 base = 0; base < n; base += 64) {
        u32 j = base + (u32)lane;
        bool lf = j < n && r.leaf[j] != 0;
        i32 root = lf ? r.root[j] : -1;
        int k = -1;
        if (lf) for (int q = 0; q < nl; q++) if (ls->l_root[q] == root) { k = q; break; }
        u64 pend = __ballot(lf && k < 0);
        while (pend) {
            int src = (int)__builtin_ctzll(pend);
            i32 rt = __shfl(root, src);
            if (nl < 64) { if (lane == 0) ls->l_root[nl] = rt; }
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
    if (lane < nl) {
        unsigned long long kv = key[lane];
        u32 j = 0xffffffffu - (u32)(kv & 0xffffffffu);
        ls->l_score[lane] = (i32)(u32)(kv >> 32) - 0x40000000; ls->l_len[lane] = r.len[j]; ls->l_leaf[lane] = (i32)j;
    }
    WSYNC();
    return nl;
}
// traceBackChains (cluster_util.cpp:306-335) for the anchor DP, lanes cooperating on the scans; lane 0 walks chains
// and emits hits.  s_flag = four LDS words.
// A chain is walked by the leader (one dependent load per element: p2) and then emitted by all lanes: hit words and chain
// scores are independent gathers.  Same effect as AnchorSink::emit; every lane keeps the sink's counters in step.
struct DeferSink { u32 nchains, first_len, pending; LNR_HD void emit(const i32 *, const i32 *, u32 n) { pending = n; } };
__device__ void emit_chain_wave(AnchorSink &sink, const Rec &r, const i32 *chain, u32 cn) {
    int lane = lane_id();
    Vec<u64> &H = *sink.hits; Vec<i32> &HS = *sink.hscore;
    u32 base = H.n, sbase = HS.n;
    u32 fit = cn, sfit = cn;
    if (base + cn > H.cap) { fit = H.cap - base; if (lane == 0) *H.ovf = 1; }
    if (sbase + cn > HS.cap) { sfit = HS.cap - sbase; if (lane == 0) *HS.ovf = 1; }
    for (u32 k = lane; k < cn; k += 64) {
        i32 idx = chain[k];
        if (k < fit) H.p[base + k] = hit2cord(sink.anchors[idx]) | (k + 1 == fit ? F_END : 0);
        if (k < sfit) HS.p[sbase + k] = r.score2[idx];
    }
    H.n = base + fit; HS.n = sbase + sfit;
    if (sink.nchains == 0) sink.first_len = cn;
    sink.nchains++;
    WSYNC();
}
__device__ JOB_INLINE void traceback_anchor_wave(Rec r, u32 n, AnchorSink &sink, i32 *chain, i32 *chain_sc, i32 *cnt, int *s_flag, LeaderScratch *ls) {
    int lane = lane_id();
    for (u32 i = lane; i < n; i += 64) cnt[i] = 0;
    WSYNC();
    for (u32 i = lane; i < n; i += 64) cnt[r.root[i]] = 1;
    WSYNC();
    u32 c = 0;
    for (u32 i = lane; i < n; i += 64) c += (u32)cnt[i];
    u32 root_num = wave_sum(c);
    const int min_len = 1, abort_score = 45, bestn = 50;
    if (root_num > 50) {
        // traceBackChains0 (cluster_util.cpp:113-211), one search per iteration.  The reference scans for the best score AND
        // the best score before it; the second is only consulted when the walk meets an element that an earlier chain took,
        // so it is computed then -- from the scores after this walk's deletions plus the largest score the walk deleted
        // (every deleted element lies before max_str), which is the value the up-front scan would have returned.
        const int delete_score = -1000;
        const bool chunked = n > TB0_CHUNKED_MIN;
        const u32 C = tb0_chunk_len(n);
        i32 bound = chunked ? tb0_chunk_bounds(r, n, C) : -1;
        for (int it = 0; it < 50; it++) {
            Tb0Scan sc = chunked ? tb0_scan_chunked(r, n, C, bound) : tb0_scan_wave(r, n);
            int max_score = sc.max_score, max_str = sc.max_str, max_len = sc.max_len;
            bool f_done = max_str == -1;
            if (sink.nchains) { if ((float)max_len > (float)sink.first_len * 0.0f) f_done = false; }
            if (f_done || max_score == 0) break;
            bool walk = max_len > min_len && max_score / (max_len - 1) > abort_score;
            u32 cn = 0;
            if (walk) {
                if (lane == 0) {
                    int hit = -1, m_del = -1;
                    u32 c2 = 0;
                    for (int j = max_str; j != -1; j = r.p2[j]) {
                        int sj = r.score[j];
                        if (sj != delete_score) {
                            chain[c2++] = j;
                            if (j != max_str && sj > m_del) m_del = sj;
                            r.score[j] = delete_score;
                        } else { hit = j; break; }
                    }
                    s_flag[0] = (int)c2; s_flag[1] = hit; s_flag[2] = m_del;
                }
                WSYNC();
                cn = (u32)s_flag[0];
                int hit = s_flag[1], m_del = s_flag[2];
                WSYNC();
                if (hit >= 0) {
                    // the bounds this scan tightens see the walk's elements as deleted: they only become the lane's bounds when
                    // the chain is accepted -- a rejected walk puts (smaller) scores back
                    i32 b2 = bound;
                    int m2 = chunked ? tb0_prefix_max_chunked(r, (u32)max_str, C, b2) : tb0_prefix_max_wave(r, 0, (u32)max_str);
                    int max_2nd = m2 > m_del ? m2 : m_del;
                    int infix = r.score2[hit];
                    if (max_score - infix < max_2nd) {
                        if (lane == 0) for (int k = max_str; k != hit; k = r.p2[k]) r.score[k] = r.score2[k] - infix;
                        cn = 0;
                        WSYNC();
                    } else bound = b2;
                }
                if (cn) emit_chain_wave(sink, r, chain, cn);
            }
            if (max_str != -1) { if (lane == 0) r.score[max_str] = delete_score; WSYNC(); }
        }
    } else {
        int nl = traceback1_table_wave(r, n, ls);
        if (lane == 0) {   // trees by score (std::sort order, cluster_util.cpp:269)
            for (int i = 0; i < nl; i++) ls->ranks[i] = ((u64)(u32)ls->l_score[i] << 32) | (u32)i;
            ref_sort(ls->ranks, (long)nl, [](const u64 &a, const u64 &b) { return (i32)(a >> 32) > (i32)(b >> 32); }, ls->st);
        }
        WSYNC();
        int lim = bestn < nl ? bestn : nl;
        for (int i = 0; i < lim; i++) {
            int t = (int)(u32)ls->ranks[i];
            int max_score = ls->l_score[t], max_len = ls->l_len[t], max_str = ls->l_leaf[t];
            int mean = max_len > 1 ? max_score / (max_len - 1) : abort_score + 1;
            if (max_len > min_len && mean > abort_score) {
                if (lane == 0) {
                    u32 cn = 0;
                    for (int j = max_str; j != -1; j = r.p2[j]) chain[cn++] = j;
                    *s_flag = (int)cn;
                }
                WSYNC();
                u32 cn = (u32)*s_flag;
                WSYNC();
                // (the f_stop test of the serial form compares against stop_ratio = 0 here and never fires)
                if (cn) emit_chain_wave(sink, r, chain, cn);
            }
        }
    }
}

// ---- the small tie-sensitive sorts of the block phases (a11-a13).  The arrays live in the job's global scratch, where lane 0's serial
// emulation of std::sort pays a memory round trip per comparison.  Up to 16 elements std::sort IS a plain insertion sort, i.e. stable: every
// lane ranks its element by counting (elements before it in the order + equal ones with a smaller index).  Longer arrays are staged in the
// LDS tree tables (idle between the tracebacks), sorted there by lane 0 with the exact emulation, and copied back.
#define JOB_STAGE_CAP 192      // l_root .. l_leaf (256 words) + ranks (64 u64), contiguous in LeaderScratch
__device__ __forceinline__ u64 *job_stage_of(LeaderScratch &ls) {
    static_assert(offsetof(LeaderScratch, l_root) % 8 == 0 && offsetof(LeaderScratch, ranks) == offsetof(LeaderScratch, l_root) + 1024, "stage = tree tables + ranks");
    return (u64 *)ls.l_root;
}
template <class Comp>
__device__ JOB_INLINE void small_sort_u64_wave(u64 *keys, u32 n, Comp comp, LeaderScratch &ls) {
    int lane = lane_id();
    if (n < 2) return;
    if (n <= 16) {
        u64 mine = (u32)lane < n ? keys[lane] : 0;
        u32 rank = 0;
        for (u32 t = 0; t < n; t++) { u64 o = __shfl(mine, (int)t); rank += (comp(o, mine) || (!comp(mine, o) && t < (u32)lane)) ? 1u : 0u; }
        WSYNC();
        if ((u32)lane < n) keys[rank] = mine;
        WSYNC();
        return;
    }
    if (n <= JOB_STAGE_CAP) {
        u64 *stage = job_stage_of(ls);
        for (u32 i = lane; i < n; i += 64) stage[i] = keys[i];
        WSYNC();
        if (lane == 0) ref_sort(stage, (long)n, comp, ls.st);
        WSYNC();
        for (u32 i = lane; i < n; i += 64) keys[i] = stage[i];
        WSYNC();
        return;
    }
    if (lane == 0) ref_sort(keys, (long)n, comp, ls.st);
    WSYNC();
}
// gather_blocks(hits, nh, nullptr, sep, 1, nh, L, 600, 0, 0) (pmpfinder.cpp:1484-1530) with all lanes: a block ends before hit i when hit i - 1
// closes a chain or the two are not consecutive; boundaries are compacted in index order.  Returns the number of blocks.
__device__ JOB_INLINE u32 gather_blocks_wave(const u64 *hits, u32 nh, UP *sep, u32 sep_cap, int *ovf) {
    if (nh < 2) return 0;
    int lane = lane_id();
    u32 nbd = 0;
    for (u32 base = 2; base < nh; base += 64) {
        u32 i = base + (u32)lane;
        bool b = false;
        if (i < nh) { u64 c0 = hits[i - 1], c1 = hits[i]; b = is_end(c0) || !consecutive(c0, c1, 600); }
        u64 m = __ballot(b);
        if (b) {
            u32 pos = nbd + (u32)__popcll(m & lanemask_lt());
            if (pos < sep_cap) sep[pos].second = i;
            if (pos + 1 < sep_cap) sep[pos + 1].first = i;
        }
        nbd += (u32)__popcll(m);
    }
    u32 nb = nbd + 1;
    if (nb > sep_cap) { if (lane == 0) *ovf = 1; nb = sep_cap; }
    if (lane == 0) { sep[0].first = 1; if (nbd < sep_cap) sep[nbd].second = nh; }
    WSYNC();
    return nb;
}
// chain_blocks_prepare (f_sort = 1) with all lanes: keys, the tie-sensitive sort, the two gathers
__device__ JOB_INLINE void chain_blocks_prepare_wave(const u64 *records, const UP *sep, const i32 *sep_score, u32 nb, BlockScratch s) {
    int lane = lane_id();
    u64 *e = (u64 *)s.sep_tmp;
    for (u32 i = lane; i < nb; i += 64) e[i] = (cord_x40(records[sep[i].first]) << 24) | (u64)i;
    WSYNC();
    small_sort_u64_wave(e, nb, [](const u64 &a, const u64 &b) { return (a >> 24) > (b >> 24); }, *s.ls);
    for (u32 i = lane; i < nb; i += 64) s.ptr[i] = (u32)(e[i] & 0xffffffu);
    WSYNC();
    for (u32 i = lane; i < nb; i += 64) { u32 q = s.ptr[i]; s.sep_tmp[i] = sep[q]; s.score_tmp[i] = sep_score[q]; }
    WSYNC();
}
// filter_blocks_hits (_filterBlocksHits cluster_util.cpp:633-719) with all lanes: the chains' block ranges are copied 64 hits at a time.
// nchains = the sink's chain count (broadcast by the caller: the sink itself was filled by lane 0).
__device__ JOB_INLINE u32 filter_blocks_hits_wave(const BlockSink &ch, u32 nchains, const u64 *hits, u64 *out) {
    if (nchains == 0) return 0xffffffffu;   // untouched
    int lane = lane_id();
    u32 n = 0, major_n = 1;
    float bound = 0.0f;
    for (u32 c = 0; c < nchains; c++) {
        i32 e0 = ch.off[c], e1 = ch.off[c + 1];
        u64 len_current = 0;
        for (i32 eb = e0; eb < e1; eb += 64) {
            i32 e = eb + lane;
            u32 l = e < e1 ? (u32)(ch.el[e].second - ch.el[e].first) : 0u;
            len_current += wave_sum(l);
        }
        bool append = true;
        if (c == 0) bound = 0.8 * len_current;
        else { append = major_n < 5 && (float)len_current > bound; if (append) ++major_n; }
        if (!append) continue;
        for (i32 eb = e0; eb < e1; eb += 64) {
            i32 e = eb + lane;
            u64 qf = 0, qs = 0;
            if (e < e1) { UP q = ch.el[e]; qf = q.first; qs = q.second; }
            int cnt = e1 - eb < 64 ? e1 - eb : 64;
            for (int t = 0; t < cnt; t++) {
                u64 f = __shfl(qf, t), sd = __shfl(qs, t);
                bool last_range = eb + t + 1 == e1;
                for (u64 k = f + (u64)lane; k < sd; k += 64) out[n + (u32)(k - f)] = (hits[k] & ~F_END) | ((last_range && k + 1 == sd) ? F_END : 0);
                n += (u32)(sd - f);
            }
        }
    }
    WSYNC();
    return n;
}

// wave-parallel twin of prefilter_chains2 (pmpfinder.cpp:2366-2446).  Cuts are visited in their (tie-sensitive) sorted
// order; for one cut the blocks are independent, so lanes take one block each.  The pieces are re-sorted by their unique
// end position afterwards, so the order in which lanes append them does not matter.  The j-loop's early stop
// ("xy_strs[j] < length(hits)" in the loop condition) is the first block whose cursor reached the end of hits at the
// start of the pass.  `sep` is the leader's Vec (its .n is broadcast through s_n); every lane returns the new count.
__device__ JOB_INLINE u32 prefilter_chains2_wave(u64 *hits, u32 nhits, Vec<UP> &sep, u32 nb, u64 *cuts, u64 *xy_strs, Vec<UP> &tmp, LeaderScratch &ls) {
    int lane = lane_id();
    const u64 mask = 1ULL << 62;
    UP *sp = sep.p;
    // cuts are sorted by the y of the hit they point at; the key travels with the element (y << 32 | end flag << 31 | hit
    // index) so that the tie-sensitive sort compares array elements only -- same comparator results, same permutation
    for (u32 i = lane; i < nb; i += 64) {
        u64 f = sp[i].first, l = sp[i].second - 1;
        cuts[2 * i] = (cord_y(hits[f]) << 32) | f;
        cuts[2 * i + 1] = (cord_y(hits[l]) << 32) | (1ULL << 31) | l;
        xy_strs[i] = f;
    }
    WSYNC();
    small_sort_u64_wave(cuts, 2 * nb, [](const u64 &a, const u64 &b) { return (a >> 32) < (b >> 32); }, ls);
    UP *tp = tmp.p;
    u32 ntmp = 0, tcap = tmp.cap;
    if (nb <= 64) {
        // The usual case: one block per lane, its cursor, end and the y at the cursor live in registers; a cut is a packed
        // element (y in the upper half), so a pass touches memory only where a block really advances.  No store of one lane
        // is read by another, hence no ordering point inside the loop.
        bool mine = (u32)lane < nb;
        u64 lower = mine ? xy_strs[lane] : ~0ULL, kend = mine ? sp[lane].second : 0;
        u64 ylow = (mine && lower < nhits) ? cord_y(hits[lower]) : 0;
        for (u32 i = 0; i < 2 * nb; i++) {
            u64 e = cuts[i];
            u64 cuty = e >> 32;
            bool is_end = ((e >> 31) & 1) != 0;
            u64 mm = __ballot(mine && lower >= nhits);
            u32 jstar = mm ? (u32)__builtin_ctzll(mm) : nb;   // first block whose cursor already sits at the end of hits
            bool emit = false;
            UP piece; piece.first = 0; piece.second = 0;
            if ((u32)lane < jstar && !(cuty < ylow)) {
                // the reference scans k upwards for the first hit with y >= cuty; inside a block y never decreases (gather_blocks keeps
                // consecutive() hits together), so that hit is a lower bound -- a bisection instead of a scan of the block per cut
                u64 lo = lower, hi_ = kend;
                while (lo < hi_) {
                    u64 mid = (lo + hi_) >> 1;
                    u64 ky = mid == lower ? ylow : cord_y(hits[mid]);
                    if (ky >= cuty) hi_ = mid; else lo = mid + 1;
                }
                if (lo < kend) {
                    u64 ky = lo == lower ? ylow : cord_y(hits[lo]);
                    u64 upper = (is_end && ky == cuty) ? lo + 1 : lo;
                    if (lower != upper) { emit = true; piece.first = lower; piece.second = upper; lower = upper; ylow = lower < nhits ? cord_y(hits[lower]) : 0; }
                }
            }
            u64 em = __ballot(emit);
            if (emit) { u32 pos = ntmp + (u32)__popcll(em & lanemask_lt()); if (pos < tcap) tp[pos] = piece; }
            ntmp += (u32)__popcll(em);
        }
        WSYNC();
    } else {
    for (u32 i = lane; i < 2 * nb; i += 64) { u64 e = cuts[i]; cuts[i] = (e & 0x7fffffffULL) | ((e >> 31) & 1 ? mask : 0); }   // back to index | end mask
    WSYNC();
    for (u32 i = 0; i < 2 * nb; i++) {
        u64 cut = cuts[i];
        u64 cuty = cord_y(hits[cut & ~mask]);
        bool is_end = (cut & mask) != 0;
        u32 jstar = nb;   // first block whose cursor already sits at the end of hits: the reference's j-loop stops there
        for (u32 jb = 0; jb < nb; jb += 64) {
            u32 j = jb + lane;
            u64 mm = __ballot(j < nb && xy_strs[j] >= nhits);
            if (mm) { jstar = jb + (u32)__builtin_ctzll(mm); break; }
        }
        for (u32 jb = 0; jb < jstar; jb += 64) {
            u32 j = jb + lane;
            bool emit = false;
            UP piece; piece.first = 0; piece.second = 0;
            if (j < jstar) {
                u64 lower = xy_strs[j];
                if (!(cuty < cord_y(hits[lower]))) {
                    u64 kend = sp[j].second;
                    u64 lo = lower, hi_ = kend;          // (bisection for the first hit with y >= cuty, as above)
                    while (lo < hi_) { u64 mid = (lo + hi_) >> 1; if (cord_y(hits[mid]) >= cuty) hi_ = mid; else lo = mid + 1; }
                    if (lo < kend) {
                        u64 upper = (is_end && cord_y(hits[lo]) == cuty) ? lo + 1 : lo;
                        if (lower != upper) { emit = true; piece.first = lower; piece.second = upper; xy_strs[j] = upper; }
                    }
                }
            }
            u64 em = __ballot(emit);
            if (emit) { u32 pos = ntmp + (u32)__popcll(em & lanemask_lt()); if (pos < tcap) tp[pos] = piece; }
            ntmp += (u32)__popcll(em);
        }
        WSYNC();
    }
    }
    if (ntmp > tcap) { if (lane == 0) *tmp.ovf = 1; WSYNC(); return 0; }   // (nothing of hits / sep is rewritten: the caller gives up or retries with larger arrays)
    if (ntmp <= 64) {
        // the pieces end at distinct positions: any correct sort gives the reference's order -- ranks by counting
        UP q; q.first = 0; q.second = ~0ULL;
        if ((u32)lane < ntmp) q = tp[lane];
        u32 rank = 0;
        for (u32 t = 0; t < ntmp; t++) { u64 o = __shfl(q.second, (int)t); rank += o < q.second ? 1u : 0u; }
        if ((u32)lane < ntmp) sp[rank] = q;
        WSYNC();
    } else {
        for (u32 i = lane; i < ntmp; i += 64) sp[i] = tp[i];
        WSYNC();
        if (lane == 0) ref_sort(sp, (long)ntmp, [](const UP &a, const UP &b) { return a.second < b.second; }, ls.st);
        WSYNC();
    }
    for (u32 i = lane; i < ntmp; i += 64) hits[sp[i].second - 1] |= F_END;
    WSYNC();
    return ntmp;
}
// wave-parallel twin of best_chains2 with getApxChainScore2 (cluster_util.cpp:469-526,586-631): serial over blocks, lanes
// over the <= 20 predecessors; among equal totals the LAST predecessor wins (ascending scan with >=).
__device__ void best_chains2_wave(const u64 *hits, const UP *sep, const i32 *sep_score, u32 nb, Rec r, unsigned long long *dbg = nullptr) {
    int lane = lane_id();
    unsigned long long ta = 0, tb = 0, tc = 0, t0_ = 0, t1_ = 0, t2_ = 0;
    (void)ta; (void)tb; (void)tc; (void)t0_; (void)t1_; (void)t2_;
    for (u32 i = 0; i < nb; i++) {
#ifdef LNR_PROF
        t0_ = clock64();
#endif
        int j_str = (int)i - 20 < 0 ? 0 : (int)i - 20;
        int j = j_str + lane;
        i64 best = -1;
        i32 si = sep_score[i];
        if (j < (int)i) {
            int sc = block_score2(hits[sep[j].first], hits[sep[i].second - 1]);
            if (sc > 0) best = ((i64)(sc + r.score[j] + si) << 32) | (i64)(u32)j;   // larger j wins ties
        }
#ifdef LNR_PROF
        t1_ = clock64(); ta += t1_ - t0_;
#endif
        best = wave_max_i64(best);
#ifdef LNR_PROF
        t2_ = clock64(); tb += t2_ - t1_;
#endif
        if (lane == 0) {
            int tot = best >= 0 ? (int)(best >> 32) : -1;
            i32 li = (i32)(sep[i].second - sep[i].first);
            if (tot > 0) {
                int mj = (int)(u32)(best & 0xffffffff);
                r.p2[i] = mj; r.score[i] = tot; r.len[i] = li + r.len[mj]; r.score2[i] = tot;
                r.root[i] = r.root[mj]; r.leaf[i] = 1; r.leaf[mj] = 0;
            } else {
                r.p2[i] = -1; r.score[i] = si; r.len[i] = li; r.score2[i] = si; r.root[i] = (i32)i; r.leaf[i] = 1;
            }
        }
        WSYNC();
#ifdef LNR_PROF
        tc += clock64() - t2_;
#endif
    }
#ifdef LNR_PROF
    if (dbg) { atomicAdd(&dbg[26], ta); atomicAdd(&dbg[29], tb); atomicAdd(&dbg[30 - 3], tc); }
#endif
}

// Workgroup form of the tiled DP for the multi-wave kernels: the before-tile chunks are dealt out over the NW waves,
// wave 0 merges the candidates and runs the in-tile steps.  Every thread of the workgroup calls it (workgroup barriers).
// candidates of this wave's share of ONE chunk of predecessors [top - cnt, top): the steps are dealt over the waves
template <int ST>
__device__ __forceinline__ i64 dp_chunk_by_steps(const u32 *xs, const u32 *ys, const i32 *score, int top, int cnt, u32 xi, u32 yi, int jl, int wave, int nw) {
    int lane = lane_id();
    i32 btot = -1, bj = 0;
    int jl_ = top - 1 - lane;
    u32 px = 0, py = 0; i32 ps = 0;
    if (lane < cnt) { px = xs[jl_]; py = ys[jl_]; ps = score[jl_]; }
    for (int s_ = wave; s_ < cnt; s_ += nw) {
        u32 qx = (u32)__builtin_amdgcn_readlane((int)px, s_), qy = (u32)__builtin_amdgcn_readlane((int)py, s_);
        i32 qs = __builtin_amdgcn_readlane(ps, s_);
        int jj = top - 1 - s_;
        dp_eval<ST, true>(btot, bj, qx, qy, qs, jj, xi, yi, jj >= jl);
    }
    return btot < 0 ? (i64)-1 : dp_key(btot, bj);
}
// Workgroup form of the tiled DP for the multi-wave kernels, software pipelined over the tiles: while wave 0 runs the
// in-tile steps of tile k (the serial part), the other waves already score tile k+1 against the predecessors that lie before
// tile k (final by then); after a barrier all waves share the one remaining chunk -- tile k itself -- by steps.  Wave 0 then
// merges the candidates and goes on with tile k+1.  Every thread of the workgroup calls it (two barriers per tile).
template <int NW, int ST>
__device__ void best_chains_block_t(const u32 *xs, const u32 *ys, u32 m, Rec r, i32 *jlo, DpTile<NW> &T) {
    int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    dp_window_bounds(xs, m, jlo, tid, NW * 64);
    __syncthreads();
    u32 ntiles = (m + DP_TILE - 1) / DP_TILE;
    for (u32 k = 0; k < ntiles; k++) {
        u32 t0 = k * DP_TILE;
        int tn = (int)(m - t0 < DP_TILE ? m - t0 : DP_TILE);
        bool more = k + 1 < ntiles;
        u32 t1 = t0 + DP_TILE;                                  // next tile
        int tn1 = more ? (int)(m - t1 < DP_TILE ? m - t1 : DP_TILE) : 0;
        u32 x1 = 0, y1 = 0; int jl1 = 0x7fffffff;
        if (more && lane < tn1) { x1 = xs[t1 + lane]; y1 = ys[t1 + lane]; jl1 = jlo[t1 + lane]; }
        if (wave == 0) {
            u32 xi = 0, yi = 0; int jl = 0x7fffffff;
            if (lane < tn) { xi = xs[t0 + lane]; yi = ys[t0 + lane]; jl = jlo[t0 + lane]; }
            i64 best = -1;
            if (k > 0) {
#pragma unroll
                for (int w = 1; w < NW; w++) { i64 v = T.wfar[k & 1][w][lane]; best = v > best ? v : best; }
#pragma unroll
                for (int w = 0; w < NW; w++) { i64 v = T.wnear[w][lane]; best = v > best ? v : best; }
            }
            dp_in_tile_finish<ST>(T.tleaf, (int)t0, tn, xi, yi, jl, best, r);
        } else if (more) {
            int lo1 = __builtin_amdgcn_readfirstlane(jl1);
            T.wfar[(k + 1) & 1][wave][lane] = dp_before_tile<ST>(xs, ys, r.score, (int)t0, lo1, x1, y1, jl1, wave - 1, NW - 1);
        }
        __syncthreads();   // tile k is final and visible; the far candidates of tile k+1 are stored
        if (more) T.wnear[wave][lane] = dp_chunk_by_steps<ST>(xs, ys, r.score, (int)t0 + tn, tn, x1, y1, jl1, wave, NW);
        __syncthreads();
    }
}
// The un-pipelined form: all waves share the predecessor chunks, then wave 0 runs the in-tile steps.  Measured better for
// the 4-wave kernel (three helper waves cannot hide the chunks behind the in-tile steps; round 0: 27.8 vs 29.0 ms).
template <int NW, int ST>
__device__ void best_chains_block_flat(const u32 *xs, const u32 *ys, u32 m, Rec r, i32 *jlo, DpTile<NW> &T) {
    int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    dp_window_bounds(xs, m, jlo, tid, NW * 64);
    __syncthreads();
    for (u32 t0 = 0; t0 < m; t0 += DP_TILE) {
        int tn = (int)(m - t0 < DP_TILE ? m - t0 : DP_TILE);
        u32 xi = 0, yi = 0; int jl = 0x7fffffff;
        if (lane < tn) { xi = xs[t0 + lane]; yi = ys[t0 + lane]; jl = jlo[t0 + lane]; }
        int lo = __builtin_amdgcn_readfirstlane(jl);
        T.wnear[wave][lane] = dp_before_tile<ST>(xs, ys, r.score, (int)t0, lo, xi, yi, jl, wave, NW);
        __syncthreads();
        if (wave == 0) {
            i64 best = -1;
#pragma unroll
            for (int w = 0; w < NW; w++) { i64 v = T.wnear[w][lane]; best = v > best ? v : best; }
            dp_in_tile_finish<ST>(T.tleaf, (int)t0, tn, xi, yi, jl, best, r);
        }
        __syncthreads();
    }
}
template <int NW>
__device__ void best_chains_block(const u32 *xs, const u32 *ys, u32 m, Rec r, int score_type, i32 *jlo, DpTile<NW> &T) {
    if (NW >= 8) {
        if (score_type) best_chains_block_t<NW, 1>(xs, ys, m, r, jlo, T);
        else best_chains_block_t<NW, 0>(xs, ys, m, r, jlo, T);
    } else {
        if (score_type) best_chains_block_flat<NW, 1>(xs, ys, m, r, jlo, T);
        else best_chains_block_flat<NW, 0>(xs, ys, m, r, jlo, T);
    }
}

// Where a job's chained hits wait for k_post: the last 12 x cap bytes (16-byte aligned) of its global scratch region.  The
// region holds 160 x cap + 1 KB (job_scratch_bytes); the job kernel's own carving stays below 140 x cap and k_post's below
// 124 x cap, both from the front.
#ifndef POST_MAX_HITS
#define POST_MAX_HITS 128
#endif
struct PostIn { u64 *hits; i32 *hscore; };
__device__ __forceinline__ PostIn post_in_of(char *region, u32 cap) {
    u64 bytes = (job_scratch_bytes(cap) + 255) & ~255ULL;
    u64 need = ((u64)cap * 12 + 15) & ~15ULL;
    PostIn p; p.hits = (u64 *)(region + bytes - need); p.hscore = (i32 *)(p.hits + cap);
    return p;
}
// Replays the allocation sequence of the pre phase (global scratch only) from the two counts it left in jstate: n1 = anchors
// after binning, m = anchors after the list filter.  Same calls in the same order -> same pointers.
__device__ __forceinline__ void job_replay(const JobArgs &A, u32 j, u32 n1, u32 m, u32 *dyn_lds, Arena &slow, Arena &ar, u64 *&a, JobScratch &S, int *ovf) {
    u32 cap = A.n_anchors[j] + 2;
    slow.init(A.scratch + A.scr_off[j], job_scratch_bytes(cap));
    ar.init((void *)dyn_lds, A.arena_lds); ar.next = &slow;
    a = A.anchors + A.anc_off[j];
    (void)slow.get<u64>(cap);
    if (n1 > 1) a = ar.get<u64>((u64)n1 + 2);
    if (m > 1) (void)slow.get<u64>((u64)m + 2);
    (void)job_carve(ar, m, S, ovf);
}
// One workgroup per read: runs the read's jobs in order (round 0: the whole read; remap round: its gaps), appending
// cords to the read's cord list exactly like consecutive apxMap_ calls do.
//   NW == 1 : one wave does everything (the bulk of the reads).
//   NW == 4 / 16: reads with many anchors.  Wave 0 runs the serial / wave-parallel phases with a large LDS arena; the other
//             waves join for the radix sort (radix_sort_block), the x-descending sort (introsort_xdesc_block) and the chaining
//             DP (best_chains_block) and otherwise wait at the workgroup barriers that frame those three phases.
struct DpShare { const u32 *xs, *ys; Rec rec; i32 *jlo; u32 m; int score_type; int abort; };
struct RadixShare { u64 *a, *alt; u32 n; };
#ifndef RADIX_BLOCK_MIN
#define RADIX_BLOCK_MIN 2048   // shortest array the workgroup forms are used for (tools/stress_parity.py runs a build with 64)
#endif
struct SortShare { u64 *a; u32 *L, *R; u64 *tasks, *stg; u32 stg_cap, n; };
#ifndef SORT_BLOCK_MIN
#define SORT_BLOCK_MIN 2048
#endif
// PHASE: 0 = the whole job; 1 = up to the filled x / y arrays (state -> A.jstate); 2 = from the traceback on (the DP ran
// in k_job_dp).  Phases 1 and 2 are launched with arena_lds = 0: every array then lives in the job's global scratch and
// the allocation sequence, replayed from the two counts in jstate, yields the same pointers in all three kernels.
template <int NW, int PHASE = 0>
__device__ void job_group_run(const JobArgs &A, u32 grp, u32 *dyn_lds) {
    __shared__ u32 s_m;
    __shared__ int s_ovf, s_flag[4];
    __shared__ u64 *s_H;
    __shared__ u32 s_nH, s_nhits;
    __shared__ LeaderScratch s_ls;   // introsort stack + tree table: one per workgroup, in LDS
    // the 256 digit counters of the wave radix sort live in the four 64-entry tree tables (used by the traceback only): every
    // static LDS byte of the single-wave kernel is one byte less for its arena at 16 workgroups per CU
    static_assert(sizeof(s_ls.l_root) + sizeof(s_ls.l_score) + sizeof(s_ls.l_len) + sizeof(s_ls.l_leaf) == 256 * sizeof(u32), "tree tables = 256 words");
    u32 *hist = (u32 *)s_ls.l_root;
    __shared__ DpTile<NW> s_tile;    // tiled chaining DP: leaf flags (+ the per-wave candidates of a multi-wave workgroup)
    __shared__ DpShare s_dp;         // NW > 1: what the helper waves need for the DP
    __shared__ RadixShare s_rs;      // NW > 1: ... and for the radix sort
    __shared__ u32 s_rhist[NW > 1 ? NW : 1][256];   // per-wave digit counts / cursors of the workgroup radix sort
    __shared__ u32 s_rtot[NW > 1 ? 256 : 1], s_rflag;
    __shared__ SortShare s_so;       // NW > 1: ... and for the x-descending sort
    __shared__ SortStack s_sstk[NW > 1 ? NW : 1];
    __shared__ SortDeferQ s_sq;
    int lane = lane_id();
    int wave = (int)(threadIdx.x >> 6);
    const bool lead = NW == 1 || wave == 0;
    u32 jb = A.grp_beg[grp], je = A.grp_beg[grp + 1];
    if (jb >= je) return;
    u32 r = A.J.read[jb];
    u64 L = A.read_len[r];
    if (threadIdx.x == 0) { s_ovf = 0; s_flag[3] = 0; }
    if (NW == 1) WSYNC(); else __syncthreads();
    Vec<u64> cords;
    cords.init(A.cords + A.cords_off[r], A.cords_cap[r], &s_ovf);
    cords.n = A.ncords[r];
    unsigned long long tk_ = 0;
#ifdef LNR_PROF
    unsigned long long lnr_job_ph[16];
    for (int q = 0; q < 16; q++) lnr_job_ph[q] = 0;
    tk_ = clock64();
    unsigned long long *tl = (lead && lane == 0 && A.tl) ? A.tl + 4 * (size_t)(A.grp_lo + blockIdx.x) : nullptr;
    if (tl) { tl[0] = wall_clock64(); tl[2] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492); tl[3] = 0; }
#endif
    (void)tk_;
    // diagnostic build: 4 stamp classes of 32 slots = (multi-wave kernel) x 2 + (re-map round)
    unsigned long long *prof = (lead && lane == 0 && A.prof) ? A.prof + 32 * ((NW > 1 ? 2 : 0) + (A.J.mode[jb] ? 1 : 0)) : nullptr;
    (void)prof;
    for (u32 j = jb; j < je; j++) {
        // ---------------- pre: binning, sorts, filter (leader wave)
        u64 *a = nullptr;
        u32 m = 0;
        JobScratch S;
        int mode = (int)A.J.mode[j];
        Arena slow, ar;
        bool ok = true;
#ifdef LNR_PROF
        for (int q = 0; q < 16; q++) lnr_job_ph[q] = 0;
#endif
        u64 *ag = nullptr, *s_alt = nullptr;
        u32 n = 0, cap = 0;
        u32 *so_L = nullptr, *so_R = nullptr; u64 *so_tasks = nullptr, *so_stg = nullptr; u32 so_stg_cap = 0;   // scratch of the x-descending sort
        if (lead && PHASE != 2) {
            ag = A.anchors + A.anc_off[j];
            n = A.n_anchors[j];
            cap = n + 2;   // scratch is sized by the anchors that passed the Y filter (known before the launch), not by the bucket entries
            LNR_TICK(prof, 0, tk_);
            // two-level arena: dynamic LDS first (re-used once binning is done), the job's global scratch behind it
            slow.init(A.scratch + A.scr_off[j], job_scratch_bytes(cap));
            u64 *alt = slow.get<u64>(cap);            // radix ping-pong buffer; before that, the survivors of the binning pre-filter
            if (NW == 1 && A.stop_after == 1) break;   // (diagnostic: start-up only)
            n = binning_wave(ag, n, alt, dyn_lds, A.lds_bytes, A.nbins);   // uses the dynamic LDS as its histogram
            LNR_TICK(prof, 1, tk_);
            if (NW == 1 && A.stop_after == 2) break;
            ar.init((void *)dyn_lds, A.arena_lds); ar.next = &slow;
            a = ag;
            if (n > 1) {
                s_alt = alt;
                a = ar.get<u64>((u64)n + 2);          // sorted anchors move next to the lanes (LDS when they fit)
                if (lane == 0) ag[0] = 0;   // filterAnchorsList pmpfinder.cpp:2031
                WSYNC();
                if (NW == 1) radix_sort_wave(ag, alt, n, hist);
            }
        }
        if (NW > 1 && PHASE != 2) {
            // the radix sort of a multi-wave workgroup is dealt over its waves (long arrays only: five barriers per pass)
            if (threadIdx.x == 0) { s_rs.a = ag; s_rs.alt = s_alt; s_rs.n = (n > 1 && n >= RADIX_BLOCK_MIN) ? n : 0; }
            __syncthreads();
            RadixShare rs = s_rs;
            if (rs.n) radix_sort_block<NW>(rs.a, rs.alt, rs.n, s_rhist, s_rtot, &s_rflag);   // ends with a workgroup barrier
            else if (lead && n > 1) radix_sort_wave(ag, s_alt, n, hist);
        }
        if (lead && PHASE != 2) {
            if (n > 1) {
                for (u32 i = lane; i < n; i += 64) a[i] = ag[i];
                WSYNC();
            }
            LNR_TICK(prof, 2, tk_);
            if (NW == 1 && A.stop_after == 3) break;
            m = n > 1 ? filter_anchor_list_wave(a, n) : n;   // filterAnchors1 (pmpfinder.cpp:2073-2091)
            WSYNC();
            LNR_TICK(prof, 14, tk_);
            if (NW == 1 && A.stop_after == 13) break;   // after the list filter
            if (m > 1) {
                // scratch of the sort: position lists in the (dead) radix buffer, task list behind it
                so_L = (u32 *)s_alt; so_R = so_L + (m + 2);
                so_tasks = slow.get<u64>((u64)m + 2);
                // free LDS behind the arena's current mark can stage the final small sorts when `a` itself is in global memory
                {
                    bool a_in_lds = (char *)a >= (char *)dyn_lds && (char *)a < (char *)dyn_lds + A.arena_lds;
                    u64 used = (ar.off + 15) & ~15ULL;
                    if (!a_in_lds && A.arena_lds > used + 4096) { so_stg = (u64 *)((char *)dyn_lds + used); so_stg_cap = (u32)((A.arena_lds - used) / 8); }
                }
                if (NW == 1) introsort_xdesc_wave(a, m, so_L, so_R, so_tasks, &s_ls, so_stg, so_stg_cap);
            }
        }
        if (NW > 1 && PHASE != 2) {
            // the x-descending sort of a multi-wave workgroup: wave 0 partitions from the top, all waves finish the sub-ranges
            if (threadIdx.x == 0) { s_so.a = a; s_so.L = so_L; s_so.R = so_R; s_so.tasks = so_tasks; s_so.stg = so_stg; s_so.stg_cap = so_stg_cap; s_so.n = (m > 1 && m >= SORT_BLOCK_MIN) ? m : 0; }
            __syncthreads();
            SortShare so = s_so;
            if (so.n) introsort_xdesc_block<NW>(so.a, so.n, so.L, so.R, so.tasks, s_sstk, &s_sq, so.stg, so.stg_cap);   // ends with a workgroup barrier
            else if (lead && m > 1) introsort_xdesc_wave(a, m, so_L, so_R, so_tasks, &s_ls, so_stg, so_stg_cap);
        }
        if (lead && PHASE != 2) {
            LNR_TICK(prof, 15, tk_);
            if (NW == 1 && A.stop_after == 4) break;
            ok = job_carve(ar, m, S, &s_ovf) && !slow.ovf;
            if (ok) {
                job_fill_xy(a, m, S, (u32)lane, 64);
                WSYNC();
            } else if (lane == 0) s_ovf = 1;
            LNR_TICK(prof, 3, tk_);
            if (PHASE == 1) {
                if (lane == 0) { A.jstate[2 * j] = n; A.jstate[2 * j + 1] = m | (ok ? 0x80000000u : 0u); }
                continue;                                 // the DP and everything behind it run in the next kernels
            }
        }
        if (lead && PHASE == 2) {
            u32 n1 = A.jstate[2 * j], mm = A.jstate[2 * j + 1];
            job_replay(A, j, n1, mm & 0x7fffffffu, dyn_lds, slow, ar, a, S, &s_ovf);
            m = mm & 0x7fffffffu;
            ok = (mm >> 31) != 0;
            if (!ok && lane == 0) s_ovf = 1;
            WSYNC();
        }
#ifdef LNR_PROF
        if (prof) {   // job count, anchors entering the DP, predecessor pairs
            unsigned long long pairs = 0;
            if (ok && m >= 2) { u32 p300 = 0; for (u32 i = 0; i < m; i++) { while (p300 < i && S.xs[p300] - S.xs[i] >= 300) p300++; u32 js = i > 20 ? i - 20 : 0; pairs += i - (p300 < js ? p300 : js); } }
            if (tl) tl[3] += ((unsigned long long)m << 32) | (pairs > 0xffffffffULL ? 0xffffffffULL : pairs);
            lnr_job_ph[10] = m; lnr_job_ph[11] = pairs; lnr_job_ph[12] = A.n_anchors[j];
            atomicAdd(&prof[10], 1ULL); atomicAdd(&prof[11], (unsigned long long)m); atomicAdd(&prof[12], pairs); atomicAdd(&prof[13], (unsigned long long)A.n_anchors[j]);
            tk_ = clock64();
        }
#endif
        // ---------------- chaining DP
        if (NW == 1) {
            if (!ok) break;
            if ((PHASE == 0 || PHASE == 3) && m >= 2) {
                best_chains_wave(S.xs, S.ys, m, S.rec, job_parm(mode).score_type, S.cnt, s_tile.tleaf);
            }
        } else {
            if (threadIdx.x == 0) {
                s_dp.xs = S.xs; s_dp.ys = S.ys; s_dp.rec = S.rec; s_dp.jlo = S.cnt; s_dp.m = ok ? m : 0;
                s_dp.score_type = job_parm(mode).score_type; s_dp.abort = ok ? 0 : 1;
            }
            __syncthreads();
            DpShare d = s_dp;
            if (d.abort) break;                        // uniform: every wave leaves together
            if (d.m >= 2) best_chains_block<NW>(d.xs, d.ys, d.m, d.rec, d.score_type, d.jlo, s_tile);   // ends with a workgroup barrier
        }
        LNR_TICK(prof, 4, tk_);
        if (NW == 1 && A.stop_after == 5) break;   // after the DP
        // ---------------- post: traceback, blocks, windows (leader wave)
        if (lead) {
            AnchorSink sink; sink.anchors = a; sink.hits = &S.hits; sink.hscore = &S.hscore; sink.first_len = 0; sink.nchains = 0;
            S.hits.n = 0; S.hscore.n = 0; S.hits.push_u(F_END); S.hscore.push_u(0);   // every lane tracks the counts
            WSYNC();
            if (m >= 2) traceback_anchor_wave(S.rec, m, sink, S.chain, S.chain_sc, S.cnt, s_flag, &s_ls);
            LNR_TICK(prof, 5, tk_);
            if (NW == 1 && A.stop_after == 6) break;
            bool handed = false;
            if (PHASE == 3 && je - jb == 1) {
                // hand-off to k_post (one lane per read runs the stages a11-a16, which are chains of dependent steps per read).
                // Only reads with few chained hits go: the block pre-filter is quadratic in the number of hit blocks, fine on one lane
                // for a typical read and hopeless for a repeat-rich one (those keep the wave-parallel forms below).  Groups of
                // several jobs (re-map round) stay here too: their jobs append to one cord list in order.
                u32 nh = S.hits.n;
                handed = nh <= POST_MAX_HITS && !s_ovf;
                if (handed) {
                    WSYNC();
                    PostIn pi = post_in_of(A.scratch + A.scr_off[j], A.n_anchors[j] + 2);
                    for (u32 i = (u32)lane; i < nh; i += 64) { pi.hits[i] = S.hits.p[i]; pi.hscore[i] = S.hscore.p[i]; }
                    if (lane == 0) { A.jstate[2 * j] = m; A.jstate[2 * j + 1] = nh | 0x80000000u; s_flag[3] = 1; }
                }
            }
            if (!handed) {
            JobCtx c;
            c.traceback_done = 1;
            c.L = L; c.read_str = A.J.str[j]; c.read_end = A.J.end[j]; c.mode = mode;
            u32 nf = A.nf[r];
            c.f1[0].p = A.f1 + A.f1_off[r]; c.f1[0].n = nf;
            c.f1[1].p = A.f1 + A.f1_off[r] + nf; c.f1[1].n = nf;
            c.g = A.g; c.bins = nullptr; c.nbins = 0; c.pair_evals = nullptr; c.prof = prof;
            // hit blocks: gather (leader) -> prefilter (lanes over blocks) -> scores -> block DP (lanes over predecessors)
            // -> traceback + rewrite (leader)
            if (lane == 0) s_nhits = S.hits.n;
            WSYNC();
            S.hits.n = s_nhits;   // the leader filled hits during the traceback: every lane now agrees on the count
            // The stages behind the anchor traceback (a11-a16) are short dependent steps on a hundred hits and a few dozen blocks; in the job's
            // global scratch every step pays a memory round trip.  Everything the anchor DP kept in the LDS arena is dead now (job_carve puts
            // hits / hscore in global memory), so the hits and the block arrays move INTO the arena when they fit: the hits, the rewritten hits,
            // the keep flags (20 B per hit) and 100 B per block for as many blocks as the rest holds.  A read with more blocks than that (seen
            // only when the counts are known) falls back to the arrays carved before -- same code, other pointers.
            u64 *hits2 = a;   // output of the block filter: the anchors are dead by now and hits never outnumber them (job_blocks_finish)
            const JobScratch S0 = S;
            u32 nb = 0;
            bool in_lds = false;
            if (PHASE != 2 && A.arena_lds) {
                Arena pl; pl.init((void *)dyn_lds, A.arena_lds);
                const u32 nh = S.hits.n;
                u64 *lh = pl.get<u64>((u64)nh + 2), *lH = pl.get<u64>((u64)nh + 2);
                i32 *lcnt = pl.get<i32>((u64)nh + 2);
                u64 rest = pl.ovf ? 0 : A.arena_lds - pl.off;
                u32 C = rest > 14 * 16 + 2 * 100 ? (u32)((rest - 14 * 16) / 100) - 2 : 0;
                if (C >= 16) {
                    if (C > nh) C = nh;       // (never more blocks or pieces than hits)
                    if (lane == 0) s_flag[1] = 0;
                    for (u32 i = lane; i < nh; i += 64) lh[i] = S.hits.p[i];
                    S.hits.p = lh;
                    S.sep.init(pl.get<UP>((u64)C + 2), C, &s_flag[1]);
                    S.tmp.init(pl.get<UP>((u64)C + 2), C, &s_flag[1]);
                    S.cuts = pl.get<u64>(2 * (u64)C + 4); S.xy_strs = pl.get<u64>((u64)C + 2);
                    S.sep_score = pl.get<i32>((u64)C + 2);
                    S.xs = pl.get<u32>((u64)C + 2); S.ys = pl.get<u32>((u64)C + 2);
                    S.rec.score = pl.get<i32>((u64)C + 2); S.rec.score2 = pl.get<i32>((u64)C + 2); S.rec.len = pl.get<i32>((u64)C + 2);
                    S.rec.p2 = pl.get<i32>((u64)C + 2); S.rec.root = pl.get<i32>((u64)C + 2); S.rec.leaf = pl.get<i32>((u64)C + 2);
                    S.chain = pl.get<i32>((u64)C + 2); S.chain_sc = pl.get<i32>((u64)C + 2);
                    S.cnt = lcnt;
                    WSYNC();
                    if (!pl.ovf) {
                        nb = gather_blocks_wave(S.hits.p, nh, S.sep.p, S.sep.cap, S.sep.ovf);
                        bool fits = !s_flag[1];
                        if (fits) {
                            S.sep.n = nb;
                            nb = prefilter_chains2_wave(S.hits.p, nh, S.sep, nb, S.cuts, S.xy_strs, S.tmp, s_ls);
                            fits = !s_flag[1];
                        }
                        if (fits) { in_lds = true; hits2 = lH; }
                    }
                    if (!in_lds) { WSYNC(); S = S0; }
                }
            }
            if (!in_lds) {
                S.tmp.n = 0;
                nb = gather_blocks_wave(S.hits.p, S.hits.n, S.sep.p, S.sep.cap, S.sep.ovf);
                S.sep.n = nb;
                nb = prefilter_chains2_wave(S.hits.p, S.hits.n, S.sep, nb, S.cuts, S.xy_strs, S.tmp, s_ls);
            }
            S.sep.n = nb;
            LNR_TICK(prof, 6, tk_);
#ifdef LNR_PROF
            (void)in_lds;
            unsigned long long tsub_ = clock64();
#endif
            if (NW == 1 && A.stop_after == 7) break;
            job_blocks_scores(S, (u32)lane, 64);
            WSYNC();
            BlockScratch bsx = job_block_scratch(S, s_ls);
            if (nb >= 2) {
                chain_blocks_prepare_wave(S.hits.p, S.sep.p, S.sep_score, nb, bsx);
                if (NW == 1 && A.stop_after == 10) break;
#ifdef LNR_PROF
                unsigned long long tdp_ = clock64();
#endif
                best_chains2_wave(S.hits.p, bsx.sep_tmp, bsx.score_tmp, nb, bsx.rec, prof);
#ifdef LNR_PROF
                if (prof) atomicAdd(&prof[28], clock64() - tdp_);
#endif
                if (NW == 1 && A.stop_after == 11) break;
            }
#ifdef LNR_PROF
            tsub_ = clock64();
#endif
            {
                BlockSink bs = job_block_sink(S);
                if (lane == 0) { if (nb >= 2) chain_blocks_trace(bs, nb, bsx); s_flag[2] = (int)bs.nchains; }
                WSYNC();
#ifdef LNR_PROF
#endif
                if (NW == 1 && A.stop_after == 12) break;
                u32 nh2 = filter_blocks_hits_wave(bs, (u32)s_flag[2], S.hits.p, hits2);
                if (lane == 0) { if (nh2 == 0xffffffffu) { s_H = S.hits.p; s_nH = S.hits.n; } else { s_H = hits2; s_nH = nh2; } }
            }
            WSYNC();
            LNR_TICK(prof, 7, tk_);
            if (NW == 1 && A.stop_after == 8) break;
            if (!s_ovf) {
                u64 *H = s_H; u32 nH = s_nH;
                if (nH >= 2) {
                    filter_hits_flags(H, nH, c.f1, c.g, S.cnt, (u32)lane, 64);    // window distance of one hit per lane
                    WSYNC();
                    if (lane == 0) s_nH = filter_hits_apply(H, nH, S.cnt);
                    WSYNC();
                    nH = s_nH;
                    LNR_TICK(prof, 8, tk_);
                    if (NW == 1 && A.stop_after == 9) break;   // after filter_hits
                    path_dst_2(H, nH, c.f1, c.g, cords, c.read_str, c.read_end, L);   // SIMT-uniform: candidates evaluated by lanes 0..2
                    LNR_TICK(prof, 9, tk_);
                }
            }
            }   // !handed
            WSYNC();
#ifdef LNR_PROF
            if (A.prof && lane == 0) {   // phase cycles of the job with the most anchors in the DP (per launch class of 4: A.prof + 128 + 16 * class)
                unsigned long long *mp = A.prof + 128 + 16 * ((NW > 1 ? 2 : 0) + (A.J.mode[jb] ? 1 : 0));
                unsigned long long old = atomicMax(&mp[10], lnr_job_ph[10]);
                if (lnr_job_ph[10] > old) { for (int q = 0; q < 10; q++) mp[q] = lnr_job_ph[q]; mp[11] = lnr_job_ph[11]; mp[12] = lnr_job_ph[12]; mp[14] = lnr_job_ph[14]; mp[15] = lnr_job_ph[15]; }
            }
            tk_ = clock64();
#endif
        }
        if (NW == 1) { if (s_ovf) break; }
        else {
            __syncthreads();              // post done: the helpers may see s_ovf and the leader may reuse the LDS
            if (s_ovf) break;             // uniform
        }
    }
    if (lead && lane == 0) { if (PHASE != 1 && !(PHASE == 3 && s_flag[3])) A.ncords[r] = cords.n; if (s_ovf) A.read_err[r] = 1; }
#ifdef LNR_PROF
    if (tl) tl[1] = wall_clock64();
#endif
}

// K_JOB_WAVES: waves per SIMD the single-wave kernel is compiled for (4: 128 VGPRs; 6: 80 VGPRs with spills -- the compiler's budget
// for 5 is 104 registers, which the hardware's granule of 8 turns into 4 waves again).  The phase functions above are inlined by force
// in the register-capped build: a call keeps the callee's own, uncapped register count (traceback 104, introsort 104).
#ifndef K_JOB_WAVES
#define K_JOB_WAVES 4
#endif
__global__ void __attribute__((amdgpu_flat_work_group_size(64, 64), amdgpu_waves_per_eu(K_JOB_WAVES, K_JOB_WAVES))) k_job(JobArgs A) {
    extern __shared__ u32 dyn_lds[];
    if (A.grp_lo + blockIdx.x >= A.grp_hi) return;
    job_group_run<1>(A, A.grp_order[A.grp_lo + blockIdx.x], dyn_lds);
}
// Split path for reads with many anchors: k_job_pre (one wave: binning .. x/y arrays) -> k_job_dp (16 waves: the chaining DP
// and nothing else, so no wave idles through the serial phases) -> k_job_post (one wave: traceback .. cords), on one stream.
__global__ void __launch_bounds__(64, 4) k_job_pre(JobArgs A) {
    extern __shared__ u32 dyn_lds[];
    if (A.grp_lo + blockIdx.x >= A.grp_hi) return;
    job_group_run<1, 1>(A, A.grp_order[A.grp_lo + blockIdx.x], dyn_lds);
}
__global__ void __launch_bounds__(64, 4) k_job_post(JobArgs A) {
    extern __shared__ u32 dyn_lds[];
    if (A.grp_lo + blockIdx.x >= A.grp_hi) return;
    job_group_run<1, 2>(A, A.grp_order[A.grp_lo + blockIdx.x], dyn_lds);
}
#define DP_SPLIT_WAVES 4
__global__ void __launch_bounds__(64 * DP_SPLIT_WAVES) k_job_dp(JobArgs A) {
    __shared__ DpTile<DP_SPLIT_WAVES> s_tile;
    __shared__ DpShare s_dp;
    if (A.grp_lo + blockIdx.x >= A.grp_hi) return;
    u32 grp = A.grp_order[A.grp_lo + blockIdx.x];
    for (u32 j = A.grp_beg[grp]; j < A.grp_beg[grp + 1]; j++) {
        if (threadIdx.x == 0) {
            u32 n1 = A.jstate[2 * j], mm = A.jstate[2 * j + 1];
            u32 m = mm & 0x7fffffffu;
            Arena slow, ar; u64 *a; JobScratch S; int ovf = 0;
            job_replay(A, j, n1, m, nullptr, slow, ar, a, S, &ovf);
            s_dp.xs = S.xs; s_dp.ys = S.ys; s_dp.rec = S.rec; s_dp.jlo = S.cnt; s_dp.m = (mm >> 31) ? m : 0;
            s_dp.score_type = job_parm((int)A.J.mode[j]).score_type; s_dp.abort = 0;
        }
        __syncthreads();
        DpShare d = s_dp;
        if (d.m >= 2) best_chains_block<DP_SPLIT_WAVES>(d.xs, d.ys, d.m, d.rec, d.score_type, d.jlo, s_tile);
        __syncthreads();
    }
}
__global__ void __launch_bounds__(1024) k_job_heavy(JobArgs A) {
    extern __shared__ u32 dyn_lds[];
    if (A.grp_lo + blockIdx.x >= A.grp_hi) return;
    job_group_run<16>(A, A.grp_order[A.grp_lo + blockIdx.x], dyn_lds);
}
__global__ void __launch_bounds__(256, 4) k_job_mid(JobArgs A) {
    extern __shared__ u32 dyn_lds[];
    if (A.grp_lo + blockIdx.x >= A.grp_hi) return;
    job_group_run<4>(A, A.grp_order[A.grp_lo + blockIdx.x], dyn_lds);
}
// the same class on two waves (LNR_MID_WAVES=2): half the wave slots per read for a longer time
__global__ void __launch_bounds__(128, 4) k_job_mid2(JobArgs A) {
    extern __shared__ u32 dyn_lds[];
    if (A.grp_lo + blockIdx.x >= A.grp_hi) return;
    job_group_run<2>(A, A.grp_order[A.grp_lo + blockIdx.x], dyn_lds);
}

// ---- split form (default): the wave-parallel stages a8-a10 (binning .. chaining DP .. traceback) per read in k_job*_a, then
// k_post with ONE LANE PER READ for the stages that are chains of dependent steps per read (a11-a16: block gathering, block
// pre-filter, block chaining, window filter, window extension).  In the fused kernels those ran one lane of 64 for 70-80 % of a
// read's time (phase stamps: extension 37 %, block chaining 19 %, gather + pre-filter 16 % at human scale) while the wave's
// other lanes idled; here 64 reads share a wave, every read of the batch is resident at once, and the stage is over when its
// longest chain is.
__global__ void __launch_bounds__(64, 4) k_job_a(JobArgs A) {
    extern __shared__ u32 dyn_lds[];
    if (A.grp_lo + blockIdx.x >= A.grp_hi) return;
    job_group_run<1, 3>(A, A.grp_order[A.grp_lo + blockIdx.x], dyn_lds);
}
__global__ void __launch_bounds__(256, 4) k_job_mid_a(JobArgs A) {
    extern __shared__ u32 dyn_lds[];
    if (A.grp_lo + blockIdx.x >= A.grp_hi) return;
    job_group_run<4, 3>(A, A.grp_order[A.grp_lo + blockIdx.x], dyn_lds);
}
__global__ void __launch_bounds__(1024) k_job_heavy_a(JobArgs A) {
    extern __shared__ u32 dyn_lds[];
    if (A.grp_lo + blockIdx.x >= A.grp_hi) return;
    job_group_run<16, 3>(A, A.grp_order[A.grp_lo + blockIdx.x], dyn_lds);
}
__global__ void __launch_bounds__(64) k_post(JobArgs A) {
    u32 pos = A.grp_lo + blockIdx.x * 64 + threadIdx.x;
    if (pos >= A.grp_hi) return;
    u32 grp = A.grp_order[pos];
    u32 jb = A.grp_beg[grp], je = A.grp_beg[grp + 1];
    if (jb >= je) return;
    u32 r = A.J.read[jb];
    if (A.read_err[r]) return;
    u64 L = A.read_len[r];
    int ovf = 0;
    Vec<u64> cords;
    cords.init(A.cords + A.cords_off[r], A.cords_cap[r], &ovf);
    cords.n = A.ncords[r];
    LeaderScratch ls;
    for (u32 j = jb; j < je && !ovf; j++) {
        u32 m = A.jstate[2 * j], st = A.jstate[2 * j + 1];
        if (!(st >> 31)) return;                        // not handed over: the job kernel ran these stages itself (or flagged the read)
        u32 nh = st & 0x7fffffffu;
        u32 cap = A.n_anchors[j] + 2;
        char *region = A.scratch + A.scr_off[j];
        PostIn pi = post_in_of(region, cap);
        Arena ar; ar.init(region, (u64)((char *)pi.hits - region));
        JobScratch S;
        if (!job_carve(ar, m, S, &ovf)) { ovf = 1; break; }
        u64 *a2 = ar.get<u64>((u64)m + 2);              // output of _filterBlocksHits (the fused kernel reuses the dead anchor array)
        if (ar.ovf) { ovf = 1; break; }
        S.hits.init(pi.hits, cap, &ovf); S.hits.n = nh;
        S.hscore.init(pi.hscore, cap, &ovf); S.hscore.n = nh;
        JobCtx c;
        c.traceback_done = 1;
        c.L = L; c.read_str = A.J.str[j]; c.read_end = A.J.end[j]; c.mode = (int)A.J.mode[j];
        u32 nf = A.nf[r];
        c.f1[0].p = A.f1 + A.f1_off[r]; c.f1[0].n = nf;
        c.f1[1].p = A.f1 + A.f1_off[r] + nf; c.f1[1].n = nf;
        c.g = A.g; c.bins = nullptr; c.nbins = 0; c.pair_evals = nullptr; c.prof = nullptr;
        u64 *H = nullptr; u32 nH = 0;
        if (job_phase3a(a2, m, S, c, nullptr, H, nH, ls)) { ovf = 1; break; }
        if (nH >= 2) {
            filter_hits_flags(H, nH, c.f1, c.g, S.cnt, 0, 1);
            nH = filter_hits_apply(H, nH, S.cnt);
            path_dst_2<false>(H, nH, c.f1, c.g, cords, c.read_str, c.read_end, L);
        }
    }
    A.ncords[r] = cords.n;
    if (ovf) A.read_err[r] = 1;
}

// Placed on the bulk stream ahead of k_job when multi-wave kernels were launched beside it: a 4- or 16-wave workgroup
// only finds room while the single-wave kernel has not filled every CU, and the three launches reach the GPU through
// different queues in no particular order.  100 us (of a ~30 ms kernel) lets the big workgroups take their places.
__global__ void k_delay(unsigned ticks_100mhz) {
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks_100mhz) __builtin_amdgcn_s_sleep(32);
}

// =================================================================== tails ====
struct TailArgs {
    const u32 *read_len; u32 n;
    const u32 *list;            // optional: the reads this launch covers (n = its length); null = reads 0..n-1
    u64 *cords; const u64 *cords_off; const u32 *cords_cap; u32 *ncords; i32 *read_err;
    char *scratch; const u64 *scr_off; const u32 *scr_cap;   // per read: bytes offset / capacity in cord slots used for sizing
    UP *gaps; const u64 *gaps_off; const u32 *gaps_cap; u32 *ngaps; u32 *remap;
    UP *gdense; u32 *gcursor; u32 *gpos;   // gaps of the reads that go to the re-map round, packed: [gpos[r], gpos[r] + ngaps[r])
    u64 *out_str, *out_end; u32 *nout;
};
__global__ void __launch_bounds__(64) k_tail_a(TailArgs T) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= T.n) return;
    if (T.list) r = T.list[r];
    T.ngaps[r] = 0; T.remap[r] = 0;
    u64 L = T.read_len[r];
    if (L <= 200 || T.read_err[r]) return;
    Arena ar; ar.init(T.scratch + T.scr_off[r], tail_scratch_bytes(T.scr_cap[r]));
    u32 nc = T.ncords[r], ng = 0, rm = 0;
    LeaderScratch ls;
    int rc = tail_a(T.cords + T.cords_off[r], nc, L, ar, T.gaps + T.gaps_off[r], T.gaps_cap[r], ng, rm, ls);
    T.ncords[r] = nc; T.ngaps[r] = ng; T.remap[r] = rm;
    if (rm && ng) {   // hand the gaps to the host densely (one small copy instead of the whole per-read table)
        u32 pos = atomicAdd(T.gcursor, ng);
        T.gpos[r] = pos;
        const UP *g = T.gaps + T.gaps_off[r];
        for (u32 k = 0; k < ng; k++) T.gdense[pos + k] = g[k];
    }
    if (rc || ar.ovf) T.read_err[r] = 2;
}
__global__ void __launch_bounds__(64) k_tail_b(TailArgs T) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= T.n) return;
    if (T.list) r = T.list[r];
    T.nout[r] = 0;
    u64 L = T.read_len[r];
    if (L <= 200 || T.read_err[r]) return;
    Arena ar; ar.init(T.scratch + T.scr_off[r], tail_scratch_bytes(T.scr_cap[r]));
    u32 no = 0;
    LeaderScratch ls;
    int rc = tail_b(T.cords + T.cords_off[r], T.ncords[r], L, ar, T.out_str + T.cords_off[r], T.out_end + T.cords_off[r], T.cords_cap[r], no, ls);
    T.nout[r] = no;
    if (rc || ar.ovf) { T.read_err[r] = 3; T.nout[r] = 0; }
}
__global__ void __launch_bounds__(64) k_gather_out(const u64 *out_str, const u64 *out_end, const u64 *cords_off, const u32 *nout, const u64 *cord_off, u32 n,
                                                   u64 *cs, u64 *ce) {
    u32 r = blockIdx.x;
    if (r >= n) return;
    u32 c = nout[r];
    const u64 *s = out_str + cords_off[r], *e = out_end + cords_off[r];
    u64 o = cord_off[r];
    for (u32 i = threadIdx.x; i < c; i += blockDim.x) { cs[o + i] = s[i]; ce[o + i] = e[i]; }
}

}  // namespace lnr
