// lnr_gap_kernels.hip -- the kernels of the gap re-mapper (mapGaps + reformCords behind apxMap, `-g > 0`): k_gap / k_gap_team
// around the per-read code of lnr_gap_hd.h, the wave-parallel exact introsort of that code, and the host-side launcher.  A translation unit
// of its own (compiled beside lnr_api.hip, linked into the same library): the two halves build in parallel.
#include <hip/hip_runtime.h>
#include "lnr_wave.h"
#include "lnr_gap_hd.h"
#include "lnr_gap_args.h"

namespace lnr {
#define SORT_SMALL 32
// ---- gap_sort_wave: introsort_xdesc_wave for any element type and comparator, the scratch taken from the read's arena.  The array,
// the swap-candidate lists and the task list live in global memory; the explicit stack is wave-uniform private state (this kernel has
// one wave per workgroup and registers to spare).  Falls back to the serial ref_sort when the arena cannot hold the lists.
template <class T, class Comp> __device__ void gap_sort_wave(T *a, u32 n, Comp comp, GapCtx &X) {
    const int lane = lane_id();
    u64 m0 = X.ar->mark();
    u32 *Lbuf = (u32 *)X.ar->get((u64)n * 4), *Rbuf = (u32 *)X.ar->get((u64)n * 4);
    u64 *tasks = (u64 *)X.ar->get(((u64)n + 64) * 8);
    if (X.ar->ovf) {                                         // (the overflow stands: the read is redone with a larger arena)
        ref_sort(a, (long)n, comp, X.ls->st);
        return;
    }
    int stk_first[64], stk_last[64], stk_depth[64];
    int sp = 0, lg = 0;
    for (u32 t = n; t > 1; t >>= 1) lg++;
    stk_first[0] = 0; stk_last[0] = (int)n; stk_depth[0] = lg * 2;
    sp = 1;
    u32 ntasks = 0;
    WSYNC();
    while (sp > 0) {
        --sp;
        u32 first = (u32)stk_first[sp], last = (u32)stk_last[sp];
        int depth = stk_depth[sp];
        while (true) {
            if (last - first <= SORT_SMALL) {
                if (lane == 0) tasks[ntasks] = (u64)first | ((u64)last << 28) | ((u64)depth << 56);
                ntasks++;
                break;
            }
            if (depth == 0) {
                if (lane == 0) tasks[ntasks] = (u64)first | ((u64)last << 28) | (1ULL << 63);
                ntasks++;
                break;
            }
            --depth;
            u32 iA = first + 1, iB = first + (last - first) / 2, iC = last - 1;
            T va = a[iA], vb = a[iB], vc = a[iC], vf = a[first];            // (uniform loads)
            u32 pick;
            if (comp(va, vb)) pick = comp(vb, vc) ? iB : (comp(va, vc) ? iC : iA);
            else pick = comp(va, vc) ? iA : (comp(vb, vc) ? iC : iB);
            T vp = pick == iA ? va : (pick == iB ? vb : vc);
            WSYNC();                                                          // every lane has read the four before one of them is overwritten
            if (lane == 0) { a[first] = vp; a[pick] = vf; }
            u32 lo = first + 1, nL = 0, nR = 0;
            for (u32 base = lo; base < last; base += 128) {
                T v2[2];
#pragma unroll
                for (int u = 0; u < 2; u++) { u32 i = base + 64 * u + lane; v2[u] = i < last ? a[i] : vp; }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    if (base + 64 * u >= last) break;       // uniform
                    u32 i = base + 64 * u + lane;
                    bool in = i < last;
                    T x = i == pick ? vf : v2[u];
                    bool fL = in && !comp(x, vp);           // the left scan stops here
                    bool fR = in && !comp(vp, x);           // the right scan stops here
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
            for (u32 k = lane; k < K; k += 64) { u32 i = Lbuf[k], j = Rbuf[nR - 1 - k]; T t = a[i]; a[i] = a[j]; a[j] = t; }
            u32 cut = last;
            if (K < nL) cut = Lbuf[K];
            if (K >= 1) { u32 r = Rbuf[nR - K]; cut = r < cut ? r : cut; }
            WSYNC();
            stk_first[sp] = (int)cut; stk_last[sp] = (int)last; stk_depth[sp] = depth;
            ++sp;
            last = cut;
        }
    }
    WSYNC();
    for (u32 b = 0; b < ntasks; b += 64) {                   // the deferred ranges, one per lane
        u32 t = b + (u32)lane;
        if (t < ntasks) {
            u64 v = tasks[t];
            if (v >> 63) rs_heap_sort(a, (long)(v & 0xfffffff), (long)((v >> 28) & 0xfffffff), comp);
            else rs_finish_range<16>(a, (long)(v & 0xfffffff), (long)((v >> 28) & 0xfffffff), (int)((v >> 56) & 0x7f), comp);
        }
    }
    WSYNC();
    X.ar->release(m0);
}

#ifndef K_GAP_WAVES
#define K_GAP_WAVES 4
#endif
__device__ void gap_worker(const GapArgs &A, GapTeam *tm, int team) {
    u32 worker = A.coop ? blockIdx.x : blockIdx.x * blockDim.x + threadIdx.x;
    char *mine = A.arena + (u64)worker * A.arena_bytes;
    for (;;) {
        u32 r;
        if (A.coop) { r = threadIdx.x == 0 ? atomicAdd(A.next, 1u) : 0u; r = (u32)__shfl((int)r, 0); }
        else r = atomicAdd(A.next, 1u);
        r += A.lo;
        if (r >= A.n) break;
        if (A.big && !A.gap_flag[r]) continue;
        if (A.big && threadIdx.x == 0) atomicAdd(A.next + 8, 1u);   // (statistics: reads of the second launch)
        u32 nc = A.nout[r];
        u64 L = A.off[r + 1] - A.off[r];
        if (L <= 200 || nc <= 1) { if (!A.big) A.gap_flag[r] = 0; continue; }
        GArena all; all.init(mine, A.arena_bytes);
        LeaderScratch *ls = (LeaderScratch *)all.get(sizeof(LeaderScratch));
        u8 *rd = (u8 *)all.get(L + 64), *rc = (u8 *)all.get(L + 64);
        u64 keep_bytes = ((u64)A.cords_cap[r] * 16 + (u64)nc * 64 + 8192) * 2;
        char *kp = (char *)all.get(keep_bytes);
        bool bad = all.ovf != 0;
        if (!bad) {
            const u8 *src = A.reads + A.off[r];
            for (u64 k = 0; k < L; k++) { u8 b = src[k]; b = b > 4 ? 4 : b; rd[k] = b; rc[L - 1 - k] = b == 4 ? 4 : 3 - b; }
            for (u32 k = 0; k < 64; k++) { rd[L + k] = 0; rc[L + k] = 0; }
            GArena keep; keep.init(kp, keep_bytes);
            GArena ar; ar.init(mine + all.off, A.arena_bytes - all.off);
            GapCtx X;
            X.ar = &ar; X.ls = ls; X.read.p = rd; X.read.len = L; X.com.p = rc; X.com.len = L;
            X.g = A.g; X.seq_off = A.seq_off; X.seq_len = A.seq_len;
            u32 nf = A.nf[r];
            X.f1[0].p = A.f1 + A.f1_off[r]; X.f1[0].n = nf; X.f1[1].p = A.f1 + A.f1_off[r] + nf; X.f1[1].n = nf;
            X.gf = A.gf;
            X.gp.f_dup = A.f_dup; X.gp.thd_gap_len_min = A.gap_len_min;
            const bool ext_in = r >= A.ext_from;
            if (ext_in) X.gp.thd_cts_major_limit = 3;
            X.coop = A.coop; X.work_cap = A.work_cap; X.team = team; X.tm = tm;
            u64 *os = A.out_str + A.cords_off[r], *oe = A.out_end + A.cords_off[r];
            GVec<u64> cs, ce; cs.init(&keep, nc * 2 + 64); ce.init(&keep, nc * 2 + 64);
            for (u32 i = 0; i < nc; i++) { cs.push(os[i]); ce.push(oe[i]); }
#ifdef LNR_GAP_DEVPROF
            unsigned long long t_read = wall_clock64();
#endif
            int rc_ = gap_map_gaps(cs, ce, keep, X);
            gap_reform_cords(cs, ce);
#ifdef LNR_GAP_DEVPROF
            if (A.prof && threadIdx.x == (A.coop ? 0u : threadIdx.x)) {
                unsigned long long *pp = A.prof + 16 * (A.big + A.last);
                t_read = wall_clock64() - t_read;
                for (int k = 0; k < 10; k++) atomicAdd(pp + k, X.prof[k]);
                atomicAdd(pp + 11, t_read); atomicAdd(pp + 12, 1ULL);
                A.prof[96 + r] = t_read | ((unsigned long long)(A.big + A.last) << 56);      // per-read time of the launch that did the read
                if (atomicMax(pp + 15, t_read) < t_read) { unsigned long long *ps = A.prof + 48 + 16 * (A.big + A.last); for (int k = 0; k < 10; k++) ps[k] = X.prof[k]; ps[10] = r; ps[11] = L; ps[12] = nc; ps[13] = ar.hw; }
            }
#endif
            bad = rc_ != 0 || ar.ovf || keep.ovf || cs.n > A.cords_cap[r] || cs.n != ce.n;
            if (!bad) {
                if (!ext_in && X.gp.thd_cts_major_limit == 3 && threadIdx.x == (A.coop ? 0u : threadIdx.x)) atomicMin(A.first_ext, r);
                if (!A.probe) {
                    for (u32 i = 0; i < cs.n; i++) { os[i] = cs[i]; oe[i] = ce[i]; }
                    A.nout[r] = cs.n;
                }
            }
        }
        A.gap_flag[r] = bad ? 1 : 0;
        if (A.last && bad) A.read_err[r] = 5;
    }
}
__global__ void __attribute__((amdgpu_flat_work_group_size(64, 64), amdgpu_waves_per_eu(K_GAP_WAVES, K_GAP_WAVES))) k_gap(GapArgs A) { gap_worker(A, nullptr, 1); }
// The launches for the flagged reads: K_GAP_TEAM waves per read.  Wave 0 is the worker; the others only serve the long rows of its
// chain DPs (gap_team_helper_loop) and leave when wave 0 has run out of reads.
__global__ void __attribute__((amdgpu_flat_work_group_size(64 * K_GAP_TEAM, 64 * K_GAP_TEAM))) k_gap_team(GapArgs A) {
    __shared__ GapTeam tm;
    if (threadIdx.x >= 64) { gap_team_helper_loop(&tm, (int)(threadIdx.x >> 6), K_GAP_TEAM); return; }
    gap_worker(A, &tm, K_GAP_TEAM);
    if (threadIdx.x == 0) tm.cmd = 0;
    __syncthreads();                                             // (A) with the exit command: the helpers leave
}


hipError_t launch_gap(const GapArgs &A, int team, unsigned grid, hipStream_t stream) {
    if (team) hipLaunchKernelGGL(k_gap_team, dim3(grid), dim3(64 * K_GAP_TEAM), 0, stream, A);
    else hipLaunchKernelGGL(k_gap, dim3(grid), dim3(64), 0, stream, A);
    return hipGetLastError();
}

}  // namespace lnr
