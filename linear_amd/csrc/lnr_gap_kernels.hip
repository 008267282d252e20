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
// the swap-candidate lists and the task list live in global memory; the explicit stack is wave-uniform private state.  Falls back to the serial
// ref_sort when the arena cannot hold the lists.
// gap_sort_core sorts a[first0, last0) the way std::sort's introsort loop + final insertion sort do (ref_sort.h), with the partitions of the
// ranges above SORT_SMALL done by all 64 lanes and the small ranges finished one per lane.  Everything a range needs lies at the range's own
// indices of Lbuf / Rbuf / tasks, so several waves can work on disjoint ranges of one array at once.  Ranges of at most `cutoff` elements are
// not sorted here but appended to `queue` (first | last << 28 | depth << 56): the team form deals them over the waves afterwards -- what
// introsort does with a range depends on that range (and its depth budget) alone.
template <class T, class Comp> __device__ void gap_sort_core(T *a, u32 first0, u32 last0, int depth0, Comp comp, u32 *Lbuf, u32 *Rbuf, u64 *tasks, u32 cutoff, u64 *queue, u32 *nqueue) {
    const int lane = lane_id();
    int stk_first[64], stk_last[64], stk_depth[64];
    int sp = 0;
    stk_first[0] = (int)first0; stk_last[0] = (int)last0; stk_depth[0] = depth0;
    sp = 1;
    u64 *tk = tasks + first0;
    u32 ntasks = 0, nq = nqueue ? *nqueue : 0;
    WSYNC();
    while (sp > 0) {
        --sp;
        u32 first = (u32)stk_first[sp], last = (u32)stk_last[sp];
        int depth = stk_depth[sp];
        while (true) {
            if (last - first <= SORT_SMALL) {
                if (lane == 0) tk[ntasks] = (u64)first | ((u64)last << 28) | ((u64)depth << 56);
                ntasks++;
                break;
            }
            if (depth == 0) {
                if (lane == 0) tk[ntasks] = (u64)first | ((u64)last << 28) | (1ULL << 63);
                ntasks++;
                break;
            }
            if (last - first <= cutoff) {                                     // (team form, wave 0: left for the waves)
                if (lane == 0) queue[nq] = (u64)first | ((u64)last << 28) | ((u64)depth << 56);
                nq++;
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
            u32 *Lb = Lbuf + first, *Rb = Rbuf + first;
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
                    if (fL) Lb[nL + __popcll(mL & lanemask_lt())] = i;
                    if (fR) Rb[nR + __popcll(mR & lanemask_lt())] = i;
                    nL += (u32)__popcll(mL); nR += (u32)__popcll(mR);
                }
            }
            WSYNC();
            u32 lim = nL < nR ? nL : nR, cnt = 0;
            for (u32 k = lane; k < lim; k += 64) cnt += Lb[k] < Rb[nR - 1 - k] ? 1u : 0u;
            u32 K = wave_sum(cnt);
            for (u32 k = lane; k < K; k += 64) { u32 i = Lb[k], j = Rb[nR - 1 - k]; T t = a[i]; a[i] = a[j]; a[j] = t; }
            u32 cut = last;
            if (K < nL) cut = Lb[K];
            if (K >= 1) { u32 r = Rb[nR - K]; cut = r < cut ? r : cut; }
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
            u64 v = tk[t];
            if (v >> 63) rs_heap_sort(a, (long)(v & 0xfffffff), (long)((v >> 28) & 0xfffffff), comp);
            else rs_finish_range<16>(a, (long)(v & 0xfffffff), (long)((v >> 28) & 0xfffffff), (int)((v >> 56) & 0x7f), comp);
        }
    }
    WSYNC();
    if (nqueue) *nqueue = nq;
}
// the team's share of a big sort: ranges from the queue, one at a time per wave (cmd 3)
__device__ void gap_sort_team_share(GapTeam *tm) {
    for (;;) {
        u32 k = 0;
        if (lane_id() == 0) k = atomicAdd(&tm->s_next, 1u);
        k = (u32)__shfl((int)k, 0);
        if (k >= tm->s_nq) break;
        u64 v = tm->s_queue[k];
        gap_sort_core<u64, GapCmp>(tm->s_a, (u32)(v & 0xfffffff), (u32)((v >> 28) & 0xfffffff), (int)((v >> 56) & 0x7f), tm->s_cmp, tm->s_L, tm->s_R, tm->s_tasks, 0, nullptr, nullptr);
    }
}
#ifndef K_GAP_SORT_TEAM_MIN
#define K_GAP_SORT_TEAM_MIN 4096
#endif
template <class T, class Comp> struct GapSortTeam { static __device__ bool run(T *, u32, Comp, GapCtx &, u32 *, u32 *, u64 *, int) { return false; } };
template <> struct GapSortTeam<u64, GapCmp> {
    // wave 0 partitions from the top down to ranges of n / 64 elements (at least 2048) and queues them; all waves of the team then sort queued ranges
    static __device__ bool run(u64 *a, u32 n, GapCmp comp, GapCtx &X, u32 *Lbuf, u32 *Rbuf, u64 *tasks, int lg2) {
        if (X.team <= 1 || n < K_GAP_SORT_TEAM_MIN) return false;
        u32 cutoff = n / 64 > 2048 ? n / 64 : 2048;
        u64 *queue = (u64 *)X.ar->get(((u64)n / SORT_SMALL + 512) * 8);       // (queued ranges are disjoint and longer than SORT_SMALL: at most n / 33 of them)
        if (X.ar->ovf) return false;
        u32 nq = 0;
        gap_sort_core<u64, GapCmp>(a, 0, n, lg2 * 2, comp, Lbuf, Rbuf, tasks, cutoff, queue, &nq);
        GapTeam *tm = X.tm;
        if (lane_id() == 0) { tm->s_a = a; tm->s_L = Lbuf; tm->s_R = Rbuf; tm->s_tasks = tasks; tm->s_queue = queue; tm->s_nq = nq; tm->s_next = 0; tm->s_cmp = comp; tm->cmd = 3; }
        __syncthreads();                                                      // (A)
        gap_sort_team_share(tm);
        __syncthreads();                                                      // (B)
        return true;
    }
};
template <class T, class Comp> __device__ void gap_sort_wave(T *a, u32 n, Comp comp, GapCtx &X) {
    // a single wave that meets a sort this long has a read no cost predictor announced: given up here (like an arena overflow) and redone by
    // a team, whose 16 waves share the sort -- the stage waits for its slowest worker, not for the sum
    if (X.hand && n >= K_GAP_SINGLE_MAX) { if (!X.ar->ovf) X.ar->ovf = 2; return; }
    u64 m0 = X.ar->mark();
    u32 *Lbuf = (u32 *)X.ar->get((u64)n * 4), *Rbuf = (u32 *)X.ar->get((u64)n * 4);
    u64 *tasks = (u64 *)X.ar->get(((u64)n + 64) * 8);
    if (X.ar->ovf) {                                         // (the overflow stands: the read is redone with a larger arena)
        ref_sort(a, (long)n, comp, X.ls->st);
        return;
    }
    if (gap_late(X)) return;
    int lg = 0;
    for (u32 t = n; t > 1; t >>= 1) lg++;
    if (!GapSortTeam<T, Comp>::run(a, n, comp, X, Lbuf, Rbuf, tasks, lg))
        gap_sort_core<T, Comp>(a, 0, n, lg * 2, comp, Lbuf, Rbuf, tasks, 0, nullptr, nullptr);
    X.ar->release(m0);
}

#ifndef K_GAP_WAVES
#define K_GAP_WAVES 4
#endif
// one read: mapGaps + reformCords on its cords in the per-read output slot, with the arena [mine, mine + arena_bytes).  Returns true when the read
// could not be done here (arena, cord slot, deadline): its apxMap cords stay and gap_flag[r] = 1 + (the request that did not fit, KiB).
// lvl: 0 first pass (one wave), 1 team with the middle arena, 2 team with the large one (statistics / profile only).
// hands_off: the caller passes a read it could not do on to another workgroup OF THE SAME LAUNCH (k_gap_all), which then owns gap_flag[r]: two
// workgroups may sit on different XCDs, whose L2s are written back in no particular order at the end of the kernel, so only one of them may
// store to the word.
__device__ bool gap_do_read(const GapArgs &A, u32 r, char *mine, u64 arena_bytes, GapTeam *tm, int team, bool flagged_only, int lvl, bool hands_off = false) {
    u32 nc = A.nout[r];
    u64 L = A.off[r + 1] - A.off[r];
    if (L <= 200 || nc <= 1) { if (!flagged_only) A.gap_flag[r] = 0; return false; }
#ifdef LNR_GAP_POISON
    // diagnostic build: the worker's whole arena is filled with a byte pattern before every read (a result that depends on what an earlier read
    // left in the arena shows up as a dependence on the pattern)
    { const u64 pat = 0x0101010101010101ULL * (u64)(LNR_GAP_POISON & 0xff);
      for (u64 o_ = (u64)(threadIdx.x & 63) * 8; o_ + 8 <= arena_bytes; o_ += 512) *(u64 *)(mine + o_) = pat; WSYNC(); }
#endif
    GArena all; all.init(mine, arena_bytes);
    LeaderScratch *ls = (LeaderScratch *)all.get(sizeof(LeaderScratch));
    u8 *rd = (u8 *)all.get(L + 64), *rc = (u8 *)all.get(L + 64);
    u64 keep_bytes = ((u64)A.cords_cap[r] * 16 + (u64)nc * 64 + 8192) * 2;
    char *kp = (char *)all.get(keep_bytes);
    bool bad = all.ovf != 0;
    u64 ar_want = 0;
    if (!bad) {
        const u8 *src = A.reads + A.off[r];
        if (A.coop) {                                            // (the wave's lanes share the copy; every lane reads the arrays afterwards)
            for (u64 k = threadIdx.x & 63; k < L; k += 64) { u8 b = src[k]; b = b > 4 ? 4 : b; rd[k] = b; rc[L - 1 - k] = b == 4 ? 4 : 3 - b; }
            rd[L + (threadIdx.x & 63)] = 0; rc[L + (threadIdx.x & 63)] = 0;
            WSYNC();
        } else {
            for (u64 k = 0; k < L; k++) { u8 b = src[k]; b = b > 4 ? 4 : b; rd[k] = b; rc[L - 1 - k] = b == 4 ? 4 : 3 - b; }
            for (u32 k = 0; k < 64; k++) { rd[L + k] = 0; rc[L + k] = 0; }
        }
        GArena keep; keep.init(kp, keep_bytes);
        GArena ar; ar.init(mine + all.off, arena_bytes - all.off);
        GapCtx X;
        X.ar = &ar; X.ls = ls; X.read.p = rd; X.read.len = L; X.com.p = rc; X.com.len = L;
        X.g = A.g; X.seq_off = A.seq_off; X.seq_len = A.seq_len;
        u32 nf = A.nf[r];
        X.f1[0].p = A.f1 + A.f1_off[r]; X.f1[0].n = nf; X.f1[1].p = A.f1 + A.f1_off[r] + nf; X.f1[1].n = nf;
        X.gf = A.gf;
        X.gp.f_dup = A.f_dup; X.gp.thd_gap_len_min = A.gap_len_min;
        const bool ext_in = r >= A.ext_from;
        if (ext_in) X.gp.thd_cts_major_limit = 3;
        X.coop = A.coop; X.work_cap = A.work_cap; X.team = team; X.tm = tm; X.hand = (A.coop && !A.big && team <= 1) ? 1 : 0;
        X.deadline = (A.cap_ticks && !flagged_only) ? wall_clock64() + A.cap_ticks : 0;
        u64 *os = A.out_str + A.cords_off[r], *oe = A.out_end + A.cords_off[r];
        GVec<u64> cs, ce; cs.init(&keep, nc * 2 + 64); ce.init(&keep, nc * 2 + 64);
        for (u32 i = 0; i < nc; i++) { cs.push(os[i]); ce.push(oe[i]); }
#ifdef LNR_GAP_DEVPROF
        unsigned long long t_read = wall_clock64();
#endif
        int rc_ = gap_map_gaps(cs, ce, keep, X);
        gap_reform_cords(cs, ce);
#ifdef LNR_GAP_DEVPROF
        if (A.prof && (A.coop ? (threadIdx.x & 63) == 0 : true)) {
            unsigned long long *pp = A.prof + 16 * (lvl);
            t_read = wall_clock64() - t_read;
            for (int k = 0; k < 10; k++) atomicAdd(pp + k, X.prof[k]);
            atomicAdd(A.prof + 63, X.prof[10]); atomicAdd(A.prof + 79, X.prof[11]);      // map along chain, split: streams + join + anchor sort | chain DP + traceback (all launches)
            atomicAdd(pp + 11, t_read); atomicAdd(pp + 12, 1ULL);
            A.prof[96 + r] = t_read | ((unsigned long long)(lvl) << 56);      // per-read time of the launch that did the read
            if (!lvl) { A.prof[96 + A.n + r] = wall_clock64() - t_read; A.prof[96 + 2 * (unsigned long long)A.n + r] = wall_clock64(); }   // first launch: start and end tick of the read
            if (atomicMax(pp + 15, t_read) < t_read) { unsigned long long *ps = A.prof + 48 + 16 * (lvl); for (int k = 0; k < 10; k++) ps[k] = X.prof[k]; ps[10] = r; ps[11] = L; ps[12] = nc; ps[13] = ar.hw; ps[14] = X.dp_t | (X.dp_mode << 60) | (X.dp_fn << 56); A.prof[90 + lvl] = X.dp_n; }
        }
#endif
        bad = rc_ != 0 || ar.ovf || keep.ovf || cs.n > A.cords_cap[r] || cs.n != ce.n;
        ar_want = ar.want > keep.want ? ar.want : keep.want;
        if (!bad) {
            if (!ext_in && X.gp.thd_cts_major_limit == 3 && (A.coop ? (threadIdx.x & 63) == 0 : true)) atomicMin(A.first_ext, r);
            if (!A.probe) {
                for (u32 i = 0; i < cs.n; i++) { os[i] = cs[i]; oe[i] = ce[i]; }
                A.nout[r] = cs.n;
            }
        }
    }
    u64 wkib = (all.want > ar_want ? all.want : ar_want) >> 10;
    if (!(bad && hands_off)) A.gap_flag[r] = bad ? 1u + (u32)(wkib < 0x3fffffffu ? wkib : 0x3fffffffu) : 0u;
    if (A.last && bad) A.read_err[r] = 5;
    return bad;
}
__device__ void gap_worker(const GapArgs &A, GapTeam *tm, int team) {
    u32 worker = A.coop ? blockIdx.x : blockIdx.x * blockDim.x + threadIdx.x;
#ifdef LNR_GAP_DEVPROF
    if (A.prof && threadIdx.x == 0 && !A.big) { unsigned long long c = atomicAdd(A.prof + 94, 1ULL) + 1; atomicMax(A.prof + 95, c); }   // workers of the first launch alive at once
#endif
    char *mine = A.arena + (u64)worker * A.arena_bytes;
    for (;;) {
        u32 r;
        if (A.coop) { r = threadIdx.x == 0 ? atomicAdd(A.next, 1u) : 0u; r = (u32)__shfl((int)r, 0); }
        else r = atomicAdd(A.next, 1u);
        if (A.big) {                                                 // the flagged reads, heaviest first
            if (r >= *A.list_n) break;
            r = A.list[r];
        } else { if (r >= A.n - A.lo) break; r = A.order ? A.order[r] : r + A.lo; }
        if (r >= A.n) break;
        if (A.big && !A.gap_flag[r]) continue;
        if (A.big && threadIdx.x == 0) atomicAdd(A.next + 8, 1u);   // (statistics: reads of the second launch)
        gap_do_read(A, r, mine, A.arena_bytes, tm, team, A.big != 0, A.big + A.last);
    }
#ifdef LNR_GAP_DEVPROF
    if (A.prof && threadIdx.x == 0 && !A.big) atomicAdd(A.prof + 94, ~0ULL);
#endif
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


// ---- the fused first stage.  One launch, workgroups of 16 waves: the first A.nteams are TEAMS (wave 0 runs reads, the others serve its DPs,
// sorts and joins), every other workgroup is 16 independent single-wave workers.  Workgroups are placed in index order, so the teams hold their
// CUs before the single waves flood the rest of the chip.  The reads come ordered by weight (k_gap_weight / k_gap_rank): teams start on the ones
// expected to be heavy at once -- their critical path, not the work, is what the stage waits for -- while the single waves take the light end;
// a single wave that cannot finish a read (arena, deadline) posts it in the queue and a team picks it up.  The stage ends when the single waves
// are through and the queue is empty.  Which worker a read ends up with does not change its result.
__global__ void __attribute__((amdgpu_flat_work_group_size(64 * K_GAP_TEAM, 64 * K_GAP_TEAM))) k_gap_all(GapArgs A) {
    __shared__ GapTeam tm;
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32 *ctl = A.next;
    const u32 m = A.n - A.lo;
    u32 nheavy = *A.n_heavy;
    if (nheavy > m) nheavy = m;
    if (blockIdx.x < A.nteams) {
        if (wave) { gap_team_helper_loop(&tm, (int)wave, K_GAP_TEAM); return; }
        char *mine = A.arena + (u64)blockIdx.x * A.arena2_bytes;
        for (;;) {                                                   // 1. the reads expected to be heavy, heaviest first
            u32 k = lane == 0 ? atomicAdd(ctl + 1, 1u) : 0u;
            k = (u32)__shfl((int)k, 0);
            if (k >= nheavy) break;
            if (lane == 0) atomicAdd(ctl + 16, 1u);
            gap_do_read(A, A.order[k], mine, A.arena2_bytes, &tm, K_GAP_TEAM, false, 1);
        }
        for (;;) {                                                   // 2. what the single waves hand over, until they are all through
            u32 k = lane == 0 ? atomicAdd(ctl + 3, 1u) : 0u;
            k = (u32)__shfl((int)k, 0);
            u32 v = 0;
            for (;;) {
                u32 done = 0, tail = 0;
                if (lane == 0) { v = atomicAdd(A.q + k, 0u); if (!v) { done = atomicAdd(ctl + 5, 0u); tail = atomicAdd(ctl + 2, 0u); } }
                v = (u32)__shfl((int)v, 0); done = (u32)__shfl((int)done, 0); tail = (u32)__shfl((int)tail, 0);
                if (v || (done && k >= tail)) break;
                __builtin_amdgcn_s_sleep(127);
            }
            if (!v) break;
            if (lane == 0) atomicAdd(ctl + 16, 1u);
            gap_do_read(A, v - 1, mine, A.arena2_bytes, &tm, K_GAP_TEAM, true, 1);
        }
        if (lane == 0) tm.cmd = 0;
        __syncthreads();                                             // (A) with the exit command: the helpers leave
    } else {
        u32 worker = (blockIdx.x - A.nteams) * K_GAP_TEAM + wave;
        char *mine = A.arena + (u64)A.nteams * A.arena2_bytes + (u64)worker * A.arena_bytes;
#ifdef LNR_GAP_DEVPROF
        if (A.prof && lane == 0) { unsigned long long c = atomicAdd(A.prof + 94, 1ULL) + 1; atomicMax(A.prof + 95, c); }
#endif
        for (;;) {
            u32 k = lane == 0 ? atomicAdd(ctl, 1u) : 0u;
            k = (u32)__shfl((int)k, 0);
            if (nheavy + k >= m) break;
            u32 r = A.order[nheavy + k];
            bool bad = gap_do_read(A, r, mine, A.arena_bytes, nullptr, 1, false, 0, true);
            if (bad && lane == 0) { u32 slot = atomicAdd(ctl + 2, 1u); atomicExch(A.q + slot, r + 1); }
        }
        if (lane == 0) { u32 e = atomicAdd(ctl + 4, 1u) + 1; if (e == A.nbulk_waves) atomicExch(ctl + 5, 1u); }
#ifdef LNR_GAP_DEVPROF
        if (A.prof && lane == 0) atomicAdd(A.prof + 94, ~0ULL);
#endif
    }
}
hipError_t launch_gap_all(const GapArgs &A, unsigned grid, hipStream_t stream) {
    hipLaunchKernelGGL(k_gap_all, dim3(grid), dim3(64 * K_GAP_TEAM), 0, stream, A);
    return hipGetLastError();
}

// flagged reads of [lo, n) -> list, heaviest first: one workgroup collects them (ballot compaction, index order) and, up to GAP_LIST_SORT_MAX of
// them, sorts (weight descending, index ascending) with a bitonic network in LDS
__global__ void __launch_bounds__(1024) k_gap_order(const u32 *gap_flag, u32 lo, u32 n, u32 *list, u32 *list_n) {
    __shared__ u64 key[GAP_LIST_SORT_MAX];
    __shared__ u32 s_cnt, s_wave[16];
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (u32 base = lo; base < n; base += 1024) {
        u32 r = base + tid;
        u32 f = r < n ? gap_flag[r] : 0u;
        u64 m = __ballot(f != 0);
        if (lane == 0) s_wave[wv] = (u32)__popcll(m);
        __syncthreads();
        u32 before = s_cnt;
        for (u32 w = 0; w < wv; w++) before += s_wave[w];
        if (f) {
            u32 pos = before + (u32)__popcll(m & ((1ULL << lane) - 1));
            list[pos] = r;
            if (pos < GAP_LIST_SORT_MAX) key[pos] = ((u64)(0xffffffffu - f) << 32) | r;      // ascending key = weight descending, then index ascending
        }
        __syncthreads();
        if (tid == 0) { u32 t = 0; for (u32 w = 0; w < 16; w++) t += s_wave[w]; s_cnt += t; }
        __syncthreads();
    }
    const u32 cnt = s_cnt;
    if (tid == 0) *list_n = cnt;
    if (cnt < 2 || cnt > GAP_LIST_SORT_MAX) return;
    u32 P = 1;
    while (P < cnt) P <<= 1;
    for (u32 i = cnt + tid; i < P; i += 1024) key[i] = ~0ULL;
    __syncthreads();
    for (u32 k = 2; k <= P; k <<= 1)
        for (u32 j = k >> 1; j > 0; j >>= 1) {
            for (u32 i = tid; i < P; i += 1024) {
                u32 x = i ^ j;
                if (x > i) {
                    u64 a = key[i], b = key[x];
                    bool up = (i & k) == 0;
                    if ((a > b) == up) { key[i] = b; key[x] = a; }
                }
            }
            __syncthreads();
        }
    for (u32 i = tid; i < cnt; i += 1024) list[i] = (u32)key[i];
}
__global__ void __launch_bounds__(64) k_gap_weight(const u8 *reads, const u64 *off, const u64 *out_str, const u64 *cords_off, const u32 *nout, u32 lo, u32 n, u32 *weight) {
    __shared__ u32 cov[1024];           // one bit per 32 bases of the read (reads of up to 2^20 bases: 32768 cells)
    __shared__ unsigned short bin[4096];
    const u32 r = lo + blockIdx.x, lane = threadIdx.x;
    if (r >= n) return;
    const u64 L = off[r + 1] - off[r];
    const u32 nc = nout[r];
    if (L <= 200 || nc <= 1 || L > (1u << 20)) { if (lane == 0) weight[r] = 0; return; }
    const u32 ncell = (u32)((L + 31) >> 5), nw = (ncell + 31) >> 5;
    for (u32 i = lane; i < nw; i += 64) cov[i] = 0;
    for (u32 i = lane; i < 4096; i += 64) bin[i] = 0;
    __syncthreads();
    const u64 *c = out_str + cords_off[r];
    for (u32 i = 1 + lane; i < nc; i += 64) {                    // a cord covers 96 bases of the read from its y (on the strand the cord is on)
        u64 v = c[i];
        u64 y = cord_y(v);
        if (cord_strand(v)) y = L > y + 96 ? L - y - 96 : 0;
        u32 c0 = (u32)(y >> 5), c1 = (u32)((y + 95) >> 5);
        for (u32 q = c0; q <= c1 && q < ncell; q++) atomicOr(&cov[q >> 5], 1u << (q & 31));
    }
    __syncthreads();
    const u8 *rd = reads + off[r];
    u32 w = 0;
    for (u64 p = lane; p + 9 <= L; p += 64) {
        u32 cell = (u32)(p >> 5);
        if ((cov[cell >> 5] >> (cell & 31)) & 1) continue;
        u32 code = 0, bad = 0;
        for (int k = 0; k < 9; k++) { u32 b = rd[p + k]; bad |= b > 3; code = code * 4 + (b & 3); }
        if (bad) continue;
        u32 h = (code ^ (code >> 7)) & 4095u;
        u32 old = (u32)atomicAdd((unsigned int *)&bin[h & ~1u], (h & 1) ? 0x10000u : 1u);       // (two 16-bit counters per word)
        old = (h & 1) ? old >> 16 : old & 0xffffu;
        w += old < 60000u ? old : 60000u;
    }
    w = wave_sum(w);
    if (lane == 0) weight[r] = w;
}
hipError_t launch_gap_weight(const u8 *reads, const u64 *off, const u64 *out_str, const u64 *cords_off, const u32 *nout, unsigned lo, unsigned n, u32 *weight, hipStream_t stream) {
    if (n > lo) hipLaunchKernelGGL(k_gap_weight, dim3(n - lo), dim3(64), 0, stream, reads, off, out_str, cords_off, nout, lo, n, weight);
    return hipGetLastError();
}
__global__ void __launch_bounds__(1024) k_gap_rank(const u32 *weight, u32 lo, u32 n, u32 *order, u32 *n_heavy, u32 heavy_w) {
    __shared__ u32 bin[1024];
    __shared__ u32 s_heavy;
    const u32 tid = threadIdx.x;
    bin[tid] = 0;
    if (tid == 0) s_heavy = 0;
    __syncthreads();
    auto key = [&](u32 r) -> u32 {                            // 32 bins per power of two, bin 0 = the heaviest
        u32 w = weight[r];
        if (w == 0) return 1023u;
        u32 lg = 31u - (u32)__builtin_clz(w);
        u32 frac = lg >= 5 ? (w >> (lg - 5)) & 31u : (w << (5 - lg)) & 31u;
        u32 k = lg * 32 + frac;
        return 1023u - (k < 1022u ? k : 1022u);
    };
    for (u32 r = lo + tid; r < n; r += 1024) { atomicAdd(&bin[key(r)], 1u); if (weight[r] >= heavy_w) atomicAdd(&s_heavy, 1u); }
    __syncthreads();
    if (tid == 0) { u32 acc = 0; for (u32 b = 0; b < 1024; b++) { u32 c = bin[b]; bin[b] = acc; acc += c; } *n_heavy = s_heavy; }
    __syncthreads();
    for (u32 r = lo + tid; r < n; r += 1024) order[atomicAdd(&bin[key(r)], 1u)] = r;
}
hipError_t launch_gap_rank(const u32 *weight, unsigned lo, unsigned n, u32 *order, u32 *n_heavy, u32 heavy_w, hipStream_t stream) {
    hipLaunchKernelGGL(k_gap_rank, dim3(1), dim3(1024), 0, stream, weight, lo, n, order, n_heavy, heavy_w);
    return hipGetLastError();
}
hipError_t launch_gap_order(const u32 *gap_flag, unsigned lo, unsigned n, u32 *list, u32 *list_n, hipStream_t stream) {
    hipLaunchKernelGGL(k_gap_order, dim3(1), dim3(1024), 0, stream, gap_flag, lo, n, list, list_n);
    return hipGetLastError();
}
hipError_t launch_gap(const GapArgs &A, int team, unsigned grid, hipStream_t stream) {
    if (team) hipLaunchKernelGGL(k_gap_team, dim3(grid), dim3(64 * K_GAP_TEAM), 0, stream, A);
    else hipLaunchKernelGGL(k_gap, dim3(grid), dim3(64), 0, stream, A);
    return hipGetLastError();
}

}  // namespace lnr
