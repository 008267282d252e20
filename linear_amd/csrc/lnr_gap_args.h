// lnr_gap_args.h -- arguments of the gap re-mapper's kernels (lnr_gap_kernels.hip) and their host-side launcher, shared with lnr_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "lnr_hd.h"

namespace lnr {

// ------------------------------------------------------------------ gap re-mapper [f1] ----
// mapGaps + reformCords (Mapper::p_calRecords with -g > 0, mapper.cpp:207-231 / gap.cpp:407-576 / cords.cpp:504-687) on the final
// cords of every read, in place in the per-read output slots.  One lane = one worker with an arena of its own; workers take reads
// from a shared counter (the work per read ranges from nothing to dozens of k-mer joins).  A read whose gaps outgrow the arena (or
// whose new cords outgrow its slot), or whose chain DPs go over the work budget (a read of N runs joins into 10^5 anchors with
// thousands of predecessors each), keeps its apxMap cords and is flagged in gap_flag; the second and the third launch (big = 1) take
// only the flagged reads, one WAVE per read with a larger (8 MB) and a large (64 MB) arena: all lanes run the read's code with the same data (stores of one value to one
// address), and the chain DP deals the predecessors of an anchor over the lanes (gap_chain_anchors).  What is still flagged
// afterwards is reported through read_err.
struct GapArgs {
    const u8 *g; const u64 *seq_off, *seq_len; GenomeFeat gf;
    const u8 *reads; const u64 *off; u32 n;
    const u32 *nf; const u64 *f1_off; const F96 *f1;
    u64 *out_str, *out_end; const u64 *cords_off; const u32 *cords_cap; u32 *nout; i32 *read_err; u32 *gap_flag;
    unsigned long long *prof;   // LNR_GAP_DEVPROF builds: [launch][16] ticks per phase, [15] = the slowest read
    char *arena; u64 arena_bytes; u32 *next; u32 gap_len_min; int f_dup; u64 work_cap;
    int coop;   // one wave per read (the launches after the first; LNR_GAP_MODE=1: the first too)
    int big;    // only the reads an earlier launch flagged
    int last;   // what this launch cannot do either is an error of the read
    // The read stream's state (DESIGN 5c "stream state"): the reference keeps ONE GapParms per thread for the whole run and the first
    // mapExtend / mapExtends of the stream leaves thd_cts_major_limit = 3 behind for every later read (mapper.cpp:233-237,447,
    // gap_util.cpp:4052,4091; read by chainTiles :1188).  Reads [lo, n) are processed; those with index >= ext_from start "extended".
    // probe: nothing is written back -- the launch only finds the first read that extends (atomicMin into *first_ext).
    u32 lo; u32 ext_from; int probe; u32 *first_ext;
    // the launches for the flagged reads take them from a list ordered heaviest first (k_gap_order: gap_flag[r] - 1 = the arena request that did not
    // fit, in KiB) -- launched in index order the read that alone sets the launch's duration started as late as a third of the way in
    const u32 *list; const u32 *list_n;
    // the fused first stage (k_gap_all): nteams workgroups are teams (arena2_bytes each, at the start of `arena`), the others hold 16 single-wave
    // workers each (arena_bytes each, behind the teams' arenas).  Teams take the reads expected to be heavy (the first *n_heavy of `order`) and
    // whatever the single waves hand over through the queue q (ctl words in `next`: 0 next light read, 1 next heavy read, 2 queue tail, 3 queue
    // head, 4 single waves through, 5 all of them through, 16 reads done by teams)
    u32 nteams, nbulk_waves; u64 arena2_bytes; const u32 *n_heavy; u32 *q;
    u64 cap_ticks;           // first launch: a read still busy after this many 10 ns ticks is left to the team launch (0 = no limit)
    const u32 *order;        // first launch: reads [lo, n) by decreasing uncovered length (k_gap_rank) -- the long ones start first, the launch's tail is short ones
};

#ifndef K_GAP_TEAM
#define K_GAP_TEAM 16         // waves per read of the launches for the flagged reads (wave 0 = the worker): one CU per read, measured 564 vs 724 ms for the slowest read with 8
#endif
// launches k_gap (team = 0: one wave per workgroup) or k_gap_team (K_GAP_TEAM waves per workgroup) on `grid` workgroups
hipError_t launch_gap(const GapArgs &A, int team, unsigned grid, hipStream_t stream);
hipError_t launch_gap_all(const GapArgs &A, unsigned grid, hipStream_t stream);     // the fused first stage: A.nteams team workgroups + (grid - A.nteams) x 16 single waves
// the flagged reads of [lo, n) into list[0 .. *list_n), heaviest first (one workgroup; more than GAP_LIST_SORT_MAX stay in index order)
#define GAP_LIST_SORT_MAX 4096
hipError_t launch_gap_order(const u32 *gap_flag, unsigned lo, unsigned n, u32 *list, u32 *list_n, hipStream_t stream);
// weight[r] = pairs of equal 9-mers inside the stretches of read r that its cords leave uncovered (hashed into 4096 bins): what the gap re-mapper's
// k-mer joins and chain DPs will be busy with -- a read over a tandem repeat or a homopolymer run scores 10^5 .. 10^7, an ordinary one a few hundred
hipError_t launch_gap_weight(const u8 *reads, const u64 *off, const u64 *out_str, const u64 *cords_off, const u32 *nout, unsigned lo, unsigned n, u32 *weight, hipStream_t stream);
// order[0 .. n - lo) = the reads of [lo, n) by decreasing weight (1024 logarithmic bins); *n_heavy = how many of them weigh heavy_w or more
hipError_t launch_gap_rank(const u32 *weight, unsigned lo, unsigned n, u32 *order, u32 *n_heavy, u32 heavy_w, hipStream_t stream);

}  // namespace lnr
