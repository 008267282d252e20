// lnr_api.hip -- C ABI (include/linear_amd.h) of the MI355X filter hot path: context, device memory,
// index build orchestration and the per-batch kernel pipeline.  Device code: lnr_kernels.hip + lnr_hd.h.
//
// There is no CPU execution path in this library: every stage runs in a HIP kernel, and every entry
// point fails (LNR_ERR_NO_DEVICE / LNR_ERR_HIP) when no GPU is usable.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>   // device-wide radix sort of the HIndex build (-i 2, once per index): AMD's native primitives library, no CUB layer
#include "lnr_kernels.hip"
#include "lnr_gap_args.h"
#include "../../include/linear_amd.h"

#include <algorithm>
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace lnr;

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool ensure(size_t bytes) {
        if (bytes <= cap && p) return true;
        bool grown = p != nullptr;          // a buffer that had to grow once gets half as much again: batch-dependent sizes creep, and
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }   // re-allocating GBs in the middle of a run costs hundreds of ms
        size_t nc = bytes + (grown ? bytes / 2 : bytes / 8) + 4096;
        if (hipMalloc(&p, nc) != hipSuccess) { p = nullptr; cap = 0; (void)hipGetLastError(); return false; }
        cap = nc;
        return true;
    }
    // pinned host staging for uploads into this buffer: a copy from pageable memory is staged by the runtime and was
    // measured to block the host for ~7 ms now and then; from pinned memory it is a plain asynchronous DMA
    void *hp = nullptr;
    size_t hcap = 0;
    void *host_stage(size_t bytes) {
        if (bytes <= hcap && hp) return hp;
        if (hp) { (void)hipHostFree(hp); hp = nullptr; hcap = 0; }
        size_t nc = bytes + bytes / 8 + 4096;
        if (hipHostMalloc(&hp, nc, hipHostMallocDefault) != hipSuccess) { hp = nullptr; hcap = 0; (void)hipGetLastError(); return nullptr; }
        hcap = nc;
        return hp;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; if (hp) (void)hipHostFree(hp); hp = nullptr; hcap = 0; }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    template <class T> T *as() const { return (T *)p; }
    void swap(DevBuf &o) { std::swap(p, o.p); std::swap(cap, o.cap); std::swap(hp, o.hp); std::swap(hcap, o.hcap); }
};

// pinned host staging (device-to-host copies from pageable memory run at a fraction of the link rate)
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool ensure(size_t bytes) {
        if (bytes <= cap && p) return true;
        bool grown = p != nullptr;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        size_t nc = bytes + (grown ? bytes / 2 : bytes / 8) + 4096;
        if (hipHostMalloc(&p, nc, hipHostMallocDefault) != hipSuccess) { p = nullptr; cap = 0; (void)hipGetLastError(); return false; }
        cap = nc;
        return true;
    }
    ~PinBuf() { if (p) (void)hipHostFree(p); }
    PinBuf() = default;
    PinBuf(const PinBuf &) = delete;
    PinBuf &operator=(const PinBuf &) = delete;
    template <class T> T *as() const { return (T *)p; }
};

// Device-to-host readbacks of the batch pipeline (counts, flags: a few MB per batch) land in pinned memory and are copied out
// after the stream sync.  They are written there by a kernel's stores, not by a DMA copy: while the copy stream uploads the next
// batch (lnr_filter_submit, 1 GB, 18 ms) a DMA readback on the compute stream was measured to queue behind that upload -- the
// seed stage took 24 ms instead of 6.7 -- and a copy into pageable memory is staged by the runtime on top of that.
struct Readback {
    struct Item { void *dst; size_t off, bytes; };
    PinBuf *pin = nullptr;
    std::vector<Item> items;
    size_t used = 0;
    bool begin(PinBuf &p, size_t total) { pin = &p; items.clear(); used = 0; return p.ensure(total + 64 * 8); }
    hipError_t add(void *dst, const void *dsrc, size_t bytes, hipStream_t st) {   // bytes: a multiple of 4, dsrc 4-byte aligned
        size_t o = (used + 15) & ~(size_t)15;
        used = o + bytes;
        items.push_back({dst, o, bytes});
        if (!bytes) return hipSuccess;
        u64 nw = bytes / 4;
        hipLaunchKernelGGL(lnr::k_words_out, dim3((u32)std::min<u64>((nw + 255) / 256, 1024)), dim3(256), 0, st, (const u32 *)dsrc, (u32 *)((char *)pin->p + o), nw);
        return hipGetLastError();
    }
    void finish() { for (auto &i : items) if (i.bytes) memcpy(i.dst, (char *)pin->p + i.off, i.bytes); }
};

// pinned host -> device for the small per-batch tables, as a kernel's loads (same reason as Readback: a DMA copy on the compute
// stream queues behind the copy stream's upload of the next batch)
static inline hipError_t words_in(void *d_dst, const void *h_pinned, size_t bytes, hipStream_t st) {
    u64 nw = (bytes + 3) / 4;
    if (!nw) return hipSuccess;
    hipLaunchKernelGGL(lnr::k_words_out, dim3((u32)std::min<u64>((nw + 255) / 256, 1024)), dim3(256), 0, st, (const u32 *)h_pinned, (u32 *)d_dst, nw);
    return hipGetLastError();
}

static const u64 SEQ_PAD = 64;
static inline u64 align_up(u64 v, u64 a) { return (v + a - 1) / a * a; }

struct Timer {
    hipEvent_t a = nullptr, b = nullptr;
    void init() { (void)hipEventCreate(&a); (void)hipEventCreate(&b); }
    void destroy() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    void start(hipStream_t s) { (void)hipEventRecord(a, s); }
    void stop(hipStream_t s) { (void)hipEventRecord(b, s); }
    double ms() { float f = 0; if (hipEventSynchronize(b) != hipSuccess) return 0; (void)hipEventElapsedTime(&f, a, b); return f; }
};

}  // namespace

struct lnr_ctx {
    lnr_opts opts;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // ---- index
    bool has_index = false;
    lnr_index_info info{};
    std::vector<u64> seq_len, seq_off, f2_off;
    u32 nbins = 0;
    size_t job_lds_bytes = 6 * 1024;    // LDS half of k_job's two-level arena (LNR_JOB_LDS_KB overrides, for tuning): 2.6 KB static + 6 KB x 16 workgroups fit a CU's 160 KB (measured: 5 KB +2 %, 7 KB +1 %)
    size_t job_stage_bytes = 0;         // LDS stage of the blocked DP's predecessor window in the fused k_job (LNR_JOB_STAGE_KB; measured slower, off)
    u32 heavy_lds_kb = 48;              // LDS arena of k_job_heavy (LNR_HEAVY_LDS_KB)
    u32 mid_cap = 6144, mid_lds_kb = 24;   // reads with at least this many anchors run on 4 waves (k_job_mid: the DP is dealt over the waves); LNR_MID_CAP, LNR_MID_LDS_KB
    u32 heavy_cap = 0xffffffffu;        // reads with at least this many anchors (after the Y filter) take the 16-wave path (LNR_HEAVY_CAP overrides)
    u32 dp_split_cap = 0xffffffffu, dp_split_cap_r1 = 0xffffffffu;   // reads with at least this many anchors take the split path pre -> 16-wave DP -> post (LNR_DP_SPLIT_CAP, LNR_DP_SPLIT_CAP_R1)
    u32 heavy_cap_r1 = 7000, mid_cap_r1 = 3000;   // the same cuts for the re-map round (LNR_HEAVY_CAP_R1, LNR_MID_CAP_R1)
    bool lane_bulk_first = true;         // two lanes: which lane goes through the re-map round first (LNR_LANE_ORDER=heavy|bulk)
    u32 stop_after = 0;                  // diagnostic: LNR_STOP_AFTER (see JobArgs)
    u32 mid_waves = 0;                   // waves per read of the middle class (LNR_MID_WAVES=2|4; 0 = 2 in round 0 on a populated table, else 4)
    bool mid_cap_env = false;            // LNR_MID_CAP given: no density-dependent default
    bool post_split = false;             // a11-a16 in k_post, one lane per read (LNR_POST_SPLIT=0: fused job kernels)
    int seed_bm = -1;                    // bucket bitmap in the seed kernel: -1 = by table density, 0 / 1 forced (LNR_SEED_BM)
    u32 prep_threads = 256;             // workgroup size of k_prep (LNR_PREP_THREADS: 64, 128 or 256)
    u32 prep_grid = 4096;               // workgroups of k_prep (LNR_PREP_GRID): they loop over the reads
    u32 bulk_delay_ticks = 10000;       // head start (100 MHz ticks) of the multi-wave kernels over the bulk kernel (LNR_BULK_DELAY_US)
    u32 split_cap = 0xffffffffu;               // reads with at least this many anchors form the "heavy lane": their re-map round starts
                                        // while the bulk of the batch is still in round 0 (LNR_SPLIT_CAP; 0xffffffff = one lane)
    // Two lanes of streams: lane 0 = heavy reads, lane 1 = the bulk.  s_multi carries the multi-wave kernels (and the
    // lane's seed / tail launches), s_bulk the single-wave kernel of the same launch.
    hipStream_t s_multi[2] = {nullptr, nullptr}, s_bulk[2] = {nullptr, nullptr}, s_tail = nullptr;   // s_tail: early tail B of the reads that skip the re-map round
    hipEvent_t ev_fork[2] = {nullptr, nullptr}, ev_join[2] = {nullptr, nullptr}, ev_start = nullptr, ev_lane[2] = {nullptr, nullptr}, ev_prep = nullptr, ev_f1 = nullptr;
    u32 cap_scale = 1;      // per-read capacities (cords, gaps) x this: raised for the re-run of a batch in which a read overflowed
    u32 cap_shrink = 1;     // diagnostic (LNR_CAP_SHRINK): capacities / this, to exercise that re-run
    u32 overflow_reruns = 0;
    u32 seed_lds_pad = 0;   // diagnostic (LNR_SEED_LDS_PAD): dynamic LDS the seed kernel does not use, to lower its waves per CU
    DevBuf hx_nkeys, hx_nvals; u32 hx_nnodes = 0; u64 hx_empty_dir = 0;   // HIndex (-i 2): dir = hdir[2^18] (head of the block of X, -1: none), hs = ysa, nodes of the large blocks
    DevBuf gap_arena, gap_flag, gap_next, d_seq_len, gap_prof, gap_first, gap_list, gap_rank, gap_weight;
    int gap_ext = 0;        // the read stream's state: 1 once a read of this context's stream went through mapExtend / mapExtends (lnr_gap_stream)   // the gap re-mapper (-g > 0): arenas of its workers, per-read retry flags, the two work counters
    DevBuf bh;              // header words of the bucket lines as a dense table (k_ix_lines; LNR_SEED_BH=0 turns it off for A/B runs)
    int use_bh = 0;        // (measured on the GRCh38 stand-in, same box, two runs each: 3.07 - 3.12 ms per launch with the table, 2.97 - 3.00 without: the early line fetch warms L2 / MALL for the DMA)
    DevBuf g, dir, hs, f2, d_seq_off, d_f2_off, bm, bl, ov;   // derived from dir / hs on every GPU: bm = bucket-non-empty bitmap, bl = bucket lines, ov = their aligned overflow lines (k_ix_lines)
    // ---- batch inputs / per-read arrays
    // host-buffer entry points: two input slots, so that the upload of the next batch (copy stream) runs under the kernels of
    // the current one (lnr_filter_submit / lnr_filter_wait)
    DevBuf in_reads[3], in_off[3];       // three input slots: one batch computed ahead + two uploads pending (lnr_filter_submit / lnr_filter_wait)
    PinBuf h_off[3];
    hipStream_t s_copy = nullptr, s_down = nullptr;   // uploads / result downloads, each on a stream of its own
    hipEvent_t ev_in[3] = {nullptr, nullptr, nullptr}, ev_down = nullptr, ev_done = nullptr;
    // the batch that has been computed but not handed out yet (lnr_filter_wait computes the NEXT submitted batch while the results of the
    // one it returns travel to the host), and the second set of device result buffers it lives in
    struct Pre { bool valid = false; lnr_status st = LNR_OK; u32 n = 0; u64 tot = 0; lnr_stats stats; std::vector<u64> coff; const void *d_str = nullptr, *d_end = nullptr; std::string err; } pre;
    DevBuf rB_off, rB_str, rB_end;
    lnr_stats stats_pub;                 // statistics of the batch handed out last (what lnr_last_stats reports)
    u32 in_n[3] = {0, 0, 0};
    int in_head = 0, in_count = 0;
    DevBuf rlen, rks, nf, f1_off, f1, pk, nm, pk_off;
    DevBuf cords, out_str, out_end, cords_off, cords_cap, ncords, nout, read_err;
    DevBuf gaps, gaps_off, gaps_cap, ngaps, remap, gdense, gcursor, gpos;
    PinBuf h_gaps, h_flags;             // pinned staging of the tail-A results
    // ---- jobs: a JobSet is one seeded job list (device arrays + host mirrors); a Launch is the per-launch state of the
    // job kernels (order, scratch); a TailBuf the per-launch state of a tail kernel.  Two of each: one per lane.
    struct JobSet {
        DevBuf j_read, j_str, j_end, j_mode, j_cap, j_look, j_anc_off, j_nanc, grp_beg, anchors, seed_ctl;
        std::vector<u32> cap, look, nanc;
        std::vector<u64> anc_off;
        u32 est_x16 = 64;               // anchors per sample x 16 the seed kernel sizes a job's first segment with (learned from the last batch)
        u64 cap_slots = 0;              // anchor buffer capacity (u64 slots), sticky
        Timer t_seed;
        PinBuf h_rb;
    } js[2];
    // (host vectors that feed asynchronous uploads live here, not on the stack: the launch functions return before the copy ran)
    struct Launch { DevBuf grp_order, j_scr_off, job_scr, jstate; std::vector<u32> h_order; std::vector<u64> h_scr_off; } ln[2];
    struct TailBuf { DevBuf off, cap, scr, list; std::vector<u64> h_off; std::vector<u32> h_cap, h_list; } tb[3];
    PinBuf h_rb[4];                     // pinned landing zones of the small readbacks (per stream that reads back)
    DevBuf prof, tl; u32 tl_round = 0, tl_n[4] = {0, 0, 0, 0}; u32 tl_nh[4] = {0, 0, 0, 0};
    // ---- results
    DevBuf r_off, r_str, r_end;
    std::vector<u64> h_cord_off, h_anchor_off, h_anchors, h_gap_off, h_gap_pairs;
    std::vector<u64> last_gaps_off;      // per-read offsets into ctx->gaps of the last batch (capacity layout)
    PinBuf h_cords_str2[2], h_cords_end2[2], h_up[2];   // results land in pinned memory (DMA at link rate, no page faults), two result slots taken in turn: the
    std::vector<u64> h_cord_off2[2]; int res_slot = 0;    // arrays handed out stay valid until the SECOND next result (a writer thread formats batch k while k + 1 runs); h_up: upload staging ring
    hipEvent_t ev_up[2] = {nullptr, nullptr};
    std::vector<u32> dbg_r0w;   // round-0 anchors per read (LNR_DEBUG_R1 diagnostic)
    u32 last_n = 0;
    u64 last_ncords = 0;
    lnr_stats stats{};
    Timer t_prep, t_job, t_tail, t_total, t_gap;
    int gap_mode = 1, gap_team = 1; u32 gap_waves = 16384, gap_arena2_mb = 64;   // LNR_GAP_TEAM=0: one wave per flagged read, no helper waves   // LNR_GAP_MODE=1: the first launch of k_gap runs one wave per read as well (LNR_GAP_WAVES of them)
    int gap_fused = 1; u32 gap_teams = 96, ncu = 0;   // the fused first stage (k_gap_all): LNR_GAP_FUSED=0 falls back to the three launches; LNR_GAP_TEAMS = team workgroups
    u32 gap_heavy_w = 60000;   // weight (k_gap_weight) from which a read is expected to need a team (LNR_GAP_HEAVY_W)
    u32 gap_cap_ms = 0;    // first launch of the gap re-mapper: milliseconds after which a read is handed to the team launch (LNR_GAP_CAP_MS, 0 = never)
    u64 gap_work_cap = 3000000;   // pair evaluations of the chain DPs one lane spends on a read before the read goes to the wave-per-read launch (LNR_GAP_WORK_CAP)
};

namespace {

#define HIPCK(call)                                                                                  \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            char b_[256];                                                                            \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            ctx->err = b_;                                                                           \
            (void)hipGetLastError();                                                                 \
            return LNR_ERR_HIP;                                                                      \
        }                                                                                            \
    } while (0)
#define ENSURE(buf, bytes)                                                                           \
    do {                                                                                             \
        if (!(buf).ensure(bytes)) {                                                                  \
            char b_[160];                                                                            \
            snprintf(b_, sizeof b_, "device allocation of %zu bytes failed (%s:%d)", (size_t)(bytes), __FILE__, __LINE__); \
            ctx->err = b_;                                                                           \
            return LNR_ERR_NOMEM;                                                                    \
        }                                                                                            \
    } while (0)
#define KCHECK() HIPCK(hipGetLastError())
#define HIPCK_CTX(c, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { (c)->err = std::string(#call) + ": " + hipGetErrorString(e__); return LNR_ERR_HIP; } } while (0)

template <class T>
lnr_status upload(lnr_ctx *ctx, DevBuf &b, const std::vector<T> &v) {
    ENSURE(b, std::max<size_t>(v.size() * sizeof(T), 16));
    if (!v.empty()) {
        void *h = b.host_stage(v.size() * sizeof(T));
        if (!h) { ctx->err = "pinned host allocation failed"; return LNR_ERR_NOMEM; }
        memcpy(h, v.data(), v.size() * sizeof(T));
        HIPCK(words_in(b.p, h, v.size() * sizeof(T), ctx->stream));
    }
    return LNR_OK;
}

// exclusive scan of n int32 on the device (in -> out), tmp = block sums
lnr_status dev_scan_i32(lnr_ctx *ctx, const i32 *in, i32 *out, u64 n, DevBuf &tmp) {
    u32 nblk = (u32)((n + SCAN_BLK - 1) / SCAN_BLK);
    ENSURE(tmp, (size_t)nblk * 4 + 16);
    hipLaunchKernelGGL(k_scan_blk, dim3(nblk), dim3(SCAN_TPB), 0, ctx->stream, in, out, n, tmp.as<i32>());
    KCHECK();
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, ctx->stream, tmp.as<i32>(), nblk);
    KCHECK();
    hipLaunchKernelGGL(k_scan_add, dim3(nblk), dim3(SCAN_TPB), 0, ctx->stream, out, n, tmp.as<i32>());
    KCHECK();
    return LNR_OK;
}

// The seed kernel's view of the DIndex, derived from dir / hs on this GPU: bucket bitmap, bucket lines and their overflow lines.
lnr_status build_seed_view(lnr_ctx *ctx) {
    u64 nb = ctx->info.dir_len - 1, nwords = (((nb + (1u << BM_GROUP_LOG2) - 1) >> BM_GROUP_LOG2) + 31) / 32;
    ENSURE(ctx->bm, nwords * 4 + 16);
    hipLaunchKernelGGL(k_ix_bitmap, dim3((u32)((nwords + 255) / 256)), dim3(256), 0, ctx->stream, ctx->dir.as<i32>(), nb, ctx->bm.as<u32>());
    KCHECK();
    DevBuf ovoff, tmp;                                        // overflow lines per bucket -> first overflow line of every bucket
    ENSURE(ovoff, (nb + 1) * 4 + 16);
    hipLaunchKernelGGL(k_ix_ovcount, dim3((u32)((nb + 1 + 255) / 256)), dim3(256), 0, ctx->stream, ctx->dir.as<i32>(), nb, ovoff.as<i32>());
    KCHECK();
    lnr_status st = dev_scan_i32(ctx, ovoff.as<i32>(), ovoff.as<i32>(), nb + 1, tmp);
    if (st != LNR_OK) return st;
    i32 nov = 0;
    HIPCK(hipMemcpyAsync(&nov, ovoff.as<i32>() + nb, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCK(hipStreamSynchronize(ctx->stream));
    if (nov < 0) { ctx->err = "overflow lines of the bucket view exceed 2^31"; return LNR_ERR_LIMIT; }
    ENSURE(ctx->ov, ((u64)nov + 1) * 128);
    ENSURE(ctx->bl, nb * 128);
    if (ctx->use_bh) ENSURE(ctx->bh, nb * 8 + 16);
    hipLaunchKernelGGL(k_ix_lines, dim3((u32)((nb * 8 + 255) / 256)), dim3(256), 0, ctx->stream, ctx->dir.as<i32>(), ctx->hs.as<u64>(), ovoff.as<i32>(), nb, ctx->bl.as<ulonglong2>(),
                       ctx->ov.as<u64>(), ctx->use_bh ? ctx->bh.as<u64>() : (u64 *)nullptr);
    KCHECK();
    HIPCK(hipStreamSynchronize(ctx->stream));                 // ovoff / tmp go out of scope
    return LNR_OK;
}

// ---- HIndex (-i 2): lookup tables from ysa (ctx->hs), at build and at adopt
lnr_status hx_derive(lnr_ctx *ctx) {
    u64 n = ctx->info.hs_len;
    if (n < 2) { ctx->err = "empty HIndex"; return LNR_ERR_ARG; }
    ctx->hx_empty_dir = n - 2;
    ENSURE(ctx->dir, ctx->info.dir_len * 4);
    HIPCK(hipMemsetAsync(ctx->dir.p, 0xff, ctx->info.dir_len * 4, ctx->stream));
    DevBuf flag, tmp;
    ENSURE(flag, (n + 1) * 4 + 16);
    hipLaunchKernelGGL(k_hx_derive, dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->hs.as<u64>(), n, ctx->dir.as<i32>(), flag.as<i32>());
    KCHECK();
    HIPCK(hipMemsetAsync(flag.as<i32>() + n, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_hx_nodes_mark, dim3(1u << HX_XBITS), dim3(256), 0, ctx->stream, ctx->hs.as<u64>(), ctx->dir.as<i32>(), flag.as<i32>());
    KCHECK();
    DevBuf excl;
    ENSURE(excl, (n + 1) * 4 + 16);
    lnr_status st = dev_scan_i32(ctx, flag.as<i32>(), excl.as<i32>(), n + 1, tmp);
    if (st != LNR_OK) return st;
    i32 nn = 0;
    HIPCK(hipMemcpyAsync(&nn, excl.as<i32>() + n, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCK(hipStreamSynchronize(ctx->stream));
    ctx->hx_nnodes = (u32)nn;
    ENSURE(ctx->hx_nkeys, (size_t)std::max(nn, 1) * 8);
    ENSURE(ctx->hx_nvals, (size_t)std::max(nn, 1) * 4);
    if (nn) {
        DevBuf k_in, v_in, cub;
        ENSURE(k_in, (size_t)nn * 8); ENSURE(v_in, (size_t)nn * 4);
        hipLaunchKernelGGL(k_hx_nodes_fill_blk, dim3(1u << HX_XBITS), dim3(256), 0, ctx->stream, ctx->hs.as<u64>(), ctx->dir.as<i32>(), flag.as<i32>(), excl.as<i32>(), k_in.as<u64>(), v_in.as<u32>());
        KCHECK();
        size_t tb = 0;   // stable sort by (X, Y20): equal keys keep ysa order, the lookup takes the first
        HIPCK(rocprim::radix_sort_pairs(nullptr, tb, k_in.as<u64>(), ctx->hx_nkeys.as<u64>(), v_in.as<u32>(), ctx->hx_nvals.as<u32>(), (size_t)nn, 0u, (unsigned)(20 + HX_XBITS), ctx->stream));
        ENSURE(cub, tb + 16);
        HIPCK(rocprim::radix_sort_pairs(cub.p, tb, k_in.as<u64>(), ctx->hx_nkeys.as<u64>(), v_in.as<u32>(), ctx->hx_nvals.as<u32>(), (size_t)nn, 0u, (unsigned)(20 + HX_XBITS), ctx->stream));
        HIPCK(hipStreamSynchronize(ctx->stream));
    }
    HIPCK(hipStreamSynchronize(ctx->stream));
    return LNR_OK;
}
// ---- HIndex build (createHIndex, index_util.cpp:1463-1476): samples per -t chunk, blocks by X, bodies descending, ysa
lnr_status build_hindex(lnr_ctx *ctx, const u64 *len, u32 nseq, u32 T) {
    std::vector<HxPiece> pieces;
    std::vector<u32> chunk_first;                                        // index of every chunk's first piece (+ end sentinel)
    u64 stage = 0;
    for (u32 j = 0; j < nseq; j++) {
        if (len[j] < HX_SPAN) { ctx->err = "sequence shorter than the HIndex shape (17 bases)"; return LNR_ERR_LIMIT; }
        u64 npos = len[j] - HX_SPAN + 1, size2 = npos / T;
        for (u32 t = 0; t < T; t++) {                                    // __createHsArray :745-760
            u64 chunk, start;
            if (t < npos - size2 * T) { chunk = size2 + 1; start = (size2 + 1) * t; }
            else { chunk = size2; start = len[j] + 1 - HX_SPAN - size2 * (T - t); }
            chunk_first.push_back((u32)pieces.size());
            u64 u = start;
            do {                                                         // (a chunk of no positions still has its hashInit: one empty piece)
                HxPiece c; c.seq_off = ctx->seq_off[j]; c.seq_id = j; c.start = start; c.chunk = chunk;
                c.u = u; c.v = std::min(u + HX_PIECE, start + chunk); if (c.v < c.u) c.v = c.u;
                if (start + chunk - c.v < 64) c.v = start + chunk;       // no sliver at the end: the last piece holds the chunk's end rule
                c.first = u == start ? 1 : 0; c.out_base = stage; c.kt0 = ~0ULL; c.kinit = start; c.nc = ~0ULL; c.slen = len[j];
                stage += (c.v - c.u) / HX_STEP + 4;
                pieces.push_back(c);
                u = c.v;
            } while (u < start + chunk);
        }
    }
    chunk_first.push_back((u32)pieces.size());
    u32 npc = (u32)pieces.size(), nchk = (u32)chunk_first.size() - 1;
    DevBuf d_pc, fileX, body, d_po, d_cp, d_fn, d_fc, d_tc, Xs, bodies, Xs2, bodies2, cub, flag, cntX, tmp;
    lnr_status s;
    if ((s = upload(ctx, d_pc, pieces)) != LNR_OK) return s;
    ENSURE(d_fn, (size_t)npc * 8 + 16); ENSURE(d_fc, (size_t)npc * 8 + 16); ENSURE(d_tc, (size_t)npc * 8 + 16);
    hipLaunchKernelGGL(k_hx_pre, dim3((npc + 63) / 64), dim3(64), 0, ctx->stream, ctx->g.as<u8>(), d_pc.as<HxPiece>(), npc, d_fn.as<u64>(), d_fc.as<u64>(), d_tc.as<u64>());
    KCHECK();
    {   // what a piece needs from its neighbours: where a jump over an N cluster lands behind it (nc), the chunk's first clean window
        // (kinit: the state its hashInit leaves) and where the first N enters a window of the chunk (kt0)
        std::vector<u64> fn(npc), fc(npc), tc(npc);
        HIPCK(hipMemcpyAsync(fn.data(), d_fn.p, (size_t)npc * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCK(hipMemcpyAsync(fc.data(), d_fc.p, (size_t)npc * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCK(hipMemcpyAsync(tc.data(), d_tc.p, (size_t)npc * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCK(hipStreamSynchronize(ctx->stream));
        u64 carried = ~0ULL;
        for (i64 q = (i64)npc - 1; q >= 0; q--) {                                // pieces are in sequence order, positions ascending
            bool seq_last = q == (i64)npc - 1 || pieces[q + 1].seq_id != pieces[q].seq_id;
            if (seq_last) carried = tc[q];                                        // (position len is always clean: padding)
            pieces[q].nc = carried;
            if (fc[q] != ~0ULL) carried = fc[q];
        }
        for (u32 c = 0; c < nchk; c++) {
            u32 p0 = chunk_first[c], p1 = chunk_first[c + 1];
            u64 kt0 = ~0ULL;
            for (u32 q = p0; q < p1; q++) if (fn[q] != ~0ULL) { kt0 = fn[q] - 16; break; }
            u64 kinit = fc[p0] != ~0ULL ? fc[p0] : pieces[p0].nc;
            for (u32 q = p0; q < p1; q++) { pieces[q].kt0 = kt0; pieces[q].kinit = kinit; }
        }
        if ((s = upload(ctx, d_pc, pieces)) != LNR_OK) return s;
    }
    ENSURE(fileX, stage * 4 + 16); ENSURE(body, stage * 8 + 16); ENSURE(d_po, (size_t)npc * sizeof(HxPieceOut) + 16);
    hipLaunchKernelGGL(k_hx_piece, dim3((npc + 63) / 64), dim3(64), 0, ctx->stream, ctx->g.as<u8>(), d_pc.as<HxPiece>(), npc, fileX.as<u32>(), body.as<u64>(), d_po.as<HxPieceOut>());
    KCHECK();
    std::vector<HxPieceOut> po(npc);
    HIPCK(hipMemcpyAsync(po.data(), d_po.p, (size_t)npc * sizeof(HxPieceOut), hipMemcpyDeviceToHost, ctx->stream));
    HIPCK(hipStreamSynchronize(ctx->stream));
    std::vector<HxCopy> cp(npc);
    u64 n = 0;
    for (u32 c = 0; c < nchk; c++) {
        u32 p0 = chunk_first[c], p1 = chunk_first[c + 1];
        bool have_prev = false; u32 prevX = 0, endX = 0; bool any_hashed = false; i64 last_emit = -1;
        for (u32 q = p0; q < p1; q++) {
            HxCopy k; k.src = pieces[q].out_base; k.n = po[q].cnt; k.patch = 0; k.patchX = 0; k.pad = 0;
            if (po[q].cnt) {
                if (q != p0 && have_prev && po[q].firstX == prevX) { k.src++; k.n--; }   // first sample of the piece repeats the X of the sample before it
                have_prev = true; prevX = po[q].lastX;
            }
            k.dst = n; n += k.n;
            if (k.n) last_emit = q;
            if (po[q].hashed) { any_hashed = true; endX = po[q].endX; }
            cp[q] = k;
        }
        if (last_emit >= 0 && any_hashed) { cp[(u32)last_emit].patch = 1; cp[(u32)last_emit].patchX = endX; }   // :801
    }
    if (n >= (1ULL << 31) - 4) { ctx->err = "too many HIndex samples"; return LNR_ERR_LIMIT; }
    if (n == 0) { ctx->err = "no HIndex samples"; return LNR_ERR_ARG; }
    ctx->info.n_samples = n;
    if ((s = upload(ctx, d_cp, cp)) != LNR_OK) return s;
    ENSURE(Xs, n * 4 + 16); ENSURE(bodies, n * 8 + 16); ENSURE(Xs2, n * 4 + 16); ENSURE(bodies2, n * 8 + 16);
    hipLaunchKernelGGL(k_hx_compact, dim3(npc), dim3(256), 0, ctx->stream, d_cp.as<HxCopy>(), npc, fileX.as<u32>(), body.as<u64>(), Xs.as<u32>(), bodies.as<u64>());
    KCHECK();
    // blocks by X ascending, bodies of a block descending (_sort_YSA_Block :600-611): sort by body descending, then stable by X.
    // (The reference's block sort is stable in file order, but the bodies of a block are re-sorted as whole words afterwards.)
    size_t tb1 = 0, tb2 = 0;
    HIPCK(rocprim::radix_sort_pairs_desc(nullptr, tb1, bodies.as<u64>(), bodies2.as<u64>(), Xs.as<u32>(), Xs2.as<u32>(), (size_t)n, 0u, 64u, ctx->stream));
    HIPCK(rocprim::radix_sort_pairs(nullptr, tb2, Xs2.as<u32>(), Xs.as<u32>(), bodies2.as<u64>(), bodies.as<u64>(), (size_t)n, 0u, (unsigned)HX_XBITS, ctx->stream));
    ENSURE(cub, std::max(tb1, tb2) + 16);
    HIPCK(rocprim::radix_sort_pairs_desc(cub.p, tb1, bodies.as<u64>(), bodies2.as<u64>(), Xs.as<u32>(), Xs2.as<u32>(), (size_t)n, 0u, 64u, ctx->stream));
    HIPCK(rocprim::radix_sort_pairs(cub.p, tb2, Xs2.as<u32>(), Xs.as<u32>(), bodies2.as<u64>(), bodies.as<u64>(), (size_t)n, 0u, (unsigned)HX_XBITS, ctx->stream));
    ENSURE(flag, (n + 1) * 4 + 16); ENSURE(cntX, ((size_t)1 << HX_XBITS) * 4);
    HIPCK(hipMemsetAsync(cntX.p, 0, ((size_t)1 << HX_XBITS) * 4, ctx->stream));
    HIPCK(hipMemsetAsync(flag.as<i32>() + n, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_hx_flags, dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, Xs.as<u32>(), n, flag.as<i32>(), cntX.as<u32>());
    KCHECK();
    if ((s = dev_scan_i32(ctx, flag.as<i32>(), flag.as<i32>(), n + 1, tmp)) != LNR_OK) return s;
    i32 ndist = 0;
    HIPCK(hipMemcpyAsync(&ndist, flag.as<i32>() + n, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCK(hipStreamSynchronize(ctx->stream));
    // _createYSA :1336-1352: with fewer than three merged blocks the reference drops its last block and leaves words of it behind in
    // file order -- a reference of a few hundred bases; not reproduced
    if (n - (u64)ndist <= 2) { ctx->err = "reference too small for -i 2 (fewer than three repeated minimizers: the reference's countMove <= 2 branch)"; return LNR_ERR_UNSUPPORTED; }
    u64 ysa_len = n + (u64)ndist + 2;
    ctx->info.hs_len = ysa_len;
    ENSURE(ctx->hs, ysa_len * 8 + 64);
    hipLaunchKernelGGL(k_hx_assemble, dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, Xs.as<u32>(), bodies.as<u64>(), n, flag.as<i32>(), cntX.as<u32>(), ctx->hs.as<u64>(), ysa_len);
    KCHECK();
    HIPCK(hipStreamSynchronize(ctx->stream));
    return hx_derive(ctx);
}

void set_index_layout(lnr_ctx *ctx, const u64 *len, u32 nseq) {
    ctx->seq_len.assign(len, len + nseq);
    ctx->seq_off.assign(nseq, 0);
    ctx->f2_off.assign(nseq + 1, 0);
    u64 o = 0, maxlen = 0;
    for (u32 i = 0; i < nseq; i++) {
        ctx->seq_off[i] = o;
        o += align_up(len[i] + SEQ_PAD, 64);
        ctx->f2_off[i + 1] = ctx->f2_off[i] + genome_feature_count(len[i]);
        maxlen = std::max(maxlen, len[i]);
    }
    ctx->info.nseq = nseq;
    ctx->info.genome_bytes = o;
    ctx->info.dir_len = ctx->opts.index_type == 2 ? ((u64)1 << HX_XBITS) + 1 : ((u64)1 << 26) + 1;
    ctx->info.f2_len = ctx->f2_off[nseq];
    ctx->nbins = (u32)((maxlen + (2ULL << 20)) / 30000 + 2);
}

lnr_status upload_index_layout(lnr_ctx *ctx) {
    lnr_status s;
    if ((s = upload(ctx, ctx->d_seq_off, ctx->seq_off)) != LNR_OK) return s;
    if ((s = upload(ctx, ctx->d_f2_off, ctx->f2_off)) != LNR_OK) return s;
    if ((s = upload(ctx, ctx->d_seq_len, ctx->seq_len)) != LNR_OK) return s;
    return LNR_OK;
}

// ------------------------------------------------------------------ jobs ----
struct HostJobs {
    std::vector<u32> read, str, end, mode, grp_beg;
    u64 nsamp = 0;
    void add(u32 r, u32 s, u32 e, u32 m) {
        read.push_back(r); str.push_back(s); end.push_back(e); mode.push_back(m);
        nsamp += seed_num_samples(s, e, (u32)job_parm((int)m).alpha);
    }
    u32 size() const { return (u32)read.size(); }
};

struct BatchHost {
    u32 n = 0;
    std::vector<u64> off;
    std::vector<u32> len, nf, cords_cap, gaps_cap;
    std::vector<u64> f1_off, cords_off, gaps_off, pk_off;
};

// host-side lap timer (LNR_DEBUG_TIMES=1 prints where the host thread spends the step)
struct Laps {
    bool on; std::chrono::steady_clock::time_point t0, t; std::string out;
    Laps() : on(getenv("LNR_DEBUG_TIMES") != nullptr) { t0 = t = std::chrono::steady_clock::now(); }
    void lap(const char *name) {
        if (!on) return;
        auto n = std::chrono::steady_clock::now();
        char b[96]; snprintf(b, sizeof b, " %s %.2f", name, std::chrono::duration<double, std::milli>(n - t).count());
        out += b; t = n;
    }
    void done() { if (on) fprintf(stderr, "[lnr] host laps (ms):%s | total %.2f\n", out.c_str(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
};
typedef lnr_ctx::JobSet JobSet;
typedef lnr_ctx::Launch Launch;
typedef lnr_ctx::TailBuf TailBuf;

JobArrays job_arrays(JobSet &S) {
    JobArrays J;
    J.read = S.j_read.as<u32>(); J.str = S.j_str.as<u32>(); J.end = S.j_end.as<u32>(); J.mode = S.j_mode.as<u32>();
    return J;
}
ReadArrays read_arrays(lnr_ctx *ctx) {
    ReadArrays R;
    R.len = ctx->rlen.as<u32>(); R.ks = ctx->rks.as<i32>();
    R.pk = ctx->pk.as<u64>(); R.nm = ctx->nm.as<u32>(); R.pk_off = ctx->pk_off.as<u64>();
    return R;
}
template <class T>
lnr_status upload_on(lnr_ctx *ctx, DevBuf &b, const std::vector<T> &v, hipStream_t st) {
    ENSURE(b, std::max<size_t>(v.size() * sizeof(T), 16));
    if (!v.empty()) {
        void *h = b.host_stage(v.size() * sizeof(T));
        if (!h) { ctx->err = "pinned host allocation failed"; return LNR_ERR_NOMEM; }
        memcpy(h, v.data(), v.size() * sizeof(T));
        HIPCK(words_in(b.p, h, v.size() * sizeof(T), st));
    }
    return LNR_OK;
}

// Read features of the whole batch on the side stream, ordered behind whatever the main stream holds right now.  They are
// not needed before the job kernels, so filter_dev issues this right behind the round-0 seed kernel: k_f1 then runs while
// the host reads the seed counts back and prepares the launch order (the GPU would idle there), not beside the seed kernel.
lnr_status launch_f1(lnr_ctx *ctx, u32 n) {
    HIPCK(hipEventRecord(ctx->ev_prep, ctx->stream));
    HIPCK(hipStreamWaitEvent(ctx->s_bulk[1], ctx->ev_prep, 0));
    hipLaunchKernelGGL(k_f1, dim3(n), dim3(256), 0, ctx->s_bulk[1], ctx->pk.as<u64>(), ctx->nm.as<u32>(), ctx->pk_off.as<u64>(), ctx->rlen.as<u32>(), ctx->nf.as<u32>(), ctx->f1_off.as<u64>(), n,
                       ctx->f1.as<F96>());
    KCHECK();
    HIPCK(hipEventRecord(ctx->ev_f1, ctx->s_bulk[1]));
    return LNR_OK;
}

// Seed lookup (k_seed_fused) of the job list `hj` into the job set S, on stream st.  Returns with the stream idle and the
// per-job counts (bucket entries, lookups, anchors, anchor offsets) mirrored on the host.
lnr_status seed_jobs(lnr_ctx *ctx, JobSet &S, const HostJobs &hj, hipStream_t st, u32 f1_reads = 0) {
    u32 nj = hj.size();
    S.cap.assign(nj, 0); S.look.assign(nj, 0); S.nanc.assign(nj, 0); S.anc_off.assign(nj, 0);
    if (nj == 0) return f1_reads ? launch_f1(ctx, f1_reads) : LNR_OK;
    lnr_status s;
    if ((s = upload_on(ctx, S.j_read, hj.read, st)) != LNR_OK) return s;
    if ((s = upload_on(ctx, S.j_str, hj.str, st)) != LNR_OK) return s;
    if ((s = upload_on(ctx, S.j_end, hj.end, st)) != LNR_OK) return s;
    if ((s = upload_on(ctx, S.j_mode, hj.mode, st)) != LNR_OK) return s;
    if ((s = upload_on(ctx, S.grp_beg, hj.grp_beg, st)) != LNR_OK) return s;
    ENSURE(S.j_cap, (size_t)nj * 4);
    ENSURE(S.j_look, (size_t)nj * 4);
    ENSURE(S.j_nanc, (size_t)nj * 4);
    ENSURE(S.j_anc_off, (size_t)nj * 8);
    ENSURE(S.seed_ctl, 64);
    if (!S.t_seed.a) S.t_seed.init();
    JobArrays J = job_arrays(S);
    ReadArrays R = read_arrays(ctx);
    // anchor buffer: every job starts with a segment of est x samples slots and moves to one of twice the size when that fills
    // up, so the buffer holds the first segments plus room for the moves; a launch that runs out is repeated with twice the room
    // The capacity is sticky and generous (grown by half when a batch needs more, never shrunk): re-allocating a buffer of a few GB
    // costs ~300 ms, which one step of a benchmark paid when the estimate crept over the old allocation's slack.
    u64 first_segs = ((hj.nsamp * S.est_x16) >> 4) + (u64)nj * 194;
    u64 need_slots = first_segs * 2 + (1u << 20);
    if (need_slots > S.cap_slots) S.cap_slots = need_slots + need_slots / 2;
    u64 anc_slots = S.cap_slots;
    bool use_bm = ctx->seed_bm < 0 ? ctx->info.hs_len < (1ULL << 25) : ctx->seed_bm != 0;
    for (int attempt = 0; ; attempt++) {
        ENSURE(S.anchors, anc_slots * 8);
        HIPCK(hipMemsetAsync(S.seed_ctl.p, 0, 64, st));
        SeedOutArrays O;
        O.cursor = S.seed_ctl.as<unsigned long long>(); O.overflow = (int *)(S.seed_ctl.as<char>() + 16); O.capacity = anc_slots;
        O.anchors = S.anchors.as<u64>(); O.anc_off = S.j_anc_off.as<u64>(); O.job_cap = S.j_cap.as<u32>(); O.job_look = S.j_look.as<u32>();
        O.n_anchors = S.j_nanc.as<u32>();
        S.t_seed.start(st);
        // the bucket bitmap answers lookups of empty buckets without touching the bucket lines; once most buckets hold entries
        // (human scale: 328 M entries in 67 M buckets) it is one more dependent load in front of every lookup and is skipped
        if (ctx->opts.index_type == 2)
            hipLaunchKernelGGL(k_seed_hindex, dim3(nj), dim3(64), 0, st, J, R, ctx->hs.as<u64>(), ctx->info.hs_len, ctx->hx_empty_dir, ctx->dir.as<i32>(), ctx->hx_nkeys.as<u64>(), ctx->hx_nvals.as<u32>(), ctx->hx_nnodes, nj, O,
                               S.est_x16);
        else
        hipLaunchKernelGGL(k_seed_fused, dim3(nj), dim3(64), ctx->seed_lds_pad, st, J, R, ctx->bl.as<ulonglong2>(), use_bm ? ctx->bm.as<u32>() : (const u32 *)nullptr, ctx->ov.as<u64>(), nj, O, S.est_x16,
                           ctx->use_bh ? ctx->bh.as<u64>() : (const u64 *)nullptr);
        KCHECK();
        S.t_seed.stop(st);
        if (f1_reads && attempt == 0) { lnr_status fs = launch_f1(ctx, f1_reads); if (fs != LNR_OK) return fs; }   // (beside the seed kernel instead: measured no faster)
        int ovf = 0;
        Readback rb;
        if (!rb.begin(S.h_rb, (size_t)nj * 20 + 256)) { ctx->err = "pinned host allocation failed"; return LNR_ERR_NOMEM; }
        HIPCK(rb.add(S.cap.data(), S.j_cap.p, (size_t)nj * 4, st));
        HIPCK(rb.add(S.look.data(), S.j_look.p, (size_t)nj * 4, st));
        HIPCK(rb.add(S.nanc.data(), S.j_nanc.p, (size_t)nj * 4, st));
        HIPCK(rb.add(S.anc_off.data(), S.j_anc_off.p, (size_t)nj * 8, st));
        HIPCK(rb.add(&ovf, S.seed_ctl.as<char>() + 16, 4, st));
        HIPCK(hipStreamSynchronize(st));
        rb.finish();
        if (!ovf) {                                   // only the successful launch is the stage's time
            ctx->stats.seed_count_ms += S.t_seed.ms();
            ctx->stats.seed_count_launches++;
            break;
        }
        if (attempt == 5) { ctx->err = "anchor buffer overflow after five resizes"; return LNR_ERR_INTERNAL; }
        anc_slots *= 2;
        S.cap_slots = anc_slots;
    }
    {   // learn the segment estimate for the next batch: 1.5 x the mean anchors per sample of this one
        u64 tot = 0;
        for (u32 j = 0; j < nj; j++) tot += S.nanc[j];
        if (hj.nsamp) {
            S.est_x16 = (u32)std::min<u64>(std::max<u64>((tot * 24) / hj.nsamp + 8, 32), 400 * 16);
        }
    }
    ctx->stats.jobs += nj;
    ctx->stats.samples += hj.nsamp;
    for (u32 j = 0; j < nj; j++) { ctx->stats.lookups += S.look[j]; ctx->stats.bucket_entries += S.cap[j] - 1; ctx->stats.anchors += S.nanc[j] - 1; }
    return LNR_OK;
}

// copy the raw anchors of a seeded job set to the host arrays (CSR by job)
lnr_status export_anchors(lnr_ctx *ctx, JobSet &S, u32 nj) {
    ctx->h_anchor_off.assign((size_t)nj + 1, 0);
    for (u32 j = 0; j < nj; j++) ctx->h_anchor_off[j + 1] = ctx->h_anchor_off[j] + S.nanc[j];
    ctx->h_anchors.resize(ctx->h_anchor_off[nj]);
    u64 used = 0;
    for (u32 j = 0; j < nj; j++) used = std::max<u64>(used, S.anc_off[j] + S.nanc[j]);
    std::vector<u64> all(used);
    if (used) HIPCK(hipMemcpy(all.data(), S.anchors.p, used * 8, hipMemcpyDeviceToHost));
    for (u32 j = 0; j < nj; j++) memcpy(ctx->h_anchors.data() + ctx->h_anchor_off[j], all.data() + S.anc_off[j], (size_t)S.nanc[j] * 8);
    return LNR_OK;
}

// Per-read job kernels for the groups `groups` of the seeded job set S (hj = its host list).  Heaviest group first (anchors
// that passed the Y filter are the work proxy), so the long tail of repeat-rich reads starts at once.  The multi-wave
// kernels go to lane's s_multi and are launched first: a multi-wave workgroup only finds a CU with enough free wave slots
// while the single-wave kernel has not flooded the chip (it refills every slot a finished wave frees -- a late heavy
// launch was measured to start only when the bulk kernel drained, 47 ms late).  The bulk kernel follows on s_bulk.  On
// return everything is enqueued and s_multi also waits for s_bulk; nothing is synchronised unless the scratch budget
// forces several slices.
lnr_status launch_jobs(lnr_ctx *ctx, JobSet &S, Launch &Lx, const HostJobs &hj, const std::vector<u32> &groups, int lane) {
    u32 ngrp = (u32)groups.size();
    if (ngrp == 0) return LNR_OK;
    Laps laps;
    u32 nj = hj.size();
    hipStream_t sm = ctx->s_multi[lane], sb = ctx->s_bulk[lane];
    u64 budget = ctx->opts.scratch_budget ? ctx->opts.scratch_budget : (64ULL << 30);
    const std::vector<u32> &nanc = S.nanc;
    std::vector<u64> w(ngrp, 0);
    for (u32 k = 0; k < ngrp; k++) for (u32 j = hj.grp_beg[groups[k]]; j < hj.grp_beg[groups[k] + 1]; j++) w[k] += nanc[j];
    std::vector<u32> order(ngrp);   // indices into `groups`, heaviest first
    {
        // counting sort by weight class (1/8-octave steps: "descending up to 9 %" is all the scheduler needs) in O(n),
        // then the small multi-wave prefix in exact order (the size-class cut below walks it)
        auto cls = [](u64 v) -> u32 {
            if (v < 8) return (u32)v;
            int lg = 63 - __builtin_clzll(v);
            return (u32)(8 * (lg - 2) + ((v >> (lg - 3)) & 7));
        };
        const u32 NCLS = 8 * 64;
        std::vector<u32> cnt(NCLS + 1, 0), gc(ngrp);
        for (u32 k = 0; k < ngrp; k++) { gc[k] = NCLS - 1 - std::min<u32>(cls(w[k]), NCLS - 1); cnt[gc[k] + 1]++; }
        for (u32 c = 0; c < NCLS; c++) cnt[c + 1] += cnt[c];
        for (u32 k = 0; k < ngrp; k++) order[cnt[gc[k]]++] = k;
        u32 nh = 0;
        while (nh < ngrp && w[order[nh]] >= std::min<u64>(std::min(std::min(ctx->heavy_cap, ctx->mid_cap), std::min(ctx->heavy_cap_r1, ctx->mid_cap_r1)), std::min(ctx->dp_split_cap, ctx->dp_split_cap_r1)) / 2) nh++;
        // (exact order only for a short prefix: at human scale every read carries > 1500 mostly random anchors, the prefix was
        // 60 % of the batch and its sort 2.3 ms of host time per step with the GPU idle; without it the class cuts are exact
        // to the 1/8 octave, which only moves a few reads between kernels)
        if (nh <= 4096) std::stable_sort(order.begin(), order.begin() + nh, [&w](u32 a, u32 b) { return w[a] > w[b]; });
    }
    std::vector<u32> &dev_order = Lx.h_order;
    dev_order.resize(ngrp);
    for (u32 k = 0; k < ngrp; k++) dev_order[k] = groups[order[k]];
    laps.lap("order");
    lnr_status s;
    if ((s = upload_on(ctx, Lx.grp_order, dev_order, sm)) != LNR_OK) return s;
    laps.lap("upload-order");
    ENSURE(Lx.j_scr_off, (size_t)nj * 8);
    ENSURE(Lx.jstate, (size_t)nj * 8 + 16);
    std::vector<u64> &scr_off = Lx.h_scr_off;
    scr_off.assign(nj, 0);
    auto grp_scr = [&](u32 g) { u64 b = 0; for (u32 j = hj.grp_beg[g]; j < hj.grp_beg[g + 1]; j++) b += align_up(job_scratch_bytes((u64)nanc[j] + 2), 256); return b; };
    u32 g0 = 0;
    while (g0 < ngrp) {
        u64 scr = 0;
        u32 g1 = g0;
        while (g1 < ngrp) {
            u64 s2 = scr + grp_scr(dev_order[g1]);
            if (g1 > g0 && s2 > budget) break;
            scr = s2; g1++;
        }
        u64 so = 0;
        for (u32 k = g0; k < g1; k++)
            for (u32 j = hj.grp_beg[dev_order[k]]; j < hj.grp_beg[dev_order[k] + 1]; j++) { scr_off[j] = so; so += align_up(job_scratch_bytes((u64)nanc[j] + 2), 256); }
        laps.lap("scr-layout");
        ENSURE(Lx.job_scr, std::max<u64>(so, 16));
        laps.lap("ensure-scr");
        {
            void *h = Lx.j_scr_off.host_stage((size_t)nj * 8);
            if (!h) { ctx->err = "pinned host allocation failed"; return LNR_ERR_NOMEM; }
            memcpy(h, scr_off.data(), (size_t)nj * 8);
            HIPCK(words_in(Lx.j_scr_off.p, h, (size_t)nj * 8, sm));
        }
        laps.lap("upload-scr");
        JobArgs A;
        A.grp_order = Lx.grp_order.as<u32>(); A.grp_beg = S.grp_beg.as<u32>(); A.J = job_arrays(S);
        A.anc_off = S.j_anc_off.as<u64>(); A.job_cap = S.j_cap.as<u32>(); A.n_anchors = S.j_nanc.as<u32>(); A.scr_off = Lx.j_scr_off.as<u64>();
        A.anchors = S.anchors.as<u64>(); A.scratch = Lx.job_scr.as<char>();
        A.read_len = ctx->rlen.as<u32>(); A.f1_off = ctx->f1_off.as<u64>(); A.nf = ctx->nf.as<u32>(); A.f1 = ctx->f1.as<F96>();
        A.g.base = ctx->f2.as<F96>(); A.g.off = ctx->d_f2_off.as<u64>(); A.g.nseq = ctx->info.nseq;
        A.cords = ctx->cords.as<u64>(); A.cords_off = ctx->cords_off.as<u64>(); A.cords_cap = ctx->cords_cap.as<u32>(); A.ncords = ctx->ncords.as<u32>();
        A.read_err = ctx->read_err.as<i32>();
        A.nbins = ctx->nbins; A.grp_lo = g0; A.grp_hi = g1;
        A.prof = nullptr; A.tl = nullptr; A.jstate = Lx.jstate.as<u32>(); A.stop_after = ctx->stop_after;
        const bool split = ctx->post_split && ctx->stop_after == 0;
        if (split) HIPCK(hipMemsetAsync(Lx.jstate.p, 0, (size_t)nj * 8, sm));   // a job the job kernel never reached reads as "not handed over"
        // dynamic LDS = the job arena; the binning histogram borrows it first and sweeps the bin range in passes of that many
        // bins, so the LDS per workgroup (hence the residency of the bulk kernel) does not depend on the reference's length
        const size_t lds_min = 4096;                 // the split path's pre / post kernels keep every array in global scratch
        size_t arena = (ctx->job_lds_bytes + 15) & ~(size_t)15;
        size_t lds = arena + ctx->job_stage_bytes;
        A.lds_bytes = (u32)lds;
        A.arena_lds = (u32)arena;
        // size classes along the (weight-descending) slice: heavy = 16 waves per read, mid = 4 waves, rest = 1 wave
        // (the re-map round leaves most of the chip idle, so it can afford wider workgroups for more of its reads)
        bool remap_round_ = nj && hj.mode[0] != 0;
        u64 hcap = remap_round_ ? ctx->heavy_cap_r1 : ctx->heavy_cap, mcap = remap_round_ ? ctx->mid_cap_r1 : ctx->mid_cap;
        // a populated table (human scale) adds ~1 500 chance anchors to every read's weight, and the 4-wave kernel holds 14 of a CU's 16
        // wave slots while it runs: fewer reads go there (measured on the GRCh38 stand-in: 6144 -> 40.0 ms, 9000 -> 38.6, 12000 -> 38.5, 20000 -> 45)
        if (!remap_round_ && !ctx->mid_cap_env && ctx->info.hs_len >= (1ULL << 25)) mcap = 9000;
        u64 scap = remap_round_ ? ctx->dp_split_cap_r1 : ctx->dp_split_cap;
        u32 gh = g0;
        while (gh < g1 && w[order[gh]] >= hcap) gh++;
        u32 gs = gh;                                    // [gh, gs): split path (pre -> 16-wave DP -> post)
        while (gs < g1 && w[order[gs]] >= scap) gs++;
        u32 gm = gs;                                    // [gs, gm): 4 waves per read
        while (gm < g1 && w[order[gm]] >= mcap) gm++;
#ifdef LNR_PROF
        if (!ctx->prof.p) { if (!ctx->prof.ensure(192 * 8)) return LNR_ERR_NOMEM; (void)hipMemsetAsync(ctx->prof.p, 0, 192 * 8, sm); }
        A.prof = ctx->prof.as<unsigned long long>();
        // timeline: up to 4 launches of up to 2^20 positions
        if (!ctx->tl.p) { if (!ctx->tl.ensure(4ULL * (1u << 20) * 32)) return LNR_ERR_NOMEM; (void)hipMemsetAsync(ctx->tl.p, 0, 4ULL * (1u << 20) * 32, sm); }
        if (ctx->tl_round < 4 && g1 <= (1u << 20)) { A.tl = ctx->tl.as<unsigned long long>() + (size_t)ctx->tl_round * (1u << 20) * 4; ctx->tl_n[ctx->tl_round] = g1; ctx->tl_nh[ctx->tl_round] = gm; }
        ctx->tl_round++;
#endif
        // streams: the 16-wave kernel gets the spare stream when the batch runs as one lane (kernels on one stream would
        // run back to back), the 4-wave kernel the lane's main stream, the single-wave kernel the lane's bulk stream
        // streams: kernels on one stream run back to back.  The 16-wave kernel stays on the lane's main stream (no event wait,
        // it reaches the GPU first); the split-path chain and the 4-wave kernel take the spare stream when the 16-wave class
        // is present and the batch runs as one lane, else the main stream; the single-wave kernel goes to the lane's bulk stream.
        bool wide2 = gm > gh;                                   // split and / or 4-wave class present
        bool one_lane = ctx->split_cap == 0xffffffffu;
        // stream of the split / 4-wave class when the 16-wave class is present too: the spare stream when the batch runs as
        // one lane; with two lanes the other lane owns that stream, so the class queues behind this lane's (short) bulk kernel
        hipStream_t s4 = (gh > g0 && wide2) ? ((lane == 1 && one_lane) ? ctx->s_multi[0] : sb) : sm;
        bool after_bulk = s4 == sb && sb != sm;
        bool fork_m = wide2 && s4 != sm, fork_b = g1 > gm && sb != sm && gm > g0;
        if (fork_m || fork_b) HIPCK(hipEventRecord(ctx->ev_fork[lane], sm));    // before any launch: nobody waits for another kernel
        if (gh > g0) {
            JobArgs H = A;
            size_t hl = (size_t)ctx->heavy_lds_kb * 1024;
            H.grp_lo = g0; H.grp_hi = gh; H.lds_bytes = (u32)hl; H.arena_lds = (u32)hl;
            if (split) {
                hipLaunchKernelGGL(k_job_heavy_a, dim3(gh - g0), dim3(1024), hl, sm, H);
                KCHECK();
                hipLaunchKernelGGL(k_post, dim3((gh - g0 + 63) / 64), dim3(64), 0, sm, H);
            } else hipLaunchKernelGGL(k_job_heavy, dim3(gh - g0), dim3(1024), hl, sm, H);
            KCHECK();
        }
        auto launch_bulk = [&]() -> lnr_status {
            if (g1 <= gm) return LNR_OK;
            hipStream_t bulk = fork_b ? sb : sm;
            if (fork_b) HIPCK(hipStreamWaitEvent(sb, ctx->ev_fork[lane], 0));
            if (gm > g0 && bulk != sm && !after_bulk) { hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, bulk, ctx->bulk_delay_ticks); KCHECK(); }
            JobArgs K = A;
            K.grp_lo = gm; K.grp_hi = g1;
            if (split) {
                hipLaunchKernelGGL(k_job_a, dim3(g1 - gm), dim3(64), lds, bulk, K);
                KCHECK();
                hipLaunchKernelGGL(k_post, dim3((g1 - gm + 63) / 64), dim3(64), 0, bulk, K);
            } else hipLaunchKernelGGL(k_job, dim3(g1 - gm), dim3(64), lds, bulk, K);
            KCHECK();
            return LNR_OK;
        };
        if (after_bulk && (s = launch_bulk()) != LNR_OK) return s;
        if (fork_m && !after_bulk) HIPCK(hipStreamWaitEvent(s4, ctx->ev_fork[lane], 0));
        if (fork_m && after_bulk && !(g1 > gm)) HIPCK(hipStreamWaitEvent(s4, ctx->ev_fork[lane], 0));
        if (gs > gh) {
            JobArgs P = A;
            P.grp_lo = gh; P.grp_hi = gs; P.lds_bytes = (u32)lds_min; P.arena_lds = 0;   // global scratch only: pointers must replay
            hipLaunchKernelGGL(k_job_pre, dim3(gs - gh), dim3(64), lds_min, s4, P);
            KCHECK();
            hipLaunchKernelGGL(k_job_dp, dim3(gs - gh), dim3(64 * DP_SPLIT_WAVES), 0, s4, P);
            KCHECK();
            hipLaunchKernelGGL(k_job_post, dim3(gs - gh), dim3(64), lds_min, s4, P);
            KCHECK();
        }
        if (gm > gs) {
            JobArgs M = A;
            size_t ml = (size_t)ctx->mid_lds_kb * 1024;
            M.grp_lo = gs; M.grp_hi = gm; M.lds_bytes = (u32)ml; M.arena_lds = (u32)ml;
            if (split) {
                hipLaunchKernelGGL(k_job_mid_a, dim3(gm - gs), dim3(256), ml, s4, M);
                KCHECK();
                hipLaunchKernelGGL(k_post, dim3((gm - gs + 63) / 64), dim3(64), 0, s4, M);
            } else if (ctx->mid_waves == 2 || (ctx->mid_waves == 0 && !remap_round_ && ctx->info.hs_len >= (1ULL << 25)))
                // round 0 at human scale is bound by wave slots (16 per CU at 128 VGPRs): two waves per read of this class hold half the slots of
                // four for a little longer (GRCh38 stand-in: 45.2 vs 47.0 ms per step)
                hipLaunchKernelGGL(k_job_mid2, dim3(gm - gs), dim3(128), ml, s4, M);
            else hipLaunchKernelGGL(k_job_mid, dim3(gm - gs), dim3(256), ml, s4, M);
            KCHECK();
        }
        if (fork_m && !after_bulk) HIPCK(hipEventRecord(ctx->ev_join[0], s4));
        if (!after_bulk && (s = launch_bulk()) != LNR_OK) return s;
        if (fork_b || (after_bulk && fork_m)) { HIPCK(hipEventRecord(ctx->ev_join[lane], sb)); HIPCK(hipStreamWaitEvent(sm, ctx->ev_join[lane], 0)); }
        if (fork_m && !after_bulk) HIPCK(hipStreamWaitEvent(sm, ctx->ev_join[0], 0));
        laps.lap("launches");
        ctx->stats.job_launches++;
        g0 = g1;
        if (g0 < ngrp) HIPCK(hipStreamSynchronize(sm));   // next slice reuses the scratch
    }
    if (laps.on && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - laps.t0).count() > 1.5) laps.done();
    return LNR_OK;
}

// per-batch host tables + prep / feature kernels.  d_reads/d_off are device pointers.
lnr_status prepare_batch(lnr_ctx *ctx, const u8 *d_reads, const u64 *d_off, u32 n, BatchHost &B, const u64 *h_off = nullptr) {
    B.n = n;
    B.off.resize((size_t)n + 1);
    if (h_off) memcpy(B.off.data(), h_off, ((size_t)n + 1) * 8);     // the host-buffer entry points know the offsets already
    else {
        Readback rb;
        if (!rb.begin(ctx->h_rb[0], ((size_t)n + 1) * 8)) { ctx->err = "pinned host allocation failed"; return LNR_ERR_NOMEM; }
        HIPCK(rb.add(B.off.data(), d_off, ((size_t)n + 1) * 8, ctx->stream));
        HIPCK(hipStreamSynchronize(ctx->stream));
        rb.finish();
    }
    B.len.resize(n); B.nf.resize(n); B.cords_cap.resize(n); B.gaps_cap.resize(n);
    B.f1_off.resize(n); B.cords_off.resize(n); B.gaps_off.resize(n); B.pk_off.resize(n);
    u64 fo = 0, co = 0, go = 0, po = 0;
    for (u32 i = 0; i < n; i++) {
        if (B.off[i + 1] < B.off[i]) { ctx->err = "read offsets not monotone"; return LNR_ERR_ARG; }
        u64 L = B.off[i + 1] - B.off[i];
        if (L >= (1ULL << 20)) { ctx->err = "read longer than 2^20-1 bases (cord y field, cords.cpp:15)"; return LNR_ERR_LIMIT; }
        B.len[i] = (u32)L;
        B.pk_off[i] = po; po += 2 * packed_words(L);   // forward + reverse-complement strand
        B.nf[i] = L > 200 ? read_feature_count(L) : 0;
        B.f1_off[i] = fo; fo += 2ULL * B.nf[i];
        B.cords_cap[i] = L > 200 ? (u32)std::min<u64>(std::max<u64>((16 * (L / 64) + 256) / ctx->cap_shrink, 8) * ctx->cap_scale, 1u << 24) : 0;
        B.cords_off[i] = co; co += B.cords_cap[i];
        B.gaps_cap[i] = L > 200 ? (u32)((L / 1000 + 4) * ctx->cap_scale) : 0;
        B.gaps_off[i] = go; go += B.gaps_cap[i];
    }
    lnr_status s;
    if ((s = upload(ctx, ctx->rlen, B.len)) != LNR_OK) return s;
    if ((s = upload(ctx, ctx->pk_off, B.pk_off)) != LNR_OK) return s;
    ENSURE(ctx->pk, std::max<u64>(po * 8, 16));
    ENSURE(ctx->nm, std::max<u64>(po * 4, 16));
    if ((s = upload(ctx, ctx->nf, B.nf)) != LNR_OK) return s;
    if ((s = upload(ctx, ctx->f1_off, B.f1_off)) != LNR_OK) return s;
    if ((s = upload(ctx, ctx->cords_cap, B.cords_cap)) != LNR_OK) return s;
    if ((s = upload(ctx, ctx->cords_off, B.cords_off)) != LNR_OK) return s;
    if ((s = upload(ctx, ctx->gaps_cap, B.gaps_cap)) != LNR_OK) return s;
    if ((s = upload(ctx, ctx->gaps_off, B.gaps_off)) != LNR_OK) return s;
    ENSURE(ctx->rks, (size_t)n * 4);
    ENSURE(ctx->f1, std::max<u64>(fo * sizeof(F96), 16));
    ENSURE(ctx->cords, std::max<u64>(co * 8, 16));
    ENSURE(ctx->out_str, std::max<u64>(co * 8, 16));
    ENSURE(ctx->out_end, std::max<u64>(co * 8, 16));
    ENSURE(ctx->gaps, std::max<u64>(go * sizeof(UP), 16));
    ENSURE(ctx->gdense, std::max<u64>(go * sizeof(UP), 16));
    ENSURE(ctx->gcursor, 16);
    ENSURE(ctx->gpos, (size_t)n * 4);
    ENSURE(ctx->ncords, (size_t)n * 4);
    ENSURE(ctx->nout, (size_t)n * 4);
    ENSURE(ctx->read_err, (size_t)n * 4);
    ENSURE(ctx->ngaps, (size_t)n * 4);
    ENSURE(ctx->remap, (size_t)n * 4);
    HIPCK(hipMemsetAsync(ctx->ncords.p, 0, (size_t)n * 4, ctx->stream));
    HIPCK(hipMemsetAsync(ctx->read_err.p, 0, (size_t)n * 4, ctx->stream));
    ctx->t_prep.start(ctx->stream);
    hipLaunchKernelGGL(k_prep, dim3(n < ctx->prep_grid ? n : ctx->prep_grid), dim3(ctx->prep_threads), 0, ctx->stream, d_reads, d_off, ctx->pk_off.as<u64>(), n, ctx->pk.as<u64>(), ctx->nm.as<u32>(), ctx->rks.as<i32>());
    KCHECK();
    ctx->t_prep.stop(ctx->stream);
    ctx->stats.reads = n;
    ctx->stats.bases = B.off[n] - B.off[0];
    return LNR_OK;
}

// Scratch layout + launch arguments of a tail kernel over the reads in `list` (null = all reads), on stream st.
// Returns with the stream idle (it reads the current cord counts back to size the scratch).
lnr_status tail_prepare(lnr_ctx *ctx, const BatchHost &B, TailBuf &tb, const std::vector<u32> *list, hipStream_t st, TailArgs &T) {
    u32 n = B.n;
    std::vector<u32> ncords(n);
    {
        Readback rb;
        PinBuf &pb = ctx->h_rb[st == ctx->s_tail ? 2 : (st == ctx->stream ? 1 : 3)];
        if (!rb.begin(pb, (size_t)n * 4)) { ctx->err = "pinned host allocation failed"; return LNR_ERR_NOMEM; }
        HIPCK(rb.add(ncords.data(), ctx->ncords.p, (size_t)n * 4, st));
        HIPCK(hipStreamSynchronize(st));   // (counts of reads another lane is still working on are not used)
        rb.finish();
    }
    tb.h_off.assign(n, 0); tb.h_cap.assign(n, 0);
    u64 o = 0;
    u32 cnt = list ? (u32)list->size() : n;
    for (u32 k = 0; k < cnt; k++) {
        u32 i = list ? (*list)[k] : k;
        tb.h_cap[i] = ncords[i] + 4; tb.h_off[i] = o; o += align_up(tail_scratch_bytes(tb.h_cap[i]), 256);
    }
    lnr_status s;
    if ((s = upload_on(ctx, tb.off, tb.h_off, st)) != LNR_OK) return s;
    if ((s = upload_on(ctx, tb.cap, tb.h_cap, st)) != LNR_OK) return s;
    ENSURE(tb.scr, std::max<u64>(o, 16));
    T.read_len = ctx->rlen.as<u32>(); T.n = cnt; T.list = nullptr;
    if (list) {
        tb.h_list = *list;
        if ((s = upload_on(ctx, tb.list, tb.h_list, st)) != LNR_OK) return s;
        T.list = tb.list.as<u32>();
    }
    T.cords = ctx->cords.as<u64>(); T.cords_off = ctx->cords_off.as<u64>(); T.cords_cap = ctx->cords_cap.as<u32>(); T.ncords = ctx->ncords.as<u32>();
    T.read_err = ctx->read_err.as<i32>();
    T.scratch = tb.scr.as<char>(); T.scr_off = tb.off.as<u64>(); T.scr_cap = tb.cap.as<u32>();
    T.gaps = ctx->gaps.as<UP>(); T.gaps_off = ctx->gaps_off.as<u64>(); T.gaps_cap = ctx->gaps_cap.as<u32>(); T.ngaps = ctx->ngaps.as<u32>(); T.remap = ctx->remap.as<u32>();
    T.gdense = ctx->gdense.as<UP>(); T.gcursor = ctx->gcursor.as<u32>(); T.gpos = ctx->gpos.as<u32>();
    T.out_str = ctx->out_str.as<u64>(); T.out_end = ctx->out_end.as<u64>(); T.nout = ctx->nout.as<u32>();
    return LNR_OK;
}

void reset_stats(lnr_ctx *ctx) { memset(&ctx->stats, 0, sizeof ctx->stats); }
void finish_stats(lnr_ctx *ctx, const BatchHost &B) {
    u64 rb = 0;
    for (u32 i = 0; i < B.n; i++) if (B.len[i] > 200) rb += (B.len[i] + 3) / 4;
    ctx->stats.seed_bytes = rb + ctx->stats.lookups * 8 + ctx->stats.bucket_entries * 8 + ctx->stats.anchors * 8;
}

// Tail A + re-map round of the reads in `list` (whose round 0 has completed on the lane's s_multi): clean / gather /
// gaps decide the remap loop (pmpfinder.cpp:2744-2749); every gap of a poorly covered read is then re-seeded with
// step 7 / score0 (pmpfinder.cpp:2749-2767).  Uses job set S and launch state Lx; returns with the launches enqueued.
lnr_status remap_round(lnr_ctx *ctx, const BatchHost &B, const std::vector<u32> &list, int lane, JobSet &S, Launch &Lx, TailBuf &tb, HostJobs &j1) {
    if (list.empty()) return LNR_OK;
    hipStream_t st = ctx->s_multi[lane];
    u32 n = B.n;
    TailArgs T;
    lnr_status s;
    if ((s = tail_prepare(ctx, B, tb, &list, st, T)) != LNR_OK) return s;
    Laps laps;
    laps.lap("tail_prepare");
    HIPCK(hipMemsetAsync(ctx->gcursor.p, 0, 4, st));
    hipLaunchKernelGGL(k_tail_a, dim3((T.n + 63) / 64), dim3(64), 0, st, T);
    KCHECK();
    if (!ctx->h_flags.ensure((size_t)n * 12 + 16)) { ctx->err = "pinned host allocation failed"; return LNR_ERR_NOMEM; }
    u32 *remap = ctx->h_flags.as<u32>(), *ngaps = remap + n, *gpos = ngaps + n, *gtot_p = gpos + n;
    auto words_out = [&](void *h_dst, const void *d_src, u64 nw) -> hipError_t {   // (kernel stores into pinned memory: see Readback)
        if (!nw) return hipSuccess;
        hipLaunchKernelGGL(k_words_out, dim3((u32)std::min<u64>((nw + 255) / 256, 1024)), dim3(256), 0, st, (const u32 *)d_src, (u32 *)h_dst, nw);
        return hipGetLastError();
    };
    HIPCK(words_out(remap, ctx->remap.p, n));
    HIPCK(words_out(ngaps, ctx->ngaps.p, n));
    HIPCK(words_out(gpos, ctx->gpos.p, n));
    HIPCK(words_out(gtot_p, ctx->gcursor.p, 1));
    HIPCK(hipStreamSynchronize(st));
    laps.lap("tail_a+flags");
    u64 gtot = *gtot_p;
    if (gtot == 0) return LNR_OK;
    if (!ctx->h_gaps.ensure(gtot * sizeof(UP))) { ctx->err = "pinned host allocation failed"; return LNR_ERR_NOMEM; }
    UP *gaps = ctx->h_gaps.as<UP>();
    HIPCK(words_out(gaps, ctx->gdense.p, gtot * sizeof(UP) / 4));
    HIPCK(hipStreamSynchronize(st));
    for (u32 i : list) {
        if (!(remap[i] && ngaps[i])) continue;
        ctx->stats.remap_reads++;
        j1.grp_beg.push_back(j1.size());
        for (u32 k = 0; k < ngaps[i]; k++) {
            UP y = forward_y(gaps[gpos[i] + k], B.len[i]);
            j1.add(i, (u32)y.first, (u32)y.second, 1);
        }
    }
    j1.grp_beg.push_back(j1.size());
    laps.lap("gaps-copy+build");
    if ((s = seed_jobs(ctx, S, j1, st)) != LNR_OK) return s;
    laps.lap("seed1(sync)");
    if (getenv("LNR_DEBUG_R1") && !ctx->dbg_r0w.empty()) {   // diagnostic: round-0 anchors of the reads that own the heavy re-map groups
        std::vector<std::pair<u64, u64> > v;
        for (u32 g = 0; g + 1 < j1.grp_beg.size(); g++) {
            u64 w = 0;
            for (u32 j = j1.grp_beg[g]; j < j1.grp_beg[g + 1]; j++) w += S.nanc[j];
            if (w >= 2048) v.push_back(std::make_pair(w, (u64)ctx->dbg_r0w[j1.read[j1.grp_beg[g]]]));
        }
        std::sort(v.begin(), v.end());
        fprintf(stderr, "[lnr] lane %d: %zu re-map groups with >= 2048 anchors; (r1 anchors, r0 anchors of the read):", lane, v.size());
        for (size_t k = 0; k < v.size(); k += std::max<size_t>(1, v.size() / 40)) fprintf(stderr, " (%llu,%llu)", (unsigned long long)v[k].first, (unsigned long long)v[k].second);
        fprintf(stderr, "\n");
    }
    std::vector<u32> all((size_t)j1.grp_beg.size() - 1);
    for (u32 g = 0; g < all.size(); g++) all[g] = g;
    s = launch_jobs(ctx, S, Lx, j1, all, lane);
    laps.lap("launch1");
    laps.done();
    return s;
}

lnr_status filter_dev(lnr_ctx *ctx, const u8 *d_reads, const u64 *d_off, u32 n, lnr_cords_dev *out, const u64 *h_off = nullptr, int attempt = 0) {
    if (!ctx->has_index) { ctx->err = "no index: call lnr_index_build or lnr_index_adopt first"; return LNR_ERR_NO_INDEX; }
    reset_stats(ctx);
    ctx->last_n = n; ctx->last_ncords = 0;
    if (out) { out->n_reads = n; out->n_cords = 0; out->d_cord_off = nullptr; out->d_cords_str = nullptr; out->d_cords_end = nullptr; }
    ENSURE(ctx->r_off, ((size_t)n + 1) * 8);
    if (n == 0) {
        HIPCK(hipMemsetAsync(ctx->r_off.p, 0, 8, ctx->stream));
        HIPCK(hipStreamSynchronize(ctx->stream));
        if (out) out->d_cord_off = ctx->r_off.as<u64>();
        return LNR_OK;
    }
    Laps laps;
    ctx->t_total.start(ctx->stream);
    BatchHost B;
    lnr_status s = prepare_batch(ctx, d_reads, d_off, n, B, h_off);
    if (s != LNR_OK) return s;
    ctx->last_gaps_off = B.gaps_off;
    laps.lap("prepare");
    // round 0: one job per read longer than 200 bases (mapper.cpp:430,440), whole read, default parameters
    HostJobs j0;
    for (u32 i = 0; i < n; i++) {
        if (B.len[i] > 200) { j0.grp_beg.push_back(j0.size()); j0.add(i, 0, B.len[i], 0); }
    }
    j0.grp_beg.push_back(j0.size());
    JobSet &S0 = ctx->js[0], &S1 = ctx->js[1];
    if ((s = seed_jobs(ctx, S0, j0, ctx->stream, n)) != LNR_OK) return s;
    laps.lap("seed0(sync)");
    // Two lanes.  Lane 0 = the reads with many anchors (they hold the long chaining jobs of both rounds), lane 1 = the bulk.
    // The reference maps read by read, so any interleaving of reads is the same computation; here lane 0 goes through
    // round 0 -> tail A -> re-map round while lane 1 is still in round 0, instead of a batch-wide barrier per round.
    u32 ngrp0 = (u32)j0.grp_beg.size() - 1;
    std::vector<u32> grp[2], reads[2];
    std::vector<char> in_heavy(n, 0);
    const bool dbg_r1 = getenv("LNR_DEBUG_R1") != nullptr;
    if (dbg_r1) ctx->dbg_r0w.assign(n, 0);
    for (u32 g = 0; g < ngrp0; g++) {
        u64 w = 0;
        for (u32 j = j0.grp_beg[g]; j < j0.grp_beg[g + 1]; j++) w += S0.nanc[j];
        int lane = w >= ctx->split_cap ? 0 : 1;
        if (dbg_r1) ctx->dbg_r0w[j0.read[j0.grp_beg[g]]] = (u32)w;
        grp[lane].push_back(g);
        if (lane == 0) in_heavy[j0.read[j0.grp_beg[g]]] = 1;
    }
    for (u32 i = 0; i < n; i++) reads[in_heavy[i] ? 0 : 1].push_back(i);   // reads without a job go with the bulk
    laps.lap("partition");
    HIPCK(hipStreamWaitEvent(ctx->stream, ctx->ev_f1, 0));   // read features ready (k_f1 ran beside the seed kernel)
    ctx->t_job.start(ctx->stream);
    HIPCK(hipEventRecord(ctx->ev_start, ctx->stream));
    HIPCK(hipStreamWaitEvent(ctx->s_multi[0], ctx->ev_start, 0));
    HIPCK(hipStreamWaitEvent(ctx->s_bulk[1], ctx->ev_start, 0));
    laps.lap("events");
    if ((s = launch_jobs(ctx, S0, ctx->ln[0], j0, grp[0], 0)) != LNR_OK) return s;
    if ((s = launch_jobs(ctx, S0, ctx->ln[1], j0, grp[1], 1)) != LNR_OK) return s;
    laps.lap("launch0");
    // Whichever lane finishes round 0 first goes through tail A and the re-map round while the other is still in round 0
    // (job set 1: the other lane still reads job set 0); the second lane follows, by then nobody reads job set 0 any more.
    // With LNR_SPLIT_CAP at a few thousand anchors lane 0 holds the long single-wave and the 4-wave jobs -- the tail of
    // round 0 -- and the bulk lane is through first (LNR_LANE_ORDER=bulk, the default); the heavy-first order is kept for
    // large split values where lane 0 is a handful of reads.
    HostJobs j1h, j1b;
    bool bulk_first = ctx->lane_bulk_first;
    int first = bulk_first ? 1 : 0, second = 1 - first;
    if (bulk_first) HIPCK(hipStreamSynchronize(ctx->s_multi[1]));
    if ((s = remap_round(ctx, B, reads[first], first, S1, ctx->ln[first], ctx->tb[first], first ? j1b : j1h)) != LNR_OK) return s;
    laps.lap("lane-a");
    // Tail B (block chaining on both strands, flags, cords_end; pmpfinder.cpp:2764-2801) of the reads that do not go through
    // the re-map round is final after tail A: it runs on its own stream while the re-map jobs (a few long reads) are busy.
    std::vector<u32> late_list, early_list;
    bool early = ctx->split_cap == 0xffffffffu && j1b.size() > 0;
    if (early) {
        std::vector<char> in_r1(n, 0);
        for (u32 q = 0; q < j1b.size(); q++) in_r1[j1b.read[q]] = 1;
        for (u32 i = 0; i < n; i++) (in_r1[i] ? late_list : early_list).push_back(i);
        TailArgs TE;
        if ((s = tail_prepare(ctx, B, ctx->tb[2], &early_list, ctx->s_tail, TE)) != LNR_OK) return s;
        if (TE.n) { hipLaunchKernelGGL(k_tail_b, dim3((TE.n + 63) / 64), dim3(64), 0, ctx->s_tail, TE); KCHECK(); }   // (every read may be in the re-map round)
        HIPCK(hipEventRecord(ctx->ev_prep, ctx->s_tail));
    }
    HIPCK(hipStreamSynchronize(ctx->s_multi[0]));
    HIPCK(hipStreamSynchronize(ctx->s_multi[1]));
    laps.lap("wait-r0");
    if ((s = remap_round(ctx, B, reads[second], second, S0, ctx->ln[second], ctx->tb[second], second ? j1b : j1h)) != LNR_OK) return s;
    laps.lap("tailA+seed1+launch1");
    HIPCK(hipEventRecord(ctx->ev_lane[0], ctx->s_multi[0]));
    HIPCK(hipStreamWaitEvent(ctx->stream, ctx->ev_lane[0], 0));   // (lane 1's multi stream is the main stream)
    ctx->t_job.stop(ctx->stream);
    HIPCK(hipStreamSynchronize(ctx->stream));
    ctx->stats.job_ms += ctx->t_job.ms();
    laps.lap("wait-r1");
    TailArgs T;
    if ((s = tail_prepare(ctx, B, ctx->tb[0], early ? &late_list : nullptr, ctx->stream, T)) != LNR_OK) return s;
    ctx->t_tail.start(ctx->stream);
    if (T.n) { hipLaunchKernelGGL(k_tail_b, dim3((T.n + 63) / 64), dim3(64), 0, ctx->stream, T); KCHECK(); }
    ctx->t_tail.stop(ctx->stream);
    if (early) HIPCK(hipStreamWaitEvent(ctx->stream, ctx->ev_prep, 0));
    int ext_state_out = ctx->gap_ext;
    if (ctx->opts.gap_len) {
        // the gap re-mapper on the final cords (k_gap): every read with small arenas, then the flagged reads with large ones
        u32 maxlen = 0;
        for (u32 i = 0; i < n; i++) maxlen = std::max(maxlen, B.len[i]);
        // arena budget of the gap re-mapper's workers: 48 GiB of the 288, never more than lnr_opts.scratch_budget (when given) nor than 80 % of what
        // is free on the device beside the arena already held -- a second context on the GPU gets fewer workers instead of LNR_ERR_NOMEM
        u64 budget = (u64)48 << 30;
        if (ctx->opts.scratch_budget && ctx->opts.scratch_budget < budget) budget = ctx->opts.scratch_budget;
        { size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) == hipSuccess) { u64 avail = (u64)(((double)fr + (double)ctx->gap_arena.cap) * 0.8); if (avail < budget) budget = avail; } else (void)hipGetLastError(); }
        if (budget < ((u64)1 << 30)) budget = (u64)1 << 30;
        u64 arena1 = align_up(((u64)512 << 10) * ctx->cap_scale + 16ULL * maxlen + sizeof(LeaderScratch) + 65536, 256);
        u64 arena2 = std::max<u64>(((u64)ctx->gap_arena2_mb << 20) * ctx->cap_scale, arena1 * 2);
        u64 arena3 = std::max<u64>(((u64)64 << 20) * ctx->cap_scale, arena2 * 2);
        u32 w1 = (u32)std::min<u64>(align_up(n, 64), std::max<u64>(64, (budget / arena1) / 64 * 64));   // workers: lanes, or waves with LNR_GAP_MODE=1
        if (ctx->gap_mode) w1 = std::min<u32>(w1, ctx->gap_waves);
        u32 w2 = (u32)std::min<u64>(std::min<u64>(n, ctx->gap_waves), std::max<u64>(1, budget / arena2));   // waves of the second launch
        u32 w3 = (u32)std::min<u64>(n, std::max<u64>(1, (budget / 2) / arena3));
        const bool fused = ctx->gap_fused && ctx->gap_mode && ctx->gap_team;
        const u32 ncu = ctx->ncu ? ctx->ncu : 256, nteams = std::min<u32>(ctx->gap_teams, ncu / 2);
        u64 fused_bytes = fused ? (u64)nteams * arena2 + (u64)ncu * K_GAP_TEAM * arena1 : 0;     // (a small chunk runs fewer teams and more single waves: bounded by every CU full of single waves)
        ENSURE(ctx->gap_arena, std::max(std::max(std::max((u64)w1 * arena1, (u64)w2 * arena2), (u64)w3 * arena3), fused_bytes));
        ENSURE(ctx->gap_flag, (size_t)n * 4);
        ENSURE(ctx->gap_next, 256);
        HIPCK(hipMemsetAsync(ctx->gap_next.p, 0, 256, ctx->stream));
        GapArgs G;
        G.g = ctx->g.as<u8>(); G.seq_off = ctx->d_seq_off.as<u64>(); G.seq_len = ctx->d_seq_len.as<u64>();
        G.gf.base = ctx->f2.as<F96>(); G.gf.off = ctx->d_f2_off.as<u64>(); G.gf.nseq = ctx->info.nseq;
        G.reads = d_reads; G.off = d_off; G.n = n;
        G.nf = ctx->nf.as<u32>(); G.f1_off = ctx->f1_off.as<u64>(); G.f1 = ctx->f1.as<F96>();
        G.out_str = ctx->out_str.as<u64>(); G.out_end = ctx->out_end.as<u64>(); G.cords_off = ctx->cords_off.as<u64>(); G.cords_cap = ctx->cords_cap.as<u32>();
        G.nout = ctx->nout.as<u32>(); G.read_err = ctx->read_err.as<i32>(); G.gap_flag = ctx->gap_flag.as<u32>();
        G.arena = (char *)ctx->gap_arena.p;
        G.prof = nullptr;
#ifdef LNR_GAP_DEVPROF
        ENSURE(ctx->gap_prof, (96 + 3 * (size_t)n) * 8);
        HIPCK(hipMemsetAsync(ctx->gap_prof.p, 0, (96 + 3 * (size_t)n) * 8, ctx->stream));
        G.prof = ctx->gap_prof.as<unsigned long long>();
#endif
        G.gap_len_min = ctx->opts.gap_len == 1 ? 50 : (ctx->opts.gap_len < 10 ? 10 : ctx->opts.gap_len);   // mapper.cpp:438-453
        G.f_dup = (int)ctx->opts.dup;
        ctx->t_gap.start(ctx->stream);
        ENSURE(ctx->gap_first, 64);
        G.first_ext = ctx->gap_first.as<u32>();
        ENSURE(ctx->gap_rank, ((size_t)n + 16) * 4);
        ENSURE(ctx->gap_weight, ((size_t)n + 16) * 4);
        ENSURE(ctx->gap_list, ((size_t)n + 16) * 4);
        G.list = ctx->gap_list.as<u32>() + 16; G.list_n = ctx->gap_list.as<u32>();
        // one "ladder" = the three launches (small arenas for every read of [lo, hi), then the flagged reads with larger ones)
        auto ladder = [&](u32 lo, u32 hi, u32 ext_from, int probe) -> hipError_t {
            hipError_t e = hipMemsetAsync(ctx->gap_next.p, 0, 256, ctx->stream);
            if (e != hipSuccess) return e;
            u32 m = hi - lo;
            G.lo = lo; G.n = hi; G.ext_from = ext_from; G.probe = probe;
            G.work_cap = ctx->gap_work_cap;
            if ((e = launch_gap_weight(G.reads, G.off, G.out_str, G.cords_off, G.nout, lo, hi, ctx->gap_weight.as<u32>(), ctx->stream)) != hipSuccess) return e;
            if ((e = launch_gap_rank(ctx->gap_weight.as<u32>(), lo, hi, ctx->gap_rank.as<u32>() + 16, ctx->gap_rank.as<u32>(), ctx->gap_heavy_w, ctx->stream)) != hipSuccess) return e;
            G.order = ctx->gap_rank.as<u32>() + 16;
            G.cap_ticks = (u64)ctx->gap_cap_ms * 100000ULL;
            G.next = ctx->gap_next.as<u32>(); G.big = 0; G.last = 0; G.coop = ctx->gap_mode;
            if (fused) {
                // one launch: teams on the reads expected to be heavy + on what the single waves hand over, single waves on the rest
                if ((e = hipMemsetAsync(ctx->gap_list.p, 0, ((size_t)n + 16) * 4, ctx->stream)) != hipSuccess) return e;
                G.nteams = std::min<u32>(nteams, std::max<u32>(1, m / 8));
                u32 bulk_wg = std::min<u32>(ncu > G.nteams ? ncu - G.nteams : 1, (m + K_GAP_TEAM - 1) / K_GAP_TEAM);
                G.nbulk_waves = bulk_wg * K_GAP_TEAM; G.arena_bytes = arena1; G.arena2_bytes = arena2;
                G.n_heavy = ctx->gap_rank.as<u32>(); G.q = ctx->gap_list.as<u32>() + 16;
                G.coop = 1;
                if ((e = launch_gap_all(G, G.nteams + bulk_wg, ctx->stream)) != hipSuccess) return e;
            } else {
                G.arena_bytes = arena1;
                u32 v1 = std::min<u32>(w1, (u32)align_up(m, 64));
                if ((e = launch_gap(G, 0, ctx->gap_mode ? v1 : v1 / 64, ctx->stream)) != hipSuccess) return e;
                G.work_cap = ~0ULL; G.cap_ticks = 0;
                if ((e = launch_gap_order(G.gap_flag, lo, hi, ctx->gap_list.as<u32>() + 16, ctx->gap_list.as<u32>(), ctx->stream)) != hipSuccess) return e;
                G.arena_bytes = arena2; G.next = ctx->gap_next.as<u32>() + 8; G.big = 1; G.coop = 1;
                if ((e = launch_gap(G, ctx->gap_team, std::min(w2, m), ctx->stream)) != hipSuccess) return e;
            }
            G.work_cap = ~0ULL; G.cap_ticks = 0; G.big = 1; G.coop = 1;
            if ((e = launch_gap_order(G.gap_flag, lo, hi, ctx->gap_list.as<u32>() + 16, ctx->gap_list.as<u32>(), ctx->stream)) != hipSuccess) return e;
            G.arena_bytes = arena3; G.next = ctx->gap_next.as<u32>() + 24; G.last = 1;
            return launch_gap(G, ctx->gap_team, std::min(w3, m), ctx->stream);
        };
        // The stream state (GapArgs): once a read of the stream has extended, every later read starts "extended" -- one ladder over the batch.
        // Until then the batch is taken in growing chunks: a probe ladder finds the chunk's first extending read r* (all reads started "not
        // extended", nothing written), then the chunk is done for good with the reads behind r* started "extended".  The state is kept in
        // the context across batches (one context = one read stream in file order, the reference's `-t 1`; lnr_gap_stream).
        int ext_state = ctx->gap_ext;
        u32 lo = 0;
        for (u32 chunk = 256; lo < n && !ext_state; chunk = chunk < (1u << 20) ? chunk * 4 : chunk) {
            u32 hi = (u32)std::min<u64>(n, (u64)lo + chunk), first = 0xffffffffu;
            HIPCK(hipMemsetAsync(ctx->gap_first.p, 0xff, 64, ctx->stream));
            HIPCK(ladder(lo, hi, 0xffffffffu, 1));
            HIPCK(hipMemcpyAsync(&first, ctx->gap_first.p, 4, hipMemcpyDeviceToHost, ctx->stream));
            HIPCK(hipStreamSynchronize(ctx->stream));
            HIPCK(ladder(lo, hi, first == 0xffffffffu ? first : first + 1, 0));
            if (first != 0xffffffffu) ext_state = 1;
            lo = hi;
        }
        if (lo < n) HIPCK(ladder(lo, n, 0, 0));
        ext_state_out = ext_state;
        ctx->t_gap.stop(ctx->stream);
#ifdef LNR_GAP_DEVPROF
        {
            unsigned long long hp[96];
            HIPCK(hipMemcpyAsync(hp, ctx->gap_prof.p, sizeof hp, hipMemcpyDeviceToHost, ctx->stream));
            HIPCK(hipStreamSynchronize(ctx->stream));
            static const char *nm[10] = {"sort k-mers", "join", "k-mer stream", "sort anchors", "chain DP", "traceback", "chain tiles", "map along chain (incl.)", "tiles from chain", "filter anchors (sorts)"};
            for (int L = 0; L < 3; L++) {
                const unsigned long long *q = hp + 16 * L;
                fprintf(stderr, "[gap prof] launch %d: reads %llu, lane/wave time %.1f ms in total, slowest read %.1f ms\n", L, q[12], q[11] / 1e5, q[15] / 1e5);
                const unsigned long long *w = hp + 48 + 16 * L;
                fprintf(stderr, "[gap prof]    slowest read: index %llu, length %llu, cords in %llu, arena high-water %llu bytes; its longest chain DP: %.1f ms, %llu anchors, score fn %llu, %s\n", w[10], w[11], w[12], w[13],
                        (double)(w[14] & ((1ULL << 56) - 1)) / 1e5, hp[90 + L], (w[14] >> 56) & 15, (w[14] >> 60) ? "by columns" : "single wave");
                for (int k = 0; k < 10; k++) fprintf(stderr, "[gap prof]    %-26s %10.1f ms  %5.1f %%   slowest read: %8.1f ms\n", nm[k], q[k] / 1e5, q[11] ? 100.0 * q[k] / q[11] : 0.0, w[k] / 1e5);
            }
            fprintf(stderr, "[gap prof] map along chain, all launches: streams + join + anchor sort %.1f ms, chain DP + traceback + tiles %.1f ms\n", hp[63] / 1e5, hp[79] / 1e5);
            fprintf(stderr, "[gap prof] first launch: at most %llu workers (waves) alive at once\n", hp[95]);
            {   // how well the weight predicts: weights of the reads the team launch did, and of the slowest / all reads of the first launch
                std::vector<u32> wt(n); std::vector<unsigned long long> pr0(n);
                HIPCK(hipMemcpy(wt.data(), ctx->gap_weight.p, (size_t)n * 4, hipMemcpyDeviceToHost));
                HIPCK(hipMemcpy(pr0.data(), (char *)ctx->gap_prof.p + 96 * 8, (size_t)n * 8, hipMemcpyDeviceToHost));
                std::vector<u32> wh, wl; std::vector<std::pair<double, u32> > slow;
                for (u32 i = 0; i < n; i++) { if (!pr0[i]) continue; if ((pr0[i] >> 56) >= 1) wh.push_back(wt[i]); else { wl.push_back(wt[i]); slow.push_back({(double)(pr0[i] & ((1ULL << 56) - 1)) / 1e5, wt[i]}); } }
                std::sort(wh.begin(), wh.end()); std::sort(wl.begin(), wl.end()); std::sort(slow.begin(), slow.end());
                if (!wh.empty() && !wl.empty()) {
                    fprintf(stderr, "[gap prof] weight of team-launch reads: min %u p10 %u p50 %u p90 %u max %u | of first-launch reads: p50 %u p90 %u p99 %u p99.9 %u max %u\n", wh[0], wh[wh.size() / 10], wh[wh.size() / 2], wh[wh.size() * 9 / 10], wh.back(),
                            wl[wl.size() / 2], wl[wl.size() * 9 / 10], wl[wl.size() * 99 / 100], wl[(size_t)(wl.size() * 0.999)], wl.back());
                    fprintf(stderr, "[gap prof] slowest first-launch reads (ms : weight):");
                    for (size_t k = 0; k < 16 && k < slow.size(); k++) fprintf(stderr, " %.0f:%u", slow[slow.size() - 1 - k].first, slow[slow.size() - 1 - k].second);
                    fprintf(stderr, "\n");
                }
            }
            {   // reads in flight over the first launch's duration (start / end ticks of every read, 10 ns)
                std::vector<unsigned long long> se(2 * (size_t)n);
                HIPCK(hipMemcpy(se.data(), (char *)ctx->gap_prof.p + (96 + (size_t)n) * 8, 2 * (size_t)n * 8, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ULL, t1 = 0;
                for (u32 i = 0; i < n; i++) if (se[i]) { t0 = std::min(t0, se[i]); t1 = std::max(t1, se[n + i]); }
                if (t1 > t0) {
                    const int NBK = 20;
                    std::vector<double> busy(NBK, 0.0);
                    double span = (double)(t1 - t0), bw = span / NBK;
                    for (u32 i = 0; i < n; i++) if (se[i]) {
                        double a = (double)(se[i] - t0), b = (double)(se[n + i] - t0);
                        for (int k = (int)(a / bw); k < NBK && k * bw < b; k++) busy[k] += std::min(b, (k + 1) * bw) - std::max(a, k * bw);
                    }
                    fprintf(stderr, "[gap prof] first launch: %.1f ms from the first read's start to the last read's end; mean reads in flight per twentieth:", span / 1e5);
                    for (int k = 0; k < NBK; k++) fprintf(stderr, " %.0f", busy[k] / bw);
                    fprintf(stderr, "\n");
                }
            }
            std::vector<unsigned long long> pr(n);
            HIPCK(hipMemcpy(pr.data(), (char *)ctx->gap_prof.p + 96 * 8, (size_t)n * 8, hipMemcpyDeviceToHost));
            for (int L = 0; L < 3; L++) {
                std::vector<double> t;
                for (u32 i = 0; i < n; i++) if (pr[i] && (int)(pr[i] >> 56) == L) t.push_back((double)(pr[i] & ((1ULL << 56) - 1)) / 1e5);
                if (t.empty()) continue;
                std::sort(t.begin(), t.end());
                double sum = 0; for (double v : t) sum += v;
                fprintf(stderr, "[gap prof] launch %d per-read ms: n %zu sum %.1f p50 %.3f p90 %.3f p99 %.3f p99.9 %.3f max %.3f | top:", L, t.size(), sum, t[t.size() / 2], t[t.size() * 9 / 10], t[t.size() * 99 / 100], t[(size_t)(t.size() * 0.999)], t.back());
                for (size_t k = 0; k < 12 && k < t.size(); k++) fprintf(stderr, " %.1f", t[t.size() - 1 - k]);
                fprintf(stderr, "\n");
            }
        }
#endif
    }
    std::vector<u32> nout(n);
    std::vector<i32> rerr(n);
    u32 gap_second = 0;
    {
        Readback rb;
        if (!rb.begin(ctx->h_rb[0], (size_t)n * 8 + 256)) { ctx->err = "pinned host allocation failed"; return LNR_ERR_NOMEM; }
        HIPCK(rb.add(nout.data(), ctx->nout.p, (size_t)n * 4, ctx->stream));
        HIPCK(rb.add(rerr.data(), ctx->read_err.p, (size_t)n * 4, ctx->stream));
        if (ctx->opts.gap_len) HIPCK(rb.add(&gap_second, ctx->gap_next.as<u32>() + 16, 4, ctx->stream));
        HIPCK(hipStreamSynchronize(ctx->stream));
        rb.finish();
    }
    ctx->stats.tail_ms += ctx->t_tail.ms();
    if (ctx->opts.gap_len) { ctx->stats.gap_ms += ctx->t_gap.ms(); ctx->stats.gap_second_pass += gap_second; }
    for (u32 i = 0; i < n; i++)
        if (rerr[i]) {
            // A read outgrew a per-read capacity (cords: 16 per 64 bases + 256; gaps: one per 1000 bases + 4 -- heuristics, generous by an
            // order of magnitude).  Nothing of the batch is handed out; the batch is run again with 4x, then 16x the capacities.
            if (attempt < 2) {
                ctx->t_total.stop(ctx->stream);
                HIPCK(hipStreamSynchronize(ctx->stream));
                ctx->cap_scale = attempt == 0 ? 4 : 16;
                ctx->overflow_reruns++;
                lnr_status rs = filter_dev(ctx, d_reads, d_off, n, out, h_off, attempt + 1);
                ctx->cap_scale = 1;
                return rs;
            }
            char b[160];
            snprintf(b, sizeof b, "device capacity overflow on read %u (stage code %d, length %u) with 16x capacities", i, rerr[i], B.len[i]);
            ctx->err = b;
            return LNR_ERR_INTERNAL;
        }
    ctx->gap_ext = ext_state_out;        // (only a batch that went through: a re-run after an overflow starts from the state the batch met)
    ctx->h_cord_off.assign((size_t)n + 1, 0);
    for (u32 i = 0; i < n; i++) ctx->h_cord_off[i + 1] = ctx->h_cord_off[i] + nout[i];
    u64 tot = ctx->h_cord_off[n];
    ENSURE(ctx->r_str, std::max<u64>(tot * 8, 16));
    ENSURE(ctx->r_end, std::max<u64>(tot * 8, 16));
    {
        void *h = ctx->r_off.host_stage(((size_t)n + 1) * 8);
        if (!h) { ctx->err = "pinned host allocation failed"; return LNR_ERR_NOMEM; }
        memcpy(h, ctx->h_cord_off.data(), ((size_t)n + 1) * 8);
        HIPCK(words_in(ctx->r_off.p, h, ((size_t)n + 1) * 8, ctx->stream));
    }
    hipLaunchKernelGGL(k_gather_out, dim3(n), dim3(64), 0, ctx->stream, ctx->out_str.as<u64>(), ctx->out_end.as<u64>(), ctx->cords_off.as<u64>(), ctx->nout.as<u32>(),
                       ctx->r_off.as<u64>(), n, ctx->r_str.as<u64>(), ctx->r_end.as<u64>());
    KCHECK();
    ctx->t_total.stop(ctx->stream);
    HIPCK(hipStreamSynchronize(ctx->stream));
    laps.lap("tailB+gather");
    laps.done();
    ctx->stats.prep_ms = ctx->t_prep.ms();
    ctx->stats.total_ms = ctx->t_total.ms();
    ctx->stats.cords = tot;
    finish_stats(ctx, B);
    ctx->last_ncords = tot;
    if (out) { out->n_cords = tot; out->d_cord_off = ctx->r_off.as<u64>(); out->d_cords_str = ctx->r_str.as<u64>(); out->d_cords_end = ctx->r_end.as<u64>(); }
    return LNR_OK;
}

lnr_status seed_dev(lnr_ctx *ctx, const u8 *d_reads, const u64 *d_off, u32 n, bool to_host) {
    if (!ctx->has_index) { ctx->err = "no index: call lnr_index_build or lnr_index_adopt first"; return LNR_ERR_NO_INDEX; }
    reset_stats(ctx);
    ctx->h_anchor_off.assign((size_t)n + 1, 0);
    ctx->h_anchors.clear();
    if (n == 0) return LNR_OK;
    ctx->t_total.start(ctx->stream);
    BatchHost B;
    lnr_status s = prepare_batch(ctx, d_reads, d_off, n, B);
    if (s != LNR_OK) return s;
    HostJobs j0;
    std::vector<u32> job_of(n, 0xffffffffu);
    for (u32 i = 0; i < n; i++) {
        if (B.len[i] >= 43) { job_of[i] = j0.size(); j0.grp_beg.push_back(j0.size()); j0.add(i, 0, B.len[i], 0); }
    }
    j0.grp_beg.push_back(j0.size());
    if ((s = seed_jobs(ctx, ctx->js[0], j0, ctx->stream, n)) != LNR_OK) return s;
    if (to_host && (s = export_anchors(ctx, ctx->js[0], j0.size())) != LNR_OK) return s;
    HIPCK(hipStreamWaitEvent(ctx->stream, ctx->ev_f1, 0));
    ctx->t_total.stop(ctx->stream);
    HIPCK(hipStreamSynchronize(ctx->stream));
    ctx->stats.prep_ms = ctx->t_prep.ms();
    ctx->stats.total_ms = ctx->t_total.ms();
    // stats.seed_bytes with every read counted
    u64 rb = 0;
    for (u32 i = 0; i < n; i++) if (job_of[i] != 0xffffffffu) rb += (B.len[i] + 3) / 4;
    ctx->stats.seed_bytes = rb + ctx->stats.lookups * 8 + ctx->stats.bucket_entries * 8 + ctx->stats.anchors * 8;
    if (to_host) {
        // re-index the per-job CSR by read (reads without a job get the bare dummy)
        std::vector<u64> off((size_t)n + 1, 0), vals;
        for (u32 i = 0; i < n; i++) {
            u64 c = job_of[i] == 0xffffffffu ? 1 : ctx->h_anchor_off[job_of[i] + 1] - ctx->h_anchor_off[job_of[i]];
            off[i + 1] = off[i] + c;
        }
        vals.assign(off[n], 0);
        for (u32 i = 0; i < n; i++)
            if (job_of[i] != 0xffffffffu) {
                u64 a = ctx->h_anchor_off[job_of[i]], c = ctx->h_anchor_off[job_of[i] + 1] - a;
                memcpy(vals.data() + off[i], ctx->h_anchors.data() + a, c * 8);
            }
        ctx->h_anchor_off.swap(off);
        ctx->h_anchors.swap(vals);
    }
    return LNR_OK;
}

// memcpy of a large block by a few threads (a pageable source is first copied into pinned staging; one thread moves ~10 GB/s)
void par_memcpy(void *dst, const void *src, size_t len) {
    const size_t MIN = 4u << 20;
    unsigned T = (unsigned)std::min<size_t>(4, len / MIN);
    if (T < 2) { memcpy(dst, src, len); return; }
    std::vector<std::thread> th;
    size_t per = (len / T + 63) & ~(size_t)63;
    for (unsigned t = 1; t < T; t++) {
        size_t o = (size_t)t * per, l = o < len ? std::min(per, len - o) : 0;
        if (l) th.emplace_back([=]() { memcpy((char *)dst + o, (const char *)src + o, l); });
    }
    memcpy(dst, src, std::min(per, len));
    for (auto &t : th) t.join();
}

// Upload of one batch into input slot `slot`, asynchronously on the copy stream; ev_in[slot] marks its end.
lnr_status submit_reads(lnr_ctx *ctx, int slot, const u8 *reads, const u64 *off, u32 n) {
    if (!off || (n && !reads)) { ctx->err = "null read buffer"; return LNR_ERR_ARG; }
    for (u32 i = 0; i < n; i++)
        if (off[i + 1] < off[i]) { ctx->err = "read offsets not monotone"; return LNR_ERR_ARG; }   // (before anything is sized by them)
    u64 base = off[0], total = off[n] - off[0];
    DevBuf &dr = ctx->in_reads[slot], &dof = ctx->in_off[slot];
    ENSURE(dr, std::max<u64>(total, 16));
    ENSURE(dof, ((size_t)n + 1) * 8);
    if (!ctx->h_off[slot].ensure(((size_t)n + 1) * 8)) { ctx->err = "pinned host allocation failed"; return LNR_ERR_NOMEM; }
    u64 *o = ctx->h_off[slot].as<u64>();
    for (u32 i = 0; i <= n; i++) o[i] = off[i] - base;
    hipStream_t sc = ctx->s_copy;
    if (total) {
        hipPointerAttribute_t at;
        bool pinned = hipPointerGetAttributes(&at, reads + base) == hipSuccess && at.type == hipMemoryTypeHost;
        if (!pinned) (void)hipGetLastError();
        if (pinned) {
            // the caller filled memory from lnr_host_alloc (or registered its own): one DMA at link rate, no staging copy
            HIPCK(hipMemcpyAsync(dr.p, reads + base, total, hipMemcpyHostToDevice, sc));
        } else {
            // pageable source: through two pinned staging buffers, so that the host copy of one chunk overlaps the DMA of the last
            const u64 CH = 32ULL << 20;
            for (int k = 0; k < 2; k++) {
                if (!ctx->h_up[k].ensure(CH)) { ctx->err = "pinned host allocation failed"; return LNR_ERR_NOMEM; }
                if (!ctx->ev_up[k]) HIPCK(hipEventCreateWithFlags(&ctx->ev_up[k], hipEventDisableTiming));
            }
            // (the staging buffers and their events belong to the context: a second submit right behind the first -- two batches in flight --
            //  must wait for the first one's DMA out of a buffer as well.  An event that was never recorded reads as complete.)
            int k = 0;
            for (u64 o2 = 0; o2 < total; o2 += CH, k ^= 1) {
                u64 len = std::min<u64>(CH, total - o2);
                HIPCK(hipEventSynchronize(ctx->ev_up[k]));                // the last DMA out of this staging buffer has finished
                par_memcpy(ctx->h_up[k].p, reads + base + o2, len);
                HIPCK(hipMemcpyAsync(dr.as<u8>() + o2, ctx->h_up[k].p, len, hipMemcpyHostToDevice, sc));
                HIPCK(hipEventRecord(ctx->ev_up[k], sc));
            }
        }
    }
    HIPCK(hipMemcpyAsync(dof.p, o, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, sc));
    HIPCK(hipEventRecord(ctx->ev_in[slot], sc));
    ctx->in_n[slot] = n;
    return LNR_OK;
}

// restores the caller's current device when an entry point returns (a context may live on another device than the one the
// caller's own HIP / torch code is using)
struct DevGuard {
    int prev = -1;
    explicit DevGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

// ======================================================================= C ABI ====
extern "C" {

void lnr_opts_default(lnr_opts *o) {
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->device = -1; o->index_type = 1; o->feature_type = 2; o->preset = 1; o->gap_len = 0; o->dup = 0; o->scratch_budget = 0;
}

const char *lnr_strerror(lnr_status s) {
    switch (s) {
        case LNR_OK: return "ok";
        case LNR_ERR_ARG: return "invalid argument";
        case LNR_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU path)";
        case LNR_ERR_HIP: return "HIP runtime error";
        case LNR_ERR_NOMEM: return "out of memory";
        case LNR_ERR_NO_INDEX: return "index not built";
        case LNR_ERR_LIMIT: return "input exceeds a format limit";
        case LNR_ERR_UNSUPPORTED: return "option not supported by this build";
        case LNR_ERR_INTERNAL: return "internal capacity overflow";
    }
    return "unknown status";
}
const char *lnr_last_error(const lnr_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

lnr_status lnr_create(const lnr_opts *opts, lnr_ctx **out) {
    if (!out) return LNR_ERR_ARG;
    *out = nullptr;
    lnr_opts o;
    if (opts) o = *opts; else lnr_opts_default(&o);
    if ((o.index_type != 1 && o.index_type != 2) || o.feature_type != 2 || (o.preset != 1 && o.preset != 2) || o.dup > 1) return LNR_ERR_UNSUPPORTED;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return LNR_ERR_NO_DEVICE; }
    int dev = o.device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) return LNR_ERR_NO_DEVICE; }
    if (dev >= ndev) return LNR_ERR_ARG;
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    if (hipSetDevice(dev) != hipSuccess) return LNR_ERR_NO_DEVICE;
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore_{prev_dev == dev ? -1 : prev_dev};
    lnr_ctx *ctx = new (std::nothrow) lnr_ctx();
    if (!ctx) return LNR_ERR_NOMEM;
    ctx->opts = o;
    ctx->device = dev;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return LNR_ERR_HIP; }
    if (const char *e = getenv("LNR_CAP_SHRINK")) { long v = atol(e); if (v >= 1 && v <= 4096) ctx->cap_shrink = (u32)v; }
    if (const char *e = getenv("LNR_GAP_MODE")) ctx->gap_mode = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LNR_GAP_FUSED")) ctx->gap_fused = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LNR_GAP_TEAMS")) { long v = atol(e); if (v >= 1 && v <= 4096) ctx->gap_teams = (u32)v; }
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, ctx->device) == hipSuccess) ctx->ncu = (u32)pr.multiProcessorCount; else (void)hipGetLastError(); }
    if (const char *e = getenv("LNR_GAP_HEAVY_W")) { long v = atol(e); if (v >= 1) ctx->gap_heavy_w = (u32)v; }
    if (const char *e = getenv("LNR_GAP_CAP_MS")) { long v = atol(e); if (v >= 0 && v <= 100000) ctx->gap_cap_ms = (u32)v; }
    if (const char *e = getenv("LNR_SEED_BH")) ctx->use_bh = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LNR_GAP_WAVES")) { long v = atol(e); if (v >= 1 && v <= (1 << 20)) ctx->gap_waves = (u32)v; }
    if (const char *e = getenv("LNR_GAP_ARENA2_MB")) { long v = atol(e); if (v >= 1 && v <= 1024) ctx->gap_arena2_mb = (u32)v; }
    if (const char *e = getenv("LNR_GAP_TEAM")) ctx->gap_team = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LNR_GAP_WORK_CAP")) { long long v = atoll(e); if (v >= 0) ctx->gap_work_cap = (u64)v; }
    if (const char *e = getenv("LNR_SEED_LDS_PAD")) { long v = atol(e); if (v >= 0 && v <= 100000) ctx->seed_lds_pad = (u32)v; }
    if (const char *e = getenv("LNR_JOB_LDS_KB")) { long kb = atol(e); if (kb >= 1 && kb <= 156) ctx->job_lds_bytes = (size_t)kb * 1024; }
    if (const char *e = getenv("LNR_JOB_STAGE_KB")) { long kb = atol(e); if (kb >= 0 && kb <= 60) ctx->job_stage_bytes = (size_t)kb * 1024; }
    if (const char *e = getenv("LNR_HEAVY_CAP")) { long v = atol(e); if (v >= 64) { ctx->heavy_cap = (u32)std::min<long>(v, 0xffffffffL); ctx->heavy_cap_r1 = ctx->heavy_cap; } }
    if (const char *e = getenv("LNR_HEAVY_LDS_KB")) { long v = atol(e); if (v >= 1 && v <= 56) ctx->heavy_lds_kb = (u32)v; }
    if (const char *e = getenv("LNR_MID_CAP")) { long v = atol(e); if (v >= 64) { ctx->mid_cap_env = true; ctx->mid_cap = (u32)std::min<long>(v, 0xffffffffL); ctx->mid_cap_r1 = ctx->mid_cap; } }
    if (const char *e = getenv("LNR_MID_LDS_KB")) { long v = atol(e); if (v >= 1 && v <= 56) ctx->mid_lds_kb = (u32)v; }
    if (const char *e = getenv("LNR_HEAVY_CAP_R1")) { long v = atol(e); if (v >= 64) ctx->heavy_cap_r1 = (u32)std::min<long>(v, 0xffffffffL); }
    if (const char *e = getenv("LNR_MID_CAP_R1")) { long v = atol(e); if (v >= 64) ctx->mid_cap_r1 = (u32)std::min<long>(v, 0xffffffffL); }
    if (const char *e = getenv("LNR_DP_SPLIT_CAP")) { long v = atol(e); if (v >= 64) { ctx->dp_split_cap = (u32)std::min<long>(v, 0xffffffffL); ctx->dp_split_cap_r1 = ctx->dp_split_cap; } }
    if (const char *e = getenv("LNR_DP_SPLIT_CAP_R1")) { long v = atol(e); if (v >= 64) ctx->dp_split_cap_r1 = (u32)std::min<long>(v, 0xffffffffL); }
    if (const char *e = getenv("LNR_MID_WAVES")) ctx->mid_waves = atoi(e) == 2 ? 2 : 4;
    if (const char *e = getenv("LNR_POST_SPLIT")) ctx->post_split = atoi(e) != 0;
    if (const char *e = getenv("LNR_SEED_BM")) ctx->seed_bm = atoi(e) ? 1 : 0;
    if (const char *e = getenv("LNR_STOP_AFTER")) { long v = atol(e); if (v >= 0 && v < 16) ctx->stop_after = (u32)v; }
    if (const char *e = getenv("LNR_PREP_GRID")) { long v = atol(e); if (v > 0) ctx->prep_grid = (u32)v; }
    if (const char *e = getenv("LNR_PREP_THREADS")) { long v = atol(e); if (v == 64 || v == 128 || v == 256) ctx->prep_threads = (u32)v; }
    if (const char *e = getenv("LNR_BULK_DELAY_US")) { long v = atol(e); if (v >= 0 && v <= 5000) ctx->bulk_delay_ticks = (u32)v * 100; }
    if (const char *e = getenv("LNR_LANE_ORDER")) ctx->lane_bulk_first = e[0] != 'h';
    if (const char *e = getenv("LNR_SPLIT_CAP")) { long v = atol(e); if (v >= 1) ctx->split_cap = (u32)std::min<long>(v, 0xffffffffL); }
    // Three streams in all: the runtime multiplexes streams onto a few hardware queues (4 by default) and two streams on
    // one queue run their kernels back to back (measured: the bulk kernel waited for the 4-wave kernel).  Lane 1 (bulk)
    // uses the main stream for its multi-wave kernels, seeds and tails and one side stream for the single-wave kernel;
    // lane 0 (heavy reads, nearly all multi-wave) runs everything on one stream.
    bool ok = hipEventCreateWithFlags(&ctx->ev_start, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ctx->ev_prep, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&ctx->ev_f1, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&ctx->s_multi[0], hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&ctx->s_bulk[1], hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&ctx->s_tail, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&ctx->s_copy, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&ctx->s_down, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&ctx->ev_down, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < 3 && ok; k++) ok = hipEventCreateWithFlags(&ctx->ev_in[k], hipEventDisableTiming) == hipSuccess;
    ctx->s_bulk[0] = ctx->s_multi[0];
    if (getenv("LNR_LANE0_BULK_STREAM")) ok = ok && hipStreamCreateWithFlags(&ctx->s_bulk[0], hipStreamNonBlocking) == hipSuccess;   // experiment: own stream for lane 0's single-wave kernel
    ctx->s_multi[1] = ctx->stream;
    for (int l = 0; l < 2 && ok; l++)
        ok = hipEventCreateWithFlags(&ctx->ev_fork[l], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ctx->ev_join[l], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&ctx->ev_lane[l], hipEventDisableTiming) == hipSuccess;
    if (!ok) { lnr_destroy(ctx); return LNR_ERR_HIP; }
    ctx->t_prep.init(); ctx->t_job.init(); ctx->t_tail.init(); ctx->t_total.init(); ctx->t_gap.init();
    *out = ctx;
    return LNR_OK;
}

void lnr_destroy(lnr_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int l = 0; l < 2; l++) {
        if (ctx->s_multi[l]) (void)hipStreamSynchronize(ctx->s_multi[l]);
        if (ctx->s_bulk[l]) (void)hipStreamSynchronize(ctx->s_bulk[l]);
    }
    ctx->t_prep.destroy(); ctx->t_job.destroy(); ctx->t_tail.destroy(); ctx->t_total.destroy();
    for (int l = 0; l < 2; l++) {
        ctx->js[l].t_seed.destroy();
        if (ctx->ev_fork[l]) (void)hipEventDestroy(ctx->ev_fork[l]);
        if (ctx->ev_join[l]) (void)hipEventDestroy(ctx->ev_join[l]);
        if (ctx->ev_lane[l]) (void)hipEventDestroy(ctx->ev_lane[l]);
    }
    if (ctx->s_tail) { (void)hipStreamSynchronize(ctx->s_tail); (void)hipStreamDestroy(ctx->s_tail); }
    if (ctx->s_copy) { (void)hipStreamSynchronize(ctx->s_copy); (void)hipStreamDestroy(ctx->s_copy); }
    if (ctx->s_down) { (void)hipStreamSynchronize(ctx->s_down); (void)hipStreamDestroy(ctx->s_down); }
    if (ctx->ev_down) (void)hipEventDestroy(ctx->ev_down);
    if (ctx->ev_done) (void)hipEventDestroy(ctx->ev_done);
    for (int k = 0; k < 3; k++) if (ctx->ev_in[k]) (void)hipEventDestroy(ctx->ev_in[k]);
    if (ctx->s_bulk[0] && ctx->s_bulk[0] != ctx->s_multi[0]) (void)hipStreamDestroy(ctx->s_bulk[0]);
    if (ctx->s_multi[0]) (void)hipStreamDestroy(ctx->s_multi[0]);
    if (ctx->s_bulk[1]) (void)hipStreamDestroy(ctx->s_bulk[1]);
    for (int k = 0; k < 2; k++) if (ctx->ev_up[k]) (void)hipEventDestroy(ctx->ev_up[k]);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_prep) (void)hipEventDestroy(ctx->ev_prep);
    if (ctx->ev_f1) (void)hipEventDestroy(ctx->ev_f1);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

lnr_status lnr_index_build(lnr_ctx *ctx, const uint8_t *const *seq, const uint64_t *len, uint32_t nseq, uint32_t T) {
    if (!ctx) return LNR_ERR_ARG;
    if (!seq || !len || nseq == 0) { ctx->err = "null/empty sequence set"; return LNR_ERR_ARG; }
    if (nseq >= 1024) { ctx->err = "at most 1023 reference sequences (cord id field; linear.cpp:107)"; return LNR_ERR_LIMIT; }
    if (T == 0) T = 1;
    for (u32 i = 0; i < nseq; i++) {
        if (!seq[i]) { ctx->err = "null sequence pointer"; return LNR_ERR_ARG; }
        if (len[i] >= (1ULL << 30) - (1ULL << 20)) { ctx->err = "sequence too long for the 30-bit x field (cords.cpp:13-14)"; return LNR_ERR_LIMIT; }
    }
    DevGuard dg_(ctx->device);
    ctx->has_index = false;
    set_index_layout(ctx, len, nseq);
    ctx->info.layout_threads = T;
    lnr_status s;
    if ((s = upload_index_layout(ctx)) != LNR_OK) return s;
    // genome: padded device copy (zero padding pins the reference's out-of-range reads to 'A')
    ENSURE(ctx->g, ctx->info.genome_bytes + 64);
    HIPCK(hipMemsetAsync(ctx->g.p, 0, ctx->info.genome_bytes + 64, ctx->stream));
    for (u32 i = 0; i < nseq; i++)
        if (len[i]) HIPCK(hipMemcpyAsync(ctx->g.as<u8>() + ctx->seq_off[i], seq[i], len[i], hipMemcpyDefault, ctx->stream));   // host or device source
    Timer tm; tm.init();
    tm.start(ctx->stream);
    {   // ordinals above 4 -> N
        u64 n16 = (ctx->info.genome_bytes + 64) / 16;
        hipLaunchKernelGGL(k_clamp_bases, dim3((u32)((n16 + 255) / 256)), dim3(256), 0, ctx->stream, ctx->g.as<u8>(), n16);
    }
    if (ctx->opts.index_type == 2) {   // HIndex: own build; genome features as for the DIndex
        auto fail = [&](lnr_status st_) { tm.destroy(); return st_; };
        lnr_status hst = build_hindex(ctx, len, nseq, T);
        if (hst != LNR_OK) return fail(hst);
        if (!ctx->f2.ensure(std::max<u64>(ctx->info.f2_len * sizeof(F96), 16))) { ctx->err = "device allocation failed during index build"; return fail(LNR_ERR_NOMEM); }
        if (ctx->info.f2_len)
            hipLaunchKernelGGL(k_f2, dim3((u32)((ctx->info.f2_len + 255) / 256)), dim3(256), 0, ctx->stream, ctx->g.as<u8>(), ctx->d_seq_off.as<u64>(), ctx->d_f2_off.as<u64>(), nseq,
                               ctx->info.f2_len, ctx->f2.as<F96>());
        tm.stop(ctx->stream);
        hipError_t he = hipStreamSynchronize(ctx->stream);
        if (he == hipSuccess) he = hipGetLastError();
        if (he != hipSuccess) { ctx->err = std::string("HIndex build: ") + hipGetErrorString(he); return fail(LNR_ERR_HIP); }
        ctx->info.build_ms = tm.ms();
        tm.destroy();
        ctx->has_index = true;
        return LNR_OK;
    }
    // chunks of the T-thread layout (index_util.cpp:1654-1666)
    std::vector<ChunkDesc> chunks;
    u64 nsamp = 0;
    for (u32 i = 0; i < nseq; i++)
        for (u32 t = 0; t < T; t++) {
            i64 ts, te;
            chunk_bounds(len[i], T, t, ts, te);
            if (ts >= te) continue;
            u64 ns = chunk_num_samples(ts, te);
            if (!ns) continue;
            ChunkDesc c; c.seq_off = ctx->seq_off[i]; c.t_str = ts; c.samp_base = nsamp; c.nsamp = (u32)ns; c.seq_id = i; c.ks = 0; c.C = 0;
            chunks.push_back(c);
            nsamp += ns;
        }
    if (nsamp >= (1ULL << 32) - 2) { ctx->err = "too many genome samples"; return LNR_ERR_LIMIT; }
    ctx->info.n_samples = nsamp;
    u64 dir_len = ctx->info.dir_len;
    ENSURE(ctx->dir, dir_len * 4);
    DevBuf d_chunks, Xs, vals, cnt, blk, scan_tmp, big, nbig;
    auto cleanup = [&]() { d_chunks.release(); Xs.release(); vals.release(); cnt.release(); blk.release(); scan_tmp.release(); big.release(); nbig.release(); tm.destroy(); };
#define IXCK(expr) do { lnr_status s_ = (expr); if (s_ != LNR_OK) { cleanup(); return s_; } } while (0)
#define IXHIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); (void)hipGetLastError(); cleanup(); return LNR_ERR_HIP; } } while (0)
#define IXENS(b, bytes) do { if (!(b).ensure(bytes)) { ctx->err = "device allocation failed during index build"; cleanup(); return LNR_ERR_NOMEM; } } while (0)
    IXENS(cnt, dir_len * 4);
    IXHIP(hipMemsetAsync(cnt.p, 0, dir_len * 4, ctx->stream));
    u64 hs_len = 0;
    if (nsamp) {
        IXCK(upload(ctx, d_chunks, chunks));
        IXENS(Xs, nsamp * 4);
        IXENS(vals, nsamp * 8);
        u32 nch = (u32)chunks.size();
        hipLaunchKernelGGL(k_ix_chunk_const, dim3(nch), dim3(256), 0, ctx->stream, ctx->g.as<u8>(), (u64)(ctx->info.genome_bytes + 64), d_chunks.as<ChunkDesc>(), nch);
        IXHIP(hipGetLastError());
        hipLaunchKernelGGL(k_ix_sample, dim3((u32)((nsamp + 255) / 256)), dim3(256), 0, ctx->stream, ctx->g.as<u8>(), d_chunks.as<ChunkDesc>(), nch, nsamp, Xs.as<u32>(), vals.as<u64>());
        IXHIP(hipGetLastError());
        u32 nrb = (u32)((nsamp + REC_BLK - 1) / REC_BLK);
        IXENS(blk, (size_t)nrb * 4);
        hipLaunchKernelGGL(k_ix_start_blk, dim3(nrb), dim3(REC_TPB), 0, ctx->stream, Xs.as<u32>(), nsamp, blk.as<u32>());
        IXHIP(hipGetLastError());
        hipLaunchKernelGGL(k_max_top, dim3(1), dim3(1024), 0, ctx->stream, blk.as<u32>(), nrb);
        IXHIP(hipGetLastError());
        hipLaunchKernelGGL(k_ix_rec, dim3(nrb), dim3(REC_TPB), 0, ctx->stream, Xs.as<u32>(), nsamp, blk.as<u32>(), cnt.as<i32>());
        IXHIP(hipGetLastError());
        hipLaunchKernelGGL(k_ix_omit, dim3((u32)((dir_len + 255) / 256)), dim3(256), 0, ctx->stream, cnt.as<i32>(), dir_len);
        IXHIP(hipGetLastError());
    }
    IXCK(dev_scan_i32(ctx, cnt.as<i32>(), ctx->dir.as<i32>(), dir_len, scan_tmp));
    i32 total = 0;
    IXHIP(hipMemcpyAsync(&total, ctx->dir.as<i32>() + (dir_len - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    IXHIP(hipStreamSynchronize(ctx->stream));
    hs_len = (u64)total;
    IXENS(ctx->hs, std::max<u64>(hs_len * 8, 16));
    if (nsamp && hs_len) {
        IXHIP(hipMemsetAsync(cnt.p, 0, dir_len * 4, ctx->stream));
        hipLaunchKernelGGL(k_ix_scatter, dim3((u32)((nsamp + 255) / 256)), dim3(256), 0, ctx->stream, Xs.as<u32>(), vals.as<u64>(), nsamp, ctx->dir.as<i32>(), cnt.as<i32>(), ctx->hs.as<u64>());
        IXHIP(hipGetLastError());
        IXENS(big, (hs_len / 33 + 2) * 4);
        IXENS(nbig, 16);
        IXHIP(hipMemsetAsync(nbig.p, 0, 4, ctx->stream));
        u64 nb = dir_len - 1;
        hipLaunchKernelGGL(k_ix_sort_small, dim3((u32)((nb + 255) / 256)), dim3(256), 0, ctx->stream, ctx->dir.as<i32>(), nb, ctx->hs.as<u64>(), big.as<u32>(), nbig.as<u32>());
        IXHIP(hipGetLastError());
        u32 hb = 0;
        IXHIP(hipMemcpyAsync(&hb, nbig.p, 4, hipMemcpyDeviceToHost, ctx->stream));
        IXHIP(hipStreamSynchronize(ctx->stream));
        if (hb) {
            hipLaunchKernelGGL(k_ix_sort_big, dim3(hb), dim3(64), 0, ctx->stream, ctx->dir.as<i32>(), ctx->hs.as<u64>(), big.as<u32>(), hb);
            IXHIP(hipGetLastError());
        }
    }
    IXCK(build_seed_view(ctx));   // bucket bitmap, bucket lines, overflow lines for the seed kernel
    // genome window features
    ENSURE(ctx->f2, std::max<u64>(ctx->info.f2_len * sizeof(F96), 16));
    if (ctx->info.f2_len) {
        hipLaunchKernelGGL(k_f2, dim3((u32)((ctx->info.f2_len + 255) / 256)), dim3(256), 0, ctx->stream, ctx->g.as<u8>(), ctx->d_seq_off.as<u64>(), ctx->d_f2_off.as<u64>(), nseq,
                           ctx->info.f2_len, ctx->f2.as<F96>());
        IXHIP(hipGetLastError());
    }
    tm.stop(ctx->stream);
    IXHIP(hipStreamSynchronize(ctx->stream));
    ctx->info.build_ms = tm.ms();
    ctx->info.hs_len = hs_len;
    cleanup();
    ctx->has_index = true;
    return LNR_OK;
}

lnr_status lnr_index_info_get(const lnr_ctx *ctx, lnr_index_info *info) {
    if (!ctx || !info) return LNR_ERR_ARG;
    if (!ctx->has_index) return LNR_ERR_NO_INDEX;
    *info = ctx->info;
    return LNR_OK;
}

lnr_status lnr_index_export(lnr_ctx *ctx, int32_t *dir, uint64_t *hs, int32_t *f2, uint64_t *f2_off) {
    if (!ctx) return LNR_ERR_ARG;
    if (!ctx->has_index) return LNR_ERR_NO_INDEX;
    DevGuard dg_(ctx->device);
    HIPCK(hipStreamSynchronize(ctx->stream));
    if (dir) HIPCK(hipMemcpy(dir, ctx->dir.p, ctx->info.dir_len * 4, hipMemcpyDeviceToHost));
    if (hs && ctx->info.hs_len) HIPCK(hipMemcpy(hs, ctx->hs.p, ctx->info.hs_len * 8, hipMemcpyDeviceToHost));
    if (f2 && ctx->info.f2_len) {
        std::vector<F96> tmp(ctx->info.f2_len);
        HIPCK(hipMemcpy(tmp.data(), ctx->f2.p, ctx->info.f2_len * sizeof(F96), hipMemcpyDeviceToHost));
        for (u64 i = 0; i < ctx->info.f2_len; i++) { f2[3 * i] = tmp[i].v0; f2[3 * i + 1] = tmp[i].v1; f2[3 * i + 2] = tmp[i].v2; }
    }
    if (f2_off) memcpy(f2_off, ctx->f2_off.data(), ctx->f2_off.size() * 8);
    return LNR_OK;
}

lnr_status lnr_index_alloc(lnr_ctx *ctx, const lnr_index_info *info, const uint64_t *seq_len) {
    if (!ctx || !info || !seq_len || info->nseq == 0 || info->nseq >= 1024) return LNR_ERR_ARG;
    DevGuard dg_(ctx->device);
    ctx->has_index = false;
    set_index_layout(ctx, seq_len, info->nseq);
    if (ctx->info.genome_bytes != info->genome_bytes || ctx->info.f2_len != info->f2_len || ctx->info.dir_len != info->dir_len) {
        ctx->err = "index info does not match the sequence lengths";
        return LNR_ERR_ARG;
    }
    ctx->info = *info;
    lnr_status s;
    if ((s = upload_index_layout(ctx)) != LNR_OK) return s;
    ENSURE(ctx->g, ctx->info.genome_bytes + 64);
    ENSURE(ctx->dir, ctx->info.dir_len * 4);
    ENSURE(ctx->hs, std::max<u64>(ctx->info.hs_len * 8, 16));
    ENSURE(ctx->f2, std::max<u64>(ctx->info.f2_len * sizeof(F96), 16));
    HIPCK(hipStreamSynchronize(ctx->stream));
    return LNR_OK;
}
lnr_status lnr_index_blob(lnr_ctx *ctx, uint32_t which, void **d_ptr, uint64_t *bytes) {
    if (!ctx || !d_ptr || !bytes) return LNR_ERR_ARG;
    switch (which) {
        case 0: *d_ptr = ctx->g.p; *bytes = ctx->info.genome_bytes; break;
        case 1: *d_ptr = ctx->dir.p; *bytes = ctx->info.dir_len * 4; break;
        case 2: *d_ptr = ctx->hs.p; *bytes = ctx->info.hs_len * 8; break;
        case 3: *d_ptr = ctx->f2.p; *bytes = ctx->info.f2_len * sizeof(F96); break;
        default: return LNR_ERR_ARG;
    }
    if (!*d_ptr) return LNR_ERR_NO_INDEX;
    return LNR_OK;
}
lnr_status lnr_index_adopt(lnr_ctx *ctx) {
    if (!ctx) return LNR_ERR_ARG;
    if (!ctx->g.p || !ctx->dir.p || !ctx->hs.p || !ctx->f2.p) return LNR_ERR_NO_INDEX;
    DevGuard dg_(ctx->device);
    { lnr_status st_ = ctx->opts.index_type == 2 ? hx_derive(ctx) : build_seed_view(ctx); if (st_ != LNR_OK) return st_; }   // derived structures: rebuilt from the received dir / hs (ysa)
    HIPCK(hipStreamSynchronize(ctx->stream));
    ctx->has_index = true;
    return LNR_OK;
}

// ---- one process, several GPUs: the index of ctxs[root] into the other contexts (RCCL between devices, device copies inside one)
namespace {
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load() {
        if (lib) return true;
        for (const char *nm : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) { lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        Broadcast = (decltype(Broadcast))dlsym(lib, "ncclBroadcast"); GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd"); GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        return CommInitAll && CommDestroy && Broadcast && GroupStart && GroupEnd;
    }
};
Rccl g_rccl;
}  // namespace

lnr_status lnr_index_broadcast(lnr_ctx *const *ctxs, uint32_t n, uint32_t root, double *seconds) {
    if (!ctxs || n == 0 || root >= n) return LNR_ERR_ARG;
    for (uint32_t i = 0; i < n; i++) if (!ctxs[i]) return LNR_ERR_ARG;
    lnr_ctx *src = ctxs[root];
    if (!src->has_index) { src->err = "lnr_index_broadcast: the root context has no index"; return LNR_ERR_NO_INDEX; }
    auto t0 = std::chrono::steady_clock::now();
    lnr_status s;
    for (uint32_t i = 0; i < n; i++) {
        if (i == root) continue;
        if (ctxs[i]->opts.index_type != src->opts.index_type) { ctxs[i]->err = "lnr_index_broadcast: contexts with different index types"; return LNR_ERR_ARG; }
        if ((s = lnr_index_alloc(ctxs[i], &src->info, src->seq_len.data())) != LNR_OK) return s;
    }
    // one representative context per device (the root for its own); RCCL between the representatives
    std::vector<uint32_t> rep;
    rep.push_back(root);
    for (uint32_t i = 0; i < n; i++) {
        bool seen = false;
        for (uint32_t r : rep) seen = seen || ctxs[r]->device == ctxs[i]->device;
        if (!seen) rep.push_back(i);
    }
    { DevGuard dg_(src->device); HIPCK_CTX(src, hipStreamSynchronize(src->stream)); }
    if (rep.size() > 1) {
        if (!g_rccl.load()) { src->err = "lnr_index_broadcast: librccl could not be loaded"; return LNR_ERR_HIP; }
        std::vector<int> devs;
        for (uint32_t r : rep) devs.push_back(ctxs[r]->device);
        std::vector<void *> comms(rep.size(), nullptr);
        int rc = g_rccl.CommInitAll(comms.data(), (int)rep.size(), devs.data());
        if (rc != 0) { src->err = std::string("ncclCommInitAll: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"); return LNR_ERR_HIP; }
        for (uint32_t which = 0; which < 4 && rc == 0; which++) {
            g_rccl.GroupStart();
            for (size_t k = 0; k < rep.size() && rc == 0; k++) {
                lnr_ctx *c = ctxs[rep[k]];
                void *p = nullptr; uint64_t bytes = 0;
                if (lnr_index_blob(c, which, &p, &bytes) != LNR_OK) { rc = -1; break; }
                (void)hipSetDevice(c->device);
                rc = g_rccl.Broadcast(p, p, (size_t)bytes, /* ncclUint8 */ 1, /* root = rep[0] */ 0, comms[k], c->stream);
            }
            int rc2 = g_rccl.GroupEnd();
            if (rc == 0) rc = rc2;
        }
        for (size_t k = 0; k < rep.size(); k++) { lnr_ctx *c = ctxs[rep[k]]; (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }
        for (void *cm : comms) if (cm) g_rccl.CommDestroy(cm);
        (void)hipSetDevice(src->device);
        if (rc != 0) { src->err = std::string("ncclBroadcast: ") + (rc > 0 && g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"); return LNR_ERR_HIP; }
    }
    // contexts that share a device with a representative: device-to-device copies from it
    for (uint32_t i = 0; i < n; i++) {
        if (i == root) continue;
        bool is_rep = false; uint32_t from = root;
        for (uint32_t r : rep) { if (r == i) is_rep = true; if (ctxs[r]->device == ctxs[i]->device) from = r; }
        if (!is_rep) {
            DevGuard dg_(ctxs[i]->device);
            for (uint32_t which = 0; which < 4; which++) {
                void *ps = nullptr, *pd = nullptr; uint64_t b1 = 0, b2 = 0;
                if (lnr_index_blob(ctxs[from], which, &ps, &b1) != LNR_OK || lnr_index_blob(ctxs[i], which, &pd, &b2) != LNR_OK || b1 != b2) return LNR_ERR_INTERNAL;
                HIPCK_CTX(ctxs[i], hipMemcpyAsync(pd, ps, b1, hipMemcpyDeviceToDevice, ctxs[i]->stream));
            }
            HIPCK_CTX(ctxs[i], hipStreamSynchronize(ctxs[i]->stream));
        }
    }
    for (uint32_t i = 0; i < n; i++) if (i != root && (s = lnr_index_adopt(ctxs[i])) != LNR_OK) return s;
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return LNR_OK;
}

lnr_status lnr_filter_batch_dev(lnr_ctx *ctx, const uint8_t *d_reads, const uint64_t *d_off, uint32_t n, lnr_cords_dev *out) {
    if (!ctx || !d_off || (n && !d_reads)) return LNR_ERR_ARG;
    DevGuard dg_(ctx->device);
    if (ctx->in_count || ctx->pre.valid) { ctx->err = "batches submitted with lnr_filter_submit are still in flight"; return LNR_ERR_ARG; }
    lnr_status st_ = filter_dev(ctx, d_reads, d_off, n, out);
    ctx->stats_pub = ctx->stats;
    return st_;
}
lnr_status lnr_last_gaps(lnr_ctx *ctx, lnr_gaps *out) {
    if (!ctx || !out) return LNR_ERR_ARG;
    DevGuard dg_(ctx->device);
    u32 n = ctx->last_n;
    out->n_reads = n; out->n_gaps = 0; out->gap_off = nullptr; out->gaps = nullptr;
    ctx->h_gap_off.assign((size_t)n + 1, 0);
    ctx->h_gap_pairs.clear();
    if (n && ctx->last_gaps_off.size() == n) {
        std::vector<u32> ng(n);
        HIPCK(hipStreamSynchronize(ctx->stream));
        HIPCK(hipMemcpy(ng.data(), ctx->ngaps.p, (size_t)n * 4, hipMemcpyDeviceToHost));
        u64 cap_tot = ctx->last_gaps_off[n - 1] + 0;
        (void)cap_tot;
        for (u32 i = 0; i < n; i++) ctx->h_gap_off[i + 1] = ctx->h_gap_off[i] + ng[i];
        ctx->h_gap_pairs.resize(2 * ctx->h_gap_off[n]);
        // the device keeps the gaps of a read at its capacity offset; gather them densely (not on the hot path)
        u64 last = ctx->last_gaps_off[n - 1];
        std::vector<UP> all;
        {
            // capacity of the last read: L / 1000 + 4 for reads longer than 200 (prepare_batch); copy a safe upper bound
            u64 total_cap = last + 4 + (1u << 10);
            if (total_cap * sizeof(UP) > ctx->gaps.cap) total_cap = ctx->gaps.cap / sizeof(UP);
            all.resize(total_cap);
            if (total_cap) HIPCK(hipMemcpy(all.data(), ctx->gaps.p, total_cap * sizeof(UP), hipMemcpyDeviceToHost));
        }
        for (u32 i = 0; i < n; i++)
            for (u32 k = 0; k < ng[i]; k++) {
                const UP &g = all[ctx->last_gaps_off[i] + k];
                ctx->h_gap_pairs[2 * (ctx->h_gap_off[i] + k)] = g.first; ctx->h_gap_pairs[2 * (ctx->h_gap_off[i] + k) + 1] = g.second;
            }
    }
    out->n_gaps = ctx->h_gap_off[n];
    out->gap_off = ctx->h_gap_off.data(); out->gaps = ctx->h_gap_pairs.data();
    return LNR_OK;
}
lnr_status lnr_cords_to_host(lnr_ctx *ctx, lnr_cords *out) {
    if (!ctx || !out) return LNR_ERR_ARG;
    DevGuard dg_(ctx->device);
    u64 tot = ctx->last_ncords;
    const int rs = ctx->res_slot;
    ctx->res_slot ^= 1;
    PinBuf &hs_ = ctx->h_cords_str2[rs], &he_ = ctx->h_cords_end2[rs];
    if (!hs_.ensure(std::max<u64>(tot * 8, 16)) || !he_.ensure(std::max<u64>(tot * 8, 16))) { ctx->err = "pinned host allocation failed"; return LNR_ERR_NOMEM; }
    if (ctx->h_cord_off.size() != (size_t)ctx->last_n + 1) ctx->h_cord_off.assign((size_t)ctx->last_n + 1, 0);
    ctx->h_cord_off2[rs] = ctx->h_cord_off;
    if (tot) {
        HIPCK(hipMemcpyAsync(hs_.p, ctx->r_str.p, tot * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCK(hipMemcpyAsync(he_.p, ctx->r_end.p, tot * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCK(hipStreamSynchronize(ctx->stream));
    }
    out->n_reads = ctx->last_n; out->n_cords = tot;
    out->cord_off = ctx->h_cord_off2[rs].data(); out->cords_str = hs_.as<u64>(); out->cords_end = he_.as<u64>();
    return LNR_OK;
}
void *lnr_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void lnr_host_free(void *p) { if (p) (void)hipHostFree(p); }

lnr_status lnr_filter_submit(lnr_ctx *ctx, const uint8_t *reads, const uint64_t *off, uint32_t n) {
    if (!ctx) return LNR_ERR_ARG;
    DevGuard dg_(ctx->device);
    if (!ctx->has_index) { ctx->err = "no index"; return LNR_ERR_NO_INDEX; }
    if (ctx->in_count + (ctx->pre.valid ? 1 : 0) >= 3) { ctx->err = "three batches already in flight: call lnr_filter_wait first"; return LNR_ERR_ARG; }
    int slot = (ctx->in_head + ctx->in_count) % 3;
    lnr_status s = submit_reads(ctx, slot, reads, off, n);
    if (s != LNR_OK) return s;
    ctx->in_count++;
    return LNR_OK;
}
namespace {
// runs the oldest submitted batch; its results stay on the device (ctx->pre) until lnr_filter_wait hands them out
lnr_status compute_submitted(lnr_ctx *ctx) {
    int slot = ctx->in_head;
    ctx->in_head = (ctx->in_head + 1) % 3; ctx->in_count--;
    lnr_ctx::Pre &P = ctx->pre;
    P.valid = true; P.tot = 0; P.n = ctx->in_n[slot];
    hipError_t e = hipStreamWaitEvent(ctx->stream, ctx->ev_in[slot], 0);
    P.st = e == hipSuccess ? filter_dev(ctx, ctx->in_reads[slot].as<u8>(), ctx->in_off[slot].as<u64>(), ctx->in_n[slot], nullptr, ctx->h_off[slot].as<u64>()) : LNR_ERR_HIP;
    P.err = ctx->err;
    P.stats = ctx->stats;
    if (P.st == LNR_OK) {
        P.tot = ctx->last_ncords;
        if (ctx->h_cord_off.size() != (size_t)P.n + 1) ctx->h_cord_off.assign((size_t)P.n + 1, 0);
        P.coff = ctx->h_cord_off;
        P.d_str = ctx->r_str.p; P.d_end = ctx->r_end.p;
    }
    return P.st;
}
}  // namespace
lnr_status lnr_filter_wait(lnr_ctx *ctx, lnr_cords *out) {
    if (!ctx || !out) return LNR_ERR_ARG;
    DevGuard dg_(ctx->device);
    if (!ctx->pre.valid) {
        if (ctx->in_count == 0) { ctx->err = "no batch in flight"; return LNR_ERR_ARG; }
        compute_submitted(ctx);
    }
    // the batch to hand out: its download starts now, on the download stream ...
    lnr_ctx::Pre P = std::move(ctx->pre);
    ctx->pre = lnr_ctx::Pre();
    if (P.st != LNR_OK) { ctx->err = P.err; return P.st; }
    const int rs = ctx->res_slot;
    ctx->res_slot ^= 1;
    PinBuf &hs_ = ctx->h_cords_str2[rs], &he_ = ctx->h_cords_end2[rs];
    if (!hs_.ensure(std::max<u64>(P.tot * 8, 16)) || !he_.ensure(std::max<u64>(P.tot * 8, 16))) { ctx->err = "pinned host allocation failed"; return LNR_ERR_NOMEM; }
    ctx->h_cord_off2[rs].swap(P.coff);
    if (P.tot) {
        HIPCK(hipEventRecord(ctx->ev_done, ctx->stream));
        HIPCK(hipStreamWaitEvent(ctx->s_down, ctx->ev_done, 0));
        HIPCK(hipMemcpyAsync(hs_.p, P.d_str, P.tot * 8, hipMemcpyDeviceToHost, ctx->s_down));
        HIPCK(hipMemcpyAsync(he_.p, P.d_end, P.tot * 8, hipMemcpyDeviceToHost, ctx->s_down));
    }
    HIPCK(hipEventRecord(ctx->ev_down, ctx->s_down));
    // ... and the next submitted batch is computed meanwhile, into the other set of device result buffers
    if (ctx->in_count > 0) {
        ctx->r_off.swap(ctx->rB_off); ctx->r_str.swap(ctx->rB_str); ctx->r_end.swap(ctx->rB_end);
        compute_submitted(ctx);
    }
    HIPCK(hipEventSynchronize(ctx->ev_down));
    ctx->stats_pub = P.stats;
    out->n_reads = P.n; out->n_cords = P.tot;
    out->cord_off = ctx->h_cord_off2[rs].data(); out->cords_str = hs_.as<u64>(); out->cords_end = he_.as<u64>();
    return LNR_OK;
}
lnr_status lnr_filter_batch(lnr_ctx *ctx, const uint8_t *reads, const uint64_t *off, uint32_t n, lnr_cords *out) {
    if (!ctx || !out) return LNR_ERR_ARG;
    if (ctx->in_count || ctx->pre.valid) { ctx->err = "batches submitted with lnr_filter_submit are still in flight"; return LNR_ERR_ARG; }
    lnr_status s = lnr_filter_submit(ctx, reads, off, n);
    if (s != LNR_OK) return s;
    return lnr_filter_wait(ctx, out);
}

lnr_status lnr_seed_lookup_batch_dev(lnr_ctx *ctx, const uint8_t *d_reads, const uint64_t *d_off, uint32_t n) {
    if (!ctx || !d_off || (n && !d_reads)) return LNR_ERR_ARG;
    DevGuard dg_(ctx->device);
    lnr_status st_ = seed_dev(ctx, d_reads, d_off, n, false);
    ctx->stats_pub = ctx->stats;
    return st_;
}
lnr_status lnr_seed_lookup_batch(lnr_ctx *ctx, const uint8_t *reads, const uint64_t *off, uint32_t n, lnr_anchors *out) {
    if (!ctx || !out) return LNR_ERR_ARG;
    DevGuard dg_(ctx->device);
    if (!ctx->has_index) { ctx->err = "no index"; return LNR_ERR_NO_INDEX; }
    if (ctx->in_count || ctx->pre.valid) { ctx->err = "batches submitted with lnr_filter_submit are still in flight"; return LNR_ERR_ARG; }
    lnr_status s = submit_reads(ctx, 0, reads, off, n);
    if (s != LNR_OK) return s;
    HIPCK(hipStreamWaitEvent(ctx->stream, ctx->ev_in[0], 0));
    if ((s = seed_dev(ctx, ctx->in_reads[0].as<u8>(), ctx->in_off[0].as<u64>(), n, true)) != LNR_OK) return s;
    ctx->stats_pub = ctx->stats;
    out->n_reads = n; out->n_anchors = ctx->h_anchor_off[n];
    out->anchor_off = ctx->h_anchor_off.data(); out->anchors = ctx->h_anchors.data();
    return LNR_OK;
}

#ifdef LNR_PROF
// diagnostic build only: cumulative per-phase cycle sums of k_job's lane 0 (16 counters)
// diagnostic build: timeline of launch `round` (4 x u64 per launch position); returns positions, *n_heavy = size of the heavy prefix
long long lnr_prof_timeline(lnr_ctx *ctx, unsigned round, unsigned long long *out, unsigned long long cap_positions, unsigned *n_heavy) {
    if (!ctx || round >= 4 || !ctx->tl.p) return -1;
    unsigned long long n = ctx->tl_n[round] < cap_positions ? ctx->tl_n[round] : cap_positions;
    if (hipSetDevice(ctx->device) != hipSuccess) return -1;
    if (hipMemcpy(out, ctx->tl.as<unsigned long long>() + (size_t)round * (1u << 20) * 4, n * 32, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (n_heavy) *n_heavy = ctx->tl_nh[round];
    return (long long)n;
}
lnr_status lnr_prof_read(lnr_ctx *ctx, unsigned long long *out16) {
    if (!ctx || !out16 || !ctx->prof.p) return LNR_ERR_ARG;
    HIPCK(hipStreamSynchronize(ctx->stream));
    HIPCK(hipMemcpy(out16, ctx->prof.p, 192 * 8, hipMemcpyDeviceToHost));
    return LNR_OK;
}
#endif

lnr_status lnr_gap_stream(lnr_ctx *ctx, int set, int *state) {
    if (!ctx || set > 1) return LNR_ERR_ARG;
    if (set >= 0 && (ctx->in_count || ctx->pre.valid)) { ctx->err = "lnr_gap_stream: batches are in flight (the next one may have been computed already)"; return LNR_ERR_ARG; }
    if (set >= 0) ctx->gap_ext = set;
    if (state) *state = ctx->gap_ext;
    return LNR_OK;
}

lnr_status lnr_set_gap(lnr_ctx *ctx, uint32_t gap_len, uint32_t dup) {
    if (!ctx || dup > 1) return LNR_ERR_ARG;
    if (ctx->in_count || ctx->pre.valid) { ctx->err = "lnr_set_gap: batches are in flight"; return LNR_ERR_ARG; }
    ctx->opts.gap_len = gap_len; ctx->opts.dup = dup; ctx->gap_ext = 0;
    return LNR_OK;
}

lnr_status lnr_last_stats(const lnr_ctx *ctx, lnr_stats *st) {
    if (!ctx || !st) return LNR_ERR_ARG;
    *st = ctx->stats_pub;
    return LNR_OK;
}

}  // extern "C"
