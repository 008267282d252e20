/* linear_amd.h -- C ABI of the MI355X-native `linear filter` hot path.
 *
 * Drop-in boundary for the compute path of xp3i4/linear's `Mapper` (reference paths are
 * relative to the reference root):
 *
 *   reference entry point                                     replaced by
 *   --------------------------------------------------------  --------------------------
 *   createFeatures(genomes, f2, type, T)  pmpfinder.h:196-197  \
 *   Mapper::createIndex -> createIndexDynamic(seqs, index,      } lnr_index_build
 *       gstr, gend, threads, efficient)   index_util.h:297-302 /
 *   body of the `for j` loop in Mapper::p_calRecords            lnr_filter_batch
 *       (mapper.cpp:438-462): _compltRvseStr + createFeatures
 *       (read) x2 + apxMap(...)           pmpfinder.h:213-225
 *   getDIndexMatchAll (stage a7)          pmpfinder.cpp:1856    lnr_seed_lookup_batch
 *
 * Sequences are SeqAn `Dna5` ordinals, one byte per base (A,C,G,T,N = 0..4): exactly the
 * storage of `String<Dna5>` (begin pointer + length), so `&read[0]` / `length(read)` bind
 * directly.  Results are the reference's 64-bit cord words
 * (main[63] recd[62] strand[61] blockEnd[60] id[50..59] x[20..49] y[0..19], cords.h:24-39),
 * `cords_str[j]` and `cords_end[j]` of every read in CSR form.
 *
 * Plain C: pointers and sizes only.  No exceptions cross this boundary; every call returns
 * LNR_OK (0) or a negative lnr_status.  A context is single-threaded and owns one GPU; use
 * one context per GPU / per process (reads shard across contexts, the index is identical).
 * The library has no CPU execution path: without a HIP device lnr_create fails with
 * LNR_ERR_NO_DEVICE.
 */
#ifndef LINEAR_AMD_H
#define LINEAR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lnr_ctx lnr_ctx;

typedef enum lnr_status {
    LNR_OK = 0,
    LNR_ERR_ARG = -1,         /* bad argument (null pointer, offsets not monotone, >= 1024 sequences, ...).  Base ordinals above 4 are not an
                                 error: they are read as N (4), in reads and in reference sequences alike */
    LNR_ERR_NO_DEVICE = -2,   /* no usable HIP device */
    LNR_ERR_HIP = -3,         /* a HIP runtime call failed; see lnr_last_error */
    LNR_ERR_NOMEM = -4,       /* device or host allocation failed */
    LNR_ERR_NO_INDEX = -5,    /* filter/seed call before lnr_index_build / lnr_index_adopt */
    LNR_ERR_LIMIT = -6,       /* input exceeds a format limit (read >= 2^20, sequence >= 2^30 - 2^20; cords.cpp:13-15) */
    LNR_ERR_UNSUPPORTED = -7, /* option outside this build (index_type not 1 or 2, feature_type != 2, dup > 1; -i 2 on a reference with fewer
                                 than three repeated minimizers) */
    LNR_ERR_INTERNAL = -8     /* device-side capacity overflow that retries could not resolve */
} lnr_status;

/* Options = the subset of the reference's `Options` (base.cpp:26-54) that reaches this path. */
typedef struct lnr_opts {
    int32_t device;            /* HIP device ordinal; -1 = current device */
    uint32_t index_type;       /* -i : 1 = DIndex (reference default), 2 = HIndex (index_util.cpp:2593-2610: shape 17/9, one sample per 8 bases) */
    uint32_t feature_type;     /* -f : 2 = 2-mer/48 window features (reference default) */
    uint32_t preset;           /* -p : 1 (reference default: chain stop ratio 0) or 2 (the same computation; only the writer's CIGAR thresholds differ,
                                  lnr_writer_set_preset).  -p 0 (stop ratio 0.7 in the anchor traceback) is not built: LNR_ERR_UNSUPPORTED */
    uint32_t gap_len;          /* -g : 0 = apxMap only; > 0 = the cords go through the gap re-mapper (mapGaps + reformCords, gap.cpp:407-576) with this
                                  minimum gap length, mapped as the reference does: 1 -> 50, 2..9 -> 10 (mapper.cpp:207-231) */
    uint32_t dup;              /* -dup : 0 | 1, the duplication add-on of the gap re-mapper (gap.cpp:303-362) */
    uint64_t scratch_budget;   /* max bytes of per-read device scratch in flight (0 = default 64 GiB of the 288 GB) */
} lnr_opts;

typedef struct lnr_index_info {
    uint32_t nseq;
    uint32_t layout_threads;   /* the reference's -t the index layout reproduces */
    uint64_t genome_bytes;     /* padded device copy of the sequences */
    uint64_t dir_len;          /* int32 entries (4^13 + 1); -i 2: 4^9 + 1 entries of a derived table (head of the block of X, -1 = none) */
    uint64_t hs_len;           /* uint64 entries; -i 2: the words of ysa, the reference's block array (the parity surface of that index) */
    uint64_t f2_len;           /* 16-byte feature entries over all sequences */
    uint64_t n_samples;        /* genome minimizer samples examined */
    double build_ms;           /* device time of the last build */
} lnr_index_info;

/* CSR result of a batch.  Host arrays owned by the context, valid until the next
 * lnr_filter_batch / lnr_seed_lookup_batch on the same context or lnr_destroy. */
typedef struct lnr_cords {
    uint32_t n_reads;
    uint64_t n_cords;
    const uint64_t *cord_off;   /* n_reads + 1 */
    const uint64_t *cords_str;  /* n_cords */
    const uint64_t *cords_end;  /* n_cords */
} lnr_cords;

/* Same, device resident (for callers that keep results on the GPU, and for benchmarking). */
typedef struct lnr_cords_dev {
    uint32_t n_reads;
    uint64_t n_cords;
    const uint64_t *d_cord_off;
    const uint64_t *d_cords_str;
    const uint64_t *d_cords_end;
} lnr_cords_dev;

typedef struct lnr_anchors {    /* stage a7 output: raw anchors per read, each list led by the dummy 0 */
    uint32_t n_reads;
    uint64_t n_anchors;
    const uint64_t *anchor_off; /* n_reads + 1 */
    const uint64_t *anchors;
} lnr_anchors;

/* Counters and device timings of the last batch call (deterministic functions of index + reads,
 * except the *_ms fields).  seed_bytes is SURVEY.md 8(d)'s algorithmic byte count
 * sum(ceil(L/4)) + lookups*8 + bucket_entries*8 + anchors*8. */
typedef struct lnr_stats {
    uint64_t reads, bases, jobs, samples, lookups, bucket_entries, anchors, remap_reads, cords;
    uint64_t seed_bytes;
    double prep_ms, seed_count_ms, seed_gather_ms, job_ms, tail_ms, total_ms;
    uint32_t seed_count_launches, seed_gather_launches, job_launches;
    uint32_t gap_second_pass;  /* reads the gap re-mapper ran a second time (a team of waves per read: out of the first launch's arena or cord slot) */
    double gap_ms;             /* device time of the gap re-mapper (-g > 0) */
} lnr_stats;

void lnr_opts_default(lnr_opts *o);
lnr_status lnr_create(const lnr_opts *opts, lnr_ctx **out);
void lnr_destroy(lnr_ctx *ctx);
const char *lnr_strerror(lnr_status s);
const char *lnr_last_error(const lnr_ctx *ctx);   /* detail of the last failure on this context */

/* Index + genome features from the reference sequences.  seq[i] may point to host memory (`&genome[i][0]` of the
 * StringSet<String<Dna5>>) or to device memory (a genome already resident in HBM); the pointer array itself is a host array.
 * layout_threads = the reference's -t whose DIndex layout is to be reproduced (index content depends on it:
 * index_util.cpp:1652-1700). */
lnr_status lnr_index_build(lnr_ctx *ctx, const uint8_t *const *seq, const uint64_t *len, uint32_t nseq, uint32_t layout_threads);
lnr_status lnr_index_info_get(const lnr_ctx *ctx, lnr_index_info *info);
/* Copy the index to host arrays (any pointer may be NULL).  f2 is written as 3 x int32 per entry. */
lnr_status lnr_index_export(lnr_ctx *ctx, int32_t *dir, uint64_t *hs, int32_t *f2, uint64_t *f2_off /* nseq+1 */);

/* Multi-GPU: the packed index is built on one rank and broadcast (RCCL) to the others.
 * lnr_index_blob_* expose the device buffers that make up the index so the host framework can
 * broadcast them in place: rank 0 calls lnr_index_build, every rank exchanges lnr_index_info +
 * the sequence lengths, the receiving ranks call lnr_index_alloc, then all ranks broadcast each
 * blob (ptr, bytes) and the receivers finish with lnr_index_adopt. */
#define LNR_INDEX_BLOBS 4       /* genome bytes, dir, hs, f2 */
lnr_status lnr_index_alloc(lnr_ctx *ctx, const lnr_index_info *info, const uint64_t *seq_len /* nseq */);
lnr_status lnr_index_blob(lnr_ctx *ctx, uint32_t which, void **d_ptr, uint64_t *bytes);
lnr_status lnr_index_adopt(lnr_ctx *ctx);
/* One process driving several GPUs (the C++ front-end: one host thread + one context per GPU): moves the index of ctxs[root] into
 * every other context (created with the same options, no index yet) through lnr_index_alloc / _blob / _adopt.  Contexts on distinct devices:
 * ncclBroadcast of the four device buffers in place over RCCL / xGMI (librccl is loaded on first use; nothing else in the library needs it);
 * a context that shares its device with another one gets device-to-device copies.  *seconds (optional): wall time of the exchange, the
 * receivers' derived tables included.  What a front-end compares it with: lnr_index_build on every context ("every GPU builds its own"). */
lnr_status lnr_index_broadcast(lnr_ctx *const *ctxs, uint32_t n, uint32_t root, double *seconds);

/* The hot path.  reads_concat = bases of all reads back to back, off[n+1] = start offsets.
 * Host-buffer form (copies in and out over PCIe): */
lnr_status lnr_filter_batch(lnr_ctx *ctx, const uint8_t *reads_concat, const uint64_t *off, uint32_t n, lnr_cords *out);
/* The same in two halves, so that one context overlaps transfers and compute: lnr_filter_submit starts the upload of a batch on a copy
 * stream and returns; lnr_filter_wait hands out the cords of the oldest submitted batch.  Up to THREE batches may be in flight, and with
 * the pattern   submit(0); submit(1); loop { submit(k + 2); wait(k); }   the GPU never idles: lnr_filter_wait(k) starts the download of
 * batch k's cords (batch k was computed during the previous wait) and computes batch k + 1 while they travel; the upload of batch k + 2
 * runs under those kernels.  (submit(k + 1); wait(k) works as well: upload overlapped, download not.)  The read buffer must stay untouched
 * until the lnr_filter_wait that RETURNS its batch has returned; the host arrays of a result stay valid until the SECOND next result of the
 * context (two result slots taken in turn: another thread may format batch k while the context runs on).  lnr_last_stats reports the batch
 * handed out last.  lnr_gap_stream(set >= 0) and lnr_set_gap need an idle context (nothing in flight).
 * A read buffer in pinned host memory (lnr_host_alloc, or the caller's own hipHostMalloc / hipHostRegister) is uploaded by one
 * DMA at link rate; a pageable one goes through the context's pinned staging buffers first. */
lnr_status lnr_filter_submit(lnr_ctx *ctx, const uint8_t *reads_concat, const uint64_t *off, uint32_t n);
lnr_status lnr_filter_wait(lnr_ctx *ctx, lnr_cords *out);
void *lnr_host_alloc(size_t bytes);   /* pinned host memory for read blocks (NULL on failure) */
void lnr_host_free(void *p);
/* Device-buffer form: d_reads_concat / d_off already in HBM; results stay in HBM.  Streams: the library works on private
 * streams.  The device inputs must be complete before the call (the caller synchronises the stream that produced them) and
 * the results are complete when the call returns.  Every entry point leaves the caller's current HIP device as it found it. */
lnr_status lnr_filter_batch_dev(lnr_ctx *ctx, const uint8_t *d_reads_concat, const uint64_t *d_off, uint32_t n, lnr_cords_dev *out);
/* apx_gaps of the last filter call -- apxMap's second output (include/pmpfinder.h:213-225), the input of the reference's gap
 * re-mapper mapGaps (src/mapper.cpp:448-453; not part of this library): per read, the uncovered stretches of the read as pairs of
 * cord words (first, second), as gather_gaps_y_ leaves them before the re-map loop (src/pmpfinder.cpp:2744).  CSR over the reads of
 * the batch; host arrays owned by the context, valid until the next call. */
typedef struct lnr_gaps {
    uint32_t n_reads;
    uint64_t n_gaps;
    const uint64_t *gap_off;    /* n_reads + 1 */
    const uint64_t *gaps;       /* 2 * n_gaps: first, second */
} lnr_gaps;
lnr_status lnr_last_gaps(lnr_ctx *ctx, lnr_gaps *out);
/* Copy the last device result to the context's host arrays. */
lnr_status lnr_cords_to_host(lnr_ctx *ctx, lnr_cords *out);

/* Stage a7 only (seed lookup on [0, L) with sampling step 15), for parity tests and the roofline measurement. */
lnr_status lnr_seed_lookup_batch(lnr_ctx *ctx, const uint8_t *reads_concat, const uint64_t *off, uint32_t n, lnr_anchors *out);
lnr_status lnr_seed_lookup_batch_dev(lnr_ctx *ctx, const uint8_t *d_reads_concat, const uint64_t *d_off, uint32_t n);

lnr_status lnr_last_stats(const lnr_ctx *ctx, lnr_stats *st);

/* The read stream's state of the gap re-mapper (gap_len > 0).  The reference keeps ONE GapParms per calculator thread for the whole run
 * (Mapper::loadOptions mapper.cpp:233-237, used in p_calRecords :447) and mapExtend / mapExtends leave it modified (gap_util.cpp:4046-4071,
 * 4088-4119).  Of the fields left behind only thd_cts_major_limit is read before it is written again (chainTiles :1188 under mapGeneric):
 * 1 until the first mapExtend / mapExtends of the thread's stream, 3 for every read after it.  A context therefore IS one read stream:
 * batches are taken in submission order, reads in batch order, exactly as `linear filter -t 1` meets them (with more threads the reference
 * itself is not reproducible: what a read sees depends on which reads its thread met before).  The state lives in the context across
 * batches.  set < 0: query only; set = 0: start a new stream (a new read file); set = 1: the stream has extended already -- what a
 * front-end that deals the batches of one file over several contexts / GPUs sets on the others once the first context reports 1.
 * *state (optional) receives the state after the call. */
lnr_status lnr_gap_stream(lnr_ctx *ctx, int set, int *state);
/* Change -g / -dup of an existing context (the index does not depend on them); starts a new read stream. */
lnr_status lnr_set_gap(lnr_ctx *ctx, uint32_t gap_len, uint32_t dup);

/* Input side (host code; replaces, for this path, the fetcher's SeqAn readRecords of src/parallel_io.cpp:433-485): FASTA or
 * FASTQ records, plain or gzip, decoded into the layout lnr_filter_batch / lnr_filter_submit take.  Characters convert as SeqAn's
 * char -> Dna5 table does (A/a 0, C/c 1, G/g 2, T/t/U/u 3, anything else N = 4).  lnr_reader_next fills dst (e.g. a block from
 * lnr_host_alloc) with up to max_reads records and at most dst_cap bases, writes off[0 .. *n_out]; *n_out == 0 at end of file.
 * A record that no longer fits is delivered first by the next call.  lnr_reader_ids: header lines of the last block (without the
 * '>' / '@'), '\0'-separated, id_off[k] = start of id k. */
typedef struct lnr_reader lnr_reader;
typedef struct lnr_writer lnr_writer;
lnr_status lnr_reader_open(const char *path, lnr_reader **out);
lnr_status lnr_reader_next(lnr_reader *r, uint8_t *dst, uint64_t dst_cap, uint64_t *off, uint32_t max_reads, uint32_t *n_out);
lnr_status lnr_reader_ids(const lnr_reader *r, const char **ids, const uint64_t **id_off);
const char *lnr_reader_error(const lnr_reader *r);
void lnr_reader_close(lnr_reader *r);

/* Output side (host threads; replaces, for this path, the calculator's tail cords2BamLink + fillBamRecords, src/mapper.cpp:463-470,
 * src/f_io.cpp:758-1011, src/align_util.cpp:301-343,452-744, and the printer's writeSam / print_cords_apf, src/f_io.cpp:100-207,
 * 313-412): the cords of a batch as SAM records (what = 1) or APF text (what = 2), byte for byte what the reference prints for
 * the same cords with -g 0 (MAPQ 255, flag 16 / 2048, SA:Z of the read's other lines; APF: a blank line before every '@' record
 * of a read that is not the first of the call, as the reference does per block).  read_ids = header lines, '\0'-separated,
 * id_off[k] = start of id k (the layout lnr_reader_ids returns); read_len[k] = bases of read k.  The text is owned by the writer
 * and valid until its next call.  lnr_writer_sam_header: @SQ per sequence, @RG ID: SM:, @PG ID:M1-3 PN:Linear CL:<command_line>. */
lnr_status lnr_writer_create(const char *const *genome_ids, const uint64_t *genome_len, uint32_t nseq, lnr_writer **out);
lnr_status lnr_writer_format(lnr_writer *w, const lnr_cords *cords, const uint64_t *read_len, const char *read_ids, const uint64_t *id_off,
                             int what, uint32_t threads, const char **text, uint64_t *size);
lnr_status lnr_writer_sam_header(lnr_writer *w, const char *command_line, const char **text, uint64_t *size);
/* -p (mapper.cpp:173-196): preset 1 splits large diagonal shifts of a CIGAR (thd_DI 80, thd_X 200); presets 0 and 2 leave the reference's
 * defaults (2^60 - 1: never).  -rg / -sn: the @RG line's ID / SM (mapper.cpp:288-324); both empty by default. */
lnr_status lnr_writer_set_preset(lnr_writer *w, uint32_t preset);
lnr_status lnr_writer_set_read_group(lnr_writer *w, const char *read_group, const char *sample_name);
void lnr_writer_destroy(lnr_writer *w);

#ifdef __cplusplus
}
#endif
#endif
