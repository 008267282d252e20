// TEST DOUBLE of the library's device half (never shipped): the ABI entry points the front-end (linear_amd/csrc/linear_filter_main.cpp) calls, with a
// fake "GPU" so that the front-end's own logic -- option table, several read files, block pipeline on N contexts, output order, the gap stream
// protocol across contexts -- runs on CPU (tests/test_cli_frontend_cpu.py).  The reader and the writer are the product's real host code
// (lnr_reader.cpp / lnr_output.cpp are linked in).  The fake filter makes cords out of a read's bases AND the stream state it meets, and a read
// that starts with 'T' flips the stream to "extended" -- so the output is only right if blocks see the state the file order implies.
#include "../include/linear_amd.h"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

struct lnr_ctx {
    lnr_opts opts; bool has_index = false; int ext = 0;
    struct Sub { const uint8_t *reads; const uint64_t *off; uint32_t n; } q[3];
    int head = 0, count = 0, slot = 0;
    std::vector<uint64_t> coff[2], cs[2], ce[2];
    std::string err;
};
extern "C" {
void lnr_opts_default(lnr_opts *o) { memset(o, 0, sizeof *o); o->device = -1; o->index_type = 1; o->feature_type = 2; o->preset = 1; }
const char *lnr_strerror(lnr_status s) { return s == LNR_OK ? "ok" : "stub error"; }
const char *lnr_last_error(const lnr_ctx *c) { return c ? c->err.c_str() : ""; }
lnr_status lnr_create(const lnr_opts *o, lnr_ctx **out) { *out = new lnr_ctx(); (*out)->opts = *o; return LNR_OK; }
void lnr_destroy(lnr_ctx *c) { delete c; }
lnr_status lnr_index_build(lnr_ctx *c, const uint8_t *const *, const uint64_t *, uint32_t, uint32_t) { c->has_index = true; return LNR_OK; }
lnr_status lnr_index_broadcast(lnr_ctx *const *cs, uint32_t n, uint32_t root, double *sec) { if (!cs[root]->has_index) return LNR_ERR_NO_INDEX; for (uint32_t i = 0; i < n; i++) cs[i]->has_index = true; if (sec) *sec = 0; return LNR_OK; }
lnr_status lnr_gap_stream(lnr_ctx *c, int set, int *state) { if (set >= 0 && c->count) { c->err = "in flight"; return LNR_ERR_ARG; } if (set >= 0) c->ext = set; if (state) *state = c->ext; return LNR_OK; }
void *lnr_host_alloc(size_t b) { return malloc(b ? b : 16); }
void lnr_host_free(void *p) { free(p); }
lnr_status lnr_filter_submit(lnr_ctx *c, const uint8_t *reads, const uint64_t *off, uint32_t n) {
    if (!c->has_index) return LNR_ERR_NO_INDEX;
    if (c->count >= 3) return LNR_ERR_ARG;
    c->q[(c->head + c->count) % 3] = {reads, off, n};
    c->count++;
    return LNR_OK;
}
lnr_status lnr_filter_wait(lnr_ctx *c, lnr_cords *out) {
    if (!c->count) return LNR_ERR_ARG;
    lnr_ctx::Sub s = c->q[c->head];
    c->head = (c->head + 1) % 3; c->count--;
    int rs = c->slot; c->slot ^= 1;
    std::vector<uint64_t> &coff = c->coff[rs], &cs = c->cs[rs], &ce = c->ce[rs];
    coff.assign(1, 0); cs.clear(); ce.clear();
    for (uint32_t i = 0; i < s.n; i++) {
        const uint8_t *r = s.reads + s.off[i];
        uint64_t L = s.off[i + 1] - s.off[i], h = 1469598103934665603ULL;
        for (uint64_t k = 0; k < L; k++) h = (h ^ r[k]) * 1099511628211ULL;
        if (L > 20) {
            uint64_t k = 1 + h % 3;
            cs.push_back(0x9000000000000000ULL); ce.push_back(0x9000000000000000ULL + ((96ULL << 20) | 96));
            for (uint64_t j = 0; j < k; j++) {
                uint64_t x = 500 + 120 * j + (c->opts.gap_len && c->ext ? 7 : 0) + h % 50, y = 100 * j;
                uint64_t v = (x << 20) | y | (j + 1 == k ? 1ULL << 60 : 0);
                cs.push_back(v); ce.push_back(v + ((96ULL << 20) | 96));
            }
        }
        coff.push_back(cs.size());
        if (c->opts.gap_len && L && r[0] == 3) c->ext = 1;                 // this read "extends": every later read of the stream sees it
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(1 + (int)(s.n % 3)));
    out->n_reads = s.n; out->n_cords = cs.size(); out->cord_off = coff.data(); out->cords_str = cs.data(); out->cords_end = ce.data();
    return LNR_OK;
}
}
