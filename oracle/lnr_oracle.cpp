// lnr_oracle.cpp -- CPU restatement of the `linear filter` hot path.
//
// THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may load it.  The product (linear_amd/) never
// links, imports or calls anything in this directory.
//
// What it restates (reference = xp3i4/linear @ v1, paths relative to the
// reference root):
//   gapped-minimizer shape   src/shape_extend.cpp:86-116,173-184,245-348
//   cord/anchor bit words    src/cords.cpp:21-37,81-90,135-141,159-198,306-331
//   DIndex build             src/index_util.cpp:1628-1803 (params :2551-2555)
//   2-mer window features    src/pmpfinder.cpp:493-652
//   seed lookup              src/pmpfinder.cpp:1856-1913, src/index_util.cpp:1509-1520
//   anchor filter            src/pmpfinder.cpp:1979-2091
//   anchor chaining DP       src/cluster_util.cpp:53-462, src/pmpfinder.cpp:2448-2481
//   hit blocks               src/pmpfinder.cpp:1484-1530,2366-2446,2506-2555, src/cluster_util.cpp:469-732
//   window extension         src/pmpfinder.cpp:680-722,883-945,1079-1178,1309-1469
//   cord blocks              src/pmpfinder.cpp:1537-1767, src/cluster_util.cpp:774-1102
//   apxMap_/apxMap           src/pmpfinder.cpp:2632-2804
//
// Parity pin: every function here is checked against the reference itself
// (oracle/_ref, built from the reference's own sources by oracle/Makefile) and
// against the golden vectors under tests/golden/ that oracle/_ref produced.
//
// Deliberate pins of reference UB (see DESIGN.md "pinned undefined behaviour"):
//   * bytes past the end of a sequence read as 0 ('A'); the reference reads
//     uninitialised allocator slack there (shape_extend.cpp:294-296 reads up to
//     3 bases past a read, pmpfinder.cpp:637-647 one base past a genome).
//   * __builtin_ctzl(0) in the Y filter is treated as "match" (pmpfinder.cpp:1893).
//
// Sorting: the reference's tie-sensitive std::sort calls are restated with
// std::sort from the same libstdc++, i.e. the identical algorithm.

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <cmath>
#include <vector>
#include <algorithm>
#include <utility>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace orc {

typedef uint64_t u64;
typedef int64_t i64;
typedef std::pair<u64, u64> UPair;

static const size_t SEQ_PAD = 64;

// ---------------------------------------------------------------- cords ----
// src/cords.cpp:8-15,21-37
static const u64 const_anchor_zero = 1ULL << 20;
static const u64 MAX_CORD_ID = (1ULL << 10) - 1;
static const u64 MAX_CORD_X = (1ULL << 30) - 1;
static const u64 F_END = 1ULL << 60;
static const u64 F_STRAND = 1ULL << 61;
static const u64 F_RECD = 1ULL << 62;
static const u64 F_MAIN = 1ULL << 63;
static const u64 VALUE_MASK = (1ULL << 60) - 1;
static const u64 VALUE_MASK_DSTR = VALUE_MASK | F_STRAND;

static inline u64 get_cord_x(u64 v) { return (v >> 20) & ((1ULL << 30) - 1); }      // cords.cpp:159
static inline u64 get_cord_y(u64 v) { return v & 0xfffffULL; }                       // cords.cpp:56,163
static inline u64 get_cord_strand(u64 v) { return (v >> 61) & 1ULL; }                // cords.cpp:116
static inline u64 get_cord_id(u64 v) { return (v >> 50) & ((1ULL << 10) - 1); }      // cords.cpp:166
static inline u64 getCordX40(u64 v) { return (v >> 20) & 0xffffffffffULL; }          // Cord::getCordX cords.cpp:50
static inline u64 create_id_x(u64 id, u64 x) { return (id << 30) + x; }              // cords.cpp:180
static inline u64 createCord(u64 x, u64 y, u64 strand) { return (x << 20) + y + (strand << 61); } // cords.cpp:61
static inline u64 create_cord(u64 id, u64 x, u64 y, u64 s) { return createCord(create_id_x(id, x), y, s); } // :195
static inline u64 shift_cord(u64 val, i64 x, i64 y) {                                // cords.cpp:135
    if (x < 0) return val - ((u64)(-x) << 20) + (u64)y;
    return val + ((u64)x << 20) + (u64)y;
}
static inline bool is_block_end(u64 v) { return (v & F_END) != 0; }
static inline void set_block_end(u64 &v) { v |= F_END; }
static inline void unset_block_end(u64 &v) { v &= ~F_END; }
static inline u64 hit2Cord_dstr(u64 hit) {                                           // cords.cpp:81-90
    u64 c = (hit + ((hit & 0xfffffULL) << 20) - (const_anchor_zero << 20)) & VALUE_MASK_DSTR;
    c &= ~(1ULL << 62);
    return c;
}
static inline u64 getAnchorX(u64 a) { return get_cord_x(hit2Cord_dstr(a)); }         // cords.cpp:463
static inline int isCordsConsecutive_(u64 c1, u64 c2, u64 thd) {                     // cords.cpp:306
    u64 x1 = get_cord_x(c1), x2 = get_cord_x(c2), y1 = get_cord_y(c1), y2 = get_cord_y(c2);
    return !get_cord_strand(c1 ^ c2) && x1 <= x2 && y1 <= y2 && x2 - x1 < thd && y2 - y1 < thd;
}
static inline bool _isRangeOverLap(u64 x11, u64 x12, u64 x21, u64 x22) {             // cords.cpp:450
    return std::max(x11, x21) < std::min(x12, x22);
}
static inline bool _isCordyOverLap(u64 c11, u64 c12, u64 c21, u64 c22, u64 read_len) { // cords.cpp:455
    return get_cord_strand(c11 ^ c21)
               ? _isRangeOverLap(get_cord_y(c11), get_cord_y(c12), read_len - 1 - get_cord_y(c21), read_len - 1 - get_cord_y(c22))
               : _isRangeOverLap(get_cord_y(c11), get_cord_y(c12), get_cord_y(c21), get_cord_y(c22));
}
static inline UPair getUPForwardy(UPair se, u64 read_len) {                          // cords.cpp:469
    if (get_cord_strand(se.first))
        return UPair(read_len - get_cord_y(se.second) - 1, read_len - get_cord_y(se.first) - 1);
    return UPair(get_cord_y(se.first), get_cord_y(se.second));
}
static void initCords(std::vector<u64> &c) { c.clear(); c.push_back(0); set_block_end(c[0]); } // cords.cpp:325

// ---------------------------------------------------------------- shape ----
struct Shape {                                                                       // include/shape_extend.h:7-22
    unsigned span = 21, weight = 13;
    u64 hValue = 0, crhValue = 0, XValue = 0, YValue = 0, strand = 0;
    int leftChar = 0, x = 0;
};
static inline u64 getMask(unsigned bit) { return (1ULL << bit) - 1; }

static u64 hashInit(Shape &me, const uint8_t *it) {                                  // shape_extend.cpp:86-116
    me.leftChar = 0; me.hValue = 0; me.crhValue = 0;
    me.x = me.leftChar - 3;
    u64 k = 0, count = 0;
    while (count < me.span) {
        if (it[k + count] == 4) { k += count + 1; count = 0; }
        else count++;
    }
    unsigned bit = 2;
    for (unsigned i = 0; i < me.span - 1; ++i) {
        u64 val = it[k + i];
        me.x += (int(val) << 1) - 3;
        me.hValue = (me.hValue << 2) + val;
        me.crhValue += ((3ULL - val) << bit);
        bit += 2;
    }
    return k;
}
static inline void hashNexth(Shape &me, const uint8_t *it) {                         // shape_extend.cpp:173-184
    u64 mask = getMask((me.span << 1) - 2);
    int v2 = it[me.span - 1];
    me.hValue = ((me.hValue & mask) << 2) + (u64)v2;
    me.crhValue = ((me.crhValue >> 2) & mask) + ((3ULL - (u64)(i64)v2) << ((me.span << 1) - 2));
    me.x += (v2 - me.leftChar) * 2;
    me.leftChar = it[0];
}
static inline void hashNextX(Shape &me, const uint8_t *it) {                         // shape_extend.cpp:245-271,282-348
    u64 t = 0, v2, v1;
    unsigned span = me.span << 1, weight = me.weight << 1;
    if (me.x > 0) { v2 = me.hValue; me.strand = 0; }
    else { v2 = me.crhValue; me.strand = 1; }
    me.XValue = getMask(me.span << 1);
    for (unsigned k = 64 - span; k <= 64 - weight; k += 2) {
        v1 = v2 << k >> (64 - weight);
        if (me.XValue > v1) { me.XValue = v1; t = k; }
    }
    me.YValue = 0;
    i64 n = 4;
    if (me.x > 0) {
        i64 d_it = (i64)(t >> 1) + me.span + me.weight - 32;
        for (i64 i = d_it; i < d_it + n; i++) {
            i64 val = it[i];
            me.YValue = val > 3 ? (me.YValue << 2) : (me.YValue << 2) + val;
        }
    } else {
        i64 d_it = -(i64)(t >> 1) - (i64)me.weight + 31;
        for (i64 i = d_it; i > d_it - n; i--) {
            i64 val = 3 - (i64)it[i];
            if (val < 0) me.YValue = me.YValue << 2;
            else me.YValue = (me.YValue << 2) + val;
        }
    }
}

// --------------------------------------------------------------- DIndex ----
struct DIndex {
    std::vector<int32_t> dir;
    std::vector<u64> hs;
    u64 fill_mismatch = 0;   // buckets whose pass-2 fill count != pass-1 count (must stay 0)
};

// src/index_util.cpp:1628-1803 with thd_min_step 8, thd_max_step 10, thd_omit_block 400, span 21
static void createDIndex(const std::vector<const uint8_t *> &seqs, const std::vector<u64> &lens, DIndex &index, unsigned threads) {
    const i64 thd_min_step = 8, thd_max_step = 10, thd_omit_block = 400;
    Shape t_shape;
    const size_t full = ((size_t)1 << t_shape.weight << t_shape.weight) + 1;
    std::vector<int32_t> &dir = index.dir;
    std::vector<u64> &hs = index.hs;
    dir.assign(full, 0);
    // The reference runs the `threads` chunks of a sequence as an OpenMP team and counts / fills with atomics
    // (index_util.cpp:1667-1700, 1743-1786: atomicInc on dir, atomic_inc_cord_y on the slot head), so the order in which
    // chunks run does not reach the result.  Here every (sequence, chunk) pair is one work item of a host-wide team
    // (build_threads, independent of the layout parameter `threads`): same counts, same slots after the per-bucket sort.
    struct Item { size_t seq; i64 t_str, t_end; };
    std::vector<Item> items;
    for (size_t i = 0; i < seqs.size(); i++) {
        std::vector<i64> t_blocks;
        for (unsigned j = 0; j < threads; j++) t_blocks.push_back((i64)(lens[i] / threads * j));
        t_blocks.push_back((i64)lens[i] - (i64)t_shape.span);
        for (unsigned t_id = 0; t_id < threads; t_id++) {
            i64 t_str = t_blocks[t_id] + t_shape.span;
            i64 t_end = t_blocks[t_id + 1] - t_shape.span;
            if (t_str >= t_end) continue;   // reference would still call hashInit; no effect
            items.push_back({i, t_str, t_end});
        }
    }
    std::sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.t_end - a.t_str > b.t_end - b.t_str; });   // longest first
    const int build_threads = omp_get_max_threads();
#pragma omp parallel for num_threads(build_threads) schedule(dynamic, 1)
    for (size_t it = 0; it < items.size(); it++) {
        const uint8_t *seq = seqs[items[it].seq];
        i64 t_str = items[it].t_str, t_end = items[it].t_end;
        i64 last_j = t_str - 1, count = 0;
        u64 preVal = ~0ULL;
        Shape shape = t_shape;
        hashInit(shape, seq + t_str);
        for (i64 j = t_str; j < t_end; j++) {
            hashNexth(shape, seq + j);
            if (++count > thd_min_step) {
                hashNextX(shape, seq + j);
                if (preVal != shape.XValue || j - last_j > thd_max_step) {
                    __atomic_fetch_add(&dir[shape.XValue], 1, __ATOMIC_RELAXED);
                    preVal = shape.XValue;
                    last_j = j;
                }
                count = 0;
            }
        }
    }
    i64 sum = 0;
    for (size_t i = 0; i < dir.size(); i++) {
        if (dir[i] > thd_omit_block) dir[i] = 0;
        sum += dir[i];
        dir[i] = (int32_t)(sum - dir[i]);
    }
    u64 EmptyVal = create_cord(seqs.size(), 0, 0, 0);
    hs.assign((size_t)sum, EmptyVal);
#pragma omp parallel for num_threads(build_threads) schedule(dynamic, 1)
    for (size_t it = 0; it < items.size(); it++) {
        size_t i = items[it].seq;
        const uint8_t *seq = seqs[i];
        i64 t_str = items[it].t_str, t_end = items[it].t_end;
        i64 last_j = t_str - 1, count = 0;
        u64 preVal = ~0ULL;
        Shape shape = t_shape;
        hashInit(shape, seq + t_str);
        for (i64 j = t_str; j < t_end; j++) {
            hashNexth(shape, seq + j);
            if (++count > thd_min_step) {
                hashNextX(shape, seq + j);
                if (preVal != shape.XValue || j - last_j > thd_max_step) {
                    if (dir[shape.XValue + 1] - dir[shape.XValue]) {
                        i64 slot_str = dir[shape.XValue];
                        i64 slot_end = dir[shape.XValue + 1];
                        u64 nv = __atomic_add_fetch(&hs[slot_str], 1, __ATOMIC_RELAXED);   // atomic_inc_cord_y
                        i64 k = slot_end - (i64)(get_cord_y(nv) & ((1ULL << 15) - 1));
                        u64 val = create_cord(i, (u64)j + const_anchor_zero, shape.YValue, shape.strand);
                        // the head slot doubles as the fill counter until its own (last) entry lands: that store must not tear
                        // against another thread's increment, so it is an atomic exchange of the whole word
                        if (k == slot_str) __atomic_store_n(&hs[k], val, __ATOMIC_RELAXED); else hs[k] = val;
                        preVal = shape.XValue;
                        last_j = j;
                    }
                }
                count = 0;
            }
        }
    }
    u64 mism = 0;
#pragma omp parallel for num_threads(build_threads) schedule(static, 1 << 16) reduction(+ : mism)
    for (size_t i = 0; i < dir.size() - 1; i++) {
        if (dir[i + 1] > dir[i]) {
            if (get_cord_id(hs[dir[i]]) >= seqs.size()) mism++;
            std::sort(hs.begin() + dir[i], hs.begin() + dir[i + 1]);
        }
    }
    index.fill_mismatch = mism;
}


// --------------------------------------------------------------- HIndex ----
// `-i 2`.  hs / ysa word formats (index_util.cpp:136-157, 241-299): head = ptr[23] << 40 | X[40], bit 63 clear;
// body = 1 << 63 | Y << 41 | reverse-strand flag << 40 | sequence id << 30 | position.
static const u64 HS_TYPEFLAG = 1ULL << 63, HS_TYPEMASK = HS_TYPEFLAG - 1, HS_MASK40 = (1ULL << 40) - 1, HS_PTRMASK = (1ULL << 23) - 1;
static const u64 HS_BODYYMASK = (1ULL << 20) - 1, HS_BODYCODEFLAG = 1ULL << 40;
static inline u64 hsHead(u64 ptr, u64 x) { return ((ptr << 40) + x) & HS_TYPEMASK; }                      // Hs::setHsHead :245
static inline u64 hsHeadPtr(u64 v) { return (v >> 40) & HS_PTRMASK; }                                      // Hs::getHeadPtr :261
static inline u64 hsHeadX(u64 v) { return v & HS_MASK40; }                                                  // Hs::getHeadX :257
static inline bool hsIsHead(u64 v) { return ((v & HS_TYPEFLAG) ^ HS_TYPEFLAG) != 0; }                      // Hs::isHead :241
static inline u64 hsBody(u64 y, u64 id, u64 pos) { return ((y << 41) | HS_TYPEFLAG) + (id << 30) + pos; }  // Hs::setHsBody :265
static inline u64 hsBodyY(u64 v) { return (v >> 41) & HS_BODYYMASK; }                                      // Hs::getHsBodyY :275
static inline u64 hsBodyS(u64 v) { return v & HS_MASK40; }                                                 // Hs::getHsBodyS :283

static inline void hashNext(Shape &me, const uint8_t *it) {                          // shape_extend.cpp:132-168 (index side of the HIndex)
    u64 v1;
    unsigned t = 0, span = me.span << 1, weight = me.weight << 1;
    u64 v2 = it[me.span - 1];
    u64 mask = getMask(span - 2);
    me.hValue = ((me.hValue & mask) << 2) + v2;
    me.crhValue = ((me.crhValue >> 2) & mask) + ((3ULL - v2) << (span - 2));
    me.XValue = getMask(span);
    me.x += (int)((v2 - (u64)(i64)me.leftChar) << 1);                                 // (v2 - leftChar) << 1 in uint64_t, added to the int
    me.leftChar = it[0];
    if (me.x > 0) { v2 = me.hValue; me.strand = 0; }
    else { v2 = me.crhValue; me.strand = 1; }
    for (unsigned k = 64 - span; k <= 64 - weight; k += 2) {
        v1 = v2 << k >> (64 - weight);
        if (me.XValue > v1) { me.XValue = v1; t = k; }
    }
    me.YValue = (v2 >> (64 - t) << (64 - t - weight)) + (v2 & ((1ULL << (64 - t - weight)) - 1)) + ((u64)t << (span - weight - 1));
}

struct XNode { u64 val1 = 0; unsigned val2 = 0; };
struct HIndex {
    std::vector<u64> ysa;
    std::vector<XNode> xstr; u64 xmask = 0;
    u64 emptyDir = 0;
    unsigned span = 17, weight = 9;                                                  // index_util.cpp:2600,2604 (thd_shape_len 17), shape_extend.cpp:55-59
};
static inline u64 xhash(u64 key) {                                                   // XNodeFunc::hash index_util.cpp:971-982
    key = (~key) + (key << 21); key = key ^ (key >> 24); key = (key + (key << 3)) + (key << 8); key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4); key = key ^ (key >> 28); key = key + (key << 31);
    return key;
}
static void requestXNode(HIndex &ix, u64 xval, unsigned val2, u64 nodeType) {         // requestXNode_noCollision index_util.cpp:1006-1018 (returnType 0)
    u64 h1 = xhash(xval) & ix.xmask, delta = 0;
    while (ix.xstr[h1].val1) { h1 = (h1 + delta + 1) & ix.xmask; delta++; }
    ix.xstr[h1].val1 = (xval << 2) + nodeType;
    ix.xstr[h1].val2 = val2;
}
// XNodeBase::mask2 is (1 << 62) - 1 evaluated in int: undefined; the reference as compiled here (g++ -O2) holds all ones, so the
// comparison below is on the whole word (checked on the library itself: oracle/_ref, symbol _DefaultXNodeBase).
static u64 getXDir(const HIndex &ix, u64 xval, u64 yval) {                            // index_util.cpp:1071-1093
    u64 val = (xval << 2) + 1, delta = 0;
    u64 h1 = xhash(xval) & ix.xmask;
    while (ix.xstr[h1].val1) {
        u64 c = ix.xstr[h1].val1 ^ val;
        if (c == 0) return ix.xstr[h1].val2;
        if (c == 2) { val = (yval << 42) + (xval << 2) + 1; h1 = xhash((yval << 40) + xval) & ix.xmask; delta = 0; }
        else { h1 = (h1 + delta + 1) & ix.xmask; delta++; }
    }
    return ix.emptyDir;
}
static void createHIndex(const std::vector<const uint8_t *> &seqs, const std::vector<u64> &lens, HIndex &ix, unsigned threads) {
    const unsigned thd_step = 8; const u64 thd_blocklimit = 1024; const float alpha = 1.6f;   // index_util.cpp:2599-2603; XString::_fullSize default alpha (def_alpha)
    Shape shape; shape.span = ix.span; shape.weight = ix.weight;
    // ---- __createHsArray (index_util.cpp:719-818): every sequence cut into `threads` chunks, each with its own rolling state
    u64 total = 0; for (u64 l : lens) total += l;
    std::vector<u64> hs(total * 2 / thd_step + 1000 + 64 * (u64)threads * lens.size(), 0);
    u64 hsRealEnd = 0;
    for (u64 j = 0; j < seqs.size(); j++) {
        const uint8_t *seq = seqs[j];
        u64 npos = lens[j] - shape.span + 1;
        u64 size2 = npos / threads;
        std::vector<u64> cnt(threads, 0), hss(threads, 0);
        for (unsigned thd_id = 0; thd_id < threads; thd_id++) {                       // (the reference runs these bodies concurrently; they share nothing)
            Shape tshape = shape;
            u64 preX = ~0ULL; i64 ptr = 0;
            u64 chunk, start;
            if (thd_id < npos - size2 * threads) { chunk = size2 + 1; start = (size2 + 1) * thd_id; }
            else { chunk = size2; start = lens[j] + 1 - tshape.span - size2 * (threads - thd_id); }
            u64 hsStart = hsRealEnd + (start << 1) / thd_step + thd_id * 10;
            hss[thd_id] = hsStart;
            u64 thd_count = 0;
            hashInit(tshape, seq + start);
            for (u64 k = start; k < start + chunk; k++) {
                if (seq[k + tshape.span - 1] == 4) {
                    k += hashInit(tshape, seq + k);
                    if (k > chunk - tshape.span + 1 + start) k = chunk - (chunk + start) % thd_step + thd_step + start;
                }
                hashNext(tshape, seq + k);
                if (k % thd_step == 0) {
                    if (tshape.XValue ^ preX) {
                        hs[hsStart + thd_count - ptr] = hsHead((u64)ptr, preX);
                        hs[hsStart + ++thd_count] = hsBody(tshape.YValue, j, k);
                        if (tshape.strand) hs[hsStart + thd_count] |= HS_BODYCODEFLAG;
                        preX = tshape.XValue;
                        ++thd_count;
                        ptr = 2;
                    }
                }
            }
            hs[hsStart + thd_count - ptr] = hsHead((u64)ptr, tshape.XValue);         // the chunk's last block is filed under the X of its last hashed position
            cnt[thd_id] = thd_count;
        }
        u64 acc = cnt[0];
        for (unsigned t = 1; t < threads; t++) {
            u64 it = hss[t];
            for (u64 k = hsRealEnd + acc; k < hsRealEnd + acc + cnt[t]; k++) hs[k] = hs[it++];
            acc += cnt[t];
        }
        hsRealEnd += acc;
    }
    hs.resize(hsRealEnd + 1);
    hs[hsRealEnd] = hsHead(0, 0);
    // ---- _hsSortX_1 (index_util.cpp:430-560): stable LSD radix sort of the blocks by the low 2 * weight bits of X
    {
        std::vector<u64> order;                                                        // block starts
        for (u64 k = 0; k < hsRealEnd; k += hsHeadPtr(hs[k])) order.push_back(k);
        const u64 xm = (1ULL << (2 * ix.weight)) - 1;
        std::stable_sort(order.begin(), order.end(), [&](u64 a, u64 b) { return (hs[a] & xm) < (hs[b] & xm); });
        std::vector<u64> out(hs.size());
        u64 w = 0;
        for (u64 b : order) { u64 p = hsHeadPtr(hs[b]); for (u64 q = 0; q < p; q++) out[w++] = hs[b + q]; }
        out[w] = hs[hsRealEnd];
        hs.swap(out);
    }
    // ---- _createYSA (index_util.cpp:1294-1461) on [0, getLength(hs))
    u64 hs_end = hs.size();
    while (hs_end > 0 && hsIsHead(hs[hs_end - 1]) && !hsHeadPtr(hs[hs_end - 1])) hs_end--;   // Hs::getLength :303-309
    u64 ptr = hsHeadPtr(hs[0]), preX = hsHeadX(hs[0]), prek = 0, k = ptr, block_size = ptr, countMove = 0;
    while (k < hs_end && hsHeadPtr(hs[k])) {
        ptr = hsHeadPtr(hs[k]);
        if (preX != hsHeadX(hs[k])) {
            hs[k - countMove] = hs[k];
            hs[prek] = (hs[prek] & HS_MASK40) + (block_size << 40);                   // setHsHeadPtr :279
            prek = k - countMove; block_size = ptr; preX = hsHeadX(hs[k]);
        } else { countMove++; block_size += ptr - 1; }
        for (u64 q = k + 1; q < k + ptr; q++) hs[q - countMove] = hs[q];
        k += ptr;
    }
    u64 hs_end_mod;
    if (countMove > 2) {
        hs_end_mod = k - countMove;
        hs[prek] = (hs[prek] & HS_MASK40) + (block_size << 40);
        if (hs.size() < k - countMove + 2) hs.resize(k - countMove + 2);
        hs[k - countMove] = hsHead(0, 0); hs[k - countMove + 1] = hsHead(0, 0);
        ix.emptyDir = k - countMove;
    } else {                                                                           // "abort the last block"
        hs_end_mod = prek;
        if (hs.size() < prek + 2) hs.resize(prek + 2);
        hs[prek] = hsHead(0, 0); hs[prek + 1] = hsHead(0, 0);
        ix.emptyDir = prek;
    }
    hs.resize(k + 2 - countMove);
    for (u64 q = 0; q < hs_end_mod; q++)
        if (hsIsHead(hs[q])) { u64 p = hsHeadPtr(hs[q]); std::sort(hs.begin() + (long)q + 1, hs.begin() + (long)(q + p), std::greater<u64>()); }   // _sort_YSA_Block: bodies are distinct words
    u64 count = 0;
    k = 0;
    while (hsHeadPtr(hs[k]) && k < hs_end_mod) {
        ptr = hsHeadPtr(hs[k]);
        if (ptr < thd_blocklimit) ++count;
        else { for (unsigned q = (unsigned)k + 1; q < k + ptr; q++) if (hsBodyY(hs[q] ^ hs[q - 1])) ++count; ++count; }
        k += ptr;
    }
    u64 len = 1; while (len < count * alpha) len <<= 1;                               // XString::_fullSize :221-232
    ix.xstr.assign(len, XNode()); ix.xmask = len - 1;
    for (u64 i = 0; i < hs_end_mod; i++) {
        if (hsIsHead(hs[i]) && hsHeadPtr(hs[i])) {
            ptr = hsHeadPtr(hs[i]);
            if (ptr < thd_blocklimit) {
                for (unsigned q = (unsigned)i + 1; q < i + ptr; q++) hs[q] &= ~(HS_BODYYMASK << 41);   // setHsBodyY(.., 0)
                requestXNode(ix, hsHeadX(hs[i]), (unsigned)(i + 1), 1);
            } else {
                u64 xval = hsHeadX(hs[i]);
                requestXNode(ix, xval, ~1u, 3);
                for (unsigned q = (unsigned)i + 1; q < i + ptr; q++)
                    if (hsBodyY(hs[q] ^ hs[q - 1])) requestXNode(ix, xval + ((hs[q] & ((1ULL << 61) - (1ULL << 41))) >> 1), q, 1);
            }
        }
    }
    ix.ysa.swap(hs);
}
static inline u64 make_anchor(u64 id, u64 x, u64 y, u64 strand) { return create_cord(id, x - y + const_anchor_zero, y, strand); }   // cords.cpp:319

// ------------------------------------------------------------- features ----
struct int96 { int v[3]; };
static const int window48 = 48;
static const short infiN = 31;
static const unsigned infi_mask30 = (1u << 31) - 1;
static const short units[25] = {                                                     // pmpfinder.cpp:543-548
    0, 6, 12, 18, infiN,
    24, (1 << 8) + 0, (1 << 8) + 6, (1 << 8) + 12, infiN,
    (1 << 8) + 18, (1 << 8) + 24, (2 << 8) + 0, (2 << 8) + 6, infiN,
    (2 << 8) + 12, (2 << 8) + 18, (2 << 8) + 24, infiN, infiN,
    infiN, infiN, infiN, infiN, infiN};
static inline void add2merInt96(int96 &val, const uint8_t *it) {                     // pmpfinder.cpp:549-555
    unsigned ordV = it[0] * 5 + it[1];
    unsigned i = units[ordV] >> 8;
    unsigned addVal = (1u << (units[ordV] & 255)) & infi_mask30;
    val.v[i] += (int)addVal;
}
static inline void inc96(int96 &a, const int96 &b) { a.v[0] += b.v[0]; a.v[1] += b.v[1]; a.v[2] += b.v[2]; }
static inline void dec96(int96 &a, const int96 &b) { a.v[0] -= b.v[0]; a.v[1] -= b.v[1]; a.v[2] -= b.v[2]; }

// serial version, used for reads: pmpfinder.cpp:556-588
static void createFeatures2_48(const uint8_t *it_str, i64 n, std::vector<int96> &f) {
    const int scpt_step = 16, scpt_bit = 4;
    int addMod3[3] = {1, 2, 0};
    int96 zero96 = {{0, 0, 0}};
    int96 buffer[3] = {zero96, zero96, zero96};
    f.assign((size_t)((n - window48) / scpt_step + 1), zero96);
    f[0] = zero96;
    for (unsigned i = 0; i < 3; i++) {
        for (unsigned j = i << scpt_bit; j < (i << scpt_bit) + scpt_step; j++) add2merInt96(buffer[i], it_str + j);
        inc96(f[0], buffer[i]);
    }
    int next = 1, ii = 0;
    for (int i = scpt_step; i < n - window48 - 1; i += scpt_step) {
        f[next] = f[next - 1];
        dec96(f[next], buffer[ii]);
        buffer[ii] = zero96;
        for (int j = i - scpt_step + window48; j < i + window48; j++) add2merInt96(buffer[ii], it_str + j);
        inc96(f[next], buffer[ii]);
        ii = addMod3[ii];
        next++;
    }
    f.resize(next);
}
// parallel version, used for genomes: pmpfinder.cpp:589-652 (thread loop run serially)
static void createFeatures2_48_par(const uint8_t *it_str, i64 n, std::vector<int96> &f, unsigned threads) {
    const int scpt_step = 16, scpt_bit = 4;
    int window = window48;
    if (n < window) { f.clear(); return; }
    int96 zero96 = {{0, 0, 0}};
    f.assign((size_t)(((n - window) >> scpt_bit) + 1), zero96);
    i64 range = (n - window) / scpt_step + 1;
    if (range < (i64)threads) { createFeatures2_48(it_str, n, f); return; }
    for (unsigned thd_id = 0; thd_id < threads; thd_id++) {
        i64 chunk_size = range / threads;
        i64 thd_begin = thd_id * (chunk_size + 1);
        unsigned id1 = (unsigned)(range - chunk_size * threads);
        if (thd_id >= id1) thd_begin = id1 + chunk_size * thd_id;
        else ++chunk_size;
        i64 thd_end = thd_begin + chunk_size;
        i64 next = thd_begin;
        thd_begin *= scpt_step;
        thd_end *= scpt_step;
        int addMod3[3] = {1, 2, 0};
        int96 buffer[3] = {zero96, zero96, zero96};
        f[next] = zero96;
        for (unsigned i = 0; i < 3; i++) {
            unsigned tmp = (unsigned)(thd_begin + (i << scpt_bit));
            for (unsigned j = tmp; j < tmp + scpt_step; j++) add2merInt96(buffer[i], it_str + j);
            inc96(f[next], buffer[i]);
        }
        int ii = 0;
        next++;
        for (int i = (int)(thd_begin + scpt_step); i < thd_end; i += scpt_step) {
            f[next] = f[next - 1];
            dec96(f[next], buffer[ii]);
            buffer[ii] = zero96;
            for (int j = i - scpt_step + window; j < i + window; j++) add2merInt96(buffer[ii], it_str + j);
            inc96(f[next], buffer[ii]);
            ii = addMod3[ii];
            next++;
        }
    }
}
static const int max31 = 31;
static const int mxu31 = (max31 << 24) + (max31 << 18) + (max31 << 12) + (max31 << 6) + max31;
static inline i64 scriptDist63_31(int s1, int s2) {                                  // pmpfinder.cpp:497-506
    int d = s1 + mxu31 - s2;
    int mask = 63;
    return std::abs((d >> 24 & mask) - max31) + std::abs((d >> 18 & mask) - max31) + std::abs((d >> 12 & mask) - max31) +
           std::abs((d >> 6 & mask) - max31) + std::abs((d & mask) - max31);
}
static inline i64 windowDist2_48(const int96 *it1, const int96 *it2) {               // pmpfinder.cpp:523-533 (scpt_num 2, int_step 3)
    i64 sum = 0;
    for (unsigned i = 0; i < 6; i += 3)
        sum += scriptDist63_31(it1[i].v[0], it2[i].v[0]) + scriptDist63_31(it1[i].v[1], it2[i].v[1]) + scriptDist63_31(it1[i].v[2], it2[i].v[2]);
    return sum;
}

// ApxMapParm2_48 (pmpfinder.cpp:158-185,211-215)
static const unsigned P_windowThreshold = 36, P_windowThresholdReject = 50, P_windowSize = 96;
static const unsigned P_sup = 6, P_med = 5, P_inf = 3, P_abort_score = 1000;
static const unsigned window_size = 96;

typedef std::vector<int96> Feat;

static unsigned _windowDist(const Feat &f1, const Feat &f2, u64 x1, u64 x2) {        // pmpfinder.cpp:680-695
    u64 d = 2 * (3 - 1);
    if (x1 + d < f1.size() && x2 + d < f2.size()) return (unsigned)windowDist2_48(&f1[x1], &f2[x2]);
    return P_abort_score;
}
// unchecked variant: pmpfinder.cpp:655-663.  Reads outside the feature arrays are
// pinned to abort_score here (the reference reads out of bounds; only reachable
// for hits within 48 bases of a reference-sequence end).
static unsigned __windowDist(const Feat &f1, const Feat &f2, u64 x1, u64 x2) {
    if (x1 + 3 < f1.size() && x2 + 3 < f2.size()) return (unsigned)windowDist2_48(&f1[x1], &f2[x2]);
    return P_abort_score;
}

// ---------------------------------------------------------------- parms ----
struct Parms {          // PMPParms as the default preset -p 1 leaves it (mapper.cpp:181-188)
    int thd_alpha = 15;           // GetIndexMatchAllParms pmpfinder.cpp:1771-1784
    int f_score_type = 0;         // ChainAnchorsHitsParms pmpfinder.cpp:2482-2503
    int thd_best_n = 50, thd_drop_score = 45, thd_min_chain_len = 1;
    unsigned thd_chain_depth = 20; u64 thd_chain_dx_depth = 300;
    float thd_stop_chain_len_ratio = 0.0f;
    void toggle(int i) { thd_alpha = i ? 7 : 15; f_score_type = i ? 1 : 0; }
};

struct Stats { u64 lookups = 0, bucket_entries = 0, anchors = 0, samples = 0; };

// ---------------------------------------------------------- seed lookup ----
static inline u64 val2Anchor(u64 e, u64 y, u64 read_len, u64 shape_strand) {         // index_util.cpp:1509-1520
    if (get_cord_strand(e) ^ shape_strand) {
        u64 cordy = read_len - 1 - y;
        return (e - (cordy << 20) + cordy - get_cord_y(e)) | F_STRAND;
    }
    return (e - (y << 20) + y - get_cord_y(e)) & ~F_STRAND;
}
static void getDIndexMatchAll(const DIndex &index, const uint8_t *read, u64 read_len, std::vector<u64> &set,
                              u64 read_str, u64 read_end, const Parms &pm, Stats &st) { // pmpfinder.cpp:1856-1913
    int dt = 0;
    Shape shape;
    u64 xpre = 0;
    hashInit(shape, read);
    for (u64 k = read_str + shape.span; k + shape.span < read_end; k++) {   // k < read_end - span (unsigned in reference; read_end>=span here)
        hashNexth(shape, read + k);
        if (++dt == pm.thd_alpha) {
            dt = 0;
            hashNextX(shape, read + k);
            st.samples++;
            if (shape.XValue ^ xpre) {
                i64 str_ = index.dir[shape.XValue];
                i64 end_ = index.dir[shape.XValue + 1];
                st.lookups++;
                st.bucket_entries += (u64)(end_ - str_);
                for (i64 i = str_; i < end_; i++) {
                    u64 hs_y = get_cord_y(index.hs[i]);
                    u64 val = hs_y ^ shape.YValue;
                    if (val == 0 || (val >> __builtin_ctzl(val)) < 4) {
                        set.push_back(val2Anchor(index.hs[i], k, read_len, shape.strand));
                        st.anchors++;
                    }
                }
                xpre = shape.XValue;
            }
        }
    }
}


static void getHIndexMatchAll(const HIndex &index, const uint8_t *read, u64 read_len, std::vector<u64> &set, u64 map_str, u64 map_end,
                              const Parms &pm, Stats &st) {                           // pmpfinder.cpp:1918-1974
    const u64 thd_delta = 64;                                                          // GetIndexMatchAllParms pmpfinder.cpp:1769-1776
    int dt = 0;
    Shape shape; shape.span = index.span; shape.weight = index.weight;
    u64 xpre = 0;
    hashInit(shape, read);
    u64 read_str = get_cord_y(map_str), read_end = get_cord_y(map_end);
    u64 idx_str = getCordX40(map_str), idx_end = getCordX40(map_end);
    for (unsigned k = (unsigned)read_str; k < read_end - shape.span; k++) {
        hashNexth(shape, read + k);
        if (++dt == pm.thd_alpha) {
            dt = 0;
            hashNextX(shape, read + k);
            st.samples++;
            if (shape.XValue ^ xpre) {
                xpre = shape.XValue;
                u64 pos = getXDir(index, shape.XValue, shape.YValue);
                u64 ptr = hsHeadPtr(index.ysa[pos - 1]);
                st.lookups++;
                if (pos == index.emptyDir || ptr >= thd_delta) { dt = 0; continue; }
                while (hsBodyY(index.ysa[pos]) == shape.YValue || hsBodyY(index.ysa[pos]) == 0) {
                    u64 idx = hsBodyS(index.ysa[pos]);
                    st.bucket_entries++;
                    if (idx >= idx_str && idx < idx_end) {
                        u64 id = (idx >> 30) & ((1ULL << 10) - 1), x = idx & ((1ULL << 30) - 1);   // _getSA_i1 / _getSA_i2 index_util.cpp:110-117
                        if (((index.ysa[pos] & HS_BODYCODEFLAG) >> 40) ^ shape.strand) set.push_back(make_anchor(id, x, read_len - 1 - k, 1));
                        else set.push_back(make_anchor(id, x, k, 0));
                        st.anchors++;
                    }
                    if (++pos > index.ysa.size() - 1) break;
                }
            }
        }
    }
}

// -------------------------------------------------------- anchor filter ----
static void binningFilter(std::vector<u64> &anchors) {                               // pmpfinder.cpp:1979-2012
    const u64 thd_accept_bin = 10, bin_size = 30000;
    std::vector<u64> bins_counter(40000, 0);   // reference: 10000 + broken resize path (App. C.8); x-field < 2^30 -> bin < 35792
    std::vector<u64> bins_pointer(anchors.size());
    for (size_t i = 0; i < anchors.size(); i++) {
        u64 bin_i = get_cord_x(anchors[i]) / bin_size;
        bins_pointer[i] = bin_i;
        ++bins_counter[bin_i];
    }
    unsigned ii = 0;
    for (size_t i = 0; i < anchors.size(); i++)
        if (bins_counter[bins_pointer[i]] > thd_accept_bin) anchors[ii++] = anchors[i];
    if (ii != 0) anchors.resize(ii);
}
static void filterAnchorsList(std::vector<u64> &anchors, std::vector<std::pair<unsigned, unsigned>> &list,
                              u64 density, u64 accept_min, unsigned err_bit) {        // pmpfinder.cpp:2019-2068
    if (anchors.size() <= 1) return;
    anchors[0] = 0;
    const u64 thd_1k_bit = 10;
    std::sort(anchors.begin(), anchors.end());   // ska_sort ascending u64 (base.cpp:570)
    u64 ak2 = anchors[1];
    u64 block_str = 1, count_anchors = 0;
    u64 min_y = ~0ULL, max_y = 0;
    for (unsigned i = 1; i < anchors.size(); i++) {
        u64 anc_y = get_cord_y(anchors[i]);
        u64 dy2 = (u64)std::llabs((i64)(anc_y - get_cord_y(ak2)));
        int f_continuous = getCordX40(anchors[i] - ak2) < (dy2 >> err_bit);
        if (f_continuous) {
            if (min_y > anc_y) min_y = anc_y;
            if (max_y < anc_y) max_y = anc_y;
            ak2 = anchors[(block_str + i) >> 1];
            ++count_anchors;
        }
        if (!f_continuous || i == anchors.size() - 1) {
            u64 thd_accept_num = std::max(((max_y - min_y) * density >> thd_1k_bit), accept_min);
            if (count_anchors > thd_accept_num) list.push_back(std::make_pair((unsigned)block_str, i));
            block_str = i;
            ak2 = anchors[i];
            min_y = anc_y;
            max_y = anc_y;
            count_anchors = 1;
        }
    }
}
static void filterAnchors1(std::vector<u64> &anchors, u64 density, u64 accept_min, unsigned err_bit) { // pmpfinder.cpp:2073-2091
    if (anchors.size() <= 1) return;
    unsigned ii = 0;
    std::vector<std::pair<unsigned, unsigned>> list;
    filterAnchorsList(anchors, list, density, accept_min, err_bit);
    for (size_t i = 0; i < list.size(); i++)
        for (unsigned j = list[i].first; j < list[i].second; j++) anchors[ii++] = anchors[j];
    anchors.resize(ii);
}

// ------------------------------------------------------------- chaining ----
struct ChainsRecord { int score, score2, len, p2anchor, root_ptr, f_leaf; };
static const int chain_end = -1;

static int getApxChainScore0(u64 a1, u64 a2) {                                       // cluster_util.cpp:337-385
    i64 dy = (i64)(get_cord_y(a1) - get_cord_y(a2));
    if (dy < 5) return -10000;
    i64 thd_min_dy = 50;
    i64 dx = (i64)(getAnchorX(a1) - getAnchorX(a2));
    i64 da = std::llabs(dx - dy);
    i64 derr = (100 * da) / std::max(std::max((i64)std::llabs(dy), (i64)std::llabs(dx)), thd_min_dy);
    if (derr >= 100) return -1000;
    int score_dy = (int)dy;
    int score_derr = (int)da;
    if (da < 30) return 100 - score_dy;
    return 100 - score_dy - score_derr;
}
static int getApxChainScore(u64 a1, u64 a2) {                                        // cluster_util.cpp:387-443
    i64 dy = (i64)(get_cord_y(a1) - get_cord_y(a2));
    if (dy < 10) return -10000;
    i64 thd_min_dy = 50;
    i64 dx = (i64)(getAnchorX(a1) - getAnchorX(a2));
    i64 da = std::llabs(dx - dy);
    i64 derr = (100 * da) / std::max(std::max((i64)std::llabs(dy), (i64)std::llabs(dx)), thd_min_dy);
    int score_derr;
    if (derr < 5) score_derr = (int)(4 * derr);
    else if (derr < 10) score_derr = (int)(6 * derr - 10);
    else if (derr < 100) score_derr = (int)(derr * derr - 5 * derr);
    else return -1000;
    int score_dy;
    dy /= 15;
    if (dy < 150) score_dy = (int)(dy / 5);
    else if (dy < 100) score_dy = (int)(dy - 30);
    else if (dy < 10000) score_dy = (int)(dy * dy / 200 + 20);
    else score_dy = 10000;
    if (da < 10) return 100 - score_dy;
    return 100 - score_dy - score_derr;
}

static void getBestChains(std::vector<u64> &anchors, std::vector<ChainsRecord> &chains, unsigned it_str, unsigned it_end,
                          u64 thd_chain_depth, u64 thd_chain_dx_depth, int score_type, u64 *pair_evals) { // cluster_util.cpp:53-111
    if (anchors.empty()) return;
    int new_score = 0, new_max_score = 0, max_j = 0;
    chains[0].score = 0; chains[0].len = 1; chains[0].p2anchor = chain_end;
    for (int i = (int)it_str; i < (int)it_end; i++) {
        int j_str = std::max(0, i - (int)thd_chain_depth);
        max_j = i;
        new_max_score = -1;
        for (int j = i - 1; j >= 0 && (j >= j_str || getAnchorX(anchors[j]) - getAnchorX(anchors[i]) < thd_chain_dx_depth); j--) {
            new_score = score_type ? getApxChainScore0(anchors[j], anchors[i]) : getApxChainScore(anchors[j], anchors[i]);
            if (pair_evals) ++*pair_evals;
            if (new_score > 0 && new_score + chains[j].score >= new_max_score) {
                max_j = j;
                new_max_score = new_score + chains[j].score;
            }
        }
        if (new_max_score > 0) {
            chains[i].p2anchor = max_j;
            chains[i].score = new_max_score;
            chains[i].len = chains[max_j].len + 1;
            chains[i].score2 = new_max_score;
            chains[i].root_ptr = chains[max_j].root_ptr;
            chains[i].f_leaf = 1;
            chains[max_j].f_leaf = 0;
        } else {
            chains[i].p2anchor = chain_end;
            chains[i].score = 0;
            chains[i].len = 1;
            chains[i].score2 = 0;
            chains[i].root_ptr = i;
            chains[i].f_leaf = 1;
        }
    }
}

template <class E>
static void traceBackChains0(std::vector<E> &elements, std::vector<std::vector<E>> &chains, std::vector<ChainsRecord> &rec,
                             std::vector<int> &chains_score, int min_len, int abort_score, int bestn, float stop_ratio) { // cluster_util.cpp:121-205
    std::vector<E> chain;
    std::vector<int> chain_score;
    int delete_score = -1000;
    int search_times = std::min(50, bestn);
    for (int i = 0; i < search_times; i++) {
        bool f_done = true;
        int max_2nd_score = -1, max_score = -1, max_str = chain_end, max_len = 0;
        for (unsigned j = 0; j < rec.size(); j++) {
            if (rec[j].score > max_score) {
                max_2nd_score = max_score;
                max_str = (int)j;
                max_score = rec[j].score;
                max_len = rec[j].len;
                f_done = false;
            }
        }
        if (!chains.empty()) {
            if (max_len > chains[0].size() * stop_ratio) f_done = false;
        }
        if (f_done || max_score == 0) break;
        if (max_len > min_len && max_score / (max_len - 1) > abort_score) {
            for (int j = max_str; j != chain_end; j = rec[j].p2anchor) {
                if (rec[j].score != delete_score) {
                    chain.push_back(elements[j]);
                    chain_score.push_back(rec[j].score2);
                    rec[j].score = delete_score;
                } else {
                    int infix = rec[j].score2;
                    if (max_score - infix < max_2nd_score) {
                        for (int k = max_str; k != j; k = rec[k].p2anchor) rec[k].score = rec[k].score2 - infix;
                        chain.clear();
                        chain_score.clear();
                    }
                    break;
                }
            }
            if (!chain.empty()) {
                chains.push_back(chain);
                chains_score.insert(chains_score.end(), chain_score.begin(), chain_score.end());
                chain.clear();
                chain_score.clear();
            }
        }
        if (max_str != chain_end) rec[max_str].score = delete_score;
    }
}
template <class E>
static void traceBackChains1(std::vector<E> &elements, std::vector<std::vector<E>> &chains, std::vector<ChainsRecord> &rec,
                             std::vector<int> &chains_score, int min_len, int abort_score, int bestn, float stop_ratio) { // cluster_util.cpp:213-304
    int f_stop = 0;
    std::vector<E> chain;
    std::vector<int> chain_score;
    std::vector<int> new_leaves(4);
    std::vector<std::vector<int>> leaves;
    for (unsigned j = 0; j < rec.size(); j++) {
        if (rec[j].f_leaf) {
            int f_new = 1;
            for (unsigned k = 0; k < leaves.size(); k++) {
                if (leaves[k][0] == rec[j].root_ptr) {
                    leaves[k].push_back((int)j);
                    if (rec[j].score > leaves[k][1]) {
                        leaves[k][1] = rec[j].score;
                        leaves[k][2] = rec[j].len;
                        leaves[k][3] = (int)j;
                    }
                    f_new = 0;
                }
            }
            if (f_new) {
                new_leaves[0] = rec[j].root_ptr;
                new_leaves[1] = rec[j].score;
                new_leaves[2] = rec[j].len;
                new_leaves[3] = (int)j;
                leaves.push_back(new_leaves);
            }
        }
    }
    std::vector<std::pair<int, int>> ranks(leaves.size());
    for (int i = 0; i < (int)leaves.size(); i++) ranks[i] = std::pair<int, int>(i, leaves[i][1]);
    std::sort(ranks.begin(), ranks.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.second > b.second; });
    for (int i = 0; i < std::min(bestn, (int)ranks.size()); i++) {
        int max_score = leaves[ranks[i].first][1];
        int max_len = leaves[ranks[i].first][2];
        int max_str = leaves[ranks[i].first][3];
        int mean_score = max_len > 1 ? max_score / (max_len - 1) : abort_score + 1;
        if (max_len > min_len && mean_score > abort_score) {
            for (int j = max_str; j != chain_end; j = rec[j].p2anchor) {
                chain.push_back(elements[j]);
                chain_score.push_back(rec[j].score2);
            }
            if (!chain.empty()) {
                if (!chains.empty()) {
                    if (float(chain.size()) / chains[0].size() < stop_ratio) f_stop = 1;
                }
                if (!f_stop) {
                    chains.push_back(chain);
                    chains_score.insert(chains_score.end(), chain_score.begin(), chain_score.end());
                    chain.clear();
                    chain_score.clear();
                }
            }
        }
    }
}
template <class E>
static void traceBackChains(std::vector<E> &elements, std::vector<std::vector<E>> &chains, std::vector<ChainsRecord> &rec,
                            std::vector<int> &chains_score, int min_len, int abort_score, int bestn, float stop_ratio) { // cluster_util.cpp:306-335
    unsigned thd_root_num = 50;
    std::vector<int> tmp_count(elements.size(), 0);
    unsigned root_num = 0;
    for (unsigned i = 0; i < rec.size(); i++) {
        if (tmp_count[rec[i].root_ptr] == 0) root_num++;
        tmp_count[rec[i].root_ptr] = 1;
    }
    if (root_num > thd_root_num) traceBackChains0(elements, chains, rec, chains_score, min_len, abort_score, bestn, stop_ratio);
    else traceBackChains1(elements, chains, rec, chains_score, min_len, abort_score, bestn, stop_ratio);
}

struct Debug {   // optional stage dumps for parity debugging
    std::vector<u64> raw_anchors, filtered_anchors, sorted_anchors, hits_chain, hits_blocks, hits_filtered, cords_path;
    bool on = false;
    int pass = 0;
};

static void chainAnchorsHits(std::vector<u64> &anchors, std::vector<u64> &hits, std::vector<int> &hits_score, const Parms &pm,
                             u64 *pair_evals, Debug *dbg) {                           // pmpfinder.cpp:2448-2481
    std::vector<std::vector<u64>> anchors_chains;
    std::sort(anchors.begin(), anchors.end(), [](const u64 &a, const u64 &b) { return getAnchorX(a) > getAnchorX(b); });
    if (dbg && dbg->on && dbg->pass == 0) dbg->sorted_anchors = anchors;
    // chainAnchorsBase cluster_util.cpp:445-462
    if (anchors.size() >= 2) {
        std::vector<ChainsRecord> rec(anchors.size());
        getBestChains(anchors, rec, 0, (unsigned)anchors.size(), pm.thd_chain_depth, pm.thd_chain_dx_depth, pm.f_score_type, pair_evals);
        traceBackChains(anchors, anchors_chains, rec, hits_score, pm.thd_min_chain_len, pm.thd_drop_score, pm.thd_best_n, pm.thd_stop_chain_len_ratio);
    }
    for (size_t i = 0; i < anchors_chains.size(); i++) {
        for (size_t j = 0; j < anchors_chains[i].size(); j++) hits.push_back(hit2Cord_dstr(anchors_chains[i][j]));
        set_block_end(hits.back());
    }
}

// --------------------------------------------------------------- blocks ----
static int gather_blocks_(std::vector<u64> &cords, std::vector<UPair> &str_ends, std::vector<UPair> &str_ends_p, u64 str_, u64 end_,
                          u64 read_len, u64 thd_large_gap, u64 thd_cord_size, int f_set_end) { // pmpfinder.cpp:1484-1530
    str_ends.clear();
    if (cords.size() < 2) return 0;
    u64 d_shift_max = thd_cord_size / 2;
    u64 d_shift = d_shift_max;
    unsigned p_str = (unsigned)str_;
    for (unsigned i = (unsigned)str_ + 1; i < end_; i++) {
        if (is_block_end(cords[i - 1]) || !isCordsConsecutive_(cords[i - 1], cords[i], thd_large_gap)) {
            d_shift = std::min(read_len - get_cord_y(cords[p_str]) - 1, d_shift_max);
            u64 b_str = shift_cord(cords[p_str], (i64)d_shift, (i64)d_shift);
            d_shift = std::min(read_len - get_cord_y(cords[i - 1]) - 1, d_shift_max);
            u64 b_end = shift_cord(cords[i - 1], (i64)d_shift, (i64)d_shift);
            str_ends.push_back(UPair(b_str, b_end));
            str_ends_p.push_back(UPair(p_str, i));
            if (f_set_end) set_block_end(cords[i - 1]);
            p_str = i;
        }
    }
    d_shift = std::min(read_len - get_cord_y(cords.back()) - 1, d_shift_max);
    u64 b_str = shift_cord(cords[p_str], (i64)d_shift, (i64)d_shift);
    d_shift = std::min(read_len - get_cord_y(cords.back()) - 1, d_shift_max);
    u64 b_end = shift_cord(cords.back(), (i64)d_shift, (i64)d_shift);
    str_ends.push_back(UPair(b_str, b_end));
    str_ends_p.push_back(UPair(p_str, cords.size()));
    return 0;
}

static void preFilterChains2(std::vector<u64> &hits, std::vector<UPair> &str_ends_p) { // pmpfinder.cpp:2366-2446 (getCordXY = get_cord_y)
    std::vector<UPair> tmp;
    std::vector<u64> xy_strs(str_ends_p.size());
    std::vector<u64> xycuts(2 * str_ends_p.size());
    const u64 mask = 1ULL << 62;
    for (size_t i = 0; i < str_ends_p.size(); i++) {
        xycuts[2 * i] = str_ends_p[i].first;
        xycuts[2 * i + 1] = (str_ends_p[i].second - 1) | mask;
    }
    for (size_t i = 0; i < xy_strs.size(); i++) xy_strs[i] = str_ends_p[i].first;
    std::sort(xycuts.begin(), xycuts.end(), [&hits, mask](const u64 &a, const u64 &b) {
        return get_cord_y(hits[a & (~mask)]) < get_cord_y(hits[b & (~mask)]);
    });
    for (size_t i = 0; i < xycuts.size(); i++) {
        u64 cuty = get_cord_y(hits[xycuts[i] & (~mask)]);
        for (size_t j = 0; j < xy_strs.size() && xy_strs[j] < hits.size(); j++) {
            if (cuty < get_cord_y(hits[xy_strs[j]])) continue;
            for (u64 k = xy_strs[j]; k < str_ends_p[j].second; k++) {
                if (xycuts[i] & mask) {
                    if (get_cord_y(hits[k]) == cuty) {
                        u64 lowery = xy_strs[j], uppery = k + 1;
                        if (lowery != uppery) { tmp.push_back(UPair(lowery, uppery)); xy_strs[j] = uppery; }
                        break;
                    } else if (get_cord_y(hits[k]) > cuty) {
                        u64 lowery = xy_strs[j], uppery = k;
                        if (lowery != uppery) { tmp.push_back(UPair(lowery, uppery)); xy_strs[j] = uppery; }
                        break;
                    }
                } else {
                    if (get_cord_y(hits[k]) >= cuty) {
                        u64 lowery = xy_strs[j], uppery = k;
                        if (lowery != uppery) { tmp.push_back(UPair(lowery, uppery)); xy_strs[j] = uppery; }
                        break;
                    }
                }
            }
        }
    }
    str_ends_p = tmp;
    std::sort(str_ends_p.begin(), str_ends_p.end(), [](const UPair &a, const UPair &b) { return a.second < b.second; });
    for (size_t i = 0; i < str_ends_p.size(); i++) set_block_end(hits[str_ends_p[i].second - 1]);
}

typedef int (*ScoreFunc2)(u64, u64, u64, u64, u64, int, float);

static int getApxChainScore2(u64 cord11, u64 cord12, u64 cord21, u64 cord22, u64 read_len, int, float) { // cluster_util.cpp:586-631
    (void)read_len; (void)cord12; (void)cord21;
    i64 thd_max_d = 20000, thd_indel_trigger = 100, thd_indel_op = 30;
    i64 dy = (i64)(get_cord_y(cord11) - get_cord_y(cord22));
    i64 dx = (i64)(get_cord_x(cord11) - get_cord_x(cord22));
    if (dx < 0 || dy < 0 || get_cord_strand(cord11 ^ cord22) || dx > thd_max_d || dy > thd_max_d) return INT_MIN;
    i64 thd_min_dy = 100;
    i64 da = std::llabs(dx - dy);
    i64 derr = (100 * da) / std::max(std::max((i64)std::llabs(dy), thd_min_dy), (i64)std::llabs(dx));
    if (da > thd_indel_trigger || derr > 50) {
        if (dx < dy) return (int)(100 - thd_indel_op - dy / 1000 - dx / 100);
        return (int)(100 - thd_indel_op - dy / 100 - dx / 1000);
    }
    return (int)(100 - dy / 95);
}
static int getChainBlockDxDy(u64 cord11, u64 cord12, u64 cord21, u64 cord22, u64 read_len, int strand, i64 &dx, i64 &dy) { // cluster_util.cpp:774-808
    if (get_cord_strand(cord11) != (unsigned)strand) {
        if (get_cord_strand(cord22) != (unsigned)strand) {
            dy = (i64)(get_cord_y(cord21) - get_cord_y(cord12));
            dx = (i64)(get_cord_x(cord21) - get_cord_x(cord12));
        } else {
            dy = (i64)(read_len - get_cord_y(cord12) - 1 - get_cord_y(cord22));
            dx = (i64)(get_cord_x(cord11) - get_cord_x(cord22));
        }
    } else {
        if (get_cord_strand(cord22) != (unsigned)strand) {
            dy = (i64)(get_cord_y(cord11) - read_len + 1 + get_cord_y(cord21));
            dx = (i64)(get_cord_x(cord11) - get_cord_x(cord22));
        } else {
            dy = (i64)(get_cord_y(cord11) - get_cord_y(cord22));
            dx = (i64)(get_cord_x(cord11) - get_cord_x(cord22));
        }
    }
    return (int)get_cord_strand(cord11 ^ cord22);
}
static int getApxChainScore3(u64 cord11, u64 cord12, u64 cord21, u64 cord22, u64 read_len, int chn_block_strand, float ins_ratio) { // cluster_util.cpp:811-863
    i64 thd_min_dy = -80;
    i64 thd_min_dx = -(i64)read_len;
    i64 dx, dy, da;
    int f_type = getChainBlockDxDy(cord11, cord12, cord21, cord22, read_len, chn_block_strand, dx, dy);
    i64 thd_max_dy = (i64)(read_len * ins_ratio);
    i64 thd_max_dx = 15000, thd_dup_trigger = -50;
    i64 dx_ = std::llabs(dx), dy_ = std::llabs(dy);
    da = dx - dy;
    int score = 0;
    if (dy < thd_min_dy || dy > thd_max_dy || dx < thd_min_dx || dx_ > thd_max_dx) score = INT_MIN;
    else {
        i64 score_dy = dy_ > 2000 ? std::min(dy_ / 25 - 50, (i64)70) : dy_ / 40;
        i64 score_dx = dx_ > 2000 ? std::min(dx_ / 25 - 50, (i64)70) : dx_ / 40;
        if (f_type == 1) { if (dx > thd_min_dx) score = (int)(75 - score_dy); }
        else if (da < -std::max(dx_ / 4, (i64)50)) {
            if (dx > thd_dup_trigger) score = (int)(80 - score_dx);
            else score = (int)(80 - score_dy);
        } else if (da > std::max(dy / 4, (i64)50)) score = (int)(80 - score_dy);
        else score = (int)(100 - score_dy);
    }
    return score;
}

static void getBestChains2(std::vector<u64> &hits, std::vector<UPair> &sep, std::vector<int> &sep_score, std::vector<ChainsRecord> &rec,
                           u64 read_len, ScoreFunc2 score2, int chn_block_strand) {   // cluster_util.cpp:469-526
    int thd_chain_depth = 20, new_score = 0, new_max_score = 0, max_j = 0;
    rec[0].score = sep_score[0];
    rec[0].len = (int)(sep[0].second - sep[0].first);
    rec[0].p2anchor = chain_end;
    for (unsigned i = 0; i < sep.size(); i++) {
        int j_str = std::max(0, int(i) - thd_chain_depth);
        max_j = (int)i;
        new_max_score = -1;
        for (unsigned j = (unsigned)j_str; j < i; j++) {
            new_score = score2(hits[sep[j].first], hits[sep[j].second - 1], hits[sep[i].first], hits[sep[i].second - 1], read_len, chn_block_strand, 1.0f);
            // int addition as in the reference; new_score > 0 guards the INT_MIN case
            if (new_score > 0 && new_score + rec[j].score + sep_score[i] >= new_max_score) {
                max_j = (int)j;
                new_max_score = new_score + rec[j].score + sep_score[i];
            }
        }
        if (new_max_score > 0) {
            rec[i].p2anchor = max_j;
            rec[i].score = new_max_score;
            rec[i].len = (int)(sep[i].second - sep[i].first) + rec[max_j].len;
            rec[i].score2 = rec[i].score;
            rec[i].root_ptr = rec[max_j].root_ptr;
            rec[i].f_leaf = 1;
            rec[max_j].f_leaf = 0;
        } else {
            rec[i].p2anchor = chain_end;
            rec[i].score = sep_score[i];
            rec[i].len = (int)(sep[i].second - sep[i].first);
            rec[i].score2 = rec[i].score;
            rec[i].root_ptr = (int)i;
            rec[i].f_leaf = 1;
        }
    }
}
static void chainBlocksBase(std::vector<std::vector<UPair>> &chains, std::vector<u64> &records, std::vector<UPair> &sep, std::vector<int> &sep_score,
                            u64 read_len, ScoreFunc2 score2, int chn_block_strand, int min_len, int abort_score, int thd_best_n, int f_sort,
                            float stop_ratio) {                                       // cluster_util.cpp:533-577
    if (sep.size() < 2) return;
    std::vector<ChainsRecord> rec;
    std::vector<int> chains_score;
    std::vector<unsigned> ptr;
    for (unsigned i = 0; i < sep.size(); i++) ptr.push_back(i);
    if (f_sort) {
        std::sort(ptr.begin(), ptr.end(), [&records, &sep](const unsigned &a, const unsigned &b) {
            return getCordX40(records[sep[a].first]) > getCordX40(records[sep[b].first]);
        });
    }
    std::vector<UPair> sep_tmp(sep.size());
    std::vector<int> score_tmp(sep_score.size());
    for (unsigned i = 0; i < sep.size(); i++) { sep_tmp[i] = sep[ptr[i]]; score_tmp[i] = sep_score[ptr[i]]; }
    rec.resize(sep_tmp.size());
    getBestChains2(records, sep_tmp, score_tmp, rec, read_len, score2, chn_block_strand);
    traceBackChains(sep_tmp, chains, rec, chains_score, min_len, abort_score, thd_best_n, stop_ratio);
}
static void _filterBlocksHits(std::vector<std::vector<UPair>> &chains, std::vector<u64> &hits, u64 read_len) { // cluster_util.cpp:633-719
    if (chains.empty()) return;
    std::vector<UPair> best_chain(chains[0].size());
    std::vector<u64> hits_tmp;
    u64 len_current = 0;
    for (unsigned i = 0; i < chains[0].size(); i++) {
        for (u64 j = chains[0][i].first; j < chains[0][i].second; j++) { hits_tmp.push_back(hits[j]); unset_block_end(hits_tmp.back()); }
        len_current += chains[0][i].second - chains[0][i].first;
        best_chain[i] = chains[0][i];
    }
    set_block_end(hits_tmp.back());
    float thd_major_bound = 0.8 * len_current;
    unsigned thd_major_limit = 5, major_n = 1;
    i64 thd_x_max_delta = (i64)(read_len * 2);
    bool f_append = false;
    for (unsigned i = 1; i < chains.size(); i++) {
        len_current = 0;
        f_append = false;
        for (unsigned j = 0; j < chains[i].size(); j++) len_current += chains[i][j].second - chains[i][j].first;
        if (major_n < thd_major_limit && len_current > thd_major_bound) { f_append = true; ++major_n; }
        else if (len_current) {}
        else {
            f_append = true;
            for (unsigned j = 0; j < chains[i].size() && f_append; j++) {
                for (unsigned k = 0; k < best_chain.size() && f_append; k++) {
                    u64 str_major = hits[best_chain[k].first], end_major = hits[best_chain[k].second - 1];
                    u64 str_current = hits[chains[i][j].first], end_current = hits[chains[i][j].second - 1];
                    i64 dx_lower = (i64)(get_cord_x(str_major) - get_cord_x(str_current));
                    i64 dx_upper = (i64)(get_cord_x(end_current) - get_cord_x(end_major));
                    f_append = dx_lower <= thd_x_max_delta && dx_upper < thd_x_max_delta &&
                               !_isCordyOverLap(str_major, end_major, str_current, end_current, read_len);
                }
            }
            for (unsigned j = 0; j < chains[i].size() && f_append; j++) best_chain.insert(best_chain.end(), chains[i].begin(), chains[i].end());
        }
        if (f_append) {
            for (unsigned j = 0; j < chains[i].size(); j++)
                for (u64 k = chains[i][j].first; k < chains[i][j].second; k++) { hits_tmp.push_back(hits[k]); unset_block_end(hits_tmp.back()); }
            set_block_end(hits_tmp.back());
        }
        set_block_end(hits_tmp.back());
    }
    hits = hits_tmp;
}
static void chainBlocksHits(std::vector<u64> &hits, std::vector<UPair> &sep, std::vector<int> &sep_score, u64 read_len) { // cluster_util.cpp:721-732
    std::vector<std::vector<UPair>> hits_chains;
    chainBlocksBase(hits_chains, hits, sep, sep_score, read_len, &getApxChainScore2, 0, 1, 0, 3, 1, 0.7f);
    _filterBlocksHits(hits_chains, hits, read_len);
}

// ------------------------------------------------------------- windows ----
static u64 previousWindow(const Feat &f1, const Feat &f2, u64 cord, float *score = nullptr) {   // pmpfinder.cpp:883-945 (score form :838-880: adds the window's distance)
    u64 genomeId = get_cord_id(cord), strand = get_cord_strand(cord);
    u64 x_suf = get_cord_x(cord) >> 4, y_suf = get_cord_y(cord) >> 4;
    u64 x_min = 0, y, new_cord = 0;
    if (y_suf < P_med || x_suf < P_sup) return 0;
    y = y_suf - P_med;
    unsigned min = ~0u;
    for (u64 x = x_suf - P_sup; x < x_suf - P_inf; x += 1) {
        unsigned tmp = __windowDist(f1, f2, y, x);
        if (tmp < min) { min = tmp; x_min = x; }
    }
    if (min > P_windowThreshold) return 0;
    if (x_suf - x_min > P_med)
        new_cord = createCord(create_id_x(genomeId, (x_suf - P_med) << 4), (x_suf - x_min - P_med + y) << 4, strand);
    else
        new_cord = createCord(create_id_x(genomeId, x_min << 4), y << 4, strand);
    if (score) *score += (float)min;
    return new_cord;
}
static u64 nextWindow(const Feat &f1, const Feat &f2, u64 cord, float *score = nullptr) {       // pmpfinder.cpp:1079-1150 (score form :995-1045)
    u64 genomeId = get_cord_id(cord), strand = get_cord_strand(cord);
    u64 x_pre = get_cord_x(cord) >> 4, y_pre = get_cord_y(cord) >> 4;
    u64 x_min = 0, y, new_cord = 0;
    unsigned min = ~0u;
    unsigned len1 = (unsigned)f1.size(), len2 = (unsigned)f2.size();
    if (y_pre + P_sup * 2 > len1 || x_pre + P_sup * 2 > len2) return 0;
    y = y_pre + P_med;
    for (u64 x = x_pre + P_inf; x < x_pre + P_sup; x += 1) {
        unsigned tmp = __windowDist(f1, f2, y, x);
        if (tmp < min) { min = tmp; x_min = x; }
    }
    if (min > P_windowThreshold) return 0;
    if (x_min - x_pre > P_med)
        new_cord = createCord(create_id_x(genomeId, (x_pre + P_med) << 4), (x_pre + P_med - x_min + y) << 4, strand);
    else
        new_cord = createCord(create_id_x(genomeId, x_min << 4), y << 4, strand);
    if (score) *score += (float)min;
    return new_cord;
}
static int extendWindow(const Feat &f1, const Feat &f2, std::vector<u64> &cords, u64 cordy_str, u64 cordy_end) { // pmpfinder.cpp:1152-1178
    unsigned cords_p_str = (unsigned)cords.size() - 1;
    u64 new_cord = 0;
    int n_new_cord = 0;
    while ((new_cord = previousWindow(f1, f2, cords.back())) && get_cord_y(new_cord) >= cordy_str) { cords.push_back(new_cord); ++n_new_cord; }
    unsigned cords_p_end = (unsigned)cords.size();
    for (unsigned k = cords_p_str; k < (cords_p_str + cords_p_end) / 2; k++) std::swap(cords[k], cords[cords.size() - k + cords_p_str - 1]);
    while ((new_cord = nextWindow(f1, f2, cords.back())) && get_cord_y(new_cord) + window_size < cordy_end) { cords.push_back(new_cord); ++n_new_cord; }
    return n_new_cord;
}
static void _filterHits(std::vector<u64> &hits, const Feat f1[2], const std::vector<Feat> &f2) { // pmpfinder.cpp:1417-1445
    unsigned distThd = f2.empty() ? 0 : P_windowThresholdReject;
    int ii_move = 0;
    for (size_t it = 1; it < hits.size(); it++) {
        unsigned dist = _windowDist(f1[get_cord_strand(hits[it])], f2[get_cord_id(hits[it])], get_cord_y(hits[it]) >> 4, get_cord_x(hits[it]) >> 4);
        if (dist < distThd) hits[it - ii_move] = hits[it];
        else ii_move++;
        if (is_block_end(hits[it])) set_block_end(hits[it - ii_move]);
    }
    hits.resize(hits.size() - ii_move);
}
static int path_dst_2(std::vector<u64> &hits, const Feat f1[2], const std::vector<Feat> &f2, std::vector<u64> &cords,
                      u64 read_str, u64 read_end, u64 read_len) {                     // pmpfinder.cpp:1309-1410
    // iterators restated as indices into hits: hitBegin = 1, hitEnd = hits.size()
    i64 hitBegin = 1, hitEnd = (i64)hits.size();
    if (hitBegin >= hitEnd - 1) return 0;
    unsigned thd_cord_size = P_windowSize;
    if (cords.empty()) initCords(cords);
    u64 ready_str, ready_end, cordy_str = 0, cordy_end = 0;
    bool f_sp_l = false, f_sp_r = false, f_block_end = false, f_append = false;
    i64 itt_next = hitBegin + 1;
    i64 itt_first = hitBegin;
    auto isFirstHit = [&hits](i64 it) { return is_block_end(hits[it - 1]); };
    auto isLastHit = [&hits](i64 it) { return is_block_end(hits[it]); };
    for (i64 itt = hitBegin; itt < hitEnd; itt = itt_next++) {
        ready_str = get_cord_strand(hits[itt]) ? read_len - read_end : read_str;
        ready_end = get_cord_strand(hits[itt]) ? read_len - read_str + 1 : read_end;
        i64 da_l = isFirstHit(itt) ? 0 : std::llabs((i64)(get_cord_x(hits[itt]) - get_cord_x(hits[itt - 1]) - get_cord_y(hits[itt]) + get_cord_y(hits[itt - 1])));
        f_sp_l = (da_l > 80) || get_cord_strand(hits[itt] ^ hits[itt - 1]);
        while (1) {
            if (itt_next >= hitEnd || isFirstHit(itt_next)) { f_block_end = 1; itt_first = itt_next; break; }
            i64 da_r = isFirstHit(itt_next) ? 0 : std::llabs((i64)(get_cord_x(hits[itt_next]) - get_cord_x(hits[itt_next - 1]) - get_cord_y(hits[itt_next]) + get_cord_y(hits[itt_next - 1])));
            f_sp_r = (da_r > 80) || get_cord_strand(hits[itt_next] ^ hits[itt_next - 1]);
            if ((get_cord_y(hits[itt]) + thd_cord_size < get_cord_y(hits[itt_next]) && get_cord_x(hits[itt]) + thd_cord_size < get_cord_x(hits[itt_next])) || f_sp_r) break;
            itt_next++;
        }
        if (!f_sp_r && !f_block_end) {
            cordy_str = f_sp_l ? hits[itt] : (isFirstHit(itt) ? ready_str : get_cord_y(cords.back()));
            cordy_end = get_cord_y(hits[itt_next]);
            cords.push_back(hits[itt]);
            unset_block_end(cords.back());
            f_append = true;
        } else {
            if (!f_sp_l && get_cord_y(hits[itt_next - 1]) >= thd_cord_size && get_cord_x(hits[itt_next - 1]) >= thd_cord_size) {
                u64 new_cord = shift_cord(hits[itt_next - 1], -(i64)thd_cord_size, -(i64)thd_cord_size);
                cordy_str = isFirstHit(itt) ? read_str : get_cord_y(new_cord);
                cordy_end = get_cord_y(hits[itt_next - 1]);
                cords.push_back(new_cord);
                unset_block_end(cords.back());
                f_append = true;
            } else f_append = false;
        }
        if (isLastHit(itt) || f_block_end) { f_block_end = true; cordy_end = ready_end; }
        if (f_append) extendWindow(f1[get_cord_strand(hits[itt])], f2[get_cord_id(hits[itt])], cords, cordy_str, cordy_end);
        if (f_block_end) set_block_end(cords.back());
        itt_next = f_block_end ? itt_first : itt_next;
        f_sp_l = false; f_sp_r = false; f_block_end = false; f_append = false;
    }
    return 0;
}

// ---------------------------------------------------------- cord blocks ----
static void clean_blocks_(std::vector<u64> &cords, u64 thd_drop_len, i64 thd_map_error = 50) { // pmpfinder.cpp:1537-1581
    if (cords.empty()) return;
    u64 ptr = 1, len = 0;
    for (unsigned i = 1; i < cords.size(); i++) {
        len++;
        if (!is_block_end(cords[i - 1])) {
            i64 dx = (i64)(get_cord_x(cords[i]) - get_cord_x(cords[ptr - 1]));
            i64 dy = (i64)(get_cord_y(cords[i]) - get_cord_y(cords[ptr - 1]));
            if (dx < 0 || dy < 0) {
                if (std::llabs(dx) < thd_map_error && std::llabs(dy) < thd_map_error) { --len; --ptr; }
                else cords[ptr] = cords[i];
            } else cords[ptr] = cords[i];
        } else cords[ptr] = cords[i];
        if (is_block_end(cords[i])) {
            ptr = len < thd_drop_len ? ptr - len : ptr;
            len = 0;
            set_block_end(cords[ptr]);
        }
        ptr++;
    }
    cords.resize(ptr);
}
static int gather_gaps_y_(std::vector<UPair> &str_ends, std::vector<UPair> &gaps, u64 read_len, u64 thd_gap_size) { // pmpfinder.cpp:1592-1667
    u64 cord_frt = shift_cord(0, 0, 0);
    u64 cord_end = shift_cord(0, 0, (i64)(read_len - 1));
    int gap_lens_sum = 0;
    if (str_ends.empty()) {
        gaps.push_back(UPair(cord_frt, cord_end));
        UPair gap_y = getUPForwardy(gaps.back(), read_len);
        gap_lens_sum += (int)(gap_y.second - gap_y.first);
        return gap_lens_sum;
    }
    std::sort(str_ends.begin(), str_ends.end(), [read_len](const UPair &i, const UPair &j) {
        u64 y1 = get_cord_strand(i.first) ? read_len - get_cord_y(i.second) - 1 : get_cord_y(i.first);
        u64 y2 = get_cord_strand(j.first) ? read_len - get_cord_y(j.second) - 1 : get_cord_y(j.first);
        return y1 < y2;
    });
    u64 f_cover = 0, cordy1 = 0, cordy2 = 0;
    UPair y1 = getUPForwardy(str_ends[0], read_len);
    UPair y2 = y1;
    if (y1.first > thd_gap_size) {
        cordy2 = get_cord_y(y1.first);
        gaps.push_back(UPair(cord_frt, cordy2));
        UPair gap_y = getUPForwardy(gaps.back(), read_len);
        gap_lens_sum += (int)(gap_y.second - gap_y.first);
    }
    for (unsigned i = 1; i < str_ends.size(); i++) {
        if (!f_cover) { y1 = getUPForwardy(str_ends[i - 1], read_len); cordy1 = get_cord_y(y1.second); }
        y2 = getUPForwardy(str_ends[i], read_len);
        cordy2 = get_cord_y(y2.first);
        if (y1.second > y2.second) f_cover = 1;
        else {
            if (y2.first > y1.second && y2.first - y1.second > thd_gap_size) {
                gaps.push_back(UPair(cordy1, cordy2));
                UPair gap_y = getUPForwardy(gaps.back(), read_len);
                gap_lens_sum += (int)(gap_y.second - gap_y.first);
            }
            f_cover = 0;
        }
    }
    u64 max_y_end = f_cover ? y1.second : y2.second;
    if (read_len - max_y_end > thd_gap_size) {
        gaps.push_back(UPair(max_y_end, cord_end));
        UPair gap_y = getUPForwardy(gaps.back(), read_len);
        gap_lens_sum += (int)(gap_y.second - gap_y.first);
    }
    return gap_lens_sum;
}
static void chainBlocksSingleStrand(std::vector<u64> &cords, std::vector<UPair> &sep, std::vector<std::vector<UPair>> &cords_chains,
                                    int strand, u64 read_len, unsigned thd_init_cord_score) { // cluster_util.cpp:936-975
    std::vector<int> sep_score(sep.size());
    if (strand) {
        std::sort(sep.begin(), sep.end(), [&cords, read_len](const UPair &a, const UPair &b) {
            u64 y1 = !get_cord_strand(cords[a.first]) ? read_len - 1 - get_cord_y(cords[a.second - 1]) : get_cord_y(cords[a.first]);
            u64 y2 = !get_cord_strand(cords[b.first]) ? read_len - 1 - get_cord_y(cords[b.second - 1]) : get_cord_y(cords[b.first]);
            return y1 > y2;
        });
    } else {
        std::sort(sep.begin(), sep.end(), [&cords, read_len](const UPair &a, const UPair &b) {
            u64 y1 = get_cord_strand(cords[a.first]) ? read_len - 1 - get_cord_y(cords[a.second - 1]) : get_cord_y(cords[a.first]);
            u64 y2 = get_cord_strand(cords[b.first]) ? read_len - 1 - get_cord_y(cords[b.second - 1]) : get_cord_y(cords[b.first]);
            return y1 > y2;
        });
    }
    for (unsigned i = 0; i < sep_score.size(); i++) sep_score[i] = (int)((sep[i].second - sep[i].first) * thd_init_cord_score);
    chainBlocksBase(cords_chains, cords, sep, sep_score, read_len, &getApxChainScore3, strand, 1, 0, 3, 0, 0.7f);
}
static int getChainBlocksBestStrand(std::vector<std::vector<UPair>> &c1, std::vector<std::vector<UPair>> &c2) { // cluster_util.cpp:979-1019
    std::vector<int> lens1(c1.size()), lens2(c2.size());
    for (unsigned i = 0; i < c1.size(); i++) {
        lens1[i] = i == 0 ? 0 : lens1[i - 1];
        for (unsigned j = 0; j < c1[i].size(); j++) lens1[i] += (int)(c1[i][j].second - c1[i][j].first);
    }
    for (unsigned i = 0; i < c2.size(); i++) {
        lens2[i] = i == 0 ? 0 : lens2[i - 1];
        for (unsigned j = 0; j < c2[i].size(); j++) lens2[i] += (int)(c2[i][j].second - c2[i][j].first);
    }
    for (unsigned i = 0; i < std::min(lens1.size(), lens2.size()); i++) {
        if (lens1[i] < lens2[i]) return 1;
        else if (lens1[i] > lens2[i]) return 0;
    }
    return 0;
}
static void revertChainBlockStrand(std::vector<std::vector<UPair>> &cc, std::vector<u64> &cords, int strand) { // cluster_util.cpp:1023-1063
    u64 swap_str = 0;
    u64 f_strand = strand ? 1 : 0;
    for (unsigned i = 0; i < cc.size(); i++) {
        cc[i].push_back(UPair(0, 0));
        u64 strand_pre = 0, strand_this = 0;
        for (unsigned j = 0; j < cc[i].size(); j++) {
            if (j == cc[i].size() - 1 || get_cord_strand(cords[cc[i][j].first]) == f_strand) strand_this = 0;
            else strand_this = 1;
            if (strand_this && !strand_pre) swap_str = j;
            if (!strand_this && strand_pre)
                for (unsigned k = (unsigned)swap_str; k < (swap_str + j) / 2; k++) std::swap(cc[i][k], cc[i][swap_str + j - 1 - k]);
            strand_pre = strand_this;
        }
        cc[i].resize(cc[i].size() - 1);
    }
}
static void _filterBlocksCords(std::vector<std::vector<UPair>> &chains, std::vector<u64> &hits, u64 thd_major_limit, int f_header) { // cluster_util.cpp:865-931
    if (chains.empty()) return;
    std::vector<u64> hits_tmp;
    u64 len_current = 0;
    if (f_header) hits_tmp.push_back(hits[0]);
    for (unsigned i = 0; i < chains[0].size(); i++) {
        for (u64 j = chains[0][i].first; j < chains[0][i].second; j++) { hits_tmp.push_back(hits[j]); unset_block_end(hits_tmp.back()); }
        len_current += chains[0][i].second - chains[0][i].first;
    }
    set_block_end(hits_tmp.back());
    float thd_major_bound = 0.8 * len_current;
    unsigned major_n = 1;
    bool f_append = false;
    for (unsigned i = 1; i < chains.size() && major_n < thd_major_limit; i++) {
        len_current = 0;
        f_append = false;
        for (unsigned j = 0; j < chains[i].size(); j++) len_current += chains[i][j].second - chains[i][j].first;
        if (len_current > thd_major_bound) { f_append = true; ++major_n; }
        if (f_append) {
            for (unsigned j = 0; j < chains[i].size(); j++)
                for (u64 k = chains[i][j].first; k < chains[i][j].second; k++) { hits_tmp.push_back(hits[k]); unset_block_end(hits_tmp.back()); }
            set_block_end(hits_tmp.back());
        }
    }
    hits = hits_tmp;
}
static void chainBlocksCords(std::vector<u64> &cords, std::vector<UPair> &sep, u64 read_len, unsigned thd_init_cord_score, u64 thd_major_limit, int f_header) { // cluster_util.cpp:1068-1102
    std::vector<std::vector<UPair>> cc1, cc2;
    std::vector<UPair> sep1(sep), sep2(sep);
    chainBlocksSingleStrand(cords, sep1, cc1, 0, read_len, thd_init_cord_score);
    chainBlocksSingleStrand(cords, sep2, cc2, 1, read_len, thd_init_cord_score);
    int best_strand = getChainBlocksBestStrand(cc1, cc2);
    if (best_strand == 0) {
        sep = sep1;
        revertChainBlockStrand(cc1, cords, best_strand);
        _filterBlocksCords(cc1, cords, thd_major_limit, f_header);
    } else {
        sep = sep2;
        revertChainBlockStrand(cc2, cords, best_strand);
        _filterBlocksCords(cc2, cords, thd_major_limit, f_header);
    }
}

// --------------------------------------------------------------- apxMap ----
struct Work {         // per-thread mutable state (mapper.cpp:423-433)
    std::vector<u64> cords_str, cords_end;
    std::vector<UPair> apx_gaps;
    Stats stats;
    u64 pair_evals = 0;
    Debug dbg;
};
struct Ctx {          // shared, read-only after build
    std::vector<std::vector<uint8_t>> seqs;   // padded copies
    std::vector<u64> lens;
    unsigned T = 1;
    DIndex index;
    HIndex hindex;
    int index_type = 1;   // -i: 1 DIndex, 2 HIndex (mapper.cpp:200)
    std::vector<Feat> f2;
    Work w;           // default single-thread work area
};

static void getAnchorHitsChains(std::vector<u64> &anchors, std::vector<u64> &hits, std::vector<int> &hits_score, u64 read_len,
                                const Parms &pm, Work &c) {                           // pmpfinder.cpp:2506-2555 with parms of :2599-2606
    const u64 density = 1, accept_min = 2, thd_large_gap = 600;
    const unsigned err_bit = 2;
    binningFilter(anchors);
    filterAnchors1(anchors, density, accept_min, err_bit);
    if (c.dbg.on && c.dbg.pass == 0) c.dbg.filtered_anchors = anchors;
    std::vector<UPair> str_ends, str_ends_p;
    std::vector<int> sep_score;
    hits_score.clear();
    hits_score.push_back(0);
    chainAnchorsHits(anchors, hits, hits_score, pm, &c.pair_evals, &c.dbg);
    if (c.dbg.on && c.dbg.pass == 0) c.dbg.hits_chain = hits;
    gather_blocks_(hits, str_ends, str_ends_p, 1, hits.size(), read_len, thd_large_gap, 0, 0);
    preFilterChains2(hits, str_ends_p);
    sep_score.resize(str_ends_p.size());
    for (unsigned i = 0; i < str_ends_p.size(); i++) sep_score[i] = hits_score[str_ends_p[i].first] - hits_score[str_ends_p[i].second - 1];
    chainBlocksHits(hits, str_ends_p, sep_score, read_len);
    if (c.dbg.on && c.dbg.pass == 0) c.dbg.hits_blocks = hits;
}

static void apxMap_(const Ctx &cx, Work &c, const uint8_t *read, u64 read_len, std::vector<u64> &hits, const Feat f1[2], std::vector<u64> &cords,
                    u64 map_str, u64 map_end, const Parms &pm) {                      // pmpfinder.cpp:2632-2707
    hits.clear();
    std::vector<u64> anchors;
    anchors.push_back(0);        // anchors.init(1)
    initCords(hits);             // initHits
    std::vector<int> hits_score;
    u64 read_str = get_cord_y(map_str), read_end = get_cord_y(map_end);
    if (cx.index_type == 2) getHIndexMatchAll(cx.hindex, read, read_len, anchors, map_str, map_end, pm, c.stats);   // getIndexMatchAll pmpfinder.cpp:2576-2583
    else getDIndexMatchAll(cx.index, read, read_len, anchors, read_str, read_end, pm, c.stats);
    if (c.dbg.on && c.dbg.pass == 0) c.dbg.raw_anchors = anchors;
    getAnchorHitsChains(anchors, hits, hits_score, read_len, pm, c);
    // cords_info bookkeeping (pmpfinder.cpp:2655-2703) has no effect on cords; omitted.
    // path_dst alg 2 (pmpfinder.cpp:1447-1469)
    if (hits.size() >= 2) {
        _filterHits(hits, f1, cx.f2);
        if (c.dbg.on && c.dbg.pass == 0) c.dbg.hits_filtered = hits;
        path_dst_2(hits, f1, cx.f2, cords, read_str, read_end, read_len);
    }
    if (c.dbg.on && c.dbg.pass == 0) c.dbg.cords_path = cords;
    c.dbg.pass++;
}

static void apxMap(const Ctx &cx, Work &c, const uint8_t *read, u64 read_len) {                      // pmpfinder.cpp:2709-2804 (f_chain = 1) + mapper.cpp:438-447
    std::vector<u64> &cords_str = c.cords_str, &cords_end = c.cords_end;
    cords_str.clear(); cords_end.clear(); c.apx_gaps.clear();
    c.dbg.pass = 0;
    if (read_len <= 200) return;                                                      // mapper.cpp:430,440
    // reverse complement + read features (base.cpp:335-344, pmpfinder.cpp:556)
    std::vector<uint8_t> com(read_len + SEQ_PAD, 0);
    static const uint8_t cpl[5] = {3, 2, 1, 0, 4};
    for (u64 k = 0; k < read_len; k++) com[k] = cpl[read[read_len - k - 1]];
    Feat f1[2];
    createFeatures2_48(read, (i64)read_len, f1[0]);
    createFeatures2_48(com.data(), (i64)read_len, f1[1]);

    Parms pm;
    i64 thd_cord_size = P_windowSize;
    i64 thd_large_gap = 1000;
    i64 thd_drop_len = 2;
    float thd_reapx_max_gap_ratio = 0.7f;
    thd_drop_len = std::min(thd_drop_len, (i64)(read_len * 0.05 / thd_cord_size));
    std::vector<u64> hit;
    u64 map_str = 0ULL;
    u64 map_end = create_cord(MAX_CORD_ID, MAX_CORD_X, read_len, 0);
    apxMap_(cx, c, read, read_len, hit, f1, cords_str, map_str, map_end, pm);
    std::vector<UPair> str_ends, str_ends_p;
    clean_blocks_(cords_str, (u64)thd_drop_len, 50);
    gather_blocks_(cords_str, str_ends, str_ends_p, 1, cords_str.size(), read_len, (u64)thd_large_gap, (u64)thd_cord_size, 1);
    int gap_lens_sum = gather_gaps_y_(str_ends, c.apx_gaps, read_len, (u64)thd_large_gap);
    if (float(gap_lens_sum) / read_len >= thd_reapx_max_gap_ratio) {
        for (unsigned i = 0; i < c.apx_gaps.size(); i++) {
            UPair y = getUPForwardy(c.apx_gaps[i], read_len);
            pm.toggle(1);
            map_str = y.first;
            map_end = create_cord(MAX_CORD_ID, MAX_CORD_X, y.second, 0);
            apxMap_(cx, c, read, read_len, hit, f1, cords_str, map_str, map_end, pm);
            pm.toggle(0);
        }
        str_ends.clear();
        str_ends_p.clear();
        gather_blocks_(cords_str, str_ends, str_ends_p, 1, cords_str.size(), read_len, (u64)thd_large_gap, (u64)thd_cord_size, 1);
    }
    chainBlocksCords(cords_str, str_ends_p, read_len, 16, 2, 1);                      // chainApxCordsBlocks alg 2, pmpfinder.cpp:1761-1764
    clean_blocks_(cords_str, (u64)thd_drop_len, 50);
    cords_end.resize(cords_str.size());
    int seg = 0;
    u64 d = shift_cord(0ULL, thd_cord_size, thd_cord_size);
    for (unsigned i = 0; i < cords_str.size(); i++) {
        if (seg) cords_str[i] |= F_RECD; else cords_str[i] &= ~F_RECD;
        cords_str[i] |= F_MAIN;
        if (is_block_end(cords_str[i])) seg = 1 - seg;
        cords_end[i] = cords_str[i] + d;
    }
}

#include "lnr_gap.inc"

}  // namespace orc

// ================================================================= C API ====
using namespace orc;
extern "C" {

static void *orc_create_i(const uint8_t *const *seqs, const uint64_t *lens, uint32_t nseq, uint32_t T, int index_type);
void *orc_create(const uint8_t *const *seqs, const uint64_t *lens, uint32_t nseq, uint32_t T) { return orc_create_i(seqs, lens, nseq, T, 1); }
void *orc_create2(const uint8_t *const *seqs, const uint64_t *lens, uint32_t nseq, uint32_t T, int index_type) { return orc_create_i(seqs, lens, nseq, T, index_type); }
static void *orc_create_i(const uint8_t *const *seqs, const uint64_t *lens, uint32_t nseq, uint32_t T, int index_type) {
    Ctx *c = new Ctx();
    c->T = T ? T : 1;
    c->index_type = index_type;
    std::vector<const uint8_t *> ptrs;
    for (uint32_t i = 0; i < nseq; i++) {
        c->seqs.emplace_back(lens[i] + SEQ_PAD, 0);
        memcpy(c->seqs.back().data(), seqs[i], lens[i]);
        c->lens.push_back(lens[i]);
    }
    for (uint32_t i = 0; i < nseq; i++) ptrs.push_back(c->seqs[i].data());
    if (index_type == 2) createHIndex(ptrs, c->lens, c->hindex, c->T);
    else createDIndex(ptrs, c->lens, c->index, c->T);
    c->f2.resize(nseq);
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t i = 0; i < nseq; i++) createFeatures2_48_par(ptrs[i], (i64)lens[i], c->f2[i], c->T);
    return c;
}
void orc_destroy(void *h) { delete (Ctx *)h; }
uint64_t orc_dir_len(void *h) { return ((Ctx *)h)->index.dir.size(); }
uint64_t orc_hs_len(void *h) { return ((Ctx *)h)->index.hs.size(); }
const int32_t *orc_dir(void *h) { return ((Ctx *)h)->index.dir.data(); }
const uint64_t *orc_hs(void *h) { return ((Ctx *)h)->index.hs.data(); }
uint64_t orc_ysa_len(void *h) { return ((Ctx *)h)->hindex.ysa.size(); }
const uint64_t *orc_ysa(void *h) { return ((Ctx *)h)->hindex.ysa.data(); }
uint64_t orc_empty_dir(void *h) { return ((Ctx *)h)->hindex.emptyDir; }
uint64_t orc_fill_mismatch(void *h) { return ((Ctx *)h)->index.fill_mismatch; }
uint64_t orc_f2_len(void *h, uint32_t id) { return ((Ctx *)h)->f2[id].size(); }
const int32_t *orc_f2(void *h, uint32_t id) { return (const int32_t *)((Ctx *)h)->f2[id].data(); }

// read features (forward strand when strand==0, reverse complement otherwise); returns entry count, writes 3 ints per entry
uint64_t orc_read_features(const uint8_t *read, uint64_t len, int strand, int32_t *out, uint64_t cap) {
    std::vector<uint8_t> s(len + SEQ_PAD, 0);
    static const uint8_t cpl[5] = {3, 2, 1, 0, 4};
    if (strand) for (u64 k = 0; k < len; k++) s[k] = cpl[read[len - k - 1]];
    else memcpy(s.data(), read, len);
    Feat f;
    createFeatures2_48(s.data(), (i64)len, f);
    u64 n = std::min<u64>(f.size(), cap);
    memcpy(out, f.data(), n * 12);
    return f.size();
}

// seed lookup only (stage a7): returns number of anchors incl. the leading dummy 0
uint64_t orc_seed_lookup(void *h, const uint8_t *read, uint64_t len, uint64_t read_str, uint64_t read_end, int alpha,
                         uint64_t *out, uint64_t cap, uint64_t *stats4) {
    Ctx *c = (Ctx *)h;
    std::vector<uint8_t> s(len + SEQ_PAD, 0);
    memcpy(s.data(), read, len);
    std::vector<u64> set;
    set.push_back(0);
    Parms pm;
    pm.thd_alpha = alpha;
    Stats st;
    if (c->index_type == 2) getHIndexMatchAll(c->hindex, s.data(), len, set, read_str, create_cord(MAX_CORD_ID, MAX_CORD_X, read_end, 0), pm, st);   // map_str = y only, map_end as apxMap builds it
    else getDIndexMatchAll(c->index, s.data(), len, set, read_str, read_end, pm, st);
    u64 n = std::min<u64>(set.size(), cap);
    if (out) memcpy(out, set.data(), n * 8);
    if (stats4) { stats4[0] = st.samples; stats4[1] = st.lookups; stats4[2] = st.bucket_entries; stats4[3] = st.anchors; }
    return set.size();
}

// full per-read path: returns number of cords
uint64_t orc_map_read(void *h, const uint8_t *read, uint64_t len) {
    Ctx *c = (Ctx *)h;
    std::vector<uint8_t> s(len + SEQ_PAD, 0);
    memcpy(s.data(), read, len);
    apxMap(*c, c->w, s.data(), len);
    return c->w.cords_str.size();
}
void orc_get_cords(void *h, uint64_t *cords_str, uint64_t *cords_end) {
    Work *c = &((Ctx *)h)->w;
    if (!c->cords_str.empty()) {
        memcpy(cords_str, c->cords_str.data(), c->cords_str.size() * 8);
        memcpy(cords_end, c->cords_end.data(), c->cords_end.size() * 8);
    }
}
// diagnostic: histogram of the bucket length of every lookup getDIndexMatchAll makes for this read (hist[0..400])
void orc_lookup_hist(void *h, const uint8_t *read, uint64_t len, uint64_t *hist401) {
    Ctx *c = (Ctx *)h;
    std::vector<uint8_t> buf(len + SEQ_PAD, 0);
    memcpy(buf.data(), read, len);
    Shape shape;
    u64 dt = 0, xpre = ~0ULL;
    if (len < 2 * shape.span) return;
    hashInit(shape, buf.data());
    for (u64 k = shape.span; k < len - shape.span; k++) {   // pmpfinder.cpp:1870-1907 with thd_alpha 15
        hashNexth(shape, buf.data() + k);
        if (++dt == 15) {
            hashNextX(shape, buf.data() + k);
            if (shape.XValue != xpre) {
                u64 n = (u64)(c->index.dir[shape.XValue + 1] - c->index.dir[shape.XValue]);
                hist401[n > 400 ? 400 : n]++;
                xpre = shape.XValue;
            }
            dt = 0;
        }
    }
}
// apx_gaps of the last orc_map_read (apxMap's output for the gap re-mapper, pmpfinder.cpp:2744): pairs of cord words

// ---- unit hooks of the gap path restatement (lnr_gap.inc), mirrored by ref_gap_* in ref_harness.cpp
static uint64_t out_u64(const std::vector<u64> &v, uint64_t *out, uint64_t cap) { for (size_t i = 0; i < v.size() && i < cap; i++) out[i] = v[i]; return v.size(); }
static std::vector<uint8_t> padded(const uint8_t *p, uint64_t n) { std::vector<uint8_t> s(n + SEQ_PAD, 0); memcpy(s.data(), p, n); return s; }
uint64_t orc_gap_anchors(const uint8_t *g, uint64_t glen, const uint8_t *r, uint64_t rlen, uint64_t gap_str, uint64_t gap_end, int shape_len, int step1, int step2, int direction,
                         int64_t anchor_lower, int64_t anchor_upper, uint64_t rvcp_const, uint64_t *out, uint64_t cap) {
    auto a = padded(g, glen), b = padded(r, rlen);
    Seq s1{a.data(), glen}, s2{b.data(), rlen};
    GapParms gp; std::vector<u64> g_hs, anc;
    g_stream_(s1, s2, g_hs, gap_str, gap_end, (unsigned)shape_len, step1, step2);
    g_create_anchors_(g_hs, anc, shape_len, direction, anchor_lower, anchor_upper, rvcp_const, gap_str, gap_end, gp);
    return out_u64(anc, out, cap);
}
uint64_t orc_gap_anchor_pair(const uint8_t *g, uint64_t glen, const uint8_t *r, uint64_t rlen, uint64_t gs, uint64_t ge, int shape_len, int step1, int step2, uint64_t rvcp_const,
                             uint64_t gap_str1, uint64_t gap_end1, uint64_t gap_str2, uint64_t gap_end2, uint64_t *out1, uint64_t *n1, uint64_t *out2, uint64_t cap) {
    auto a = padded(g, glen), b = padded(r, rlen);
    Seq s1{a.data(), glen}, s2{b.data(), rlen};
    GapParms gp; std::vector<u64> g_hs, a1, a2;
    g_stream_(s1, s2, g_hs, gs, ge, (unsigned)shape_len, step1, step2);
    g_CreateExtendAnchorsPair_(g_hs, a1, a2, shape_len, rvcp_const, gap_str1, gap_end1, gap_str2, gap_end2, gp);
    *n1 = out_u64(a1, out1, cap);
    return out_u64(a2, out2, cap);
}
uint64_t orc_gap_canchors(const uint8_t *g, uint64_t glen, const uint8_t *r, uint64_t rlen, uint64_t s1s, uint64_t s1e, uint64_t s2s, uint64_t s2e, int step1, int step2, int shape_len,
                          int64_t anchor_lower, int64_t anchor_upper, uint64_t *out, uint64_t cap) {
    auto a = padded(g, glen), b = padded(r, rlen);
    Seq s1{a.data(), glen}, s2{b.data(), rlen};
    std::vector<u64> g_hs, anc;
    c_stream_(s1, g_hs, s1s, s1e, step1, shape_len, 0);
    c_stream_(s2, g_hs, s2s, s2e, step2, shape_len, 1);
    c_createAnchors2(g_hs, anc, (int)g_hs.size(), anchor_lower, anchor_upper);
    return out_u64(anc, out, cap);
}

// alt bit 0: the chain metrics mapGap_ swaps in around its extensions (gap.cpp:124-130); bit 1: thd_cts_major_limit 3 (the stream state "extended")
static void gp_alt(GapParms &gp, int alt) { if (alt & 1) { gp.chn1_min_len = 1; gp.chn1_abort = 0; gp.chn1_fn = 2; gp.chn2_abort = 0; gp.chn2_fn = 3; } if (alt & 2) gp.thd_cts_major_limit = 3; }
uint64_t orc_gap_chains(const uint64_t *anchors, uint64_t n, uint64_t read_len, int alt, int direction, uint64_t gap_str, uint64_t gap_end, int closest, uint64_t *out, uint64_t cap, int *pr) {
    std::vector<u64> a(anchors, anchors + n), tiles;
    GapParms gp; gp_alt(gp, alt); gp.direction = direction;
    g_CreateChainsFromAnchors_(a, tiles, read_len, gp);
    if (closest) { std::pair<int, int> r = getClosestExtensionChain_(tiles, gap_str, gap_end, closest == 2, gp); pr[0] = r.first; pr[1] = r.second; }
    return out_u64(tiles, out, cap);
}

uint64_t orc_gap_map(void *h, const uint8_t *read, uint64_t len, int which, uint64_t gs1, uint64_t ge1, uint64_t gs2, uint64_t ge2, int direction, int alt, uint64_t *out_str, uint64_t *out_end,
                     uint64_t *n2, uint64_t cap) {
    Ctx *c = (Ctx *)h;
    auto rd = padded(read, len);
    std::vector<uint8_t> com(len + SEQ_PAD, 0);
    static const uint8_t cpl[5] = {3, 2, 1, 0, 4};
    for (u64 k = 0; k < len; k++) com[k] = cpl[rd[len - k - 1]];
    Feat f1[2];
    createFeatures2_48(rd.data(), (i64)len, f1[0]);
    createFeatures2_48(com.data(), (i64)len, f1[1]);
    GapFeat F{f1, &c->f2};
    u64 id = get_cord_id(gs1);
    GapSeqs Q{Seq{c->seqs[id].data(), c->lens[id]}, Seq{rd.data(), len}, Seq{com.data(), len}};
    GapParms gp; gp_alt(gp, alt); gp.read_len = len; gp.ref_len = c->lens[id];
    std::vector<u64> ts1, te1, ts2, te2;
    if (which == 1) mapGeneric(Q, F, ts1, te1, gs1, ge1, gp);
    else if (which == 2) mapExtend(Q, F, ts1, te1, gs1, ge1, direction, gp);
    else mapExtends(Q, F, ts1, te1, ts2, te2, gs1, ge1, gs2, ge2, gp);
    u64 n1 = ts1.size();
    for (u64 i = 0; i < n1 && i < cap; i++) { out_str[i] = ts1[i]; out_end[i] = i < te1.size() ? te1[i] : 0; }
    *n2 = ts2.size();
    for (u64 i = 0; i < ts2.size() && n1 + i < cap; i++) { out_str[n1 + i] = ts2[i]; out_end[n1 + i] = i < te2.size() ? te2[i] : 0; }
    return n1 | ((u64)te1.size() << 32);
}

// apxMap + mapGaps + reformCords as Mapper::p_calRecords runs them for -g gap_len [-dup] (mapper.cpp:207-231,438-453).
// THE STREAM STATE.  The Mapper keeps ONE GapParms per thread for the whole run (mapper.cpp:233-237, used :447) and mapExtend / mapExtends leave
// fields of it changed (gap_util.cpp:4046-4071,4088-4119).  Of those only thd_cts_major_limit is ever read before it is written again:
// `direction` is set by every caller of getClosestExtensionChain_ (gap_util.cpp:3667,3671,3935), f_gmsa_direction is never read, the two
// thd_ctfas2_* values are re-set to their defaults.  thd_cts_major_limit is 1 until the first mapExtend / mapExtends of the thread's read
// stream and 3 from then on (read by chainTiles, :1188, under mapGeneric).  `*ext_state` carries that bit in and out: 0 = nothing extended
// yet, 1 = extended.  With one thread (`-t 1`, file order) this is the program's result; with more threads the reference itself is not
// reproducible (which reads a thread meets before its first extension depends on the schedule).
static void gap_parms_for(GapParms &gp, uint32_t gap_len, int f_dup, int ext_state) {
    gp.f_dup = f_dup;
    gp.thd_gap_len_min = gap_len == 1 ? 50 : (gap_len < 10 ? 10 : gap_len);
    if (ext_state) gp.thd_cts_major_limit = 3;
}
uint64_t orc_map_read_g2(void *h, const uint8_t *read, uint64_t len, uint32_t gap_len, int f_dup, int *ext_state) {
    Ctx *c = (Ctx *)h;
    auto rd = padded(read, len);
    apxMap(*c, c->w, rd.data(), len);
    if (len <= 200 || gap_len == 0) return c->w.cords_str.size();
    std::vector<uint8_t> com(len + SEQ_PAD, 0);
    static const uint8_t cpl[5] = {3, 2, 1, 0, 4};
    for (u64 k = 0; k < len; k++) com[k] = cpl[rd[len - k - 1]];
    Feat f1[2];
    createFeatures2_48(rd.data(), (i64)len, f1[0]);
    createFeatures2_48(com.data(), (i64)len, f1[1]);
    GapFeat F{f1, &c->f2};
    GapGenome G{&c->seqs, &c->lens};
    GapParms gp;
    gap_parms_for(gp, gap_len, f_dup, ext_state ? *ext_state : 0);
    mapGaps(G, Seq{rd.data(), len}, Seq{com.data(), len}, c->w.cords_str, c->w.cords_end, c->w.apx_gaps, F, gp);
    reformCords(c->w.cords_str, c->w.cords_end);
    if (ext_state) *ext_state = gp.thd_cts_major_limit == 3;
    return c->w.cords_str.size();
}
uint64_t orc_map_read_g(void *h, const uint8_t *read, uint64_t len, uint32_t gap_len, int f_dup) { return orc_map_read_g2(h, read, len, gap_len, f_dup, nullptr); }
// the same with the trace of every mapExtend / mapExtends / mapGeneric call of mapGap_ (layout: gap_trace_call in lnr_gap.inc); returns the words of the trace
uint64_t orc_map_read_g_trace(void *h, const uint8_t *read, uint64_t len, uint32_t gap_len, int f_dup, int ext_state, uint64_t *out, uint64_t cap) {
    std::vector<u64> t;
    gap_trace = &t;
    orc_map_read_g2(h, read, len, gap_len, f_dup, &ext_state);
    gap_trace = nullptr;
    for (size_t i = 0; i < t.size() && i < cap; i++) out[i] = t[i];
    return t.size();
}
int orc_gap_score(int which, uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t read_len, int strand) {
    switch (which) {
        case 1: return getGapAnchorsChainScore(a, b);
        case 2: return getGapAnchorsChainScore2(a, b);
        case 3: return getGapBlocksChainScore2(a, b, c, d, read_len, strand);
        default: return getGapBlocksChainScore3(a, b, c, d, read_len, strand);
    }
}
uint64_t orc_gap_xdrop(uint64_t *chain, uint64_t n, int direction, int f_erase, int *ret) {
    std::vector<u64> c(chain, chain + n);
    GapParms gp;
    *ret = dropChainGapX(c, ganc_x, ganc_y, direction, f_erase != 0, gp);
    for (size_t i = 0; i < c.size(); i++) chain[i] = c[i];
    return c.size();
}
uint64_t orc_get_gaps(void *h, uint64_t *out_pairs, uint64_t cap_pairs) {
    Work &w = ((Ctx *)h)->w;
    for (size_t i = 0; i < w.apx_gaps.size() && i < cap_pairs; i++) { out_pairs[2 * i] = w.apx_gaps[i].first; out_pairs[2 * i + 1] = w.apx_gaps[i].second; }
    return w.apx_gaps.size();
}
void orc_reset_stats(void *h) { Work *c = &((Ctx *)h)->w; c->stats = Stats(); c->pair_evals = 0; }
void orc_get_stats(void *h, uint64_t *out5) {
    Work *c = &((Ctx *)h)->w;
    out5[0] = c->stats.samples; out5[1] = c->stats.lookups; out5[2] = c->stats.bucket_entries; out5[3] = c->stats.anchors; out5[4] = c->pair_evals;
}
void orc_debug(void *h, int on) { ((Ctx *)h)->w.dbg.on = on != 0; }
// stage: 0 raw anchors, 1 filtered anchors, 2 x-desc sorted anchors, 3 hits after anchor chaining, 4 hits after block chaining,
//        5 hits after window filter, 6 cords after path_dst   (all from the first apxMap_ pass of the last orc_map_read)
uint64_t orc_debug_get(void *h, int stage, uint64_t *out, uint64_t cap) {
    Work *c = &((Ctx *)h)->w;
    std::vector<u64> *v = nullptr;
    switch (stage) {
        case 0: v = &c->dbg.raw_anchors; break;
        case 1: v = &c->dbg.filtered_anchors; break;
        case 2: v = &c->dbg.sorted_anchors; break;
        case 3: v = &c->dbg.hits_chain; break;
        case 4: v = &c->dbg.hits_blocks; break;
        case 5: v = &c->dbg.hits_filtered; break;
        case 6: v = &c->dbg.cords_path; break;
        default: return 0;
    }
    u64 n = std::min<u64>(v->size(), cap);
    if (out && n) memcpy(out, v->data(), n * 8);
    return v->size();
}

// batch helper: maps reads [0,n) with `threads` OpenMP threads (per-thread Work), writes CSR in read order.
// stats5 (optional) accumulates samples, lookups, bucket entries, anchors, DP pair evaluations over the batch.
uint64_t orc_map_batch_g(void *h, const uint8_t *reads, const uint64_t *off, uint32_t n, int threads, uint64_t *cord_off,
                         uint64_t *cords_str, uint64_t *cords_end, uint64_t cap, uint64_t *stats5, uint32_t gap_len, int f_dup);
uint64_t orc_map_batch_g2(void *h, const uint8_t *reads, const uint64_t *off, uint32_t n, int threads, uint64_t *cord_off,
                          uint64_t *cords_str, uint64_t *cords_end, uint64_t cap, uint64_t *stats5, uint32_t gap_len, int f_dup, int *ext_state);
uint64_t orc_map_batch(void *h, const uint8_t *reads, const uint64_t *off, uint32_t n, int threads, uint64_t *cord_off,
                       uint64_t *cords_str, uint64_t *cords_end, uint64_t cap, uint64_t *stats5) {
    return orc_map_batch_g(h, reads, off, n, threads, cord_off, cords_str, cords_end, cap, stats5, 0, 0);
}
uint64_t orc_map_batch_g(void *h, const uint8_t *reads, const uint64_t *off, uint32_t n, int threads, uint64_t *cord_off,
                         uint64_t *cords_str, uint64_t *cords_end, uint64_t cap, uint64_t *stats5, uint32_t gap_len, int f_dup) {
    int st = 0;     // a fresh stream: the batch is the whole read file
    return orc_map_batch_g2(h, reads, off, n, threads, cord_off, cords_str, cords_end, cap, stats5, gap_len, f_dup, &st);
}
// The batch in FILE ORDER as one thread of the reference would meet it (`-t 1`), on `threads` host threads: apxMap of every read in
// parallel; then the gap re-mapper read by read until the stream state flips (see orc_map_read_g2) and in parallel from there on -- every
// later read starts from the flipped state whatever the schedule.  *ext_state in / out.
uint64_t orc_map_batch_g2(void *h, const uint8_t *reads, const uint64_t *off, uint32_t n, int threads, uint64_t *cord_off,
                          uint64_t *cords_str, uint64_t *cords_end, uint64_t cap, uint64_t *stats5, uint32_t gap_len, int f_dup, int *ext_state) {
    Ctx *c = (Ctx *)h;
    if (threads < 1) threads = 1;
    std::vector<std::vector<u64>> rs(n), re(n);
    std::vector<std::vector<UPair>> rg(gap_len ? n : 0);
    std::vector<Work> works(threads);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 16)
    for (uint32_t i = 0; i < n; i++) {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        Work &w = works[tid];
        u64 len = off[i + 1] - off[i];
        std::vector<uint8_t> s(len + SEQ_PAD, 0);
        memcpy(s.data(), reads + off[i], len);
        apxMap(*c, w, s.data(), len);
        rs[i] = w.cords_str;
        re[i] = w.cords_end;
        if (gap_len) rg[i] = w.apx_gaps;
    }
    if (gap_len) {
        auto gap_read = [&](uint32_t i, int entry) -> int {
            u64 len = off[i + 1] - off[i];
            if (len <= 200) return entry;
            std::vector<uint8_t> s(len + SEQ_PAD, 0), com(len + SEQ_PAD, 0);
            memcpy(s.data(), reads + off[i], len);
            static const uint8_t cpl[5] = {3, 2, 1, 0, 4};
            for (u64 k = 0; k < len; k++) com[k] = cpl[s[len - k - 1]];
            Feat f1[2];
            createFeatures2_48(s.data(), (i64)len, f1[0]);
            createFeatures2_48(com.data(), (i64)len, f1[1]);
            GapFeat F{f1, &c->f2};
            GapGenome G{&c->seqs, &c->lens};
            GapParms gp;
            gap_parms_for(gp, gap_len, f_dup, entry);
            mapGaps(G, Seq{s.data(), len}, Seq{com.data(), len}, rs[i], re[i], rg[i], F, gp);
            reformCords(rs[i], re[i]);
            return gp.thd_cts_major_limit == 3;
        };
        uint32_t i0 = 0;
        int st = ext_state ? *ext_state : 0;
        while (i0 < n && !st) st = gap_read(i0++, 0);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 16)
        for (uint32_t i = i0; i < n; i++) gap_read(i, 1);
        if (ext_state) *ext_state = st;
    }
    u64 tot = 0;
    cord_off[0] = 0;
    for (uint32_t i = 0; i < n; i++) {
        u64 m = rs[i].size();
        if (tot + m <= cap && cords_str && m) {
            memcpy(cords_str + tot, rs[i].data(), m * 8);
            memcpy(cords_end + tot, re[i].data(), m * 8);
        }
        tot += m;
        cord_off[i + 1] = tot;
    }
    if (stats5) {
        for (auto &w : works) {
            stats5[0] += w.stats.samples; stats5[1] += w.stats.lookups; stats5[2] += w.stats.bucket_entries;
            stats5[3] += w.stats.anchors; stats5[4] += w.pair_evals;
        }
    }
    return tot;
}
}  // extern "C"
