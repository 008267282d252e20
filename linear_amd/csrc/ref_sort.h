// ref_sort.h -- bit-exact re-implementation of libstdc++'s std::sort (introsort) that
// runs on the device.
//
// Why it exists: several sorts on the reference's hot path use comparators that are not
// total orders (ties): anchors by genome x only (reference src/pmpfinder.cpp:2465), cut
// lists (:2384), gap lists (:1610), tree ranks (src/cluster_util.cpp:269), block order
// (:558, :945, :956).  The reference's results depend on the permutation libstdc++'s
// introsort happens to produce for tied keys, and swapping in a stable sort changes the
// output.  To emit the same cords on the GPU we run the same
// algorithm: median-of-3 introsort with threshold 16, depth limit 2*floor(log2 n),
// heapsort fallback, final (un)guarded insertion sort -- as published in
// libstdc++ 11 bits/stl_algo.h (std::__sort) and bits/stl_heap.h.
//
// Recursion is replaced by an explicit stack (sub-ranges are disjoint, so processing
// order does not change the result).  tests/test_stage_logic_host.py::test_ref_sort_equals_libstdcxx_sort_with_ties fuzzes this against
// std::sort with heavy ties on the CPU build.
#pragma once
#include <stdint.h>

#ifndef LNR_HD
#if defined(__HIPCC__)
#define LNR_HD __host__ __device__
#else
#define LNR_HD
#endif
#endif

namespace lnr {

template <class T>
LNR_HD inline void rs_swap(T &a, T &b) { T t = a; a = b; b = t; }

// std::__unguarded_linear_insert
template <class T, class Comp>
LNR_HD inline void rs_unguarded_linear_insert(T *a, long last, Comp comp) {
    T val = a[last];
    long next = last - 1;
    while (comp(val, a[next])) {
        a[last] = a[next];
        last = next;
        --next;
    }
    a[last] = val;
}
// std::__insertion_sort on [first,last)
template <class T, class Comp>
LNR_HD inline void rs_insertion_sort(T *a, long first, long last, Comp comp) {
    if (first == last) return;
    for (long i = first + 1; i != last; ++i) {
        if (comp(a[i], a[first])) {
            T val = a[i];
            for (long k = i; k > first; --k) a[k] = a[k - 1];   // move_backward(first, i, i+1)
            a[first] = val;
        } else
            rs_unguarded_linear_insert(a, i, comp);
    }
}
// std::__adjust_heap (+ __push_heap) on the heap that starts at a[first]
template <class T, class Comp>
LNR_HD inline void rs_adjust_heap(T *a, long first, long holeIndex, long len, T value, Comp comp) {
    const long topIndex = holeIndex;
    long secondChild = holeIndex;
    while (secondChild < (len - 1) / 2) {
        secondChild = 2 * (secondChild + 1);
        if (comp(a[first + secondChild], a[first + (secondChild - 1)])) secondChild--;
        a[first + holeIndex] = a[first + secondChild];
        holeIndex = secondChild;
    }
    if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
        secondChild = 2 * (secondChild + 1);
        a[first + holeIndex] = a[first + (secondChild - 1)];
        holeIndex = secondChild - 1;
    }
    long parent = (holeIndex - 1) / 2;
    while (holeIndex > topIndex && comp(a[first + parent], value)) {
        a[first + holeIndex] = a[first + parent];
        holeIndex = parent;
        parent = (holeIndex - 1) / 2;
    }
    a[first + holeIndex] = value;
}
// std::__partial_sort(first, last, last) == heap sort of the whole range
template <class T, class Comp>
LNR_HD inline void rs_heap_sort(T *a, long first, long last, Comp comp) {
    long len = last - first;
    if (len >= 2) {   // __make_heap
        long parent = (len - 2) / 2;
        while (true) {
            T value = a[first + parent];
            rs_adjust_heap(a, first, parent, len, value, comp);
            if (parent == 0) break;
            parent--;
        }
    }
    // __heap_select's scan over [middle,last) is empty (middle == last)
    while (last - first > 1) {   // __sort_heap -> __pop_heap(first, last, last)
        --last;
        T value = a[last];
        a[last] = a[first];
        rs_adjust_heap(a, first, 0, last - first, value, comp);
    }
}

// Explicit stack of the introsort loop.  Depth <= 2*lg(n) + 1 entries; callers on the GPU place it in LDS
// (one per wave) instead of per-lane private memory, which would cost occupancy.
struct SortStack { int first[64], last[64], depth[64]; };   // 2 lg n + 1 entries: enough below 2^31 elements

// std::sort(a, a+n, comp)
template <class T, class Comp>
LNR_HD inline void ref_sort(T *a, long n, Comp comp, SortStack &st) {
    if (n <= 0) return;
    int *stk_first = st.first, *stk_last = st.last, *stk_depth = st.depth;
    int sp = 0;
    int lg = 0;
    for (long t = n; t > 1; t >>= 1) lg++;
    stk_first[0] = 0; stk_last[0] = n; stk_depth[0] = lg * 2; sp = 1;
    while (sp > 0) {
        --sp;
        long first = stk_first[sp], last = stk_last[sp];
        int depth_limit = stk_depth[sp];
        while (last - first > 16) {
            if (depth_limit == 0) { rs_heap_sort(a, first, last, comp); break; }
            --depth_limit;
            // __unguarded_partition_pivot
            long mid = first + (last - first) / 2;
            {   // __move_median_to_first(first, first+1, mid, last-1)
                long A = first + 1, B = mid, C = last - 1;
                if (comp(a[A], a[B])) {
                    if (comp(a[B], a[C])) rs_swap(a[first], a[B]);
                    else if (comp(a[A], a[C])) rs_swap(a[first], a[C]);
                    else rs_swap(a[first], a[A]);
                } else if (comp(a[A], a[C])) rs_swap(a[first], a[A]);
                else if (comp(a[B], a[C])) rs_swap(a[first], a[C]);
                else rs_swap(a[first], a[B]);
            }
            long lo = first + 1, hi = last;
            while (true) {   // __unguarded_partition(first+1, last, pivot=first)
                while (comp(a[lo], a[first])) ++lo;
                --hi;
                while (comp(a[first], a[hi])) --hi;
                if (!(lo < hi)) break;
                rs_swap(a[lo], a[hi]);
                ++lo;
            }
            long cut = lo;
            // recurse on [cut,last), continue with [first,cut)
            stk_first[sp] = (int)cut; stk_last[sp] = (int)last; stk_depth[sp] = depth_limit; ++sp;
            last = cut;
        }
    }
    // __final_insertion_sort
    if (n > 16) {
        rs_insertion_sort(a, 0, 16, comp);
        for (long i = 16; i != n; ++i) rs_unguarded_linear_insert(a, i, comp);
    } else
        rs_insertion_sort(a, 0, n, comp);
}

}  // namespace lnr

// ---------------------------------------------------------------------------------------------------
// List formulation of std::__unguarded_partition, the form the GPU kernel executes with all lanes
// (lnr_kernels.hip: introsort_wave).  For the range [lo, hi) and pivot p (sitting at lo-1):
//   L = positions, ascending,  of elements x with !comp(x, p)      (where the left scan stops)
//   R = positions, descending, of elements x with !comp(p, x)      (where the right scan stops)
// The serial algorithm swaps the pairs (L_k, R_k) for k = 0..K-1, K = number of leading pairs with
// L_k < R_k (L grows, R shrinks, so the predicate is monotone), and returns
//   cut = min(L_K if it exists, R_{K-1} if K >= 1)
// because positions inside (L_{k-1}, R_{k-1}) are untouched when step k scans them, and R_{K-1} holds an
// element >= p after its swap.  An element equal to the pivot is in both lists but can never be the left
// member of one swapped pair and the right member of another.  ref_sort_model() is ref_sort() with this
// partition and with the kernel's task split (large ranges partitioned by lists, ranges <= small_limit
// finished independently: introsort loop + insertion sort of the range); tests fuzz it against std::sort.
namespace lnr {

template <class T, class Comp>
LNR_HD inline long rs_partition_lists(T *a, long lo, long hi, const T &p, Comp comp, long *Lbuf, long *Rbuf) {
    long nL = 0, nR = 0;
    for (long i = lo; i < hi; i++) { if (!comp(a[i], p)) Lbuf[nL++] = i; if (!comp(p, a[i])) Rbuf[nR++] = i; }
    // R_k = Rbuf[nR-1-k]
    long K = 0;
    long lim = nL < nR ? nL : nR;
    while (K < lim && Lbuf[K] < Rbuf[nR - 1 - K]) K++;
    for (long k = 0; k < K; k++) rs_swap(a[Lbuf[k]], a[Rbuf[nR - 1 - k]]);
    long cut = hi;   // cannot stay hi: the median-of-3 guarantees a stop inside the range
    if (K < nL) cut = Lbuf[K];
    if (K >= 1 && Rbuf[nR - K] < cut) cut = Rbuf[nR - K];
    return cut;
}
// serial introsort loop on [first,last) followed by the insertion sort of that range
// STK = stack entries: a range of s elements needs at most min(depth, s - 16) entries.
template <int STK, class T, class Comp>
LNR_HD inline void rs_finish_range(T *a, long first0, long last0, int depth0, Comp comp) {
    int stk_first[STK], stk_last[STK];
    int stk_depth[STK];
    int sp = 0;
    stk_first[0] = first0; stk_last[0] = last0; stk_depth[0] = depth0; sp = 1;
    while (sp > 0) {
        --sp;
        long first = stk_first[sp], last = stk_last[sp];
        int depth_limit = stk_depth[sp];
        while (last - first > 16) {
            if (depth_limit == 0) { rs_heap_sort(a, first, last, comp); break; }
            --depth_limit;
            long mid = first + (last - first) / 2;
            long A = first + 1, B = mid, C = last - 1;
            if (comp(a[A], a[B])) {
                if (comp(a[B], a[C])) rs_swap(a[first], a[B]);
                else if (comp(a[A], a[C])) rs_swap(a[first], a[C]);
                else rs_swap(a[first], a[A]);
            } else if (comp(a[A], a[C])) rs_swap(a[first], a[A]);
            else if (comp(a[B], a[C])) rs_swap(a[first], a[C]);
            else rs_swap(a[first], a[B]);
            long lo = first + 1, hi = last;
            while (true) {
                while (comp(a[lo], a[first])) ++lo;
                --hi;
                while (comp(a[first], a[hi])) --hi;
                if (!(lo < hi)) break;
                rs_swap(a[lo], a[hi]);
                ++lo;
            }
            stk_first[sp] = (int)lo; stk_last[sp] = (int)last; stk_depth[sp] = depth_limit; ++sp;
            last = lo;
        }
    }
    rs_insertion_sort(a, first0, last0, comp);
}
template <class T, class Comp>
LNR_HD inline void ref_sort_model(T *a, long n, Comp comp, long small_limit, long *Lbuf, long *Rbuf) {
    if (n <= 0) return;
    long stk_first[96], stk_last[96];
    int stk_depth[96];
    int sp = 0, lg = 0;
    for (long t = n; t > 1; t >>= 1) lg++;
    stk_first[0] = 0; stk_last[0] = n; stk_depth[0] = lg * 2; sp = 1;
    while (sp > 0) {
        --sp;
        long first = stk_first[sp], last = stk_last[sp];
        int depth_limit = stk_depth[sp];
        while (true) {
            if (last - first <= small_limit) { rs_finish_range<96>(a, first, last, depth_limit, comp); break; }
            if (depth_limit == 0) { rs_heap_sort(a, first, last, comp); break; }
            --depth_limit;
            long mid = first + (last - first) / 2;
            long A = first + 1, B = mid, C = last - 1;
            if (comp(a[A], a[B])) {
                if (comp(a[B], a[C])) rs_swap(a[first], a[B]);
                else if (comp(a[A], a[C])) rs_swap(a[first], a[C]);
                else rs_swap(a[first], a[A]);
            } else if (comp(a[A], a[C])) rs_swap(a[first], a[A]);
            else if (comp(a[B], a[C])) rs_swap(a[first], a[C]);
            else rs_swap(a[first], a[B]);
            T p = a[first];
            long cut = rs_partition_lists(a, first + 1, last, p, comp, Lbuf, Rbuf);
            stk_first[sp] = cut; stk_last[sp] = last; stk_depth[sp] = depth_limit; ++sp;
            last = cut;
        }
    }
}

}  // namespace lnr
