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
// order does not change the result).  tests/test_ref_sort.py fuzzes this against
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

// std::sort(a, a+n, comp)
template <class T, class Comp>
LNR_HD inline void ref_sort(T *a, long n, Comp comp) {
    if (n <= 0) return;
    // __introsort_loop with an explicit stack; depth <= 2*lg(n) entries suffice
    long stk_first[96], stk_last[96];
    int stk_depth[96];
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
            stk_first[sp] = cut; stk_last[sp] = last; stk_depth[sp] = depth_limit; ++sp;
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
