// wc_argsort.hpp -- the element order GNU libstdc++'s std::sort produces, reproduced step by step.
//
// Harvest's mergeF0 (reference src/harvest.cpp:508-513) argsorts the voiced sections by start frame with std::sort,
// which is not stable, and the loop that follows skips order[0] (:517): when two sections start at the same frame
// (backward extension lets more than one reach frame 0) the contour depends on which of them the sort leaves first.
// For more than 16 sections that is decided by the introsort partitioning of libstdc++, the library the reference is
// built with (gcc), so the device code walks the same algorithm: introsort loop with median-of-three pivot and
// unguarded partition down to runs of 16, heap sort at the depth limit, then the final (un)guarded insertion sort
// (bits/stl_algo.h: __introsort_loop, __unguarded_partition_pivot, __final_insertion_sort; bits/stl_heap.h:
// __make_heap, __adjust_heap, __pop_heap).  tests/cpp/argsort_check.cpp compares it with std::sort itself on
// tie-heavy keys.
#pragma once

#if defined(__HIPCC__)
#define WC_ARGSORT_FN __host__ __device__ inline
#else
#define WC_ARGSORT_FN inline
#endif

namespace wc_argsort {

struct ByKey {
	const int *key;
	int stride;
	WC_ARGSORT_FN bool operator()(int a, int b) const { return key[a * stride] < key[b * stride]; }
};

template <class C> WC_ARGSORT_FN void unguarded_linear_insert(int *last, C less) {
	const int val = *last;
	int *next = last - 1;
	while (less(val, *next)) { *last = *next; last = next; --next; }
	*last = val;
}
template <class C> WC_ARGSORT_FN void insertion_sort(int *first, int *last, C less) {
	if (first == last) return;
	for (int *i = first + 1; i != last; ++i) {
		if (less(*i, *first)) {
			const int val = *i;
			for (int *p = i; p != first; --p) *p = *(p - 1);
			*first = val;
		} else {
			unguarded_linear_insert(i, less);
		}
	}
}
template <class C> WC_ARGSORT_FN void adjust_heap(int *first, long hole, long len, int value, C less) {
	const long top = hole;
	long child = hole;
	while (child < (len - 1) / 2) {
		child = 2 * (child + 1);
		if (less(first[child], first[child - 1])) child--;
		first[hole] = first[child];
		hole = child;
	}
	if ((len & 1) == 0 && child == (len - 2) / 2) {
		child = 2 * (child + 1);
		first[hole] = first[child - 1];
		hole = child - 1;
	}
	long parent = (hole - 1) / 2;  // __push_heap
	while (hole > top && less(first[parent], value)) {
		first[hole] = first[parent];
		hole = parent;
		parent = (hole - 1) / 2;
	}
	first[hole] = value;
}
template <class C> WC_ARGSORT_FN void heap_sort(int *first, int *last, C less) {  // __partial_sort(first, last, last)
	const long len = last - first;
	if (len >= 2) {
		for (long parent = (len - 2) / 2;; --parent) {
			adjust_heap(first, parent, len, first[parent], less);
			if (parent == 0) break;
		}
	}
	while (last - first > 1) {
		--last;
		const int value = *last;
		*last = *first;
		adjust_heap(first, 0, last - first, value, less);
	}
}
template <class C> WC_ARGSORT_FN void move_median_to_first(int *result, int *a, int *b, int *c, C less) {
	int *pick;
	if (less(*a, *b)) {
		if (less(*b, *c)) pick = b;
		else if (less(*a, *c)) pick = c;
		else pick = a;
	} else if (less(*a, *c)) pick = a;
	else if (less(*b, *c)) pick = c;
	else pick = b;
	const int t = *result; *result = *pick; *pick = t;
}
template <class C> WC_ARGSORT_FN int *unguarded_partition(int *first, int *last, int *pivot, C less) {
	for (;;) {
		while (less(*first, *pivot)) ++first;
		--last;
		while (less(*pivot, *last)) --last;
		if (!(first < last)) return first;
		const int t = *first; *first = *last; *last = t;
		++first;
	}
}

// std::sort(order, order + n, less) of libstdc++; the recursion of __introsort_loop (always on the right part,
// at most 2 * lg(n) deep) is unrolled onto a small explicit stack
template <class C> WC_ARGSORT_FN void sort_like_libstdcxx(int *order, int n, C less) {
	if (n <= 0) return;
	int lg = 0;
	while ((1L << (lg + 1)) <= n) ++lg;
	struct Range { int *first, *last; int depth; };
	Range stack[64];
	int sp = 0;
	stack[sp++] = Range{order, order + n, 2 * lg};
	while (sp > 0) {
		Range r = stack[--sp];
		// one activation of __introsort_loop(first, last, depth); right-hand recursions run first, as there
		while (r.last - r.first > 16) {
			if (r.depth == 0) { heap_sort(r.first, r.last, less); break; }
			--r.depth;
			int *mid = r.first + (r.last - r.first) / 2;
			move_median_to_first(r.first, r.first + 1, mid, r.last - 1, less);
			int *cut = unguarded_partition(r.first + 1, r.last, r.first, less);
			// recursion on [cut, last) happens before the loop continues with [first, cut); the two ranges are
			// disjoint, so running the left one later from the stack gives the same array
			stack[sp++] = Range{r.first, cut, r.depth};
			r.first = cut;
		}
	}
	if (n > 16) {  // __final_insertion_sort
		insertion_sort(order, order + 16, less);
		for (int *i = order + 16; i != order + n; ++i) unguarded_linear_insert(i, less);
	} else {
		insertion_sort(order, order + n, less);
	}
}

}  // namespace wc_argsort
