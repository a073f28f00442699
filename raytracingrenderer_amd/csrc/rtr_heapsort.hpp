// The sequential pieces of libstdc++'s std::sort (bits/stl_algo.h, bits/stl_heap.h), operation by
// operation, over any record store with get(i) / set(i, record): the median-of-three move of
// __unguarded_partition_pivot and the heap sort (__partial_sort over the whole range) that
// __introsort_loop falls back to when its depth budget is spent.  Heap sort is not stable, so the
// order it leaves equal keys in depends on every single move; the tree builder needs that order.
// Compiles for the host too (tools/introsort_model.cpp checks it against std::partial_sort).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define RTR_SORT_FN __host__ __device__ inline
#else
#define RTR_SORT_FN inline
#endif

namespace rtr {

struct SortRec {
  float key;
  uint32_t id;
};

// std::__move_median_to_first(result, a, b, c) with iter_swap spelled out
template <class A>
RTR_SORT_FN void median_to_first(const A &v, int result, int a, int b, int c) {
  const SortRec ra = v.get(a), rb = v.get(b), rc = v.get(c), rr = v.get(result);
  int pick;
  if (ra.key < rb.key) {
    if (rb.key < rc.key) pick = 1;
    else if (ra.key < rc.key) pick = 2;
    else pick = 0;
  } else if (ra.key < rc.key) pick = 0;
  else if (rb.key < rc.key) pick = 2;
  else pick = 1;
  const int at = pick == 0 ? a : (pick == 1 ? b : c);
  const SortRec chosen = pick == 0 ? ra : (pick == 1 ? rb : rc);
  v.set(result, chosen);
  v.set(at, rr);
}

// std::__adjust_heap followed by its std::__push_heap
template <class A>
RTR_SORT_FN void adjust_heap(const A &v, int first, int hole, int len, SortRec value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (v.get(first + child).key < v.get(first + child - 1).key) child--;
    v.set(first + hole, v.get(first + child));
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    v.set(first + hole, v.get(first + child - 1));
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && v.get(first + parent).key < value.key) {
    v.set(first + hole, v.get(first + parent));
    hole = parent;
    parent = (hole - 1) / 2;
  }
  v.set(first + hole, value);
}

// std::__partial_sort(first, last, last): __heap_select (here only its __make_heap; the loop over
// [middle, last) is empty) and __sort_heap
template <class A>
RTR_SORT_FN void heap_sort(const A &v, int first, int last) {
  const int len = last - first;
  if (len >= 2) {
    int parent = (len - 2) / 2;
    while (true) {
      adjust_heap(v, first, parent, len, v.get(first + parent));
      if (parent == 0) break;
      parent--;
    }
  }
  while (last - first > 1) {
    --last;
    const SortRec value = v.get(last);
    v.set(last, v.get(first));
    adjust_heap(v, first, 0, last - first, value);
  }
}

}  // namespace rtr
