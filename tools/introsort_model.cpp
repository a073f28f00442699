// Model of libstdc++'s std::sort (bits/stl_algo.h: __introsort_loop, __unguarded_partition_pivot,
// __final_insertion_sort) in the data-parallel form the device tree builder uses, checked against
// std::sort itself on records whose keys tie.  Every step below is written as "what each element
// can compute from scans", executed sequentially here.
//
//   g++ -O2 -std=c++17 -o /tmp/introsort_model tools/introsort_model.cpp && /tmp/introsort_model [trials]
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../raytracingrenderer_amd/csrc/rtr_heapsort.hpp"

struct Rec {
  float key;
  int id;
};
static bool less(const Rec &a, const Rec &b) { return a.key < b.key; }

struct Seg {
  int f, l, depth;
};

// one Hoare partition of [f+1, l) around the key at f, by ranks
static int partition_by_ranks(std::vector<Rec> &a, int f, int l) {
  const float p = a[f].key;
  std::vector<int> PL, PR;
  for (int i = f + 1; i < l; i++)
    if (!(a[i].key < p)) PL.push_back(i);
  for (int i = l - 1; i > f; i--)
    if (!(p < a[i].key)) PR.push_back(i);
  size_t K = 0;
  while (K < PL.size() && K < PR.size() && PL[K] < PR[K]) K++;
  std::vector<Rec> b = a;
  for (size_t k = 0; k < K; k++) {
    a[PL[k]] = b[PR[k]];
    a[PR[k]] = b[PL[k]];
  }
  int cut = INT32_MAX;
  if (K < PL.size()) cut = std::min(cut, PL[K]);
  if (K >= 1) cut = std::min(cut, PR[K - 1]);
  return cut;
}

// the record store the device code's sequential pieces (rtr_heapsort.hpp) work on
struct Store {
  std::vector<Rec> *v;
  rtr::SortRec get(int i) const { return rtr::SortRec{(*v)[i].key, (uint32_t)(*v)[i].id}; }
  void set(int i, rtr::SortRec r) const { (*v)[i] = Rec{r.key, (int)r.id}; }
};

static int lg(int n) { return 31 - __builtin_clz(n); }

static long g_heaps = 0;

static void model_sort(std::vector<Rec> &a, int lo, int hi) {
  if (hi - lo < 2) return;
  std::vector<Seg> work{{lo, hi, 2 * lg(hi - lo)}};
  while (!work.empty()) {
    std::vector<Seg> next;
    for (Seg s : work) {  // all segments of a round are independent
      if (s.l - s.f <= 16) continue;
      if (s.depth == 0) {
        rtr::heap_sort(Store{&a}, s.f, s.l);
        g_heaps++;
        continue;
      }
      rtr::median_to_first(Store{&a}, s.f, s.f + 1, s.f + (s.l - s.f) / 2, s.l - 1);
      int cut = partition_by_ranks(a, s.f, s.l);
      next.push_back({cut, s.l, s.depth - 1});
      next.push_back({s.f, cut, s.depth - 1});
    }
    work.swap(next);
  }
  // the final insertion sort is a stable sort
  std::stable_sort(a.begin() + lo, a.begin() + hi, less);
}

int main(int argc, char **argv) {
  std::mt19937 rng(7);
  long cases = 0;
  const int trials = argc > 1 ? atoi(argv[1]) : 4000;
  for (int trial = 0; trial < trials; trial++) {
    int n = 1 + rng() % (trial % 50 == 0 ? 200000 : 3000);
    int distinct = 1 + rng() % (1 + rng() % (2 * n));
    std::vector<Rec> a(n);
    int mode = trial % 5;
    for (int i = 0; i < n; i++) {
      float k;
      if (mode == 0) k = (float)(rng() % distinct);
      else if (mode == 1) k = (float)(i / (1 + distinct % 7));           // sorted with ties
      else if (mode == 2) k = (float)((n - i) / (1 + distinct % 5));     // reversed with ties
      else if (mode == 3) k = (float)((i * 7919) % distinct);            // strided
      else k = (i & 1) ? (float)(i % distinct) : -(float)(i % 17);       // mixed, some -0/+0
      if (k == 0 && (rng() & 1)) k = -0.0f;
      a[i] = {k, i};
    }
    std::vector<Rec> want = a, got = a;
    std::sort(want.begin(), want.end(), less);
    model_sort(got, 0, n);
    for (int i = 0; i < n; i++)
      if (want[i].id != got[i].id) {
        printf("MISMATCH trial %d n %d at %d\n", trial, n, i);
        return 1;
      }
    cases++;
  }
  printf("ok: %ld cases equal to std::sort, %ld heap sorts\n", cases, g_heaps);
  return 0;
}
