// BVH::selfBuildBVH (reference src/BVH.cpp:19-64) on the device, with the reference's topology
// bit for bit -- the order in which its std::sort leaves objects whose keys tie included.
//
// The reference's tree is defined by std::sort: at every node the objects are sorted by box
// centre along the longest axis of the node's box and cut in the middle.  Which of several
// objects with EQUAL keys lands in which half is whatever libstdc++'s introsort leaves (it is
// not stable), so a tree with the same leaves in the same places needs that very permutation.
// This file computes it in parallel.  std::sort (bits/stl_algo.h) is
//
//   __introsort_loop:  while the range has more than 16 elements: if the depth budget
//                      (2 * floor(log2 n)) is spent, heap-sort the range; else move the median of
//                      {first+1, middle, last-1} to `first`, partition [first+1, last) around it
//                      (__unguarded_partition), recurse into [cut, last), continue with [first, cut);
//   __final_insertion_sort over the whole range.
//
//   * Hoare's partition is a data-parallel operation.  Its two scans only ever look at elements
//     that no swap has touched yet, so the k-th swap exchanges the k-th element from the left that
//     is not less than the pivot (position L_k) with the k-th from the right that is not greater
//     (R_k), for as long as L_k < R_k; with K such swaps the scans end at
//     cut = min(L_{K+1}, R_K).  L_k and R_k are ranks -- prefix sums of two flags.
//   * The ranges of one round of the recursion are disjoint, so all of them are partitioned at once,
//     round after round, until none has more than 16 elements.
//   * The insertion sort at the end moves an element left past strictly greater ones only: it is
//     a stable sort, and a stable sort's result does not depend on how it is computed.  The ranges
//     the recursion leaves are ordered among themselves, so every element's final place is its
//     range's start + the number of elements of that range that sort before it.
//   * The heap sort of a range whose depth budget ran out (not rare: runs of tied keys in sorted
//     input get there) is sequential; one lane does it, operation by operation as libstdc++ does.
//
// Checked against std::sort itself: tools/introsort_model.cpp (the formulation, on the CPU) and the
// tree tests, which compare every array of the result with the host builder's and the reference's.
//
// Shape of the computation.  Level L of the tree has 2^L ranges of the object array, of
// floor(n / 2^L) or one more objects.  Levels whose ranges exceed kT objects are processed by
// kernels over the whole array (box of every range by packed 64-bit atomics, keys, rounds of
// partitions with one workgroup per range, then one workgroup per left-over range of at most kT
// objects finishing its sort in LDS); from the first level whose ranges fit, ONE workgroup per range
// builds the whole subtree below it in LDS.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtr.h"
#include "rtr_heapsort.hpp"

namespace rtr {
int fail_msg(int code, const std::string &msg);  // rtr_api.hip: sets rtr_last_error()
}

namespace {

constexpr int kT = 2048;          // objects a workgroup sorts in LDS
constexpr int kWG = 256;          // its lanes
constexpr int kEPL = kT / kWG;    // consecutive positions per lane in the scan step
constexpr int kBigWG = 1024;      // lanes of the workgroup that partitions one large range
constexpr int kBigE = 4;          // positions per lane and trip there
static_assert(kBigE * (kBigWG / 64) == 64, "one wave scans the per-wave totals of a trip");
constexpr int kInsertion = 16;    // libstdc++'s _S_threshold
constexpr uint8_t kFinal = 0xFF;  // range is final, to be ranked (at most kInsertion elements)
constexpr uint8_t kSorted = 0xFE; // range is final and sorted (heap sort)

struct QsRange {
  int f, l, depth, pad;
};

struct NodeOut {
  float *bmin[3], *bmax[3];
  int32_t *left, *right, *object;
};

// ---- ranges of the tree: floor half to the left, the rest to the right (src/BVH.cpp:58-59) ----
__device__ inline void range_of_position(int n, int L, int p, int &k, int &lo, int &s) {
  lo = 0, s = n, k = 0;
  for (int b = 0; b < L; b++) {
    const int h = s >> 1;
    if (p >= lo + h) lo += h, s -= h, k = 2 * k + 1;
    else s = h, k = 2 * k;
  }
}
__host__ __device__ inline void range_of_index(int n, int L, int k, int &lo, int &s) {
  lo = 0, s = n;
  for (int b = L - 1; b >= 0; b--) {
    const int h = s >> 1;
    if ((k >> b) & 1) lo += h, s -= h;
    else s = h;
  }
}
// like range_of_position, inside one range of m objects; alive = the walk did not pass a range of
// one object before depth j (such a position became a leaf further up)
__device__ inline bool subrange_of_position(int m, int j, int p, int &k, int &lo, int &s) {
  lo = 0, s = m, k = 0;
  for (int b = 0; b < j; b++) {
    if (s == 1) return false;
    const int h = s >> 1;
    if (p >= lo + h) lo += h, s -= h, k = 2 * k + 1;
    else s = h, k = 2 * k;
  }
  return true;
}

// Node numbers: level by level.  Levels 0 .. Lc = floor(log2 n) are complete (2^L nodes); ranges
// of level Lc hold one or two objects, and the children of those with two follow, in order.
__device__ inline int node_number(int L, int Lc, int k, int lo) {
  if (L <= Lc) return (1 << L) - 1 + k;
  const int lo_parent = lo - (k & 1), k_parent = k >> 1;
  return (2 << Lc) - 1 + 2 * (lo_parent - k_parent) + (k & 1);
}
__device__ inline int left_child_number(int L, int Lc, int k, int lo) {
  return L < Lc ? (2 << L) - 1 + 2 * k : (2 << Lc) - 1 + 2 * (lo - k);
}

__device__ inline int floor_log2(int v) { return 31 - __clz(v); }

// ---- Bound3::merge (include/Bound3.h:14-20) as an order-free reduction ----
// `min.x < b.min.x ? min.x : b.min.x` keeps the LATER operand on a tie, which shows when a zero of
// either sign ties with the other: the merged plane carries the sign of the last zero merged.  A
// plane travels as (order-preserving code of the value with -0 folded onto +0) : (position, sign
// bit), so that a 64-bit minimum / maximum picks the value first and among equal values the later
// position, whose sign rides along.
__device__ inline uint32_t ordered_code(float v) {
  const uint32_t u = __float_as_uint(v == 0.f ? 0.f : v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline unsigned long long pack_min(float v, uint32_t pos) {
  return ((unsigned long long)ordered_code(v) << 32) | (((0x7fffffffu - pos) << 1) | (__float_as_uint(v) >> 31));
}
__device__ inline unsigned long long pack_max(float v, uint32_t pos) {
  return ((unsigned long long)ordered_code(v) << 32) | ((pos << 1) | (__float_as_uint(v) >> 31));
}
__device__ inline float unpack_plane(unsigned long long q) {
  const uint32_t o = (uint32_t)(q >> 32);
  const float v = __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
  return (v == 0.f && (q & 1)) ? -0.f : v;
}
constexpr unsigned long long kMinIdentity = ~0ull, kMaxIdentity = 0ull;

__device__ inline int longest_axis(const float lo[3], const float hi[3]) {
  const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
  return ex > ey ? (ex > ez ? 0 : 2) : (ey > ez ? 1 : 2);  // src/BVH.cpp:36-39
}
// Bound3::getCenterPos (include/Bound3.h:22): min * 0.5 + max * 0.5 in fp32, not fused
__device__ inline float centre_key(float lo, float hi) { return lo * 0.5f + hi * 0.5f; }

// ---------------------------------------------------------------------------------------------
// kernels over the whole array (levels whose ranges do not fit a workgroup)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_tb_init(int n, const float *box, uint32_t *idx, uint32_t *bad) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  idx[p] = p;
  bool ok = true;
  for (int a = 0; a < 6; a++) {
    const uint32_t u = __float_as_uint(box[(size_t)a * n + p]);
    ok = ok && ((u & 0x7f800000u) != 0x7f800000u);
  }
  if (!ok) atomicOr(bad, 1u);
}

__global__ void __launch_bounds__(256) k_tb_fill64(unsigned long long *a, int count_min, int count_all) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < count_all) a[i] = i < count_min ? kMinIdentity : kMaxIdentity;
}

// boxes of all ranges of level L: planes[a << L | k], a = 0..2 minima, 3..5 maxima
__global__ void __launch_bounds__(256)
k_tb_level_boxes(int n, int L, const uint32_t *idx, const float *box, unsigned long long *planes) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const bool live = p < n;
  int k = -1, lo, s;
  unsigned long long v[6];
  if (live) {
    range_of_position(n, L, p, k, lo, s);
    const uint32_t o = idx[p];
    for (int a = 0; a < 3; a++) {
      v[a] = pack_min(box[(size_t)a * n + o], (uint32_t)p);
      v[3 + a] = pack_max(box[(size_t)(3 + a) * n + o], (uint32_t)p);
    }
  } else {
    for (int a = 0; a < 3; a++) v[a] = kMinIdentity, v[3 + a] = kMaxIdentity;
  }
  const int k0 = __shfl(k, 0);
  if (k0 < 0) return;  // lane 0 of a wave holds its lowest position
  if (__all(!live || k == k0)) {
    for (int off = 32; off > 0; off >>= 1)
      for (int a = 0; a < 3; a++) {
        const unsigned long long m = __shfl_xor(v[a], off), x = __shfl_xor(v[3 + a], off);
        v[a] = m < v[a] ? m : v[a];
        v[3 + a] = x > v[3 + a] ? x : v[3 + a];
      }
    if ((threadIdx.x & 63) == 0)
      for (int a = 0; a < 3; a++) {
        atomicMin(&planes[((size_t)a << L) + k0], v[a]);
        atomicMax(&planes[((size_t)(3 + a) << L) + k0], v[3 + a]);
      }
  } else if (live) {
    for (int a = 0; a < 3; a++) {
      atomicMin(&planes[((size_t)a << L) + k], v[a]);
      atomicMax(&planes[((size_t)(3 + a) << L) + k], v[3 + a]);
    }
  }
}

__device__ inline void push_range(QsRange *list, int *count, int f, int l, int depth) {
  const int at = atomicAdd(count, 1);
  list[at] = QsRange{f, l, depth, 0};
}

// node records of level L, sort keys of its objects, and the ranges to sort
__global__ void __launch_bounds__(256)
k_tb_level_keys(int n, int L, int Lc, const uint32_t *idx, const float *box,
                const unsigned long long *planes, float *key, NodeOut out, QsRange *big, int *n_big,
                QsRange *small, int *n_small) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  int k, lo, s;
  range_of_position(n, L, p, k, lo, s);
  float bl[3], bh[3];
  for (int a = 0; a < 3; a++) {
    bl[a] = unpack_plane(planes[((size_t)a << L) + k]);
    bh[a] = unpack_plane(planes[((size_t)(3 + a) << L) + k]);
  }
  const int axis = longest_axis(bl, bh);
  const uint32_t o = idx[p];
  key[p] = centre_key(box[(size_t)axis * n + o], box[(size_t)(3 + axis) * n + o]);
  if (p != lo) return;
  const int id = node_number(L, Lc, k, lo);
  for (int a = 0; a < 3; a++) out.bmin[a][id] = bl[a], out.bmax[a][id] = bh[a];
  const int lc = left_child_number(L, Lc, k, lo);  // ranges of these levels hold at least two objects
  out.left[id] = lc;
  out.right[id] = lc + 1;
  out.object[id] = -1;
  if (s > kT) push_range(big, n_big, lo, lo + s, 2 * floor_log2(s));
  else push_range(small, n_small, lo, lo + s, 2 * floor_log2(s));
}

struct GlobalRecords {
  float *key;
  uint32_t *idx;
  __device__ rtr::SortRec get(int i) const { return rtr::SortRec{key[i], idx[i]}; }
  __device__ void set(int i, rtr::SortRec r) const { key[i] = r.key, idx[i] = r.id; }
};

// One round of the introsort recursion for one range of more than kT objects: median of three to
// the front, Hoare partition by ranks, the two parts queued.  Two passes over the range: ranks
// (positions of the swap candidates written to PL / PR), then the swaps.
__global__ void __launch_bounds__(kBigWG)
k_tb_partition(const QsRange *in, uint32_t *idx, float *key, uint32_t *PL, uint32_t *PR, QsRange *big,
               int *n_big, QsRange *small, int *n_small) {
  __shared__ uint32_t wave_total[2][kBigE * (kBigWG / 64)];
  __shared__ int sK;
  const QsRange rg = in[blockIdx.x];
  const int f = rg.f, l = rg.l, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  GlobalRecords rec{key, idx};
  if (rg.depth == 0) {
    if (tid == 0) rtr::heap_sort(rec, f, l);
    return;
  }
  if (tid == 0) {
    rtr::median_to_first(rec, f, f + 1, f + (l - f) / 2, l - 1);
    sK = 0;
  }
  __syncthreads();
  const float pivot = key[f];
  const int first = f + 1, count = l - first;
  int base_l = 0, base_r = 0;  // candidates seen in earlier trips
  for (int c0 = 0, trip = 0; c0 < count; c0 += kBigWG * kBigE, trip++) {
    bool fl[kBigE], fr[kBigE];
    uint32_t rank[kBigE];  // candidates before this lane in its wave: left | right << 16
    for (int e = 0; e < kBigE; e++) {
      const int p = first + c0 + e * kBigWG + tid;
      const bool in_range = p < l;
      const float kv = in_range ? key[p] : 0.f;
      fl[e] = in_range && !(kv < pivot);
      fr[e] = in_range && !(pivot < kv);
      const unsigned long long ml = __ballot(fl[e]), mr = __ballot(fr[e]);
      const unsigned long long below = (1ull << lane) - 1;
      rank[e] = (uint32_t)__popcll(ml & below) | ((uint32_t)__popcll(mr & below) << 16);
      if (lane == 0)
        wave_total[trip & 1][e * (kBigWG / 64) + wave] = (uint32_t)__popcll(ml) | ((uint32_t)__popcll(mr) << 16);
    }
    __syncthreads();
    // exclusive prefix over the 64 (sub-trip, wave) totals, in position order, by every wave
    uint32_t t = wave_total[trip & 1][lane], incl = t;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = __shfl_up(incl, off);
      if (lane >= off) incl += up;
    }
    const uint32_t excl = incl - t;
    for (int e = 0; e < kBigE; e++) {
      const uint32_t before = __shfl(excl, e * (kBigWG / 64) + wave) + rank[e];
      const int p = first + c0 + e * kBigWG + tid;
      if (fl[e]) PL[first + base_l + (int)(before & 0xffffu)] = (uint32_t)p;
      if (fr[e]) PR[first + base_r + (int)(before >> 16)] = (uint32_t)p;
    }
    const uint32_t all = __shfl(incl, 63);
    base_l += (int)(all & 0xffffu);
    base_r += (int)(all >> 16);
  }
  __syncthreads();
  // PR holds the right candidates left to right: the k-th from the right is PR[first + cr - 1 - k]
  const int cl = base_l, cr = base_r, m = cl < cr ? cl : cr;
  for (int k0 = tid; k0 < m; k0 += kBigWG) {
    const uint32_t a = PL[first + k0], b = PR[first + cr - 1 - k0];
    const bool sw = a < b;
    if (sw) {
      const float ka = key[a], kb = key[b];
      const uint32_t ia = idx[a], ib = idx[b];
      key[a] = kb, key[b] = ka, idx[a] = ib, idx[b] = ia;
    }
    const bool next = (k0 + 1 < m) && PL[first + k0 + 1] < PR[first + cr - 2 - k0];
    if (sw && !next) sK = k0 + 1;
  }
  __syncthreads();
  if (tid == 0) {
    const int K = sK;
    uint32_t cut = 0xffffffffu;
    if (K < cl) cut = PL[first + K];
    if (K >= 1) {
      const uint32_t r = PR[first + cr - K];
      cut = r < cut ? r : cut;
    }
    const int c = (int)cut, d = rg.depth - 1;
    const int parts[2][2] = {{c, l}, {f, c}};
    for (int h = 0; h < 2; h++) {
      const int size = parts[h][1] - parts[h][0];
      if (size > kT) push_range(big, n_big, parts[h][0], parts[h][1], d);
      else if (size >= 2) push_range(small, n_small, parts[h][0], parts[h][1], d);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// sorting in LDS: any number of disjoint ranges of [0, m) at once
// ---------------------------------------------------------------------------------------------
struct LdsSort {
  float key[kT];
  uint32_t idx[kT];
  uint32_t scan[kT];       // inclusive counts of left | right << 16 candidates up to a position
  uint32_t cand[kT];       // two uint16 arrays: left candidates, right candidates (by range slot)
  uint16_t start[kT], end[kT];  // the range a position belongs to
  uint16_t cut[kT];        // by range start
  uint8_t depth[kT];       // by range start: depth budget left, kFinal or kSorted
  uint32_t wave_total[kWG / 64];
  int any[2];
  __device__ uint16_t *pl() { return (uint16_t *)cand; }
  __device__ uint16_t *pr() { return (uint16_t *)cand + kT; }
};

struct LdsRecords {
  LdsSort *s;
  __device__ rtr::SortRec get(int i) const { return rtr::SortRec{s->key[i], s->idx[i]}; }
  __device__ void set(int i, rtr::SortRec r) const { s->key[i] = r.key, s->idx[i] = r.id; }
};

__device__ inline uint8_t part_state(int size, int depth) {
  return size > kInsertion ? (uint8_t)depth : kFinal;
}

// Sorts every range described by start / end / depth as std::sort would.  All lanes of the
// workgroup call it; key, idx, start, end and depth[start] are set and a barrier has passed.
__device__ void lds_sort(LdsSort &S, int m) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint16_t *PL = S.pl(), *PR = S.pr();
  LdsRecords rec{&S};
  for (int round = 0;; round++) {
    if (tid == 0) S.any[round & 1] = 0;
    // range heads: heap sort when the budget is spent, else the median of three to the front
    for (int p = tid; p < m; p += kWG) {
      if (S.start[p] != p) continue;
      const int d = S.depth[p], l = S.end[p];
      if (d == 0) {
        rtr::heap_sort(rec, p, l);
        S.depth[p] = kSorted;
      } else if (d < 0x80) {
        rtr::median_to_first(rec, p, p + 1, p + (l - p) / 2, l - 1);
      }
    }
    __syncthreads();
    // candidates and their counts; a lane owns kEPL consecutive positions
    bool fl[kEPL], fr[kEPL];
    uint32_t own[kEPL], total = 0;
    for (int e = 0; e < kEPL; e++) {
      const int p = tid * kEPL + e;
      fl[e] = fr[e] = false;
      if (p < m) {
        const int f = S.start[p];
        if (p > f && S.depth[f] < 0x80) {
          const float pivot = S.key[f], kv = S.key[p];
          fl[e] = !(kv < pivot);
          fr[e] = !(pivot < kv);
        }
      }
      total += (fl[e] ? 1u : 0u) + (fr[e] ? 0x10000u : 0u);
      own[e] = total;
    }
    uint32_t incl = total;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = __shfl_up(incl, off);
      if (lane >= off) incl += up;
    }
    if (lane == 63) S.wave_total[wave] = incl;
    __syncthreads();
    uint32_t before = incl - total;
    for (int w = 0; w < wave; w++) before += S.wave_total[w];
    for (int e = 0; e < kEPL; e++) {
      const int p = tid * kEPL + e;
      if (p < m) S.scan[p] = before + own[e];
    }
    __syncthreads();
    for (int e = 0; e < kEPL; e++) {
      const int p = tid * kEPL + e;
      if (!(fl[e] || fr[e])) continue;
      const int f = S.start[p], l = S.end[p];
      const uint32_t at_f = S.scan[f], here = S.scan[p], at_end = S.scan[l - 1];
      if (fl[e]) PL[f + 1 + (int)((here & 0xffffu) - (at_f & 0xffffu)) - 1] = (uint16_t)p;
      if (fr[e]) {
        const int cr = (int)((at_end >> 16) - (at_f >> 16)), seen = (int)((here >> 16) - (at_f >> 16));
        PR[f + 1 + cr - seen] = (uint16_t)p;  // k-th from the right at slot k
      }
    }
    __syncthreads();
    // the swaps; the lane at the last swap (or at the first pair, when there is none) knows the cut
    for (int p = tid; p < m; p += kWG) {
      const int f = S.start[p];
      if (p == f || S.depth[f] >= 0x80) continue;
      const int l = S.end[p];
      const uint32_t at_f = S.scan[f], at_end = S.scan[l - 1];
      const int cl = (int)((at_end & 0xffffu) - (at_f & 0xffffu)), cr = (int)((at_end >> 16) - (at_f >> 16));
      const int mm = cl < cr ? cl : cr, k0 = p - f - 1;
      if (k0 >= mm) continue;
      const int a = PL[p], b = PR[p];
      const bool sw = a < b;
      if (sw) {
        const float ka = S.key[a], kb = S.key[b];
        const uint32_t ia = S.idx[a], ib = S.idx[b];
        S.key[a] = kb, S.key[b] = ka, S.idx[a] = ib, S.idx[b] = ia;
      }
      const bool next = (k0 + 1 < mm) && PL[p + 1] < PR[p + 1];
      if (sw && !next) {
        int c = b;  // R_K
        if (k0 + 1 < cl && (int)PL[p + 1] < c) c = PL[p + 1];
        S.cut[f] = (uint16_t)c;
      } else if (k0 == 0 && !sw) {
        S.cut[f] = (uint16_t)a;  // no swap at all: L_1
      }
    }
    __syncthreads();
    // every position moves to its part
    int nf[kEPL], nl[kEPL], od[kEPL];
    for (int e = 0; e < kEPL; e++) {
      const int p = e * kWG + tid;
      od[e] = 0xFF;
      if (p >= m) continue;
      const int f = S.start[p];
      od[e] = S.depth[f];
      if (od[e] >= 0x80) continue;
      const int c = S.cut[f], l = S.end[p];
      nf[e] = p < c ? f : c;
      nl[e] = p < c ? c : l;
    }
    __syncthreads();
    for (int e = 0; e < kEPL; e++) {
      const int p = e * kWG + tid;
      if (od[e] >= 0x80) continue;
      S.start[p] = (uint16_t)nf[e];
      S.end[p] = (uint16_t)nl[e];
      if (p == nf[e]) {
        const uint8_t st = part_state(nl[e] - nf[e], od[e] - 1);
        S.depth[p] = st;
        if (st < 0x80) S.any[round & 1] = 1;
      }
    }
    __syncthreads();
    if (!S.any[round & 1]) break;
  }
  // the final insertion sort = a stable sort of ranges that are ordered among themselves
  uint32_t *tk = S.scan, *ti = S.cand;
  for (int p = tid; p < m; p += kWG) {
    const int f = S.start[p], l = S.end[p];
    int dst = p;
    if (S.depth[f] != kSorted && l - f > 1) {
      const float kv = S.key[p];
      int r = 0;
      for (int q = f; q < l; q++) {
        const float kq = S.key[q];
        r += (kq < kv || (kq == kv && q < p)) ? 1 : 0;
      }
      dst = f + r;
    }
    tk[dst] = __float_as_uint(S.key[p]);
    ti[dst] = S.idx[p];
  }
  __syncthreads();
  for (int p = tid; p < m; p += kWG) {
    S.key[p] = __uint_as_float(tk[p]);
    S.idx[p] = ti[p];
  }
  __syncthreads();
}

// a left-over range of at most kT objects of a level that is sorted by kernels over the whole array
__global__ void __launch_bounds__(kWG)
k_tb_sort_small(const QsRange *list, uint32_t *idx, const float *key) {
  __shared__ LdsSort S;
  const QsRange rg = list[blockIdx.x];
  const int m = rg.l - rg.f, tid = threadIdx.x;
  for (int p = tid; p < m; p += kWG) {
    S.key[p] = key[rg.f + p];
    S.idx[p] = idx[rg.f + p];
    S.start[p] = 0;
    S.end[p] = (uint16_t)m;
  }
  if (tid == 0) S.depth[0] = part_state(m, rg.depth);
  __syncthreads();
  lds_sort(S, m);
  for (int p = tid; p < m; p += kWG) idx[rg.f + p] = S.idx[p];
}

// The whole subtree under range kb of level Ls (at most kT objects), level by level in LDS.
__global__ void __launch_bounds__(kWG)
k_tb_subtree(int n, int Ls, int Lc, uint32_t *idx, const float *box, NodeOut out, uint32_t *root_order) {
  __shared__ LdsSort S;
  __shared__ uint32_t gid[kT];                // object of a slot; S.idx holds slots
  __shared__ uint8_t axis_of[kT];             // by range start
  __shared__ unsigned long long planes[6][64];
  const int kb = blockIdx.x, tid = threadIdx.x;
  int LO, m;
  range_of_index(n, Ls, kb, LO, m);
  for (int p = tid; p < m; p += kWG) {
    gid[p] = idx[LO + p];
    S.idx[p] = p;
  }
  __syncthreads();
  for (int j = 0;; j++) {
    const int L = Ls + j;
    int smax = m, smin = m;  // largest and smallest range at depth j
    bool level_exists = true;
    for (int b = 0; b < j; b++) {
      if (smax < 2) level_exists = false;
      smax -= smax >> 1;
      smin >>= 1;
    }
    if (!level_exists) break;
    const bool by_atomics = smin > 32;  // then at most 64 ranges
    if (by_atomics) {
      for (int i = tid; i < 6 * 64; i += kWG) planes[i / 64][i % 64] = i < 3 * 64 ? kMinIdentity : kMaxIdentity;
      __syncthreads();
      for (int p = tid; p < m; p += kWG) {
        int k, lo, s;
        subrange_of_position(m, j, p, k, lo, s);
        const uint32_t o = gid[S.idx[p]];
        for (int a = 0; a < 3; a++) {
          atomicMin(&planes[a][k], pack_min(box[(size_t)a * n + o], (uint32_t)p));
          atomicMax(&planes[3 + a][k], pack_max(box[(size_t)(3 + a) * n + o], (uint32_t)p));
        }
      }
      __syncthreads();
    }
    // range heads: the node's record and the sort axis
    for (int p = tid; p < m; p += kWG) {
      int k, lo, s;
      if (!subrange_of_position(m, j, p, k, lo, s) || p != lo) continue;
      float bl[3], bh[3];
      if (by_atomics) {
        for (int a = 0; a < 3; a++) bl[a] = unpack_plane(planes[a][k]), bh[a] = unpack_plane(planes[3 + a][k]);
      } else {
        // Bound3() and Bound3::merge, object after object (include/Bound3.h:10-20)
        for (int a = 0; a < 3; a++) bl[a] = 3.402823466e+38f, bh[a] = -3.402823466e+38f;
        for (int q = lo; q < lo + s; q++) {
          const uint32_t o = gid[S.idx[q]];
          for (int a = 0; a < 3; a++) {
            const float vl = box[(size_t)a * n + o], vh = box[(size_t)(3 + a) * n + o];
            bl[a] = bl[a] < vl ? bl[a] : vl;
            bh[a] = bh[a] > vh ? bh[a] : vh;
          }
        }
      }
      const int kg = (kb << j) | k, log = LO + lo;
      const int id = node_number(L, Lc, kg, log);
      for (int a = 0; a < 3; a++) out.bmin[a][id] = bl[a], out.bmax[a][id] = bh[a];
      if (s == 1) {
        out.left[id] = -1, out.right[id] = -1;
        out.object[id] = (int32_t)gid[S.idx[p]];
        axis_of[p] = 3;
      } else {
        const int lc = left_child_number(L, Lc, kg, log);
        out.left[id] = lc, out.right[id] = lc + 1;
        out.object[id] = -1;
        axis_of[p] = (uint8_t)longest_axis(bl, bh);
      }
    }
    __syncthreads();
    for (int p = tid; p < m; p += kWG) {
      int k, lo, s;
      const bool alive = subrange_of_position(m, j, p, k, lo, s);
      if (!alive || s == 1) {
        S.start[p] = (uint16_t)p, S.end[p] = (uint16_t)(p + 1), S.depth[p] = kFinal;
        continue;
      }
      const int axis = axis_of[lo];
      const uint32_t o = gid[S.idx[p]];
      S.key[p] = centre_key(box[(size_t)axis * n + o], box[(size_t)(3 + axis) * n + o]);
      S.start[p] = (uint16_t)lo, S.end[p] = (uint16_t)(lo + s);
      if (p == lo) S.depth[p] = part_state(s, 2 * floor_log2(s));
    }
    __syncthreads();
    lds_sort(S, m);
    if (L == 0)
      for (int p = tid; p < m; p += kWG) root_order[p] = gid[S.idx[p]];
  }
}

#define TB_TRY(expr)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      rc = rtr::fail_msg(RTR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));  \
      goto done;                                                                              \
    }                                                                                         \
  } while (0)

}  // namespace

extern "C" int rtr_build_bvh(int device, int32_t n_objects, const float *const obj_bmin[3],
                             const float *const obj_bmax[3], float *const node_bmin[3],
                             float *const node_bmax[3], int32_t *node_left, int32_t *node_right,
                             int32_t *node_object, int32_t *root_order) {
  if (n_objects < 1 || n_objects > (1 << 25))
    return rtr::fail_msg(RTR_ERR_INVALID, "rtr_build_bvh: n_objects must be in [1, 2^25]");
  if (!obj_bmin || !obj_bmax || !node_bmin || !node_bmax || !node_left || !node_right || !node_object)
    return rtr::fail_msg(RTR_ERR_INVALID, "rtr_build_bvh: null array");
  for (int a = 0; a < 3; a++)
    if (!obj_bmin[a] || !obj_bmax[a] || !node_bmin[a] || !node_bmax[a])
      return rtr::fail_msg(RTR_ERR_INVALID, "rtr_build_bvh: null array");
  const int n = n_objects, n_nodes = 2 * n - 1;
  int Lc = 0, Ls = 0;
  while ((2 << Lc) <= n) Lc++;                               // floor(log2 n)
  while (((n + (1 << Ls) - 1) >> Ls) > kT) Ls++;             // first level whose ranges fit a workgroup
  int rc = RTR_OK;
  float *d_box = nullptr, *d_key = nullptr, *d_nodes = nullptr;
  uint32_t *d_idx = nullptr, *d_pl = nullptr, *d_pr = nullptr, *d_root = nullptr, *d_flags = nullptr;
  int32_t *d_links = nullptr;
  unsigned long long *d_planes = nullptr;
  QsRange *d_list[3] = {nullptr, nullptr, nullptr};  // two lists of large ranges (in, out), one of small ones
  int *d_counts = nullptr;                           // [0], [1]: large lists; [2]: small list
  int h_counts[3];
  NodeOut out;
  const size_t list_cap = (size_t)n / 2 + 2;
  const int grid_n = (n + 255) / 256;
  if (hipSetDevice(device) != hipSuccess) return rtr::fail_msg(RTR_ERR_DEVICE, "rtr_build_bvh: no such device");
  TB_TRY(hipMalloc((void **)&d_box, sizeof(float) * 6 * n));
  TB_TRY(hipMalloc((void **)&d_key, sizeof(float) * n));
  TB_TRY(hipMalloc((void **)&d_idx, sizeof(uint32_t) * n));
  TB_TRY(hipMalloc((void **)&d_root, sizeof(uint32_t) * n));
  TB_TRY(hipMalloc((void **)&d_nodes, sizeof(float) * 6 * n_nodes));
  TB_TRY(hipMalloc((void **)&d_links, sizeof(int32_t) * 3 * n_nodes));
  TB_TRY(hipMalloc((void **)&d_flags, sizeof(uint32_t)));
  TB_TRY(hipMemset(d_flags, 0, sizeof(uint32_t)));
  for (int a = 0; a < 3; a++) {
    TB_TRY(hipMemcpy(d_box + (size_t)a * n, obj_bmin[a], sizeof(float) * n, hipMemcpyHostToDevice));
    TB_TRY(hipMemcpy(d_box + (size_t)(3 + a) * n, obj_bmax[a], sizeof(float) * n, hipMemcpyHostToDevice));
    out.bmin[a] = d_nodes + (size_t)a * n_nodes;
    out.bmax[a] = d_nodes + (size_t)(3 + a) * n_nodes;
  }
  out.left = d_links, out.right = d_links + n_nodes, out.object = d_links + 2 * (size_t)n_nodes;
  k_tb_init<<<grid_n, 256>>>(n, d_box, d_idx, d_flags);
  {
    uint32_t bad = 0;
    TB_TRY(hipMemcpy(&bad, d_flags, sizeof(bad), hipMemcpyDeviceToHost));
    if (bad) {
      rc = rtr::fail_msg(RTR_ERR_INVALID, "rtr_build_bvh: a box plane is infinite or not a number");
      goto done;
    }
  }
  if (Ls > 0) {
    TB_TRY(hipMalloc((void **)&d_pl, sizeof(uint32_t) * n));
    TB_TRY(hipMalloc((void **)&d_pr, sizeof(uint32_t) * n));
    TB_TRY(hipMalloc((void **)&d_planes, sizeof(unsigned long long) * 6 * ((size_t)1 << (Ls - 1))));
    for (int i = 0; i < 3; i++) TB_TRY(hipMalloc((void **)&d_list[i], sizeof(QsRange) * list_cap));
    TB_TRY(hipMalloc((void **)&d_counts, sizeof(int) * 3));
  }
  for (int L = 0; L < Ls; L++) {
    const int ranges = 1 << L;
    k_tb_fill64<<<(6 * ranges + 255) / 256, 256>>>(d_planes, 3 * ranges, 6 * ranges);
    k_tb_level_boxes<<<grid_n, 256>>>(n, L, d_idx, d_box, d_planes);
    TB_TRY(hipMemsetAsync(d_counts, 0, sizeof(int) * 3, 0));
    k_tb_level_keys<<<grid_n, 256>>>(n, L, Lc, d_idx, d_box, d_planes, d_key, out, d_list[0], d_counts,
                                     d_list[2], d_counts + 2);
    int cur = 0;
    for (int round = 0;; round++) {
      TB_TRY(hipMemcpy(h_counts, d_counts, sizeof(h_counts), hipMemcpyDeviceToHost));
      const int n_in = h_counts[cur];
      if (n_in == 0) break;
      if (round > 2 * 26 + 2) {
        rc = rtr::fail_msg(RTR_ERR_DEVICE, "rtr_build_bvh: the sort did not end within its depth budget");
        goto done;
      }
      k_tb_partition<<<n_in, kBigWG>>>(d_list[cur], d_idx, d_key, d_pl, d_pr, d_list[1 - cur],
                                       d_counts + (1 - cur), d_list[2], d_counts + 2);
      TB_TRY(hipMemsetAsync(d_counts + cur, 0, sizeof(int), 0));
      cur = 1 - cur;
    }
    if (h_counts[2] > 0) k_tb_sort_small<<<h_counts[2], kWG>>>(d_list[2], d_idx, d_key);
    if (L == 0) TB_TRY(hipMemcpyAsync(d_root, d_idx, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, 0));
  }
  k_tb_subtree<<<1 << Ls, kWG>>>(n, Ls, Lc, d_idx, d_box, out, d_root);
  TB_TRY(hipGetLastError());
  TB_TRY(hipDeviceSynchronize());
  for (int a = 0; a < 3; a++) {
    TB_TRY(hipMemcpy(node_bmin[a], out.bmin[a], sizeof(float) * n_nodes, hipMemcpyDeviceToHost));
    TB_TRY(hipMemcpy(node_bmax[a], out.bmax[a], sizeof(float) * n_nodes, hipMemcpyDeviceToHost));
  }
  TB_TRY(hipMemcpy(node_left, out.left, sizeof(int32_t) * n_nodes, hipMemcpyDeviceToHost));
  TB_TRY(hipMemcpy(node_right, out.right, sizeof(int32_t) * n_nodes, hipMemcpyDeviceToHost));
  TB_TRY(hipMemcpy(node_object, out.object, sizeof(int32_t) * n_nodes, hipMemcpyDeviceToHost));
  if (root_order) TB_TRY(hipMemcpy(root_order, d_root, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
done:
  (void)hipFree(d_box), (void)hipFree(d_key), (void)hipFree(d_idx), (void)hipFree(d_root);
  (void)hipFree(d_nodes), (void)hipFree(d_links), (void)hipFree(d_flags), (void)hipFree(d_pl);
  (void)hipFree(d_pr), (void)hipFree(d_planes), (void)hipFree(d_counts);
  for (int i = 0; i < 3; i++) (void)hipFree(d_list[i]);
  return rc;
}
