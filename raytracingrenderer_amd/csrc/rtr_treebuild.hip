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

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/rtr.h"
#include "rtr_heapsort.hpp"

namespace rtr {
int fail_msg(int code, const std::string &msg);  // rtr_api.hip: sets rtr_last_error()
}

namespace {

constexpr int kT = 2048;          // objects a workgroup sorts in LDS
constexpr int kWG = 1024;         // its lanes
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
// Nodes are numbered depth first, left child first (the order BVH::intersect visits them): a
// subtree over s objects has 2s - 1 nodes, so the left child of node i is i + 1 and the right child
// i + 2 * floor(s / 2); every walk down the ranges carries the node number along.
__device__ inline void range_of_position(int n, int L, int p, int &k, int &lo, int &s, int &id) {
  lo = 0, s = n, k = 0, id = 0;
  for (int b = 0; b < L; b++) {
    const int h = s >> 1;
    if (p >= lo + h) lo += h, s -= h, k = 2 * k + 1, id += 2 * h;
    else s = h, k = 2 * k, id += 1;
  }
}
__host__ __device__ inline void range_of_index(int n, int L, int k, int &lo, int &s, int &id) {
  lo = 0, s = n, id = 0;
  for (int b = L - 1; b >= 0; b--) {
    const int h = s >> 1;
    if ((k >> b) & 1) lo += h, s -= h, id += 2 * h;
    else s = h, id += 1;
  }
}
// like range_of_position, inside one range of m objects (id counts from that range's node);
// alive = the walk did not pass a range of one object before depth j (such a position became a
// leaf further up)
__device__ inline bool subrange_of_position(int m, int j, int p, int &k, int &lo, int &s, int &id) {
  lo = 0, s = m, k = 0, id = 0;
  for (int b = 0; b < j; b++) {
    if (s == 1) return false;
    const int h = s >> 1;
    if (p >= lo + h) lo += h, s -= h, k = 2 * k + 1, id += 2 * h;
    else s = h, k = 2 * k, id += 1;
  }
  return true;
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

// boxes of all ranges of level L: planes[a << L | k], a = 0..2 minima, 3..5 maxima.  A wave whose
// positions are all in one range reduces before it touches memory, and so does a workgroup whose
// waves are: the upper levels would otherwise queue a hundred thousand atomics on six words.
constexpr int kBoxWG = 1024;
__global__ void __launch_bounds__(kBoxWG)
k_tb_level_boxes(int n, int L, const uint32_t *idx, const float *box, unsigned long long *planes) {
  __shared__ unsigned long long wave_planes[kBoxWG / 64][6];
  __shared__ int wave_range[kBoxWG / 64];  // the wave's range, -1: none (all beyond n), -2: several
  const int p = blockIdx.x * kBoxWG + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool live = p < n;
  int k = -1, lo, s, id;
  unsigned long long v[6];
  for (int a = 0; a < 3; a++) v[a] = kMinIdentity, v[3 + a] = kMaxIdentity;
  if (live) {
    range_of_position(n, L, p, k, lo, s, id);
    const uint32_t o = idx[p];
    for (int a = 0; a < 3; a++) {
      v[a] = pack_min(box[(size_t)a * n + o], (uint32_t)p);
      v[3 + a] = pack_max(box[(size_t)(3 + a) * n + o], (uint32_t)p);
    }
  }
  const int k0 = __shfl(k, 0);  // lane 0 of a wave holds its lowest position
  const bool uniform = k0 >= 0 && __all(!live || k == k0);
  if (uniform) {
    for (int off = 32; off > 0; off >>= 1)
      for (int a = 0; a < 3; a++) {
        const unsigned long long m = __shfl_xor(v[a], off), x = __shfl_xor(v[3 + a], off);
        v[a] = m < v[a] ? m : v[a];
        v[3 + a] = x > v[3 + a] ? x : v[3 + a];
      }
    if (lane == 0)
      for (int a = 0; a < 6; a++) wave_planes[wave][a] = v[a];
  } else if (live) {
    for (int a = 0; a < 3; a++) {
      atomicMin(&planes[((size_t)a << L) + k], v[a]);
      atomicMax(&planes[((size_t)(3 + a) << L) + k], v[3 + a]);
    }
  }
  if (lane == 0) wave_range[wave] = uniform ? k0 : (k0 < 0 ? -1 : -2);
  __syncthreads();
  // plane a of the workgroup's first range by lane a of wave 0; other waves' ranges by themselves
  const int kw = wave_range[0];
  if (wave == 0 && lane < 6 && kw >= 0) {
    unsigned long long r = wave_planes[0][lane];
    for (int w = 1; w < kBoxWG / 64; w++)
      if (wave_range[w] == kw) {
        const unsigned long long o = wave_planes[w][lane];
        r = lane < 3 ? (o < r ? o : r) : (o > r ? o : r);
      }
    if (lane < 3) atomicMin(&planes[((size_t)lane << L) + kw], r);
    else atomicMax(&planes[((size_t)lane << L) + kw], r);
  }
  if (wave > 0 && lane < 6 && wave_range[wave] >= 0 && wave_range[wave] != kw) {
    const unsigned long long r = wave_planes[wave][lane];
    if (lane < 3) atomicMin(&planes[((size_t)lane << L) + wave_range[wave]], r);
    else atomicMax(&planes[((size_t)lane << L) + wave_range[wave]], r);
  }
}

struct GlobalRecords {
  float *key;
  uint32_t *idx;
  __device__ rtr::SortRec get(int i) const { return rtr::SortRec{key[i], idx[i]}; }
  __device__ void set(int i, rtr::SortRec r) const { key[i] = r.key, idx[i] = r.id; }
};

// A range of more than kT objects on its way through one round of the introsort recursion.  Its
// partition is cut into tiles of kTile positions, one workgroup each; tile numbers run through
// all ranges of the round (tile_base = tiles of the ranges queued before it).
constexpr int kTile = kBigWG * kBigE;
struct BigRange {
  int f, l, depth, tile_base;
  int K, cl, cr, pad;  // swaps made, candidates on either side (set by the swap pass)
};
struct SortQueues {
  BigRange *big[2];                 // this round's and the next round's large ranges
  unsigned long long *big_count;    // [2]: ranges << 32 | tiles
  QsRange *small;                   // ranges of at most kT objects, finished in LDS
  int *small_count;
  uint32_t *tile_counts;            // per tile: left | right << 16 candidates
  uint32_t *flags;                  // bit 0: a box plane is not finite; bit 1: a heap sort too long for one lane
};
// A heap sort is one lane's work, a dependent chain of memory accesses: ranges beyond this many
// objects (only key sequences built to defeat the median of three get there) are refused instead
// of holding the device for seconds.
constexpr int kHeapSortMax = 1 << 17;

__device__ inline void push_small(const SortQueues &q, int f, int l, int depth) {
  q.small[atomicAdd(q.small_count, 1)] = QsRange{f, l, depth, 0};
}
// Queues [f, l) for the next round: the median of three goes to the front now (the range is at
// rest between rounds), so that every tile of the round finds the pivot at f.  A range whose depth
// budget is spent is heap-sorted here, by this one lane, and not queued.
__device__ inline void push_big(const SortQueues &q, int to, GlobalRecords rec, int f, int l, int depth) {
  if (depth == 0) {
    if (l - f > kHeapSortMax) atomicOr(q.flags, 2u);
    else rtr::heap_sort(rec, f, l);
    return;
  }
  rtr::median_to_first(rec, f, f + 1, f + (l - f) / 2, l - 1);
  const int tiles = (l - f - 1 + kTile - 1) / kTile;
  const unsigned long long old = atomicAdd(&q.big_count[to], (1ull << 32) | (unsigned long long)tiles);
  q.big[to][(int)(old >> 32)] = BigRange{f, l, depth, (int)(old & 0xffffffffu), 0, 0, 0, 0};
}

// node records of level L and the sort keys of its objects
__global__ void __launch_bounds__(256)
k_tb_level_keys(int n, int L, const uint32_t *idx, const float *box,
                const unsigned long long *planes, float *key, NodeOut out) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  int k, lo, s, id;
  range_of_position(n, L, p, k, lo, s, id);
  float bl[3], bh[3];
  for (int a = 0; a < 3; a++) {
    bl[a] = unpack_plane(planes[((size_t)a << L) + k]);
    bh[a] = unpack_plane(planes[((size_t)(3 + a) << L) + k]);
  }
  const int axis = longest_axis(bl, bh);
  const uint32_t o = idx[p];
  key[p] = centre_key(box[(size_t)axis * n + o], box[(size_t)(3 + axis) * n + o]);
  if (p != lo) return;
  for (int a = 0; a < 3; a++) out.bmin[a][id] = bl[a], out.bmax[a][id] = bh[a];
  out.left[id] = id + 1;  // ranges of these levels hold at least two objects
  out.right[id] = id + 2 * (s >> 1);
  out.object[id] = -1;
}

// the ranges of level L, queued for their sort (after every key of the level is written)
__global__ void __launch_bounds__(256)
k_tb_level_ranges(int n, int L, SortQueues q, uint32_t *idx, float *key) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= (1 << L)) return;
  int lo, s, id;
  range_of_index(n, L, k, lo, s, id);
  if (s > kT) push_big(q, 0, GlobalRecords{key, idx}, lo, lo + s, 2 * floor_log2(s));
  else push_small(q, lo, lo + s, 2 * floor_log2(s));
}

// which range a tile belongs to: the last one whose tile_base is not beyond it
__device__ inline int range_of_tile(const BigRange *list, int count, int tile) {
  int lo = 0, hi = count - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (list[mid].tile_base <= tile) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

// One round of the introsort recursion over all large ranges, three passes of one workgroup per
// tile: candidates counted (k_tb_count), their positions written in rank order to PL / PR
// (k_tb_scatter), the swaps (k_tb_swap); then one lane per range queues the two parts (k_tb_finish).
struct TileFlags {
  uint32_t left, right;  // bit e: position e * kBigWG + lane of the tile is a candidate
};
__device__ inline TileFlags tile_flags(const BigRange &rg, int t, const float *key) {
  const float pivot = key[rg.f];
  TileFlags fl{0u, 0u};
  for (int e = 0; e < kBigE; e++) {
    const int p = rg.f + 1 + t * kTile + e * kBigWG + (int)threadIdx.x;
    if (p < rg.l) {
      const float kv = key[p];
      fl.left |= (!(kv < pivot) ? 1u : 0u) << e;
      fl.right |= (!(pivot < kv) ? 1u : 0u) << e;
    }
  }
  return fl;
}

__global__ void __launch_bounds__(kBigWG)
k_tb_count(const BigRange *list, int count, const float *key, uint32_t *tile_counts) {
  __shared__ int s_range;
  __shared__ uint32_t s_sum[2];
  if (threadIdx.x == 0) s_range = range_of_tile(list, count, blockIdx.x), s_sum[0] = s_sum[1] = 0;
  __syncthreads();
  const BigRange rg = list[s_range];
  const TileFlags fl = tile_flags(rg, blockIdx.x - rg.tile_base, key);
  uint32_t cl = __popc(fl.left), cr = __popc(fl.right);
  for (int off = 32; off > 0; off >>= 1) cl += __shfl_xor(cl, off), cr += __shfl_xor(cr, off);
  if ((threadIdx.x & 63) == 0) atomicAdd(&s_sum[0], cl), atomicAdd(&s_sum[1], cr);
  __syncthreads();
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = s_sum[0] | (s_sum[1] << 16);
}

// sum of the candidates of tiles [from, to), by the whole workgroup: {left, right}
__device__ inline void sum_tiles(const uint32_t *tile_counts, int from, int to, uint32_t *s_sum,
                                 uint32_t &left, uint32_t &right) {
  uint32_t cl = 0, cr = 0;
  for (int t = from + (int)threadIdx.x; t < to; t += kBigWG) {
    const uint32_t c = tile_counts[t];
    cl += c & 0xffffu, cr += c >> 16;
  }
  for (int off = 32; off > 0; off >>= 1) cl += __shfl_xor(cl, off), cr += __shfl_xor(cr, off);
  if (threadIdx.x == 0) s_sum[0] = s_sum[1] = 0;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) atomicAdd(&s_sum[0], cl), atomicAdd(&s_sum[1], cr);
  __syncthreads();
  left = s_sum[0], right = s_sum[1];
  __syncthreads();
}

__global__ void __launch_bounds__(kBigWG)
k_tb_scatter(const BigRange *list, int count, const float *key, const uint32_t *tile_counts,
             uint32_t *PL, uint32_t *PR) {
  __shared__ int s_range;
  __shared__ uint32_t s_sum[2];
  __shared__ uint32_t wave_total[kBigE * (kBigWG / 64)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_range = range_of_tile(list, count, blockIdx.x);
  __syncthreads();
  const BigRange rg = list[s_range];
  const int t = blockIdx.x - rg.tile_base, first = rg.f + 1;
  uint32_t base_l, base_r;  // candidates in the range's earlier tiles
  sum_tiles(tile_counts, rg.tile_base, blockIdx.x, s_sum, base_l, base_r);
  const TileFlags fl = tile_flags(rg, t, key);
  uint32_t rank[kBigE];  // candidates before this lane in its wave: left | right << 16
  const unsigned long long below = (1ull << lane) - 1;
  for (int e = 0; e < kBigE; e++) {
    const unsigned long long ml = __ballot((fl.left >> e) & 1), mr = __ballot((fl.right >> e) & 1);
    rank[e] = (uint32_t)__popcll(ml & below) | ((uint32_t)__popcll(mr & below) << 16);
    if (lane == 0) wave_total[e * (kBigWG / 64) + wave] = (uint32_t)__popcll(ml) | ((uint32_t)__popcll(mr) << 16);
  }
  __syncthreads();
  // exclusive prefix over the 64 (sub-trip, wave) totals, in position order, by every wave
  const uint32_t own = wave_total[lane];
  uint32_t incl = own;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(incl, off);
    if (lane >= off) incl += up;
  }
  const uint32_t excl = incl - own;
  for (int e = 0; e < kBigE; e++) {
    const uint32_t before = __shfl(excl, e * (kBigWG / 64) + wave) + rank[e];
    const uint32_t p = (uint32_t)(first + t * kTile + e * kBigWG + tid);
    if ((fl.left >> e) & 1) PL[first + (int)base_l + (int)(before & 0xffffu)] = p;
    if ((fl.right >> e) & 1) PR[first + (int)base_r + (int)(before >> 16)] = p;
  }
}

// PR holds the right candidates left to right: the k-th from the right is PR[first + cr - 1 - k]
__global__ void __launch_bounds__(kBigWG)
k_tb_swap(BigRange *list, int count, float *key, uint32_t *idx, const uint32_t *tile_counts,
          const uint32_t *PL, const uint32_t *PR) {
  __shared__ int s_range;
  __shared__ uint32_t s_sum[2];
  const int tid = threadIdx.x;
  if (tid == 0) s_range = range_of_tile(list, count, blockIdx.x);
  __syncthreads();
  // (other tiles of the range write its K, cl and cr meanwhile: only the fields set when it was queued are read)
  const int rf = list[s_range].f, rl = list[s_range].l, tile_base = list[s_range].tile_base;
  const int t = blockIdx.x - tile_base, first = rf + 1;
  const int tiles = (rl - first + kTile - 1) / kTile;
  uint32_t ucl, ucr;
  sum_tiles(tile_counts, tile_base, tile_base + tiles, s_sum, ucl, ucr);
  const int cl = (int)ucl, cr = (int)ucr, m = cl < cr ? cl : cr;
  if (t == 0 && tid == 0) list[s_range].cl = cl, list[s_range].cr = cr;
  for (int e = 0; e < kBigE; e++) {
    const int k0 = t * kTile + e * kBigWG + tid;
    if (k0 >= m) break;
    const uint32_t a = PL[first + k0], b = PR[first + cr - 1 - k0];
    const bool sw = a < b;
    if (sw) {
      const float ka = key[a], kb = key[b];
      const uint32_t ia = idx[a], ib = idx[b];
      key[a] = kb, key[b] = ka, idx[a] = ib, idx[b] = ia;
    }
    const bool next = (k0 + 1 < m) && PL[first + k0 + 1] < PR[first + cr - 2 - k0];
    if (sw && !next) list[s_range].K = k0 + 1;  // the last swap; K stays 0 when there is none
  }
}

__global__ void __launch_bounds__(256)
k_tb_finish(const BigRange *list, int count, int to, SortQueues q, float *key, uint32_t *idx,
            const uint32_t *PL, const uint32_t *PR) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= count) return;
  const BigRange rg = list[r];
  const int first = rg.f + 1, K = rg.K;
  uint32_t cut = 0xffffffffu;  // min(L_{K+1}, R_K)
  if (K < rg.cl) cut = PL[first + K];
  if (K >= 1) {
    const uint32_t rk = PR[first + rg.cr - K];
    cut = rk < cut ? rk : cut;
  }
  const int c = (int)cut, d = rg.depth - 1;
  const int parts[2][2] = {{c, rg.l}, {rg.f, c}};
  for (int h = 0; h < 2; h++) {
    const int size = parts[h][1] - parts[h][0];
    if (size > kT) push_big(q, to, GlobalRecords{key, idx}, parts[h][0], parts[h][1], d);
    else if (size >= 2) push_small(q, parts[h][0], parts[h][1], d);
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
k_tb_subtree(int n, int Ls, uint32_t *idx, const float *box, NodeOut out, uint32_t *root_order) {
  __shared__ LdsSort S;
  __shared__ uint32_t gid[kT];                // object of a slot; S.idx holds slots
  __shared__ uint8_t axis_of[kT];             // by range start
  __shared__ unsigned long long planes[6][64];
  const int kb = blockIdx.x, tid = threadIdx.x;
  int LO, m, root_id;
  range_of_index(n, Ls, kb, LO, m, root_id);
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
        int k, lo, s, rel;
        subrange_of_position(m, j, p, k, lo, s, rel);
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
      int k, lo, s, rel;
      if (!subrange_of_position(m, j, p, k, lo, s, rel) || p != lo) continue;
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
      const int id = root_id + rel;
      for (int a = 0; a < 3; a++) out.bmin[a][id] = bl[a], out.bmax[a][id] = bh[a];
      if (s == 1) {
        out.left[id] = -1, out.right[id] = -1;
        out.object[id] = (int32_t)gid[S.idx[p]];
        axis_of[p] = 3;
      } else {
        out.left[id] = id + 1, out.right[id] = id + 2 * (s >> 1);
        out.object[id] = -1;
        axis_of[p] = (uint8_t)longest_axis(bl, bh);
      }
    }
    __syncthreads();
    for (int p = tid; p < m; p += kWG) {
      int k, lo, s, rel;
      const bool alive = subrange_of_position(m, j, p, k, lo, s, rel);
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
  int Ls = 0;
  while (((n + (1 << Ls) - 1) >> Ls) > kT) Ls++;             // first level whose ranges fit a workgroup
  int rc = RTR_OK;
  float *d_box = nullptr, *d_key = nullptr, *d_nodes = nullptr;
  uint32_t *d_idx = nullptr, *d_pl = nullptr, *d_pr = nullptr, *d_root = nullptr, *d_flags = nullptr;
  int32_t *d_links = nullptr;
  unsigned long long *d_planes = nullptr;
  char *d_all = nullptr;
  SortQueues q;
  memset(&q, 0, sizeof(q));
  NodeOut out;
  const size_t list_cap = (size_t)n / 2 + 2;
  const int grid_n = (n + 255) / 256;
  // RTR_TREE_TIMES=1: where the call's time goes, on stderr (a development aid)
  const bool times = getenv("RTR_TREE_TIMES") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto mark = [&](const char *what) {
    if (!times) return;
    (void)hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "rtr_build_bvh %-12s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  if (device >= 0 && hipSetDevice(device) != hipSuccess)
    return rtr::fail_msg(RTR_ERR_DEVICE, "rtr_build_bvh: no such device");
  // one allocation for everything (a dozen separate ones cost more than the kernels of a small tree)
  {
    const size_t big_cap = (size_t)n / kT + 2, tile_cap = (size_t)n / kTile + big_cap;
    size_t total = 0;
    auto take = [&](size_t bytes) {
      const size_t at = total;
      total += (bytes + 255) & ~(size_t)255;
      return at;
    };
    const size_t o_box = take(sizeof(float) * 6 * n), o_key = take(sizeof(float) * n);
    const size_t o_idx = take(sizeof(uint32_t) * n), o_root = take(sizeof(uint32_t) * n);
    const size_t o_nodes = take(sizeof(float) * 6 * n_nodes), o_links = take(sizeof(int32_t) * 3 * n_nodes);
    const size_t o_flags = take(sizeof(uint32_t));
    size_t o_pl = 0, o_pr = 0, o_planes = 0, o_big[2] = {0, 0}, o_bigc = 0, o_small = 0, o_smallc = 0, o_tiles = 0;
    if (Ls > 0) {
      o_pl = take(sizeof(uint32_t) * n), o_pr = take(sizeof(uint32_t) * n);
      o_planes = take(sizeof(unsigned long long) * 6 * ((size_t)1 << (Ls - 1)));
      o_big[0] = take(sizeof(BigRange) * big_cap), o_big[1] = take(sizeof(BigRange) * big_cap);
      o_bigc = take(sizeof(unsigned long long) * 2);
      o_small = take(sizeof(QsRange) * list_cap), o_smallc = take(sizeof(int));
      o_tiles = take(sizeof(uint32_t) * tile_cap);
    }
    TB_TRY(hipMalloc((void **)&d_all, total));
    d_box = (float *)(d_all + o_box), d_key = (float *)(d_all + o_key);
    d_idx = (uint32_t *)(d_all + o_idx), d_root = (uint32_t *)(d_all + o_root);
    d_nodes = (float *)(d_all + o_nodes), d_links = (int32_t *)(d_all + o_links);
    d_flags = (uint32_t *)(d_all + o_flags);
    if (Ls > 0) {
      d_pl = (uint32_t *)(d_all + o_pl), d_pr = (uint32_t *)(d_all + o_pr);
      d_planes = (unsigned long long *)(d_all + o_planes);
      q.big[0] = (BigRange *)(d_all + o_big[0]), q.big[1] = (BigRange *)(d_all + o_big[1]);
      q.big_count = (unsigned long long *)(d_all + o_bigc);
      q.small = (QsRange *)(d_all + o_small), q.small_count = (int *)(d_all + o_smallc);
      q.tile_counts = (uint32_t *)(d_all + o_tiles);
      q.flags = d_flags;
    }
  }
  TB_TRY(hipMemset(d_flags, 0, sizeof(uint32_t)));
  for (int a = 0; a < 3; a++) {
    TB_TRY(hipMemcpy(d_box + (size_t)a * n, obj_bmin[a], sizeof(float) * n, hipMemcpyHostToDevice));
    TB_TRY(hipMemcpy(d_box + (size_t)(3 + a) * n, obj_bmax[a], sizeof(float) * n, hipMemcpyHostToDevice));
    out.bmin[a] = d_nodes + (size_t)a * n_nodes;
    out.bmax[a] = d_nodes + (size_t)(3 + a) * n_nodes;
  }
  out.left = d_links, out.right = d_links + n_nodes, out.object = d_links + 2 * (size_t)n_nodes;
  mark("alloc+upload");
  k_tb_init<<<grid_n, 256>>>(n, d_box, d_idx, d_flags);
  {
    uint32_t bad = 0;
    TB_TRY(hipMemcpy(&bad, d_flags, sizeof(bad), hipMemcpyDeviceToHost));
    if (bad) {
      rc = rtr::fail_msg(RTR_ERR_INVALID, "rtr_build_bvh: a box plane is infinite or not a number");
      goto done;
    }
  }
  for (int L = 0; L < Ls; L++) {
    const int ranges = 1 << L;
    k_tb_fill64<<<(6 * ranges + 255) / 256, 256>>>(d_planes, 3 * ranges, 6 * ranges);
    k_tb_level_boxes<<<(n + kBoxWG - 1) / kBoxWG, kBoxWG>>>(n, L, d_idx, d_box, d_planes);
    TB_TRY(hipMemsetAsync(q.big_count, 0, sizeof(unsigned long long) * 2, 0));
    TB_TRY(hipMemsetAsync(q.small_count, 0, sizeof(int), 0));
    k_tb_level_keys<<<grid_n, 256>>>(n, L, d_idx, d_box, d_planes, d_key, out);
    k_tb_level_ranges<<<(ranges + 255) / 256, 256>>>(n, L, q, d_idx, d_key);
    int cur = 0;
    for (int round = 0;; round++) {
      unsigned long long queued = 0;
      TB_TRY(hipMemcpy(&queued, q.big_count + cur, sizeof(queued), hipMemcpyDeviceToHost));
      const int n_ranges = (int)(queued >> 32), n_tiles = (int)(queued & 0xffffffffu);
      if (n_ranges == 0) break;
      if (round > 2 * 26 + 2) {
        rc = rtr::fail_msg(RTR_ERR_DEVICE, "rtr_build_bvh: the sort did not end within its depth budget");
        goto done;
      }
      TB_TRY(hipMemsetAsync(q.big_count + (1 - cur), 0, sizeof(unsigned long long), 0));
      k_tb_count<<<n_tiles, kBigWG>>>(q.big[cur], n_ranges, d_key, q.tile_counts);
      k_tb_scatter<<<n_tiles, kBigWG>>>(q.big[cur], n_ranges, d_key, q.tile_counts, d_pl, d_pr);
      k_tb_swap<<<n_tiles, kBigWG>>>(q.big[cur], n_ranges, d_key, d_idx, q.tile_counts, d_pl, d_pr);
      k_tb_finish<<<(n_ranges + 255) / 256, 256>>>(q.big[cur], n_ranges, 1 - cur, q, d_key, d_idx, d_pl, d_pr);
      cur = 1 - cur;
    }
    int n_small = 0;
    TB_TRY(hipMemcpy(&n_small, q.small_count, sizeof(int), hipMemcpyDeviceToHost));
    if (n_small > 0) k_tb_sort_small<<<n_small, kWG>>>(q.small, d_idx, d_key);
    if (L == 0) TB_TRY(hipMemcpyAsync(d_root, d_idx, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, 0));
  }
  mark("upper levels");
  k_tb_subtree<<<1 << Ls, kWG>>>(n, Ls, d_idx, d_box, out, d_root);
  TB_TRY(hipGetLastError());
  {
    uint32_t flags = 0;
    TB_TRY(hipMemcpy(&flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost));
    if (flags & 2u) {
      rc = rtr::fail_msg(RTR_ERR_UNSUPPORTED,
                         "rtr_build_bvh: the reference's sort heap-sorts a range of more than 2^17 objects "
                         "for these boxes, which is not done on the device");
      goto done;
    }
  }
  mark("subtrees");
  for (int a = 0; a < 3; a++) {
    TB_TRY(hipMemcpy(node_bmin[a], out.bmin[a], sizeof(float) * n_nodes, hipMemcpyDeviceToHost));
    TB_TRY(hipMemcpy(node_bmax[a], out.bmax[a], sizeof(float) * n_nodes, hipMemcpyDeviceToHost));
  }
  TB_TRY(hipMemcpy(node_left, out.left, sizeof(int32_t) * n_nodes, hipMemcpyDeviceToHost));
  TB_TRY(hipMemcpy(node_right, out.right, sizeof(int32_t) * n_nodes, hipMemcpyDeviceToHost));
  TB_TRY(hipMemcpy(node_object, out.object, sizeof(int32_t) * n_nodes, hipMemcpyDeviceToHost));
  if (root_order) TB_TRY(hipMemcpy(root_order, d_root, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
  mark("download");
done:
  (void)hipFree(d_all);
  mark("free");
  return rc;
}
