// Scene upload: from the arrays of rtr_scene_desc to the records the traversal kernels read,
// on the device.
//
// The caller's tree is the reference's pointer tree (struct BVHNode, include/BVH.h:9-22) as
// index arrays, numbered in any order.  The kernels want it in depth-first pre-order with
// skip links, every node's parent, one record per leaf in depth-first leaf order, and
// primitives as float4 records.  None of that needs a sequential walk:
//
//   * a node's position in pre-order is the number of nodes visited before it = on the way
//     from the root down to it, one per step, plus the whole left subtree whenever the step
//     goes right.  In a tree whose inner nodes all have two children a subtree with L leaves
//     has 2L - 1 nodes, so "leaves under every node" is all that has to be known: counts climb
//     from the leaves, the second child to reach a node carrying the sum on (k_build_count), then
//     every node walks up once and sums (k_build_place).  The same walk gives a leaf's number among
//     the leaves.  Trees are at most 64 levels deep, so a walk is short.
//   * skip link = position + 2L - 1; parent = position of the parent (k_build_parent).
//
// What is malformed about a tree (a child index out of range, a node with two parents, a
// leaf without a primitive, nodes the root does not reach, more than 64 levels) is found on
// the way and reported as bits of BuildFlags::errors.
#pragma once
#include "rtr_shade.hpp"

namespace rtr {

enum : uint32_t {
  kBuildChildRange = 1u,    // child index out of range
  kBuildTwoParents = 2u,    // node reachable twice (not a tree)
  kBuildLeafPrim = 4u,      // leaf without a valid prim
  kBuildOneChild = 8u,      // inner node with one child
  kBuildUnreachable = 16u,  // unreachable nodes in the tree
  kBuildTooDeep = 32u,      // tree deeper than 64 levels
  kBuildPrimIndex = 64u,    // prim_index out of range
  kBuildPrimKind = 128u,    // unknown prim_kind
  kBuildPrimMaterial = 256u // prim_material out of range
};

struct BuildFlags {
  uint32_t errors;
  uint32_t slow_planes;  // != 0: some box plane is outside the division-free test's range
  uint32_t not_nested;   // != 0: some box does not contain its child's box
  uint32_t n_items, n_pairs;  // trace_all2's record table (small scenes)
  uint32_t far_prims;    // != 0: a vertex coordinate or a sphere's extent beyond 2^20 in magnitude (or not a number)
};

// The description's arrays in device memory (copies, freed after the build).
struct BuildInput {
  const float *bmin[3], *bmax[3];
  const int32_t *left, *right, *prim;
  const int32_t *prim_kind, *prim_index, *prim_material;
  const float *p0[3], *p1[3], *p2[3], *tn[3], *tri_area;
  const float *sc[3], *sr;
  int32_t n_nodes, root, n_prims, n_tris, n_sphs, n_materials;
};

struct BuildTemp {
  int32_t *parent_old;     // by caller's node index; -1: none (yet)
  uint32_t *leaves_under;  // by caller's node index
  int32_t *place;          // caller's node index -> pre-order position
};

struct BuildOutput {
  float4 *nodes, *tris, *sphs, *leaves;
  int32_t *parent, *node_info, *tri_prim, *sph_prim, *prim_kind_row, *prim_material;
  float *tri_area;
  const DeviceMaterial *mats;
  int32_t info_fits;  // prim ids and material ids fit node_info's bit fields
};

constexpr int kBuildBlock = 256;

RTR_DEV void build_error(BuildFlags *f, uint32_t bit) { atomicOr(&f->errors, bit); }

// prim id -> (kind << 30) | row and material; triangle / sphere row -> prim id
__global__ void __launch_bounds__(kBuildBlock)
k_build_prims(BuildInput in, BuildOutput out, BuildFlags *flags) {
  const int p = blockIdx.x * kBuildBlock + threadIdx.x;
  if (p >= in.n_prims) return;
  const int kind = in.prim_kind[p], row = in.prim_index[p], m = in.prim_material[p];
  if (kind == 0) {
    if (row < 0 || row >= in.n_tris) return build_error(flags, kBuildPrimIndex);
    out.tri_prim[row] = p;
  } else if (kind == 1) {
    if (row < 0 || row >= in.n_sphs) return build_error(flags, kBuildPrimIndex);
    out.sph_prim[row] = p;
  } else {
    return build_error(flags, kBuildPrimKind);
  }
  if (m < 0 || m >= in.n_materials) return build_error(flags, kBuildPrimMaterial);
  out.prim_kind_row[p] = (kind << 30) | row;
  out.prim_material[p] = m;
}

// primitives as records: {p0, n.x} {p1, n.y} {p2, n.z}; {c, r}
__global__ void __launch_bounds__(kBuildBlock)
k_build_geometry(BuildInput in, BuildOutput out, BuildFlags *flags) {
  const int t = blockIdx.x * kBuildBlock + threadIdx.x;
  // |v| <= 2^20, false for a NaN
  auto near = [](float v) { return fabsf(v) <= 1048576.0f; };
  bool ok = true;
  if (t < in.n_tris) {
    out.tris[3 * (size_t)t] = make_float4(in.p0[0][t], in.p0[1][t], in.p0[2][t], in.tn[0][t]);
    out.tris[3 * (size_t)t + 1] = make_float4(in.p1[0][t], in.p1[1][t], in.p1[2][t], in.tn[1][t]);
    out.tris[3 * (size_t)t + 2] = make_float4(in.p2[0][t], in.p2[1][t], in.p2[2][t], in.tn[2][t]);
    out.tri_area[t] = in.tri_area[t];
    for (int a = 0; a < 3; a++) ok = ok && near(in.p0[a][t]) && near(in.p1[a][t]) && near(in.p2[a][t]);
  }
  if (t < in.n_sphs) {
    out.sphs[t] = make_float4(in.sc[0][t], in.sc[1][t], in.sc[2][t], in.sr[t]);
    for (int a = 0; a < 3; a++) ok = ok && near(fabsf(in.sc[a][t]) + fabsf(in.sr[t]));
  }
  if (!ok) flags->far_prims = 1u;
}

// every node names its children's parent; what is wrong with a node on its own
__global__ void __launch_bounds__(kBuildBlock)
k_build_links(BuildInput in, BuildTemp tmp, BuildFlags *flags) {
  const int i = blockIdx.x * kBuildBlock + threadIdx.x;
  if (i >= in.n_nodes) return;
  const int l = in.left[i], r = in.right[i];
  if (l < 0 && r < 0) {
    const int p = in.prim[i];
    if (p < 0 || p >= in.n_prims) build_error(flags, kBuildLeafPrim);
    tmp.leaves_under[i] = 1u;
    return;
  }
  if (l < 0 || r < 0) return build_error(flags, kBuildOneChild);
  if (l >= in.n_nodes || r >= in.n_nodes) return build_error(flags, kBuildChildRange);
  if (l == in.root || r == in.root) build_error(flags, kBuildTwoParents);
  if (atomicCAS(&tmp.parent_old[l], -1, i) != -1) build_error(flags, kBuildTwoParents);
  if (atomicCAS(&tmp.parent_old[r], -1, i) != -1) build_error(flags, kBuildTwoParents);
}

// Leaves under every node, bottom up: a leaf climbs with its count of one; at every node the first
// child to arrive leaves its count and stops, the second takes the sum and climbs on -- one atomic
// addition per arrival and no word that more than two threads touch (every leaf adding itself to
// every ancestor put a million additions on the root's word alone).  The addition carries the
// count in its low 31 bits and an arrival mark in the top bit: the first arrival finds zero, the
// second finds the first one's count under a set top bit, and its own mark on top of that carries
// out of the word, which is left holding the plain sum.
// A node that only one child reaches (a cycle above it, a subtree the root does not reach) is
// never completed: the count at the root then falls short, which the caller checks.
__global__ void __launch_bounds__(kBuildBlock)
k_build_count(BuildInput in, BuildTemp tmp, BuildFlags *flags) {
  const int i = blockIdx.x * kBuildBlock + threadIdx.x;
  if (i >= in.n_nodes) return;
  if (!(in.left[i] < 0 && in.right[i] < 0)) return;
  int c = i;
  uint32_t count = 1u;
  for (int steps = 0; c != in.root; steps++) {
    const int p = tmp.parent_old[c];
    if (p < 0) return build_error(flags, kBuildUnreachable);
    // (also what ends a climb round a cycle of nodes that name each other as children)
    if (steps >= kMaxDepth - 1) return build_error(flags, kBuildTooDeep);
    const uint32_t before = atomicAdd(&tmp.leaves_under[p], count | 0x80000000u);
    if (before == 0u) return;
    count += before & 0x7fffffffu;
    c = p;
  }
}

// Smallest float T with sqrtf(T) >= r, so that sqrtf(x) < r  <=>  x < T for every x (sqrtf is
// correctly rounded, hence monotonic).  r <= 0 or NaN: never true -> T = 0.
RTR_DEV float sqrt_threshold(float r) {
  if (!(r > 0.0f)) return 0.0f;
  if (r == __int_as_float(0x7f800000)) return r;
  float t = (float)((double)r * (double)r);
  while (sqrtf(t) >= r && t > 0.0f) t = __uint_as_float(__float_as_uint(t) - 1u);
  while (sqrtf(t) < r) t = __uint_as_float(__float_as_uint(t) + 1u);
  return t;
}

// every node finds its place, writes its record; leaves write theirs
__global__ void __launch_bounds__(kBuildBlock)
k_build_place(BuildInput in, BuildTemp tmp, BuildOutput out, BuildFlags *flags) {
  const int i = blockIdx.x * kBuildBlock + threadIdx.x;
  if (i >= in.n_nodes) return;
  uint32_t at = 0, leaf_no = 0;
  int c = i;
  for (int steps = 0; c != in.root; steps++) {
    const int p = tmp.parent_old[c];
    if (p < 0 || steps >= kMaxDepth - 1) {  // not under the root, or 64 levels and more below it
      build_error(flags, p < 0 ? kBuildUnreachable : kBuildTooDeep);
      tmp.place[i] = -1;
      return;
    }
    if (in.right[p] == c) {
      const uint32_t lu = tmp.leaves_under[in.left[p]];
      at += 2u * lu;  // the parent, and the 2 lu - 1 nodes of its left subtree
      leaf_no += lu;
    } else {
      at += 1u;
    }
    c = p;
  }
  tmp.place[i] = (int32_t)at;
  const bool is_leaf = in.left[i] < 0 && in.right[i] < 0;
  const uint32_t size = 2u * tmp.leaves_under[i] - 1u;
  const int32_t skip = (int32_t)(at + size);
  float4 lo = make_float4(in.bmin[0][i], in.bmin[1][i], in.bmin[2][i], __int_as_float(skip));
  int32_t payload = -1;
  if (is_leaf) {
    const int p = in.prim[i];
    if (p >= 0 && p < in.n_prims) payload = (in.prim_kind[p] << 30) | in.prim_index[p];
  }
  float4 hi = make_float4(in.bmax[0][i], in.bmax[1][i], in.bmax[2][i], __int_as_float(payload));
  const float planes[6] = {lo.x, lo.y, lo.z, hi.x, hi.y, hi.z};
  bool fast = true;
  for (float v : planes) fast = fast && fast_coordinate(v);
  if (!fast) flags->slow_planes = 1u;
  out.nodes[2 * (size_t)at] = lo;
  out.nodes[2 * (size_t)at + 1] = hi;
  int32_t info = 0;
  if (is_leaf && payload >= 0) {
    // the leaf's record (rtr_trace.hpp: trace_all)
    const int row = payload & 0x3fffffff;
    float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
    if (payload >> 30) {
      if (row < in.n_sphs) {
        const float4 q = make_float4(in.sc[0][row], in.sc[1][row], in.sc[2][row], in.sr[row]);
        r0 = q;
        r1.x = q.w * q.w;
        r1.y = sqrt_threshold(q.w);
      }
    } else if (row < in.n_tris) {
      const float4 q0 = make_float4(in.p0[0][row], in.p0[1][row], in.p0[2][row], in.tn[0][row]);
      const float4 q1 = make_float4(in.p1[0][row], in.p1[1][row], in.p1[2][row], in.tn[1][row]);
      const float4 q2 = make_float4(in.p2[0][row], in.p2[1][row], in.p2[2][row], in.tn[2][row]);
      r0 = q0;
      r1 = make_float4(q1.x - q0.x, q1.y - q0.y, q1.z - q0.z, q1.w);
      r2 = make_float4(q2.x - q0.x, q2.y - q0.y, q2.z - q0.z, q2.w);
    }
    r3.z = __int_as_float((int32_t)at);
    r3.w = __int_as_float(payload);
    float4 *rec = out.leaves + 4 * (size_t)leaf_no;
    rec[0] = r0; rec[1] = r1; rec[2] = r2; rec[3] = r3;
    if (out.info_fits) {
      // prim id | material << 16 | emissive << 30 (small scenes, see hit_info)
      const int p = in.prim[i];
      const int m = in.prim_material[p];
      if (m >= 0 && m < in.n_materials)
        info = p | (m << 16) | ((int32_t)mat_emissive(out.mats[m]) << 30);
    }
  }
  out.node_info[at] = info;
}

// parents in the new numbering; does every box contain its children's boxes, plane by plane?
__global__ void __launch_bounds__(kBuildBlock)
k_build_parent(BuildInput in, BuildTemp tmp, BuildOutput out, BuildFlags *flags) {
  const int i = blockIdx.x * kBuildBlock + threadIdx.x;
  if (i >= in.n_nodes) return;
  const int at = tmp.place[i];
  if (at < 0) return;
  if (i == in.root) {
    out.parent[at] = -1;
    return;
  }
  const int p = tmp.parent_old[i];
  out.parent[at] = tmp.place[p];
  const bool inside = in.bmin[0][p] <= in.bmin[0][i] && in.bmin[1][p] <= in.bmin[1][i] &&
                      in.bmin[2][p] <= in.bmin[2][i] && in.bmax[0][p] >= in.bmax[0][i] &&
                      in.bmax[1][p] >= in.bmax[1][i] && in.bmax[2][p] >= in.bmax[2][i];
  if (!inside) flags->not_nested = 1u;
}

// What only scenes small enough for LDS use; one thread walks the few leaf records.
//  * Sphere leaves are numbered in leaf order (the bit a ray marks in trace_all2).
//  * The record table of trace_all2: the leaf records again, but of a mirrored pair only the
//    first, marked with 1 + the pair's number in its first spare word; the second triangle's
//    {p0, e1, e2} go to pair_second.  Leaves k and k + 1 are a mirrored pair when both are
//    triangles with e1' = -e1, e2' = -e2 and the same normal -- the two halves of a
//    parallelogram (p0 p1 p2 / p3 p2 p1).  trace_all2 then takes the second triangle's
//    direction-dependent terms over from the first; the conditions below are the ones its
//    argument needs: non-zero components negated bit for bit, zero components +0 in both,
//    the non-zero components of magnitude in [2^-60, 2^60], normals equal as numbers.
__device__ inline bool mirrored_component(float a, float b) {
  const uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
  if (ua == 0u) return ub == 0u;
  if (ub != (ua ^ 0x80000000u)) return false;
  const float m = fabsf(a);
  return m >= 8.673617379884035e-19f && m <= 1152921504606846976.0f;  // false for -0, infinity, NaN
}
// The second triangle of a pair takes its s1 from the first one by plain negation, which gives a
// zero component the other sign than its own evaluation would.  That sign reaches nothing but b1
// when b1 is a zero, and a zero b1 of either sign reaches nothing but the hit's position
// p0 w + p1 b1 + p2 b2 (src/Trangle.cpp:23), where p1 * (+-0) is absorbed by a non-zero sum, and by
// a sum of zeros as well when no vertex coordinate is -0: w + b2 = 1 then, so one of the two other
// terms has a positive weight, and is non-zero or +0, which makes the sum +0 whatever the other
// zeros' signs.  Hence: no pair when a coordinate of the second triangle is -0.
__device__ inline bool has_negative_zero(float4 v) {
  return __float_as_uint(v.x) == 0x80000000u || __float_as_uint(v.y) == 0x80000000u ||
         __float_as_uint(v.z) == 0x80000000u;
}
__global__ void k_build_items(float4 *leaves, int n_leaves, const float4 *tris, float4 *items,
                              float4 *pair_second, BuildFlags *flags) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int32_t j = 0;
  for (int k = 0; k < n_leaves; k++)
    if (__float_as_int(leaves[4 * k + 3].w) >> 30) leaves[4 * k + 1].z = __int_as_float(j++);
  int32_t n = 0, n_pairs = 0;
  for (int k = 0; k < n_leaves; k++) {
    const float4 *a = leaves + 4 * k, *b = a + 4;
    float4 *it = items + 4 * n++;
    it[0] = a[0]; it[1] = a[1]; it[2] = a[2]; it[3] = a[3];
    if ((__float_as_int(a[3].w) >> 30) || k + 1 >= n_leaves || (__float_as_int(b[3].w) >> 30)) continue;
    const bool edges = mirrored_component(a[1].x, b[1].x) && mirrored_component(a[1].y, b[1].y) &&
                       mirrored_component(a[1].z, b[1].z) && mirrored_component(a[2].x, b[2].x) &&
                       mirrored_component(a[2].y, b[2].y) && mirrored_component(a[2].z, b[2].z);
    const bool normals = a[0].w == b[0].w && a[1].w == b[1].w && a[2].w == b[2].w;  // false for NaN
    const float4 *vb = tris + 3 * (__float_as_int(b[3].w) & 0x3fffffff);  // the second triangle's p0, p1, p2
    const bool zeros = !has_negative_zero(vb[0]) && !has_negative_zero(vb[1]) && !has_negative_zero(vb[2]);
    if (edges && normals && zeros) {
      pair_second[3 * n_pairs] = b[0];
      pair_second[3 * n_pairs + 1] = b[1];
      pair_second[3 * n_pairs + 2] = b[2];
      it[3].x = __int_as_float(++n_pairs);
      k++;
    }
  }
  for (int q = 0; q < 4; q++) items[4 * n + q] = make_float4(0, 0, 0, 0);  // read past the end
  flags->n_items = (uint32_t)n;
  flags->n_pairs = (uint32_t)n_pairs;
}

} // namespace rtr
