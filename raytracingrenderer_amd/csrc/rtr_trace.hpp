// Closest-hit traversal and primitive tests for gfx950.
//
// Layout in HBM (built by rtr_scene_create):
//   nodes : 2 x float4 per node, nodes numbered in depth-first pre-order (left
//           child first), so the left child of inner node i is i+1:
//             [2i]   = { bmin.x, bmin.y, bmin.z, bits(skip) }
//             [2i+1] = { bmax.x, bmax.y, bmax.z, bits(leaf)  }
//           skip = index of the first node after i's subtree (n_nodes at the
//           end) -- where traversal continues when i's box is missed;
//           leaf = -1 for an inner node, else (kind << 30) | row.
//   tris  : 3 x float4 per triangle: {p0, n.x} {p1, n.y} {p2, n.z}
//   sphs  : 1 x float4 per sphere:   {c, r}
// A lane gathers whole nodes, so a node is one 32-byte record (two 16-byte
// loads from one cache line) rather than eight separate arrays.
//
// Traversal is stackless: visiting nodes in pre-order and jumping over missed
// subtrees through `skip` reaches exactly the leaves the reference's recursion
// (src/BVH.cpp:66-80) reaches, in the same left-to-right order, with no
// per-lane stack at all.  The reference merges the two subtrees' results with
// "strictly nearer keeps the left one" (src/BVH.cpp:75-76); folding the leaves
// left to right with the same rule gives the same winner whenever all hit
// distances are ordered.  A hit distance can be NaN (degenerate triangle,
// src/Trangle.cpp:13-17 with s_e == 0); then the tree-shaped merge and the fold
// can differ, so such rays are re-traced by trace_tree(), which evaluates the
// merge tree literally.
#pragma once
#include "rtr_math.hpp"

namespace rtr {

struct DeviceScene {
  const float4 *nodes;
  const float4 *tris;
  const float4 *sphs;
  const int32_t *tri_prim;  // triangle row -> prim id
  const int32_t *sph_prim;  // sphere row   -> prim id
  const int32_t *prim_kind_row;  // prim id -> (kind << 30) | row
  const int32_t *prim_material;
  int32_t n_nodes;
  int32_t n_prims;
  int32_t fast_planes;  // 1: every box plane is 0 or has magnitude in [2^-20, 2^20]
  int32_t n_tris, n_sphs;
  // small scenes only (see trace_all): one record per leaf in depth-first order,
  // and every node's parent
  const float4 *leaves;
  const int32_t *parent;
  int32_t n_leaves;
  int32_t nested_boxes;  // 1: every node's box contains its children's boxes, plane by plane
  // small scenes only (see trace_all2): the leaf records without the second triangles of
  // mirrored pairs, and those triangles' {p0, e1, e2} by pair
  const float4 *items;
  const float4 *pair_second;
  int32_t n_items, n_pairs;
  int32_t near_prims;  // 1: every vertex coordinate and every sphere's extent is within 2^20 in magnitude
};

// Where traversal reads geometry from.  GlobalGeom gathers from HBM through the
// caches; LdsGeom reads a copy the workgroup staged in LDS (scenes small enough
// to fit: lower and steadier latency than the vector L1, which the streaming
// path state keeps evicting).
struct GlobalGeom {
  const float4 *nodes, *tris, *sphs;
  RTR_DEV float4 node_lo(int i) const { return nodes[2 * i]; }
  RTR_DEV float4 node_hi(int i) const { return nodes[2 * i + 1]; }
  RTR_DEV float4 tri(int row, int k) const { return tris[3 * row + k]; }
  RTR_DEV float4 sph(int row) const { return sphs[row]; }
};
struct LdsGeom {
  const float4 *nodes, *tris, *sphs;  // point into __shared__ arrays
  RTR_DEV float4 node_lo(int i) const { return nodes[2 * i]; }
  RTR_DEV float4 node_hi(int i) const { return nodes[2 * i + 1]; }
  RTR_DEV float4 tri(int row, int k) const { return tris[3 * row + k]; }
  RTR_DEV float4 sph(int row) const { return sphs[row]; }
};
RTR_DEV GlobalGeom global_geom(const DeviceScene &sc) { return GlobalGeom{sc.nodes, sc.tris, sc.sphs}; }

struct HitRec {
  float t, b1, b2;
  int32_t leaf;  // (kind << 30) | row, or -1 for a miss
};

constexpr int kMaxDepth = 64;

// (double)x < L for a float x and a double literal L  <=>  x < ceil_to_float(L)
constexpr float ceil_to_float(double L) {
  float f = (float)L;
  if ((double)f >= L) return f;
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  u = f > 0 ? u + 1 : u - 1;
  return __builtin_bit_cast(float, u);
}
constexpr float kTriEps = ceil_to_float(-1e-5);   // src/Trangle.cpp:17
constexpr float kSphEps = ceil_to_float(1e-3);    // src/Sphere.cpp:29


// Bound3::intersect, src/Bound3.cpp:3-22, with the direction already
// normalised by the caller (the reference normalises the same vector on every
// call, src/Bound3.cpp:5; the result is the same each time).
RTR_DEV bool slab(float4 lo, float4 hi, V3 o, V3 d) {
  float ax = (lo.x - o.x) / d.x, bx = (hi.x - o.x) / d.x;
  float nx = ax > bx ? bx : ax, fx = ax > bx ? ax : bx;
  float ay = (lo.y - o.y) / d.y, by = (hi.y - o.y) / d.y;
  float ny = ay > by ? by : ay, fy = ay > by ? ay : by;
  float az = (lo.z - o.z) / d.z, bz = (hi.z - o.z) / d.z;
  float nz = az > bz ? bz : az, fz = az > bz ? az : bz;
  float tnear = nx > ny ? (nx > nz ? nx : nz) : (ny > nz ? ny : nz);
  float tfar = fx < fy ? (fx < fz ? fx : fz) : (fy < fz ? fy : fz);
  return tnear <= tfar;
}

// Trangle::intersect, src/Trangle.cpp:3-28.  `dir` is the ray direction as
// given (used un-normalised by the back-face test, :5), `dn` its normalisation.
template <class Geom>
RTR_DEV bool hit_triangle(const Geom &g, int row, V3 o, V3 dir, V3 dn, float &t, float &b1,
                          float &b2) {
  float4 q0 = g.tri(row, 0), q1 = g.tri(row, 1), q2 = g.tri(row, 2);
  V3 p0 = xyz(q0), p1 = xyz(q1), p2 = xyz(q2), n = mk(q0.w, q1.w, q2.w);
  if (dot(-dir, n) < 0) return false;
  V3 e1 = p1 - p0, e2 = p2 - p0, s = o - p0;
  V3 s1 = cross(dn, e2), s2 = cross(s, e1);
  float se = dot(s1, e1);
  float inv = 1 / se;
  t = inv * dot(s2, e2);
  b1 = inv * dot(s1, s);
  b2 = inv * dot(s2, dn);
  // float operands against double literals, src/Trangle.cpp:17
  if ((double)t < -1e-5 || (double)b1 < -1e-5 || (double)b2 < -1e-5 ||
      (double)(1 - b1 - b2) < -1e-5)
    return false;
  return true;
}

// Sphere::intersect, src/Sphere.cpp:3-39.
template <class Geom>
RTR_DEV bool hit_sphere(const Geom &g, int row, V3 org, V3 d, float &t) {
  float4 q = g.sph(row);
  V3 c = xyz(q);
  float r = q.w;
  V3 o = org - c;
  if (length(o) < r) return false;
  float a = d.x * d.x + d.y * d.y + d.z * d.z;
  float b = 2 * (o.x * d.x + o.y * d.y + o.z * d.z);
  float cc = o.x * o.x + o.y * o.y + o.z * o.z - r * r;
  float delta = b * b - 4 * a * cc;
  if (delta < 0) return false;
  float sq = sqrtf(delta);
  float t1 = (-b + sq) / 2 / a;
  float t2 = (-b - sq) / 2 / a;
  if ((double)t1 < 1e-3) return false;
  t = (t2 < 0) ? t1 : t2;
  return true;
}

template <class Geom>
RTR_DEV bool hit_leaf(const Geom &g, int leaf, V3 o, V3 dir, V3 dn, HitRec &h) {
  int row = leaf & 0x3fffffff;
  h.leaf = leaf;
  if (leaf >> 30) {
    h.b1 = 0;
    h.b2 = 0;
    return hit_sphere(g, row, o, dn, h.t);
  }
  return hit_triangle(g, row, o, dir, dn, h.t, h.b1, h.b2);
}

// Stack of the literal merge-tree evaluation below: one per wave, in LDS.  It is
// only touched by rays that saw a NaN hit distance, one lane at a time, so it
// costs no registers and no scratch memory on the common path.
struct MergeStack {
  int node[kMaxDepth];
  float lt[kMaxDepth], lb1[kMaxDepth], lb2[kMaxDepth];
  int lleaf[kMaxDepth];
  unsigned char phase[kMaxDepth];
};

// Literal evaluation of the reference's merge tree (src/BVH.cpp:66-80) with an
// explicit stack; only used for rays on which a NaN hit distance was seen.
template <class Geom>
RTR_DEV HitRec trace_tree(const Geom &g, V3 o, V3 dir, V3 dn, MergeStack &st) {
  // frame: node index, phase (0 = enter, 1 = left done, 2 = right done), left result
  int sp = 0;
  st.node[0] = 0;
  st.phase[0] = 0;
  HitRec ret;
  ret.leaf = -1;
  ret.t = 0; ret.b1 = 0; ret.b2 = 0;
  while (sp >= 0) {
    int i = st.node[sp];
    int ph = st.phase[sp];
    if (ph == 0) {
      float4 lo = g.node_lo(i), hi = g.node_hi(i);
      int leaf = (int)wbits(hi);
      ret.leaf = -1;
      if (!slab(lo, hi, o, dn)) { sp--; continue; }
      if (leaf >= 0) {
        HitRec h;
        if (hit_leaf(g, leaf, o, dir, dn, h)) ret = h;
        sp--;
        continue;
      }
      if (sp + 1 >= kMaxDepth) { ret.leaf = -1; return ret; }  // rejected at scene creation
      st.phase[sp] = 1;
      st.node[sp + 1] = i + 1;
      st.phase[sp + 1] = 0;
      sp++;
    } else if (ph == 1) {
      st.lt[sp] = ret.t; st.lb1[sp] = ret.b1; st.lb2[sp] = ret.b2; st.lleaf[sp] = ret.leaf;
      st.phase[sp] = 2;
      st.node[sp + 1] = (int)wbits(g.node_lo(i + 1));  // right child = skip of the left child
      st.phase[sp + 1] = 0;
      sp++;
    } else {
      HitRec l{st.lt[sp], st.lb1[sp], st.lb2[sp], st.lleaf[sp]}, r = ret, out;
      out.leaf = -1; out.t = 0; out.b1 = 0; out.b2 = 0;
      out = (l.leaf >= 0) ? l : out;
      out = (r.leaf >= 0) ? ((out.leaf >= 0) ? ((out.t < r.t) ? out : r) : r) : out;
      ret = out;
      sp--;
    }
  }
  return ret;
}

// Bound3::intersect without IEEE division instructions, same boolean.
//
// The reference divides six times per node and an IEEE fp32 division costs as much as
// seventeen plain instructions on gfx950.  With y = RN(1/d) formed once per ray and axis,
// each quotient is rebuilt by ONE residual correction:
//   q0 = RN(a*y); r = fma(-d, q0, a) (exact); q = fma(r, y, q0)  =  RN(a/d).
// That this is the correctly rounded quotient whenever y is the correctly rounded reciprocal
// and nothing under- or overflows on the way is not argued here but was checked for every pair
// of significands there is -- 2^46 pairs, tools/ubench/exact_division.hip, no mismatch
// (rounding is symmetric in sign and, away from under- and overflow, sees significands only) --
// so min/max of the six quotients and the final compare give the reference's result bit for
// bit.  The fma calls are explicit; contraction stays off for everything else.
//
// Preconditions (else the IEEE-division test above is used): direction
// components finite with magnitude in [2^-20, 2]; origin components and box planes
// 0 or of magnitude in [2^-20, 2^20].  Then every a = plane - o is 0 or at least
// 2^-44, quotients lie in [2^-45, 2^41] and the residuals stay normal numbers.
// All values are finite, so plain min/max equal the reference's swap-and-select.
// (A zero quotient may come out as +0 where IEEE gives -0; the two compare equal in
// every min, max and <= of the test, so its boolean is unaffected.)
// (div_corrected and reciprocal_fast themselves: rtr_math.hpp)
// Two quotients at a time: gfx950 issues packed fp32 multiply / fma (two IEEE
// operations per lane and instruction) at the rate of the scalar forms, and the
// two planes of an axis share d and y.
typedef float f32x2 __attribute__((ext_vector_type(2)));
RTR_DEV f32x2 div_corrected2(f32x2 a, float d, float y) {
  const f32x2 yy = {y, y}, nd = {-d, -d};
  const f32x2 q = a * yy;
  const f32x2 r = __builtin_elementwise_fma(nd, q, a);
  return __builtin_elementwise_fma(r, yy, q);
}

// 1.0f / d, by the short sequence when every lane of the wave may take it
RTR_DEV f32x2 reciprocal_pair(f32x2 d) {
  const bool slow = !(reciprocal_operand_fast(d.x) && reciprocal_operand_fast(d.y));
  if (__ballot(slow) == 0) return f32x2{reciprocal_fast(d.x), reciprocal_fast(d.y)};
  return f32x2{1 / d.x, 1 / d.y};
}

RTR_DEV bool slab_fast(float4 lo, float4 hi, V3 o, V3 d, V3 y) {
  const f32x2 px = {lo.x, hi.x}, py = {lo.y, hi.y}, pz = {lo.z, hi.z};
  const f32x2 ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
  const f32x2 qx = div_corrected2(px - ox, d.x, y.x);
  const f32x2 qy = div_corrected2(py - oy, d.y, y.y);
  const f32x2 qz = div_corrected2(pz - oz, d.z, y.z);
  float tn = fmaxf(fmaxf(fminf(qx.x, qx.y), fminf(qy.x, qy.y)), fminf(qz.x, qz.y));
  float tf = fminf(fminf(fmaxf(qx.x, qx.y), fmaxf(qy.x, qy.y)), fmaxf(qz.x, qz.y));
  return tn <= tf;
}

constexpr float kFastLo = 9.5367431640625e-07f;  // 2^-20
constexpr float kFastHi = 1048576.0f;            // 2^20
RTR_DEV bool fast_coordinate(float v) {
  float a = fabsf(v);
  return a == 0.0f || (a >= kFastLo && a <= kFastHi);
}
RTR_DEV bool fast_ray(V3 o, V3 dn) {
  return fast_coordinate(o.x) && fast_coordinate(o.y) && fast_coordinate(o.z) &&
         fabsf(dn.x) >= kFastLo && fabsf(dn.y) >= kFastLo && fabsf(dn.z) >= kFastLo &&
         fabsf(dn.x) <= 2.0f && fabsf(dn.y) <= 2.0f && fabsf(dn.z) <= 2.0f;
}

// slab_fast behind a cheap filter.  With y = RN(1/d), q' = RN(a*y) differs from the IEEE
// quotient q = RN(a/d) by at most 3 ulp: |q' - q| <= 2^-22 |q'| (y and the product each
// carry one rounding, q another; nothing under- or overflows under the preconditions
// of slab_fast).  tnear and tfar are a max and a min of such values, so each is within
// 2^-22 m of the reference's, m being the largest of the six |q'|.  If the computed gap
// tfar' - tnear' exceeds 2^-20 m in magnitude, the reference's "tnear <= tfar" has the
// sign of that gap; otherwise (a few tests in a million) the exact sequence decides.
RTR_DEV bool slab_filtered(float4 lo, float4 hi, V3 o, V3 d, V3 y) {
  const f32x2 px = {lo.x, hi.x}, py = {lo.y, hi.y}, pz = {lo.z, hi.z};
  const f32x2 ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
  const f32x2 yx = {y.x, y.x}, yy = {y.y, y.y}, yz = {y.z, y.z};
  const f32x2 qx = (px - ox) * yx, qy = (py - oy) * yy, qz = (pz - oz) * yz;
  const float nx = fminf(qx.x, qx.y), ny = fminf(qy.x, qy.y), nz = fminf(qz.x, qz.y);
  const float fx = fmaxf(qx.x, qx.y), fy = fmaxf(qy.x, qy.y), fz = fmaxf(qz.x, qz.y);
  const float tn = fmaxf(fmaxf(nx, ny), nz), tf = fminf(fminf(fx, fy), fz);
  const float m = fmaxf(fmaxf(fmaxf(fabsf(nx), fabsf(ny)), fabsf(nz)),
                        fmaxf(fmaxf(fabsf(fx), fabsf(fy)), fabsf(fz)));
  const float gap = tf - tn;
  if (fabsf(gap) > m * 9.5367431640625e-07f) return gap > 0.0f;  // 2^-20
  return slab_fast(lo, hi, o, d, y);
}

// ---------------------------------------------------------------------------
// Wave-cooperative closest hit.
//
// Per-lane traversal leaves most lanes idle during primitive tests: at any
// moment only the few lanes standing on a leaf need the (long) triangle or
// sphere code, yet the whole wave steps through it.  Box tests never depend on
// primitive results (the reference culls nothing by distance), so the two are
// decoupled here:
//   * each lane walks the tree doing box tests only; when it reaches a leaf whose
//     box it hits, the pair (lane, node) is appended to a per-wave ring in LDS
//     (one ring per primitive kind, positions from a 64-wide ballot);
//   * as soon as a ring holds 64 pairs the whole wave tests them, one pair per
//     lane, reading the ray of the pair's lane from LDS -- all 64 lanes busy with
//     the same primitive kind; leftovers are drained at the end.
// A test that hits publishes key = (orderable(t) << 32) | ~node with a 64-bit
// LDS atomic min on its ray's slot.  The reference folds leaves left to right and
// replaces the best hit unless it is strictly nearer (src/BVH.cpp:75-76); for
// ordered distances that is "smallest t, and among equal t the LAST leaf in
// depth-first order", which is exactly the minimum of these keys, in any order of
// evaluation.  The pair that owns the minimum afterwards writes the full hit
// record.  Rays that met a NaN distance take the literal merge-tree path.
struct WaveShared {
  union {
    struct {
      float4 ray0[64], ray1[64], ray2[64];  // {o, d.x} {d.y, d.z, dn.x, dn.y} {dn.z,-,-,-}
      unsigned long long key[64];
      float4 hit[64];  // {t, b1, b2, bits(leaf)}
      uint32_t ring[2][128];
    } w;
    MergeStack merge;  // only used after everything above is dead
  };
  unsigned long long nan_mask;
};

RTR_DEV unsigned long long hit_key(float t, int node) {
  uint32_t u = __float_as_uint(t + 0.0f);  // -0 -> +0: the reference compares them equal
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | (uint32_t)(~node);
}

// One round: up to 64 pending (lane, node) pairs of one primitive kind.
template <int kKind, class Geom>
RTR_DEV void test_round(const Geom &g, WaveShared &ws, unsigned head, unsigned count) {
  const unsigned lane = threadIdx.x & 63u;
  if (lane < count) {
    uint32_t e = ws.w.ring[kKind][(head + lane) & 127u];
    unsigned r = e & 63u;
    int node = (int)(e >> 6);
    float4 a = ws.w.ray0[r], b = ws.w.ray1[r], c = ws.w.ray2[r];
    V3 o = mk(a.x, a.y, a.z), d = mk(a.w, b.x, b.y), dn = mk(b.z, b.w, c.x);
    int leaf = (int)wbits(g.node_hi(node));
    int row = leaf & 0x3fffffff;
    float t = 0, b1 = 0, b2 = 0;
    bool hit = kKind ? hit_sphere(g, row, o, dn, t) : hit_triangle(g, row, o, d, dn, t, b1, b2);
    if (hit) {
      if (t != t) {
        atomicOr(&ws.nan_mask, 1ull << r);
      } else {
        unsigned long long key = hit_key(t, node);
        atomicMin(&ws.w.key[r], key);
        // LDS executes a wave's operations in order: every lane's min is in place here
        if (ws.w.key[r] == key) ws.w.hit[r] = make_float4(t, b1, b2, __int_as_float(leaf));
      }
    }
  }
}

// Scene::intersect for the 64 rays of a wave.  Must be called by all 64 lanes
// (valid = false for lanes without a ray).
//
// The number of nodes a ray visits varies by orders of magnitude (nothing is culled by
// distance), so with can_split lanes whose ray is finished -- or that never had one --
// take the upper half of the node range of lanes that still walk.  Any node may start a
// walk when boxes are nested (can_split says they are) and the ray's arithmetic is
// finite: a box that passes implies that all its ancestors pass (their near/far
// quotients enclose the child's, rounding being monotone), so nodes below a missed box
// miss as well and no test is made that the reference would not make; results merge
// through the order-independent key minimum of the ray's slot.
#ifndef RTR_WAVE_SPLIT_AT
#define RTR_WAVE_SPLIT_AT 32
#endif
#ifndef RTR_WAVE_SPLIT_MIN
#define RTR_WAVE_SPLIT_MIN 8
#endif
template <class Geom>
RTR_DEV HitRec trace_wave(const Geom &g, int n_nodes, bool fast_planes, bool can_split, bool valid,
                          V3 o, V3 dir, V3 dn, WaveShared &ws) {
  const unsigned lane = threadIdx.x & 63u;
  const unsigned long long below = (1ull << lane) - 1ull;
  ws.w.ray0[lane] = make_float4(o.x, o.y, o.z, dir.x);
  ws.w.ray1[lane] = make_float4(dir.y, dir.z, dn.x, dn.y);
  ws.w.ray2[lane] = make_float4(dn.z, 0.0f, 0.0f, 0.0f);  // .yzw: hand-over of split ranges
  ws.w.key[lane] = ~0ull;
  if (lane == 0) ws.nan_mask = 0;
  // the walk's ray: the lane's own (slot r == lane) or, after a split, another slot's
  bool fast = fast_planes && fast_ray(o, dn);
  V3 wo = o, wdn = dn;
  V3 inv = mk(1.0f / dn.x, 1.0f / dn.y, 1.0f / dn.z);
  unsigned r = lane;
  unsigned head[2] = {0, 0}, tail[2] = {0, 0};  // wave-uniform ring cursors
  int i = 0, end = valid ? n_nodes : 0;
  for (;;) {
    const bool alive = i < end;
    const unsigned long long am = __ballot(alive);
    if (am == 0) break;
    if (can_split && (unsigned)__popcll(am) <= RTR_WAVE_SPLIT_AT) {
      const bool victim = alive && fast && end - i >= RTR_WAVE_SPLIT_MIN;
      const unsigned long long vm = __ballot(victim), tm = ~am;
      const unsigned nv = (unsigned)__popcll(vm), nt = (unsigned)__popcll(tm);
      const unsigned pairs = nv < nt ? nv : nt;
      if (pairs) {
        const unsigned vr = (unsigned)__popcll(vm & below), tr = (unsigned)__popcll(tm & below);
        if (victim && vr < pairs) {
          const int mid = i + (end - i) / 2;
          float4 *x = &ws.w.ray2[vr];
          x->y = __int_as_float(mid);
          x->z = __int_as_float(end);
          x->w = __int_as_float((int)r);
          end = mid;
        }
        if (!alive && tr < pairs) {  // LDS operations of a wave execute in order
          const float4 x = ws.w.ray2[tr];
          i = __float_as_int(x.y);
          end = __float_as_int(x.z);
          r = (unsigned)__float_as_int(x.w);
          const float4 a0 = ws.w.ray0[r], a1 = ws.w.ray1[r];
          wo = mk(a0.x, a0.y, a0.z);
          wdn = mk(a1.z, a1.w, ws.w.ray2[r].x);
          inv = mk(1.0f / wdn.x, 1.0f / wdn.y, 1.0f / wdn.z);
          fast = true;
        }
        continue;
      }
    }
    bool cand = false;
    int kind = 0, node = i;
    if (alive) {
      float4 lo = g.node_lo(i), hi = g.node_hi(i);
      int skip = (int)wbits(lo);
      int leaf = (int)wbits(hi);
      bool pass;
      if (fast) pass = slab_filtered(lo, hi, wo, wdn, inv);
      else pass = slab(lo, hi, wo, wdn);
      if (pass && leaf < 0) {
        i = i + 1;
      } else {
        i = skip;  // missed box, or a leaf (whose subtree is itself)
        cand = pass;
        kind = leaf >> 30;
      }
    }
    unsigned long long m0 = __ballot(cand && kind == 0), m1 = __ballot(cand && kind != 0);
    if (m0) {
      if (cand && kind == 0)
        ws.w.ring[0][(tail[0] + __popcll(m0 & below)) & 127u] = r | ((uint32_t)node << 6);
      tail[0] += (unsigned)__popcll(m0);
      if (tail[0] - head[0] >= 64u) {
        test_round<0>(g, ws, head[0], 64u);
        head[0] += 64u;
      }
    }
    if (m1) {
      if (cand && kind != 0)
        ws.w.ring[1][(tail[1] + __popcll(m1 & below)) & 127u] = r | ((uint32_t)node << 6);
      tail[1] += (unsigned)__popcll(m1);
      if (tail[1] - head[1] >= 64u) {
        test_round<1>(g, ws, head[1], 64u);
        head[1] += 64u;
      }
    }
  }
  if (tail[0] != head[0]) test_round<0>(g, ws, head[0], tail[0] - head[0]);
  if (tail[1] != head[1]) test_round<1>(g, ws, head[1], tail[1] - head[1]);

  HitRec best;
  best.leaf = -1;
  best.t = 0; best.b1 = 0; best.b2 = 0;
  if (ws.w.key[lane] != ~0ull) {
    float4 h = ws.w.hit[lane];
    best.t = h.x; best.b1 = h.y; best.b2 = h.z; best.leaf = __float_as_int(h.w);
  }
  // rare: a NaN distance was seen on some rays -> literal merge tree, one lane at a time
  // (the merge stack overlays the rings, which are dead by now)
  unsigned long long need = ws.nan_mask;
  while (need) {
    int l = __ffsll((long long)need) - 1;
    if ((int)lane == l) best = trace_tree(g, o, dir, dn, ws.merge);
    need &= need - 1;
  }
  return best;
}

// ---------------------------------------------------------------------------
// Closest hit over a stream of rays, with lanes refilled as their rays finish.
//
// The reference's traversal culls nothing by distance, so the number of nodes a ray
// visits varies by orders of magnitude with the ray (a ray that skims a large mesh
// walks thousands of nodes, one that only meets the walls a few dozen).  With one
// fixed ray per lane a wave lasts as long as its longest ray and most lanes idle
// (31 % of lanes active on a 1 M-triangle height field).  Here a wave keeps pulling
// rays from the stream: whenever few enough of its lanes are still walking, the
// pending primitive tests are flushed, finished lanes write their result and take
// the next rays.
//
// Box tests go through slab_filtered.
//
// A refill is kept short, because the whole wave stands still for it: ring entries
// carry the leaf, so a flush needs one round trip to memory, not two; both rings are
// flushed in one pass; a test that becomes its ray's best hit writes the hit's
// position and normal while the primitive is in registers, so retiring a ray loads
// nothing; and the next 64 rays of the wave are reserved and loaded into registers
// long before they are handed out (lane to lane, ds_bpermute), so that neither the
// stream counter nor the ray arrays are waited for.
struct StreamWave {
  float4 ray0[64], ray1[64];  // {o, d.x} {d.y, d.z, dn.x, dn.y}
  float rayz[64];             // dn.z
  unsigned long long key[64];
  float4 hit0[64], hit1[64];  // {position, t} {normal as stored, bits(leaf)}
  uint2 ring[2][128];         // {ray slot | node << 6, leaf}
  unsigned int parts[64];     // lanes currently walking a part of the slot's node range
  unsigned long long nan_mask;
};
// Per workgroup.  The stack of the literal merge-tree evaluation is shared by the
// waves of a workgroup (rays of other lanes stay live while one lane re-traces, so it
// cannot overlay the rings as in WaveShared); a wave takes `merge_lock` to use it.
struct StreamShared {
  StreamWave wave[4];
  MergeStack merge;
  unsigned merge_lock;
};

// Trangle::intersect / Sphere::intersect on primitives already in registers.
RTR_DEV bool hit_triangle_regs(float4 q0, float4 q1, float4 q2, V3 o, V3 dir, V3 dn, float &t,
                               float &b1, float &b2) {
  V3 p0 = xyz(q0), p1 = xyz(q1), p2 = xyz(q2), n = mk(q0.w, q1.w, q2.w);
  if (dot(-dir, n) < 0) return false;
  V3 e1 = p1 - p0, e2 = p2 - p0, s = o - p0;
  V3 s1 = cross(dn, e2), s2 = cross(s, e1);
  float se = dot(s1, e1);
  float inv = 1 / se;
  t = inv * dot(s2, e2);
  b1 = inv * dot(s1, s);
  b2 = inv * dot(s2, dn);
  // float operands against the double literal -1e-5 (src/Trangle.cpp:17), as float compares
  if (t < kTriEps || b1 < kTriEps || b2 < kTriEps || (1 - b1 - b2) < kTriEps) return false;
  return true;
}
RTR_DEV bool hit_sphere_regs(float4 q, V3 org, V3 d, float &t) {
  V3 c = xyz(q);
  float r = q.w;
  V3 o = org - c;
  if (length(o) < r) return false;
  float a = d.x * d.x + d.y * d.y + d.z * d.z;
  float b = 2 * (o.x * d.x + o.y * d.y + o.z * d.z);
  float cc = o.x * o.x + o.y * o.y + o.z * o.z - r * r;
  float delta = b * b - 4 * a * cc;
  if (delta < 0) return false;
  float sq = sqrtf(delta);
  float t1 = (-b + sq) / 2 / a;
  float t2 = (-b - sq) / 2 / a;
  if (t1 < kSphEps) return false;
  t = (t2 < 0) ? t1 : t2;
  return true;
}

// One pass over up to 64 pending (lane, leaf) pairs: c0 from the triangle ring, then c1
// from the sphere ring (c0 + c1 <= 64).
RTR_DEV void stream_round(const DeviceScene &sc, StreamWave &ws, unsigned head0, unsigned c0,
                          unsigned head1, unsigned c1) {
  const unsigned lane = threadIdx.x & 63u;
  if (lane >= c0 + c1) return;
  const bool is_tri = lane < c0;
  const uint2 e = is_tri ? ws.ring[0][(head0 + lane) & 127u] : ws.ring[1][(head1 + lane - c0) & 127u];
  const unsigned r = e.x & 63u;
  const int node = (int)(e.x >> 6);
  const int leaf = (int)e.y;
  const int row = leaf & 0x3fffffff;
  // all loads first: one round trip for both kinds
  const float4 *first = is_tri ? sc.tris + 3 * (size_t)row : sc.sphs + row;
  const float4 q0 = first[0];
  float4 q1 = q0, q2 = q0;
  if (is_tri) {
    q1 = first[1];
    q2 = first[2];
  }
  const float4 a = ws.ray0[r], b = ws.ray1[r];
  const V3 o = mk(a.x, a.y, a.z), d = mk(a.w, b.x, b.y), dn = mk(b.z, b.w, ws.rayz[r]);
  float t = 0, b1 = 0, b2 = 0;
  const bool hit = is_tri ? hit_triangle_regs(q0, q1, q2, o, d, dn, t, b1, b2) : hit_sphere_regs(q0, o, dn, t);
  if (!hit) return;
  if (t != t) {
    atomicOr(&ws.nan_mask, 1ull << r);
    return;
  }
  const unsigned long long key = hit_key(t, node);
  atomicMin(&ws.key[r], key);
  // LDS executes a wave's operations in order: every lane's min is in place here
  if (ws.key[r] != key) return;
  V3 pos, nrm;
  if (is_tri) {  // as hit_point()
    pos = xyz(q0) * (1 - b1 - b2) + xyz(q1) * b1 + xyz(q2) * b2;
    nrm = mk(q0.w, q1.w, q2.w);
  } else {
    pos = o + dn * t;
    nrm = unit(pos - xyz(q0));
  }
  ws.hit0[r] = make_float4(pos.x, pos.y, pos.z, t);
  ws.hit1[r] = make_float4(nrm.x, nrm.y, nrm.z, __int_as_float(leaf));
}

#ifndef RTR_REFILL_AT
#define RTR_REFILL_AT 40
#endif
// once the stream is dry: split node ranges when at most this many lanes still walk,
// and only ranges of at least RTR_SPLIT_MIN nodes
#ifndef RTR_SPLIT_AT
#define RTR_SPLIT_AT 32
#endif
#ifndef RTR_SPLIT_MIN
#define RTR_SPLIT_MIN 128
#endif

RTR_DEV void hit_point(const DeviceScene &sc, const HitRec &h, V3 o, V3 dn, V3 &pos, V3 &nrm);

#ifdef RTR_STREAM_STATS
// development only: [0] wave iterations, [1] lane visits, [2] refills, [3] rays,
// [4] test rounds, [5] pairs tested
__device__ unsigned long long g_stream_stats[8];
// (v is evaluated by every lane, so that a ballot in it sees the whole wave)
#ifdef RTR_SECTION_TIMES
#define RTR_STAT(k, v) do { } while (0)
#define RTR_CAUSE(k, v) do { } while (0)
#elif defined(RTR_CAUSE_STATS)
#define RTR_STAT(k, v) do { } while (0)
#define RTR_CAUSE(k, v) do { const unsigned long long v_ = (unsigned long long)(v); if ((threadIdx.x & 63u) == 0) atomicAdd(&g_stream_stats[k], v_); } while (0)
#else
#define RTR_STAT(k, v) do { const unsigned long long v_ = (unsigned long long)(v); if ((threadIdx.x & 63u) == 0) atomicAdd(&g_stream_stats[k], v_); } while (0)
#endif
#else
#define RTR_STAT(k, v) do { } while (0)
#endif
// Development only (-DRTR_STREAM_STATS -DRTR_SECTION_TIMES): shader cycles a wave spends in each
// section of the fused kernel's round, summed over waves into g_stream_stats[0..5]; [6] rounds.
struct SectionClock {
#ifdef RTR_SECTION_TIMES
  unsigned long long last, acc[7];
  RTR_DEV void start() { last = __builtin_readcyclecounter(); for (int i = 0; i < 7; i++) acc[i] = 0; }
  RTR_DEV void mark(int i) { const unsigned long long t = __builtin_readcyclecounter(); acc[i] += t - last; last = t; }
  RTR_DEV void count(int i) { acc[i] += 1; }
  RTR_DEV void flush() { if ((threadIdx.x & 63u) == 0) for (int i = 0; i < 7; i++) atomicAdd(&g_stream_stats[i], acc[i]); }
#else
  RTR_DEV void start() {}
  RTR_DEV void mark(int) {}
  RTR_DEV void count(int) {}
  RTR_DEV void flush() {}
#endif
};

RTR_DEV float lane_read(float v, unsigned src_lane) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), __float_as_int(v)));
}
RTR_DEV uint32_t lane_read(uint32_t v, unsigned src_lane) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}
RTR_DEV unsigned long long lane_read(unsigned long long v, unsigned src_lane) {
  return ((unsigned long long)lane_read((uint32_t)(v >> 32), src_lane) << 32) | lane_read((uint32_t)v, src_lane);
}

// Src provides, for a wave:
//   unsigned grab(Index &base, bool &more)  -- all lanes call it; reserves up to 64 rays
//       base .. base + n - 1 of the stream and returns n (wave-uniform); clears `more`
//       when nothing is left beyond them (only then may n be 0);
//   void load(Index idx, V3 &o, V3 &dir, Tag &tag) -- ray idx and whatever names its
//       result slot (Tag is a 4- or 8-byte integer);
//   void report(boxes, prims) -- once per wave at the end: box tests and primitive tests it made;
//   void store(Tag tag, int leaf, float t, V3 pos, V3 nrm) -- the result: leaf is -1
//       for a miss, else (kind << 30) | row with the hit's distance, position and normal
//       (Intersection::postion / ::normal of src/Trangle.cpp:23-24, src/Sphere.cpp:25-26).
// Must be called by all 64 lanes of the wave; returns when the stream is exhausted
// and all of this wave's rays are stored.
template <class Src>
RTR_DEV void trace_stream(const DeviceScene &sc, Src &src, StreamShared &sh) {
  typedef typename Src::Index Index;
  typedef typename Src::Tag Tag;
  const GlobalGeom g = global_geom(sc);
  const int n_nodes = sc.n_nodes;
  const bool fast_planes = sc.fast_planes != 0;
  const bool can_split = sc.nested_boxes != 0;
  StreamWave &ws = sh.wave[threadIdx.x >> 6];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned long long below = (1ull << lane) - 1ull;
  if (lane == 0) ws.nan_mask = 0;
  ws.parts[lane] = 0;
  // `have`: this lane's slot holds a ray whose result is not stored yet (tag names it).
  // The lane walks nodes [i, end) for the ray in slot r: its own (r == lane), or, once
  // the stream has run dry, half of another lane's remaining range.
  bool have = false, fast = false;
  Tag tag = 0;
  unsigned r = lane;
  V3 o = mk(0, 0, 0), dn = mk(1, 1, 1), inv = mk(1, 1, 1);
  unsigned head0 = 0, tail0 = 0, head1 = 0, tail1 = 0;  // wave-uniform ring cursors
  // rays reserved ahead: lane l holds ray stage_base + l, for l < stage_n; the first
  // stage_used of them have been handed out
  bool more = true;  // the stream may hold rays beyond the reserved ones
  Index stage_base = 0;
  unsigned stage_n = 0, stage_used = 0;
  V3 so = mk(0, 0, 0), sd = mk(1, 1, 1);
  Tag stag = 0;
  int i = 0, end = 0;
  // the wave's work, counted in scalar registers: boxes tested, primitives tested
  unsigned long long n_boxes = 0, n_prims = 0;
  for (;;) {
    const bool alive = i < end;
    const unsigned long long alive_mask = __ballot(alive);
    const unsigned n_alive = (unsigned)__popcll(alive_mask);
    const bool supply = more || stage_used < stage_n;
    bool event = n_alive == 0 || (supply && n_alive <= RTR_REFILL_AT);
    if (!event && !supply && can_split && n_alive <= RTR_SPLIT_AT)
      event = __ballot(alive && fast && end - i >= RTR_SPLIT_MIN) != 0;
    if (event) {
      RTR_STAT(2, 1);
      // flush the pending primitive tests, so that finished rays are final
      {
        const unsigned c0 = tail0 - head0, c1 = tail1 - head1;
        n_prims += c0 + c1;
        RTR_STAT(4, (c0 + c1 > 64u) ? 2 : (c0 + c1 > 0u ? 1 : 0));
        RTR_STAT(5, c0 + c1);
        if (c0 + c1 > 64u) {
          stream_round(sc, ws, head0, c0, head1, 0u);
          stream_round(sc, ws, head0, 0u, head1, c1);
        } else if (c0 + c1 > 0u) {
          stream_round(sc, ws, head0, c0, head1, c1);
        }
        head0 = tail0;
        head1 = tail1;
      }
      // a lane that reached the end of its range leaves its ray; a ray no lane works on
      // any more is finished (LDS operations of a wave execute in order)
      if (!alive && end > 0) {
        atomicSub(&ws.parts[r], 1u);
        end = 0;
        i = 0;
      }
      const bool done = have && ws.parts[lane] == 0;
      int leaf = -1;
      float t = 0;
      V3 pos = mk(0, 0, 0), nrm = mk(0, 0, 0);
      if (done && ws.key[lane] != ~0ull) {
        const float4 h0 = ws.hit0[lane], h1 = ws.hit1[lane];
        pos = xyz(h0);
        t = h0.w;
        nrm = xyz(h1);
        leaf = __float_as_int(h1.w);
      }
      // rare: a NaN distance was seen -> literal merge tree, one lane at a time
      unsigned long long need = ws.nan_mask & __ballot(done);
      if (need) {
        if (lane == 0) {
          ws.nan_mask &= ~need;
          while (atomicCAS(&sh.merge_lock, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(2);
        }
        while (need) {
          int l = __ffsll((long long)need) - 1;
          if ((int)lane == l) {
            const float4 a0 = ws.ray0[lane], a1 = ws.ray1[lane];
            const V3 ro = mk(a0.x, a0.y, a0.z), rd = mk(a0.w, a1.x, a1.y), rn = mk(a1.z, a1.w, ws.rayz[lane]);
            HitRec h = trace_tree(g, ro, rd, rn, sh.merge);
            leaf = h.leaf;
            t = h.t;
            pos = mk(0, 0, 0);
            nrm = mk(0, 0, 0);
            if (h.leaf >= 0) hit_point(sc, h, ro, rn, pos, nrm);
          }
          need &= need - 1;
        }
        if (lane == 0) atomicExch(&sh.merge_lock, 0u);
      }
      if (done) {
        src.store(tag, leaf, t, pos, nrm);
        have = false;
      }
      // hand out reserved rays; when they run out reserve the next 64 (their loads are
      // only waited for if lanes are still idle now, else at a later refill)
      for (int pass = 0; pass < 2 && supply; pass++) {
        const unsigned long long idle = __ballot(!have);
        const unsigned want = (unsigned)__popcll(idle);
        const unsigned avail = stage_n - stage_used;
        const unsigned take = want < avail ? want : avail;
        if (take) {
          const unsigned rank = (unsigned)__popcll(idle & below);
          const unsigned from = (stage_used + rank) & 63u;
          const V3 no = mk(lane_read(so.x, from), lane_read(so.y, from), lane_read(so.z, from));
          const V3 nd = mk(lane_read(sd.x, from), lane_read(sd.y, from), lane_read(sd.z, from));
          const Tag nt = lane_read(stag, from);
          if (!have && rank < take) {
            tag = nt;
            o = no;
            dn = unit(nd);
            inv = mk(1.0f / dn.x, 1.0f / dn.y, 1.0f / dn.z);
            fast = fast_planes && fast_ray(o, dn);
            ws.ray0[lane] = make_float4(o.x, o.y, o.z, nd.x);
            ws.ray1[lane] = make_float4(nd.y, nd.z, dn.x, dn.y);
            ws.rayz[lane] = dn.z;
            ws.key[lane] = ~0ull;
            ws.parts[lane] = n_nodes > 0 ? 1u : 0u;  // an empty tree: finished at once
            r = lane;
            i = 0;
            end = n_nodes;
            have = true;
          }
          stage_used += take;
          RTR_STAT(3, take);
        }
        if (stage_used < stage_n || !more) break;
        const unsigned got = src.grab(stage_base, more);
        stage_n = got;
        stage_used = 0;
        if (lane < got) src.load(stage_base + lane, so, sd, stag);
        if (want == take) break;
      }
      // The stream is dry: idle lanes take the upper half of the node range of lanes
      // that still walk.  Any node may start a walk when boxes are nested and the ray's
      // arithmetic is finite (fast): a box that passes implies that all its ancestors
      // pass (their near/far quotients enclose the child's, rounding being monotone), so
      // nodes below a missed box miss as well and no test is made that the reference
      // would not make; results merge through the order-independent key minimum.
      if (!(more || stage_used < stage_n) && can_split) {
        const bool victim = i < end && fast && end - i >= RTR_SPLIT_MIN;
        const bool thief = !(i < end);
        const unsigned long long vm = __ballot(victim), tm = __ballot(thief);
        const unsigned nv = (unsigned)__popcll(vm), nt = (unsigned)__popcll(tm);
        const unsigned pairs = nv < nt ? nv : nt;
        if (pairs) {
          RTR_STAT(6, pairs);
          // the rings are empty here: their storage carries the hand-over
          const unsigned vr = (unsigned)__popcll(vm & below), tr = (unsigned)__popcll(tm & below);
          if (victim && vr < pairs) {
            const int mid = i + (end - i) / 2;
            ws.ring[0][vr] = make_uint2((uint32_t)mid, (uint32_t)end);
            ws.ring[1][vr].x = r;
            atomicAdd(&ws.parts[r], 1u);
            end = mid;
          }
          if (thief && tr < pairs) {
            const uint2 range = ws.ring[0][tr];
            r = ws.ring[1][tr].x;
            i = (int)range.x;
            end = (int)range.y;
            const float4 a0 = ws.ray0[r], a1 = ws.ray1[r];
            o = mk(a0.x, a0.y, a0.z);
            dn = mk(a1.z, a1.w, ws.rayz[r]);
            inv = mk(1.0f / dn.x, 1.0f / dn.y, 1.0f / dn.z);
            fast = true;
          }
        }
      }
      if ((__ballot(have) | __ballot(i < end)) == 0) break;  // stream dry, nothing in flight
      continue;
    }
    n_boxes += n_alive;
    RTR_STAT(0, 1);
    RTR_STAT(1, n_alive);
    bool cand = false;
    int kind = 0, node = i, cleaf = 0;
    if (alive) {
      float4 lo = g.node_lo(i), hi = g.node_hi(i);
      int skip = (int)wbits(lo);
      int leaf = (int)wbits(hi);
      bool pass;
      if (fast) pass = slab_filtered(lo, hi, o, dn, inv);
      else pass = slab(lo, hi, o, dn);
      if (pass && leaf < 0) {
        i = i + 1;
      } else {
        i = skip;  // missed box, or a leaf (whose subtree is itself)
        cand = pass;
        kind = leaf >> 30;
        cleaf = leaf;
      }
    }
    const unsigned long long m0 = __ballot(cand && kind == 0), m1 = __ballot(cand && kind != 0);
    if (m0) {
      if (cand && kind == 0)
        ws.ring[0][(tail0 + __popcll(m0 & below)) & 127u] = make_uint2(r | ((uint32_t)node << 6), (uint32_t)cleaf);
      tail0 += (unsigned)__popcll(m0);
      if (tail0 - head0 >= 64u) {
        RTR_STAT(4, 1);
        RTR_STAT(5, 64);
        n_prims += 64u;
        stream_round(sc, ws, head0, 64u, head1, 0u);
        head0 += 64u;
      }
    }
    if (m1) {
      if (cand && kind != 0)
        ws.ring[1][(tail1 + __popcll(m1 & below)) & 127u] = make_uint2(r | ((uint32_t)node << 6), (uint32_t)cleaf);
      tail1 += (unsigned)__popcll(m1);
      if (tail1 - head1 >= 64u) {
        RTR_STAT(4, 1);
        RTR_STAT(5, 64);
        n_prims += 64u;
        stream_round(sc, ws, head0, 0u, head1, 64u);
        head1 += 64u;
      }
    }
  }
  src.report(n_boxes, n_prims);
}

// Geometry staged in LDS by a whole workgroup (kernels call this before tracing,
// every thread of the block, then __syncthreads()).
constexpr int kLdsNodes = 64, kLdsTris = 48, kLdsSphs = 32, kLdsLeaves = kLdsTris + kLdsSphs;
struct LdsScene {
  float4 nodes[2 * kLdsNodes];
  float4 tris[3 * kLdsTris];
  float4 sphs[kLdsSphs];
  int32_t parent[kLdsNodes];
  // by leaf number k (depth-first leaf order): {node, leaf payload}
  int2 leaf_of[kLdsLeaves];
  // by sphere number j (depth-first order among the sphere leaves): {c, r} and
  // {r*r, T, bits(k), -} of the leaf record (see trace_all)
  float4 sph_leaf[kLdsSphs], sph_aux[kLdsSphs];
  // by mirrored pair: the second triangle's p0, e1, e2 (see trace_all2)
  float4 pair_second[3 * (kLdsTris / 2)];
};
RTR_DEV void stage_scene(const DeviceScene &sc, LdsScene &s, int tid, int nthreads) {
  for (int k = tid; k < 2 * sc.n_nodes; k += nthreads) s.nodes[k] = sc.nodes[k];
  for (int k = tid; k < 3 * sc.n_tris; k += nthreads) s.tris[k] = sc.tris[k];
  for (int k = tid; k < sc.n_sphs; k += nthreads) s.sphs[k] = sc.sphs[k];
  for (int k = tid; k < sc.n_nodes; k += nthreads) s.parent[k] = sc.parent[k];
  for (int k = tid; k < 3 * sc.n_pairs; k += nthreads) s.pair_second[k] = sc.pair_second[k];
  for (int k = tid; k < sc.n_leaves; k += nthreads) {
    const float4 r0 = sc.leaves[4 * k], r1 = sc.leaves[4 * k + 1], r3 = sc.leaves[4 * k + 3];
    const int leaf = __float_as_int(r3.w);
    s.leaf_of[k] = make_int2(__float_as_int(r3.z), leaf);
    if (leaf >> 30) {
      const int j = __float_as_int(r1.z);
      s.sph_leaf[j] = r0;
      s.sph_aux[j] = make_float4(r1.x, r1.y, __int_as_float(k), 0.0f);
    }
  }
}

// ---------------------------------------------------------------------------
// Closest hit in a scene of a few dozen primitives: test them all.
//
// The reference reaches a leaf only through boxes the ray's line crosses, but it
// culls nothing by distance, and in a scene this small a line crosses most boxes:
// walking the tree costs more box tests than there are primitives, plus the
// bookkeeping of a walk.  Its result is  min over {leaves whose ancestors' boxes
// all pass}  under the left-to-right fold of src/BVH.cpp:75-76.  Here every lane
//   1. folds ALL leaves in depth-first order with that same rule -- the leaf index
//      is wave-uniform, so leaf data arrives through the scalar cache in SGPRs, no
//      lane is ever idle and there is no per-wave queue, atomic or LDS traffic;
//   2. runs the reference's box test on the winner's own chain of boxes (leaf to
//      root).  If they all pass, the winner belongs to the reference's candidate
//      set, and the minimum of a superset that lies in the subset is the subset's
//      minimum, ties included: an equal-distance later leaf would have replaced it
//      in step 1 as well.
// Rays whose winner fails a box (a hit through the 1e-5 slack of the triangle test
// just outside a box), and rays that met a NaN distance, are re-traced by the
// literal tree evaluation, one lane at a time.
//
// Leaf record, 4 x float4, in depth-first leaf order:
//   triangle: {p0, n.x} {e1 = p1-p0, n.y} {e2 = p2-p0, n.z} {-, -, bits(node), bits(leaf)}
//   sphere:   {c, r}    {r*r, T, bits(j), -}  {-}           {-, -, bits(node), bits(leaf)}
// (j numbers the sphere leaves in depth-first order.)
// e1, e2 and r*r are the reference's own fp32 intermediates (src/Trangle.cpp:8-9,
// src/Sphere.cpp:13), formed once on the host instead of once per ray.  T is the
// smallest float whose correctly rounded square root is >= r, so that
// `length(o) < r` (src/Sphere.cpp:6) is `dot(o,o) < T` without the square root.
struct LeafRecord {
  float a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3, d0, d1;
  int node, leaf;
};
typedef const LeafRecord __attribute__((address_space(4))) const_leaf;

RTR_DEV bool quotient_operand_fast(float n) {
  // [2^-40, 2^40]; zero is left to the IEEE division: the corrections lose the sign
  // of a zero quotient, and the root is returned as the hit distance
  float m = fabsf(n);
  return m >= 9.094947017729282e-13f && m <= 1099511627776.0f;
}

RTR_DEV LeafRecord load_leaf(const const_leaf *tab, int k) {
  LeafRecord r;
  r.a0 = tab[k].a0; r.a1 = tab[k].a1; r.a2 = tab[k].a2; r.a3 = tab[k].a3;
  r.b0 = tab[k].b0; r.b1 = tab[k].b1; r.b2 = tab[k].b2; r.b3 = tab[k].b3;
  r.c0 = tab[k].c0; r.c1 = tab[k].c1; r.c2 = tab[k].c2; r.c3 = tab[k].c3;
  r.d0 = tab[k].d0; r.d1 = tab[k].d1; r.node = tab[k].node; r.leaf = tab[k].leaf;
  return r;
}

// A leaf record on its way into sixteen scalar registers: leaf_fetch issues the load, leaf_ready
// waits for it (for every scalar and LDS access the wave has in flight, that is), leaf_unpack
// names the fields.  The compiler does not know that the registers are not valid in between:
// a record must not be unpacked before leaf_ready.
typedef int leaf_regs __attribute__((ext_vector_type(16)));
RTR_DEV leaf_regs leaf_fetch(const const_leaf *rec) {
  leaf_regs r;
  asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(r) : "s"(rec));
  return r;
}
RTR_DEV void leaf_ready(leaf_regs &r) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r)); }
RTR_DEV LeafRecord leaf_unpack(const leaf_regs &v) {
  LeafRecord r;
  r.a0 = __int_as_float(v[0]); r.a1 = __int_as_float(v[1]); r.a2 = __int_as_float(v[2]); r.a3 = __int_as_float(v[3]);
  r.b0 = __int_as_float(v[4]); r.b1 = __int_as_float(v[5]); r.b2 = __int_as_float(v[6]); r.b3 = __int_as_float(v[7]);
  r.c0 = __int_as_float(v[8]); r.c1 = __int_as_float(v[9]); r.c2 = __int_as_float(v[10]); r.c3 = __int_as_float(v[11]);
  r.d0 = __int_as_float(v[12]); r.d1 = __int_as_float(v[13]); r.node = v[14]; r.leaf = v[15];
  return r;
}

RTR_DEV HitRec trace_all(const DeviceScene &sc, const LdsScene &lds, bool valid, V3 o, V3 dir,
                         V3 dn, MergeStack &merge) {
  const unsigned lane = threadIdx.x & 63u;
  const const_leaf *tab = (const const_leaf *)sc.leaves;  // uniform index -> scalar loads
  const int n_leaves = sc.n_leaves;
  // per-ray invariants of src/Sphere.cpp:10-14
  const float qa = dn.x * dn.x + dn.y * dn.y + dn.z * dn.z;
  const float qa4 = 4 * qa;
  // n / qa by one fma correction of n * RN(1/qa) (div_corrected) is the IEEE quotient
  // when qa is in [0.5, 2] and |n| is in [2^-40, 2^40]
  const bool a_fast = qa >= 0.5f && qa <= 2.0f;
  const float ya = 1.0f / qa;
  HitRec best;
  best.leaf = -1;
  best.t = 0; best.b1 = 0; best.b2 = 0;
  int best_node = -1;
  bool nan_seen = false;
  LeafRecord r = load_leaf(tab, 0);  // the table ends with one padding record
  for (int k = 0; k < n_leaves; k++) {
    const LeafRecord nx = load_leaf(tab, k + 1);  // in flight while this leaf is tested
    const int leaf = r.leaf;
    float t = 0, b1 = 0, b2 = 0;
    bool hit;
    if (leaf >> 30) {
      // Sphere::intersect, src/Sphere.cpp:3-39
      const V3 oc = o - mk(r.a0, r.a1, r.a2);
      const float l2 = oc.x * oc.x + oc.y * oc.y + oc.z * oc.z;
      const float b = 2 * (oc.x * dn.x + oc.y * dn.y + oc.z * dn.z);
      const float cc = l2 - r.b0;
      const float delta = b * b - qa4 * cc;
      // most waves have no ray that reaches this sphere: the square root and the
      // two quotients are only formed when some lane needs them
      const bool reach = valid && !(l2 < r.b1) && !(delta < 0);
      hit = false;
      if (__ballot(reach) != 0) {
        const float sq = sqrtf(delta);
        const float n1 = (-b + sq) / 2, n2 = (-b - sq) / 2;
        float t1, t2;
        const bool slow = reach && !(a_fast && quotient_operand_fast(n1) && quotient_operand_fast(n2));
        if (__ballot(slow) == 0) {
          t1 = div_corrected(n1, qa, ya);
          t2 = div_corrected(n2, qa, ya);
        } else {
          t1 = n1 / qa;
          t2 = n2 / qa;
        }
        hit = reach && !(t1 < kSphEps);
        t = (t2 < 0) ? t1 : t2;
      }
    } else {
      // Trangle::intersect, src/Trangle.cpp:3-28
      const V3 p0 = mk(r.a0, r.a1, r.a2), n = mk(r.a3, r.b3, r.c3);
      const V3 e1 = mk(r.b0, r.b1, r.b2), e2 = mk(r.c0, r.c1, r.c2);
      const V3 s = o - p0;
      const V3 s1 = cross(dn, e2), s2 = cross(s, e1);
      const float se = dot(s1, e1);
      const float inv = 1 / se;
      t = inv * dot(s2, e2);
      b1 = inv * dot(s1, s);
      b2 = inv * dot(s2, dn);
      hit = !(dot(-dir, n) < 0) &&
            !(t < kTriEps || b1 < kTriEps || b2 < kTriEps || (1 - b1 - b2) < kTriEps);
    }
    const bool is_nan = t != t;
    nan_seen = nan_seen || (hit && is_nan);
    // src/BVH.cpp:75-76, folded left to right: replaced unless strictly nearer
    const bool take = hit && !is_nan && !(best.leaf >= 0 && best.t < t);
    best.t = take ? t : best.t;
    best.b1 = take ? b1 : best.b1;
    best.b2 = take ? b2 : best.b2;
    best.leaf = take ? leaf : best.leaf;
    best_node = take ? r.node : best_node;
    r = nx;
  }
  // The winner's boxes, leaf to root: Bound3::intersect, src/Bound3.cpp:3-22.
  // When boxes are nested (a parent's planes enclose its children's, checked at
  // scene creation) and all quotients are finite, the leaf's own box decides: with
  // lo_P <= lo_C and hi_P >= hi_C, every step of the test is monotonic (rounded
  // subtraction, rounded division by the same d, min, max), so per axis
  // near_P <= near_C and far_P >= far_C, and  max near_C <= min far_C  implies the
  // same for the parent, all the way to the root.
  const bool fast = sc.fast_planes != 0 && fast_ray(o, dn);
  const bool leaf_only = fast && sc.nested_boxes != 0;
  const V3 inv = mk(1.0f / dn.x, 1.0f / dn.y, 1.0f / dn.z);
  bool ok = true;
  for (int node = best_node; node >= 0; node = leaf_only ? -1 : lds.parent[node]) {
    const float4 lo = lds.nodes[2 * node], hi = lds.nodes[2 * node + 1];
    bool pass;
    if (fast) pass = slab_fast(lo, hi, o, dn, inv);
    else pass = slab(lo, hi, o, dn);
    ok = ok && pass;
  }
  unsigned long long need = __ballot(valid && (nan_seen || !ok));
  if (need) {
    LdsGeom g{lds.nodes, lds.tris, lds.sphs};
    while (need) {
      int l = __ffsll((long long)need) - 1;
      if ((int)lane == l) best = trace_tree(g, o, dir, dn, merge);
      need &= need - 1;
    }
  }
  return best;
}

// Intersection::postion / ::normal of a recorded hit (src/Trangle.cpp:23-24,
// src/Sphere.cpp:25-26 / :33-34).
RTR_DEV void hit_point(const DeviceScene &sc, const HitRec &h, V3 o, V3 dn, V3 &pos, V3 &nrm) {
  int row = h.leaf & 0x3fffffff;
  if (h.leaf >> 30) {
    float4 q = sc.sphs[row];
    pos = o + dn * h.t;
    nrm = unit(pos - xyz(q));
  } else {
    float4 q0 = sc.tris[3 * row], q1 = sc.tris[3 * row + 1], q2 = sc.tris[3 * row + 2];
    pos = xyz(q0) * (1 - h.b1 - h.b2) + xyz(q1) * h.b1 + xyz(q2) * h.b2;
    nrm = mk(q0.w, q1.w, q2.w);
  }
}

RTR_DEV int hit_prim(const DeviceScene &sc, const HitRec &h) {
  if (h.leaf < 0) return -1;
  int row = h.leaf & 0x3fffffff;
  return (h.leaf >> 30) ? sc.sph_prim[row] : sc.tri_prim[row];
}

// ---------------------------------------------------------------------------
// trace_all for two rays that leave the same point (the shadow ray and the
// extension ray of one path vertex).
//
// Everything that depends on the origin only is formed once (s, s x e1 and the
// numerator of t for triangles; o - c, its square and c for spheres), and the
// per-ray arithmetic runs in packed fp32 instructions with ray A in the low half
// and ray B in the high half: v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 round each
// half exactly like the scalar forms, and issue at the same rate.  The sequence of
// operations per ray is that of trace_all.
struct V3x2 {
  f32x2 x, y, z;
};
RTR_DEV f32x2 splat(float v) { return f32x2{v, v}; }
RTR_DEV V3x2 pair(V3 a, V3 b) { return V3x2{{a.x, b.x}, {a.y, b.y}, {a.z, b.z}}; }
RTR_DEV V3x2 splat(V3 a) { return V3x2{splat(a.x), splat(a.y), splat(a.z)}; }
RTR_DEV f32x2 dot(V3x2 a, V3x2 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RTR_DEV V3x2 cross(V3x2 a, V3x2 b) {
  return V3x2{a.y * b.z - a.z * b.y, a.z * b.x - b.z * a.x, a.x * b.y - a.y * b.x};
}
RTR_DEV f32x2 div_corrected_pairs(f32x2 a, f32x2 d, f32x2 y) {
  const f32x2 nd = -d;
  const f32x2 q = a * y;
  const f32x2 r = __builtin_elementwise_fma(nd, q, a);
  return __builtin_elementwise_fma(r, y, q);
}

// node index of a leaf payload in a staged scene (only used after the rare literal
// tree evaluation, which reports the leaf but not where it sits)
RTR_DEV int node_of_leaf(const LdsScene &lds, int n_nodes, int leaf) {
  if (leaf < 0) return -1;
  for (int i = 0; i < n_nodes; i++)
    if (__float_as_int(lds.nodes[2 * i + 1].w) == leaf) return i;
  return -1;
}

// Best hit of one ray while the leaves are folded.  k is the leaf's number in depth-first
// leaf order (-1: nothing hit yet, with t = +inf).
struct BestHit {
  float t, b1, b2;
  int k;
  RTR_DEV void init() {
    t = __int_as_float(0x7f800000);
    b1 = 0; b2 = 0;
    k = -1;
  }
  // src/BVH.cpp:75-76 folded left to right -- the best hit is replaced unless it is strictly
  // nearer -- for leaves that arrive in increasing k.  tt is the distance of a hit and NaN
  // otherwise: "tt <= t" is false for a NaN (best is +inf while nothing is hit, and never NaN).
  RTR_DEV void fold_next(float tt, float bb1, float bb2, int kk) {
    const bool take = tt <= t;
    t = take ? tt : t;
    b1 = take ? bb1 : b1;
    b2 = take ? bb2 : b2;
    k = take ? kk : k;
  }
};

// The hardware's minimum / maximum of three (a NaN operand is passed over, like fminf does).
// Written out because the compiler's fminf first re-canonicalises operands that come out of
// packed arithmetic (one more instruction each at the rate of a compare).
RTR_DEV float min3_raw(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
RTR_DEV float min_raw(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
RTR_DEV float min3_abs(float a, float b, float c) {  // min(a, |b|, |c|)
  float r;
  asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
RTR_DEV float min3_abs3(float a, float b, float c) {  // min(|a|, |b|, |c|)
  float r;
  asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
RTR_DEV float max3_abs3(float a, float b, float c) {  // max(|a|, |b|, |c|)
  float r;
  asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
RTR_DEV float max3_abs(float a, float b, float c) {  // max(a, |b|, |c|)
  float r;
  asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// keeps the compiler from folding a select into the comparisons around it (which would
// bring the lane masks back into scalar registers and scalar instructions)
RTR_DEV float opaque(float v) {
  asm volatile("" : "+v"(v));
  return v;
}
RTR_DEV uint32_t opaque(uint32_t v) {
  asm volatile("" : "+v"(v));
  return v;
}

// colour / d1 / d2 with two positive scalars (src/Render.cpp:104: Le f cos cos' / dis^2 / pdf;
// :125: f cos / pdf / P_RR): six IEEE divisions at the price of seventeen plain instructions
// each.  The short way -- RN(1 / d) by reciprocal_fast, each quotient by div_corrected -- is
// taken when, for every lane of the wave, both divisors are in [2^-20, 2^20] and every
// component is zero or of magnitude in [2^-60, 2^60]: then the first quotients are zero or in
// [2^-80, 2^80], the second ones in [2^-100, 2^100], and no product or residual leaves the normal
// range.  Zero components are common (a light below the horizon of the surface) and of either
// sign; div_corrected turns -0 into +0, so the numerator's sign is put back (the divisors
// are positive: the quotient's sign is the numerator's).  A NaN component passes the guard unseen
// by the maximum and comes out as a NaN either way.
RTR_DEV V3 div_div(V3 x, float d1, float d2) {
#ifndef RTR_EXP_IEEE_COLOUR_DIV
  const uint32_t bx = __float_as_uint(x.x), by = __float_as_uint(x.y), bz = __float_as_uint(x.z);
  // twice the magnitude's bit pattern less one wraps a zero to the top (see unit())
  const uint32_t least = min3_u32(bx + bx - 1u, by + by - 1u, bz + bz - 1u);
  const float most = max3_abs3(x.x, x.y, x.z);
  const uint32_t lo = 0x35800000u, span = 0x49800000u - 0x35800000u;  // 2^-20 ... 2^20; negative numbers and NaNs lie above
  const bool bad = least < 2u * 0x21800000u - 1u || !(most <= 1152921504606846976.0f) ||
                   __float_as_uint(d1) - lo > span || __float_as_uint(d2) - lo > span;
  if (__ballot(bad) == 0) {
    const float y1 = reciprocal_fast(d1), y2 = reciprocal_fast(d2);
    const float qx = div_corrected(div_corrected(x.x, d1, y1), d2, y2);
    const float qy = div_corrected(div_corrected(x.y, d1, y1), d2, y2);
    const float qz = div_corrected(div_corrected(x.z, d1, y1), d2, y2);
    return mk(__builtin_copysignf(qx, x.x), __builtin_copysignf(qy, x.y), __builtin_copysignf(qz, x.z));
  }
#endif
  return mk(x.x / d1 / d2, x.y / d1 / d2, x.z / d1 / d2);
}

// trace_all for two rays that leave the same point, in two phases.
//
// 1. One pass over the leaf table (wave-uniform leaf, data in SGPRs, packed arithmetic as
//    described above).  Triangles are tested and folded.  Of a sphere only the cheap part
//    runs: does the ray's line meet it at all (origin outside, discriminant >= 0,
//    src/Sphere.cpp:5-15)?  The answer goes into one bit per sphere and ray.  Few rays
//    reach a given sphere, but among the 128 rays of a wave nearly always one does, so
//    running the square root, the two quotients and the fold for the whole wave (as
//    trace_all does) was mostly idle work.
// 2. Every lane takes its next marked sphere -- lanes work on different spheres and on
//    either of their two rays -- reads its record from LDS and finishes the test; the
//    loop ends when no lane has a mark left (three or four rounds on the default scene,
//    instead of one per sphere).  Leaves now arrive out of order, so the fold spells the
//    tie rule out: the reference keeps, among equal distances, the LAST leaf in depth-first
//    order ("replaced unless strictly nearer"), i.e. the larger k.
//
// Scalar instructions are the scarce resource of this loop (a scalar instruction holds its
// wave up for longer than a vector instruction does, and four waves per SIMD do not hide
// it: tools/ubench/issue_cycles.hip), so everything that decides per lane is kept in vector
// registers: a hit is folded as "distance, or NaN for no hit", conditions that only switch a
// test off become operands (+inf as the threshold of a triangle that faces away), and
// whatever can only go wrong for a few rays in a billion (a reciprocal outside the short
// sequence's range, a NaN or infinite distance) is accumulated and looked at once after the
// pass: such a ray is re-traced by the literal tree evaluation.
struct HitPair {
  HitRec a, b;
  int node_a, node_b;  // the leaves' node indices (-1 for a miss)
};

//
// kBounded: the scene's planes are in the division-free box test's range, its boxes nested and its
// primitives within 2^20 of the origin (DeviceScene::fast_planes, nested_boxes, near_prims).  A ray
// is then either re-traced for its own sake -- an origin component beyond 2^20 or a direction
// component outside [2^-20, 2] -- or cannot produce an infinite or undefined distance at all:
// s and the edges are below 2^21, s1 below 2^23, se below 2^46, s2 below 2^43, the numerator of t
// below 2^66 and t, with |1 / se| <= 2^60 (se_lo), below 2^126; b1 and b2 below 2^106; a sphere's b
// below 2^25, its discriminant below 2^51 and the numerators of its roots below 2^26.  The sum of
// distances, the largest |se|, discriminant and numerator are not accumulated for such a scene, and the winners' boxes take the one-box form unconditionally.
template <bool kBounded>
RTR_DEV HitPair trace_all2(const DeviceScene &sc, const float4 *item_table, int n_items, bool has_pairs,
                           const LdsScene &lds, bool valid_a, bool valid_b,
                           V3 o, V3 dir_a, V3 dn_a, V3 dir_b, V3 dn_b, MergeStack &merge, SectionClock &clk) {
  const unsigned lane = threadIdx.x & 63u;
  const V3x2 dir = pair(dir_a, dir_b), dn = pair(dn_a, dn_b);
  const V3x2 ndir = V3x2{-dir.x, -dir.y, -dir.z};
  const f32x2 qa = dn.x * dn.x + dn.y * dn.y + dn.z * dn.z;
  const f32x2 qa4 = splat(4.0f) * qa;
  BestHit ba, bb;
  ba.init();
  bb.init();
  uint32_t reach_a = 0, reach_b = 0;  // bit j: the ray's line meets sphere j, not behind the origin
  const bool a_half = qa.x >= 0.5f && qa.y >= 0.5f;
  // n / qa by one fma correction of n * RN(1/qa) (div_corrected) is the IEEE quotient
  // when qa is in [0.5, 2] and |n| is in [2^-40, 2^40]
  const bool a_fast = a_half && qa.x <= 2.0f && qa.y <= 2.0f;
  // RN(1 / qa): qa in [0.5, 2] is well inside reciprocal_fast's range
  f32x2 ya;
  if (__ballot(!a_fast) == 0) ya = f32x2{reciprocal_fast(qa.x), reciprocal_fast(qa.y)};
  else ya = f32x2{1.0f / qa.x, 1.0f / qa.y};
  asm volatile("" : "+v"(ya));  // formed once here, not again in every round of phase 2
  // Mirrored pairs.  Two triangles that split a parallelogram (p0 p1 p2 and p3 p2 p1, like every
  // wall of the reference's scene, src/main.cpp:26-39) have e1' = -e1 and e2' = -e2.  When two
  // such leaves are neighbours, the negations are exact (zero components +0 in both, the
  // others of magnitude in [2^-60, 2^60]) and the normals agree, the scene upload leaves the
  // second one out of the record table, marks the first and keeps the second's own p0, e1, e2
  // by pair.  Of the second triangle's test,
  //   s1' = cross(dn, e2'): every product changes sign, so a non-zero component is the first
  //        triangle's negated -- provided no product underflows: direction components of
  //        magnitude in [2^-20, 2] -- and a zero component (both products zero, or equal) is a
  //        zero, whose sign nothing downstream can see (below, where the second test is made);
  //   se' = dot(s1', e1'): the same non-zero products in the same order (zero products may
  //        differ in sign, which a non-zero sum does not see): the same bits unless se is zero;
  //   1 / se' follows, and so does the facing test (normals equal as numbers: `< 0` of the
  //        sum does not see the sign of a zero term).
  // Everything else is computed from the second triangle's own data.  A ray outside the two
  // conditions is re-traced literally.
  const const_leaf *items = (const const_leaf *)item_table;
  // What the pass accumulates instead of deciding per triangle:
  //  * se_lo, se_hi: the smallest and the largest |se| of both rays.  1 / se is always formed
  //    by the short sequence (reciprocal_fast); when some |se| was outside its range [2^-60,
  //    2^60] -- zero included -- both rays are re-traced.
  //  * t_sum: the sum of all distances computed, per ray.  It is finite unless some distance
  //    was a NaN or infinite (or the sum overflowed): then the ray is re-traced.  Within
  //    reciprocal_fast's range and with a finite origin and direction a triangle's t, b1 and
  //    b2 are finite, so a NaN distance cannot hide behind finite ones.
  float se_lo = __int_as_float(0x7f800000), se_hi = 0.0f;
  f32x2 t_sum = splat(0.0f);
  const float inf = __int_as_float(0x7f800000), nan = __int_as_float(0x7fc00000);
  // one triangle for both rays; thr: kTriEps, or +inf for a ray the triangle faces away from
  // (then no finite minimum passes; an infinite one comes with an infinite t and a re-trace)
  auto finish = [&](V3 s, V3 e1, V3 e2, const V3x2 &s1, f32x2 inv, f32x2 thr, int k) {
    const V3 s2 = cross(s, e1);
    const float tnum = dot(s2, e2);
    const f32x2 t = inv * splat(tnum);
    const f32x2 b1 = inv * dot(s1, splat(s));
    const f32x2 b2 = inv * dot(splat(s2), dn);
    const f32x2 w = splat(1.0f) - b1 - b2;
    if constexpr (!kBounded) t_sum = t_sum + t;
    // !(t < eps || b1 < eps || b2 < eps || w < eps), src/Trangle.cpp:17: a NaN fails no
    // comparison there and is passed over by the minimum here
    const float ma = min_raw(min3_raw(t.x, b1.x, b2.x), w.x);
    const float mb = min_raw(min3_raw(t.y, b1.y, b2.y), w.y);
    const float ta = opaque(!(ma < thr.x) ? t.x : nan);
    const float tb = opaque(!(mb < thr.y) ? t.y : nan);
    ba.fold_next(ta, b1.x, b2.x, k);
    bb.fold_next(tb, b1.y, b2.y, k);
  };
  // Sphere::intersect, src/Sphere.cpp:3-15, the cheap part.  A sphere behind the ray needs no
  // second look: with the origin outside (cc > 0) and b > 0 the larger root is negative or a
  // rounding error above zero -- sqrt(delta) is at most one ulp above b, so for b < 8192 and
  // a >= 0.5 the root is below 2^-11 < 1e-3 and the reference's `t1 < 1e-3`
  // (src/Sphere.cpp:19) rejects it.  "0 < b < 8192" is one unsigned comparison of the bit
  // patterns: bits(b) - 1 < bits(8192) - 1 (a zero wraps around, negative numbers and NaNs
  // have larger patterns); with the limit at 0 the comparison is switched off.
  const uint32_t behind_limit = a_half ? 0x46000000u - 1u : 0u;
  uint32_t sphere_bit = 1u;
  auto mark_sphere = [&](const LeafRecord &r) {
    const V3 oc = o - mk(r.a0, r.a1, r.a2);
    const float l2 = oc.x * oc.x + oc.y * oc.y + oc.z * oc.z;
    const float cc = l2 - r.b0;
    const f32x2 b = splat(2.0f) * (splat(oc.x) * dn.x + splat(oc.y) * dn.y + splat(oc.z) * dn.z);
    const f32x2 delta = b * b - qa4 * splat(cc);
    const uint32_t bit = opaque(!(l2 < r.b1) ? sphere_bit : 0u);  // origin outside
    const uint32_t limit = opaque(cc > 0 ? behind_limit : 0u);
    uint32_t ra = opaque((__float_as_uint(b.x) - 1u < limit) ? 0u : bit);
    uint32_t rb = opaque((__float_as_uint(b.y) - 1u < limit) ? 0u : bit);
    ra = opaque(delta.x < 0 ? 0u : ra);
    rb = opaque(delta.y < 0 ? 0u : rb);
    reach_a |= ra;
    reach_b |= rb;
    sphere_bit += sphere_bit;  // spheres are numbered in table order
  };
  auto test_triangles = [&](const LeafRecord &r, int k) {
    // Trangle::intersect, src/Trangle.cpp:3-28
    const V3 p0 = mk(r.a0, r.a1, r.a2), n = mk(r.a3, r.b3, r.c3);
    const V3 e1 = mk(r.b0, r.b1, r.b2), e2 = mk(r.c0, r.c1, r.c2);
    const V3x2 s1 = cross(dn, splat(e2));
    const f32x2 se = dot(s1, splat(e1));
    se_lo = min3_abs(se_lo, se.x, se.y);
    if constexpr (!kBounded) se_hi = max3_abs(se_hi, se.x, se.y);
    const f32x2 inv = f32x2{reciprocal_fast(se.x), reciprocal_fast(se.y)};
    const f32x2 facing = dot(ndir, splat(n));
    const f32x2 thr = f32x2{opaque(facing.x < 0 ? inf : kTriEps), opaque(facing.y < 0 ? inf : kTriEps)};
    finish(o - p0, e1, e2, s1, inv, thr, k);
    const int pair = __float_as_int(r.d0);  // 1 + the pair's number, 0: a triangle on its own
    if (pair != 0) {
      const float4 q0 = lds.pair_second[3 * pair - 3], q1 = lds.pair_second[3 * pair - 2],
                   q2 = lds.pair_second[3 * pair - 1];
      // s1' enters the second test through dot(s1', s) alone.  Its non-zero components are the first
      // triangle's negated; a zero component's sign (the same zero as the first triangle's, where
      // plain negation gives the other one) can only show when every product of that sum is a
      // zero: then b1 is a zero of one sign or the other.  No comparison, minimum or t_sum sees
      // that sign, and the hit's position does not either for the pairs the upload forms
      // (k_build_items: no vertex coordinate of the second triangle is -0).
      const V3x2 m1 = V3x2{-s1.x, -s1.y, -s1.z};
      finish(o - xyz(q0), xyz(q1), xyz(q2), m1, inv, thr, k + 1);
    }
  };
  int k = 0;  // the leaf's number in depth-first leaf order
  auto process = [&](const LeafRecord &r) {
    if (r.leaf >> 30) {
      mark_sphere(r);
      k++;
    } else {
      test_triangles(r, k);
      k += __float_as_int(r.d0) != 0 ? 2 : 1;
    }
  };
  // Two records per trip, each fetched while the other is worked on and without copying a
  // record from one set of scalar registers to another (the table ends with a padding
  // record).  The fetch and the wait for it are written out: scalar loads return out of
  // order, so any wait is for all of them, and left to itself the compiler moves a record's
  // fetch (or part of it) down to its first use, where the wait then meets a load just issued.
  // Here every wait is for a fetch issued one record earlier.
  {
    leaf_regs ra = leaf_fetch(items), rb;
    int i = 0;
    for (; i + 1 < n_items; i += 2) {
      leaf_ready(ra);
      rb = leaf_fetch(items + i + 1);
#ifdef RTR_EXP_SALU  // development switch: extra scalar instructions per leaf record, to price them
      {
        unsigned xx;
#pragma unroll
        for (int qq = 0; qq < 2 * RTR_EXP_SALU; qq++) asm volatile("s_mov_b32 %0, 0" : "=s"(xx));
      }
#endif
#ifdef RTR_EXP_VALU  // development switch: extra plain vector instructions per leaf record
      {
        float xx = 0;
#pragma unroll
        for (int qq = 0; qq < 2 * RTR_EXP_VALU; qq++) asm volatile("v_mov_b32 %0, 0" : "=v"(xx));
      }
#endif
      process(leaf_unpack(ra));
      leaf_ready(rb);
      ra = leaf_fetch(items + i + 2);
      process(leaf_unpack(rb));
    }
    leaf_ready(ra);
    if (i < n_items) process(leaf_unpack(ra));
  }
  reach_a = valid_a ? reach_a : 0u;
  reach_b = valid_b ? reach_b : 0u;
  clk.mark(2);
  RTR_STAT(0, 1);
  RTR_STAT(3, __popcll(__ballot(valid_a)));
  RTR_STAT(4, __popcll(__ballot(valid_b)));
  RTR_STAT(5, __popcll(__ballot(valid_a || valid_b)));
  // phase 2: the marked spheres, src/Sphere.cpp:3-39 -- first the first ray's, then the second
  // ray's (one loop for both would have every lane select its ray's direction, best hit and
  // reciprocal by a lane mask, and write the result back through two more).
  //  * sqrt(delta) by sqrt_ranged; d_lo, d_hi: the smallest and largest non-zero delta it was
  //    given (its range is [2^-100, 2^100]; a marked sphere has delta >= 0);
  //  * (-b +- sq) / 2 / a by div_corrected, which needs a in [0.5, 2] and a numerator that is
  //    +0 or of magnitude in [2^-40, 2^40]: n_tiny is set by a numerator below 2^-40 that is not
  //    zero, n_hi is the largest magnitude.  (+0 is common -- a ray that leaves a sphere it
  //    sits on -- and goes through div_corrected as +0, like the IEEE quotient; -0 only comes
  //    out of b = +0 and sq = +0, where the other root is +0 as well and the sphere is not
  //    hit.)
  //  * the distances go into t_sum like the triangles'.
  float d_lo = inf, d_hi = 0.0f, n_hi = 0.0f;
  uint32_t n_tiny = 0u;
  auto marked_spheres = [&](uint32_t reach, V3 d, float a, float y, BestHit &best, float &tsum) {
    for (;;) {
      const bool act = reach != 0u;
      if (__ballot(act) == 0) break;
      RTR_STAT(1, 1);
      RTR_STAT(2, __popcll(__ballot(act)));
      const int j = act ? __ffs((int)reach) - 1 : 0;
      reach &= reach - 1u;  // 0 stays 0
      const float4 q = lds.sph_leaf[j], aux = lds.sph_aux[j];
      const int k = __float_as_int(aux.z);
      // the arithmetic of phase 1 again, for this ray alone: same operations, same bits
      const V3 oc = o - xyz(q);
      const float l2 = oc.x * oc.x + oc.y * oc.y + oc.z * oc.z;
      const float b = 2 * (oc.x * d.x + oc.y * d.y + oc.z * d.z);
      const float cc = l2 - aux.x;
      // (a lane without a mark works on sphere 0 with a harmless discriminant)
      const float delta = opaque(act ? b * b - 4 * a * cc : 1.0f);
      // delta == 0 is not rare (a shadow ray that grazes the light: one light sample in a
      // thousand): its root is 0, taken without the sequence
      const bool grazing = delta == 0.0f;
      const float ds = opaque(grazing ? 1.0f : delta);
      d_lo = min_raw(d_lo, ds);
      if constexpr (!kBounded) d_hi = max3_abs(d_hi, ds, ds);  // (bounded: delta is below 2^51)
      const float sq = opaque(grazing ? delta : sqrt_ranged(ds));
      const float n1 = (-b + sq) / 2, n2 = (-b - sq) / 2;
      if constexpr (!kBounded) n_hi = max3_abs(n_hi, n1, n2);  // (bounded: the numerators are below 2^26)
      n_tiny |= opaque((tiny_below(n1, 0x2b800000u) || tiny_below(n2, 0x2b800000u)) ? 1u : 0u);
      const float t1 = div_corrected(n1, a, y);
      const float t2 = div_corrected(n2, a, y);
      const float t = (t2 < 0) ? t1 : t2;
      if constexpr (!kBounded) tsum = tsum + (act ? t : 0.0f);
      // a hit's distance, or NaN: no hit (src/Sphere.cpp:19), or a lane without a mark
      float tt = opaque(!(t1 < kSphEps) ? t : nan);
      tt = opaque(act ? tt : nan);
      // equal distances: the later leaf wins (see above); then "tt <= best.t" decides
      const int kk = opaque(tt == best.t ? (uint32_t)best.k : 0xffffffffu);
      tt = opaque(k > kk ? tt : nan);
      const bool take = tt <= best.t;
      // (b1, b2 keep what an earlier triangle left: nothing reads them for a sphere)
      best.t = take ? tt : best.t;
      best.k = take ? k : best.k;
    }
  };
  float ts_a = t_sum.x, ts_b = t_sum.y;
  marked_spheres(reach_a, dn_a, qa.x, ya.x, ba, ts_a);
  marked_spheres(reach_b, dn_b, qa.y, ya.y, bb, ts_b);
  clk.mark(3);
  // the verdict on the short cuts taken (`<` and `>=` are false for a NaN)
  const bool shared_ok = se_lo >= 8.673617379884035e-19f && se_hi <= 1152921504606846976.0f &&  // 2^-60, 2^60
                         d_lo >= 7.888609052210118e-31f && d_hi <= 1.2676506002282294e30f &&   // 2^-100, 2^100
                         n_hi <= 1099511627776.0f && n_tiny == 0u &&                           // 2^40
                         (a_fast || (reach_a | reach_b) == 0u);
  bool retrace_a = !(shared_ok && fabsf(ts_a) < inf), retrace_b = !(shared_ok && fabsf(ts_b) < inf);
  // direction components of magnitude in [2^-20, 2]: what the mirrored pairs and the division-free
  // box test below both ask of a ray
  const bool dir_ok_a = min3_abs3(dn_a.x, dn_a.y, dn_a.z) >= kFastLo && max3_abs3(dn_a.x, dn_a.y, dn_a.z) <= 2.0f;
  const bool dir_ok_b = min3_abs3(dn_b.x, dn_b.y, dn_b.z) >= kFastLo && max3_abs3(dn_b.x, dn_b.y, dn_b.z) <= 2.0f;
  if (has_pairs) {
    retrace_a = retrace_a || !dir_ok_a;
    retrace_b = retrace_b || !dir_ok_b;
  }
  HitPair out;
  {
    const int2 la = lds.leaf_of[ba.k >= 0 ? ba.k : 0], lb = lds.leaf_of[bb.k >= 0 ? bb.k : 0];
    out.a = HitRec{ba.k >= 0 ? ba.t : 0.0f, ba.b1, ba.b2, ba.k >= 0 ? la.y : -1};
    out.b = HitRec{bb.k >= 0 ? bb.t : 0.0f, bb.b1, bb.b2, bb.k >= 0 ? lb.y : -1};
    out.node_a = ba.k >= 0 ? la.x : -1;
    out.node_b = bb.k >= 0 ? lb.x : -1;
  }
  // Both winners' boxes (see trace_all).  Where the planes are in the division-free test's range
  // and boxes are nested, the leaf's own box decides and the test is that one box by slab_fast;
  // a ray outside its preconditions (a direction component outside [2^-20, 2], an origin component
  // that is neither zero nor of magnitude in [2^-20, 2^20]) is re-traced.  Elsewhere: leaf to root,
  // with the IEEE divisions when some lane needs them.
  bool ok_a, ok_b;
  if (kBounded || (sc.fast_planes != 0 && sc.nested_boxes != 0)) {
    // origin: twice the magnitude's bit pattern less one wraps a zero to the top, so the least of
    // the three is at least 2 * bits(2^-20) - 1 unless some component is tiny and not zero
    const uint32_t ox = __float_as_uint(o.x), oy = __float_as_uint(o.y), oz = __float_as_uint(o.z);
    const bool o_ok = min3_u32(ox + ox - 1u, oy + oy - 1u, oz + oz - 1u) >= 2u * 0x35800000u - 1u &&
                      max3_abs3(o.x, o.y, o.z) <= kFastHi;
    retrace_a = retrace_a || !(o_ok && dir_ok_a);
    retrace_b = retrace_b || !(o_ok && dir_ok_b);
    auto leaf_box_ok = [&](int node, V3 d) {
      const V3 inv = mk(reciprocal_fast(d.x), reciprocal_fast(d.y), reciprocal_fast(d.z));
      const int at = node >= 0 ? node : 0;
      const float4 lo = lds.nodes[2 * at], hi = lds.nodes[2 * at + 1];
      return slab_fast(lo, hi, o, d, inv) || node < 0;
    };
    ok_a = leaf_box_ok(out.node_a, dn_a);
    ok_b = leaf_box_ok(out.node_b, dn_b);
  } else {
    auto chain_ok = [&](int node, V3 d) {
      const bool fast = sc.fast_planes != 0 && fast_ray(o, d);
      // RN(1/d): a fast ray's direction components are within reciprocal_fast's range
      V3 inv;
      if (__ballot(!fast) == 0) inv = mk(reciprocal_fast(d.x), reciprocal_fast(d.y), reciprocal_fast(d.z));
      else inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
      bool ok = true;
      while (node >= 0) {
        const float4 lo = lds.nodes[2 * node], hi = lds.nodes[2 * node + 1];
        const bool pass = fast ? slab_fast(lo, hi, o, d, inv) : slab(lo, hi, o, d);
        ok = ok && pass;
        node = lds.parent[node];
      }
      return ok;
    };
    ok_a = chain_ok(out.node_a, dn_a);
    ok_b = chain_ok(out.node_b, dn_b);
  }
  unsigned long long need_a = __ballot(valid_a && (retrace_a || !ok_a));
  unsigned long long need_b = __ballot(valid_b && (retrace_b || !ok_b));
  RTR_STAT(6, __popcll(need_a) + __popcll(need_b));
#ifdef RTR_CAUSE_STATS
  RTR_CAUSE(6, __popcll(need_a) + __popcll(need_b));
#endif
#ifdef RTR_CAUSE_STATS  // development: why rays were flagged for the literal evaluation
  RTR_CAUSE(0, 1);
  RTR_CAUSE(1, __popcll(__ballot(!(se_lo >= 8.673617379884035e-19f && se_hi <= 1152921504606846976.0f))));
  RTR_CAUSE(2, __popcll(__ballot(!(d_lo >= 7.888609052210118e-31f && d_hi <= 1.2676506002282294e30f))));
  RTR_CAUSE(3, __popcll(__ballot(!(n_hi <= 1099511627776.0f))));
  RTR_CAUSE(4, __popcll(__ballot(n_tiny != 0u)));
  RTR_CAUSE(5, __popcll(__ballot(!(fabsf(ts_a) < inf) || !(fabsf(ts_b) < inf))));
  RTR_CAUSE(7, __popcll(__ballot(!(a_fast || (reach_a | reach_b) == 0u))));
#endif
  if (need_a | need_b) {
    LdsGeom g{lds.nodes, lds.tris, lds.sphs};
    while (need_a) {
      int l = __ffsll((long long)need_a) - 1;
      if ((int)lane == l) {
        out.a = trace_tree(g, o, dir_a, dn_a, merge);
        out.node_a = node_of_leaf(lds, sc.n_nodes, out.a.leaf);
      }
      need_a &= need_a - 1;
    }
    while (need_b) {
      int l = __ffsll((long long)need_b) - 1;
      if ((int)lane == l) {
        out.b = trace_tree(g, o, dir_b, dn_b, merge);
        out.node_b = node_of_leaf(lds, sc.n_nodes, out.b.leaf);
      }
      need_b &= need_b - 1;
    }
  }
  clk.mark(4);
  return out;
}

} // namespace rtr
