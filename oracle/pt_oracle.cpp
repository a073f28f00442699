/*
 * pt_oracle.cpp -- CPU restatement of the reference path tracer's hot path.
 *
 * TEST INFRASTRUCTURE ONLY: the shipped library never includes, links or calls
 * this file.  It is the checker the HIP kernels are compared against, and the
 * "port" CPU baseline of the benchmark.
 *
 * The restatement follows the reference expression by expression, including
 * the float/double promotion each literal causes and the order of every
 * addition, because the closest-hit result is compared bit for bit.  It must
 * be compiled without FMA contraction (see oracle/Makefile).
 *
 * Parity pin: checked against the reference built from its own sources
 * (oracle/_ref/libref.so) by tests/test_oracle_vs_ref.py, and against the
 * vectors in tests/golden/ that were produced by that build.
 */
#include "pt_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

const double kPi = 3.1415926; /* include/Global.h:4 */

struct V3 {
  float x, y, z;
};
inline V3 mk(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
inline V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
/* include/Vector.h:139-141: products summed left to right */
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* include/Vector.h:147-149 */
inline V3 cross(V3 a, V3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - b.z * a.x, a.x * b.y - a.y * b.x);
}
/* include/Vector.h:79 */
inline float length(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
/* include/Vector.h:84-87: three divisions by the length */
inline V3 unit(V3 a) {
  float l = length(a);
  return mk(a.x / l, a.y / l, a.z / l);
}
/* std::max / std::min comparison directions */
inline float maxf(float a, float b) { return (a < b) ? b : a; }
inline float minf(float a, float b) { return (b < a) ? b : a; }
/* include/Global.h:15-18 */
inline float clampf(float lo, float hi, float v) { return maxf(lo, minf(hi, v)); }

/* ---- portable sin/cos: fixed IEEE fp32 sequence, valid on [0, 2*pi] ---- */
void sincos_portable(float x, float *s_out, float *c_out) {
  float kf = floorf(x * 0.636619772f + 0.5f); /* nearest multiple of pi/2 */
  int k = (int)kf;
  /* pi/2 split so that kf*hi is exact for kf <= 8 */
  float r = x - kf * 1.5703125f;
  r = r - kf * 4.837512969970703125e-4f;
  r = r - kf * 7.54978995489188e-8f;
  float z = r * r;
  float sp = -1.9515295891e-4f;
  sp = sp * z + 8.3321608736e-3f;
  sp = sp * z + -1.6666654611e-1f;
  float s = r + r * z * sp;
  float cp = 2.443315711809948e-5f;
  cp = cp * z + -1.388731625493765e-3f;
  cp = cp * z + 4.166664568298827e-2f;
  float c = 1.0f - 0.5f * z + z * z * cp;
  switch (k & 3) {
    case 0: *s_out = s; *c_out = c; break;
    case 1: *s_out = c; *c_out = -s; break;
    case 2: *s_out = -s; *c_out = -c; break;
    default: *s_out = -c; *c_out = s; break;
  }
}

struct Math {
  int mode;
  void sincos(float x, float *s, float *c) const {
    if (mode == ORC_MATH_LIBM) {
      *s = sinf(x);
      *c = cosf(x);
    } else {
      sincos_portable(x, s, c);
    }
  }
  float sq(float t) const { return mode == ORC_MATH_LIBM ? powf(t, 2) : t * t; }
};

/* ---- geometry ---- */
struct Box {
  V3 lo, hi;
  /* include/Bound3.h:10 */
  Box() {
    lo = mk(std::numeric_limits<float>::max(), std::numeric_limits<float>::max(),
            std::numeric_limits<float>::max());
    hi = mk(std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest(),
            std::numeric_limits<float>::lowest());
  }
  Box(V3 a, V3 b) : lo(a), hi(b) {}
  /* include/Bound3.h:14-20 */
  void merge(const Box &b) {
    V3 nlo = mk(lo.x < b.lo.x ? lo.x : b.lo.x, lo.y < b.lo.y ? lo.y : b.lo.y,
                lo.z < b.lo.z ? lo.z : b.lo.z);
    V3 nhi = mk(hi.x > b.hi.x ? hi.x : b.hi.x, hi.y > b.hi.y ? hi.y : b.hi.y,
                hi.z > b.hi.z ? hi.z : b.hi.z);
    lo = nlo;
    hi = nhi;
  }
  /* include/Bound3.h:22: min*0.5 + max*0.5 with 0.5 narrowed to float */
  V3 center() const { return lo * 0.5f + hi * 0.5f; }
};

/* src/Bound3.cpp:3-22 */
bool slab(const Box &b, V3 org, V3 dir) {
  V3 d = unit(dir);
  float ax = (b.lo.x - org.x) / d.x, bx = (b.hi.x - org.x) / d.x;
  if (ax > bx) std::swap(ax, bx);
  float ay = (b.lo.y - org.y) / d.y, by = (b.hi.y - org.y) / d.y;
  if (ay > by) std::swap(ay, by);
  float az = (b.lo.z - org.z) / d.z, bz = (b.hi.z - org.z) / d.z;
  if (az > bz) std::swap(az, bz);
  float tnear = ax > ay ? (ax > az ? ax : az) : (ay > az ? ay : az);
  float tfar = bx < by ? (bx < bz ? bx : bz) : (by < bz ? by : bz);
  return tnear <= tfar;
}

struct Mat {
  int kind;
  V3 emission, kd, ks;
  float ior, roughness;
  /* include/Material.h:65 */
  bool emissive() const {
    return (double)emission.x > 1e-4 || (double)emission.y > 1e-4 || (double)emission.z > 1e-4;
  }
};

struct Hit {
  bool hit = false;
  float t = 0;
  V3 pos{0, 0, 0}, nrm{0, 0, 0};
  int prim = -1;
};

struct Tri {
  V3 p0, p1, p2, n;
  float area;
  Box box;
  int mat, prim, object;
};
struct Sph {
  V3 c;
  float r, area;
  Box box;
  int mat, prim, object;
};

struct Node {
  Box box;
  int item = -1; /* leaf payload (object id at top level, triangle index in a mesh) */
  int left = -1, right = -1;
};

struct Tree {
  std::vector<Node> nodes;
  int root = -1;
};

struct Object {
  int kind; /* 0 sphere, 1 mesh */
  int mat;
  int sphere = -1;
  std::vector<int> tris; /* mesh: triangle indices, construction order */
  Tree tree;             /* mesh: nested BVH */
  Box box;
  float area;
};

struct Counters {
  uint64_t queries = 0, boxes = 0, tris = 0, spheres = 0;
};
Counters g_counters;

} // namespace

struct orc_scene {
  V3 cam_pos, cam_front, cam_up;
  float fov;
  std::vector<Mat> mats;
  std::vector<Tri> tris;
  std::vector<Sph> sphs;
  std::vector<Object> objs;
  std::vector<int> lights; /* object ids, add order (include/Scene.h:37-42) */
  Tree top;
  int n_prims = 0;
  std::vector<int> prim_mat;
  Math math{ORC_MATH_LIBM};
};

namespace {

/* src/BVH.cpp:19-64.  `items` is sorted in place exactly like objs there. */
template <class BoxOf>
int build_tree(Tree &tr, std::vector<int> &items, const BoxOf &box_of) {
  if (items.empty()) return -1;
  int id = (int)tr.nodes.size();
  tr.nodes.emplace_back();
  Box b;
  for (size_t i = 0; i < items.size(); i++) b.merge(box_of(items[i]));
  tr.nodes[id].box = b;
  if (items.size() == 1) {
    tr.nodes[id].item = items[0];
    return id;
  }
  float xl = b.hi.x - b.lo.x, yl = b.hi.y - b.lo.y, zl = b.hi.z - b.lo.z;
  int axis = xl > yl ? (xl > zl ? 0 : 2) : (yl > zl ? 1 : 2);
  if (axis == 0)
    std::sort(items.begin(), items.end(),
              [&](int a, int c) { return box_of(a).center().x < box_of(c).center().x; });
  else if (axis == 1)
    std::sort(items.begin(), items.end(),
              [&](int a, int c) { return box_of(a).center().y < box_of(c).center().y; });
  else
    std::sort(items.begin(), items.end(),
              [&](int a, int c) { return box_of(a).center().z < box_of(c).center().z; });
  std::vector<int> l(items.begin(), items.begin() + items.size() / 2);
  std::vector<int> r(items.begin() + items.size() / 2, items.end());
  int li = build_tree(tr, l, box_of);
  int ri = build_tree(tr, r, box_of);
  tr.nodes[id].left = li;
  tr.nodes[id].right = ri;
  return id;
}

/* src/Trangle.cpp:3-28 */
Hit hit_triangle(const Tri &T, V3 org, V3 dir, Counters *cn) {
  if (cn) cn->tris++;
  Hit h;
  if (dot(-dir, T.n) < 0) return h;
  V3 e1 = T.p1 - T.p0, e2 = T.p2 - T.p0, s = org - T.p0;
  V3 dn = unit(dir);
  V3 s1 = cross(dn, e2), s2 = cross(s, e1);
  float se = dot(s1, e1);
  float t = 1 / se * dot(s2, e2);
  float b1 = 1 / se * dot(s1, s);
  float b2 = 1 / se * dot(s2, dn);
  if (t < -1e-5 || b1 < -1e-5 || b2 < -1e-5 || (1 - b1 - b2) < -1e-5) return h;
  h.hit = true;
  h.t = t;
  h.pos = T.p0 * (1 - b1 - b2) + T.p1 * b1 + T.p2 * b2;
  h.nrm = T.n;
  h.prim = T.prim;
  return h;
}

/* src/Sphere.cpp:3-39 */
Hit hit_sphere(const Sph &S, V3 org, V3 dir, Counters *cn) {
  if (cn) cn->spheres++;
  Hit h;
  if (length(org - S.c) < S.r) return h;
  V3 o = org - S.c;
  V3 d = unit(dir);
  float a = d.x * d.x + d.y * d.y + d.z * d.z;
  float b = 2 * (o.x * d.x + o.y * d.y + o.z * d.z);
  float c = o.x * o.x + o.y * o.y + o.z * o.z - S.r * S.r;
  float delta = b * b - 4 * a * c;
  if (delta < 0) return h;
  float t1 = (-b + sqrtf(delta)) / 2 / a;
  float t2 = (-b - sqrtf(delta)) / 2 / a;
  if (t1 < 1e-3) return h;
  float t = (t2 < 0) ? t1 : t2;
  h.hit = true;
  h.t = t;
  h.pos = org + d * t;
  h.nrm = unit(h.pos - S.c);
  h.prim = S.prim;
  return h;
}

/* src/BVH.cpp:75-76: strict-less keeps the left result, ties and NaN go right */
inline Hit merge_hits(const Hit &l, const Hit &r) {
  Hit out;
  out = l.hit ? l : out;
  out = r.hit ? (out.hit ? (out.t < r.t ? out : r) : r) : out;
  return out;
}

Hit walk_mesh(const orc_scene *s, const Object &m, int node, V3 org, V3 dir, Counters *cn) {
  Hit h;
  const Node &nd = m.tree.nodes[node];
  if (cn) cn->boxes++;
  if (slab(nd.box, org, dir)) {
    if (nd.left < 0 && nd.right < 0) {
      h = hit_triangle(s->tris[nd.item], org, dir, cn);
    } else {
      Hit l = walk_mesh(s, m, nd.left, org, dir, cn);
      Hit r = walk_mesh(s, m, nd.right, org, dir, cn);
      h = merge_hits(l, r);
    }
  }
  return h;
}

/* src/BVH.cpp:66-80 over the top tree; include/Trangle.h:99-101 for meshes */
Hit walk_top(const orc_scene *s, int node, V3 org, V3 dir, Counters *cn) {
  Hit h;
  const Node &nd = s->top.nodes[node];
  if (cn) cn->boxes++;
  if (slab(nd.box, org, dir)) {
    if (nd.left < 0 && nd.right < 0) {
      const Object &ob = s->objs[nd.item];
      if (ob.kind == 0)
        h = hit_sphere(s->sphs[ob.sphere], org, dir, cn);
      else
        h = walk_mesh(s, ob, ob.tree.root, org, dir, cn);
    } else {
      Hit l = walk_top(s, nd.left, org, dir, cn);
      Hit r = walk_top(s, nd.right, org, dir, cn);
      h = merge_hits(l, r);
    }
  }
  return h;
}

inline Hit closest(const orc_scene *s, V3 org, V3 dir, Counters *cn) {
  if (cn) cn->queries++;
  if (s->top.root < 0) return Hit();
  return walk_top(s, s->top.root, org, dir, cn);
}

/* ---- materials: src/Material.cpp ---- */

/* src/Material.cpp:44-64 */
float fresnel_kr(V3 I, V3 N, float ior) {
  float cosi = clampf(-1.0f, 1.0f, dot(I, N));
  float etai = 1, etat = ior;
  if (cosi > 0) std::swap(etai, etat);
  float sint = etai / etat * sqrtf(maxf(0.f, 1 - cosi * cosi));
  if (sint >= 1) return 1;
  float cost = sqrtf(maxf(0.f, 1 - sint * sint));
  cosi = fabsf(cosi);
  float Rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
  float Rp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
  return (Rs * Rs + Rp * Rp) / 2;
}

/* src/Material.cpp:3-14 */
float ggx_d(float roughness, V3 N, V3 H) {
  float a = roughness * roughness;
  float a2 = a * a;
  float nh = maxf(dot(unit(N), unit(H)), 0.0f);
  float nh2 = nh * nh;
  float den = (float)((double)nh2 * ((double)a2 - 1.0) + 1.0);
  den = (float)(kPi * (double)den * (double)den);
  return a2 / maxf(den, 0.0001f);
}

/* src/Material.cpp:24-42 */
float smith_g(float roughness, V3 N, V3 V, V3 L) {
  float r = (float)((double)roughness + 1.0);
  float k = (float)((double)(r * r) / 8.0);
  float nv = maxf(dot(N, V), 0.0f);
  float nl = maxf(dot(N, L), 0.0f);
  float g2 = nv / (float)((double)nv * (1.0 - (double)k) + (double)k);
  float g1 = nl / (float)((double)nl * (1.0 - (double)k) + (double)k);
  return g1 * g2;
}

/* src/Material.cpp:66-95 */
V3 mat_eval(const Mat &m, V3 wi, V3 wo, V3 n) {
  V3 res = mk(0, 0, 0);
  switch (m.kind) {
    case ORC_DIFFUSE: {
      float c = dot(unit(wo), unit(n));
      float f = (float)(c > 0.0f ? 1.0f / kPi : (double)0.0f);
      res = m.kd * f;
    } break;
    case ORC_SPECULAR: {
      V3 h = unit(wi) + unit(wo);
      h = unit(h);
      V3 N = unit(n);
      res = ((double)dot(h, N) > (double)1.0f - 1e-3) ? m.kd : mk(0, 0, 0);
    } break;
    case ORC_MICROFACET: {
      float kr = fresnel_kr(unit(wi), unit(n), m.ior);
      float d = ggx_d(m.roughness, unit(n), unit(unit(wi) + unit(wo)));
      float g = smith_g(m.roughness, unit(n), unit(wi), unit(wo));
      float co = dot(unit(wo), unit(n));
      float ci = dot(unit(wi), unit(n));
      res = m.ks * (kr * d * g / 4.0f / maxf(co * ci, 0.0001f)) + m.kd * (1 - kr) / (float)kPi;
    } break;
  }
  return res;
}

/* src/Material.cpp:149-168 */
float mat_pdf(const Mat &m) {
  switch (m.kind) {
    case ORC_DIFFUSE: return (float)(0.5f / kPi);
    case ORC_SPECULAR: return 1.0f;
    case ORC_MICROFACET: return (float)(0.5f / kPi);
  }
  return 0;
}

/* src/Material.cpp:97-147; include/Material.h:80-84 for the mirror.
 * draw order: alph first, then theta. */
template <class Rng>
V3 mat_sample(const Mat &m, const Math &mm, V3 wi, V3 n, Rng &rng) {
  if (m.kind == ORC_SPECULAR) {
    V3 i = unit(-wi);
    V3 nn = unit(n);
    return i - 2.0f * dot(nn, i) * nn;
  }
  float alph = (float)((double)(rng() * 2.0f) * kPi);
  float theta = (float)((double)(rng() * 2.0f) * kPi);
  float sa, ca, st, ct;
  mm.sincos(alph, &sa, &ca);
  mm.sincos(theta, &st, &ct);
  V3 wo = mk(ca * ct, st, sa * ct);
  if (dot(wo, n) < 0) wo = -wo;
  return wo;
}

/* ---- light sampling ---- */

/* src/Sphere.cpp:41-48; theta is drawn before phi */
template <class Rng>
void sample_sphere(const Sph &S, const Math &mm, Rng &rng, Hit &out, float &pdf) {
  float theta = (float)(2.0 * kPi * (double)rng());
  float phi = (float)(kPi * (double)rng());
  float st, ct, sp, cp;
  mm.sincos(theta, &st, &ct);
  mm.sincos(phi, &sp, &cp);
  V3 d = mk(cp, sp * ct, sp * st);
  out.pos = S.c + S.r * d;
  out.nrm = d;
  out.prim = S.prim;
  pdf = 1.0f / S.area;
}

/* src/Trangle.cpp:42-49 */
template <class Rng>
void sample_triangle(const Tri &T, Rng &rng, Hit &out, float &pdf) {
  float b1 = rng();
  float b2 = (float)((double)rng() * (1.0 - (double)b1));
  out.pos = (1 - b1 - b2) * T.p0 + b1 * T.p1 + b2 * T.p2;
  out.nrm = T.n;
  out.prim = T.prim;
  pdf = 1.0f / T.area;
}

/* src/BVH.cpp:82-95: the walk does not stop after the first sampled leaf, so
 * every later leaf re-samples and the last leaf in DFS order wins. */
template <class Rng>
void mesh_sample_walk(const orc_scene *s, const Object &m, int node, Rng &rng, Hit &out,
                      float &pdf, float &rest) {
  const Node &nd = m.tree.nodes[node];
  if (nd.left < 0 && nd.right < 0) {
    const Tri &T = s->tris[nd.item];
    rest -= T.area;
    if ((double)rest < 1e-3) {
      sample_triangle(T, rng, out, pdf);
      pdf *= T.area;
    }
  } else {
    mesh_sample_walk(s, m, nd.left, rng, out, pdf, rest);
    mesh_sample_walk(s, m, nd.right, rng, out, pdf, rest);
  }
}

template <class Rng>
void sample_object(const orc_scene *s, int obj, Rng &rng, Hit &out, float &pdf) {
  const Object &ob = s->objs[obj];
  if (ob.kind == 0) {
    sample_sphere(s->sphs[ob.sphere], s->math, rng, out, pdf);
  } else {
    /* include/Trangle.h:103-108 */
    pdf = -1.0f;
    float rest = rng() * ob.area;
    mesh_sample_walk(s, ob, ob.tree.root, rng, out, pdf, rest);
    pdf /= ob.area;
  }
}

/* src/Scene.cpp:16-35.  Caller presets pdf = -1 (src/Render.cpp:91). */
template <class Rng>
void sample_light(const orc_scene *s, Rng &rng, Hit &out, float &pdf) {
  float area = 0;
  for (size_t i = 0; i < s->lights.size(); i++) area += s->objs[s->lights[i]].area;
  float pick = rng() * area;
  for (size_t i = 0; i < s->lights.size(); i++) {
    if ((double)pick < 1e-3) {
      sample_object(s, s->lights[i], rng, out, pdf);
      break;
    } else {
      pick -= s->objs[s->lights[i]].area;
    }
  }
  if (pdf < 0 && !s->lights.empty()) sample_object(s, s->lights.back(), rng, out, pdf);
}

/* The next-event term of src/Render.cpp:90-108.  Returns true when it adds. */
template <class Rng>
bool nee_term(const orc_scene *s, const Hit &v, const Mat &vm, V3 wi, V3 N, Rng &rng, V3 &term,
              uint64_t *rays) {
  Hit lp;
  float lpdf = -1.0f;
  sample_light(s, rng, lp, lpdf);
  if (!(lpdf > 0)) return false;
  V3 l = unit(lp.pos - v.pos);
  if (rays) (*rays)++;
  Hit th = closest(s, v.pos, l, nullptr);
  if (!th.hit) return false;
  float dl = length(th.pos - lp.pos);
  if (!((double)dl >= -1e-3 && (double)dl <= 1e-3)) return false;
  V3 l2 = unit(l);
  const Mat &lm = s->mats[s->prim_mat[lp.prim]];
  term = lm.emission * mat_eval(vm, wi, l2, N) * dot(N, l2) * dot(unit(lp.nrm), -l2) /
         s->math.sq(th.t) / lpdf;
  return true;
}

/* src/Render.cpp:77-130, literally (the recursion re-intersects the probe ray) */
template <class Rng>
V3 trace_recursive(const orc_scene *s, V3 org, V3 dir, Rng &rng, float rr, int depth,
                   int max_depth, uint64_t *st) {
  V3 res = mk(0, 0, 0);
  if (st) st[2]++;
  Hit v = closest(s, org, dir, nullptr);
  V3 wi = -unit(dir);
  V3 N = unit(v.nrm);
  if (!v.hit) return res;
  const Mat &vm = s->mats[s->prim_mat[v.prim]];
  if (vm.emissive()) return vm.emission;
  if (st) st[3]++;
  V3 term;
  uint64_t nee_rays = 0;
  if (nee_term(s, v, vm, wi, N, rng, term, &nee_rays)) res = res + term;
  if (st) { st[1] += nee_rays; st[2] += nee_rays; }
  if (rng() < rr) {
    V3 wo = mat_sample(vm, s->math, wi, N, rng);
    float pdf = mat_pdf(vm);
    V3 w1 = unit(wo);
    if (st) { st[1]++; st[2]++; }
    Hit probe = closest(s, v.pos, w1, nullptr);
    if (probe.hit && !s->mats[s->prim_mat[probe.prim]].emissive()) {
      if (max_depth <= 0 || depth + 1 < max_depth) {
        V3 rec = trace_recursive(s, v.pos, w1, rng, rr, depth + 1, max_depth, st);
        res = res + rec * mat_eval(vm, wi, unit(w1), N) * dot(N, unit(wo)) / pdf / rr;
      }
    }
  }
  return res;
}

/* ---- counter-based stream ---- */
inline void philox_round(uint32_t c[4], uint32_t k0, uint32_t k1) {
  uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  uint32_t n1 = (uint32_t)p1;
  uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
void philox10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  memcpy(out, c, sizeof(c));
}

/* Uniforms of one path vertex: number `dim` is word dim&3 of Philox block
 * (pixel, sample, bounce, dim>>2) under key (seed, 0). */
struct VertexStream {
  uint32_t seed, pixel, sample, bounce, dim;
  uint32_t cache[4];
  int cached_block;
  uint32_t key1; /* 0: path tracing; 1: photon paths (pixel = path index, sample = 0) */
  VertexStream(uint32_t se, uint32_t px, uint32_t sa, uint32_t bo, uint32_t k1 = 0)
      : seed(se), pixel(px), sample(sa), bounce(bo), dim(0), cached_block(-1), key1(k1) {}
  float operator()() {
    int blk = (int)(dim >> 2);
    if (blk != cached_block) {
      uint32_t c[4] = {pixel, sample, bounce, (uint32_t)blk};
      uint32_t k[2] = {seed, key1};
      philox10(c, k, cache);
      cached_block = blk;
    }
    uint32_t w = cache[dim & 3];
    dim++;
    return (float)(w >> 8) * 5.9604644775390625e-8f; /* 2^-24 */
  }
};

struct CallbackRng {
  orc_rng_fn fn;
  void *user;
  float operator()() { return fn(user); }
};

/* Reference recursion, but each vertex draws from its own counter stream. */
V3 trace_recursive_stream(const orc_scene *s, V3 org, V3 dir, const Hit *known, uint32_t seed,
                          uint32_t pixel, uint32_t sample, float rr, int depth, int max_depth,
                          uint64_t *st) {
  V3 res = mk(0, 0, 0);
  Hit v;
  if (known) {
    v = *known;
  } else {
    v = closest(s, org, dir, nullptr);
  }
  if (st) st[2]++;
  V3 wi = -unit(dir);
  V3 N = unit(v.nrm);
  if (!v.hit) return res;
  const Mat &vm = s->mats[s->prim_mat[v.prim]];
  if (vm.emissive()) return vm.emission;
  if (st) st[3]++;
  VertexStream rng(seed, pixel, sample, (uint32_t)depth);
  V3 term;
  uint64_t nee_rays = 0;
  if (nee_term(s, v, vm, wi, N, rng, term, &nee_rays)) res = res + term;
  if (st) { st[1] += nee_rays; st[2] += nee_rays; }
  if (rng() < rr) {
    V3 wo = mat_sample(vm, s->math, wi, N, rng);
    float pdf = mat_pdf(vm);
    V3 w1 = unit(wo);
    if (st) { st[1]++; st[2]++; }
    Hit probe = closest(s, v.pos, w1, nullptr);
    if (probe.hit && !s->mats[s->prim_mat[probe.prim]].emissive()) {
      if (max_depth <= 0 || depth + 1 < max_depth) {
        V3 rec = trace_recursive_stream(s, v.pos, w1, &probe, seed, pixel, sample, rr, depth + 1,
                                        max_depth, st);
        res = res + rec * mat_eval(vm, wi, unit(w1), N) * dot(N, unit(wo)) / pdf / rr;
      }
    }
  }
  return res;
}

/* Forward form of the same estimator: radiance accumulates as
 * L += beta (*) nee, beta = beta (*) f * cos / pdf / rr.  Same draws, same
 * rays, different rounding from the recursion.  This is what the wavefront
 * kernels compute. */
V3 trace_forward(const orc_scene *s, V3 org, V3 dir, const Hit &first, uint32_t seed,
                 uint32_t pixel, uint32_t sample, float rr, int max_depth, uint64_t *st) {
  V3 L = mk(0, 0, 0);
  if (st) st[2]++;
  if (!first.hit) return L;
  Hit v = first;
  V3 wi = -unit(dir);
  {
    const Mat &m0 = s->mats[s->prim_mat[v.prim]];
    if (m0.emissive()) return m0.emission;
  }
  V3 beta = mk(1, 1, 1);
  for (int depth = 0;; depth++) {
    V3 N = unit(v.nrm);
    const Mat &vm = s->mats[s->prim_mat[v.prim]];
    if (st) st[3]++;
    VertexStream rng(seed, pixel, sample, (uint32_t)depth);
    V3 term;
    uint64_t nee_rays = 0;
    if (nee_term(s, v, vm, wi, N, rng, term, &nee_rays)) L = L + beta * term;
    if (st) { st[1] += nee_rays; st[2] += nee_rays; }
    if (!(rng() < rr)) break;
    V3 wo = mat_sample(vm, s->math, wi, N, rng);
    float pdf = mat_pdf(vm);
    V3 w1 = unit(wo);
    if (st) { st[1]++; st[2]++; }
    Hit probe = closest(s, v.pos, w1, nullptr);
    if (!probe.hit || s->mats[s->prim_mat[probe.prim]].emissive()) break;
    if (max_depth > 0 && depth + 1 >= max_depth) break;
    if (st) st[2]++; /* the reference re-traces the probe ray on entry */
    beta = beta * mat_eval(vm, wi, unit(w1), N) * dot(N, unit(wo)) / pdf / rr;
    wi = -unit(w1);
    v = probe;
  }
  return L;
}

/* ---- photon emission paths: the body of Render::createPhotonMap's thread function,
 * src/Render.cpp:410-461 ---- */
struct PhotonRec {
  V3 pos, wi, power;
};

/* One pass of the outer `while (k < endIndex)` loop: sample a light point, then bounce.
 * rng_of(b) is the draw source of bounce b (the light sample draws from rng_of(0) first).
 * Stores the photons the reference stores in photonMap, at most `room` of them (the
 * reference's `k < endIndex`); returns their number, or -1 when the scene has no light
 * (the reference leaves its loop then, :413-415). */
template <class RngOf>
int photon_path(const orc_scene *s, RngOf &rng_of, PhotonRec *out, int room, uint64_t *rays) {
  Hit inter;
  float pdf = -1.0f;
  sample_light(s, rng_of(0), inter, pdf);
  if (inter.prim < 0) return -1;
  const Mat *im = &s->mats[s->prim_mat[inter.prim]];
  PhotonRec pre;
  pre.pos = inter.pos;
  pre.wi = unit(inter.nrm);
  pre.power = im->emission;
  int k = 0, max_deep = 64;
  for (int b = 0;; b++) {
    max_deep--;
    auto &rng = rng_of(b);
    V3 pre_wo = mat_sample(*im, s->math, pre.wi, unit(inter.nrm), rng);
    /* Vector3f::norm() is the squared length, compared with a double literal (:427) */
    if ((double)(pre_wo.x * pre_wo.x + pre_wo.y * pre_wo.y + pre_wo.z * pre_wo.z) < 1e-5) break;
    pre_wo = unit(pre_wo);
    if (rays) (*rays)++;
    Hit next = closest(s, pre.pos, pre_wo, nullptr);
    if (!next.hit) break;
    const Mat *nm = &s->mats[s->prim_mat[next.prim]];
    if (nm->emissive()) break;
    PhotonRec ph;
    ph.pos = next.pos;
    ph.wi = -pre_wo;
    if (!im->emissive()) {
      ph.power = mat_eval(*im, pre.wi, pre_wo, unit(inter.nrm)) * pre.power *
                 dot(unit(inter.nrm), pre_wo) * dot(unit(next.nrm), ph.wi);
    } else {
      ph.power = pre.power * dot(unit(inter.nrm), pre_wo) * dot(unit(next.nrm), ph.wi);
    }
    if (im->kind == 0) out[k++] = ph; /* stored only when the vertex left is DIFFUSE (:451) */
    pre = ph;
    inter = next;
    im = nm;
    /* :461 -- the roulette draw is only made when the first two conditions hold */
    if (!(k < room && max_deep >= 0)) break;
    if (!(rng() < 0.6f || im->kind != 0)) break;
  }
  return k;
}

/* every bounce of path `path` draws from its own counter stream */
struct PhotonStreams {
  uint32_t seed, path;
  int bounce;
  VertexStream cur;
  PhotonStreams(uint32_t se, uint32_t pa) : seed(se), path(pa), bounce(0), cur(se, pa, 0, 0, 1) {}
  VertexStream &operator()(int b) {
    if (b != bounce) {
      cur = VertexStream(seed, path, 0, (uint32_t)b, 1);
      bounce = b;
    }
    return cur;
  }
};
template <class Rng>
struct OneSource {
  Rng &r;
  Rng &operator()(int) { return r; }
};

struct CameraSetup {
  V3 eye, front, hstep, vstep;
};
/* src/Render.cpp:639-649 */
CameraSetup camera_setup(const orc_scene *s, int H) {
  CameraSetup c;
  c.eye = s->cam_pos;
  c.front = s->cam_front;
  V3 right = unit(cross(s->cam_front, s->cam_up));
  double scale = (double)tanf((float)((double)s->fov / 2.0 / kPi / 2.0)) * 2.0 / (double)(float)H;
  c.vstep = (float)scale * s->cam_up;
  c.hstep = (float)scale * right;
  return c;
}
/* src/Render.cpp:654 */
inline V3 pixel_dir(const CameraSetup &c, int W, int H, int i, int j) {
  return unit(c.front + (float)(-W / 2 + i) * c.hstep + (float)(-H / 2 + j) * c.vstep);
}

} // namespace

/* ======================= C interface ======================= */
extern "C" {

orc_scene *orc_scene_new(const float p[3], const float f[3], const float u[3], float fov) {
  orc_scene *s = new orc_scene();
  s->cam_pos = mk(p[0], p[1], p[2]);
  s->cam_front = mk(f[0], f[1], f[2]);
  s->cam_up = mk(u[0], u[1], u[2]);
  s->fov = fov;
  return s;
}
void orc_scene_free(orc_scene *s) { delete s; }
void orc_set_math(orc_scene *s, int mode) { s->math.mode = mode; }

int orc_add_material(orc_scene *s, int kind, const float e[3], const float kd[3],
                     const float ks[3], float ior, float roughness) {
  Mat m;
  m.kind = kind;
  m.emission = mk(e[0], e[1], e[2]);
  m.kd = mk(kd[0], kd[1], kd[2]);
  m.ks = mk(ks[0], ks[1], ks[2]);
  /* include/Material.h:58-59 */
  m.ior = (float)(ior > 0 ? (double)ior : 1e-5);
  m.roughness = (float)(roughness > 0 ? (double)(roughness < 1.0f ? roughness : 1.0f) : 1e-5);
  s->mats.push_back(m);
  return (int)s->mats.size() - 1;
}

int orc_add_sphere(orc_scene *s, int material, const float c[3], float radius) {
  Sph S;
  S.c = mk(c[0], c[1], c[2]);
  S.r = radius > 0 ? radius : 0;
  S.box = Box(S.c - mk(S.r, S.r, S.r), S.c + mk(S.r, S.r, S.r));
  S.area = (float)(4.0 * kPi * (double)S.r * (double)S.r);
  S.mat = material;
  S.prim = -1;
  S.object = (int)s->objs.size();
  s->sphs.push_back(S);
  Object ob;
  ob.kind = 0;
  ob.mat = material;
  ob.sphere = (int)s->sphs.size() - 1;
  ob.box = S.box;
  ob.area = S.area;
  s->objs.push_back(ob);
  if (s->mats[material].emissive()) s->lights.push_back(S.object);
  return S.object;
}

int orc_add_mesh(orc_scene *s, int material, const float *pos, int n_pos, const uint32_t *idx,
                 int n_idx) {
  (void)n_pos;
  Object ob;
  ob.kind = 1;
  ob.mat = material;
  ob.area = 0;
  int object = (int)s->objs.size();
  for (int i = 0; i + 2 < n_idx; i += 3) {
    Tri T;
    const float *a = pos + 3 * idx[i], *b = pos + 3 * idx[i + 1], *c = pos + 3 * idx[i + 2];
    T.p0 = mk(a[0], a[1], a[2]);
    T.p1 = mk(b[0], b[1], b[2]);
    T.p2 = mk(c[0], c[1], c[2]);
    /* include/Trangle.h:28-37 with the NaN normal the mesh passes (:77,:87) */
    T.n = unit(cross(T.p1 - T.p0, T.p2 - T.p0));
    T.area = length(cross(T.p1 - T.p0, T.p2 - T.p0));
    /* src/Trangle.cpp:30-40 */
    V3 lo, hi;
    lo.x = T.p0.x < T.p1.x ? (T.p0.x < T.p2.x ? T.p0.x : T.p2.x) : (T.p1.x < T.p2.x ? T.p1.x : T.p2.x);
    lo.y = T.p0.y < T.p1.y ? (T.p0.y < T.p2.y ? T.p0.y : T.p2.y) : (T.p1.y < T.p2.y ? T.p1.y : T.p2.y);
    lo.z = T.p0.z < T.p1.z ? (T.p0.z < T.p2.z ? T.p0.z : T.p2.z) : (T.p1.z < T.p2.z ? T.p1.z : T.p2.z);
    hi.x = T.p0.x > T.p1.x ? (T.p0.x > T.p2.x ? T.p0.x : T.p2.x) : (T.p1.x > T.p2.x ? T.p1.x : T.p2.x);
    hi.y = T.p0.y > T.p1.y ? (T.p0.y > T.p2.y ? T.p0.y : T.p2.y) : (T.p1.y > T.p2.y ? T.p1.y : T.p2.y);
    hi.z = T.p0.z > T.p1.z ? (T.p0.z > T.p2.z ? T.p0.z : T.p2.z) : (T.p1.z > T.p2.z ? T.p1.z : T.p2.z);
    T.box = Box(lo, hi);
    T.mat = material;
    T.prim = -1;
    T.object = object;
    ob.area += T.area;
    s->tris.push_back(T);
    ob.tris.push_back((int)s->tris.size() - 1);
  }
  std::vector<int> items = ob.tris;
  ob.tree.root = build_tree(ob.tree, items, [&](int t) -> const Box & { return s->tris[t].box; });
  ob.box = ob.tree.nodes[ob.tree.root].box;
  s->objs.push_back(ob);
  if (s->mats[material].emissive()) s->lights.push_back(object);
  return object;
}

void orc_build(orc_scene *s) {
  /* unified primitive ids: objects in add order, a mesh's triangles in index order */
  int p = 0;
  s->prim_mat.clear();
  for (size_t o = 0; o < s->objs.size(); o++) {
    Object &ob = s->objs[o];
    if (ob.kind == 0) {
      s->sphs[ob.sphere].prim = p++;
      s->prim_mat.push_back(ob.mat);
    } else {
      for (size_t k = 0; k < ob.tris.size(); k++) {
        s->tris[ob.tris[k]].prim = p++;
        s->prim_mat.push_back(ob.mat);
      }
    }
  }
  s->n_prims = p;
  s->top = Tree();
  std::vector<int> items(s->objs.size());
  for (size_t i = 0; i < items.size(); i++) items[i] = (int)i;
  s->top.root = build_tree(s->top, items, [&](int o) -> const Box & { return s->objs[o].box; });
}

orc_scene *orc_default_scene(void) {
  /* src/main.cpp:7-84 */
  const float cam[3] = {256, 256, 512}, front[3] = {0, 0, -1}, up[3] = {0, 1, 0};
  orc_scene *s = orc_scene_new(cam, front, up, 45);
  V3 le = 8.0f * mk(0.747f + 0.058f, 0.747f + 0.258f, 0.747f) +
          15.6f * mk(0.740f + 0.287f, 0.740f + 0.160f, 0.740f) +
          18.4f * mk(0.737f + 0.642f, 0.737f + 0.159f, 0.737f);
  const float zero[3] = {0, 0, 0};
  auto diffuse = [&](float r, float g, float b) {
    float kd[3] = {r, g, b};
    return orc_add_material(s, ORC_DIFFUSE, zero, kd, zero, 1e-5f, 1e-5f);
  };
  float lev[3] = {le.x, le.y, le.z};
  int m_light = orc_add_material(s, ORC_DIFFUSE, lev, zero, zero, 1e-5f, 1e-5f);
  int m_white = diffuse(0.6f, 0.6f, 0.6f);
  int m_red = diffuse(0.6f, 0.01f, 0.01f);
  int m_green = diffuse(0.01f, 0.6f, 0.01f);
  int m_blue = diffuse(0.01f, 0.01f, 0.6f);
  int m_yellow = diffuse(0.6f, 0.6f, 0.01f);
  int m_pink = diffuse(0.01f, 0.6f, 0.6f);
  int m_cyan = diffuse(0.6f, 0.01f, 0.6f);
  const float one[3] = {1, 1, 1};
  int m_spec = orc_add_material(s, ORC_SPECULAR, zero, one, zero, 1e-5f, 1e-5f);
  const float mk3[3] = {0.6f, 0.01f, 0.01f};
  int m_micro = orc_add_material(s, ORC_MICROFACET, zero, mk3, mk3, 0, 0.5f);
  const float cube[8 * 3] = {0, 0, -512, 512, 0, -512, 0, 512, -512, 512, 512, -512,
                             0, 0, 0,    512, 0, 0,    0, 512, 0,    512, 512, 0};
  const uint32_t back[6] = {0, 1, 2, 3, 2, 1}, top[6] = {0, 4, 1, 5, 1, 4},
                 bottom[6] = {2, 3, 6, 7, 6, 3}, left[6] = {0, 2, 4, 6, 4, 2},
                 right[6] = {1, 5, 3, 7, 3, 5};
  const float lc[3] = {256, 256, -256};
  orc_add_sphere(s, m_light, lc, 20);
  orc_add_mesh(s, m_red, cube, 8, back, 6);
  orc_add_mesh(s, m_white, cube, 8, top, 6);
  orc_add_mesh(s, m_yellow, cube, 8, bottom, 6);
  orc_add_mesh(s, m_pink, cube, 8, left, 6);
  orc_add_mesh(s, m_cyan, cube, 8, right, 6);
  const float c_red[3] = {256, 512 - 64, -256}, c_blue[3] = {64, 256, -256},
              c_green[3] = {512 - 64, 256, -256}, c_micro[3] = {256, 64, -256};
  orc_add_sphere(s, m_red, c_red, 32);
  orc_add_sphere(s, m_blue, c_blue, 32);
  orc_add_sphere(s, m_green, c_green, 32);
  orc_add_sphere(s, m_micro, c_micro, 32);
  const float s1[3] = {64, 64, -512 + 64}, s2[3] = {64, 512 - 64, -512 + 64},
              s3[3] = {512 - 64, 64, -512 + 64}, s4[3] = {512 - 64, 512 - 64, -512 + 64};
  orc_add_sphere(s, m_spec, s1, 64);
  orc_add_sphere(s, m_spec, s2, 64);
  orc_add_sphere(s, m_spec, s3, 64);
  orc_add_sphere(s, m_spec, s4, 64);
  (void)m_white;
  orc_build(s);
  return s;
}

int orc_num_prims(const orc_scene *s) { return s->n_prims; }
int orc_num_objects(const orc_scene *s) { return (int)s->objs.size(); }
int orc_num_lights(const orc_scene *s) { return (int)s->lights.size(); }
float orc_object_area(const orc_scene *s, int o) { return s->objs[o].area; }

namespace {
struct Flat {
  std::vector<float> lo, hi;
  std::vector<int32_t> child, prim;
};
int flat_mesh(const orc_scene *s, const Object &m, int node, Flat &f) {
  const Node &nd = m.tree.nodes[node];
  int id = (int)f.prim.size();
  f.lo.insert(f.lo.end(), {nd.box.lo.x, nd.box.lo.y, nd.box.lo.z});
  f.hi.insert(f.hi.end(), {nd.box.hi.x, nd.box.hi.y, nd.box.hi.z});
  f.child.push_back(-1);
  f.child.push_back(-1);
  f.prim.push_back(-1);
  if (nd.left < 0 && nd.right < 0) {
    f.prim[id] = s->tris[nd.item].prim;
  } else {
    int l = flat_mesh(s, m, nd.left, f);
    int r = flat_mesh(s, m, nd.right, f);
    f.child[2 * id] = l;
    f.child[2 * id + 1] = r;
  }
  return id;
}
int flat_top(const orc_scene *s, int node, Flat &f) {
  const Node &nd = s->top.nodes[node];
  if (nd.left < 0 && nd.right < 0) {
    const Object &ob = s->objs[nd.item];
    if (ob.kind == 1) return flat_mesh(s, ob, ob.tree.root, f); /* same box as the leaf */
    int id = (int)f.prim.size();
    f.lo.insert(f.lo.end(), {nd.box.lo.x, nd.box.lo.y, nd.box.lo.z});
    f.hi.insert(f.hi.end(), {nd.box.hi.x, nd.box.hi.y, nd.box.hi.z});
    f.child.push_back(-1);
    f.child.push_back(-1);
    f.prim.push_back(s->sphs[ob.sphere].prim);
    return id;
  }
  int id = (int)f.prim.size();
  f.lo.insert(f.lo.end(), {nd.box.lo.x, nd.box.lo.y, nd.box.lo.z});
  f.hi.insert(f.hi.end(), {nd.box.hi.x, nd.box.hi.y, nd.box.hi.z});
  f.child.push_back(-1);
  f.child.push_back(-1);
  f.prim.push_back(-1);
  int l = flat_top(s, nd.left, f);
  int r = flat_top(s, nd.right, f);
  f.child[2 * id] = l;
  f.child[2 * id + 1] = r;
  return id;
}
void top_leaves(const orc_scene *s, int node, std::vector<int32_t> &out) {
  const Node &nd = s->top.nodes[node];
  if (nd.left < 0 && nd.right < 0) {
    out.push_back(nd.item);
  } else {
    top_leaves(s, nd.left, out);
    top_leaves(s, nd.right, out);
  }
}
} // namespace

namespace {
/* finds flattened node `want` by the walk of flat_top / flat_mesh and overwrites its box */
bool set_box_mesh(Object &m, int node, int &next, int want, const Box &b) {
  Node &nd = m.tree.nodes[node];
  if (next++ == want) {
    nd.box = b;
    return true;
  }
  if (nd.left < 0 && nd.right < 0) return false;
  return set_box_mesh(m, nd.left, next, want, b) || set_box_mesh(m, nd.right, next, want, b);
}
bool set_box_top(orc_scene *s, int node, int &next, int want, const Box &b) {
  Node &nd = s->top.nodes[node];
  if (nd.left < 0 && nd.right < 0) {
    Object &ob = s->objs[nd.item];
    if (ob.kind == 1) {
      /* the top-level leaf and the mesh's root share one flattened node: both boxes */
      if (next == want) nd.box = b;
      return set_box_mesh(ob, ob.tree.root, next, want, b);
    }
    if (next++ == want) {
      nd.box = b;
      return true;
    }
    return false;
  }
  if (next++ == want) {
    nd.box = b;
    return true;
  }
  return set_box_top(s, nd.left, next, want, b) || set_box_top(s, nd.right, next, want, b);
}
} // namespace

int orc_set_node_box(orc_scene *s, int flat_index, const float lo[3], const float hi[3]) {
  int next = 0;
  if (s->top.root < 0) return -1;
  Box b(mk(lo[0], lo[1], lo[2]), mk(hi[0], hi[1], hi[2]));
  return set_box_top(s, s->top.root, next, flat_index, b) ? 0 : -1;
}

int orc_flatten(const orc_scene *s, float *bmin3, float *bmax3, int32_t *child2, int32_t *prim) {
  Flat f;
  if (s->top.root >= 0) flat_top(s, s->top.root, f);
  int n = (int)f.prim.size();
  if (bmin3) memcpy(bmin3, f.lo.data(), sizeof(float) * 3 * n);
  if (bmax3) memcpy(bmax3, f.hi.data(), sizeof(float) * 3 * n);
  if (child2) memcpy(child2, f.child.data(), sizeof(int32_t) * 2 * n);
  if (prim) memcpy(prim, f.prim.data(), sizeof(int32_t) * n);
  return n;
}

int orc_top_leaf_order(const orc_scene *s, int32_t *object_ids) {
  std::vector<int32_t> v;
  if (s->top.root >= 0) top_leaves(s, s->top.root, v);
  if (object_ids) memcpy(object_ids, v.data(), sizeof(int32_t) * v.size());
  return (int)v.size();
}

void orc_intersect(const orc_scene *s, const float *org3, const float *dir3, int64_t n,
                   int32_t *prim, float *t, float *pos3, float *nrm3) {
  for (int64_t i = 0; i < n; i++) {
    V3 o = mk(org3[3 * i], org3[3 * i + 1], org3[3 * i + 2]);
    V3 d = mk(dir3[3 * i], dir3[3 * i + 1], dir3[3 * i + 2]);
    Hit h = closest(s, o, d, &g_counters);
    prim[i] = h.hit ? h.prim : -1;
    t[i] = h.t;
    if (pos3) { pos3[3 * i] = h.pos.x; pos3[3 * i + 1] = h.pos.y; pos3[3 * i + 2] = h.pos.z; }
    if (nrm3) { nrm3[3 * i] = h.nrm.x; nrm3[3 * i + 1] = h.nrm.y; nrm3[3 * i + 2] = h.nrm.z; }
  }
}

// include/Scene.h:49-56
void orc_check_connectable(const orc_scene *s, const float *a3, const float *b3, int64_t n,
                           uint8_t *connected) {
  for (int64_t i = 0; i < n; i++) {
    V3 a = mk(a3[3 * i], a3[3 * i + 1], a3[3 * i + 2]);
    V3 b = mk(b3[3 * i], b3[3 * i + 1], b3[3 * i + 2]);
    V3 ab = b - a;
    Hit h = closest(s, a, unit(ab), nullptr);
    // std::abs(float) against the double literal 1e-3
    connected[i] = (h.hit && (double)fabsf(h.t - length(ab)) < 1e-3) ? 1 : 0;
  }
}

int orc_slab(const float bmin[3], const float bmax[3], const float org[3], const float dir[3]) {
  Box b(mk(bmin[0], bmin[1], bmin[2]), mk(bmax[0], bmax[1], bmax[2]));
  return slab(b, mk(org[0], org[1], org[2]), mk(dir[0], dir[1], dir[2])) ? 1 : 0;
}

void orc_counters(uint64_t out[4], int reset) {
  out[0] = g_counters.queries;
  out[1] = g_counters.boxes;
  out[2] = g_counters.tris;
  out[3] = g_counters.spheres;
  if (reset) g_counters = Counters();
}

void orc_material_eval(const orc_scene *s, int material, const float *wi3, const float *wo3,
                       const float *n3, int64_t count, float *out3) {
  const Mat &m = s->mats[material];
  for (int64_t i = 0; i < count; i++) {
    V3 r = mat_eval(m, mk(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]),
                    mk(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]),
                    mk(n3[3 * i], n3[3 * i + 1], n3[3 * i + 2]));
    out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
  }
}

void orc_material_pdf(const orc_scene *s, int material, const float *, const float *,
                      const float *, int64_t count, float *out) {
  for (int64_t i = 0; i < count; i++) out[i] = mat_pdf(s->mats[material]);
}

namespace {
struct ArrayRng {
  const float *u;
  int n, used;
  float operator()() { return used < n ? u[used++] : (used++, 0.0f); }
};
} // namespace

void orc_material_sample(const orc_scene *s, int material, const float *wi3, const float *n3,
                         const float *u2, int64_t count, float *wo3) {
  const Mat &m = s->mats[material];
  for (int64_t i = 0; i < count; i++) {
    ArrayRng rng{u2 + 2 * i, 2, 0};
    V3 r = mat_sample(m, s->math, mk(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]),
                      mk(n3[3 * i], n3[3 * i + 1], n3[3 * i + 2]), rng);
    wo3[3 * i] = r.x; wo3[3 * i + 1] = r.y; wo3[3 * i + 2] = r.z;
  }
}

int orc_sample_light(const orc_scene *s, const float *u, int n_u, float pos[3], float nrm[3],
                     float *pdf, int32_t *prim) {
  ArrayRng rng{u, n_u, 0};
  Hit h;
  float p = -1.0f;
  sample_light(s, rng, h, p);
  pos[0] = h.pos.x; pos[1] = h.pos.y; pos[2] = h.pos.z;
  nrm[0] = h.nrm.x; nrm[1] = h.nrm.y; nrm[2] = h.nrm.z;
  *pdf = p;
  *prim = h.prim;
  return rng.used;
}

void orc_path_trace(const orc_scene *s, const float org[3], const float dir[3], orc_rng_fn fn,
                    void *user, float rgb[3]) {
  CallbackRng rng{fn, user};
  V3 r = trace_recursive(s, mk(org[0], org[1], org[2]), mk(dir[0], dir[1], dir[2]), rng, 0.6f, 0,
                         0, nullptr);
  rgb[0] = r.x; rgb[1] = r.y; rgb[2] = r.z;
}

void orc_path_trace_u(const orc_scene *s, const float org[3], const float dir[3], const float *u,
                      int n_u, float rgb[3], int32_t *used) {
  ArrayRng rng{u, n_u, 0};
  V3 r = trace_recursive(s, mk(org[0], org[1], org[2]), mk(dir[0], dir[1], dir[2]), rng, 0.6f, 0,
                         0, nullptr);
  rgb[0] = r.x; rgb[1] = r.y; rgb[2] = r.z;
  if (used) *used = rng.used;
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  philox10(ctr, key, out);
}

float orc_stream_uniform(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce,
                         uint32_t dim) {
  VertexStream v(seed, pixel, sample, bounce);
  v.dim = dim;
  return v();
}

void orc_primary_ray(const orc_scene *s, int W, int H, int i, int j, float org[3], float dir[3]) {
  CameraSetup c = camera_setup(s, H);
  V3 d = pixel_dir(c, W, H, i, j);
  org[0] = c.eye.x; org[1] = c.eye.y; org[2] = c.eye.z;
  dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}

void orc_render(const orc_scene *s, const orc_render_params *p, float *rgb, uint64_t stats[4]) {
  const int W = p->width, H = p->height, spp = p->spp;
  const int chunk = p->chunk_spp > 0 ? p->chunk_spp : spp;
  const int c0 = p->col_begin > 0 ? p->col_begin : 0;
  const int c1 = p->col_end > 0 ? (p->col_end < W ? p->col_end : W) : W;
  int nthreads = p->threads > 0 ? p->threads : (int)std::thread::hardware_concurrency();
  if (nthreads < 1) nthreads = 1;
  const CameraSetup cam = camera_setup(s, H);
  std::atomic<int> next_col(c0);
  std::vector<std::vector<uint64_t>> tstats(nthreads, std::vector<uint64_t>(4, 0));
  auto worker = [&](int tid) {
    uint64_t *st = tstats[tid].data();
    for (;;) {
      int i = next_col.fetch_add(1);
      if (i >= c1) break;
      for (int j = 0; j < H; j++) {
        V3 d = pixel_dir(cam, W, H, i, j);
        const uint32_t pixel = (uint32_t)(i * H + j);
        Hit first = closest(s, cam.eye, d, nullptr); /* same for every sample */
        st[1]++;
        V3 color = mk(0, 0, 0);
        for (int k0 = 0; k0 < spp; k0 += chunk) {
          V3 part = mk(0, 0, 0);
          int k1 = k0 + chunk < spp ? k0 + chunk : spp;
          for (int k = k0; k < k1; k++) {
            V3 L = p->recursive
                       ? trace_recursive_stream(s, cam.eye, d, &first, p->seed, pixel, (uint32_t)k,
                                                p->rr_prob, 0, p->max_depth, st)
                       : trace_forward(s, cam.eye, d, first, p->seed, pixel, (uint32_t)k,
                                       p->rr_prob, p->max_depth, st);
            part = part + L;
            st[0]++;
          }
          color = color + part;
        }
        color = color / (float)spp;
        rgb[3 * (size_t)pixel] = color.x;
        rgb[3 * (size_t)pixel + 1] = color.y;
        rgb[3 * (size_t)pixel + 2] = color.z;
      }
    }
  };
  if (nthreads == 1) {
    worker(0);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(worker, t);
    for (auto &t : th) t.join();
  }
  if (stats) {
    for (int k = 0; k < 4; k++) {
      stats[k] = 0;
      for (int t = 0; t < nthreads; t++) stats[k] += tstats[t][k];
    }
  }
}

int32_t orc_photon_path(const orc_scene *s, uint32_t seed, uint32_t path, int32_t room, float *pos3,
                        float *wi3, float *power3) {
  std::vector<PhotonRec> buf(66);
  PhotonStreams src(seed, path);
  int k = photon_path(s, src, buf.data(), room < 65 ? room : 65, nullptr);
  for (int j = 0; j < k; j++) {
    const V3 v[3] = {buf[j].pos, buf[j].wi, buf[j].power};
    float *dst[3] = {pos3, wi3, power3};
    for (int a = 0; a < 3; a++)
      if (dst[a]) { dst[a][3 * j] = v[a].x; dst[a][3 * j + 1] = v[a].y; dst[a][3 * j + 2] = v[a].z; }
  }
  return k;
}

int64_t orc_create_photon_map(const orc_scene *s, uint32_t seed, int64_t photon_num, float *pos3,
                              float *wi3, float *power3, uint32_t *paths_used) {
  std::vector<PhotonRec> buf(66);
  int64_t k = 0;
  uint32_t path = 0;
  int barren = 0;
  while (k < photon_num) {
    PhotonStreams src(seed, path);
    int64_t room = photon_num - k;
    int got = photon_path(s, src, buf.data(), room < 65 ? (int)room : 65, nullptr);
    if (got < 0) break;
    barren = got == 0 ? barren + 1 : 0;
    if (barren > (1 << 22)) break; /* no path stores anything: the reference would never return */
    for (int j = 0; j < got; j++, k++) {
      const V3 v[3] = {buf[j].pos, buf[j].wi, buf[j].power};
      float *dst[3] = {pos3, wi3, power3};
      for (int a = 0; a < 3; a++) { dst[a][3 * k] = v[a].x; dst[a][3 * k + 1] = v[a].y; dst[a][3 * k + 2] = v[a].z; }
    }
    path++;
    if (path == 0xffffffffu) break;
  }
  if (paths_used) *paths_used = path;
  return k;
}

int32_t orc_create_photon_map_u(const orc_scene *s, const float *u, int n_u, int32_t photon_num,
                                float *pos3, float *wi3, float *power3, int32_t *used) {
  ArrayRng rng{u, n_u, 0};
  OneSource<ArrayRng> src{rng};
  std::vector<PhotonRec> buf(66);
  int32_t k = 0;
  while (k < photon_num) {
    int room = photon_num - k;
    int got = photon_path(s, src, buf.data(), room < 65 ? room : 65, nullptr);
    if (got < 0) break;
    for (int j = 0; j < got; j++, k++) {
      const V3 v[3] = {buf[j].pos, buf[j].wi, buf[j].power};
      float *dst[3] = {pos3, wi3, power3};
      for (int a = 0; a < 3; a++) { dst[a][3 * k] = v[a].x; dst[a][3 * k + 1] = v[a].y; dst[a][3 * k + 2] = v[a].z; }
    }
    if (rng.used > n_u + 4096) break; /* ran far past the supplied draws */
  }
  if (used) *used = rng.used;
  return k;
}

unsigned char orc_tonemap(float v) {
  /* src/Render.cpp:10: (unsigned char)(255 * std::pow(clamp(0,1,v), 0.6f)) */
  return (unsigned char)(255 * powf(clampf(0, 1, v), 0.6f));
}

} /* extern "C" */
