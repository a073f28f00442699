// Materials and light sampling on the device: Material::eval / sample / pdf
// (src/Material.cpp), Sphere::sample, Trangle::sample, TrangleMesh::sample,
// Scene::samplingLight.  Expressions keep the reference's float/double
// promotions: where the reference multiplies by the double literal MY_PI
// (include/Global.h:4) the product is formed in fp64 and narrowed, exactly once.
#pragma once
#include "rtr_trace.hpp"

namespace rtr {

constexpr double kPi = 3.1415926;  // include/Global.h:4

struct DeviceMaterial {  // 3 x float4
  float4 e_kind;   // emission.xyz, bits(kind | emissive << 8)
  float4 kd_ior;   // kd.xyz, ior
  float4 ks_rough; // ks.xyz, roughness
};

struct DeviceLights {
  const float4 *light;        // per light: { area, bits(kind), bits(sphere row), 1 / area } or
                              //            { area, bits(kind), bits(tri_begin), bits(tri_end) }
  const float4 *light_le;     // per light: emission of the primitive its samples land on
  const int32_t *light_tris;  // triangle ROWS (not prim ids) in mesh leaf order
  const float *tri_area;      // per triangle row
  int32_t n_lights;
  float total_area;           // sum in Scene::lights order, fp32 (src/Scene.cpp:17-20)
};

// Where shading reads its tables from: HBM through the caches, or copies a
// workgroup staged in LDS (scenes small enough; a dependent chain of global loads
// costs a path vertex more than all its arithmetic when few waves share a SIMD).
constexpr int kLdsMats = 32, kLdsLights = 16, kLdsLightTris = kLdsTris;
struct LdsShade {
  DeviceMaterial mats[kLdsMats];
  float4 light[kLdsLights], light_le[kLdsLights];
  int32_t light_tris[kLdsLightTris];
  float tri_area[kLdsTris];
  int32_t node_info[kLdsNodes];  // leaf nodes: prim id | material << 16 | emissive << 30
  uint2 philox_keys[10];         // the generator's round keys for (seed, 0): see philox4x32_10
};
struct GlobalTables {
  const DeviceScene &sc;
  const DeviceLights &L;
  const DeviceMaterial *mats;
  RTR_DEV DeviceMaterial mat(int m) const { return mats[m]; }
  RTR_DEV float4 light(int l) const { return L.light[l]; }
  RTR_DEV float4 light_le(int l) const { return L.light_le[l]; }
  RTR_DEV int light_tri(int k) const { return L.light_tris[k]; }
  RTR_DEV float tri_area(int row) const { return L.tri_area[row]; }
  RTR_DEV float4 tri(int row, int k) const { return sc.tris[3 * row + k]; }
  RTR_DEV float4 sph(int row) const { return sc.sphs[row]; }
  RTR_DEV int n_lights() const { return L.n_lights; }
  RTR_DEV float total_area() const { return L.total_area; }
  RTR_DEV const uint2 *philox_keys() const { return nullptr; }
};
struct LdsTables {
  const LdsScene &g;
  const LdsShade &s;
  int32_t nl;
  float area;
  RTR_DEV DeviceMaterial mat(int m) const { return s.mats[m]; }
  RTR_DEV float4 light(int l) const { return s.light[l]; }
  RTR_DEV float4 light_le(int l) const { return s.light_le[l]; }
  RTR_DEV int light_tri(int k) const { return s.light_tris[k]; }
  RTR_DEV float tri_area(int row) const { return s.tri_area[row]; }
  RTR_DEV float4 tri(int row, int k) const { return g.tris[3 * row + k]; }
  RTR_DEV float4 sph(int row) const { return g.sphs[row]; }
  RTR_DEV int n_lights() const { return nl; }
  RTR_DEV float total_area() const { return area; }
  RTR_DEV const uint2 *philox_keys() const { return s.philox_keys; }
};
__host__ __device__ inline bool fits_lds_shade(int n_materials, int n_lights, int n_light_tris) {
  return n_materials <= kLdsMats && n_lights <= kLdsLights && n_light_tris <= kLdsLightTris;
}

RTR_DEV int mat_kind(const DeviceMaterial &m) { return (int)(wbits(m.e_kind) & 0xff); }
RTR_DEV bool mat_emissive(const DeviceMaterial &m) { return (wbits(m.e_kind) >> 8) & 1; }

// src/Material.cpp:44-64
RTR_DEV float fresnel_kr(V3 I, V3 N, float ior) {
  float cosi = clampf(-1.0f, 1.0f, dot(I, N));
  float etai = 1, etat = ior;
  if (cosi > 0) { float tmp = etai; etai = etat; etat = tmp; }
  float sint = etai / etat * sqrtf(maxf(0.f, 1 - cosi * cosi));
  if (sint >= 1) return 1;
  float cost = sqrtf(maxf(0.f, 1 - sint * sint));
  cosi = fabsf(cosi);
  float Rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
  float Rp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
  return (Rs * Rs + Rp * Rp) / 2;
}

// src/Material.cpp:3-14
RTR_DEV float ggx_d(float roughness, V3 N, V3 H) {
  float a = roughness * roughness;
  float a2 = a * a;
  float nh = maxf(dot(unit_near(N), unit_near(H)), 0.0f);  // both come out of a normalisation
  float nh2 = nh * nh;
  float den = (float)((double)nh2 * ((double)a2 - 1.0) + 1.0);
  den = (float)(kPi * (double)den * (double)den);
  return a2 / maxf(den, 0.0001f);
}

// src/Material.cpp:24-42
RTR_DEV float smith_g(float roughness, V3 N, V3 V, V3 L) {
  float r = (float)((double)roughness + 1.0);
  float k = (float)((double)(r * r) / 8.0);
  float nv = maxf(dot(N, V), 0.0f);
  float nl = maxf(dot(N, L), 0.0f);
  float g2 = nv / (float)((double)nv * (1.0 - (double)k) + (double)k);
  float g1 = nl / (float)((double)nl * (1.0 - (double)k) + (double)k);
  return g1 * g2;
}

// x / MY_PI for the three components of a colour (src/Material.cpp:92): three IEEE divisions by
// the same constant, by one residual correction each of x * RN(1 / pi) (div_corrected) when, for
// every lane of the wave, every component is zero or of magnitude in [2^-60, 2^60]; the numerator's
// sign is put back (a zero of either sign is common).
RTR_DEV V3 div_by_pi(V3 x) {
  const float pi = (float)kPi;
  const uint32_t bx = __float_as_uint(x.x), by = __float_as_uint(x.y), bz = __float_as_uint(x.z);
  const uint32_t least = min3_u32(bx + bx - 1u, by + by - 1u, bz + bz - 1u);  // a zero wraps to the top
  const bool bad = least < 2u * 0x21800000u - 1u || !(max3_abs3(x.x, x.y, x.z) <= 1152921504606846976.0f);
  if (__ballot(bad) == 0) {
    const float y = 1.0f / pi;  // folded by the compiler: the correctly rounded reciprocal
    return mk(__builtin_copysignf(div_corrected(x.x, pi, y), x.x), __builtin_copysignf(div_corrected(x.y, pi, y), x.y),
              __builtin_copysignf(div_corrected(x.z, pi, y), x.z));
  }
  return mk(x.x / pi, x.y / pi, x.z / pi);
}

// Material::eval, src/Material.cpp:66-95.  The reference normalises wi, wo and n
// again in every branch; ui = unit(wi) and un = unit(n) are passed in so that a
// vertex which evaluates its material twice forms them once.
RTR_DEV V3 mat_eval_units(const DeviceMaterial &m, V3 ui, V3 wo, V3 un, V3 &uo) {
  V3 res = mk(0, 0, 0);
  int kind = mat_kind(m);
#ifdef RTR_EXP_NO_MICROFACET  // development switch (wrong image): what the microfacet branch costs
  if (kind == 2) kind = 0;
#endif
  V3 kd = xyz(m.kd_ior);
  uo = unit_near(wo);  // (callers pass a normalised direction) handed back: the next vertex's unit(wi) is its negation
  if (kind == 0) {
    float c = dot(uo, un);
    float f = c > 0.0f ? (float)(1.0f / kPi) : 0.0f;
    res = kd * f;
  } else {
    // both remaining kinds start from the normalised half vector (src/Material.cpp:76-77, :86)
    const V3 h = unit(ui + uo);
    if (kind == 1) {
      res = ((double)dot(h, un) > (double)1.0f - 1e-3) ? kd : mk(0, 0, 0);
    } else {
      V3 ks = xyz(m.ks_rough);
      float rough = m.ks_rough.w;
      float kr = fresnel_kr(ui, un, m.kd_ior.w);
      float d = ggx_d(rough, un, h);
      float g = smith_g(rough, un, ui, uo);
      float co = dot(uo, un);
      float ci = dot(ui, un);
      res = ks * (kr * d * g / 4.0f / maxf(co * ci, 0.0001f)) + div_by_pi(kd * (1 - kr));
    }
  }
  return res;
}
RTR_DEV V3 mat_eval_units(const DeviceMaterial &m, V3 ui, V3 wo, V3 un) {
  V3 uo;
  return mat_eval_units(m, ui, wo, un, uo);
}
RTR_DEV V3 mat_eval(const DeviceMaterial &m, V3 wi, V3 wo, V3 n) {
  return mat_eval_units(m, unit(wi), wo, unit(n));
}

// Material::pdf, src/Material.cpp:149-168
RTR_DEV float mat_pdf(const DeviceMaterial &m) {
  return mat_kind(m) == 1 ? 1.0f : (float)(0.5f / kPi);
}

// Material::sample, src/Material.cpp:97-147 (mirror: include/Material.h:80-84).
// u1 drives alph, u2 drives theta -- the reference's draw order.
RTR_DEV V3 mat_sample_uv(V3 n, float u1, float u2) {
  float alph = (float)((double)(u1 * 2.0f) * kPi);
  float theta = (float)((double)(u2 * 2.0f) * kPi);
  float sa, ca, st, ct;
  sincos_fixed(alph, sa, ca);
  sincos_fixed(theta, st, ct);
  V3 wo = mk(ca * ct, st, sa * ct);
  if (dot(wo, n) < 0) wo = -wo;
  return wo;
}
RTR_DEV V3 mat_mirror(V3 wi, V3 n) {
  V3 i = unit(-wi);
  V3 nn = unit(n);
  return i - 2.0f * dot(nn, i) * nn;
}
// the same from ui = unit(wi), un = unit(n): unit(-wi) is -unit(wi) bit for bit
// (the length is the same and IEEE division is odd in its numerator)
RTR_DEV V3 mat_mirror_units(V3 ui, V3 un) {
  V3 i = -ui;
  return i - 2.0f * dot(un, i) * un;
}

struct LightSample {
  V3 pos, nrm;
  float pdf;
  int32_t leaf;   // (kind << 30) | row of the sampled primitive, -1 if none
  int32_t light;  // index of the light the sample was taken from
};

// Sphere::sample, src/Sphere.cpp:41-48 (theta is drawn before phi)
template <class Tab, class Rng>
RTR_DEV void sample_sphere(const Tab &T, int row, float one_by_area, Rng &rng, LightSample &ls) {
  float theta = (float)(2.0 * kPi * (double)rng.next());
  float phi = (float)(kPi * (double)rng.next());
  float st, ct, sp, cp;
  sincos_fixed(theta, st, ct);
  sincos_fixed(phi, sp, cp);
  V3 d = mk(cp, sp * ct, sp * st);
  float4 q = T.sph(row);
  ls.pos = xyz(q) + q.w * d;
  ls.nrm = d;
  ls.leaf = (1 << 30) | row;
  ls.pdf = one_by_area;  // 1.0f / area, divided once when the scene was uploaded
}

// Trangle::sample, src/Trangle.cpp:42-49
template <class Tab, class Rng>
RTR_DEV void sample_triangle(const Tab &T, int row, float area, Rng &rng, LightSample &ls) {
  float b1 = rng.next();
  float b2 = (float)((double)rng.next() * (1.0 - (double)b1));
  float4 q0 = T.tri(row, 0), q1 = T.tri(row, 1), q2 = T.tri(row, 2);
  ls.pos = (1 - b1 - b2) * xyz(q0) + b1 * xyz(q1) + b2 * xyz(q2);
  ls.nrm = mk(q0.w, q1.w, q2.w);
  ls.leaf = row;
  ls.pdf = 1.0f / area;
}

// Object::sample for light l: Sphere::sample, or TrangleMesh::sample
// (include/Trangle.h:103-108) whose walk (src/BVH.cpp:82-95) re-samples at every
// leaf from the first selected one on, so the LAST leaf decides; the draws the
// earlier leaves would have consumed are skipped by advancing the counter.
template <class Tab, class Rng>
RTR_DEV void sample_object(const Tab &T, int l, Rng &rng, LightSample &ls) {
  float4 rec = T.light(l);
  float area = rec.x;
  ls.light = l;
  if (__float_as_int(rec.y) == 1) {
    sample_sphere(T, __float_as_int(rec.z), rec.w, rng, ls);
    return;
  }
  ls.pdf = -1.0f;
  float rest = rng.next() * area;
  int b = __float_as_int(rec.z), e = __float_as_int(rec.w);
  int first = -1;
  for (int k = b; k < e; k++) {
    rest -= T.tri_area(T.light_tri(k));
    if ((double)rest < 1e-3) { first = k; break; }
  }
  if (first >= 0) {
    rng.dim += 2u * (uint32_t)(e - 1 - first);  // draws of the leaves that get overwritten
    int row = T.light_tri(e - 1);
    float a = T.tri_area(row);
    sample_triangle(T, row, a, rng, ls);
    ls.pdf *= a;
  }
  ls.pdf /= area;
}

// Scene::samplingLight, src/Scene.cpp:16-35 (caller presets pdf = -1,
// src/Render.cpp:91).
template <class Tab, class Rng>
RTR_DEV void sample_light(const Tab &T, Rng &rng, LightSample &ls) {
  ls.pdf = -1.0f;
  ls.leaf = -1;
  ls.light = 0;
  ls.pos = mk(0, 0, 0);
  ls.nrm = mk(0, 0, 0);
  const int n_lights = T.n_lights();
  float pick = rng.next() * T.total_area();
  for (int i = 0; i < n_lights; i++) {
    if ((double)pick < 1e-3) {
      sample_object(T, i, rng, ls);
      break;
    } else {
      pick -= T.light(i).x;
    }
  }
  if (ls.pdf < 0 && n_lights > 0) sample_object(T, n_lights - 1, rng, ls);
}

} // namespace rtr
