// Device-side fp32 vector math for the path-tracing kernels (gfx950).
//
// Every expression keeps the reference's operation order (include/Vector.h) and
// is compiled with -ffp-contract=off: closest-hit results must equal the
// reference's bit for bit, so a*b+c must stay two roundings, divisions must be
// IEEE divisions (never reciprocal multiplies) and sqrt must be correctly
// rounded (hipcc's default for fp32 when fast-math is off).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RTR_DEV __device__ __forceinline__

namespace rtr {

struct V3 {
  float x, y, z;
};

RTR_DEV V3 mk(float x, float y, float z) { return V3{x, y, z}; }
RTR_DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RTR_DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RTR_DEV V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
RTR_DEV V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
RTR_DEV V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
RTR_DEV V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
#ifdef RTR_EXP_CHEAP_DIV  // development switch (wrong bits): what the IEEE vector / scalar divisions cost
RTR_DEV V3 operator/(V3 a, float s) { const float y = __builtin_amdgcn_rcpf(s); return mk(a.x * y, a.y * y, a.z * y); }
#else
RTR_DEV V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
#endif
// include/Vector.h:139-141 -- products summed left to right
RTR_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// include/Vector.h:147-149
RTR_DEV V3 cross(V3 a, V3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - b.z * a.x, a.x * b.y - a.y * b.x);
}
// include/Vector.h:79
RTR_DEV float length(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
// RN(sqrt(s)) for a normal s in [2^-100, 2^100] without the range handling of sqrtf: from the
// hardware's reciprocal square root y, g = s y is the root to about an ulp and h = y / 2 the
// factor that turns the exact residual s - g g into its correction.  That one correction is the
// correctly rounded root for every float of the range: rtr_self_test compares it with sqrtf for
// all of them (Markstein's longer sequence, with a coupled refinement of g and h first, gave the
// same bits everywhere).
RTR_DEV float sqrt_from_rsq(float s, float y) {
  const float g = s * y, h = 0.5f * y;
  const float d = __builtin_fmaf(-g, g, s);
  return __builtin_fmaf(d, h, g);
}
RTR_DEV float sqrt_ranged(float s) { return sqrt_from_rsq(s, __builtin_amdgcn_rsqf(s)); }
RTR_DEV bool tiny_below(float x, uint32_t limit_bits) {  // 0 < |x| < the float with these bits
  return ((__float_as_uint(x) & 0x7fffffffu) - 1u) < limit_bits - 1u;
}
RTR_DEV bool tiny_nonzero(float x) { return tiny_below(x, 0x21800000u); }  // 0 < |x| < 2^-60
RTR_DEV uint32_t min3_u32(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// RN(1/d) without an IEEE division.  hipcc expands 1.0f / d into v_div_scale (twice), v_rcp, six
// fma steps, v_div_fmas and v_div_fixup.  The hardware's estimate is good to one ulp, and ONE
// Newton step in fused arithmetic, y1 = y0 + (1 - d y0) y0, already lands on the correctly rounded
// reciprocal of every float whose magnitude is in [2^-60, 2^60] (reciprocal_operand_fast): not
// argued but checked -- the self-test entry point compares it with 1.0f / d for every float.
RTR_DEV bool reciprocal_operand_fast(float d) {
  const float m = fabsf(d);
  return m >= 8.673617379884035e-19f && m <= 1152921504606846976.0f;  // 2^-60, 2^60
}
RTR_DEV float reciprocal_fast(float d) {
  const float y0 = __builtin_amdgcn_rcpf(d);
  return __builtin_fmaf(__builtin_fmaf(-d, y0, 1.0f), y0, y0);
}
// a / d from y = RN(1/d) by ONE residual correction: q0 = RN(a y), r = a - d q0 (exact in an fma),
// q = RN(q0 + r y).  Equal to the IEEE quotient whenever no product or residual leaves the normal
// range -- checked for every pair of significands (tools/ubench/exact_division.hip).  The sign
// of a zero quotient is lost (-0 comes out as +0).
RTR_DEV float div_corrected(float a, float d, float y) {
  const float q = a * y;
  const float r = __builtin_fmaf(-d, q, a);
  return __builtin_fmaf(r, y, q);
}

// include/Vector.h:84-87 -- three divisions by the length.
//
// An IEEE fp32 division costs as much as seventeen plain instructions on gfx950 and sqrtf
// twenty-two, and the reference normalises eleven vectors per path vertex.  The quotients a / l
// of one normalisation share l: RN(1 / l) by reciprocal_fast, each quotient by div_corrected, the
// numerator's sign put back (zero components are common -- wall normals -- and of either sign).
// Taken when, for every lane of the wave, the sum of squares is in [2^-100, 2^100] and no
// component is a non-zero number below 2^-60: then l is in [2^-50, 2^50], every quotient is zero
// or of magnitude in [2^-110, 1], and no product or residual leaves the normal range; otherwise the
// IEEE operations.  rtr_self_test runs it on adversarial operands (numerators next to midpoints).
//
// The guard is written for the instruction rates of gfx950 (tools/ubench/issue_cycles.hip:
// additions and subtractions, integer ones included, issue at twice the rate of comparisons,
// minima, shifts to the left and anything that reads a scalar register): x + x as integers
// drops the sign and leaves twice the magnitude's bit pattern, one less wraps a zero around to
// the largest number, the smallest of the three is below 2 * bits(2^-60) - 1 exactly when some
// component is tiny and not zero; the range of s is one subtraction and one unsigned comparison.
RTR_DEV V3 unit(V3 a) {
  const float s = a.x * a.x + a.y * a.y + a.z * a.z;
#ifdef RTR_EXP_CHEAP_UNIT  // development switch (wrong bits): what the exact normalisation costs
  { const float y = __builtin_amdgcn_rsqf(s); return mk(a.x * y, a.y * y, a.z * y); }
#endif
  const uint32_t bx = __float_as_uint(a.x), by = __float_as_uint(a.y), bz = __float_as_uint(a.z);
  const uint32_t least = min3_u32(bx + bx - 1u, by + by - 1u, bz + bz - 1u);
  // bits(2^-100) = 0x0d800000, bits(2^100) = 0x71800000: negative numbers and NaNs wrap or lie above
  const bool bad = least < 2u * 0x21800000u - 1u || __float_as_uint(s) - 0x0d800000u > 0x71800000u - 0x0d800000u;
  if (__ballot(bad) == 0) {
    const float l = sqrt_ranged(s);
    const float y = reciprocal_fast(l);
    return mk(__builtin_copysignf(div_corrected(a.x, l, y), a.x), __builtin_copysignf(div_corrected(a.y, l, y), a.y),
              __builtin_copysignf(div_corrected(a.z, l, y), a.z));
  }
  const float l = sqrtf(s);
  return mk(a.x / l, a.y / l, a.z / l);
}
// unit() of a vector that is a normalised vector already, up to rounding -- most of what the
// reference normalises: it normalises the results of its own normalisations again (a ray's direction
// by the ray's maker, by Bound3::intersect and by Material::eval; a normal by the primitive, by
// Render::pathTracing and by Material::eval).  When the sum of squares s lies within a few dozen
// ulps of one (kNearOneLo ... kNearOneHi, checked for the wave; else unit()):
//   * RN(sqrt(s)) is (bits(s) >> 1) + 0x1fc00000: the bit pattern halved and re-biased is exact
//     on that window;
//   * a / l for l = 1 + e that close to one: q0 = RN(a - a e), its exact residual r = a - l q0,
//     and q = RN(q0 + r y) with y = RN(1 - e (1 + 2^-20)), a reciprocal of l rounded away from the
//     tie that l = 1 - 2^-24 would otherwise meet.  The numerator's sign is put back for a = -0.
// Both are compared with sqrtf and the IEEE quotient for EVERY s of the window and every
// numerator a float can be (zero, or of magnitude in [2^-100, 2]) by rtr_self_test -- the
// window is small enough for that, which is why no rounding argument is needed here.  Components
// that are tiny and not zero (below 2^-60, as in unit()) take unit().
constexpr uint32_t kNearOneLo = 0x3f7fffc0u, kNearOneHi = 0x3f800020u;  // 1 - 2^-18 ... 1 + 2^-18
RTR_DEV float near_one_sqrt(float s) { return __uint_as_float((__float_as_uint(s) >> 1) + 0x1fc00000u); }
struct NearOne {
  float l, e, y;
  RTR_DEV explicit NearOne(float len) : l(len), e(len - 1.0f), y(__builtin_fmaf(-(len - 1.0f), 1.00000095367431640625f, 1.0f)) {}
  RTR_DEV float quotient(float a) const {
    const float q0 = __builtin_fmaf(-a, e, a);
    const float r = __builtin_fmaf(-l, q0, a);
    const float q = __builtin_fmaf(r, y, q0);
    return __uint_as_float((__float_as_uint(q) & 0x7fffffffu) | (__float_as_uint(a) & 0x80000000u));
  }
};
RTR_DEV V3 unit_near(V3 a) {
  const float s = a.x * a.x + a.y * a.y + a.z * a.z;
#ifdef RTR_EXP_CHEAP_UNIT
  return unit(a);
#endif
  const uint32_t bx = __float_as_uint(a.x), by = __float_as_uint(a.y), bz = __float_as_uint(a.z);
  const uint32_t least = min3_u32(bx + bx - 1u, by + by - 1u, bz + bz - 1u);
  const bool bad = least < 2u * 0x21800000u - 1u || __float_as_uint(s) - kNearOneLo > kNearOneHi - kNearOneLo;
  if (__ballot(bad) != 0) return unit(a);
  const NearOne d(near_one_sqrt(s));
  return mk(d.quotient(a.x), d.quotient(a.y), d.quotient(a.z));
}
// every component of unit((1, 1, 1)): 1 / RN(sqrt(3)) = 0x3f13cd3a
constexpr float kUnitOfOnes = 0.5773502588272095f;
// `disLen >= -1e-3 && disLen <= 1e-3` (src/Render.cpp:97-98) with disLen = |v| an fp32 length
// compared as a double: the largest float not above 0.001 is 0x3a83126e, sqrtf is correctly
// rounded and hence monotonic, and kNearLight2 is the smallest float whose root exceeds that
// one -- so the test is  dot(v, v) < kNearLight2,  without the root (a NaN fails both forms;
// rtr_self_test checks the constant against sqrtf).
RTR_DEV float near_light_threshold() { return __uint_as_float(0x358637bdu); }
RTR_DEV bool near_light(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z < near_light_threshold(); }

// std::max / std::min comparison directions (NaN behaviour differs from fmaxf)
RTR_DEV float maxf(float a, float b) { return (a < b) ? b : a; }
RTR_DEV float minf(float a, float b) { return (b < a) ? b : a; }
// include/Global.h:15-18
RTR_DEV float clampf(float lo, float hi, float v) { return maxf(lo, minf(hi, v)); }

RTR_DEV V3 xyz(float4 v) { return mk(v.x, v.y, v.z); }
RTR_DEV float4 pack(V3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
RTR_DEV float4 pack(V3 v, uint32_t w) { return make_float4(v.x, v.y, v.z, __uint_as_float(w)); }
RTR_DEV uint32_t wbits(float4 v) { return __float_as_uint(v.w); }

// sin and cos of x in [0, 2*pi] as a fixed sequence of fp32 multiplies and adds:
// Cody-Waite reduction by pi/2 (three-term split, products with k <= 4 exact)
// and the classic single-precision minimax polynomials on [-pi/4, pi/4].  The
// same sequence evaluated on any IEEE-754 machine gives the same bits, which is
// what lets a CPU checker reproduce the kernels' sample values exactly; device
// libm and host libm differ in the last ulp.
RTR_DEV void sincos_fixed(float x, float &s_out, float &c_out) {
#ifdef RTR_EXP_CHEAP_SINCOS  // development switch (wrong values)
  s_out = x * 0.1f; c_out = 1.0f - x * 0.1f; return;
#endif
  float kf = floorf(x * 0.636619772f + 0.5f);
  int k = (int)kf;
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
  // quadrant k & 3: sine and cosine change places in the odd ones; the sine is negated in quadrants
  // 2 and 3, the cosine in 1 and 2 -- bit 1 of k and of k + 1, moved onto the sign bit (a negation
  // is a flip of that bit, for zeros and NaNs too)
  const bool odd = (k & 1) != 0;
  const float ss = odd ? c : s, cc = odd ? s : c;
  s_out = __uint_as_float(__float_as_uint(ss) ^ (((uint32_t)k & 2u) << 30));
  c_out = __uint_as_float(__float_as_uint(cc) ^ (((uint32_t)(k + 1) & 2u) << 30));
}

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3").
// round_keys (optional): the ten rounds' keys {k0 + r * 0x9E3779B9, k1 + r * 0xBB67AE85} in a table
// (LDS): the keys are the same for every lane, so computed here they sit in scalar registers, and an
// exclusive-or with a scalar operand issues at half rate; read from LDS they arrive in vector
// registers over a pipe the kernel leaves idle.
RTR_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                           uint32_t k1, uint32_t out[4], const uint2 *round_keys = nullptr) {
#ifdef RTR_EXP_CHEAP_RNG  // development switch (a poor generator)
  out[0] = c0 * 0x9E3779B9u + c1 + k0; out[1] = out[0] * 0x85EBCA6Bu + c2; out[2] = out[1] * 0xC2B2AE35u + c3; out[3] = out[2] * 0x27D4EB2Fu + k1;
  out[0] ^= out[3] >> 15; out[1] ^= out[0] >> 13; out[2] ^= out[1] >> 16; out[3] ^= out[2] >> 14;
  return;
#endif
#pragma unroll
  for (int r = 0; r < 10; r++) {
    // one 64-bit product per multiplier: a single v_mad_u64_u32 yields both halves
    // (separate high and low multiplies cost about half as much again)
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0, n2;
    if (round_keys) {
      const uint2 rk = round_keys[r];
      n0 = hi1 ^ c1 ^ rk.x, n2 = hi0 ^ c3 ^ rk.y;
    } else {
      n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// top 24 bits -> [0,1)
RTR_DEV float u01(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-8f; }

// Uniform number `dim` of path vertex `bounce` of sample `sample` of pixel
// `pixel`: word dim&3 of block (pixel, sample, bounce, dim>>2), key (seed, key1).
// key1 = 0: path tracing; 1: photon emission paths (pixel = path number, sample = 0).
struct VertexStream {
  uint32_t seed, pixel, sample, bounce, dim;
  uint32_t w[4];
  int block;
  uint32_t key1;
  const uint2 *round_keys = nullptr;  // philox4x32_10's table for (seed, key1), if there is one
  RTR_DEV VertexStream(uint32_t se, uint32_t px, uint32_t sa, uint32_t bo, uint32_t k1 = 0u)
      : seed(se), pixel(px), sample(sa), bounce(bo), dim(0), block(-1), key1(k1) {}
  RTR_DEV float next() {
    int b = (int)(dim >> 2);
    if (b != block) {
      philox4x32_10(pixel, sample, bounce, (uint32_t)b, seed, key1, w, round_keys);
      block = b;
    }
    uint32_t d = dim & 3u;
    uint32_t v = d == 0 ? w[0] : d == 1 ? w[1] : d == 2 ? w[2] : w[3];
    dim++;
    return u01(v);
  }
};

} // namespace rtr
