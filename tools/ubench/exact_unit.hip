// Development check: a correctly rounded fp32 square root and fp32 quotients by way of a
// double-precision reciprocal, against the compiler's IEEE sequences.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
void report_rcp_error();

__device__ __forceinline__ float sqrt_short(float s) {
  // Markstein's square root: y ~ 1/sqrt(s), then g -> sqrt(s), h -> 1/(2 sqrt(s))
  const float y = __builtin_amdgcn_rsqf(s);
  float g = s * y, h = 0.5f * y;
  const float r = __builtin_fmaf(-h, g, 0.5f);
  g = __builtin_fmaf(g, r, g);
  h = __builtin_fmaf(h, r, h);
  const float d = __builtin_fmaf(-g, g, s);
  return __builtin_fmaf(d, h, g);
}

__device__ __forceinline__ double reciprocal_f64(double l) {
  double y = __builtin_amdgcn_rcp(l);
  y = __builtin_fma(__builtin_fma(-l, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-l, y, 1.0), y, y);
  return y;
}

// all floats with bits in [lo, hi): sqrt_short against sqrtf
__global__ void k_sqrt(uint32_t lo, unsigned long long *bad, uint32_t *first) {
  const uint32_t bits = lo + blockIdx.x * 256u + threadIdx.x;
  const float s = __uint_as_float(bits);
  const float a = sqrt_short(s), b = __builtin_sqrtf(s);
  if (__float_as_uint(a) != __float_as_uint(b)) {
    if (atomicAdd(bad, 1ull) == 0) *first = bits;
  }
}

__device__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// quotients: numerator patterns x random denominators in [2^-50, 2^50]
__global__ void k_div(uint32_t round, unsigned long long *bad, uint32_t *first) {
  const uint32_t id = blockIdx.x * 256u + threadIdx.x;
  const uint32_t r0 = mix(id * 2654435761u + round), r1 = mix(r0 + 0x9e3779b9u), r2 = mix(r1 ^ round);
  // denominator: random mantissa, exponent in [-50, 50]
  const uint32_t e = 127u - 50u + r0 % 101u;
  const float l = __uint_as_float((e << 23) | (r1 & 0x7fffffu));
  const double y = reciprocal_f64((double)l);
  float a;
  const uint32_t kind = r2 & 7u;
  if (kind == 0) a = __uint_as_float(r1 ^ (r2 << 7));                       // any bit pattern
  else if (kind == 1) a = __uint_as_float((r2 >> 3) & 0x807fffffu);         // denormals and zeros
  else if (kind == 2) a = l * __uint_as_float(0x3f000000u | (r2 >> 9));     // |a| <= l, like a normalisation
  else {
    // close to a midpoint of two floats: q = m + ulp/2, a = RN(q * l) and neighbours
    const uint32_t qb = (mix(r2) & 0x3fffffffu) | 0x00800000u;
    const double q = ((double)__uint_as_float(qb) + (double)__uint_as_float(qb + 1u)) * 0.5;
    const float a0 = (float)(q * (double)l);
    const int off = (int)(kind) - 5;
    a = __uint_as_float(__float_as_uint(a0) + (uint32_t)off);
    if (r2 & 0x80000000u) a = -a;
  }
  // the guarded domain: a numerator is zero or at least 2^-60 in magnitude
  const uint32_t au = __float_as_uint(a) & 0x7fffffffu;
  if (au != 0u && au < 0x21800000u) a = 0.0f * a;
  const float want = a / l;
  const float got = (float)((double)a * y);
  const bool same = __float_as_uint(want) == __float_as_uint(got) || (want != want && got != got);
  if (!same) {
    if (atomicAdd(bad, 1ull) == 0) { first[0] = __float_as_uint(a); first[1] = __float_as_uint(l); }
  }
}

int main() {
  report_rcp_error();
  unsigned long long *bad, h = 0;
  uint32_t *first, hf[2] = {0, 0};
  (void)hipMalloc(&bad, 8); (void)hipMalloc(&first, 8);
  (void)hipMemset(bad, 0, 8); (void)hipMemset(first, 0, 8);
  // floats in [2^-100, 2^100): bits from (127-100)<<23 to (127+100)<<23
  const uint32_t lo = (127u - 100u) << 23, hi = (127u + 100u) << 23;
  for (uint32_t b = lo; b < hi; b += 1u << 28) {
    const uint32_t n = (hi - b < (1u << 28)) ? hi - b : (1u << 28);
    hipLaunchKernelGGL(k_sqrt, dim3(n / 256), dim3(256), 0, 0, b, bad, first);
  }
  (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hf, first, 8, hipMemcpyDeviceToHost);
  printf("square root, every float in [2^-100, 2^100): %llu mismatches (first at bits %08x)\n", h, hf[0]);
  (void)hipMemset(bad, 0, 8); (void)hipMemset(first, 0, 8);
  const int rounds = 32;
  for (int r = 0; r < rounds; r++) hipLaunchKernelGGL(k_div, dim3(1 << 20), dim3(256), 0, 0, (uint32_t)r * 7919u + 1u, bad, first);
  (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hf, first, 8, hipMemcpyDeviceToHost);
  printf("quotients, %.1f G pairs: %llu mismatches (first a %08x l %08x)\n", rounds * 268.435456e6 / 1e9, h, hf[0], hf[1]);
  return 0;
}

// The hardware's double-precision reciprocal estimate on the only arguments unit() gives it
// (floats converted to double: 2^23 significands): largest |1 - l y0|, as bits of a double.
__global__ void k_rcp_error(unsigned long long *worst) {
  const uint32_t m = blockIdx.x * 256u + threadIdx.x;
  const double l = (double)__uint_as_float(0x3f800000u | m);
  const double y0 = __builtin_amdgcn_rcp(l);
  const double e = __builtin_fabs(__builtin_fma(-l, y0, 1.0));
  atomicMax(worst, (unsigned long long)__double_as_longlong(e));
}
void report_rcp_error() {
  unsigned long long *w, h = 0;
  (void)hipMalloc(&w, 8); (void)hipMemset(w, 0, 8);
  hipLaunchKernelGGL(k_rcp_error, dim3(1 << 15), dim3(256), 0, 0, w);
  (void)hipMemcpy(&h, w, 8, hipMemcpyDeviceToHost);
  double e; memcpy(&e, &h, 8);
  printf("v_rcp_f64 on float significands: largest |1 - l y0| = %.3e = 2^%.2f\n", e, log2(e));
}
