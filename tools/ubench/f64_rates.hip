// Development micro-benchmark: issue cost of the fp64 instructions a float division by way of
// double precision would use, relative to v_fma_f32 (wave64, 8 independent chains, 4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed, int iters) {
  float f[8];
  double d[8];
  for (int i = 0; i < 8; i++) { f[i] = (float)(seed * (threadIdx.x + 1 + i)); d[i] = (double)f[i] + 0.25; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) f[i] = __builtin_fmaf(f[i], 1.0001f, 0.5f);
      if (OP == 1) d[i] = __builtin_fma(d[i], 1.0001, 0.5);
      if (OP == 2) d[i] = d[i] * 1.0001;
      if (OP == 3) d[i] = __builtin_amdgcn_rcp(d[i]);
      if (OP == 4) { d[i] = (double)f[i]; asm volatile("" : "+v"(d[i])); f[i] = __uint_as_float(__float_as_uint(f[i]) + 1u); }
      if (OP == 5) { f[i] = (float)d[i]; asm volatile("" : "+v"(f[i])); d[i] = __longlong_as_double(__double_as_longlong(d[i]) + 1); }
      if (OP == 6) f[i] = f[i] / 1.0001f;  // the compiler's IEEE fp32 division
      if (OP == 7) f[i] = __builtin_sqrtf(f[i]);
    }
  }
  uint32_t r = 0;
  for (int i = 0; i < 8; i++) r ^= __float_as_uint(f[i]) ^ (uint32_t)__double_as_longlong(d[i]) ^ (uint32_t)(__double_as_longlong(d[i]) >> 32);
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int OP>
float run(uint32_t *d, int iters) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, d, 12345u, 10);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, d, 12345u, iters);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  uint32_t *d;
  (void)hipMalloc(&d, 1024 * 256 * 4);
  const int iters = 5000;
  const char *names[] = {"v_fma_f32", "v_fma_f64", "v_mul_f64", "v_rcp_f64", "v_cvt_f64_f32 (+int add)", "v_cvt_f32_f64 (+64-bit add)",
                         "IEEE fp32 division", "IEEE sqrtf"};
  float t[8];
  t[0] = run<0>(d, iters); t[1] = run<1>(d, iters); t[2] = run<2>(d, iters); t[3] = run<3>(d, iters);
  t[4] = run<4>(d, iters); t[5] = run<5>(d, iters); t[6] = run<6>(d, iters); t[7] = run<7>(d, iters);
  for (int i = 0; i < 8; i++)
    printf("%-30s %8.3f ms  %.2f x fma  %.2f ns per operation, wave and SIMD\n", names[i], t[i], t[i] / t[0],
           t[i] * 1e6 / (4.0 * iters * 8));
  return 0;
}
