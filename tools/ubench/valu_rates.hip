// Development micro-benchmark: issue cost of a few VALU instructions on gfx950,
// relative to v_fma_f32 (wave64, 8 independent chains, 4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed, int iters) {
  uint32_t a[8];
  float f[8];
  for (int i = 0; i < 8; i++) { a[i] = seed * (threadIdx.x + 1 + i); f[i] = (float)a[i]; }
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[4];
  for (int i = 0; i < 4; i++) p[i] = f2{f[2 * i], f[2 * i + 1]};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) f[i] = __builtin_fmaf(f[i], 1.0001f, 0.5f);
      if (OP == 1) a[i] = __umulhi(a[i], 0xD2511F53u) ^ (uint32_t)it;
      if (OP == 2) a[i] = a[i] * 0xCD9E8D57u + 1u;
      if (OP == 3) { unsigned long long w = (unsigned long long)a[i] * 0xD2511F53u; a[i] = (uint32_t)(w >> 32) ^ (uint32_t)w; }
      if (OP == 4) f[i] = __builtin_amdgcn_rcpf(f[i]);
      if (OP == 5 && i < 4) p[i] = __builtin_elementwise_fma(p[i], f2{1.0001f, 1.0001f}, f2{0.5f, 0.5f});
      if (OP == 6) f[i] = f[i] * 1.0001f;
      if (OP == 7) { double d = (double)f[i] * 3.1415926; f[i] = (float)d; }
    }
  }
  uint32_t r = 0;
  for (int i = 0; i < 8; i++) r ^= a[i] ^ __float_as_uint(f[i]);
  for (int i = 0; i < 4; i++) r ^= __float_as_uint(p[i].x) ^ __float_as_uint(p[i].y);
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

__global__ void __launch_bounds__(256) kclk(unsigned long long *out, int iters) {
  float f[8];
  for (int i = 0; i < 8; i++) f[i] = (float)(threadIdx.x + i);
  unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) f[i] = __builtin_fmaf(f[i], 1.0001f, 0.5f);
  }
  unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < 8; i++) s += f[i];
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = (unsigned long long)s; }
}

template <int OP>
float run(uint32_t *d, int iters, int blocks = 1024) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  uint32_t *d;
  hipMalloc(&d, 256 * 16 * 256 * 4);
  const int iters = 20000;
  const char *names[] = {"v_fma_f32 x8", "v_mul_hi_u32(+xor) x8", "v_mul_lo_u32(+add) x8", "64-bit product x8",
                         "v_rcp_f32 x8", "v_pk_fma_f32 x4", "v_mul_f32 x8", "f64 mul + cvts x8"};
  float t[8];
  t[0] = run<0>(d, iters); t[1] = run<1>(d, iters); t[2] = run<2>(d, iters); t[3] = run<3>(d, iters);
  t[4] = run<4>(d, iters); t[5] = run<5>(d, iters); t[6] = run<6>(d, iters); t[7] = run<7>(d, iters);
  // 1024 blocks x 4 waves over 1024 SIMDs: 4 waves per SIMD, each issuing iters x 8 (x 4 packed) instructions
  const int per_it[] = {8, 8, 8, 8, 8, 4, 8, 8};
  for (int i = 0; i < 8; i++)
    printf("%-28s %8.3f ms  (%.2f x fma)  %.2f ns per wave-instruction and SIMD\n", names[i], t[i], t[i] / t[0],
           t[i] * 1e6 / (4.0 * iters * per_it[i]));
  for (int blocks : {256, 512, 1024, 2048}) {
    float a = run<0>(d, iters, blocks), b = run<5>(d, iters, blocks);
    printf("blocks %4d (%d waves/SIMD): fma x8 %.3f ms, pk_fma x4 %.3f ms\n", blocks, blocks / 256, a, b);
  }
  unsigned long long *o, h[3];
  hipMalloc(&o, 24);
  hipLaunchKernelGGL(kclk, dim3(1024), dim3(256), 0, 0, o, iters);
  hipMemcpy(h, o, 24, hipMemcpyDeviceToHost);
  int wc_khz = 0;
  hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
  printf("cycle counter %llu, wall clock %llu ticks at %d kHz -> %.0f MHz shader clock if readcyclecounter counts shader cycles\n",
         h[0], h[1], wc_khz, (double)h[0] / ((double)h[1] / wc_khz / 1e3) / 1e6 * 1e3 / 1e3);
  return 0;
}
