// Is  q = fma(fma(-d, a*y, a), y, a*y)  with y = RN(1/d) the IEEE quotient a/d?
//
// Checked for EVERY pair of significands: a = 1.ma and d = 1.md, ma and md over all 2^23 values
// each (2^46 pairs).  Rounding to nearest is symmetric in sign and, away from under- and overflow,
// the arithmetic only sees significands (scaling a or d by a power of two scales every
// intermediate by a power of two), so this covers every pair of normal operands whose products,
// residual and quotient stay in the normal range.
//
// Also counted: the same with a second correction step (the form used until now), and whether the
// short reciprocal (v_rcp_f32 + one fused Newton step) equals 1.0f / d for every significand.
//
// usage: exact_division [first block] [blocks]   (64 blocks of 2^17 divisors cover everything;
// one JSON line per block, then a summary line).  Build: hipcc --offload-arch=gfx950 -O3
// -ffp-contract=off -o exact_division exact_division.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                       \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

struct Tally {
  unsigned long long one_step, two_steps, reciprocal, pairs;
  uint32_t first_a, first_d;  // a mismatching pair of the one-step form, if any
};

constexpr unsigned kDivisorsPerBlock = 1u << 17;  // per launch
constexpr unsigned kSlices = 8;                   // of the numerators' 2^23 significands
constexpr unsigned kPerSlice = (1u << 23) / kSlices;

__global__ void __launch_bounds__(256) k_check(uint32_t md_base, Tally *tally) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t md = md_base + (tid & (kDivisorsPerBlock - 1u));
  const uint32_t ma0 = (tid / kDivisorsPerBlock) * kPerSlice;
  const float d = __uint_as_float(0x3f800000u | md);
  const float y = 1.0f / d;  // IEEE: RN(1/d)
  const float y0 = __builtin_amdgcn_rcpf(d);
  const float ys = __builtin_fmaf(__builtin_fmaf(-d, y0, 1.0f), y0, y0);
  unsigned bad1 = 0, bad2 = 0;
  uint32_t fa = 0;
  for (uint32_t i = 0; i < kPerSlice; i++) {
    const float a = __uint_as_float(0x3f800000u | (ma0 + i));
    const float want = a / d;
    const float q0 = a * y;
    const float r0 = __builtin_fmaf(-d, q0, a);
    const float q1 = __builtin_fmaf(r0, y, q0);
    const float r1 = __builtin_fmaf(-d, q1, a);
    const float q2 = __builtin_fmaf(r1, y, q1);
    if (__float_as_uint(q1) != __float_as_uint(want)) {
      bad1++;
      fa = __float_as_uint(a);
    }
    bad2 += __float_as_uint(q2) != __float_as_uint(want) ? 1u : 0u;
  }
  if (bad1) {
    atomicAdd(&tally->one_step, (unsigned long long)bad1);
    tally->first_a = fa;
    tally->first_d = __float_as_uint(d);
  }
  if (bad2) atomicAdd(&tally->two_steps, (unsigned long long)bad2);
  if (ma0 == 0 && __float_as_uint(ys) != __float_as_uint(y)) atomicAdd(&tally->reciprocal, 1ull);
  if ((threadIdx.x & 63u) == 0) atomicAdd(&tally->pairs, 64ull * kPerSlice);
}

int main(int argc, char **argv) {
  const unsigned first = argc > 1 ? (unsigned)atoi(argv[1]) : 0u;
  const unsigned blocks = argc > 2 ? (unsigned)atoi(argv[2]) : 64u;
  Tally *dev = nullptr, sum = {}, h = {};
  CHECK(hipMalloc(&dev, sizeof(Tally)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (unsigned b = first; b < first + blocks && b < 64u; b++) {
    CHECK(hipMemset(dev, 0, sizeof(Tally)));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_check, dim3(kDivisorsPerBlock * kSlices / 256), dim3(256), 0, 0, b * kDivisorsPerBlock, dev);
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(&h, dev, sizeof(Tally), hipMemcpyDeviceToHost));
    printf("{\"divisor_significands\": [%u, %u], \"pairs\": %llu, \"one_step_mismatches\": %llu, "
           "\"two_step_mismatches\": %llu, \"short_reciprocal_mismatches\": %llu, \"ms\": %.1f}\n",
           b * kDivisorsPerBlock, (b + 1) * kDivisorsPerBlock - 1, h.pairs, h.one_step, h.two_steps, h.reciprocal, ms);
    fflush(stdout);
    sum.pairs += h.pairs;
    sum.one_step += h.one_step;
    sum.two_steps += h.two_steps;
    sum.reciprocal += h.reciprocal;
    if (h.one_step && !sum.first_a) {
      sum.first_a = h.first_a;
      sum.first_d = h.first_d;
    }
  }
  printf("{\"summary\": true, \"pairs\": %llu, \"one_step_mismatches\": %llu, \"two_step_mismatches\": %llu, "
         "\"short_reciprocal_mismatches\": %llu, \"example_a_bits\": \"0x%08x\", \"example_d_bits\": \"0x%08x\"}\n",
         sum.pairs, sum.one_step, sum.two_steps, sum.reciprocal, sum.first_a, sum.first_d);
  CHECK(hipFree(dev));
  return 0;
}
