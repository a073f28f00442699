/*
 * TEST INFRASTRUCTURE.  Checks the claim the traversal kernels rely on: for fp32
 * a, d inside the stated ranges, with y = RN(1/d),
 *   q0 = a*y; r = fma(-d,q0,a); q = fma(r,y,q0)
 * equals the IEEE quotient a/d bit for bit (zeros compare equal regardless of sign).
 * Prints the number of mismatches; exit status 0 iff none.
 *   usage: check_division [samples] [seed]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t s[2];
static uint64_t next(void) { /* xorshift128+ */
  uint64_t a = s[0], b = s[1];
  s[0] = b;
  a ^= a << 23;
  s[1] = a ^ b ^ (a >> 17) ^ (b >> 26);
  return s[1] + b;
}
static float from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static float corrected(float a, float d, float y) {
  float q = a * y;
  float r = fmaf(-d, q, a);
  return fmaf(r, y, q);
}

static long bad = 0, total = 0;
static void check(float a, float d) {
  float y = 1.0f / d;
  float want = a / d, got = corrected(a, d, y);
  total++;
  if (bits(want) != bits(got) && !(want == 0.0f && got == 0.0f)) {
    if (bad < 10) printf("MISMATCH a=%a d=%a want=%a got=%a\n", a, d, want, got);
    bad++;
  }
}

/* random float with exponent in [elo, ehi] (unbiased exponent), random sign/mantissa */
static float rnd(int elo, int ehi, int pattern) {
  uint64_t r = next();
  uint32_t man = (uint32_t)(r & 0x7fffff);
  if (pattern == 1) man = 0x7fffff;           /* all ones */
  if (pattern == 2) man = 0;                  /* power of two */
  if (pattern == 3) man = (uint32_t)(r & 7);  /* tiny mantissa */
  if (pattern == 4) man = 0x7fffff - (uint32_t)(r & 7);
  int e = elo + (int)((r >> 23) % (uint64_t)(ehi - elo + 1));
  uint32_t sign = (uint32_t)((r >> 60) & 1) << 31;
  return from_bits(sign | ((uint32_t)(e + 127) << 23) | man);
}

int main(int argc, char **argv) {
  long n = argc > 1 ? atol(argv[1]) : 20000000;
  s[0] = 0x9E3779B97F4A7C15ull ^ (argc > 2 ? (uint64_t)atol(argv[2]) : 1);
  s[1] = 0xD1B54A32D192ED03ull;
  /* kernel preconditions: |d| in [2^-20, 2], a = 0 or |a| in [2^-44, 2^21] */
  for (long i = 0; i < n; i++) check(rnd(-44, 20, 0), rnd(-20, 0, 0));
  for (int pa = 0; pa < 5; pa++)
    for (int pd = 0; pd < 5; pd++)
      for (long i = 0; i < n / 50; i++) check(rnd(-44, 20, pa), rnd(-20, 0, pd));
  /* quotients that are exactly representable or exact ties' neighbours */
  for (long i = 0; i < n / 10; i++) {
    float d = rnd(-20, 0, 0), q = rnd(-20, 20, 0);
    float a = d * q; /* rounded product: a/d is within an ulp of q */
    if (fabsf(a) >= 0x1p-44f && fabsf(a) <= 0x1p21f) {
      check(a, d);
      check(nextafterf(a, 0), d);
      check(nextafterf(a, INFINITY), d);
    }
  }
  check(0.0f, 0.5f); check(-0.0f, 0.5f); check(0.0f, -0.5f);
  printf("checked %ld quotients, %ld mismatches\n", total, bad);
  return bad ? 1 : 0;
}
