/* ORACLE (test infrastructure only): replays the fma-based division by a constant that csrc/anymal_terrain.hip uses for
 * `points / horizontal_scale` (anymal_terrain.py:586-593) against IEEE fp32 division, bit for bit, over random numerators
 * and a set of divisors.  Built and run by tests/test_helpers_known_answers.py (gcc -O2 -ffp-contract=off; fmaf is exact
 * with or without hardware FMA).  Usage: div_check [samples per divisor] */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static float div_by_const(float x, float h, float inv) {
  const float q0 = x * inv;
  const float q1 = fmaf(fmaf(-q0, h, x), inv, q0);
  return fmaf(fmaf(-q1, h, x), inv, q1);
}
static uint64_t s = 88172645463325252ull;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main(int argc, char** argv) {
  const float hs[] = {0.1f, 0.25f, 0.3f, 0.05f, 1.0f/3.0f, 0.7f, 1.9999999f, 1.0000001f, 0.123456f};
  long bad = 0, total = 0;
  const long samples = argc > 1 ? atol(argv[1]) : 4000000;
  for (unsigned k = 0; k < sizeof(hs)/sizeof(hs[0]); ++k) {
    const float h = hs[k], inv = 1.0f / h;
    for (long i = 0; i < samples; ++i) {
      // x in [-16, 400): uniform bits over a range of exponents
      uint32_t bits = (uint32_t)rnd();
      float m = 1.0f + (float)(bits & 0x7fffff) / 8388608.0f;  // [1,2)
      int e = (int)((bits >> 23) & 15) - 6;                    // 2^-6 .. 2^9
      float x = ldexpf(m, e);
      if (bits >> 31) x = -x;
      const float a = x / h, b = div_by_const(x, h, inv);
      if (memcmp(&a, &b, 4) != 0) { if (bad < 5) printf("h=%.9g x=%.9g div=%.9g fast=%.9g\n", h, x, a, b); ++bad; }
      ++total;
    }
  }
  printf("total %ld mismatches %ld\n", total, bad);
  return bad != 0;
}
