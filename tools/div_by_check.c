// div_by() of ansel_amd/csrc/bilat.hip on the host: is the quotient's sequence with the denominator's part hoisted --
// y1 = fma(fma(-d, y, 1), y, y) once; q = n * y1, q1 = fma(fma(-d, q, n), y1, q), result = fma(fma(-d, q1, n), y1, q1)
// per numerator -- the correctly rounded n / d for EVERY reciprocal seed y within 2 ulp of 1 / d (v_rcp_f32: 1 ulp),
// for n = 0 and n in [2^-40, 2^8), d in [2^-3, 2^24)?  (Every operation scales exactly with the operands' exponents, so
// the ranges stand for any operands whose quotient and residuals stay normal.)
//   gcc -O2 -ffp-contract=off -o div_by_check tools/div_by_check.c -lm && ./div_by_check [cases, default 4e8]
// prints "<bad> bad of <cases x 5>"; tests/test_div_by.py runs 2e7 cases.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
static float fdiv_inv(float n, float d, float y)
{
  float e = fmaf(-d, y, 1.0f);
  float y1 = fmaf(e, y, y);
  float q = n * y1;
  float r = fmaf(-d, q, n);
  float q1 = fmaf(r, y1, q);
  float r1 = fmaf(-d, q1, n);
  return fmaf(r1, y1, q1);
}
int main(int argc, char **argv)
{
  const long cases = argc > 1 ? atol(argv[1]) : 400000000L;
  uint64_t s = 88172645463325252ull;
  long bad = 0, tot = 0;
  for(long it = 0; it < cases; it++)
  {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    uint32_t a = (uint32_t)s, b = (uint32_t)(s >> 32);
    // n in [2^-40, 2^8), d in [2^-3, 2^24)
    uint32_t nb = ((87u + (a >> 23) % 48u) << 23) | (a & 0x7fffff);
    uint32_t db = ((124u + (b >> 23) % 27u) << 23) | (b & 0x7fffff);
    float n, d; memcpy(&n, &nb, 4); memcpy(&d, &db, 4);
    if((it & 15) == 0) n = 0.0f;
    float y0 = 1.0f / d;
    float want = n / d;
    for(int k = -2; k <= 2; k++)
    {
      float y = y0;
      for(int t = 0; t < abs(k); t++) y = nextafterf(y, k < 0 ? 0.0f : INFINITY);
      float got = fdiv_inv(n, d, y);
      tot++;
      if(memcmp(&got, &want, 4)) { if(bad < 10) printf("n=%a d=%a y=%a k=%d got %a want %a\n", n, d, y, k, got, want); bad++; }
    }
  }
  printf("%ld bad of %ld\n", bad, tot);
  return bad != 0;
}
