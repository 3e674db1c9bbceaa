// Host build of ansel_amd/csrc/devmath.h for tests/test_devmath.py: compares the restated
// glibc algorithms with the libm this process links, bit for bit.
//   devmath_host <func> <count> <seed>     prints "<mismatches> <count>"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cstdint>
#include "devmath.h"

static uint64_t s[2];
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t next()
{
  const uint64_t s0 = s[0];
  uint64_t s1 = s[1];
  const uint64_t r = s0 + s1;
  s1 ^= s0;
  s[0] = rotl(s0, 24) ^ s1 ^ (s1 << 16);
  s[1] = rotl(s1, 37);
  return r;
}
static inline float rnd_bits() { return ansel_math::asfloat((uint32_t)(next() >> 32)); }
static inline float rnd_range(float lo, float hi) { return lo + (hi - lo) * (float)((next() >> 40) * (1.0 / 16777216.0)); }

extern "C" {
// vector entry points, also used through ctypes
void devmath_powf(const float *x, const float *y, float *o, size_t n) { for(size_t i = 0; i < n; i++) o[i] = ansel_math::powf_exact(x[i], y[i]); }
void devmath_log2f(const float *x, float *o, size_t n) { for(size_t i = 0; i < n; i++) o[i] = ansel_math::log2f_exact(x[i]); }
void devmath_exp2f(const float *x, float *o, size_t n) { for(size_t i = 0; i < n; i++) o[i] = ansel_math::exp2f_exact(x[i]); }
void devmath_expf(const float *x, float *o, size_t n) { for(size_t i = 0; i < n; i++) o[i] = ansel_math::expf_exact(x[i]); }
void devmath_atan2f(const float *y, const float *x, float *o, size_t n) { for(size_t i = 0; i < n; i++) o[i] = ansel_math::atan2f_exact(y[i], x[i]); }
void devmath_hypotf(const float *x, const float *y, float *o, size_t n) { for(size_t i = 0; i < n; i++) o[i] = ansel_math::hypotf_exact(x[i], y[i]); }
void devmath_sinf(const float *x, float *o, size_t n) { for(size_t i = 0; i < n; i++) o[i] = ansel_math::sinf_exact(x[i]); }
void devmath_cosf(const float *x, float *o, size_t n) { for(size_t i = 0; i < n; i++) o[i] = ansel_math::cosf_exact(x[i]); }
void devmath_logf(const float *x, float *o, size_t n) { for(size_t i = 0; i < n; i++) o[i] = ansel_math::logf_exact(x[i]); }
void libm_logf(const float *x, float *o, size_t n) { for(size_t i = 0; i < n; i++) o[i] = logf(x[i]); }
void libm_sinf(const float *x, float *o, size_t n) { for(size_t i = 0; i < n; i++) o[i] = sinf(x[i]); }
void libm_cosf(const float *x, float *o, size_t n) { for(size_t i = 0; i < n; i++) o[i] = cosf(x[i]); }
void libm_fmodf(const float *x, const float *y, float *o, size_t n) { for(size_t i = 0; i < n; i++) o[i] = fmodf(x[i], y[i]); }
void libm_atan2f(const float *y, const float *x, float *o, size_t n) { for(size_t i = 0; i < n; i++) o[i] = atan2f(y[i], x[i]); }
void libm_hypotf(const float *x, const float *y, float *o, size_t n) { for(size_t i = 0; i < n; i++) o[i] = hypotf(x[i], y[i]); }
void libm_powf(const float *x, const float *y, float *o, size_t n) { for(size_t i = 0; i < n; i++) o[i] = powf(x[i], y[i]); }
void libm_log2f(const float *x, float *o, size_t n) { for(size_t i = 0; i < n; i++) o[i] = log2f(x[i]); }
void libm_exp2f(const float *x, float *o, size_t n) { for(size_t i = 0; i < n; i++) o[i] = exp2f(x[i]); }
void libm_expf(const float *x, float *o, size_t n) { for(size_t i = 0; i < n; i++) o[i] = expf(x[i]); }
}

static bool same(float a, float b)
{
  if(std::isnan(a) && std::isnan(b)) return true;
  return ansel_math::asuint(a) == ansel_math::asuint(b);
}

int main(int argc, char **argv)
{
  if(argc < 4) return 2;
  const char *fn = argv[1];
  const size_t n = strtoull(argv[2], nullptr, 10);
  s[0] = strtoull(argv[3], nullptr, 10) * 0x9E3779B97F4A7C15ull + 1;
  s[1] = 0xD1B54A32D192ED03ull;
  size_t bad = 0;
  for(size_t i = 0; i < n; i++)
  {
    float a, b, r0, r1;
    const int mode = (int)(i & 3);
    if(!strcmp(fn, "powf"))
    {
      // mode 0: arbitrary bit patterns; 1-3: the ranges the pipe uses
      if(mode == 0) { a = rnd_bits(); b = rnd_bits(); }
      else if(mode == 1) { a = rnd_range(0.f, 4.f); b = rnd_range(0.2f, 6.f); }
      else if(mode == 2) { a = rnd_range(0.f, 1.f); b = rnd_range(-3.f, 3.f); }
      else { a = fabsf(rnd_bits()); b = rnd_range(-8.f, 8.f); }
      r0 = powf(a, b); r1 = ansel_math::powf_exact(a, b);
    }
    else if(!strcmp(fn, "atan2f") || !strcmp(fn, "hypotf"))
    {
      // mode 0: arbitrary bit patterns; 1: chroma-like small components; 2: one operand dominating; 3: around the axes
      if(mode == 0) { a = rnd_bits(); b = rnd_bits(); }
      else if(mode == 1) { a = rnd_range(-0.5f, 0.5f); b = rnd_range(-0.5f, 0.5f); }
      else if(mode == 2) { a = rnd_range(-1e-3f, 1e-3f); b = rnd_range(-200.f, 200.f); if(i & 4) { const float t = a; a = b; b = t; } }
      else { a = rnd_range(-2.f, 2.f); b = (i & 4) ? 1.0f : ((i & 8) ? 0.0f : -0.0f); if(i & 16) { const float t = a; a = b; b = t; } }
      if(!strcmp(fn, "atan2f")) { r0 = atan2f(a, b); r1 = ansel_math::atan2f_exact(a, b); }
      else { r0 = hypotf(a, b); r1 = ansel_math::hypotf_exact(a, b); }
    }
    else if(!strcmp(fn, "sinf") || !strcmp(fn, "cosf"))
    {
      // mode 0: arbitrary bit patterns; 1: one turn (the hue angles of the callers); 2: the fast-reduction range; 3: beyond it
      if(mode == 0) a = rnd_bits();
      else if(mode == 1) a = rnd_range(-6.5f, 6.5f);
      else if(mode == 2) a = rnd_range(-120.f, 120.f);
      else a = rnd_range(-1e7f, 1e7f);
      b = 0;
      if(!strcmp(fn, "sinf")) { r0 = sinf(a); r1 = ansel_math::sinf_exact(a); }
      else { r0 = cosf(a); r1 = ansel_math::cosf_exact(a); }
    }
    else if(!strcmp(fn, "log2f"))
    {
      a = (mode == 0) ? rnd_bits() : ((mode == 1) ? rnd_range(0.f, 2.f) : fabsf(rnd_bits()));
      b = 0; r0 = log2f(a); r1 = ansel_math::log2f_exact(a);
    }
    else if(!strcmp(fn, "logf"))
    {
      // mode 0: arbitrary bit patterns; 1: the uniform deviates of the Box-Muller caller; 2: positive patterns; 3: around 1
      a = (mode == 0) ? rnd_bits() : ((mode == 1) ? rnd_range(0.f, 1.f) : ((mode == 2) ? fabsf(rnd_bits()) : rnd_range(0.9f, 1.1f)));
      b = 0; r0 = logf(a); r1 = ansel_math::logf_exact(a);
    }
    else if(!strcmp(fn, "exp2f"))
    {
      a = (mode == 0) ? rnd_bits() : rnd_range(-160.f, 140.f);
      b = 0; r0 = exp2f(a); r1 = ansel_math::exp2f_exact(a);
    }
    else
    {
      a = (mode == 0) ? rnd_bits() : rnd_range(-110.f, 95.f);
      b = 0; r0 = expf(a); r1 = ansel_math::expf_exact(a);
    }
    if(!same(r0, r1))
    {
      if(bad < 5) fprintf(stderr, "%s(%a, %a): libm %a  restated %a\n", fn, a, b, r0, r1);
      bad++;
    }
  }
  printf("%zu %zu\n", bad, n);
  return bad ? 1 : 0;
}
