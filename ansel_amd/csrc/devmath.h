// devmath.h -- single-precision powf / log2f / exp2f / expf that return, bit for bit, what the
// glibc 2.35 libm of this image returns on x86-64 -- on the GPU.
//
// Why not the ROCm device-libs functions: the reference's CPU path gets its transcendentals from
// libm (filmic: log2f + powf per channel, src/iop/filmicrgb.c:1047-1051, :2124-2146; colour TRCs:
// powf above white, src/colorprofiles/iop_profile.h:558-562; color calibration: powf,
// src/iop/channelmixerrgb.c:665, src/pixel/chromatic_adaptation.h:205), their results feed further
// arithmetic, and the parity bar is 1 ULP at the END of each module.  OCML's powf/log2f are
// accurate to 1-2 ULP, glibc's to < 0.52 ULP: differences of that size are amplified by the
// filmic spline and by the 3x3 matrices that follow.  So this header restates glibc's algorithms
// (ARM optimized-routines, Szabolcs Nagy: table-driven log2 in double, exp2 in double, one final
// rounding to float) with the same tables (libm_tables.h) and the same polynomial evaluation
// order.  MI355X runs v_fma_f64 at half the f32 rate, so a powf costs ~25 double operations.
//
// Contraction: on x86-64 with FMA+AVX2 (this image's Xeon, the MI355X hosts' EPYCs) glibc
// dispatches powf/log2f/exp2f/expf to its *_fma builds, in which gcc contracts every `a * b + c`
// of the source into one fused operation.  The fma() calls below are exactly those.  (The two
// variants differ in the last bit of the float result about once in 10^8 calls.)
//
// tests/test_devmath.py: host build of this header vs libm on 10^8 arguments (bit-exact), and
// on the GPU box the device build vs the host's libm on the same arguments.
#pragma once
#include <stdint.h>
#include <math.h>
#include "libm_tables.h"

#if defined(__HIPCC__)
#define ANSEL_HD __host__ __device__ __forceinline__
#define ANSEL_HDM __host__ __device__ __forceinline__ // member functions
#else
#define ANSEL_HD static inline
#define ANSEL_HDM inline
#endif

namespace ansel_math
{

#if defined(__HIP_DEVICE_COMPILE__)
ANSEL_LIBM_TABLES(static __device__ const)
#else
ANSEL_LIBM_TABLES(static const)
#endif

ANSEL_HD uint32_t asuint(const float f)
{
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  return u;
}
ANSEL_HD float asfloat(const uint32_t u)
{
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}
ANSEL_HD uint64_t asuint64(const double f)
{
  uint64_t u;
  __builtin_memcpy(&u, &f, 8);
  return u;
}
ANSEL_HD double asdouble(const uint64_t u)
{
  double f;
  __builtin_memcpy(&f, &u, 8);
  return f;
}

// ---- where the lookup tables are read from ------------------------------------------------------------------------------------
// tabs_global: the constant arrays above (on the device: global memory behind the scalar / vector caches -- a per-lane index makes
// every lookup a vector-memory round trip, two dependent ones per powf).  tabs_lds (device kernels that call powf / log2f / exp2f per
// pixel; round 6): the same 1 KB of tables copied once per workgroup into LDS (stage()), a lookup an LDS read.  Same words either way.
struct tabs_global
{
  ANSEL_HDM double powlog(const int j) const { return k_powf_log2_tab[j]; }
  ANSEL_HDM uint64_t exp2(const int j) const { return k_exp2f_tab[j]; }
  ANSEL_HDM double log2(const int j) const { return k_log2f_tab[j]; }
};
#if defined(__HIPCC__)
// the workgroup's copy (one per kernel that stages it; a kernel that never names it gets no LDS for it)
static __shared__ uint64_t libm_lds_words[32 + 32 + 32];
struct tabs_lds
{
  __device__ __forceinline__ double powlog(const int j) const { return reinterpret_cast<const double *>(libm_lds_words)[j]; }
  __device__ __forceinline__ uint64_t exp2(const int j) const { return libm_lds_words[32 + j]; }
  __device__ __forceinline__ double log2(const int j) const { return reinterpret_cast<const double *>(libm_lds_words + 64)[j]; }
};
// EVERY thread of the workgroup calls it at the top of the kernel, before any exit (threads beyond the tables' 96 words idle), then the
// workgroup meets at a barrier -- stage_tables_and_sync() -- before the first lookup
__device__ __forceinline__ void stage_tables(const int tid)
{
  static_assert(sizeof(k_powf_log2_tab) == 32 * 8 && sizeof(k_exp2f_tab) == 32 * 8 && sizeof(k_log2f_tab) == 32 * 8, "three tables of 32 words");
  if(tid < 32) libm_lds_words[tid] = asuint64(k_powf_log2_tab[tid]);
  else if(tid < 64) libm_lds_words[tid] = k_exp2f_tab[tid - 32];
  else if(tid < 96) libm_lds_words[tid] = asuint64(k_log2f_tab[tid - 64]);
}
__device__ __forceinline__ void stage_tables_and_sync(const int tid)
{
  stage_tables(tid);
  __syncthreads();
}
#endif
// A translation unit whose kernels ALL stage the tables defines ANSEL_MATH_DEFAULT_TABS as tabs_lds in front of this header: every
// lookup of its code then reads the workgroup's copy (the per-pixel device functions of px_*.h take no table argument)
#ifndef ANSEL_MATH_DEFAULT_TABS
#define ANSEL_MATH_DEFAULT_TABS tabs_global
#endif
#if defined(__HIPCC__)
// the first statement of every kernel of such a translation unit (a no-op elsewhere)
__device__ __forceinline__ void stage_default_tables(const int tid)
{
  if constexpr(__is_same(ANSEL_MATH_DEFAULT_TABS, tabs_lds)) stage_tables_and_sync(tid);
}
#endif

// ---- log2f: sysdeps/ieee754/flt-32/e_log2f.c ------------------------------------------
// The table-driven core, valid for a positive normal (or pre-normalised) bit pattern ix.
template <class Tabs = ANSEL_MATH_DEFAULT_TABS> ANSEL_HD float log2f_core(const uint32_t ix, const Tabs tabs = Tabs())
{
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (tmp >> (23 - 4)) % 16;
  const uint32_t top = tmp & 0xff800000u;
  const uint32_t iz = ix - top;
  const int k = (int32_t)tmp >> 23;
  const double invc = tabs.log2(2 * i), logc = tabs.log2(2 * i + 1);
  const double z = (double)asfloat(iz);
  const double r = fma(z, invc, -1.0);
  const double y0 = logc + (double)k;
  const double r2 = r * r;
  double y = fma(k_log2f_poly[1], r, k_log2f_poly[2]);
  y = fma(k_log2f_poly[0], r2, y);
  const double p = fma(k_log2f_poly[3], r, y0);
  y = fma(y, r2, p);
  return (float)y;
}

// glibc's control flow folded for a SIMT machine: one predicate separates the positive normal
// arguments (straight-line code, the only path a wave of image data ever takes) from everything
// glibc special-cases (zero, negative, inf, NaN, subnormal), which share one cold block.
template <class Tabs = ANSEL_MATH_DEFAULT_TABS> ANSEL_HD float log2f_exact(const float x, const Tabs tabs = Tabs())
{
  uint32_t ix = asuint(x);
  if(ix - 0x00800000u >= 0x7f800000u - 0x00800000u)
  {
    if(ix * 2 == 0) return -INFINITY;                              // log2(+-0) = -inf
    if(ix == 0x7f800000u) return x;                                // log2(inf) = inf
    if((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return NAN;    // x < 0 or NaN
    ix = asuint(x * 0x1p23f);                                      // subnormal: normalise
    ix -= 23u << 23;
  }
  const float r = log2f_core(ix, tabs);
  return ix == 0x3f800000u ? 0.0f : r;                             // log2(1) is exactly +0
}

// ---- logf: sysdeps/ieee754/flt-32/e_logf.c (the FMA build the x86-64 ifunc selects) ----------------------
// Caller: the Box-Muller noise that seeds diffuse-or-sharpen's inpainting (src/iop/noise_generator.h:81-93).
ANSEL_HD float logf_exact(const float x)
{
  uint32_t ix = asuint(x);
  if(ix == 0x3f800000u) return 0.0f;
  if(ix - 0x00800000u >= 0x7f800000u - 0x00800000u)
  {
    if(ix * 2 == 0) return -INFINITY;
    if(ix == 0x7f800000u) return x;
    if((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return NAN;
    ix = asuint(x * 0x1p23f);
    ix -= 23u << 23;
  }
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (tmp >> (23 - 4)) % 16;
  const int k = (int32_t)tmp >> 23;
  const uint32_t iz = ix - (tmp & 0xff800000u);
  const double invc = k_logf_tab[2 * i], logc = k_logf_tab[2 * i + 1];
  const double z = (double)asfloat(iz);
  const double r = fma(z, invc, -1.0);
  const double y0 = fma((double)k, k_logf_ln2, logc);
  const double r2 = r * r;
  double y = fma(k_logf_poly[1], r, k_logf_poly[2]);
  y = fma(k_logf_poly[0], r2, y);
  y = fma(y, r2, y0 + r);
  return (float)y;
}

// ---- exp2 of a double argument, rounded to float: exp2_inline() of e_powf.c -------------
template <class Tabs = ANSEL_MATH_DEFAULT_TABS> ANSEL_HD float exp2_from_double(const double xd, const uint32_t sign_bias, const Tabs tabs = Tabs())
{
  double kd = xd + k_exp2f_shift_scaled;
  const uint64_t ki = asuint64(kd);
  kd -= k_exp2f_shift_scaled;
  const double r = xd - kd;
  uint64_t t = tabs.exp2((int)(ki % 32));
  const uint64_t ski = ki + sign_bias;
  t += ski << (52 - 5);
  const double s = asdouble(t);
  const double z = fma(k_exp2f_poly[0], r, k_exp2f_poly[1]);
  const double r2 = r * r;
  double y = fma(k_exp2f_poly[2], r, 1.0);
  y = fma(z, r2, y);
  y = y * s;
  return (float)y;
}

// checkint() of e_powf.c: 0 = not an integer, 1 = odd integer, 2 = even integer
ANSEL_HD int powf_checkint(const uint32_t iy)
{
  const int e = iy >> 23 & 0xff;
  if(e < 0x7f) return 0;
  if(e > 0x7f + 23) return 2;
  if(iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
  if(iy & (1u << (0x7f + 23 - e))) return 1;
  return 2;
}

ANSEL_HD bool powf_zeroinfnan(const uint32_t ix) { return 2 * ix - 1 >= 2u * 0x7f800000u - 1; }

// ---- powf: sysdeps/ieee754/flt-32/e_powf.c ------------------------------------------------
// log2_inline() + the range checks + exp2_inline() for |x| given as a positive normal (or
// pre-normalised) bit pattern ix: straight-line code, overflow / underflow handled by selects.
template <class Tabs = ANSEL_MATH_DEFAULT_TABS> ANSEL_HD float powf_core(const uint32_t ix, const float y, const uint32_t sign_bias, const Tabs tabs = Tabs())
{
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (tmp >> (23 - 4)) % 16;
  const uint32_t top = tmp & 0xff800000u;
  const uint32_t iz = ix - top;
  const int k = (int32_t)top >> 23;
  const double invc = tabs.powlog(2 * i), logc = tabs.powlog(2 * i + 1);
  const double z = (double)asfloat(iz);
  const double r = fma(z, invc, -1.0);
  const double y0 = logc + (double)k;
  const double r2 = r * r;
  double yy = fma(k_powf_log2_poly[0], r, k_powf_log2_poly[1]);
  const double p = fma(k_powf_log2_poly[2], r, k_powf_log2_poly[3]);
  const double r4 = r2 * r2;
  double q = fma(k_powf_log2_poly[4], r, y0);
  q = fma(p, r2, q);
  yy = fma(yy, r4, q);
  const double ylogx = (double)y * yy;
  const float res = exp2_from_double(ylogx, sign_bias, tabs);
  // |y * log2(x)| >= 126: overflow above 0x1.fffffffd1d571p+6, underflow at or below -150
  const bool big = (asuint64(ylogx) >> 47 & 0xffff) >= asuint64(126.0) >> 47;
  const bool of = big && ylogx > 0x1.fffffffd1d571p+6;
  const bool uf = big && ylogx <= -150.0;
  const float inf_s = sign_bias ? -INFINITY : INFINITY;
  const float zero_s = sign_bias ? -0.0f : 0.0f;
  return of ? inf_s : (uf ? zero_s : res);
}

// everything glibc special-cases: x zero / negative / inf / NaN / subnormal, y zero / inf / NaN
template <class Tabs = ANSEL_MATH_DEFAULT_TABS> ANSEL_HD float powf_special(const float x, const float y, const Tabs tabs = Tabs())
{
  uint32_t sign_bias = 0;
  uint32_t ix = asuint(x);
  const uint32_t iy = asuint(y);
  if(powf_zeroinfnan(iy))
  {
    if(2 * iy == 0) return 1.0f;
    if(ix == 0x3f800000u) return 1.0f;
    if(2 * ix > 2u * 0x7f800000u || 2 * iy > 2u * 0x7f800000u) return x + y;
    if(2 * ix == 2 * 0x3f800000u) return 1.0f;
    if((2 * ix < 2 * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;
    return y * y;
  }
  if(powf_zeroinfnan(ix))
  {
    float x2 = x * x;
    if((ix & 0x80000000u) && powf_checkint(iy) == 1)
    {
      x2 = -x2;
      sign_bias = 1;
    }
    if(2 * ix == 0 && (iy & 0x80000000u)) return sign_bias ? -INFINITY : INFINITY;
    return (iy & 0x80000000u) ? 1 / x2 : x2;
  }
  if(ix & 0x80000000u)
  {
    const int yint = powf_checkint(iy);
    if(yint == 0) return NAN;
    if(yint == 1) sign_bias = 1u << (5 + 11);
    ix &= 0x7fffffffu;
  }
  if(ix < 0x00800000u)
  {
    ix = asuint(x * 0x1p23f);
    ix &= 0x7fffffffu;
    ix -= 23u << 23;
  }
  return powf_core(ix, y, sign_bias, tabs);
}

template <class Tabs = ANSEL_MATH_DEFAULT_TABS> ANSEL_HD float powf_exact(const float x, const float y, const Tabs tabs = Tabs())
{
  const uint32_t ix = asuint(x);
  const uint32_t iy = asuint(y);
  // glibc's powf(x, 1.0f) returns x for every one of the 2^32 bit patterns of x (checked
  // exhaustively against this image's libm; NaNs stay NaN): a uniform-exponent shortcut for the
  // modules whose default exponent is 1 (color calibration gamut compression)
  if(iy == 0x3f800000u && x == x) return x;
  if(ix - 0x00800000u >= 0x7f800000u - 0x00800000u || powf_zeroinfnan(iy)) return powf_special(x, y, tabs);
  return powf_core(ix, y, 0, tabs);
}

// ---- exp2f: sysdeps/ieee754/flt-32/e_exp2f.c ----------------------------------------------
template <class Tabs = ANSEL_MATH_DEFAULT_TABS> ANSEL_HD float exp2f_exact(const float x, const Tabs tabs = Tabs())
{
  const double xd = (double)x;
  const uint32_t abstop = (asuint(x) >> 20) & 0x7ff;
  if(abstop >= (asuint(128.0f) >> 20))
  {
    if(asuint(x) == asuint(-INFINITY)) return 0.0f;
    if(abstop >= (asuint(INFINITY) >> 20)) return x + x;
    if(x > 0.0f) return INFINITY;
    if(x <= -150.0f) return 0.0f;
  }
  double kd = xd + k_exp2f_shift_scaled;
  const uint64_t ki = asuint64(kd);
  kd -= k_exp2f_shift_scaled;
  const double r = xd - kd;
  uint64_t t = tabs.exp2((int)(ki % 32));
  t += ki << (52 - 5);
  const double s = asdouble(t);
  const double z = fma(k_exp2f_poly[0], r, k_exp2f_poly[1]);
  const double r2 = r * r;
  double y = fma(k_exp2f_poly[2], r, 1.0);
  y = fma(z, r2, y);
  y = y * s;
  return (float)y;
}

// ---- expf: sysdeps/ieee754/flt-32/e_expf.c --------------------------------------------------
template <class Tabs = ANSEL_MATH_DEFAULT_TABS> ANSEL_HD float expf_exact(const float x, const Tabs tabs = Tabs())
{
  const double xd = (double)x;
  const uint32_t abstop = (asuint(x) >> 20) & 0x7ff;
  if(abstop >= (asuint(88.0f) >> 20))
  {
    if(asuint(x) == asuint(-INFINITY)) return 0.0f;
    if(abstop >= (asuint(INFINITY) >> 20)) return x + x;
    if(x > 0x1.62e42ep6f) return INFINITY;
    if(x < -0x1.9fe368p6f) return 0.0f;
  }
  const double z = k_exp2f_invln2_scaled * xd;
  double kd = z + k_exp2f_shift;
  const uint64_t ki = asuint64(kd);
  kd -= k_exp2f_shift;
  const double r = z - kd;
  uint64_t t = tabs.exp2((int)(ki % 32));
  t += ki << (52 - 5);
  const double s = asdouble(t);
  const double zz = fma(k_exp2f_poly_scaled[0], r, k_exp2f_poly_scaled[1]);
  const double r2 = r * r;
  double y = fma(k_exp2f_poly_scaled[2], r, 1.0);
  y = fma(zz, r2, y);
  y = y * s;
  return (float)y;
}

// ---- atanf / atan2f / hypotf ----------------------------------------------------------------
// glibc 2.35 has no multiarch build of these three on x86-64: atanf and atan2f are the fdlibm
// single-precision routines (sysdeps/ieee754/flt-32/s_atanf.c, e_atan2f.c) compiled without
// contraction, hypotf is one double-precision square root (sysdeps/ieee754/flt-32/e_hypotf.c).
// Callers: the JzCzhz / LCh hue and chroma of the parametric blend masks
// (src/common/colorspaces_inline_conversions.h:775-781, src/develop/blends/*.c).
ANSEL_HD float atanf_exact(float x)
{
  const float atanhi[4] = { 4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f };
  const float atanlo[4] = { 5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f };
  const float aT[11] = { 3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
                         9.0908870101e-02f, -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f,
                         4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f };
  const int32_t hx = (int32_t)asuint(x);
  const int32_t ix = hx & 0x7fffffff;
  int id;
  if(ix >= 0x4c000000) // |x| >= 2^25
  {
    if(ix > 0x7f800000) return x + x; // NaN
    return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
  }
  if(ix < 0x3ee00000) // |x| < 0.4375
  {
    if(ix < 0x31000000) return x; // |x| < 2^-29
    id = -1;
  }
  else
  {
    x = fabsf(x);
    if(ix < 0x3f980000) // |x| < 1.1875
    {
      if(ix < 0x3f300000) // 7/16 <= |x| < 11/16
      {
        id = 0;
        x = (2.0f * x - 1.0f) / (2.0f + x);
      }
      else // 11/16 <= |x| < 19/16
      {
        id = 1;
        x = (x - 1.0f) / (x + 1.0f);
      }
    }
    else
    {
      if(ix < 0x401c0000) // |x| < 2.4375
      {
        id = 2;
        x = (x - 1.5f) / (1.0f + 1.5f * x);
      }
      else // 2.4375 <= |x| < 2^25
      {
        id = 3;
        x = -1.0f / x;
      }
    }
  }
  const float z = x * x;
  const float w = z * z;
  const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if(id < 0) return x - x * (s1 + s2);
  const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
  return hx < 0 ? -r : r;
}

ANSEL_HD float atan2f_exact(const float y, const float x)
{
  const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
              pi_lo = -8.7422776573e-08f;
  const int32_t hx = (int32_t)asuint(x), hy = (int32_t)asuint(y);
  const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if(ix > 0x7f800000 || iy > 0x7f800000) return x + y; // NaN
  if(hx == 0x3f800000) return atanf_exact(y);           // x = 1
  const int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2); // 2 * sign(x) + sign(y)
  if(iy == 0)
  {
    switch(m)
    {
      case 0:
      case 1: return y;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  if(ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if(ix == 0x7f800000)
  {
    if(iy == 0x7f800000)
    {
      switch(m)
      {
        case 0: return pi_o_4 + tiny;
        case 1: return -pi_o_4 - tiny;
        case 2: return 3.0f * pi_o_4 + tiny;
        default: return -3.0f * pi_o_4 - tiny;
      }
    }
    switch(m)
    {
      case 0: return 0.0f;
      case 1: return -0.0f;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  if(iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  const int32_t k = (iy - ix) >> 23;
  float z;
  if(k > 60) z = pi_o_2 + 0.5f * pi_lo;     // |y / x| > 2^60
  else if(hx < 0 && k < -60) z = 0.0f;      // |y| / x < -2^60
  else z = atanf_exact(fabsf(y / x));
  switch(m)
  {
    case 0: return z;
    case 1: return asfloat(asuint(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

ANSEL_HD float hypotf_exact(const float x, const float y)
{
  const uint32_t ax = asuint(x) & 0x7fffffffu, ay = asuint(y) & 0x7fffffffu;
  if(ax >= 0x7f800000u || ay >= 0x7f800000u)
  {
    // an infinity wins over a quiet NaN, a signalling NaN over everything
    const bool sx = ax > 0x7f800000u && ax < 0x7fc00000u, sy = ay > 0x7f800000u && ay < 0x7fc00000u;
    if((ax == 0x7f800000u || ay == 0x7f800000u) && !sx && !sy) return INFINITY;
    return x + y;
  }
  const double dx = (double)x, dy = (double)y;
  return (float)sqrt(dx * dx + dy * dy);
}

// ---- sinf / cosf: sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h --------------------------
// (ARM optimized-routines: binary64 polynomials after a reduction by pi/2).  On x86-64 both are ifuncs and an
// FMA-capable CPU runs the *_fma build, in which gcc contracts every a + b * c of the source: the fma() calls
// below.  Callers: dt_LCH_2_Lab() of the Lab blend operators hue / color / chroma / coloradjust
// (src/common/colorspaces_inline_conversions.h:608-620, src/develop/blends/blendif_lab.c:843-975).
ANSEL_HD uint32_t sincosf_abstop12(const float x) { return (asuint(x) >> 20) & 0x7ff; }

// sinf_poly(): n even -> sine polynomial, odd -> cosine; `alt` selects __sincosf_table[1] (cosine negated)
ANSEL_HD float sincosf_poly(const double x, const double x2, const bool alt, const int n)
{
  const double sg = alt ? -1.0 : 1.0;
  const double c0 = sg * 0x1p0, c1 = sg * -0x1.ffffffd0c621cp-2, c2 = sg * 0x1.55553e1068f19p-5,
               c3 = sg * -0x1.6c087e89a359dp-10, c4 = sg * 0x1.99343027bf8c3p-16;
  const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
  if((n & 1) == 0)
  {
    const double x3 = x * x2;
    const double t1 = fma(x2, s3, s2);
    const double x7 = x3 * x2;
    const double t = fma(x3, s1, x);
    return (float)fma(x7, t1, t);
  }
  const double x4 = x2 * x2;
  const double u2 = fma(x2, c4, c3);
  const double u1 = fma(x2, c1, c0);
  const double x6 = x4 * x2;
  const double u = fma(x4, c2, u1);
  return (float)fma(x6, u2, u);
}

// reduce_fast(): |x| < 120
ANSEL_HD double sincosf_reduce_fast(const double x, int *np)
{
  const double r = x * 0x1.45f306dc9c883p+23; // 2/pi * 2^24
  const int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return fma(-(double)n, 0x1.921fb54442d18p+0, x);
}

// reduce_large(): 120 <= |x| < inf, 192 bits of 4/pi
ANSEL_HD double sincosf_reduce_large(uint32_t xi, int *np)
{
  const uint32_t inv_pio4[24] = { 0xa2,       0xa2f9,     0xa2f983,   0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
                                  0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0,
                                  0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041 };
  const int idx = (xi >> 26) & 15;
  const int shift = (xi >> 23) & 7;
  xi = (xi & 0xffffff) | 0x800000;
  xi <<= shift;
  uint64_t res0 = (uint32_t)(xi * inv_pio4[idx]);
  const uint64_t res1 = (uint64_t)xi * inv_pio4[idx + 4];
  const uint64_t res2 = (uint64_t)xi * inv_pio4[idx + 8];
  res0 = (res2 >> 32) | (res0 << 32);
  res0 += res1;
  const uint64_t n = (res0 + (1ULL << 61)) >> 62;
  res0 -= n << 62;
  *np = (int)n;
  return (double)(int64_t)res0 * 0x1.921fb54442d18p-62;
}

template <int COS> ANSEL_HD float sincosf_exact(const float y)
{
  const double sign[4] = { 1.0, -1.0, -1.0, 1.0 };
  double x = (double)y;
  int n;
  if(sincosf_abstop12(y) < sincosf_abstop12(0x1.921fb6p-1f))
  {
    if(sincosf_abstop12(y) < sincosf_abstop12(0x1p-12f)) return COS ? 1.0f : y;
    return sincosf_poly(x, x * x, false, COS);
  }
  if(sincosf_abstop12(y) < sincosf_abstop12(120.0f))
  {
    x = sincosf_reduce_fast(x, &n);
    const double sg = sign[n & 3];
    return sincosf_poly(x * sg, x * x, (n & 2) != 0, n ^ COS);
  }
  if(sincosf_abstop12(y) < sincosf_abstop12(INFINITY))
  {
    const uint32_t xi = asuint(y);
    const int sgn = (int)(xi >> 31);
    x = sincosf_reduce_large(xi, &n);
    const double sg = sign[(n + sgn) & 3];
    return sincosf_poly(x * sg, x * x, ((n + sgn) & 2) != 0, n ^ COS);
  }
  return (y - y) / (y - y); // __math_invalidf(): NaN
}

ANSEL_HD float sinf_exact(const float x) { return sincosf_exact<0>(x); }
ANSEL_HD float cosf_exact(const float x) { return sincosf_exact<1>(x); }

} // namespace ansel_math
