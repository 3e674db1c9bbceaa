// ieee_inrange.h -- correctly rounded binary32 division and square root WITHOUT the range scaffolding, for operands a
// kernel has shown to be in range (gfx950 device code only).
//
// hipcc expands `a / b` (no fast-math, denormals on) into eleven instructions:
//     v_div_scale_f32 d, b, b, a ; v_div_scale_f32 n, vcc, a, b, a ; v_rcp_f32 r, d ;
//     e = fma(-d, r, 1) ; r = fma(e, r, r) ; q = n * r ; e2 = fma(-d, q, n) ; q = fma(e2, r, q) ; e3 = fma(-d, q, n) ;
//     v_div_fmas_f32 q, e3, r, q ; v_div_fixup_f32 q, q, b, a
// and `sqrtf(x)` into sixteen (scale x by 2^32 below 2^-96, v_sqrt_f32, two one-ulp corrections by fma residuals, scale
// back, pass +-0 / +inf through).  The two v_div_scale, the scale bit of v_div_fmas and v_div_fixup only ACT when an
// operand or the quotient leaves the range in which the eight instructions between them round correctly (CDNA4 ISA,
// V_DIV_SCALE_F32 / V_DIV_FIXUP_F32):
//     d is scaled   when b is subnormal, |b| >= 2^126 (1 / b subnormal), or exponent(a) - exponent(b) >= 96;
//     n is scaled   when |a| < 2^-103 (the residuals a - b q would be subnormal), or a / b is subnormal;
//     fixup acts    when a or b is NaN, infinite or zero, or the quotient over- / underflows.
// Everywhere else v_div_scale returns its operand, VCC = 0 makes v_div_fmas a plain fma and v_div_fixup returns q: the
// eight instructions of div_core() below ARE the expansion, same operations on the same operands in the same order, so the
// same bits.  Likewise sqrt_core() is the expansion's path for x in {+0} U [2^-96, +inf): the nine instructions between
// the scaling and the class test (for +0 the two corrections see a NaN and a zero residual and leave +0).
//
// The callers (diffuse.hip) state, per division site, why the operands are in range or test it once per wave; the
// tests compare both cores with the host's IEEE division / square root over the whole in-range domain's edges and
// 10^7 random operands (tests/test_gpu_devmath.py::test_inrange_*), and the kernels that use them against the oracle.
#pragma once

namespace ansel_ieee
{

// a / b for: b normal, 2^-126 <= |b| < 2^126; a == +-0 (returns a zero whose SIGN is not the quotient's) or
// 2^-103 <= |a| and -125 <= exponent(a) - exponent(b) < 96; a NaN gives a NaN (any payload)
__device__ __forceinline__ float div_core(const float a, const float b)
{
  float r = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  float q = a * r;
  const float e2 = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(e2, r, q);
  const float e3 = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(e3, r, q);
}

// 1.0f / b for 2^-126 <= |b| < 2^126 (the product n * r of the expansion is 1.0f * r == r)
__device__ __forceinline__ float rcp_core(const float b)
{
  float r = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  const float e2 = __builtin_fmaf(-b, r, 1.0f);
  const float q = __builtin_fmaf(e2, r, r);
  const float e3 = __builtin_fmaf(-b, q, 1.0f);
  return __builtin_fmaf(e3, r, q);
}

// n / d for a divisor that many numerators share (wave-uniform): y1 = rcp_refined(d) once -- v_rcp_f32 and the expansion's Newton
// step --, then the expansion's five operations per numerator and v_div_fixup_f32 for the numerators they do not take (infinite: the
// quotient is an infinity, not the NaN of inf - inf; NaN; zero, with the quotient's sign).  Correctly rounded for d normal in
// [2^-126, 2^126) and |n| in [2^-103, 2^96 |d|) (div_core()'s domain); outside it a quotient may come out an ulp off -- the caller's
// business (nlm3_body.h CENTER: the weight behind such a quotient is 0 or 1 either way).
__device__ __forceinline__ float rcp_refined(const float d)
{
  const float y = __builtin_amdgcn_rcpf(d);
  return __builtin_fmaf(__builtin_fmaf(-d, y, 1.0f), y, y);
}
__device__ __forceinline__ float div_uniform(const float n, const float d, const float y1)
{
  const float q = n * y1;
  const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q, n), y1, q);
  const float q2 = __builtin_fmaf(__builtin_fmaf(-d, q1, n), y1, q1);
  return __builtin_amdgcn_div_fixupf(q2, d, n);
}

// sqrtf(x) for x == +0 or 2^-96 <= x < +inf
__device__ __forceinline__ float sqrt_core(const float x)
{
  const float s = __builtin_amdgcn_sqrtf(x);
  const float down = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
  const float r_down = __builtin_fmaf(-down, s, x);
  float t = (0.0f >= r_down) ? down : s;
  const float r_up = __builtin_fmaf(-up, s, x);
  t = (0.0f < r_up) ? up : t;
  return t;
}

// x == +0 or 2^-96 <= x < +inf -- sqrt_core()'s domain -- on the bit pattern: everything else (negative numbers and -0, NaNs,
// +inf, the subnormals and the normals below 2^-96) fails.  (A first version tested the class of x * 2^-30: one instruction
// fewer, and wrong -- a nonzero x below 2^-120 scales to zero and passed; tests/test_gpu_devmath.py caught it.)
__device__ __forceinline__ bool zero_or_above_2m96(const float x)
{
  const unsigned u = __float_as_uint(x);
  return u == 0u || (u - 0x0f800000u) <= (0x7f7fffffu - 0x0f800000u);
}

} // namespace ansel_ieee
