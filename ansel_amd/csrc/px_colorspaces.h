// px_colorspaces.h -- per-pixel body of colorin / colorout (matrix path), shared by the standalone
// kernels (colorspaces.hip) and the fused pointwise chain (pipe_fused.hip).  See colorspaces.hip for
// the reference citations.
#pragma once
#include "hip_common.h"
#include "devmath.h"

namespace ansel
{

struct conv_args
{
  float m[3][4];  // rows of the source->target (or source->clip) matrix
  float cm[3][4]; // rows of the clip->target matrix
  float cs[3][3]; // source power-law fits {a, b, c} per channel
  float ct[3][3]; // target
  const float *ls[3];
  const float *lt[3];
  int decode[3], encode[3];
  int clipping, blue_mapping;
};

__device__ __forceinline__ float extrapolate_lut(const float *__restrict__ lut, const float v)
{
  // iop_profile.h:537-547, lutsize = 0x10000
  const float a = v * 65535.0f;
  const float ft = a > 0.0f ? (a < 65535.0f ? a : 65535.0f) : 0.0f; // CLAMPS(v * (lutsize - 1), 0, lutsize - 1)
  const int t = (int)((ft < 65534.0f) ? ft : 65534.0f);
  const float f = ft - (float)t;
  // the two adjacent samples in ONE 8-byte gather (dword-aligned): the tone-curve stage is bound by
  // the number of L1/L2 requests, one per lane per load instruction
  typedef float f2u_t __attribute__((ext_vector_type(2), aligned(4)));
  const f2u_t l = *reinterpret_cast<const f2u_t *>(lut + t);
  return l.x * (1.0f - f) + l.y * f;
}

__device__ __forceinline__ float eval_trc(const float x, const float *__restrict__ lut, const float *coeff)
{
  return (x < 1.0f) ? extrapolate_lut(lut, x) : coeff[1] * ansel_math::powf_exact(x * coeff[0], coeff[2]);
}

// dt_mat3x4_mul_vec4(), src/system/simd.h:188-197, on all four lanes (the 4th matrix column is 0,
// so lane 3 is 0*x + 0*y + 0*z: a signed zero, or NaN for a non-finite input -- kept as is)
__device__ __forceinline__ float4 mat3x4(const float x, const float y, const float z, const float m[3][4])
{
  float4 o;
  o.x = m[0][0] * x;
  o.y = m[1][0] * x;
  o.z = m[2][0] * x;
  o.w = 0.0f * x;
  o.x = m[0][1] * y + o.x;
  o.y = m[1][1] * y + o.y;
  o.z = m[2][1] * y + o.z;
  o.w = 0.0f * y + o.w;
  o.x = m[0][2] * z + o.x;
  o.y = m[1][2] * z + o.y;
  o.z = m[2][2] * z + o.z;
  o.w = 0.0f * z + o.w;
  return o;
}

// ---- the RGB <-> Lab glue of the pipe (pixelpipe_cpu.c:59-75), per pixel: colorspaces.hip and the fused chain ----
// lab_f() with cbrt_5f() + cbrta_halleyf(), src/common/colorspaces_inline_conversions.h:50-73
__device__ __forceinline__ float lab_f(const float x)
{
  const float epsilon = 216.0f / 24389.0f, kappa = 24389.0f / 27.0f;
  if(!(x > epsilon)) return (kappa * x + 16.0f) / 116.0f;
  const float a = __uint_as_float(__float_as_uint(x) / 3u + 709921077u);
  const float a3 = a * a * a;
  return a * (a3 + x + x) / (a3 + a3 + x);
}
// lab_f_inv(), :88-94
__device__ __forceinline__ float lab_f_inv(const float x)
{
  const float epsilon = 0.20689655172413796f, kappa = 24389.0f / 27.0f;
  return (x > epsilon) ? x * x * x : (116.0f * x - 16.0f) / kappa;
}
// _transform_rgb_to_lab_matrix(), src/colorprofiles/iop_profile.c:405-418 + dt_XYZ_to_Lab()
__device__ __forceinline__ float4 px_rgb_to_lab(const float4 p, const float m[3][4])
{
  const float4 xyz = mat3x4(p.x, p.y, p.z, m);
  const float f0 = lab_f(xyz.x / 0.9642f), f1 = lab_f(xyz.y / 1.0f), f2 = lab_f(xyz.z / 0.8249f);
  return make_float4(116.0f * f1 - 16.0f, 500.0f * (f0 - f1), 200.0f * (f1 - f2), p.w);
}
// _transform_lab_to_rgb_matrix(), :423-450 + dt_Lab_to_XYZ()
__device__ __forceinline__ float4 px_lab_to_rgb(const float4 p, const float m[3][4])
{
  const float fy = (p.x + 16.0f) / 116.0f;
  const float fx = p.y / 500.0f + fy;
  const float fz = fy - p.z / 200.0f;
  const float4 rgb = mat3x4(0.9642f * lab_f_inv(fx), 1.0f * lab_f_inv(fy), 0.8249f * lab_f_inv(fz), m);
  return make_float4(rgb.x, rgb.y, rgb.z, p.w);
}

__device__ __forceinline__ float clamp01(const float v)
{
  // CLAMP(v, 0.0f, 1.0f) of glib
  return v > 1.0f ? 1.0f : (v < 0.0f ? 0.0f : v);
}


// the whole conversion of one pixel; lane 3 of the result is the 0*x + 0*y + 0*z the reference's
// 4-lane product leaves there
// (the four switches are compile-time constants at the standalone kernels' call sites and
// wave-uniform run-time values in the fused chain; forceinline folds the former)
__device__ __forceinline__ float4 px_conversion(const float4 p, const conv_args &a, const bool DECODE,
                                                const bool ENCODE, const bool CLIP, const bool HOOK)
{
  float s0 = p.x, s1 = p.y, s2 = p.z;
  if(DECODE)
  {
    if(a.decode[0]) s0 = eval_trc(s0, a.ls[0], a.cs[0]);
    if(a.decode[1]) s1 = eval_trc(s1, a.ls[1], a.cs[1]);
    if(a.decode[2]) s2 = eval_trc(s2, a.ls[2], a.cs[2]);
  }
  if(HOOK)
  {
    // apply_blue_mapping(), colorin.c:690-709
    const float YY = s0 + s1 + s2;
    if(YY > 0.0f)
    {
      const float zz = s2 / YY;
      const float bound_z = 0.5f, bound_Y = 0.5f, amount = 0.11f;
      if(zz > bound_z)
      {
        // fminf(1.0, YY / bound_Y): double literal, so fminf receives (float)1.0
        const float t = (zz - bound_z) / (1.0f - bound_z) * fminf(1.0f, YY / bound_Y);
        s1 += t * amount;
        s2 -= t * amount;
      }
    }
  }
  float4 v = mat3x4(s0, s1, s2, a.m);
  if(CLIP) v = mat3x4(clamp01(v.x), clamp01(v.y), clamp01(v.z), a.cm);
  if(ENCODE)
  {
    if(a.encode[0]) v.x = eval_trc(v.x, a.lt[0], a.ct[0]);
    if(a.encode[1]) v.y = eval_trc(v.y, a.lt[1], a.ct[1]);
    if(a.encode[2]) v.z = eval_trc(v.z, a.lt[2], a.ct[2]);
  }
  return v;
}

// run-time dispatch for the fused chain (all flags are wave-uniform)
__device__ __forceinline__ float4 px_conversion_rt(const float4 p, const conv_args &a)
{
  const bool decode = a.decode[0] | a.decode[1] | a.decode[2];
  const bool encode = a.encode[0] | a.encode[1] | a.encode[2];
  if(!decode && !a.clipping && !a.blue_mapping)
    return encode ? px_conversion(p, a, false, true, false, false) : px_conversion(p, a, false, false, false, false);
  return px_conversion(p, a, decode, encode, a.clipping != 0, a.blue_mapping != 0);
}

// host: dt_hip_conversion_t -> kernel arguments; returns the template key (decode<<2 | encode<<1 | clip)
int conversion_fill_args(const dt_hip_conversion_t *d, conv_args &a);

} // namespace ansel
