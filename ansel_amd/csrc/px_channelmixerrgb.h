// px_channelmixerrgb.h -- per-pixel body of color calibration, shared by the standalone kernel
// (channelmixerrgb.hip) and the fused pointwise chain (pipe_fused.hip).  Reference citations in
// channelmixerrgb.hip.
#pragma once
#include "hip_common.h"
#include "devmath.h"

namespace ansel
{

#define NORM_MIN 1.52587890625e-05f
#define INVERSE_SQRT_3 0.5773502691896258f

struct f3
{
  float x, y, z;
};

struct cm_args
{
  float XYZ_to_RGB[3][3], RGB_to_XYZ[3][3], MIX[3][3];
  float illuminant[3], saturation[3], lightness[3], grey[3];
  float p, gamut;
  int apply_grey, version;
  int kind, clip; // used by the fused chain only
};

__device__ __forceinline__ f3 mat3(const float m[3][3], const f3 v)
{
  f3 o;
  o.x = m[0][0] * v.x;
  o.y = m[1][0] * v.x;
  o.z = m[2][0] * v.x;
  o.x = m[0][1] * v.y + o.x;
  o.y = m[1][1] * v.y + o.y;
  o.z = m[2][1] * v.y + o.z;
  o.x = m[0][2] * v.z + o.x;
  o.y = m[1][2] * v.z + o.y;
  o.z = m[2][2] * v.z + o.z;
  return o;
}

// src/pixel/chromatic_adaptation.h:45-95
__device__ const float k_XYZ_to_Bradford[3][3] = { { 0.8951f, 0.2664f, -0.1614f }, { -0.7502f, 1.7135f, 0.0367f }, { 0.0389f, -0.0685f, 1.0296f } };
__device__ const float k_Bradford_to_XYZ[3][3] = { { 0.9870f, -0.1471f, 0.1600f }, { 0.4323f, 0.5184f, 0.0493f }, { -0.0085f, 0.0400f, 0.9685f } };
__device__ const float k_XYZ_to_CAT16[3][3] = { { 0.401288f, 0.650173f, -0.051461f }, { -0.250268f, 1.204414f, 0.045854f }, { -0.002079f, 0.048952f, 0.953127f } };
__device__ const float k_CAT16_to_XYZ[3][3] = { { 1.862068f, -1.011255f, 0.149187f }, { 0.38752f, 0.621447f, -0.008974f }, { -0.015841f, -0.034123f, 1.049964f } };

__device__ __forceinline__ float max_zero1(const float v) { return isfinite(v) ? (v > 0.0f ? v : 0.0f) : 0.f; }
__device__ __forceinline__ f3 max_zero(const f3 v) { return { max_zero1(v.x), max_zero1(v.y), max_zero1(v.z) }; }

__device__ __forceinline__ float scale_of(const float scaling)
{
  const bool valid = (scaling > NORM_MIN) && !isnan(scaling);
  return valid ? (scaling + NORM_MIN) : NORM_MIN;
}

template <int KIND> __device__ __forceinline__ f3 xyz_to_lms(const f3 v)
{
  if(KIND == DT_HIP_ADAPTATION_FULL_BRADFORD || KIND == DT_HIP_ADAPTATION_LINEAR_BRADFORD) return mat3(k_XYZ_to_Bradford, v);
  if(KIND == DT_HIP_ADAPTATION_CAT16) return mat3(k_XYZ_to_CAT16, v);
  return v;
}
template <int KIND> __device__ __forceinline__ f3 lms_to_xyz(const f3 v)
{
  if(KIND == DT_HIP_ADAPTATION_FULL_BRADFORD || KIND == DT_HIP_ADAPTATION_LINEAR_BRADFORD) return mat3(k_Bradford_to_XYZ, v);
  if(KIND == DT_HIP_ADAPTATION_CAT16) return mat3(k_CAT16_to_XYZ, v);
  return v;
}

__device__ __forceinline__ f3 gamut_mapping(const f3 input, const float compression, const bool CLIP)
{
  const float sum = input.x + input.y + input.z;
  const float Y = input.y;
  if(!(sum > 0.f && Y > 0.f)) return { 0.f, 0.f, 0.f };
  float x = input.x / sum;
  float y = input.y / sum;
  const float to_uv = -2.f * x + 12.f * y + 3.f;
  float u = 4.f * x / to_uv, v = 9.f * y / to_uv; // CIE 1976 u'v' of the chromaticity
  const float D50u = 0.20915914598542354f, D50v = 0.488075320769787f;
  const float du = D50u - u, dv = D50v - v;
  const float Delta = Y * (du * du + dv * dv);
  const float correction = (compression == 0.0f) ? 0.f : ansel_math::powf_exact(Delta, compression);
  const float tmp_u = __builtin_fmaf(correction, du, u);
  const float tmp_v = __builtin_fmaf(correction, dv, v);
  u = (u > D50u) ? fmaxf(tmp_u, D50u) : fminf(tmp_u, D50u);
  v = (v > D50v) ? fmaxf(tmp_v, D50v) : fminf(tmp_v, D50v);
  const float to_xy = 6.f * u - 16.f * v + 12.f;
  x = 9.f * u / to_xy, y = 4.f * v / to_xy; // and back
  if(CLIP)
  {
    x = fmaxf(x, 0.0f), y = fmaxf(y, 0.0f);
  }
  y = fmaxf(y, NORM_MIN);
  const float xy_sum = x + y;
  if(xy_sum >= 1.f)
  {
    x /= xy_sum;
    y /= xy_sum;
  }
  return { Y * x / y, Y, Y * (1.f - x - y) / y };
}

__device__ __forceinline__ float enorm(const f3 v) { return fmaxf(sqrtf(v.x * v.x + v.y * v.y + v.z * v.z), NORM_MIN); }

__device__ __forceinline__ f3 luma_chroma(const f3 in, const float sat[3], const float light[3], const int version)
{
  float norm = enorm(in);
  const float avg = fmaxf((in.x + in.y + in.z) / 3.0f, NORM_MIN);
  if(!(norm > 0.f && avg > 0.f)) return in;
  float mix = 0.f;
  mix += in.x * light[0];
  mix += in.y * light[1];
  mix += in.z * light[2];
  if(version == 2) norm *= INVERSE_SQRT_3;
  float o[3] = { in.x / norm, in.y / norm, in.z / norm };
  float desat = 0.f;
  if(version == 0)
  {
#pragma unroll
    for(int c = 0; c < 3; c++) desat += (1.0f - o[c]) * (1.0f - o[c]) * sat[c];
  }
  else
  {
    float sp = 0.f;
#pragma unroll
    for(int c = 0; c < 3; c++) sp += o[c] * sat[c];
    desat = sp / 3.f;
  }
#pragma unroll
  for(int c = 0; c < 3; c++)
  {
    const float floor_c = (o[c] < 0.0f) ? o[c] : 0.0f;
    const float headroom = 1.0f - o[c];
    o[c] = fmaxf(__builtin_fmaf(headroom, desat, o[c]), floor_c);
  }
  if(version == 2) norm /= enorm({ o[0], o[1], o[2] }) * INVERSE_SQRT_3;
  norm *= fmaxf(mix / avg + 1.f, 0.f);
  return { o[0] * norm, o[1] * norm, o[2] * norm };
}


// KIND is a template parameter (it selects matrices and the adaptation formula); CLIP is a
// compile-time constant at the standalone kernel's call site and wave-uniform in the fused chain
template <int KIND> __device__ __forceinline__ float4 px_channelmixerrgb(const float4 p, const cm_args &a, const bool CLIP)
{
  f3 one, two = { p.x, p.y, p.z };
  if(CLIP) two = max_zero(two);
  if(KIND == DT_HIP_ADAPTATION_FULL_BRADFORD || KIND == DT_HIP_ADAPTATION_LINEAR_BRADFORD)
  {
    one = mat3(a.RGB_to_XYZ, two);
    const float Y = one.y;
    const float s = scale_of(Y);
    two = mat3(k_XYZ_to_Bradford, one);
    two = { two.x / s, two.y / s, two.z / s };
    // bradford_adapt_D50(), chromatic_adaptation.h:209-217
    f3 t = { two.x / a.illuminant[0], two.y / a.illuminant[1], two.z / a.illuminant[2] };
    if(KIND == DT_HIP_ADAPTATION_FULL_BRADFORD) t.z = (t.z > 0.f) ? ansel_math::powf_exact(t.z, a.p) : t.z;
    t = { 0.996078f * t.x, 1.020646f * t.y, 0.818155f * t.z };
    one = { t.x * s, t.y * s, t.z * s };
    two = mat3(a.MIX, one);
    one = mat3(k_Bradford_to_XYZ, two);
  }
  else if(KIND == DT_HIP_ADAPTATION_CAT16)
  {
    one = mat3(a.RGB_to_XYZ, two);
    const float Y = one.y;
    const float s = scale_of(Y);
    two = mat3(k_XYZ_to_CAT16, one);
    two = { two.x / s, two.y / s, two.z / s };
    // CAT16_adapt_D50(lms, illuminant, 1.0f, TRUE) = lms * D50 / illuminant
    const f3 t = { two.x * 0.994535f / a.illuminant[0], two.y * 1.000997f / a.illuminant[1], two.z * 0.833036f / a.illuminant[2] };
    one = { t.x * s, t.y * s, t.z * s };
    two = mat3(a.MIX, one);
    one = mat3(k_CAT16_to_XYZ, two);
  }
  else if(KIND == DT_HIP_ADAPTATION_XYZ)
  {
    one = mat3(a.RGB_to_XYZ, two);
    const float Y = one.y;
    const float s = scale_of(Y);
    f3 t = { one.x / s, one.y / s, one.z / s };
    t = { t.x * 0.9642119944211994f / a.illuminant[0], t.y * 1.0f / a.illuminant[1], t.z * 0.8251882845188288f / a.illuminant[2] };
    two = { t.x * s, t.y * s, t.z * s };
    one = mat3(a.MIX, two);
  }
  else
  {
    one = mat3(a.MIX, two);
    one = mat3(a.RGB_to_XYZ, one);
  }

  two = gamut_mapping(one, a.gamut, CLIP);
  one = (KIND == DT_HIP_ADAPTATION_RGB) ? mat3(a.XYZ_to_RGB, two) : xyz_to_lms<KIND>(two);
  if(CLIP) one = max_zero(one);
  two = luma_chroma(one, a.saturation, a.lightness, a.version);
  if(CLIP) two = max_zero(two);

  float4 o;
  if(a.apply_grey)
  {
    const float grey_mix = fmaxf(two.x * a.grey[0] + two.y * a.grey[1] + two.z * a.grey[2], 0.0f);
    o = make_float4(grey_mix, grey_mix, grey_mix, p.w);
  }
  else
  {
    one = (KIND == DT_HIP_ADAPTATION_RGB) ? mat3(a.RGB_to_XYZ, two) : lms_to_xyz<KIND>(two);
    if(CLIP) one = max_zero(one);
    two = mat3(a.XYZ_to_RGB, one);
    if(CLIP) two = max_zero(two);
    o = make_float4(two.x, two.y, two.z, p.w);
  }
  return o;
}

// host: dt_hip_channelmixerrgb_data_t -> kernel arguments
void channelmixerrgb_fill_args(const dt_hip_channelmixerrgb_data_t *d, cm_args &a);

} // namespace ansel
