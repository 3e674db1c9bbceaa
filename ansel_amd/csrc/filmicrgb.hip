// filmicrgb.hip -- filmic RGB tone mapping on gfx950, highlight reconstruction bypassed (the
// default of every new edit: hl_deprecated, src/iop/filmicrgb.c:2733, :4104).
//
//   filmic_agx()        filmicrgb.c:2495-2587   colour science v8 "AgX" (versions 5..9; default 7)
//   filmic_v5()         filmicrgb.c:2247-2300   v7
//   filmic_chroma_v4()  filmicrgb.c:2153-2198   v6, norm-preserving
//   filmic_split_v4()   filmicrgb.c:2201-2244   v6, per channel
// with log_tonemapping :1047, filmic_spline :1063-1160, get_pixel_norm_simd :976-1035, the Ych /
// gamut-mapping helpers :1740-2030, filmic_v4_prepare_matrices :2033-2064, the AgX bracket
// :2344-2459 and filmic_agx_compress_negatives :2461-2492.
//
// One pointwise kernel per colour science, 16 B in + 16 B out per pixel (32 B/px algorithmic).
// It is the one ALU-heavy pointwise stage of the pipe: per pixel 3 log2f + 3..9 powf (all in
// binary64 inside devmath.h, to return glibc's bits), ~10 3x3 products, ~25 divisions and 3 sqrt.
// The reference measured 1.03 s on CPU and 0.22 s on its OpenCL path for 24 MP (filmicrgb.c:2684).
//
// The per-call matrix preparation the reference does at the top of each of these functions runs
// on the host here (filmic_prepare below), in the same binary32 operation order.
#include "hip_common.h"
#include "devmath.h"
#include <float.h>

using namespace ansel;

namespace
{

#define CIE_Y_1931_to_CIE_Y_2006(x) (1.05785528f * (x))
#define INVERSE_SQRT_3 0.5773502691896258f

struct m3
{
  float r[3][3];
};

struct fargs
{
  m3 input, output, export_input, export_output, inset, outset;
  float luma[3];
  float norm_min, norm_max;
  float display_black, display_white;
  float grey_source, black_source, dynamic_range, output_power, saturation, beta_hue;
  // spline
  float M1[3], M2[3], M3[3], M4[3], M5[3];
  float latitude_min, latitude_max, y0, y4;
  int type0, type1;
  int preserve_color;
};

struct v4
{
  float x, y, z, w;
};

// dt_mat3x4_mul_vec4 (src/system/simd.h:188-197); lane 3 carries 0*x + 0*y + 0*z
__device__ __forceinline__ v4 mat3(const m3 &m, const v4 v)
{
  v4 o;
  o.x = m.r[0][0] * v.x;
  o.y = m.r[1][0] * v.x;
  o.z = m.r[2][0] * v.x;
  o.w = 0.0f * v.x;
  o.x = m.r[0][1] * v.y + o.x;
  o.y = m.r[1][1] * v.y + o.y;
  o.z = m.r[2][1] * v.y + o.z;
  o.w = 0.0f * v.y + o.w;
  o.x = m.r[0][2] * v.z + o.x;
  o.y = m.r[1][2] * v.z + o.y;
  o.z = m.r[2][2] * v.z + o.z;
  o.w = 0.0f * v.z + o.w;
  return o;
}

__device__ __forceinline__ float min_(const float a, const float b) { return a < b ? a : b; }         // glib MIN
__device__ __forceinline__ float max_(const float a, const float b) { return a > b ? a : b; }         // glib MAX
__device__ __forceinline__ float clamp_glib(const float x, const float lo, const float hi) { return x > hi ? hi : (x < lo ? lo : x); }
__device__ __forceinline__ float clampf(const float a, const float mn, const float mx) { return a >= mn ? (a <= mx ? a : mx) : mn; }
__device__ __forceinline__ float clamp_simd(const float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

// ---- Yrg / Ych: src/common/colorspaces_inline_conversions.h:1033-1074 ----------------------
__device__ __forceinline__ v4 LMS_to_Yrg(const v4 LMS)
{
  const float Y = 0.68990272f * LMS.x + 0.34832189f * LMS.y;
  const float a = LMS.x + LMS.y + LMS.z;
  const float inv_a = (a == 0.f) ? 0.f : 1.f / a;
  const float l = LMS.x * inv_a, m = LMS.y * inv_a, s = LMS.z * inv_a;
  // LMS_to_gradingRGB_simd(): rows of LMS_D65_to_filmlightRGB_D65
  float r0 = 1.0877193f * l;
  float r1 = -0.0877193f * l;
  r0 = -0.66666667f * m + r0;
  r1 = 1.66666667f * m + r1;
  r0 = 0.02061856f * s + r0;
  r1 = -0.05154639f * s + r1;
  return { Y, r0, r1, 0.f };
}

__device__ __forceinline__ v4 Yrg_to_LMS(const float Y, const float r, const float g)
{
  const float b = 1.f - r - g;
  // gradingRGB_to_LMS_simd(): rows of filmlightRGB_D65_to_LMS_D65
  float l0 = 0.95f * r;
  float l1 = 0.05f * r;
  float l2 = 0.00f * r;
  l0 = 0.38f * g + l0;
  l1 = 0.62f * g + l1;
  l2 = 0.00f * g + l2;
  l0 = 0.00f * b + l0;
  l1 = 0.03f * b + l1;
  l2 = 0.97f * b + l2;
  const float denom = 0.68990272f * l0 + 0.34832189f * l1;
  const float a = (denom == 0.f) ? 0.f : Y / denom;
  return { l0 * a, l1 * a, l2 * a, 0.f };
}

__device__ __forceinline__ v4 pipe_RGB_to_Ych(const v4 in, const m3 &M)
{
  const v4 Yrg = LMS_to_Yrg(mat3(M, in));
  const float r = Yrg.y - 0.21902143f;
  const float g = Yrg.z - 0.54371398f;
  const float c = sqrtf(g * g + r * r); // dt_fast_hypotf(g, r)
  const float cos_h = c != 0.f ? r / c : 1.f;
  const float sin_h = c != 0.f ? g / c : 0.f;
  return { Yrg.x, c, cos_h, sin_h };
}

__device__ __forceinline__ v4 Ych_to_pipe_RGB(const v4 in, const m3 &M)
{
  return mat3(M, Yrg_to_LMS(in.x, in.y * in.z + 0.21902143f, in.y * in.w + 0.54371398f));
}

// ---- tone curve ------------------------------------------------------------------------------
__device__ __forceinline__ float log_tonemapping(const float x, const fargs &a)
{
  return clamp_simd((ansel_math::log2f_exact(x / a.grey_source) - a.black_source) / a.dynamic_range);
}

__device__ __forceinline__ float filmic_spline(const float x, const fargs &a)
{
  using ansel_math::powf_exact;
  float result;
  if(x < a.latitude_min)
  {
    if(a.type0 == 3)
    {
      if(a.M5[0] != 0.f)
        result = a.M3[2] + fmaxf(0.f, a.M3[0] * powf_exact(fmaxf(x, 0.f), a.M4[0]));
      else
      {
        const float ty = a.latitude_min * a.M2[2] + a.M1[2];
        const float u = a.M2[2] * (x - a.latitude_min) / a.M1[0];
        result = a.M1[0] * (u / powf_exact(1.f + powf_exact(u, a.M2[0]), 1.f / a.M2[0])) + ty;
      }
    }
    else if(a.type0 == 0)
      result = a.M1[0] + x * (a.M2[0] + x * (a.M3[0] + x * (a.M4[0] + x * a.M5[0])));
    else if(a.type0 == 1)
      result = a.M1[0] + x * (a.M2[0] + x * (a.M3[0] + x * a.M4[0]));
    else
    {
      const float xi = a.latitude_min - x;
      const float rat = xi * (xi * a.M2[0] + 1.f);
      result = a.M4[0] - a.M1[0] * rat / (rat + a.M3[0]);
    }
  }
  else if(x > a.latitude_max)
  {
    if(a.type1 == 3)
    {
      if(a.M5[1] != 0.f)
        result = a.M4[2] - fmaxf(0.f, a.M3[1] * powf_exact(fmaxf(1.f - x, 0.f), a.M4[1]));
      else
      {
        const float ty = a.latitude_max * a.M2[2] + a.M1[2];
        const float u = a.M2[2] * (x - a.latitude_max) / a.M1[1];
        result = a.M1[1] * (u / powf_exact(1.f + powf_exact(u, a.M2[1]), 1.f / a.M2[1])) + ty;
      }
    }
    else if(a.type1 == 0)
      result = a.M1[1] + x * (a.M2[1] + x * (a.M3[1] + x * (a.M4[1] + x * a.M5[1])));
    else if(a.type1 == 1)
      result = a.M1[1] + x * (a.M2[1] + x * (a.M3[1] + x * a.M4[1]));
    else
    {
      const float xi = x - a.latitude_max;
      const float rat = xi * (xi * a.M2[1] + 1.f);
      result = a.M4[1] + a.M1[1] * rat / (rat + a.M3[1]);
    }
  }
  else
    result = a.M1[2] + x * a.M2[2];
  return result;
}

__device__ __forceinline__ float tone_channel(const float v, const fargs &a, const float lo)
{
  const float mapped = log_tonemapping(v, a);
  return ansel_math::powf_exact(clampf(filmic_spline(mapped, a), lo, a.y4), a.output_power);
}

__device__ __forceinline__ v4 RGB_tone_mapping_v4(const v4 p, const fargs &a)
{
  return { tone_channel(p.x, a, 0.f), tone_channel(p.y, a, 0.f), tone_channel(p.z, a, 0.f), p.w };
}

__device__ __forceinline__ float pixel_norm(const v4 p, const int variant, const fargs &a)
{
  switch(variant)
  {
    case 1: return fmaxf(fmaxf(p.x, p.y), p.z);
    case 3:
    {
      float numerator = 0.0f, denominator = 0.0f;
      const float c[3] = { p.x, p.y, p.z };
#pragma unroll
      for(int k = 0; k < 3; k++)
      {
        const float value = fabsf(c[k]);
        const float sq = value * value;
        numerator += sq * value;
        denominator += sq;
      }
      return numerator / fmaxf(denominator, 1e-12f);
    }
    case 4: return sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);
    case 5: return sqrtf(p.x * p.x + p.y * p.y + p.z * p.z) * INVERSE_SQRT_3;
    default: return a.luma[0] * p.x + a.luma[1] * p.y + a.luma[2] * p.z;
  }
}

__device__ __forceinline__ v4 norm_tone_mapping_v4(const v4 p, const int type, const fargs &a)
{
  float norm = clampf(pixel_norm(p, type, a), a.norm_min, a.norm_max);
  const v4 ratios = { p.x / norm, p.y / norm, p.z / norm, p.w / norm };
  norm = log_tonemapping(norm, a);
  norm = ansel_math::powf_exact(clampf(filmic_spline(norm, a), a.y0, a.y4), a.output_power);
  return { ratios.x * norm, ratios.y * norm, ratios.z * norm, ratios.w * norm };
}

// ---- gamut mapping ----------------------------------------------------------------------------
__device__ __forceinline__ v4 filmic_desaturate_v4(const v4 Yo, v4 Yf, const float saturation)
{
  const float chroma_original = Yo.y * Yo.x;
  float chroma_final = Yf.y * Yf.x;
  const float delta_chroma = saturation * (chroma_original - chroma_final);
  const bool filmic_brightens = (Yf.x > Yo.x);
  const bool filmic_resat = (chroma_original < chroma_final);
  const bool filmic_desat = (chroma_original > chroma_final);
  const bool user_resat = (saturation > 0.f);
  const bool user_desat = (saturation < 0.f);
  chroma_final = (filmic_brightens && filmic_resat) ? (chroma_original + chroma_final) / 2.f
                 : ((user_resat && filmic_desat) || user_desat) ? chroma_final + delta_chroma
                                                                : chroma_final;
  Yf.y = fmaxf(chroma_final / Yf.x, 0.f);
  return Yf;
}

__device__ __forceinline__ float clip_chroma_white_raw(const float c[3], const float target_white, const float Y,
                                                       const float cos_h, const float sin_h)
{
  const float denominator_Y_coeff = c[0] * (0.979381443298969f * cos_h + 0.391752577319588f * sin_h)
                                    + c[1] * (0.0206185567010309f * cos_h + 0.608247422680412f * sin_h)
                                    - c[2] * (cos_h + sin_h);
  const float denominator_target_term = target_white * (0.68285981628866f * cos_h + 0.482137060515464f * sin_h);
  if(denominator_Y_coeff == 0.f) return FLT_MAX;
  const float Y_asymptote = denominator_target_term / denominator_Y_coeff;
  if(Y <= Y_asymptote) return FLT_MAX;
  const float denominator = Y * denominator_Y_coeff - denominator_target_term;
  const float numerator = -0.427506877216495f
                          * (Y * (c[0] + 0.856492345150334f * c[1] + 0.554995960637719f * c[2])
                             - 0.988237752433297f * target_white);
  return numerator / denominator;
}

__device__ __forceinline__ float clip_chroma_white(const float c[3], const float target_white, const float Y,
                                                   const float cos_h, const float sin_h)
{
  const float eps = 1e-3f;
  const float max_Y = CIE_Y_1931_to_CIE_Y_2006(target_white);
  const float delta_Y = max_(max_Y - Y, 0.f);
  float max_chroma;
  if(delta_Y < eps)
    max_chroma = delta_Y / (eps * max_Y) * clip_chroma_white_raw(c, target_white, (1.f - eps) * max_Y, cos_h, sin_h);
  else
    max_chroma = clip_chroma_white_raw(c, target_white, Y, cos_h, sin_h);
  return max_chroma >= 0.f ? max_chroma : FLT_MAX;
}

__device__ __forceinline__ float clip_chroma_black(const float c[3], const float cos_h, const float sin_h)
{
  const float denominator = c[0] * (0.979381443298969f * cos_h + 0.391752577319588f * sin_h)
                            + c[1] * (0.0206185567010309f * cos_h + 0.608247422680412f * sin_h)
                            - c[2] * (cos_h + sin_h);
  if(denominator == 0.f) return FLT_MAX;
  const float numerator = -0.427506877216495f * (c[0] + 0.856492345150334f * c[1] + 0.554995960637719f * c[2]);
  const float max_chroma = numerator / denominator;
  return max_chroma >= 0.f ? max_chroma : FLT_MAX;
}

__device__ __forceinline__ float clip_chroma(const m3 &mo, const float target_white, const float Y, const float cos_h,
                                             const float sin_h, const float chroma)
{
  const float wr = clip_chroma_white(mo.r[0], target_white, Y, cos_h, sin_h);
  const float wg = clip_chroma_white(mo.r[1], target_white, Y, cos_h, sin_h);
  const float wb = clip_chroma_white(mo.r[2], target_white, Y, cos_h, sin_h);
  const float max_chroma_white = min_(min_(wr, wg), wb);
  const float br = clip_chroma_black(mo.r[0], cos_h, sin_h);
  const float bg = clip_chroma_black(mo.r[1], cos_h, sin_h);
  const float bb = clip_chroma_black(mo.r[2], cos_h, sin_h);
  const float max_chroma_black = min_(min_(br, bg), bb);
  return min_(min_(chroma, max_chroma_black), max_chroma_white);
}

__device__ __forceinline__ v4 gamut_check_Yrg(const v4 Ych)
{
  const float Yrg1 = Ych.y * Ych.z + 0.21902143f;
  const float Yrg2 = Ych.y * Ych.w + 0.54371398f;
  float max_c = Ych.y;
  if(Yrg1 < 0.f) max_c = fminf(-0.21902143f / Ych.z, max_c);
  if(Yrg2 < 0.f) max_c = fminf(-0.54371398f / Ych.w, max_c);
  if(Yrg1 + Yrg2 > 1.f) max_c = fminf((1.f - 0.21902143f - 0.54371398f) / (Ych.z + Ych.w), max_c);
  return { Ych.x, max_c, Ych.z, Ych.w };
}

__device__ __forceinline__ v4 gamut_check_RGB(const m3 &mi, const m3 &mo, const float display_black,
                                              const float display_white, const v4 Ych_in)
{
  v4 b = Ych_to_pipe_RGB(Ych_in, mo);
  const float min_pix = min_(min_(b.x, b.y), b.z);
  const float black_offset = max_(-min_pix, 0.f);
  b = { b.x + black_offset, b.y + black_offset, b.z + black_offset, b.w + black_offset };
  const v4 Ych_brightened = pipe_RGB_to_Ych(b, mi);
  const float Y = clamp_glib((Ych_in.x + Ych_brightened.x) / 2.f, CIE_Y_1931_to_CIE_Y_2006(display_black),
                             CIE_Y_1931_to_CIE_Y_2006(display_white));
  const float new_chroma = clip_chroma(mo, display_white, Y, Ych_in.z, Ych_in.w, Ych_in.y);
  v4 o = Ych_to_pipe_RGB({ Y, new_chroma, Ych_in.z, Ych_in.w }, mo);
  o.x = clamp_glib(o.x, 0.f, display_white);
  o.y = clamp_glib(o.y, 0.f, display_white);
  o.z = clamp_glib(o.z, 0.f, display_white);
  o.w = clamp_glib(o.w, 0.f, display_white);
  return o;
}

template <bool EXPORT>
__device__ __forceinline__ v4 gamut_mapping(v4 Yf, const v4 Yo, const fargs &a, const float saturation)
{
  Yf.z = Yo.z;
  Yf.w = Yo.w;
  Yf.x = clamp_glib(Yf.x, CIE_Y_1931_to_CIE_Y_2006(a.display_black), CIE_Y_1931_to_CIE_Y_2006(a.display_white));
  Yf = filmic_desaturate_v4(Yo, Yf, saturation);
  Yf = gamut_check_Yrg(Yf);
  if(!EXPORT) return gamut_check_RGB(a.input, a.output, a.display_black, a.display_white, Yf);
  const v4 pix_out = gamut_check_RGB(a.export_input, a.export_output, a.display_black, a.display_white, Yf);
  return mat3(a.output, mat3(a.export_input, pix_out));
}

__device__ __forceinline__ v4 agx_compress_negatives(const v4 p, const float luma[3])
{
  const float input_y = p.x * luma[0] + p.y * luma[1] + p.z * luma[2];
  const float max_rgb = fmaxf(fmaxf(p.x, p.y), p.z);
  const float min_rgb = fminf(fminf(p.x, p.y), p.z);
  const float o0 = max_rgb - p.x, o1 = max_rgb - p.y, o2 = max_rgb - p.z;
  const float opponent_y = o0 * luma[0] + o1 * luma[1] + o2 * luma[2];
  const float max_opponent = fmaxf(fmaxf(o0, o1), o2);
  const float y_compensated = max_opponent - opponent_y + input_y;
  const float offset = fmaxf(-min_rgb, 0.f);
  const v4 s = { p.x + offset, p.y + offset, p.z + offset, p.w + offset };
  const float max_shifted = fmaxf(fmaxf(s.x, s.y), s.z);
  const float q0 = max_shifted - s.x, q1 = max_shifted - s.y, q2 = max_shifted - s.z;
  const float max_opponent_shifted = fmaxf(fmaxf(q0, q1), q2);
  const float y_opponent_shifted = q0 * luma[0] + q1 * luma[1] + q2 * luma[2];
  float y_new = s.x * luma[0] + s.y * luma[1] + s.z * luma[2];
  y_new += max_opponent_shifted - y_opponent_shifted;
  const float ratio = (y_new > y_compensated && y_new > 1e-6f) ? y_compensated / y_new : 1.f;
  return { s.x * ratio, s.y * ratio, s.z * ratio, s.w * ratio };
}

enum { MODE_AGX = 0, MODE_V5 = 1, MODE_SPLIT_V4 = 2, MODE_CHROMA_V4 = 3 };

template <int MODE, bool EXPORT>
__global__ __launch_bounds__(256) void filmic_kernel(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                      const size_t npixels, const fargs a)
{
  for(size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < npixels; k += (size_t)gridDim.x * blockDim.x)
  {
    const float4 pi = in[k];
    v4 pix_in = { pi.x, pi.y, pi.z, pi.w };
    v4 res;
    if(MODE == MODE_AGX)
    {
      pix_in.x = isnan(pix_in.x) ? 0.f : clampf(pix_in.x, -1e6f, 1e6f);
      pix_in.y = isnan(pix_in.y) ? 0.f : clampf(pix_in.y, -1e6f, 1e6f);
      pix_in.z = isnan(pix_in.z) ? 0.f : clampf(pix_in.z, -1e6f, 1e6f);
      const v4 compressed = agx_compress_negatives(pix_in, a.luma);
      const v4 Yo = pipe_RGB_to_Ych(compressed, a.input);
      v4 rendering = mat3(a.inset, compressed);
      rendering = RGB_tone_mapping_v4(rendering, a);
      const v4 pix_out = mat3(a.outset, rendering);
      v4 Yf = pipe_RGB_to_Ych(pix_out, a.input);
      const float chroma_final = fminf(Yo.y, Yf.y);
      const float r_mix = a.beta_hue * Yo.y * Yo.z + (1.f - a.beta_hue) * chroma_final * Yf.z;
      const float g_mix = a.beta_hue * Yo.y * Yo.w + (1.f - a.beta_hue) * chroma_final * Yf.w;
      const float norm_mix = sqrtf(g_mix * g_mix + r_mix * r_mix);
      v4 Yref = Yo;
      Yref.z = (norm_mix > 1e-9f) ? r_mix / norm_mix : Yo.z;
      Yref.w = (norm_mix > 1e-9f) ? g_mix / norm_mix : Yo.w;
      Yf.y = chroma_final;
      res = gamut_mapping<EXPORT>(Yf, Yref, a, 0.f);
    }
    else if(MODE == MODE_V5)
    {
      const v4 naive = RGB_tone_mapping_v4(pix_in, a);
      const v4 mx = norm_tone_mapping_v4(pix_in, 1, a);
      const float ws = 0.5f + a.saturation, wn = 0.5f - a.saturation;
      v4 po = { ws * mx.x, ws * mx.y, ws * mx.z, ws * mx.w };
      po = { wn * naive.x + po.x, wn * naive.y + po.y, wn * naive.z + po.z, wn * naive.w + po.w };
      const v4 Yo = pipe_RGB_to_Ych(pix_in, a.input);
      v4 Yf = pipe_RGB_to_Ych(po, a.input);
      Yf.y = fminf(Yo.y, Yf.y);
      res = gamut_mapping<EXPORT>(Yf, Yo, a, 0.f);
    }
    else if(MODE == MODE_SPLIT_V4)
    {
      const v4 po = RGB_tone_mapping_v4(pix_in, a);
      const v4 Yo = pipe_RGB_to_Ych(pix_in, a.input);
      v4 Yf = pipe_RGB_to_Ych(po, a.input);
      Yf.y = fminf(Yo.y, Yf.y);
      res = gamut_mapping<EXPORT>(Yf, Yo, a, a.saturation);
    }
    else
    {
      const v4 po = norm_tone_mapping_v4(pix_in, a.preserve_color, a);
      const v4 Yo = pipe_RGB_to_Ych(pix_in, a.input);
      const v4 Yf = pipe_RGB_to_Ych(po, a.input);
      res = gamut_mapping<EXPORT>(Yf, Yo, a, a.saturation);
    }
    nt_store(out + k, make_float4(res.x, res.y, res.z, res.w));
  }
}

// ---------------------------------------------------------------------------------------------
// host: per-call preparation (binary32, reference operation order)
// ---------------------------------------------------------------------------------------------
typedef float mat_t[4][4];

const mat_t XYZ_D50_to_D65_CAT16 = { { 9.89466254e-01f, -4.00304626e-02f, 4.40530317e-02f, 0.f },
                                     { -5.40518733e-03f, 1.00666069e+00f, -1.75551955e-03f, 0.f },
                                     { -4.03920992e-04f, 1.50768030e-02f, 1.30210211e+00f, 0.f } };
const mat_t XYZ_D65_to_D50_CAT16 = { { 1.01085433e+00f, 4.07086103e-02f, -3.41445825e-02f, 0.f },
                                     { 5.42814201e-03f, 9.93581926e-01f, 1.15592039e-03f, 0.f },
                                     { 2.50722468e-04f, -1.14918759e-02f, 7.67964947e-01f, 0.f } };
const mat_t XYZ_D65_to_LMS_2006_D65 = { { 0.257085f, 0.859943f, -0.031061f, 0.f },
                                        { -0.394427f, 1.175800f, 0.106423f, 0.f },
                                        { 0.064856f, -0.076250f, 0.559067f, 0.f } };
const mat_t LMS_2006_D65_to_XYZ_D65 = { { 1.80794659f, -1.29971660f, 0.34785879f, 0.f },
                                        { 0.61783960f, 0.39595453f, -0.04104687f, 0.f },
                                        { -0.12546960f, 0.20478038f, 1.74274183f, 0.f } };
const mat_t filmlightRGB_D65_to_LMS_D65 = { { 0.95f, 0.38f, 0.00f, 0.f }, { 0.05f, 0.62f, 0.03f, 0.f }, { 0.00f, 0.00f, 0.97f, 0.f } };
const mat_t LMS_D65_to_filmlightRGB_D65 = { { 1.0877193f, -0.66666667f, 0.02061856f, 0.f },
                                            { -0.0877193f, 1.66666667f, -0.05154639f, 0.f },
                                            { 0.f, 0.f, 1.03092784f, 0.f } };

// dt_colormatrix_mul(), src/math/matrices.h:167-179
void mat_mul(mat_t dst, const mat_t m1, const mat_t m2)
{
  mat_t t;
  for(int k = 0; k < 3; ++k)
    for(int i = 0; i < 4; i++)
    {
      float sum = 0.0f;
      for(int j = 0; j < 3; j++) sum += m1[k][j] * m2[j][i];
      t[k][i] = sum;
    }
  for(int i = 0; i < 4; i++) t[3][i] = 0.f;
  memcpy(dst, t, sizeof(mat_t));
}

// mat3SSEinv(), src/math/matrices.h:37-66
int mat_inv(mat_t dst, const mat_t src)
{
#define A(y, x) src[(y - 1)][(x - 1)]
#define B(y, x) dst[(y - 1)][(x - 1)]
  const float det = A(1, 1) * (A(3, 3) * A(2, 2) - A(3, 2) * A(2, 3)) - A(2, 1) * (A(3, 3) * A(1, 2) - A(3, 2) * A(1, 3))
                    + A(3, 1) * (A(2, 3) * A(1, 2) - A(2, 2) * A(1, 3));
  if(fabsf(det) < 1e-7f) return 1;
  const float invDet = 1.f / det;
  B(1, 1) = invDet * (A(3, 3) * A(2, 2) - A(3, 2) * A(2, 3));
  B(1, 2) = -invDet * (A(3, 3) * A(1, 2) - A(3, 2) * A(1, 3));
  B(1, 3) = invDet * (A(2, 3) * A(1, 2) - A(2, 2) * A(1, 3));
  B(2, 1) = -invDet * (A(3, 3) * A(2, 1) - A(3, 1) * A(2, 3));
  B(2, 2) = invDet * (A(3, 3) * A(1, 1) - A(3, 1) * A(1, 3));
  B(2, 3) = -invDet * (A(2, 3) * A(1, 1) - A(2, 1) * A(1, 3));
  B(3, 1) = invDet * (A(3, 2) * A(2, 1) - A(3, 1) * A(2, 2));
  B(3, 2) = -invDet * (A(3, 2) * A(1, 1) - A(3, 1) * A(1, 2));
  B(3, 3) = invDet * (A(2, 2) * A(1, 1) - A(2, 1) * A(1, 2));
#undef A
#undef B
  return 0;
}

// dot_product() (matrices.h:201-206) with scalar_product()'s accumulation (math.h:186-194)
void dotp(const float v[4], const mat_t M, float o[4])
{
  for(int i = 0; i < 3; i++)
  {
    float acc = 0.f;
    for(int c = 0; c < 3; c++) acc += v[c] * M[i][c];
    o[i] = acc;
  }
}

void agx_xyz_D50_to_Yrg(const float xyz_D50[4], float Yrg[4])
{
  float xyz_D65[4] = { 0.f }, lms[4] = { 0.f };
  dotp(xyz_D50, XYZ_D50_to_D65_CAT16, xyz_D65);
  dotp(xyz_D65, XYZ_D65_to_LMS_2006_D65, lms);
  const float Y = 0.68990272f * lms[0] + 0.34832189f * lms[1];
  const float a = lms[0] + lms[1] + lms[2];
  float n[4] = { 0.f }, rgb[4] = { 0.f };
  for(int c = 0; c < 4; c++) n[c] = (a == 0.f) ? 0.f : lms[c] / a;
  dotp(n, LMS_D65_to_filmlightRGB_D65, rgb);
  Yrg[0] = Y;
  Yrg[1] = rgb[0];
  Yrg[2] = rgb[1];
}

void agx_Yrg_to_xyz_D50(const float Yrg[4], float xyz_D50[4])
{
  const float Y = Yrg[0], r = Yrg[1], g = Yrg[2];
  const float b = 1.f - r - g;
  const float rgb[4] = { r, g, b, 0.f };
  float lms[4] = { 0.f }, LMS[4] = { 0.f }, xyz_D65[4] = { 0.f };
  dotp(rgb, filmlightRGB_D65_to_LMS_D65, lms);
  const float denom = (0.68990272f * lms[0] + 0.34832189f * lms[1]);
  const float a = (denom == 0.f) ? 0.f : Y / denom;
  for(int c = 0; c < 4; c++) LMS[c] = lms[c] * a;
  dotp(LMS, LMS_2006_D65_to_XYZ_D65, xyz_D65);
  dotp(xyz_D65, XYZ_D65_to_D50_CAT16, xyz_D50);
}

// _filmic_agx_build_displaced(), filmicrgb.c:2344-2388
bool agx_build_displaced(const mat_t work_in, const mat_t work_out, const float inset[3], const float rotation[3], mat_t M)
{
  float white_xyz[4] = { 0.f }, white_Yrg[4] = { 0.f };
  for(int r = 0; r < 3; r++)
    for(int c = 0; c < 3; c++) white_xyz[r] += work_in[r][c];
  agx_xyz_D50_to_Yrg(white_xyz, white_Yrg);
  mat_t P = { { 0.f } };
  for(int i = 0; i < 3; i++)
  {
    const float primary_xyz[4] = { work_in[0][i], work_in[1][i], work_in[2][i], 0.f };
    float primary_Yrg[4] = { 0.f };
    agx_xyz_D50_to_Yrg(primary_xyz, primary_Yrg);
    const float dr = primary_Yrg[1] - white_Yrg[1];
    const float dg = primary_Yrg[2] - white_Yrg[2];
    const float in_i = inset[i];
    const float scale = 1.f - (in_i >= 0.f ? (in_i <= 0.9f ? in_i : 0.9f) : 0.f);
    const float cos_a = cosf(rotation[i]);
    const float sin_a = sinf(rotation[i]);
    const float displaced_Yrg[4] = { primary_Yrg[0], white_Yrg[1] + scale * (cos_a * dr - sin_a * dg),
                                     white_Yrg[2] + scale * (sin_a * dr + cos_a * dg), 0.f };
    float displaced_xyz[4] = { 0.f };
    agx_Yrg_to_xyz_D50(displaced_Yrg, displaced_xyz);
    for(int r = 0; r < 3; r++) P[r][i] = displaced_xyz[r];
  }
  mat_t Pinv = { { 0.f } };
  if(mat_inv(Pinv, P)) return false;
  float s[4] = { 0.f };
  dotp(white_xyz, Pinv, s);
  for(int r = 0; r < 3; r++)
    for(int c = 0; c < 3; c++) P[r][c] *= s[c];
  mat_mul(M, work_out, P);
  return true;
}

void mat_identity(mat_t M)
{
  for(int r = 0; r < 4; r++)
    for(int c = 0; c < 4; c++) M[r][c] = (r == c && r < 3) ? 1.f : 0.f;
}

// filmic_agx_prepare_bracket(), filmicrgb.c:2390-2459
void agx_prepare_bracket(const mat_t work_in, const mat_t work_out, const int variant, mat_t inset, mat_t outset)
{
  // { inset[3], rotation[3], outset[3], outset_rotation[3] } per bleach variant, versions 5..9
  static const float K[5][12] = {
    { +0.5991055f, +0.6000000f, +0.3300009f, +0.0571015f, +0.1999891f, +0.0886110f, 0.761433f, 0.752267f, 0.465293f, -0.0034297f, +0.1952448f, -0.0480109f },
    { +0.6410825f, +0.6898110f, +0.3194529f, +0.0405734f, +0.1631286f, +0.0350584f, 0.784757f, 0.789387f, 0.445403f, -0.0057845f, +0.1593207f, -0.0592955f },
    { +0.6509540f, +0.7488775f, +0.3517703f, +0.0278602f, +0.1214671f, -0.0228829f, 0.793082f, 0.815169f, 0.460318f, -0.0053781f, +0.1187604f, -0.0794801f },
    { +0.6379749f, +0.7878689f, +0.3753822f, +0.0106096f, +0.0582598f, -0.0696729f, 0.790237f, 0.831376f, 0.465406f, -0.0080070f, +0.0571100f, -0.0912220f },
    { +0.5770235f, +0.8102094f, +0.4000390f, -0.0081060f, -0.0034008f, -0.1035236f, 0.766420f, 0.838020f, 0.465130f, -0.0122011f, -0.0021732f, -0.0971215f },
  };
  const float *k = K[(variant >= 5 && variant <= 9) ? variant - 5 : 0];
  mat_t rec = { { 0.f } };
  if(!agx_build_displaced(work_in, work_out, k + 0, k + 3, inset) || !agx_build_displaced(work_in, work_out, k + 6, k + 9, rec)
     || mat_inv(outset, rec))
  {
    mat_identity(inset);
    mat_identity(outset);
  }
}

void to_mat(mat_t m, const float a[3][4])
{
  memset(m, 0, sizeof(mat_t));
  for(int r = 0; r < 3; r++)
    for(int c = 0; c < 3; c++) m[r][c] = a[r][c];
}

void to_m3(m3 &o, const mat_t m)
{
  for(int r = 0; r < 3; r++)
    for(int c = 0; c < 3; c++) o.r[r][c] = m[r][c];
}

void filmic_prepare(const dt_hip_filmicrgb_data_t *d, fargs &a)
{
  memset(&a, 0, sizeof(a));
  mat_t work_in, work_out, exp_in, exp_out, tmp, m;
  to_mat(work_in, d->work_matrix_in);
  to_mat(work_out, d->work_matrix_out);
  to_mat(exp_in, d->export_matrix_in);
  to_mat(exp_out, d->export_matrix_out);
  // filmic_v4_prepare_matrices(), filmicrgb.c:2033-2064
  mat_mul(tmp, XYZ_D50_to_D65_CAT16, work_in);
  mat_mul(m, XYZ_D65_to_LMS_2006_D65, tmp);
  to_m3(a.input, m);
  mat_mul(tmp, XYZ_D65_to_D50_CAT16, LMS_2006_D65_to_XYZ_D65);
  mat_mul(m, work_out, tmp);
  to_m3(a.output, m);
  if(d->use_output_profile)
  {
    mat_mul(tmp, XYZ_D65_to_D50_CAT16, LMS_2006_D65_to_XYZ_D65);
    mat_mul(m, exp_out, tmp);
    to_m3(a.export_output, m);
    mat_mul(tmp, XYZ_D50_to_D65_CAT16, exp_in);
    mat_mul(m, XYZ_D65_to_LMS_2006_D65, tmp);
    to_m3(a.export_input, m);
  }
  if(d->version >= 5)
  {
    mat_t inset = { { 0.f } }, outset = { { 0.f } };
    agx_prepare_bracket(work_in, work_out, d->version, inset, outset);
    to_m3(a.inset, inset);
    to_m3(a.outset, outset);
  }
  for(int c = 0; c < 3; c++) a.luma[c] = work_in[1][c];
  // exp_tonemapping_v2(), filmicrgb.c:1054-1060
  a.norm_min = d->grey_source * exp2f(d->dynamic_range * 0.f + d->black_source);
  a.norm_max = d->grey_source * exp2f(d->dynamic_range * 1.f + d->black_source);
  a.display_white = powf(d->spline.y[4], d->output_power);
  a.display_black = powf(d->spline.y[0], d->output_power);
  a.grey_source = d->grey_source;
  a.black_source = d->black_source;
  a.dynamic_range = d->dynamic_range;
  a.output_power = d->output_power;
  a.saturation = d->saturation;
  a.beta_hue = d->agx_beta_hue;
  for(int k = 0; k < 3; k++)
  {
    a.M1[k] = d->spline.M1[k];
    a.M2[k] = d->spline.M2[k];
    a.M3[k] = d->spline.M3[k];
    a.M4[k] = d->spline.M4[k];
    a.M5[k] = d->spline.M5[k];
  }
  a.latitude_min = d->spline.latitude_min;
  a.latitude_max = d->spline.latitude_max;
  a.y0 = d->spline.y[0];
  a.y4 = d->spline.y[4];
  a.type0 = d->spline.type[0];
  a.type1 = d->spline.type[1];
  a.preserve_color = d->preserve_color;
}

template <int MODE>
void launch_m(const bool exp, const unsigned grid, hipStream_t s, const float4 *in, float4 *out, const size_t np, const fargs &a)
{
  if(exp)
    filmic_kernel<MODE, true><<<grid, 256, 0, s>>>(in, out, np, a);
  else
    filmic_kernel<MODE, false><<<grid, 256, 0, s>>>(in, out, np, a);
}

} // namespace

extern "C" int dt_hip_iop_filmicrgb_process(int devid, const dt_hip_piece_t *piece, const dt_hip_filmicrgb_data_t *d,
                                            dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  if(piece->channels != 4) return DT_HIP_INVALID_ARG;
  if(d->version < 3 || d->version > 9)
  {
    set_last_error("filmicrgb: colour science %d (v3/v4/v5, 2019-2021) is not implemented on device", d->version);
    return DT_HIP_INVALID_ARG;
  }
  const size_t np = (size_t)piece->roi_out.width * piece->roi_out.height;
  if(np == 0) return DT_HIP_SUCCESS;
  fargs a;
  filmic_prepare(d, a);
  const unsigned grid = stream_grid(np, 256);
  hipStream_t s = stream_of(devid);
  const float4 *in = (const float4 *)dev_in;
  float4 *out = (float4 *)dev_out;
  const bool exp = d->use_output_profile != 0;
  launch_scope ls(devid, "filmicrgb");
  if(d->version >= 5)
    launch_m<MODE_AGX>(exp, grid, s, in, out, np, a);
  else if(d->version == 4)
    launch_m<MODE_V5>(exp, grid, s, in, out, np, a);
  else if(d->preserve_color == 0)
    launch_m<MODE_SPLIT_V4>(exp, grid, s, in, out, np, a);
  else
    launch_m<MODE_CHROMA_V4>(exp, grid, s, in, out, np, a);
  return check_launch("filmicrgb");
}
