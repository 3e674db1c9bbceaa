// px_filmicrgb.h -- per-pixel body of filmic RGB, shared by the standalone kernel (filmicrgb.hip)
// and the fused pointwise chain (pipe_fused.hip).  Reference citations in filmicrgb.hip.
#pragma once
#include "hip_common.h"
#include "devmath.h"
#include <float.h>

namespace ansel
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
  float inv_M2[2]; // 1.f / M2[side], the sigmoid's outer exponent (uniform, computed once on the host)
  float latitude_min, latitude_max, y0, y4;
  int type0, type1;
  int preserve_color;
  int mode, use_export; // MODE_* of the colour science, export-profile gamut mapping
  // colour sciences v1..v3: commit_params(), filmicrgb.c:4101-4102, and the version (0..2)
  float sigma_toe, sigma_shoulder;
  int legacy_version;
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
  const float hc = c != 0.f ? r / c : 1.f;
  const float hs = c != 0.f ? g / c : 0.f;
  return { Yrg.x, c, hc, hs };
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

// filmic_spline(), filmicrgb.c:1063-1160.  Same arithmetic per lane as the reference's three-way
// branch, organised for a 64-wide wave: toe, latitude and shoulder pixels sit side by side in real
// images, so a branch per zone would run every powf of every zone for the whole wave (2 + 1 per
// channel with the default "perceptual" curves).  Instead each lane selects the operands of its
// FIRST power (toe sigmoid: u^p; slope-matched power toe / shoulder: x^q or (1-x)^q) and all lanes
// that need one run it together; the toe sigmoid's second power runs the same way.  Polynomial and
// rational curves have no powf and keep their closed forms.
__device__ __forceinline__ float filmic_spline(const float x, const fargs &a)
{
  using ansel_math::powf_exact;
  const bool toe = x < a.latitude_min;
  const bool shoulder = !toe && x > a.latitude_max;
  const bool sig0 = a.type0 == 3, sig1 = a.type1 == 3;     // uniform
  const bool pow0 = a.M5[0] != 0.f, pow1 = a.M5[1] != 0.f; // uniform: slope-matched power curve instead of the sigmoid
  float result = a.M1[2] + x * a.M2[2]; // latitude
  if(sig0 | sig1)
  {
    const bool t_sig = toe && sig0, s_sig = shoulder && sig1;
    // first power.  One division for the sigmoid argument u = slope * (x - x_t) / scale, whichever
    // side the lane is on
    const float xt = t_sig ? a.latitude_min : a.latitude_max;
    const float sc = t_sig ? a.M1[0] : a.M1[1];
    const bool powcurve = t_sig ? pow0 : pow1;
    const float u = a.M2[2] * (x - xt) / sc;
    const float base1 = powcurve ? fmaxf(t_sig ? x : 1.f - x, 0.f) : u;
    const float e1 = t_sig ? (pow0 ? a.M4[0] : a.M2[0]) : (pow1 ? a.M4[1] : a.M2[1]);
    float p1 = 0.f;
    if(t_sig | s_sig) p1 = powf_exact(base1, e1);
    // second power: the generalized sigmoid u / (1 + u^p)^(1/p)
    const bool need2 = (t_sig | s_sig) && !powcurve;
    float p2 = 1.f;
    if(need2) p2 = powf_exact(1.f + p1, t_sig ? a.inv_M2[0] : a.inv_M2[1]);
    const float u0 = u, u1 = u;
    if(t_sig)
    {
      const float ty = a.latitude_min * a.M2[2] + a.M1[2];
      result = pow0 ? a.M3[2] + fmaxf(0.f, a.M3[0] * p1) : a.M1[0] * (u0 / p2) + ty;
    }
    if(s_sig)
    {
      const float ty = a.latitude_max * a.M2[2] + a.M1[2];
      result = pow1 ? a.M4[2] - fmaxf(0.f, a.M3[1] * p1) : a.M1[1] * (u1 / p2) + ty;
    }
  }
  if(toe && !sig0)
  {
    if(a.type0 == 0)
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
  if(shoulder && !sig1)
  {
    if(a.type1 == 0)
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
  const float chroma_in = Yo.y * Yo.x;
  float chroma_out = Yf.y * Yf.x;
  const float chroma_shift = saturation * (chroma_in - chroma_out);
  const bool curve_brightens = (Yf.x > Yo.x);
  const bool curve_resat = (chroma_in < chroma_out);
  const bool curve_desat = (chroma_in > chroma_out);
  const bool asked_resat = (saturation > 0.f);
  const bool asked_desat = (saturation < 0.f);
  chroma_out = (curve_brightens && curve_resat) ? (chroma_in + chroma_out) / 2.f
                 : ((asked_resat && curve_desat) || asked_desat) ? chroma_out + chroma_shift
                                                                : chroma_out;
  Yf.y = fmaxf(chroma_out / Yf.x, 0.f);
  return Yf;
}

__device__ __forceinline__ float clip_chroma_white_raw(const float c[3], const float target_white, const float Y,
                                                       const float hc, const float hs)
{
  const float y_slope = c[0] * (0.979381443298969f * hc + 0.391752577319588f * hs)
                                    + c[1] * (0.0206185567010309f * hc + 0.608247422680412f * hs)
                                    - c[2] * (hc + hs);
  const float white_term = target_white * (0.68285981628866f * hc + 0.482137060515464f * hs);
  if(y_slope == 0.f) return FLT_MAX;
  const float y_pole = white_term / y_slope;
  if(Y <= y_pole) return FLT_MAX;
  const float denominator = Y * y_slope - white_term;
  const float numerator = -0.427506877216495f
                          * (Y * (c[0] + 0.856492345150334f * c[1] + 0.554995960637719f * c[2])
                             - 0.988237752433297f * target_white);
  return numerator / denominator;
}

__device__ __forceinline__ float clip_chroma_white(const float c[3], const float target_white, const float Y,
                                                   const float hc, const float hs)
{
  const float eps = 1e-3f;
  const float max_Y = CIE_Y_1931_to_CIE_Y_2006(target_white);
  const float delta_Y = max_(max_Y - Y, 0.f);
  float limit_c;
  if(delta_Y < eps)
    limit_c = delta_Y / (eps * max_Y) * clip_chroma_white_raw(c, target_white, (1.f - eps) * max_Y, hc, hs);
  else
    limit_c = clip_chroma_white_raw(c, target_white, Y, hc, hs);
  return limit_c >= 0.f ? limit_c : FLT_MAX;
}

__device__ __forceinline__ float clip_chroma_black(const float c[3], const float hc, const float hs)
{
  const float denominator = c[0] * (0.979381443298969f * hc + 0.391752577319588f * hs)
                            + c[1] * (0.0206185567010309f * hc + 0.608247422680412f * hs)
                            - c[2] * (hc + hs);
  if(denominator == 0.f) return FLT_MAX;
  const float numerator = -0.427506877216495f * (c[0] + 0.856492345150334f * c[1] + 0.554995960637719f * c[2]);
  const float limit_c = numerator / denominator;
  return limit_c >= 0.f ? limit_c : FLT_MAX;
}

__device__ __forceinline__ float clip_chroma(const m3 &mo, const float target_white, const float Y, const float hc,
                                             const float hs, const float chroma)
{
  const float wr = clip_chroma_white(mo.r[0], target_white, Y, hc, hs);
  const float wg = clip_chroma_white(mo.r[1], target_white, Y, hc, hs);
  const float wb = clip_chroma_white(mo.r[2], target_white, Y, hc, hs);
  const float limit_white = min_(min_(wr, wg), wb);
  const float br = clip_chroma_black(mo.r[0], hc, hs);
  const float bg = clip_chroma_black(mo.r[1], hc, hs);
  const float bb = clip_chroma_black(mo.r[2], hc, hs);
  const float limit_black = min_(min_(br, bg), bb);
  return min_(min_(chroma, limit_black), limit_white);
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

__device__ __forceinline__ v4 gamut_mapping(v4 Yf, const v4 Yo, const fargs &a, const float saturation, const bool EXPORT)
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
  const float o_max = fmaxf(fmaxf(o0, o1), o2);
  const float y_target = o_max - opponent_y + input_y;
  const float offset = fmaxf(-min_rgb, 0.f);
  const v4 s = { p.x + offset, p.y + offset, p.z + offset, p.w + offset };
  const float max_shifted = fmaxf(fmaxf(s.x, s.y), s.z);
  const float q0 = max_shifted - s.x, q1 = max_shifted - s.y, q2 = max_shifted - s.z;
  const float q_max = fmaxf(fmaxf(q0, q1), q2);
  const float y_opponent_shifted = q0 * luma[0] + q1 * luma[1] + q2 * luma[2];
  float y_new = s.x * luma[0] + s.y * luma[1] + s.z * luma[2];
  y_new += q_max - y_opponent_shifted;
  const float ratio = (y_new > y_target && y_new > 1e-6f) ? y_target / y_new : 1.f;
  return { s.x * ratio, s.y * ratio, s.z * ratio, s.w * ratio };
}

enum { MODE_AGX = 0, MODE_V5 = 1, MODE_SPLIT_V4 = 2, MODE_CHROMA_V4 = 3, MODE_SPLIT_LEGACY = 4, MODE_CHROMA_V1 = 5, MODE_CHROMA_V2_V3 = 6 };

// ---- colour sciences v1..v3 (2019-2020), kept for old edits ------------------------------------
#define FILMIC_NORM_MIN 1.52587890625e-05f // src/math/math.h:37

// filmic_desaturate_v1(), filmicrgb.c:1163-1174
__device__ __forceinline__ float filmic_desaturate_v1(const float x, const fargs &a)
{
  const float to_toe = x;
  const float to_shoulder = 1.0f - x;
  const float key_toe = ansel_math::expf_exact(-0.5f * to_toe * to_toe / a.sigma_toe);
  const float key_shoulder = ansel_math::expf_exact(-0.5f * to_shoulder * to_shoulder / a.sigma_shoulder);
  return 1.0f - clamp_simd((key_toe + key_shoulder) / a.saturation);
}

// filmic_desaturate_v2(), filmicrgb.c:1178-1189
__device__ __forceinline__ float filmic_desaturate_v2(const float x, const fargs &a)
{
  const float to_toe = x;
  const float to_shoulder = 1.0f - x;
  const float sat2 = 0.5f / sqrtf(a.saturation);
  const float key_toe = ansel_math::expf_exact(-to_toe * to_toe / a.sigma_toe * sat2);
  const float key_shoulder = ansel_math::expf_exact(-to_shoulder * to_shoulder / a.sigma_shoulder * sat2);
  return (a.saturation - (key_toe + key_shoulder) * (a.saturation));
}

// linear_saturation(), filmicrgb.c:1193-1196
__device__ __forceinline__ float linear_saturation(const float x, const float luminance, const float saturation)
{
  return luminance + saturation * (x - luminance);
}

__device__ __forceinline__ float curve_to_display(const float x, const fargs &a)
{
  return ansel_math::powf_exact(clampf(filmic_spline(x, a), a.y0, a.y4), a.output_power);
}

// filmic_split_v1() :1534-1571 and filmic_split_v2_v3() :1574-1611 (they differ by the desaturation); the reference
// leaves the output's alpha unwritten, the input's is passed through
__device__ __forceinline__ v4 filmic_split_legacy(const v4 p, const fargs &a)
{
  const float tx = log_tonemapping(fmaxf(p.x, FILMIC_NORM_MIN), a);
  const float ty = log_tonemapping(fmaxf(p.y, FILMIC_NORM_MIN), a);
  const float tz = log_tonemapping(fmaxf(p.z, FILMIC_NORM_MIN), a);
  const float lum = a.luma[0] * tx + a.luma[1] * ty + a.luma[2] * tz;
  const float desaturation = a.legacy_version ? filmic_desaturate_v2(lum, a) : filmic_desaturate_v1(lum, a);
  return { curve_to_display(linear_saturation(tx, lum, desaturation), a), curve_to_display(linear_saturation(ty, lum, desaturation), a),
           curve_to_display(linear_saturation(tz, lum, desaturation), a), p.w };
}

// filmic_chroma_v1(), filmicrgb.c:1614-1667
__device__ __forceinline__ v4 filmic_chroma_v1(const v4 p, const fargs &a)
{
  float norm = fmaxf(pixel_norm(p, a.preserve_color, a), FILMIC_NORM_MIN);
  v4 ratios = { p.x / norm, p.y / norm, p.z / norm, p.w / norm };
  const float min_ratios = fminf(fminf(ratios.x, ratios.y), ratios.z);
  if(min_ratios < 0.0f) ratios = { ratios.x - min_ratios, ratios.y - min_ratios, ratios.z - min_ratios, ratios.w - min_ratios };
  norm = log_tonemapping(norm, a);
  const float desaturation = filmic_desaturate_v1(norm, a);
  ratios = { ratios.x * norm, ratios.y * norm, ratios.z * norm, ratios.w * norm };
  const float lum = a.luma[0] * ratios.x + a.luma[1] * ratios.y + a.luma[2] * ratios.z;
  ratios.x = linear_saturation(ratios.x, lum, desaturation) / norm;
  ratios.y = linear_saturation(ratios.y, lum, desaturation) / norm;
  ratios.z = linear_saturation(ratios.z, lum, desaturation) / norm;
  norm = curve_to_display(norm, a);
  return { ratios.x * norm, ratios.y * norm, ratios.z * norm, ratios.w * norm };
}

// filmic_chroma_v2_v3(), filmicrgb.c:1670-1737
__device__ __forceinline__ v4 filmic_chroma_v2_v3(const v4 p, const fargs &a)
{
  float norm = fmaxf(pixel_norm(p, a.preserve_color, a), FILMIC_NORM_MIN);
  v4 ratios = { p.x / norm, p.y / norm, p.z / norm, p.w / norm };
  const float min_ratios = fminf(fminf(ratios.x, ratios.y), ratios.z);
  if(min_ratios < 0.0f) ratios = { ratios.x - min_ratios, ratios.y - min_ratios, ratios.z - min_ratios, ratios.w - min_ratios };
  norm = log_tonemapping(norm, a);
  const float desaturation = filmic_desaturate_v2(norm, a);
  norm = curve_to_display(norm, a);
  ratios.x = fmaxf(ratios.x + (1.0f - ratios.x) * (1.0f - desaturation), 0.0f);
  ratios.y = fmaxf(ratios.y + (1.0f - ratios.y) * (1.0f - desaturation), 0.0f);
  ratios.z = fmaxf(ratios.z + (1.0f - ratios.z) * (1.0f - desaturation), 0.0f);
  if(a.legacy_version == 2) norm /= fmaxf(pixel_norm(ratios, a.preserve_color, a), FILMIC_NORM_MIN);
  v4 o = { ratios.x * norm, ratios.y * norm, ratios.z * norm, ratios.w * norm };
  const float max_pix = fmaxf(fmaxf(o.x, o.y), o.z);
  if(max_pix > 1.0f)
  {
    ratios = { fmaxf(ratios.x + (1.0f - max_pix), 0.0f), fmaxf(ratios.y + (1.0f - max_pix), 0.0f),
               fmaxf(ratios.z + (1.0f - max_pix), 0.0f), fmaxf(ratios.w + (1.0f - max_pix), 0.0f) };
    o = { ratios.x * norm, ratios.y * norm, ratios.z * norm, ratios.w * norm };
  }
  return o;
}


template <int MODE> __device__ __forceinline__ float4 px_filmicrgb(const float4 pi, const fargs &a, const bool EXPORT)
{
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
    const float chroma_out = fminf(Yo.y, Yf.y);
    const float r_mix = a.beta_hue * Yo.y * Yo.z + (1.f - a.beta_hue) * chroma_out * Yf.z;
    const float g_mix = a.beta_hue * Yo.y * Yo.w + (1.f - a.beta_hue) * chroma_out * Yf.w;
    const float norm_mix = sqrtf(g_mix * g_mix + r_mix * r_mix);
    v4 Yref = Yo;
    Yref.z = (norm_mix > 1e-9f) ? r_mix / norm_mix : Yo.z;
    Yref.w = (norm_mix > 1e-9f) ? g_mix / norm_mix : Yo.w;
    Yf.y = chroma_out;
    res = gamut_mapping(Yf, Yref, a, 0.f, EXPORT);
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
    res = gamut_mapping(Yf, Yo, a, 0.f, EXPORT);
  }
  else if(MODE == MODE_SPLIT_V4)
  {
    const v4 po = RGB_tone_mapping_v4(pix_in, a);
    const v4 Yo = pipe_RGB_to_Ych(pix_in, a.input);
    v4 Yf = pipe_RGB_to_Ych(po, a.input);
    Yf.y = fminf(Yo.y, Yf.y);
    res = gamut_mapping(Yf, Yo, a, a.saturation, EXPORT);
  }
  else if(MODE == MODE_SPLIT_LEGACY)
    res = filmic_split_legacy(pix_in, a);
  else if(MODE == MODE_CHROMA_V1)
    res = filmic_chroma_v1(pix_in, a);
  else if(MODE == MODE_CHROMA_V2_V3)
    res = filmic_chroma_v2_v3(pix_in, a);
  else
  {
    const v4 po = norm_tone_mapping_v4(pix_in, a.preserve_color, a);
    const v4 Yo = pipe_RGB_to_Ych(pix_in, a.input);
    const v4 Yf = pipe_RGB_to_Ych(po, a.input);
    res = gamut_mapping(Yf, Yo, a, a.saturation, EXPORT);
  }
  return make_float4(res.x, res.y, res.z, res.w);
}

// host: dt_hip_filmicrgb_data_t -> kernel arguments (per-call matrix preparation included);
// returns DT_HIP_INVALID_ARG for a colour science outside 0..9
int filmicrgb_fill_args(const dt_hip_filmicrgb_data_t *d, fargs &a);

} // namespace ansel
