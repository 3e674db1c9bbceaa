// blend.hip -- the blend stage on gfx950 (all four blend colourspaces: RGB scene, RGB display, Lab, raw): opacity mask and
// blend operator in ONE pass.
//
// Reference: dt_develop_blend_process(), src/develop/blend.c:657-900, which runs after the process()
// of every blending-capable module (src/develop/pixelpipe_cpu.c:137-228): fill / build the mask
// (dt_develop_blendif_rgb_jzczhz_make_mask(), src/develop/blends/blendif_rgb_jzczhz.c:196-324),
// post-process it (_develop_blend_process_mask_tone_curve(), blend.c:626-655), then blend the
// module's input into its output (dt_develop_blendif_rgb_jzczhz_blend(), :878-960).  The reference
// makes one pass over the frame per active mask channel, one for the combination, one for the tone
// curve, copies the output and makes one more pass for the operator -- a 4-byte mask plane and a
// 16-byte copy through memory each time.  Without a spatial mask operation (feathering, blur, drawn
// shapes: refused here) every one of those steps is pointwise, so this kernel does them all on the
// pixel in registers: 16 B in + 16 B out read, 16 B written, 48 B per pixel.
//
// Arithmetic: the parametric channels Jz, Cz, hz go through two powf per LMS channel, one atan2f and
// one hypotf (src/common/colorspaces_inline_conversions.h:672-781); those are the glibc-exact ones
// of devmath.h.  The per-channel parameter table, the masking profile and exp2f / expf of the
// uniform parameters are prepared on the host with the host's libm, as the reference does.
#include "hip_common.h"
#include "devmath.h"

#include <math.h>

using namespace ansel;

namespace
{

#define PARAM_ITEMS 6 // DEVELOP_BLENDIF_PARAMETER_ITEMS
#define RGB_MASK 0x77FFu
#define LAB_MASK 0x3377u
#define GRAY_OUT 4

enum
{
  MODE_MULTIPLY = 0x04,
  MODE_AVERAGE = 0x05,
  MODE_ADD = 0x06,
  MODE_SUBTRACT = 0x07,
  MODE_DIFFERENCE = 0x08,
  MODE_LIGHTNESS = 0x10,
  MODE_CHROMATICITY = 0x11,
  MODE_DIFFERENCE2 = 0x17,
  MODE_RGB_R = 0x21,
  MODE_RGB_G = 0x22,
  MODE_RGB_B = 0x23,
  MODE_SUBTRACT_INVERSE = 0x25,
  MODE_DIVIDE = 0x26,
  MODE_DIVIDE_INVERSE = 0x27,
  MODE_GEOMETRIC_MEAN = 0x28,
  MODE_HARMONIC_MEAN = 0x29,
};

struct blend_args
{
  int owidth, oheight, iwidth, xoffs, yoffs;
  float constant; // the mask when it does not depend on the pixel
  float global_opacity, seed;
  int inclusive, inversed;
  unsigned blendif; // inclusive-combine inversion applied
  float parameters[PARAM_ITEMS * DT_HIP_BLENDIF_SIZE];
  float luma[3];
  float mT[3][4]; // masking profile: RGB -> XYZ D65, transposed
  int tone;
  float e, brightness, opacity;
  unsigned mode;
  int reverse;
  float p;
};

// _blendif_compute_factor(), blendif_rgb_jzczhz.c:42-73
__device__ __forceinline__ float compute_factor(const float value, const unsigned invert, const float *p)
{
  float factor;
  if(value <= p[0]) factor = 0.0f;
  else if(value < p[1]) factor = (value - p[0]) * p[4];
  else if(value <= p[2]) factor = 1.0f;
  else if(value < p[3]) factor = 1.0f - (value - p[2]) * p[5];
  else factor = 0.0f;
  return invert ? 1.0f - factor : factor;
}

// dt_ioppr_rgb_matrix_to_xyz() on a linear profile, dt_XYZ_2_JzAzBz(), dt_JzAzBz_2_JzCzhz()
__device__ __forceinline__ void rgb_to_JzCzhz(const float4 rgb, const float (&mT)[3][4], float (&JzCzhz)[3])
{
  const float b = 1.15f, g = 0.66f, c1 = 0.8359375f, c2 = 18.8515625f, c3 = 18.6875f, n = 0.159301758f,
              p = 134.034375f, d = -0.56f, d0 = 1.6295499532821566e-11f;
  const float M[3][3] = { { 0.41478972f, 0.579999f, 0.0146480f },
                          { -0.2015100f, 1.120649f, 0.0531008f },
                          { -0.0166008f, 0.264800f, 0.6684799f } };
  const float A_T[3][3] = { { 0.5f, 3.524000f, 0.199076f }, { 0.5f, -4.066708f, 1.096799f }, { 0.0f, 0.542708f, -1.295875f } };
  float D65[3];
#pragma unroll
  for(int c = 0; c < 3; c++)
  {
    float o = mT[0][c] * rgb.x;
    o = mT[1][c] * rgb.y + o;
    D65[c] = mT[2][c] * rgb.z + o;
  }
  float XYZ[3], LMS[3], Jab[3];
  XYZ[0] = b * D65[0] - (b - 1.0f) * D65[2];
  XYZ[1] = g * D65[1] - (g - 1.0f) * D65[0];
  XYZ[2] = D65[2];
#pragma unroll
  for(int i = 0; i < 3; i++)
  {
    LMS[i] = M[i][0] * XYZ[0] + M[i][1] * XYZ[1] + M[i][2] * XYZ[2];
    LMS[i] = ansel_math::powf_exact(fmaxf(LMS[i] / 10000.f, 0.0f), n);
    LMS[i] = ansel_math::powf_exact((c1 + c2 * LMS[i]) / (1.0f + c3 * LMS[i]), p);
  }
#pragma unroll
  for(int c = 0; c < 3; c++) Jab[c] = A_T[0][c] * LMS[0] + A_T[1][c] * LMS[1] + A_T[2][c] * LMS[2];
  Jab[0] = fmaxf(((1.0f + d) * Jab[0]) / (1.0f + d * Jab[0]) - d0, 0.f);
  const float var_H = ansel_math::atan2f_exact(Jab[2], Jab[1]) / (2.0f * 3.14159265358979324f);
  JzCzhz[0] = Jab[0];
  JzCzhz[1] = ansel_math::hypotf_exact(Jab[1], Jab[2]);
  JzCzhz[2] = var_H >= 0.0f ? var_H : 1.0f + var_H;
}

// _blendif_combine_channels(), blendif_rgb_jzczhz.c:151-194; OUT selects the output-side channels
__device__ __forceinline__ void rgb_to_hsl(const float (&RGB)[3], float (&HSL)[3]);

// HSL: the RGB (display) colourspace, whose channels 8..10 are H, S, L instead of Jz, Cz, hz
template <int OUT, bool HSL> __device__ __forceinline__ float combine_channels(const float4 px, float temp, const blend_args &a)
{
  const unsigned blendif = OUT ? a.blendif >> GRAY_OUT : a.blendif; // uniform
  const float *const params = a.parameters + (OUT ? PARAM_ITEMS * GRAY_OUT : 0);
  if(blendif & 1u)
  {
    const float value = a.luma[0] * px.x + a.luma[1] * px.y + a.luma[2] * px.z;
    temp *= compute_factor(value, (blendif >> 16) & 1u, params);
  }
  if(blendif & 2u) temp *= compute_factor(px.x, (blendif >> 16) & 2u, params + PARAM_ITEMS * 1);
  if(blendif & 4u) temp *= compute_factor(px.y, (blendif >> 16) & 4u, params + PARAM_ITEMS * 2);
  if(blendif & 8u) temp *= compute_factor(px.z, (blendif >> 16) & 8u, params + PARAM_ITEMS * 3);
  if(blendif & ((1u << 8) | (1u << 9) | (1u << 10)))
  {
    float JzCzhz[3];
    if(HSL)
    {
      const float rgb[3] = { px.x, px.y, px.z };
      rgb_to_hsl(rgb, JzCzhz);
    }
    else
      rgb_to_JzCzhz(px, a.mT, JzCzhz);
    float factor = 1.0f;
#pragma unroll
    for(unsigned i = 0; i < 3; i++)
      factor *= compute_factor(JzCzhz[i], (blendif >> 16) & (1u << (8 + i)), params + PARAM_ITEMS * (8 + i));
    temp *= factor;
  }
  return temp;
}

// _develop_blend_process_mask_tone_curve(), blend.c:626-655
__host__ __device__ __forceinline__ float tone_curve(const float m, const blend_args &a)
{
  const float mask_epsilon = 16 * 1.19209290e-7f;
  float x = m / a.opacity;
  x = 2.f * x - 1.f;
  if(1.f - a.brightness <= 0.f) x = m <= mask_epsilon ? -1.f : 1.f;
  else if(1.f + a.brightness <= 0.f) x = m >= 1.f - mask_epsilon ? 1.f : -1.f;
  else if(a.brightness > 0.f)
  {
    x = (x + a.brightness) / (1.f - a.brightness);
    x = fminf(x, 1.f);
  }
  else
  {
    x = (x + a.brightness) / (1.f + a.brightness);
    x = fmaxf(x, -1.f);
  }
  const float r = ((x * a.e / (1.f + (a.e - 1.f) * fabsf(x))) / 2.f + 0.5f) * a.opacity;
  return r > 1.f ? 1.f : (r < 0.f ? 0.f : r);
}

// C fmax() on a float promoted to double, converted back (exact): max with NaN returning the other
__device__ __forceinline__ float fmax_d(const float x, const float y) { return (float)fmax((double)x, (double)y); }

// the _blend_* row functions (blendif_rgb_jzczhz.c:328-585), one pixel: a = bottom, b = top layer
__device__ __forceinline__ float4 blend_pixel(const unsigned mode, const float4 a4, const float4 b4, const float p, const float lo)
{
  const float a[3] = { a4.x, a4.y, a4.z }, b[3] = { b4.x, b4.y, b4.z };
  float o[3];
  switch(mode)
  {
    case MODE_MULTIPLY:
#pragma unroll
      for(int k = 0; k < 3; k++) o[k] = a[k] * (1.0f - lo) + (a[k] * b[k] * p) * lo;
      break;
    case MODE_AVERAGE:
#pragma unroll
      for(int k = 0; k < 3; k++) o[k] = a[k] * (1.0f - lo) + (a[k] + b[k]) / 2.0f * lo;
      break;
    case MODE_ADD:
#pragma unroll
      for(int k = 0; k < 3; k++) o[k] = a[k] * (1.0f - lo) + (a[k] + p * b[k]) * lo;
      break;
    case MODE_SUBTRACT:
#pragma unroll
      for(int k = 0; k < 3; k++) o[k] = a[k] * (1.0f - lo) + fmaxf(a[k] - p * b[k], 0.0f) * lo;
      break;
    case MODE_SUBTRACT_INVERSE:
#pragma unroll
      for(int k = 0; k < 3; k++) o[k] = a[k] * (1.0f - lo) + fmaxf(b[k] - p * a[k], 0.0f) * lo;
      break;
    case MODE_DIFFERENCE:
    case MODE_DIFFERENCE2:
#pragma unroll
      for(int k = 0; k < 3; k++) o[k] = a[k] * (1.0f - lo) + fabsf(a[k] - b[k]) * lo;
      break;
    case MODE_DIVIDE:
#pragma unroll
      for(int k = 0; k < 3; k++) o[k] = a[k] * (1.0f - lo) + a[k] / fmaxf(p * b[k], 1e-6f) * lo;
      break;
    case MODE_DIVIDE_INVERSE:
#pragma unroll
      for(int k = 0; k < 3; k++) o[k] = a[k] * (1.0f - lo) + b[k] / fmaxf(p * a[k], 1e-6f) * lo;
      break;
    case MODE_GEOMETRIC_MEAN:
#pragma unroll
      for(int k = 0; k < 3; k++) o[k] = a[k] * (1.0f - lo) + sqrtf(fmax_d(a[k] * b[k], 0.0f)) * lo;
      break;
    case MODE_HARMONIC_MEAN:
#pragma unroll
      for(int k = 0; k < 3; k++)
        o[k] = a[k] * (1.0f - lo) + 2.0f * a[k] * b[k] / (fmaxf(a[k], 5e-7f) + fmaxf(b[k], 5e-7f)) * lo;
      break;
    case MODE_CHROMATICITY:
    case MODE_LIGHTNESS:
    {
      const float norm_a = fmax_d(sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), 1e-6f);
      const float norm_b = fmax_d(sqrtf(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]), 1e-6f);
      if(mode == MODE_CHROMATICITY)
      {
#pragma unroll
        for(int k = 0; k < 3; k++) o[k] = a[k] * (1.0f - lo) + b[k] * norm_a / norm_b * lo;
      }
      else
      {
#pragma unroll
        for(int k = 0; k < 3; k++) o[k] = a[k] * (1.0f - lo) + a[k] * norm_b / norm_a * lo;
      }
      break;
    }
    case MODE_RGB_R:
      o[0] = a[0] * (1.0f - lo) + p * b[0] * lo;
      o[1] = a[1];
      o[2] = a[2];
      break;
    case MODE_RGB_G:
      o[0] = a[0];
      o[1] = a[1] * (1.0f - lo) + p * b[1] * lo;
      o[2] = a[2];
      break;
    case MODE_RGB_B:
      o[0] = a[0];
      o[1] = a[1];
      o[2] = a[2] * (1.0f - lo) + p * b[2] * lo;
      break;
    default: // normal
#pragma unroll
      for(int k = 0; k < 3; k++) o[k] = a[k] * (1.0f - lo) + b[k] * lo;
      break;
  }
  return make_float4(o[0], o[1], o[2], lo);
}


// ---- Lab (src/develop/blends/blendif_lab.c) -----------------------------------------------------
enum
{
  LAB_LIGHTEN = 0x02, LAB_DARKEN = 0x03, LAB_MULTIPLY = 0x04, LAB_AVERAGE = 0x05, LAB_ADD = 0x06, LAB_SUBTRACT = 0x07,
  LAB_DIFFERENCE = 0x08, LAB_SCREEN = 0x09, LAB_OVERLAY = 0x0A, LAB_SOFTLIGHT = 0x0B, LAB_HARDLIGHT = 0x0C,
  LAB_VIVIDLIGHT = 0x0D, LAB_LINEARLIGHT = 0x0E, LAB_PINLIGHT = 0x0F, LAB_LIGHTNESS = 0x10, LAB_CHROMATICITY = 0x11,
  LAB_HUE = 0x12, LAB_COLOR = 0x13, LAB_COLORADJUST = 0x16, LAB_DIFFERENCE2 = 0x17, LAB_BOUNDED = 0x19,
  LAB_LAB_LIGHTNESS = 0x1A, LAB_LAB_COLOR = 0x1B, LAB_LAB_L = 0x1E, LAB_LAB_A = 0x1F, LAB_LAB_B = 0x20,
};

// _blendif_combine_channels() of blendif_lab.c:139-173: L / 100, a / 256, b / 256, and C, h of dt_Lab_2_LCH()
// (src/common/colorspaces_inline_conversions.h:594-606)
template <int OUT> __device__ __forceinline__ float combine_channels_lab(const float4 px, float temp, const blend_args &a)
{
  const unsigned blendif = OUT ? a.blendif >> GRAY_OUT : a.blendif; // uniform
  const float *const params = a.parameters + (OUT ? PARAM_ITEMS * GRAY_OUT : 0);
  if(blendif & 1u) temp *= compute_factor(px.x / 100.0f, (blendif >> 16) & 1u, params);
  if(blendif & 2u) temp *= compute_factor(px.y / 256.0f, (blendif >> 16) & 2u, params + PARAM_ITEMS * 1);
  if(blendif & 4u) temp *= compute_factor(px.z / 256.0f, (blendif >> 16) & 4u, params + PARAM_ITEMS * 2);
  if(blendif & ((1u << 8) | (1u << 9)))
  {
    const float c_scale = 1.0f / (128.0f * 1.41421354f); // 1.0f / (128.0f * sqrtf(2.0f))
    float var_H = ansel_math::atan2f_exact(px.z, px.y);
    if(var_H > 0.0f) var_H = var_H / (2.0f * 3.14159265358979324f);
    else var_H = 1.0f - fabsf(var_H) / (2.0f * 3.14159265358979324f);
    const float C = ansel_math::hypotf_exact(px.y, px.z);
    float factor = 1.0f;
    factor *= compute_factor(C * c_scale, (blendif >> 16) & (1u << 8), params + PARAM_ITEMS * 8);
    factor *= compute_factor(var_H, (blendif >> 16) & (1u << 9), params + PARAM_ITEMS * 9);
    temp *= factor;
  }
  return temp;
}

__device__ __forceinline__ float CL(const float x, const float lo, const float hi) { return fminf(fmaxf(x, lo), hi); } // _CLAMP()

// dt_Lab_2_LCH() / dt_LCH_2_Lab(), src/common/colorspaces_inline_conversions.h:594-620
__device__ __forceinline__ void lab_to_lch(const float (&Lab)[3], float (&LCH)[3])
{
  float var_H = ansel_math::atan2f_exact(Lab[2], Lab[1]);
  if(var_H > 0.0f) var_H = var_H / (2.0f * 3.14159265358979324f);
  else var_H = 1.0f - fabsf(var_H) / (2.0f * 3.14159265358979324f);
  LCH[0] = Lab[0];
  LCH[1] = ansel_math::hypotf_exact(Lab[1], Lab[2]);
  LCH[2] = var_H;
}

__device__ __forceinline__ void lch_to_lab(const float (&LCH)[3], float (&Lab)[3])
{
  Lab[0] = LCH[0];
  Lab[1] = ansel_math::cosf_exact(2.0f * 3.14159265358979324f * LCH[2]) * LCH[1];
  Lab[2] = ansel_math::sinf_exact(2.0f * 3.14159265358979324f * LCH[2]) * LCH[1];
}

// the _blend_* row functions of blendif_lab.c:320-1068, one pixel: a = bottom, b = top layer
__device__ __forceinline__ float4 blend_pixel_lab(const unsigned mode, const float4 a4, const float4 b4, const float lo)
{
  const float min[3] = { 0.0f, -1.0f, -1.0f }, max[3] = { 1.0f, 1.0f, 1.0f };
  const float scale[3] = { 1 / 100.0f, 1 / 128.0f, 1 / 128.0f }, rescale[3] = { 100.0f, 128.0f, 128.0f };
  const float a[3] = { a4.x, a4.y, a4.z }, b[3] = { b4.x, b4.y, b4.z };
  float ta[3], tb[3];
#pragma unroll
  for(int c = 0; c < 3; c++)
  {
    ta[c] = a[c] * scale[c];
    tb[c] = b[c] * scale[c];
  }
  const float lo2 = lo * lo;
  // the lightness-based operators work on L shifted into [0, lmax]
  const float lmin = 0.0f, lmax = max[0] + fabsf(min[0]);
  const float la = CL(ta[0] + fabsf(min[0]), lmin, lmax), lb = CL(tb[0] + fabsf(min[0]), lmin, lmax);
  const float halfmax = lmax / 2.0f, doublemax = lmax * 2.0f;
  const float f = fmaxf(ta[0], 0.01f);
  bool chroma_follows = false; // a, b scaled by the lightness ratio with opacity lo2 (overlay .. linearlight)
  switch(mode)
  {
    case LAB_BOUNDED:
#pragma unroll
      for(int x = 0; x < 3; x++) tb[x] = CL(ta[x] * (1.0f - lo) + tb[x] * lo, min[x], max[x]);
      break;
    case LAB_LIGHTEN:
    case LAB_DARKEN:
    {
      const float pick = mode == LAB_LIGHTEN ? (ta[0] > tb[0] ? ta[0] : tb[0]) : (ta[0] < tb[0] ? ta[0] : tb[0]);
      tb[0] = CL(ta[0] * (1.0f - lo) + pick * lo, min[0], max[0]);
      tb[1] = CL(ta[1] * (1.0f - fabsf(tb[0] - ta[0])) + 0.5f * (ta[1] + tb[1]) * fabsf(tb[0] - ta[0]), min[1], max[1]);
      tb[2] = CL(ta[2] * (1.0f - fabsf(tb[0] - ta[0])) + 0.5f * (ta[2] + tb[2]) * fabsf(tb[0] - ta[0]), min[2], max[2]);
      break;
    }
    case LAB_MULTIPLY:
      tb[0] = CL(ta[0] * (1.0f - lo) + (ta[0] * tb[0]) * lo, min[0], max[0]);
      tb[1] = CL(ta[1] * (1.0f - lo) + (ta[1] + tb[1]) * tb[0] / f * lo, min[1], max[1]);
      tb[2] = CL(ta[2] * (1.0f - lo) + (ta[2] + tb[2]) * tb[0] / f * lo, min[2], max[2]);
      break;
    case LAB_AVERAGE:
#pragma unroll
      for(int x = 0; x < 3; x++) tb[x] = CL(ta[x] * (1.0f - lo) + (ta[x] + tb[x]) / 2.0f * lo, min[x], max[x]);
      break;
    case LAB_ADD:
#pragma unroll
      for(int x = 0; x < 3; x++) tb[x] = CL(ta[x] * (1.0f - lo) + (ta[x] + tb[x]) * lo, min[x], max[x]);
      break;
    case LAB_SUBTRACT:
#pragma unroll
      for(int x = 0; x < 3; x++)
        tb[x] = CL(ta[x] * (1.0f - lo) + ((tb[x] + ta[x]) - (fabsf(min[x] + max[x]))) * lo, min[x], max[x]);
      break;
    case LAB_DIFFERENCE:
#pragma unroll
      for(int x = 0; x < 3; x++)
      {
        const float xmax = max[x] + fabsf(min[x]);
        const float xa = CL(ta[x] + fabsf(min[x]), lmin, xmax), xb = CL(tb[x] + fabsf(min[x]), lmin, xmax);
        tb[x] = CL(xa * (1.0f - lo) + fabsf(xa - xb) * lo, lmin, xmax) - fabsf(min[x]);
      }
      break;
    case LAB_DIFFERENCE2:
#pragma unroll
      for(int x = 0; x < 3; x++) tb[x] = fabsf(ta[x] - tb[x]) / fabsf(max[x] - min[x]);
      tb[0] = fmaxf(tb[0], fmaxf(tb[1], tb[2]));
      tb[0] = CL(ta[0] * (1.0f - lo) + tb[0] * lo, min[0], max[0]);
      tb[1] = 0.0f;
      tb[2] = 0.0f;
      break;
    case LAB_SCREEN:
      tb[0] = CL(la * (1.0f - lo) + ((lmax - (lmax - la) * (lmax - lb))) * lo, lmin, lmax) - fabsf(min[0]);
      tb[1] = CL(ta[1] * (1.0f - lo) + 0.5f * (ta[1] + tb[1]) * tb[0] / f * lo, min[1], max[1]);
      tb[2] = CL(ta[2] * (1.0f - lo) + 0.5f * (ta[2] + tb[2]) * tb[0] / f * lo, min[2], max[2]);
      break;
    case LAB_OVERLAY:
      tb[0] = CL(la * (1.0f - lo2)
                     + (la > halfmax ? lmax - (lmax - doublemax * (la - halfmax)) * (lmax - lb) : (doublemax * la) * lb) * lo2,
                 lmin, lmax)
              - fabsf(min[0]);
      chroma_follows = true;
      break;
    case LAB_SOFTLIGHT:
      tb[0] = CL(la * (1.0f - lo2) + (lb > halfmax ? lmax - (lmax - la) * (lmax - (lb - halfmax)) : la * (lb + halfmax)) * lo2,
                 lmin, lmax)
              - fabsf(min[0]);
      chroma_follows = true;
      break;
    case LAB_HARDLIGHT:
      tb[0] = CL(la * (1.0f - lo2)
                     + (lb > halfmax ? lmax - (lmax - doublemax * (la - halfmax)) * (lmax - lb) : doublemax * la * lb) * lo2,
                 lmin, lmax)
              - fabsf(min[0]);
      chroma_follows = true;
      break;
    case LAB_VIVIDLIGHT:
      tb[0] = CL(la * (1.0f - lo2)
                     + (lb > halfmax ? (lb >= lmax ? lmax : la / (doublemax * (lmax - lb)))
                                     : (lb <= lmin ? lmin : lmax - (lmax - la) / (doublemax * lb)))
                           * lo2,
                 lmin, lmax)
              - fabsf(min[0]);
      chroma_follows = true;
      break;
    case LAB_LINEARLIGHT:
      tb[0] = CL(la * (1.0f - lo2) + (la + doublemax * lb - lmax) * lo2, lmin, lmax) - fabsf(min[0]);
      chroma_follows = true;
      break;
    case LAB_PINLIGHT:
      tb[0] = CL(la * (1.0f - lo2) + (lb > halfmax ? fmaxf(la, doublemax * (lb - halfmax)) : fminf(la, doublemax * lb)) * lo2,
                 lmin, lmax)
              - fabsf(min[0]);
      tb[1] = CL(ta[1], min[1], max[1]);
      tb[2] = CL(ta[2], min[2], max[2]);
      break;
    case LAB_LIGHTNESS:
      tb[0] = CL(ta[0] * (1.0f - lo) + tb[0] * lo, min[0], max[0]);
      tb[1] = CL(ta[1], min[1], max[1]);
      tb[2] = CL(ta[2], min[2], max[2]);
      break;
    case LAB_CHROMATICITY:
    case LAB_HUE:
    case LAB_COLOR:
    case LAB_COLORADJUST:
    {
      // blendif_lab.c:843-975: through LCh, hue blended along the shortest way round the colour circle; fmodf is
      // exact in any correct implementation
      float tta[3], ttb[3];
#pragma unroll
      for(int x = 0; x < 3; x++)
      {
        ta[x] = CL(ta[x], min[x], max[x]);
        tb[x] = CL(tb[x], min[x], max[x]);
      }
      lab_to_lch(ta, tta);
      lab_to_lch(tb, ttb);
      if(mode != LAB_COLORADJUST) ttb[0] = tta[0];
      const float chroma = (tta[1] * (1.0f - lo)) + ttb[1] * lo;
      const float d = fabsf(tta[2] - ttb[2]);
      const float sh = d > 0.5f ? -lo * (1.0f - d) / d : lo;
      const float hue = fmodf((tta[2] * (1.0f - sh)) + ttb[2] * sh + 1.0f, 1.0f);
      ttb[1] = mode == LAB_HUE ? tta[1] : chroma;
      ttb[2] = mode == LAB_CHROMATICITY ? tta[2] : hue;
      lch_to_lab(ttb, tb);
#pragma unroll
      for(int x = 0; x < 3; x++) tb[x] = CL(tb[x], min[x], max[x]);
      break;
    }
    case LAB_LAB_LIGHTNESS:
    case LAB_LAB_L:
      tb[0] = ta[0] * (1.0f - lo) + tb[0] * lo;
      tb[1] = ta[1];
      tb[2] = ta[2];
      break;
    case LAB_LAB_A:
      tb[0] = ta[0];
      tb[1] = ta[1] * (1.0f - lo) + tb[1] * lo;
      tb[2] = ta[2];
      break;
    case LAB_LAB_B:
      tb[0] = ta[0];
      tb[1] = ta[1];
      tb[2] = ta[2] * (1.0f - lo) + tb[2] * lo;
      break;
    case LAB_LAB_COLOR:
      tb[0] = ta[0];
      tb[1] = ta[1] * (1.0f - lo) + tb[1] * lo;
      tb[2] = ta[2] * (1.0f - lo) + tb[2] * lo;
      break;
    default: // normal, unbounded
#pragma unroll
      for(int x = 0; x < 3; x++) tb[x] = ta[x] * (1.0f - lo) + tb[x] * lo;
      break;
  }
  if(chroma_follows)
  {
    tb[1] = CL(ta[1] * (1.0f - lo2) + (ta[1] + tb[1]) * tb[0] / f * lo2, min[1], max[1]);
    tb[2] = CL(ta[2] * (1.0f - lo2) + (ta[2] + tb[2]) * tb[0] / f * lo2, min[2], max[2]);
  }
  return make_float4(tb[0] * rescale[0], tb[1] * rescale[1], tb[2] * rescale[2], lo);
}


// ---- raw, one channel (src/develop/blends/blendif_raw.c:66-288) -----------------------------------------
__device__ __forceinline__ float clamp01(const float x) { return fminf(fmaxf(x, 0.0f), 1.0f); } // clamp_simd()

__device__ __forceinline__ float blend_value_raw(const unsigned mode, const float a, const float b, const float lo)
{
  const float lo2 = lo * lo;
  const float la = clamp01(a), lb = clamp01(b);
  switch(mode)
  {
    case LAB_BOUNDED: return clamp01(a * (1.0f - lo) + b * lo);
    case LAB_LIGHTEN: return clamp01(a * (1.0f - lo) + fmaxf(a, b) * lo);
    case LAB_DARKEN: return clamp01(a * (1.0f - lo) + fminf(a, b) * lo);
    case LAB_MULTIPLY: return clamp01(a * (1.0f - lo) + (a * b) * lo);
    case LAB_AVERAGE: return clamp01(a * (1.0f - lo) + (a + b) / 2.0f * lo);
    case LAB_ADD: return clamp01(a * (1.0f - lo) + (a + b) * lo);
    case LAB_SUBTRACT: return clamp01(a * (1.0f - lo) + ((b + a) - 1.0f) * lo);
    case LAB_DIFFERENCE:
    case LAB_DIFFERENCE2: return clamp01(a * (1.0f - lo) + fabsf(a - b) * lo);
    case LAB_SCREEN: return clamp01(la * (1.0f - lo) + (1.0f - (1.0f - la) * (1.0f - lb)) * lo);
    case LAB_OVERLAY:
      return clamp01(la * (1.0f - lo2) + (la > 0.5f ? 1.0f - (1.0f - 2.0f * (la - 0.5f)) * (1.0f - lb) : 2.0f * la * lb) * lo2);
    case LAB_SOFTLIGHT:
      return clamp01(la * (1.0f - lo2) + (lb > 0.5f ? 1.0f - (1.0f - la) * (1.0f - (lb - 0.5f)) : la * (lb + 0.5f)) * lo2);
    case LAB_HARDLIGHT:
      return clamp01(la * (1.0f - lo2) + (lb > 0.5f ? 1.0f - (1.0f - 2.0f * (la - 0.5f)) * (1.0f - lb) : 2.0f * la * lb) * lo2);
    case LAB_VIVIDLIGHT:
      return clamp01(la * (1.0f - lo2)
                     + (lb > 0.5f ? (lb >= 1.0f ? 1.0f : la / (2.0f * (1.0f - lb)))
                                  : (lb <= 0.0f ? 0.0f : 1.0f - (1.0f - la) / (2.0f * lb)))
                           * lo2);
    case LAB_LINEARLIGHT: return clamp01(la * (1.0f - lo2) + (la + 2.0f * lb - 1.0f) * lo2);
    case LAB_PINLIGHT:
      return clamp01(la * (1.0f - lo2) + (lb > 0.5f ? fmaxf(la, 2.0f * (lb - 0.5f)) : fminf(la, 2.0f * lb)) * lo2);
    default: return a * (1.0f - lo) + b * lo; // normal, unbounded
  }
}

// the mask of the raw colourspace never depends on the photosite (blendif_raw.c:36-62)
__global__ __launch_bounds__(256) void blend_raw_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                        const float *__restrict__ plane, const blend_args q)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= (size_t)q.owidth * q.oheight) return;
  const int y = (int)(k / q.owidth), x = (int)(k - (size_t)y * q.owidth);
  const float a = in[(size_t)(y + q.yoffs) * q.iwidth + q.xoffs + x];
  const float b = out[k];
  float m = plane ? plane[k] : q.constant; // the blurred mask plane, or the same value everywhere
  if(plane && q.tone) m = tone_curve(m, q);
  out[k] = q.reverse ? blend_value_raw(q.mode, b, a, m) : blend_value_raw(q.mode, a, b, m);
}


// ---- RGB (display): src/develop/blends/blendif_rgb_hsl.c; HSL / HSV conversions of
//      src/common/colorspaces_inline_conversions.h:421-565 -----------------------------------------------
__device__ __forceinline__ float rgb_hue(const float (&RGB)[3], const float max, const float delta) // _dt_RGB_2_Hue()
{
  float hue;
  if(RGB[0] == max) hue = (RGB[1] - RGB[2]) / delta;
  else if(RGB[1] == max) hue = 2.0f + (RGB[2] - RGB[0]) / delta;
  else hue = 4.0f + (RGB[0] - RGB[1]) / delta;
  hue /= 6.0f;
  if(hue < 0.0f) hue += 1.0f;
  if(hue > 1.0f) hue -= 1.0f;
  return hue;
}

__device__ __forceinline__ void hue_to_rgb(float (&RGB)[3], const float H, const float C, const float min) // _dt_Hue_2_RGB()
{
  const float h = H * 6.0f;
  const float i = floorf(h);
  const float f = h - i;
  const float fc = f * C;
  const float top = C + min;
  const float inc = fc + min;
  const float dec = top - fc;
  // (size_t)i of the reference on x86: anything but 0 .. 4 (negative, >= 5, NaN) takes the last branch
  const int sector = (i >= 0.0f && i < 5.0f) ? (int)i : 5;
  RGB[0] = sector == 0 || sector == 5 ? top : (sector == 1 ? dec : (sector == 4 ? inc : min));
  RGB[1] = sector == 1 || sector == 2 ? top : (sector == 0 ? inc : (sector == 3 ? dec : min));
  RGB[2] = sector == 3 || sector == 4 ? top : (sector == 2 ? inc : (sector == 5 ? dec : min));
}

__device__ __forceinline__ void rgb_to_hsl(const float (&RGB)[3], float (&HSL)[3]) // dt_RGB_2_HSL()
{
  const float min = fminf(RGB[0], fminf(RGB[1], RGB[2]));
  const float max = fmaxf(RGB[0], fmaxf(RGB[1], RGB[2]));
  const float delta = max - min;
  const float L = (max + min) / 2.0f;
  float H = 0.0f, S = 0.0f;
  if(fabsf(max) > 1e-6f && fabsf(delta) > 1e-6f)
  {
    if(L < 0.5f) S = delta / (max + min);
    else S = delta / (2.0f - max - min);
    H = rgb_hue(RGB, max, delta);
  }
  HSL[0] = H;
  HSL[1] = S;
  HSL[2] = L;
}

__device__ __forceinline__ void hsl_to_rgb(const float (&HSL)[3], float (&RGB)[3]) // dt_HSL_2_RGB()
{
  const float L = HSL[2];
  float C;
  if(L < 0.5f) C = L * HSL[1];
  else C = (1.0f - L) * HSL[1];
  const float m = L - C;
  hue_to_rgb(RGB, HSL[0], 2.0f * C, m);
}

__device__ __forceinline__ void rgb_to_hsv(const float (&RGB)[3], float (&HSV)[3]) // dt_RGB_2_HSV()
{
  const float min = fminf(RGB[0], fminf(RGB[1], RGB[2]));
  const float max = fmaxf(RGB[0], fmaxf(RGB[1], RGB[2]));
  const float delta = max - min;
  float S = 0.0f, H = 0.0f;
  if(fabsf(max) > 1e-6f && fabsf(delta) > 1e-6f)
  {
    S = delta / max;
    H = rgb_hue(RGB, max, delta);
  }
  HSV[0] = H;
  HSV[1] = S;
  HSV[2] = max;
}

__device__ __forceinline__ void hsv_to_rgb(const float (&HSV)[3], float (&RGB)[3]) // dt_HSV_2_RGB()
{
  const float C = HSV[1] * HSV[2];
  const float m = HSV[2] - C;
  hue_to_rgb(RGB, HSV[0], C, m);
}

enum { DSP_HSV_VALUE = 0x1C, DSP_HSV_COLOR = 0x1D };

__device__ __forceinline__ float blend_value_raw(const unsigned mode, const float a, const float b, const float lo);

// the _blend_* row functions of blendif_rgb_hsl.c:348-913, one pixel
__device__ __forceinline__ float4 blend_pixel_display(const unsigned mode, const float4 a4, const float4 b4, const float lo)
{
  const float a[3] = { a4.x, a4.y, a4.z }, b[3] = { b4.x, b4.y, b4.z };
  float out[3];
  switch(mode)
  {
    case LAB_LIGHTNESS:
    case LAB_CHROMATICITY:
    case LAB_HUE:
    case LAB_COLOR:
    case LAB_COLORADJUST:
    {
      float ta[3], tb[3], tta[3], ttb[3];
#pragma unroll
      for(int k = 0; k < 3; k++)
      {
        ta[k] = fminf(fmaxf(a[k], 0.0f), 1.0f);
        tb[k] = fminf(fmaxf(b[k], 0.0f), 1.0f);
      }
      rgb_to_hsl(ta, tta);
      rgb_to_hsl(tb, ttb);
      const float d = fabsf(tta[0] - ttb[0]);
      const float sh = d > 0.5f ? -lo * (1.0f - d) / d : lo;
      const float hue = fmodf((tta[0] * (1.0f - sh)) + ttb[0] * sh + 1.0f, 1.0f);
      const float sat = (tta[1] * (1.0f - lo)) + ttb[1] * lo;
      const float lig = (tta[2] * (1.0f - lo)) + ttb[2] * lo;
      if(mode == LAB_LIGHTNESS) { ttb[0] = tta[0]; ttb[1] = tta[1]; ttb[2] = lig; }
      else if(mode == LAB_CHROMATICITY) { ttb[0] = tta[0]; ttb[1] = sat; ttb[2] = tta[2]; }
      else if(mode == LAB_HUE) { ttb[0] = hue; ttb[1] = tta[1]; ttb[2] = tta[2]; }
      else if(mode == LAB_COLOR) { ttb[0] = hue; ttb[1] = sat; ttb[2] = tta[2]; }
      else { ttb[0] = hue; ttb[1] = sat; } // coloradjust: lightness of the module output
      hsl_to_rgb(ttb, out);
#pragma unroll
      for(int k = 0; k < 3; k++) out[k] = fminf(fmaxf(out[k], 0.0f), 1.0f);
      break;
    }
    case DSP_HSV_VALUE:
    {
      float ta[3], tb[3];
      rgb_to_hsv(a, ta);
      rgb_to_hsv(b, tb);
      tb[0] = ta[0];
      tb[1] = ta[1];
      tb[2] = ta[2] * (1.0f - lo) + tb[2] * lo;
      hsv_to_rgb(tb, out);
      break;
    }
    case DSP_HSV_COLOR:
    {
      float ta[3], tb[3];
      rgb_to_hsv(a, ta);
      rgb_to_hsv(b, tb);
      const float xa = ta[1] * ansel_math::cosf_exact(2.0f * 3.14159265358979324f * ta[0]);
      const float ya = ta[1] * ansel_math::sinf_exact(2.0f * 3.14159265358979324f * ta[0]);
      const float xb = tb[1] * ansel_math::cosf_exact(2.0f * 3.14159265358979324f * tb[0]);
      const float yb = tb[1] * ansel_math::sinf_exact(2.0f * 3.14159265358979324f * tb[0]);
      const float xc = xa * (1.0f - lo) + xb * lo;
      const float yc = ya * (1.0f - lo) + yb * lo;
      tb[0] = ansel_math::atan2f_exact(yc, xc) / (2.0f * 3.14159265358979324f);
      if(tb[0] < 0.0f) tb[0] += 1.0f;
      tb[1] = sqrtf(xc * xc + yc * yc);
      tb[2] = ta[2];
      hsv_to_rgb(tb, out);
      break;
    }
    case MODE_RGB_R:
      out[0] = a[0] * (1.0f - lo) + b[0] * lo;
      out[1] = a[1];
      out[2] = a[2];
      break;
    case MODE_RGB_G:
      out[0] = a[0];
      out[1] = a[1] * (1.0f - lo) + b[1] * lo;
      out[2] = a[2];
      break;
    case MODE_RGB_B:
      out[0] = a[0];
      out[1] = a[1];
      out[2] = a[2] * (1.0f - lo) + b[2] * lo;
      break;
    default: // the per-channel operators: the formulas of the one-channel colourspace on each of R, G, B
#pragma unroll
      for(int k = 0; k < 3; k++) out[k] = blend_value_raw(mode, a[k], b[k], lo);
      break;
  }
  return make_float4(out[0], out[1], out[2], lo);
}

// ---- mask blur: dt_gaussian_blur(), src/pixel/gaussian.c:176-326, one channel, order 0, clamped to [0, 1] on the way
//      in (CLAMPF, src/math/math.h:91).  Two recursive passes, each a forward and a backward second-order recurrence:
//      sequential along the pass direction, one lane per column (vertical) or per row (horizontal).
struct gauss_args
{
  int width, height;
  float a0, a1, a2, a3, b1, b2, coefp, coefn;
};

__device__ __forceinline__ float clampf01(const float a) { return a >= 0.0f ? (a <= 1.0f ? a : 1.0f) : 0.0f; }

// The recursive gaussian of dt_gaussian_blur() (src/pixel/gaussian.c:133-326, one channel) down the columns of a plane:
// a lane walks the `n` samples of its column, `stride` apart -- consecutive lanes = consecutive columns, coalesced at
// every step.  The causal and the anticausal filter of a column run in two different waves (threads 0..63 / 64..127 of
// a workgroup) into two planes; the kernel that transposes the result for the next pass adds them (forward + backward:
// the reference's `+=`).  Nothing a sample's fetch needs depends on the recurrence, so a lane fetches GAUSS_U samples at
// once and then runs them through the filter: one memory round trip per GAUSS_U rows instead of one per row.  The pass
// along the rows is the same kernel on the transposed plane (one lane per ROW of a row-major plane touches 64 cache
// lines per step).  24 MP, sigma 10: 5.6 ms as two one-lane-per-line kernels with a fetch per step, 1.3 ms this way.
#define GAUSS_U 32
__global__ __launch_bounds__(128) void gauss_vertical(const float *__restrict__ src, float *__restrict__ fwd, float *__restrict__ bwd,
                                                      const int width, const int n, const gauss_args g)
{
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  if(i >= width) return;
  const size_t stride = (size_t)width;
  src += i;
  if(threadIdx.x < 64)
  {
    float *const dst = fwd + i;
    float xp = clampf01(src[0]), yb = xp * g.coefp, yp = yb;
    for(int j0 = 0; j0 < n; j0 += GAUSS_U)
    {
      float x[GAUSS_U];
#pragma unroll
      for(int k = 0; k < GAUSS_U; k++) x[k] = j0 + k < n ? src[(size_t)(j0 + k) * stride] : 0.0f;
#pragma unroll
      for(int k = 0; k < GAUSS_U; k++)
        if(j0 + k < n)
        {
          const float xc = clampf01(x[k]);
          const float yc = (g.a0 * xc) + (g.a1 * xp) - (g.b1 * yp) - (g.b2 * yb);
          dst[(size_t)(j0 + k) * stride] = yc;
          xp = xc;
          yb = yp;
          yp = yc;
        }
    }
  }
  else
  {
    float *const dst = bwd + i;
    float xn = clampf01(src[(size_t)(n - 1) * stride]), xa = xn, yn = xn * g.coefn, ya = yn;
    for(int j0 = n - 1; j0 > -1; j0 -= GAUSS_U)
    {
      float x[GAUSS_U];
#pragma unroll
      for(int k = 0; k < GAUSS_U; k++) x[k] = j0 - k > -1 ? src[(size_t)(j0 - k) * stride] : 0.0f;
#pragma unroll
      for(int k = 0; k < GAUSS_U; k++)
        if(j0 - k > -1)
        {
          const float xc = clampf01(x[k]);
          const float yc = (g.a2 * xn) + (g.a3 * xa) - (g.b1 * yn) - (g.b2 * ya);
          xa = xn;
          xn = xc;
          ya = yn;
          yn = yc;
          dst[(size_t)(j0 - k) * stride] = yc;
        }
    }
  }
}

// dst (width rows of height) = (fwd + bwd)^T, 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void gauss_sum_transpose(const float *__restrict__ fwd, const float *__restrict__ bwd,
                                                           float *__restrict__ dst, const int width, const int height)
{
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
#pragma unroll
  for(int r = ty; r < 32; r += 8)
    if(x0 + tx < width && y0 + r < height)
    {
      const size_t k = (size_t)(y0 + r) * width + x0 + tx;
      tile[r][tx] = fwd[k] + bwd[k];
    }
  __syncthreads();
#pragma unroll
  for(int r = ty; r < 32; r += 8)
    if(y0 + tx < height && x0 + r < width) dst[(size_t)(x0 + r) * height + y0 + tx] = tile[tx][r];
}

__global__ __launch_bounds__(256) void fill_plane(float *__restrict__ p, const size_t n, const float v)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(k < n) p[k] = v;
}

// the parametric mask of a pixel, before the post operations (make_mask(), blendif_*.c)
template <int CS>
__device__ __forceinline__ float parametric_mask(const float4 pa, const float4 pb, const blend_args &a, const float seed)
{
  float temp = 1.0f;
  if(CS == DT_HIP_BLEND_CS_LAB)
  {
    temp = combine_channels_lab<0>(pa, temp, a);
    temp = combine_channels_lab<1>(pb, temp, a);
  }
  else
  {
    temp = combine_channels<0, CS == DT_HIP_BLEND_CS_RGB_DISPLAY>(pa, temp, a);
    temp = combine_channels<1, CS == DT_HIP_BLEND_CS_RGB_DISPLAY>(pb, temp, a);
  }
  if(a.inclusive) return a.inversed ? a.global_opacity * (1.0f - seed) * temp : a.global_opacity * (1.0f - (1.0f - seed) * temp);
  return a.inversed ? a.global_opacity * (1.0f - seed * temp) : a.global_opacity * seed * temp;
}

// The mask of a blend with a host-rendered form mask (drawn forms, a raster mask, the details refinement: blend.c:740-790,
// uploaded as one plane as in blend.c:1278-1325), before the post operations.  kind 4: a raster mask alone, form * opacity
// (blend.c:740-745); 3: make_mask() not conditional (blendif_rgb_jzczhz.c:228-240); 2: every conditional channel, the form
// mask in the place of the constant one
template <int CS>
__global__ __launch_bounds__(256) void form_mask_kernel(const float4 *__restrict__ in, const float4 *__restrict__ out,
                                                        const float *__restrict__ form, float *__restrict__ plane,
                                                        const blend_args a, const int kind)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= (size_t)a.owidth * a.oheight) return;
  const float f = form[k];
  float m;
  if(kind == 4) m = f * a.opacity;
  else if(kind == 3) m = a.inversed ? a.global_opacity * (1.0f - f) : f * a.global_opacity;
  else
  {
    const int y = (int)(k / a.owidth), x = (int)(k - (size_t)y * a.owidth);
    m = parametric_mask<CS>(in[(size_t)(y + a.yoffs) * a.iwidth + a.xoffs + x], out[k], a, f);
  }
  plane[k] = m;
}

// the mask plane of a frame, for the post operations that are not pointwise (blur)
template <int CS>
__global__ __launch_bounds__(256) void blend_mask_kernel(const float4 *__restrict__ in, const float4 *__restrict__ out,
                                                         float *__restrict__ plane, const blend_args a_by_value)
{
  constexpr int at = kernarg_offset_after<blend_args, const float4 *, const float4 *, float *>(); // after the three pointers
  static_assert(at == 24, "the blend kernels' by-value parameter block follows three pointers");
  const blend_args &a = kernarg_at<blend_args>(at);
  (void)a_by_value;
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= (size_t)a.owidth * a.oheight) return;
  const int y = (int)(k / a.owidth), x = (int)(k - (size_t)y * a.owidth);
  plane[k] = parametric_mask<CS>(in[(size_t)(y + a.yoffs) * a.iwidth + a.xoffs + x], out[k], a, a.seed);
}

// MASK: 0 = the same value everywhere, 1 = parametric, computed here, 2 = read from the (blurred) mask plane
template <int CS, int MASK>
__global__ __launch_bounds__(256) void blend_kernel(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                    const float *__restrict__ plane, const blend_args a_by_value)
{
  constexpr int at = kernarg_offset_after<blend_args, const float4 *, const float4 *, float *>(); // after the three pointers
  static_assert(at == 24, "the blend kernels' by-value parameter block follows three pointers");
  const blend_args &a = kernarg_at<blend_args>(at);
  (void)a_by_value;
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= (size_t)a.owidth * a.oheight) return;
  const int y = (int)(k / a.owidth), x = (int)(k - (size_t)y * a.owidth);
  const float4 pa = in[(size_t)(y + a.yoffs) * a.iwidth + a.xoffs + x];
  const float4 pb = out[k];
  float m = a.constant;
  if(MASK == 2)
  {
    m = plane[k];
    if(a.tone) m = tone_curve(m, a);
  }
  if(MASK == 1)
  {
    m = parametric_mask<CS>(pa, pb, a, a.seed);
    if(a.tone) m = tone_curve(m, a);
  }
  float4 r;
  if(CS == DT_HIP_BLEND_CS_LAB) r = a.reverse ? blend_pixel_lab(a.mode, pb, pa, m) : blend_pixel_lab(a.mode, pa, pb, m);
  else if(CS == DT_HIP_BLEND_CS_RGB_DISPLAY) r = a.reverse ? blend_pixel_display(a.mode, pb, pa, m) : blend_pixel_display(a.mode, pa, pb, m);
  else r = a.reverse ? blend_pixel(a.mode, pb, pa, a.p, m) : blend_pixel(a.mode, pa, pb, a.p, m);
  nt_store(out + k, r);
}

} // namespace

namespace ansel
{
// dt_develop_blend_get_mask_usage(), blend.c:262-320: is any parametric channel of the blend's colourspace away from its
// full range
static bool blend_is_parametric(const dt_hip_blend_data_t *d)
{
  if(!(d->mask_mode & DT_HIP_MASK_PARAMETRIC)) return false;
  const unsigned CH_MASK = d->blend_cst == DT_HIP_BLEND_CS_LAB ? LAB_MASK : RGB_MASK;
  for(unsigned ch = 0; ch < DT_HIP_BLENDIF_SIZE; ch++)
  {
    const unsigned bit = 1u << ch;
    if(!(CH_MASK & bit) || !(d->blendif & bit)) continue;
    const float *c = &d->blendif_parameters[ch * 4];
    if(fabsf(c[0]) > 1e-6f || fabsf(c[1]) > 1e-6f || fabsf(c[2] - 1.0f) > 1e-6f || fabsf(c[3] - 1.0f) > 1e-6f) return true;
  }
  return false;
}

// does this blend run _refine_with_detail_mask() (blend.c:789)?  Only an enabled blend with a mask source other than a
// raster mask alone, a non-zero threshold and the raw detail mask at hand (:379) -- what the row-band walker refuses
int blend_refines_with_detail_mask(const dt_hip_blend_data_t *d)
{
  if(!d || !(d->mask_mode & DT_HIP_MASK_ENABLED) || d->details == 0.f || !d->detail_mask) return 0;
  const bool form = d->form_mask != nullptr, parametric = blend_is_parametric(d);
  const bool raster_only = form && (d->mask_mode & DT_HIP_MASK_RASTER) && !(d->mask_mode & DT_HIP_MASK_SHAPE) && !parametric;
  return (form || parametric) && !raster_only;
}
} // namespace ansel

extern "C" int dt_hip_develop_blend_process(int devid, const dt_hip_piece_t *piece, const dt_hip_blend_data_t *d,
                                            dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  const bool lab = d->blend_cst == DT_HIP_BLEND_CS_LAB, raw = d->blend_cst == DT_HIP_BLEND_CS_RAW;
  const bool display = d->blend_cst == DT_HIP_BLEND_CS_RGB_DISPLAY;
  if(d->blend_cst < DT_HIP_BLEND_CS_RAW || d->blend_cst > DT_HIP_BLEND_CS_RGB_SCENE)
  {
    set_last_error("blend: unknown colourspace %d", d->blend_cst);
    return DT_HIP_INVALID_ARG;
  }
  const unsigned CH_MASK = lab ? LAB_MASK : RGB_MASK;
  if(piece->channels != (raw ? 1 : 4)) return DT_HIP_INVALID_ARG;
  // drawn / raster masks and the details threshold: rendered and refined by the host into ONE plane, as the reference's
  // device blend receives them (blend.c:1278-1325)
  const float *form = (const float *)d->form_mask;
  float *refined = nullptr; // the form mask times the detail mask, built here when the raw detail mask is at hand
  if((d->mask_mode & (DT_HIP_MASK_SHAPE | DT_HIP_MASK_RASTER)) && !form)
  {
    set_last_error("blend: a drawn / raster mask needs the host-rendered form mask (form_mask)");
    return DT_HIP_INVALID_ARG;
  }
  // a details threshold is read where the reference reads it (blend.c:732-790): behind use_masks && !raster_only, and
  // _refine_with_detail_mask() returns silently when the pipe holds no raw detail mask (:379) -- so a uniform, disabled
  // or raster-only blend that carries a stale `details` value, or one without the plane, blends as if it were 0
  if(form && raw)
  {
    set_last_error("blend: form masks in the raw colourspace are not built");
    return DT_HIP_INVALID_ARG;
  }
  if(!(d->mask_mode & DT_HIP_MASK_ENABLED)) return DT_HIP_SUCCESS; // blend.c:673
  blend_args a;
  memset(&a, 0, sizeof(a));
  a.xoffs = piece->roi_out.x - piece->roi_in.x;
  a.yoffs = piece->roi_out.y - piece->roi_in.y;
  a.iwidth = piece->roi_in.width;
  a.owidth = piece->roi_out.width;
  a.oheight = piece->roi_out.height;
  if(a.owidth <= 0 || a.oheight <= 0) return DT_HIP_SUCCESS;
  // "skipped blending: roi's do not match", blend.c:697-702
  if(piece->roi_out.scale != piece->roi_in.scale || a.xoffs < 0 || a.yoffs < 0
     || ((a.xoffs > 0 || a.yoffs > 0)
         && (a.owidth + a.xoffs > a.iwidth || a.oheight + a.yoffs > piece->roi_in.height)))
    return DT_HIP_SUCCESS;

  const float opacity = fminf(fmaxf(d->opacity / 100.0f, 0.0f), 1.0f);
  const bool parametric = blend_is_parametric(d);
  // make_mask(), blendif_rgb_jzczhz.c:196-324: which of its three cases
  const unsigned any_channel_active = d->blendif & CH_MASK;
  const unsigned mask_inclusive = d->mask_combine & DT_HIP_COMBINE_INCL;
  const unsigned mask_inversed = d->mask_combine & DT_HIP_COMBINE_INV;
  const unsigned blendif = d->blendif ^ (mask_inclusive ? CH_MASK << 16 : 0);
  const unsigned canceling_channel = (blendif >> 16) & ~blendif & CH_MASK;
  const float global_opacity = fminf(fmaxf(d->opacity / 100.0f, 0.0f), 1.0f); // clamp_simd()
  const float seed = mask_inclusive ? 0.0f : 1.0f; // the form mask of a parametric-only blend, blend.c:749-757
  bool per_pixel = false;
  a.constant = opacity;
  // blend.c:732-760: which mask sources there are; a raster mask alone is form * opacity, without make_mask() and
  // without post operations
  const bool use_masks = form || parametric;
  const bool raster_only = form && (d->mask_mode & DT_HIP_MASK_RASTER) && !(d->mask_mode & DT_HIP_MASK_SHAPE) && !parametric;
  const bool post = use_masks && !raster_only;
  if(d->details != 0.f && d->detail_mask && post)
  {
    // _refine_with_detail_mask(), blend.c:361-425 (:789): the form mask -- or the neutral fill of a parametric-only blend,
    // :749-757 -- times the blurred sigmoid of the raw detail mask
    if(raw)
    {
      set_last_error("blend: the details threshold in the raw colourspace is not built");
      return DT_HIP_INVALID_ARG;
    }
    refined = (float *)dt_hip_alloc_device_buffer(devid, (size_t)a.owidth * a.oheight * sizeof(float));
    if(!refined) return DT_HIP_SYSMEM_ALLOCATION;
    const int rerr = detail_refine_launch(devid, (const float *)d->detail_mask, form, seed, d->details, a.owidth, a.oheight, refined);
    if(rerr != DT_HIP_SUCCESS)
    {
      dt_hip_release_mem_object(refined);
      return rerr;
    }
    form = refined;
  }
  int form_kind = 0; // form_mask_kernel: 0 = no form plane involved
  if(raster_only)
    form_kind = 4;
  else if(use_masks)
  {
    if(!(d->mask_mode & DT_HIP_MASK_PARAMETRIC) || (!canceling_channel && !any_channel_active))
    {
      if(form) form_kind = 3;
      else a.constant = mask_inversed ? global_opacity * (1.0f - seed) : seed * global_opacity;
    }
    else if(canceling_channel || !any_channel_active)
      a.constant = ((mask_inversed == 0) ^ (mask_inclusive == 0)) ? global_opacity : 0.0f;
    else
    {
      per_pixel = true;
      if(form) form_kind = 2;
    }
  }
  a.global_opacity = global_opacity;
  a.seed = seed;
  a.inclusive = mask_inclusive != 0;
  a.inversed = mask_inversed != 0;
  a.blendif = blendif;
  a.opacity = opacity;
  a.brightness = d->brightness;
  a.e = expf(3.f * d->contrast);
  a.tone = post && (fabsf(d->contrast) >= 0.01f || fabsf(d->brightness) >= 0.01f) && opacity > 1e-4f;
  a.mode = d->blend_mode & 0xFFu;
  a.reverse = (d->blend_mode & DT_HIP_BLEND_REVERSE) == DT_HIP_BLEND_REVERSE;
  a.p = exp2f(d->blend_parameter);
  // post operations follow make_mask() only (blend.c:759-900): feathering and blur in the order of
  // _develop_mask_get_post_operations() (blend.c:427-469), then the tone curve
  const bool blur = post && d->blur_radius > 0.1f;
  const bool feather = post && !raw && d->feathering_radius > 0.1f; // one-channel buffers are never feathered, blend.c:431
  const bool feather_before = d->feathering_guide == DT_HIP_MASK_GUIDE_IN_BEFORE_BLUR
                              || d->feathering_guide == DT_HIP_MASK_GUIDE_OUT_BEFORE_BLUR;
  const bool feather_out = d->feathering_guide == DT_HIP_MASK_GUIDE_OUT_BEFORE_BLUR
                           || d->feathering_guide == DT_HIP_MASK_GUIDE_OUT_AFTER_BLUR;
  if(feather && !feather_out && (a.xoffs || a.yoffs || a.iwidth != a.owidth || piece->roi_in.height != a.oheight))
  {
    // blend.c:823-824 hands the region copy ch * yoffs and ch * oheight where it takes rows: the reference reads past
    // its input there
    set_last_error("blend: feathering guided by the module's input needs roi_in == roi_out (the reference reads outside "
                   "its input otherwise)");
    if(refined) dt_hip_release_mem_object(refined);
    return DT_HIP_INVALID_ARG;
  }
  const bool spatial = blur || feather;
  const size_t np = (size_t)a.owidth * a.oheight;
  hipStream_t s = stream_of(devid);
  // the blurred mask plane: `plane` holds the mask, `scratch` / `scratch2` the causal and the anticausal half of a pass
  float *plane = nullptr, *scratch = nullptr, *scratch2 = nullptr;
  if(spatial || form_kind)
  {
    plane = (float *)dt_hip_alloc_device_buffer(devid, np * sizeof(float));
    if(blur) scratch = (float *)dt_hip_alloc_device_buffer(devid, np * sizeof(float));
    if(blur) scratch2 = (float *)dt_hip_alloc_device_buffer(devid, np * sizeof(float));
    if(!plane || (blur && (!scratch || !scratch2)))
    {
      if(plane) dt_hip_release_mem_object(plane);
      if(scratch) dt_hip_release_mem_object(scratch);
      if(scratch2) dt_hip_release_mem_object(scratch2);
      if(refined) dt_hip_release_mem_object(refined);
      return DT_HIP_SYSMEM_ALLOCATION;
    }
  }
  auto blur_plane = [&]() {
    // compute_gauss_params(), gaussian.c:44-95, order 0; sigma = blur_radius * roi_out->scale (blend.c:871)
    gauss_args g;
    g.width = a.owidth;
    g.height = a.oheight;
    const float sigma = d->blur_radius * (float)piece->roi_out.scale;
    const float alpha = 1.695f / sigma;
    const float ema = expf(-alpha);
    const float ema2 = expf(-2.0f * alpha);
    g.b1 = -2.0f * ema;
    g.b2 = ema2;
    const float k = (1.0f - ema) * (1.0f - ema) / (1.0f + (2.0f * alpha * ema) - ema2);
    g.a0 = k;
    g.a1 = k * (alpha - 1.0f) * ema;
    g.a2 = k * (alpha + 1.0f) * ema;
    g.a3 = -k * ema2;
    g.coefp = (g.a0 + g.a1) / (1.0f + g.b1 + g.b2);
    g.coefn = (g.a2 + g.a3) / (1.0f + g.b1 + g.b2);
    launch_scope ls(devid, "blend_mask_blur");
    const dim3 tiles((a.owidth + 31) / 32, (a.oheight + 31) / 32), tiles_t((a.oheight + 31) / 32, (a.owidth + 31) / 32);
    gauss_vertical<<<(a.owidth + 63) / 64, 128, 0, s>>>(plane, scratch, scratch2, a.owidth, a.oheight, g); // gaussian.c: columns first
    gauss_sum_transpose<<<tiles, 256, 0, s>>>(scratch, scratch2, plane, a.owidth, a.oheight);
    gauss_vertical<<<(a.oheight + 63) / 64, 128, 0, s>>>(plane, scratch, scratch2, a.oheight, a.owidth, g); // the rows
    gauss_sum_transpose<<<tiles_t, 256, 0, s>>>(scratch, scratch2, plane, a.oheight, a.owidth);
  };
  // the spatial post operations on the plane; false after a failure (error code in `post_err`)
  int post_err = DT_HIP_SUCCESS;
  auto spatial_ops = [&]() {
    const bool feather_first = feather && blur && feather_before;
    for(int step = 0; step < 2 && post_err == DT_HIP_SUCCESS; step++)
    {
      if(blur && (feather_first ? step == 1 : step == 0)) blur_plane();
      if(feather && (feather_first ? step == 0 : step == 1))
      {
        // _develop_blend_process_feather(), blend.c:603-623
        int w = (int)(2 * d->feathering_radius * (float)piece->roi_out.scale + 0.5f);
        if(w < 1) w = 1;
        post_err = guided_filter_launch(devid, feather_out ? (const float4 *)dev_out : (const float4 *)dev_in, plane, a.owidth,
                                        a.oheight, w, 1.0f, lab ? 1.0f : 100.0f, 0.0f, 1.0f);
      }
    }
  };
  auto release_planes = [&]() {
    if(plane) dt_hip_release_mem_object(plane);   // stream-ordered: re-used only by later launches
    if(scratch) dt_hip_release_mem_object(scratch);
    if(scratch2) dt_hip_release_mem_object(scratch2);
    if(refined) dt_hip_release_mem_object(refined);
  };
  if(raw)
  {
    // dt_develop_blendif_raw_make_mask(), blendif_raw.c:36-62: global opacity, optionally inverted -- the
    // conditions of a parametric mask have no channels to look at
    if(parametric) a.constant = mask_inversed ? global_opacity * (1.0f - seed) : seed * global_opacity;
    if(blur)
    {
      fill_plane<<<pixel_grid(np), 256, 0, s>>>(plane, np, a.constant);
      blur_plane();
    }
    else if(a.tone)
      a.constant = tone_curve(a.constant, a);
    {
      launch_scope ls(devid, "blend_raw");
      blend_raw_kernel<<<pixel_grid(np), 256, 0, s>>>((const float *)dev_in, (float *)dev_out, plane, a);
    }
    release_planes();
    return check_launch("blend_raw");
  }
  if(per_pixel)
  {
    // dt_develop_blendif_process_parameters(), blend.c:214-260
    for(size_t i = 0, j = 0; i < DT_HIP_BLENDIF_SIZE; i++, j += PARAM_ITEMS)
    {
      float *p = a.parameters + j;
      if(d->blendif & (1u << i))
      {
        const float *bp = d->blendif_parameters + i * 4;
        const float boost = exp2f(d->blendif_boost_factors[i]);
        const float offset = (lab && (i == 1 || i == 2 || i == 5 || i == 6)) ? 0.5f : 0.0f; // Lab a, b in / out
        for(int k = 0; k < 4; k++) p[k] = (bp[k] - offset) * boost;
        p[4] = 1.0f / fmaxf(0.001f, p[1] - p[0]);
        p[5] = 1.0f / fmaxf(0.001f, p[3] - p[2]);
        if(bp[0] <= 0.0f && bp[1] <= 0.0f) p[0] = p[1] = -INFINITY;
        if(bp[2] >= 1.0f && bp[3] >= 1.0f) p[2] = p[3] = INFINITY;
      }
      else
      {
        p[0] = p[1] = -INFINITY;
        p[2] = p[3] = INFINITY;
        p[4] = p[5] = 0.0f;
      }
    }
    // dt_develop_blendif_init_masking_profile(), blend.c:322-353: Bradford D50 -> D65 times RGB -> XYZ
    static const float Mb[3][3] = { { 0.9555766f, -0.0230393f, 0.0631636f },
                                    { -0.0282895f, 1.0099416f, 0.0210077f },
                                    { 0.0122982f, -0.0204830f, 1.3299098f } };
    for(int y = 0; y < 3; y++)
      for(int c = 0; c < 3; c++)
      {
        float sum = 0.0f;
        for(int i = 0; i < 3; i++) sum += Mb[y][i] * d->matrix_in[i][c];
        a.mT[c][y] = sum;
      }
    for(int c = 0; c < 3; c++) a.luma[c] = d->matrix_in[1][c];
  }
  const float4 *const in = (const float4 *)dev_in;
  float4 *const out = (float4 *)dev_out;
  const unsigned grid = pixel_grid(np);
  const int cs = lab ? DT_HIP_BLEND_CS_LAB : (display ? DT_HIP_BLEND_CS_RGB_DISPLAY : DT_HIP_BLEND_CS_RGB_SCENE);
  int mask = per_pixel ? 1 : 0;
  if(form_kind)
  {
    // the mask from the host-rendered form plane, as a plane (the post operations follow below)
    launch_scope ls(devid, "blend_mask");
    if(cs == DT_HIP_BLEND_CS_LAB) form_mask_kernel<DT_HIP_BLEND_CS_LAB><<<grid, 256, 0, s>>>(in, out, form, plane, a, form_kind);
    else if(cs == DT_HIP_BLEND_CS_RGB_DISPLAY)
      form_mask_kernel<DT_HIP_BLEND_CS_RGB_DISPLAY><<<grid, 256, 0, s>>>(in, out, form, plane, a, form_kind);
    else form_mask_kernel<DT_HIP_BLEND_CS_RGB_SCENE><<<grid, 256, 0, s>>>(in, out, form, plane, a, form_kind);
    mask = 2;
  }
  else if(spatial)
  {
    // the mask as a plane (before the post operations), feathered / blurred in place
    launch_scope ls(devid, "blend_mask");
    if(!per_pixel) fill_plane<<<grid, 256, 0, s>>>(plane, np, a.constant);
    else if(cs == DT_HIP_BLEND_CS_LAB) blend_mask_kernel<DT_HIP_BLEND_CS_LAB><<<grid, 256, 0, s>>>(in, out, plane, a);
    else if(cs == DT_HIP_BLEND_CS_RGB_DISPLAY) blend_mask_kernel<DT_HIP_BLEND_CS_RGB_DISPLAY><<<grid, 256, 0, s>>>(in, out, plane, a);
    else blend_mask_kernel<DT_HIP_BLEND_CS_RGB_SCENE><<<grid, 256, 0, s>>>(in, out, plane, a);
    mask = 2;
  }
  else if(!per_pixel && a.tone)
    a.constant = tone_curve(a.constant, a); // the same value for every pixel
  if(spatial) spatial_ops();
  if(post_err != DT_HIP_SUCCESS)
  {
    release_planes();
    return post_err;
  }
  {
    launch_scope ls(devid, "blend_kernel");
#define BLEND_LAUNCH(CS_)                                                                   \
  do                                                                                        \
  {                                                                                         \
    if(mask == 2) blend_kernel<CS_, 2><<<grid, 256, 0, s>>>(in, out, plane, a);             \
    else if(mask == 1) blend_kernel<CS_, 1><<<grid, 256, 0, s>>>(in, out, plane, a);        \
    else blend_kernel<CS_, 0><<<grid, 256, 0, s>>>(in, out, plane, a);                      \
  } while(0)
    if(cs == DT_HIP_BLEND_CS_LAB) BLEND_LAUNCH(DT_HIP_BLEND_CS_LAB);
    else if(cs == DT_HIP_BLEND_CS_RGB_DISPLAY) BLEND_LAUNCH(DT_HIP_BLEND_CS_RGB_DISPLAY);
    else BLEND_LAUNCH(DT_HIP_BLEND_CS_RGB_SCENE);
#undef BLEND_LAUNCH
  }
  release_planes();
  return check_launch("blend_kernel");
}
