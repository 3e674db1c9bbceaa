/* oracle/src/blend.c -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * CPU restatement of the blend stage, all four blend colourspaces (RGB scene, RGB display, Lab, raw), for the mask sources
 * the device path supports (uniform opacity, parametric mask, mask tone curve).  It follows
 *   dt_develop_blend_process()                  src/develop/blend.c:657-900 (driver)
 *   dt_develop_blend_get_mask_usage()           src/develop/blend.c:262-320 (is the parametric mask in use)
 *   dt_develop_blendif_process_parameters()     src/develop/blend.c:214-260
 *   dt_develop_blendif_init_masking_profile()   src/develop/blend.c:322-353
 *   _develop_blend_process_mask_tone_curve()    src/develop/blend.c:626-655
 *   dt_develop_blendif_rgb_jzczhz_make_mask()   src/develop/blends/blendif_rgb_jzczhz.c:196-324
 *   _blendif_* channel functions                src/develop/blends/blendif_rgb_jzczhz.c:42-194
 *   _blend_* operators, _choose_blend_func()    src/develop/blends/blendif_rgb_jzczhz.c:328-650
 *   dt_develop_blendif_rgb_jzczhz_blend()       src/develop/blends/blendif_rgb_jzczhz.c:878-960
 *   dt_XYZ_2_JzAzBz(), dt_JzAzBz_2_JzCzhz()     src/common/colorspaces_inline_conversions.h:672-781
 *   Lab: make_mask, channel functions, operators, blend   src/develop/blends/blendif_lab.c:56-300, :302-1068, :1302-1420
 *   dt_Lab_2_LCH()                              src/common/colorspaces_inline_conversions.h:594-606
 *   raw: make_mask, operators, blend            src/develop/blends/blendif_raw.c:36-62, :66-353, :355-412
 *   RGB (display): mask channels, operators     src/develop/blends/blendif_rgb_hsl.c:56-216, :348-1008
 *   dt_RGB_2_HSL() ... dt_HSV_2_RGB()           src/common/colorspaces_inline_conversions.h:421-565
 * One pass per pixel instead of the reference's one pass per mask channel: every step is pointwise,
 * so the order of the passes does not enter the arithmetic.
 * Pinned by tests/test_oracle_vs_ref.py against oracle/_ref (the reference's own functions). */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PARAM_ITEMS 6 /* DEVELOP_BLENDIF_PARAMETER_ITEMS */
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

typedef struct blend_ctx_t
{
  float parameters[PARAM_ITEMS * DT_HIP_BLENDIF_SIZE];
  float luma[3];           /* matrix_in row 1 */
  float xyz_d65_T[3][4];   /* masking profile matrix_out_transposed: RGB -> XYZ D65 */
  unsigned blendif;        /* with the inclusive-combine inversion applied */
} blend_ctx_t;

/* _blendif_compute_factor() */
static float compute_factor(const float value, const unsigned invert, const float *p)
{
  float factor;
  if(value <= p[0]) factor = 0.0f;
  else if(value < p[1]) factor = (value - p[0]) * p[4];
  else if(value <= p[2]) factor = 1.0f;
  else if(value < p[3]) factor = 1.0f - (value - p[2]) * p[5];
  else factor = 0.0f;
  return invert ? 1.0f - factor : factor;
}

/* dt_ioppr_rgb_matrix_to_xyz() on a linear profile + dt_XYZ_2_JzAzBz() + dt_JzAzBz_2_JzCzhz() */
static void rgb_to_JzCzhz(const float *rgb, const float mT[3][4], float JzCzhz[3])
{
  const float b = 1.15f, g = 0.66f, c1 = 0.8359375f, c2 = 18.8515625f, c3 = 18.6875f, n = 0.159301758f,
              p = 134.034375f, d = -0.56f, d0 = 1.6295499532821566e-11f;
  static const float M[3][3] = { { 0.41478972f, 0.579999f, 0.0146480f },
                                 { -0.2015100f, 1.120649f, 0.0531008f },
                                 { -0.0166008f, 0.264800f, 0.6684799f } };
  static const float A_T[3][3] = { { 0.5f, 3.524000f, 0.199076f }, { 0.5f, -4.066708f, 1.096799f }, { 0.0f, 0.542708f, -1.295875f } };
  float D65[3];
  for(int c = 0; c < 3; c++)
  {
    float o = mT[0][c] * rgb[0];
    o = mT[1][c] * rgb[1] + o;
    D65[c] = mT[2][c] * rgb[2] + o;
  }
  float XYZ[3], LMS[3], Jab[3];
  XYZ[0] = b * D65[0] - (b - 1.0f) * D65[2];
  XYZ[1] = g * D65[1] - (g - 1.0f) * D65[0];
  XYZ[2] = D65[2];
  for(int i = 0; i < 3; i++)
  {
    LMS[i] = M[i][0] * XYZ[0] + M[i][1] * XYZ[1] + M[i][2] * XYZ[2];
    LMS[i] = powf(fmaxf(LMS[i] / 10000.f, 0.0f), n);
    LMS[i] = powf((c1 + c2 * LMS[i]) / (1.0f + c3 * LMS[i]), p);
  }
  for(int c = 0; c < 3; c++) Jab[c] = A_T[0][c] * LMS[0] + A_T[1][c] * LMS[1] + A_T[2][c] * LMS[2];
  Jab[0] = fmaxf(((1.0f + d) * Jab[0]) / (1.0f + d * Jab[0]) - d0, 0.f);
  const float var_H = atan2f(Jab[2], Jab[1]) / (2.0f * 3.14159265358979324f);
  JzCzhz[0] = Jab[0];
  JzCzhz[1] = hypotf(Jab[1], Jab[2]);
  JzCzhz[2] = var_H >= 0.0f ? var_H : 1.0f + var_H;
}

/* _blendif_combine_channels(): `blendif` and `params` already shifted for the output side */
static void rgb_to_hsl(const float *RGB, float *HSL);

/* `hsl`: the RGB (display) colourspace, whose channels 8..10 are H, S, L instead of Jz, Cz, hz */
static float combine_channels(const float *px, float temp, const unsigned blendif, const float *params, const blend_ctx_t *x,
                              const int hsl)
{
  if(blendif & 1u)
  {
    const float value = x->luma[0] * px[0] + x->luma[1] * px[1] + x->luma[2] * px[2];
    temp *= compute_factor(value, (blendif >> 16) & 1u, params);
  }
  for(unsigned c = 1; c <= 3; c++)
    if(blendif & (1u << c)) temp *= compute_factor(px[c - 1], (blendif >> 16) & (1u << c), params + PARAM_ITEMS * c);
  if(blendif & ((1u << 8) | (1u << 9) | (1u << 10)))
  {
    float JzCzhz[3];
    if(hsl) rgb_to_hsl(px, JzCzhz);
    else rgb_to_JzCzhz(px, x->xyz_d65_T, JzCzhz);
    float factor = 1.0f;
    for(unsigned i = 0; i < 3; i++)
      factor *= compute_factor(JzCzhz[i], (blendif >> 16) & (1u << (8 + i)), params + PARAM_ITEMS * (8 + i));
    temp *= factor;
  }
  return temp;
}

/* _develop_blend_process_mask_tone_curve(), one value */
static float tone_curve(const float m, const float e, const float brightness, const float opacity)
{
  const float mask_epsilon = 16 * 1.19209290e-7f;
  float x = m / opacity;
  x = 2.f * x - 1.f;
  if(1.f - brightness <= 0.f) x = m <= mask_epsilon ? -1.f : 1.f;
  else if(1.f + brightness <= 0.f) x = m >= 1.f - mask_epsilon ? 1.f : -1.f;
  else if(brightness > 0.f)
  {
    x = (x + brightness) / (1.f - brightness);
    x = fminf(x, 1.f);
  }
  else
  {
    x = (x + brightness) / (1.f + brightness);
    x = fmaxf(x, -1.f);
  }
  const float r = ((x * e / (1.f + (e - 1.f) * fabsf(x))) / 2.f + 0.5f) * opacity;
  return r > 1.f ? 1.f : (r < 0.f ? 0.f : r);
}

/* the _blend_* row functions, one pixel: a = bottom layer, b = top layer, lo = local opacity */
static void blend_pixel(const unsigned mode, const float *a, const float *b, const float p, const float lo, float *out)
{
  switch(mode)
  {
    case MODE_MULTIPLY:
      for(int k = 0; k < 3; k++) out[k] = a[k] * (1.0f - lo) + (a[k] * b[k] * p) * lo;
      break;
    case MODE_AVERAGE:
      for(int k = 0; k < 3; k++) out[k] = a[k] * (1.0f - lo) + (a[k] + b[k]) / 2.0f * lo;
      break;
    case MODE_ADD:
      for(int k = 0; k < 3; k++) out[k] = a[k] * (1.0f - lo) + (a[k] + p * b[k]) * lo;
      break;
    case MODE_SUBTRACT:
      for(int k = 0; k < 3; k++) out[k] = a[k] * (1.0f - lo) + fmaxf(a[k] - p * b[k], 0.0f) * lo;
      break;
    case MODE_SUBTRACT_INVERSE:
      for(int k = 0; k < 3; k++) out[k] = a[k] * (1.0f - lo) + fmaxf(b[k] - p * a[k], 0.0f) * lo;
      break;
    case MODE_DIFFERENCE:
    case MODE_DIFFERENCE2:
      for(int k = 0; k < 3; k++) out[k] = a[k] * (1.0f - lo) + fabsf(a[k] - b[k]) * lo;
      break;
    case MODE_DIVIDE:
      for(int k = 0; k < 3; k++) out[k] = a[k] * (1.0f - lo) + a[k] / fmaxf(p * b[k], 1e-6f) * lo;
      break;
    case MODE_DIVIDE_INVERSE:
      for(int k = 0; k < 3; k++) out[k] = a[k] * (1.0f - lo) + b[k] / fmaxf(p * a[k], 1e-6f) * lo;
      break;
    case MODE_GEOMETRIC_MEAN:
      /* fmax(), the double one, on a float product: the conversion back is exact */
      for(int k = 0; k < 3; k++) out[k] = a[k] * (1.0f - lo) + sqrtf((float)fmax(a[k] * b[k], 0.0f)) * lo;
      break;
    case MODE_HARMONIC_MEAN:
      for(int k = 0; k < 3; k++)
        out[k] = a[k] * (1.0f - lo) + 2.0f * a[k] * b[k] / (fmaxf(a[k], 5e-7f) + fmaxf(b[k], 5e-7f)) * lo;
      break;
    case MODE_CHROMATICITY:
    case MODE_LIGHTNESS:
    {
      const float norm_a = (float)fmax(sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), 1e-6f);
      const float norm_b = (float)fmax(sqrtf(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]), 1e-6f);
      if(mode == MODE_CHROMATICITY)
        for(int k = 0; k < 3; k++) out[k] = a[k] * (1.0f - lo) + b[k] * norm_a / norm_b * lo;
      else
        for(int k = 0; k < 3; k++) out[k] = a[k] * (1.0f - lo) + a[k] * norm_b / norm_a * lo;
      break;
    }
    case MODE_RGB_R:
      out[0] = a[0] * (1.0f - lo) + p * b[0] * lo;
      out[1] = a[1];
      out[2] = a[2];
      break;
    case MODE_RGB_G:
      out[0] = a[0];
      out[1] = a[1] * (1.0f - lo) + p * b[1] * lo;
      out[2] = a[2];
      break;
    case MODE_RGB_B:
      out[0] = a[0];
      out[1] = a[1];
      out[2] = a[2] * (1.0f - lo) + p * b[2] * lo;
      break;
    default: /* normal */
      for(int k = 0; k < 3; k++) out[k] = a[k] * (1.0f - lo) + b[k] * lo;
      break;
  }
  out[3] = lo;
}

/* ---- Lab ---------------------------------------------------------------------------------------- */
enum
{
  LAB_LIGHTEN = 0x02, LAB_DARKEN = 0x03, LAB_MULTIPLY = 0x04, LAB_AVERAGE = 0x05, LAB_ADD = 0x06, LAB_SUBTRACT = 0x07,
  LAB_DIFFERENCE = 0x08, LAB_SCREEN = 0x09, LAB_OVERLAY = 0x0A, LAB_SOFTLIGHT = 0x0B, LAB_HARDLIGHT = 0x0C,
  LAB_VIVIDLIGHT = 0x0D, LAB_LINEARLIGHT = 0x0E, LAB_PINLIGHT = 0x0F, LAB_LIGHTNESS = 0x10, LAB_CHROMATICITY = 0x11,
  LAB_HUE = 0x12, LAB_COLOR = 0x13, LAB_COLORADJUST = 0x16, LAB_DIFFERENCE2 = 0x17, LAB_BOUNDED = 0x19,
  LAB_LAB_LIGHTNESS = 0x1A, LAB_LAB_COLOR = 0x1B, LAB_LAB_L = 0x1E, LAB_LAB_A = 0x1F, LAB_LAB_B = 0x20,
};

static int lab_mode_supported(const unsigned mode)
{
  (void)mode;
  return 1;
}

/* dt_Lab_2_LCH() / dt_LCH_2_Lab(), src/common/colorspaces_inline_conversions.h:594-620 */
static void lab_to_lch(const float *Lab, float *LCH)
{
  float var_H = atan2f(Lab[2], Lab[1]);
  if(var_H > 0.0f) var_H = var_H / (2.0f * 3.14159265358979324f);
  else var_H = 1.0f - fabsf(var_H) / (2.0f * 3.14159265358979324f);
  LCH[0] = Lab[0];
  LCH[1] = hypotf(Lab[1], Lab[2]);
  LCH[2] = var_H;
}

static void lch_to_lab(const float *LCH, float *Lab)
{
  Lab[0] = LCH[0];
  Lab[1] = cosf(2.0f * 3.14159265358979324f * LCH[2]) * LCH[1];
  Lab[2] = sinf(2.0f * 3.14159265358979324f * LCH[2]) * LCH[1];
}

/* _blendif_combine_channels() of blendif_lab.c:139-173 */
static float combine_channels_lab(const float *px, float temp, const unsigned blendif, const float *params)
{
  if(blendif & 1u) temp *= compute_factor(px[0] / 100.0f, (blendif >> 16) & 1u, params);
  if(blendif & 2u) temp *= compute_factor(px[1] / 256.0f, (blendif >> 16) & 2u, params + PARAM_ITEMS * 1);
  if(blendif & 4u) temp *= compute_factor(px[2] / 256.0f, (blendif >> 16) & 4u, params + PARAM_ITEMS * 2);
  if(blendif & ((1u << 8) | (1u << 9)))
  {
    const float c_scale = 1.0f / (128.0f * sqrtf(2.0f));
    /* dt_Lab_2_LCH() */
    float var_H = atan2f(px[2], px[1]);
    if(var_H > 0.0f) var_H = var_H / (2.0f * 3.14159265358979324f);
    else var_H = 1.0f - fabsf(var_H) / (2.0f * 3.14159265358979324f);
    const float C = hypotf(px[1], px[2]);
    float factor = 1.0f;
    factor *= compute_factor(C * c_scale, (blendif >> 16) & (1u << 8), params + PARAM_ITEMS * 8);
    factor *= compute_factor(var_H, (blendif >> 16) & (1u << 9), params + PARAM_ITEMS * 9);
    temp *= factor;
  }
  return temp;
}

static float CL(const float x, const float lo, const float hi) { return fminf(fmaxf(x, lo), hi); } /* _CLAMP() */

/* the _blend_* row functions of blendif_lab.c:320-1068, one pixel: a = bottom layer, b = top layer */
static void blend_pixel_lab(const unsigned mode, const float *a, const float *b, const float lo, float *out)
{
  static const float min[4] = { 0.0f, -1.0f, -1.0f, 0.0f }, max[4] = { 1.0f, 1.0f, 1.0f, 1.0f };
  static const float scale[3] = { 1 / 100.0f, 1 / 128.0f, 1 / 128.0f }, rescale[3] = { 100.0f, 128.0f, 128.0f };
  float ta[3], tb[3];
  for(int c = 0; c < 3; c++)
  {
    ta[c] = a[c] * scale[c];
    tb[c] = b[c] * scale[c];
  }
  const float lo2 = lo * lo;
  /* the lightness-based operators work on L shifted into [0, lmax] */
  const float lmin = 0.0f, lmax = max[0] + fabsf(min[0]);
  const float la = CL(ta[0] + fabsf(min[0]), lmin, lmax), lb = CL(tb[0] + fabsf(min[0]), lmin, lmax);
  const float halfmax = lmax / 2.0f, doublemax = lmax * 2.0f;
  const float f = fmaxf(ta[0], 0.01f);
  int chroma_follows = 0; /* a, b scaled by the lightness ratio with opacity lo2 (overlay .. linearlight) */
  switch(mode)
  {
    case LAB_BOUNDED:
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
      for(int x = 0; x < 3; x++) tb[x] = CL(ta[x] * (1.0f - lo) + (ta[x] + tb[x]) / 2.0f * lo, min[x], max[x]);
      break;
    case LAB_ADD:
      for(int x = 0; x < 3; x++) tb[x] = CL(ta[x] * (1.0f - lo) + (ta[x] + tb[x]) * lo, min[x], max[x]);
      break;
    case LAB_SUBTRACT:
      for(int x = 0; x < 3; x++)
        tb[x] = CL(ta[x] * (1.0f - lo) + ((tb[x] + ta[x]) - (fabsf(min[x] + max[x]))) * lo, min[x], max[x]);
      break;
    case LAB_DIFFERENCE:
      for(int x = 0; x < 3; x++)
      {
        const float xmax = max[x] + fabsf(min[x]);
        const float xa = CL(ta[x] + fabsf(min[x]), lmin, xmax), xb = CL(tb[x] + fabsf(min[x]), lmin, xmax);
        tb[x] = CL(xa * (1.0f - lo) + fabsf(xa - xb) * lo, lmin, xmax) - fabsf(min[x]);
      }
      break;
    case LAB_DIFFERENCE2:
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
      chroma_follows = 1;
      break;
    case LAB_SOFTLIGHT:
      tb[0] = CL(la * (1.0f - lo2) + (lb > halfmax ? lmax - (lmax - la) * (lmax - (lb - halfmax)) : la * (lb + halfmax)) * lo2,
                 lmin, lmax)
              - fabsf(min[0]);
      chroma_follows = 1;
      break;
    case LAB_HARDLIGHT:
      tb[0] = CL(la * (1.0f - lo2)
                     + (lb > halfmax ? lmax - (lmax - doublemax * (la - halfmax)) * (lmax - lb) : doublemax * la * lb) * lo2,
                 lmin, lmax)
              - fabsf(min[0]);
      chroma_follows = 1;
      break;
    case LAB_VIVIDLIGHT:
      tb[0] = CL(la * (1.0f - lo2)
                     + (lb > halfmax ? (lb >= lmax ? lmax : la / (doublemax * (lmax - lb)))
                                     : (lb <= lmin ? lmin : lmax - (lmax - la) / (doublemax * lb)))
                           * lo2,
                 lmin, lmax)
              - fabsf(min[0]);
      chroma_follows = 1;
      break;
    case LAB_LINEARLIGHT:
      tb[0] = CL(la * (1.0f - lo2) + (la + doublemax * lb - lmax) * lo2, lmin, lmax) - fabsf(min[0]);
      chroma_follows = 1;
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
      /* blendif_lab.c:843-975: through LCh, hue blended along the shortest way round the colour circle */
      float tta[3], ttb[3];
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
    default: /* normal, unbounded */
      for(int x = 0; x < 3; x++) tb[x] = ta[x] * (1.0f - lo) + tb[x] * lo;
      break;
  }
  if(chroma_follows)
  {
    tb[1] = CL(ta[1] * (1.0f - lo2) + (ta[1] + tb[1]) * tb[0] / f * lo2, min[1], max[1]);
    tb[2] = CL(ta[2] * (1.0f - lo2) + (ta[2] + tb[2]) * tb[0] / f * lo2, min[2], max[2]);
  }
  for(int c = 0; c < 3; c++) out[c] = tb[c] * rescale[c];
  out[3] = lo;
}

/* ---- mask blur: dt_gaussian_blur(), src/pixel/gaussian.c:44-95 (parameters), :176-326 (the two recursive
 * passes), one channel, order 0, values clamped to [0, 1] on the way in (CLAMPF, src/math/math.h:91) ---------- */
static float clampf01(const float a) { return a >= 0.0f ? (a <= 1.0f ? a : 1.0f) : 0.0f; }

static void gaussian_blur_mask(float *buf, const int width, const int height, const float sigma)
{
  const float alpha = 1.695f / sigma;
  const float ema = expf(-alpha);
  const float ema2 = expf(-2.0f * alpha);
  const float b1 = -2.0f * ema, b2 = ema2;
  const float k = (1.0f - ema) * (1.0f - ema) / (1.0f + (2.0f * alpha * ema) - ema2);
  const float a0 = k, a1 = k * (alpha - 1.0f) * ema, a2 = k * (alpha + 1.0f) * ema, a3 = -k * ema2;
  const float coefp = (a0 + a1) / (1.0f + b1 + b2), coefn = (a2 + a3) / (1.0f + b1 + b2);
  float *temp = (float *)malloc(sizeof(float) * (size_t)width * height);
  if(!temp) return;
  /* vertical, column by column: forward then backward, both from the unblurred input */
#pragma omp parallel for schedule(static)
  for(int i = 0; i < width; i++)
  {
    float xp = clampf01(buf[i]), yb = xp * coefp, yp = yb;
    for(int j = 0; j < height; j++)
    {
      const size_t o = (size_t)j * width + i;
      const float xc = clampf01(buf[o]);
      const float yc = (a0 * xc) + (a1 * xp) - (b1 * yp) - (b2 * yb);
      temp[o] = yc;
      xp = xc;
      yb = yp;
      yp = yc;
    }
    float xn = clampf01(buf[(size_t)(height - 1) * width + i]), xa = xn, yn = xn * coefn, ya = yn;
    for(int j = height - 1; j > -1; j--)
    {
      const size_t o = (size_t)j * width + i;
      const float xc = clampf01(buf[o]);
      const float yc = (a2 * xn) + (a3 * xa) - (b1 * yn) - (b2 * ya);
      xa = xn;
      xn = xc;
      ya = yn;
      yn = yc;
      temp[o] += yc;
    }
  }
  /* horizontal, line by line, from the vertically blurred plane */
#pragma omp parallel for schedule(static)
  for(int j = 0; j < height; j++)
  {
    const float *t = temp + (size_t)j * width;
    float *out = buf + (size_t)j * width;
    float xp = clampf01(t[0]), yb = xp * coefp, yp = yb;
    for(int i = 0; i < width; i++)
    {
      const float xc = clampf01(t[i]);
      const float yc = (a0 * xc) + (a1 * xp) - (b1 * yp) - (b2 * yb);
      out[i] = yc;
      xp = xc;
      yb = yp;
      yp = yc;
    }
    float xn = clampf01(t[width - 1]), xa = xn, yn = xn * coefn, ya = yn;
    for(int i = width - 1; i > -1; i--)
    {
      const float xc = clampf01(t[i]);
      const float yc = (a2 * xn) + (a3 * xa) - (b1 * yn) - (b2 * ya);
      xa = xn;
      xn = xc;
      ya = yn;
      yn = yc;
      out[i] += yc;
    }
  }
  free(temp);
}

/* ---- raw (one channel) ----------------------------------------------------------------------------
 * the _blend_* row functions of src/develop/blends/blendif_raw.c:66-288, one photosite */
static float clamp01(const float x) { return fminf(fmaxf(x, 0.0f), 1.0f); } /* clamp_simd() */

static float blend_value_raw(const unsigned mode, const float a, const float b, const float lo)
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
    default: return a * (1.0f - lo) + b * lo; /* normal, unbounded */
  }
}

/* dt_develop_blend_process() for a one-channel buffer: the mask never depends on the photosite
 * (dt_develop_blendif_raw_make_mask(), blendif_raw.c:36-62: global opacity, optionally inverted) */
static int blend_raw(const dt_hip_piece_t *piece, const dt_hip_blend_data_t *d, const float *in, float *out)
{
  const int xoffs = piece->roi_out.x - piece->roi_in.x, yoffs = piece->roi_out.y - piece->roi_in.y;
  const int iwidth = piece->roi_in.width, iheight = piece->roi_in.height;
  const int owidth = piece->roi_out.width, oheight = piece->roi_out.height;
  if(piece->roi_out.scale != piece->roi_in.scale || xoffs < 0 || yoffs < 0
     || ((xoffs > 0 || yoffs > 0) && (owidth + xoffs > iwidth || oheight + yoffs > iheight)))
    return 0;
  const float opacity = fminf(fmaxf(d->opacity / 100.0f, 0.0f), 1.0f);
  int parametric = 0;
  if(d->mask_mode & DT_HIP_MASK_PARAMETRIC)
    for(unsigned ch = 0; ch < DT_HIP_BLENDIF_SIZE; ch++)
    {
      const unsigned bit = 1u << ch;
      if(!(RGB_MASK & bit) || !(d->blendif & bit)) continue;
      const float *c = &d->blendif_parameters[ch * 4];
      if(fabsf(c[0]) > 1e-6f || fabsf(c[1]) > 1e-6f || fabsf(c[2] - 1.0f) > 1e-6f || fabsf(c[3] - 1.0f) > 1e-6f) parametric = 1;
    }
  float m = opacity;
  if(parametric)
  {
    const float seed = (d->mask_combine & DT_HIP_COMBINE_INCL) ? 0.0f : 1.0f;
    const float global_opacity = fminf(fmaxf(0.0f, (d->opacity / 100.0f)), 1.0f);
    m = (d->mask_combine & DT_HIP_COMBINE_INV) ? global_opacity * (1.0f - seed) : seed * global_opacity;
  }
  /* post operations (blur, then the tone curve), only on a parametric mask: blend.c:759-900 */
  float *plane = NULL;
  if(parametric && d->blur_radius > 0.1f)
  {
    plane = (float *)malloc(sizeof(float) * (size_t)owidth * oheight);
    if(!plane) return 1;
    for(size_t k = 0; k < (size_t)owidth * oheight; k++) plane[k] = m;
    gaussian_blur_mask(plane, owidth, oheight, d->blur_radius * (float)piece->roi_out.scale);
  }
  const int tone = parametric && (fabsf(d->contrast) >= 0.01f || fabsf(d->brightness) >= 0.01f) && opacity > 1e-4f;
  const float e = expf(3.f * d->contrast);
  const unsigned mode = d->blend_mode & 0xFFu;
  const int reverse = (d->blend_mode & DT_HIP_BLEND_REVERSE) == DT_HIP_BLEND_REVERSE;
#pragma omp parallel for schedule(static)
  for(int y = 0; y < oheight; y++)
    for(int x = 0; x < owidth; x++)
    {
      const float a = in[(size_t)(y + yoffs) * iwidth + xoffs + x];
      float *bo = out + (size_t)y * owidth + x;
      const float b = *bo;
      float mm = plane ? plane[(size_t)y * owidth + x] : m;
      if(tone) mm = tone_curve(mm, e, d->brightness, opacity);
      *bo = reverse ? blend_value_raw(mode, b, a, mm) : blend_value_raw(mode, a, b, mm);
    }
  free(plane);
  return 0;
}

/* ---- RGB (display) --------------------------------------------------------------------------------
 * src/develop/blends/blendif_rgb_hsl.c; HSL / HSV conversions of src/common/colorspaces_inline_conversions.h:421-565 */
static float rgb_hue(const float *RGB, const float max, const float delta) /* _dt_RGB_2_Hue() */
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

static void hue_to_rgb(float *RGB, const float H, const float C, const float min) /* _dt_Hue_2_RGB() */
{
  const float h = H * 6.0f;
  const float i = floorf(h);
  const float f = h - i;
  const float fc = f * C;
  const float top = C + min;
  const float inc = fc + min;
  const float dec = top - fc;
  /* (size_t)i of the reference: anything but 0 .. 4 (negative, >= 5, NaN) takes the last branch */
  const int sector = (i >= 0.0f && i < 5.0f) ? (int)i : 5;
  if(sector == 0) { RGB[0] = top; RGB[1] = inc; RGB[2] = min; }
  else if(sector == 1) { RGB[0] = dec; RGB[1] = top; RGB[2] = min; }
  else if(sector == 2) { RGB[0] = min; RGB[1] = top; RGB[2] = inc; }
  else if(sector == 3) { RGB[0] = min; RGB[1] = dec; RGB[2] = top; }
  else if(sector == 4) { RGB[0] = inc; RGB[1] = min; RGB[2] = top; }
  else { RGB[0] = top; RGB[1] = min; RGB[2] = dec; }
}

static void rgb_to_hsl(const float *RGB, float *HSL) /* dt_RGB_2_HSL() */
{
  const float min = fminf(RGB[0], fminf(RGB[1], RGB[2]));
  const float max = fmaxf(RGB[0], fmaxf(RGB[1], RGB[2]));
  const float delta = max - min;
  const float L = (max + min) / 2.0f;
  float H, S;
  if(fabsf(max) > 1e-6f && fabsf(delta) > 1e-6f)
  {
    if(L < 0.5f) S = delta / (max + min);
    else S = delta / (2.0f - max - min);
    H = rgb_hue(RGB, max, delta);
  }
  else
  {
    H = 0.0f;
    S = 0.0f;
  }
  HSL[0] = H;
  HSL[1] = S;
  HSL[2] = L;
}

static void hsl_to_rgb(const float *HSL, float *RGB) /* dt_HSL_2_RGB() */
{
  const float L = HSL[2];
  float C;
  if(L < 0.5f) C = L * HSL[1];
  else C = (1.0f - L) * HSL[1];
  const float m = L - C;
  hue_to_rgb(RGB, HSL[0], 2.0f * C, m);
}

static void rgb_to_hsv(const float *RGB, float *HSV) /* dt_RGB_2_HSV() */
{
  const float min = fminf(RGB[0], fminf(RGB[1], RGB[2]));
  const float max = fmaxf(RGB[0], fmaxf(RGB[1], RGB[2]));
  const float delta = max - min;
  float S, H;
  if(fabsf(max) > 1e-6f && fabsf(delta) > 1e-6f)
  {
    S = delta / max;
    H = rgb_hue(RGB, max, delta);
  }
  else
  {
    S = 0.0f;
    H = 0.0f;
  }
  HSV[0] = H;
  HSV[1] = S;
  HSV[2] = max;
}

static void hsv_to_rgb(const float *HSV, float *RGB) /* dt_HSV_2_RGB() */
{
  const float C = HSV[1] * HSV[2];
  const float m = HSV[2] - C;
  hue_to_rgb(RGB, HSV[0], C, m);
}

enum { DSP_HSV_VALUE = 0x1C, DSP_HSV_COLOR = 0x1D };

/* the _blend_* row functions of blendif_rgb_hsl.c:348-913, one pixel */
static void blend_pixel_display(const unsigned mode, const float *a, const float *b, const float lo, float *out)
{
  switch(mode)
  {
    case LAB_LIGHTNESS:
    case LAB_CHROMATICITY:
    case LAB_HUE:
    case LAB_COLOR:
    case LAB_COLORADJUST:
    {
      float ta[3], tb[3], tta[3], ttb[3];
      for(int k = 0; k < 3; k++)
      {
        ta[k] = clamp01(a[k]);
        tb[k] = clamp01(b[k]);
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
      else { ttb[0] = hue; ttb[1] = sat; } /* coloradjust: lightness of the module output */
      hsl_to_rgb(ttb, out);
      for(int k = 0; k < 3; k++) out[k] = clamp01(out[k]);
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
      const float xa = ta[1] * cosf(2.0f * 3.14159265358979324f * ta[0]);
      const float ya = ta[1] * sinf(2.0f * 3.14159265358979324f * ta[0]);
      const float xb = tb[1] * cosf(2.0f * 3.14159265358979324f * tb[0]);
      const float yb = tb[1] * sinf(2.0f * 3.14159265358979324f * tb[0]);
      const float xc = xa * (1.0f - lo) + xb * lo;
      const float yc = ya * (1.0f - lo) + yb * lo;
      tb[0] = atan2f(yc, xc) / (2.0f * 3.14159265358979324f);
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
    default: /* the per-channel operators: the formulas of the one-channel colourspace on each of R, G, B */
      for(int k = 0; k < 3; k++) out[k] = blend_value_raw(mode, a[k], b[k], lo);
      break;
  }
  out[3] = lo;
}

int oracle_develop_blend(const dt_hip_piece_t *piece, const dt_hip_blend_data_t *d, const void *in_, void *out_)
{
  if(!piece || !d || !in_ || !out_) return 1;
  if(d->blend_cst == DT_HIP_BLEND_CS_RAW)
  {
    /* one channel: no feathering whatever the radius (mask_feather needs >= 3 channels, blend.c:431) */
    if((d->mask_mode & (DT_HIP_MASK_SHAPE | DT_HIP_MASK_RASTER)) || d->details != 0.f || piece->channels != 1) return 1;
    if(!(d->mask_mode & DT_HIP_MASK_ENABLED)) return 0;
    return blend_raw(piece, d, (const float *)in_, (float *)out_);
  }
  const int lab = d->blend_cst == DT_HIP_BLEND_CS_LAB, display = d->blend_cst == DT_HIP_BLEND_CS_RGB_DISPLAY;
  if((d->blend_cst != DT_HIP_BLEND_CS_RGB_SCENE && !lab && !display) || piece->channels != 4) return 1;
  /* drawn / raster masks and the details threshold: rendered and refined by the host into ONE plane, as the reference's
   * device blend receives them (blend.c:1278-1325) */
  if((d->mask_mode & (DT_HIP_MASK_SHAPE | DT_HIP_MASK_RASTER)) && !d->form_mask) return 1;
  /* a details threshold without the raw detail mask: _refine_with_detail_mask() returns silently (blend.c:379) */
  const float *form = (const float *)d->form_mask;
  float *refined = NULL;
  if(lab && !lab_mode_supported(d->blend_mode & 0xFFu)) return 1;
  const unsigned CH_MASK = lab ? LAB_MASK : RGB_MASK;
  if(!(d->mask_mode & DT_HIP_MASK_ENABLED)) return 0;
  const float *in = (const float *)in_;
  float *out = (float *)out_;
  const int xoffs = piece->roi_out.x - piece->roi_in.x, yoffs = piece->roi_out.y - piece->roi_in.y;
  const int iwidth = piece->roi_in.width, iheight = piece->roi_in.height;
  const int owidth = piece->roi_out.width, oheight = piece->roi_out.height;
  if(piece->roi_out.scale != piece->roi_in.scale || xoffs < 0 || yoffs < 0
     || ((xoffs > 0 || yoffs > 0) && (owidth + xoffs > iwidth || oheight + yoffs > iheight)))
    return 0; /* "skipped blending: roi's do not match", blend.c:697-702 */

  const float opacity = fminf(fmaxf(d->opacity / 100.0f, 0.0f), 1.0f);

  /* dt_develop_blend_get_mask_usage(): parametric part */
  int parametric = 0;
  if(d->mask_mode & DT_HIP_MASK_PARAMETRIC)
    for(unsigned ch = 0; ch < DT_HIP_BLENDIF_SIZE; ch++)
    {
      const unsigned bit = 1u << ch;
      if(!(CH_MASK & bit) || !(d->blendif & bit)) continue;
      const float *c = &d->blendif_parameters[ch * 4];
      if(fabsf(c[0]) > 1e-6f || fabsf(c[1]) > 1e-6f || fabsf(c[2] - 1.0f) > 1e-6f || fabsf(c[3] - 1.0f) > 1e-6f) parametric = 1;
    }

  /* make_mask(): which of its three cases */
  const unsigned any_channel_active = d->blendif & CH_MASK;
  const unsigned mask_inclusive = d->mask_combine & DT_HIP_COMBINE_INCL;
  const unsigned mask_inversed = d->mask_combine & DT_HIP_COMBINE_INV;
  const unsigned blendif = d->blendif ^ (mask_inclusive ? CH_MASK << 16 : 0);
  const unsigned canceling_channel = (blendif >> 16) & ~blendif & CH_MASK;
  const float global_opacity = fminf(fmaxf(d->opacity / 100.0f, 0.0f), 1.0f);
  const float seed = mask_inclusive ? 0.0f : 1.0f; /* the form mask of a parametric-only blend, blend.c:749-757 */
  /* blend.c:732-745: a raster mask alone is form * opacity, no make_mask(), no post operations */
  const int raster_only = form && (d->mask_mode & DT_HIP_MASK_RASTER) && !(d->mask_mode & DT_HIP_MASK_SHAPE) && !parametric;
  const int use_masks = form || parametric;
  if(d->details != 0.f && d->detail_mask && use_masks && !raster_only)
  {
    /* _refine_with_detail_mask(), blend.c:361-425 (:789): the form mask -- or the neutral fill of a parametric-only
     * blend, :749-757 -- times the blurred sigmoid of the raw detail mask (same geometry: nothing to warp) */
    const size_t n = (size_t)owidth * oheight;
    refined = (float *)malloc(sizeof(float) * n);
    if(!refined) return 1;
    oracle_detail_mask((const float *)d->detail_mask, refined, owidth, oheight, d->details);
    for(size_t k = 0; k < n; k++) refined[k] = (form ? form[k] : seed) * refined[k];
    form = refined;
  }
  int kind; /* 0 uniform, 1 constant after make_mask, 2 per pixel, 3 unconditional on a form plane, 4 raster only */
  float constant = opacity;
  if(!use_masks) kind = 0;
  else if(raster_only) kind = 4;
  else if(!(d->mask_mode & DT_HIP_MASK_PARAMETRIC) || (!canceling_channel && !any_channel_active))
  {
    /* make_mask(), blendif_rgb_jzczhz.c:228-240: not conditional */
    kind = form ? 3 : 1;
    constant = mask_inversed ? global_opacity * (1.0f - seed) : seed * global_opacity;
  }
  else if(canceling_channel || !any_channel_active)
  {
    kind = 1;
    constant = ((mask_inversed == 0) ^ (mask_inclusive == 0)) ? global_opacity : 0.0f;
  }
  else kind = 2;

  blend_ctx_t x;
  memset(&x, 0, sizeof(x));
  x.blendif = blendif;
  if(kind == 2)
  {
    /* dt_develop_blendif_process_parameters() */
    for(size_t i = 0, j = 0; i < DT_HIP_BLENDIF_SIZE; i++, j += PARAM_ITEMS)
    {
      float *p = x.parameters + j;
      if(d->blendif & (1u << i))
      {
        const float *bp = d->blendif_parameters + i * 4;
        const float boost = exp2f(d->blendif_boost_factors[i]);
        const float offset = (lab && (i == 1 || i == 2 || i == 5 || i == 6)) ? 0.5f : 0.0f; /* a, b in / out */
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
    /* dt_develop_blendif_init_masking_profile(): Bradford D50 -> D65 times the profile's RGB -> XYZ */
    static const float Mb[3][3] = { { 0.9555766f, -0.0230393f, 0.0631636f },
                                    { -0.0282895f, 1.0099416f, 0.0210077f },
                                    { 0.0122982f, -0.0204830f, 1.3299098f } };
    for(int y = 0; y < 3; y++)
      for(int c = 0; c < 3; c++)
      {
        float sum = 0.0f;
        for(int i = 0; i < 3; i++) sum += Mb[y][i] * d->matrix_in[i][c];
        x.xyz_d65_T[c][y] = sum;
      }
    for(int c = 0; c < 3; c++) x.luma[c] = d->matrix_in[1][c];
  }

  const int post = use_masks && !raster_only; /* the post operations follow make_mask() only, blend.c:746-900 */
  const int tone = post && (fabsf(d->contrast) >= 0.01f || fabsf(d->brightness) >= 0.01f) && opacity > 1e-4f;
  const float e = expf(3.f * d->contrast);
  const float p = exp2f(d->blend_parameter);
  const unsigned mode = d->blend_mode & 0xFFu;
  const int reverse = (d->blend_mode & DT_HIP_BLEND_REVERSE) == DT_HIP_BLEND_REVERSE;

  /* with a spatial post operation the mask becomes a plane: build it, feather / blur it in the order of
   * _develop_mask_get_post_operations() (blend.c:427-469), then tone curve + operator per pixel */
  float *plane = NULL;
  const int blur = post && d->blur_radius > 0.1f;
  const int feather = post && d->feathering_radius > 0.1f; /* piece->channels == 4 here */
  const int feather_before = d->feathering_guide == DT_HIP_MASK_GUIDE_IN_BEFORE_BLUR
                             || d->feathering_guide == DT_HIP_MASK_GUIDE_OUT_BEFORE_BLUR;
  const int feather_out = d->feathering_guide == DT_HIP_MASK_GUIDE_OUT_BEFORE_BLUR
                          || d->feathering_guide == DT_HIP_MASK_GUIDE_OUT_AFTER_BLUR;
  const int spatial = blur || feather;
  if(spatial)
  {
    plane = (float *)malloc(sizeof(float) * (size_t)owidth * oheight);
    if(!plane)
    {
      free(refined);
      return 1;
    }
  }
  for(int pass = spatial ? 0 : 1; pass < 2; pass++)
  {
  if(pass == 1 && spatial)
  {
    const int feather_first = feather && blur && feather_before;
    for(int step = 0; step < 2; step++)
    {
      const int do_feather = feather && (feather_first ? step == 0 : step == 1);
      const int do_blur = blur && (feather_first ? step == 1 : step == 0);
      if(do_blur) gaussian_blur_mask(plane, owidth, oheight, d->blur_radius * (float)piece->roi_out.scale);
      if(do_feather)
      {
        /* _develop_blend_process_feather(), blend.c:603-623 */
        int w = (int)(2 * d->feathering_radius * (float)piece->roi_out.scale + 0.5f);
        if(w < 1) w = 1;
        const float guide_weight = lab ? 1.0f : 100.0f;
        float *bak = (float *)malloc(sizeof(float) * (size_t)owidth * oheight);
        const float *guide = out; /* FEATHER_OUT: the module's output, not yet blended */
        int err = !bak;
        if(!err && !feather_out)
        {
          /* FEATHER_IN: the module's input.  When roi_in != roi_out the reference copies the region with the row
           * offset AND the row count multiplied by the channel count (blend.c:823-824 passes ch * yoffs, ch * oheight
           * to a helper that takes rows): it reads past its input.  Not reproduced: refused. */
          if(xoffs || yoffs || iwidth != owidth || iheight != oheight) err = 1;
          guide = in;
        }
        if(!err)
        {
          memcpy(bak, plane, sizeof(float) * (size_t)owidth * oheight);
          err = oracle_guided_filter(guide, bak, plane, owidth, oheight, 4, w, 1.f, guide_weight, 0.f, 1.f);
        }
        free(bak);
        if(err)
        {
          free(plane);
          free(refined);
          return 1;
        }
      }
    }
  }
#pragma omp parallel for schedule(static)
  for(int y = 0; y < oheight; y++)
    for(int xx = 0; xx < owidth; xx++)
    {
      const float *a = in + ((size_t)(y + yoffs) * iwidth + xoffs + xx) * 4;
      float *bo = out + ((size_t)y * owidth + xx) * 4;
      const float b[4] = { bo[0], bo[1], bo[2], bo[3] };
      float m = constant;
      const float sd = form ? form[(size_t)y * owidth + xx] : seed; /* the form mask at this pixel */
      if(pass == 1 && spatial) m = plane[(size_t)y * owidth + xx];
      else if(kind == 4) m = sd * opacity;
      else if(kind == 3) m = mask_inversed ? global_opacity * (1.0f - sd) : sd * global_opacity;
      else if(kind == 2)
      {
        float temp = 1.0f;
        if(lab)
        {
          temp = combine_channels_lab(a, temp, x.blendif, x.parameters);
          temp = combine_channels_lab(b, temp, x.blendif >> GRAY_OUT, x.parameters + PARAM_ITEMS * GRAY_OUT);
        }
        else
        {
          temp = combine_channels(a, temp, x.blendif, x.parameters, &x, display);
          temp = combine_channels(b, temp, x.blendif >> GRAY_OUT, x.parameters + PARAM_ITEMS * GRAY_OUT, &x, display);
        }
        if(mask_inclusive)
          m = mask_inversed ? global_opacity * (1.0f - sd) * temp : global_opacity * (1.0f - (1.0f - sd) * temp);
        else
          m = mask_inversed ? global_opacity * (1.0f - sd * temp) : global_opacity * sd * temp;
      }
      if(pass == 0)
      {
        plane[(size_t)y * owidth + xx] = m;
        continue;
      }
      if(tone) m = tone_curve(m, e, d->brightness, opacity);
      if(lab)
      {
        if(reverse) blend_pixel_lab(mode, b, a, m, bo);
        else blend_pixel_lab(mode, a, b, m, bo);
      }
      else if(display)
      {
        if(reverse) blend_pixel_display(mode, b, a, m, bo);
        else blend_pixel_display(mode, a, b, m, bo);
      }
      else if(reverse) blend_pixel(mode, b, a, p, m, bo);
      else blend_pixel(mode, a, b, p, m, bo);
    }
  }
  free(plane);
  free(refined);
  return 0;
}
