/* oracle/src/filmic.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of filmic RGB tone mapping with the deprecated highlight reconstruction bypassed
 * (hl_deprecated, the default of every new edit: src/iop/filmicrgb.c:2733, :4104):
 *   filmic_agx()        filmicrgb.c:2495-2587   colour science v8 "AgX" (versions 5..9), the default
 *   filmic_v5()         filmicrgb.c:2247-2300   colour science v7
 *   filmic_chroma_v4()  filmicrgb.c:2153-2198   colour science v6, chroma preservation on
 *   filmic_split_v4()   filmicrgb.c:2201-2244   colour science v6, per-channel
 *   filmic_split_v1() :1534, filmic_split_v2_v3() :1574, filmic_chroma_v1() :1614, filmic_chroma_v2_v3() :1670
 *                       the colour sciences of 2019-2020 (versions 0..2), still run for old edits
 * and what they call: log_tonemapping :1047, filmic_spline :1063-1160, get_pixel_norm_simd
 * :976-1035, pipe_RGB_to_Ych_simd :1740, Ych_to_pipe_RGB_simd :1765, filmic_desaturate_v4 :1781,
 * clip_chroma* :1826-1925, gamut_check_* :1928-1984, gamut_mapping_simd :1986-2030,
 * filmic_v4_prepare_matrices :2033, filmic_agx_prepare_bracket :2390, filmic_agx_compress_negatives
 * :2461; LMS/Yrg conversions of src/common/colorspaces_inline_conversions.h:902-1074; matrix
 * helpers of src/math/matrices.h.  Per-call matrix preparation is part of process() in the
 * reference and is restated here too (binary32, same accumulation order).
 */
#include <float.h>
#include <math.h>
#include <string.h>
#include "oracle.h"

typedef float mat_t[4][4];
typedef struct { float v[4]; } v4;

#define MIN_(a, b) (((a) < (b)) ? (a) : (b))
#define MAX_(a, b) (((a) > (b)) ? (a) : (b))
#define CLAMP_(x, lo, hi) (((x) > (hi)) ? (hi) : (((x) < (lo)) ? (lo) : (x)))            /* glib */
#define CLAMPF_(a, mn, mx) ((a) >= (mn) ? ((a) <= (mx) ? (a) : (mx)) : (mn))             /* math.h:91 */
#define CIE_Y_1931_to_CIE_Y_2006(x) (1.05785528f * (x))
#define INVERSE_SQRT_3 0.5773502691896258f

/* src/pixel/chromatic_adaptation.h:264-283, src/common/colorspaces_inline_conversions.h:902-911 */
static const mat_t XYZ_D50_to_D65_CAT16 = { { 9.89466254e-01f, -4.00304626e-02f, 4.40530317e-02f, 0.f },
                                            { -5.40518733e-03f, 1.00666069e+00f, -1.75551955e-03f, 0.f },
                                            { -4.03920992e-04f, 1.50768030e-02f, 1.30210211e+00f, 0.f } };
static const mat_t XYZ_D65_to_D50_CAT16 = { { 1.01085433e+00f, 4.07086103e-02f, -3.41445825e-02f, 0.f },
                                            { 5.42814201e-03f, 9.93581926e-01f, 1.15592039e-03f, 0.f },
                                            { 2.50722468e-04f, -1.14918759e-02f, 7.67964947e-01f, 0.f } };
static const mat_t XYZ_D65_to_LMS_2006_D65 = { { 0.257085f, 0.859943f, -0.031061f, 0.f },
                                               { -0.394427f, 1.175800f, 0.106423f, 0.f },
                                               { 0.064856f, -0.076250f, 0.559067f, 0.f } };
static const mat_t LMS_2006_D65_to_XYZ_D65 = { { 1.80794659f, -1.29971660f, 0.34785879f, 0.f },
                                               { 0.61783960f, 0.39595453f, -0.04104687f, 0.f },
                                               { -0.12546960f, 0.20478038f, 1.74274183f, 0.f } };
static const mat_t filmlightRGB_D65_to_LMS_D65 = { { 0.95f, 0.38f, 0.00f, 0.f }, { 0.05f, 0.62f, 0.03f, 0.f }, { 0.00f, 0.00f, 0.97f, 0.f } };
static const mat_t LMS_D65_to_filmlightRGB_D65 = { { 1.0877193f, -0.66666667f, 0.02061856f, 0.f },
                                                   { -0.0877193f, 1.66666667f, -0.05154639f, 0.f },
                                                   { 0.f, 0.f, 1.03092784f, 0.f } };

/* dt_mat3x4_mul_vec4 (src/system/simd.h:188-197) with the rows of the UNTRANSPOSED matrix */
static inline v4 mat3(const mat_t m, const v4 in)
{
  v4 o;
  for(int i = 0; i < 4; i++)
  {
    const float m0 = i < 3 ? m[i][0] : 0.f, m1 = i < 3 ? m[i][1] : 0.f, m2 = i < 3 ? m[i][2] : 0.f;
    float a = m0 * in.v[0];
    a = m1 * in.v[1] + a;
    a = m2 * in.v[2] + a;
    o.v[i] = a;
  }
  return o;
}

/* dt_colormatrix_mul, src/math/matrices.h:167-179 */
static void mat_mul(mat_t dst, const mat_t m1, const mat_t m2)
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

/* mat3SSEinv, src/math/matrices.h:37-66 */
static int mat_inv(mat_t dst, const mat_t src)
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

/* dot_product (matrices.h:201-206) = rows . v with scalar_product's accumulation (math.h:186-194) */
static void dotp(const float v[4], const mat_t M, float out[4])
{
  for(int i = 0; i < 3; i++)
  {
    float acc = 0.f;
    for(int c = 0; c < 3; c++) acc += v[c] * M[i][c];
    out[i] = acc;
  }
}

/* ---- Yrg / Ych ------------------------------------------------------------------------- */
static inline v4 LMS_to_Yrg(const v4 LMS)
{
  const float Y = 0.68990272f * LMS.v[0] + 0.34832189f * LMS.v[1];
  const float a = LMS.v[0] + LMS.v[1] + LMS.v[2];
  const float inv_a = (a == 0.f) ? 0.f : 1.f / a;
  const v4 lms = { { LMS.v[0] * inv_a, LMS.v[1] * inv_a, LMS.v[2] * inv_a, 0.f } };
  const v4 rgb = mat3(LMS_D65_to_filmlightRGB_D65, lms);
  return (v4){ { Y, rgb.v[0], rgb.v[1], 0.f } };
}

static inline v4 Yrg_to_LMS(const v4 Yrg)
{
  const v4 rgb = { { Yrg.v[1], Yrg.v[2], 1.f - Yrg.v[1] - Yrg.v[2], 0.f } };
  const v4 lms = mat3(filmlightRGB_D65_to_LMS_D65, rgb);
  const float denom = 0.68990272f * lms.v[0] + 0.34832189f * lms.v[1];
  const float a = (denom == 0.f) ? 0.f : Yrg.v[0] / denom;
  return (v4){ { lms.v[0] * a, lms.v[1] * a, lms.v[2] * a, 0.f } };
}

static inline v4 pipe_RGB_to_Ych(const v4 in, const mat_t M)
{
  const v4 Yrg = LMS_to_Yrg(mat3(M, in));
  const float r = Yrg.v[1] - 0.21902143f;
  const float g = Yrg.v[2] - 0.54371398f;
  const float c = sqrtf(g * g + r * r); /* dt_fast_hypotf(g, r) */
  const float cos_h = c != 0.f ? r / c : 1.f;
  const float sin_h = c != 0.f ? g / c : 0.f;
  return (v4){ { Yrg.v[0], c, cos_h, sin_h } };
}

static inline v4 Ych_to_pipe_RGB(const v4 in, const mat_t M)
{
  const v4 Yrg = { { in.v[0], in.v[1] * in.v[2] + 0.21902143f, in.v[1] * in.v[3] + 0.54371398f, 0.f } };
  return mat3(M, Yrg_to_LMS(Yrg));
}

/* ---- tone curve ------------------------------------------------------------------------- */
static inline float clamp_simd(const float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

static inline float log_tonemapping(const float x, const float grey, const float black, const float dynamic_range)
{
  return clamp_simd((log2f(x / grey) - black) / dynamic_range);
}

static inline float filmic_spline(const float x, const dt_hip_filmic_spline_t *s)
{
  const float *M1 = s->M1, *M2 = s->M2, *M3 = s->M3, *M4 = s->M4, *M5 = s->M5;
  const float latitude_min = s->latitude_min, latitude_max = s->latitude_max;
  float result;
  if(x < latitude_min)
  {
    if(s->type[0] == 3)
    {
      if(M5[0] != 0.f)
        result = M3[2] + fmaxf(0.f, M3[0] * powf(fmaxf(x, 0.f), M4[0]));
      else
      {
        const float ty = latitude_min * M2[2] + M1[2];
        const float u = M2[2] * (x - latitude_min) / M1[0];
        result = M1[0] * (u / powf(1.f + powf(u, M2[0]), 1.f / M2[0])) + ty;
      }
    }
    else if(s->type[0] == 0)
      result = M1[0] + x * (M2[0] + x * (M3[0] + x * (M4[0] + x * M5[0])));
    else if(s->type[0] == 1)
      result = M1[0] + x * (M2[0] + x * (M3[0] + x * M4[0]));
    else
    {
      const float xi = latitude_min - x;
      const float rat = xi * (xi * M2[0] + 1.f);
      result = M4[0] - M1[0] * rat / (rat + M3[0]);
    }
  }
  else if(x > latitude_max)
  {
    if(s->type[1] == 3)
    {
      if(M5[1] != 0.f)
        result = M4[2] - fmaxf(0.f, M3[1] * powf(fmaxf(1.f - x, 0.f), M4[1]));
      else
      {
        const float ty = latitude_max * M2[2] + M1[2];
        const float u = M2[2] * (x - latitude_max) / M1[1];
        result = M1[1] * (u / powf(1.f + powf(u, M2[1]), 1.f / M2[1])) + ty;
      }
    }
    else if(s->type[1] == 0)
      result = M1[1] + x * (M2[1] + x * (M3[1] + x * (M4[1] + x * M5[1])));
    else if(s->type[1] == 1)
      result = M1[1] + x * (M2[1] + x * (M3[1] + x * M4[1]));
    else
    {
      const float xi = x - latitude_max;
      const float rat = xi * (xi * M2[1] + 1.f);
      result = M4[1] + M1[1] * rat / (rat + M3[1]);
    }
  }
  else
    result = M1[2] + x * M2[2];
  return result;
}

/* ---- per-call preparation ------------------------------------------------------------- */
typedef struct
{
  mat_t input, output, export_input, export_output; /* filmic_v4_prepare_matrices */
  mat_t inset, outset;                               /* filmic_agx_prepare_bracket */
  float luma[3];
  float norm_min, norm_max;
  float display_black, display_white;
  int use_output_profile;
  float sigma_toe, sigma_shoulder; /* commit_params(), filmicrgb.c:4101-4102 */
} prep_t;

static void to_mat(mat_t m, const float a[3][4])
{
  memset(m, 0, sizeof(mat_t));
  for(int r = 0; r < 3; r++)
    for(int c = 0; c < 3; c++) m[r][c] = a[r][c];
}

static void agx_xyz_D50_to_Yrg(const float xyz_D50[4], float Yrg[4])
{
  float xyz_D65[4] = { 0.f }, lms[4] = { 0.f };
  dotp(xyz_D50, XYZ_D50_to_D65_CAT16, xyz_D65);
  dotp(xyz_D65, XYZ_D65_to_LMS_2006_D65, lms);
  /* LMS_to_Yrg(), the array flavour: colorspaces_inline_conversions.h:1014-1031 */
  const float Y = 0.68990272f * lms[0] + 0.34832189f * lms[1];
  const float a = lms[0] + lms[1] + lms[2];
  float n[4] = { 0.f }, rgb[4] = { 0.f };
  for(int c = 0; c < 4; c++) n[c] = (a == 0.f) ? 0.f : lms[c] / a;
  dotp(n, LMS_D65_to_filmlightRGB_D65, rgb);
  Yrg[0] = Y;
  Yrg[1] = rgb[0];
  Yrg[2] = rgb[1];
}

static void agx_Yrg_to_xyz_D50(const float Yrg[4], float xyz_D50[4])
{
  /* Yrg_to_LMS(), array flavour: colorspaces_inline_conversions.h:1045-1063 */
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

/* _filmic_agx_build_displaced(), filmicrgb.c:2344-2388 */
static int agx_build_displaced(const mat_t work_in, const mat_t work_out, const float inset[3], const float rotation[3], mat_t M)
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
    const float scale = 1.f - CLAMPF_(inset[i], 0.f, 0.9f);
    const float cos_a = cosf(rotation[i]);
    const float sin_a = sinf(rotation[i]);
    const float displaced_Yrg[4] = { primary_Yrg[0], white_Yrg[1] + scale * (cos_a * dr - sin_a * dg),
                                     white_Yrg[2] + scale * (sin_a * dr + cos_a * dg), 0.f };
    float displaced_xyz[4] = { 0.f };
    agx_Yrg_to_xyz_D50(displaced_Yrg, displaced_xyz);
    for(int r = 0; r < 3; r++) P[r][i] = displaced_xyz[r];
  }
  mat_t Pinv = { { 0.f } };
  if(mat_inv(Pinv, P)) return 0;
  float s[4] = { 0.f };
  dotp(white_xyz, Pinv, s);
  for(int r = 0; r < 3; r++)
    for(int c = 0; c < 3; c++) P[r][c] *= s[c];
  mat_mul(M, work_out, P);
  return 1;
}

static void mat_identity(mat_t M)
{
  for(int r = 0; r < 4; r++)
    for(int c = 0; c < 4; c++) M[r][c] = (r == c && r < 3) ? 1.f : 0.f;
}

/* filmic_agx_prepare_bracket(), filmicrgb.c:2390-2459: constants per bleach variant */
static void agx_prepare_bracket(const mat_t work_in, const mat_t work_out, const int variant, mat_t inset, mat_t outset)
{
  static const float K[5][12] = {
    /* version 5: no bleach */
    { +0.5991055f, +0.6000000f, +0.3300009f, +0.0571015f, +0.1999891f, +0.0886110f, 0.761433f, 0.752267f, 0.465293f, -0.0034297f, +0.1952448f, -0.0480109f },
    /* 6: low */
    { +0.6410825f, +0.6898110f, +0.3194529f, +0.0405734f, +0.1631286f, +0.0350584f, 0.784757f, 0.789387f, 0.445403f, -0.0057845f, +0.1593207f, -0.0592955f },
    /* 7: medium (default) */
    { +0.6509540f, +0.7488775f, +0.3517703f, +0.0278602f, +0.1214671f, -0.0228829f, 0.793082f, 0.815169f, 0.460318f, -0.0053781f, +0.1187604f, -0.0794801f },
    /* 8: high */
    { +0.6379749f, +0.7878689f, +0.3753822f, +0.0106096f, +0.0582598f, -0.0696729f, 0.790237f, 0.831376f, 0.465406f, -0.0080070f, +0.0571100f, -0.0912220f },
    /* 9: extra */
    { +0.5770235f, +0.8102094f, +0.4000390f, -0.0081060f, -0.0034008f, -0.1035236f, 0.766420f, 0.838020f, 0.465130f, -0.0122011f, -0.0021732f, -0.0971215f },
  };
  const int row = (variant >= 5 && variant <= 9) ? variant - 5 : 0;
  const float *k = K[row];
  mat_t rec = { { 0.f } };
  if(!agx_build_displaced(work_in, work_out, k + 0, k + 3, inset) || !agx_build_displaced(work_in, work_out, k + 6, k + 9, rec)
     || mat_inv(outset, rec))
  {
    mat_identity(inset);
    mat_identity(outset);
  }
}

static void prepare(const dt_hip_filmicrgb_data_t *d, prep_t *p)
{
  mat_t work_in, work_out, exp_in, exp_out, tmp;
  to_mat(work_in, d->work_matrix_in);
  to_mat(work_out, d->work_matrix_out);
  to_mat(exp_in, d->export_matrix_in);
  to_mat(exp_out, d->export_matrix_out);
  memset(p, 0, sizeof(*p));
  /* filmic_v4_prepare_matrices(), filmicrgb.c:2033-2064 */
  mat_mul(tmp, XYZ_D50_to_D65_CAT16, work_in);
  mat_mul(p->input, XYZ_D65_to_LMS_2006_D65, tmp);
  mat_mul(tmp, XYZ_D65_to_D50_CAT16, LMS_2006_D65_to_XYZ_D65);
  mat_mul(p->output, work_out, tmp);
  p->use_output_profile = d->use_output_profile;
  if(p->use_output_profile)
  {
    mat_mul(tmp, XYZ_D65_to_D50_CAT16, LMS_2006_D65_to_XYZ_D65);
    mat_mul(p->export_output, exp_out, tmp);
    mat_mul(tmp, XYZ_D50_to_D65_CAT16, exp_in);
    mat_mul(p->export_input, XYZ_D65_to_LMS_2006_D65, tmp);
  }
  if(d->version >= 5) agx_prepare_bracket(work_in, work_out, d->version, p->inset, p->outset);
  for(int c = 0; c < 3; c++) p->luma[c] = work_in[1][c];
  /* exp_tonemapping_v2(), filmicrgb.c:1054-1060 */
  p->norm_min = d->grey_source * exp2f(d->dynamic_range * 0.f + d->black_source);
  p->norm_max = d->grey_source * exp2f(d->dynamic_range * 1.f + d->black_source);
  p->display_white = powf(d->spline.y[4], d->output_power);
  p->display_black = powf(d->spline.y[0], d->output_power);
  p->sigma_toe = powf(d->spline.latitude_min / 3.0f, 2.0f);
  p->sigma_shoulder = powf((1.0f - d->spline.latitude_max) / 3.0f, 2.0f);
}

/* ---- gamut mapping -------------------------------------------------------------------- */
static inline v4 filmic_desaturate_v4(const v4 Ych_original, v4 Ych_final, const float saturation)
{
  const float chroma_original = Ych_original.v[1] * Ych_original.v[0];
  float chroma_final = Ych_final.v[1] * Ych_final.v[0];
  const float delta_chroma = saturation * (chroma_original - chroma_final);
  const int filmic_brightens = (Ych_final.v[0] > Ych_original.v[0]);
  const int filmic_resat = (chroma_original < chroma_final);
  const int filmic_desat = (chroma_original > chroma_final);
  const int user_resat = (saturation > 0.f);
  const int user_desat = (saturation < 0.f);
  chroma_final = (filmic_brightens && filmic_resat) ? (chroma_original + chroma_final) / 2.f
                 : ((user_resat && filmic_desat) || user_desat) ? chroma_final + delta_chroma
                                                                : chroma_final;
  Ych_final.v[1] = fmaxf(chroma_final / Ych_final.v[0], 0.f);
  return Ych_final;
}

static inline float clip_chroma_white_raw(const float coeffs[4], const float target_white, const float Y, const float cos_h, const float sin_h)
{
  const float denominator_Y_coeff = coeffs[0] * (0.979381443298969f * cos_h + 0.391752577319588f * sin_h)
                                    + coeffs[1] * (0.0206185567010309f * cos_h + 0.608247422680412f * sin_h)
                                    - coeffs[2] * (cos_h + sin_h);
  const float denominator_target_term = target_white * (0.68285981628866f * cos_h + 0.482137060515464f * sin_h);
  if(denominator_Y_coeff == 0.f) return FLT_MAX;
  const float Y_asymptote = denominator_target_term / denominator_Y_coeff;
  if(Y <= Y_asymptote) return FLT_MAX;
  const float denominator = Y * denominator_Y_coeff - denominator_target_term;
  const float numerator = -0.427506877216495f
                          * (Y * (coeffs[0] + 0.856492345150334f * coeffs[1] + 0.554995960637719f * coeffs[2])
                             - 0.988237752433297f * target_white);
  return numerator / denominator;
}

static inline float clip_chroma_white(const float coeffs[4], const float target_white, const float Y, const float cos_h, const float sin_h)
{
  const float eps = 1e-3f;
  const float max_Y = CIE_Y_1931_to_CIE_Y_2006(target_white);
  const float delta_Y = MAX_(max_Y - Y, 0.f);
  float max_chroma;
  if(delta_Y < eps)
    max_chroma = delta_Y / (eps * max_Y) * clip_chroma_white_raw(coeffs, target_white, (1.f - eps) * max_Y, cos_h, sin_h);
  else
    max_chroma = clip_chroma_white_raw(coeffs, target_white, Y, cos_h, sin_h);
  return max_chroma >= 0.f ? max_chroma : FLT_MAX;
}

static inline float clip_chroma_black(const float coeffs[4], const float cos_h, const float sin_h)
{
  const float denominator = coeffs[0] * (0.979381443298969f * cos_h + 0.391752577319588f * sin_h)
                            + coeffs[1] * (0.0206185567010309f * cos_h + 0.608247422680412f * sin_h)
                            - coeffs[2] * (cos_h + sin_h);
  if(denominator == 0.f) return FLT_MAX;
  const float numerator = -0.427506877216495f * (coeffs[0] + 0.856492345150334f * coeffs[1] + 0.554995960637719f * coeffs[2]);
  const float max_chroma = numerator / denominator;
  return max_chroma >= 0.f ? max_chroma : FLT_MAX;
}

static inline float clip_chroma(const mat_t matrix_out, const float target_white, const float Y, const float cos_h, const float sin_h, const float chroma)
{
  const float wr = clip_chroma_white(matrix_out[0], target_white, Y, cos_h, sin_h);
  const float wg = clip_chroma_white(matrix_out[1], target_white, Y, cos_h, sin_h);
  const float wb = clip_chroma_white(matrix_out[2], target_white, Y, cos_h, sin_h);
  const float max_chroma_white = MIN_(MIN_(wr, wg), wb);
  const float br = clip_chroma_black(matrix_out[0], cos_h, sin_h);
  const float bg = clip_chroma_black(matrix_out[1], cos_h, sin_h);
  const float bb = clip_chroma_black(matrix_out[2], cos_h, sin_h);
  const float max_chroma_black = MIN_(MIN_(br, bg), bb);
  return MIN_(MIN_(chroma, max_chroma_black), max_chroma_white);
}

static inline v4 gamut_check_Yrg(const v4 Ych)
{
  const float Yrg1 = Ych.v[1] * Ych.v[2] + 0.21902143f;
  const float Yrg2 = Ych.v[1] * Ych.v[3] + 0.54371398f;
  float max_c = Ych.v[1];
  if(Yrg1 < 0.f) max_c = fminf(-0.21902143f / Ych.v[2], max_c);
  if(Yrg2 < 0.f) max_c = fminf(-0.54371398f / Ych.v[3], max_c);
  if(Yrg1 + Yrg2 > 1.f) max_c = fminf((1.f - 0.21902143f - 0.54371398f) / (Ych.v[2] + Ych.v[3]), max_c);
  return (v4){ { Ych.v[0], max_c, Ych.v[2], Ych.v[3] } };
}

static inline v4 gamut_check_RGB(const mat_t matrix_in, const mat_t matrix_out, const float display_black, const float display_white, const v4 Ych_in)
{
  v4 RGB_brightened = Ych_to_pipe_RGB(Ych_in, matrix_out);
  const float min_pix = MIN_(MIN_(RGB_brightened.v[0], RGB_brightened.v[1]), RGB_brightened.v[2]);
  const float black_offset = MAX_(-min_pix, 0.f);
  for(int c = 0; c < 4; c++) RGB_brightened.v[c] += black_offset;
  const v4 Ych_brightened = pipe_RGB_to_Ych(RGB_brightened, matrix_in);
  const float Y = CLAMP_((Ych_in.v[0] + Ych_brightened.v[0]) / 2.f, CIE_Y_1931_to_CIE_Y_2006(display_black), CIE_Y_1931_to_CIE_Y_2006(display_white));
  const float new_chroma = clip_chroma(matrix_out, display_white, Y, Ych_in.v[2], Ych_in.v[3], Ych_in.v[1]);
  v4 RGB_out = Ych_to_pipe_RGB((v4){ { Y, new_chroma, Ych_in.v[2], Ych_in.v[3] } }, matrix_out);
  for(int c = 0; c < 4; c++) RGB_out.v[c] = CLAMP_(RGB_out.v[c], 0.f, display_white);
  return RGB_out;
}

static inline v4 gamut_mapping(v4 Ych_final, const v4 Ych_original, const prep_t *p, const float saturation)
{
  Ych_final.v[2] = Ych_original.v[2];
  Ych_final.v[3] = Ych_original.v[3];
  Ych_final.v[0] = CLAMP_(Ych_final.v[0], CIE_Y_1931_to_CIE_Y_2006(p->display_black), CIE_Y_1931_to_CIE_Y_2006(p->display_white));
  Ych_final = filmic_desaturate_v4(Ych_original, Ych_final, saturation);
  Ych_final = gamut_check_Yrg(Ych_final);
  if(!p->use_output_profile) return gamut_check_RGB(p->input, p->output, p->display_black, p->display_white, Ych_final);
  const v4 pix_out = gamut_check_RGB(p->export_input, p->export_output, p->display_black, p->display_white, Ych_final);
  const v4 LMS = mat3(p->export_input, pix_out);
  return mat3(p->output, LMS);
}

/* ---- tone mapping variants ---------------------------------------------------------------- */
static inline float pixel_norm(const v4 px, const int variant, const prep_t *p)
{
  /* get_pixel_norm_simd(), filmicrgb.c:976-1035, linear work profile */
  switch(variant)
  {
    case 1: return fmaxf(fmaxf(px.v[0], px.v[1]), px.v[2]);
    case 3:
    {
      float numerator = 0.0f, denominator = 0.0f;
      for(int c = 0; c < 3; c++)
      {
        const float value = fabsf(px.v[c]);
        const float sq = value * value;
        numerator += sq * value;
        denominator += sq;
      }
      return numerator / fmaxf(denominator, 1e-12f);
    }
    case 4: return sqrtf(px.v[0] * px.v[0] + px.v[1] * px.v[1] + px.v[2] * px.v[2]);
    case 5: return sqrtf(px.v[0] * px.v[0] + px.v[1] * px.v[1] + px.v[2] * px.v[2]) * INVERSE_SQRT_3;
    default: return p->luma[0] * px.v[0] + p->luma[1] * px.v[1] + p->luma[2] * px.v[2];
  }
}

static inline v4 norm_tone_mapping_v4(const v4 pix_in, const int type, const dt_hip_filmicrgb_data_t *d, const prep_t *p)
{
  float norm = CLAMPF_(pixel_norm(pix_in, type, p), p->norm_min, p->norm_max);
  v4 ratios;
  for(int c = 0; c < 4; c++) ratios.v[c] = pix_in.v[c] / norm;
  norm = log_tonemapping(norm, d->grey_source, d->black_source, d->dynamic_range);
  const float sp = filmic_spline(norm, &d->spline);
  norm = powf(CLAMPF_(sp, d->spline.y[0], d->spline.y[4]), d->output_power);
  for(int c = 0; c < 4; c++) ratios.v[c] = ratios.v[c] * norm;
  return ratios;
}

static inline v4 RGB_tone_mapping_v4(const v4 pix_in, const dt_hip_filmicrgb_data_t *d)
{
  v4 o = pix_in;
  for(int c = 0; c < 3; c++)
  {
    const float mapped = log_tonemapping(pix_in.v[c], d->grey_source, d->black_source, d->dynamic_range);
    const float sp = filmic_spline(mapped, &d->spline);
    o.v[c] = powf(CLAMPF_(sp, 0.f, d->spline.y[4]), d->output_power);
  }
  return o;
}

/* ---- colour sciences v1..v3 (2019-2020) -------------------------------------------------- */
#define NORM_MIN_ 1.52587890625e-05f /* src/math/math.h:37 */

/* filmic_desaturate_v1(), filmicrgb.c:1163-1174 */
static inline float desaturate_v1(const float x, const prep_t *p, const float saturation)
{
  const float radius_toe = x;
  const float radius_shoulder = 1.0f - x;
  const float key_toe = expf(-0.5f * radius_toe * radius_toe / p->sigma_toe);
  const float key_shoulder = expf(-0.5f * radius_shoulder * radius_shoulder / p->sigma_shoulder);
  return 1.0f - clamp_simd((key_toe + key_shoulder) / saturation);
}

/* filmic_desaturate_v2(), filmicrgb.c:1178-1189 */
static inline float desaturate_v2(const float x, const prep_t *p, const float saturation)
{
  const float radius_toe = x;
  const float radius_shoulder = 1.0f - x;
  const float sat2 = 0.5f / sqrtf(saturation);
  const float key_toe = expf(-radius_toe * radius_toe / p->sigma_toe * sat2);
  const float key_shoulder = expf(-radius_shoulder * radius_shoulder / p->sigma_shoulder * sat2);
  return (saturation - (key_toe + key_shoulder) * (saturation));
}

/* linear_saturation(), filmicrgb.c:1193-1196 */
static inline float linear_saturation(const float x, const float luminance, const float saturation)
{
  return luminance + saturation * (x - luminance);
}

static inline float curve_to_display(const float x, const dt_hip_filmicrgb_data_t *d)
{
  const float sp = filmic_spline(x, &d->spline);
  return powf(CLAMPF_(sp, d->spline.y[0], d->spline.y[4]), d->output_power);
}

/* filmic_split_v1() :1534-1571 (v23 == 0) and filmic_split_v2_v3() :1574-1611: the alpha of the output is not
 * written by the reference; it is passed through here */
static inline v4 legacy_split(const v4 pix_in, const int v23, const dt_hip_filmicrgb_data_t *d, const prep_t *p)
{
  v4 temp = pix_in, o = pix_in;
  for(int c = 0; c < 3; c++)
    temp.v[c] = log_tonemapping(fmaxf(pix_in.v[c], NORM_MIN_), d->grey_source, d->black_source, d->dynamic_range);
  const float lum = p->luma[0] * temp.v[0] + p->luma[1] * temp.v[1] + p->luma[2] * temp.v[2];
  const float desaturation = v23 ? desaturate_v2(lum, p, d->saturation) : desaturate_v1(lum, p, d->saturation);
  for(int c = 0; c < 3; c++) o.v[c] = curve_to_display(linear_saturation(temp.v[c], lum, desaturation), d);
  return o;
}

/* filmic_chroma_v1(), filmicrgb.c:1614-1667 */
static inline v4 legacy_chroma_v1(const v4 pix_in, const dt_hip_filmicrgb_data_t *d, const prep_t *p)
{
  v4 ratios, o;
  float norm = fmaxf(pixel_norm(pix_in, d->preserve_color, p), NORM_MIN_);
  for(int c = 0; c < 4; c++) ratios.v[c] = pix_in.v[c] / norm;
  const float min_ratios = fminf(fminf(ratios.v[0], ratios.v[1]), ratios.v[2]);
  if(min_ratios < 0.0f)
    for(int c = 0; c < 4; c++) ratios.v[c] -= min_ratios;
  norm = log_tonemapping(norm, d->grey_source, d->black_source, d->dynamic_range);
  const float desaturation = desaturate_v1(norm, p, d->saturation);
  for(int c = 0; c < 4; c++) ratios.v[c] *= norm;
  const float lum = p->luma[0] * ratios.v[0] + p->luma[1] * ratios.v[1] + p->luma[2] * ratios.v[2];
  for(int c = 0; c < 3; c++) ratios.v[c] = linear_saturation(ratios.v[c], lum, desaturation) / norm;
  norm = curve_to_display(norm, d);
  for(int c = 0; c < 4; c++) o.v[c] = ratios.v[c] * norm;
  return o;
}

/* filmic_chroma_v2_v3(), filmicrgb.c:1670-1737 (v3: d->version == 2) */
static inline v4 legacy_chroma_v2_v3(const v4 pix_in, const dt_hip_filmicrgb_data_t *d, const prep_t *p)
{
  v4 ratios, o;
  float norm = fmaxf(pixel_norm(pix_in, d->preserve_color, p), NORM_MIN_);
  for(int c = 0; c < 4; c++) ratios.v[c] = pix_in.v[c] / norm;
  const float min_ratios = fminf(fminf(ratios.v[0], ratios.v[1]), ratios.v[2]);
  if(min_ratios < 0.0f)
    for(int c = 0; c < 4; c++) ratios.v[c] -= min_ratios;
  norm = log_tonemapping(norm, d->grey_source, d->black_source, d->dynamic_range);
  const float desaturation = desaturate_v2(norm, p, d->saturation);
  norm = curve_to_display(norm, d);
  for(int c = 0; c < 3; c++) ratios.v[c] = fmaxf(ratios.v[c] + (1.0f - ratios.v[c]) * (1.0f - desaturation), 0.0f);
  if(d->version == 2) norm /= fmaxf(pixel_norm(ratios, d->preserve_color, p), NORM_MIN_);
  for(int c = 0; c < 4; c++) o.v[c] = ratios.v[c] * norm;
  const float max_pix = fmaxf(fmaxf(o.v[0], o.v[1]), o.v[2]);
  if(max_pix > 1.0f)
    for(int c = 0; c < 4; c++)
    {
      ratios.v[c] = fmaxf(ratios.v[c] + (1.0f - max_pix), 0.0f);
      o.v[c] = ratios.v[c] * norm;
    }
  return o;
}

static inline v4 agx_compress_negatives(const v4 pix, const float luma[3])
{
  const float input_y = pix.v[0] * luma[0] + pix.v[1] * luma[1] + pix.v[2] * luma[2];
  const float max_rgb = fmaxf(fmaxf(pix.v[0], pix.v[1]), pix.v[2]);
  const float min_rgb = fminf(fminf(pix.v[0], pix.v[1]), pix.v[2]);
  float opp[4];
  for(int c = 0; c < 4; c++) opp[c] = max_rgb - pix.v[c];
  const float opponent_y = opp[0] * luma[0] + opp[1] * luma[1] + opp[2] * luma[2];
  const float max_opponent = fmaxf(fmaxf(opp[0], opp[1]), opp[2]);
  const float y_compensated = max_opponent - opponent_y + input_y;
  const float offset = fmaxf(-min_rgb, 0.f);
  v4 shifted;
  for(int c = 0; c < 4; c++) shifted.v[c] = pix.v[c] + offset;
  const float max_shifted = fmaxf(fmaxf(shifted.v[0], shifted.v[1]), shifted.v[2]);
  float os[4];
  for(int c = 0; c < 4; c++) os[c] = max_shifted - shifted.v[c];
  const float max_opponent_shifted = fmaxf(fmaxf(os[0], os[1]), os[2]);
  const float y_opponent_shifted = os[0] * luma[0] + os[1] * luma[1] + os[2] * luma[2];
  float y_new = shifted.v[0] * luma[0] + shifted.v[1] * luma[1] + shifted.v[2] * luma[2];
  y_new += max_opponent_shifted - y_opponent_shifted;
  const float ratio = (y_new > y_compensated && y_new > 1e-6f) ? y_compensated / y_new : 1.f;
  for(int c = 0; c < 4; c++) shifted.v[c] = shifted.v[c] * ratio;
  return shifted;
}

int oracle_filmicrgb(const dt_hip_piece_t *piece, const dt_hip_filmicrgb_data_t *d, const void *ivoid, void *ovoid)
{
  if(d->version < 0 || d->version > 9) return 1;
  const float *const in = (const float *)ivoid;
  float *const out = (float *)ovoid;
  const size_t npixels = (size_t)piece->roi_out.width * piece->roi_out.height;
  prep_t p;
  prepare(d, &p);
  #pragma omp parallel for schedule(static)
  for(size_t k = 0; k < npixels; k++)
  {
    v4 pix_in, res;
    memcpy(pix_in.v, in + 4 * k, sizeof(pix_in.v));
    if(d->version >= 5)
    {
      /* filmic_agx(), filmicrgb.c:2533-2585 */
      for(int c = 0; c < 3; c++) pix_in.v[c] = isnan(pix_in.v[c]) ? 0.f : CLAMPF_(pix_in.v[c], -1e6f, 1e6f);
      const v4 compressed = agx_compress_negatives(pix_in, p.luma);
      const v4 Ych_original = pipe_RGB_to_Ych(compressed, p.input);
      v4 rendering = mat3(p.inset, compressed);
      rendering = RGB_tone_mapping_v4(rendering, d);
      const v4 pix_out = mat3(p.outset, rendering);
      v4 Ych_final = pipe_RGB_to_Ych(pix_out, p.input);
      const float chroma_final = fminf(Ych_original.v[1], Ych_final.v[1]);
      const float beta_hue = d->agx_beta_hue;
      const float r_mix = beta_hue * Ych_original.v[1] * Ych_original.v[2] + (1.f - beta_hue) * chroma_final * Ych_final.v[2];
      const float g_mix = beta_hue * Ych_original.v[1] * Ych_original.v[3] + (1.f - beta_hue) * chroma_final * Ych_final.v[3];
      const float norm_mix = sqrtf(g_mix * g_mix + r_mix * r_mix);
      v4 Ych_reference = Ych_original;
      Ych_reference.v[2] = (norm_mix > 1e-9f) ? r_mix / norm_mix : Ych_original.v[2];
      Ych_reference.v[3] = (norm_mix > 1e-9f) ? g_mix / norm_mix : Ych_original.v[3];
      Ych_final.v[1] = chroma_final;
      res = gamut_mapping(Ych_final, Ych_reference, &p, 0.f);
    }
    else if(d->version == 4)
    {
      /* filmic_v5(), filmicrgb.c:2270-2298 */
      const v4 naive_rgb = RGB_tone_mapping_v4(pix_in, d);
      const v4 max_rgb = norm_tone_mapping_v4(pix_in, 1, d, &p);
      v4 pix_out;
      for(int c = 0; c < 4; c++) pix_out.v[c] = (0.5f + d->saturation) * max_rgb.v[c];
      for(int c = 0; c < 4; c++) pix_out.v[c] = (0.5f - d->saturation) * naive_rgb.v[c] + pix_out.v[c];
      const v4 Ych_original = pipe_RGB_to_Ych(pix_in, p.input);
      v4 Ych_final = pipe_RGB_to_Ych(pix_out, p.input);
      Ych_final.v[1] = fminf(Ych_original.v[1], Ych_final.v[1]);
      res = gamut_mapping(Ych_final, Ych_original, &p, 0.f);
    }
    else if(d->version < 3)
    {
      /* process(), filmicrgb.c:2862-2887 */
      if(d->preserve_color == 0)
        res = legacy_split(pix_in, d->version != 0, d, &p);
      else
        res = d->version == 0 ? legacy_chroma_v1(pix_in, d, &p) : legacy_chroma_v2_v3(pix_in, d, &p);
    }
    else if(d->preserve_color == 0)
    {
      /* filmic_split_v4(), filmicrgb.c:2222-2242 */
      const v4 pix_out = RGB_tone_mapping_v4(pix_in, d);
      const v4 Ych_original = pipe_RGB_to_Ych(pix_in, p.input);
      v4 Ych_final = pipe_RGB_to_Ych(pix_out, p.input);
      Ych_final.v[1] = fminf(Ych_original.v[1], Ych_final.v[1]);
      res = gamut_mapping(Ych_final, Ych_original, &p, d->saturation);
    }
    else
    {
      /* filmic_chroma_v4(), filmicrgb.c:2175-2196 */
      const v4 pix_out = norm_tone_mapping_v4(pix_in, d->preserve_color, d, &p);
      const v4 Ych_original = pipe_RGB_to_Ych(pix_in, p.input);
      const v4 Ych_final = pipe_RGB_to_Ych(pix_out, p.input);
      res = gamut_mapping(Ych_final, Ych_original, &p, d->saturation);
    }
    memcpy(out + 4 * k, res.v, sizeof(res.v));
  }
  return 0;
}

/* ---- known-answer hooks: the helpers the reference's only hot-path unit test exercises
 * (tests/unittests/iop/test_filmicrgb.c:89-320), exported so tests/test_filmic_kat.py can run the
 * same closed-form expectations against this restatement ---------------------------------------- */
float oracle_kat_clamp_simd(const float x) { return clamp_simd(x); }

float oracle_kat_pixel_norm(const float px[4], const int variant)
{
  prep_t p;
  memset(&p, 0, sizeof(p));
  p.luma[0] = 0.2225045f; /* the no-profile fallback of get_pixel_norm_simd(), filmicrgb.c:1033 */
  p.luma[1] = 0.7168786f;
  p.luma[2] = 0.0606169f;
  const v4 v = { { px[0], px[1], px[2], px[3] } };
  return pixel_norm(v, variant, &p);
}

float oracle_kat_log_tonemapping(const float x, const float grey, const float black, const float dynamic_range)
{
  return log_tonemapping(x, grey, black, dynamic_range);
}

/* test_filmicrgb.c:350-455 and :459-520 */
float oracle_kat_desaturate_v1(const float x, const float sigma_toe, const float sigma_shoulder, const float saturation)
{
  prep_t p;
  memset(&p, 0, sizeof(p));
  p.sigma_toe = sigma_toe;
  p.sigma_shoulder = sigma_shoulder;
  return desaturate_v1(x, &p, saturation);
}

float oracle_kat_linear_saturation(const float x, const float luminance, const float saturation)
{
  return linear_saturation(x, luminance, saturation);
}
