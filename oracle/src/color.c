/* oracle/src/color.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of the colour modules of the export pipe:
 *   colorin / colorout  matrix branch of dt_colorspaces_apply_conversion():
 *                       _apply_matrix(), src/colorprofiles/conversion.c:593-682,
 *                       _apply_target_curves(), :546-582, dt_ioppr_eval_trc() and friends,
 *                       src/colorprofiles/iop_profile.h:537-580, apply_blue_mapping(),
 *                       src/iop/colorin.c:690-709
 *   color calibration   loop_switch(), src/iop/channelmixerrgb.c:766-960 with gamut_mapping()
 *                       :642-706, luma_chroma() :707-763 and the adaptation helpers of
 *                       src/pixel/chromatic_adaptation.h
 *
 * Arithmetic: binary32, one rounding per operation; dt_mat3x4_mul_vec4() (src/system/simd.h:188-197)
 * is mul, mul+add, mul+add per lane, unfused; DT_FMA() (src/math/math.h:59-65) is a true fmaf()
 * as on any FMA-capable x86 target.
 */
#include <math.h>
#include <string.h>
#include "oracle.h"

#define NORM_MIN 1.52587890625e-05f /* src/math/math.h:37 */
#define INVERSE_SQRT_3 0.5773502691896258f

typedef struct { float v[4]; } px_t;

/* dt_mat3x4_mul_vec4 with the rows of the (untransposed) matrix m: out[i] = (m[i][0]*x + m[i][1]*y) + m[i][2]*z,
 * lane 3 = 0*x + 0*y + 0*z */
static inline px_t mat3(const float m[3][4], const px_t in)
{
  px_t o;
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

static inline float lut_lerp(const float *lut, const float v)
{
  /* extrapolate_lut(), iop_profile.h:537-547 with lutsize = 0x10000 */
  const float a = v * 65535.0f;
  const float ft = a > 0.0f ? (a < 65535.0f ? a : 65535.0f) : 0.0f;
  const int t = (ft < 65534.0f) ? ft : 65534.0f;
  const float f = ft - t;
  return lut[t] * (1.0f - f) + lut[t + 1] * f;
}

static inline float eval_trc(const float x, const float *lut, const float coeff[3])
{
  /* dt_ioppr_eval_trc(), iop_profile.h:577-580; eval_exp(), :558-562 */
  return (x < 1.0f) ? lut_lerp(lut, x) : coeff[1] * powf(x * coeff[0], coeff[2]);
}

static inline float clamp_glib(const float x, const float lo, const float hi) { return x > hi ? hi : (x < lo ? lo : x); }

static int conversion(const dt_hip_piece_t *piece, const dt_hip_conversion_t *d, const float *in, float *out)
{
  const size_t npixels = (size_t)piece->roi_out.width * piece->roi_out.height;
  const int decode = d->lut_source[0] != NULL && d->nonlinear_source > 0;
  const int encode = d->lut_target[0] != NULL && d->nonlinear_target > 0;
  const float *ls[3], *lt[3];
  for(int c = 0; c < 3; c++)
  {
    ls[c] = (const float *)d->lut_source[c];
    lt[c] = (const float *)d->lut_target[c];
  }
  #pragma omp parallel for schedule(static)
  for(size_t k = 0; k < npixels; k++)
  {
    px_t s;
    for(int c = 0; c < 3; c++)
      s.v[c] = (decode && ls[c][0] >= 0.0f) ? eval_trc(in[4 * k + c], ls[c], d->coeffs_source[c]) : in[4 * k + c];
    s.v[3] = 0.0f;
    if(d->blue_mapping)
    {
      /* apply_blue_mapping(), colorin.c:690-709 */
      const float YY = s.v[0] + s.v[1] + s.v[2];
      if(YY > 0.0f)
      {
        const float zz = s.v[2] / YY;
        const float bound_z = 0.5f, bound_Y = 0.5f, amount = 0.11f;
        if(zz > bound_z)
        {
          const float t = (zz - bound_z) / (1.0f - bound_z) * fminf(1.0, YY / bound_Y);
          s.v[1] += t * amount;
          s.v[2] -= t * amount;
        }
      }
    }
    px_t v = mat3(d->matrix, s);
    if(d->has_clipping)
    {
      px_t c = { { clamp_glib(v.v[0], 0.f, 1.f), clamp_glib(v.v[1], 0.f, 1.f), clamp_glib(v.v[2], 0.f, 1.f), 0.f } };
      v = mat3(d->clip_matrix, c);
    }
    if(encode)
      for(int c = 0; c < 3; c++)
        if(lt[c][0] >= 0.0f) v.v[c] = eval_trc(v.v[c], lt[c], d->coeffs_target[c]);
    memcpy(out + 4 * k, v.v, sizeof(v.v));
  }
  return 0;
}

int oracle_colorin(const dt_hip_piece_t *piece, const dt_hip_conversion_t *d, const void *in, void *out)
{
  return conversion(piece, d, (const float *)in, (float *)out);
}

int oracle_colorout(const dt_hip_piece_t *piece, const dt_hip_conversion_t *d, const void *in, void *out)
{
  return conversion(piece, d, (const float *)in, (float *)out);
}

/* ---- color calibration ------------------------------------------------------------------ */

/* src/pixel/chromatic_adaptation.h:45-95, rows of the untransposed matrices */
static const float XYZ_to_Bradford_LMS[3][4] = { { 0.8951f, 0.2664f, -0.1614f, 0.f },
                                                 { -0.7502f, 1.7135f, 0.0367f, 0.f },
                                                 { 0.0389f, -0.0685f, 1.0296f, 0.f } };
static const float Bradford_LMS_to_XYZ[3][4] = { { 0.9870f, -0.1471f, 0.1600f, 0.f },
                                                 { 0.4323f, 0.5184f, 0.0493f, 0.f },
                                                 { -0.0085f, 0.0400f, 0.9685f, 0.f } };
static const float XYZ_to_CAT16_LMS[3][4] = { { 0.401288f, 0.650173f, -0.051461f, 0.f },
                                              { -0.250268f, 1.204414f, 0.045854f, 0.f },
                                              { -0.002079f, 0.048952f, 0.953127f, 0.f } };
static const float CAT16_LMS_to_XYZ[3][4] = { { 1.862068f, -1.011255f, 0.149187f, 0.f },
                                              { 0.38752f, 0.621447f, -0.008974f, 0.f },
                                              { -0.015841f, -0.034123f, 1.049964f, 0.f } };

static inline px_t max_zero(px_t v)
{
  /* dt_simd_max_zero(), simd.h:115-122 */
  for(int c = 0; c < 4; c++) v.v[c] = isfinite(v.v[c]) ? (v.v[c] > 0.0f ? v.v[c] : 0.0f) : 0.f;
  return v;
}

static inline px_t downscale(px_t v, const float scaling)
{
  /* _downscale_vector_simd(), chromatic_adaptation.h:316-321 */
  const int valid = (scaling > NORM_MIN) && !isnan(scaling);
  const float dv = valid ? (scaling + NORM_MIN) : NORM_MIN;
  for(int c = 0; c < 4; c++) v.v[c] = v.v[c] / dv;
  return v;
}

static inline px_t upscale(px_t v, const float scaling)
{
  const int valid = (scaling > NORM_MIN) && !isnan(scaling);
  const float m = valid ? (scaling + NORM_MIN) : NORM_MIN;
  for(int c = 0; c < 4; c++) v.v[c] = v.v[c] * m;
  return v;
}

static inline px_t xyz_to_lms(const px_t v, const int kind)
{
  if(kind == DT_HIP_ADAPTATION_FULL_BRADFORD || kind == DT_HIP_ADAPTATION_LINEAR_BRADFORD) return mat3(XYZ_to_Bradford_LMS, v);
  if(kind == DT_HIP_ADAPTATION_CAT16) return mat3(XYZ_to_CAT16_LMS, v);
  return v;
}

static inline px_t lms_to_xyz(const px_t v, const int kind)
{
  if(kind == DT_HIP_ADAPTATION_FULL_BRADFORD || kind == DT_HIP_ADAPTATION_LINEAR_BRADFORD) return mat3(Bradford_LMS_to_XYZ, v);
  if(kind == DT_HIP_ADAPTATION_CAT16) return mat3(CAT16_LMS_to_XYZ, v);
  return v;
}

/* gamut_mapping(), channelmixerrgb.c:642-706 */
static inline px_t gamut_mapping(const px_t input, const float compression, const int clip)
{
  px_t o = { { 0.f, 0.f, 0.f, 0.f } };
  const float sum = input.v[0] + input.v[1] + input.v[2];
  const float Y = input.v[1];
  if(sum > 0.f && Y > 0.f)
  {
    float x = input.v[0] / sum;
    float y = input.v[1] / sum;
    const float uv_denominator = -2.f * x + 12.f * y + 3.f;
    float u = 4.f * x / uv_denominator;
    float v = 9.f * y / uv_denominator;
    const float D50[2] = { 0.20915914598542354f, 0.488075320769787f };
    const float delta[2] = { D50[0] - u, D50[1] - v };
    const float Delta = Y * (delta[0] * delta[0] + delta[1] * delta[1]);
    const float correction = (compression == 0.0f) ? 0.f : powf(Delta, compression);
    const float tmp_u = fmaf(correction, delta[0], u);
    const float tmp_v = fmaf(correction, delta[1], v);
    u = (u > D50[0]) ? fmaxf(tmp_u, D50[0]) : fminf(tmp_u, D50[0]);
    v = (v > D50[1]) ? fmaxf(tmp_v, D50[1]) : fminf(tmp_v, D50[1]);
    const float xy_denominator = 6.f * u - 16.f * v + 12.f;
    x = 9.f * u / xy_denominator;
    y = 4.f * v / xy_denominator;
    if(clip)
    {
      x = fmaxf(x, 0.0f);
      y = fmaxf(y, 0.0f);
    }
    y = fmaxf(y, NORM_MIN);
    const float scale = x + y;
    if(scale >= 1.f)
    {
      x /= scale;
      y /= scale;
    }
    o.v[0] = Y * x / y;
    o.v[1] = Y;
    o.v[2] = Y * (1.f - x - y) / y;
    o.v[3] = 0.f;
  }
  return o;
}

/* luma_chroma(), channelmixerrgb.c:707-763 */
static inline px_t luma_chroma(const px_t input, const float saturation[4], const float lightness[4], const int version)
{
  px_t output;
  /* euclidean_norm(), math.h:208-211 */
  float norm = fmaxf(sqrtf(input.v[0] * input.v[0] + input.v[1] * input.v[1] + input.v[2] * input.v[2]), NORM_MIN);
  const float avg = fmaxf((input.v[0] + input.v[1] + input.v[2]) / 3.0f, NORM_MIN);
  if(norm > 0.f && avg > 0.f)
  {
    /* scalar_product(), math.h:186-194: acc = 0; acc += v1[c] * v2[c] for c = 0..2 */
    float mix = 0.f;
    for(int c = 0; c < 3; c++) mix += input.v[c] * lightness[c];
    if(version == 2) norm *= INVERSE_SQRT_3;
    for(int c = 0; c < 3; c++) output.v[c] = input.v[c] / norm;
    float coeff_ratio = 0.f;
    if(version == 0)
    {
      for(int c = 0; c < 3; c++) coeff_ratio += (1.0f - output.v[c]) * (1.0f - output.v[c]) * saturation[c];
    }
    else
    {
      float sp = 0.f;
      for(int c = 0; c < 3; c++) sp += output.v[c] * saturation[c];
      coeff_ratio = sp / 3.f;
    }
    for(int c = 0; c < 3; c++)
    {
      const float min_ratio = (output.v[c] < 0.0f) ? output.v[c] : 0.0f;
      const float output_inverse = 1.0f - output.v[c];
      output.v[c] = fmaxf(fmaf(output_inverse, coeff_ratio, output.v[c]), min_ratio);
    }
    if(version == 2)
    {
      const float n2 = fmaxf(sqrtf(output.v[0] * output.v[0] + output.v[1] * output.v[1] + output.v[2] * output.v[2]), NORM_MIN);
      norm /= n2 * INVERSE_SQRT_3;
    }
    norm *= fmaxf(1.f + mix / avg, 0.f);
    for(int c = 0; c < 3; c++) output.v[c] *= norm;
  }
  else
  {
    for(int c = 0; c < 3; c++) output.v[c] = input.v[c];
  }
  output.v[3] = 0.f; /* lane 3 of the reference's luma_output is never read */
  return output;
}

int oracle_channelmixerrgb(const dt_hip_piece_t *piece, const dt_hip_channelmixerrgb_data_t *d, const void *ivoid, void *ovoid)
{
  const float *const in = (const float *)ivoid;
  float *const out = (float *)ovoid;
  const size_t npixels = (size_t)piece->roi_out.width * piece->roi_out.height;
  const int kind = d->adaptation, clip = d->clip;
  /* pre-computed adaptation white points, chromatic_adaptation.h:199-260 */
  static const float bradford_D50[4] = { 0.996078f, 1.020646f, 0.818155f, 0.f };
  static const float cat16_D50[4] = { 0.994535f, 1.000997f, 0.833036f, 0.f };
  static const float xyz_D50[4] = { 0.9642119944211994f, 1.0f, 0.8251882845188288f, 0.f };
  #pragma omp parallel for schedule(static)
  for(size_t k = 0; k < npixels; k++)
  {
    px_t in_v, one, two;
    memcpy(in_v.v, in + 4 * k, sizeof(in_v.v));
    two = clip ? max_zero(in_v) : in_v;
    switch(kind)
    {
      case DT_HIP_ADAPTATION_FULL_BRADFORD:
      case DT_HIP_ADAPTATION_LINEAR_BRADFORD:
      {
        one = mat3(d->RGB_to_XYZ, two);
        const float Y = one.v[1];
        two = downscale(mat3(XYZ_to_Bradford_LMS, one), Y);
        /* bradford_adapt_D50(), chromatic_adaptation.h:209-217 */
        px_t t;
        for(int c = 0; c < 4; c++) t.v[c] = two.v[c] / d->illuminant[c];
        if(kind == DT_HIP_ADAPTATION_FULL_BRADFORD) t.v[2] = (t.v[2] > 0.f) ? powf(t.v[2], d->p) : t.v[2];
        for(int c = 0; c < 4; c++) t.v[c] = bradford_D50[c] * t.v[c];
        one = upscale(t, Y);
        two = mat3(d->MIX, one);
        one = mat3(Bradford_LMS_to_XYZ, two);
        break;
      }
      case DT_HIP_ADAPTATION_CAT16:
      {
        one = mat3(d->RGB_to_XYZ, two);
        const float Y = one.v[1];
        two = downscale(mat3(XYZ_to_CAT16_LMS, one), Y);
        /* CAT16_adapt_D50(..., 1.0f, TRUE): lms * D50 / illuminant, chromatic_adaptation.h:236-243 */
        px_t t;
        for(int c = 0; c < 4; c++) t.v[c] = two.v[c] * cat16_D50[c] / d->illuminant[c];
        one = upscale(t, Y);
        two = mat3(d->MIX, one);
        one = mat3(CAT16_LMS_to_XYZ, two);
        break;
      }
      case DT_HIP_ADAPTATION_XYZ:
      {
        one = mat3(d->RGB_to_XYZ, two);
        const float Y = one.v[1];
        px_t t = downscale(one, Y);
        for(int c = 0; c < 4; c++) t.v[c] = t.v[c] * xyz_D50[c] / d->illuminant[c]; /* XYZ_adapt_D50() */
        two = upscale(t, Y);
        one = mat3(d->MIX, two);
        break;
      }
      default:
      {
        one = mat3(d->MIX, two);
        one = mat3(d->RGB_to_XYZ, one);
        break;
      }
    }
    two = gamut_mapping(one, d->gamut, clip);
    if(kind == DT_HIP_ADAPTATION_RGB)
      one = mat3(d->XYZ_to_RGB, two);
    else
      one = xyz_to_lms(two, kind);
    if(clip) one = max_zero(one);
    two = luma_chroma(one, d->saturation, d->lightness, d->version);
    if(clip) two = max_zero(two);
    if(d->apply_grey)
    {
      const float grey_mix = fmaxf(two.v[0] * d->grey[0] + two.v[1] * d->grey[1] + two.v[2] * d->grey[2], 0.0f);
      out[4 * k + 0] = out[4 * k + 1] = out[4 * k + 2] = grey_mix;
      out[4 * k + 3] = in_v.v[3];
    }
    else
    {
      if(kind == DT_HIP_ADAPTATION_RGB)
        one = mat3(d->RGB_to_XYZ, two);
      else
        one = lms_to_xyz(two, kind);
      if(clip) one = max_zero(one);
      two = mat3(d->XYZ_to_RGB, one);
      if(clip) two = max_zero(two);
      out[4 * k + 0] = two.v[0];
      out[4 * k + 1] = two.v[1];
      out[4 * k + 2] = two.v[2];
      out[4 * k + 3] = in_v.v[3];
    }
  }
  return 0;
}


/* ---- RGB <-> Lab glue around Lab modules ----------------------------------------------------
 * dt_ioppr_transform_image_colorspace() for a linear matrix profile:
 * _transform_rgb_to_lab_matrix() / _transform_lab_to_rgb_matrix(), src/colorprofiles/iop_profile.c:377-463,
 * with dt_XYZ_to_Lab() / dt_Lab_to_XYZ(), lab_f() / lab_f_inv(), cbrt_5f(), cbrta_halleyf() of
 * src/common/colorspaces_inline_conversions.h:50-106.  m = RGB -> XYZ(D50) (resp. XYZ -> RGB), rows. */
static const float LAB_D50[3] = { 0.9642f, 1.0f, 0.8249f };

static inline float lab_f_(const float x)
{
  const float epsilon = 216.0f / 24389.0f, kappa = 24389.0f / 27.0f;
  if(!(x > epsilon)) return (kappa * x + 16.0f) / 116.0f;
  union { float f; uint32_t u; } b;
  b.f = x;
  b.u = b.u / 3 + 709921077u;
  const float a = b.f, a3 = a * a * a;
  return a * (a3 + x + x) / (a3 + a3 + x);
}

static inline float lab_f_inv_(const float x)
{
  const float epsilon = 0.20689655172413796f, kappa = 24389.0f / 27.0f;
  return (x > epsilon) ? x * x * x : (116.0f * x - 16.0f) / kappa;
}

/* _apply_tonecurves(), src/colorprofiles/iop_profile.c:332-372, on one pixel: the channels that have a curve */
static inline void lab_curves(const dt_hip_lab_data_t *d, float v[4])
{
  if(!d->nonlinearlut) return;
  for(int c = 0; c < 3; c++)
    if(d->lut[c] && d->lut_first[c] >= 0.0f) v[c] = eval_trc(v[c], (const float *)d->lut[c], d->unbounded_coeffs[c]);
}

int oracle_rgb_to_lab(const dt_hip_piece_t *piece, const dt_hip_lab_data_t *d, const void *in_, void *out_)
{
  const float *in = (const float *)in_;
  float *out = (float *)out_;
  const float(*m)[4] = d->matrix;
  const size_t n = (size_t)piece->roi_out.width * piece->roi_out.height;
  #pragma omp parallel for schedule(static)
  for(size_t k = 0; k < n; k++)
  {
    px_t p;
    for(int c = 0; c < 4; c++) p.v[c] = in[4 * k + c];
    lab_curves(d, p.v); /* iop_profile.c:389-393 */
    const px_t xyz = mat3(m, p);
    float f[3];
    for(int i = 0; i < 3; i++) f[i] = lab_f_(xyz.v[i] / LAB_D50[i]);
    out[4 * k + 0] = 116.0f * f[1] - 16.0f;
    out[4 * k + 1] = 500.0f * (f[0] - f[1]);
    out[4 * k + 2] = 200.0f * (f[1] - f[2]);
    out[4 * k + 3] = in[4 * k + 3]; /* the pipe converts in place: alpha is left as it was */
  }
  return 0;
}

int oracle_lab_to_rgb(const dt_hip_piece_t *piece, const dt_hip_lab_data_t *d, const void *in_, void *out_)
{
  const float *in = (const float *)in_;
  float *out = (float *)out_;
  const float(*m)[4] = d->matrix;
  const size_t n = (size_t)piece->roi_out.width * piece->roi_out.height;
  #pragma omp parallel for schedule(static)
  for(size_t k = 0; k < n; k++)
  {
    const float fy = (in[4 * k] + 16.0f) / 116.0f;
    const float fx = in[4 * k + 1] / 500.0f + fy;
    const float fz = fy - in[4 * k + 2] / 200.0f;
    px_t xyz;
    xyz.v[0] = LAB_D50[0] * lab_f_inv_(fx);
    xyz.v[1] = LAB_D50[1] * lab_f_inv_(fy);
    xyz.v[2] = LAB_D50[2] * lab_f_inv_(fz);
    xyz.v[3] = 0.0f;
    px_t rgb = mat3(m, xyz);
    lab_curves(d, rgb.v); /* iop_profile.c:455-462 */
    for(int c = 0; c < 3; c++) out[4 * k + c] = rgb.v[c];
    out[4 * k + 3] = in[4 * k + 3];
  }
  return 0;
}
