/* oracle/ref_wrap/ref_lab.c -- TEST INFRASTRUCTURE ONLY.
 * The colourspace glue the pixelpipe runs around Lab modules (pixelpipe_cpu.c:59-75 ->
 * dt_ioppr_transform_image_colorspace(), src/colorprofiles/iop_profile.c:540-596): the matrix paths
 * _transform_rgb_to_lab_matrix() / _transform_lab_to_rgb_matrix() (:377-463) and _apply_tonecurves() (:332-372),
 * lifted verbatim. */
#include "ref_piece.h"
#include "common/colorspaces_inline_conversions.h"

typedef int dt_colorspaces_color_profile_type_t;
typedef int dt_colorspaces_color_mode_t;
typedef enum dt_iop_color_intent_t { DT_INTENT_PERCEPTUAL = 0 } dt_iop_color_intent_t;
#define DT_IOP_COLOR_ICC_LEN 512
#include "gen/iop_profile.inc"
#include "gen/iop_profile_info.inc"
#include "gen/iop_profile_c.inc"

static void fill_profile(dt_iop_order_iccprofile_info_t *pi, const dt_hip_lab_data_t *d, const int inverse)
{
  static float linear[4] = { -1.0f, 0.f, 0.f, 0.f };
  const float(*m)[4] = d->matrix;
  memset(pi, 0, sizeof(*pi));
  pi->nonlinearlut = d->nonlinearlut;
  pi->lutsize = 0x10000;
  for(int k = 0; k < 3; k++)
  {
    /* the work profile's curves (host buffers here): lut_in ahead of the matrix, lut_out behind it */
    float *const curve = (d->nonlinearlut && d->lut[k]) ? (float *)d->lut[k] : linear;
    pi->lut_in[k] = inverse ? linear : curve;
    pi->lut_out[k] = inverse ? curve : linear;
    for(int c = 0; c < 3; c++)
    {
      pi->unbounded_coeffs_in[k][c] = d->unbounded_coeffs[k][c];
      pi->unbounded_coeffs_out[k][c] = d->unbounded_coeffs[k][c];
    }
  }
  /* the functions read the TRANSPOSED matrices (row j = column j of the 3x3) */
  for(int r = 0; r < 3; r++)
    for(int c = 0; c < 3; c++)
    {
      if(inverse) pi->matrix_out_transposed[c][r] = m[r][c];
      else pi->matrix_in_transposed[c][r] = m[r][c];
    }
}

/* in place, like the pixelpipe does it (the conversion leaves the alpha of the output untouched) */
int ref_rgb_to_lab(const dt_hip_piece_t *v, const dt_hip_lab_data_t *d, const void *in, void *out_)
{
  const int width = v->roi_out.width, height = v->roi_out.height;
  float *out = (float *)out_;
  ref_reset_fp_mode();
  dt_iop_order_iccprofile_info_t pi;
  fill_profile(&pi, d, 0);
  memcpy(out, in, sizeof(float) * 4 * (size_t)width * height);
  _transform_rgb_to_lab_matrix(out, out, width, height, &pi);
  return 0;
}

int ref_lab_to_rgb(const dt_hip_piece_t *v, const dt_hip_lab_data_t *d, const void *in, void *out_)
{
  const int width = v->roi_out.width, height = v->roi_out.height;
  float *out = (float *)out_;
  ref_reset_fp_mode();
  dt_iop_order_iccprofile_info_t pi;
  fill_profile(&pi, d, 1);
  _transform_lab_to_rgb_matrix((const float *)in, out, width, height, &pi);
  return 0;
}
