/* oracle/ref_wrap/ref_color.c -- TEST INFRASTRUCTURE ONLY.
 * The reference's colorin/colorout matrix path (_apply_matrix, src/colorprofiles/conversion.c)
 * and color calibration loop (loop_switch, src/iop/channelmixerrgb.c), lifted verbatim at build
 * time, behind C entry points that take the C-ABI structs of include/ansel_hip.h. */
#include "ref_piece.h"
#include "common/colorspaces_inline_conversions.h"
#include "pixel/chromatic_adaptation.h"

/* handles of lcms2 the conversion struct carries but the matrix path never touches */
typedef void *cmsHTRANSFORM;
typedef void *cmsHPROFILE;
typedef int dt_colorspaces_color_profile_type_t;

typedef struct dt_colorspaces_conversion_t dt_colorspaces_conversion_t; /* conversion.h:63 */
#include "gen/conversion_h.inc"
#include "gen/iop_profile.inc"
#include "gen/conversion.inc"
#include "gen/colorin.inc"

static int ref_conversion(const dt_hip_piece_t *v, const dt_hip_conversion_t *d, const float *in, float *out)
{
  ref_reset_fp_mode();
  struct dt_colorspaces_conversion_t c;
  memset(&c, 0, sizeof(c));
  c.is_matrix = TRUE;
  c.has_clipping = d->has_clipping;
  for(int r = 0; r < 3; r++)
    for(int k = 0; k < 3; k++)
    {
      c.matrix[r][k] = d->matrix[r][k];
      c.clip_matrix[r][k] = d->clip_matrix[r][k];
      c.coeffs_source[r][k] = d->coeffs_source[r][k];
      c.coeffs_target[r][k] = d->coeffs_target[r][k];
    }
  for(int k = 0; k < 3; k++)
  {
    c.lut_source[k] = (float *)d->lut_source[k];
    c.lut_target[k] = (float *)d->lut_target[k];
  }
  c.nonlinear_source = d->nonlinear_source;
  c.nonlinear_target = d->nonlinear_target;
  _apply_matrix(&c, in, out, (size_t)v->roi_out.width * v->roi_out.height, d->blue_mapping ? apply_blue_mapping : NULL);
  return 0;
}

int ref_colorin(const dt_hip_piece_t *v, const dt_hip_conversion_t *d, const void *in, void *out)
{
  return ref_conversion(v, d, (const float *)in, (float *)out);
}

int ref_colorout(const dt_hip_piece_t *v, const dt_hip_conversion_t *d, const void *in, void *out)
{
  return ref_conversion(v, d, (const float *)in, (float *)out);
}

/* ---- color calibration ------------------------------------------------------------- */
#include "gen/channelmixerrgb.inc"

int ref_channelmixerrgb(const dt_hip_piece_t *v, const dt_hip_channelmixerrgb_data_t *d, const void *in, void *out)
{
  ref_reset_fp_mode();
  dt_colormatrix_t XYZ_to_RGB = { { 0.f } }, RGB_to_XYZ = { { 0.f } }, MIX = { { 0.f } };
  for(int r = 0; r < 3; r++)
    for(int k = 0; k < 4; k++)
    {
      XYZ_to_RGB[r][k] = d->XYZ_to_RGB[r][k];
      RGB_to_XYZ[r][k] = d->RGB_to_XYZ[r][k];
      MIX[r][k] = d->MIX[r][k];
    }
  dt_aligned_pixel_t illuminant, saturation, lightness, grey;
  for(int k = 0; k < 4; k++)
  {
    illuminant[k] = d->illuminant[k];
    saturation[k] = d->saturation[k];
    lightness[k] = d->lightness[k];
    grey[k] = d->grey[k];
  }
  loop_switch((const float *)in, (float *)out, v->roi_out.width, v->roi_out.height, 4, XYZ_to_RGB, RGB_to_XYZ, MIX,
              illuminant, saturation, lightness, grey, d->p, d->gamut, d->clip, d->apply_grey,
              (dt_adaptation_t)d->adaptation, (dt_iop_channelmixer_rgb_version_t)d->version);
  return 0;
}
