/* oracle/ref_wrap/ref_finalscale.c -- TEST INFRASTRUCTURE ONLY.
 * finalscale: process() of src/iop/finalscale.c is five lines around dt_iop_clip_and_zoom_roi()
 * (src/develop/imageop_math.c:146-152), which is two lines around dt_interpolation_resample_roi();
 * the resampler itself (src/pixel/interpolation.c) is compiled from where it lies. */
#include "ref_piece.h"
#include "pixel/interpolation.h"

void ref_set_interpolator(const char *name);

int ref_finalscale(const dt_hip_piece_t *v, const dt_hip_finalscale_data_t *d, const void *in, void *out)
{
  ref_reset_fp_mode();
  static const char *names[3] = { "bilinear", "bicubic", "mitchell" };
  if(d->interpolation < 0 || d->interpolation > 2) return 1;
  ref_set_interpolator(names[d->interpolation]);
  dt_dev_pixelpipe_iop_t piece;
  ref_fill_piece(&piece, v, NULL);
  /* finalscale.c:120-129 */
  dt_iop_roi_t roi_in = piece.roi_in, roi_out = piece.roi_out;
  roi_in.x = roi_in.y = roi_out.x = roi_out.y = 0;
  /* imageop_math.c:150-151 */
  const struct dt_interpolation *itor = dt_interpolation_new(DT_INTERPOLATION_USERPREF);
  dt_interpolation_resample_roi(itor, (float *)out, &roi_out, (const float *)in, &roi_in);
  return 0;
}

/* initialscale: process(), src/iop/initialscale.c:120-127: the regions go to the resampler as they are */
int ref_initialscale(const dt_hip_piece_t *v, const dt_hip_finalscale_data_t *d, const void *in, void *out)
{
  ref_reset_fp_mode();
  static const char *names[3] = { "bilinear", "bicubic", "mitchell" };
  if(d->interpolation < 0 || d->interpolation > 2) return 1;
  ref_set_interpolator(names[d->interpolation]);
  dt_dev_pixelpipe_iop_t piece;
  ref_fill_piece(&piece, v, NULL);
  const struct dt_interpolation *itor = dt_interpolation_new(DT_INTERPOLATION_USERPREF);
  dt_interpolation_resample_roi(itor, (float *)out, &piece.roi_out, (const float *)in, &piece.roi_in);
  return 0;
}
