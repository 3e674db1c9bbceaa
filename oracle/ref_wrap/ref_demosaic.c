/* oracle/ref_wrap/ref_demosaic.c -- TEST INFRASTRUCTURE ONLY.
 * Includes the reference's RCD and PPG demosaic translation-unit fragments verbatim
 * (src/iop/demosaic/rcd.c, src/iop/demosaic/ppg.c are #include'd by src/iop/demosaic.c
 * in the reference as well, demosaic.c:1013-1022). */
#include "ref_piece.h"
#include "gen/imageop_math.inc"
#include "gen/demosaic.inc" /* intp(), pre_median helpers the fragments use */

#define INLINE inline
#include "iop/demosaic/rcd.c"
#include "iop/demosaic/ppg.c"

/* dt_rawspeed_crop_dcraw_filters -> rawspeed ColorFilterArray::shiftDcrawFilter
 * (src/imageio/imageio_rawspeed.cc:146-151; rawspeed is an un-vendored submodule).
 * Published algorithm: the dcraw filter word holds 8 rows x 2 columns of 2-bit colours;
 * shifting the pattern origin by (x, y) rotates rows by y (mod 8) and swaps the two column
 * entries of every row when x is odd.  tests/test_filters.py pins it against the identity
 * FC(r + y, c + x, f) == FC(r, c, shifted) the reference's call sites rely on. */
void amaze_demosaic_RT(const dt_dev_pixelpipe_iop_t *piece, const float *const in, float *out,
                       const dt_iop_roi_t *const roi_in, const dt_iop_roi_t *const roi_out, const int filters);

uint32_t ref_shift_dcraw_filters(uint32_t filters, uint32_t x, uint32_t y)
{
  if(!filters || filters == 9u) return filters;
  uint32_t out = 0;
  for(int r = 0; r < 8; r++)
    for(int c = 0; c < 2; c++)
    {
      const uint32_t col = FC(r + y, c + x, filters);
      out |= col << (((r << 1 & 14) + (c & 1)) << 1);
    }
  return out;
}

int ref_demosaic(const dt_hip_piece_t *v, const dt_hip_demosaic_data_t *d, const void *in, void *out)
{
  ref_reset_fp_mode();
  dt_dev_pixelpipe_iop_t piece;
  ref_fill_piece(&piece, v, NULL);
  dt_iop_roi_t roi = piece.roi_in;
  dt_iop_roi_t roo = piece.roi_out;
  roo.x = roo.y = 0;
  const uint32_t filters = ref_shift_dcraw_filters(v->filters, piece.roi_in.x, piece.roi_in.y);
  /* the optional steps of process(), demosaic.c:1137-1250, in its order */
  float *geq = NULL;
  float *aux = NULL;
  if(d->green_eq)
  {
    const size_t bytes = sizeof(float) * (size_t)roi.width * roi.height;
    geq = (float *)malloc(bytes);
    if(d->green_eq == 3) aux = (float *)malloc(bytes);
    if(!geq || (d->green_eq == 3 && !aux) || d->green_eq > 3) return 1;
    if(d->green_eq >= 2)
      green_equilibration_favg(aux ? aux : geq, (const float *)in, roi.width, roi.height, v->filters, roi.x, roi.y);
    if(d->green_eq & 1)
      green_equilibration_lavg(geq, aux ? aux : (const float *)in, roi.width, roi.height, v->filters, roi.x, roi.y,
                               d->green_eq_threshold);
    in = geq;
  }
  int rc = 0;
  if(d->demosaicing_method == DT_HIP_DEMOSAIC_RCD)
    rcd_demosaic(&piece, (float *)out, (const float *)in, &roo, &roi, filters);
  else if(d->demosaicing_method == DT_HIP_DEMOSAIC_AMAZE)
    amaze_demosaic_RT(&piece, (const float *)in, (float *)out, &roi, &roo, filters);
  else if(d->demosaicing_method == DT_HIP_DEMOSAIC_PPG)
    rc = demosaic_ppg((float *)out, (const float *)in, &roo, &roi, filters, d->median_thrs);
  else
    rc = 1;
  free(geq);
  free(aux);
  if(rc == 0 && d->color_smoothing) color_smoothing((float *)out, &roo, (int)d->color_smoothing);
  return rc;
}
