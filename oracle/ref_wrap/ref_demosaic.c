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
#include "iop/demosaic/passthrough.c"

/* VNG4 and the dual demosaic (src/iop/demosaic/vng.c:34-221, dual.c:35-110): the two functions lifted as they are; what
 * they call beyond the pixel code is a timer, a log line and the detail-mask functions of src/develop/masks/detail.c, which
 * ref_blend.c holds (lifted from there) */
#include <limits.h>
#define FILTERS_ARE_4BAYER(filters) 0 /* CYGM / RGBE sensors: not a Bayer RGB mosaic */
typedef struct dt_times_t { double clock, user; } dt_times_t;
static inline void dt_get_times(dt_times_t *t) { t->clock = t->user = 0.0; }
static inline unsigned int ref_no_debug_flags(void) { return 0; }
#define dt_get_debug_flags ref_no_debug_flags
#ifndef DT_DEBUG_DEMOSAIC
#define DT_DEBUG_DEMOSAIC 0
#endif
#ifndef DT_DEBUG_PERF
#define DT_DEBUG_PERF 0
#endif
#ifndef DT_DEV_PIXELPIPE_FULL
#define DT_DEV_PIXELPIPE_FULL 2
#endif
#ifndef DT_DEV_PIXELPIPE_DISPLAY_PASSTHRU
#define DT_DEV_PIXELPIPE_DISPLAY_PASSTHRU 4
#endif
void dt_masks_calc_rawdetail_mask(float *const restrict src, float *const restrict mask, float *const restrict tmp,
                                  const int width, const int height, const dt_aligned_pixel_t wb);
void dt_masks_calc_detail_mask(float *const restrict src, float *const restrict out, float *const restrict tmp, const int width,
                               const int height, const float threshold, const gboolean detail);
#include "gen/demosaic_vng.inc"
#include "gen/demosaic_dual.inc"

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
  const void *const in0 = in;
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
  const uint32_t method = d->demosaicing_method & ~(uint32_t)DT_HIP_DEMOSAIC_DUAL;
  /* demosaic.c:1111-1118: in front of the Bayer branch, on `pixels` */
  if(d->demosaicing_method == DT_HIP_DEMOSAIC_PASSTHROUGH_MONOCHROME)
    passthrough_monochrome((float *)out, (const float *)in0, &roo, &roi);
  else if(d->demosaicing_method == DT_HIP_DEMOSAIC_PASSTHROUGH_COLOR)
    passthrough_color((float *)out, (const float *)in0, &roo, &roi, v->filters, NULL);
  else if(method == DT_HIP_DEMOSAIC_RCD)
    rcd_demosaic(&piece, (float *)out, (const float *)in, &roo, &roi, filters);
  else if(method == DT_HIP_DEMOSAIC_AMAZE)
    amaze_demosaic_RT(&piece, (const float *)in, (float *)out, &roi, &roo, filters);
  else if(method == DT_HIP_DEMOSAIC_PPG)
    rc = demosaic_ppg((float *)out, (const float *)in, &roo, &roi, filters, d->median_thrs);
  else if(d->demosaicing_method == DT_HIP_DEMOSAIC_VNG4)
    rc = vng_interpolate((float *)out, (const float *)in, &roo, &roi, v->filters, NULL, FALSE);
  else
    rc = 1;
  if(rc == 0 && (d->demosaicing_method & DT_HIP_DEMOSAIC_DUAL))
  {
    /* demosaic.c:1243-1247: `pixels`, the mosaic as the module received it, not the green-equilibrated copy */
    dt_dev_pixelpipe_t pipe;
    memset(&pipe, 0, sizeof(pipe));
    for(int c = 0; c < 4; c++) piece.dsc_in.temperature.coeffs[c] = d->wb_coeffs[c];
    rc = dual_demosaic(&pipe, &piece, (float *)out, (const float *)in0, &roo, &roi, v->filters, NULL, FALSE, d->dual_thrs);
  }
  free(geq);
  free(aux);
  if(rc == 0 && d->color_smoothing) color_smoothing((float *)out, &roo, (int)d->color_smoothing);
  return rc;
}
