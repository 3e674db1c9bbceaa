/* oracle/ref_wrap/ref_basic.c -- TEST INFRASTRUCTURE ONLY.
 * The reference's own rawprepare / temperature / highlights(clip) / exposure process()
 * bodies, lifted verbatim at build time by oracle/extract.py (see oracle/Makefile), behind
 * C entry points that take the C-ABI structs of include/ansel_hip.h. */
#include "ref_piece.h"

/* ---- rawprepare: src/iop/rawprepare.c -------------------------------------------- */
typedef struct dt_dng_gain_map_t
{
  uint32_t map_points_h, map_points_v;
  double map_spacing_h, map_spacing_v;
  double map_origin_h, map_origin_v;
  float *map_gain;
} dt_dng_gain_map_t;
#define process ref_rawprepare_process_impl
#include "gen/rawprepare.inc"
#undef process

int ref_rawprepare(const dt_hip_piece_t *v, const dt_hip_rawprepare_data_t *d, const void *in, void *out)
{
  ref_reset_fp_mode();
  dt_iop_rawprepare_data_t data;
  memset(&data, 0, sizeof(data));
  data.x = d->x; data.y = d->y; data.width = d->width; data.height = d->height;
  for(int k = 0; k < 4; k++) { data.sub[k] = d->sub[k]; data.div[k] = d->div[k]; }
  dt_dev_pixelpipe_iop_t piece;
  ref_fill_piece(&piece, v, &data);
  dt_dev_pixelpipe_t pipe = { 0 };
  return ref_rawprepare_process_impl(NULL, &pipe, &piece, in, out);
}

/* ---- temperature: src/iop/temperature.c ------------------------------------------ */
typedef struct dt_iop_temperature_data_t
{
  float coeffs[4];
} dt_iop_temperature_data_t; /* src/iop/temperature.c:118-121 */
#include "gen/imageop_math.inc"
#define process ref_temperature_process_impl
#include "gen/temperature.inc"
#undef process

int ref_temperature(const dt_hip_piece_t *v, const dt_hip_temperature_data_t *d, const void *in, void *out)
{
  ref_reset_fp_mode();
  dt_iop_temperature_data_t data;
  for(int k = 0; k < 4; k++) data.coeffs[k] = d->coeffs[k];
  dt_dev_pixelpipe_iop_t piece;
  ref_fill_piece(&piece, v, &data);
  dt_dev_pixelpipe_t pipe = { 0 };
  return ref_temperature_process_impl(NULL, &pipe, &piece, in, out);
}

/* ---- highlights, clip mode: src/iop/highlights.c + highlights/clip.c -------------- */
typedef struct dt_iop_highlights_data_t
{
  int mode;
  float clip;
} dt_iop_highlights_data_t;
enum { DT_IOP_HIGHLIGHTS_CLIP = 0, DT_IOP_HIGHLIGHTS_LCH = 1, DT_IOP_HIGHLIGHTS_INPAINT = 2,
       DT_IOP_HIGHLIGHTS_LAPLACIAN = 3, DT_IOP_HIGHLIGHTS_HARMONIC = 4 };
#define DT_HL_MIN_CLIPPED_PIXELS 25 /* src/iop/highlights/common.h:218 */
#include "gen/highlights.inc"
#include "gen/highlights_clip.inc"

/* follows process(), src/iop/highlights.c:680-789, clip branch only */
int ref_highlights(const dt_hip_piece_t *v, const dt_hip_highlights_data_t *d, const void *in, void *out)
{
  ref_reset_fp_mode();
  if(d->mode != DT_IOP_HIGHLIGHTS_CLIP) return 1;
  dt_iop_highlights_data_t data = { d->mode, d->clip };
  dt_dev_pixelpipe_iop_t piece;
  ref_fill_piece(&piece, v, &data);
  dt_aligned_pixel_t pmax;
  for(int c = 0; c < 4; c++)
    pmax[c] = (piece.dsc_in.processed_maximum[c] > 0.f) ? piece.dsc_in.processed_maximum[c] : 1.0f;
  const float clip = data.clip * fminf(pmax[0], fminf(pmax[1], pmax[2]));
  dt_aligned_pixel_t count_thresholds;
  _hl_count_thresholds(data.mode, data.clip, pmax, clip, count_thresholds);
  if(_hl_count_clipped(&piece, in, &piece.roi_out, count_thresholds) < DT_HL_MIN_CLIPPED_PIXELS)
  {
    _hl_copy_input(&piece, in, out, &piece.roi_out);
    return 0;
  }
  process_clip(&piece, in, out, &piece.roi_in, &piece.roi_out, clip);
  return 0;
}

/* ---- exposure: src/iop/exposure.c ------------------------------------------------ */
typedef struct dt_iop_exposure_data_t
{
  int deflicker;
  float black;
  float scale;
} dt_iop_exposure_data_t; /* the fields process() reads, src/iop/exposure.c:151-157 */
#define process ref_exposure_process_impl
#include "gen/exposure.inc"
#undef process

int ref_exposure(const dt_hip_piece_t *v, const dt_hip_exposure_data_t *d, const void *in, void *out)
{
  ref_reset_fp_mode();
  dt_iop_exposure_data_t data = { 0, d->black, d->scale };
  dt_dev_pixelpipe_iop_t piece;
  ref_fill_piece(&piece, v, &data);
  dt_dev_pixelpipe_t pipe = { 0 };
  return ref_exposure_process_impl(NULL, &pipe, &piece, in, out);
}

/* ---- export float -> u16 / u8: src/imageio/imageio_core.c:706-737 ---------------- */
#include "gen/imageio_core.inc"
int ref_export_convert_u16(int width, int height, const float *in, uint16_t *out)
{
  ref_reset_fp_mode();
  _export_final_buffer_to_uint16(in, out, width, height);
  return 0;
}
int ref_export_convert_u8(int width, int height, const float *in, uint8_t *out)
{
  ref_reset_fp_mode();
  _clamp_float_to_uint8(in, out, width, height);
  return 0;
}
