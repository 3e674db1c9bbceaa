/* oracle/ref_wrap/ref_blend.c -- TEST INFRASTRUCTURE ONLY.
 * The blend stage for the "RGB (scene)" blend colourspace.  Lifted verbatim at build time:
 *   src/develop/blends/blendif_rgb_jzczhz.c   the parametric mask (make_mask and its channel functions)
 *                                             and the blend operators (blend and its row functions)
 *   src/develop/blend.c                       dt_develop_blendif_process_parameters(),
 *                                             dt_develop_blendif_init_masking_profile(),
 *                                             _develop_blend_process_mask_tone_curve()
 *   src/develop/blend.h                       the parameter struct and its enums
 * The driver below follows dt_develop_blend_process() (blend.c:657-900) for the mask sources this
 * build supports (uniform, parametric); it contains no pixel arithmetic of its own. */
#define REF_REAL_IMAGEBUF 1
#include "ref_piece.h"
#include "common/imagebuf.h"
#include "common/colorspaces_inline_conversions.h"
#include "math/openmp_maths.h"
#include "pixel/gaussian.h"
#include "pixel/guided_filter.h"

typedef int dt_colorspaces_color_profile_type_t;
typedef int dt_colorspaces_color_mode_t;
typedef enum dt_iop_color_intent_t { DT_INTENT_PERCEPTUAL = 0 } dt_iop_color_intent_t;
#define DT_IOP_COLOR_ICC_LEN 512
#include "gen/iop_profile.inc"
#include "gen/iop_profile_info.inc"
#include "gen/iop_profile_xyz.inc"

typedef char dt_dev_operation_t[20];
typedef int dt_dev_pixelpipe_display_mask_t;
#define DT_DEV_PIXELPIPE_DISPLAY_NONE 0
#define DT_DEV_PIXELPIPE_DISPLAY_ANY 0x3fc /* never requested here */
typedef struct dt_iop_module_so_t dt_iop_module_so_t;
#include "gen/blend_h.inc"

/* the profile dt_develop_blendif_init_masking_profile() copies: the work profile of the pipe */
static __thread const dt_iop_order_iccprofile_info_t *ref_blend_work_profile;
#define dt_ioppr_get_pipe_current_profile_info(module, pipe) (ref_blend_work_profile)
#define dt_ioppr_get_iop_work_profile_info(module, iop) (ref_blend_work_profile)
#include "gen/blend_c.inc"

/* GUI channel display is never requested on an export */
static void _display_channel(const float *a, float *b, const float *mask, size_t stride, int channel,
                             const float *boost, const dt_iop_order_iccprofile_info_t *profile)
{
  (void)a; (void)b; (void)mask; (void)stride; (void)channel; (void)boost; (void)profile;
}
/* the row-function type of blendif_rgb_jzczhz.c:37-38 (a function typedef, which extract.py does not lift) */
typedef void(_blend_row_func)(const float *const restrict a, const float *const restrict b, const float p,
                              float *const restrict out, const float *const restrict mask, const size_t stride);
#include "gen/blendif_rgb_jzczhz.inc"

/* src/develop/blends/blendif_lab.c, compiled in ref_blend_lab.c */
void dt_develop_blendif_lab_make_mask(const struct dt_dev_pixelpipe_iop_t *piece, const float *const restrict a,
                                      const float *const restrict b, float *const restrict mask);
void dt_develop_blendif_lab_blend(const struct dt_dev_pixelpipe_t *pipe, const struct dt_dev_pixelpipe_iop_t *piece,
                                  const float *const a, float *const b, const float *const restrict mask,
                                  const dt_dev_pixelpipe_display_mask_t request_mask_display);

/* src/develop/blends/blendif_rgb_hsl.c, compiled in ref_blend_hsl.c */
void dt_develop_blendif_rgb_hsl_make_mask(const struct dt_dev_pixelpipe_t *pipe, const struct dt_dev_pixelpipe_iop_t *piece,
                                          const float *const restrict a, const float *const restrict b, float *const restrict mask);
void dt_develop_blendif_rgb_hsl_blend(const struct dt_dev_pixelpipe_t *pipe, const struct dt_dev_pixelpipe_iop_t *piece,
                                      const float *const restrict a, float *const restrict b, const float *const restrict mask,
                                      const dt_dev_pixelpipe_display_mask_t request_mask_display);

/* src/develop/blends/blendif_raw.c, compiled in ref_blend_raw.c */
void dt_develop_blendif_raw_make_mask(const struct dt_dev_pixelpipe_iop_t *piece, const float *const restrict a,
                                      const float *const restrict b, float *const restrict mask);
void dt_develop_blendif_raw_blend(const struct dt_dev_pixelpipe_t *pipe, const struct dt_dev_pixelpipe_iop_t *piece,
                                  const float *const restrict a, float *const restrict b, const float *const restrict mask,
                                  const dt_dev_pixelpipe_display_mask_t request_mask_display);

/* dt_develop_blend_get_mask_usage(), blend.c:262-320: the parametric part */
static int parametric_used(const dt_develop_blend_params_t *params)
{
  if(!(params->mask_mode & DEVELOP_MASK_PARAMETRIC)) return 0;
  const float threshold_epsilon = 1e-6f;
  const uint32_t channel_mask = params->blend_cst == DEVELOP_BLEND_CS_LAB ? DEVELOP_BLENDIF_Lab_MASK : DEVELOP_BLENDIF_RGB_MASK;
  uint32_t active_channels = 0;
  for(uint32_t ch = 0; ch < DEVELOP_BLENDIF_SIZE; ch++)
  {
    const uint32_t bit = 1u << ch;
    if(!(channel_mask & bit) || !(params->blendif & bit)) continue;
    const float *channel = &params->blendif_parameters[ch * 4];
    if(fabsf(channel[0]) > threshold_epsilon || fabsf(channel[1]) > threshold_epsilon
       || fabsf(channel[2] - 1.0f) > threshold_epsilon || fabsf(channel[3] - 1.0f) > threshold_epsilon)
      active_channels |= bit;
  }
  return active_channels != 0;
}

#include "gen/masks_detail.inc"

/* the hidden stage behind demosaic, src/iop/detailmask.c:111-150 */
int ref_detailmask(const dt_hip_piece_t *v, const dt_hip_detailmask_data_t *h, const void *in, void *out)
{
  ref_reset_fp_mode();
  const int width = v->roi_out.width, height = v->roi_out.height;
  if(!h->mask || width < 3 || height < 3) return 1;
  memcpy(out, in, sizeof(float) * 4 * (size_t)width * height);
  float *tmp = (float *)malloc(sizeof(float) * (size_t)width * height);
  if(!tmp) return 1;
  dt_aligned_pixel_t wb = { h->wb[0], h->wb[1], h->wb[2], 1.0f };
  dt_masks_calc_rawdetail_mask((float *)out, (float *)h->mask, tmp, width, height, wb);
  free(tmp);
  return 0;
}

/* `in` = the module's input (roi_in), `out` = the module's output (roi_out), blended in place */
int ref_develop_blend(const dt_hip_piece_t *v, const dt_hip_blend_data_t *h, const void *in, void *out)
{
  ref_reset_fp_mode();
  if(h->blend_cst < DEVELOP_BLEND_CS_RAW || h->blend_cst > DEVELOP_BLEND_CS_RGB_SCENE) return -1;
  /* drawn / raster masks (and the details refinement of them) come rendered, as one plane: what
   * _develop_blend_init_raster_mask() / _init_drawn_mask() / _refine_with_detail_mask() leave in `mask` (blend.c:740-790) */
  const float *const form = (const float *)h->form_mask;
  if((h->mask_mode & (DEVELOP_MASK_SHAPE | DEVELOP_MASK_RASTER)) && !form) return -1;
  if(form && h->blend_cst == DEVELOP_BLEND_CS_RAW) return -1;
  dt_develop_blend_params_t d;
  memset(&d, 0, sizeof(d));
  d.mask_mode = h->mask_mode;
  d.blend_cst = h->blend_cst;
  d.blend_mode = h->blend_mode;
  d.blend_parameter = h->blend_parameter;
  d.opacity = h->opacity;
  d.mask_combine = h->mask_combine;
  d.blendif = h->blendif;
  d.contrast = h->contrast;
  d.blur_radius = h->blur_radius;
  d.feathering_radius = h->feathering_radius;
  d.feathering_guide = h->feathering_guide;
  d.brightness = h->brightness;
  memcpy(d.blendif_parameters, h->blendif_parameters, sizeof(d.blendif_parameters));
  memcpy(d.blendif_boost_factors, h->blendif_boost_factors, sizeof(d.blendif_boost_factors));

  dt_iop_order_iccprofile_info_t work;
  static float linear[4] = { -1.0f, 0.f, 0.f, 0.f };
  memset(&work, 0, sizeof(work));
  work.nonlinearlut = 0;
  work.lutsize = 0x10000;
  for(int k = 0; k < 3; k++)
  {
    work.lut_in[k] = linear;
    work.lut_out[k] = linear;
    for(int c = 0; c < 3; c++)
    {
      work.matrix_in[k][c] = h->matrix_in[k][c];
      work.matrix_in_transposed[c][k] = h->matrix_in[k][c];
    }
  }
  ref_blend_work_profile = &work;

  dt_iop_module_t module;
  memset(&module, 0, sizeof(module));
  dt_dev_pixelpipe_iop_t piece;
  ref_fill_piece(&piece, v, NULL);
  piece.module = &module;
  piece.blendop_data = &d;
  dt_dev_pixelpipe_t pipe;
  memset(&pipe, 0, sizeof(pipe));
  pipe.type = DT_DEV_PIXELPIPE_EXPORT;

  if(!(d.mask_mode & DEVELOP_MASK_ENABLED)) return 0;
  const int owidth = piece.roi_out.width, oheight = piece.roi_out.height;
  const size_t buffsize = (size_t)owidth * oheight;
  const float opacity = fminf(fmaxf(d.opacity / 100.0f, 0.0f), 1.0f);
  float *mask = dt_pixelpipe_cache_alloc_align_float(buffsize, &pipe);
  if(!mask) return 1;
  /* blend.c:732-760 */
  const int raster_used = form && (d.mask_mode & DEVELOP_MASK_RASTER), drawn_used = form && (d.mask_mode & DEVELOP_MASK_SHAPE);
  const int use_masks = form || parametric_used(&d);
  const int raster_only = raster_used && !drawn_used && !parametric_used(&d);
  if(!use_masks)
    dt_iop_image_fill(mask, opacity, owidth, oheight, 1);
  else if(raster_only)
  {
    memcpy(mask, form, sizeof(float) * buffsize);
    dt_iop_image_mul_const(mask, opacity, owidth, oheight, 1);
  }
  else
  {
    if(form)
      memcpy(mask, form, sizeof(float) * buffsize);
    else
    {
      const float fill = (d.mask_combine & DEVELOP_COMBINE_INCL) ? 0.0f : 1.0f;
      dt_iop_image_fill(mask, fill, owidth, oheight, 1);
    }
    if(h->details != 0.f && h->detail_mask)
    {
      /* _refine_with_detail_mask(), blend.c:361-425, with the raw detail mask in the geometry of roi_out
       * (dt_dev_distort_detail_mask() hands `lum` back unchanged then) */
      const int detail = (h->details > 0.0f);
      const float threshold = _detail_mask_threshold(h->details, detail);
      float *tmp = dt_pixelpipe_cache_alloc_align_float(buffsize, &pipe);
      float *lum = dt_pixelpipe_cache_alloc_align_float(buffsize, &pipe);
      if(!tmp || !lum) return 1;
      dt_masks_calc_detail_mask((float *)h->detail_mask, lum, tmp, owidth, oheight, threshold, detail);
      for(size_t idx = 0; idx < buffsize; idx++) mask[idx] = mask[idx] * lum[idx];
      dt_pixelpipe_cache_free_align(tmp);
      dt_pixelpipe_cache_free_align(lum);
    }
    if(d.blend_cst == DEVELOP_BLEND_CS_LAB)
      dt_develop_blendif_lab_make_mask(&piece, (const float *)in, (const float *)out, mask);
    else if(d.blend_cst == DEVELOP_BLEND_CS_RAW)
      dt_develop_blendif_raw_make_mask(&piece, (const float *)in, (const float *)out, mask);
    else if(d.blend_cst == DEVELOP_BLEND_CS_RGB_DISPLAY)
      dt_develop_blendif_rgb_hsl_make_mask(&pipe, &piece, (const float *)in, (const float *)out, mask);
    else
      dt_develop_blendif_rgb_jzczhz_make_mask(&pipe, &piece, (const float *)in, (const float *)out, mask);
    /* the post operations, blend.c:815-885, in the order the reference's own _develop_mask_get_post_operations() gives */
    _develop_mask_post_processing ops[3];
    const size_t nops = _develop_mask_get_post_operations(&d, &piece, ops);
    const int rois_equal = piece.roi_in.width == owidth && piece.roi_in.height == oheight
                           && piece.roi_in.x == piece.roi_out.x && piece.roi_in.y == piece.roi_out.y;
    for(size_t k = 0; k < nops; k++)
    {
      if(ops[k] == DEVELOP_MASK_POST_FEATHER_IN || ops[k] == DEVELOP_MASK_POST_FEATHER_OUT)
      {
        const float guide_weight = (d.blend_cst == DEVELOP_BLEND_CS_LAB) ? 1.0f : 100.0f; /* dt_iop_colorspace_is_rgb() */
        /* FEATHER_IN with roi_in != roi_out: the reference's region copy reads past its input (blend.c:823): not run */
        if(ops[k] == DEVELOP_MASK_POST_FEATHER_IN && !rois_equal)
        {
          dt_pixelpipe_cache_free_align(mask);
          return -1;
        }
        const float *guide = ops[k] == DEVELOP_MASK_POST_FEATHER_IN ? (const float *)in : (const float *)out;
        if(_develop_blend_process_feather(guide, mask, owidth, oheight, 4, guide_weight, d.feathering_radius, piece.roi_out.scale))
        {
          dt_pixelpipe_cache_free_align(mask);
          return 1;
        }
      }
      else if(ops[k] == DEVELOP_MASK_POST_BLUR)
      {
        /* blend.c:869-881 */
        const float sigma = d.blur_radius * piece.roi_out.scale;
        const float mmax[] = { 1.0f };
        const float mmin[] = { 0.0f };
        dt_gaussian_t *g = dt_gaussian_init(owidth, oheight, 1, mmax, mmin, sigma, 0);
        if(g)
        {
          dt_gaussian_blur(g, mask, mask);
          dt_gaussian_free(g);
        }
      }
      else if(ops[k] == DEVELOP_MASK_POST_TONE_CURVE)
        _develop_blend_process_mask_tone_curve(mask, buffsize, d.contrast, d.brightness, opacity);
    }
  }
  if(d.blend_cst == DEVELOP_BLEND_CS_LAB)
    dt_develop_blendif_lab_blend(&pipe, &piece, (const float *)in, (float *)out, mask, DT_DEV_PIXELPIPE_DISPLAY_NONE);
  else if(d.blend_cst == DEVELOP_BLEND_CS_RAW)
    dt_develop_blendif_raw_blend(&pipe, &piece, (const float *)in, (float *)out, mask, DT_DEV_PIXELPIPE_DISPLAY_NONE);
  else if(d.blend_cst == DEVELOP_BLEND_CS_RGB_DISPLAY)
    dt_develop_blendif_rgb_hsl_blend(&pipe, &piece, (const float *)in, (float *)out, mask, DT_DEV_PIXELPIPE_DISPLAY_NONE);
  else
    dt_develop_blendif_rgb_jzczhz_blend(&pipe, &piece, (const float *)in, (float *)out, mask, DT_DEV_PIXELPIPE_DISPLAY_NONE);
  dt_pixelpipe_cache_free_align(mask);
  return 0;
}
