/* integration_stubs.h -- TEST INFRASTRUCTURE (tests/test_boundary_compile.py).
 *
 * The process_cl() / tiling_callback() / dt_develop_blend_process_cl() bodies INTEGRATION.md section 2 tells a maintainer
 * to write, as code a compiler sees: every `dt_iop_<op>_data_t` / `dt_develop_blend_params_t` below is the REFERENCE'S OWN
 * definition, lifted out of src/iop/<op>.c and src/develop/blend.h at build time (oracle/extract.py -> _gen/ref_types.h,
 * never committed), so a field the reference renames or reorders breaks this build and not a user's picture.
 * The harness (boundary_stubs.c) runs exposure, diffuse or sharpen, the blend, the RCD demosaic (a mosaic-stage module:
 * 2 x 2 alignment, one channel in, four out) and denoise (profiled) wavelets (overlap 128) through the reference's own
 * pixelpipe_process_on_GPU() / default_process_tiling_cl(); the other stubs are compiled and referenced so that they
 * are type-checked.  Round 4: the six heavy iops whose piece->data is the hard kind -- demosaic, denoise (profiled),
 * colorin, colorout, color calibration, filmic -- over the reference's own data structs and, for the two colour modules, the
 * reference's own accessor API of the prepared conversion (colorprofiles/conversion.h, included as it is). */
#pragma once
#include <string.h>

#include "ansel_hip.h"
/* GUI-free reference headers, as they are */
#include "system/mem_alloc.h"              /* DT_ALIGNED_PIXEL, DT_ALIGNED_ARRAY */
#include "system/simd.h"                   /* dt_aligned_pixel_t */
#include "math/matrices.h"                 /* dt_colormatrix_t */
#include "pixel/chromatic_adaptation.h"    /* dt_adaptation_t */
#include "pixel/illuminants.h"             /* dt_illuminant_t */
#include "iop/noise_generator.h"           /* dt_noise_distribution_t */
#include "colorprofiles/profile_types.h"   /* dt_colorspaces_color_profile_type_t, DT_IOP_COLOR_ICC_LEN */
/* (colorprofiles/iop_profile.h pulls <CL/cl.h> in under HAVE_OPENCL: dt_iop_order_iccprofile_info_t -- matrix_in / matrix_out of
 * the work profile -- is lifted into _gen/ref_types.h like the modules' own types) */
#include "colorprofiles/conversion.h"      /* dt_colorspaces_conversion_t and its accessors */
typedef struct dt_draw_curve_t dt_draw_curve_t; /* src/common/curve_tools.h: the GUI's spline handles, opaque to process_cl() */
#include "_gen/ref_types.h"

static inline void dt_hip_piece_view(const dt_dev_pixelpipe_iop_t *piece, dt_hip_piece_t *v)
{
  memset(v, 0, sizeof(*v));
  v->roi_in = (dt_hip_roi_t){ piece->roi_in.x, piece->roi_in.y, piece->roi_in.width, piece->roi_in.height, piece->roi_in.scale };
  v->roi_out = (dt_hip_roi_t){ piece->roi_out.x, piece->roi_out.y, piece->roi_out.width, piece->roi_out.height, piece->roi_out.scale };
  v->filters = piece->dsc_in.filters;
  v->channels = piece->dsc_in.channels;
  v->datatype = piece->dsc_in.datatype == TYPE_UINT16 ? DT_HIP_TYPE_UINT16 : DT_HIP_TYPE_FLOAT;
  for(int c = 0; c < 4; c++) v->processed_maximum[c] = piece->dsc_in.processed_maximum[c];
}

/* the dt_develop_tiling_t a module's tiling_callback() fills has the fields of dt_hip_tiling_t in the same order */
static inline void tiling_from_hip(const dt_hip_tiling_t *t, struct dt_develop_tiling_t *tiling)
{
  tiling->factor = t->factor;
  tiling->factor_cl = t->factor_cl;
  tiling->maxbuf = t->maxbuf;
  tiling->maxbuf_cl = t->maxbuf_cl;
  tiling->overhead = t->overhead;
  tiling->overlap = t->overlap;
  tiling->xalign = t->xalign;
  tiling->yalign = t->yalign;
}

/* what the application's pipe knows and this harness' pipe slice does not carry: in a port these are pipe->dev->image_storage.exif_iso,
 * dt_image_is_matrix_correction_supported(&pipe->dev->image_storage) and dt_ioppr_get_pipe_output_profile_info(pipe) */
static float g_stub_exif_iso = 100.0f;
static int g_stub_matrix_correction_supported = 1;
static const dt_iop_order_iccprofile_info_t *g_stub_output_profile = NULL;
static inline float stub_pipe_exif_iso(const struct dt_dev_pixelpipe_t *pipe) { (void)pipe; return g_stub_exif_iso; }
static inline int stub_image_is_matrix_correction_supported(const struct dt_dev_pixelpipe_t *pipe) { (void)pipe; return g_stub_matrix_correction_supported; }
static inline const dt_iop_order_iccprofile_info_t *stub_get_pipe_output_profile_info(const struct dt_dev_pixelpipe_t *pipe) { (void)pipe; return g_stub_output_profile; }

/* ---- exposure (src/iop/exposure.c, replaces :450-498) */
static int stub_exposure_process_cl(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                                    const struct dt_dev_pixelpipe_iop_t *piece, void *dev_in, void *dev_out)
{
  (void)self;
  const dt_iop_exposure_data_t *d = (const dt_iop_exposure_data_t *)piece->data;
  dt_hip_piece_t v;
  dt_hip_piece_view(piece, &v);
  const dt_hip_exposure_data_t hd = { d->black, d->scale };
  return dt_hip_iop_exposure_process(pipe->devid, &v, &hd, dev_in, dev_out) == DT_HIP_SUCCESS;
}

/* ---- diffuse or sharpen (src/iop/diffuse.c:1436-1580; piece->data IS the params struct, :132-138) */
static void stub_diffuse_data(const struct dt_dev_pixelpipe_iop_t *piece, dt_hip_diffuse_data_t *hd)
{
  const dt_iop_diffuse_data_t *d = (const dt_iop_diffuse_data_t *)piece->data;
  *hd = (dt_hip_diffuse_data_t){ d->iterations, d->sharpness, d->radius, d->regularization, d->variance_threshold,
                                 d->anisotropy_first, d->anisotropy_second, d->anisotropy_third, d->anisotropy_fourth,
                                 d->threshold, d->first, d->second, d->third, d->fourth, d->radius_center,
                                 1.0f /* pipe->iscale of a full-resolution export */ };
}
static int stub_diffuse_process_cl(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                                   const struct dt_dev_pixelpipe_iop_t *piece, void *dev_in, void *dev_out)
{
  (void)self;
  dt_hip_piece_t v;
  dt_hip_piece_view(piece, &v);
  dt_hip_diffuse_data_t hd;
  stub_diffuse_data(piece, &hd);
  return dt_hip_iop_diffuse_process(pipe->devid, &v, &hd, dev_in, dev_out) == DT_HIP_SUCCESS;
}
static void stub_diffuse_tiling_callback(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                                         const struct dt_dev_pixelpipe_iop_t *piece, struct dt_develop_tiling_t *tiling)
{
  (void)self;
  (void)pipe;
  dt_hip_piece_t v;
  dt_hip_piece_view(piece, &v);
  dt_hip_diffuse_data_t hd;
  stub_diffuse_data(piece, &hd);
  dt_hip_tiling_t t;
  dt_hip_iop_diffuse_tiling(&v, &hd, &t);
  tiling_from_hip(&t, tiling);
}

/* ---- denoise (non-local means) (src/iop/nlmeans.c:224-397) */
static int stub_nlmeans_process_cl(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                                   const struct dt_dev_pixelpipe_iop_t *piece, void *dev_in, void *dev_out)
{
  (void)self;
  const dt_iop_nlmeans_data_t *d = (const dt_iop_nlmeans_data_t *)piece->data;
  dt_hip_piece_t v;
  dt_hip_piece_view(piece, &v);
  const dt_hip_nlmeans_data_t hd = { d->radius, d->strength, d->luma, d->chroma };
  return dt_hip_iop_nlmeans_process(pipe->devid, &v, &hd, dev_in, dev_out) == DT_HIP_SUCCESS;
}

/* ---- local contrast (src/iop/bilat.c:206-248): both modes of the module */
static int stub_bilat_process_cl(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                                 const struct dt_dev_pixelpipe_iop_t *piece, void *dev_in, void *dev_out)
{
  (void)self;
  const dt_iop_bilat_data_t *d = (const dt_iop_bilat_data_t *)piece->data;
  dt_hip_piece_t v;
  dt_hip_piece_view(piece, &v);
  const dt_hip_bilat_data_t hd = { d->mode == s_mode_bilateral ? 0 : 1, d->sigma_r, d->sigma_s, d->detail, d->midtone, 1.0f };
  return dt_hip_iop_bilat_process(pipe->devid, &v, &hd, dev_in, dev_out) == DT_HIP_SUCCESS;
}


/* ---- demosaic (src/iop/demosaic.c:1256-1530, demosaic/rcd.c:568-850): RCD, PPG, AMaZE, VNG4, the dual methods, passthrough */
static void stub_demosaic_data(const struct dt_dev_pixelpipe_t *pipe, const struct dt_dev_pixelpipe_iop_t *piece, dt_hip_demosaic_data_t *hd)
{
  const dt_iop_demosaic_data_t *d = (const dt_iop_demosaic_data_t *)piece->data;
  _Static_assert(DT_IOP_DEMOSAIC_RCD == DT_HIP_DEMOSAIC_RCD && DT_IOP_DEMOSAIC_AMAZE == DT_HIP_DEMOSAIC_AMAZE
                     && DT_IOP_DEMOSAIC_PPG == DT_HIP_DEMOSAIC_PPG && DT_IOP_DEMOSAIC_VNG4 == DT_HIP_DEMOSAIC_VNG4
                     && (int)DT_IOP_DEMOSAIC_RCD_VNG == (DT_HIP_DEMOSAIC_DUAL | DT_HIP_DEMOSAIC_RCD),
                 "the method codes are the reference's");
  memset(hd, 0, sizeof(*hd));
  hd->green_eq = d->green_eq;
  hd->color_smoothing = d->color_smoothing;
  hd->demosaicing_method = d->demosaicing_method; /* DEMOSAIC_DUAL included */
  hd->median_thrs = d->median_thrs;
  hd->green_eq_threshold = 0.0001f * stub_pipe_exif_iso(pipe); /* the `threshold` of demosaic.c:1049 */
  hd->dual_thrs = d->dual_thrs;
  for(int c = 0; c < 4; c++) hd->wb_coeffs[c] = piece->dsc_in.temperature.coeffs[c]; /* the dual methods' detail mask */
}
static int stub_demosaic_process_cl(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                                    const struct dt_dev_pixelpipe_iop_t *piece, void *dev_in, void *dev_out)
{
  (void)self;
  dt_hip_piece_t v;
  dt_hip_piece_view(piece, &v);
  dt_hip_demosaic_data_t hd;
  stub_demosaic_data(pipe, piece, &hd);
  /* DT_HIP_INVALID_ARG for X-Trans, LMMSE, the half-size downsample -> FALSE -> the host's process() */
  return dt_hip_iop_demosaic_process(pipe->devid, &v, &hd, dev_in, dev_out) == DT_HIP_SUCCESS;
}
static void stub_demosaic_tiling_callback(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                                          const struct dt_dev_pixelpipe_iop_t *piece, struct dt_develop_tiling_t *tiling)
{
  (void)self;
  dt_hip_piece_t v;
  dt_hip_piece_view(piece, &v);
  dt_hip_demosaic_data_t hd;
  stub_demosaic_data(pipe, piece, &hd);
  dt_hip_tiling_t t;
  dt_hip_iop_demosaic_tiling(&v, &hd, &t);
  tiling_from_hip(&t, tiling);
}

/* ---- denoise (profiled) (src/iop/denoiseprofile.c:2138-2700): dt_hip_denoiseprofile_data_t is the reference's struct minus the
 *      curve handles, plus the buffer descriptor's white-balance coefficients */
static void stub_denoiseprofile_data(const struct dt_dev_pixelpipe_iop_t *piece, dt_hip_denoiseprofile_data_t *hd)
{
  const dt_iop_denoiseprofile_data_t *d = (const dt_iop_denoiseprofile_data_t *)piece->data;
  _Static_assert(sizeof(hd->force) == sizeof(d->force), "DT_DENOISE_PROFILE_NONE x DT_IOP_DENOISE_PROFILE_BANDS");
  _Static_assert(MODE_NLMEANS == DT_HIP_DENOISEPROFILE_NLMEANS && MODE_WAVELETS == DT_HIP_DENOISEPROFILE_WAVELETS
                     && MODE_NLMEANS_AUTO == DT_HIP_DENOISEPROFILE_NLMEANS_AUTO && MODE_WAVELETS_AUTO == DT_HIP_DENOISEPROFILE_WAVELETS_AUTO,
                 "the mode codes are the reference's");
  memset(hd, 0, sizeof(*hd));
  hd->radius = d->radius; hd->nbhood = d->nbhood; hd->strength = d->strength; hd->shadows = d->shadows; hd->bias = d->bias;
  hd->scattering = d->scattering; hd->central_pixel_weight = d->central_pixel_weight; hd->overshooting = d->overshooting;
  memcpy(hd->a, d->a, sizeof(hd->a));
  memcpy(hd->b, d->b, sizeof(hd->b));
  hd->mode = d->mode;
  memcpy(hd->force, d->force, sizeof(hd->force));
  hd->wb_adaptive_anscombe = d->wb_adaptive_anscombe;
  hd->fix_anscombe_and_nlmeans_norm = d->fix_anscombe_and_nlmeans_norm;
  hd->use_new_vst = d->use_new_vst;
  hd->wavelet_color_mode = d->wavelet_color_mode;
  for(int c = 0; c < 4; c++) hd->wb_coeffs[c] = piece->dsc_in.temperature.coeffs[c];
}
static int stub_denoiseprofile_process_cl(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                                          const struct dt_dev_pixelpipe_iop_t *piece, void *dev_in, void *dev_out)
{
  (void)self;
  const dt_iop_denoiseprofile_data_t *d = (const dt_iop_denoiseprofile_data_t *)piece->data;
  if(d->mode == MODE_VARIANCE) return 0; /* a GUI helper: the host's process() */
  dt_hip_piece_t v;
  dt_hip_piece_view(piece, &v);
  dt_hip_denoiseprofile_data_t hd;
  stub_denoiseprofile_data(piece, &hd);
  return dt_hip_iop_denoiseprofile_process(pipe->devid, &v, &hd, dev_in, dev_out) == DT_HIP_SUCCESS;
}
static void stub_denoiseprofile_tiling_callback(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                                                const struct dt_dev_pixelpipe_iop_t *piece, struct dt_develop_tiling_t *tiling)
{
  (void)self;
  (void)pipe;
  dt_hip_piece_t v;
  dt_hip_piece_view(piece, &v);
  dt_hip_denoiseprofile_data_t hd;
  stub_denoiseprofile_data(piece, &hd);
  dt_hip_tiling_t t;
  dt_hip_iop_denoiseprofile_tiling(&v, &hd, &t);
  tiling_from_hip(&t, tiling);
}

/* ---- colorin / colorout (src/iop/colorin.c:597-682, colorout.c:292-369): the prepared conversion through its accessors
 *      (conversion.c:756-834).  The curves are uploaded per call here; a port caches the three buffers with the piece
 *      (commit_params() builds the conversion once, the curves do not change until the next commit) */
static int stub_conversion(const int devid, const dt_colorspaces_conversion_t *conv, const int source_curves, dt_hip_conversion_t *hd,
                           void *owned[3])
{
  memset(hd, 0, sizeof(*hd));
  owned[0] = owned[1] = owned[2] = NULL;
  dt_colormatrix_t m;
  if(!dt_colorspaces_conversion_is_matrix(conv) || !dt_colorspaces_conversion_matrix(conv, m)) return 0; /* lcms2: the host path */
  for(int r = 0; r < 3; r++)
    for(int c = 0; c < 4; c++) hd->matrix[r][c] = m[r][c];
  hd->has_clipping = dt_colorspaces_conversion_has_clipping(conv);
  if(hd->has_clipping)
  {
    dt_colorspaces_conversion_clip_matrix(conv, m);
    for(int r = 0; r < 3; r++)
      for(int c = 0; c < 4; c++) hd->clip_matrix[r][c] = m[r][c];
  }
  const float *const coeffs = source_curves ? dt_colorspaces_conversion_source_coeffs(conv) : dt_colorspaces_conversion_target_coeffs(conv);
  for(int ch = 0; ch < 3; ch++)
  {
    const float *const curve = source_curves ? dt_colorspaces_conversion_source_curve(conv, ch) : dt_colorspaces_conversion_target_curve(conv, ch);
    const int nonlinear = curve && curve[0] >= 0.0f; /* lut[0] < 0: the channel is linear (conversion.c:552-555) */
    if(nonlinear) owned[ch] = dt_hip_copy_host_to_device_constant(devid, sizeof(float) * DT_HIP_LUT_SAMPLES, (void *)curve);
    if(nonlinear && !owned[ch]) return 0;
    if(source_curves)
    {
      hd->lut_source[ch] = owned[ch];
      hd->lut_source_first[ch] = curve ? curve[0] : -1.0f;
      hd->nonlinear_source |= nonlinear;
      for(int k = 0; k < 3; k++) hd->coeffs_source[ch][k] = coeffs ? coeffs[3 * ch + k] : 0.0f;
    }
    else
    {
      hd->lut_target[ch] = owned[ch];
      hd->lut_target_first[ch] = curve ? curve[0] : -1.0f;
      hd->nonlinear_target |= nonlinear;
      for(int k = 0; k < 3; k++) hd->coeffs_target[ch][k] = coeffs ? coeffs[3 * ch + k] : 0.0f;
    }
  }
  return 1;
}
static int stub_colorin_process_cl(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                                   const struct dt_dev_pixelpipe_iop_t *piece, void *dev_in, void *dev_out)
{
  (void)self;
  const dt_iop_colorin_data_t *d = (const dt_iop_colorin_data_t *)piece->data;
  dt_hip_piece_t v;
  dt_hip_piece_view(piece, &v);
  dt_hip_conversion_t hd;
  void *owned[3];
  int ok = stub_conversion(pipe->devid, d->conversion, 1, &hd, owned);
  /* colorin.c:618: the legacy blue mapping only for images whose matrix correction is supported */
  hd.blue_mapping = d->blue_mapping && stub_image_is_matrix_correction_supported(pipe);
  if(ok) ok = dt_hip_iop_colorin_process(pipe->devid, &v, &hd, dev_in, dev_out) == DT_HIP_SUCCESS;
  for(int ch = 0; ch < 3; ch++) dt_hip_release_mem_object(owned[ch]); /* stream-ordered: after the launch that reads them */
  return ok;
}
static int stub_colorout_process_cl(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                                    const struct dt_dev_pixelpipe_iop_t *piece, void *dev_in, void *dev_out)
{
  (void)self;
  const dt_iop_colorout_data_t *d = (const dt_iop_colorout_data_t *)piece->data;
  if(!d->conversion) return 0; /* a Lab output: the module is a nop on the host */
  dt_hip_piece_t v;
  dt_hip_piece_view(piece, &v);
  dt_hip_conversion_t hd;
  void *owned[3];
  int ok = stub_conversion(pipe->devid, d->conversion, 0, &hd, owned);
  if(ok) ok = dt_hip_iop_colorout_process(pipe->devid, &v, &hd, dev_in, dev_out) == DT_HIP_SUCCESS;
  for(int ch = 0; ch < 3; ch++) dt_hip_release_mem_object(owned[ch]);
  return ok;
}

/* ---- color calibration (src/iop/channelmixerrgb.c:2080-): loop_switch()'s own arguments */
static int stub_channelmixerrgb_process_cl(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                                           const struct dt_dev_pixelpipe_iop_t *piece, void *dev_in, void *dev_out)
{
  (void)self;
  const dt_iop_channelmixer_rbg_data_t *d = (const dt_iop_channelmixer_rbg_data_t *)piece->data;
  const dt_iop_order_iccprofile_info_t *const work = dt_ioppr_get_pipe_work_profile_info(pipe);
  if(!work) return 0; /* no work profile: the module does nothing but copy, on the host (channelmixerrgb.c:1967) */
  _Static_assert(DT_ADAPTATION_LINEAR_BRADFORD == DT_HIP_ADAPTATION_LINEAR_BRADFORD && DT_ADAPTATION_CAT16 == DT_HIP_ADAPTATION_CAT16
                     && DT_ADAPTATION_FULL_BRADFORD == DT_HIP_ADAPTATION_FULL_BRADFORD && DT_ADAPTATION_XYZ == DT_HIP_ADAPTATION_XYZ
                     && DT_ADAPTATION_RGB == DT_HIP_ADAPTATION_RGB, "the adaptation codes are the reference's");
  dt_hip_piece_t v;
  dt_hip_piece_view(piece, &v);
  dt_hip_channelmixerrgb_data_t hd;
  memset(&hd, 0, sizeof(hd));
  for(int r = 0; r < 3; r++)
    for(int c = 0; c < 4; c++)
    {
      hd.XYZ_to_RGB[r][c] = work->matrix_out[r][c];
      hd.RGB_to_XYZ[r][c] = work->matrix_in[r][c];
      hd.MIX[r][c] = d->MIX[r][c];
    }
  for(int c = 0; c < 4; c++)
  {
    hd.illuminant[c] = d->illuminant[c];
    hd.saturation[c] = d->saturation[c];
    hd.lightness[c] = d->lightness[c];
    hd.grey[c] = d->grey[c];
  }
  hd.p = d->p;
  hd.gamut = d->gamut;
  hd.clip = d->clip;
  hd.apply_grey = d->apply_grey;
  hd.adaptation = d->adaptation;
  hd.version = d->version;
  return dt_hip_iop_channelmixerrgb_process(pipe->devid, &v, &hd, dev_in, dev_out) == DT_HIP_SUCCESS;
}

/* ---- filmic (src/iop/filmicrgb.c:3119-): every colour science; the deprecated highlight reconstruction keeps its CPU path */
static int stub_filmicrgb_process_cl(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                                     const struct dt_dev_pixelpipe_iop_t *piece, void *dev_in, void *dev_out)
{
  (void)self;
  const dt_iop_filmicrgb_data_t *d = (const dt_iop_filmicrgb_data_t *)piece->data;
  if(!d->hl_deprecated) return 0; /* highlight reconstruction in use: the host's process() (filmicrgb.c:1430, :2733-2830) */
  const dt_iop_order_iccprofile_info_t *const work = dt_ioppr_get_pipe_work_profile_info(pipe);
  const dt_iop_order_iccprofile_info_t *const export_profile = stub_get_pipe_output_profile_info(pipe);
  if(!work) return 0;
  dt_hip_piece_t v;
  dt_hip_piece_view(piece, &v);
  dt_hip_filmicrgb_data_t hd;
  memset(&hd, 0, sizeof(hd));
  hd.white_source = d->white_source; hd.grey_source = d->grey_source; hd.black_source = d->black_source;
  hd.dynamic_range = d->dynamic_range;
  hd.saturation = d->saturation;
  hd.output_power = d->output_power;
  hd.agx_beta_hue = d->agx_beta_hue;
  hd.preserve_color = d->preserve_color;
  hd.version = d->version;
  hd.use_output_profile = export_profile != NULL;
  for(int c = 0; c < 4; c++)
  {
    hd.spline.M1[c] = d->spline.M1[c]; hd.spline.M2[c] = d->spline.M2[c]; hd.spline.M3[c] = d->spline.M3[c];
    hd.spline.M4[c] = d->spline.M4[c]; hd.spline.M5[c] = d->spline.M5[c];
  }
  hd.spline.latitude_min = d->spline.latitude_min;
  hd.spline.latitude_max = d->spline.latitude_max;
  for(int k = 0; k < 5; k++) { hd.spline.y[k] = d->spline.y[k]; hd.spline.x[k] = d->spline.x[k]; }
  hd.spline.type[0] = d->spline.type[0];
  hd.spline.type[1] = d->spline.type[1];
  for(int r = 0; r < 3; r++)
    for(int c = 0; c < 4; c++)
    {
      hd.work_matrix_in[r][c] = work->matrix_in[r][c];
      hd.work_matrix_out[r][c] = work->matrix_out[r][c];
      if(export_profile)
      {
        hd.export_matrix_in[r][c] = export_profile->matrix_in[r][c];
        hd.export_matrix_out[r][c] = export_profile->matrix_out[r][c];
      }
    }
  return dt_hip_iop_filmicrgb_process(pipe->devid, &v, &hd, dev_in, dev_out) == DT_HIP_SUCCESS;
}

/* ---- the blend stage (src/develop/pixelpipe_gpu.c:360-367 -> dt_develop_blend_process_cl(), blend.c:1180-): uniform and
 *      parametric masks of a linear-matrix work profile; drawn / raster masks and the details threshold need the host's
 *      mask renderer and the pipe's detail mask, which this harness does not have -> 1, the host path */
static float g_stub_work_matrix_in[3][4]; /* work profile's matrix_in: dt_ioppr_get_pipe_current_profile_info() in the application */
static int stub_develop_blend_process_cl(struct dt_iop_module_t *self, struct dt_dev_pixelpipe_t *pipe,
                                         const struct dt_dev_pixelpipe_iop_t *piece, void *dev_in, void *dev_out)
{
  (void)self;
  const dt_develop_blend_params_t *d = (const dt_develop_blend_params_t *)piece->blendop_data;
  /* 0 = done, 1 = error (blend.c:1118-1128, :1189; pixelpipe_gpu.c:431 `if(dt_develop_blend_process_cl(...)) goto error;`) */
  if(!d || !(d->mask_mode & DEVELOP_MASK_ENABLED)) return 0;
  if(d->mask_mode & (DEVELOP_MASK_SHAPE | DEVELOP_MASK_RASTER)) return 1;
  dt_hip_piece_t v;
  dt_hip_piece_view(piece, &v);
  dt_hip_blend_data_t hd = { d->mask_mode, d->blend_cst, d->blend_mode, d->blend_parameter, d->opacity, d->mask_combine,
                             d->blendif, d->feathering_radius, d->blur_radius, d->details, d->contrast, d->brightness };
  _Static_assert(sizeof(hd.blendif_parameters) == sizeof(d->blendif_parameters), "DEVELOP_BLENDIF_SIZE");
  _Static_assert(sizeof(hd.blendif_boost_factors) == sizeof(d->blendif_boost_factors), "DEVELOP_BLENDIF_SIZE");
  memcpy(hd.blendif_parameters, d->blendif_parameters, sizeof(hd.blendif_parameters));
  memcpy(hd.blendif_boost_factors, d->blendif_boost_factors, sizeof(hd.blendif_boost_factors));
  memcpy(hd.matrix_in, g_stub_work_matrix_in, sizeof(hd.matrix_in));
  hd.feathering_guide = d->feathering_guide;
  hd.form_mask = NULL;
  hd.detail_mask = NULL;
  return dt_hip_develop_blend_process(pipe->devid, &v, &hd, dev_in, dev_out) == DT_HIP_SUCCESS ? 0 : 1;
}

/* referenced so that every stub is compiled even where the harness does not run it */
static inline void *integration_stub_table(int i)
{
  void *const t[] = { (void *)stub_exposure_process_cl, (void *)stub_diffuse_process_cl, (void *)stub_diffuse_tiling_callback,
                      (void *)stub_nlmeans_process_cl, (void *)stub_bilat_process_cl, (void *)stub_develop_blend_process_cl,
                      (void *)stub_demosaic_process_cl, (void *)stub_demosaic_tiling_callback, (void *)stub_denoiseprofile_process_cl,
                      (void *)stub_denoiseprofile_tiling_callback, (void *)stub_colorin_process_cl, (void *)stub_colorout_process_cl,
                      (void *)stub_channelmixerrgb_process_cl, (void *)stub_filmicrgb_process_cl };
  return t[i % 14];
}
