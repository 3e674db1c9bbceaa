/* integration_stubs.h -- TEST INFRASTRUCTURE (tests/test_boundary_compile.py).
 *
 * The process_cl() / tiling_callback() / dt_develop_blend_process_cl() bodies INTEGRATION.md section 2 tells a maintainer
 * to write, as code a compiler sees: every `dt_iop_<op>_data_t` / `dt_develop_blend_params_t` below is the REFERENCE'S OWN
 * definition, lifted out of src/iop/<op>.c and src/develop/blend.h at build time (oracle/extract.py -> _gen/ref_types.h,
 * never committed), so a field the reference renames or reorders breaks this build and not a user's picture.
 * The harness (boundary_stubs.c) runs exposure, diffuse or sharpen and the blend through the reference's own
 * pixelpipe_process_on_GPU() / default_process_tiling_cl(); the other stubs are compiled and referenced so that they
 * are type-checked. */
#pragma once
#include <string.h>

#include "ansel_hip.h"
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
                      (void *)stub_nlmeans_process_cl, (void *)stub_bilat_process_cl, (void *)stub_develop_blend_process_cl };
  return t[i % 6];
}
