/* oracle/ref_wrap/ref_diffuse.c -- TEST INFRASTRUCTURE ONLY.
 * The reference's diffuse-or-sharpen module (src/iop/diffuse.c: process, wavelets_process,
 * heat_PDE_diffusion and helpers; src/pixel/bspline.h: decompose_2D_Bspline), lifted verbatim at
 * build time, behind a C entry point that takes the C-ABI structs of include/ansel_hip.h. */
#include "ref_piece.h"
#include "pixel/bspline.h"
#include "iop/noise_generator.h"

typedef void *GtkWidget;
#define process ref_diffuse_module_process /* every module calls its entry point process() */
#include "gen/diffuse.inc"

int ref_diffuse(const dt_hip_piece_t *v, const dt_hip_diffuse_data_t *d, const void *in, void *out)
{
  ref_reset_fp_mode();
  dt_iop_diffuse_params_t p;
  memset(&p, 0, sizeof(p));
  p.iterations = d->iterations;
  p.sharpness = d->sharpness;
  p.radius = d->radius;
  p.regularization = d->regularization;
  p.variance_threshold = d->variance_threshold;
  p.anisotropy_first = d->anisotropy_first;
  p.anisotropy_second = d->anisotropy_second;
  p.anisotropy_third = d->anisotropy_third;
  p.anisotropy_fourth = d->anisotropy_fourth;
  p.threshold = d->threshold;
  p.first = d->first;
  p.second = d->second;
  p.third = d->third;
  p.fourth = d->fourth;
  p.radius_center = d->radius_center;
  dt_dev_pixelpipe_iop_t piece;
  ref_fill_piece(&piece, v, &p);
  dt_dev_pixelpipe_t pipe;
  memset(&pipe, 0, sizeof(pipe));
  pipe.iscale = d->iscale;
  return process(NULL, &pipe, &piece, in, out);
}
