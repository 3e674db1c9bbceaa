/* oracle/ref_wrap/ref_denoiseprofile.c -- TEST INFRASTRUCTURE ONLY.
 * The reference's profiled denoise, wavelets mode (src/iop/denoiseprofile.c: process_wavelets and
 * the transforms it calls), lifted verbatim at build time; eaw_dn_decompose()/eaw_synthesize() come
 * from src/pixel/eaw.c and dt_iop_alloc_image_buffers() from src/common/imagebuf.c, both compiled
 * from where they lie (oracle/Makefile REF_DIRECT). */
#define REF_REAL_IMAGEBUF 1
#include "ref_piece.h"
#include "pixel/eaw.h"
#include "pixel/nlmeans_core.h"
#include "common/imagebuf.h"

typedef void dt_draw_curve_t;
typedef void *GtkWidget;
#define dt_dev_pixelpipe_has_preview_output(dev, pipe, roi) (0)
#define debug_dump_PFM(p, n, b, w, h, s) /* DEBUG_SCALES is off in the reference build, denoiseprofile.c:416-418 */
#include "gen/denoiseprofile.inc"

int ref_denoiseprofile(const dt_hip_piece_t *v, const dt_hip_denoiseprofile_data_t *h, const void *in, void *out)
{
  ref_reset_fp_mode();
  if(h->mode != DT_HIP_DENOISEPROFILE_WAVELETS && h->mode != DT_HIP_DENOISEPROFILE_NLMEANS) return 1;
  dt_iop_denoiseprofile_data_t d;
  memset(&d, 0, sizeof(d));
  d.radius = h->radius;
  d.nbhood = h->nbhood;
  d.strength = h->strength;
  d.shadows = h->shadows;
  d.bias = h->bias;
  d.scattering = h->scattering;
  d.central_pixel_weight = h->central_pixel_weight;
  d.overshooting = h->overshooting;
  for(int k = 0; k < 3; k++)
  {
    d.a[k] = h->a[k];
    d.b[k] = h->b[k];
  }
  d.mode = h->mode == DT_HIP_DENOISEPROFILE_WAVELETS ? MODE_WAVELETS : MODE_NLMEANS;
  for(int c = 0; c < 6; c++)
    for(int b = 0; b < DT_IOP_DENOISE_PROFILE_BANDS; b++) d.force[c][b] = h->force[c][b];
  d.wb_adaptive_anscombe = h->wb_adaptive_anscombe;
  d.fix_anscombe_and_nlmeans_norm = h->fix_anscombe_and_nlmeans_norm;
  d.use_new_vst = h->use_new_vst;
  d.wavelet_color_mode = h->wavelet_color_mode;
  dt_dev_pixelpipe_iop_t piece;
  ref_fill_piece(&piece, v, &d);
  for(int k = 0; k < 4; k++) piece.dsc_in.temperature.coeffs[k] = h->wb_coeffs[k];
  dt_dev_pixelpipe_t pipe;
  memset(&pipe, 0, sizeof(pipe));
  pipe.type = DT_DEV_PIXELPIPE_EXPORT;
  if(d.mode == MODE_NLMEANS)
    return process_nlmeans_cpu(&pipe, &piece, in, out, &piece.roi_in, &piece.roi_out, nlmeans_denoise);
  return process_wavelets(NULL, &pipe, &piece, in, out, &piece.roi_in, &piece.roi_out, eaw_dn_decompose,
                          eaw_synthesize);
}
