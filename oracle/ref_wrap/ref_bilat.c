/* oracle/ref_wrap/ref_bilat.c -- TEST INFRASTRUCTURE ONLY.
 * The reference's local contrast module, both modes: process() of src/iop/bilat.c lifted verbatim at
 * build time, over src/pixel/bilateral.c and src/pixel/locallaplacian.c compiled from where they lie. */
#include "ref_piece.h"
#include "pixel/bilateral.h"

typedef void *GtkWidget;
#include "pixel/locallaplacian.h"
#define process ref_bilat_module_process /* every module calls its entry point process() */
#include "gen/bilat.inc"

int ref_bilat(const dt_hip_piece_t *v, const dt_hip_bilat_data_t *h, const void *in, void *out)
{
  ref_reset_fp_mode();
  if(h->mode != DT_HIP_BILAT_BILATERAL && h->mode != DT_HIP_BILAT_LOCAL_LAPLACIAN) return 1;
  dt_iop_bilat_params_t p;
  memset(&p, 0, sizeof(p));
  p.mode = h->mode == DT_HIP_BILAT_BILATERAL ? s_mode_bilateral : s_mode_local_laplacian;
  p.sigma_r = h->sigma_r;
  p.sigma_s = h->sigma_s;
  p.detail = h->detail;
  p.midtone = h->midtone;
  dt_dev_pixelpipe_iop_t piece;
  ref_fill_piece(&piece, v, &p);
  dt_dev_pixelpipe_t pipe;
  memset(&pipe, 0, sizeof(pipe));
  pipe.iscale = h->iscale;
  return process(NULL, &pipe, &piece, in, out);
}
