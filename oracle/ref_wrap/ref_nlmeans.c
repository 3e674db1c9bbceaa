/* oracle/ref_wrap/ref_nlmeans.c -- TEST INFRASTRUCTURE ONLY.
 * The reference's denoise (non-local means) module: process_cpu() of src/iop/nlmeans.c lifted
 * verbatim at build time, over nlmeans_denoise() of src/pixel/nlmeans_core.c compiled from where it
 * lies (oracle/Makefile REF_DIRECT). */
#define REF_REAL_IMAGEBUF 1
#include "ref_piece.h"
#include "pixel/nlmeans_core.h"

typedef void *GtkWidget;
#define dt_dev_pixelpipe_has_preview_output(dev, pipe, roi) (0)
#include "gen/nlmeans.inc"

int ref_nlmeans(const dt_hip_piece_t *v, const dt_hip_nlmeans_data_t *h, const void *in, void *out)
{
  ref_reset_fp_mode();
  dt_iop_nlmeans_params_t p = { h->radius, h->strength, h->luma, h->chroma };
  dt_dev_pixelpipe_iop_t piece;
  ref_fill_piece(&piece, v, &p);
  dt_dev_pixelpipe_t pipe;
  memset(&pipe, 0, sizeof(pipe));
  pipe.type = DT_DEV_PIXELPIPE_EXPORT;
  process_cpu(&pipe, &piece, in, out, &piece.roi_in, &piece.roi_out, nlmeans_denoise);
  return 0;
}
