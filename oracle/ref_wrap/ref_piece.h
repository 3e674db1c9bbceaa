/* TEST INFRASTRUCTURE ONLY: fill the shimmed dt_dev_pixelpipe_iop_t from the C-ABI view */
#ifndef REF_PIECE_H
#define REF_PIECE_H
#include "ref_host.h"
#include "ansel_hip.h"

void ref_reset_fp_mode(void);

static inline void ref_fill_piece(dt_dev_pixelpipe_iop_t *piece, const dt_hip_piece_t *v, void *data)
{
  memset(piece, 0, sizeof(*piece));
  piece->data = data;
  piece->roi_in = (dt_iop_roi_t){ v->roi_in.x, v->roi_in.y, v->roi_in.width, v->roi_in.height, v->roi_in.scale };
  piece->roi_out = (dt_iop_roi_t){ v->roi_out.x, v->roi_out.y, v->roi_out.width, v->roi_out.height, v->roi_out.scale };
  piece->buf_in = piece->roi_in;
  piece->buf_out = piece->roi_out;
  piece->dsc_in.filters = v->filters;
  piece->dsc_in.channels = v->channels;
  piece->dsc_in.datatype = (v->datatype == DT_HIP_TYPE_UINT16) ? TYPE_UINT16 : TYPE_FLOAT;
  for(int c = 0; c < 4; c++) piece->dsc_in.processed_maximum[c] = v->processed_maximum[c];
  piece->dsc_out = piece->dsc_in;
}
#endif
