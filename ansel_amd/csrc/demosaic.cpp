// demosaic.cpp -- device entry of the demosaic module: the part of process()/process_cl()
// (src/iop/demosaic.c:1041-1253, :1443-1530) that picks the interpolation for a Bayer mosaic.
#include "hip_common.h"

namespace ansel
{
int rcd_demosaic_launch(int devid, const dt_hip_piece_t *piece, uint32_t filters, const float *in, float4 *out,
                        const rcd_band_t *band);
int ppg_demosaic_launch(int devid, const dt_hip_piece_t *piece, uint32_t filters, const float *in, float4 *out);
int amaze_demosaic_launch(int devid, const dt_hip_piece_t *piece, uint32_t filters, const float *in, float4 *out);
}
extern "C" uint32_t dt_hip_crop_dcraw_filters(uint32_t filters, uint32_t crop_x, uint32_t crop_y);

using namespace ansel;

namespace ansel
{
int dt_hip_iop_demosaic_process_band(int devid, const dt_hip_piece_t *piece, const dt_hip_demosaic_data_t *d,
                                     const rcd_band_t *band, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  if(!piece->filters || piece->filters == 9u || piece->channels != 1)
  {
    set_last_error("demosaic: only Bayer mosaics are implemented on device");
    return DT_HIP_INVALID_ARG;
  }
  if(d->green_eq || d->color_smoothing || d->median_thrs != 0.0f)
  {
    set_last_error("demosaic: green equilibration / colour smoothing / median are not implemented on device");
    return DT_HIP_INVALID_ARG;
  }
  if(piece->roi_out.width != piece->roi_in.width || piece->roi_out.height != piece->roi_in.height)
  {
    // full demosaic runs at scale 1 with identical in/out geometry (demosaic.c:1050-1053)
    set_last_error("demosaic: roi_in and roi_out differ");
    return DT_HIP_INVALID_ARG;
  }
  // dt_dev_get_roi_filters(), src/develop/imageop.c:139-142
  const uint32_t filters = dt_hip_crop_dcraw_filters(piece->filters, piece->roi_in.x, piece->roi_in.y);
  switch(d->demosaicing_method)
  {
    case DT_HIP_DEMOSAIC_RCD:
      return rcd_demosaic_launch(devid, piece, filters, (const float *)dev_in, (float4 *)dev_out, band);
    case DT_HIP_DEMOSAIC_PPG:
      if(band)
      {
        set_last_error("demosaic: PPG has no row-band mode");
        return DT_HIP_INVALID_ARG;
      }
      return ppg_demosaic_launch(devid, piece, filters, (const float *)dev_in, (float4 *)dev_out);
    case DT_HIP_DEMOSAIC_AMAZE:
      if(band)
      {
        set_last_error("demosaic: AMaZE has no row-band mode");
        return DT_HIP_INVALID_ARG;
      }
      return amaze_demosaic_launch(devid, piece, filters, (const float *)dev_in, (float4 *)dev_out);
    default:
      set_last_error("demosaic: method %u is not implemented on device", d->demosaicing_method);
      return DT_HIP_INVALID_ARG;
  }
}

} // namespace ansel

extern "C" {

int dt_hip_iop_demosaic_process(int devid, const dt_hip_piece_t *piece, const dt_hip_demosaic_data_t *d,
                                dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  return dt_hip_iop_demosaic_process_band(devid, piece, d, nullptr, dev_in, dev_out);
}

// tiling_callback(), src/iop/demosaic.c:1930-1990: RCD overlap 10, PPG/AMaZE 5; 2x2 alignment
void dt_hip_iop_demosaic_tiling(const dt_hip_piece_t *piece, const dt_hip_demosaic_data_t *d, dt_hip_tiling_t *tiling)
{
  (void)piece;
  tiling->factor = 1.0f + 4.0f; // in (1 ch) + out (4 ch), in units of the input buffer
  tiling->factor_cl = tiling->factor;
  tiling->maxbuf = 1.0f;
  tiling->maxbuf_cl = 1.0f;
  tiling->overhead = 0;
  tiling->xalign = 2;
  tiling->yalign = 2;
  tiling->overlap = (d->demosaicing_method == DT_HIP_DEMOSAIC_RCD) ? 10 : 5; // PPG and AMaZE: 5
}

} // extern "C"
