// demosaic.cpp -- device entry of the demosaic module: the part of process()/process_cl()
// (src/iop/demosaic.c:1041-1253, :1443-1530) that picks the interpolation for a Bayer mosaic.
#include "hip_common.h"
#include <math.h>

namespace ansel
{
int rcd_demosaic_launch(int devid, const dt_hip_piece_t *piece, uint32_t filters, const float *in, float4 *out,
                        const rcd_band_t *band);
int ppg_demosaic_launch(int devid, const dt_hip_piece_t *piece, uint32_t filters, const float *in, const float *in_pass1,
                        float4 *out);
int green_eq_lavg_launch(int devid, const float *in, float *out, int width, int height, uint32_t filters, int x, int y, float thr);
int green_eq_favg_launch(int devid, const float *in, float *out, int width, int height, uint32_t filters, int x, int y);
int pre_median_launch(int devid, const float *in, float *out, int width, int height, uint32_t filters, float threshold);
int color_smoothing_launch(int devid, float4 *img, int width, int height, int passes);
int passthrough_launch(int devid, const float *in, float4 *out, int width, int height, uint32_t filters, bool color);
int amaze_demosaic_launch(int devid, const dt_hip_piece_t *piece, uint32_t filters, const float *in, float4 *out, const rcd_band_t *band);
int vng4_demosaic_launch(int devid, const dt_hip_piece_t *piece, const float *in, float4 *out);
int dual_demosaic_launch(int devid, const dt_hip_piece_t *piece, const float *raw, float4 *rgb, float dual_threshold, const float wb[4]);
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
  if(d->green_eq > 3)
  {
    set_last_error("demosaic: green_eq %u is not a dt_iop_demosaic_greeneq_t", d->green_eq);
    return DT_HIP_INVALID_ARG;
  }
  const bool dual = (d->demosaicing_method & DT_HIP_DEMOSAIC_DUAL) != 0;
  const uint32_t method = d->demosaicing_method & ~(uint32_t)DT_HIP_DEMOSAIC_DUAL;
  if(dual && (method != DT_HIP_DEMOSAIC_RCD && method != DT_HIP_DEMOSAIC_AMAZE))
  {
    set_last_error("demosaic: the dual methods are RCD + VNG4 and AMaZE + VNG4 (method %u)", d->demosaicing_method);
    return DT_HIP_INVALID_ARG;
  }
  if(dual && !(d->dual_thrs == d->dual_thrs))
  {
    set_last_error("demosaic: the dual threshold is not a number");
    return DT_HIP_INVALID_ARG;
  }
  // the detail mask of the dual methods divides by the white-balance coefficients (dual.c:84): zeros -- what a caller that
  // never filled the field passes -- would make every luminance inf, the mask 1 everywhere and the blend silently the plain
  // RCD / AMaZE result.  The reference's coefficients are the image's, finite and positive
  if(dual && d->dual_thrs > 0.0f)
    for(int c = 0; c < 3; c++)
      if(!(d->wb_coeffs[c] > 0.0f) || !(d->wb_coeffs[c] <= 3.402823466e38f))
      {
        set_last_error("demosaic: the dual methods need the white-balance coefficients of the buffer descriptor (wb_coeffs[%d] = %g)",
                       c, (double)d->wb_coeffs[c]);
        return DT_HIP_INVALID_ARG;
      }
  const bool pass = method == DT_HIP_DEMOSAIC_PASSTHROUGH_MONOCHROME || method == DT_HIP_DEMOSAIC_PASSTHROUGH_COLOR;
  if(band && (dual || pass || method == DT_HIP_DEMOSAIC_VNG4))
  {
    set_last_error("demosaic: VNG4, the dual methods and the passthrough modes have no row-band mode (the detail mask's blur reads "
                   "across bands; the passthrough modes are not export states)");
    return DT_HIP_INVALID_ARG;
  }
  if(d->color_smoothing > 5 || (d->median_thrs != 0.0f && d->demosaicing_method != DT_HIP_DEMOSAIC_PPG))
  {
    set_last_error("demosaic: %u smoothing passes / a median threshold with method %u are not states of the module",
                   d->color_smoothing, d->demosaicing_method);
    return DT_HIP_INVALID_ARG;
  }
  if(piece->roi_out.width != piece->roi_in.width || piece->roi_out.height != piece->roi_in.height)
  {
    // full demosaic runs at scale 1 with identical in/out geometry (demosaic.c:1050-1053)
    set_last_error("demosaic: roi_in and roi_out differ");
    return DT_HIP_INVALID_ARG;
  }
  if(band && (d->green_eq || d->color_smoothing))
  {
    set_last_error("demosaic: green equilibration / colour smoothing have no row-band mode");
    return DT_HIP_INVALID_ARG;
  }
  const int w = piece->roi_in.width, h = piece->roi_in.height;
  if(w <= 0 || h <= 0) return DT_HIP_SUCCESS;
  // dt_dev_get_roi_filters(), src/develop/imageop.c:139-142
  const uint32_t filters = dt_hip_crop_dcraw_filters(piece->filters, piece->roi_in.x, piece->roi_in.y);
  const float *in = (const float *)dev_in;
  float *geq = nullptr, *med = nullptr;
  int err = DT_HIP_SUCCESS;
  float *aux = nullptr;
  if(d->green_eq && !pass) // (the passthrough modes take the mosaic as it came in, demosaic.c:1111-1118)
  {
    // demosaic.c:1137-1163: _FULL = favg, _LOCAL = lavg, _BOTH = favg then lavg
    geq = (float *)dt_hip_alloc_device_buffer(devid, (size_t)w * h * sizeof(float));
    if(d->green_eq == 3) aux = (float *)dt_hip_alloc_device_buffer(devid, (size_t)w * h * sizeof(float));
    if(!geq || (d->green_eq == 3 && !aux)) err = DT_HIP_SYSMEM_ALLOCATION;
    if(err == DT_HIP_SUCCESS && d->green_eq >= 2)
      err = green_eq_favg_launch(devid, in, aux ? aux : geq, w, h, piece->filters, piece->roi_in.x, piece->roi_in.y);
    if(err == DT_HIP_SUCCESS && (d->green_eq & 1))
      err = green_eq_lavg_launch(devid, aux ? aux : in, geq, w, h, piece->filters, piece->roi_in.x, piece->roi_in.y,
                                 d->green_eq_threshold);
    in = geq;
  }
  if(err == DT_HIP_SUCCESS)
    switch(method)
    {
      case DT_HIP_DEMOSAIC_PASSTHROUGH_MONOCHROME:
      case DT_HIP_DEMOSAIC_PASSTHROUGH_COLOR:
        err = passthrough_launch(devid, (const float *)dev_in, (float4 *)dev_out, w, h, piece->filters,
                                 method == DT_HIP_DEMOSAIC_PASSTHROUGH_COLOR);
        break;
      case DT_HIP_DEMOSAIC_VNG4:
        err = vng4_demosaic_launch(devid, piece, in, (float4 *)dev_out);
        break;
      case DT_HIP_DEMOSAIC_RCD:
        err = rcd_demosaic_launch(devid, piece, filters, in, (float4 *)dev_out, band);
        break;
      case DT_HIP_DEMOSAIC_PPG:
        if(band)
        {
          set_last_error("demosaic: PPG has no row-band mode");
          err = DT_HIP_INVALID_ARG;
          break;
        }
        if(d->median_thrs > 0.0f)
        {
          med = (float *)dt_hip_alloc_device_buffer(devid, (size_t)w * h * sizeof(float));
          err = med ? pre_median_launch(devid, in, med, w, h, filters, d->median_thrs) : DT_HIP_SYSMEM_ALLOCATION;
        }
        if(err == DT_HIP_SUCCESS) err = ppg_demosaic_launch(devid, piece, filters, med ? med : in, in, (float4 *)dev_out);
        break;
      case DT_HIP_DEMOSAIC_AMAZE:
        err = amaze_demosaic_launch(devid, piece, filters, in, (float4 *)dev_out, band);
        break;
      default:
        set_last_error("demosaic: method %u is not implemented on device", d->demosaicing_method);
        err = DT_HIP_INVALID_ARG;
    }
  // demosaic.c:1243-1247: the blend with VNG4 of the mosaic as the module received it (not the green-equilibrated copy)
  if(err == DT_HIP_SUCCESS && dual)
    err = dual_demosaic_launch(devid, piece, (const float *)dev_in, (float4 *)dev_out, d->dual_thrs, d->wb_coeffs);
  if(err == DT_HIP_SUCCESS && d->color_smoothing)
    err = color_smoothing_launch(devid, (float4 *)dev_out, piece->roi_out.width, piece->roi_out.height, (int)d->color_smoothing);
  if(geq) dt_hip_release_mem_object(geq);
  if(aux) dt_hip_release_mem_object(aux);
  if(med) dt_hip_release_mem_object(med);
  return err;
}

} // namespace ansel

extern "C" {

int dt_hip_iop_demosaic_process(int devid, const dt_hip_piece_t *piece, const dt_hip_demosaic_data_t *d,
                                dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  return dt_hip_iop_demosaic_process_band(devid, piece, d, nullptr, dev_in, dev_out);
}

// tiling_callback(), src/iop/demosaic.c:1916-1990, Bayer methods: factor in units of the larger (output) buffer as the
// reference counts them -- in + out + max(tmp + green-eq copy, smoothing copy); RCD overlap 10, PPG / AMaZE 5; 2x2
// alignment.  factor_cl: what this implementation holds -- the mosaic and its green-equalised / median-filtered
// copies are a quarter of an output buffer each, colour smoothing runs in place
void dt_hip_iop_demosaic_tiling(const dt_hip_piece_t *piece, const dt_hip_demosaic_data_t *d, dt_hip_tiling_t *tiling)
{
  const float ioratio = (float)piece->roi_out.width * piece->roi_out.height / ((float)piece->roi_in.width * piece->roi_in.height);
  const float smooth = d->color_smoothing ? ioratio : 0.0f;
  const float greeneq = (piece->filters != 9u && d->green_eq) ? 0.25f : 0.0f;
  tiling->factor = 1.0f + ioratio + fmaxf(1.0f + greeneq, smooth);
  tiling->factor_cl = ioratio + 0.25f + greeneq + (d->median_thrs > 0.0f ? 0.25f : 0.0f);
  tiling->maxbuf = 1.0f;
  tiling->maxbuf_cl = 1.0f;
  tiling->overhead = 0;
  tiling->xalign = 2;
  tiling->yalign = 2;
  const uint32_t method = d->demosaicing_method & ~(uint32_t)DT_HIP_DEMOSAIC_DUAL;
  tiling->overlap = method == DT_HIP_DEMOSAIC_RCD ? 10 : 5; // PPG and AMaZE: 5
  if(method == DT_HIP_DEMOSAIC_VNG4)
  {
    // demosaic.c:1995-2002: VNG4
    tiling->xalign = tiling->yalign = 6;
    tiling->overlap = 6;
    tiling->factor_cl += 1.0f; // the linear interpolation
  }
  if(d->demosaicing_method & DT_HIP_DEMOSAIC_DUAL)
  {
    // demosaic.c:2004-2011: "make sure VNG4 is also possible"
    tiling->factor += 1.0f;
    tiling->xalign = tiling->xalign > 6 ? tiling->xalign : 6;
    tiling->yalign = tiling->yalign > 6 ? tiling->yalign : 6;
    tiling->overlap = tiling->overlap > 6 ? tiling->overlap : 6;
    tiling->factor_cl += 2.0f + 0.75f; // VNG4's linear interpolation and result; luminance, raw mask, blend mask
  }
}

} // extern "C"
