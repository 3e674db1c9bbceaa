// demosaic_ppg.hip -- PPG demosaic of a Bayer mosaic: demosaic_ppg(), src/iop/demosaic/ppg.c:20-217
// (the median pre-filter is demosaic_extras.hip's pre_median; pass 1 then reads the unfiltered mosaic).  One launch, one thread per finished float4 pixel: see ppg_device.h.
// Algorithmic bytes: 4 read + 16 written per pixel.
#include "hip_common.h"
#include "ppg_device.h"

using namespace ansel;

namespace
{
__global__ __launch_bounds__(256) void ppg_full(float4 *__restrict__ out, const ppg_ctx k)
{
  // 64 x 4 pixel patch per workgroup: a wave covers 64 consecutive pixels of one row, the four
  // waves four consecutive rows, so the 7x7 footprints overlap in L1
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if(i >= k.w || j >= k.h) return;
  const float4 v = ppg_pixel<false>(k, j, i);
  float4 *const o = out + (size_t)j * k.w + i;
  if(ring_lt(k, j, i, 3))
  {
    // the outer 3 px come from the first pass (ppg.c:30-57), which stores three channels: whatever the
    // caller's buffer holds in the fourth stays there
    o->x = v.x;
    o->y = v.y;
    o->z = v.z;
  }
  else
    *o = v;
}
} // namespace

namespace ansel
{
int ppg_demosaic_launch(int devid, const dt_hip_piece_t *piece, uint32_t filters, const float *in, const float *in_pass1,
                        float4 *out)
{
  ppg_ctx k;
  k.in = in;
  k.in1 = in_pass1;
  k.iw = piece->roi_in.width;
  k.ih = piece->roi_in.height;
  k.w = piece->roi_out.width;
  k.h = piece->roi_out.height;
  k.ox = 0; // process() zeroes roo.x/y before calling demosaic_ppg (demosaic.c:1050-1052)
  k.oy = 0;
  k.filters = filters;
  if(k.w <= 0 || k.h <= 0) return DT_HIP_SUCCESS;
  dim3 grid((k.w + 63) / 64, (k.h + 3) / 4);
  launch_scope ls(devid, "ppg_full");
  ppg_full<<<grid, 256, 0, stream_of(devid)>>>(out, k);
  return check_launch("ppg_full");
}
} // namespace ansel
