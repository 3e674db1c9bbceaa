/* oracle/src/demosaic_ppg.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * demosaic_ppg(), src/iop/demosaic/ppg.c:20-217, median pre-filter off (thrs == 0, the default),
 * through the per-pixel form in ppg_core.h.  Alpha: the reference writes 0 for ring >= 3
 * (ppg.c:117) and leaves the caller's buffer untouched for ring < 3; so does this. */
#include "ppg_core.h"

int oracle_demosaic_ppg(float *out, const float *in, const dt_hip_roi_t *roi_out, const dt_hip_roi_t *roi_in,
                        uint32_t filters)
{
  const ppg_ctx_t k = { in, roi_in->width, roi_in->height, roi_out->width, roi_out->height,
                        roi_out->x, roi_out->y, filters, 0 };
  for(int j = 0; j < k.oh; j++)
    for(int i = 0; i < k.ow; i++)
    {
      float *o = out + 4 * ((size_t)j * k.ow + i);
      ppg_pixel(&k, j, i, o);
      if(!ppg_ring_lt(&k, j, i, 3)) o[3] = 0.0f;
    }
  return 0;
}
