/* oracle/src/demosaic_ppg.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * demosaic_ppg(), src/iop/demosaic/ppg.c:20-217, through the per-pixel form in ppg_core.h; with thrs > 0 the
 * green sites first go through pre_median() (src/iop/demosaic/basic.c:136-186), except for what pass 1 reads.  Alpha: the reference writes 0 for ring >= 3
 * (ppg.c:117) and leaves the caller's buffer untouched for ring < 3; so does this. */
#include <stdlib.h>
#include <string.h>
#include "ppg_core.h"

/* pre_median_b(), basic.c:136-180, one pass: every green site at least 3 px from the border becomes the median of
 * the 9 greens of its diamond (rows -2..2 with column offsets {0}, {-1, 1}, {-2, 0, 2}, {-1, 1}, {0}) that lie within
 * `threshold` of it; the others are pushed out of the way by + 64.  The sort is the reference's exchange sort. */
static void pre_median(float *out, const float *in, const int width, const int height, const uint32_t filters,
                       const float threshold)
{
  static const int dy[9] = { -2, -1, -1, 0, 0, 0, 1, 1, 2 }, dx[9] = { 0, -1, 1, -2, 0, 2, -1, 1, 0 };
  memcpy(out, in, sizeof(float) * (size_t)width * height);
#pragma omp parallel for
  for(int row = 3; row < height - 3; row++)
  {
    const int f3 = oracle_fc(row, 3, filters);
    for(int col = (f3 != 1 && f3 != 3) ? 4 : 3; col < width - 3; col += 2)
    {
      const float *c = in + (size_t)width * row + col;
      float v[9];
      int close = 0;
      for(int k = 0; k < 9; k++)
      {
        const float s = c[width * dy[k] + dx[k]];
        const int near_ = fabsf(s - c[0]) < threshold;
        v[k] = near_ ? s : 64.0f + s;
        close += near_;
      }
      for(int a = 0; a < 8; a++)
        for(int b = a + 1; b < 9; b++)
          if(v[a] > v[b])
          {
            const float t = v[a];
            v[a] = v[b];
            v[b] = t;
          }
      out[(size_t)width * row + col] = (close == 1) ? v[4] - 64.0f : v[(close - 1) / 2];
    }
  }
}

int oracle_demosaic_ppg(float *out, const float *in, const dt_hip_roi_t *roi_out, const dt_hip_roi_t *roi_in,
                        uint32_t filters, float median_thrs)
{
  float *med = NULL;
  if(median_thrs > 0.0f)
  {
    med = (float *)malloc(sizeof(float) * (size_t)roi_in->width * roi_in->height);
    if(!med) return 1;
    pre_median(med, in, roi_in->width, roi_in->height, filters, median_thrs);
  }
  const ppg_ctx_t k = { med ? med : in, roi_in->width, roi_in->height, roi_out->width, roi_out->height,
                        roi_out->x, roi_out->y, filters, 0, in };
#pragma omp parallel for
  for(int j = 0; j < k.oh; j++)
    for(int i = 0; i < k.ow; i++)
    {
      float *o = out + 4 * ((size_t)j * k.ow + i);
      ppg_pixel(&k, j, i, o);
      if(!ppg_ring_lt(&k, j, i, 3)) o[3] = 0.0f;
    }
  free(med);
  return 0;
}
