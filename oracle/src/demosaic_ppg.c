/* oracle/src/demosaic_ppg.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * demosaic_ppg(), src/iop/demosaic/ppg.c:20-217, through the per-pixel form in ppg_core.h; with thrs > 0 the
 * green sites first go through pre_median() (src/iop/demosaic/basic.c:136-186), except for what pass 1 reads.  Alpha: the reference writes 0 for ring >= 3
 * (ppg.c:117) and leaves the caller's buffer untouched for ring < 3; so does this. */
#include <stdlib.h>
#include <string.h>
#include "ppg_core.h"

/* pre_median_b(), basic.c:136-180, one pass: every green site at least 3 px from the border becomes the median of
 * the 9 greens of its diamond that lie within `threshold` of it (the others are pushed out of the way by + 64) */
static void pre_median(float *out, const float *in, const int width, const int height, const uint32_t filters,
                       const float threshold)
{
  memcpy(out, in, sizeof(float) * (size_t)width * height);
  const int lim[5] = { 0, 1, 2, 1, 0 };
  for(int row = 3; row < height - 3; row++)
  {
    float med[9];
    int col = 3;
    if(oracle_fc(row, col, filters) != 1 && oracle_fc(row, col, filters) != 3) col++;
    for(; col < width - 3; col += 2)
    {
      const float *pixi = in + (size_t)width * row + col;
      int cnt = 0;
      for(int k = 0, i = 0; i < 5; i++)
        for(int j = -lim[i]; j <= lim[i]; j += 2)
        {
          if(fabsf(pixi[width * (i - 2) + j] - pixi[0]) < threshold)
          {
            med[k++] = pixi[width * (i - 2) + j];
            cnt++;
          }
          else
            med[k++] = 64.0f + pixi[width * (i - 2) + j];
        }
      for(int i = 0; i < 8; i++)
        for(int ii = i + 1; ii < 9; ii++)
          if(med[i] > med[ii])
          {
            const float t = med[i];
            med[i] = med[ii];
            med[ii] = t;
          }
      out[(size_t)width * row + col] = (cnt == 1 ? med[4] - 64.0f : med[(cnt - 1) / 2]);
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
