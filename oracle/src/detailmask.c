/* oracle/src/detailmask.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of the detail masks (src/develop/masks/detail.c):
 *   dt_masks_calc_rawdetail_mask() :283-316   the raw detail mask the hidden "detailmask" stage behind demosaic leaves
 *                                              (src/iop/detailmask.c:111-150)
 *   dt_masks_calc_detail_mask()    :325-335   sigmoid around the threshold + dt_masks_blur_9x9() :224-243, what
 *                                              _refine_with_detail_mask() (src/develop/blend.c:361-425) multiplies a
 *                                              blend's form mask with
 *   dt_masks_extend_border() :96-123, dt_masks_blur_9x9_coeff() :159-196, _detail_mask_threshold() blend.c:355-359,
 *   dt_fast_expf() src/math/math.h:254-267
 */
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* the value of border pixel (row, col) after dt_masks_extend_border(): that of the nearest pixel of the interior */
static inline int inner(const int v, const int border, const int size)
{
  return v < border ? border : (v > size - border - 1 ? size - border - 1 : v);
}

static void extend_border(float *mask, const int width, const int height, const int border)
{
  for(int row = 0; row < height; row++)
    for(int col = 0; col < width; col++)
    {
      const int r = inner(row, border, height), c = inner(col, border, width);
      if(r != row || c != col) mask[(size_t)row * width + col] = mask[(size_t)r * width + c];
    }
}

void oracle_rawdetail_mask(const float *src, float *mask, const int width, const int height, const float wb[3])
{
  float *tmp = (float *)malloc(sizeof(float) * (size_t)width * height);
  for(size_t idx = 0; idx < (size_t)width * height; idx++)
  {
    const float val
        = 0.333333333f * (fmaxf(src[4 * idx], 0.0f) / wb[0] + fmaxf(src[4 * idx + 1], 0.0f) / wb[1] + fmaxf(src[4 * idx + 2], 0.0f) / wb[2]);
    tmp[idx] = sqrtf(val);
  }
  const float scale = 1.0f / 16.0f;
  #pragma omp parallel for schedule(static)
  for(int row = 1; row < height - 1; row++)
    for(int col = 1; col < width - 1; col++)
    {
      const size_t idx = (size_t)row * width + col;
      /* scharr operator */
      const float gx = 47.0f * (tmp[idx - width - 1] - tmp[idx - width + 1]) + 162.0f * (tmp[idx - 1] - tmp[idx + 1])
                       + 47.0f * (tmp[idx + width - 1] - tmp[idx + width + 1]);
      const float gy = 47.0f * (tmp[idx - width - 1] - tmp[idx + width - 1]) + 162.0f * (tmp[idx - width] - tmp[idx + width])
                       + 47.0f * (tmp[idx - width + 1] - tmp[idx + width + 1]);
      const float a = gx / 256.0f, b = gy / 256.0f;
      mask[idx] = scale * sqrtf(a * a + b * b);
    }
  extend_border(mask, width, height, 1);
  free(tmp);
}

int oracle_detailmask(const dt_hip_piece_t *piece, const dt_hip_detailmask_data_t *d, const void *in, void *out)
{
  const int width = piece->roi_out.width, height = piece->roi_out.height;
  if(!d->mask || piece->channels != 4 || width < 3 || height < 3) return 1;
  memcpy(out, in, sizeof(float) * 4 * (size_t)width * height);
  oracle_rawdetail_mask((const float *)out, (float *)d->mask, width, height, d->wb);
  return 0;
}

static inline float fast_expf(const float x)
{
  const float t = 1065353216.0f + x * 11401300.0f; /* i1 + x * (i2 - i1) */
  /* the conversion of an out-of-range value or a NaN is INT_MIN on the reference's target (cvttss2si) */
  const int k0 = (t > -2147483648.0f && t < 2147483648.0f) ? (int)t : INT_MIN;
  union { int k; float f; } u;
  u.k = k0 > 0 ? k0 : 0;
  return u.f;
}

/* dt_masks_blur_9x9_coeff(), detail.c:159-196 */
static void blur_9x9_coeff(float c[13], const float sigma)
{
  float kernel[9][9];
  const float temp = -2.0f * (sigma * sigma);
  const float range = (3.0f * 1.5f) * (3.0f * 1.5f);
  float sum = 0.0f;
  for(int k = -4; k <= 4; k++)
    for(int j = -4; j <= 4; j++)
    {
      const float d2 = (float)k * (float)k + (float)j * (float)j;
      if(d2 <= range)
      {
        kernel[k + 4][j + 4] = expf(d2 / temp);
        sum += kernel[k + 4][j + 4];
      }
      else
        kernel[k + 4][j + 4] = 0.0f;
    }
  for(int i = 0; i < 9; i++)
    for(int j = 0; j < 9; j++) kernel[i][j] /= sum;
  /* the thirteen distinct weights, by (|dy|, |dx|): 00 10 11 20 21 22 30 31 32 33 40 41 42 */
  static const int at[13][2] = { { 4, 4 }, { 3, 4 }, { 3, 3 }, { 2, 4 }, { 2, 3 }, { 2, 2 }, { 1, 4 },
                                 { 1, 3 }, { 1, 2 }, { 1, 1 }, { 0, 4 }, { 0, 3 }, { 0, 2 } };
  for(int k = 0; k < 13; k++) c[k] = kernel[at[k][0]][at[k][1]];
}

/* the ring of a weight: the offsets (dy, dx) whose samples the reference adds, in its order (FAST_BLUR_9, detail.c:205-218) */
typedef struct { int n; signed char o[8][2]; } ring_t;
static const ring_t RINGS[13] = {
  /* blurmat[12] */ { 8, { { -4, -2 }, { -4, 2 }, { -2, -4 }, { -2, 4 }, { 2, -4 }, { 2, 4 }, { 4, -2 }, { 4, 2 } } },
  /* blurmat[11] */ { 8, { { -4, -1 }, { -4, 1 }, { -1, -4 }, { -1, 4 }, { 1, -4 }, { 1, 4 }, { 4, -1 }, { 4, 1 } } },
  /* blurmat[10] */ { 4, { { -4, 0 }, { 0, -4 }, { 0, 4 }, { 4, 0 } } },
  /* blurmat[9]  */ { 4, { { -3, -3 }, { -3, 3 }, { 3, -3 }, { 3, 3 } } },
  /* blurmat[8]  */ { 8, { { -3, -2 }, { -3, 2 }, { -2, -3 }, { -2, 3 }, { 2, -3 }, { 2, 3 }, { 3, -2 }, { 3, 2 } } },
  /* blurmat[7]  */ { 8, { { -3, -1 }, { -3, 1 }, { -1, -3 }, { -1, 3 }, { 1, -3 }, { 1, 3 }, { 3, -1 }, { 3, 1 } } },
  /* blurmat[6]  */ { 4, { { -3, 0 }, { 0, -3 }, { 0, 3 }, { 3, 0 } } },
  /* blurmat[5]  */ { 4, { { -2, -2 }, { -2, 2 }, { 2, -2 }, { 2, 2 } } },
  /* blurmat[4]  */ { 8, { { -2, -1 }, { -2, 1 }, { -1, -2 }, { -1, 2 }, { 1, -2 }, { 1, 2 }, { 2, -1 }, { 2, 1 } } },
  /* blurmat[3]  */ { 4, { { -2, 0 }, { 0, -2 }, { 0, 2 }, { 2, 0 } } },
  /* blurmat[2]  */ { 4, { { -1, -1 }, { -1, 1 }, { 1, -1 }, { 1, 1 } } },
  /* blurmat[1]  */ { 4, { { -1, 0 }, { 0, -1 }, { 0, 1 }, { 1, 0 } } },
  /* blurmat[0]  */ { 1, { { 0, 0 } } },
};

/* dt_masks_calc_detail_mask(): `out` = the blurred sigmoid of the raw detail mask `rm`; level = the blend's details
 * threshold (!= 0) */
void oracle_detail_mask_threshold(const float *rm, float *out, int width, int height, float threshold, int detail);
void oracle_detail_mask(const float *rm, float *out, const int width, const int height, const float level)
{
  const int detail = level > 0.0f;
  /* _detail_mask_threshold(), blend.c:355-359 */
  const float threshold = 0.005f * (detail ? powf(level, 2.0f) : 1.0f - powf(fabs(level), 0.5f));
  oracle_detail_mask_threshold(rm, out, width, height, threshold, detail);
}

/* dt_masks_calc_detail_mask(src, out, tmp, width, height, threshold, detail), src/develop/masks/detail.c:325-335; rm == out
 * is allowed (the reference's dual demosaic calls it in place) */
void oracle_detail_mask_threshold(const float *rm, float *out, const int width, const int height, const float threshold,
                                  const int detail)
{
  float *tmp = (float *)malloc(sizeof(float) * (size_t)width * height);
  for(size_t idx = 0; idx < (size_t)width * height; idx++)
  {
    /* calcBlendFactor(), detail.c:317-323 */
    const float blend = 1.0f / (1.0f + fast_expf(16.0f - (16.0f / threshold) * rm[idx]));
    tmp[idx] = detail ? blend : 1.0f - blend;
  }
  float blurmat[13];
  blur_9x9_coeff(blurmat, 2.0f);
  #pragma omp parallel for schedule(static)
  for(int row = 4; row < height - 4; row++)
    for(int col = 4; col < width - 4; col++)
    {
      const float *const src = tmp + (size_t)row * width + col;
      float acc = 0.0f;
      for(int k = 0; k < 13; k++)
      {
        const ring_t *r = &RINGS[k];
        float ring = src[r->o[0][0] * width + r->o[0][1]];
        for(int s = 1; s < r->n; s++) ring = ring + src[r->o[s][0] * width + r->o[s][1]];
        const float term = blurmat[12 - k] * ring;
        acc = k ? acc + term : term;
      }
      out[(size_t)row * width + col] = fminf(1.0f, fmaxf(0.0f, acc));
    }
  extend_border(out, width, height, 4);
  free(tmp);
}
