/* oracle/src/guided_filter.c -- TEST INFRASTRUCTURE ONLY (oracle/src/oracle.h).
 *
 * The guided filter that feathers a blend mask: guided_filter(), src/pixel/guided_filter.c:369-402, its per-tile
 * body guided_filter_tiling(), :122-330, and the compensated box means it is built on, src/pixel/box_filters.c.
 *
 * What the result depends on, all of it reproduced:
 *  - the TILE GRID: targets of max(3 w, 512) pixels a side, each filtered on its own source region (the target grown
 *    by 2 w, clipped at the image): the box means restart at every source border (guided_filter.c:126-129, :387-399);
 *  - the box mean of a scan line is a sliding Kahan sum divided by the number of samples under the window
 *    (box_filters.c:408-454 for rows; :577-643, :706-764, :829-888 for columns -- the 16-, 4- and 1-wide column
 *    variants perform the same operations per scalar column);
 *  - except in the 1-wide variant, whose tail ADDS the sample leaving the window instead of subtracting it
 *    (box_filters.c:630-640).  It runs on the scalar columns a 16-wide and a 4-wide pass leave over (:982-996): the
 *    last (9 * source width) % 4 of the nine-plane variance image, never on a four-plane image.
 * PARITY PIN: tests/test_oracle_vs_ref.py compares this file with the reference's own guided_filter.c + box_filters.c
 * (compiled in place into oracle/_ref/libansel_ref.so) bit for bit. */
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* Kahan_sum(), src/math/math.h:105-111 */
static inline float kahan(const float m, float *c, const float add)
{
  const float t1 = add - (*c);
  const float t2 = m + t1;
  *c = (t2 - m) - t1;
  return t2;
}

/* one scan line of n samples, `stride` floats apart, replaced by its box means; v: n floats of scratch */
static void box_mean_line(float *line, const size_t stride, const int n, const int radius, const int tail_adds, float *v)
{
  for(int i = 0; i < n; i++) v[i] = line[(size_t)i * stride];
  float L = 0.0f, c = 0.0f;
  int hits = 0;
  for(int x = 0; x < (radius < n ? radius : n); x++)
  {
    hits++;
    L = kahan(L, &c, v[x]);
  }
  int x;
  for(x = 0; x <= radius && x + radius < n; x++)
  {
    hits++;
    L = kahan(L, &c, v[x + radius]);
    line[(size_t)x * stride] = L / (float)hits;
  }
  for(; x <= radius && x < n; x++) line[(size_t)x * stride] = L / (float)hits;
  for(; x + radius < n; x++)
  {
    L = kahan(L, &c, -v[x - radius - 1]);
    L = kahan(L, &c, v[x + radius]);
    line[(size_t)x * stride] = L / (float)hits;
  }
  for(; x < n; x++)
  {
    hits--;
    L = kahan(L, &c, tail_adds ? v[x - radius - 1] : -v[x - radius - 1]);
    line[(size_t)x * stride] = L / (float)hits;
  }
}

/* rows: blur_horizontal_4ch_Kahan() / blur_horizontal_Nch_Kahan(), one line per plane of the interleaved image */
static void box_mean_rows(float *img, const int height, const int width, const int planes, const int radius)
{
#pragma omp parallel
  {
    float *v = (float *)malloc(sizeof(float) * (size_t)width);
#pragma omp for schedule(static)
    for(int j = 0; j < height; j++)
      for(int c = 0; c < planes; c++) box_mean_line(img + (size_t)j * width * planes + c, planes, width, radius, 0, v);
    free(v);
  }
}

/* columns: box_mean_vert_1ch_Kahan() over planes * width scalar columns */
static void box_mean_columns(float *img, const int height, const int width, const int planes, const int radius)
{
  const size_t cols = (size_t)planes * width;
#pragma omp parallel
  {
    float *v = (float *)malloc(sizeof(float) * (size_t)height);
#pragma omp for schedule(static)
    for(size_t k = 0; k < cols; k++) box_mean_line(img + k, cols, height, radius, k >= (cols & ~(size_t)3), v);
    free(v);
  }
}

static inline int imin(const int a, const int b) { return a < b ? a : b; }
static inline int imax(const int a, const int b) { return a > b ? a : b; }

/* guided_filter_tiling(), guided_filter.c:122-330 */
static int filter_tile(const float *guide, const float *in, float *out, const int iw, const int ih, const int ch,
                       const int tl, const int tr, const int tlo, const int tup, const int w, const float eps,
                       const float guide_weight, const float min, const float max)
{
  const int sl = imax(tl - 2 * w, 0), sr = imin(tr + 2 * w, iw), slo = imax(tlo - 2 * w, 0), sup = imin(tup + 2 * w, ih);
  const int width = sr - sl, height = sup - slo;
  const size_t size = (size_t)width * height;
  float *mean = (float *)malloc(sizeof(float) * size * 4), *var = (float *)malloc(sizeof(float) * size * 9);
  if(!mean || !var)
  {
    free(mean);
    free(var);
    return 1;
  }
#pragma omp parallel for schedule(static)
  for(int j = 0; j < height; j++)
    for(int i = 0; i < width; i++)
    {
      const float *px = guide + ((size_t)(slo + j) * iw + sl + i) * ch;
      const float p0 = px[0] * guide_weight, p1 = px[1] * guide_weight, p2 = px[2] * guide_weight;
      const float input = in[(size_t)(slo + j) * iw + sl + i];
      float *m = mean + ((size_t)j * width + i) * 4, *v = var + ((size_t)j * width + i) * 9;
      m[0] = input;
      m[1] = p0;
      m[2] = p1;
      m[3] = p2;
      v[0] = p0 * input;
      v[1] = p1 * input;
      v[2] = p2 * input;
      v[3] = p0 * p0;
      v[4] = p0 * p1;
      v[5] = p0 * p2;
      v[6] = p1 * p1;
      v[7] = p1 * p2;
      v[8] = p2 * p2;
    }
  box_mean_rows(mean, height, width, 4, w);
  box_mean_rows(var, height, width, 9, w);
  box_mean_columns(mean, height, width, 4, w);
  box_mean_columns(var, height, width, 9, w);
  /* the coefficients a_r, a_g, a_b, b over the means: Cramer's rule on the regularised covariance, :223-287 */
#pragma omp parallel for schedule(static)
  for(size_t i = 0; i < size; i++)
  {
    float *m = mean + i * 4;
    const float *v = var + i * 9;
    const float inp_mean = m[0], guide_r = m[1], guide_g = m[2], guide_b = m[3];
    const float S00 = v[3] - (guide_r * guide_r) + eps;
    const float S01 = v[4] - (guide_r * guide_g);
    const float S02 = v[5] - (guide_r * guide_b);
    const float S11 = v[6] - (guide_g * guide_g) + eps;
    const float S12 = v[7] - (guide_g * guide_b);
    const float S22 = v[8] - (guide_b * guide_b) + eps;
    const float det0 = S00 * (S11 * S22 - S12 * S12) - S01 * (S01 * S22 - S02 * S12) + S02 * (S01 * S12 - S02 * S11);
    float ar, ag, ab, b;
    if(__builtin_fabsf(det0) > 4.f * 1.1920928955078125e-7f)
    {
      const float cov_r = v[0] - guide_r * inp_mean;
      const float cov_g = v[1] - guide_g * inp_mean;
      const float cov_b = v[2] - guide_b * inp_mean;
      const float det1 = cov_r * (S11 * S22 - S12 * S12) - S01 * (cov_g * S22 - cov_b * S12) + S02 * (cov_g * S12 - cov_b * S11);
      const float det2 = S00 * (cov_g * S22 - cov_b * S12) - cov_r * (S01 * S22 - S02 * S12) + S02 * (S01 * cov_b - S02 * cov_g);
      const float det3 = S00 * (S11 * cov_b - S12 * cov_g) - S01 * (S01 * cov_b - S02 * cov_g) + cov_r * (S01 * S12 - S02 * S11);
      ar = det1 / det0;
      ag = det2 / det0;
      ab = det3 / det0;
      b = inp_mean - ar * guide_r - ag * guide_g - ab * guide_b;
    }
    else
    {
      ar = ag = ab = 0.f;
      b = inp_mean;
    }
    m[0] = ar;
    m[1] = ag;
    m[2] = ab;
    m[3] = b;
  }
  free(var);
  /* dt_box_mean(a_b, 4 | KAHAN, w, 1): rows, then columns (dt_box_mean_4ch_Kahan(), box_filters.c:1004-1026) */
  box_mean_rows(mean, height, width, 4, w);
  box_mean_columns(mean, height, width, 4, w);
#pragma omp parallel for schedule(static)
  for(int j = tlo; j < tup; j++)
    for(int i = tl; i < tr; i++)
    {
      const float *px = guide + ((size_t)j * iw + i) * ch;
      const float *ab = mean + ((size_t)(j - slo) * width + (i - sl)) * 4;
      float res = guide_weight * (ab[0] * px[0] + ab[1] * px[1] + ab[2] * px[2]);
      res += ab[3];
      out[(size_t)j * iw + i] = (res > max) ? max : ((res < min) ? min : res); /* CLAMP() of glib */
    }
  free(mean);
  return 0;
}

/* guided_filter(), guided_filter.c:369-402.  guide: width x height pixels of ch floats; in / out: width x height */
int oracle_guided_filter(const float *guide, const float *in, float *out, const int width, const int height, const int ch,
                         const int w, const float sqrt_eps, const float guide_weight, const float min, const float max)
{
  if(ch < 3 || w < 1) return 1;
  const int tile_width = imax(3 * w, 512), tile_height = imax(3 * w, 512);
  const float eps = sqrt_eps * sqrt_eps;
  for(int j = 0; j < height; j += tile_height)
    for(int i = 0; i < width; i += tile_width)
      if(filter_tile(guide, in, out, width, height, ch, i, imin(i + tile_width, width), j, imin(j + tile_height, height), w, eps,
                     guide_weight, min, max))
        return 1;
  return 0;
}
