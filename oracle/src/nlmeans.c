/* oracle/src/nlmeans.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of the non-local-means core and of the denoise (non-local means) module:
 *   nlmeans_denoise()        src/pixel/nlmeans_core.c:315-532
 *   define_patches()         :108-150 with scatter() :95-105
 *   init_column_sums()       :208-262, pixel_difference() :160-170, diff_of_pixels_diff() :172-184
 *   compute_slice_height()   :264-295, compute_slice_width() :297-313
 *   dt_fast_mexp2f()         src/math/math.h:290-301
 *   process_cpu() (nlmeans)  src/iop/nlmeans.c:416-457
 *
 * The reference walks the frame in chunks (about 72 x 60 pixels, a pure function of the frame size),
 * and inside a chunk keeps, per patch offset, a sliding column sum of squared pixel differences that
 * is updated incrementally from row to row, plus a sliding row sum ("distortion") updated from column
 * to column.  Both are binary32 recurrences, so their rounding depends on the chunk grid and on the
 * walking order; the restatement keeps exactly those recurrences but separates them into three steps
 * per (chunk, patch) -- (A) the column sums of every row of the chunk, top to bottom; (B) the
 * distortion and weight of every pixel, left to right; (C) the weighted accumulation -- which is
 * also how the device executes it (ansel_amd/csrc/nlmeans.hip).
 */
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "nlmeans_core.h"

#define SLICE_WIDTH 72  /* nlmeans_core.c:55 */
#define SLICE_HEIGHT 60 /* nlmeans_core.c:56 */

static inline int imin(const int a, const int b) { return a < b ? a : b; }
static inline int imax(const int a, const int b) { return a > b ? a : b; }

/* float -> int as the reference's target converts (cvttss2si): out of range and NaN give INT_MIN */
static inline int cvtt(const float v)
{
  if(!(v >= -2147483648.0f && v < 2147483648.0f)) return INT_MIN;
  return (int)v;
}

/* dt_fast_mexp2f(), math.h:290-301 (two's-complement wrap of the int addition, as compiled) */
static inline float mexp2(const float x)
{
  const int i1 = 0x3f800000, i2 = 0x3f000000;
  const int k0 = (int)((unsigned)i1 + (unsigned)cvtt(x * (float)(i2 - i1)));
  union { float f; int i; } k;
  k.i = k0 >= 0x800000 ? k0 : 0;
  return k.f;
}

static inline float pixdiff(const float *a, const float *b, const float *norm)
{
  float s[3];
  for(int i = 0; i < 3; i++)
  {
    const float d = a[i] - b[i];
    s[i] = d * d * norm[i];
  }
  return s[0] + s[1] + s[2];
}

static inline float pixdiff2(const float *a, const float *b, const float *c, const float *d, const float *norm)
{
  float s[3];
  for(int i = 0; i < 3; i++)
  {
    const float d1 = a[i] - b[i], d2 = c[i] - d[i];
    s[i] = (d1 * d1 - d2 * d2) * norm[i];
  }
  return s[0] + s[1] + s[2];
}

static int sgn(const int a) { return (a > 0) - (a < 0); }

static int scatter(const float scale, const float scattering, const int i1, const int i2)
{
  const int a1 = abs(i1), a2 = abs(i2);
  return (int)(scale * ((a1 * a1 * a1 + 7.0 * a1 * sqrt(a2)) * sgn(i1) * scattering / 6.0 + i1));
}

int oracle_nlmeans_slice_height(const int height)
{
  if(height % SLICE_HEIGHT == 0) return SLICE_HEIGHT;
  int best = height % SLICE_HEIGHT, best_incr = 0;
  for(int incr = 1; incr < 10; incr++)
  {
    const int plus = height % (SLICE_HEIGHT + incr);
    if(plus == 0) return SLICE_HEIGHT + incr;
    if(plus > best)
    {
      best_incr = incr;
      best = plus;
    }
    const int minus = height % (SLICE_HEIGHT - incr);
    if(minus == 0) return SLICE_HEIGHT - incr;
    if(minus > best)
    {
      best_incr = -incr;
      best = minus;
    }
  }
  return SLICE_HEIGHT + best_incr;
}

int oracle_nlmeans_slice_width(const int width)
{
  int sl = SLICE_WIDTH;
  int rem = width % sl;
  if(rem < SLICE_WIDTH / 2 && (width % (sl - 4)) > rem)
  {
    sl -= 4;
    rem = width % sl;
    if(rem < SLICE_WIDTH / 2 && (width % (sl - 4)) > rem) sl -= 4;
  }
  return sl;
}

void oracle_nlmeans_core(const float *in, float *out, const int W, const int H, const oracle_nlm_params_t *pr)
{
  const int radius = pr->patch_radius, K = pr->search_radius;
  const int stride = 4 * W;
  const int npatch = (2 * K + 1) * (2 * K + 1);
  int *prow = (int *)malloc(sizeof(int) * npatch), *pcol = (int *)malloc(sizeof(int) * npatch);
  int np = 0;
  for(int ri = -K; ri <= K; ri++)
    for(int ci = -K; ci <= K; ci++)
    {
      prow[np] = scatter(pr->scale, pr->scattering, ri, ci);
      pcol[np] = scatter(pr->scale, pr->scattering, ci, ri);
      np++;
    }
  const float cpn = pr->center_weight * (2 * radius + 1) * (2 * radius + 1); /* compute_center_pixel_norm() */
  const float center_norm[4] = { cpn, cpn, cpn, 1.0f };
  const int chk_h = oracle_nlmeans_slice_height(H), chk_w = oracle_nlmeans_slice_width(W);
  const int nchy = (H + chk_h - 1) / chk_h, nchx = (W + chk_w - 1) / chk_w;

#pragma omp parallel for collapse(2) schedule(dynamic)
  for(int cy = 0; cy < nchy; cy++)
    for(int cx = 0; cx < nchx; cx++)
    {
      const int top = cy * chk_h, left = cx * chk_w;
      const int bot = imin(top + chk_h, H), right = imin(left + chk_w, W);
      const int csw = chk_w + 2 * radius + 1; /* columns left - radius - 1 .. right + radius - 1 */
      float *cs = (float *)malloc(sizeof(float) * (size_t)csw * (bot - top));
      float *wt = (float *)malloc(sizeof(float) * (size_t)chk_w * (bot - top));
      for(int i = top; i < bot; i++) memset(out + 4 * ((size_t)i * W + left), 0, sizeof(float) * 4 * (right - left));
      for(int p = 0; p < np; p++)
      {
        const int srow = prow[p], scol = pcol[p], offset = srow * stride + scol * 4;
        const int row_min = imax(top, imax(0, -srow)), row_max = imin(bot, H - imax(0, srow));
        const int row_top = imax(row_min, imax(radius, radius - srow));
        const int row_bot = imin(row_max, H - 1 - imax(radius, radius + srow));
        const int col_min = imax(left, -scol), col_max = imin(right, W - scol);
        const int pc_min = left - imin(radius, imin(left, left + scol));
        const int pc_max = right + imin(radius, imin(W - right, W - (right + scol)));
        if(row_min >= row_max) continue;
#define CS(r, c) cs[(size_t)((r) - top) * csw + ((c) - (left - radius - 1))]
        /* (A) column sums: from scratch on the first row (init_column_sums), then incrementally */
        {
          const int row = row_min;
          const int rmin = row - imin(radius, imin(row, row + srow));
          const int rmax = row + imin(radius, imin(H - 1 - row, H - 1 - (row + srow)));
          for(int c = left - radius - 1; c < right + radius; c++) CS(row, c) = 0.0f;
          for(int c = pc_min; c < pc_max; c++)
          {
            float sum = 0;
            for(int r = rmin; r <= rmax; r++)
            {
              const float *px = in + (size_t)r * stride + 4 * c;
              sum += pixdiff(px, px + offset, pr->norm);
            }
            CS(row, c) = sum;
          }
        }
        for(int row = row_min; row + 1 < row_max; row++)
        {
          for(int c = left - radius - 1; c < right + radius; c++) CS(row + 1, c) = CS(row, c);
          if(row < imin(row_top, row_bot))
          {
            const float *b = in + (size_t)(row + 1 + radius) * stride;
            for(int c = pc_min; c < pc_max; c++) CS(row + 1, c) += pixdiff(b + 4 * c, b + 4 * c + offset, pr->norm);
          }
          else if(row < row_bot)
          {
            const float *t = in + (size_t)(row - radius) * stride, *b = in + (size_t)(row + 1 + radius) * stride;
            for(int c = pc_min; c < pc_max; c++)
              CS(row + 1, c) += pixdiff2(b + 4 * c, b + 4 * c + offset, t + 4 * c, t + 4 * c + offset, pr->norm);
          }
          else if(row >= row_top)
          {
            const float *t = in + (size_t)(row - radius) * stride;
            for(int c = pc_min; c < pc_max; c++) CS(row + 1, c) -= pixdiff(t + 4 * c, t + 4 * c + offset, pr->norm);
          }
        }
        /* (B) sliding row sum and weight of every pixel of the chunk this patch reaches */
        for(int row = row_min; row < row_max; row++)
        {
          float distortion = 0.0f;
          for(int i = col_min - radius; i < imin(col_min + radius, col_max); i++) distortion += CS(row, i);
          const float *irow = in + (size_t)row * stride;
          for(int c = col_min; c < col_max; c++)
          {
            distortion += (CS(row, c + radius) - CS(row, c - radius - 1));
            float w;
            if(pr->center_weight < 0)
              w = mexp2(distortion * pr->sharpness);
            else
            {
              const float dis = (distortion + pixdiff(irow + 4 * c, irow + 4 * c + offset, center_norm))
                                / (1.0f + pr->center_weight);
              w = mexp2(fmaxf(0.0f, dis * pr->sharpness - 2.0f));
            }
            wt[(size_t)(row - top) * chk_w + (c - left)] = w;
          }
        }
        /* (C) accumulate the patch-centre pixels, weight in channel 3 */
        for(int row = row_min; row < row_max; row++)
          for(int c = col_min; c < col_max; c++)
          {
            const float w = wt[(size_t)(row - top) * chk_w + (c - left)];
            const float *px = in + (size_t)row * stride + 4 * c + offset;
            float *o = out + 4 * ((size_t)row * W + c);
            o[0] += px[0] * w;
            o[1] += px[1] * w;
            o[2] += px[2] * w;
            o[3] += 1.0f * w;
          }
#undef CS
      }
      /* normalise (and blend with the input), nlmeans_core.c:490-521 */
      const int skip_blend = (pr->luma == 1.0 && pr->chroma == 1.0);
      const float weight[4] = { pr->luma, pr->chroma, pr->chroma, 1.0f };
      const float invert[4] = { 1.0f - pr->luma, 1.0f - pr->chroma, 1.0f - pr->chroma, 0.0f };
      for(int row = top; row < bot; row++)
        for(int c = left; c < right; c++)
        {
          float *o = out + 4 * ((size_t)row * W + c);
          const float *ip = in + (size_t)row * stride + 4 * c;
          if(skip_blend)
            for(int k = 0; k < 4; k++) o[k] /= o[3];
          else
            for(int k = 0; k < 4; k++) o[k] = (ip[k] * invert[k]) + (o[k] / o[3] * weight[k]);
        }
      free(cs);
      free(wt);
    }
  free(prow);
  free(pcol);
}

/* process_cpu(), src/iop/nlmeans.c:416-457 */
int oracle_nlmeans(const dt_hip_piece_t *piece, const dt_hip_nlmeans_data_t *d, const void *in, void *out)
{
  const float scale = (float)fmin(piece->roi_in.scale, 2.0f);
  const float max_L = 120.0f, max_C = 512.0f;
  const float nL = 1.0f / max_L, nC = 1.0f / max_C;
  oracle_nlm_params_t p;
  memset(&p, 0, sizeof(p));
  p.scattering = 0;
  p.scale = scale;
  p.luma = d->luma;
  p.chroma = d->chroma;
  p.center_weight = -1;
  p.sharpness = 3000.0f / (1.0f + d->strength);
  p.patch_radius = (int)ceilf(d->radius * scale);
  p.search_radius = (int)ceilf(7 * scale);
  p.norm[0] = nL * nL;
  p.norm[1] = nC * nC;
  p.norm[2] = nC * nC;
  p.norm[3] = 1.0f;
  oracle_nlmeans_core((const float *)in, (float *)out, piece->roi_out.width, piece->roi_out.height, &p);
  return 0;
}
