/* oracle/src/finalscale.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of the export resampler behind finalscale:
 *   process()                        src/iop/finalscale.c:117-131 (roi origins zeroed)
 *   _interpolation_resample_plain()  src/pixel/interpolation.c:898-1030
 *   _prepare_resampling_plan()       :711-895, BORDER_REPLICATE (:62)
 *   _compute_upsampling_kernel()     :320-344   _compute_downsampling_kernel() :354-392
 *   _maketaps_bilinear/bicubic/mitchell :175-296 -- taps are produced four at a time and the tap
 *   position advances by a float accumulation (vt += 4 * interval), which is kept
 *   ceil_fast()                      src/math/math.h:324-334 (x > 0: -(float)(int)-x + 1, so an exact
 *                                    integer maps to x + 1)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static inline float ceil_fast(const float x) { return x <= 0.f ? (float)(int)x : -((float)(int)-x) + 1.f; }
static inline float max_zero(const float v) { return isfinite(v) ? (v > 0.0f ? v : 0.0f) : 0.0f; }

static const int half_width[3] = { 1, 2, 2 };

static float tap_value(const int kind, const float vt)
{
  const float a = fabsf(vt);
  if(kind == DT_HIP_INTERPOLATION_BILINEAR) return 1.0f - a;
  if(kind == DT_HIP_INTERPOLATION_BICUBIC)
  {
    const float t2 = vt * vt, t5 = 5.0f * a;
    const float r12 = (a * (t5 - 8.0f - t2) + 4.0f) * 0.5f;
    const float r01 = ((3.0f * t2 - t5) * a + 2.0f) * 0.5f;
    return a <= 1.0f ? r01 : r12;
  }
  const float a2 = a * a, a3 = a2 * a;
  const float r01 = (7.0f / 6.0f) * a3 - 2.0f * a2 + (8.0f / 9.0f);
  const float r12 = 2.0f * a2 - (7.0f / 18.0f) * a3 - (10.0f / 3.0f) * a + (16.0f / 9.0f);
  return a <= 1.0f ? r01 : r12;
}

/* taps[k], k < num_taps, as the 4-wide loops of _maketaps_*() produce them */
static void make_taps(const int kind, float *taps, const int num_taps, const float first_tap, const float interval)
{
  const float iter = 4.0f * interval;
  float vt[4];
  for(int c = 0; c < 4; c++) vt[c] = first_tap + (float)c * interval;
  const int runs = (num_taps + 3) / 4;
  for(int i = 0; i < runs; i++)
    for(int c = 0; c < 4; c++)
    {
      taps[4 * i + c] = tap_value(kind, vt[c]);
      vt[c] += iter;
    }
}

typedef struct
{
  int *length, *start, *index;
  float *kernel;
} plan_t;

static void plan_free(plan_t *p)
{
  free(p->length);
  free(p->start);
  free(p->index);
  free(p->kernel);
}

static void plan_build(plan_t *p, const int kind, const int in, const int in_x0, const int out, const int out_x0, const float scale)
{
  const int w = half_width[kind];
  const int maxtaps = scale > 1.f ? 2 * w : (int)ceil_fast((float)2 * (float)w / scale);
  p->length = (int *)malloc(sizeof(int) * out);
  p->start = (int *)malloc(sizeof(int) * out);
  p->index = (int *)malloc(sizeof(int) * (size_t)(maxtaps + 4) * out);
  p->kernel = (float *)malloc(sizeof(float) * (size_t)(maxtaps + 4) * out);
  float *scratch = (float *)malloc(sizeof(float) * (maxtaps + 8));
  int k = 0;
  for(int x = 0; x < out; x++)
  {
    int first, taps;
    if(scale > 1.f)
    {
      const float fx = (float)(out_x0 + x) / scale - in_x0;
      first = (int)floorf(fx) - w + 1;
      taps = 2 * w;
      make_taps(kind, scratch, taps, fx - (float)first, -1.0f);
    }
    else
    {
      /* _compute_downsampling_kernel(.., out_x0 + x): the first tap is an ABSOLUTE input index, in_x0 is not subtracted
       * (interpolation.c:853) */
      const int xo = out_x0 + x;
      const float xin = ceil_fast(((float)xo - (float)w) / scale);
      first = (int)xin;
      const float t = xin * scale - (float)xo;
      taps = (int)(((float)w - t) / scale);
      make_taps(kind, scratch, taps, t, scale);
    }
    p->length[x] = taps;
    p->start[x] = k;
    float norm = 0.f;
    for(int t = 0; t < taps; t++) norm += scratch[t];
    norm = 1.f / norm;
    for(int t = 0; t < taps; t++)
    {
      p->kernel[k] = scratch[t] * norm;
      int i = first + t;
      p->index[k] = i < 0 ? 0 : (i > in - 1 ? in - 1 : i);
      k++;
    }
  }
  free(scratch);
}

/* dt_interpolation_resample(), interpolation.c:897-1044, on 4-channel pixels; origins: use the regions' x / y */
static int resample(const dt_hip_piece_t *piece, const int kind, const float *in, float *out, const int origins)
{
  if(kind < 0 || kind > 2) return 1;
  const int iw = piece->roi_in.width, ih = piece->roi_in.height, ow = piece->roi_out.width, oh = piece->roi_out.height;
  const int ix0 = origins ? piece->roi_in.x : 0, iy0 = origins ? piece->roi_in.y : 0;
  const int ox0 = origins ? piece->roi_out.x : 0, oy0 = origins ? piece->roi_out.y : 0;
  const float so = (float)piece->roi_out.scale, si = (float)piece->roi_in.scale;
  if(so == 1.f || so == si)
  {
    /* :915-931: a crop */
    const int x0 = ox0 - ix0, y0 = oy0 - iy0;
    if(x0 < 0 || y0 < 0 || x0 + ow > iw || y0 + oh > ih) return 1; /* the reference would read outside its input */
    for(int y = 0; y < oh; y++) memcpy(out + 4 * (size_t)ow * y, in + 4 * ((size_t)iw * (y + y0) + x0), sizeof(float) * 4 * ow);
    return 0;
  }
  const float scale = so / si;
  plan_t h, v;
  plan_build(&h, kind, iw, ix0, ow, ox0, scale);
  plan_build(&v, kind, ih, iy0, oh, oy0, scale);
#pragma omp parallel for
  for(int oy = 0; oy < oh; oy++)
    for(int ox = 0; ox < ow; ox++)
    {
      float vs[4] = { 0.f, 0.f, 0.f, 0.f };
      for(int iy = 0; iy < v.length[oy]; iy++)
      {
        const size_t base = (size_t)v.index[v.start[oy] + iy] * iw * 4;
        float vhs[4] = { 0.f, 0.f, 0.f, 0.f };
        for(int ix = 0; ix < h.length[ox]; ix++)
        {
          const float *px = in + base + (size_t)h.index[h.start[ox] + ix] * 4;
          const float htap = h.kernel[h.start[ox] + ix];
          for(int c = 0; c < 4; c++) vhs[c] += px[c] * htap;
        }
        const float vtap = v.kernel[v.start[oy] + iy];
        for(int c = 0; c < 4; c++) vs[c] += vhs[c] * vtap;
      }
      for(int c = 0; c < 4; c++) out[4 * ((size_t)oy * ow + ox) + c] = max_zero(vs[c]);
    }
  plan_free(&h);
  plan_free(&v);
  return 0;
}

/* finalscale: process(), src/iop/finalscale.c:117-131 -- the regions' origins are zeroed, only the sizes and scales count */
int oracle_finalscale(const dt_hip_piece_t *piece, const dt_hip_finalscale_data_t *d, const void *in_, void *out_)
{
  return resample(piece, d->interpolation, (const float *)in_, (float *)out_, 0);
}

/* initialscale: process(), src/iop/initialscale.c:120-127 -- dt_iop_clip_and_zoom_roi() with the regions as they are:
 * roi_in is the whole input buffer at scale 1 (modify_roi_in(), :72-83), roi_out a region of the scaled image */
int oracle_initialscale(const dt_hip_piece_t *piece, const dt_hip_finalscale_data_t *d, const void *in_, void *out_)
{
  return resample(piece, d->interpolation, (const float *)in_, (float *)out_, 1);
}
