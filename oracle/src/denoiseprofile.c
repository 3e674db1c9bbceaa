/* oracle/src/denoiseprofile.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of denoise (profiled), wavelets mode:
 *   process_wavelets()            src/iop/denoiseprofile.c:1289-1447
 *   compute_wb_factors()          :1097-1128      set_up_conversion_matrices()  :1163-1221
 *   precondition / _v2 / _Y0U0V0  :852-870, :916-933, :1021-1051
 *   backtransform / _v2 / _Y0U0V0 :872-897, :996-1019, :1053-1090
 *   variance_stabilizing_xform()  :1223-1287
 *   eaw_dn_decompose()            src/pixel/eaw.c:242-327 with dn_weight() :181-195, fast_mexp2f() src/math/math.h:306-317
 *   eaw_synthesize()              src/pixel/eaw.c:157-175
 *
 * ONE DELIBERATE DIFFERENCE.  The reference sums the squared detail coefficients of a band with an
 * OpenMP float reduction over rows (eaw.c:253-255): its value depends on the number of host threads
 * (and stalls near 2^24 on one thread at 24 MP), so there is no single reference value to match.
 * The oracle -- and the device -- define the sum canonically: every det*det (binary32 product) is
 * widened to binary64 and added in this fixed order, then rounded once to binary32:
 *   1. a row is cut into segments of 256 pixels; inside a segment each run of 64 pixels is reduced
 *      by halving (v[m] += v[m + off], off = 32,16,..,1; missing pixels are 0) and the four run sums
 *      are added left to right;
 *   2. the segment sums, numbered row-major, are dealt round-robin to 1024 accumulators (segment s
 *      goes to accumulator s % 1024, added in increasing s) and the accumulators are reduced by
 *      halving (off = 512,..,1).
 * tests/test_oracle_vs_ref.py bounds the effect against the reference run with its default threads.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "nlmeans_core.h"

#define BANDS DT_HIP_DENOISEPROFILE_BANDS
#define P_FULCRUM 0.05f /* denoiseprofile.c:118 */

static inline float max_first(const float a, const float b) { return a > b ? a : b; } /* MAX(a, b) */
static inline float min_first(const float a, const float b) { return a < b ? a : b; } /* MIN(a, b) */
static inline int clampi(const int v, const int lo, const int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* fast_mexp2f(), math.h:306-317 (the float-constant variant the denoisers keep for compatibility) */
static inline float mexp2(const float x)
{
  const float i1 = 1065353216.0f, i2 = 1056964608.0f;
  const float k0 = i1 + x * (i2 - i1);
  union { float f; int i; } k;
  k.i = k0 >= 8388608.0f ? (int)k0 : 0;
  return k.f;
}

typedef struct
{
  int max_scale;
  float wb[4], p[4], aa[4], bb[4];
  float toY[3][4], toRGB[3][4];
  float a_v2, b_v2, bias;
  int vst; /* 0 legacy, 1 v2 RGB, 2 Y0U0V0 */
} dn_setup_t;

/* invert_matrix(), denoiseprofile.c:1132-1161 */
static int invert3(const float in[3][4], float out[3][4])
{
  const float A = in[1][1] * in[2][2] - in[1][2] * in[2][1];
  const float B = -in[1][0] * in[2][2] + in[1][2] * in[2][0];
  const float C = in[1][0] * in[2][1] - in[1][1] * in[2][0];
  const float D = -in[0][1] * in[2][2] + in[0][2] * in[2][1];
  const float E = in[0][0] * in[2][2] - in[0][2] * in[2][0];
  const float F = -in[0][0] * in[2][1] + in[0][1] * in[2][0];
  const float G = in[0][1] * in[1][2] - in[0][2] * in[1][1];
  const float H = -in[0][0] * in[1][2] + in[0][2] * in[1][0];
  const float I = in[0][0] * in[1][1] - in[0][1] * in[1][0];
  const float det = in[0][0] * A + in[0][1] * B + in[0][2] * C;
  if(det == 0.0f) return 0;
  const float r = 1.0f / det;
  out[0][0] = r * A; out[0][1] = r * D; out[0][2] = r * G; out[0][3] = 0.0f;
  out[1][0] = r * B; out[1][1] = r * E; out[1][2] = r * H; out[1][3] = 0.0f;
  out[2][0] = r * C; out[2][1] = r * F; out[2][2] = r * I; out[2][3] = 0.0f;
  return 1;
}

/* nlm = 0: process_wavelets() :1331-1378; nlm = 1: nlmeans_precondition() :1510-1546 */
static int dn_setup(const dt_hip_piece_t *piece, const dt_hip_denoiseprofile_data_t *d, dn_setup_t *s, const int nlm)
{
  memset(s, 0, sizeof(*s));
  const float in_scale = fminf((float)piece->roi_in.scale, 1.0f);
  /* number of bands, denoiseprofile.c:1301-1317 */
  const float big = (float)(piece->roi_in.height > piece->roi_in.width ? piece->roi_in.height : piece->roi_in.width);
  const float supp0 = min_first((float)(2 * (2u << (BANDS - 1)) + 1), big * 0.2f);
  const float i0 = log2f((supp0 - 1.0f) * .5f);
  int max_scale = 0;
  for(; max_scale < BANDS; max_scale++)
  {
    const float supp = (float)(2 * (2u << max_scale) + 1);
    const float supp_in = supp * (1.0f / in_scale);
    const float i_in = log2f((supp_in - 1) * .5f) - 1.0f;
    const float t = 1.0f - (i_in + .5f) / i0;
    if(t < 0.0f) break;
  }
  s->max_scale = max_scale;

  /* compute_wb_factors() with weights {2, 1, 2, 0}, :1097-1128 */
  const float weights[4] = { nlm ? 1.0f : 2.0f, 1.0f, nlm ? 1.0f : 2.0f, 0.0f };
  const float wb_mean = (d->wb_coeffs[0] + d->wb_coeffs[1] + d->wb_coeffs[2]) / 3.0f;
  float *wb = s->wb;
  wb[0] = wb[1] = wb[2] = wb[3] = wb_mean;
  if(d->fix_anscombe_and_nlmeans_norm)
  {
    if(wb_mean != 0.0f && d->wb_adaptive_anscombe)
      for(int i = 0; i < 3; i++) wb[i] = d->wb_coeffs[i];
    else if(wb_mean == 0.0f)
      for(int i = 0; i < 4; i++) wb[i] = 1.0f;
  }
  else
    for(int i = 0; i < 4; i++) wb[i] = weights[i] * piece->processed_maximum[i];

  /* adaptive p, :1339-1343: binary64 expression stored to binary32 */
  for(int i = 0; i < 3; i++)
  {
    const double v = (double)d->shadows + 0.1 * (double)logf(in_scale / wb[i]);
    s->p[i] = (float)(v > 0.0 ? v : 0.0);
  }
  s->p[3] = 0.0f;
  const float compensate_p = P_FULCRUM / powf(P_FULCRUM, d->shadows);

  /* set_up_conversion_matrices(), :1163-1221 */
  float toY[3][4] = { { 1.0f / 3.0f, 1.0f / 3.0f, 1.0f / 3.0f, 0.0f }, { 0.5f, 0.0f, -0.5f, 0.0f }, { 0.25f, -0.5f, 0.25f, 0.0f } };
  float toRGB[3][4] = { { 0 } };
  float sum_invwb = 1.0f / wb[0] + 1.0f / wb[1] + 1.0f / wb[2];
  sum_invwb *= sqrtf(3);
  toY[0][0] = sum_invwb / wb[0];
  toY[0][1] = sum_invwb / wb[1];
  toY[0][2] = sum_invwb / wb[2];
  toY[0][3] = 0.0f;
  const float sdU = sqrtf(0.5f * 0.5f * wb[0] * wb[0] + 0.5f * 0.5f * wb[2] * wb[2]);
  const float sdV = sqrtf(0.25f * 0.25f * wb[0] * wb[0] + 0.5f * 0.5f * wb[1] * wb[1] + 0.25f * 0.25f * wb[2] * wb[2]);
  for(int c = 0; c < 3; c++)
  {
    toY[1][c] /= sdU;
    toY[2][c] /= sdV;
  }
  toY[1][3] = toY[2][3] = 0.0f;
  if(!invert3(toY, toRGB))
  {
    const float sdY = sqrtf(1.0f / 9.0f * (wb[0] * wb[0] + wb[1] * wb[1] + wb[2] * wb[2]));
    toY[0][0] = toY[0][1] = toY[0][2] = 1.0f / (3.0f * sdY);
    toY[0][3] = 0.0f;
    invert3(toY, toRGB);
  }
  const float compensate_strength = (d->wavelet_color_mode == DT_HIP_DENOISEPROFILE_RGB) ? 1.0f : 2.5f;
  const float gain = nlm ? d->strength * in_scale : d->strength * compensate_strength * in_scale;
  for(int k = 0; k < 3; k++)
    for(int c = 0; c < 4; c++)
    {
      toY[k][c] /= gain;
      toRGB[k][c] *= gain;
    }
  for(int i = 0; i < 4; i++) wb[i] *= gain;
  memcpy(s->toY, toY, sizeof(toY));
  memcpy(s->toRGB, toRGB, sizeof(toRGB));
  for(int i = 0; i < 3; i++)
  {
    s->aa[i] = d->a[1] * wb[i];
    s->bb[i] = d->b[1] * wb[i];
  }
  s->aa[3] = nlm ? d->a[1] * wb[3] : 0.0f;
  s->bb[3] = nlm ? d->b[1] * wb[3] : 0.0f;
  s->a_v2 = d->a[1] * compensate_p;
  s->b_v2 = d->b[1];
  s->bias = (float)((double)d->bias - 0.5 * (double)logf(in_scale));
  s->vst = !d->use_new_vst ? 0 : ((nlm || d->wavelet_color_mode == DT_HIP_DENOISEPROFILE_RGB) ? 1 : 2);
  return 0;
}

/* ---- forward transforms ---- */
static void precondition(const dn_setup_t *s, const float *in, float *buf, const size_t npix)
{
  if(s->vst == 0)
  {
    float k[4];
    for(int c = 0; c < 3; c++) k[c] = (s->bb[c] / s->aa[c]) * (s->bb[c] / s->aa[c]) + 3.f / 8.f;
    k[3] = 0.0f;
#pragma omp parallel for
    for(size_t j = 0; j < npix; j++)
      for(int c = 0; c < 4; c++)
      {
        const float dd = fmaxf(0.0f, in[4 * j + c] / s->aa[c] + k[c]);
        buf[4 * j + c] = 2.0f * sqrtf(dd);
      }
    return;
  }
  float expon[4], scale[4];
  const float sa = sqrtf(s->a_v2);
  for(int c = 0; c < 3; c++) expon[c] = -s->p[c] / 2 + 1;
  expon[3] = 1.0f;
  if(s->vst == 1)
  {
    for(int c = 0; c < 3; c++) scale[c] = (-s->p[c] + 2) * sa; /* "denom" */
    scale[3] = 1.0f;
#pragma omp parallel for
    for(size_t j = 0; j < npix; j++)
      for(int c = 0; c < 4; c++)
        buf[4 * j + c] = 2.0f * powf(max_first(in[4 * j + c] / s->wb[c] + s->b_v2, 0.0f), expon[c]) / scale[c];
    return;
  }
  for(int c = 0; c < 3; c++) scale[c] = 2.0f / ((-s->p[c] + 2) * sa);
  scale[3] = 1.0f;
#pragma omp parallel for
  for(size_t j = 0; j < npix; j++)
  {
    float t[4];
    for(int c = 0; c < 4; c++) t[c] = powf(max_first(in[4 * j + c] + s->b_v2, 0.0f), expon[c]) * scale[c];
    for(int c = 0; c < 3; c++)
    {
      float sum = 0.0f;
      for(int k = 0; k < 4; k++) sum += s->toY[c][k] * t[k];
      buf[4 * j + c] = sum;
    }
    buf[4 * j + 3] = 0;
  }
}

static void backtransform(const dn_setup_t *s, float *buf, const size_t npix)
{
  if(s->vst == 0)
  {
    float k[4];
    for(int c = 0; c < 3; c++) k[c] = (s->bb[c] / s->aa[c]) * (s->bb[c] / s->aa[c]) + 1.f / 8.f;
    k[3] = 0.0f;
    const float sqrt_3_2 = sqrtf(3.0f / 2.0f);
#pragma omp parallel for
    for(size_t j = 0; j < npix; j++)
      for(int c = 0; c < 4; c++)
      {
        const float x = buf[4 * j + c], x2 = x * x;
        buf[4 * j + c] = (x < 0.5f) ? 0.0f
                                    : s->aa[c] * (1.f / 4.f * x2 + 1.f / 4.f * sqrt_3_2 / x - 11.f / 8.f / x2
                                                  + 5.f / 8.f * sqrt_3_2 / (x * x2) - k[c]);
      }
    return;
  }
  float expon[4], scale[4];
  const float sa = sqrtf(s->a_v2);
  for(int c = 0; c < 3; c++) expon[c] = 1.0f / (1.0f - s->p[c] / 2.0f);
  expon[3] = 1.0f;
  if(s->vst == 1)
  {
    for(int c = 0; c < 3; c++) scale[c] = 4.0f / (sa * (2.0f - s->p[c])); /* "denom" */
    scale[3] = 1.0f;
#pragma omp parallel for
    for(size_t j = 0; j < npix; j++)
      for(int c = 0; c < 4; c++)
      {
        const float x = max_first(buf[4 * j + c], 0.0f);
        const float delta = x * x + s->bias;
        const float z1 = (x + sqrtf(max_first(delta, 0.0f))) / scale[c];
        buf[4 * j + c] = s->wb[c] * (powf(z1, expon[c]) - s->b_v2);
      }
    return;
  }
  float bias_wb[4];
  for(int c = 0; c < 3; c++)
  {
    bias_wb[c] = s->bias * s->wb[c];
    scale[c] = (sa * (2.0f - s->p[c])) / 4.0f;
  }
  bias_wb[3] = 0.0f;
  scale[3] = 1.0f;
#pragma omp parallel for
  for(size_t j = 0; j < npix; j++)
  {
    float rgb[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    for(int k = 0; k < 3; k++)
      for(int c = 0; c < 4; c++) rgb[k] += s->toRGB[k][c] * buf[4 * j + c];
    for(int c = 0; c < 4; c++)
    {
      const float x = max_first(rgb[c], 0.0f);
      const float delta = x * x + bias_wb[c];
      const float z1 = (x + sqrtf(max_first(delta, 0.0f))) * scale[c];
      buf[4 * j + c] = powf(z1, expon[c]) - s->b_v2;
    }
  }
}

/* Summation order of the squared details: 0 = canonical (above); 1 = what the reference computes
 * on ONE OpenMP thread -- binary32 accumulation over rows in dwt_interleave_rows() order
 * (src/pixel/dwt.h:93-104), pixels left to right.  Mode 1 exists so that tests can pin every other
 * operation of this file bit-for-bit against oracle/_ref run single-threaded. */
static int g_sum_order = 0;
void oracle_denoiseprofile_sum_order(const int mode) { g_sum_order = mode; }

static int interleaved_row(const int rowid, const int height, const int stride)
{
  if(height <= stride) return rowid;
  const int per_pass = (height + stride - 1) / stride;
  const int long_passes = height % stride;
  if(long_passes == 0 || rowid < long_passes * per_pass) return (rowid / per_pass) + stride * (rowid % per_pass);
  const int r2 = rowid - long_passes * per_pass;
  return long_passes + (r2 / (per_pass - 1)) + stride * (r2 % (per_pass - 1));
}

/* ---- one band: edge-aware a-trous step + canonical sum of squared details ---- */
static void decompose(float *coarse, const float *in, float *detail, float sum_sq[4], const int scale,
                      const float inv_sigma2, const int w, const int h)
{
  static const float filter[5] = { 1.0f / 16.0f, 4.0f / 16.0f, 6.0f / 16.0f, 4.0f / 16.0f, 1.0f / 16.0f };
  const int mult = 1 << scale;
  const int nseg = (w + 255) / 256;
  double *partial = (double *)calloc((size_t)h * nseg * 4, sizeof(double));
#pragma omp parallel for
  for(int j = 0; j < h; j++)
  {
    double *sq = (double *)malloc(sizeof(double) * 4 * 256);
    for(int seg = 0; seg < nseg; seg++)
    {
      memset(sq, 0, sizeof(double) * 4 * 256);
      for(int l = 0; l < 256 && seg * 256 + l < w; l++)
      {
        const int i = seg * 256 + l;
        const float *px = in + 4 * ((size_t)j * w + i);
        float sum[4] = { 0.f, 0.f, 0.f, 0.f }, wgt[4] = { 0.f, 0.f, 0.f, 0.f };
        for(int jj = 0; jj < 5; jj++)
        {
          const int y = clampi(j + mult * (jj - 2), 0, h - 1);
          for(int ii = 0; ii < 5; ii++)
          {
            const int x = clampi(i + mult * (ii - 2), 0, w - 1);
            const float *px2 = in + 4 * ((size_t)y * w + x);
            /* dn_weight(), eaw.c:181-195 */
            float sqr[3];
            for(int c = 0; c < 3; c++)
            {
              const float diff = px[c] - px2[c];
              sqr[c] = diff * diff;
            }
            const float dot = (sqr[0] + sqr[1] + sqr[2]) * inv_sigma2;
            const float arg = dot * 0.02f - 9.0f;
            const float wp = mexp2(0 > arg ? 0.0f : arg);
            const float wt = (filter[ii] * filter[jj]) * wp;
            for(int c = 0; c < 4; c++)
            {
              const float pd = wt * px2[c];
              wgt[c] += wt;
              sum[c] += pd;
            }
          }
        }
        const size_t o = 4 * ((size_t)j * w + i);
        for(int c = 0; c < 4; c++)
        {
          sum[c] /= wgt[c];
          coarse[o + c] = sum[c];
          const float det = px[c] - sum[c];
          detail[o + c] = det;
          sq[4 * l + c] = (double)(det * det);
        }
      }
      /* step 1 of the canonical order */
      for(int c = 0; c < 4; c++)
      {
        double run[4];
        for(int q = 0; q < 4; q++)
        {
          double v[64];
          for(int m = 0; m < 64; m++) v[m] = sq[4 * (64 * q + m) + c];
          for(int off = 32; off >= 1; off >>= 1)
            for(int m = 0; m < off; m++) v[m] += v[m + off];
          run[q] = v[0];
        }
        partial[4 * ((size_t)j * nseg + seg) + c] = ((run[0] + run[1]) + run[2]) + run[3];
      }
    }
    free(sq);
  }
  if(g_sum_order == 1)
  {
    float acc[4] = { 0.f, 0.f, 0.f, 0.f };
    for(int rowid = 0; rowid < h; rowid++)
    {
      const int j = interleaved_row(rowid, h, mult);
      for(int i = 0; i < w; i++)
        for(int c = 0; c < 4; c++)
        {
          const float dv = detail[4 * ((size_t)j * w + i) + c];
          acc[c] += dv * dv;
        }
    }
    for(int c = 0; c < 4; c++) sum_sq[c] = acc[c];
    free(partial);
    return;
  }
  /* step 2 */
  const size_t np = (size_t)h * nseg;
  for(int c = 0; c < 4; c++)
  {
    double acc[1024];
    for(int t = 0; t < 1024; t++)
    {
      double a = 0.0;
      for(size_t s = t; s < np; s += 1024) a += partial[4 * s + c];
      acc[t] = a;
    }
    for(int off = 512; off >= 1; off >>= 1)
      for(int t = 0; t < off; t++) acc[t] += acc[t + off];
    sum_sq[c] = (float)acc[0];
  }
  free(partial);
}

/* variance_stabilizing_xform(), denoiseprofile.c:1223-1287 */
static void band_threshold(float thrs[4], const int scale, const int max_scale, const size_t npixels,
                           const float sum_y2[4], const dt_hip_denoiseprofile_data_t *d)
{
  const float varf = sqrtf(2.0f + 2.0f * 4.0f * 4.0f + 6.0f * 6.0f) / 16.0f;
  const float sigma_band = powf(varf, scale) * 1.0f;
  const float sb2 = sigma_band * sigma_band;
  const float n1 = (float)npixels - 1.0f;
  float std_x[4], adjt[4] = { 8.0f, 8.0f, 8.0f, 0.0f };
  for(int c = 0; c < 3; c++) std_x[c] = sqrtf(max_first(1e-6f, sum_y2[c] / n1 - sb2));
  std_x[3] = 1.0f;
  const int band = BANDS - (scale + (BANDS - max_scale) + 1);
  if(d->wavelet_color_mode == DT_HIP_DENOISEPROFILE_RGB)
  {
    float f = d->force[0][band];
    f *= f;
    f *= 4;
    for(int c = 0; c < 4; c++) adjt[c] *= f;
    for(int c = 0; c < 3; c++)
    {
      f = d->force[1 + c][band];
      f *= f;
      f *= 4;
      adjt[c] *= f;
    }
  }
  else
  {
    float f = d->force[4][band];
    f *= f;
    f *= 4;
    adjt[0] *= f;
    f = d->force[5][band];
    f *= f;
    f *= 4;
    adjt[1] *= f;
    adjt[2] *= f;
  }
  for(int c = 0; c < 4; c++) thrs[c] = adjt[c] * sb2 / std_x[c];
}

int oracle_denoiseprofile_bands(const dt_hip_piece_t *piece, const dt_hip_denoiseprofile_data_t *d)
{
  dn_setup_t s;
  dn_setup(piece, d, &s, 0);
  return s.max_scale;
}

/* process_nlmeans_cpu(), denoiseprofile.c:1599-1648, with nlmeans_norm() :1457-1472 and
 * nlmeans_scattering() :1476-1500 (export pipe: no preview output, not a thumbnail) */
static int denoise_nlmeans(const dt_hip_piece_t *piece, const dt_hip_denoiseprofile_data_t *d, const float *in, float *out)
{
  const int w = piece->roi_in.width, h = piece->roi_in.height;
  const size_t npix = (size_t)w * h;
  const float scale = fminf(fminf((float)piece->roi_in.scale, 2.0f), 1.0f);
  const int P = (int)ceilf(d->radius * scale);
  int K = (int)d->nbhood;
  float scattering = d->scattering;
  {
    const int maxk = (int)((K * K * K + 7.0 * K * sqrt(K)) * scattering / 6.0 + K);
    const float kf = (float)K * scale;
    const int k4 = K < 4 ? K : 4;
    K = (int)((float)k4 > kf ? (float)k4 : kf);
    scattering = (float)((maxk - K) * 6.0 / (K * K * K + 7.0 * K * sqrt(K)));
  }
  float norm = .045f / ((2 * P + 1) * (2 * P + 1));
  if(!d->fix_anscombe_and_nlmeans_norm) norm = .015f / (2 * P + 1);
  dn_setup_t s;
  dn_setup(piece, d, &s, 1);
  float *pre = (float *)malloc(sizeof(float) * 4 * npix);
  precondition(&s, in, pre, npix);
  oracle_nlm_params_t p;
  memset(&p, 0, sizeof(p));
  p.scattering = scattering;
  p.scale = scale;
  p.luma = 1.0f;
  p.chroma = 1.0f;
  p.center_weight = d->central_pixel_weight * scale;
  p.sharpness = norm;
  p.patch_radius = P;
  p.search_radius = K;
  p.norm[0] = p.norm[1] = p.norm[2] = p.norm[3] = 1.0f;
  oracle_nlmeans_core(pre, out, w, h, &p);
  free(pre);
  backtransform(&s, out, npix);
  return 0;
}

int oracle_denoiseprofile(const dt_hip_piece_t *piece, const dt_hip_denoiseprofile_data_t *d, const void *in_, void *out_)
{
  if(DT_HIP_DENOISEPROFILE_IS_NLMEANS(d->mode)) return denoise_nlmeans(piece, d, (const float *)in_, (float *)out_);
  if(!DT_HIP_DENOISEPROFILE_IS_WAVELETS(d->mode)) return 1;
  const int w = piece->roi_in.width, h = piece->roi_in.height;
  const size_t npix = (size_t)w * h;
  const float *in = (const float *)in_;
  float *out = (float *)out_;
  dn_setup_t s;
  dn_setup(piece, d, &s, 0);
  const int max_mult = 1 << (s.max_scale - 1);
  if(w < 2 * max_mult || h < 2 * max_mult)
  {
    memcpy(out, in, sizeof(float) * 4 * npix);
    return 0;
  }
  if(w < 4 * max_mult) return 1; /* the reference reads before the row start here (eaw.c:308-323): undefined */
  float *precond = (float *)malloc(sizeof(float) * 4 * npix), *tmp = (float *)malloc(sizeof(float) * 4 * npix);
  float *det = (float *)malloc(sizeof(float) * 4 * npix);
  precondition(&s, in, precond, npix);
  float *b1 = precond, *b2 = tmp;
  memset(out, 0, sizeof(float) * 4 * npix);
  for(int scale = 0; scale < s.max_scale; scale++)
  {
    const float varf = sqrtf(2.0f + 2.0f * 4.0f * 4.0f + 6.0f * 6.0f) / 16.0f;
    const float sigma_band = powf(varf, scale) * 1.0f;
    float sum_y2[4], thrs[4];
    decompose(b2, b1, det, sum_y2, scale, 1.0f / (sigma_band * sigma_band), w, h);
    band_threshold(thrs, scale, s.max_scale, npix, sum_y2, d);
    /* eaw_synthesize() with boost 1, eaw.c:157-175 */
#pragma omp parallel for
    for(size_t k = 0; k < npix; k++)
      for(int c = 0; c < 4; c++)
      {
        const float v = det[4 * k + c];
        const float amount = max_first(v - thrs[c], 0.0f) + min_first(v + thrs[c], 0.0f);
        out[4 * k + c] = out[4 * k + c] + (1.0f * amount);
      }
    float *t = b2;
    b2 = b1;
    b1 = t;
  }
  for(size_t k = 0; k < 4 * npix; k++) out[k] += b1[k];
  backtransform(&s, out, npix);
  free(precond);
  free(tmp);
  free(det);
  return 0;
}
