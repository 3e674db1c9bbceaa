/* oracle/src/locallaplacian.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of local contrast in local-laplacian mode (the module's default mode):
 *   process()                   src/iop/bilat.c:352-357
 *   local_laplacian_internal()  src/pixel/locallaplacian.c:354-563, regular mode (no preview boundary)
 *   ll_pad_input() :204-279 (padding by replication), gauss_reduce() :173-200, gauss_expand() :160-171
 *   with ll_expand_gaussian() :80-118 and the boundary fills :121-144, apply_curve()/curve_scalar()
 *   :295-351, ll_laplacian() :282-293
 *
 * The luminance plane (L / 100) is padded by 2^last_level on every side, six copies of it are pushed
 * through the tone curve centred on six grey levels, each becomes a Gaussian pyramid, and the output
 * pyramid is assembled coarse to fine from the Laplacian coefficients of the two curves that bracket
 * the local grey level.  Where the reference's expressions carry double literals the arithmetic is
 * binary64, as there.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define MAX_LEVELS 30
#define NUM_GAMMA 6

static inline int dl(int size, const int level)
{
  for(int l = 0; l < level; l++) size = (size - 1) / 2 + 1;
  return size;
}
static inline float clampf(const float v, const float lo, const float hi) { return v > lo ? (v < hi ? v : hi) : lo; }
static inline int clampi(const int v, const int lo, const int hi) { return v > lo ? (v < hi ? v : hi) : lo; }

/* ll_expand_gaussian(), locallaplacian.c:80-118 */
static inline float expand_at(const float *c, const int i, const int j, const int wd)
{
  const int cw = (wd - 1) / 2 + 1;
  const int ind = (j / 2) * cw + i / 2;
  switch((i & 1) + 2 * (j & 1))
  {
    case 0:
      return (float)(4. / 256.
                     * (double)(6.0f * (c[ind - cw] + c[ind - 1] + 6.0f * c[ind] + c[ind + 1] + c[ind + cw]) + c[ind - cw - 1]
                                + c[ind - cw + 1] + c[ind + cw - 1] + c[ind + cw + 1]));
    case 1:
      return (float)(4. / 256.
                     * (24.0 * (double)(c[ind] + c[ind + 1])
                        + 4.0 * (double)(c[ind - cw] + c[ind - cw + 1] + c[ind + cw] + c[ind + cw + 1])));
    case 2:
      return (float)(4. / 256.
                     * (24.0 * (double)(c[ind] + c[ind + cw])
                        + 4.0 * (double)(c[ind - 1] + c[ind + 1] + c[ind + cw - 1] + c[ind + cw + 1])));
    default: return .25f * (c[ind] + c[ind + 1] + c[ind + cw] + c[ind + cw + 1]);
  }
}

/* ll_fill_boundary1(), :121-131 */
static void fill_boundary1(float *b, const int wd, const int ht)
{
  for(int j = 1; j < ht - 1; j++) b[j * wd] = b[j * wd + 1];
  for(int j = 1; j < ht - 1; j++) b[j * wd + wd - 1] = b[j * wd + wd - 2];
  memcpy(b, b + wd, sizeof(float) * wd);
  memcpy(b + (size_t)wd * (ht - 1), b + (size_t)wd * (ht - 2), sizeof(float) * wd);
}

/* ll_fill_boundary2(), :133-144 */
static void fill_boundary2(float *b, const int wd, const int ht)
{
  for(int j = 1; j < ht - 1; j++) b[j * wd] = b[j * wd + 1];
  if(wd & 1)
    for(int j = 1; j < ht - 1; j++) b[j * wd + wd - 1] = b[j * wd + wd - 2];
  else
    for(int j = 1; j < ht - 1; j++) b[j * wd + wd - 1] = b[j * wd + wd - 2] = b[j * wd + wd - 3];
  memcpy(b, b + wd, sizeof(float) * wd);
  if(!(ht & 1)) memcpy(b + (size_t)wd * (ht - 2), b + (size_t)wd * (ht - 3), sizeof(float) * wd);
  memcpy(b + (size_t)wd * (ht - 1), b + (size_t)wd * (ht - 2), sizeof(float) * wd);
}

/* gauss_reduce(), :173-200 */
static void reduce(const float *in, float *coarse, const int wd, const int ht)
{
  const int cw = (wd - 1) / 2 + 1, ch = (ht - 1) / 2 + 1;
  const float w[5] = { 1.f / 16.f, 4.f / 16.f, 6.f / 16.f, 4.f / 16.f, 1.f / 16.f };
  memset(coarse, 0, sizeof(float) * (size_t)cw * ch);
#pragma omp parallel for
  for(int j = 1; j < ch - 1; j++)
    for(int i = 1; i < cw - 1; i++)
    {
      float acc = 0.0f;
      for(int jj = -2; jj <= 2; jj++)
        for(int ii = -2; ii <= 2; ii++) acc += in[(size_t)(2 * j + jj) * wd + 2 * i + ii] * w[ii + 2] * w[jj + 2];
      coarse[(size_t)j * cw + i] = acc;
    }
  fill_boundary1(coarse, cw, ch);
}

/* gauss_expand(), :160-171 */
static void expand(const float *coarse, float *fine, const int wd, const int ht)
{
#pragma omp parallel for
  for(int j = 1; j < ((ht - 1) & ~1); j++)
    for(int i = 1; i < ((wd - 1) & ~1); i++) fine[(size_t)j * wd + i] = expand_at(coarse, i, j, wd);
  fill_boundary2(fine, wd, ht);
}

/* curve_scalar(), :295-325 */
static inline float curve(const float x, const float g, const float sigma, const float shadows, const float highlights,
                          const float clarity)
{
  const float c = x - g;
  float val;
  if(c > 2 * sigma)
    val = g + sigma + shadows * (c - sigma);
  else if(c < -2 * sigma)
    val = g - sigma + highlights * (c + sigma);
  else if(c > 0.0f)
  {
    const float t = clampf(c / (2.0f * sigma), 0.0f, 1.0f);
    const float t2 = t * t, mt = 1.0f - t;
    val = g + sigma * 2.0f * mt * t + t2 * (sigma + sigma * shadows);
  }
  else
  {
    const float t = clampf(-c / (2.0f * sigma), 0.0f, 1.0f);
    const float t2 = t * t, mt = 1.0f - t;
    val = g - sigma * 2.0f * mt * t + t2 * (-sigma - sigma * highlights);
  }
  val += clarity * c * expf((float)((double)(-c * c) / (2.0 * (double)sigma * (double)sigma / (double)3.0f)));
  return val;
}

/* pad_by_replication(), :146-158 */
static void pad_rows(float *buf, const int w, const int h, const int padding)
{
  for(int j = 0; j < padding; j++)
  {
    memcpy(buf + (size_t)w * j, buf + (size_t)padding * w, sizeof(float) * w);
    memcpy(buf + (size_t)w * (h - padding + j), buf + (size_t)w * (h - padding - 1), sizeof(float) * w);
  }
}

int oracle_local_laplacian(const float *input, float *out, const int wd, const int ht, const float sigma,
                           const float shadows, const float highlights, const float clarity)
{
  if(wd <= 1 || ht <= 1) return 0;
  const int m = wd < ht ? wd : ht;
  const int num_levels = 31 - __builtin_clz(m) < MAX_LEVELS ? 31 - __builtin_clz(m) : MAX_LEVELS;
  const int last_level = num_levels - 1;
  const int max_supp = 1 << last_level;
  const int w = 2 * max_supp + wd, h = 2 * max_supp + ht;
  float *padded[MAX_LEVELS] = { 0 }, *output[MAX_LEVELS] = { 0 }, *buf[NUM_GAMMA][MAX_LEVELS] = { { 0 } };
  for(int l = 0; l <= last_level; l++)
  {
    const size_t n = (size_t)dl(w, l) * dl(h, l);
    padded[l] = (float *)calloc(n, sizeof(float));
    output[l] = (float *)calloc(n, sizeof(float));
    for(int k = 0; k < NUM_GAMMA; k++) buf[k][l] = (float *)calloc(n, sizeof(float));
  }
  /* ll_pad_input(), replication branch, :262-273 */
  for(int j = 0; j < ht; j++)
  {
    float *row = padded[0] + (size_t)(j + max_supp) * w;
    for(int i = 0; i < max_supp; i++) row[i] = input[4 * (size_t)wd * j] * 0.01f;
    for(int i = 0; i < wd; i++) row[i + max_supp] = input[4 * ((size_t)wd * j + i)] * 0.01f;
    for(int i = wd + max_supp; i < w; i++) row[i] = input[4 * ((size_t)j * wd + wd - 1)] * 0.01f;
  }
  pad_rows(padded[0], w, h, max_supp);
  /* Gaussian pyramid of the padded input; its coarsest level seeds the output, :405-407 */
  for(int l = 1; l < last_level; l++) reduce(padded[l - 1], padded[l], dl(w, l - 1), dl(h, l - 1));
  reduce(padded[last_level - 1], output[last_level], dl(w, last_level - 1), dl(h, last_level - 1));
  float gamma[NUM_GAMMA];
  for(int k = 0; k < NUM_GAMMA; k++) gamma[k] = (k + .5f) / (float)NUM_GAMMA;
  for(int k = 0; k < NUM_GAMMA; k++)
  {
    /* apply_curve(), :328-351 */
    float *o = buf[k][0];
    const float *in = padded[0];
#pragma omp parallel for
    for(int j = max_supp; j < h - max_supp; j++)
    {
      for(int i = max_supp; i < w - max_supp; i++)
        o[(size_t)j * w + i] = curve(in[(size_t)j * w + i], gamma[k], sigma, shadows, highlights, clarity);
      float *row = o + (size_t)j * w;
      for(int i = 0; i < max_supp; i++) row[i] = row[max_supp];
      for(int i = w - max_supp; i < w; i++) row[i] = row[w - max_supp - 1];
    }
    pad_rows(o, w, h, max_supp);
    for(int l = 1; l <= last_level; l++) reduce(buf[k][l - 1], buf[k][l], dl(w, l - 1), dl(h, l - 1));
  }
  /* assemble the output pyramid coarse to fine, :499-523 */
  for(int l = last_level - 1; l >= 0; l--)
  {
    const int pw = dl(w, l), ph = dl(h, l);
    expand(output[l + 1], output[l], pw, ph);
#pragma omp parallel for
    for(int j = 0; j < ph; j++)
      for(int i = 0; i < pw; i++)
      {
        const float v = padded[l][(size_t)j * pw + i];
        int hi = 1;
        for(; hi < NUM_GAMMA - 1 && gamma[hi] <= v; hi++)
          ;
        const int lo = hi - 1;
        const float a = clampf((v - gamma[lo]) / (gamma[hi] - gamma[lo]), 0.0f, 1.0f);
        const int ci = clampi(i, 1, ((pw - 1) & ~1) - 1), cj = clampi(j, 1, ((ph - 1) & ~1) - 1);
        const float l0 = buf[lo][l][(size_t)j * pw + i] - expand_at(buf[lo][l + 1], ci, cj, pw);
        const float l1 = buf[hi][l][(size_t)j * pw + i] - expand_at(buf[hi][l + 1], ci, cj, pw);
        output[l][(size_t)j * pw + i] += l0 * (1.0f - a) + l1 * a;
      }
  }
  for(int j = 0; j < ht; j++)
    for(int i = 0; i < wd; i++)
    {
      out[4 * ((size_t)j * wd + i) + 0] = 100.0f * output[0][(size_t)(j + max_supp) * w + max_supp + i];
      out[4 * ((size_t)j * wd + i) + 1] = input[4 * ((size_t)j * wd + i) + 1];
      out[4 * ((size_t)j * wd + i) + 2] = input[4 * ((size_t)j * wd + i) + 2];
    }
  for(int l = 0; l <= last_level; l++)
  {
    free(padded[l]);
    free(output[l]);
    for(int k = 0; k < NUM_GAMMA; k++) free(buf[k][l]);
  }
  return 0;
}
