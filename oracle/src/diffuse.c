/* oracle/src/diffuse.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of the diffuse-or-sharpen module:
 *   process()                src/iop/diffuse.c:1155-1258   iteration ping-pong, scale count
 *   wavelets_process()       src/iop/diffuse.c:978-1106    a-trous analysis, coarse-to-fine PDE synthesis
 *   decompose_2D_Bspline()   src/pixel/bspline.h:351-377   5-tap B-spline, dilation 2^s, clamped borders,
 *                            with _bspline_vertical_pass() :120-136, _bspline_horizontal() :139-154,
 *                            sparse_scalar_product() :86-118 (negatives clipped after EACH pass)
 *   heat_PDE_diffusion()     src/iop/diffuse.c:760-968     3x3 stencil at dilation 2^s on HF and LF
 *   scale bookkeeping        src/pixel/bspline.h:55-80     equivalent_sigma_at_step(), num_steps_to_reach_equivalent_sigma()
 *   dt_fast_expf()           src/math/math.h:254-267       integer-trick exp
 *   dt_simd_max_zero()       src/system/simd.h:108-114     non-finite -> 0, else MAX(v, 0)
 *
 * The luminance-masked branch (threshold > 0: build_mask / inpaint_mask, diffuse.c:1106-1152) seeds the
 * masked area with Box-Muller noise from src/iop/noise_generator.h:36-93 (splitmix32 seeds from the pixel's
 * position in the buffer, xoshiro128+); pixels outside the mask pass through every PDE step as HF + LF
 * (diffuse.c:802, 927-937).
 *
 * Arithmetic: binary32, one rounding per operation, in the reference's order: the vector code there
 * is GCC vector-extension arithmetic, lane-wise identical to the scalar form below.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define BSPLINE_SIGMA 1.0553651328015339f /* bspline.h:39 */
#define MAX_SCALES 10                     /* diffuse.c:75 */
#define PDE_KAPPA 0.25f                   /* diffuse.c:625 */

static inline float sq(const float x) { return x * x; }
/* the MAX() macro of the reference: (a > b ? a : b) -- NaN in b passes through */
static inline float max_first(const float a, const float b) { return a > b ? a : b; }
static inline float max_zero(const float v) { return isfinite(v) ? max_first(v, 0.0f) : 0.0f; }
static inline int clampi(const int v, const int lo, const int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* bspline.h:55-66 */
static float sigma_at_step(const float sigma, const unsigned s)
{
  float acc = sigma;
  for(unsigned k = 1; k <= s; k++) acc = sqrtf(sq(acc) + sq(exp2f((float)k) * sigma));
  return acc;
}

/* bspline.h:68-80 */
static unsigned steps_to_sigma(const float sigma_filter, const float sigma_final)
{
  unsigned s = 0;
  float radius = sigma_filter;
  while(radius < sigma_final)
  {
    ++s;
    radius = sqrtf(sq(radius) + sq((float)(1 << s) * sigma_filter));
  }
  return s + 1;
}

/* math.h:254-267; the int conversion of an out-of-range float is what cvttss2si returns (INT_MIN),
 * which the k0 > 0 test maps to 0 */
static inline float fast_expf(const float x)
{
  const float t = 1065353216.0f + x * 11401300.0f; /* (float)0x3f800000 + x * (float)(0x402DF854 - 0x3f800000) */
  int k;
  if(!(t > -2147483648.0f && t < 2147483648.0f)) k = 0;
  else k = (int)t;
  if(k < 0) k = 0;
  union { int i; float f; } u;
  u.i = k;
  return u.f;
}

static inline float tap5(const float a, const float b, const float c, const float d, const float e)
{
  /* sparse_scalar_product(): left-to-right sum of the five weighted taps, then MAX(0, .) */
  const float s = 0.0625f * a + 0.25f * b + 0.375f * c + 0.25f * d + 0.0625f * e;
  return max_first(0.0f, s);
}

static void decompose(const float *in, float *hf, float *lf, const int w, const int h, const int mult)
{
#pragma omp parallel
  {
    float *row = (float *)malloc(sizeof(float) * 4 * (size_t)w);
#pragma omp for
    for(int i = 0; i < h; i++)
    {
      const float *r0 = in + 4 * (size_t)w * clampi(i - 2 * mult, 0, h - 1);
      const float *r1 = in + 4 * (size_t)w * clampi(i - mult, 0, h - 1);
      const float *r2 = in + 4 * (size_t)w * i;
      const float *r3 = in + 4 * (size_t)w * clampi(i + mult, 0, h - 1);
      const float *r4 = in + 4 * (size_t)w * clampi(i + 2 * mult, 0, h - 1);
      for(int k = 0; k < 4 * w; k++) row[k] = tap5(r0[k], r1[k], r2[k], r3[k], r4[k]);
      for(int j = 0; j < w; j++)
      {
        const int j0 = 4 * clampi(j - 2 * mult, 0, w - 1), j1 = 4 * clampi(j - mult, 0, w - 1);
        const int j3 = 4 * clampi(j + mult, 0, w - 1), j4 = 4 * clampi(j + 2 * mult, 0, w - 1);
        const size_t o = 4 * ((size_t)i * w + j);
        for(int c = 0; c < 4; c++)
        {
          const float low = tap5(row[j0 + c], row[j1 + c], row[4 * j + c], row[j3 + c], row[j4 + c]);
          lf[o + c] = low;
          hf[o + c] = in[o + c] - low;
        }
      }
    }
    free(row);
  }
}

typedef struct
{
  float anisotropy[4]; /* squared user values, compute_anisotropy_factor() diffuse.c:970-976 */
  int kind[4];         /* 0 isotrope, 1 isophote, 2 gradient: check_isotropy_mode() :151-161 */
  float variance_threshold, regularization, abcd[4], strength;
} pde_t;

static void kernel9(const int kind, const float c2, const float cs, const float cos2, const float sin2, float k[9])
{
  if(kind == 0)
  {
    /* isotrope_laplacian(), diffuse.c:705-723 */
    k[0] = k[2] = k[6] = k[8] = 0.25f;
    k[1] = k[3] = k[5] = k[7] = 0.5f;
    k[4] = -3.0f;
    return;
  }
  float a00, a11, a01;
  if(kind == 1)
  {
    /* rotation_matrix_isophote(), :646-659 */
    a00 = cos2 + c2 * sin2;
    a11 = c2 * cos2 + sin2;
    a01 = (c2 - 1.0f) * cs;
  }
  else
  {
    /* rotation_matrix_gradient(), :661-674 */
    a00 = c2 * cos2 + sin2;
    a11 = cos2 + c2 * sin2;
    a01 = (1.0f - c2) * cs;
  }
  /* build_matrix(), :677-703 */
  const float b11 = a01 * 0.5f, b13 = -b11, b22 = -2.0f * (a00 + a11);
  k[0] = b11; k[1] = a11; k[2] = b13;
  k[3] = a00; k[4] = b22; k[5] = a00;
  k[6] = b13; k[7] = a11; k[8] = b11;
}

/* direction of a 2-vector as {cos^2, sin^2, cos*sin} and its magnitude, diffuse.c:851-866 */
static inline float direction(float gx, float gy, float *cos2, float *sin2, float *cs)
{
  const float mag = sqrtf(gx * gx + gy * gy);
  const float nonzero = (mag != 0.0f) ? 1.0f : 0.0f;
  const float inv = 1.0f / (mag + (1.0f - nonzero));
  gx = gx * inv + (1.0f - nonzero);
  gy = gy * inv;
  *cos2 = gx * gx;
  *sin2 = gy * gy;
  *cs = gx * gy;
  return mag;
}

/* ---- noise_generator.h:36-93 ---- */
static uint32_t splitmix32(const uint64_t seed)
{
  uint64_t result = (seed ^ (seed >> 33)) * 0x62a9d9ed799705f5ul;
  result = (result ^ (result >> 28)) * 0xcb24d0a5c88c35b3ul;
  return (uint32_t)(result >> 32);
}

static float xoshiro128plus(uint32_t state[4])
{
  const unsigned int result = state[0] + state[3];
  const unsigned int t = state[1] << 9;
  state[2] ^= state[0];
  state[3] ^= state[1];
  state[1] ^= state[2];
  state[0] ^= state[3];
  state[2] ^= t;
  state[3] = (state[3] << 11) | (state[3] >> (32 - 11));
  return (float)(result >> 8) * 0x1.0p-24f;
}

static float gaussian_noise(const float mu, const float sigma, const int flip, uint32_t state[4])
{
  const float u1 = fmaxf(xoshiro128plus(state), FLT_MIN);
  const float u2 = xoshiro128plus(state);
  const float noise = (flip) ? sqrtf(-2.0f * logf(u1)) * cosf(2.f * M_PI * u2) : sqrtf(-2.0f * logf(u1)) * sinf(2.f * M_PI * u2);
  return noise * sigma + mu;
}

/* build_mask() + inpaint_mask(), diffuse.c:1106-1152.  k is the FLOAT index of the pixel; the reference derives its
 * "row" as k / width and its "column" as k - row (sic), which is what seeds the generator */
static void build_and_inpaint(const float *in, float *inpainted, uint8_t *mask, const float threshold, const size_t width,
                              const size_t height)
{
#pragma omp parallel for
  for(size_t k = 0; k < height * width * 4; k += 4)
  {
    mask[k / 4] = (in[k] > threshold || in[k + 1] > threshold || in[k + 2] > threshold);
    if(mask[k / 4])
    {
      const uint32_t i = k / width;
      const uint32_t j = k - i;
      uint32_t state[4] = { splitmix32(j + 1), splitmix32((uint64_t)(j + 1) * (i + 3)), splitmix32(1337), splitmix32(666) };
      xoshiro128plus(state);
      xoshiro128plus(state);
      xoshiro128plus(state);
      xoshiro128plus(state);
      for(int c = 0; c < 4; c++) inpainted[k + c] = fabsf(gaussian_noise(in[k + c], in[k + c], i % 2 || j % 2, state));
    }
    else
      for(int c = 0; c < 4; c++) inpainted[k + c] = in[k + c];
  }
}

static void pde(const float *hf, const float *lf, const uint8_t *mask, float *out, const int w, const int h, const int mult,
                const pde_t *p)
{
#pragma omp parallel for
  for(int i = 0; i < h; i++)
  {
    const size_t rows[3] = { (size_t)clampi(i - mult, 0, h - 1) * w, (size_t)i * w, (size_t)clampi(i + mult, 0, h - 1) * w };
    for(int j = 0; j < w; j++)
    {
      const int cols[3] = { clampi(j - mult, 0, w - 1), j, clampi(j + mult, 0, w - 1) };
      float H[9][4], L[9][4];
      for(int ii = 0; ii < 3; ii++)
        for(int jj = 0; jj < 3; jj++)
        {
          const size_t n = 4 * (rows[ii] + cols[jj]);
          for(int c = 0; c < 4; c++)
          {
            H[3 * ii + jj][c] = hf[n + c];
            L[3 * ii + jj][c] = lf[n + c];
          }
        }
      float *o = out + 4 * ((size_t)i * w + j);
      if(mask && !mask[(size_t)i * w + j])
      {
        /* outside the mask: only copy input to output, diffuse.c:927-937 */
        for(int c = 0; c < 4; c++) o[c] = max_zero(H[4][c] + L[4][c]);
        continue;
      }
      for(int c = 0; c < 4; c++)
      {
        /* HF/LF energy over the 3x3 support, diffuse.c:823-840 */
        float energy = 0.0f;
        for(int k = 0; k < 9; k++)
        {
          const float safe = max_zero(L[k][c] - 1e-8f) + 1e-8f;
          const float ratio = H[k][c] / safe;
          energy += ratio * ratio;
        }
        energy = max_zero(p->variance_threshold + energy * p->regularization - 1e-8f) + 1e-8f;
        /* centred differences, find_gradients() :628-635 (x vertical, y horizontal) */
        float cos2g, sin2g, csg, cos2l, sin2l, csl;
        const float mg = direction((L[7][c] - L[1][c]) * 0.5f, (L[5][c] - L[3][c]) * 0.5f, &cos2g, &sin2g, &csg);
        const float ml = direction((H[7][c] - H[1][c]) * 0.5f, (H[5][c] - H[3][c]) * 0.5f, &cos2l, &sin2l, &csl);
        const float c2[4] = { fast_expf(-mg * p->anisotropy[0]), fast_expf(-ml * p->anisotropy[1]),
                              fast_expf(-mg * p->anisotropy[2]), fast_expf(-ml * p->anisotropy[3]) };
        float k1[9], k2[9], k3[9], k4[9];
        kernel9(p->kind[0], c2[0], csg, cos2g, sin2g, k1);
        kernel9(p->kind[1], c2[1], csl, cos2l, sin2l, k2);
        kernel9(p->kind[2], c2[2], csg, cos2g, sin2g, k3);
        kernel9(p->kind[3], c2[3], csl, cos2l, sin2l, k4);
        float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
        for(int k = 0; k < 9; k++)
        {
          d0 = k1[k] * L[k][c] + d0;
          d1 = k2[k] * L[k][c] + d1;
          d2 = k3[k] * H[k][c] + d2;
          d3 = k4[k] * H[k][c] + d3;
        }
        float update = d0 * p->abcd[0];
        update = d1 * p->abcd[1] + update;
        update = d2 * p->abcd[2] + update;
        update = d3 * p->abcd[3] + update;
        const float acc = H[4][c] * p->strength + update / energy;
        o[c] = max_zero(acc + L[4][c]);
      }
    }
  }
}

static int kind_of(const float a) { return a == 0.0f ? 0 : (a > 0.0f ? 1 : 2); }

int oracle_diffuse_scales(const dt_hip_piece_t *piece, const dt_hip_diffuse_data_t *d)
{
  const float zoom = (float)(d->iscale / piece->roi_in.scale);
  const float final_radius = (float)(d->radius + d->radius_center) * 2.0f / zoom;
  const int s = (int)steps_to_sigma(BSPLINE_SIGMA, final_radius);
  return clampi(s, 1, MAX_SCALES);
}

int oracle_diffuse(const dt_hip_piece_t *piece, const dt_hip_diffuse_data_t *d, const void *in_, void *out_)
{
  const int w = piece->roi_out.width, h = piece->roi_out.height;
  const size_t plane = sizeof(float) * 4 * (size_t)w * h;
  const float zoom = (float)(d->iscale / piece->roi_in.scale);
  const int scales = oracle_diffuse_scales(piece, d);
  const int iterations = (int)ceilf((float)d->iterations) > 1 ? (int)ceilf((float)d->iterations) : 1;

  float *hf[MAX_SCALES] = { 0 };
  float *lf_a = (float *)malloc(plane), *lf_b = (float *)malloc(plane);
  float *t1 = (float *)malloc(plane), *t2 = (float *)malloc(plane);
  int ok = lf_a && lf_b && t1 && t2;
  for(int s = 0; s < scales; s++) ok &= (hf[s] = (float *)malloc(plane)) != NULL;

  pde_t p;
  const float user_aniso[4] = { d->anisotropy_first, d->anisotropy_second, d->anisotropy_third, d->anisotropy_fourth };
  for(int k = 0; k < 4; k++)
  {
    p.anisotropy[k] = sq(user_aniso[k]);
    p.kind[k] = kind_of(user_aniso[k]);
  }
  const float regularization = powf(10.0f, d->regularization) - 1.0f;
  p.variance_threshold = powf(10.0f, d->variance_threshold);
  const float speed[4] = { d->first, d->second, d->third, d->fourth };

  const float *src = (const float *)in_;
  uint8_t *mask = NULL;
  if(ok && d->threshold > 0.0f)
  {
    /* diffuse.c:1207-1219: the iterations start from the inpainted copy in "temp1" */
    mask = (uint8_t *)malloc((size_t)w * h);
    ok = mask != NULL;
    if(ok) build_and_inpaint(src, t1, mask, d->threshold, (size_t)w, (size_t)h);
    src = t1;
  }
  for(int it = 0; ok && it < iterations; it++)
  {
    /* iteration ping-pong, diffuse.c:1223-1249 */
    float *dst = (it == iterations - 1) ? (float *)out_ : ((it % 2 == 0) ? t2 : t1);
    /* analysis: scale s reads the previous low-pass, diffuse.c:1011-1047 */
    const float *level = src;
    float *residual = lf_a, *spare = lf_b;
    for(int s = 0; s < scales; s++)
    {
      float *low = (s % 2 == 0) ? lf_a : lf_b;
      decompose(level, hf[s], low, w, h, 1 << s);
      level = low;
      residual = low;
    }
    spare = (residual == lf_b) ? lf_a : lf_b;
    /* synthesis, coarse to fine, diffuse.c:1051-1103 */
    const float *cur = residual;
    float *pp[2] = { spare, residual };
    int count = 0;
    for(int s = scales - 1; s >= 0; s--, count++)
    {
      const float real_radius = sigma_at_step(BSPLINE_SIGMA, (unsigned)s) * zoom;
      p.regularization = regularization / 9.0f * sq(real_radius);
      const float norm = expf(-sq(real_radius - (float)d->radius_center) / sq((float)d->radius));
      for(int k = 0; k < 4; k++) p.abcd[k] = speed[k] * PDE_KAPPA * norm;
      p.strength = d->sharpness * norm + 1.0f;
      float *to = (s == 0) ? dst : pp[count % 2];
      pde(hf[s], cur, mask, to, w, h, 1 << s, &p);
      cur = to;
    }
    src = dst;
  }
  free(lf_a); free(lf_b); free(t1); free(t2); free(mask);
  for(int s = 0; s < scales; s++) free(hf[s]);
  return ok ? 0 : 1;
}
