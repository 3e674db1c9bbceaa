/* oracle/src/bilat.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of local contrast in bilateral-grid mode:
 *   process()                 src/iop/bilat.c:330-361
 *   dt_bilateral_grid_size()  src/pixel/bilateral.c:50-74
 *   dt_bilateral_splat()      :183-256    dt_bilateral_blur() :341-352 with blur_line() :299-338,
 *   blur_line_z() :258-297    dt_bilateral_slice() :356-393
 *
 * dt_bilateral_splat() gives every OpenMP thread its own slice of image rows and grid rows, sums in
 * binary32 and then merges the slices: the rounding of the grid depends on the thread count.  The
 * restatement accumulates every grid cell in pixel row-major order, which is what the reference does
 * when it runs on ONE thread (tests pin it against oracle/_ref that way).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define MAX_RES_S 3000 /* bilateral.c:47 */
#define MAX_RES_R 50

typedef struct
{
  int size_x, size_y, size_z, width, height;
  float sigma_s, sigma_r;
} grid_t;

static inline float clampf(const float v, const float lo, const float hi) { return v > lo ? (v < hi ? v : hi) : lo; }
static inline int clampi(const int v, const int lo, const int hi) { return v > lo ? (v < hi ? v : hi) : lo; }

static void grid_size(grid_t *b, const int width, const int height, const float L_range, float sigma_s, const float sigma_r)
{
  if(sigma_s < 0.5) sigma_s = 0.5;
  const float _x = (float)clampi((int)roundf(width / sigma_s), 4, MAX_RES_S);
  const float _y = (float)clampi((int)roundf(height / sigma_s), 4, MAX_RES_S);
  const float _z = (float)clampi((int)roundf(L_range / sigma_r), 4, MAX_RES_R);
  const float sy = height / _y, sx = width / _x;
  b->sigma_s = sy > sx ? sy : sx;
  b->sigma_r = L_range / _z;
  b->size_x = (int)ceilf(width / b->sigma_s) + 1;
  b->size_y = (int)ceilf(height / b->sigma_s) + 1;
  b->size_z = (int)ceilf(L_range / b->sigma_r) + 1;
  b->width = width;
  b->height = height;
}

/* image_to_grid() / image_to_relgrid(), bilateral.c:127-155: cell index and fraction on one axis */
static inline int axis(const float v, const float sigma, const int size, float *frac)
{
  const float x = clampf(v / sigma, 0.0f, (float)(size - 1));
  const int xi = (int)x < size - 2 ? (int)x : size - 2;
  *frac = x - xi;
  return xi;
}

static void blur_line(float *buf, const int offset1, const int offset2, const int offset3, const int size1,
                      const int size2, const int size3)
{
  const float w0 = 6.f / 16.f, w1 = 4.f / 16.f, w2 = 1.f / 16.f;
#pragma omp parallel for
  for(int k = 0; k < size1; k++)
  {
    size_t index = (size_t)k * offset1;
    for(int j = 0; j < size2; j++)
    {
      float tmp1 = buf[index];
      buf[index] = buf[index] * w0 + w1 * buf[index + offset3] + w2 * buf[index + 2 * offset3];
      index += offset3;
      float tmp2 = buf[index];
      buf[index] = buf[index] * w0 + w1 * (buf[index + offset3] + tmp1) + w2 * buf[index + 2 * offset3];
      index += offset3;
      for(int i = 2; i < size3 - 2; i++)
      {
        const float tmp3 = buf[index];
        buf[index] = buf[index] * w0 + w1 * (buf[index + offset3] + tmp2) + w2 * (buf[index + 2 * offset3] + tmp1);
        index += offset3;
        tmp1 = tmp2;
        tmp2 = tmp3;
      }
      const float tmp3 = buf[index];
      buf[index] = buf[index] * w0 + w1 * (buf[index + offset3] + tmp2) + w2 * tmp1;
      index += offset3;
      buf[index] = buf[index] * w0 + w1 * tmp3 + w2 * tmp2;
      index += offset3;
      index += offset2 - offset3 * size3;
    }
  }
}

static void blur_line_z(float *buf, const int offset1, const int offset2, const int offset3, const int size1,
                        const int size2, const int size3)
{
  const float w1 = 4.f / 16.f, w2 = 2.f / 16.f;
#pragma omp parallel for
  for(int k = 0; k < size1; k++)
  {
    size_t index = (size_t)k * offset1;
    for(int j = 0; j < size2; j++)
    {
      float tmp1 = buf[index];
      buf[index] = w1 * buf[index + offset3] + w2 * buf[index + 2 * offset3];
      index += offset3;
      float tmp2 = buf[index];
      buf[index] = w1 * (buf[index + offset3] - tmp1) + w2 * buf[index + 2 * offset3];
      index += offset3;
      for(int i = 2; i < size3 - 2; i++)
      {
        const float tmp3 = buf[index];
        buf[index] = +w1 * (buf[index + offset3] - tmp2) + w2 * (buf[index + 2 * offset3] - tmp1);
        index += offset3;
        tmp1 = tmp2;
        tmp2 = tmp3;
      }
      const float tmp3 = buf[index];
      buf[index] = w1 * (buf[index + offset3] - tmp2) - w2 * tmp1;
      index += offset3;
      buf[index] = -w1 * tmp3 - w2 * tmp2;
      index += offset3;
      index += offset2 - offset3 * size3;
    }
  }
}

void oracle_bilat_grid(const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d, int dims[3], float sigmas[2])
{
  grid_t b;
  const float scale = (float)(d->iscale / piece->roi_in.scale);
  grid_size(&b, piece->roi_in.width, piece->roi_in.height, 100.0f, d->sigma_s / scale, d->sigma_r);
  dims[0] = b.size_x;
  dims[1] = b.size_y;
  dims[2] = b.size_z;
  sigmas[0] = b.sigma_s;
  sigmas[1] = b.sigma_r;
}

int oracle_local_laplacian(const float *input, float *out, int wd, int ht, float sigma, float shadows, float highlights,
                           float clarity);

int oracle_bilat(const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d, const void *in_, void *out_)
{
  if(d->mode == DT_HIP_BILAT_LOCAL_LAPLACIAN) /* bilat.c:352-357 */
  {
    /* a 2- or 3-pixel side makes the reference read outside its buffers: undefined, refused (as the device does) */
    {
      const int m = piece->roi_in.width < piece->roi_in.height ? piece->roi_in.width : piece->roi_in.height;
      if(m == 2 || m == 3) return 1;
    }
    return oracle_local_laplacian((const float *)in_, (float *)out_, piece->roi_in.width, piece->roi_in.height, d->midtone,
                                  d->sigma_s, d->sigma_r, d->detail);
  }
  if(d->mode != DT_HIP_BILAT_BILATERAL) return 1;
  const float *in = (const float *)in_;
  float *out = (float *)out_;
  const int width = piece->roi_in.width, height = piece->roi_in.height;
  const float scale = (float)(d->iscale / piece->roi_in.scale);
  grid_t b;
  grid_size(&b, width, height, 100.0f, d->sigma_s / scale, d->sigma_r);
  /* the reference's blur touches four entries of every grid line unconditionally (bilateral.c:266-340): shorter
   * lines are written past their end there -- undefined, refused (as the device does) */
  if(b.size_x < 4 || b.size_y < 4 || b.size_z < 4) return 1;
  const int ox = b.size_z, oy = b.size_x * b.size_z, oz = 1;
  /* two spare grid rows like the reference's slice buffer (slicerows = size_y + 2 on one thread) */
  float *buf = (float *)calloc((size_t)oy * (b.size_y + 2), sizeof(float));
  if(!buf) return 1;

  /* splat, row-major */
  const float s2 = b.sigma_s * b.sigma_s;
  for(int j = 0; j < height; j++)
  {
    float yf;
    const int yi = axis((float)j, b.sigma_s, b.size_y, &yf);
    for(int i = 0; i < width; i++)
    {
      float xf, zf;
      const float L = in[4 * ((size_t)j * width + i)];
      const int xi = axis((float)i, b.sigma_s, b.size_x, &xf);
      const int zi = axis(L, b.sigma_r, b.size_z, &zf);
      const size_t gi = (size_t)yi * oy + (size_t)xi * ox + zi;
      const float contrib[4] = { (1.0f - xf) * (1.0f - yf) * 100.0f / s2, xf * (1.0f - yf) * 100.0f / s2,
                                 (1.0f - xf) * yf * 100.0f / s2, xf * yf * 100.0f / s2 };
      const size_t off[4] = { 0, (size_t)ox, (size_t)oy, (size_t)ox + oy };
      for(int k = 0; k < 4; k++)
      {
        buf[gi + off[k]] += (contrib[k] * (1.0f - zf));
        buf[gi + off[k] + oz] += (contrib[k] * zf);
      }
    }
  }
  /* blur: x, y (6-4-1 binomial), z (derivative) */
  blur_line(buf, oz, oy, ox, b.size_z, b.size_y, b.size_x);
  blur_line(buf, oz, ox, oy, b.size_z, b.size_x, b.size_y);
  blur_line_z(buf, ox, oy, oz, b.size_x, b.size_y, b.size_z);
  /* slice */
  const float norm = -d->detail * b.sigma_r * 0.04f;
#pragma omp parallel for
  for(int j = 0; j < height; j++)
    for(int i = 0; i < width; i++)
    {
      const size_t index = 4 * ((size_t)j * width + i);
      float xf, yf, zf;
      const float L = in[index];
      const int xi = axis((float)i, b.sigma_s, b.size_x, &xf);
      const int yi = axis((float)j, b.sigma_s, b.size_y, &yf);
      const int zi = axis(L, b.sigma_r, b.size_z, &zf);
      const size_t gi = ((size_t)xi + (size_t)yi * b.size_x) * b.size_z + zi;
      const float Lout = fmaxf(0.0f, L
                         + norm * (buf[gi] * (1.0f - xf) * (1.0f - yf) * (1.0f - zf)
                                   + buf[gi + ox] * (xf) * (1.0f - yf) * (1.0f - zf)
                                   + buf[gi + oy] * (1.0f - xf) * (yf) * (1.0f - zf)
                                   + buf[gi + ox + oy] * (xf) * (yf) * (1.0f - zf)
                                   + buf[gi + oz] * (1.0f - xf) * (1.0f - yf) * (zf)
                                   + buf[gi + ox + oz] * (xf) * (1.0f - yf) * (zf)
                                   + buf[gi + oy + oz] * (1.0f - xf) * (yf) * (zf)
                                   + buf[gi + ox + oy + oz] * (xf) * (yf) * (zf)));
      out[index] = Lout;
      out[index + 1] = in[index + 1];
      out[index + 2] = in[index + 2];
      out[index + 3] = in[index + 3];
    }
  free(buf);
  return 0;
}
