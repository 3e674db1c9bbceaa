// px_bilat.h -- local contrast, bilateral-grid mode: the per-pixel slice, shared by the module's own kernel (bilat.hip bilat_slice) and
// by the fused RGBA run behind the module (rgb_chain_kernel.h, round 6: the slice is pointwise given the blurred grid, so it becomes the
// first stage of the run that follows -- a float4 plane less to write and to read).  dt_bilateral_slice(), src/pixel/bilateral.c:356-393.
#pragma once
#include "hip_common.h"

namespace ansel
{

struct bilat_grid_t
{
  int size_x, size_y, size_z, width, height;
  float sigma_s, sigma_r;
};
// what a fused run needs of the module: the blurred grid, its geometry, -detail x sigma_r x 0.04
struct bilat_slice_args
{
  bilat_grid_t b;
  const float *grid;
  float norm;
};

__host__ __device__ __forceinline__ float bilat_clampf(const float v, const float lo, const float hi)
{
  return v > lo ? (v < hi ? v : hi) : lo; // CLAMPS(), src/math/math.h
}
// image_to_grid() / image_to_relgrid(), bilateral.c:127-155: cell index and fraction on one axis
__device__ __forceinline__ int bilat_axis(const float v, const float sigma, const int size, float &frac)
{
  const float x = bilat_clampf(v / sigma, 0.0f, (float)(size - 1));
  const int xi = (int)x < size - 2 ? (int)x : size - 2;
  frac = x - xi;
  return xi;
}
// the lightness of the pixel at column i, row j of the frame
__device__ __forceinline__ float bilat_slice_lightness(const float L, const int i, const int j, const float *__restrict__ buf,
                                                       const bilat_grid_t &b, const float norm)
{
  const int ox = b.size_z, oy = b.size_x * b.size_z, oz = 1;
  float xf, yf, zf;
  const int xi = bilat_axis((float)i, b.sigma_s, b.size_x, xf);
  const int yi = bilat_axis((float)j, b.sigma_s, b.size_y, yf);
  const int zi = bilat_axis(L, b.sigma_r, b.size_z, zf);
  const size_t gi = ((size_t)xi + (size_t)yi * b.size_x) * b.size_z + zi;
  return fmaxf(0.0f, L
                     + norm * (buf[gi] * (1.0f - xf) * (1.0f - yf) * (1.0f - zf)
                               + buf[gi + ox] * (xf) * (1.0f - yf) * (1.0f - zf)
                               + buf[gi + oy] * (1.0f - xf) * (yf) * (1.0f - zf)
                               + buf[gi + ox + oy] * (xf) * (yf) * (1.0f - zf)
                               + buf[gi + oz] * (1.0f - xf) * (1.0f - yf) * (zf)
                               + buf[gi + ox + oz] * (xf) * (1.0f - yf) * (zf)
                               + buf[gi + oy + oz] * (1.0f - xf) * (yf) * (zf)
                               + buf[gi + ox + oy + oz] * (xf) * (yf) * (zf)));
}

} // namespace ansel
