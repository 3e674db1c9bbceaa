// demosaic_extras.hip -- the optional steps process() runs around the interpolation (src/iop/demosaic.c:1137-1250):
//   green_equilibration_lavg()  src/iop/demosaic/basic.c:248-293   on the mosaic, before
//   green_equilibration_favg()  src/iop/demosaic/basic.c:296-329   on the mosaic, before (alone or ahead of _lavg)
//   pre_median()                src/iop/demosaic/basic.c:136-186   on the green sites, inside demosaic_ppg() (ppg.c:58-67)
//   color_smoothing()           src/iop/demosaic/basic.c:191-243   on the output, after
// All three are local: one thread per pixel, reads through L1/L2.  Algorithmic bytes: 8 / 8 / 2 x (16 + 16 + 16) per pass.
#include "hip_common.h"

using namespace ansel;

namespace
{

__device__ __forceinline__ int fc(const int row, const int col, const uint32_t filters)
{
  return filters >> (((row << 1 & 14) + (col & 1)) << 1) & 3; // FC(), src/develop/imageop_math.h:190
}

// mean absolute difference of the six pairs of four greens: (0,1) (0,2) (0,3) (1,2) (2,3) (1,3), left to right
__device__ __forceinline__ float mean_abs_pairs(const float g[4])
{
  float acc = fabsf(g[0] - g[1]);
  acc = acc + fabsf(g[0] - g[2]);
  acc = acc + fabsf(g[0] - g[3]);
  acc = acc + fabsf(g[1] - g[2]);
  acc = acc + fabsf(g[2] - g[3]);
  acc = acc + fabsf(g[1] - g[3]);
  return acc / 6.0f;
}

__global__ __launch_bounds__(256) void green_eq_lavg(const float *__restrict__ in, float *__restrict__ out, const int width,
                                                     const int height, const int oj, const int oi, const float thr)
{
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if(i >= width || j >= height) return;
  const size_t w = (size_t)width, p = (size_t)j * w + i;
  float v = in[p];
  if(j >= oj && j < height - 2 && ((j - oj) & 1) == 0 && i >= oi && i < width - 2 && ((i - oi) & 1) == 0)
  {
    // the four diagonal greens and the four axial greens at distance 2; sums in the reference's operand order
    const float diag[4] = { in[p - w - 1], in[p - w + 1], in[p + w - 1], in[p + w + 1] };
    const float axial[4] = { in[p - 2 * w], in[p + 2 * w], in[p - 2], in[p + 2] };
    const float m_diag = (((diag[0] + diag[1]) + diag[2]) + diag[3]) / 4.0f;
    const float m_axial = (((axial[0] + axial[1]) + axial[2]) + axial[3]) / 4.0f;
    if((m_axial > 0.0f) && (m_diag > 0.0f) && (m_diag / m_axial < 2.0f))
    {
      const float flat_d = mean_abs_pairs(diag), flat_a = mean_abs_pairs(axial);
      if((v < 0.95f) && (flat_d < thr) && (flat_a < thr)) v = v * m_diag / m_axial;
    }
  }
  out[p] = v;
}

// green_equilibration_favg(): the greens of the lattice's rows are scaled by sum(greens of the other rows) / sum(own).
// The reference forms the two sums in binary64 in an OpenMP reduction, so its low bits follow the host's thread count;
// here each sum is carried as an unevaluated pair (TwoSum) in a fixed order -- ~2^-100 relative, i.e. the correctly
// rounded binary64 sum for all practical purposes, which every order of the reference's additions approximates to a few
// ulp of binary64.  After the product is rounded to binary32 a pixel can differ from a given run of the reference only
// when in * ratio falls within ~1e-15 of a rounding boundary (expected: about one pixel in 1e8).
struct dd_t
{
  double hi, lo;
};
__device__ __forceinline__ dd_t dd_add(const dd_t a, const double b)
{
  const double s = a.hi + b;
  const double bb = s - a.hi;
  const double e = (a.hi - (s - bb)) + (b - bb);
  return { s, isfinite(s) ? a.lo + e : 0.0 }; // an infinite or NaN sum stays what binary64 addition makes of it
}
__device__ __forceinline__ dd_t dd_merge(const dd_t a, const dd_t b)
{
  dd_t r = dd_add(a, b.hi);
  r.lo += b.lo;
  const double s = r.hi + r.lo; // renormalise
  return { s, isfinite(s) ? r.lo - (s - r.hi) : 0.0 };
}

#define FAVG_BLOCKS 1024
// partial sums: workgroup b takes the lattice rows b, b + FAVG_BLOCKS, ...; a lane the sites lane, lane + 256, ... of a row
__global__ __launch_bounds__(256) void green_eq_favg_sums(const float *__restrict__ in, const int width, const int height, const int oi,
                                                          const int g2_offset, dd_t *__restrict__ partial)
{
  __shared__ dd_t red[2][256];
  const int sites = (width - 1 - g2_offset - oi + 1) / 2; // i = oi, oi + 2, ... < width - 1 - g2_offset
  const int rows = height / 2;                            // j = 0, 2, ... < height - 1
  dd_t s1 = { 0.0, 0.0 }, s2 = { 0.0, 0.0 };
  for(int r = blockIdx.x; r < rows; r += FAVG_BLOCKS)
  {
    const float *row = in + (size_t)(2 * r) * width;
    for(int k = threadIdx.x; k < sites; k += 256)
    {
      const int i = oi + 2 * k;
      s1 = dd_add(s1, (double)row[i]);
      s2 = dd_add(s2, (double)row[width + i + g2_offset]);
    }
  }
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  __syncthreads();
  for(int half = 128; half > 0; half >>= 1)
  {
    if((int)threadIdx.x < half)
    {
      red[0][threadIdx.x] = dd_merge(red[0][threadIdx.x], red[0][threadIdx.x + half]);
      red[1][threadIdx.x] = dd_merge(red[1][threadIdx.x], red[1][threadIdx.x + half]);
    }
    __syncthreads();
  }
  if(threadIdx.x == 0)
  {
    partial[blockIdx.x] = red[0][0];
    partial[FAVG_BLOCKS + blockIdx.x] = red[1][0];
  }
}

// the partials in order, then { gr_ratio, 1 } or { 0, 0 } where "sum1 > 0.0 && sum2 > 0.0" fails (the copy is the result)
__global__ __launch_bounds__(64) void green_eq_favg_ratio(const dd_t *__restrict__ partial, double *__restrict__ ratio)
{
  if(threadIdx.x != 0) return;
  dd_t s1 = { 0.0, 0.0 }, s2 = { 0.0, 0.0 };
  for(int b = 0; b < FAVG_BLOCKS; b++)
  {
    s1 = dd_merge(s1, partial[b]);
    s2 = dd_merge(s2, partial[FAVG_BLOCKS + b]);
  }
  const double sum1 = s1.hi, sum2 = s2.hi;
  const bool valid = sum1 > 0.0 && sum2 > 0.0;
  ratio[0] = valid ? sum2 / sum1 : 0.0;
  ratio[1] = valid ? 1.0 : 0.0;
}

__global__ __launch_bounds__(256) void green_eq_favg_apply(const float *__restrict__ in, float *__restrict__ out, const int width,
                                                           const int height, const int oi, const int g2_offset,
                                                           const double *__restrict__ ratio)
{
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if(i >= width || j >= height) return;
  const size_t p = (size_t)j * width + i;
  float v = in[p];
  const double gr_ratio = ratio[0];
  if(ratio[1] != 0.0 && (j & 1) == 0 && j < height - 1 && i >= oi && ((i - oi) & 1) == 0 && i < width - 1 - g2_offset)
    v = (float)((double)v * gr_ratio);
  out[p] = v;
}

// pre_median_b(), one pass: the exchange order of the reference's sort is kept (it decides where a NaN ends up)
__global__ __launch_bounds__(256) void pre_median(const float *__restrict__ in, float *__restrict__ out, const int width,
                                                  const int height, const uint32_t filters, const float threshold)
{
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), row = blockIdx.y * 4 + (threadIdx.x >> 6);
  if(col >= width || row >= height) return;
  const size_t p = (size_t)row * width + col;
  float v = in[p];
  const int f3 = fc(row, 3, filters);
  const int col0 = (f3 != 1 && f3 != 3) ? 4 : 3;
  if(row >= 3 && row < height - 3 && col >= col0 && col < width - 3 && ((col - col0) & 1) == 0)
  {
    float med[9];
    int cnt = 0;
    // the diamond of same-colour neighbours, rows -2..2 with column offsets {0}, {-1, 1}, {-2, 0, 2}, {-1, 1}, {0} (:142-168)
    const int dy[9] = { -2, -1, -1, 0, 0, 0, 1, 1, 2 }, dx[9] = { 0, -1, 1, -2, 0, 2, -1, 1, 0 };
#pragma unroll
    for(int k = 0; k < 9; k++)
    {
      const float s = in[p + (ptrdiff_t)width * dy[k] + dx[k]];
      if(fabsf(s - v) < threshold)
      {
        med[k] = s;
        cnt++;
      }
      else
        med[k] = 64.0f + s;
    }
#pragma unroll
    for(int i = 0; i < 8; i++)
#pragma unroll
      for(int ii = i + 1; ii < 9; ii++)
        if(med[i] > med[ii])
        {
          const float t = med[i];
          med[i] = med[ii];
          med[ii] = t;
        }
    // med[(cnt - 1) / 2] with cnt = 1..9 (the centre always counts unless it is NaN: cnt 0 -> med[0], C truncation)
    const int idx = (cnt - 1) / 2;
    float pick = med[0];
#pragma unroll
    for(int q = 1; q < 5; q++) pick = idx == q ? med[q] : pick;
    v = (cnt == 1) ? med[4] - 64.0f : pick;
  }
  out[p] = v;
}

// color_smoothing(): first the channel goes to alpha over the whole frame, then the 3x3 median of (alpha - green)
__global__ __launch_bounds__(256) void smooth_stash(float4 *__restrict__ img, const size_t npixels, const int c)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= npixels) return;
  float4 v = img[k];
  v.w = c == 0 ? v.x : v.z;
  img[k] = v;
}

__global__ __launch_bounds__(256) void smooth_median(float4 *__restrict__ img, const int width, const int height, const int c)
{
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if(i < 1 || j < 1 || i >= width - 1 || j >= height - 1) return;
  const size_t p = (size_t)j * width + i;
  float med[9];
#pragma unroll
  for(int dj = -1; dj <= 1; dj++)
#pragma unroll
    for(int di = -1; di <= 1; di++)
    {
      // alpha and green of the neighbours: neither is written by this launch (it stores red or blue only)
      const float *const n = reinterpret_cast<const float *>(img + (p + (ptrdiff_t)dj * width + di));
      med[3 * (dj + 1) + (di + 1)] = n[3] - n[1];
    }
#define SWAPmed(I, J) if(med[I] > med[J]) { const float t_ = med[I]; med[I] = med[J]; med[J] = t_; }
  SWAPmed(1, 2) SWAPmed(4, 5) SWAPmed(7, 8) SWAPmed(0, 1) SWAPmed(3, 4) SWAPmed(6, 7) SWAPmed(1, 2) SWAPmed(4, 5)
  SWAPmed(7, 8) SWAPmed(0, 3) SWAPmed(5, 8) SWAPmed(4, 7) SWAPmed(3, 6) SWAPmed(1, 4) SWAPmed(2, 5) SWAPmed(4, 7)
  SWAPmed(4, 2) SWAPmed(6, 4) SWAPmed(4, 2)
#undef SWAPmed
  const float own_green = reinterpret_cast<const float *>(img + p)[1];
  const float r = fmaxf(med[4] + own_green, 0.0f);
  float *const o = reinterpret_cast<float *>(img + p);
  o[c] = r;
}

} // namespace

namespace ansel
{

int green_eq_lavg_launch(int devid, const float *in, float *out, int width, int height, uint32_t filters, int x, int y, float thr)
{
  auto FCh = [&](int row, int col) { return (int)(filters >> (((row << 1 & 14) + (col & 1)) << 1) & 3); };
  int oj = 2, oi = 2;
  if(FCh(oj + y, oi + x) != 1) oj++;
  if(FCh(oj + y, oi + x) != 1) oi++;
  if(FCh(oj + y, oi + x) != 1) oj--;
  launch_scope ls(devid, "green_eq_lavg");
  green_eq_lavg<<<dim3((width + 63) / 64, (height + 3) / 4), 256, 0, stream_of(devid)>>>(in, out, width, height, oj, oi, thr);
  return check_launch("green_eq_lavg");
}

int green_eq_favg_launch(int devid, const float *in, float *out, int width, int height, uint32_t filters, int x, int y)
{
  auto FCh = [&](int row, int col) { return (int)(filters >> (((row << 1 & 14) + (col & 1)) << 1) & 3); };
  const int oi = (FCh(y, x) & 1) != 1 ? 1 : 0;
  const int g2_offset = oi ? -1 : 1;
  hipStream_t s = stream_of(devid);
  dd_t *partial = (dd_t *)dt_hip_alloc_device_buffer(devid, sizeof(dd_t) * 2 * FAVG_BLOCKS + 2 * sizeof(double));
  if(!partial) return DT_HIP_SYSMEM_ALLOCATION;
  double *ratio = (double *)(partial + 2 * FAVG_BLOCKS);
  {
    launch_scope ls(devid, "green_eq_favg_sums");
    green_eq_favg_sums<<<FAVG_BLOCKS, 256, 0, s>>>(in, width, height, oi, g2_offset, partial);
    green_eq_favg_ratio<<<1, 64, 0, s>>>(partial, ratio);
  }
  {
    launch_scope ls(devid, "green_eq_favg_apply");
    green_eq_favg_apply<<<dim3((width + 63) / 64, (height + 3) / 4), 256, 0, s>>>(in, out, width, height, oi, g2_offset, ratio);
  }
  dt_hip_release_mem_object(partial);
  return check_launch("green_eq_favg");
}

// passthrough_monochrome() / passthrough_color(), src/iop/demosaic/passthrough.c:21-67: the photosite in all three channels,
// or in its own (the colour from the unshifted filters at the output position) with 0 in the other two; alpha is left as it is
template <bool COLOR>
__global__ __launch_bounds__(256) void passthrough(const float *__restrict__ in, float *__restrict__ out, const int width,
                                                   const size_t npixels, const uint32_t filters)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if(i >= npixels) return;
  const float v = in[i];
  typedef float v3f_t __attribute__((ext_vector_type(3)));
  v3f_t o = { v, v, v };
  if(COLOR)
  {
    const int row = (int)(i / (size_t)width), col = (int)(i % (size_t)width);
    const int c = filters >> ((((row << 1) & 14) + (col & 1)) << 1) & 3;
    o = v3f_t{ c == 0 ? v : 0.0f, c == 1 ? v : 0.0f, c == 2 ? v : 0.0f };
  }
  __builtin_memcpy(out + 4 * i, &o, 12);
}

int passthrough_launch(int devid, const float *in, float4 *out, int width, int height, uint32_t filters, bool color)
{
  const size_t np = (size_t)width * height;
  launch_scope ls(devid, "passthrough");
  if(color)
    passthrough<true><<<pixel_grid(np), 256, 0, stream_of(devid)>>>(in, (float *)out, width, np, filters);
  else
    passthrough<false><<<pixel_grid(np), 256, 0, stream_of(devid)>>>(in, (float *)out, width, np, filters);
  return check_launch("passthrough");
}

int pre_median_launch(int devid, const float *in, float *out, int width, int height, uint32_t filters, float threshold)
{
  launch_scope ls(devid, "pre_median");
  pre_median<<<dim3((width + 63) / 64, (height + 3) / 4), 256, 0, stream_of(devid)>>>(in, out, width, height, filters, threshold);
  return check_launch("pre_median");
}

int color_smoothing_launch(int devid, float4 *img, int width, int height, int passes)
{
  hipStream_t s = stream_of(devid);
  const size_t np = (size_t)width * height;
  launch_scope ls(devid, "color_smoothing");
  for(int pass = 0; pass < passes; pass++)
    for(int c = 0; c < 3; c += 2)
    {
      smooth_stash<<<pixel_grid(np), 256, 0, s>>>(img, np, c);
      smooth_median<<<dim3((width + 63) / 64, (height + 3) / 4), 256, 0, s>>>(img, width, height, c);
    }
  return check_launch("color_smoothing");
}

} // namespace ansel
