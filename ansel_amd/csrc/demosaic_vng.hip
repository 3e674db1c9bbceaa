// demosaic_vng.hip -- VNG4 interpolation of a Bayer mosaic and the dual demosaic that blends it with RCD / AMaZE.
//
// Reference: lin_interpolate(), src/iop/demosaic/basic.c:22-111; vng_interpolate(), src/iop/demosaic/vng.c:34-221 (Bayer:
// four colours -- the second green of a 2 x 2 cell apart -- mixed into one green at the end); dual_demosaic(),
// src/iop/demosaic/dual.c:35-110.
//
// The reference walks the frame row by row and keeps its results in a ring of three rows before storing them, so that
// every gradient and every average of the VNG step reads the LINEAR interpolation of its 5 x 5 neighbourhood, never a
// finished pixel: the step is a pure function of the linear interpolation, one thread per pixel here.  What is kept:
// the order of every sum (neighbours in row-major order in the linear step; the terms of dcraw's table in table order
// into each of the eight gradients; the directions below the threshold in order 0..7 into the averages), the integer
// weights converted where the reference converts them, the per-phase code tables (built on the host from the same
// table, one per (row mod 8, column mod 2)).  A wave holds pixels of ONE phase (columns of one parity): no divergence
// between the two term lists of a row.
#include "hip_common.h"

#include <math.h>
#include <stdlib.h>
#include <vector>

using namespace ansel;

namespace ansel
{
int color_smoothing_launch(int devid, float4 *img, int width, int height, int passes);
int dual_blend_mask_launch(int devid, float4 *rgb, const float wb[3], float threshold, int width, int height, float *mask);
}

namespace
{

__host__ __device__ __forceinline__ int fc(const int row, const int col, const uint32_t filters)
{
  return filters >> ((((row) << 1 & 14) + ((col)&1)) << 1) & 3; // FC(), src/develop/imageop_math.h:190-193
}

// the dcraw filter word with the second green of a 2 x 2 cell as colour 3, vng.c:61-68
uint32_t filters4_of(const uint32_t filters) { return (filters & 3) == 1 ? filters | 0x03030303u : filters | 0x0c0c0c0cu; }

__device__ __forceinline__ float chan(const float4 p, const int c) { return c == 0 ? p.x : (c == 1 ? p.y : (c == 2 ? p.z : p.w)); }

// lin_interpolate(): the frame's outermost ring is the mean of the adjoining photosites of each colour (:28-58), the
// interior the weighted 3 x 3 sums of the lookup table (:72-109: neighbours in row-major order, weight 1 << ((y == 0) + (x == 0)))
__global__ __launch_bounds__(256) void vng_lin(const float *__restrict__ in, float4 *__restrict__ out, const int width,
                                               const int height, const int rx, const int ry, const uint32_t filters4)
{
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), row = blockIdx.y * 4 + (threadIdx.x >> 6);
  if(col >= width || row >= height) return;
  const int f = fc(row + ry, col + rx, filters4);
  const float own = in[(size_t)row * width + col];
  float sum[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
  float res[4];
  if(row == 0 || col == 0 || row == height - 1 || col == width - 1)
  {
    int count[4] = { 0, 0, 0, 0 };
    for(int y = row - 1; y != row + 2; y++)
      for(int x = col - 1; x != col + 2; x++)
        if(y >= 0 && x >= 0 && y < height && x < width)
        {
          const int c = fc(y + ry, x + rx, filters4);
          const float v = in[(size_t)y * width + x];
#pragma unroll
          for(int k = 0; k < 4; k++)
            if(k == c)
            {
              sum[k] += v;
              count[k]++;
            }
        }
#pragma unroll
    for(int c = 0; c < 4; c++) res[c] = (c != f && count[c] != 0) ? sum[c] / (float)count[c] : own;
  }
  else
  {
    int wsum[4] = { 0, 0, 0, 0 };
#pragma unroll
    for(int y = -1; y <= 1; y++)
#pragma unroll
      for(int x = -1; x <= 1; x++)
      {
        const int weight = 1 << ((y == 0) + (x == 0));
        const int c = fc(row + y + ry, col + x + rx, filters4);
        if(c == f) continue;
        const float v = in[(size_t)(row + y) * width + col + x] * (float)weight;
#pragma unroll
        for(int k = 0; k < 4; k++)
          if(k == c)
          {
            sum[k] += v;
            wsum[k] += weight;
          }
      }
#pragma unroll
    for(int c = 0; c < 4; c++) res[c] = c != f ? sum[c] / (float)wsum[c] : own;
  }
  out[(size_t)row * width + col] = make_float4(res[0], res[1], res[2], res[3]);
}

// one term of dcraw's table that survives for a phase (vng.c:97-115), and one of the eight neighbours of the averages
struct vng_term
{
  signed char y1, x1, y2, x2;
  unsigned char color, weight, grads, pad;
};
struct vng_hood
{
  signed char y, x;
  unsigned char far, pad; // far: the pixel's own colour is taken half-way from the site two steps away (:118-127)
};
struct vng_code
{
  int nterms;
  vng_term terms[64];
  vng_hood hood[8];
};

__global__ __launch_bounds__(256) void vng_main(const float4 *__restrict__ lin, float4 *__restrict__ out, const int width,
                                                const int height, const int rx, const int ry, const uint32_t filters4,
                                                const vng_code *__restrict__ codes)
{
  // blockIdx.z = column parity: the 64 lanes of a wave are pixels of one phase
  const int col = 2 * (blockIdx.x * 64 + (threadIdx.x & 63)) + blockIdx.z, row = blockIdx.y * 4 + (threadIdx.x >> 6);
  if(col >= width || row >= height) return;
  const size_t idx = (size_t)row * width + col;
  float4 px = lin[idx];
  if(row >= 2 && col >= 2 && row < height - 2 && col < width - 2)
  {
    const vng_code &code = codes[((row + ry) & 7) * 2 + ((col + rx) & 1)];
    float grad[8] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
    for(int t = 0; t < code.nterms; t++)
    {
      const vng_term tm = code.terms[t];
      const float a = chan(lin[idx + (ptrdiff_t)tm.y1 * width + tm.x1], tm.color);
      const float b = chan(lin[idx + (ptrdiff_t)tm.y2 * width + tm.x2], tm.color);
      const float diff = fabsf(a - b) * (float)tm.weight;
#pragma unroll
      for(int g = 0; g < 8; g++)
        if(tm.grads & (1 << g)) grad[g] += diff;
    }
    float grad_lo = grad[0], grad_hi = grad[0];
#pragma unroll
    for(int g = 1; g < 8; g++)
    {
      if(grad_lo > grad[g]) grad_lo = grad[g];
      if(grad_hi < grad[g]) grad_hi = grad[g];
    }
    if(!(grad_hi == 0)) // `if(grad_hi == 0) keep the linear interpolation`, :151-155 (a NaN maximum goes on)
    {
      const float cut = grad_lo + (grad_hi * 0.5f);
      const int color = fc(row + ry, col + rx, filters4);
      const float own = chan(px, color);
      float sum[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
      int num = 0;
#pragma unroll
      for(int g = 0; g < 8; g++)
        if(grad[g] <= cut)
        {
          const vng_hood hd = code.hood[g];
          const float4 q = lin[idx + (ptrdiff_t)hd.y * width + hd.x];
          float far = 0.0f;
          if(hd.far) far = chan(lin[idx + (ptrdiff_t)(2 * hd.y) * width + 2 * hd.x], color);
#pragma unroll
          for(int c = 0; c < 4; c++)
            sum[c] += (c == color && hd.far) ? (own + far) * 0.5f : chan(q, c);
          num++;
        }
      const float sc = chan(make_float4(sum[0], sum[1], sum[2], sum[3]), color);
      float res[4];
#pragma unroll
      for(int c = 0; c < 4; c++)
      {
        float tot = own;
        if(c != color) tot += (sum[c] - sc) / (float)num;
        res[c] = tot;
      }
      px = make_float4(res[0], res[1], res[2], res[3]);
    }
  }
  // VNG4: the two greens become one (:205-208); the fourth lane keeps the second green
  px.y = (px.y + px.w) / 2.0f;
  out[idx] = px;
}

// out = mask * (high - low) + low, all four lanes: intp(), demosaic.c:250-257, dual.c:99-106
__global__ __launch_bounds__(256) void dual_blend(float4 *__restrict__ rgb, const float4 *__restrict__ vng,
                                                  const float *__restrict__ mask, const size_t n)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= n) return;
  const float a = mask[k];
  const float4 b = rgb[k], c = vng[k];
  rgb[k] = make_float4(a * (b.x - c.x) + c.x, a * (b.y - c.y) + c.y, a * (b.z - c.z) + c.z, a * (b.w - c.w) + c.w);
}

const signed char k_terms[] = {
  -2, -2, +0, -1, 1, 0x01, -2, -2, +0, +0, 2, 0x01, -2, -1, -1, +0, 1, 0x01, -2, -1, +0, -1, 1, 0x02, -2, -1, +0, +0, 1, 0x03,
  -2, -1, +0, +1, 2, 0x01, -2, +0, +0, -1, 1, 0x06, -2, +0, +0, +0, 2, 0x02, -2, +0, +0, +1, 1, 0x03, -2, +1, -1, +0, 1, 0x04,
  -2, +1, +0, -1, 2, 0x04, -2, +1, +0, +0, 1, 0x06, -2, +1, +0, +1, 1, 0x02, -2, +2, +0, +0, 2, 0x04, -2, +2, +0, +1, 1, 0x04,
  -1, -2, -1, +0, 1, (signed char)0x80, -1, -2, +0, -1, 1, 0x01, -1, -2, +1, -1, 1, 0x01, -1, -2, +1, +0, 2, 0x01,
  -1, -1, -1, +1, 1, (signed char)0x88, -1, -1, +1, -2, 1, 0x40, -1, -1, +1, -1, 1, 0x22, -1, -1, +1, +0, 1, 0x33,
  -1, -1, +1, +1, 2, 0x11, -1, +0, -1, +2, 1, 0x08, -1, +0, +0, -1, 1, 0x44, -1, +0, +0, +1, 1, 0x11, -1, +0, +1, -2, 2, 0x40,
  -1, +0, +1, -1, 1, 0x66, -1, +0, +1, +0, 2, 0x22, -1, +0, +1, +1, 1, 0x33, -1, +0, +1, +2, 2, 0x10, -1, +1, +1, -1, 2, 0x44,
  -1, +1, +1, +0, 1, 0x66, -1, +1, +1, +1, 1, 0x22, -1, +1, +1, +2, 1, 0x10, -1, +2, +0, +1, 1, 0x04, -1, +2, +1, +0, 2, 0x04,
  -1, +2, +1, +1, 1, 0x04, +0, -2, +0, +0, 2, (signed char)0x80, +0, -1, +0, +1, 2, (signed char)0x88, +0, -1, +1, -2, 1, 0x40,
  +0, -1, +1, +0, 1, 0x11, +0, -1, +2, -2, 1, 0x40, +0, -1, +2, -1, 1, 0x20, +0, -1, +2, +0, 1, 0x30, +0, -1, +2, +1, 2, 0x10,
  +0, +0, +0, +2, 2, 0x08, +0, +0, +2, -2, 2, 0x40, +0, +0, +2, -1, 1, 0x60, +0, +0, +2, +0, 2, 0x20, +0, +0, +2, +1, 1, 0x30,
  +0, +0, +2, +2, 2, 0x10, +0, +1, +1, +0, 1, 0x44, +0, +1, +1, +2, 1, 0x10, +0, +1, +2, -1, 2, 0x40, +0, +1, +2, +0, 1, 0x60,
  +0, +1, +2, +1, 1, 0x20, +0, +1, +2, +2, 1, 0x10, +1, -2, +1, +0, 1, (signed char)0x80, +1, -1, +1, +1, 1, (signed char)0x88,
  +1, +0, +1, +2, 1, 0x08, +1, +0, +2, -1, 1, 0x40, +1, +0, +2, +1, 1, 0x10
};
const signed char k_chood[] = { -1, -1, -1, 0, -1, +1, 0, +1, +1, +1, +1, 0, +1, -1, 0, -1 };

// the per-phase code of vng.c:86-129, phases = (row mod 8, column mod 2) of the SENSOR
void build_codes(const uint32_t filters4, vng_code codes[16])
{
  for(int row = 0; row < 8; row++)
    for(int col = 0; col < 2; col++)
    {
      vng_code &c = codes[row * 2 + col];
      memset(&c, 0, sizeof(c));
      const signed char *cp = k_terms;
      for(int t = 0; t < 64; t++)
      {
        const int y1 = *cp++, x1 = *cp++, y2 = *cp++, x2 = *cp++, weight = *cp++, grads = (unsigned char)*cp++;
        const int color = fc(row + y1, col + x1, filters4);
        if(fc(row + y2, col + x2, filters4) != color) continue;
        const int diag = (fc(row, col + 1, filters4) == color && fc(row + 1, col, filters4) == color) ? 2 : 1;
        if(abs(y1 - y2) == diag && abs(x1 - x2) == diag) continue;
        vng_term &tm = c.terms[c.nterms++];
        tm.y1 = (signed char)y1;
        tm.x1 = (signed char)x1;
        tm.y2 = (signed char)y2;
        tm.x2 = (signed char)x2;
        tm.color = (unsigned char)color;
        tm.weight = (unsigned char)weight;
        tm.grads = (unsigned char)grads;
      }
      cp = k_chood;
      for(int g = 0; g < 8; g++)
      {
        const int y = *cp++, x = *cp++;
        const int color = fc(row, col, filters4);
        c.hood[g].y = (signed char)y;
        c.hood[g].x = (signed char)x;
        c.hood[g].far = (fc(row + y, col + x, filters4) != color && fc(row + y * 2, col + x * 2, filters4) == color) ? 1 : 0;
      }
    }
}

} // namespace

namespace ansel
{

// vng_interpolate(out, in, ..., only_vng_linear = FALSE) for a Bayer mosaic: `filters` the word of the SENSOR, the roi
// origin added to every coordinate as the reference does (fcol(row + roi_in->y, col + roi_in->x, ...))
int vng4_demosaic_launch(int devid, const dt_hip_piece_t *piece, const float *in, float4 *out)
{
  const int w = piece->roi_in.width, h = piece->roi_in.height, rx = piece->roi_in.x, ry = piece->roi_in.y;
  const uint32_t filters4 = filters4_of(piece->filters);
  vng_code host_codes[16];
  build_codes(filters4, host_codes);
  vng_code *codes = (vng_code *)dt_hip_alloc_device_buffer(devid, sizeof(host_codes));
  float4 *lin = (float4 *)dt_hip_alloc_device_buffer(devid, (size_t)w * h * sizeof(float4));
  int err = (codes && lin) ? DT_HIP_SUCCESS : DT_HIP_SYSMEM_ALLOCATION;
  hipStream_t s = stream_of(devid);
  if(err == DT_HIP_SUCCESS
     && (hipMemcpyAsync(codes, host_codes, sizeof(host_codes), hipMemcpyHostToDevice, s) != hipSuccess
         || hipStreamSynchronize(s) != hipSuccess)) // host_codes lives on this stack
    err = DT_HIP_DEFAULT_ERROR;
  if(err == DT_HIP_SUCCESS)
  {
    {
      launch_scope ls(devid, "vng_lin");
      vng_lin<<<dim3((w + 63) / 64, (h + 3) / 4), 256, 0, s>>>(in, lin, w, h, rx, ry, filters4);
    }
    {
      launch_scope ls(devid, "vng_main");
      vng_main<<<dim3(((w + 1) / 2 + 63) / 64, (h + 3) / 4, 2), 256, 0, s>>>(lin, out, w, h, rx, ry, filters4, codes);
    }
    err = check_launch("vng4");
  }
  if(codes) dt_hip_release_mem_object(codes);
  if(lin) dt_hip_release_mem_object(lin);
  return err;
}

// dual_demosaic(), dual.c:40-110: rgb = the high-frequency interpolation going in, the blend coming out; raw = the mosaic
// as the module received it
int dual_demosaic_launch(int devid, const dt_hip_piece_t *piece, const float *raw, float4 *rgb, const float dual_threshold,
                         const float wb[4])
{
  const int w = piece->roi_in.width, h = piece->roi_in.height;
  if(w < 16 || h < 16) return DT_HIP_SUCCESS;       // :47
  if(!(dual_threshold > 0.0f)) return DT_HIP_SUCCESS; // :50 (`dual_threshold <= 0.0f`; a NaN threshold blends in the reference:
                                                      // not a state commit_params() leaves, refused by the caller)
  const size_t n = (size_t)w * h;
  float4 *vng = (float4 *)dt_hip_alloc_device_buffer(devid, n * sizeof(float4));
  float *mask = (float *)dt_hip_alloc_device_buffer(devid, n * sizeof(float));
  int err = (vng && mask) ? DT_HIP_SUCCESS : DT_HIP_SYSMEM_ALLOCATION;
  if(err == DT_HIP_SUCCESS) err = vng4_demosaic_launch(devid, piece, raw, vng);
  if(err == DT_HIP_SUCCESS) err = color_smoothing_launch(devid, vng, w, h, 2);
  const float contrastf = 0.005f * powf(dual_threshold, 1.1f); // slider2contrast(), :35-38
  if(err == DT_HIP_SUCCESS) err = dual_blend_mask_launch(devid, rgb, wb, contrastf, w, h, mask);
  if(err == DT_HIP_SUCCESS)
  {
    launch_scope ls(devid, "dual_blend");
    dual_blend<<<pixel_grid(n), 256, 0, stream_of(devid)>>>(rgb, vng, mask, n);
    err = check_launch("dual_blend");
  }
  if(vng) dt_hip_release_mem_object(vng);
  if(mask) dt_hip_release_mem_object(mask);
  return err;
}

} // namespace ansel
