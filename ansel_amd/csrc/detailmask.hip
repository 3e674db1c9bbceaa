// detailmask.hip -- the detail masks (src/develop/masks/detail.c) on gfx950.
//
//   dt_masks_calc_rawdetail_mask()  :283-316  the hidden "detailmask" stage behind demosaic (src/iop/detailmask.c:111-150):
//                                             Scharr gradient magnitude of sqrt(mean of the white-balance-normalised RGB)
//   dt_masks_calc_detail_mask()     :325-335  sigmoid around the blend's details threshold + dt_masks_blur_9x9() :224-243:
//                                             what _refine_with_detail_mask() (src/develop/blend.c:361-425) multiplies a
//                                             form mask with
//   dt_masks_extend_border()        :96-123   a border pixel takes the value of the nearest pixel of the interior, so both
//                                             stencils run at the clamped position and no border pass exists
//
// Two launches per use: a one-float plane (the luminance resp. the sigmoid) and the stencil over it, read through L1/L2
// (4 B/px planes: 16 + 4 + 4 + 4 B/px for the stage, 4 + 4 + 4 + 4 B/px for the refinement).
#include "hip_common.h"

#include <math.h>

using namespace ansel;

namespace
{

// dt_fast_expf(), src/math/math.h:254-267.  The float -> int conversion of an out-of-range or NaN value is INT_MIN on the
// reference's target (cvttss2si), which its k0 > 0 test turns into 0.
__device__ __forceinline__ float fast_expf(const float x)
{
  const float t = 1065353216.0f + x * 11401300.0f;
  int k = (t > -2147483648.0f && t < 2147483648.0f) ? (int)t : 0;
  k = k > 0 ? k : 0;
  return __int_as_float(k);
}

__device__ __forceinline__ int inner(const int v, const int border, const int size)
{
  return v < border ? border : (v > size - border - 1 ? size - border - 1 : v);
}

// in -> out (the stage copies its input) and sqrt(Y0) of the output into `lum`
__global__ __launch_bounds__(256) void rawdetail_luminance(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                           float *__restrict__ lum, const size_t n, const float wb0,
                                                           const float wb1, const float wb2)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= n) return;
  const float4 p = in[k];
  out[k] = p;
  const float val = 0.333333333f * (fmaxf(p.x, 0.0f) / wb0 + fmaxf(p.y, 0.0f) / wb1 + fmaxf(p.z, 0.0f) / wb2);
  lum[k] = sqrtf(val);
}

__global__ __launch_bounds__(256) void rawdetail_scharr(const float *__restrict__ tmp, float *__restrict__ mask, const int width,
                                                        const int height)
{
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), row = blockIdx.y * 4 + (threadIdx.x >> 6);
  if(col >= width || row >= height) return;
  const size_t w = (size_t)width, idx = (size_t)inner(row, 1, height) * w + inner(col, 1, width);
  const float gx = 47.0f * (tmp[idx - w - 1] - tmp[idx - w + 1]) + 162.0f * (tmp[idx - 1] - tmp[idx + 1])
                   + 47.0f * (tmp[idx + w - 1] - tmp[idx + w + 1]);
  const float gy = 47.0f * (tmp[idx - w - 1] - tmp[idx + w - 1]) + 162.0f * (tmp[idx - w] - tmp[idx + w])
                   + 47.0f * (tmp[idx - w + 1] - tmp[idx + w + 1]);
  const float a = gx / 256.0f, b = gy / 256.0f;
  mask[(size_t)row * w + col] = (1.0f / 16.0f) * sqrtf(a * a + b * b);
}

// calcBlendFactor(), detail.c:317-323
__global__ __launch_bounds__(256) void detail_sigmoid(const float *__restrict__ rm, float *__restrict__ tmp, const size_t n,
                                                      const float threshold, const int detail)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= n) return;
  const float blend = 1.0f / (1.0f + fast_expf(16.0f - (16.0f / threshold) * rm[k]));
  tmp[k] = detail ? blend : 1.0f - blend;
}

struct blur_args
{
  float c[13]; // dt_masks_blur_9x9_coeff(): by (|dy|, |dx|) 00 10 11 20 21 22 30 31 32 33 40 41 42
};

// The 9 x 9 blur of dt_masks_blur_9x9() (detail.c:224-243) at the clamped position, clipped to [0, 1], times the form
// mask (or `fill`): the refined form mask of _refine_with_detail_mask().  The kernel has thirteen distinct weights, one
// per ring of samples at the same (|dy|, |dx|); the reference adds the samples of a ring in a fixed order, multiplies the
// sum by the ring's weight and adds the thirteen products from the widest ring inwards.  RING[] lists the rings in that
// order, each with its samples' (dy, dx) in the order they are added; the loops unroll into straight-line code.
struct ring_t
{
  int weight, n;       // index into blur_args::c, number of samples
  signed char at[8][2]; // (dy, dx)
};
__device__ constexpr ring_t RING[13] = {
  { 12, 8, { { -4, -2 }, { -4, 2 }, { -2, -4 }, { -2, 4 }, { 2, -4 }, { 2, 4 }, { 4, -2 }, { 4, 2 } } },
  { 11, 8, { { -4, -1 }, { -4, 1 }, { -1, -4 }, { -1, 4 }, { 1, -4 }, { 1, 4 }, { 4, -1 }, { 4, 1 } } },
  { 10, 4, { { -4, 0 }, { 0, -4 }, { 0, 4 }, { 4, 0 } } },
  { 9, 4, { { -3, -3 }, { -3, 3 }, { 3, -3 }, { 3, 3 } } },
  { 8, 8, { { -3, -2 }, { -3, 2 }, { -2, -3 }, { -2, 3 }, { 2, -3 }, { 2, 3 }, { 3, -2 }, { 3, 2 } } },
  { 7, 8, { { -3, -1 }, { -3, 1 }, { -1, -3 }, { -1, 3 }, { 1, -3 }, { 1, 3 }, { 3, -1 }, { 3, 1 } } },
  { 6, 4, { { -3, 0 }, { 0, -3 }, { 0, 3 }, { 3, 0 } } },
  { 5, 4, { { -2, -2 }, { -2, 2 }, { 2, -2 }, { 2, 2 } } },
  { 4, 8, { { -2, -1 }, { -2, 1 }, { -1, -2 }, { -1, 2 }, { 1, -2 }, { 1, 2 }, { 2, -1 }, { 2, 1 } } },
  { 3, 4, { { -2, 0 }, { 0, -2 }, { 0, 2 }, { 2, 0 } } },
  { 2, 4, { { -1, -1 }, { -1, 1 }, { 1, -1 }, { 1, 1 } } },
  { 1, 4, { { -1, 0 }, { 0, -1 }, { 0, 1 }, { 1, 0 } } },
  { 0, 1, { { 0, 0 } } },
};

__global__ __launch_bounds__(256) void detail_refine(const float *__restrict__ src, const float *__restrict__ form,
                                                     float *__restrict__ out, const int width, const int height, const float fill,
                                                     const blur_args a)
{
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), row = blockIdx.y * 4 + (threadIdx.x >> 6);
  if(col >= width || row >= height) return;
  const float *const s = src + (size_t)inner(row, 4, height) * width + inner(col, 4, width);
  float v = 0.0f;
#pragma unroll
  for(int k = 0; k < 13; k++)
  {
    float ring = s[RING[k].at[0][0] * width + RING[k].at[0][1]];
#pragma unroll
    for(int j = 1; j < RING[k].n; j++) ring = ring + s[RING[k].at[j][0] * width + RING[k].at[j][1]];
    const float term = a.c[RING[k].weight] * ring;
    v = k ? v + term : term;
  }
  const float lum = fminf(1.0f, fmaxf(0.0f, v));
  const size_t k = (size_t)row * width + col;
  out[k] = (form ? form[k] : fill) * lum;
}

// dt_masks_blur_9x9_coeff(), detail.c:159-196 (binary32, the host's expf as in the reference)
void blur_9x9_coeff(float c[13], const float sigma)
{
  float kernel[9][9];
  const float temp = -2.0f * (sigma * sigma);
  const float range = (3.0f * 1.5f) * (3.0f * 1.5f);
  float sum = 0.0f;
  for(int k = -4; k <= 4; k++)
    for(int j = -4; j <= 4; j++)
    {
      const float d2 = (float)k * (float)k + (float)j * (float)j;
      kernel[k + 4][j + 4] = d2 <= range ? expf(d2 / temp) : 0.0f;
      if(d2 <= range) sum += kernel[k + 4][j + 4];
    }
  for(int i = 0; i < 9; i++)
    for(int j = 0; j < 9; j++) kernel[i][j] /= sum;
  static const int at[13][2] = { { 4, 4 }, { 3, 4 }, { 3, 3 }, { 2, 4 }, { 2, 3 }, { 2, 2 }, { 1, 4 },
                                 { 1, 3 }, { 1, 2 }, { 1, 1 }, { 0, 4 }, { 0, 3 }, { 0, 2 } };
  for(int k = 0; k < 13; k++) c[k] = kernel[at[k][0]][at[k][1]];
}

} // namespace

namespace ansel
{

// _refine_with_detail_mask(), blend.c:361-425: refined = (form or fill) * blur(sigmoid(raw detail mask)); all planes
// width x height floats, `refined` the caller's
int detail_refine_launch(int devid, const float *rawdetail, const float *form, float fill, float level, int width, int height,
                         float *refined)
{
  if(width < 9 || height < 9)
  {
    set_last_error("blend: the details threshold needs a frame of at least 9 x 9 (the 9 x 9 blur of the detail mask)");
    return DT_HIP_INVALID_ARG;
  }
  const size_t n = (size_t)width * height;
  float *tmp = (float *)dt_hip_alloc_device_buffer(devid, n * sizeof(float));
  if(!tmp) return DT_HIP_SYSMEM_ALLOCATION;
  const int detail = level > 0.0f;
  // _detail_mask_threshold(), blend.c:355-359
  const float threshold = 0.005f * (detail ? powf(level, 2.0f) : 1.0f - powf(fabs(level), 0.5f));
  blur_args a;
  blur_9x9_coeff(a.c, 2.0f);
  hipStream_t s = stream_of(devid);
  {
    launch_scope ls(devid, "detail_sigmoid");
    detail_sigmoid<<<pixel_grid(n), 256, 0, s>>>(rawdetail, tmp, n, threshold, detail);
  }
  {
    launch_scope ls(devid, "detail_refine");
    detail_refine<<<dim3((width + 63) / 64, (height + 3) / 4), 256, 0, s>>>(tmp, form, refined, width, height, fill, a);
  }
  dt_hip_release_mem_object(tmp);
  return check_launch("detail_refine");
}

// The blend mask of the dual demosaic (dual.c:83-85): dt_masks_calc_rawdetail_mask() of the high-frequency image with the
// white-balance coefficients of the buffer, then dt_masks_calc_detail_mask(..., threshold, detail = TRUE) in place.
int dual_blend_mask_launch(int devid, float4 *rgb, const float wb[3], float threshold, int width, int height, float *mask)
{
  if(width < 9 || height < 9) return DT_HIP_INVALID_ARG;
  const size_t n = (size_t)width * height;
  float *lum = (float *)dt_hip_alloc_device_buffer(devid, n * sizeof(float));
  float *raw = (float *)dt_hip_alloc_device_buffer(devid, n * sizeof(float));
  if(!lum || !raw)
  {
    if(lum) dt_hip_release_mem_object(lum);
    if(raw) dt_hip_release_mem_object(raw);
    return DT_HIP_SYSMEM_ALLOCATION;
  }
  blur_args a;
  blur_9x9_coeff(a.c, 2.0f);
  hipStream_t s = stream_of(devid);
  {
    launch_scope ls(devid, "rawdetail_luminance");
    rawdetail_luminance<<<pixel_grid(n), 256, 0, s>>>(rgb, rgb, lum, n, wb[0], wb[1], wb[2]); // the copy is onto itself
  }
  {
    launch_scope ls(devid, "rawdetail_scharr");
    rawdetail_scharr<<<dim3((width + 63) / 64, (height + 3) / 4), 256, 0, s>>>(lum, raw, width, height);
  }
  {
    launch_scope ls(devid, "detail_sigmoid");
    detail_sigmoid<<<pixel_grid(n), 256, 0, s>>>(raw, lum, n, threshold, 1);
  }
  {
    launch_scope ls(devid, "detail_refine");
    detail_refine<<<dim3((width + 63) / 64, (height + 3) / 4), 256, 0, s>>>(lum, nullptr, mask, width, height, 1.0f, a);
  }
  dt_hip_release_mem_object(lum);
  dt_hip_release_mem_object(raw);
  return check_launch("dual_blend_mask");
}

} // namespace ansel

extern "C" int dt_hip_iop_detailmask_process(int devid, const dt_hip_piece_t *piece, const dt_hip_detailmask_data_t *d,
                                             dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out || !d->mask || piece->channels != 4) return DT_HIP_INVALID_ARG;
  const int width = piece->roi_out.width, height = piece->roi_out.height;
  if(width < 3 || height < 3)
  {
    set_last_error("detailmask: the Scharr operator needs a frame of at least 3 x 3");
    return DT_HIP_INVALID_ARG;
  }
  const size_t n = (size_t)width * height;
  float *lum = (float *)dt_hip_alloc_device_buffer(devid, n * sizeof(float));
  if(!lum) return DT_HIP_SYSMEM_ALLOCATION;
  hipStream_t s = stream_of(devid);
  {
    launch_scope ls(devid, "rawdetail_luminance");
    rawdetail_luminance<<<pixel_grid(n), 256, 0, s>>>((const float4 *)dev_in, (float4 *)dev_out, lum, n, d->wb[0], d->wb[1], d->wb[2]);
  }
  {
    launch_scope ls(devid, "rawdetail_scharr");
    rawdetail_scharr<<<dim3((width + 63) / 64, (height + 3) / 4), 256, 0, s>>>(lum, (float *)d->mask, width, height);
  }
  dt_hip_release_mem_object(lum);
  return check_launch("detailmask");
}
