// locallaplacian.hip -- local contrast, local-laplacian mode (the module's default), on gfx950.
//
// Reference: local_laplacian_internal(), src/pixel/locallaplacian.c:354-563, regular mode (an export
// has no preview boundary), called from process(), src/iop/bilat.c:352-357.  (OpenCL twin:
// locallaplaciancl.c; same stages.)
//
// L / 100 is padded by 2^top_level on every side; six copies go through the tone curve centred on six
// grey levels and become Gaussian pyramids; the output pyramid is assembled coarse to fine from the
// Laplacian coefficients of the two curves bracketing the local grey level.  Every stage is a pure
// per-pixel function of the planes written by earlier stages -- the reference's "compute the interior,
// then copy the border rows and columns" steps are clamped reads here -- so each stage is one launch,
// the six curves run in one launch (blockIdx.z), and nothing returns to the host.
#include "hip_common.h"
#include "devmath.h"

#include <math.h>

using namespace ansel;

namespace
{

#define LL_MAX_LEVELS 30
#define LL_NUM_GAMMA 6

inline int dl(int size, const int level)
{
  for(int l = 0; l < level; l++) size = (size - 1) / 2 + 1;
  return size;
}
__device__ __forceinline__ float clampf(const float v, const float lo, const float hi) { return v > lo ? (v < hi ? v : hi) : lo; }
__device__ __forceinline__ int clampi(const int v, const int lo, const int hi) { return v > lo ? (v < hi ? v : hi) : lo; }

// ll_expand_gaussian(), locallaplacian.c:80-118; binary64 where the reference has double literals
__device__ __forceinline__ float expand_at(const float *__restrict__ c, const int i, const int j, const int wd)
{
  const int half_w = (wd - 1) / 2 + 1;
  const int at = (j / 2) * half_w + i / 2;
  switch((i & 1) + 2 * (j & 1))
  {
    case 0:
      return (float)(4. / 256.
                     * (double)(6.0f * (c[at - half_w] + c[at - 1] + 6.0f * c[at] + c[at + 1] + c[at + half_w]) + c[at - half_w - 1]
                                + c[at - half_w + 1] + c[at + half_w - 1] + c[at + half_w + 1]));
    case 1:
      return (float)(4. / 256.
                     * (24.0 * (double)(c[at] + c[at + 1])
                        + 4.0 * (double)(c[at - half_w] + c[at - half_w + 1] + c[at + half_w] + c[at + half_w + 1])));
    case 2:
      return (float)(4. / 256.
                     * (24.0 * (double)(c[at] + c[at + half_w])
                        + 4.0 * (double)(c[at - 1] + c[at + 1] + c[at + half_w - 1] + c[at + half_w + 1])));
    default: return .25f * (c[at] + c[at + 1] + c[at + half_w] + c[at + half_w + 1]);
  }
}

// ll_pad_input(), replication branch, :262-273
__global__ __launch_bounds__(256) void ll_pad(const float4 *__restrict__ in, float *__restrict__ padded, const int wd,
                                              const int ht, const int w, const int h, const int support)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= (size_t)w * h) return;
  const int j = (int)(k / w), i = (int)(k - (size_t)j * w);
  const int sj = clampi(j - support, 0, ht - 1), si = clampi(i - support, 0, wd - 1);
  padded[k] = in[(size_t)sj * wd + si].x * 0.01f;
}

// curve_scalar(), :295-325
__device__ __forceinline__ float curve(const float x, const float g, const float sigma, const float shadows,
                                       const float highlights, const float clarity)
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
  val += clarity * c * ansel_math::expf_exact((float)((double)(-c * c) / (2.0 * (double)sigma * (double)sigma / (double)3.0f)));
  return val;
}

struct planes6
{
  float *p[LL_NUM_GAMMA];
};

// apply_curve(), :328-351, for the six grey levels (blockIdx.z)
__global__ __launch_bounds__(256) void ll_curve(const float *__restrict__ padded, const planes6 out, const int w, const int h,
                                                const int support, const float sigma, const float shadows,
                                                const float highlights, const float clarity)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= (size_t)w * h) return;
  const int j = (int)(k / w), i = (int)(k - (size_t)j * w);
  const int sj = clampi(j, support, h - support - 1), si = clampi(i, support, w - support - 1);
  const float g = ((float)blockIdx.z + .5f) / (float)LL_NUM_GAMMA;
  out.p[blockIdx.z][k] = curve(padded[(size_t)sj * w + si], g, sigma, shadows, highlights, clarity);
}

// gauss_reduce() + ll_fill_boundary1(), :173-200, :121-131; blockIdx.z selects one of up to six planes
__global__ __launch_bounds__(256) void ll_reduce(const planes6 in, const planes6 coarse, const int wd, const int ht)
{
  const int half_w = (wd - 1) / 2 + 1, half_h = (ht - 1) / 2 + 1;
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= (size_t)half_w * half_h) return;
  const int jj0 = (int)(k / half_w), ii0 = (int)(k - (size_t)jj0 * half_w);
  const int j = clampi(jj0, 1, half_h - 2), i = clampi(ii0, 1, half_w - 2);
  const float *__restrict__ src = in.p[blockIdx.z];
  float acc = 0.0f;
#pragma unroll
  for(int jj = -2; jj <= 2; jj++)
  {
    const float wj = (jj == -2 || jj == 2) ? 1.f / 16.f : (jj == 0 ? 6.f / 16.f : 4.f / 16.f);
#pragma unroll
    for(int ii = -2; ii <= 2; ii++)
    {
      const float wi = (ii == -2 || ii == 2) ? 1.f / 16.f : (ii == 0 ? 6.f / 16.f : 4.f / 16.f);
      acc += src[(size_t)(2 * j + jj) * wd + 2 * i + ii] * wi * wj;
    }
  }
  coarse.p[blockIdx.z][k] = acc;
}

struct assemble_args
{
  const float *padded;     // level l of the input pyramid
  const float *coarse_out; // output[l + 1]
  float *out;              // output[l]
  const float *fine[LL_NUM_GAMMA], *coarse[LL_NUM_GAMMA];
  int pw, ph;
};

// gauss_expand() + ll_fill_boundary2() + the coefficient blend, :499-523
__global__ __launch_bounds__(256) void ll_assemble(const assemble_args a)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= (size_t)a.pw * a.ph) return;
  const int j = (int)(k / a.pw), i = (int)(k - (size_t)j * a.pw);
  const int ci = clampi(i, 1, ((a.pw - 1) & ~1) - 1), cj = clampi(j, 1, ((a.ph - 1) & ~1) - 1);
  const float base = expand_at(a.coarse_out, ci, cj, a.pw);
  const float v = a.padded[k];
  int hi = 1;
  for(; hi < LL_NUM_GAMMA - 1 && ((float)hi + .5f) / (float)LL_NUM_GAMMA <= v; hi++)
    ;
  const int lo = hi - 1;
  const float glo = ((float)lo + .5f) / (float)LL_NUM_GAMMA, ghi = ((float)hi + .5f) / (float)LL_NUM_GAMMA;
  const float t = clampf((v - glo) / (ghi - glo), 0.0f, 1.0f);
  float l0 = 0.f, l1 = 0.f;
#pragma unroll
  for(int g = 0; g < LL_NUM_GAMMA; g++)
  {
    if(g == lo) l0 = a.fine[g][k] - expand_at(a.coarse[g], ci, cj, a.pw);
    if(g == hi) l1 = a.fine[g][k] - expand_at(a.coarse[g], ci, cj, a.pw);
  }
  a.out[k] = base + (l0 * (1.0f - t) + l1 * t);
}

// :524-530 (alpha is not written)
__global__ __launch_bounds__(256) void ll_finish(const float4 *__restrict__ in, float *__restrict__ out,
                                                 const float *__restrict__ level0, const int wd, const int ht, const int w,
                                                 const int support)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= (size_t)wd * ht) return;
  const int j = (int)(k / wd), i = (int)(k - (size_t)j * wd);
  const float4 px = in[k];
  out[4 * k + 0] = 100.0f * level0[(size_t)(j + support) * w + support + i];
  out[4 * k + 1] = px.y;
  out[4 * k + 2] = px.z;
}

} // namespace

namespace ansel
{

int local_laplacian_launch(int devid, const float4 *in, float4 *out, int wd, int ht, float sigma, float shadows,
                           float highlights, float clarity)
{
  if(wd <= 1 || ht <= 1) return DT_HIP_SUCCESS; // locallaplacian.c:366
  const int m = wd < ht ? wd : ht;
  const int nl = 31 - __builtin_clz((unsigned)m);
  const int num_levels = nl < LL_MAX_LEVELS ? nl : LL_MAX_LEVELS;
  const int top_level = num_levels - 1;
  if(top_level < 1)
  {
    set_last_error("local laplacian: frame too small for a pyramid");
    return DT_HIP_INVALID_ARG;
  }
  const int support = 1 << top_level;
  const int w = 2 * support + wd, h = 2 * support + ht;
  float *padded[LL_MAX_LEVELS] = { nullptr }, *output[LL_MAX_LEVELS] = { nullptr }, *buf[LL_NUM_GAMMA][LL_MAX_LEVELS] = { { nullptr } };
  bool ok = true;
  for(int l = 0; l <= top_level; l++)
  {
    const size_t bytes = (size_t)dl(w, l) * dl(h, l) * sizeof(float);
    if(l < top_level) ok &= (padded[l] = (float *)dt_hip_alloc_device_buffer(devid, bytes)) != nullptr;
    ok &= (output[l] = (float *)dt_hip_alloc_device_buffer(devid, bytes)) != nullptr;
    for(int k = 0; k < LL_NUM_GAMMA; k++) ok &= (buf[k][l] = (float *)dt_hip_alloc_device_buffer(devid, bytes)) != nullptr;
  }
  int err = ok ? DT_HIP_SUCCESS : DT_HIP_SYSMEM_ALLOCATION;
  hipStream_t s = stream_of(devid);
  if(err == DT_HIP_SUCCESS)
  {
    {
      launch_scope ls(devid, "ll_pad");
      ll_pad<<<pixel_grid((size_t)w * h), 256, 0, s>>>(in, padded[0], wd, ht, w, h, support);
    }
    {
      // Gaussian pyramid of the padded input; its coarsest level seeds the output pyramid, :405-407
      launch_scope ls(devid, "ll_reduce");
      for(int l = 1; l <= top_level; l++)
      {
        planes6 src = { { padded[l - 1] } }, dst = { { l < top_level ? padded[l] : output[top_level] } };
        const size_t n = (size_t)dl(w, l) * dl(h, l);
        ll_reduce<<<dim3(pixel_grid(n), 1, 1), 256, 0, s>>>(src, dst, dl(w, l - 1), dl(h, l - 1));
      }
    }
    {
      launch_scope ls(devid, "ll_curve");
      planes6 dst;
      for(int k = 0; k < LL_NUM_GAMMA; k++) dst.p[k] = buf[k][0];
      ll_curve<<<dim3(pixel_grid((size_t)w * h), 1, LL_NUM_GAMMA), 256, 0, s>>>(padded[0], dst, w, h, support, sigma, shadows,
                                                                              highlights, clarity);
    }
    {
      launch_scope ls(devid, "ll_reduce");
      for(int l = 1; l <= top_level; l++)
      {
        planes6 src, dst;
        for(int k = 0; k < LL_NUM_GAMMA; k++)
        {
          src.p[k] = buf[k][l - 1];
          dst.p[k] = buf[k][l];
        }
        const size_t n = (size_t)dl(w, l) * dl(h, l);
        ll_reduce<<<dim3(pixel_grid(n), 1, LL_NUM_GAMMA), 256, 0, s>>>(src, dst, dl(w, l - 1), dl(h, l - 1));
      }
    }
    {
      launch_scope ls(devid, "ll_assemble");
      for(int l = top_level - 1; l >= 0; l--)
      {
        assemble_args a;
        a.padded = padded[l];
        a.coarse_out = output[l + 1];
        a.out = output[l];
        for(int k = 0; k < LL_NUM_GAMMA; k++)
        {
          a.fine[k] = buf[k][l];
          a.coarse[k] = buf[k][l + 1];
        }
        a.pw = dl(w, l);
        a.ph = dl(h, l);
        ll_assemble<<<pixel_grid((size_t)a.pw * a.ph), 256, 0, s>>>(a);
      }
    }
    {
      launch_scope ls(devid, "ll_finish");
      ll_finish<<<pixel_grid((size_t)wd * ht), 256, 0, s>>>(in, (float *)out, output[0], wd, ht, w, support);
    }
    err = check_launch("local laplacian");
  }
  for(int l = 0; l <= top_level; l++)
  {
    if(padded[l]) dt_hip_release_mem_object(padded[l]);
    if(output[l]) dt_hip_release_mem_object(output[l]);
    for(int k = 0; k < LL_NUM_GAMMA; k++)
      if(buf[k][l]) dt_hip_release_mem_object(buf[k][l]);
  }
  return err;
}

} // namespace ansel
