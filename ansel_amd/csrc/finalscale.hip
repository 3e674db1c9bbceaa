// finalscale.hip -- the export resampler on gfx950.
//
// Reference: process(), src/iop/finalscale.c:117-131 -> dt_iop_clip_and_zoom_roi()
// (src/develop/imageop_math.c:146-152) -> _interpolation_resample_plain(), src/pixel/interpolation.c:898-1030
// (OpenCL twin: dt_interpolation_resample_cl(), :1105-1277, which sums the taps in a different order).
// The host builds the two 1-D resampling plans exactly as _prepare_resampling_plan() (:711-895) does --
// kernels bilinear / bicubic / Mitchell (:175-296), BORDER_REPLICATE -- and uploads them; one thread
// per output pixel then accumulates  sum_y ( sum_x in * htap ) * vtap  in the reference's order.
#include "hip_common.h"

#include <math.h>
#include <vector>

using namespace ansel;

namespace
{

struct plan_dev
{
  const int *length, *start, *index;
  const float *kernel;
};

__device__ __forceinline__ float max_zero(const float v) { return isfinite(v) ? (v > 0.0f ? v : 0.0f) : 0.0f; }

__global__ __launch_bounds__(256) void resample(const float4 *__restrict__ in, float4 *__restrict__ out, const int iw,
                                                const int ow, const int oh, const plan_dev h, const plan_dev v)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= (size_t)ow * oh) return;
  const int oy = (int)(k / ow), ox = (int)(k - (size_t)oy * ow);
  const int hl = h.length[ox], hs = h.start[ox], vl = v.length[oy], vs0 = v.start[oy];
  float4 vs = make_float4(0.f, 0.f, 0.f, 0.f);
  for(int iy = 0; iy < vl; iy++)
  {
    const float4 *const row = in + (size_t)v.index[vs0 + iy] * iw;
    float4 vhs = make_float4(0.f, 0.f, 0.f, 0.f);
    for(int ix = 0; ix < hl; ix++)
    {
      const float4 px = row[h.index[hs + ix]];
      const float htap = h.kernel[hs + ix];
      vhs.x += px.x * htap;
      vhs.y += px.y * htap;
      vhs.z += px.z * htap;
      vhs.w += px.w * htap;
    }
    const float vtap = v.kernel[vs0 + iy];
    vs.x += vhs.x * vtap;
    vs.y += vhs.y * vtap;
    vs.z += vhs.z * vtap;
    vs.w += vhs.w * vtap;
  }
  nt_store(out + k, make_float4(max_zero(vs.x), max_zero(vs.y), max_zero(vs.z), max_zero(vs.w)));
}

// ---- host: the resampling plan ---------------------------------------------------------------

// ceil_fast(), src/math/math.h:324-334 (an exact positive integer maps to x + 1)
inline float ceil_fast(const float x) { return x <= 0.f ? (float)(int)x : -((float)(int)-x) + 1.f; }

const int k_half_width[3] = { 1, 2, 2 }; // dt_interpolator[], interpolation.c:298-314

float tap_value(const int kind, const float vt)
{
  const float a = fabsf(vt);
  if(kind == DT_HIP_INTERPOLATION_BILINEAR) return 1.0f - a; // :175-194
  if(kind == DT_HIP_INTERPOLATION_BICUBIC)                   // :200-232
  {
    const float t2 = vt * vt, t5 = 5.0f * a;
    const float r12 = (a * (t5 - 8.0f - t2) + 4.0f) * 0.5f;
    const float r01 = ((3.0f * t2 - t5) * a + 2.0f) * 0.5f;
    return a <= 1.0f ? r01 : r12;
  }
  const float a2 = a * a, a3 = a2 * a; // Mitchell-Netravali B = C = 1/3, :253-296
  const float r01 = (7.0f / 6.0f) * a3 - 2.0f * a2 + (8.0f / 9.0f);
  const float r12 = 2.0f * a2 - (7.0f / 18.0f) * a3 - (10.0f / 3.0f) * a + (16.0f / 9.0f);
  return a <= 1.0f ? r01 : r12;
}

// the 4-wide loops of _maketaps_*(): the tap position advances by a float accumulation
void make_taps(const int kind, float *taps, const int num_taps, const float first_tap, const float interval)
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

struct plan_host
{
  std::vector<int> length, start, index;
  std::vector<float> kernel;
};

// _prepare_resampling_plan(), interpolation.c:711-895 (finalscale: x0 = 0 on both sides, finalscale.c:123-127)
void plan_build(plan_host &p, const int kind, const int in, const int in_x0, const int out, const int out_x0, const float scale)
{
  const int w = k_half_width[kind];
  const int maxtaps = scale > 1.f ? 2 * w : (int)ceil_fast((float)2 * (float)w / scale);
  std::vector<float> scratch(maxtaps + 8);
  p.length.resize(out);
  p.start.resize(out);
  for(int x = 0; x < out; x++)
  {
    int first, taps;
    if(scale > 1.f)
    {
      const float fx = (float)(out_x0 + x) / scale - in_x0; // _compute_upsampling_kernel(), :320-344
      first = (int)floorf(fx) - w + 1;
      taps = 2 * w;
      make_taps(kind, scratch.data(), taps, fx - (float)first, -1.0f);
    }
    else
    {
      // _compute_downsampling_kernel(.., out_x0 + x), :354-392: the first tap is an absolute input index (in_x0 is not
      // subtracted, :853)
      const int xo = out_x0 + x;
      const float xin = ceil_fast(((float)xo - (float)w) / scale);
      first = (int)xin;
      const float t = xin * scale - (float)xo;
      taps = (int)(((float)w - t) / scale);
      if(taps > maxtaps + 4) taps = maxtaps + 4;
      make_taps(kind, scratch.data(), taps, t, scale);
    }
    p.length[x] = taps;
    p.start[x] = (int)p.kernel.size();
    float norm = 0.f;
    for(int t = 0; t < taps; t++) norm += scratch[t];
    norm = 1.f / norm;
    for(int t = 0; t < taps; t++)
    {
      p.kernel.push_back(scratch[t] * norm);
      const int i = first + t;
      p.index.push_back(i < 0 ? 0 : (i > in - 1 ? in - 1 : i)); // BORDER_REPLICATE, :62, :95-104
    }
  }
}

// dt_interpolation_resample(), interpolation.c:897-1044, on RGBA pixels; origins: the regions' x / y count (initialscale)
int resample_launch(int devid, const dt_hip_piece_t *piece, const int kind, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out,
                    const bool origins, const char *tag)
{
  const int iw = piece->roi_in.width, ih = piece->roi_in.height, ow = piece->roi_out.width, oh = piece->roi_out.height;
  if(iw <= 0 || ih <= 0 || ow <= 0 || oh <= 0) return DT_HIP_SUCCESS;
  const int ix0 = origins ? piece->roi_in.x : 0, iy0 = origins ? piece->roi_in.y : 0;
  const int ox0 = origins ? piece->roi_out.x : 0, oy0 = origins ? piece->roi_out.y : 0;
  const float so = (float)piece->roi_out.scale, si = (float)piece->roi_in.scale;
  if(!(so > 0.f) || !(si > 0.f)) return DT_HIP_INVALID_ARG;
  if(so == 1.f || so == si) // interpolation.c:915-931: a crop
  {
    const int x0 = ox0 - ix0, y0 = oy0 - iy0;
    if(x0 < 0 || y0 < 0 || x0 + ow > iw || y0 + oh > ih)
    {
      set_last_error("%s: the 1:1 region [%d, %d) x [%d, %d) lies outside the %d x %d input", tag, x0, x0 + ow, y0, y0 + oh, iw, ih);
      return DT_HIP_INVALID_ARG;
    }
    return dt_hip_enqueue_copy_region(devid, dev_in, iw, x0, y0, dev_out, ow, 0, 0, ow, oh, 16);
  }
  const float scale = so / si;
  plan_host h, v;
  plan_build(h, kind, iw, ix0, ow, ox0, scale);
  plan_build(v, kind, ih, iy0, oh, oy0, scale);
  // one upload: [h.length h.start h.index v.length v.start v.index | h.kernel v.kernel]
  const size_t n_int = h.length.size() + h.start.size() + h.index.size() + v.length.size() + v.start.size() + v.index.size();
  const size_t n_flt = h.kernel.size() + v.kernel.size();
  std::vector<int> blob(n_int + n_flt);
  size_t o = 0;
  const size_t o_hl = o; memcpy(&blob[o], h.length.data(), h.length.size() * 4); o += h.length.size();
  const size_t o_hs = o; memcpy(&blob[o], h.start.data(), h.start.size() * 4); o += h.start.size();
  const size_t o_hi = o; memcpy(&blob[o], h.index.data(), h.index.size() * 4); o += h.index.size();
  const size_t o_vl = o; memcpy(&blob[o], v.length.data(), v.length.size() * 4); o += v.length.size();
  const size_t o_vs = o; memcpy(&blob[o], v.start.data(), v.start.size() * 4); o += v.start.size();
  const size_t o_vi = o; memcpy(&blob[o], v.index.data(), v.index.size() * 4); o += v.index.size();
  const size_t o_hk = o; memcpy(&blob[o], h.kernel.data(), h.kernel.size() * 4); o += h.kernel.size();
  const size_t o_vk = o; memcpy(&blob[o], v.kernel.data(), v.kernel.size() * 4); o += v.kernel.size();
  int *dev = (int *)dt_hip_alloc_device_buffer(devid, blob.size() * 4);
  if(!dev) return DT_HIP_SYSMEM_ALLOCATION;
  hipStream_t s = stream_of(devid);
  if(hipMemcpyAsync(dev, blob.data(), blob.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess
     || hipStreamSynchronize(s) != hipSuccess) // blob is a stack-lifetime host buffer
  {
    dt_hip_release_mem_object(dev);
    return DT_HIP_DEFAULT_ERROR;
  }
  plan_dev ph = { dev + o_hl, dev + o_hs, dev + o_hi, (const float *)(dev + o_hk) };
  plan_dev pv = { dev + o_vl, dev + o_vs, dev + o_vi, (const float *)(dev + o_vk) };
  {
    launch_scope ls(devid, tag);
    resample<<<pixel_grid((size_t)ow * oh), 256, 0, s>>>((const float4 *)dev_in, (float4 *)dev_out, iw, ow, oh, ph, pv);
  }
  dt_hip_release_mem_object(dev);
  return check_launch(tag);
}


} // namespace

extern "C" {

int dt_hip_iop_finalscale_process(int devid, const dt_hip_piece_t *piece, const dt_hip_finalscale_data_t *d,
                                  dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out || piece->channels != 4) return DT_HIP_INVALID_ARG;
  if(d->interpolation < 0 || d->interpolation > 2)
  {
    set_last_error("finalscale: unknown interpolator %d", d->interpolation);
    return DT_HIP_INVALID_ARG;
  }
  return resample_launch(devid, piece, d->interpolation, dev_in, dev_out, false, "finalscale_resample");
}

// initialscale: process(), src/iop/initialscale.c:120-127 -- the same resampler with the regions as they are (roi_in:
// the whole input buffer at scale 1, modify_roi_in() :72-83; roi_out: a region of the scaled image)
int dt_hip_iop_initialscale_process(int devid, const dt_hip_piece_t *piece, const dt_hip_finalscale_data_t *d,
                                    dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out || piece->channels != 4) return DT_HIP_INVALID_ARG;
  if(d->interpolation < 0 || d->interpolation > 2)
  {
    set_last_error("initialscale: unknown interpolator %d", d->interpolation);
    return DT_HIP_INVALID_ARG;
  }
  return resample_launch(devid, piece, d->interpolation, dev_in, dev_out, true, "initialscale_resample");
}

} // extern "C"
