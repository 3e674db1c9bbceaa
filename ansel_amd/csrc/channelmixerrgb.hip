// channelmixerrgb.hip -- color calibration on gfx950: loop_switch(), src/iop/channelmixerrgb.c:766-960
// with gamut_mapping() :642-706, luma_chroma() :707-763 and the chromatic adaptation helpers of
// src/pixel/chromatic_adaptation.h (Bradford linear / full, CAT16, XYZ, none).
//
// Pointwise, 16 B in + 16 B out per pixel (32 B/px algorithmic).  About ten 3x3 products, six
// divisions, two square roots and (full Bradford, gamut compression) two powf per pixel: still
// HBM-bound on MI355X (~250 f32 ops against 32 B).  The adaptation kind and the clip flag are
// template parameters -- the reference "forces loop unswitching in a controlled way" the same way
// (channelmixerrgb.c:2030-2070) -- the saturation-algorithm version and the grey switch are uniform
// branches.
//
// Numerics: dt_mat3x4_mul_vec4() is mul, mul+add, mul+add (unfused, -ffp-contract=off); DT_FMA()
// (src/math/math.h:59-65) is a real fma; powf is glibc's (devmath.h).
#include "px_channelmixerrgb.h"

using namespace ansel;

namespace
{

template <int KIND, bool CLIP>
__global__ __launch_bounds__(256) void channelmixerrgb(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                        const size_t npixels, const cm_args a)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; // one pixel per thread (pixel_grid)
  if(k < npixels)
    nt_store(out + k, px_channelmixerrgb<KIND>(in[k], a, CLIP));
}

template <int KIND>
void launch_k(const bool clip, const unsigned grid, hipStream_t s, const float4 *in, float4 *out, const size_t np,
              const cm_args &a)
{
  if(clip)
    channelmixerrgb<KIND, true><<<grid, 256, 0, s>>>(in, out, np, a);
  else
    channelmixerrgb<KIND, false><<<grid, 256, 0, s>>>(in, out, np, a);
}

} // namespace

namespace ansel
{
void channelmixerrgb_fill_args(const dt_hip_channelmixerrgb_data_t *d, cm_args &a)
{
  for(int r = 0; r < 3; r++)
    for(int c = 0; c < 3; c++)
    {
      a.XYZ_to_RGB[r][c] = d->XYZ_to_RGB[r][c];
      a.RGB_to_XYZ[r][c] = d->RGB_to_XYZ[r][c];
      a.MIX[r][c] = d->MIX[r][c];
    }
  for(int c = 0; c < 3; c++)
  {
    a.illuminant[c] = d->illuminant[c];
    a.saturation[c] = d->saturation[c];
    a.lightness[c] = d->lightness[c];
    a.grey[c] = d->grey[c];
  }
  a.p = d->p;
  a.gamut = d->gamut;
  a.apply_grey = d->apply_grey;
  a.version = d->version;
  a.kind = d->adaptation;
  a.clip = d->clip != 0;
}
} // namespace ansel

extern "C" int dt_hip_iop_channelmixerrgb_process(int devid, const dt_hip_piece_t *piece,
                                                  const dt_hip_channelmixerrgb_data_t *d, dt_hip_mem_t dev_in,
                                                  dt_hip_mem_t dev_out)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  if(piece->channels != 4) return DT_HIP_INVALID_ARG;
  const size_t np = (size_t)piece->roi_out.width * piece->roi_out.height;
  if(np == 0) return DT_HIP_SUCCESS;
  cm_args a;
  channelmixerrgb_fill_args(d, a);
  const unsigned grid = pixel_grid(np);
  hipStream_t s = stream_of(devid);
  const float4 *in = (const float4 *)dev_in;
  float4 *out = (float4 *)dev_out;
  const bool clip = d->clip != 0;
  launch_scope ls(devid, "channelmixerrgb");
  switch(d->adaptation)
  {
    case DT_HIP_ADAPTATION_FULL_BRADFORD: launch_k<DT_HIP_ADAPTATION_FULL_BRADFORD>(clip, grid, s, in, out, np, a); break;
    case DT_HIP_ADAPTATION_LINEAR_BRADFORD: launch_k<DT_HIP_ADAPTATION_LINEAR_BRADFORD>(clip, grid, s, in, out, np, a); break;
    case DT_HIP_ADAPTATION_CAT16: launch_k<DT_HIP_ADAPTATION_CAT16>(clip, grid, s, in, out, np, a); break;
    case DT_HIP_ADAPTATION_XYZ: launch_k<DT_HIP_ADAPTATION_XYZ>(clip, grid, s, in, out, np, a); break;
    case DT_HIP_ADAPTATION_RGB: launch_k<DT_HIP_ADAPTATION_RGB>(clip, grid, s, in, out, np, a); break;
    default:
      // DT_ADAPTATION_LAST: process() runs no loop at all (channelmixerrgb.c:2071-2075)
      return DT_HIP_SUCCESS;
  }
  return check_launch("channelmixerrgb");
}
