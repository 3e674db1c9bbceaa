// colorspaces.hip -- colorin / colorout on gfx950: the matrix branch of
// dt_colorspaces_apply_conversion(), src/colorprofiles/conversion.c:744-770:
//
//   [source tone curves] -> [legacy blue mapping hook] -> 3x3 -> [clamp to [0,1] -> 3x3] -> [target tone curves]
//
//   _apply_matrix()          conversion.c:593-682
//   _apply_target_curves()   conversion.c:546-582
//   dt_ioppr_eval_trc()      src/colorprofiles/iop_profile.h:577-580  (65536-entry LUT with lerp
//                            below 1.0, fitted power law b*(a*x)^c above)
//   apply_blue_mapping()     src/iop/colorin.c:690-709
//
// One kernel, 16 B in + 16 B out per pixel (32 B/px algorithmic).  The reference runs the target
// curves as a second pass over the output only so that its compiler contracts the matrix loop the
// same way with and without curves (conversion.c:585-591); with contraction off there is nothing
// to preserve and the curves are applied in the same thread -- same arithmetic, half the traffic.
// The three 256 KiB curves are read with plain loads and stay resident in the 4 MiB XCD L2.
// The lcms2 branch of the reference (non-matrix profiles) has no device path there either
// (colorin.c:887-888 clears process_cl_ready); dt_hip_iop_color*_process is only called for
// is_matrix conversions.
#include "px_colorspaces.h"

using namespace ansel;

namespace
{

template <bool DECODE, bool ENCODE, bool CLIP, bool HOOK>
__global__ __launch_bounds__(256) void apply_matrix(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                     const size_t npixels, const conv_args a)
{
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; // one pixel per thread (pixel_grid)
  if(k < npixels)
    nt_store(out + k, px_conversion(in[k], a, DECODE, ENCODE, CLIP, HOOK));
}

template <bool D, bool E, bool C>
void launch_h(const bool hook, const unsigned grid, hipStream_t s, const float4 *in, float4 *out, const size_t np,
              const conv_args &a)
{
  if(hook)
    apply_matrix<D, E, C, true><<<grid, 256, 0, s>>>(in, out, np, a);
  else
    apply_matrix<D, E, C, false><<<grid, 256, 0, s>>>(in, out, np, a);
}

int conversion_process(const int devid, const char *tag, const dt_hip_piece_t *piece, const dt_hip_conversion_t *d,
                       dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  if(piece->channels != 4) return DT_HIP_INVALID_ARG;
  const size_t np = (size_t)piece->roi_out.width * piece->roi_out.height;
  if(np == 0) return DT_HIP_SUCCESS;
  conv_args a;
  const int key = conversion_fill_args(d, a);
  const bool hook = d->blue_mapping != 0;
  const unsigned grid = pixel_grid(np);
  hipStream_t s = stream_of(devid);
  const float4 *in = (const float4 *)dev_in;
  float4 *out = (float4 *)dev_out;
  launch_scope ls(devid, tag);
  switch(key)
  {
    case 0: launch_h<false, false, false>(hook, grid, s, in, out, np, a); break;
    case 1: launch_h<false, false, true>(hook, grid, s, in, out, np, a); break;
    case 2: launch_h<false, true, false>(hook, grid, s, in, out, np, a); break;
    case 3: launch_h<false, true, true>(hook, grid, s, in, out, np, a); break;
    case 4: launch_h<true, false, false>(hook, grid, s, in, out, np, a); break;
    case 5: launch_h<true, false, true>(hook, grid, s, in, out, np, a); break;
    case 6: launch_h<true, true, false>(hook, grid, s, in, out, np, a); break;
    default: launch_h<true, true, true>(hook, grid, s, in, out, np, a); break;
  }
  return check_launch(tag);
}

} // namespace

namespace ansel
{
int conversion_fill_args(const dt_hip_conversion_t *d, conv_args &a)
{
  memset(&a, 0, sizeof(a));
  for(int r = 0; r < 3; r++)
    for(int c = 0; c < 3; c++)
    {
      a.m[r][c] = d->matrix[r][c];
      a.cm[r][c] = d->clip_matrix[r][c];
      a.cs[r][c] = d->coeffs_source[r][c];
      a.ct[r][c] = d->coeffs_target[r][c];
    }
  // gate on the buffer, not on the count (conversion.c:610-615)
  const bool decode = d->lut_source[0] != nullptr && d->nonlinear_source > 0;
  const bool encode = d->lut_target[0] != nullptr && d->nonlinear_target > 0;
  bool any_encode = false;
  for(int c = 0; c < 3; c++)
  {
    a.ls[c] = (const float *)d->lut_source[c];
    a.lt[c] = (const float *)d->lut_target[c];
    a.decode[c] = decode && a.ls[c] && d->lut_source_first[c] >= 0.0f;
    a.encode[c] = encode && a.lt[c] && d->lut_target_first[c] >= 0.0f;
    any_encode |= a.encode[c] != 0;
  }
  a.clipping = d->has_clipping;
  a.blue_mapping = d->blue_mapping;
  return (decode ? 4 : 0) | (any_encode ? 2 : 0) | (d->has_clipping ? 1 : 0);
}
} // namespace ansel

extern "C" {

int dt_hip_iop_colorin_process(int devid, const dt_hip_piece_t *piece, const dt_hip_conversion_t *d,
                               dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  return conversion_process(devid, "colorin", piece, d, dev_in, dev_out);
}

int dt_hip_iop_colorout_process(int devid, const dt_hip_piece_t *piece, const dt_hip_conversion_t *d,
                                dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  return conversion_process(devid, "colorout", piece, d, dev_in, dev_out);
}

} // extern "C"

namespace
{
struct lab_args
{
  float m[3][4];
  // _apply_tonecurves(), iop_profile.c:332-372: the channels that have a curve
  const float *lut[3];
  float coeff[3][3];
  int curve[3];
};

__device__ __forceinline__ float4 lab_curves(float4 p, const lab_args &a)
{
  if(a.curve[0]) p.x = eval_trc(p.x, a.lut[0], a.coeff[0]);
  if(a.curve[1]) p.y = eval_trc(p.y, a.lut[1], a.coeff[1]);
  if(a.curve[2]) p.z = eval_trc(p.z, a.lut[2], a.coeff[2]);
  return p;
}

// _transform_rgb_to_lab_matrix(), src/colorprofiles/iop_profile.c:377-418 + dt_XYZ_to_Lab(); TRC: the profile's input
// curves ahead of the matrix (:389-393)
template <bool TRC>
__global__ __launch_bounds__(256) void rgb_to_lab(const float4 *in, float4 *out, const size_t n, const lab_args a)
{
  for(size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x)
    out[k] = px_rgb_to_lab(TRC ? lab_curves(in[k], a) : in[k], a.m);
}

// _transform_lab_to_rgb_matrix(), :423-463 + dt_Lab_to_XYZ(); TRC: the profile's output curves behind the matrix (:455-462)
template <bool TRC>
__global__ __launch_bounds__(256) void lab_to_rgb(const float4 *in, float4 *out, const size_t n, const lab_args a)
{
  for(size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x)
  {
    const float4 rgb = px_lab_to_rgb(in[k], a.m);
    out[k] = TRC ? lab_curves(rgb, a) : rgb;
  }
}

int lab_launch(int devid, const dt_hip_piece_t *piece, const dt_hip_lab_data_t *d, dt_hip_mem_t dev_in,
               dt_hip_mem_t dev_out, const bool to_lab)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out || piece->channels != 4) return DT_HIP_INVALID_ARG;
  const size_t n = (size_t)piece->roi_out.width * piece->roi_out.height;
  if(n == 0) return DT_HIP_SUCCESS;
  lab_args a;
  memset(&a, 0, sizeof(a));
  memcpy(a.m, d->matrix, sizeof(a.m));
  bool trc = false;
  if(d->nonlinearlut)
    for(int c = 0; c < 3; c++)
    {
      a.lut[c] = (const float *)d->lut[c];
      a.curve[c] = a.lut[c] && d->lut_first[c] >= 0.0f;
      for(int k = 0; k < 3; k++) a.coeff[c][k] = d->unbounded_coeffs[c][k];
      trc |= a.curve[c] != 0;
    }
  hipStream_t s = stream_of(devid);
  const unsigned grid = stream_grid(n, 256);
  const float4 *in = (const float4 *)dev_in;
  float4 *out = (float4 *)dev_out;
  launch_scope ls(devid, to_lab ? "rgb_to_lab" : "lab_to_rgb");
  if(to_lab && trc)
    rgb_to_lab<true><<<grid, 256, 0, s>>>(in, out, n, a);
  else if(to_lab)
    rgb_to_lab<false><<<grid, 256, 0, s>>>(in, out, n, a);
  else if(trc)
    lab_to_rgb<true><<<grid, 256, 0, s>>>(in, out, n, a);
  else
    lab_to_rgb<false><<<grid, 256, 0, s>>>(in, out, n, a);
  return check_launch(to_lab ? "rgb_to_lab" : "lab_to_rgb");
}
} // namespace

extern "C" {
int dt_hip_transform_rgb_to_lab(int devid, const dt_hip_piece_t *piece, const dt_hip_lab_data_t *d, dt_hip_mem_t dev_in,
                                dt_hip_mem_t dev_out)
{
  return lab_launch(devid, piece, d, dev_in, dev_out, true);
}
int dt_hip_transform_lab_to_rgb(int devid, const dt_hip_piece_t *piece, const dt_hip_lab_data_t *d, dt_hip_mem_t dev_in,
                                dt_hip_mem_t dev_out)
{
  return lab_launch(devid, piece, d, dev_in, dev_out, false);
}
} // extern "C"

