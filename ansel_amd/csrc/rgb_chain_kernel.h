// rgb_chain_kernel.h -- the fused RGBA chain kernel (see pipe_fused.hip), instantiated per
// {color-calibration adaptation, filmic colour science} in pipe_fused_rgb_*.hip so that the
// translation units compile in parallel.  Stage order is the reference's pipe order
// (src/develop/iop_order.c:565-: exposure 25 < colorin 31 < channelmixerrgb 33 < filmicrgb 49 <
// colorout): exposure -> colorin -> color calibration -> filmic -> colorout -> [float -> u16].
#pragma once
#include "px_colorspaces.h"
#include "px_channelmixerrgb.h"
#include "px_filmicrgb.h"
#include "px_bilat.h"

namespace ansel
{

struct chain_args
{
  int has_exposure, has_colorin, has_colorout, to_u16;
  int cm_clip, filmic_export;
  int pre_lab, post_lab; // the pipe's Lab -> RGB glue in front of the chain / RGB -> Lab behind it
  float lab_pre[3][4], lab_post[3][4];
  float exp_black, exp_scale;
  conv_args colorin, colorout;
  cm_args cm;
  fargs filmic;
  // local contrast's slice (bilateral-grid mode) in front of everything: the run then reads the MODULE's input (px_bilat.h)
  int pre_bilat;
  bilat_slice_args bilat;
};

// offset of the by-value chain_args in the kernarg segment of rgb_chain(): after two pointers and a size_t
constexpr int CHAIN_ARGS_KERNARG_OFFSET = ansel::kernarg_offset_after<chain_args, const float4 *, void *, size_t>();
static_assert(CHAIN_ARGS_KERNARG_OFFSET == 24, "rgb_chain: the by-value chain_args follows two pointers and a size_t");
constexpr int CM_NONE = -1;
constexpr int FM_NONE = -1;

template <int CM, int FM>
__global__ __launch_bounds__(256) void rgb_chain(const float4 *__restrict__ in, void *__restrict__ out,
                                                  const size_t npixels, const chain_args a_by_value)
{
  // the ~340 dwords of parameters are read where they are used (hip_common.h kernarg_at())
  const chain_args &a = kernarg_at<chain_args>(CHAIN_ARGS_KERNARG_OFFSET);
  (void)a_by_value;
  ansel_math::stage_default_tables(threadIdx.x); // (pipe_fused_rgb_*.hip: the lookup tables of the ~12 powf / log2f a pixel, in LDS)
  // one pixel per thread: the ~250 uniform parameters of the five stages are then used once per wave
  // instead of staying live across a grid-stride loop (which spilled 770 SGPRs to VGPR lanes)
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(k < npixels)
  {
    float4 v = in[k];
    if(a.pre_bilat)
    {
      const int j = (int)(k / (size_t)a.bilat.b.width), i = (int)(k - (size_t)j * a.bilat.b.width);
      v.x = bilat_slice_lightness(v.x, i, j, a.bilat.grid, a.bilat.b, a.bilat.norm);
    }
    if(a.pre_lab) v = px_lab_to_rgb(v, a.lab_pre);
    if(a.has_exposure)
    {
      // exposure.c:521-524 on all four lanes
      v.x = (v.x - a.exp_black) * a.exp_scale;
      v.y = (v.y - a.exp_black) * a.exp_scale;
      v.z = (v.z - a.exp_black) * a.exp_scale;
      v.w = (v.w - a.exp_black) * a.exp_scale;
    }
    if(a.has_colorin) v = px_conversion_rt(v, a.colorin);
    if(CM != CM_NONE) v = px_channelmixerrgb<(CM == CM_NONE ? 0 : CM)>(v, a.cm, a.cm_clip != 0);
    if(FM != FM_NONE) v = px_filmicrgb<(FM == FM_NONE ? 0 : FM)>(v, a.filmic, a.filmic_export != 0);
    if(a.has_colorout) v = px_conversion_rt(v, a.colorout);
    if(a.post_lab) v = px_rgb_to_lab(v, a.lab_post);
    if(a.to_u16)
    {
      // _export_final_buffer_to_uint16(), src/imageio/imageio_core.c:729-737 (glib CLAMP)
      const float x = roundf(v.x * 65535.f), y = roundf(v.y * 65535.f), z = roundf(v.z * 65535.f), w = roundf(v.w * 65535.f);
      typedef unsigned short v4us_t __attribute__((ext_vector_type(4)));
      const v4us_t ov = { (unsigned short)(int)(x > 65535.f ? 65535.f : (x < 0.f ? 0.f : x)),
                          (unsigned short)(int)(y > 65535.f ? 65535.f : (y < 0.f ? 0.f : y)),
                          (unsigned short)(int)(z > 65535.f ? 65535.f : (z < 0.f ? 0.f : z)),
                          (unsigned short)(int)(w > 65535.f ? 65535.f : (w < 0.f ? 0.f : w)) };
      if(a.to_u16 == 2)
      {
        // the scanline a format writer hands to its library: 3 samples per pixel (tiff.c:322-339)
        unsigned short *const o = reinterpret_cast<unsigned short *>(out) + 3 * k;
        __builtin_nontemporal_store(ov.x, o);
        __builtin_nontemporal_store(ov.y, o + 1);
        __builtin_nontemporal_store(ov.z, o + 2);
      }
      else
        // streaming store: written once, read by nobody on the device (PMC: 1.14 GB of write requests per
        // 0.81 GB written, against 1.58 GB with a cached store; pairing lanes into 16-byte stores changed
        // nothing and cost 3 %)
        __builtin_nontemporal_store(ov, reinterpret_cast<v4us_t *>(out) + k);
    }
    else
      nt_store(reinterpret_cast<float4 *>(out) + k, v);
  }
}

template <int FM>
int rgb_chain_launch_fm(const int cm_kind, const unsigned grid, hipStream_t s, const float4 *in, void *out,
                        const size_t np, const chain_args &a)
{
  switch(cm_kind)
  {
    case CM_NONE: rgb_chain<CM_NONE, FM><<<grid, 256, 0, s>>>(in, out, np, a); break;
    case DT_HIP_ADAPTATION_LINEAR_BRADFORD: rgb_chain<DT_HIP_ADAPTATION_LINEAR_BRADFORD, FM><<<grid, 256, 0, s>>>(in, out, np, a); break;
    case DT_HIP_ADAPTATION_CAT16: rgb_chain<DT_HIP_ADAPTATION_CAT16, FM><<<grid, 256, 0, s>>>(in, out, np, a); break;
    case DT_HIP_ADAPTATION_FULL_BRADFORD: rgb_chain<DT_HIP_ADAPTATION_FULL_BRADFORD, FM><<<grid, 256, 0, s>>>(in, out, np, a); break;
    case DT_HIP_ADAPTATION_XYZ: rgb_chain<DT_HIP_ADAPTATION_XYZ, FM><<<grid, 256, 0, s>>>(in, out, np, a); break;
    case DT_HIP_ADAPTATION_RGB: rgb_chain<DT_HIP_ADAPTATION_RGB, FM><<<grid, 256, 0, s>>>(in, out, np, a); break;
    default: return DT_HIP_INVALID_ARG;
  }
  return DT_HIP_SUCCESS;
}

// one per translation unit
int rgb_chain_launch_none(int cm_kind, unsigned grid, hipStream_t s, const float4 *in, void *out, size_t np, const chain_args &a);
int rgb_chain_launch_agx(int cm_kind, unsigned grid, hipStream_t s, const float4 *in, void *out, size_t np, const chain_args &a);
int rgb_chain_launch_v5(int cm_kind, unsigned grid, hipStream_t s, const float4 *in, void *out, size_t np, const chain_args &a);
int rgb_chain_launch_split_v4(int cm_kind, unsigned grid, hipStream_t s, const float4 *in, void *out, size_t np, const chain_args &a);
int rgb_chain_launch_chroma_v4(int cm_kind, unsigned grid, hipStream_t s, const float4 *in, void *out, size_t np, const chain_args &a);

} // namespace ansel
