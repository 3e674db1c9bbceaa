// nlmeans_core_params.h -- dt_nlmeans_param_t (src/pixel/nlmeans_core.h:31-50) as the device core takes it
#pragma once
#include "hip_common.h"

namespace ansel
{
struct nlm_core_params_t
{
  float scattering, scale, luma, chroma, center_weight, sharpness;
  int patch_radius, search_radius;
  float norm[4];
  const band_view_t *band; // nullptr: the buffers are the frame
  // round 6 (pipe.cpp: local contrast's bilateral grid behind the module): every output pixel's lightness cell, (fraction, index as
  // float bits) per pixel -- what bilat.hip's bilat_zcells would compute from the output in a pass of its own -- or nullptr
  float2 *cell_out;
  float cell_sigma_r;
  int cell_size_z;
};
int nlmeans_core_launch(int devid, const float4 *in, float4 *out, int width, int height, const nlm_core_params_t &p);
int nlmeans_core_halo_rows(int frame_h, const nlm_core_params_t &p);
} // namespace ansel
