// pipe_fused_rgb_split_v4.hip -- instantiations of the fused RGBA chain (rgb_chain_kernel.h) for filmic
// mode MODE_SPLIT_V4, one per color-calibration adaptation.
#include "rgb_chain_kernel.h"
namespace ansel
{
int rgb_chain_launch_split_v4(int cm_kind, unsigned grid, hipStream_t s, const float4 *in, void *out, size_t np, const chain_args &a)
{
  return rgb_chain_launch_fm<MODE_SPLIT_V4>(cm_kind, grid, s, in, out, np, a);
}
} // namespace ansel
