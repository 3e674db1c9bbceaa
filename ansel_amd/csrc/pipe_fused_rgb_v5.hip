// pipe_fused_rgb_v5.hip -- instantiations of the fused RGBA chain (rgb_chain_kernel.h) for filmic
// mode MODE_V5, one per color-calibration adaptation.
// powf / log2f / expf look their tables up in the workgroup's LDS copy (devmath.h): rgb_chain() stages it
#define ANSEL_MATH_DEFAULT_TABS tabs_lds
#include "rgb_chain_kernel.h"
namespace ansel
{
int rgb_chain_launch_v5(int cm_kind, unsigned grid, hipStream_t s, const float4 *in, void *out, size_t np, const chain_args &a)
{
  return rgb_chain_launch_fm<MODE_V5>(cm_kind, grid, s, in, out, np, a);
}
} // namespace ansel
