/* oracle/ref_wrap/ref_guided.c -- TEST INFRASTRUCTURE: the reference's own guided filter (src/pixel/guided_filter.c and
 * src/pixel/box_filters.c, compiled from where they lie by oracle/Makefile) behind a plain C entry point */
#include "pixel/guided_filter.h"

int ref_guided_filter(const float *guide, const float *in, float *out, int width, int height, int ch, int w, float sqrt_eps,
                      float guide_weight, float min, float max)
{
  return guided_filter(guide, in, out, width, height, ch, w, sqrt_eps, guide_weight, min, max);
}
