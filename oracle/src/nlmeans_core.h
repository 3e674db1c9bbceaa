/* oracle/src/nlmeans_core.h -- TEST INFRASTRUCTURE ONLY: dt_nlmeans_param_t of
 * src/pixel/nlmeans_core.h:31-50, reduced to the fields the CPU core reads */
#ifndef ORACLE_NLMEANS_CORE_H
#define ORACLE_NLMEANS_CORE_H
typedef struct oracle_nlm_params_t
{
  float scattering, scale, luma, chroma, center_weight, sharpness;
  int patch_radius, search_radius;
  float norm[4];
} oracle_nlm_params_t;
void oracle_nlmeans_core(const float *in, float *out, int width, int height, const oracle_nlm_params_t *p);
int oracle_nlmeans_slice_height(int height);
int oracle_nlmeans_slice_width(int width);
#endif
