// testhooks.hip -- device-side self-test entry points (not part of include/ansel_hip.h).
#include "hip_common.h"
#include "devmath.h"
#include "ieee_inrange.h"

#include <atomic>
#include <string.h>

using namespace ansel;

// ---- dispatch overrides (hip_common.h dispatch_key_t) ----
namespace
{
std::atomic<int> g_dispatch[DISPATCH_KEYS];
const char *const g_dispatch_names[DISPATCH_KEYS] = { "nlm_v2", "nlm_fused", "amaze_unfused", "amaze_slab", "amaze_blocks",
                                                       "bilat_blur_split" };
} // namespace
namespace ansel
{
int dispatch_override(const dispatch_key_t key) { return g_dispatch[key].load(std::memory_order_relaxed); }
} // namespace ansel
extern "C" int dt_hip_test_dispatch(const char *key, int value)
{
  if(!key) return DT_HIP_INVALID_ARG;
  for(int k = 0; k < DISPATCH_KEYS; k++)
    if(!strcmp(key, g_dispatch_names[k]))
    {
      g_dispatch[k].store(value);
      return DT_HIP_SUCCESS;
    }
  return DT_HIP_INVALID_ARG;
}

// ---- test hooks: the device build of devmath.h on plain arrays (tests/test_gpu_devmath.py) ----
namespace
{
template <int FN>
__global__ void devmath_test(const float *__restrict__ x, const float *__restrict__ y, float *__restrict__ o, const size_t n)
{
  for(size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x)
  {
    float r;
    if(FN == 0) r = ansel_math::powf_exact(x[k], y[k]);
    else if(FN == 1) r = ansel_math::log2f_exact(x[k]);
    else if(FN == 2) r = ansel_math::exp2f_exact(x[k]);
    else if(FN == 3) r = ansel_math::expf_exact(x[k]);
    else if(FN == 4) r = ansel_math::atan2f_exact(x[k], y[k]);
    else if(FN == 5) r = ansel_math::hypotf_exact(x[k], y[k]);
    else if(FN == 6) r = ansel_math::sinf_exact(x[k]);
    else if(FN == 7) r = ansel_math::cosf_exact(x[k]);
    else if(FN == 9) r = ansel_math::logf_exact(x[k]);
    else if(FN == 10) r = ansel_ieee::div_core(x[k], y[k]);
    else if(FN == 11) r = ansel_ieee::rcp_core(x[k]);
    else if(FN == 12) r = ansel_ieee::sqrt_core(x[k]);
    else if(FN == 13) r = ansel_ieee::zero_or_above_2m96(x[k]) ? 1.0f : 0.0f;
    else if(FN == 14) r = ansel_ieee::div_uniform(x[k], y[k], ansel_ieee::rcp_refined(y[k]));
    else r = fmodf(x[k], y[k]); // the device library's: fmod is exact, any correct implementation agrees
    o[k] = r;
  }
}
template <int FN> int devmath_launch(int devid, const void *x, const void *y, void *o, size_t n)
{
  if(!valid_device(devid)) return DT_HIP_INVALID_ARG;
  devmath_test<FN><<<stream_grid(n, 256), 256, 0, stream_of(devid)>>>((const float *)x, (const float *)y, (float *)o, n);
  return check_launch("devmath_test");
}
} // namespace
extern "C" {
int dt_hip_test_powf(int devid, const void *x, const void *y, void *o, size_t n) { return devmath_launch<0>(devid, x, y, o, n); }
int dt_hip_test_log2f(int devid, const void *x, const void *y, void *o, size_t n) { return devmath_launch<1>(devid, x, y, o, n); }
int dt_hip_test_exp2f(int devid, const void *x, const void *y, void *o, size_t n) { return devmath_launch<2>(devid, x, y, o, n); }
int dt_hip_test_expf(int devid, const void *x, const void *y, void *o, size_t n) { return devmath_launch<3>(devid, x, y, o, n); }
int dt_hip_test_atan2f(int devid, const void *x, const void *y, void *o, size_t n) { return devmath_launch<4>(devid, x, y, o, n); }
int dt_hip_test_hypotf(int devid, const void *x, const void *y, void *o, size_t n) { return devmath_launch<5>(devid, x, y, o, n); }
int dt_hip_test_sinf(int devid, const void *x, const void *y, void *o, size_t n) { return devmath_launch<6>(devid, x, y, o, n); }
int dt_hip_test_cosf(int devid, const void *x, const void *y, void *o, size_t n) { return devmath_launch<7>(devid, x, y, o, n); }
int dt_hip_test_fmodf(int devid, const void *x, const void *y, void *o, size_t n) { return devmath_launch<8>(devid, x, y, o, n); }
int dt_hip_test_logf(int devid, const void *x, const void *y, void *o, size_t n) { return devmath_launch<9>(devid, x, y, o, n); }
// ieee_inrange.h: the division / reciprocal / square root without range scaffolding, on operands INSIDE their stated domains
int dt_hip_test_div_core(int devid, const void *x, const void *y, void *o, size_t n) { return devmath_launch<10>(devid, x, y, o, n); }
int dt_hip_test_rcp_core(int devid, const void *x, const void *y, void *o, size_t n) { return devmath_launch<11>(devid, x, y, o, n); }
int dt_hip_test_sqrt_core(int devid, const void *x, const void *y, void *o, size_t n) { return devmath_launch<12>(devid, x, y, o, n); }
int dt_hip_test_zero_or_above_2m96(int devid, const void *x, const void *y, void *o, size_t n) { return devmath_launch<13>(devid, x, y, o, n); }
int dt_hip_test_div_uniform(int devid, const void *x, const void *y, void *o, size_t n) { return devmath_launch<14>(devid, x, y, o, n); }
}
