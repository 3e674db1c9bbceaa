// tests/native/nlm2_host.cpp -- TEST INFRASTRUCTURE: the body of the gfx950 kernel nlm_chunks_v2
// (ansel_amd/csrc/nlm2_body.h) compiled for the host.  A workgroup is 1024 OS threads meeting at a std::barrier
// where the kernel has __syncthreads(), its LDS a heap block: every index, every hand-off between the pipelined
// phases, every table slot of the schedule runs exactly as on the device (minus the timing), so the CPU suite can
// compare the kernel with the oracle bit for bit before a GPU is involved.  -ffp-contract=off like the device build.
//
//   g++ -O2 -std=c++20 -ffp-contract=off -fPIC -shared -pthread -I ansel_amd/csrc tests/native/nlm2_host.cpp -o tests/native/libnlm2_host.so
#include <barrier>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <thread>
#include <vector>

#include "nlm2_body.h"
#include "nlm3_body.h"
#include "nlm_tail_body.h"

namespace
{

struct F4
{
  float x, y, z, w;
};
struct I2
{
  int x, y;
};

// the fields of nlm_args (nlmeans.hip) the body reads
struct Args
{
  int W, H;
  int chk_w, chk_h, nchx;
  int radius, npatch;
  float sharpness;
  float norm[3];
  float luma, chroma;
  int skip_blend;
  int reach;
  int cy0, out_row0, out_row1;
  int variant;
  float center_weight, cpn; // CENTER bodies (denoise (profiled)'s weight): nlm_args of nlmeans.hip
};

// what the lanes of a wave exchange (DPP on the device): a slot per thread, two sets used in turn, one wave barrier per exchange
struct WaveExchange
{
  float slot[2][1024];
  std::barrier<> *wave_bar[16];
  WaveExchange()
  {
    for(auto &b : wave_bar) b = new std::barrier<>(64);
  }
  ~WaveExchange()
  {
    for(auto &b : wave_bar) delete b;
  }
};

struct HostEnv
{
  int tid_, bid_;
  float *lds_;
  std::barrier<> *bar_;
  WaveExchange *xch_ = nullptr;
  mutable int xch_turn_ = 0;
  // DPP row_shr:1: the value of the lane to the left within a row of 16 lanes, 0 for the row's first lane.  Every lane of
  // the wave calls it (the device executes it with all lanes enabled)
  float lane_shr1(const float v) const
  {
    const int set = xch_turn_ & 1;
    xch_turn_++;
    xch_->slot[set][tid_] = v;
    xch_->wave_bar[tid_ >> 6]->arrive_and_wait();
    return (tid_ & 15) ? xch_->slot[set][tid_ - 1] : 0.0f;
  }
  int tid() const { return tid_; }
  int bid() const { return bid_; }
  float *lds() const { return lds_; }
  void sync() const { bar_->arrive_and_wait(); }
  void prio_high() const {}
  template <class A> void store_cell(const A &, long, float) const {} // (the device's extra store for the bilateral grid behind the module)
  void sched_fence() const {}
  // ds_write_addtid_b32: base + offset + 4 * lane
  template <int OFF> void st_addtid(float *const wave_base, const int lane, const float v) const { wave_base[OFF / 4 + lane] = v; }
  // eight rows of the column recurrence (nlm3_body.h column_chain): v = v + term[t], stored at row t, for t = T0 .. T0 + 7
  template <int T0, int ROWBYTES> void chain8(float *const wave_base, const int lane, float &v, const float *const term) const
  {
    for(int t = T0; t < T0 + 8; t++)
    {
      v = v + term[t];
      wave_base[(size_t)t * (ROWBYTES / 4) + lane] = v;
    }
  }
  bool any(const bool) const { return true; } // a wave-level vote on the device; every slot stays live here
  static constexpr bool TIMED = false;
  long long clock() const { return 0; }
  // v_cvt_i32_f32: truncation, saturating, NaN -> 0
  static int cvt_i32_sat(const float v)
  {
    if(v != v) return 0;
    if(v >= 2147483648.0f) return 0x7fffffff;
    if(v <= -2147483648.0f) return (int)0x80000000;
    return (int)v;
  }
  static float max_num(const float a, const float b) { return fmaxf(a, b); } // v_max_f32: the number, if one is a NaN
  static float min_num(const float a, const float b) { return fminf(a, b); }
  // the device's quotient by a wave-uniform divisor (nlmeans.hip div_uniform(): correctly rounded, checked on the GPU)
  static float rcp_refined(const float d) { return 1.0f / d; }
  static float div_uniform(const float n, const float d, const float) { return n / d; }
  static float int_as_float(const int v)
  {
    float f;
    memcpy(&f, &v, sizeof(f));
    return f;
  }
};

int sgn(const int v) { return (v > 0) - (v < 0); }
int scatter(const float scale, const float scattering, const int i1, const int i2)
{
  const int a1 = abs(i1), a2 = abs(i2);
  return (int)(scale * ((a1 * a1 * a1 + 7.0 * a1 * sqrt((double)a2)) * sgn(i1) * scattering / 6.0 + i1));
}

template <int P, int WP, int TP, bool DEEP, bool CENTER = false>
void run(const F4 *in, F4 *out, const Args &a, const I2 *patches, const int nchunks, const size_t lds_floats)
{
  std::vector<float> lds(lds_floats + 4096, 0.0f);
  std::barrier<> bar(NL2_THREADS);
  std::vector<std::thread> pool;
  pool.reserve(NL2_THREADS);
  for(int t = 0; t < NL2_THREADS; t++)
    pool.emplace_back([&, t]() {
      for(int b = 0; b < nchunks; b++)
      {
        HostEnv env{ t, b, lds.data(), &bar };
        nlm2::body<P, WP, TP, DEEP, CENTER>(env, in, out, a, patches);
        bar.arrive_and_wait(); // the next chunk reuses the LDS block
      }
    });
  for(auto &th : pool) th.join();
}

} // namespace

// Runs the kernel body over every chunk of the frame (only interior chunks write; the rest of `out` is left as it
// is).  Returns 0, or a negative number when the configuration is outside what nlm_chunks_v2 takes (the same tests as
// nlmeans_core_launch()).  chk_w / chk_h: the frame's chunk grid (oracle_nlmeans_slice_width / _height).
static int nlm2_host_run_(const float *in, float *out, int W, int H, int chk_w, int chk_h, int patch_radius,
                          int search_radius, float scale, float scattering, float sharpness, const float *norm,
                          float luma, float chroma, int *interior_chunks, const float center_weight);
extern "C" int nlm2_host_run(const float *in, float *out, int W, int H, int chk_w, int chk_h, int patch_radius,
                             int search_radius, float scale, float scattering, float sharpness, const float *norm,
                             float luma, float chroma, int *interior_chunks)
{
  return nlm2_host_run_(in, out, W, H, chk_w, chk_h, patch_radius, search_radius, scale, scattering, sharpness, norm, luma, chroma,
                        interior_chunks, -1.0f);
}
// ... with the weight of denoise (profiled)'s non-local-means mode (center_weight >= 0: nlm2_body.h, CENTER)
extern "C" int nlm2_host_run_center(const float *in, float *out, int W, int H, int chk_w, int chk_h, int patch_radius,
                                    int search_radius, float scale, float scattering, float sharpness, const float *norm,
                                    float luma, float chroma, int *interior_chunks, float center_weight)
{
  return nlm2_host_run_(in, out, W, H, chk_w, chk_h, patch_radius, search_radius, scale, scattering, sharpness, norm, luma, chroma,
                        interior_chunks, center_weight);
}
static int nlm2_host_run_(const float *in, float *out, int W, int H, int chk_w, int chk_h, int patch_radius,
                          int search_radius, float scale, float scattering, float sharpness, const float *norm,
                          float luma, float chroma, int *interior_chunks, const float center_weight)
{
  const char *const var_env = getenv("ANSEL_NLM2_VARIANT");
  std::vector<I2> patches;
  int max_shift = 0;
  for(int ri = -search_radius; ri <= search_radius; ri++)
    for(int ci = -search_radius; ci <= search_radius; ci++)
    {
      const int r = scatter(scale, scattering, ri, ci), c = scatter(scale, scattering, ci, ri);
      patches.push_back(I2{ r, c });
      max_shift = std::max(max_shift, std::max(abs(r), abs(c)));
    }
  Args a;
  memset(&a, 0, sizeof(a));
  a.W = W;
  a.H = H;
  a.chk_w = chk_w;
  a.chk_h = chk_h;
  a.nchx = (W + chk_w - 1) / chk_w;
  const int nchy = (H + chk_h - 1) / chk_h;
  a.radius = patch_radius;
  a.npatch = (int)patches.size();
  a.sharpness = sharpness;
  for(int k = 0; k < 3; k++) a.norm[k] = norm[k];
  a.luma = luma;
  a.chroma = chroma;
  a.skip_blend = (luma == 1.0 && chroma == 1.0);
  a.reach = patch_radius + 1 + max_shift;
  a.cy0 = 0;
  a.out_row0 = 0;
  a.out_row1 = H;
  a.variant = var_env ? atoi(var_env) : 0;
  a.center_weight = center_weight;
  a.cpn = center_weight * (2 * patch_radius + 1) * (2 * patch_radius + 1); // compute_center_pixel_norm(), nlmeans_core.c:147-153
  const bool center = !(center_weight < 0);
  const int S = 2 * patch_radius + 1, ncol = chk_w + 2 * patch_radius;
  if(patch_radius < 1 || patch_radius > 3) return -1;
  // the same choices as nlmeans_core_launch(): the tight layout when the chunk fits it, four tables when they fit LDS
  // (ANSEL_NLM2_LAYOUT = loose / ANSEL_NLM2_DEEP = 0 force the other paths for the tests)
  const char *const force_layout = getenv("ANSEL_NLM2_LAYOUT"), *const force_deep = getenv("ANSEL_NLM2_DEEP");
  const bool tight = chk_w + 2 * a.reach <= NL2_WP_TIGHT && ncol + 1 <= NL2_TP_TIGHT && !(force_layout && !strcmp(force_layout, "loose"));
  const int WP = tight ? NL2_WP_TIGHT : NL2_WP_LOOSE, TP = tight ? NL2_TP_TIGHT : NL2_TP_LOOSE;
  if(chk_w + 2 * a.reach > WP || chk_h > NL2_SERIAL / 2 || ncol + 1 > TP || chk_w * chk_h > NL2_PAR * NL2_PX || a.npatch > 4096) return -2;
  if(ncol * S > NL2_PAR) return -3;
  const int nseg = NL2_PAR / (ncol * S), m0 = (chk_h - 2) / S + 1;
  if((m0 + nseg - 1) / nseg > NL2_MSEG) return -4;
  const bool deep = nlm2::lds_floats(4, chk_h, a.reach, a.npatch, WP, TP) * sizeof(float) <= 160 * 1024 && chk_h <= 64
                    && !(force_deep && !strcmp(force_deep, "0"));
  const size_t lds_floats = nlm2::lds_floats(deep ? 4 : 2, chk_h, a.reach, a.npatch, WP, TP);
  if(lds_floats * sizeof(float) > 160 * 1024) return -5;
  int interior = 0;
  for(int cy = 0; cy < nchy; cy++)
    for(int cx = 0; cx < a.nchx; cx++)
    {
      const int top = cy * chk_h, left = cx * chk_w;
      const int bot = std::min(top + chk_h, H), right = std::min(left + chk_w, W);
      if(top >= a.reach && bot + a.reach <= H && left >= a.reach && right + a.reach <= W && bot - top == chk_h && right - left == chk_w)
        interior++;
    }
  if(interior_chunks) *interior_chunks = interior;
  const int nchunks = a.nchx * nchy;
  const F4 *const fin = (const F4 *)in;
  F4 *const fout = (F4 *)out;
#define RUN_(P_, C_)                                                                                                      \
  do                                                                                                                      \
  {                                                                                                                       \
    if(tight && deep) run<P_, NL2_WP_TIGHT, NL2_TP_TIGHT, true, C_>(fin, fout, a, patches.data(), nchunks, lds_floats);    \
    else if(tight) run<P_, NL2_WP_TIGHT, NL2_TP_TIGHT, false, C_>(fin, fout, a, patches.data(), nchunks, lds_floats);      \
    else if(deep) run<P_, NL2_WP_LOOSE, NL2_TP_LOOSE, true, C_>(fin, fout, a, patches.data(), nchunks, lds_floats);        \
    else run<P_, NL2_WP_LOOSE, NL2_TP_LOOSE, false, C_>(fin, fout, a, patches.data(), nchunks, lds_floats);                \
  } while(0)
#define RUN(P_)                  \
  do                             \
  {                              \
    if(center) RUN_(P_, true);   \
    else RUN_(P_, false);        \
  } while(0)
  if(patch_radius == 1) RUN(1);
  else if(patch_radius == 2) RUN(2);
  else RUN(3);
#undef RUN
#undef RUN_
  return (tight ? 1 : 0) | (deep ? 2 : 0); /* which path ran */
}

// ---- the third version (ansel_amd/csrc/nlm3_body.h, launched as nlm_chunks_v3): same harness.  Returns 1 when it ran,
//      0 when the configuration is not one it takes (the launch falls back to the second version then), < 0 on error.
namespace
{
template <int NPXL, int MSEG, bool FUSED = false, int P = 2, bool CENTER = false>
void run3(const F4 *in, F4 *out, const Args &a, const I2 *patches, const int nchunks, const size_t lds_floats, const int ndx,
          const bool border = false)
{
  std::vector<float> lds(lds_floats + 4096, 0.0f);
  WaveExchange xch;
  float *base = lds.data();
  while((uintptr_t)base & 15) base++; // the kernel's 16-byte LDS accesses
  std::barrier<> bar(NL3_THREADS);
  std::vector<std::thread> pool;
  pool.reserve(NL3_THREADS);
  for(int t = 0; t < NL3_THREADS; t++)
    pool.emplace_back([&, t]() {
      for(int b = 0; b < nchunks; b++)
      {
        HostEnv env{ t, b, base, &bar, &xch };
        nlm3::body<NPXL, MSEG, false, FUSED, false, P, CENTER>(env, in, out, a, patches, ndx);
        bar.arrive_and_wait();
        if(border)
        {
          HostEnv envb{ t, b, base, &bar, &xch };
          nlm3::body<NPXL, MSEG, true, FUSED, false, P, CENTER>(envb, in, out, a, patches, ndx); // the chunks of the outermost ring it takes
          bar.arrive_and_wait();
        }
      }
    });
  for(auto &th : pool) th.join();
}
} // namespace

static int nlm3_host_run_(const float *in, float *out, int W, int H, int chk_w, int chk_h, int patch_radius,
                          int search_radius, float scale, float scattering, float sharpness, const float *norm,
                          float luma, float chroma, int *interior_chunks, const bool border, const bool fused = false,
                          const float center_weight = -1.0f);
// round 6: patch radius 1 or 2, either weight (center_weight < 0: denoise (non-local means)'s), third version or fused, with or
// without the outermost ring
extern "C" int nlm3_host_run_ex(const float *in, float *out, int W, int H, int chk_w, int chk_h, int patch_radius,
                                int search_radius, float scale, float scattering, float sharpness, const float *norm,
                                float luma, float chroma, int *chunks, float center_weight, int border, int fused)
{
  return nlm3_host_run_(in, out, W, H, chk_w, chk_h, patch_radius, search_radius, scale, scattering, sharpness, norm, luma,
                        chroma, chunks, border != 0, fused != 0, center_weight);
}
// ---- the fused variant (nlm3_body.h FUSED, launched as nlm_chunks_v4): three tables, the row recurrence in the C role
extern "C" int nlm4_host_run(const float *in, float *out, int W, int H, int chk_w, int chk_h, int patch_radius,
                             int search_radius, float scale, float scattering, float sharpness, const float *norm,
                             float luma, float chroma, int *interior_chunks)
{
  return nlm3_host_run_(in, out, W, H, chk_w, chk_h, patch_radius, search_radius, scale, scattering, sharpness, norm, luma,
                        chroma, interior_chunks, false, true);
}
extern "C" int nlm4_host_run_all(const float *in, float *out, int W, int H, int chk_w, int chk_h, int patch_radius,
                                 int search_radius, float scale, float scattering, float sharpness, const float *norm,
                                 float luma, float chroma, int *chunks)
{
  return nlm3_host_run_(in, out, W, H, chk_w, chk_h, patch_radius, search_radius, scale, scattering, sharpness, norm, luma,
                        chroma, chunks, true, true);
}
extern "C" int nlm3_host_run(const float *in, float *out, int W, int H, int chk_w, int chk_h, int patch_radius,
                             int search_radius, float scale, float scattering, float sharpness, const float *norm,
                             float luma, float chroma, int *interior_chunks)
{
  return nlm3_host_run_(in, out, W, H, chk_w, chk_h, patch_radius, search_radius, scale, scattering, sharpness, norm, luma,
                        chroma, interior_chunks, false);
}
// ... and the border ring with the BORDER body; *chunks = the chunks written (interior + border ring taken)
extern "C" int nlm3_host_run_all(const float *in, float *out, int W, int H, int chk_w, int chk_h, int patch_radius,
                                 int search_radius, float scale, float scattering, float sharpness, const float *norm,
                                 float luma, float chroma, int *chunks)
{
  return nlm3_host_run_(in, out, W, H, chk_w, chk_h, patch_radius, search_radius, scale, scattering, sharpness, norm, luma,
                        chroma, chunks, true);
}
static int nlm3_host_run_(const float *in, float *out, int W, int H, int chk_w, int chk_h, int patch_radius,
                          int search_radius, float scale, float scattering, float sharpness, const float *norm,
                          float luma, float chroma, int *interior_chunks, const bool border, const bool fused,
                          const float center_weight)
{
  std::vector<I2> patches;
  int max_shift = 0;
  for(int ri = -search_radius; ri <= search_radius; ri++)
    for(int ci = -search_radius; ci <= search_radius; ci++)
    {
      const int r = scatter(scale, scattering, ri, ci), c = scatter(scale, scattering, ci, ri);
      patches.push_back(I2{ r, c });
      max_shift = std::max(max_shift, std::max(abs(r), abs(c)));
    }
  Args a;
  memset(&a, 0, sizeof(a));
  a.W = W;
  a.H = H;
  a.chk_w = chk_w;
  a.chk_h = chk_h;
  a.nchx = (W + chk_w - 1) / chk_w;
  const int nchy = (H + chk_h - 1) / chk_h;
  a.radius = patch_radius;
  a.npatch = (int)patches.size();
  a.sharpness = sharpness;
  for(int k = 0; k < 3; k++) a.norm[k] = norm[k];
  a.luma = luma;
  a.chroma = chroma;
  a.skip_blend = (luma == 1.0 && chroma == 1.0);
  a.reach = patch_radius + 1 + max_shift;
  a.cy0 = 0;
  a.out_row0 = 0;
  a.out_row1 = H;
  a.center_weight = center_weight;
  a.cpn = center_weight * (2 * patch_radius + 1) * (2 * patch_radius + 1); // compute_center_pixel_norm(), nlmeans_core.c:147-153
  const bool center = !(center_weight < 0);
  int ndx = 0;
  // the instantiations the launch has (nlmeans.hip): patch radius 2 on <9, 6> / fused <9, 7>, patch radius 1 on <9, 7> either way
  const bool p1 = patch_radius == 1;
  const bool takes = p1 ? (fused ? nlm3::fits_fused<9, 7, 1>(chk_w, chk_h, patch_radius, a.reach) : nlm3::fits<9, 7, 1>(chk_w, chk_h, patch_radius, a.reach))
                        : (fused ? nlm3::fits_fused<9, 7>(chk_w, chk_h, patch_radius, a.reach) : nlm3::fits<9, 6>(chk_w, chk_h, patch_radius, a.reach));
  if(!takes || !nlm3::regular_grid(patches.data(), a.npatch, &ndx)) return 0;
  const size_t lds_floats = p1 ? (fused ? nlm3::lds_floats_fused<9, 1>(chk_h, a.reach) : nlm3::lds_floats<9, 1>(chk_h, a.reach))
                               : (fused ? nlm3::lds_floats_fused<9>(chk_h, a.reach) : nlm3::lds_floats<9>(chk_h, a.reach));
  if(lds_floats * sizeof(float) > 160 * 1024) return 0;
  int interior = 0;
  for(int cy = 0; cy < nchy; cy++)
    for(int cx = 0; cx < a.nchx; cx++)
    {
      const int top = cy * chk_h, left = cx * chk_w;
      const int bot = std::min(top + chk_h, H), right = std::min(left + chk_w, W);
      if(top >= a.reach && bot + a.reach <= H && left >= a.reach && right + a.reach <= W && bot - top == chk_h && right - left == chk_w)
        interior++;
      else if(border && nlm3::border_fits(right - left, bot - top))
        interior++;
    }
  if(interior_chunks) *interior_chunks = interior;
  const F4 *const fin = (const F4 *)in;
  F4 *const fout = (F4 *)out;
  const int nch = a.nchx * nchy;
  if(p1)
  {
    if(fused && center) run3<9, 7, true, 1, true>(fin, fout, a, patches.data(), nch, lds_floats, ndx, border);
    else if(fused) run3<9, 7, true, 1, false>(fin, fout, a, patches.data(), nch, lds_floats, ndx, border);
    else if(center) run3<9, 7, false, 1, true>(fin, fout, a, patches.data(), nch, lds_floats, ndx, border);
    else run3<9, 7, false, 1, false>(fin, fout, a, patches.data(), nch, lds_floats, ndx, border);
  }
  else if(center)
  {
    if(fused) run3<9, 7, true, 2, true>(fin, fout, a, patches.data(), nch, lds_floats, ndx, border);
    else run3<9, 6, false, 2, true>(fin, fout, a, patches.data(), nch, lds_floats, ndx, border);
  }
  else if(fused) run3<9, 7, true>(fin, fout, a, patches.data(), nch, lds_floats, ndx, border);
  else run3<9, 6>(fin, fout, a, patches.data(), nch, lds_floats, ndx, border);
  return 1;
}

// ---- a tall chunk grid (65 - 69 rows; round 5): the fused body on the first 64 rows of every interior chunk, exporting the
//      column sums behind them (nlm3_body.h TALL), then nlm_tail_body.h on the rows that are left.  Interior chunks only
//      (the device's launch hands the outermost ring to the first version's body).  Returns 1 when it ran, 0 when the
//      configuration is not one the pair takes.
static int nlm_tall_host_run_(const float *in, float *out, int W, int H, int chk_w, int chk_h, int patch_radius,
                              int search_radius, float scale, float scattering, float sharpness, const float *norm,
                              float luma, float chroma, int *interior_chunks, const bool border, const float center_weight = -1.0f);
// round 6: patch radius 1 or 2, either weight
extern "C" int nlm_tall_host_run_ex(const float *in, float *out, int W, int H, int chk_w, int chk_h, int patch_radius,
                                    int search_radius, float scale, float scattering, float sharpness, const float *norm,
                                    float luma, float chroma, int *chunks, float center_weight, int border)
{
  return nlm_tall_host_run_(in, out, W, H, chk_w, chk_h, patch_radius, search_radius, scale, scattering, sharpness, norm, luma, chroma,
                            chunks, border != 0, center_weight);
}
extern "C" int nlm_tall_host_run(const float *in, float *out, int W, int H, int chk_w, int chk_h, int patch_radius,
                                 int search_radius, float scale, float scattering, float sharpness, const float *norm,
                                 float luma, float chroma, int *interior_chunks)
{
  return nlm_tall_host_run_(in, out, W, H, chk_w, chk_h, patch_radius, search_radius, scale, scattering, sharpness, norm, luma, chroma,
                            interior_chunks, false);
}
// ... and the outermost ring with the BORDER bodies; *chunks = the chunks written
extern "C" int nlm_tall_host_run_all(const float *in, float *out, int W, int H, int chk_w, int chk_h, int patch_radius,
                                     int search_radius, float scale, float scattering, float sharpness, const float *norm,
                                     float luma, float chroma, int *chunks)
{
  return nlm_tall_host_run_(in, out, W, H, chk_w, chk_h, patch_radius, search_radius, scale, scattering, sharpness, norm, luma, chroma,
                            chunks, true);
}
namespace
{
template <int P, bool CENTER>
void run_tall(const F4 *fin, F4 *fout, const Args &a, const I2 *patches, const int nchunks, const int ndx, const size_t head_floats,
              const size_t tail_floats, const bool border)
{
  const size_t per_chunk = (size_t)a.npatch * NLT_SEED_PITCH;
  std::vector<float> seeds(per_chunk * nchunks, __builtin_nanf(""));
  {
    std::vector<float> lds(head_floats + 4096, 0.0f);
    WaveExchange xch;
    float *base = lds.data();
    while((uintptr_t)base & 15) base++;
    std::barrier<> bar(NL3_THREADS);
    std::vector<std::thread> pool;
    for(int t = 0; t < NL3_THREADS; t++)
      pool.emplace_back([&, t]() {
        for(int b = 0; b < nchunks; b++)
        {
          HostEnv env{ t, b, base, &bar, &xch };
          nlm3::body<9, 7, false, true, true, P, CENTER>(env, fin, fout, a, patches, ndx, seeds.data() + per_chunk * b);
          bar.arrive_and_wait();
          if(border)
          {
            HostEnv envb{ t, b, base, &bar, &xch };
            nlm3::body<9, 7, true, true, true, P, CENTER>(envb, fin, fout, a, patches, ndx, seeds.data() + per_chunk * b);
            bar.arrive_and_wait();
          }
        }
      });
    for(auto &th : pool) th.join();
  }
  {
    std::vector<float> lds(tail_floats + 4096, 0.0f);
    float *base = lds.data();
    while((uintptr_t)base & 15) base++;
    std::barrier<> bar(NLT_THREADS);
    WaveExchange xch;
    std::vector<std::thread> pool;
    for(int t = 0; t < NLT_THREADS; t++)
      pool.emplace_back([&, t]() {
        for(int b = 0; b < nchunks; b++)
        {
          HostEnv env{ t, b, base, &bar, &xch };
          nlmt::body<false, P, CENTER>(env, fin, fout, a, patches, seeds.data() + per_chunk * b);
          bar.arrive_and_wait();
          if(border)
          {
            HostEnv envb{ t, b, base, &bar, &xch };
            nlmt::body<true, P, CENTER>(envb, fin, fout, a, patches, seeds.data() + per_chunk * b);
            bar.arrive_and_wait();
          }
        }
      });
    for(auto &th : pool) th.join();
  }
}
} // namespace
static int nlm_tall_host_run_(const float *in, float *out, int W, int H, int chk_w, int chk_h, int patch_radius,
                              int search_radius, float scale, float scattering, float sharpness, const float *norm,
                              float luma, float chroma, int *interior_chunks, const bool border, const float center_weight)
{
  std::vector<I2> patches;
  int max_shift = 0;
  for(int ri = -search_radius; ri <= search_radius; ri++)
    for(int ci = -search_radius; ci <= search_radius; ci++)
    {
      const int r = scatter(scale, scattering, ri, ci), c = scatter(scale, scattering, ci, ri);
      patches.push_back(I2{ r, c });
      max_shift = std::max(max_shift, std::max(abs(r), abs(c)));
    }
  Args a;
  memset(&a, 0, sizeof(a));
  a.W = W;
  a.H = H;
  a.chk_w = chk_w;
  a.chk_h = chk_h;
  a.nchx = (W + chk_w - 1) / chk_w;
  const int nchy = (H + chk_h - 1) / chk_h;
  a.radius = patch_radius;
  a.npatch = (int)patches.size();
  a.sharpness = sharpness;
  for(int k = 0; k < 3; k++) a.norm[k] = norm[k];
  a.luma = luma;
  a.chroma = chroma;
  a.skip_blend = (luma == 1.0 && chroma == 1.0);
  a.reach = patch_radius + 1 + max_shift;
  a.cy0 = 0;
  a.out_row0 = 0;
  a.out_row1 = H;
  a.center_weight = center_weight;
  a.cpn = center_weight * (2 * patch_radius + 1) * (2 * patch_radius + 1);
  const bool center = !(center_weight < 0), p1 = patch_radius == 1;
  int ndx = 0;
  const bool takes = p1 ? nlmt::fits<1>(chk_w, chk_h, patch_radius, a.reach, a.npatch) && nlm3::fits_fused<9, 7, 1>(chk_w, NLT_HEAD_ROWS, patch_radius, a.reach)
                        : nlmt::fits<2>(chk_w, chk_h, patch_radius, a.reach, a.npatch) && nlm3::fits_fused<9, 7>(chk_w, NLT_HEAD_ROWS, patch_radius, a.reach);
  if(!takes || !nlm3::regular_grid(patches.data(), a.npatch, &ndx)) return 0;
  const size_t head_floats = p1 ? nlm3::lds_floats_fused<9, 1>(NLT_HEAD_ROWS, a.reach) : nlm3::lds_floats_fused<9>(NLT_HEAD_ROWS, a.reach);
  const size_t tail_floats = nlmt::lds_floats(chk_h - NLT_HEAD_ROWS, a.reach, a.npatch);
  if(head_floats * sizeof(float) > 160 * 1024 || tail_floats * sizeof(float) > 64 * 1024) return 0;
  int interior = 0;
  for(int cy = 0; cy < nchy; cy++)
    for(int cx = 0; cx < a.nchx; cx++)
    {
      const int top = cy * chk_h, left = cx * chk_w;
      const int bot = std::min(top + chk_h, H), right = std::min(left + chk_w, W);
      if(top >= a.reach && bot + a.reach <= H && left >= a.reach && right + a.reach <= W && bot - top == chk_h && right - left == chk_w)
        interior++;
      else if(border && nlm3::border_fits(right - left, std::min(bot - top, NLT_HEAD_ROWS)))
        interior++;
    }
  if(interior_chunks) *interior_chunks = interior;
  const int nchunks = a.nchx * nchy;
  const F4 *const fin = (const F4 *)in;
  F4 *const fout = (F4 *)out;
  if(p1 && center) run_tall<1, true>(fin, fout, a, patches.data(), nchunks, ndx, head_floats, tail_floats, border);
  else if(p1) run_tall<1, false>(fin, fout, a, patches.data(), nchunks, ndx, head_floats, tail_floats, border);
  else if(center) run_tall<2, true>(fin, fout, a, patches.data(), nchunks, ndx, head_floats, tail_floats, border);
  else run_tall<2, false>(fin, fout, a, patches.data(), nchunks, ndx, head_floats, tail_floats, border);
  return 1;
}
