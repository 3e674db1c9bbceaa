// diffuse.hip -- diffuse or sharpen on gfx950.
//
// Reference: process(), src/iop/diffuse.c:1155-1258; wavelets_process() :978-1106;
// decompose_2D_Bspline(), src/pixel/bspline.h:351-377; heat_PDE_diffusion(), diffuse.c:760-968.
// (process_cl() :1436-1580 runs the same two steps as OpenCL kernels diffuse_pde / filmic_bspline_*.)
//
// Per iteration: `scales` a-trous B-spline analyses (dilation 2^s) that each leave a detail plane
// HF[s] and a low-pass plane, then `scales` PDE updates from coarse to fine, each a 3x3 stencil at
// dilation 2^s over HF[s] and the running low-pass.  Everything is a float4 plane resident in HBM;
// algorithmic traffic is 48 B/px per analysis (in -> LF + HF) and 48 B/px per PDE (HF + LF -> out).
//
// Device layout of the analysis: the reference blurs vertically into a row buffer, clips negatives,
// blurs that row horizontally, clips again.  Here one workgroup owns a piece of one row: the
// vertically blurred, clipped samples it needs are built in LDS (5 global taps each, L2/MALL hits
// because workgroups walk the rows in dilation order), the horizontal taps then come from LDS.  At
// dilation m the horizontal taps of column j are j-2m..j+2m, so a workgroup takes R adjacent
// columns x T steps of m (R*16 B = one 128-byte line per step) and pays a (T+4)/T halo.  Border
// taps clamp to column 0 / width-1 whatever the dilation (bspline.h:143-149), so those two columns
// are blurred by every workgroup as well.
#include "hip_common.h"

#include <type_traits>
#include "devmath.h"
#include "ieee_inrange.h"
#include "px_colorspaces.h"

#include <math.h>
#include <cstdlib>

namespace ansel
{
// diffuse_bspline.hip: one a-trous B-spline analysis in -> (hf, lf) at dilation mult
int bspline_launch_decompose(int devid, hipStream_t s, const float4 *in, float4 *hf, float4 *lf, int w, int h, int mult);
#ifdef ANSEL_HIP_MEASURING
// ... two scales (dilations mult and 2 mult, mult 1 or 4) in one pass: in -> low1, low2 (diffuse_bspline.hip; no faster)
int bspline_launch_decompose2(int devid, hipStream_t s, const float4 *in, float4 *low1, float4 *low2, int w, int h, int mult);
#endif
}
using namespace ansel;

namespace
{

#define DIFFUSE_MAX_SCALES 10             // MAX_NUM_SCALES, diffuse.c:75
#define BSPLINE_SIGMA 1.0553651328015339f // B_SPLINE_SIGMA, bspline.h:39
#define PDE_KAPPA 0.25f                   // KAPPA, diffuse.c:625

// the MAX(a, b) macro of the reference: a > b ? a : b (a NaN in b passes through)
__device__ __forceinline__ float max_first(const float a, const float b) { return a > b ? a : b; }
// dt_simd_max_zero(), src/system/simd.h:108-114
__device__ __forceinline__ float max_zero(const float v) { return isfinite(v) ? max_first(v, 0.0f) : 0.0f; }
__device__ __forceinline__ int clampi(const int v, const int lo, const int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// row walked by workgroup row `b`: rows of one dilation class back to back (0, m, 2m, ... then
// 1, m+1, ...), the order dwt_interleave_rows() (src/pixel/dwt.h:93-104) gives the CPU for the
// same reason -- consecutive workgroups share 4 of their 5 vertical taps.  Returns -1 past the end.
__device__ __forceinline__ int walk_row(const int b, const int height, const int mult)
{
  if(height <= mult) return b < height ? b : -1;
  const int per_pass = (height + mult - 1) / mult;
  const int row = (b % per_pass) * mult + b / per_pass;
  return row < height ? row : -1;
}

struct pde_args
{
  int width, height, mult;
  float anisotropy[4];
  int kind[4]; // 0 isotrope, 1 isophote, 2 gradient (check_isotropy_mode(), diffuse.c:151-161)
  int same02, same13; // orders 1 and 3 (2 and 4) have the same kind and anisotropy: one kernel for both
  float variance_threshold, regularization;
  float abcd[4], strength;
  int wskip;
  int post_lab;        // the pipe's RGB -> Lab glue behind the module, applied where the last pass stores (strip kernel)
  float post_m[3][4];
  int approx_div;      // measuring builds only (ANSEL_HIP_PDE_APPROX_DIV): PDE_APPROX below
  int off;             // measuring builds only (ANSEL_HIP_PDE_OFF): parts of the strip kernel switched off, PDE_OFF below
};

// What would the north star's 1 ULP buy?  Measuring builds can run the strip kernel with every division as v_rcp, one
// product and ONE residual correction (4 instructions instead of 11; <= 1 ulp) and the square root as the bare v_sqrt_f32
// (1 ulp): tools/pde_div_ab.py times both arms and commits the ULP histogram of the module's output beside the
// milliseconds (profiles/r05_pde_div_ab.json).  Never in the product: PDE_APPROX() is the constant 0 there.
#ifdef ANSEL_HIP_MEASURING
#define PDE_APPROX(a) ((a).approx_div != 0)
// timing experiments (wrong results): 1 no barrier, 2 no fetches behind the strip's first rows (the registers keep what they
// hold), 4 no squared-ratio ring (no LDS stores / reads; the energy is a constant), 8 no store, 16 no update arithmetic (the
// output is a sample) -- tools/pde_off_ab.py times the strip kernel with each switched off in turn
#define PDE_OFF(a, bit) (((a).off & (bit)) != 0)
#else
#define PDE_APPROX(a) false
#define PDE_OFF(a, bit) false
#endif
__device__ __forceinline__ float div_1ulp(const float x, const float y)
{
  const float r = __builtin_amdgcn_rcpf(y);
  const float q = x * r;
  return __builtin_fmaf(__builtin_fmaf(-y, q, x), r, q);
}

// dt_fast_expf(), src/math/math.h:254-267.  The float -> int conversion of an out-of-range or NaN
// value is INT_MIN on the reference's target (cvttss2si), which its k0 > 0 test turns into 0.
__device__ __forceinline__ float fast_expf(const float x)
{
  const float t = 1065353216.0f + x * 11401300.0f;
  int k = (t > -2147483648.0f && t < 2147483648.0f) ? (int)t : 0;
  k = k > 0 ? k : 0;
  return __int_as_float(k);
}

// diffuse.c:851-866: magnitude of a 2-vector and {cos^2, sin^2, cos*sin} of its argument
// Round 5: the square root and the division without their range scaffolding (ieee_inrange.h) where every lane of the
// wave has gx gx + gy gy == +0 or in [2^-96, +inf) -- one multiplication, one class test and one ballot per call: the
// magnitude is then +0 or in [2^-48, 2^64), the divisor 1 or that magnitude -- inside the range in which the division's
// scale / fix-up instructions do nothing -- and the numerator is 1.  Same operations otherwise, so the same bits; a wave
// with a gradient outside (below 2^-48: dark pixels one ulp apart; not finite) takes the long forms.  The two forms
// rejoin at once: a channel's update stays one block for the scheduler, and only the magnitude and the reciprocal leave
// the branch (the non-zero flag is formed behind it: with it inside, the modes with two directions needed 130 registers).
__device__ __forceinline__ float direction(float gx, float gy, float &cos2, float &sin2, float &cs, const bool approx = false)
{
  const float s2 = gx * gx + gy * gy;
  float mag, inv;
  if(approx)
  {
    mag = __builtin_amdgcn_sqrtf(s2);
    inv = div_1ulp(1.0f, mag + (1.0f - ((mag != 0.0f) ? 1.0f : 0.0f)));
  }
  else if(__builtin_amdgcn_ballot_w64(!ansel_ieee::zero_or_above_2m96(s2)) == 0ull)
  {
    mag = ansel_ieee::sqrt_core(s2);
    inv = ansel_ieee::rcp_core(mag + (1.0f - ((mag != 0.0f) ? 1.0f : 0.0f)));
  }
  else
  {
    mag = sqrtf(s2);
    inv = 1.0f / (mag + (1.0f - ((mag != 0.0f) ? 1.0f : 0.0f)));
  }
  const float nonzero = (mag != 0.0f) ? 1.0f : 0.0f;
  gx = gx * inv + (1.0f - nonzero);
  gy = gy * inv;
  cos2 = gx * gx;
  sin2 = gy * gy;
  cs = gx * gy;
  return mag;
}

// compute_kernel(), diffuse.c:725-757: the five distinct weights of the 3 x 3 kernel of one order (k[0] the corners
// 0 and 8, k[1] top and bottom, k[2] = -k[0] the other diagonal, k[3] left and right, k[4] the centre)
struct kernel5
{
  float k0, k1, k2, k3, k4;
};
__device__ __forceinline__ kernel5 order_kernel(const int kind, const float c2, const float cs, const float cos2, const float sin2)
{
  kernel5 w;
  if(kind == 0)
  {
    w.k0 = 0.25f; w.k1 = 0.5f; w.k2 = 0.25f; w.k3 = 0.5f; w.k4 = -3.0f; // isotrope_laplacian(), :705-723
  }
  else
  {
    float a00, a11, a01;
    if(kind == 1)
    {
      a00 = cos2 + c2 * sin2; // rotation_matrix_isophote(), :646-659
      a11 = c2 * cos2 + sin2;
      a01 = (c2 - 1.0f) * cs;
    }
    else
    {
      a00 = c2 * cos2 + sin2; // rotation_matrix_gradient(), :661-674
      a11 = cos2 + c2 * sin2;
      a01 = (1.0f - c2) * cs;
    }
    w.k0 = a01 * 0.5f; // build_matrix(), :677-703
    w.k1 = a11;
    w.k2 = -w.k0;
    w.k3 = a00;
    w.k4 = -2.0f * (a00 + a11);
  }
  return w;
}

// ... convolved on the spot with the 9 samples p[] (k = 0..8 order of the reference's accumulation loop, :913-919):
// d = kern[k] * p[k] + d
__device__ __forceinline__ float convolve(const kernel5 w, const float p[9])
{
  float d = 0.0f;
  d = w.k0 * p[0] + d;
  d = w.k1 * p[1] + d;
  d = w.k2 * p[2] + d;
  d = w.k3 * p[3] + d;
  d = w.k4 * p[4] + d;
  d = w.k3 * p[5] + d;
  d = w.k2 * p[6] + d;
  d = w.k1 * p[7] + d;
  d = w.k0 * p[8] + d;
  return d;
}

// (HF / LF)^2 of one sample, diffuse.c:829-841.  Each pixel is a neighbour of nine centres: the workgroup computes
// the squared ratio ONCE per sample of its rows (diffuse_pde stages them in LDS) instead of nine times -- the
// same three operations on the same operands, so the same binary32 value
__device__ __forceinline__ float ratio2(const float h, const float l)
{
  const float safe = max_zero(l - 1e-8f) + 1e-8f;
  const float ratio = h / safe;
  return ratio * ratio;
}
// The three colour channels' squared ratios of one sample, the division's range scaffolding left out where a wave's
// operands allow it (round 5; ieee_inrange.h).  The divisor is max_zero(l - 1e-8) + 1e-8: finite, never NaN, >= 1e-8 > 2^-27.
// One test per sample -- the largest of |h| and of the divisors over the three channels <= 2^64 -- and one ballot per
// wave: then every divisor is a normal number below 2^126 and exponent(h) - exponent(divisor) <= 64 + 27 < 96, which is
// all of div_core()'s domain except the small numerators -- |h| < 2^-103, or a subnormal quotient -- and the zeros and
// NaNs.  Those do not reach the result: a quotient below 2^-76 in magnitude, however its last bits fall, squares to +0
// (2^-152 is below half the smallest subnormal), as +-0 does; and a NaN numerator gives a NaN either way, whose only
// reader is the energy sum that max_zero() (not finite -> 0) closes.  (v_max3 skips a NaN numerator; an infinite one fails
// the test.)  A wave that fails -- magnitudes no image carries -- divides the long way.
__device__ __forceinline__ float4 ratio2_rgb(const float4 h, const float4 l, const float w_ratio, const bool approx = false)
{
  if(approx)
  {
    const float ax = div_1ulp(h.x, max_zero(l.x - 1e-8f) + 1e-8f), ay = div_1ulp(h.y, max_zero(l.y - 1e-8f) + 1e-8f),
                az = div_1ulp(h.z, max_zero(l.z - 1e-8f) + 1e-8f);
    return make_float4(ax * ax, ay * ay, az * az, w_ratio);
  }
  const float sx = max_zero(l.x - 1e-8f) + 1e-8f, sy = max_zero(l.y - 1e-8f) + 1e-8f, sz = max_zero(l.z - 1e-8f) + 1e-8f;
  const float top = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(h.x), __builtin_fabsf(h.y)), __builtin_fabsf(h.z)),
                                    __builtin_fmaxf(__builtin_fmaxf(sx, sy), sz));
  float rx, ry, rz;
  if(__builtin_amdgcn_ballot_w64(!(top <= 0x1p64f)) == 0ull)
  {
    rx = ansel_ieee::div_core(h.x, sx);
    ry = ansel_ieee::div_core(h.y, sy);
    rz = ansel_ieee::div_core(h.z, sz);
  }
  else
  {
    rx = h.x / sx;
    ry = h.y / sy;
    rz = h.z / sz;
  }
  return make_float4(rx * rx, ry * ry, rz * rz, w_ratio);
}

// The four orders' kinds and the two "same kernel" flags are parameters, and branching on them (uniformly) cuts a channel's
// update into a dozen basic blocks that the scheduler cannot interleave with each other or with the next channel.  MODE >= 0
// spells them at compile time -- kind[0] + 3 kind[1] + 9 kind[2] + 27 kind[3] + 81 same02 + 162 same13 -- for the
// combinations the presets use (the strip kernel is instantiated for them); MODE < 0 reads the arguments.
// (the presets of init_presets(), diffuse.c:298-583, by the signs of their four anisotropies)
#define PDE_MODE_ISOTROPIC (81 + 162)                 // the module's defaults, bloom: four isotropic orders
#define PDE_MODE_DEBLUR (1 + 9 + 81 + 162)            // lens deblur, dehaze, denoise: orders 1 and 3 along the isophotes, 2 and 4 isotropic
#define PDE_MODE_ISOPHOTE (1 + 3 + 9 + 27 + 81 + 162) // surface blur, sharpen demosaicing: all four along the isophotes
#define PDE_MODE_GRADIENT (2 + 6 + 18 + 54 + 81 + 162) // simulate line drawing: all four along the gradients
#define PDE_MODE_WATERCOLOR (9 + 27)                  // simulate watercolor: orders 3 and 4 along the isophotes
#define PDE_MODE_CONTRAST (2 + 54)                    // add local contrast: orders 1 and 4 along the gradients
#define PDE_MODE_INPAINT (27)                         // inpaint highlights: order 4 along the isophotes
#define PDE_MODE_FAST (9)                             // fast sharpness, fast local contrast: order 3 along the isophotes
#define PDE_MODES(X) X(PDE_MODE_ISOTROPIC) X(PDE_MODE_DEBLUR) X(PDE_MODE_ISOPHOTE) X(PDE_MODE_GRADIENT) X(PDE_MODE_WATERCOLOR) \
  X(PDE_MODE_CONTRAST) X(PDE_MODE_INPAINT) X(PDE_MODE_FAST)
template <int MODE> struct pde_mode
{
  static __device__ __forceinline__ int kind(const int i, const pde_args &a)
  {
    if constexpr(MODE < 0) return a.kind[i];
    else return (i == 0 ? MODE : (i == 1 ? MODE / 3 : (i == 2 ? MODE / 9 : MODE / 27))) % 3;
  }
  static __device__ __forceinline__ bool same02(const pde_args &a)
  {
    if constexpr(MODE < 0) return a.same02 != 0;
    else return (MODE / 81) % 2 != 0;
  }
  static __device__ __forceinline__ bool same13(const pde_args &a)
  {
    if constexpr(MODE < 0) return a.same13 != 0;
    else return (MODE / 162) % 2 != 0;
  }
};
inline int pde_mode_of(const pde_args &a)
{
  return a.kind[0] + 3 * a.kind[1] + 9 * a.kind[2] + 27 * a.kind[3] + 81 * (a.same02 ? 1 : 0) + 162 * (a.same13 ? 1 : 0);
}

// energy: the sum of the squared ratios of the nine samples, added in the order of the reference's loop
template <int MODE = -1>
__device__ __forceinline__ float pde_channel(const float H[9], const float L[9], float energy, const pde_args &a)
{
  typedef pde_mode<MODE> M;
  // HF/LF energy over the 3x3 support, diffuse.c:823-845
  energy = max_zero(a.variance_threshold + energy * a.regularization - 1e-8f) + 1e-8f;
  // the direction of the low-frequency gradient steers orders 1 and 3, that of the high-frequency one orders 2 and 4; an
  // isotropic order (anisotropy 0, the module's default for all four) reads neither the angle nor the magnitude, so a
  // direction nobody reads is not computed (uniform branches: the kinds are parameters)
  float cos2g = 0.f, sin2g = 0.f, csg = 0.f, cos2l = 0.f, sin2l = 0.f, csl = 0.f, mg = 0.f, ml = 0.f;
  if(M::kind(0, a) | M::kind(2, a)) mg = direction((L[7] - L[1]) * 0.5f, (L[5] - L[3]) * 0.5f, cos2g, sin2g, csg, PDE_APPROX(a));
  if(M::kind(1, a) | M::kind(3, a)) ml = direction((H[7] - H[1]) * 0.5f, (H[5] - H[3]) * 0.5f, cos2l, sin2l, csl, PDE_APPROX(a));
  // orders 1 and 3 share the direction of the low-frequency gradient, orders 2 and 4 that of the high-frequency one; with
  // the same anisotropy (the presets' case) they share the kernel too (uniform: the anisotropies are parameters)
  const kernel5 w0 = order_kernel(M::kind(0, a), fast_expf(-mg * a.anisotropy[0]), csg, cos2g, sin2g);
  const kernel5 w1 = order_kernel(M::kind(1, a), fast_expf(-ml * a.anisotropy[1]), csl, cos2l, sin2l);
  const kernel5 w2 = M::same02(a) ? w0 : order_kernel(M::kind(2, a), fast_expf(-mg * a.anisotropy[2]), csg, cos2g, sin2g);
  const kernel5 w3 = M::same13(a) ? w1 : order_kernel(M::kind(3, a), fast_expf(-ml * a.anisotropy[3]), csl, cos2l, sin2l);
  const float d0 = convolve(w0, L);
  const float d1 = convolve(w1, L);
  const float d2 = convolve(w2, H);
  const float d3 = convolve(w3, H);
  float update = d0 * a.abcd[0];
  update = d1 * a.abcd[1] + update;
  update = d2 * a.abcd[2] + update;
  update = d3 * a.abcd[3] + update;
  const float acc = H[4] * a.strength + (PDE_APPROX(a) ? div_1ulp(update, energy) : update / energy);
  return max_zero(acc + L[4]);
}

// The fourth channel is a channel like the others to the reference (it loops over four), and in a pipe it is all +0 where
// this module runs: colorin's matrix leaves 0 x + 0 y + 0 z there.  With +0 in all 18 samples of a support the update is
// +0 whatever the parameters are -- every squared ratio is +0 / 1e-8, the gradients are +0 (direction (1, 0), magnitude 0),
// each convolution adds +-0 products onto +0, update / energy is +-0 or NaN and max_zero(that + +0) is +0 either way -- so a
// wave whose lanes all see that (one ballot) skips the channel: a quarter of the kernel's arithmetic.  Any other bit
// pattern, -0 included, takes the full path.
__device__ __forceinline__ bool alpha_is_blank(const float4 H4[9], const float4 L4[9])
{
  unsigned bits = 0;
#pragma unroll
  for(int k = 0; k < 9; k++) bits |= __float_as_uint(H4[k].w) | __float_as_uint(L4[k].w);
  return __builtin_amdgcn_ballot_w64(bits != 0) == 0ull;
}

// SHARED (dilations up to PDE_SHARED_MULT, where the three column sets of a 256-pixel segment overlap): every thread
// squares the ratios of the CENTRE column of its support -- three samples it has fetched anyway -- into LDS, the first
// and last `mult` threads also those of their left / right column, and the nine ratios of a support are read back
// from there: 12 - 24 divisions per pixel instead of 36, no extra fetch.  Else every thread computes its own nine.
#define PDE_SHARED_MULT 128
template <bool SHARED>
__global__ __launch_bounds__(256) void diffuse_pde(const float4 *__restrict__ hf, const float4 *__restrict__ lf,
                                                   float4 *__restrict__ out, const pde_args a, const int final_pass,
                                                   const unsigned char *__restrict__ mask)
{
  extern __shared__ float4 r2s[]; // SHARED: [3][256 + 2 * mult]; slot x of a row = column clamp(seg - mult + x)
  const int row = walk_row(blockIdx.y, a.height, a.mult);
  const int bx = xcd_col(); // hip_common.h: column block pinned to an XCD for 64 rows
  const int col = bx * (int)blockDim.x + (int)threadIdx.x;
  if(row < 0 || bx * (int)blockDim.x >= a.width) return; // uniform over the workgroup
  const size_t rows[3] = { (size_t)clampi(row - a.mult, 0, a.height - 1) * a.width, (size_t)row * a.width,
                           (size_t)clampi(row + a.mult, 0, a.height - 1) * a.width };
  // threads past the end of the row keep fetching (clamped): their samples are the clamped columns of their neighbours
  const int cols[3] = { clampi(col - a.mult, 0, a.width - 1), clampi(col, 0, a.width - 1), clampi(col + a.mult, 0, a.width - 1) };
  float4 H4[9], L4[9];
#pragma unroll
  for(int ii = 0; ii < 3; ii++)
#pragma unroll
    for(int jj = 0; jj < 3; jj++)
    {
      H4[3 * ii + jj] = hf[rows[ii] + cols[jj]];
      L4[3 * ii + jj] = lf[rows[ii] + cols[jj]];
    }
  const int tw = 256 + 2 * a.mult;
  const int tx = threadIdx.x;
  float4 energy = make_float4(0.f, 0.f, 0.f, 0.f); // energy += ratio * ratio over k = 0..8, diffuse.c:829-841
  if(SHARED)
  {
#pragma unroll
    for(int ii = 0; ii < 3; ii++)
    {
      const float4 h = H4[3 * ii + 1], l = L4[3 * ii + 1];
      r2s[ii * tw + tx + a.mult] = make_float4(ratio2(h.x, l.x), ratio2(h.y, l.y), ratio2(h.z, l.z), ratio2(h.w, l.w));
    }
    if(tx < a.mult)
    {
#pragma unroll
      for(int ii = 0; ii < 3; ii++)
      {
        const float4 h = H4[3 * ii], l = L4[3 * ii];
        r2s[ii * tw + tx] = make_float4(ratio2(h.x, l.x), ratio2(h.y, l.y), ratio2(h.z, l.z), ratio2(h.w, l.w));
      }
    }
    if(tx >= 256 - a.mult)
    {
#pragma unroll
      for(int ii = 0; ii < 3; ii++)
      {
        const float4 h = H4[3 * ii + 2], l = L4[3 * ii + 2];
        r2s[ii * tw + tx + 2 * a.mult] = make_float4(ratio2(h.x, l.x), ratio2(h.y, l.y), ratio2(h.z, l.z), ratio2(h.w, l.w));
      }
    }
    __syncthreads();
    if(col >= a.width) return;
#pragma unroll
    for(int ii = 0; ii < 3; ii++)
    {
#pragma unroll
      for(int jj = 0; jj < 3; jj++)
      {
        const float4 r = r2s[ii * tw + tx + jj * a.mult];
        energy.x += r.x;
        energy.y += r.y;
        energy.z += r.z;
        energy.w += r.w;
      }
      // three reads in flight, not nine (they sit on top of the 72 registers of the support): the next row's reads stay
      // behind this statement, which needs the sums of this row
      asm volatile("" : "+v"(energy.x), "+v"(energy.y), "+v"(energy.z), "+v"(energy.w) : : "memory");
    }
  }
  else
  {
    if(col >= a.width) return;
#pragma unroll
    for(int k = 0; k < 9; k++)
    {
      energy.x += ratio2(H4[k].x, L4[k].x);
      energy.y += ratio2(H4[k].y, L4[k].y);
      energy.z += ratio2(H4[k].z, L4[k].z);
      energy.w += ratio2(H4[k].w, L4[k].w);
    }
  }
  if(mask && !mask[(size_t)row * a.width + col])
  {
    // outside the luminance mask: "only copy input to output", diffuse.c:927-937
    const size_t c = (size_t)row * a.width + col;
    const float4 h = H4[4], l = L4[4];
    const float4 o = make_float4(max_zero(h.x + l.x), max_zero(h.y + l.y), max_zero(h.z + l.z), max_zero(h.w + l.w));
    if(final_pass) nt_store(out + c, o);
    else out[c] = o;
    return;
  }
  float H[9], L[9];
  float4 o;
#pragma unroll
  for(int k = 0; k < 9; k++) { H[k] = H4[k].x; L[k] = L4[k].x; }
  o.x = pde_channel(H, L, energy.x, a);
#pragma unroll
  for(int k = 0; k < 9; k++) { H[k] = H4[k].y; L[k] = L4[k].y; }
  o.y = pde_channel(H, L, energy.y, a);
#pragma unroll
  for(int k = 0; k < 9; k++) { H[k] = H4[k].z; L[k] = L4[k].z; }
  o.z = pde_channel(H, L, energy.z, a);
  if(alpha_is_blank(H4, L4))
    o.w = 0.0f;
  else
  {
#pragma unroll
    for(int k = 0; k < 9; k++) { H[k] = H4[k].w; L[k] = L4[k].w; }
    o.w = pde_channel(H, L, energy.w, a);
  }
  const size_t idx = rows[1] + col;
  if(final_pass) nt_store(out + idx, o);
  else out[idx] = o;
}

// The same update on STRIPS: a workgroup keeps its 256 columns for up to `strip` rows of one dilation class (rows c,
// c + m, c + 2 m, ...).  The 3 x 3 supports of consecutive rows of a class share two of their three rows, so a lane
// rolls the three rows of its three columns through registers and fetches ONE new row (6 float4) per output row
// instead of three (18), and the squared ratios go into a ring of four LDS rows, one NEW row per output row: 4 - 8
// divisions per pixel where the per-row kernel has 12 - 24.  Same operands, same operations, same order.
// HSUB: the high-frequency plane is not in memory -- `hf` is the low-pass plane of this scale, `hsub` that of the next
// coarser one, and a support sample is their difference, the subtraction decompose_2D_Bspline() stores (bspline.h:369-374:
// same operands, same operation).  The analysis then writes 16 B per pixel and scale instead of 32, and this kernel, which
// waits for its arithmetic and not for its fetches, reads 48 instead of 32.
// DMA (round 5, HSUB only): the kernel has two bounds of the same size -- its bytes and its update's arithmetic (11.7 and
// 12.0 ms of 16.9) -- that overlapped by half: a wave fetched its row, waited, then computed with nothing in flight.  Here the rows
// arrive by LDS-DMA (global_load_lds_dwordx4: no destination registers, so the fetch of support row kk + 3 is in flight during ALL of
// output row kk's arithmetic, and nothing the register allocator does can touch data that has not landed).  Each WAVE owns a landing zone
// of [3 planes][64 + 2 mult] samples -- its own 64 columns and the mult either side: what its lanes' three support columns span -- fills
// it itself (each lane one sample per plane, the first 2 mult lanes one more) and reads it back after ITS OWN `s_waitcnt vmcnt(0)`: no
// barrier beyond the ratio ring's.  A sample is fetched once per wave (1.03 - 1.5 times per workgroup) instead of by three lanes.  The
// fetches are inline assembly (hipcc waits vmcnt(0) at every barrier for an LDS-DMA it knows of); the only vector-memory operation
// the compiler sees in the loop is the store.
#define PDE_RING 4
#define PDE_DMA_MAX_MULT 16
// three planes' pieces of one landing-zone row segment: lane l's 16 bytes at base + voff go to LDS byte dst + 16 l.  M0 carries the
// destination and is the compiler's: saved and restored inside the statement.  lgkmcnt(0) first: the zone's previous contents have been
// READ (the ds_reads have returned) before anything is sent to overwrite them.  s_nop 4: an SGPR operand a VALU wrote (readfirstlane)
// is five wait states from a vector-memory instruction reading it, and nothing pads an inline statement.
// The statement is gfx950's: global_load_lds_dwordx4 exists there and not before, and the s_nop padding is that target's hazard table.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "pde_dma_planes(): LDS-DMA of 16 bytes per lane and its hazard padding are written for gfx950 only"
#endif
template <int NPL>
__device__ __forceinline__ void pde_dma_planes(const float4 *const p0, const float4 *const p1, const float4 *const p2, const unsigned voff,
                                               const unsigned d0, const unsigned d1, const unsigned d2)
{
  unsigned keep;
  if constexpr(NPL == 3)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 4\n\ts_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                 "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(p0), "s"(p1), "s"(p2), "s"(d0), "s"(d1), "s"(d2)
                 : "memory");
  else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 4\n\ts_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(p0), "s"(p2), "s"(d0), "s"(d2)
                 : "memory");
}
template <bool HSUB, int MODE, bool DMA = false>
__global__ __launch_bounds__(256, 3) void diffuse_pde_strip(const float4 *__restrict__ hf, const float4 *__restrict__ hsub,
                                                         const float4 *__restrict__ lf,
                                                         float4 *__restrict__ out, const pde_args a, const int final_pass,
                                                         const unsigned char *__restrict__ mask, const int strip,
                                                         const int strips_per_class)
{
  extern __shared__ float4 r2s[]; // [PDE_RING][256 + 2 * mult]; slot x of a row = column clamp(seg - mult + x)
  const int mult = a.mult;
  const int cls = blockIdx.y / strips_per_class, k0s = (blockIdx.y - cls * strips_per_class) * strip;
  const int n_cls = (a.height - cls + mult - 1) / mult; // rows of this class
  if(k0s >= n_cls) return;
  const int nrows = (strip < n_cls - k0s) ? strip : n_cls - k0s;
  const int r_first = cls + k0s * mult;
  const int tx = threadIdx.x, tw = 256 + 2 * mult;
  const int col = (int)blockIdx.x * 256 + tx;
  const bool live = col < a.width;
  // lanes past the end of the row keep fetching (clamped): their samples are the clamped columns of their neighbours
  const int cols[3] = { clampi(col - mult, 0, a.width - 1), clampi(col, 0, a.width - 1), clampi(col + mult, 0, a.width - 1) };
  // support row v of the strip (output row kk reads v = kk, kk + 1, kk + 2) = frame row r_first + (v - 1) mult, clamped
#define PDE_ROW(v) ((size_t)clampi(r_first + ((v) - 1) * mult, 0, a.height - 1) * a.width)
  // the three support rows of the window live in three register sets used in rotation -- support row v in set v % 3, the
  // row loop unrolled by three -- so that moving the window down a row moves no register (it was 48 v_mov per row)
  float4 Hw[3][3], Lw[3][3];
  // DMA: this wave's landing zone behind the ring, [NPL][zw] samples; slot x of a plane's row = column clamp(col0 - mult + x) where
  // col0 is the wave's first column: lane l reads slots l, l + mult, l + 2 mult = its three support columns cols[]
  constexpr int NPL = HSUB ? 3 : 2;
  const int lane = tx & 63, zw = 64 + 2 * mult;
  float4 *const zone = r2s + PDE_RING * tw + (tx >> 6) * NPL * zw;
  // the zone's LDS byte address (what M0 takes): the pointer in the LDS address space, not the low half of a generic one
  using lds_f4_ptr = __attribute__((address_space(3))) float4 *;
  const unsigned zone_lds = DMA ? (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(__UINTPTR_TYPE__)(lds_f4_ptr)zone) : 0u;
  const unsigned off_main = (unsigned)cols[0] * 16u, off_halo = (unsigned)clampi(col - mult + 64, 0, a.width - 1) * 16u;
  // send for support row v: every lane one sample per plane, the first 2 mult lanes the zone's last 2 mult slots as well
  auto dma_row = [&](const int v) {
    const size_t y = PDE_ROW(v);
    const unsigned pl = (unsigned)zw * 16u;
    pde_dma_planes<NPL>(hf + y, hsub + y, lf + y, off_main, zone_lds, zone_lds + pl, zone_lds + (NPL - 1) * pl);
    if(lane < 2 * mult)
      pde_dma_planes<NPL>(hf + y, hsub + y, lf + y, off_halo, zone_lds + 1024u, zone_lds + pl + 1024u, zone_lds + (NPL - 1) * pl + 1024u);
  };
  // the output of the row step before, stored behind the NEXT row step's fetches and barrier (row_step())
  struct
  {
    float4 o;
    size_t idx;
    bool have;
  } pending;
  pending.have = false;
  // fetch support row v into set `slot` and leave its squared ratios in ring row v % PDE_RING
  auto fetch_row = [&](const int v, auto slot_tag) {
    constexpr int SL = decltype(slot_tag)::value;
    const size_t y = PDE_ROW(v);
    // ALL nine fetches of the row are issued before the first is used (round 5).  The compiler had scheduled the first
    // subtraction c - low between the fifth and the sixth fetch: `s_waitcnt vmcnt(0)` there -- two memory round trips per row
    // step instead of one, and, the vector-memory counter being in order, the first of them also waited for the store the row
    // step before had just issued.  That store now goes out in the middle of THIS row step (row_step(): `pending`), a whole
    // update's arithmetic ahead of the next wait for fetches.
    float4 c[3], low[3];
    if constexpr(DMA)
    {
      // everything this wave sent for has landed (and the row step before's store has left)
      asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
#pragma unroll
      for(int jj = 0; jj < 3; jj++)
      {
        c[jj] = zone[lane + jj * mult];
        if(HSUB) low[jj] = zone[zw + lane + jj * mult];
        Lw[SL][jj] = zone[(NPL - 1) * zw + lane + jj * mult];
      }
    }
    else if(PDE_OFF(a, 2) && v >= 2)
    {
#pragma unroll
      for(int jj = 0; jj < 3; jj++)
      {
        c[jj] = Hw[SL][jj];
        low[jj] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    else
    {
#pragma unroll
      for(int jj = 0; jj < 3; jj++)
      {
        c[jj] = hf[y + cols[jj]];
        if(HSUB) low[jj] = hsub[y + cols[jj]];
        Lw[SL][jj] = lf[y + cols[jj]];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for(int jj = 0; jj < 3; jj++)
      Hw[SL][jj] = HSUB ? make_float4(c[jj].x - low[jj].x, c[jj].y - low[jj].y, c[jj].z - low[jj].z, c[jj].w - low[jj].w) : c[jj];
    float4 *const ring = r2s + (v % PDE_RING) * tw;
    // the fourth channel's squared ratio is +0 when both samples are +0 (0 / 1e-8, squared): a wave whose samples all
    // are -- what a pipe hands this module, see alpha_is_blank() -- skips that division (a uniform branch)
    auto ratios = [&](const float4 h, const float4 l) {
      const bool blank = a.wskip && __builtin_amdgcn_ballot_w64((__float_as_uint(h.w) | __float_as_uint(l.w)) != 0) == 0ull;
      return ratio2_rgb(h, l, blank ? 0.0f : ratio2(h.w, l.w), PDE_APPROX(a));
    };
    if(PDE_OFF(a, 4)) return;
    ring[tx + mult] = ratios(Hw[SL][1], Lw[SL][1]);
    if(tx < mult) ring[tx] = ratios(Hw[SL][0], Lw[SL][0]);
    if(tx >= 256 - mult) ring[tx + 2 * mult] = ratios(Hw[SL][2], Lw[SL][2]);
  };
  // output row kk of the strip: its support rows kk, kk + 1 (fetched) and kk + 2 (fetched here), in sets T, T + 1, T + 2 mod 3
  auto row_step = [&](auto t_tag, const int kk) {
    constexpr int T = decltype(t_tag)::value, S0 = T, S1 = (T + 1) % 3, S2 = (T + 2) % 3;
    // (a mode spelled at compile time whose orders read at most ONE of the two gradient directions)
    constexpr bool WIDE_READS = MODE >= 0 && !(((MODE % 3) | ((MODE / 9) % 3)) && (((MODE / 3) % 3) | ((MODE / 27) % 3)));
    fetch_row(kk + 2, std::integral_constant<int, S2>());
    if constexpr(DMA)
    {
      // the row step before's pixel leaves, then the fetch of the NEXT step's row: both in flight through all of this step's arithmetic
      if(pending.have)
      {
        if(final_pass) nt_store(out + pending.idx, pending.o);
        else out[pending.idx] = pending.o;
        pending.have = false;
      }
      if(kk + 1 < nrows) dma_row(kk + 3);
    }
    if(!PDE_OFF(a, 1)) __syncthreads();
    if(!live) return;
    const float4 H4[9] = { Hw[S0][0], Hw[S0][1], Hw[S0][2], Hw[S1][0], Hw[S1][1], Hw[S1][2], Hw[S2][0], Hw[S2][1], Hw[S2][2] };
    const float4 L4[9] = { Lw[S0][0], Lw[S0][1], Lw[S0][2], Lw[S1][0], Lw[S1][1], Lw[S1][2], Lw[S2][0], Lw[S2][1], Lw[S2][2] };
    const int row = r_first + kk * mult;
    const bool blank = alpha_is_blank(H4, L4);
    // the output of the row step before goes out HERE: every register of the fetched row has been read by now (the blank test
    // reads the last of them), so no wait for a fetch comes behind this store -- the compiler merges the counters of the two
    // paths of a branch conservatively, and a store in front of the first use of a fetched register made that use wait for
    // the store as well (`s_waitcnt vmcnt(0)`)
    __builtin_amdgcn_sched_barrier(0);
    if(pending.have && !PDE_OFF(a, 8))
    {
      if(final_pass) nt_store(out + pending.idx, pending.o);
      else out[pending.idx] = pending.o;
      pending.have = false;
    }
    __builtin_amdgcn_sched_barrier(0);
    float4 energy = make_float4(0.f, 0.f, 0.f, 0.f);
    if(PDE_OFF(a, 4)) energy = make_float4(0.5f, 0.5f, 0.5f, 0.5f);
    else
#pragma unroll
    for(int ii = 0; ii < 3; ii++)
    {
      const float4 *const ring = r2s + ((kk + ii) % PDE_RING) * tw + tx;
#pragma unroll
      for(int jj = 0; jj < 3; jj++)
      {
        const float4 r = ring[jj * mult];
        energy.x += r.x;
        energy.y += r.y;
        energy.z += r.z;
        if(!(blank && a.wskip)) energy.w += r.w; // (nobody reads it otherwise)
      }
      // three reads in flight, not nine (they sit on top of the 72 registers of the support) -- six + three where the mode
      // leaves the registers (one direction or none: 122 of 128; two directions would need 134): a round trip through the
      // LDS fewer per row step
      if(!WIDE_READS || ii >= 1) asm volatile("" : "+v"(energy.x), "+v"(energy.y), "+v"(energy.z), "+v"(energy.w) : : "memory");
    }
    const size_t idx = (size_t)row * a.width + col;
    float4 o;
    if(mask && !mask[idx])
    {
      // outside the luminance mask: "only copy input to output", diffuse.c:927-937
      const float4 h = H4[4], l = L4[4];
      o = make_float4(max_zero(h.x + l.x), max_zero(h.y + l.y), max_zero(h.z + l.z), max_zero(h.w + l.w));
    }
    else if(PDE_OFF(a, 16))
      o = make_float4(H4[4].x + energy.x, L4[4].y + energy.y, H4[0].z + energy.z, L4[8].w);
    else
    {
      float H[9], L[9];
#pragma unroll
      for(int k = 0; k < 9; k++) { H[k] = H4[k].x; L[k] = L4[k].x; }
      o.x = pde_channel<MODE>(H, L, energy.x, a);
#pragma unroll
      for(int k = 0; k < 9; k++) { H[k] = H4[k].y; L[k] = L4[k].y; }
      o.y = pde_channel<MODE>(H, L, energy.y, a);
#pragma unroll
      for(int k = 0; k < 9; k++) { H[k] = H4[k].z; L[k] = L4[k].z; }
      o.z = pde_channel<MODE>(H, L, energy.z, a);
      if(blank)
        o.w = 0.0f;
      else
      {
#pragma unroll
        for(int k = 0; k < 9; k++) { H[k] = H4[k].w; L[k] = L4[k].w; }
        o.w = pde_channel<MODE>(H, L, energy.w, a);
      }
    }
    if(a.post_lab) o = px_rgb_to_lab(o, a.post_m);
    pending.o = o;
    pending.idx = idx;
    pending.have = true;
  };
  if constexpr(DMA) dma_row(0);
  fetch_row(0, std::integral_constant<int, 0>());
  if constexpr(DMA) dma_row(1);
  fetch_row(1, std::integral_constant<int, 1>());
  if constexpr(DMA) dma_row(2);
  for(int kk = 0; kk < nrows; kk += 3)
  {
    row_step(std::integral_constant<int, 0>(), kk);
    if(kk + 1 < nrows) row_step(std::integral_constant<int, 1>(), kk + 1);
    if(kk + 2 < nrows) row_step(std::integral_constant<int, 2>(), kk + 2);
  }
  if constexpr(DMA) asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); // nothing of this wave's is on its way to an LDS it has left
  if(pending.have)
  {
    if(final_pass) nt_store(out + pending.idx, pending.o);
    else out[pending.idx] = pending.o;
  }
#undef PDE_ROW
}

// ---- build_mask() + inpaint_mask(), diffuse.c:1106-1152, with the generators of src/iop/noise_generator.h:36-93 ----
__device__ __forceinline__ uint32_t splitmix32(const uint64_t seed)
{
  uint64_t result = (seed ^ (seed >> 33)) * 0x62a9d9ed799705f5ull;
  result = (result ^ (result >> 28)) * 0xcb24d0a5c88c35b3ull;
  return (uint32_t)(result >> 32);
}

__device__ __forceinline__ float xoshiro128plus(uint32_t state[4])
{
  const uint32_t result = state[0] + state[3];
  const uint32_t t = state[1] << 9;
  state[2] ^= state[0];
  state[3] ^= state[1];
  state[1] ^= state[2];
  state[0] ^= state[3];
  state[2] ^= t;
  state[3] = (state[3] << 11) | (state[3] >> 21);
  return (float)(result >> 8) * 0x1.0p-24f;
}

// gaussian_noise(), noise_generator.h:81-93: Box-Muller with the C library's logf / cosf / sinf (devmath.h restates
// them) and the angle formed in binary64 (2.f * M_PI * u2) before it is narrowed for the call
__device__ __forceinline__ float gaussian_noise(const float mu, const float sigma, const bool flip, uint32_t state[4])
{
  const float x1 = xoshiro128plus(state);
  const float u1 = x1 > 1.17549435e-38f ? x1 : 1.17549435e-38f; // fmaxf(x, FLT_MIN); x is never NaN
  const float u2 = xoshiro128plus(state);
  const float angle = (float)(6.283185307179586 * (double)u2);
  const float radius = sqrtf(-2.0f * ansel_math::logf_exact(u1));
  const float noise = flip ? radius * ansel_math::cosf_exact(angle) : radius * ansel_math::sinf_exact(angle);
  return noise * sigma + mu;
}

// k = the FLOAT index of the pixel in the frame; the reference takes k / width for its "row" and k - row for its
// "column" (sic) and seeds the generator with them.  first_pixel: the frame index of in[0] (a row band of a frame)
__global__ __launch_bounds__(256) void diffuse_inpaint(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                       unsigned char *__restrict__ mask, const float threshold,
                                                       const size_t npixels, const size_t width, const size_t first_pixel)
{
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(p >= npixels) return;
  const float4 v = in[p];
  const bool m = v.x > threshold || v.y > threshold || v.z > threshold;
  mask[p] = m ? 1 : 0;
  float4 o = v;
  if(m)
  {
    const size_t k = 4 * (p + first_pixel);
    const uint32_t i = (uint32_t)(k / width);
    const uint32_t j = (uint32_t)(k - i);
    uint32_t state[4] = { splitmix32((uint32_t)(j + 1)), splitmix32((uint64_t)(uint32_t)(j + 1) * (uint32_t)(i + 3)),
                          splitmix32(1337), splitmix32(666) };
    xoshiro128plus(state);
    xoshiro128plus(state);
    xoshiro128plus(state);
    xoshiro128plus(state);
    const bool flip = (i % 2) || (j % 2);
    o.x = fabsf(gaussian_noise(v.x, v.x, flip, state));
    o.y = fabsf(gaussian_noise(v.y, v.y, flip, state));
    o.z = fabsf(gaussian_noise(v.z, v.z, flip, state));
    o.w = fabsf(gaussian_noise(v.w, v.w, flip, state));
  }
  out[p] = o;
}

inline float sqf(const float x) { return x * x; }

// equivalent_sigma_at_step(), bspline.h:55-66
float sigma_at_step(const float sigma, const unsigned s)
{
  float acc = sigma;
  for(unsigned k = 1; k <= s; k++) acc = sqrtf(sqf(acc) + sqf(exp2f((float)k) * sigma));
  return acc;
}

// num_steps_to_reach_equivalent_sigma(), bspline.h:68-80
unsigned steps_to_sigma(const float sigma_filter, const float sigma_final)
{
  unsigned s = 0;
  float radius = sigma_filter;
  while(radius < sigma_final)
  {
    ++s;
    radius = sqrtf(sqf(radius) + sqf((float)(1 << s) * sigma_filter));
  }
  return s + 1;
}

int scales_of(const dt_hip_piece_t *piece, const dt_hip_diffuse_data_t *d)
{
  const float zoom = (float)(d->iscale / piece->roi_in.scale);
  const float final_radius = (float)(d->radius + d->radius_center) * 2.0f / zoom;
  const int s = (int)steps_to_sigma(BSPLINE_SIGMA, final_radius);
  return s < 1 ? 1 : (s > DIFFUSE_MAX_SCALES ? DIFFUSE_MAX_SCALES : s);
}

} // namespace

namespace ansel
{
// Rows of input the own rows of a band depend on: per iteration the decompositions reach 2 * (1 + 2 + ... +
// 2^(scales-1)) rows and the PDE passes, coarse to fine, one dilation each
int diffuse_halo_rows(const dt_hip_piece_t *piece, const dt_hip_diffuse_data_t *d)
{
  if(!(d->iscale > 0.0f) || !(piece->roi_in.scale > 0.0)) return -1;
  const int scales = scales_of(piece, d);
  const int it_req = (int)ceilf((float)d->iterations);
  const int iterations = it_req > 1 ? it_req : 1;
  return iterations * 3 * ((1 << scales) - 1);
}
} // namespace ansel

extern "C" {

int dt_hip_iop_diffuse_process(int devid, const dt_hip_piece_t *piece, const dt_hip_diffuse_data_t *d,
                               dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  return diffuse_process_rows(devid, piece, d, 0, dev_in, dev_out);
}

} // extern "C"

namespace ansel
{

// first_row: the frame row of the buffer's first row (a row band, pipe.cpp) -- it only enters the seeds of the
// inpainting noise, which the reference derives from the pixel's index in the frame
static int diffuse_run(int devid, const dt_hip_piece_t *piece, const dt_hip_diffuse_data_t *d, int first_row, dt_hip_mem_t dev_in,
                       dt_hip_mem_t dev_out, const dt_hip_lab_data_t *post_lab);
int diffuse_process_rows(int devid, const dt_hip_piece_t *piece, const dt_hip_diffuse_data_t *d, int first_row,
                         dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  return diffuse_run(devid, piece, d, first_row, dev_in, dev_out, nullptr);
}
// the module followed by the pipe's RGB -> Lab glue: the conversion is the tail of the module's last kernel
int diffuse_process_post_lab(int devid, const dt_hip_piece_t *piece, const dt_hip_diffuse_data_t *d, dt_hip_mem_t dev_in,
                             dt_hip_mem_t dev_out, const dt_hip_lab_data_t *lab)
{
  if(!lab || measuring_env("ANSEL_HIP_PDE_PER_ROW")) return DT_HIP_INVALID_ARG; // only the strip kernel has the tail
  return diffuse_run(devid, piece, d, 0, dev_in, dev_out, lab);
}
static int diffuse_run(int devid, const dt_hip_piece_t *piece, const dt_hip_diffuse_data_t *d, int first_row, dt_hip_mem_t dev_in,
                       dt_hip_mem_t dev_out, const dt_hip_lab_data_t *post_lab)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  if(piece->channels != 4)
  {
    set_last_error("diffuse: needs a 4-channel float input");
    return DT_HIP_INVALID_ARG;
  }
  const int w = piece->roi_out.width, h = piece->roi_out.height;
  if(w <= 0 || h <= 0) return DT_HIP_SUCCESS;
  if(!(d->iscale > 0.0f) || !(piece->roi_in.scale > 0.0)) return DT_HIP_INVALID_ARG;
  const size_t plane = (size_t)w * h * sizeof(float4);
  const float zoom = (float)(d->iscale / piece->roi_in.scale);
  const int scales = scales_of(piece, d);
  const int it_req = (int)ceilf((float)d->iterations);
  const int iterations = it_req > 1 ? it_req : 1;

  // planes: HF[scales], two low-pass ping-pong, two iteration ping-pong (diffuse.c:1167-1195)
  // LF chain (every scale on the strip kernel): hf[s] holds the LOW-pass plane of scale s + 1 instead, nothing stores a
  // high-frequency plane, and one plane besides them serves the synthesis' ping-pong (a plane fewer than the reference)
  static const bool per_row_pde = measuring_env("ANSEL_HIP_PDE_PER_ROW") != nullptr; // the per-row kernel, for A/B timing
  static const bool hf_planes = measuring_env("ANSEL_HIP_DIFFUSE_HF_PLANES") != nullptr; // the stored-HF path, for A/B timing
  static const bool no_dma = measuring_env("ANSEL_HIP_PDE_NO_DMA") != nullptr; // the rows through registers (round 4's fetch), for A/B timing
  const bool lf_chain = !per_row_pde && !hf_planes && (1 << (scales - 1)) <= PDE_SHARED_MULT;
  float4 *hf[DIFFUSE_MAX_SCALES] = { nullptr };
  float4 *lf[2] = { nullptr, nullptr }, *tmp[2] = { nullptr, nullptr };
  bool ok = true;
  for(int s = 0; s < scales; s++) ok &= (hf[s] = (float4 *)dt_hip_alloc_device_buffer(devid, plane)) != nullptr;
  for(int k = 0; k < (lf_chain ? 1 : 2); k++) ok &= (lf[k] = (float4 *)dt_hip_alloc_device_buffer(devid, plane)) != nullptr;
  if(iterations > 1)
    for(int k = 0; k < 2 && k < iterations - 1; k++)
      ok &= (tmp[k] = (float4 *)dt_hip_alloc_device_buffer(devid, plane)) != nullptr;
  int err = ok ? DT_HIP_SUCCESS : DT_HIP_SYSMEM_ALLOCATION;

  pde_args a;
  memset(&a, 0, sizeof(a));
  a.width = w;
  a.height = h;
  a.wskip = measuring_env("ANSEL_HIP_PDE_NO_WSKIP") ? 0 : 1;
  {
    const char *const approx_env = measuring_env("ANSEL_HIP_PDE_APPROX_DIV"); // read per call: tools/pde_div_ab.py flips it
    a.approx_div = approx_env && atoi(approx_env) != 0;
    const char *const off_env = measuring_env("ANSEL_HIP_PDE_OFF"); // read per call: tools/pde_off_ab.py
    a.off = off_env ? atoi(off_env) : 0;
  }
  const float user_aniso[4] = { d->anisotropy_first, d->anisotropy_second, d->anisotropy_third, d->anisotropy_fourth };
  for(int k = 0; k < 4; k++)
  {
    a.anisotropy[k] = sqf(user_aniso[k]); // compute_anisotropy_factor(), diffuse.c:970-976
    a.kind[k] = user_aniso[k] == 0.0f ? 0 : (user_aniso[k] > 0.0f ? 1 : 2);
  }
  a.same02 = a.kind[0] == a.kind[2] && a.anisotropy[0] == a.anisotropy[2];
  a.same13 = a.kind[1] == a.kind[3] && a.anisotropy[1] == a.anisotropy[3];
  const float regularization = powf(10.0f, d->regularization) - 1.0f; // diffuse.c:999-1000
  a.variance_threshold = powf(10.0f, d->variance_threshold);
  const float speed[4] = { d->first, d->second, d->third, d->fourth };

  hipStream_t st = stream_of(devid);
  const float4 *src = (const float4 *)dev_in;
  unsigned char *mask = nullptr;
  float4 *inpainted = nullptr;
  if(err == DT_HIP_SUCCESS && d->threshold > 0.0f)
  {
    // diffuse.c:1207-1219: mask of the pixels above the threshold, noise-seeded copy as the first iteration's input
    // (the reference reuses "temp1" for it; the second iteration's output may overwrite it, so may ours)
    const size_t np = (size_t)w * h;
    mask = (unsigned char *)dt_hip_alloc_device_buffer(devid, np);
    inpainted = tmp[1] ? tmp[1] : (tmp[1] = (float4 *)dt_hip_alloc_device_buffer(devid, plane));
    if(!mask || !inpainted) err = DT_HIP_SYSMEM_ALLOCATION;
    else
    {
      launch_scope ls(devid, "diffuse_inpaint");
      diffuse_inpaint<<<pixel_grid(np), 256, 0, st>>>(src, inpainted, mask, d->threshold, np, (size_t)w, (size_t)first_row * w);
      err = check_launch("diffuse_inpaint");
      src = inpainted;
    }
  }
  for(int it = 0; err == DT_HIP_SUCCESS && it < iterations; it++)
  {
    // iteration ping-pong, diffuse.c:1223-1249 (tmp[0] = "temp2", tmp[1] = "temp1")
    float4 *dst = (it == iterations - 1) ? (float4 *)dev_out : tmp[it % 2];
    const float4 *level = src;
    float4 *residual = lf[0];
    // the chain reads the iteration's input until its last pass: not when that pass writes the same plane
    const bool chain = lf_chain && (const float4 *)dst != src;
    if(!chain && !lf[1])
    {
      lf[1] = (float4 *)dt_hip_alloc_device_buffer(devid, plane);
      if(!lf[1])
      {
        err = DT_HIP_SYSMEM_ALLOCATION;
        break;
      }
    }
    for(int s = 0; s < scales && err == DT_HIP_SUCCESS; s++)
    {
#ifdef ANSEL_HIP_MEASURING
      // the chain keeps every low-pass plane: two scales in one pass where the pair kernel has the dilation (1 + 2, 4 + 8)
      static const bool pairs = measuring_env("ANSEL_HIP_BSPLINE_PAIRS") != nullptr;
      if(chain && pairs && s + 1 < scales && (s == 0 || s == 2))
      {
        err = bspline_launch_decompose2(devid, st, level, hf[s], hf[s + 1], w, h, 1 << s);
        s++;
        level = residual = hf[s];
        continue;
      }
#endif
      float4 *low = chain ? hf[s] : lf[s % 2];
      err = bspline_launch_decompose(devid, st, level, chain ? nullptr : hf[s], low, w, h, 1 << s);
      level = low;
      residual = low;
    }
    float4 *pp[2] = { residual == lf[1] ? lf[0] : lf[1], residual };
    const float4 *cur = residual;
    int count = 0;
    for(int s = scales - 1; s >= 0 && err == DT_HIP_SUCCESS; s--, count++)
    {
      // per-band constants, diffuse.c:1053-1074
      const float real_radius = sigma_at_step(BSPLINE_SIGMA, (unsigned)s) * zoom;
      a.regularization = regularization / 9.0f * sqf(real_radius);
      const float norm = expf(-sqf(real_radius - (float)d->radius_center) / sqf((float)d->radius));
      for(int k = 0; k < 4; k++) a.abcd[k] = speed[k] * PDE_KAPPA * norm;
      a.strength = d->sharpness * norm + 1.0f;
      a.mult = 1 << s;
      a.post_lab = 0;
      if(post_lab && s == 0 && it == iterations - 1)
      {
        a.post_lab = 1;
        memcpy(a.post_m, post_lab->matrix, sizeof(a.post_m));
      }
      // chain: scale s reads the low-pass planes of scales s (hf[s - 1], or the iteration's input) and s + 1 (hf[s]) and
      // the running sum; free to write are lf[0] and, once scale s + 1 is done, hf[s + 1]: they alternate
      float4 *to = (s == 0) ? dst : (chain ? (count % 2 == 0 ? lf[0] : hf[s + 1]) : pp[count % 2]);
      const int rows = (h <= a.mult) ? h : ((h + a.mult - 1) / a.mult) * a.mult;
      {
        launch_scope ls(devid, "diffuse_pde");
        // gridDim.x padded to a multiple of 8: a column block stays on one XCD, the rows above and below hit its L2
        const dim3 grid(xcd_pad((w + 255) / 256), rows);
        const bool per_row = per_row_pde;
        if(a.mult <= PDE_SHARED_MULT && !per_row)
        {
          const int classes = h < a.mult ? h : a.mult, per_class = (h + a.mult - 1) / a.mult;
          const int gx = (w + 255) / 256;
          int strip = 32;
#ifdef ANSEL_HIP_MEASURING
          static const char *const strip_env = measuring_env("ANSEL_HIP_PDE_STRIP"); // rows a workgroup keeps its columns for, for A/B timing
          if(strip_env) strip = atoi(strip_env);
#endif
          while(strip > 4 && (size_t)gx * classes * ((per_class + strip - 1) / strip) < 2048) strip /= 2;
          const int spc = (per_class + strip - 1) / strip;
          size_t ring = (size_t)PDE_RING * (256 + 2 * a.mult) * sizeof(float4);
#ifdef ANSEL_HIP_MEASURING
          // how much the kernel needs its occupancy: LDS asked for and not used (bytes), for A/B timing
          static const char *const pad_env = measuring_env("ANSEL_HIP_PDE_LDS_PAD");
          if(pad_env) ring += (size_t)atoi(pad_env);
#endif
          static const bool generic = measuring_env("ANSEL_HIP_PDE_GENERIC") != nullptr; // the kinds read at run time, for A/B timing
          const int mode = generic ? -1 : pde_mode_of(a);
          const dim3 sgrid(gx, classes * spc);
          const float4 *const h0 = chain ? (s == 0 ? src : hf[s - 1]) : hf[s], *const h1 = chain ? hf[s] : nullptr;
#define PDE_LAUNCH(HS, MD) diffuse_pde_strip<HS, MD><<<sgrid, 256, ring, st>>>(h0, h1, cur, to, a, s == 0, mask, strip, spc)
          // the rows by LDS-DMA into per-wave landing zones behind the ring (the kernel's comment): the low-pass chain (three planes a row) at the dilations whose halo is at most a quarter of a wave
          // (with a luminance mask the byte the row step loads makes the compiler wait for every vector-memory operation in flight,
          // the row fetch included, in the middle of a step: masked runs keep the form -- same result -- but not its overlap)
          const bool dma = chain && a.mult <= PDE_DMA_MAX_MULT && !no_dma;
          const size_t ring_dma = ring + (size_t)4 * 3 * (64 + 2 * a.mult) * sizeof(float4);
#define PDE_LAUNCH_DMA(MD) diffuse_pde_strip<true, MD, true><<<sgrid, 256, ring_dma, st>>>(h0, h1, cur, to, a, s == 0, mask, strip, spc)
          bool launched = false;
#define PDE_CASE(MD)                     \
  if(!launched && mode == (MD))          \
  {                                      \
    if(dma) PDE_LAUNCH_DMA(MD);          \
    else if(chain) PDE_LAUNCH(true, MD); \
    else PDE_LAUNCH(false, MD);          \
    launched = true;                     \
  }
          PDE_MODES(PDE_CASE)
#undef PDE_CASE
          if(!launched)
          {
            if(dma) PDE_LAUNCH_DMA(-1);
            else if(chain) PDE_LAUNCH(true, -1);
            else PDE_LAUNCH(false, -1);
          }
#undef PDE_LAUNCH_DMA
#undef PDE_LAUNCH
        }
        else if(a.mult <= PDE_SHARED_MULT)
          diffuse_pde<true><<<grid, 256, (size_t)3 * (256 + 2 * a.mult) * sizeof(float4), st>>>(hf[s], cur, to, a, s == 0, mask);
        else
          diffuse_pde<false><<<grid, 256, 0, st>>>(hf[s], cur, to, a, s == 0, mask);
      }
      err = check_launch("diffuse_pde");
      cur = to;
    }
    src = dst;
  }
  for(int s = 0; s < scales; s++)
    if(hf[s]) dt_hip_release_mem_object(hf[s]);
  for(int k = 0; k < 2; k++)
  {
    if(lf[k]) dt_hip_release_mem_object(lf[k]);
    if(tmp[k]) dt_hip_release_mem_object(tmp[k]);
  }
  if(mask) dt_hip_release_mem_object(mask);
  return err;
}

} // namespace ansel

extern "C" {

// tiling_callback(), diffuse.c:585-610
void dt_hip_iop_diffuse_tiling(const dt_hip_piece_t *piece, const dt_hip_diffuse_data_t *d, dt_hip_tiling_t *tiling)
{
  const int scales = scales_of(piece, d);
  tiling->factor = 6.0625f + scales;
  tiling->factor_cl = 6.0625f + scales;
  tiling->maxbuf = 1.0f;
  tiling->maxbuf_cl = 1.0f;
  tiling->overhead = 0;
  tiling->overlap = 1 << scales;
  tiling->xalign = 1;
  tiling->yalign = 1;
}

} // extern "C"
