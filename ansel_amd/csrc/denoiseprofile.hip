// denoiseprofile.hip -- denoise (profiled), wavelets mode, on gfx950.
//
// Reference: process_wavelets(), src/iop/denoiseprofile.c:1289-1447 (process_wavelets_cl() :2150-2480
// is the same chain as OpenCL kernels): variance-stabilising transform -> max_scale bands of the
// edge-aware a-trous step eaw_dn_decompose() (src/pixel/eaw.c:242-327) -> BayesShrink threshold from
// the band's sum of squared details (variance_stabilizing_xform(), :1223-1287) -> soft-threshold
// accumulate eaw_synthesize() (eaw.c:157-175) -> + residue -> inverse transform.
//
// Launches per frame: 1 precondition, per band {decompose, threshold}, 1 finish.  Every band keeps its
// detail plane (7 x 1.6 GB at 100 MP, of 288 GB; on the unsplit frame since round 6: band 0's detail and every band's COARSE
// plane, the details of bands 1 .. being differences of those, formed where they are read -- three floats a pixel under the Y0U0V0
// transform, whose fourth channel is +0: denoiseprofile_run()), so the soft-threshold accumulation of ALL bands
// (eaw_synthesize(), in band order, from a zero accumulator: the same additions in the same order), the
// residue add and the inverse transform are one pass over the frame -- 144 B/px instead of the 384 B/px
// of one read-modify-write pass per band.  Nothing returns to the host between launches: the band
// thresholds are computed on the device from the reduced sums and consumed through a 112-byte buffer.
// The round-robin accumulation of the segment sums runs on 16 workgroups (one CU reads 12.8 MB at ~55 GB/s).
//
// The sum of squared details is an OpenMP float reduction in the reference, so its value depends on
// the host's thread count.  Here (and in oracle/src/denoiseprofile.c) it is the binary64 sum of the
// binary32 products det*det in a fixed order:
//   1. per 256-pixel row segment (= one workgroup): halving reduction inside each wave of 64
//      (v[m] += v[m + off], off = 32..1), then the four wave sums left to right;
//   2. segment sums, numbered row-major, dealt round-robin to 1024 accumulators (increasing segment
//      number), which are then reduced by halving (off = 512..1); rounded once to binary32.
// powf (the variance-stabilising transforms, three per sample each way; the fused run's colour stages) looks its tables up in the
// workgroup's LDS copy: every kernel of this file that calls it stages them first (devmath.h stage_default_tables())
#define ANSEL_MATH_DEFAULT_TABS tabs_lds
#include "hip_common.h"
#include "pipe_fused.h"
#include "rgb_chain_kernel.h"
#include "devmath.h"
#include "nlmeans_core_params.h"

#include <math.h>

using namespace ansel;

namespace
{

#define BANDS DT_HIP_DENOISEPROFILE_BANDS
#define P_FULCRUM 0.05f // DT_IOP_DENOISE_PROFILE_P_FULCRUM, denoiseprofile.c:118

__host__ __device__ __forceinline__ float max_first(const float a, const float b) { return a > b ? a : b; } // MAX(a, b)
__host__ __device__ __forceinline__ float min_first(const float a, const float b) { return a < b ? a : b; } // MIN(a, b)
__device__ __forceinline__ int clampi(const int v, const int lo, const int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct dn_setup
{
  int max_scale;
  int vst; // 0 legacy Anscombe, 1 v2 RGB, 2 v2 Y0U0V0
  float wb[4], p[4], aa[4], bb[4];
  float toY[3][4], toRGB[3][4];
  float a_v2, b_v2, bias;
};

// per-channel constants of the forward / inverse transform, prepared on the host
struct vst_args
{
  int vst;
  float k[4];      // legacy: sigma^2 + 3/8 (forward) or + 1/8 (inverse)
  float aa[4];     // legacy: a
  float expon[4];  // v2
  float scale[4];  // v2: "denom" (RGB) or "scale" (Y0U0V0)
  float wb[4];     // v2 RGB
  float bias_wb[4];
  float b, bias, sqrt_3_2;
  float m[3][4];   // Y0U0V0: toY0U0V0 (forward) / toRGB (inverse)
};

__device__ __forceinline__ float chan(const float4 &v, const int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

// precondition(), precondition_v2(), precondition_Y0U0V0(): denoiseprofile.c:852-870, :916-933, :1021-1051
// t3_of_zero (Y0U0V0 only; pass have_t3 = false otherwise): the transformed fourth channel of a pixel whose alpha is +0 -- the same
// expression on the same operands, formed once per thread: a wave whose pixels all carry +0 there (what a pipe hands this module)
// takes it instead of a fourth powf per pixel (a uniform branch)
__device__ __forceinline__ float dn_vst_y0u0v0_alpha(const float alpha, const vst_args &a)
{
  return ansel_math::powf_exact(max_first(alpha + a.b, 0.0f), a.expon[3]) * a.scale[3];
}
__device__ __forceinline__ float4 dn_precondition_pixel(const float4 px, const vst_args &a, const float t3_of_zero = 0.0f,
                                                        const bool have_t3 = false)
{
  float o[4];
  if(a.vst == 0)
  {
#pragma unroll
    for(int c = 0; c < 4; c++)
    {
      const float d = fmaxf(0.0f, chan(px, c) / a.aa[c] + a.k[c]);
      o[c] = 2.0f * sqrtf(d);
    }
  }
  else if(a.vst == 1)
  {
#pragma unroll
    for(int c = 0; c < 4; c++)
      o[c] = 2.0f * ansel_math::powf_exact(max_first(chan(px, c) / a.wb[c] + a.b, 0.0f), a.expon[c]) / a.scale[c];
  }
  else
  {
    float t[4];
#pragma unroll
    for(int c = 0; c < 3; c++) t[c] = ansel_math::powf_exact(max_first(chan(px, c) + a.b, 0.0f), a.expon[c]) * a.scale[c];
    if(have_t3 && __builtin_amdgcn_ballot_w64(__float_as_uint(px.w) != 0u) == 0ull) t[3] = t3_of_zero;
    else t[3] = dn_vst_y0u0v0_alpha(px.w, a);
#pragma unroll
    for(int c = 0; c < 3; c++)
    {
      float sum = 0.0f;
#pragma unroll
      for(int k = 0; k < 4; k++) sum += a.m[c][k] * t[k];
      o[c] = sum;
    }
    o[3] = 0.0f;
  }
  return make_float4(o[0], o[1], o[2], o[3]);
}

__global__ __launch_bounds__(256) void dn_precondition(const float4 *__restrict__ in, float4 *__restrict__ buf,
                                                       const size_t npix, const vst_args a)
{
  ansel_math::stage_default_tables(threadIdx.x);
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; // one pixel per thread (pixel_grid)
  if(j < npix) buf[j] = dn_precondition_pixel(in[j], a);
}

// out[k] += residue[k] (denoiseprofile.c:1423-1425) followed by backtransform(), backtransform_v2(),
// backtransform_Y0U0V0(): :872-897, :996-1019, :1053-1090
struct synth_args
{
  int nbands;                 // 0: `out` already holds the accumulator (the non-local-means path)
  const float4 *detail[BANDS]; // the bands' detail planes, finest first
  const float *thrs;          // [nbands][4]
  // round 6 (the unsplit frame): non-null = only band 0's detail is a plane; detail[b], b >= 1, is band b's COARSE plane, coarse0 is band
  // 0's, and the detail of band b is formed here, coarse b - 1 minus coarse b -- the subtraction eaw_dn_decompose() stores
  // (eaw.c:232-238: det = px - sum; px IS the coarse value of the band before), same operands, same operation -- so that the
  // decompositions of bands 1 .. write one plane instead of two (16 B/px less each) and two planes fewer are held.  The residue is the
  // last band's coarse plane: read once.  (Band 0's input is the transformed frame, which is never stored: its detail stays a plane.)
  const float4 *coarse0;
  // the planes hold three floats a pixel, their fourth channel being +0 (the launches that wrote them were ALPHA0 ones: dn_decompose_strip P3)
  int planes3;
  // non-null: the launch runs only if the word is 0 (gate_sense 1) / is not 0 (gate_sense 2) -- the decompositions' alpha flag: the
  // synthesis over the three-float planes, or the one over the four-channel sequence's planes (denoiseprofile_run())
  const unsigned *gate;
  int gate_sense;
};
__device__ __forceinline__ bool dn_gate_closed(const synth_args &sy)
{
  if(!sy.gate) return false;
  const unsigned raised = *sy.gate;
  return sy.gate_sense == 1 ? raised != 0u : raised == 0u;
}

// eaw_synthesize() with boost 1, eaw.c:157-175, for one band on the accumulator in registers
__device__ __forceinline__ void synthesize_band(float4 &acc, const float4 d, const float *__restrict__ t)
{
  acc.x = acc.x + (1.0f * (max_first(d.x - t[0], 0.0f) + min_first(d.x + t[0], 0.0f)));
  acc.y = acc.y + (1.0f * (max_first(d.y - t[1], 0.0f) + min_first(d.y + t[1], 0.0f)));
  acc.z = acc.z + (1.0f * (max_first(d.z - t[2], 0.0f) + min_first(d.z + t[2], 0.0f)));
  acc.w = acc.w + (1.0f * (max_first(d.w - t[3], 0.0f) + min_first(d.w + t[3], 0.0f)));
}

// the synthesis of all bands, the residue and the inverse transform of pixel j
__device__ __forceinline__ float4 dn_finish_pixel(const float4 *__restrict__ out, const float4 *__restrict__ residue, const size_t j,
                                                  const vst_args &a, const synth_args &sy)
{
  {
    float4 acc;
    float4 res = make_float4(0.f, 0.f, 0.f, 0.f);
    if(sy.nbands > 0)
    {
      // every band's sample and the residue are fetched before the first is used: with a fetch inside each band's
      // (uniform) branch a wave had ONE 16-byte fetch per lane in flight at a time, and eight waves per SIMD of that do
      // not cover the latency of 8 TB/s (a band the module does not have fetches the first one again)
      float4 d[BANDS];
      float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f);
      if(sy.planes3) // uniform
      {
#pragma unroll
        for(int b = 0; b < BANDS; b++)
        {
          const float *const p3 = reinterpret_cast<const float *>(sy.detail[b < sy.nbands ? b : 0]) + 3 * j;
          d[b] = make_float4(p3[0], p3[1], p3[2], 0.0f);
        }
        if(sy.coarse0)
        {
          const float *const p3 = reinterpret_cast<const float *>(sy.coarse0) + 3 * j;
          c0 = make_float4(p3[0], p3[1], p3[2], 0.0f);
        }
      }
      else
      {
#pragma unroll
        for(int b = 0; b < BANDS; b++) d[b] = sy.detail[b < sy.nbands ? b : 0][j];
        if(sy.coarse0) c0 = sy.coarse0[j];
      }
      if(sy.coarse0)
      {
        res = c0; // (the residue of a single band; else the last band's coarse plane, which d[] holds)
#pragma unroll
        for(int b = 1; b < BANDS; b++)
          if(b < sy.nbands) res = d[b];
        // from the coarsest band down, so that d[b - 1] is still the coarse plane when d[b] becomes the detail
#pragma unroll
        for(int b = BANDS - 1; b >= 2; b--)
          if(b < sy.nbands) d[b] = make_float4(d[b - 1].x - d[b].x, d[b - 1].y - d[b].y, d[b - 1].z - d[b].z, d[b - 1].w - d[b].w);
        if(1 < sy.nbands) d[1] = make_float4(c0.x - d[1].x, c0.y - d[1].y, c0.z - d[1].z, c0.w - d[1].w);
      }
      else if(residue) res = residue[j];
      // the accumulator of denoiseprofile.c:1398 starts zeroed; bands are added finest first (:1400-1421)
      acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for(int b = 0; b < BANDS; b++)
        if(b < sy.nbands) synthesize_band(acc, d[b], sy.thrs + 4 * b);
    }
    else
    {
      acc = out[j];
      if(residue) res = residue[j];
    }
    float v[4] = { acc.x, acc.y, acc.z, acc.w };
    if(residue)
    {
      v[0] = acc.x + res.x;
      v[1] = acc.y + res.y;
      v[2] = acc.z + res.z;
      v[3] = acc.w + res.w;
    }
    float o[4];
    if(a.vst == 0)
    {
#pragma unroll
      for(int c = 0; c < 4; c++)
      {
        const float x = v[c], x2 = x * x;
        o[c] = (x < 0.5f) ? 0.0f
                          : a.aa[c] * (1.f / 4.f * x2 + 1.f / 4.f * a.sqrt_3_2 / x - 11.f / 8.f / x2
                                       + 5.f / 8.f * a.sqrt_3_2 / (x * x2) - a.k[c]);
      }
    }
    else if(a.vst == 1)
    {
#pragma unroll
      for(int c = 0; c < 4; c++)
      {
        const float x = max_first(v[c], 0.0f);
        const float delta = x * x + a.bias;
        const float z1 = (x + sqrtf(max_first(delta, 0.0f))) / a.scale[c];
        o[c] = a.wb[c] * (ansel_math::powf_exact(z1, a.expon[c]) - a.b);
      }
    }
    else
    {
      float rgb[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
      for(int k = 0; k < 3; k++)
#pragma unroll
        for(int c = 0; c < 4; c++) rgb[k] += a.m[k][c] * v[c];
#pragma unroll
      for(int c = 0; c < 4; c++)
      {
        const float x = max_first(rgb[c], 0.0f);
        const float delta = x * x + a.bias_wb[c];
        const float z1 = (x + sqrtf(max_first(delta, 0.0f))) * a.scale[c];
        o[c] = ansel_math::powf_exact(z1, a.expon[c]) - a.b;
      }
    }
    return make_float4(o[0], o[1], o[2], o[3]);
  }
}

__global__ __launch_bounds__(256) void dn_finish(float4 *__restrict__ out, const float4 *__restrict__ residue,
                                                 const size_t npix, const vst_args a, const synth_args sy)
{
  if(dn_gate_closed(sy)) return;
  ansel_math::stage_default_tables(threadIdx.x);
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; // one pixel per thread (pixel_grid)
  if(j < npix) nt_store(out + j, dn_finish_pixel(out, residue, j, a, sy));
}

// dn_finish followed by the pointwise run the executor has fused behind the module (exposure, colorin, color
// calibration, colorout -- no filmic here): dn_finish streams 144 B per pixel at HBM speed with 450 instructions per
// pixel, the run's 560 fit under the same memory time, and its 32 B per pixel of traffic and its launch disappear
// (2.63 + 1.19 -> 2.9 ms at 100 MP).  The stages are the device functions of rgb_chain (rgb_chain_kernel.h).
struct dn_chain_kernargs // the kernarg segment of dn_finish_chain, for the offset of its last member
{
  float4 *out;
  const float4 *residue;
  size_t npix;
  vst_args a;
  synth_args sy;
  chain_args c;
};
template <int CM>
__global__ __launch_bounds__(256) void dn_finish_chain(float4 *__restrict__ out, const float4 *__restrict__ residue,
                                                       const size_t npix, const vst_args a, const synth_args sy,
                                                       const chain_args c_by_value)
{
  const chain_args &c = kernarg_at<chain_args>((int)offsetof(dn_chain_kernargs, c));
  (void)c_by_value;
  if(dn_gate_closed(sy)) return;
  ansel_math::stage_default_tables(threadIdx.x);
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(j < npix)
  {
    float4 v = dn_finish_pixel(out, residue, j, a, sy);
    if(c.has_exposure)
    {
      v.x = (v.x - c.exp_black) * c.exp_scale;
      v.y = (v.y - c.exp_black) * c.exp_scale;
      v.z = (v.z - c.exp_black) * c.exp_scale;
      v.w = (v.w - c.exp_black) * c.exp_scale;
    }
    if(c.has_colorin) v = px_conversion_rt(v, c.colorin);
    if(CM != CM_NONE) v = px_channelmixerrgb<(CM == CM_NONE ? 0 : CM)>(v, c.cm, c.cm_clip != 0);
    if(c.has_colorout) v = px_conversion_rt(v, c.colorout);
    if(c.post_lab) v = px_rgb_to_lab(v, c.lab_post);
    out[j] = v;
  }
}

// fast_mexp2f(), src/math/math.h:306-317
__device__ __forceinline__ float mexp2(const float x)
{
  // x * -2^23 is exact (a power of two; an overflow is -inf either way), so the fused form rounds once to the same value
  // as the reference's product-then-sum: one instruction instead of two
  const float k0 = fmaf(x, 1056964608.0f - 1065353216.0f, 1065353216.0f);
  return __int_as_float(k0 >= 8388608.0f ? (int)k0 : 0);
}

// mexp2(0 > arg ? 0 : arg) -- dn_weight()'s argument, eaw.c:181-195 -- with the clamp at zero moved behind the
// conversion: a negative argument gives k0 above 0x3f800000 (its value for +-0), where an integer minimum brings it
// back; a NaN converts to 0 (v_cvt_i32_f32) and fails the test as it does in the reference; -inf saturates to
// INT_MAX and is cut to 0x3f800000 like every negative argument.  Five instructions instead of six.
__device__ __forceinline__ float mexp2_of_clamped(const float arg)
{
  const float k0 = fmaf(arg, 1056964608.0f - 1065353216.0f, 1065353216.0f);
  int k;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(k) : "v"(k0)); // truncating, saturating, NaN -> 0: the instruction's semantics, not C's
  k = min(k, 0x3f800000);
  return __int_as_float(k >= 0x800000 ? k : 0);
}

__device__ __forceinline__ int walk_row(const int b, const int height, const int mult)
{
  if(height <= mult) return b < height ? b : -1;
  const int per_pass = (height + mult - 1) / mult;
  const int row = (b % per_pass) * mult + b / per_pass;
  return row < height ? row : -1;
}

__device__ __forceinline__ double wave_sum_halving(double v)
{
#pragma unroll
  for(int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
  return v; // lane 0 holds the halving-tree sum of the wave
}

#ifdef ANSEL_HIP_MEASURING // the per-row kernel: A/B timing only (ANSEL_HIP_DN_PER_ROW); the strips below are the product's
// eaw_dn_decompose(), eaw.c:242-327: one workgroup = one 256-pixel segment of one row
__global__ __launch_bounds__(256) void dn_decompose(const float4 *__restrict__ in, float4 *__restrict__ coarse,
                                                    float4 *__restrict__ detail, double *__restrict__ partial,
                                                    const int width, const int height, const int mult,
                                                    const float inv_sigma2, const int nseg, const int in_row0,
                                                    const int in_rows)
{
  // `height` rows are computed.  Normally they are the whole input (in_row0 = 0, in_rows = height); on a row band
  // (pipe.cpp) the input additionally holds in_row0 halo rows above and in_rows - in_row0 - height below them
  __shared__ double runs[4][4];
  const int bx = xcd_col(); // hip_common.h: the column block, pinned to an XCD for 64 rows of the walk
  if(bx >= nseg) return;
  const int row = walk_row(blockIdx.y, height, mult);
  if(row < 0) return;
  const int col = bx * 256 + threadIdx.x;
  double sq[4] = { 0.0, 0.0, 0.0, 0.0 };
  if(col < width)
  {
    const float4 px = in[(size_t)(row + in_row0) * width + col];
    float sum[4] = { 0.f, 0.f, 0.f, 0.f }, wgt = 0.f;
#pragma unroll
    for(int jj = 0; jj < 5; jj++)
    {
      const size_t y = (size_t)clampi(row + in_row0 + mult * (jj - 2), 0, in_rows - 1) * width;
      const float fj = jj == 0 || jj == 4 ? 0.0625f : (jj == 2 ? 0.375f : 0.25f);
#pragma unroll
      for(int ii = 0; ii < 5; ii++)
      {
        const int x = clampi(col + mult * (ii - 2), 0, width - 1);
        const float fi = ii == 0 || ii == 4 ? 0.0625f : (ii == 2 ? 0.375f : 0.25f);
        // straight from L1 / L2: staging the five tap rows in LDS was measured (2.96 ms against 2.85 ms at 100 MP) --
        // the kernel is bound by the ~28 VALU instructions of each tap, not by its fetch
        const float4 p2 = in[y + x];
        // dn_weight(), eaw.c:181-195
        const float dx = px.x - p2.x, dy = px.y - p2.y, dz = px.z - p2.z;
        const float dot = (dx * dx + dy * dy + dz * dz) * inv_sigma2;
        const float arg = dot * 0.02f - 9.0f;
        const float wp = mexp2_of_clamped(arg);
        const float w = (fi * fj) * wp;
        wgt += w; // the reference keeps one weight sum per channel; they are identical
        sum[0] += w * p2.x;
        sum[1] += w * p2.y;
        sum[2] += w * p2.z;
        sum[3] += w * p2.w;
      }
    }
    float c4[4], d4[4];
    const float pin[4] = { px.x, px.y, px.z, px.w };
#pragma unroll
    for(int c = 0; c < 4; c++)
    {
      c4[c] = sum[c] / wgt;
      d4[c] = pin[c] - c4[c];
      sq[c] = (double)(d4[c] * d4[c]);
    }
    const size_t o = (size_t)row * width + col;
    coarse[o] = make_float4(c4[0], c4[1], c4[2], c4[3]);
    if(detail) detail[o] = make_float4(d4[0], d4[1], d4[2], d4[3]); // (nullptr: the synthesis forms it from two coarse planes)
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for(int c = 0; c < 4; c++)
  {
    const double s = wave_sum_halving(sq[c]);
    if(lane == 0) runs[wave][c] = s;
  }
  __syncthreads();
  if(threadIdx.x < 4)
  {
    const int c = threadIdx.x;
    const double seg = ((runs[0][c] + runs[1][c]) + runs[2][c]) + runs[3][c];
    partial[4 * ((size_t)row * nseg + bx) + c] = seg;
  }
}
#endif // ANSEL_HIP_MEASURING

// The same step on STRIPS: one workgroup = one 256-pixel segment x up to `strip` rows of one dilation class (rows c,
// c + mult, c + 2 mult, ...: consecutive rows of the class share four of their five tap rows).  The tap rows live in a
// ring of six LDS rows of 256 + 4 mult pixels (the segment and the 2 mult columns either side, clamped like the
// reference's indices): per output row ONE new row is fetched -- while the previous output row is computed -- and the 25
// taps are LDS reads, against 25 fetches per pixel through the vector L1 (64 B/clk per CU: ~400 cycles of the ~1 080 a
// wave of the per-row kernel took).  Same taps in the same order, so the same binary32 values; the partial sums of
// detail^2 leave in the same per-(row, segment) slots.
// PRE: `in` is the module's input and the variance-stabilising transform is applied to every sample as it enters the
// ring (the finest scale of a whole frame: a row is fetched 1.14 times per strip, which costs less than the pass that
// wrote and re-read the transformed plane: 0.89 + 1.9 -> 2.4 ms at 100 MP)
#define DN_RING 6
// dn_weight(), eaw.c:181-195, without the filter coefficient
__device__ __forceinline__ float dn_photometric(const float4 px, const float4 p2, const float inv_sigma2)
{
  const float dx = px.x - p2.x, dy = px.y - p2.y, dz = px.z - p2.z;
  const float dot = (dx * dx + dy * dy + dz * dz) * inv_sigma2;
  const float arg = dot * 0.02f - 9.0f;
  return mexp2_of_clamped(arg);
}
// The halving-tree sums of FOUR quantities over the wave in one tree (round 4; wave_sum_halving() four times was 72 of the
// kernel's ~830 instructions per pixel): the tree adds lane i + 32 onto lane i, then i + 16, ... -- in the first step only
// half the lanes do useful work, in the second a quarter.  v_permlane32_swap trades the upper half of one quantity for the
// lower half of another, so that ONE addition forms the first level of both (lanes 0-31: q0[i] + q0[i + 32], lanes 32-63:
// q1[i - 32] + q1[i], the same two operands as the tree's); v_permlane16_swap does the same with the 16-lane rows for
// the second level of all four; the last four levels stay inside a row (DPP row_shl).  Every sum adds the same pairs as
// wave_sum_halving() -- binary64 addition commutes -- so the same bits.  Returns, in lanes 0 / 16 / 32 / 48, the sums of
// q0 / q2 / q1 / q3.
__device__ __forceinline__ double wave_sum4_halving(const double q0, const double q1, const double q2, const double q3)
{
  union d2
  {
    double d;
    unsigned u[2];
  };
  auto level = [](const double a, const double b, const bool rows16) {
    d2 x, y, rx, ry;
    x.d = a;
    y.d = b;
#pragma unroll
    for(int h = 0; h < 2; h++)
    {
      const auto r = rows16 ? __builtin_amdgcn_permlane16_swap(x.u[h], y.u[h], false, false)
                            : __builtin_amdgcn_permlane32_swap(x.u[h], y.u[h], false, false);
      rx.u[h] = r[0];
      ry.u[h] = r[1];
    }
    return rx.d + ry.d;
  };
  const double s01 = level(q0, q1, false), s23 = level(q2, q3, false);
  d2 t;
  t.d = level(s01, s23, true);
#define DN_ROW_SHL(n)                                                                                     \
  {                                                                                                       \
    d2 o;                                                                                                 \
    o.u[0] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)t.u[0], 0x100 + (n), 0xf, 0xf, true);          \
    o.u[1] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)t.u[1], 0x100 + (n), 0xf, 0xf, true);          \
    t.d = t.d + o.d;                                                                                      \
  }
  DN_ROW_SHL(8)
  DN_ROW_SHL(4)
  DN_ROW_SHL(2)
  DN_ROW_SHL(1)
#undef DN_ROW_SHL
  return t.d;
}

// MULT: the dilation at compile time (1 .. 64: the seven scales of a frame), so that the 25 taps of a lane are one base
// address per tap row + immediates (the counters had 11 % of this kernel's VALU instructions as integer address
// arithmetic); 0: read `mult_arg`
// ALPHA0: the fourth channel of every sample this launch reads is +0 -- what the Y0U0V0 transform leaves there (it sets the
// channel to 0.0f), and what a decomposition leaves in its coarse plane when its input's was: the 25 weighted taps of the
// channel are then +0 (a weight is finite and >= 0), the coarse value is +0 / wgt -- +0 unless the weights sum to 0 or NaN, which a
// pixel that is not finite makes them do -- and the detail +0 - that.  The launch leaves the channel's 50 multiply-adds and its
// division out (the division is made where a wave holds such a pixel) and RAISES `alpha_flag` when a coarse alpha it writes is
// not +0: the next scale's ALPHA0 launch (flag_sense 1) then leaves at once and the four-channel launch behind it (flag_sense
// 2) runs instead.  flag_sense 0: run whatever the flag says (the first scale: its input is +0 by construction).
// (the kernel's arguments as the kernarg segment lays them out: where `fa` sits, for kernarg_at())
struct dn_strip_kernargs
{
  const float4 *in;
  float4 *coarse, *detail;
  double *partial;
  int width, height, mult_arg;
  float inv_sigma2;
  int nseg, in_row0, in_rows, strip, strips_per_class;
  vst_args fa;
  unsigned *alpha_flag;
  int flag_sense;
};
// Waves per SIMD the register allocation aims at: eight (64 VGPRs) where the ring lets eight workgroups share a CU -- the
// kernel answers to its occupancy, profiles/r04_negative_results.txt item 8 --, as many as the ring allows at the large
// dilations (MULT 32 / 64: 12 - 16 bytes x 6 x (256 + 4 MULT) of LDS a workgroup, five to three of them a CU), and six for
// the transform-applying launch at a dilation only known at run time (it would need 66 registers: no scratch instead)
template <bool PRE, int MULT, bool ALPHA0> constexpr int dn_strip_waves()
{
  if(PRE && MULT == 0) return 6;
  const int lds = DN_RING * (256 + 4 * (MULT ? MULT : 1)) * (ALPHA0 ? 12 : 16);
  const int wgs = (160 * 1024) / lds;
  return wgs >= 8 ? 8 : (wgs < 1 ? 1 : wgs);
}
// P3 (ALPHA0 launches of the unsplit frame, round 6): the planes this launch reads and writes hold THREE floats a pixel -- the fourth
// channel of every one of them is +0 by the launch's premise, so it is not stored (12 instead of 16 B/px either way; the module's
// own input, which the transform-applying launch reads, stays float4).  A launch that finds the premise broken raises the flag and
// the caller's four-channel sequence behind it redoes the module (denoiseprofile_run()).
template <bool PRE, int MULT, bool ALPHA0, bool P3 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(dn_strip_waves<PRE, MULT, ALPHA0>(), 8))) void dn_decompose_strip(const float4 *__restrict__ in, float4 *__restrict__ coarse,
                                                          float4 *__restrict__ detail, double *__restrict__ partial,
                                                          const int width, const int height, const int mult_arg,
                                                          const float inv_sigma2, const int nseg, const int in_row0,
                                                          const int in_rows, const int strip, const int strips_per_class,
                                                          const vst_args fa_by_value, unsigned *__restrict__ alpha_flag, const int flag_sense)
{
  // the transform's ~40 parameters are read from the kernarg segment where a fetch uses them: as a by-value argument they
  // sat in scalar registers for the whole row loop and 53 of them were spilled to vector lanes (hip_common.h kernarg_at())
  const vst_args &fa = kernarg_at<vst_args>((int)offsetof(dn_strip_kernargs, fa));
  (void)fa_by_value;
  // PRE: three powf a fetched sample, two dependent table lookups each -- as vector-memory round trips they were what the
  // transform-applying launch waited for (round 5's counters: 68 % of its cycles); from the workgroup's LDS copy now
  if(PRE) ansel_math::stage_default_tables(threadIdx.x);
  if(flag_sense == 1 && *alpha_flag) return;
  if(flag_sense == 2 && !*alpha_flag) return;
  unsigned alpha_bits = 0; // ALPHA0: the coarse alphas this lane wrote
  const int mult = MULT ? MULT : mult_arg;
  constexpr bool LATE2 = PRE && MULT == 1;
  const bool have_t3 = PRE && fa.vst == 2;
  const float t3_of_zero = have_t3 ? dn_vst_y0u0v0_alpha(0.0f, fa) : 0.0f;
#define DN_FETCH(p) (PRE ? dn_precondition_pixel((p), fa, t3_of_zero, have_t3) : (p))
  static_assert(!P3 || ALPHA0, "three-float planes are the ALPHA0 launches'");
  // sample `i` of the input: a float4, or (P3, not the module's own input) three floats and +0
  auto in_at = [&](const size_t i) -> float4 {
    if(P3 && !PRE)
    {
      const float *const p3 = reinterpret_cast<const float *>(in) + 3 * i;
      return make_float4(p3[0], p3[1], p3[2], 0.0f);
    }
    return in[i];
  };
  auto plane_put = [&](float4 *const plane, const size_t i, const float v0, const float v1, const float v2, const float v3) {
    if(P3)
    {
      float *const p3 = reinterpret_cast<float *>(plane) + 3 * i;
      p3[0] = v0;
      p3[1] = v1;
      p3[2] = v2;
    }
    else
      plane[i] = make_float4(v0, v1, v2, v3);
  };
  __shared__ double runs[2][4][4];
  const int bx = blockIdx.x;
  const int cls = blockIdx.y / strips_per_class, k0 = (blockIdx.y - cls * strips_per_class) * strip;
  const int n_cls = (height - cls + mult - 1) / mult; // rows of this class
  if(k0 >= n_cls) return;
  const int nrows = min(strip, n_cls - k0);
  const int r_first = cls + k0 * mult;
  const int tw = 256 + 4 * mult;
  // the rows' samples: [DN_RING][256 + 4 * mult] float4 -- or, for an ALPHA0 launch, three planes of floats a row (the fourth
  // channel is +0 and is not kept): 12 bytes a sample instead of 16 are two more workgroups a CU (the kernel loses 11 % with
  // 8 KB of LDS more a workgroup, measured: profiles/r04_negative_results.txt item 8), and a tap's three ds_read_b32 move through
  // the LDS in 6 cycles where its ds_read_b128 took 8
  extern __shared__ float4 ring[];
  float *const ringf = reinterpret_cast<float *>(ring);
  auto ring_put = [&](const int slot, const int c, const float4 v) {
    if(ALPHA0)
    {
      ringf[(slot * 3 + 0) * tw + c] = v.x;
      ringf[(slot * 3 + 1) * tw + c] = v.y;
      ringf[(slot * 3 + 2) * tw + c] = v.z;
    }
    else
      ring[slot * tw + c] = v;
  };
  auto ring_get = [&](const int slot, const int c) -> float4 {
    if(ALPHA0) return make_float4(ringf[(slot * 3 + 0) * tw + c], ringf[(slot * 3 + 1) * tw + c], ringf[(slot * 3 + 2) * tw + c], 0.0f);
    return ring[slot * tw + c];
  };
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = bx * 256 + tid;
  // the one or two ring entries this thread fetches per row
  const int ecol0 = clampi(bx * 256 - 2 * mult + tid, 0, width - 1), ecol1 = clampi(bx * 256 - 2 * mult + tid + 256, 0, width - 1);
  const bool second = tid + 256 < tw;
  // ring row q of the strip = frame row r_first + (q - 2) mult, clamped into the input buffer like the reference's taps
#define DN_IN_ROW(q) ((size_t)clampi(r_first + ((q) - 2) * mult + in_row0, 0, in_rows - 1) * width)
#pragma unroll
  for(int q = 0; q < 4; q++)
  {
    const size_t y = DN_IN_ROW(q);
    ring_put(q, tid, DN_FETCH(in_at(y + ecol0)));
    if(second) ring_put(q, tid + 256, DN_FETCH(in_at(y + ecol1)));
  }
  float4 n0, n1 = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    const size_t y = DN_IN_ROW(4);
    n0 = in_at(y + ecol0);
    if(!LATE2 && second) n1 = in_at(y + ecol1);
  }
  int s0 = 0; // ring slot of row k
  // the weights of the taps straight above, left by this lane one and two rows ago; for the strip's first two rows, whose
  // rows above belong to the strip in front, computed here from the ring rows that are in already
  float up1 = 0.f, up2 = 0.f, up2_next = 0.f;
  __syncthreads();
  {
    const float4 c0 = ring_get(2, tid + 2 * mult), c1 = ring_get(3, tid + 2 * mult);
    const float4 a0 = ring_get(0, tid + 2 * mult), a1 = ring_get(1, tid + 2 * mult);
    up2 = dn_photometric(c0, a0, inv_sigma2);      // row 0 <- two rows above
    up1 = dn_photometric(c0, a1, inv_sigma2);      // row 0 <- the row above
    up2_next = dn_photometric(c1, a1, inv_sigma2); // row 1 <- two rows above
  }
  for(int k = 0; k < nrows; k++)
  {
    {
      const int sl = s0 + 4 >= DN_RING ? s0 + 4 - DN_RING : s0 + 4;
      ring_put(sl, tid, DN_FETCH(n0));
      // LATE2: the (few) lanes with a second ring entry fetch it here, where it is transformed, instead of holding it in
      // four more registers across the 25 taps -- what pushed the launch that applies the transform over its 64 registers
      // (12 - 24 bytes of scratch, folded reloads in the tap loop: round 4's review).  At dilation 1 they are four lanes of
      // the workgroup's first wave; the seven other workgroups of the CU cover the fetch
      if(LATE2) { if(second) ring_put(sl, tid + 256, DN_FETCH(in_at(DN_IN_ROW(k + 4) + ecol1))); }
      else if(second) ring_put(sl, tid + 256, DN_FETCH(n1));
    }
    __syncthreads();
    if(k > 0 && tid < 4)
    {
      // the previous row's segment sum (its four run sums were complete at the barrier)
      const int c = tid, pr = (k - 1) & 1;
      const double seg = ((runs[pr][0][c] + runs[pr][1][c]) + runs[pr][2][c]) + runs[pr][3][c];
      partial[4 * ((size_t)(r_first + (k - 1) * mult) * nseg + bx) + c] = seg;
    }
    if(k + 1 < nrows)
    {
      const size_t y = DN_IN_ROW(k + 5);
      n0 = in_at(y + ecol0);
      if(!LATE2 && second) n1 = in_at(y + ecol1);
    }
    const int row = r_first + k * mult;
    double sq[4] = { 0.0, 0.0, 0.0, 0.0 };
    if(col < width)
    {
      const int sc = s0 + 2 >= DN_RING ? s0 + 2 - DN_RING : s0 + 2;
      const float4 px = ring_get(sc, tid + 2 * mult);
      float sum[4] = { 0.f, 0.f, 0.f, 0.f }, wgt = 0.f, down1 = 0.f, down2 = 0.f;
#pragma unroll
      for(int jj = 0; jj < 5; jj++)
      {
        const int sl = s0 + jj >= DN_RING ? s0 + jj - DN_RING : s0 + jj;
        const float fj = jj == 0 || jj == 4 ? 0.0625f : (jj == 2 ? 0.375f : 0.25f);
        float4 tap[5];
#pragma unroll
        for(int ii = 0; ii < 5; ii++) tap[ii] = ring_get(sl, tid + ii * mult);
#pragma unroll
        for(int ii = 0; ii < 5; ii++)
        {
          const float fi = ii == 0 || ii == 4 ? 0.0625f : (ii == 2 ? 0.375f : 0.25f);
          const float4 p2 = tap[ii];
          // dn_weight(), eaw.c:181-195.  It squares the differences of the two pixels, so the weight of this pixel's tap
          // straight above is bit for bit the weight the pixel up there computed for its tap straight below: this lane,
          // one or two rows ago (the strip's first two rows: the prologue) -- no branch in the 25 taps
          float wp;
          if(ii == 2 && jj == 0) wp = up2;
          else if(ii == 2 && jj == 1) wp = up1;
          else wp = dn_photometric(px, p2, inv_sigma2);
          if(ii == 2 && jj == 3) down1 = wp;
          if(ii == 2 && jj == 4) down2 = wp;
          const float w = (fi * fj) * wp;
          wgt += w;
          sum[0] += w * p2.x;
          sum[1] += w * p2.y;
          sum[2] += w * p2.z;
          if(!ALPHA0) sum[3] += w * p2.w;
        }
        // five reads in flight, not twenty-five: the next row's reads stay behind the sums of this one
        asm volatile("" : "+v"(wgt), "+v"(sum[0]), "+v"(sum[1]), "+v"(sum[2]), "+v"(sum[3]) : : "memory");
      }
      up1 = down1;
      up2 = up2_next;
      up2_next = down2;
      float c4[4], d4[4];
      const float pin[4] = { px.x, px.y, px.z, px.w };
#pragma unroll
      for(int c = 0; c < 4; c++)
      {
        if(ALPHA0 && c == 3)
        {
          // +0 / wgt: +0 for a positive sum of weights (at most 1: no infinity), the division's own result where a lane of
          // the wave holds another (a uniform branch)
          c4[3] = __builtin_amdgcn_ballot_w64(!(wgt > 0.0f)) == 0ull ? 0.0f : 0.0f / wgt;
          alpha_bits |= __float_as_uint(c4[3]);
        }
        else
          c4[c] = sum[c] / wgt;
        d4[c] = pin[c] - c4[c];
        sq[c] = (double)(d4[c] * d4[c]);
      }
      const size_t o = (size_t)row * width + col;
      plane_put(coarse, o, c4[0], c4[1], c4[2], c4[3]);
      if(detail) plane_put(detail, o, d4[0], d4[1], d4[2], d4[3]); // (nullptr: the synthesis forms it from two coarse planes)
    }
    {
      const double s = wave_sum4_halving(sq[0], sq[1], sq[2], sq[3]);
      if((lane & 15) == 0) runs[k & 1][wave][((lane >> 5) & 1) | ((lane >> 3) & 2)] = s; // lanes 0 / 16 / 32 / 48: channels 0 / 2 / 1 / 3
    }
    s0 = s0 + 1 == DN_RING ? 0 : s0 + 1;
  }
#undef DN_IN_ROW
#undef DN_FETCH
  __syncthreads();
  if(tid < 4)
  {
    const int c = tid, pr = (nrows - 1) & 1;
    const double seg = ((runs[pr][0][c] + runs[pr][1][c]) + runs[pr][2][c]) + runs[pr][3][c];
    partial[4 * ((size_t)(r_first + (nrows - 1) * mult) * nseg + bx) + c] = seg;
  }
  if(ALPHA0 && alpha_flag && __builtin_amdgcn_ballot_w64(alpha_bits != 0) != 0ull && lane == 0) atomicOr(alpha_flag, 1u);
}

// one a-trous step of `height` rows (see dn_decompose for in_row0 / in_rows)
static void launch_decompose(hipStream_t st, const float4 *in, float4 *coarse, float4 *detail, double *partial, const int width,
                             const int height, const int mult, const float inv_sigma2, const int nseg, const int in_row0,
                             const int in_rows, const vst_args *pre = nullptr, unsigned *alpha_flag = nullptr, const int mode = 0)
{
  // alpha_flag != nullptr: the caller's input alpha is +0 by construction at its first scale (the Y0U0V0 transform, vst 2) and
  // the flag carries "still +0" from scale to scale: the three-channel launch first, the four-channel one behind it.
  // mode 1 (the unsplit frame's first sequence): the ALPHA0 launch on three-float planes, whatever the flag says; mode 2 (its
  // second sequence): the four-channel launch on float4 planes, which leaves at once unless the flag is up
#ifdef ANSEL_HIP_MEASURING
  static const bool per_row = getenv("ANSEL_HIP_DN_PER_ROW") != nullptr; // the per-row kernel, for A/B timing
  if(per_row && !pre)
  {
    const int rows = (height <= mult) ? height : ((height + mult - 1) / mult) * mult;
    dn_decompose<<<dim3(xcd_pad(nseg), rows), 256, 0, st>>>(in, coarse, detail, partial, width, height, mult, inv_sigma2, nseg,
                                                            in_row0, in_rows);
    return;
  }
#endif
  const int classes = height < mult ? height : mult, per_class = (height + mult - 1) / mult;
  int strip = 32;
  while(strip > 4 && (size_t)nseg * classes * ((per_class + strip - 1) / strip) < 2048) strip /= 2;
  const int strips_per_class = (per_class + strip - 1) / strip;
  const dim3 grid(nseg, classes * strips_per_class);
  // (an ALPHA0 launch keeps three floats a sample: DN_LDS())
  size_t lds = (size_t)DN_RING * (256 + 4 * mult) * sizeof(float);
#ifdef ANSEL_HIP_MEASURING
  // how much the kernel needs its occupancy: LDS asked for and not used (bytes), for A/B timing
  static const char *const pad_env = measuring_env("ANSEL_HIP_DN_LDS_PAD");
  const size_t lds_pad = pad_env ? (size_t)atoi(pad_env) : 0;
#else
  const size_t lds_pad = 0;
#endif
#define DN_LDS(A0_) (lds * ((A0_) ? 3 : 4) + lds_pad)
  vst_args none;
  memset(&none, 0, sizeof(none));
#define DN_LAUNCH_(PRE_, M_, A0_, FA_, SENSE_)                                                                                          \
  dn_decompose_strip<PRE_, M_, A0_><<<grid, 256, DN_LDS(A0_), st>>>(in, coarse, detail, partial, width, height, mult, inv_sigma2, nseg, in_row0, \
                                                            in_rows, strip, strips_per_class, FA_, alpha_flag, SENSE_)
#define DN_LAUNCH(PRE_, M_, FA_)                  \
  do                                              \
  {                                               \
    if(mode == 1)                                 \
      dn_decompose_strip<PRE_, M_, true, true><<<grid, 256, DN_LDS(true), st>>>(in, coarse, detail, partial, width, height, mult, inv_sigma2, nseg, in_row0, \
                                                            in_rows, strip, strips_per_class, FA_, alpha_flag, 0); \
    else if(mode == 2)                            \
      DN_LAUNCH_(PRE_, M_, false, FA_, 2);        \
    else if(alpha_flag && (PRE_))                 \
      DN_LAUNCH_(PRE_, M_, true, FA_, 0);         \
    else if(alpha_flag)                           \
    {                                             \
      DN_LAUNCH_(PRE_, M_, true, FA_, 1);         \
      DN_LAUNCH_(PRE_, M_, false, FA_, 2);        \
    }                                             \
    else                                          \
      DN_LAUNCH_(PRE_, M_, false, FA_, 0);        \
  } while(0)
  if(pre)
  {
    if(mult == 1) DN_LAUNCH(true, 1, *pre);
    else DN_LAUNCH(true, 0, *pre);
  }
  else
    switch(mult)
    {
      case 1: DN_LAUNCH(false, 1, none); break;
      case 2: DN_LAUNCH(false, 2, none); break;
      case 4: DN_LAUNCH(false, 4, none); break;
      case 8: DN_LAUNCH(false, 8, none); break;
      case 16: DN_LAUNCH(false, 16, none); break;
      case 32: DN_LAUNCH(false, 32, none); break;
      case 64: DN_LAUNCH(false, 64, none); break;
      default: DN_LAUNCH(false, 0, none); break;
    }
#undef DN_LAUNCH
#undef DN_LAUNCH_
#undef DN_LDS
}

struct thr_args
{
  size_t n_partial;
  float n1;        // (float)npixels - 1.0f
  float sb2;       // sigma_band^2
  float adjt[4];   // 8 x the band's force factors
};

// step 2 of the canonical sum.  Segment sums, numbered row-major, go round-robin to 1024 accumulators, each adding
// its segments in increasing order; accumulator t is thread t % 64 of workgroup t / 64 (the order of the additions
// does not depend on which workgroup holds an accumulator; sixteen CUs read the table sixteen times faster than one)
#define THR_GROUPS 16
// (blockIdx.y = the band: the tables and accumulators of all bands of a frame in one launch)
__global__ __launch_bounds__(64) void dn_band_sums(const double *__restrict__ partial_all, const size_t n_partial,
                                                  double *__restrict__ acc_all /* [bands][4][1024] */)
{
  const double *const partial = partial_all + (size_t)blockIdx.y * n_partial * 4;
  double *const acc = acc_all + (size_t)blockIdx.y * 4 * 1024;
  const int t = blockIdx.x * 64 + threadIdx.x;
  double s[4] = { 0.0, 0.0, 0.0, 0.0 };
  typedef double d4 __attribute__((ext_vector_type(4)));
  const d4 *const p4 = (const d4 *)partial;
  size_t k = t;
  for(; k + 7 * 1024 < n_partial; k += 8 * 1024)
  {
    d4 v[8];
#pragma unroll
    for(int u = 0; u < 8; u++) v[u] = p4[k + (size_t)u * 1024];
#pragma unroll
    for(int u = 0; u < 8; u++)
    {
      s[0] += v[u].x;
      s[1] += v[u].y;
      s[2] += v[u].z;
      s[3] += v[u].w;
    }
  }
  for(; k < n_partial; k += 1024)
  {
    const d4 v = p4[k];
    s[0] += v.x;
    s[1] += v.y;
    s[2] += v.z;
    s[3] += v.w;
  }
#pragma unroll
  for(int c = 0; c < 4; c++) acc[c * 1024 + t] = s[c];
}

// the 1024 accumulators reduced by halving (off = 512..1), rounded once to binary32, then
// variance_stabilizing_xform(), denoiseprofile.c:1223-1287
struct thr_args_all
{
  thr_args band[BANDS];
};
__global__ __launch_bounds__(1024) void dn_band_threshold(const double *__restrict__ sums_all /* [bands][4][1024] */,
                                                          const thr_args_all all, float *__restrict__ thrs_all)
{
  __shared__ double acc[4][1024];
  const double *const sums = sums_all + (size_t)blockIdx.x * 4 * 1024;
  const thr_args &a = all.band[blockIdx.x];
  float *const thrs = thrs_all + 4 * blockIdx.x;
  const int t = threadIdx.x;
#pragma unroll
  for(int c = 0; c < 4; c++) acc[c][t] = sums[c * 1024 + t];
  __syncthreads();
  for(int off = 512; off >= 1; off >>= 1)
  {
    if(t < off)
    {
#pragma unroll
      for(int c = 0; c < 4; c++) acc[c][t] += acc[c][t + off];
    }
    __syncthreads();
  }
  if(t < 4)
  {
    const float sum_y2 = (float)acc[t][0];
    const float std_x = t < 3 ? sqrtf(max_first(1e-6f, sum_y2 / a.n1 - a.sb2)) : 1.0f;
    thrs[t] = a.adjt[t] * a.sb2 / std_x;
  }
}

// ---- host side ---------------------------------------------------------------------------------

// invert_matrix(), denoiseprofile.c:1132-1161
bool invert3(const float in[3][4], float out[3][4])
{
  const float A = in[1][1] * in[2][2] - in[1][2] * in[2][1];
  const float B = -in[1][0] * in[2][2] + in[1][2] * in[2][0];
  const float C = in[1][0] * in[2][1] - in[1][1] * in[2][0];
  const float D = -in[0][1] * in[2][2] + in[0][2] * in[2][1];
  const float E = in[0][0] * in[2][2] - in[0][2] * in[2][0];
  const float F = -in[0][0] * in[2][1] + in[0][1] * in[2][0];
  const float G = in[0][1] * in[1][2] - in[0][2] * in[1][1];
  const float H = -in[0][0] * in[1][2] + in[0][2] * in[1][0];
  const float I = in[0][0] * in[1][1] - in[0][1] * in[1][0];
  const float det = in[0][0] * A + in[0][1] * B + in[0][2] * C;
  if(det == 0.0f) return false;
  const float r = 1.0f / det;
  out[0][0] = r * A; out[0][1] = r * D; out[0][2] = r * G; out[0][3] = 0.0f;
  out[1][0] = r * B; out[1][1] = r * E; out[1][2] = r * H; out[1][3] = 0.0f;
  out[2][0] = r * C; out[2][1] = r * F; out[2][2] = r * I; out[2][3] = 0.0f;
  return true;
}

// nlm false: process_wavelets() :1331-1378; nlm true: nlmeans_precondition() :1510-1546
void setup(const dt_hip_piece_t *piece, const dt_hip_denoiseprofile_data_t *d, dn_setup &s, const bool nlm)
{
  memset(&s, 0, sizeof(s));
  const float in_scale = fminf((float)piece->roi_in.scale, 1.0f);
  // number of bands, denoiseprofile.c:1301-1317 (piece->buf_in == roi_in for a full-frame export)
  const float big = (float)(piece->roi_in.height > piece->roi_in.width ? piece->roi_in.height : piece->roi_in.width);
  const float supp0 = min_first((float)(2 * (2u << (BANDS - 1)) + 1), big * 0.2f);
  const float i0 = log2f((supp0 - 1.0f) * .5f);
  int max_scale = 0;
  for(; max_scale < BANDS; max_scale++)
  {
    const float supp = (float)(2 * (2u << max_scale) + 1);
    const float supp_in = supp * (1.0f / in_scale);
    const float i_in = log2f((supp_in - 1) * .5f) - 1.0f;
    const float t = 1.0f - (i_in + .5f) / i0;
    if(t < 0.0f) break;
  }
  s.max_scale = max_scale;

  // compute_wb_factors() with weights {2, 1, 2, 0}, :1097-1128
  const float weights[4] = { nlm ? 1.0f : 2.0f, 1.0f, nlm ? 1.0f : 2.0f, 0.0f };
  const float wb_mean = (d->wb_coeffs[0] + d->wb_coeffs[1] + d->wb_coeffs[2]) / 3.0f;
  float *wb = s.wb;
  wb[0] = wb[1] = wb[2] = wb[3] = wb_mean;
  if(d->fix_anscombe_and_nlmeans_norm)
  {
    if(wb_mean != 0.0f && d->wb_adaptive_anscombe)
      for(int i = 0; i < 3; i++) wb[i] = d->wb_coeffs[i];
    else if(wb_mean == 0.0f)
      for(int i = 0; i < 4; i++) wb[i] = 1.0f;
  }
  else
    for(int i = 0; i < 4; i++) wb[i] = weights[i] * piece->processed_maximum[i];

  // adaptive p, :1339-1343 (a binary64 expression stored to binary32)
  for(int i = 0; i < 3; i++)
  {
    const double v = (double)d->shadows + 0.1 * (double)logf(in_scale / wb[i]);
    s.p[i] = (float)(v > 0.0 ? v : 0.0);
  }
  s.p[3] = 0.0f;
  const float compensate_p = P_FULCRUM / powf(P_FULCRUM, d->shadows);

  // set_up_conversion_matrices(), :1163-1221
  float toY[3][4] = { { 1.0f / 3.0f, 1.0f / 3.0f, 1.0f / 3.0f, 0.0f }, { 0.5f, 0.0f, -0.5f, 0.0f }, { 0.25f, -0.5f, 0.25f, 0.0f } };
  float toRGB[3][4];
  memset(toRGB, 0, sizeof(toRGB));
  float sum_invwb = 1.0f / wb[0] + 1.0f / wb[1] + 1.0f / wb[2];
  sum_invwb *= sqrtf(3);
  toY[0][0] = sum_invwb / wb[0];
  toY[0][1] = sum_invwb / wb[1];
  toY[0][2] = sum_invwb / wb[2];
  toY[0][3] = 0.0f;
  const float sdU = sqrtf(0.5f * 0.5f * wb[0] * wb[0] + 0.5f * 0.5f * wb[2] * wb[2]);
  const float sdV = sqrtf(0.25f * 0.25f * wb[0] * wb[0] + 0.5f * 0.5f * wb[1] * wb[1] + 0.25f * 0.25f * wb[2] * wb[2]);
  for(int c = 0; c < 3; c++)
  {
    toY[1][c] /= sdU;
    toY[2][c] /= sdV;
  }
  toY[1][3] = toY[2][3] = 0.0f;
  if(!invert3(toY, toRGB))
  {
    const float sdY = sqrtf(1.0f / 9.0f * (wb[0] * wb[0] + wb[1] * wb[1] + wb[2] * wb[2]));
    toY[0][0] = toY[0][1] = toY[0][2] = 1.0f / (3.0f * sdY);
    toY[0][3] = 0.0f;
    invert3(toY, toRGB);
  }
  const float compensate_strength = (d->wavelet_color_mode == DT_HIP_DENOISEPROFILE_RGB) ? 1.0f : 2.5f;
  const float gain = nlm ? d->strength * in_scale : d->strength * compensate_strength * in_scale;
  for(int k = 0; k < 3; k++)
    for(int c = 0; c < 4; c++)
    {
      toY[k][c] /= gain;
      toRGB[k][c] *= gain;
    }
  for(int i = 0; i < 4; i++) wb[i] *= gain;
  memcpy(s.toY, toY, sizeof(toY));
  memcpy(s.toRGB, toRGB, sizeof(toRGB));
  for(int i = 0; i < 3; i++)
  {
    s.aa[i] = d->a[1] * wb[i];
    s.bb[i] = d->b[1] * wb[i];
  }
  s.aa[3] = nlm ? d->a[1] * wb[3] : 0.0f;
  s.bb[3] = nlm ? d->b[1] * wb[3] : 0.0f;
  s.a_v2 = d->a[1] * compensate_p;
  s.b_v2 = d->b[1];
  s.bias = (float)((double)d->bias - 0.5 * (double)logf(in_scale));
  s.vst = !d->use_new_vst ? 0 : ((nlm || d->wavelet_color_mode == DT_HIP_DENOISEPROFILE_RGB) ? 1 : 2);
}

void forward_args(const dn_setup &s, vst_args &a)
{
  memset(&a, 0, sizeof(a));
  a.vst = s.vst;
  a.b = s.b_v2;
  const float sa = sqrtf(s.a_v2);
  for(int c = 0; c < 4; c++)
  {
    a.aa[c] = s.aa[c];
    a.wb[c] = s.wb[c];
  }
  if(s.vst == 0)
  {
    for(int c = 0; c < 3; c++) a.k[c] = (s.bb[c] / s.aa[c]) * (s.bb[c] / s.aa[c]) + 3.f / 8.f;
    a.k[3] = 0.0f;
    return;
  }
  for(int c = 0; c < 3; c++) a.expon[c] = -s.p[c] / 2 + 1;
  a.expon[3] = 1.0f;
  for(int c = 0; c < 3; c++) a.scale[c] = s.vst == 1 ? (-s.p[c] + 2) * sa : 2.0f / ((-s.p[c] + 2) * sa);
  a.scale[3] = 1.0f;
  memcpy(a.m, s.toY, sizeof(a.m));
}

void inverse_args(const dn_setup &s, vst_args &a)
{
  memset(&a, 0, sizeof(a));
  a.vst = s.vst;
  a.b = s.b_v2;
  a.bias = s.bias;
  a.sqrt_3_2 = sqrtf(3.0f / 2.0f);
  const float sa = sqrtf(s.a_v2);
  for(int c = 0; c < 4; c++)
  {
    a.aa[c] = s.aa[c];
    a.wb[c] = s.wb[c];
  }
  if(s.vst == 0)
  {
    for(int c = 0; c < 3; c++) a.k[c] = (s.bb[c] / s.aa[c]) * (s.bb[c] / s.aa[c]) + 1.f / 8.f;
    a.k[3] = 0.0f;
    return;
  }
  for(int c = 0; c < 3; c++) a.expon[c] = 1.0f / (1.0f - s.p[c] / 2.0f);
  a.expon[3] = 1.0f;
  for(int c = 0; c < 3; c++)
  {
    a.scale[c] = s.vst == 1 ? 4.0f / (sa * (2.0f - s.p[c])) : (sa * (2.0f - s.p[c])) / 4.0f;
    a.bias_wb[c] = s.bias * s.wb[c];
  }
  a.scale[3] = 1.0f;
  a.bias_wb[3] = 0.0f;
  memcpy(a.m, s.toRGB, sizeof(a.m));
}

// the host part of variance_stabilizing_xform(): everything that does not depend on the sums
void threshold_args(const dt_hip_denoiseprofile_data_t *d, const int scale, const int max_scale, const size_t npixels,
                    thr_args &t)
{
  const float varf = sqrtf(2.0f + 2.0f * 4.0f * 4.0f + 6.0f * 6.0f) / 16.0f;
  const float sigma_band = powf(varf, scale) * 1.0f;
  t.sb2 = sigma_band * sigma_band;
  t.n1 = (float)npixels - 1.0f;
  float adjt[4] = { 8.0f, 8.0f, 8.0f, 0.0f };
  const int band = BANDS - (scale + (BANDS - max_scale) + 1);
  if(d->wavelet_color_mode == DT_HIP_DENOISEPROFILE_RGB)
  {
    float f = d->force[0][band];
    f *= f;
    f *= 4;
    for(int c = 0; c < 4; c++) adjt[c] *= f;
    for(int c = 0; c < 3; c++)
    {
      f = d->force[1 + c][band];
      f *= f;
      f *= 4;
      adjt[c] *= f;
    }
  }
  else
  {
    float f = d->force[4][band];
    f *= f;
    f *= 4;
    adjt[0] *= f;
    f = d->force[5][band];
    f *= f;
    f *= 4;
    adjt[1] *= f;
    adjt[2] *= f;
  }
  for(int c = 0; c < 4; c++) t.adjt[c] = adjt[c];
}

// nlmeans_norm() :1457-1472 and nlmeans_scattering() :1476-1500 for an export pipe (no preview output, not a
// thumbnail): the dt_nlmeans_param_t of process_nlmeans_cpu(), denoiseprofile.c:1599-1648
nlm_core_params_t nlm_params_of(const dt_hip_piece_t *piece, const dt_hip_denoiseprofile_data_t *d)
{
  const float scale = fminf(fminf((float)piece->roi_in.scale, 2.0f), 1.0f);
  const int P = (int)ceilf(d->radius * scale);
  int K = (int)d->nbhood;
  float scattering = d->scattering;
  {
    const int maxk = (int)((K * K * K + 7.0 * K * sqrt((double)K)) * scattering / 6.0 + K);
    const float kf = (float)K * scale;
    const int k4 = K < 4 ? K : 4;
    K = (int)((float)k4 > kf ? (float)k4 : kf);
    scattering = (float)((maxk - K) * 6.0 / (K * K * K + 7.0 * K * sqrt((double)K)));
  }
  float norm = .045f / ((2 * P + 1) * (2 * P + 1));
  if(!d->fix_anscombe_and_nlmeans_norm) norm = .015f / (2 * P + 1);
  nlm_core_params_t p;
  memset(&p, 0, sizeof(p));
  p.scattering = scattering;
  p.scale = scale;
  p.luma = 1.0f;
  p.chroma = 1.0f;
  p.center_weight = d->central_pixel_weight * scale;
  p.sharpness = norm;
  p.patch_radius = P;
  p.search_radius = K;
  p.norm[0] = p.norm[1] = p.norm[2] = p.norm[3] = 1.0f;
  return p;
}

// process_nlmeans_cpu(), denoiseprofile.c:1599-1648.  On a row band (hip_common.h band_view_t) dev_in holds
// buf_rows rows from frame row band->buf_row0 on and dev_out the band's own rows.
int denoise_nlmeans(int devid, const dt_hip_piece_t *piece, const dt_hip_denoiseprofile_data_t *d, const band_view_t *band,
                    const int buf_rows, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  const int w = piece->roi_in.width, h = piece->roi_in.height;
  const size_t npix_in = (size_t)w * (band ? buf_rows : h);
  const size_t npix_out = (size_t)w * (band ? band->row1 - band->row0 : h);
  dn_setup s;
  setup(piece, d, s, true);
  float4 *pre = (float4 *)dt_hip_alloc_device_buffer(devid, npix_in * sizeof(float4));
  if(!pre) return DT_HIP_SYSMEM_ALLOCATION;
  hipStream_t st = stream_of(devid);
  {
    vst_args fa;
    forward_args(s, fa);
    launch_scope ls(devid, "dn_precondition");
    dn_precondition<<<pixel_grid(npix_in), 256, 0, st>>>((const float4 *)dev_in, pre, npix_in, fa);
  }
  nlm_core_params_t p = nlm_params_of(piece, d);
  p.band = band;
  int err = nlmeans_core_launch(devid, pre, (float4 *)dev_out, w, h, p);
  dt_hip_release_mem_object(pre);
  if(err != DT_HIP_SUCCESS) return err;
  {
    vst_args ia;
    inverse_args(s, ia);
    launch_scope ls(devid, "dn_finish");
    synth_args none;
    memset(&none, 0, sizeof(none));
    dn_finish<<<pixel_grid(npix_out), 256, 0, st>>>((float4 *)dev_out, nullptr, npix_out, ia, none);
  }
  return check_launch("dn_finish");
}

// what process_wavelets() refuses or copies through, on the FRAME's geometry: 1 = run, 0 = copy, < 0 error
int wavelets_runnable(const dt_hip_piece_t *piece, const dn_setup &s)
{
  const int w = piece->roi_in.width, h = piece->roi_in.height;
  if(s.max_scale < 1)
  {
    set_last_error("denoiseprofile: frame too small for a single wavelet band");
    return -1;
  }
  const int max_mult = 1 << (s.max_scale - 1);
  if(w < 2 * max_mult || h < 2 * max_mult) return 0; // denoiseprofile.c:1325-1329: too small, copy through
  if(w < 4 * max_mult)
  {
    // eaw.c:308-323 reads before the start of the row in this case (undefined in the reference)
    set_last_error("denoiseprofile: %d columns is less than 4x the coarsest dilation %d", w, max_mult);
    return -1;
  }
  return 1;
}

} // namespace

namespace ansel
{

// ---- row bands (pipe.cpp) --------------------------------------------------------------
// The band computes its OWN rows of every wavelet scale.  Scale k reads 2 * 2^k rows of scale k - 1 on either side, so
// before each decomposition the neighbours' rows of the current coarse plane are fetched: the module input comes in
// with 2 halo rows, and every decomposition writes its coarse plane into the middle of a buffer laid out for the next,
// twice as large, exchange.  No row is computed twice.
struct dn_band_job_t
{
  int devid, w, frame_h, max_scale, row0, rows;
  dt_hip_denoiseprofile_data_t d;
  dn_setup s;
  float4 *cur;          // coarse plane of scale next_scale - 1: [cur_top][rows][cur_bottom] rows
  int cur_top, cur_bottom;
  float4 *det[BANDS];   // own rows
  double *sums;         // [max_scale][frame_h * nseg][4]
  double *local;        // own rows' partial sums of one scale
  float *thrs;
  int next_scale;
  bool sums_requested;
};

void denoiseprofile_band_abort(dn_band_job_t *j)
{
  if(!j) return;
  if(j->cur) dt_hip_release_mem_object(j->cur);
  for(int k = 0; k < BANDS; k++)
    if(j->det[k]) dt_hip_release_mem_object(j->det[k]);
  if(j->sums) dt_hip_release_mem_object(j->sums);
  if(j->local) dt_hip_release_mem_object(j->local);
  if(j->thrs) dt_hip_release_mem_object(j->thrs);
  delete j;
}

// rows of the module INPUT a band needs from each neighbour; the wavelets then ask for 2 * 2^k rows of their own
// coarse planes before every later scale (denoiseprofile_band_step)
int denoiseprofile_halo_rows(const dt_hip_piece_t *piece, const dt_hip_denoiseprofile_data_t *d)
{
  if(DT_HIP_DENOISEPROFILE_IS_NLMEANS(d->mode)) return nlmeans_core_halo_rows(piece->roi_in.height, nlm_params_of(piece, d));
  dn_setup s;
  setup(piece, d, s, false);
  if(wavelets_runnable(piece, s) != 1) return -1;
  return 2;
}

static int clip_halo(const int h, const int avail) { return h < avail ? h : avail; }

int denoiseprofile_band_begin(int devid, const dt_hip_piece_t *piece, const dt_hip_denoiseprofile_data_t *d,
                              const band_view_t *band, const int buf_rows, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out,
                              dn_band_job_t **job)
{
  *job = nullptr;
  if(!valid_device(devid) || !piece || !d || !band || !dev_in || buf_rows <= 0) return DT_HIP_INVALID_ARG;
  if(piece->channels != 4 || !(piece->roi_in.scale > 0.0)) return DT_HIP_INVALID_ARG;
  if(DT_HIP_DENOISEPROFILE_IS_NLMEANS(d->mode)) return denoise_nlmeans(devid, piece, d, band, buf_rows, dev_in, dev_out);
  if(!DT_HIP_DENOISEPROFILE_IS_WAVELETS(d->mode)) return DT_HIP_INVALID_ARG;
  dn_band_job_t *j = new dn_band_job_t;
  memset(j, 0, sizeof(*j));
  j->devid = devid;
  j->w = piece->roi_in.width;
  j->frame_h = band->frame_h;
  j->row0 = band->row0;
  j->rows = band->row1 - band->row0;
  j->d = *d;
  setup(piece, d, j->s, false);
  if(wavelets_runnable(piece, j->s) != 1)
  {
    delete j;
    return DT_HIP_INVALID_ARG; // the planner (denoiseprofile_halo_rows) does not send such a frame here
  }
  j->max_scale = j->s.max_scale;
  const int w = j->w, nseg = (w + 255) / 256;
  const size_t npix = (size_t)w * buf_rows;
  const size_t n_frame = (size_t)band->frame_h * nseg; // partial sums of one wavelet band of the frame
  const size_t sums_bytes = (size_t)j->max_scale * n_frame * 4 * sizeof(double);
  j->cur_top = band->row0 - band->buf_row0;
  j->cur_bottom = buf_rows - j->cur_top - j->rows;
  bool ok = (j->cur = (float4 *)dt_hip_alloc_device_buffer(devid, npix * sizeof(float4))) != nullptr;
  ok &= (j->sums = (double *)dt_hip_alloc_device_buffer(devid, sums_bytes)) != nullptr;
  ok &= (j->local = (double *)dt_hip_alloc_device_buffer(devid, (size_t)j->rows * nseg * 4 * sizeof(double))) != nullptr;
  ok &= (j->thrs = (float *)dt_hip_alloc_device_buffer(devid, BANDS * 4 * sizeof(float))) != nullptr;
  if(!ok || j->cur_bottom < 0)
  {
    denoiseprofile_band_abort(j);
    return DT_HIP_SYSMEM_ALLOCATION;
  }
  hipStream_t st = stream_of(devid);
  if(hipMemsetAsync(j->sums, 0, sums_bytes, st) != hipSuccess)
  {
    denoiseprofile_band_abort(j);
    return DT_HIP_DEFAULT_ERROR;
  }
  {
    // pointwise, so the halo rows of the input become halo rows of the preconditioned plane
    vst_args fa;
    forward_args(j->s, fa);
    launch_scope ls(devid, "dn_precondition");
    dn_precondition<<<pixel_grid(npix), 256, 0, st>>>((const float4 *)dev_in, j->cur, npix, fa);
  }
  const int err = check_launch("dn_precondition");
  if(err != DT_HIP_SUCCESS)
  {
    denoiseprofile_band_abort(j);
    return err;
  }
  *job = j;
  return DT_HIP_SUCCESS;
}

// One decomposition per call.  Returns 1 with *halo_buf / *halo_rows set when the coarse plane just written needs the
// neighbours' rows before the next call, 2 with *sums / *sum_count set once every scale is done (the frame-wide table of
// partial sums to all-reduce), 0 when there is nothing left but denoiseprofile_band_finish(); < 0 on error (job freed).
int denoiseprofile_band_step(dn_band_job_t *j, dt_hip_mem_t *halo_buf, int *halo_rows, double **sums, size_t *sum_count)
{
  *halo_buf = nullptr;
  *halo_rows = 0;
  *sums = nullptr;
  *sum_count = 0;
  const int devid = j->devid, w = j->w, nseg = (w + 255) / 256;
  const size_t n_frame = (size_t)j->frame_h * nseg;
  if(j->next_scale >= j->max_scale)
  {
    if(j->sums_requested) return 0;
    j->sums_requested = true;
    *sums = j->sums;
    *sum_count = (size_t)j->max_scale * n_frame * 4;
    return 2;
  }
  hipStream_t st = stream_of(devid);
  const int scale = j->next_scale, mult = 1 << scale;
  const bool last = scale + 1 == j->max_scale;
  // the coarse plane of this scale, with room for the rows the next scale reads beyond the band
  const int h_next = last ? 0 : 2 * (2 << scale);
  const int top = clip_halo(h_next, j->row0), bottom = clip_halo(h_next, j->frame_h - j->row0 - j->rows);
  float4 *coarse = (float4 *)dt_hip_alloc_device_buffer(devid, (size_t)(top + j->rows + bottom) * w * sizeof(float4));
  j->det[scale] = (float4 *)dt_hip_alloc_device_buffer(devid, (size_t)j->rows * w * sizeof(float4));
  if(!coarse || !j->det[scale])
  {
    if(coarse) dt_hip_release_mem_object(coarse);
    denoiseprofile_band_abort(j);
    return DT_HIP_SYSMEM_ALLOCATION;
  }
  const float varf = sqrtf(2.0f + 2.0f * 4.0f * 4.0f + 6.0f * 6.0f) / 16.0f;
  const float sigma_band = powf(varf, scale) * 1.0f;
  {
    launch_scope ls(devid, "dn_decompose");
    launch_decompose(st, j->cur, coarse + (size_t)top * w, j->det[scale], j->local, w, j->rows, mult,
                     1.0f / (sigma_band * sigma_band), nseg, j->cur_top, j->cur_top + j->rows + j->cur_bottom);
  }
  int err = check_launch("denoiseprofile band decompose");
  // the own rows' partial sums at their place in the frame's table
  if(err == DT_HIP_SUCCESS
     && hipMemcpyAsync(j->sums + ((size_t)scale * n_frame + (size_t)j->row0 * nseg) * 4, j->local,
                       (size_t)j->rows * nseg * 4 * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess)
    err = DT_HIP_DEFAULT_ERROR;
  dt_hip_release_mem_object(j->cur); // stream-ordered
  j->cur = coarse;
  j->cur_top = top;
  j->cur_bottom = bottom;
  j->next_scale++;
  if(err != DT_HIP_SUCCESS)
  {
    denoiseprofile_band_abort(j);
    return err;
  }
  if(!last && (top || bottom))
  {
    *halo_buf = coarse;
    *halo_rows = h_next;
    return 1;
  }
  return denoiseprofile_band_step(j, halo_buf, halo_rows, sums, sum_count);
}

// thresholds from the reduced sums, synthesis, inverse transform on the band's own rows.  Frees the job
int denoiseprofile_band_finish(dn_band_job_t *j, dt_hip_mem_t dev_out)
{
  if(!j || !dev_out) return DT_HIP_INVALID_ARG;
  const int devid = j->devid, nseg = (j->w + 255) / 256;
  const size_t npix = (size_t)j->w * j->rows, n_frame = (size_t)j->frame_h * nseg;
  hipStream_t st = stream_of(devid);
  float4 *out = (float4 *)dev_out;
  int err = DT_HIP_SUCCESS;
  synth_args sy;
  memset(&sy, 0, sizeof(sy));
  sy.nbands = j->max_scale;
  sy.thrs = j->thrs;
  double *accs = (double *)dt_hip_alloc_device_buffer(devid, (size_t)BANDS * 4 * 1024 * sizeof(double));
  if(!accs) err = DT_HIP_SYSMEM_ALLOCATION;
  thr_args_all all;
  memset(&all, 0, sizeof(all));
  for(int scale = 0; scale < j->max_scale; scale++)
  {
    all.band[scale].n_partial = n_frame;
    threshold_args(&j->d, scale, j->max_scale, (size_t)j->w * j->frame_h, all.band[scale]);
    sy.detail[scale] = j->det[scale];
  }
  if(err == DT_HIP_SUCCESS)
  {
    launch_scope ls(devid, "dn_band_threshold");
    dn_band_sums<<<dim3(THR_GROUPS, j->max_scale), 64, 0, st>>>(j->sums, n_frame, accs);
    dn_band_threshold<<<j->max_scale, 1024, 0, st>>>(accs, all, j->thrs);
    err = check_launch("denoiseprofile band threshold");
  }
  if(err == DT_HIP_SUCCESS)
  {
    // every band's soft threshold, the residue (the last coarse plane has no halo: j->cur holds the band's own rows)
    // and the inverse transform in one pass
    vst_args ia;
    inverse_args(j->s, ia);
    launch_scope ls(devid, "dn_finish");
    dn_finish<<<pixel_grid(npix), 256, 0, st>>>(out, j->cur + (size_t)j->cur_top * j->w, npix, ia, sy);
    err = check_launch("dn_finish");
  }
  if(accs) dt_hip_release_mem_object(accs);
  denoiseprofile_band_abort(j);
  return err;
}

} // namespace ansel

extern "C" {

// chain: nullptr, or the pointwise run applied in the last kernel (wavelets only)
static int denoiseprofile_run(int devid, const dt_hip_piece_t *piece, const dt_hip_denoiseprofile_data_t *d, dt_hip_mem_t dev_in,
                              dt_hip_mem_t dev_out, const rgb_group_t *chain)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  if(!DT_HIP_DENOISEPROFILE_IS_WAVELETS(d->mode) && !DT_HIP_DENOISEPROFILE_IS_NLMEANS(d->mode))
  {
    set_last_error("denoiseprofile: mode %d is not implemented on device (wavelets and non-local means are)", d->mode);
    return DT_HIP_INVALID_ARG;
  }
  if(piece->channels != 4 || !(piece->roi_in.scale > 0.0)) return DT_HIP_INVALID_ARG;
  const int w = piece->roi_in.width, h = piece->roi_in.height;
  if(w <= 0 || h <= 0) return DT_HIP_SUCCESS;
  const size_t npix = (size_t)w * h, plane = npix * sizeof(float4);
  if(DT_HIP_DENOISEPROFILE_IS_NLMEANS(d->mode))
    return chain ? DT_HIP_INVALID_ARG : denoise_nlmeans(devid, piece, d, nullptr, h, dev_in, dev_out);
  // the run's arguments first: an unsupported combination must fail before anything is launched
  chain_args ca;
  int cm_kind = CM_NONE, fm = FM_NONE;
  if(chain)
  {
    if(chain->pre_lab || chain->to_u16 || chain->width != w || chain->height != h) return DT_HIP_INVALID_ARG;
    if(rgb_group_fill_args(*chain, ca, cm_kind, fm) != DT_HIP_SUCCESS || fm != FM_NONE) return DT_HIP_INVALID_ARG;
  }
  dn_setup s;
  setup(piece, d, s, false);
  const int runnable = wavelets_runnable(piece, s);
  if(runnable < 0) return DT_HIP_INVALID_ARG;
  hipStream_t st = stream_of(devid);
  if(!runnable) return chain ? DT_HIP_INVALID_ARG : dt_hip_enqueue_copy_buffer_to_buffer(devid, dev_in, dev_out, 0, 0, plane);
  const int nseg = (w + 255) / 256;
  const size_t n_partial = (size_t)h * nseg;
  // the tables of partial sums of ALL bands: the thresholds are needed by dn_finish only, so they are reduced together
  // behind the last decomposition (seven pairs of latency-bound launches in a row cost 0.39 ms per frame)
  double *partial = (double *)dt_hip_alloc_device_buffer(devid, (size_t)BANDS * n_partial * 4 * sizeof(double));
  float *thrs = (float *)dt_hip_alloc_device_buffer(devid, BANDS * 4 * sizeof(float));
  double *accs = (double *)dt_hip_alloc_device_buffer(devid, (size_t)BANDS * 4 * 1024 * sizeof(double));
  int err = (partial && thrs && accs) ? DT_HIP_SUCCESS : DT_HIP_SYSMEM_ALLOCATION;
  // Round 6: one COARSE plane per band, all alive until the single synthesis pass at the end, and ONE detail plane -- band 0's, whose
  // input (the transformed frame) is never stored.  The detail of every other band is the difference of two coarse planes and is formed
  // where it is read (synth_args::coarse0): bands 1 .. write 16 B/px instead of 32, and the frame holds 1 + bands planes instead of 2 + bands
  synth_args sy;
  memset(&sy, 0, sizeof(sy));
  sy.nbands = s.max_scale;
  sy.thrs = thrs;
  float4 *coarse[BANDS] = { nullptr };
  float4 *det0 = (float4 *)dt_hip_alloc_device_buffer(devid, plane);
  if(!det0) err = DT_HIP_SYSMEM_ALLOCATION;
  for(int k = 0; k < s.max_scale && err == DT_HIP_SUCCESS; k++)
  {
    coarse[k] = (float4 *)dt_hip_alloc_device_buffer(devid, plane);
    if(!coarse[k]) err = DT_HIP_SYSMEM_ALLOCATION;
    sy.detail[k] = k == 0 ? det0 : coarse[k];
  }
  sy.coarse0 = coarse[0];
  float4 *out = (float4 *)dev_out;
  // the variance-stabilising transform is applied by the first decomposition as it fetches the module's input (dn_decompose_strip<true>)
  vst_args fa;
  forward_args(s, fa);
  const float4 *b1 = (const float4 *)dev_in;
  // the Y0U0V0 transform sets the fourth channel to 0.0f: the decompositions leave it out for as long as it stays +0
  unsigned *alpha_flag = nullptr;
  if(err == DT_HIP_SUCCESS && s.vst == 2)
  {
    alpha_flag = (unsigned *)dt_hip_alloc_device_buffer(devid, sizeof(unsigned));
    if(!alpha_flag) err = DT_HIP_SYSMEM_ALLOCATION;
    else if(hipMemsetAsync(alpha_flag, 0, sizeof(unsigned), st) != hipSuccess) err = DT_HIP_DEFAULT_ERROR;
  }
  // Y0U0V0 (round 6): TWO sequences over the same planes.  First every scale as an ALPHA0 launch on three-float planes (12 B/px read
  // and written instead of 16, and 108 instead of 144 read by the synthesis), whatever the flag says; a launch that writes a coarse
  // alpha other than +0 -- a pixel that is not finite made its weights sum to NaN -- raises the flag.  Then every scale again as the
  // four-channel launch on float4 planes in the same memory: each leaves at once unless the flag is up (8 launches of ~5 us on a
  // frame of finite pixels, where the per-scale fallback launches of the rounds before were 6), and redoes the module if it is.
  // The synthesis is launched for either kind of plane, gated the same way.  Other transforms: one sequence, four channels.
  const int sequences = alpha_flag ? 2 : 1;
  for(int seq = 0; seq < sequences && err == DT_HIP_SUCCESS; seq++)
  {
    b1 = (const float4 *)dev_in;
    for(int scale = 0; scale < s.max_scale && err == DT_HIP_SUCCESS; scale++)
    {
      const int mult = 1 << scale;
      const float varf = sqrtf(2.0f + 2.0f * 4.0f * 4.0f + 6.0f * 6.0f) / 16.0f;
      const float sigma_band = powf(varf, scale) * 1.0f;
      {
        launch_scope ls(devid, "dn_decompose");
        launch_decompose(st, b1, coarse[scale], scale == 0 ? det0 : nullptr, partial + (size_t)scale * n_partial * 4, w, h, mult,
                         1.0f / (sigma_band * sigma_band), nseg, 0, h, scale == 0 ? &fa : nullptr, alpha_flag, alpha_flag ? seq + 1 : 0);
      }
      err = check_launch("denoiseprofile band");
      b1 = coarse[scale]; // the coarse plane just written is the next scale's input, and after the last one the residue
    }
  }
  if(err == DT_HIP_SUCCESS)
  {
    thr_args_all all;
    memset(&all, 0, sizeof(all));
    for(int scale = 0; scale < s.max_scale; scale++)
    {
      all.band[scale].n_partial = n_partial;
      threshold_args(d, scale, s.max_scale, npix, all.band[scale]);
    }
    launch_scope ls(devid, "dn_band_threshold");
    dn_band_sums<<<dim3(THR_GROUPS, s.max_scale), 64, 0, st>>>(partial, n_partial, accs);
    dn_band_threshold<<<s.max_scale, 1024, 0, st>>>(accs, all, thrs);
    err = check_launch("denoiseprofile thresholds");
  }
  if(err == DT_HIP_SUCCESS)
  {
    vst_args ia;
    inverse_args(s, ia);
    const unsigned grid = pixel_grid(npix);
    for(int seq = 0; seq < sequences; seq++)
    {
      launch_scope ls(devid, chain ? "dn_finish_chain" : "dn_finish"); // (one per launch: the profiling tables average per launch)
      if(alpha_flag)
      {
        // the synthesis over the first sequence's three-float planes if the flag stayed down, over the second's float4 planes if not
        sy.planes3 = seq == 0;
        sy.gate = alpha_flag;
        sy.gate_sense = seq + 1;
      }
      if(!chain) dn_finish<<<grid, 256, 0, st>>>(out, b1, npix, ia, sy);
      else
        switch(cm_kind)
        {
          case CM_NONE: dn_finish_chain<CM_NONE><<<grid, 256, 0, st>>>(out, b1, npix, ia, sy, ca); break;
          case 0: dn_finish_chain<0><<<grid, 256, 0, st>>>(out, b1, npix, ia, sy, ca); break;
          case 1: dn_finish_chain<1><<<grid, 256, 0, st>>>(out, b1, npix, ia, sy, ca); break;
          case 2: dn_finish_chain<2><<<grid, 256, 0, st>>>(out, b1, npix, ia, sy, ca); break;
          case 3: dn_finish_chain<3><<<grid, 256, 0, st>>>(out, b1, npix, ia, sy, ca); break;
          default: dn_finish_chain<4><<<grid, 256, 0, st>>>(out, b1, npix, ia, sy, ca); break;
        }
    }
    err = check_launch("dn_finish");
  }
  if(det0) dt_hip_release_mem_object(det0);
  for(int k = 0; k < BANDS; k++)
    if(coarse[k]) dt_hip_release_mem_object(coarse[k]);
  if(partial) dt_hip_release_mem_object(partial);
  if(thrs) dt_hip_release_mem_object(thrs);
  if(accs) dt_hip_release_mem_object(accs);
  if(alpha_flag) dt_hip_release_mem_object(alpha_flag);
  return err;
}

int dt_hip_iop_denoiseprofile_process(int devid, const dt_hip_piece_t *piece, const dt_hip_denoiseprofile_data_t *d,
                                      dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  return denoiseprofile_run(devid, piece, d, dev_in, dev_out, nullptr);
}
} // extern "C"

namespace ansel
{
int denoiseprofile_process_chain(int devid, const dt_hip_piece_t *piece, const dt_hip_denoiseprofile_data_t *d,
                                 dt_hip_mem_t dev_in, dt_hip_mem_t dev_out, const rgb_group_t *chain)
{
  if(!chain) return DT_HIP_INVALID_ARG;
  return denoiseprofile_run(devid, piece, d, dev_in, dev_out, chain);
}
} // namespace ansel

extern "C" {


// tiling_callback(), src/iop/denoiseprofile.c:796-848.  factor / overlap as the reference states them for the host;
// factor_cl = the planes this implementation holds on the device: wavelets in + out + band 0's detail + one coarse
// plane per band (the partial sums are W / 64 of a plane), non-local means in + out + the preconditioned copy (the tables live in LDS)
void dt_hip_iop_denoiseprofile_tiling(const dt_hip_piece_t *piece, const dt_hip_denoiseprofile_data_t *d,
                                      dt_hip_tiling_t *tiling)
{
  memset(tiling, 0, sizeof(*tiling));
  tiling->maxbuf = tiling->maxbuf_cl = 1.0f;
  tiling->xalign = tiling->yalign = 1;
  if(DT_HIP_DENOISEPROFILE_IS_NLMEANS(d->mode))
  {
    const float scale = fminf(fminf((float)piece->roi_in.scale, 2.0f), 1.0f);
    const int P = (int)ceilf(d->radius * scale), K = (int)ceilf(d->nbhood * scale);
    const int K_scattered = (int)ceilf(d->scattering * (K * K * K + 7.0 * K * sqrt((double)K)) / 6.0) + K;
    tiling->factor = 2.0f + 0.25f;
    tiling->factor_cl = 3.0f;
    tiling->overlap = (unsigned)(P + K_scattered);
  }
  else
  {
    dn_setup s;
    setup(piece, d, s, false);
    tiling->factor = 5.0f;
    tiling->factor_cl = 3.0f + (float)s.max_scale + 1.0f / 64.0f; // in, out, band 0's detail, one coarse plane per band
    tiling->overlap = 1u << s.max_scale;
  }
}

} // extern "C"
