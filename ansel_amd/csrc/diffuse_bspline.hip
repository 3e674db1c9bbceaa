// diffuse_bspline.hip -- the a-trous B-spline analysis of diffuse or sharpen (decompose_2D_Bspline(),
// src/pixel/bspline.h:351-377) on gfx950; see diffuse.hip for the module.  A translation unit of its own because it
// is the one kernel of the library that is FASTER with the SLP vectoriser on (its float4 taps stay whole; 1.12 against
// 1.19 ms per plane at 100 MP), while every arithmetic-bound kernel loses to v_pk_*_f32 (ansel_amd/build.py).
#include "hip_common.h"

#include <cstdlib>

using namespace ansel;

namespace
{

// the MAX(a, b) macro of the reference: a > b ? a : b (a NaN in b passes through)
__device__ __forceinline__ float max_first(const float a, const float b) { return a > b ? a : b; }
__device__ __forceinline__ int clampi(const int v, const int lo, const int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// sparse_scalar_product(), bspline.h:86-118: left-to-right weighted sum, then MAX(0, .)
__device__ __forceinline__ float tap5(const float a, const float b, const float c, const float d, const float e)
{
  const float s = 0.0625f * a + 0.25f * b + 0.375f * c + 0.25f * d + 0.0625f * e;
  return max_first(0.0f, s);
}
__device__ __forceinline__ float4 tap5(const float4 a, const float4 b, const float4 c, const float4 d, const float4 e)
{
  return make_float4(tap5(a.x, b.x, c.x, d.x, e.x), tap5(a.y, b.y, c.y, d.y, e.y), tap5(a.z, b.z, c.z, d.z, e.z),
                     tap5(a.w, b.w, c.w, d.w, e.w));
}

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

#ifdef ANSEL_HIP_MEASURING // the per-row kernel: A/B timing only (ANSEL_HIP_BSPLINE_PER_ROW); the strips below are the product's
__device__ __forceinline__ float4 vertical5(const float4 *__restrict__ in, const int width, const int height,
                                            const int row, const int col, const int mult, float4 *centre)
{
  const float4 a = in[(size_t)clampi(row - 2 * mult, 0, height - 1) * width + col];
  const float4 b = in[(size_t)clampi(row - mult, 0, height - 1) * width + col];
  const float4 c = in[(size_t)row * width + col];
  const float4 d = in[(size_t)clampi(row + mult, 0, height - 1) * width + col];
  const float4 e = in[(size_t)clampi(row + 2 * mult, 0, height - 1) * width + col];
  if(centre) *centre = c;
  return tap5(a, b, c, d, e);
}

// R adjacent columns x T steps of the dilation; R * T == 256
template <int R, int T>
__global__ __launch_bounds__(256) void bspline_decompose(const float4 *__restrict__ in, float4 *__restrict__ hf,
                                                         float4 *__restrict__ lf, const int width, const int height,
                                                         const int mult, const int groups, const int gx)
{
  __shared__ float4 vert[(T + 4) * R + 2];
  const int bx = xcd_col(); // hip_common.h: the column block, pinned to an XCD for 64 rows of the walk
  if(bx >= gx) return;
  const int row = walk_row(blockIdx.y, height, mult);
  if(row < 0) return;
  // bx = step tile * groups + residue group
  const int group = bx % groups, tile = bx / groups;
  const int r0 = group * R, k0 = tile * T;
  const int tid = threadIdx.x;
  const int r = tid % R, k = tid / R;
  const int col = r0 + r + (k0 + k) * mult;
  float4 centre = make_float4(0.f, 0.f, 0.f, 0.f);
  // own sample -> slot (k + 2) * R + r
  if(col < width) vert[(k + 2) * R + r] = vertical5(in, width, height, row, col, mult, &centre);
  // halo: steps k0-2, k0-1, k0+T, k0+T+1
  if(tid < 4 * R)
  {
    const int hr = tid % R, hs = tid / R; // hs 0..3
    const int hk = hs < 2 ? hs - 2 : T + hs - 2;
    const int hcol = r0 + hr + (k0 + hk) * mult;
    if(hcol >= 0 && hcol < width) vert[(hk + 2) * R + hr] = vertical5(in, width, height, row, hcol, mult, nullptr);
  }
  else if(tid == 4 * R)
    vert[(T + 4) * R] = vertical5(in, width, height, row, 0, mult, nullptr);
  else if(tid == 4 * R + 1)
    vert[(T + 4) * R + 1] = vertical5(in, width, height, row, width - 1, mult, nullptr);
  __syncthreads();
  if(col >= width) return;
  float4 t[5];
#pragma unroll
  for(int s = -2; s <= 2; s++)
  {
    const int c = col + s * mult;
    t[s + 2] = c < 0 ? vert[(T + 4) * R] : (c > width - 1 ? vert[(T + 4) * R + 1] : vert[(k + s + 2) * R + r]);
  }
  const float4 low = tap5(t[0], t[1], t[2], t[3], t[4]);
  const size_t o = (size_t)row * width + col;
  lf[o] = low;
  nt_store(hf + o, make_float4(centre.x - low.x, centre.y - low.y, centre.z - low.z, centre.w - low.w));
}
#endif // ANSEL_HIP_MEASURING

// The same analysis on STRIPS: a workgroup keeps its R x T columns for up to `strip` rows of one dilation class (rows c,
// c + m, c + 2 m, ...), whose five vertical taps overlap in four rows: every lane rolls the five samples of its column
// (and of the halo / border column it also serves) through registers and fetches ONE new row per output row -- issued
// before the current row is filtered -- instead of five through L2.  The vertically blurred row lives in two LDS
// buffers used alternately, one barrier per row.  Same taps, same order, same binary32 values.
template <int R, int T>
__global__ __launch_bounds__(256) void bspline_decompose_strip(const float4 *__restrict__ in, float4 *__restrict__ hf,
                                                               float4 *__restrict__ lf, const int width, const int height,
                                                               const int mult, const int groups, const int strip,
                                                               const int strips_per_class)
{
  __shared__ float4 vert[2][(T + 4) * R + 2];
  const int bx = blockIdx.x;
  const int cls = blockIdx.y / strips_per_class, k0s = (blockIdx.y - cls * strips_per_class) * strip;
  const int n_cls = (height - cls + mult - 1) / mult; // rows of this class
  if(k0s >= n_cls) return;
  const int nrows = (strip < n_cls - k0s) ? strip : n_cls - k0s;
  const int r_first = cls + k0s * mult;
  const int group = bx % groups, tile = bx / groups;
  const int r0 = group * R, k0 = tile * T;
  const int tid = threadIdx.x;
  const int r = tid % R, k = tid / R;
  const int col = r0 + r + (k0 + k) * mult;
  const bool own = col < width;
  // the second column this lane blurs vertically: a halo step (k0-2, k0-1, k0+T, k0+T+1) or a border column
  int col2 = -1, slot2 = 0;
  if(tid < 4 * R)
  {
    const int hr = tid % R, hs = tid / R;
    const int hk = hs < 2 ? hs - 2 : T + hs - 2;
    const int hcol = r0 + hr + (k0 + hk) * mult;
    if(hcol >= 0 && hcol < width)
    {
      col2 = hcol;
      slot2 = (hk + 2) * R + hr;
    }
  }
  else if(tid == 4 * R)
  {
    col2 = 0;
    slot2 = (T + 4) * R;
  }
  else if(tid == 4 * R + 1)
  {
    col2 = width - 1;
    slot2 = (T + 4) * R + 1;
  }
  const bool second = col2 >= 0;
  // tap q of the strip = frame row r_first + q mult, clamped (bspline.h:143-149)
#define BS_ROW(q) ((size_t)clampi(r_first + (q) * mult, 0, height - 1) * width)
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 a = z, b = z, c = z, d = z, e = z, a2 = z, b2 = z, c2 = z, d2 = z, e2 = z;
  if(own)
  {
    a = in[BS_ROW(-2) + col];
    b = in[BS_ROW(-1) + col];
    c = in[BS_ROW(0) + col];
    d = in[BS_ROW(1) + col];
    e = in[BS_ROW(2) + col];
  }
  if(second)
  {
    a2 = in[BS_ROW(-2) + col2];
    b2 = in[BS_ROW(-1) + col2];
    c2 = in[BS_ROW(0) + col2];
    d2 = in[BS_ROW(1) + col2];
    e2 = in[BS_ROW(2) + col2];
  }
  for(int kk = 0; kk < nrows; kk++)
  {
    const int row = r_first + kk * mult;
    const bool more = kk + 1 < nrows;
    float4 n1 = z, n2 = z;
    if(more && own) n1 = in[BS_ROW(kk + 3) + col];
    if(more && second) n2 = in[BS_ROW(kk + 3) + col2];
    float4 *const V = vert[kk & 1];
    if(own) V[(k + 2) * R + r] = tap5(a, b, c, d, e);
    if(second) V[slot2] = tap5(a2, b2, c2, d2, e2);
    __syncthreads();
    if(own)
    {
      float4 t[5];
#pragma unroll
      for(int s = -2; s <= 2; s++)
      {
        const int cc = col + s * mult;
        t[s + 2] = cc < 0 ? V[(T + 4) * R] : (cc > width - 1 ? V[(T + 4) * R + 1] : V[(k + s + 2) * R + r]);
      }
      const float4 low = tap5(t[0], t[1], t[2], t[3], t[4]);
      const size_t o = (size_t)row * width + col;
      lf[o] = low;
      // hf == nullptr: the caller keeps every low-pass plane and its PDE kernel forms centre - low itself (diffuse.hip)
      if(hf) nt_store(hf + o, make_float4(c.x - low.x, c.y - low.y, c.z - low.z, c.w - low.w));
    }
    a = b;
    b = c;
    c = d;
    d = e;
    e = n1;
    a2 = b2;
    b2 = c2;
    c2 = d2;
    d2 = e2;
    e2 = n2;
  }
#undef BS_ROW
}

} // namespace

namespace ansel
{
int bspline_launch_decompose(int devid, hipStream_t s, const float4 *in, float4 *hf, float4 *lf, int w, int h, int mult)
{
  const int steps = (w + mult - 1) / mult; // steps of the dilation across a row
  launch_scope ls(devid, "diffuse_decompose");
#ifdef ANSEL_HIP_MEASURING
  static const bool per_row = getenv("ANSEL_HIP_BSPLINE_PER_ROW") != nullptr; // the per-row kernels, for A/B timing
  if(per_row)
  {
    const int rows = (h <= mult) ? h : ((h + mult - 1) / mult) * mult;
    if(mult == 1)
      bspline_decompose<1, 256><<<dim3(xcd_pad((steps + 255) / 256), rows), 256, 0, s>>>(in, hf, lf, w, h, mult, 1, (steps + 255) / 256);
    else if(mult == 2)
      bspline_decompose<2, 128><<<dim3(xcd_pad((steps + 127) / 128), rows), 256, 0, s>>>(in, hf, lf, w, h, mult, 1, (steps + 127) / 128);
    else if(mult == 4)
      bspline_decompose<4, 64><<<dim3(xcd_pad((steps + 63) / 64), rows), 256, 0, s>>>(in, hf, lf, w, h, mult, 1, (steps + 63) / 64);
    else
    {
      const int groups = mult / 8;
      bspline_decompose<8, 32><<<dim3(xcd_pad(((steps + 31) / 32) * groups), rows), 256, 0, s>>>(in, hf, lf, w, h, mult, groups,
                                                                                            ((steps + 31) / 32) * groups);
    }
    return check_launch("diffuse_decompose");
  }
#endif
  const int classes = h < mult ? h : mult, per_class = (h + mult - 1) / mult;
  const int gx = mult == 1 ? (steps + 255) / 256 : (mult == 2 ? (steps + 127) / 128 : (mult == 4 ? (steps + 63) / 64 : ((steps + 31) / 32) * (mult / 8)));
  int strip = 32;
  while(strip > 4 && (size_t)gx * classes * ((per_class + strip - 1) / strip) < 2048) strip /= 2;
  const int spc = (per_class + strip - 1) / strip;
  const dim3 grid(gx, classes * spc);
  if(mult == 1) bspline_decompose_strip<1, 256><<<grid, 256, 0, s>>>(in, hf, lf, w, h, mult, 1, strip, spc);
  else if(mult == 2) bspline_decompose_strip<2, 128><<<grid, 256, 0, s>>>(in, hf, lf, w, h, mult, 1, strip, spc);
  else if(mult == 4) bspline_decompose_strip<4, 64><<<grid, 256, 0, s>>>(in, hf, lf, w, h, mult, 1, strip, spc);
  else bspline_decompose_strip<8, 32><<<grid, 256, 0, s>>>(in, hf, lf, w, h, mult, mult / 8, strip, spc);
  return check_launch("diffuse_decompose");
}
} // namespace ansel
