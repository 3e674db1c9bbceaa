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


#ifdef ANSEL_HIP_MEASURING
// MEASURING BUILD ONLY -- equal in time to the two single-scale launches it replaces (profiles/r04_negative_results.txt,
// item 6): either form writes 2.5 - 2.6 TB/s, as a copy does, and the pair saves reads only.
// TWO consecutive scales in one pass (round 4): low1 = blur_m(in), low2 = blur_2m(low1).  The analysis moves its bytes at
// the device's copy rate (4.8 - 5.1 TB/s), and run scale by scale it reads every low-pass plane back that it has just written:
// 16 + 16 bytes per pixel and scale.  Here a workgroup owns 256 adjacent columns for up to `strip` rows of one dilation class
// of the COARSER scale (rows c, c + 2m, c + 4m, ...), one lane per column of the 256 + 12m the first scale's vertical pass
// needs.  Per output row of low2:
//   X  the lane's five samples of `in` (rows 2m apart from the last step's: three stay, two new ones were fetched a step
//      ahead) -> vertical taps of low1's next row into LDS (VA); the lane's five samples of low1 (the five rows low2's row
//      taps: 2m apart, the same class -- produced by this very lane, in registers) -> vertical taps of low2's row (VC)
//   -- one barrier (VA and VC alternate between two buffers) --
//   Y  horizontal taps of VA: low1's next row at the lane's column (256 + 8m of them), kept and, where the strip owns the
//      row, written; horizontal taps of VC: low2's row, written.
// in -> (low1, low2): 16 bytes read and 32 written per pixel instead of 32 + 32.  Same taps, same order, same clamping
// (bspline.h:118-149: rows and columns clamp to the frame's, whatever the dilation), same binary32 values.  A row of low1
// whose (virtual) index lies outside the frame is the row at the clamped index: formed from scratch (five direct fetches) and
// not written -- the strip that owns row 0 / height - 1 writes it.
__device__ __forceinline__ float4 bs_row_taps(const float4 *__restrict__ V, const int col, const int step, const int origin,
                                              const int width)
{
  float4 t[5];
#pragma unroll
  for(int j = 0; j < 5; j++) t[j] = V[clampi(col + (j - 2) * step, 0, width - 1) - origin];
  return tap5(t[0], t[1], t[2], t[3], t[4]);
}

template <int M, int AHEAD>
__global__ __launch_bounds__(320) void bspline_decompose2_strip(const float4 *__restrict__ in, float4 *__restrict__ low1,
                                                                float4 *__restrict__ low2, const int width, const int height,
                                                                const int strip, const int strips_per_class)
{
  constexpr int M2 = 2 * M, WA = 256 + 12 * M, WB = 256 + 8 * M;
  static_assert(WA <= 320, "a lane a column");
  __shared__ float4 VA[2][WA], VC[2][WB];
  const int c0 = (int)blockIdx.x * 256;
  const int cls = blockIdx.y / strips_per_class, k0s = (blockIdx.y - cls * strips_per_class) * strip;
  const int n_cls = (height - cls + M2 - 1) / M2; // rows of this class of the coarser scale
  if(k0s >= n_cls) return;
  const int nrows = (strip < n_cls - k0s) ? strip : n_cls - k0s;
  const int r_first = cls + k0s * M2;
  const int tid = threadIdx.x;
  const int ca = c0 - 6 * M + tid, cb = c0 - 4 * M + tid;
  const bool on_a = tid < WA && ca >= 0 && ca < width;                 // a column of the first vertical pass
  const bool on_b = tid < WB && cb >= 0 && cb < width;                 // ... of low1's row / the second vertical pass
  const bool on_c = on_b && tid >= 4 * M && tid < 4 * M + 256;          // ... of the 256 this workgroup writes
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 a = z, b = z, c = z, d = z, e = z; // `in` at rows sigma - 2m .. sigma + 2m (clamped)
  // ... and the two new rows (sigma + m, sigma + 2m) of the next AHEAD steps, in flight: a step is a few hundred cycles, a
  // fetch from memory several thousand -- one step ahead and four workgroups a CU left the analysis waiting (3.9 TB/s)
  float4 nd[AHEAD], ne[AHEAD];
  float4 r0 = z, r1 = z, r2 = z, r3 = z, r4 = z;            // low1 at the five rows low2's current row taps
  bool rolling = false;
#define BS_AT(r) in[(size_t)clampi((r), 0, height - 1) * width + ca]
  const int ca_safe = clampi(ca, 0, width - 1);
#define BS_SAFE(r) in[(size_t)clampi((r), 0, height - 1) * width + ca_safe]
#pragma unroll
  for(int j = 0; j < AHEAD; j++)
  {
    const int sj = r_first + (j - 2) * M2;
    nd[j] = BS_SAFE(sj + M);
    ne[j] = BS_SAFE(sj + 2 * M);
  }
  // entry q = low1's row at the clamped index of sigma_q = r_first + (q - 2) 2m; output row k taps entries k .. k + 4.
  // One step; J is the queue slot of its pair of rows, refilled at once for the step AHEAD later (the loop below is unrolled
  // AHEAD times so that the slots are fixed registers: shifting a queue would be a use of every fetch in flight)
#define BS_STEP(kk, J) \
  { \
    const int k = (kk); \
    const int q = k + 5, sig = r_first + (q - 2) * M2, buf = q & 1; \
    const bool produce = q <= nrows + 3; \
    if(produce) \
    { \
      if(sig >= 0 && sig < height) \
      { \
        if(rolling) \
        { \
          a = c; \
          b = d; \
          c = e; \
        } \
        else \
        { \
          if(on_a) \
          { \
            a = BS_AT(sig - 2 * M); \
            b = BS_AT(sig - M); \
            c = BS_AT(sig); \
          } \
          rolling = true; \
        } \
        d = nd[J]; \
        e = ne[J]; \
        if(on_a) VA[buf][tid] = tap5(a, b, c, d, e); \
      } \
      else \
      { \
        /* outside the frame: the row at the clamped index, from scratch (the window of the rows inside starts afresh) */ \
        rolling = false; \
        const int tau = clampi(sig, 0, height - 1); \
        if(on_a) VA[buf][tid] = tap5(BS_AT(tau - 2 * M), BS_AT(tau - M), BS_AT(tau), BS_AT(tau + M), BS_AT(tau + 2 * M)); \
      } \
    } \
    /* every lane, every step (rows and columns clamp): a condition around a fetch makes the compiler wait for it at once */ \
    nd[J] = BS_SAFE(sig + AHEAD * M2 + M); \
    ne[J] = BS_SAFE(sig + AHEAD * M2 + 2 * M); \
    if(k >= 0 && on_b) VC[buf][tid] = tap5(r0, r1, r2, r3, r4); \
    __syncthreads(); \
    if(k >= 0 && on_c) low2[(size_t)(r_first + k * M2) * width + cb] = bs_row_taps(VC[buf], cb, M2, c0 - 4 * M, width); \
    if(produce) \
    { \
      r0 = r1; \
      r1 = r2; \
      r2 = r3; \
      r3 = r4; \
      if(on_b) \
      { \
        r4 = bs_row_taps(VA[buf], cb, M, c0 - 6 * M, width); \
        if(on_c && q >= 2 && q - 2 < nrows) low1[(size_t)sig * width + cb] = r4; \
      } \
    } \
  }
  for(int k0 = -5; k0 < nrows; k0 += AHEAD)
  {
    BS_STEP(k0, 0)
    if constexpr(AHEAD > 1)
      if(k0 + 1 < nrows) BS_STEP(k0 + 1, 1)
    if constexpr(AHEAD > 2)
      if(k0 + 2 < nrows) BS_STEP(k0 + 2, 2)
    if constexpr(AHEAD > 3)
      if(k0 + 3 < nrows) BS_STEP(k0 + 3, 3)
  }
#undef BS_STEP
#undef BS_AT
#undef BS_SAFE
}
#endif // ANSEL_HIP_MEASURING

} // namespace

namespace ansel
{
#ifdef ANSEL_HIP_MEASURING
// two scales at once (dilations mult and 2 mult; mult 1 or 4): in -> low1, low2.  DT_HIP_INVALID_ARG for another dilation
int bspline_launch_decompose2(int devid, hipStream_t s, const float4 *in, float4 *low1, float4 *low2, int w, int h, int mult)
{
  (void)devid;
  if(mult != 1 && mult != 4) return DT_HIP_INVALID_ARG;
  const int m2 = 2 * mult;
  const int classes = h < m2 ? h : m2, per_class = (h + m2 - 1) / m2;
  const int gx = (w + 255) / 256;
  // four rows of low1 a strip are formed twice (its neighbours' too): long strips
  int strip = 64;
  while(strip > 4 && (size_t)gx * classes * ((per_class + strip - 1) / strip) < 2048) strip /= 2;
  const int spc = (per_class + strip - 1) / strip;
  const dim3 grid(gx, classes * spc);
  launch_scope ls(devid, "diffuse_decompose");
  // how far ahead the fetches run
  static const char *const ahead_env = measuring_env("ANSEL_HIP_BSPLINE_AHEAD");
  const int ahead = ahead_env ? atoi(ahead_env) : 1;
#define BS_PAIR(MM) \
  switch(ahead) \
  { \
    case 2: bspline_decompose2_strip<MM, 2><<<grid, 320, 0, s>>>(in, low1, low2, w, h, strip, spc); break; \
    case 4: bspline_decompose2_strip<MM, 4><<<grid, 320, 0, s>>>(in, low1, low2, w, h, strip, spc); break; \
    case 3: bspline_decompose2_strip<MM, 3><<<grid, 320, 0, s>>>(in, low1, low2, w, h, strip, spc); break; \
    default: bspline_decompose2_strip<MM, 1><<<grid, 320, 0, s>>>(in, low1, low2, w, h, strip, spc); break; \
  }
  if(mult == 1)
  {
    BS_PAIR(1);
  }
  else
  {
    BS_PAIR(4);
  }
#undef BS_PAIR
  return check_launch("diffuse_decompose");
}
#endif // ANSEL_HIP_MEASURING

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
