// bilat.hip -- local contrast, bilateral-grid mode, on gfx950.
//
// Reference: process(), src/iop/bilat.c:330-361 -> dt_bilateral_init / _splat / _blur / _slice,
// src/pixel/bilateral.c:157-393 (OpenCL twin: bilateral_cl.c, whose splat uses float atomics in
// LDS and is therefore order-free; the pin here is the CPU path).
//
// splat   The CPU scatters every pixel into 8 grid cells, per OpenMP slice, and merges the slices;
//         on one thread that is: every cell accumulates its contributions in pixel row-major order.
//         The device GATHERS in that order: one thread per grid node (x, y) walks the ~(2 sigma_s)^2
//         pixels that can reach the node, row by row, and adds into the node's z column kept in LDS.
//         No atomics, no dependence on launch geometry, bit-identical to the one-thread reference.
// blur    x, y: 1-4-6-4-1; z: derivative -- as out-of-place 5-tap stencils, one launch (bilat_blur_yz).
// slice   trilinear lookup per pixel, L += -detail * sigma_r * 0.04 * grid (px_bilat.h: also the first stage of a fused run behind the module).
#include "hip_common.h"
#include "pipe_fused.h"
#include "px_bilat.h"

#include <math.h>

using namespace ansel;

namespace
{

#define MAX_RES_S 3000 // DT_COMMON_BILATERAL_MAX_RES_S, bilateral.c:47
#define MAX_RES_R 50   // DT_COMMON_BILATERAL_MAX_RES_R
#define SPLAT_THREADS 64

typedef ansel::bilat_grid_t grid_t;

__host__ __device__ __forceinline__ float clampf(const float v, const float lo, const float hi) { return bilat_clampf(v, lo, hi); }
inline int clampi_h(const int v, const int lo, const int hi) { return v > lo ? (v < hi ? v : hi) : lo; }

// image_to_grid() / image_to_relgrid(), bilateral.c:127-155: cell index and fraction on one axis (px_bilat.h)
__device__ __forceinline__ int axis(const float v, const float sigma, const int size, float &frac) { return bilat_axis(v, sigma, size, frac); }

#ifdef ANSEL_HIP_MEASURING // the first gather: A/B timing only (ANSEL_HIP_BILAT_SPLAT_V1); bilat_splat2 below is the product's
// dt_bilateral_splat(), bilateral.c:183-256, gathered per grid node
// Row bands (a frame over several GPUs, pipe.cpp): a band splats its OWN rows [row_lo, row_hi) of the frame on top of
// what the bands above it have accumulated (`accumulate`: the node's z column starts from `buf` instead of zero) --
// rows ascend from band to band, so every cell still adds its contributions in pixel row-major order, the binary32
// partial sums travelling through `buf` unchanged.  `in` is the band's first row.  Whole frame: 0, height, 0.
// in: the L plane of the rows (bilat_lightness): a lane walks consecutive columns, so with 4-byte samples a 128-byte line
// serves 32 of its steps instead of the 8 it serves with the float4 pixels (5.4 -> 3.7 ms at 60 MP, 2.7 with the batched fetch below; what is left is the
// read-modify-write chain of a node's cells in LDS, one pixel after the other -- the order IS the result)
__global__ __launch_bounds__(SPLAT_THREADS) void bilat_splat(const float *__restrict__ in, float *__restrict__ buf,
                                                             const grid_t b, const int row_lo, const int row_hi,
                                                             const int accumulate)
{
  extern __shared__ float acc[]; // [size_z][SPLAT_THREADS]
  const int tid = threadIdx.x;
  const int node = blockIdx.x * SPLAT_THREADS + tid;
  const bool live = node < b.size_x * b.size_y;
  const int Y = live ? node / b.size_x : 0, X = live ? node - Y * b.size_x : 0;
  for(int z = 0; z < b.size_z; z++)
    acc[z * SPLAT_THREADS + tid] = (accumulate && live) ? buf[(size_t)(X + Y * b.size_x) * b.size_z + z] : 0.0f;
  if(live)
  {
    const float s2 = b.sigma_s * b.sigma_s;
    const int i0 = max(0, (int)floorf((X - 1) * b.sigma_s) - 2), i1 = min(b.width - 1, (int)ceilf((X + 1) * b.sigma_s) + 2);
    const int j0 = max(row_lo, (int)floorf((Y - 1) * b.sigma_s) - 2), j1 = min(row_hi - 1, (int)ceilf((Y + 1) * b.sigma_s) + 2);
    for(int j = j0; j <= j1; j++)
    {
      float yf;
      const int yi = axis((float)j, b.sigma_s, b.size_y, yf);
      if(yi != Y && yi != Y - 1) continue;
      const float wy = yi == Y ? (1.0f - yf) : yf;
      // sixteen samples are fetched before the first is used: the cell a sample lands in depends on its value, so one
      // fetch per step put a full memory round trip (~700 cycles with one wave per SIMD) in front of every addition
      // (3.7 -> 2.7 ms at 60 MP; the z column in registers with compare-and-select instead of LDS: 3.6 ms, not used)
      const float *const rowp = in + (size_t)(j - row_lo) * b.width;
      for(int ib = i0; ib <= i1; ib += 16)
      {
        float Lv[16];
#pragma unroll
        for(int u = 0; u < 16; u++) Lv[u] = rowp[min(ib + u, i1)];
#pragma unroll
        for(int u = 0; u < 16; u++)
        {
          const int i = ib + u;
          if(i > i1) continue;
          float xf, zf;
          const int xi = axis((float)i, b.sigma_s, b.size_x, xf);
          if(xi != X && xi != X - 1) continue;
          const float wx = xi == X ? (1.0f - xf) : xf;
          const float L = Lv[u];
          const int zi = axis(L, b.sigma_r, b.size_z, zf);
          const float contrib = wx * wy * 100.0f / s2; // (1-xf)*(1-yf)*100/s2 and its three siblings
          acc[zi * SPLAT_THREADS + tid] += (contrib * (1.0f - zf));
          acc[(zi + 1) * SPLAT_THREADS + tid] += (contrib * zf);
        }
      }
    }
    float *const cell = buf + (size_t)(X + Y * b.size_x) * b.size_z;
    for(int z = 0; z < b.size_z; z++) cell[z] = acc[z * SPLAT_THREADS + tid];
  }
}
#endif // ANSEL_HIP_MEASURING

// The second version of the gather.  What bound the first: 644 waves at 100 MP -- fewer than the chip has SIMDs -- each
// walking ~10^4 pixels alone on its SIMD at ~80 dependent instructions a pixel, three exact divisions among them (the
// pixel's column and lightness by the cell sizes, the contribution by sigma_s^2).  None of the three needs to be where
// the chain is:
//   * the lightness cell (zi, zf) of a pixel is the same for the four nodes it reaches: bilat_zcells computes it once per
//     pixel, in parallel, where the first version only copied L out of the pixel;
//   * the column weight wx depends on (node column, pixel column) only: every lane tabulates it once for the columns
//     of its footprint (LDS, [column][lane]) -- 0 for the columns at the footprint's rim that belong to the cell beyond,
//     whose contributions are then +0 and leave the sums (all >= +0) as they are, so the walk needs no branch;
//   * x / sigma_s^2 has a wave-uniform denominator: its reciprocal and the Newton step on it are computed once, and a
//     quotient is the remaining five operations of the correctly rounded sequence (div_by, below);
//   * a wave's lanes are nodes of ONE grid row, so the row loop, the row weight and the row pointer are scalar.
// ~20 instructions a pixel and the two read-add-write round trips of the cells, in pixel order as before.
struct inv_t
{
  float d, y1; // the denominator and its refined reciprocal
};
__device__ __forceinline__ inv_t inv_of(const float d)
{
  const float y = __builtin_amdgcn_rcpf(d); // v_rcp_f32: within 1 ulp
  inv_t r = { d, fmaf(fmaf(-d, y, 1.0f), y, y) };
  return r;
}
// n / d, correctly rounded -- i.e. the quotient `/` gives -- for n = 0 and for operands and quotients well inside the
// normal range (the caller's business): the sequence the compiler expands a division to, less its range scaling and its
// special-case fix-up, and with the denominator's part hoisted.  tools/div_by_check.c: no difference from n / d in 2 10^9
// random cases, with reciprocal seeds up to 2 ulp off.
__device__ __forceinline__ float div_by(const float n, const inv_t i)
{
  const float q = n * i.y1;
  const float q1 = fmaf(fmaf(-i.d, q, n), i.y1, q);
  return fmaf(fmaf(-i.d, q1, n), i.y1, q1);
}

// (zf, zi) per pixel: image_to_grid()'s third axis, bilateral.c:127-155
__global__ __launch_bounds__(256) void bilat_zcells(const float4 *__restrict__ in, float2 *__restrict__ zc, const size_t n,
                                                    const float sigma_r, const int size_z)
{
  for(size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x)
  {
    float zf;
    const int zi = axis(in[k].x, sigma_r, size_z, zf);
    zc[k] = make_float2(zf, __int_as_float(zi));
  }
}

#ifndef SPLAT_GROUP
#define SPLAT_GROUP 2
#endif
// blockIdx.y = the grid row Y, blockIdx.x = a block of 64 grid columns.  TABLE: the column weights fit in LDS
template <bool TABLE, bool FAST_DIV>
__global__ __launch_bounds__(SPLAT_THREADS) void bilat_splat2(const float2 *__restrict__ zc, float *__restrict__ buf,
                                                              const grid_t b, const int row_lo, const int row_hi,
                                                              const int accumulate)
{
  extern __shared__ float acc[]; // [size_z][SPLAT_THREADS], then the column weights [footprint column][SPLAT_THREADS]
  float *const wxt = acc + b.size_z * SPLAT_THREADS;
  const int tid = threadIdx.x;
  const int Y = blockIdx.y, X = blockIdx.x * SPLAT_THREADS + tid;
  const bool live = X < b.size_x;
  for(int z = 0; z < b.size_z; z++)
    acc[z * SPLAT_THREADS + tid] = (accumulate && live) ? buf[(size_t)(X + Y * b.size_x) * b.size_z + z] : 0.0f;
  const float s2 = b.sigma_s * b.sigma_s;
  const inv_t is2 = inv_of(s2);
  // a dead lane walks the last node's columns and adds +0
  const int Xn = live ? X : b.size_x - 1;
  const int i0 = max(0, (int)floorf((Xn - 1) * b.sigma_s) - 2), i1 = min(b.width - 1, (int)ceilf((Xn + 1) * b.sigma_s) + 2);
  const int j0 = max(row_lo, (int)floorf((Y - 1) * b.sigma_s) - 2), j1 = min(row_hi - 1, (int)ceilf((Y + 1) * b.sigma_s) + 2);
  auto column_weight = [&](const int i) {
    float xf;
    const int xi = axis((float)i, b.sigma_s, b.size_x, xf);
    return !live ? 0.0f : (xi == X ? 1.0f - xf : (xi == X - 1 ? xf : 0.0f));
  };
  if(TABLE)
    for(int i = i0; i <= i1; i++) wxt[(i - i0) * SPLAT_THREADS + tid] = column_weight(i);
  // the widest footprint of the wave: lanes with a narrower one add +0 beyond theirs
  int span = i1 - i0 + 1;
#pragma unroll
  for(int off = 32; off >= 1; off >>= 1) span = max(span, __shfl_xor(span, off, 64));
  span = __builtin_amdgcn_readfirstlane(span); // (the same in every lane: the walk's control flow is scalar)
  // the rows of the footprint that reach this grid row (uniform), each with its weight
  auto row_weight = [&](const int j, float &wy) {
    float yf;
    const int yi = axis((float)j, b.sigma_s, b.size_y, yf);
    wy = yi == Y ? (1.0f - yf) : yf;
    return yi == Y || yi == Y - 1;
  };
  auto next_row = [&](int j, float &wy) {
    while(j <= j1 && !row_weight(j, wy)) j++;
    return j;
  };
  // The sixteen contributions of a block FIRST, their column weights fetched in one round trip: the weights live in the same LDS
  // allocation as the cells, so a weight read between two cells' read-add-write sequences has to stay between them (the compiler
  // cannot know that no cell is written there) -- which made every pixel TWO dependent LDS round trips (the weight, then the cells:
  // ~290 cycles a pixel), with the quotient's five dependent operations between them.
  auto contributions = [&](float(&contrib)[16], const float wy, const int ub) {
#pragma unroll
    for(int u = 0; u < 16; u++)
    {
      const int i = i0 + ub + u;
      float wx = TABLE ? wxt[min(ub + u, i1 - i0) * SPLAT_THREADS + tid] : column_weight(min(i, i1));
      if(i > i1) wx = 0.0f;
      const float num = wx * wy * 100.0f;
      contrib[u] = FAST_DIV ? div_by(num, is2) : num / s2; // (1-xf)*(1-yf)*100/s2 and its three siblings
    }
  };
  auto fetch = [&](float2(&cell)[16], const int j, const int ub) {
    const float2 *const rowp = zc + (size_t)(j - row_lo) * b.width;
#pragma unroll
    for(int u = 0; u < 16; u++)
    {
      // a relaxed atomic load of wavefront scope IS the plain global_load_dwordx2 -- and stays where it is written: the compiler
      // sinks an ordinary load of read-only memory towards its first use (the NEXT round's additions), across any barrier it is given
      const unsigned long long bits = __hip_atomic_load((const unsigned long long *)(rowp + min(i0 + ub + u, i1)), __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_WAVEFRONT);
      cell[u] = make_float2(__uint_as_float((unsigned)bits), __uint_as_float((unsigned)(bits >> 32)));
    }
  };
  // SPLAT_GROUP pixels per LDS round trip: the cells of all of them are read first; a pixel whose cell one of the pixels before it
  // in the group has just updated takes that pixel's sum instead of the value read (the latest such pixel's: it holds the earlier
  // ones' contributions already), and the sums are written back in pixel order -- every cell receives the same additions in the
  // same order as with one read-add-write per pixel.  Two: bilat --size 100MP 1.69 -> 1.60 ms; four 1.83, eight 2.31 (the selects
  // cost a wave alone on its SIMD more issue slots than the round trips they save: profiles/r06_negative_results.txt, 13).
  auto add = [&](const float2(&cell)[16], const float(&contrib)[16]) {
#pragma unroll
    for(int u0 = 0; u0 < 16; u0 += SPLAT_GROUP)
    {
      int zi[SPLAT_GROUP];
      float lo[SPLAT_GROUP], hi[SPLAT_GROUP], sum_lo[SPLAT_GROUP], sum_hi[SPLAT_GROUP];
#pragma unroll
      for(int g = 0; g < SPLAT_GROUP; g++)
      {
        const float zf = cell[u0 + g].x;
        zi[g] = __float_as_int(cell[u0 + g].y);
        lo[g] = contrib[u0 + g] * (1.0f - zf);
        hi[g] = contrib[u0 + g] * zf;
        sum_lo[g] = acc[zi[g] * SPLAT_THREADS + tid];
        sum_hi[g] = acc[(zi[g] + 1) * SPLAT_THREADS + tid];
      }
#pragma unroll
      for(int g = 0; g < SPLAT_GROUP; g++)
      {
#pragma unroll
        for(int p = 0; p < g; p++)
        {
          sum_lo[g] = zi[g] == zi[p] ? sum_lo[p] : (zi[g] == zi[p] + 1 ? sum_hi[p] : sum_lo[g]);
          sum_hi[g] = zi[g] == zi[p] ? sum_hi[p] : (zi[g] + 1 == zi[p] ? sum_lo[p] : sum_hi[g]);
        }
        sum_lo[g] += lo[g];
        sum_hi[g] += hi[g];
      }
#pragma unroll
      for(int g = 0; g < SPLAT_GROUP; g++)
      {
        acc[zi[g] * SPLAT_THREADS + tid] = sum_lo[g];
        acc[(zi[g] + 1) * SPLAT_THREADS + tid] = sum_hi[g];
      }
    }
  };
  // the block behind (j, ub) in the walk; j > j1: none
  auto advance = [&](int &j, int &ub, float &wy) {
    ub += 16;
    if(ub >= span)
    {
      ub = 0;
      j = next_row(j + 1, wy);
    }
  };
  // One walk over blocks of 16 columns, row after row.  A block's (zf, zi) pairs are fetched while the block before it is added (two
  // sets of registers, the loop unrolled by two so that no set is copied): the walk is one chain of LDS round trips (a cell's
  // contributions meet in pixel order), and a fetch at the head of every block put a round trip to memory into it every sixteen pixels.
  float wy = 0.0f;
  int j = next_row(j0, wy), ub = 0;
  float2 cell_a[16], cell_b[16];
  float contrib[16];
  if(j <= j1) fetch(cell_a, j, 0);
  while(j <= j1)
  {
    contributions(contrib, wy, ub);
    advance(j, ub, wy);
    // (behind the last block: any row of the footprint, unconditionally -- with the fetch under a branch the compiler's wait-count
    // pass takes the path without it and drains every fetch in flight before the additions use the block fetched a round before)
    fetch(cell_b, min(j, j1), ub);
    __builtin_amdgcn_sched_barrier(0); // the fetches are ISSUED here (left to itself the compiler sinks them behind the additions)
    add(cell_a, contrib);
    if(j > j1) break;
    contributions(contrib, wy, ub);
    advance(j, ub, wy);
    fetch(cell_a, min(j, j1), ub);
    __builtin_amdgcn_sched_barrier(0);
    add(cell_b, contrib);
  }
  if(live)
  {
    float *const cell = buf + (size_t)(X + Y * b.size_x) * b.size_z;
    for(int z = 0; z < b.size_z; z++) cell[z] = acc[z * SPLAT_THREADS + tid];
  }
}

// ---- dt_bilateral_blur(), src/pixel/bilateral.c:266-352, as out-of-place stencils -------------------------------------------------
// The reference walks every grid line in place, carrying the two samples it has just overwritten in scalars: each result is a
// function of the line's ORIGINAL samples i - 2 .. i + 2 only, and the walk's first two and last two steps are that stencil with
// the taps beyond the line's ends left out.  So a pass is a 5-tap stencil that any lane can evaluate for any cell -- what has to be
// kept is each result's operands and the order they meet in, (centre + w1 (next + previous)) + w2 (second next + second previous):
//   along x and y:  6/16, 4/16, 1/16           (bilateral.c:299-338)
//   along z:        the -2 derivative 4/16 (next - previous) + 2/16 (second next - second previous)   (:258-297)
// `at(k)` is sample k of the line, n >= 4 its length (bilat_grid_of() refuses shorter lines: the reference writes past them).
template <class At> __device__ __forceinline__ float bilat_smooth_tap(const int i, const int n, const At at)
{
  constexpr float centre = 6.f / 16.f, near = 4.f / 16.f, far = 1.f / 16.f;
  const float c = at(i) * centre;
  if(i == 0) return (c + near * at(1)) + far * at(2);
  if(i == 1) return (c + near * (at(2) + at(0))) + far * at(3);
  if(i == n - 1) return (c + near * at(n - 2)) + far * at(n - 3);
  if(i == n - 2) return (c + near * (at(n - 1) + at(n - 3))) + far * at(n - 4);
  return (c + near * (at(i + 1) + at(i - 1))) + far * (at(i + 2) + at(i - 2));
}
template <class At> __device__ __forceinline__ float bilat_derivative_tap(const int i, const int n, const At at)
{
  constexpr float near = 4.f / 16.f, far = 2.f / 16.f;
  if(i == 0) return near * at(1) + far * at(2);
  if(i == 1) return near * (at(2) - at(0)) + far * at(3);
  if(i == n - 1) return -near * at(n - 2) - far * at(n - 3);
  if(i == n - 2) return near * (at(n - 1) - at(n - 3)) - far * at(n - 4);
  return near * (at(i + 1) - at(i - 1)) + far * (at(i + 2) - at(i - 2));
}

// The three passes in ONE launch, src -> dst (two buffers: every cell reads its neighbours' unblurred values).  Layout [y][x][z], z
// contiguous.  A workgroup owns `npw` whole z-lines of one grid row y -- a contiguous run of the row, so lanes walk memory in order:
//   1. per cell, the y-pass over the x-pass: the x-smoothed value of the five rows y - 2 .. y + 2 at the cell's (x, z), each from
//      its five x-taps (25 reads of a grid that lives in L2; the x-pass of a row is recomputed by the five rows that use it -- a
//      grid is 10^-3 of the frame), smoothed along y, into LDS;
//   2. barrier; the z-derivative along each z-line from LDS -> dst.
// XPASS false: src holds the x-pass already (bilat_blur_x below; grids too large to recompute it five times).
template <bool XPASS>
__global__ __launch_bounds__(256) void bilat_blur_yz(const float *__restrict__ src, float *__restrict__ dst, const int size_x,
                                                     const int size_y, const int size_z, const int npw)
{
  extern __shared__ float lines[]; // [nodes of this workgroup][size_z]
  const int y = blockIdx.y, node0 = blockIdx.x * npw;
  const int nn = min(npw, size_x - node0), ncell = nn * size_z;
  const size_t pitch_x = (size_t)size_z, pitch_y = (size_t)size_x * size_z;
  for(int c = threadIdx.x; c < ncell; c += 256)
  {
    const int x = node0 + c / size_z, z = c - (c / size_z) * size_z;
    const auto row_value = [&](const int yy) {
      const float *const p = src + yy * pitch_y + z;
      if(!XPASS) return p[x * pitch_x];
      return bilat_smooth_tap(x, size_x, [&](const int k) { return p[k * pitch_x]; });
    };
    lines[c] = bilat_smooth_tap(y, size_y, row_value);
  }
  __syncthreads();
  float *const out_row = dst + y * pitch_y + node0 * pitch_x;
  for(int c = threadIdx.x; c < ncell; c += 256)
  {
    const int node = c / size_z, z = c - node * size_z;
    const float *const line = lines + node * size_z;
    out_row[c] = bilat_derivative_tap(z, size_z, [&](const int k) { return line[k]; });
  }
}
// the x-pass alone, src -> dst, one lane per cell
__global__ __launch_bounds__(256) void bilat_blur_x(const float *__restrict__ src, float *__restrict__ dst, const int size_x,
                                                    const int size_z, const size_t cells)
{
  const size_t row_cells = (size_t)size_x * size_z;
  for(size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (size_t)gridDim.x * blockDim.x)
  {
    const size_t in_row = c % row_cells;
    const int x = (int)(in_row / size_z);
    const float *const p = src + (c - (size_t)x * size_z); // the line's sample 0 at this cell's (y, z)
    dst[c] = bilat_smooth_tap(x, size_x, [&](const int k) { return p[(size_t)k * size_z]; });
  }
}

// dt_bilateral_slice(), bilateral.c:356-393
// `rows` rows from frame row `row0` on (a row band; the whole frame: 0, height); in / out hold those rows
__global__ __launch_bounds__(256) void bilat_slice(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                   const float *__restrict__ buf, const grid_t b, const float norm,
                                                   const int row0, const int rows)
{
  const size_t n = (size_t)b.width * rows;
  for(size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x)
  {
    const int jl = (int)(p / b.width), i = (int)(p - (size_t)jl * b.width);
    const int j = row0 + jl;
    const float4 px = in[p];
    const float Lout = bilat_slice_lightness(px.x, i, j, buf, b, norm);
    nt_store(out + p, make_float4(Lout, px.y, px.z, px.w));
  }
}

// dt_bilateral_grid_size(), bilateral.c:50-74
void grid_size(grid_t &b, const int width, const int height, const float L_range, float sigma_s, const float sigma_r)
{
  if(sigma_s < 0.5) sigma_s = 0.5;
  const float _x = (float)clampi_h((int)roundf(width / sigma_s), 4, MAX_RES_S);
  const float _y = (float)clampi_h((int)roundf(height / sigma_s), 4, MAX_RES_S);
  const float _z = (float)clampi_h((int)roundf(L_range / sigma_r), 4, MAX_RES_R);
  const float sy = height / _y, sx = width / _x;
  b.sigma_s = sy > sx ? sy : sx;
  b.sigma_r = L_range / _z;
  b.size_x = (int)ceilf(width / b.sigma_s) + 1;
  b.size_y = (int)ceilf(height / b.sigma_s) + 1;
  b.size_z = (int)ceilf(L_range / b.sigma_r) + 1;
  b.width = width;
  b.height = height;
}

} // namespace

namespace ansel
{
int local_laplacian_launch(int devid, const float4 *in, float4 *out, int wd, int ht, float sigma, float shadows,
                           float highlights, float clarity);
}

namespace
{
// the grid of the FRAME `piece` describes (bilat.c:339-346)
int bilat_grid_of(const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d, grid_t &b)
{
  const int width = piece->roi_in.width, height = piece->roi_in.height;
  if(!(d->iscale > 0.0f) || !(piece->roi_in.scale > 0.0) || !(d->sigma_r > 0.0f)) return DT_HIP_INVALID_ARG;
  const float scale = (float)(d->iscale / piece->roi_in.scale); // dt_dev_get_module_scale(), bilat.c:339
  grid_size(b, width, height, 100.0f, d->sigma_s / scale, d->sigma_r);
  if(b.size_x < 4 || b.size_y < 4 || b.size_z < 4)
  {
    // blur_line() / blur_line_z() (src/pixel/bilateral.c:266-340) touch four entries of every grid line
    // unconditionally: on a shorter line the reference writes past it (heap corruption on the CPU)
    set_last_error("bilat: a %d x %d x %d grid is below the 4 entries per line the reference's blur assumes", b.size_x,
                   b.size_y, b.size_z);
    return DT_HIP_INVALID_ARG;
  }
  return DT_HIP_SUCCESS;
}

#ifdef ANSEL_HIP_MEASURING
__global__ __launch_bounds__(256) void bilat_lightness(const float4 *__restrict__ in, float *__restrict__ L, const size_t n)
{
  for(size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) L[k] = in[k].x;
}
#endif

// zc_pre: the rows' lightness cells, already computed (pipe.cpp: by the non-local-means kernels in front of the module), or nullptr
int bilat_splat_rows(int devid, const grid_t &b, float *buf, const float4 *in_rows, int row_lo, int row_hi, int accumulate,
                     const float2 *zc_pre = nullptr)
{
  hipStream_t s = stream_of(devid);
  const size_t n = (size_t)b.width * (row_hi - row_lo);
#ifdef ANSEL_HIP_MEASURING
  const int nodes = b.size_x * b.size_y;
  static const bool v1 = getenv("ANSEL_HIP_BILAT_SPLAT_V1") != nullptr; // the first gather, for A/B timing
  if(v1)
  {
    float *L = (float *)dt_hip_alloc_device_buffer(devid, n * sizeof(float));
    if(!L) return DT_HIP_SYSMEM_ALLOCATION;
    {
      launch_scope ls(devid, "bilat_splat");
      bilat_lightness<<<stream_grid(n, 256), 256, 0, s>>>(in_rows, L, n);
      bilat_splat<<<(nodes + SPLAT_THREADS - 1) / SPLAT_THREADS, SPLAT_THREADS, (size_t)b.size_z * SPLAT_THREADS * sizeof(float), s>>>(
          L, buf, b, row_lo, row_hi, accumulate);
    }
    dt_hip_release_mem_object(L); // stream-ordered
    return DT_HIP_SUCCESS;
  }
#endif
  float2 *zc = zc_pre ? const_cast<float2 *>(zc_pre) : (float2 *)dt_hip_alloc_device_buffer(devid, n * sizeof(float2));
  if(!zc) return DT_HIP_SYSMEM_ALLOCATION;
  {
    launch_scope ls(devid, "bilat_splat");
    if(!zc_pre) bilat_zcells<<<stream_grid(n, 256), 256, 0, s>>>(in_rows, zc, n, b.sigma_r, b.size_z);
    const size_t cells = (size_t)b.size_z * SPLAT_THREADS * sizeof(float);
    const size_t table = (size_t)((int)ceilf(2.0f * b.sigma_s) + 8) * SPLAT_THREADS * sizeof(float);
    // div_by()'s premise: numerators are 0 or >= 2^-46 * 100 (two fractions of at least one ulp of a grid coordinate
    // below 2^12 each), so with sigma_s^2 inside [2^-3, 2^40] every quotient is a normal number far from either end
    const float s2 = b.sigma_s * b.sigma_s;
    const bool fast_div = s2 >= 0.125f && s2 <= 1099511627776.0f;
    const dim3 grid((b.size_x + SPLAT_THREADS - 1) / SPLAT_THREADS, b.size_y);
    if(cells + table <= 64 * 1024)
    {
      if(fast_div) bilat_splat2<true, true><<<grid, SPLAT_THREADS, cells + table, s>>>(zc, buf, b, row_lo, row_hi, accumulate);
      else bilat_splat2<true, false><<<grid, SPLAT_THREADS, cells + table, s>>>(zc, buf, b, row_lo, row_hi, accumulate);
    }
    else
    {
      if(fast_div) bilat_splat2<false, true><<<grid, SPLAT_THREADS, cells, s>>>(zc, buf, b, row_lo, row_hi, accumulate);
      else bilat_splat2<false, false><<<grid, SPLAT_THREADS, cells, s>>>(zc, buf, b, row_lo, row_hi, accumulate);
    }
  }
  if(!zc_pre) dt_hip_release_mem_object(zc); // stream-ordered
  return DT_HIP_SUCCESS;
}

// dt_bilateral_blur() of the complete grid, then the slice of `rows` rows from frame row `row0`.  The blur is out of place: the
// fused launch leaves it in a second grid the slice reads (buf keeps the splat); a grid of more than BLUR_FUSED_CELLS cells takes
// its x-pass in a launch of its own (second grid), and the y- and z-passes bring it back into buf
constexpr size_t BLUR_FUSED_CELLS = (size_t)1 << 22;
// the blur: *blurred = the grid that holds it (buf or *second), *second = the grid to release once nothing reads `blurred` any more
int bilat_blur(int devid, const grid_t &b, float *buf, const float **blurred, float **second_out)
{
  hipStream_t s = stream_of(devid);
  const size_t cells = (size_t)b.size_x * b.size_y * b.size_z;
  float *const second = (float *)dt_hip_alloc_device_buffer(devid, cells * sizeof(float));
  if(!second) return DT_HIP_SYSMEM_ALLOCATION;
  *second_out = second;
  launch_scope ls(devid, "bilat_blur");
  // z-lines per workgroup: ~2048 cells (eight per lane), whole lines
  const int npw = b.size_z >= 2048 ? 1 : 2048 / b.size_z;
  const dim3 grid((b.size_x + npw - 1) / npw, b.size_y);
  const size_t lds = (size_t)npw * b.size_z * sizeof(float);
  if(cells <= BLUR_FUSED_CELLS && !dispatch_override(DISPATCH_BILAT_BLUR_SPLIT))
  {
    bilat_blur_yz<true><<<grid, 256, lds, s>>>(buf, second, b.size_x, b.size_y, b.size_z, npw);
    *blurred = second;
  }
  else
  {
    bilat_blur_x<<<stream_grid(cells, 256), 256, 0, s>>>(buf, second, b.size_x, b.size_z, cells);
    bilat_blur_yz<false><<<grid, 256, lds, s>>>(second, buf, b.size_x, b.size_y, b.size_z, npw);
    *blurred = buf;
  }
  return DT_HIP_SUCCESS;
}
int bilat_blur_and_slice(int devid, const grid_t &b, float *buf, const dt_hip_bilat_data_t *d, const float4 *in_rows,
                         float4 *out_rows, int row0, int rows)
{
  hipStream_t s = stream_of(devid);
  const float *blurred = nullptr;
  float *second = nullptr;
  const int berr = bilat_blur(devid, b, buf, &blurred, &second);
  if(berr != DT_HIP_SUCCESS) return berr;
  {
    const float norm = -d->detail * b.sigma_r * 0.04f;
    launch_scope ls(devid, "bilat_slice");
    bilat_slice<<<stream_grid((size_t)b.width * rows, 256), 256, 0, s>>>(in_rows, out_rows, blurred, b, norm, row0, rows);
  }
  dt_hip_release_mem_object(second); // stream-ordered
  return check_launch("bilat");
}
} // namespace

namespace ansel
{
// ---- row bands (pipe.cpp): the grid is ONE accumulation over the frame in pixel order, so the
// bands splat one after the other into the same grid (a relay: band k starts from what bands 0..k-1 left, a few
// hundred KB travelling from GPU to GPU), the last band's grid is broadcast, and every band blurs its copy and slices
// its own rows.  Bit-identical to the unsplit module.
int bilat_band_supported(const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d)
{
  grid_t b;
  return d->mode == DT_HIP_BILAT_BILATERAL && piece->channels == 4 && bilat_grid_of(piece, d, b) == DT_HIP_SUCCESS;
}
// allocates the zeroed grid of the frame; *bytes = its size
int bilat_band_begin(int devid, const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d, dt_hip_mem_t *grid, size_t *bytes)
{
  grid_t b;
  const int gerr = bilat_grid_of(piece, d, b);
  if(gerr != DT_HIP_SUCCESS) return gerr;
  *bytes = (size_t)b.size_x * b.size_y * b.size_z * sizeof(float);
  *grid = dt_hip_alloc_device_buffer(devid, *bytes);
  if(!*grid) return DT_HIP_SYSMEM_ALLOCATION;
  if(hipMemsetAsync(*grid, 0, *bytes, stream_of(devid)) != hipSuccess) return DT_HIP_DEFAULT_ERROR;
  return DT_HIP_SUCCESS;
}
// the band's rows [row0, row0 + rows) on top of what `grid` holds
int bilat_band_splat(int devid, const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d, dt_hip_mem_t grid,
                     dt_hip_mem_t in_rows, int row0, int rows)
{
  grid_t b;
  const int gerr = bilat_grid_of(piece, d, b);
  if(gerr != DT_HIP_SUCCESS) return gerr;
  const int serr = bilat_splat_rows(devid, b, (float *)grid, (const float4 *)in_rows, row0, row0 + rows, 1);
  return serr != DT_HIP_SUCCESS ? serr : check_launch("bilat_splat");
}
// `grid` = the complete splat of the frame: blur it (in place, this band's copy) and slice the band's rows
int bilat_band_finish(int devid, const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d, dt_hip_mem_t grid,
                      dt_hip_mem_t in_rows, dt_hip_mem_t out_rows, int row0, int rows)
{
  grid_t b;
  const int gerr = bilat_grid_of(piece, d, b);
  if(gerr != DT_HIP_SUCCESS) return gerr;
  return bilat_blur_and_slice(devid, b, (float *)grid, d, (const float4 *)in_rows, (float4 *)out_rows, row0, rows);
}
} // namespace ansel

namespace ansel
{
// The module in bilateral-grid mode with the fused RGBA run `chain` behind it (pipe.cpp): splat and blur as the module runs them, the slice
// as the first stage of the run's kernel -- the module's output plane is neither written nor read.  DT_HIP_INVALID_ARG: not this
// mode (the caller runs the two one after the other).
// the grid's third axis for the frame `piece` describes: what a kernel that writes lightness cells for the module needs
int bilat_cell_params(const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d, float *sigma_r, int *size_z)
{
  if(!piece || !d || d->mode != DT_HIP_BILAT_BILATERAL || piece->channels != 4) return DT_HIP_INVALID_ARG;
  grid_t b;
  if(bilat_grid_of(piece, d, b) != DT_HIP_SUCCESS) return DT_HIP_INVALID_ARG;
  *sigma_r = b.sigma_r;
  *size_z = b.size_z;
  return DT_HIP_SUCCESS;
}
// the module (bilateral-grid mode) on a frame whose lightness cells `cells` exist already
int bilat_process_cells(int devid, const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out,
                        dt_hip_mem_t cells)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out || !cells || piece->channels != 4) return DT_HIP_INVALID_ARG;
  if(d->mode != DT_HIP_BILAT_BILATERAL) return DT_HIP_INVALID_ARG;
  const int height = piece->roi_in.height;
  grid_t b;
  if(bilat_grid_of(piece, d, b) != DT_HIP_SUCCESS) return DT_HIP_INVALID_ARG;
  const size_t ncells = (size_t)b.size_x * b.size_y * b.size_z;
  float *buf = (float *)dt_hip_alloc_device_buffer(devid, ncells * sizeof(float));
  if(!buf) return DT_HIP_SYSMEM_ALLOCATION;
  int err = bilat_splat_rows(devid, b, buf, (const float4 *)dev_in, 0, height, 0, (const float2 *)cells);
  if(err == DT_HIP_SUCCESS) err = bilat_blur_and_slice(devid, b, buf, d, (const float4 *)dev_in, (float4 *)dev_out, 0, height);
  dt_hip_release_mem_object(buf);
  return err;
}
int bilat_process_chain(int devid, const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out,
                        const rgb_group_t *chain, dt_hip_mem_t pixel_cells)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out || !chain || piece->channels != 4) return DT_HIP_INVALID_ARG;
  if(d->mode != DT_HIP_BILAT_BILATERAL) return DT_HIP_INVALID_ARG;
  const int width = piece->roi_in.width, height = piece->roi_in.height;
  if(width <= 0 || height <= 0 || chain->width != width || chain->height != height) return DT_HIP_INVALID_ARG;
  grid_t b;
  if(bilat_grid_of(piece, d, b) != DT_HIP_SUCCESS) return DT_HIP_INVALID_ARG;
  const size_t cells = (size_t)b.size_x * b.size_y * b.size_z;
  float *buf = (float *)dt_hip_alloc_device_buffer(devid, cells * sizeof(float));
  if(!buf) return DT_HIP_SYSMEM_ALLOCATION;
  int err = bilat_splat_rows(devid, b, buf, (const float4 *)dev_in, 0, height, 0, (const float2 *)pixel_cells);
  const float *blurred = nullptr;
  float *second = nullptr;
  if(err == DT_HIP_SUCCESS) err = bilat_blur(devid, b, buf, &blurred, &second);
  if(err == DT_HIP_SUCCESS)
  {
    bilat_slice_args sl;
    sl.b = b;
    sl.grid = blurred;
    sl.norm = -d->detail * b.sigma_r * 0.04f;
    err = rgb_group_launch(devid, *chain, dev_in, dev_out, &sl);
  }
  if(second) dt_hip_release_mem_object(second); // stream-ordered
  dt_hip_release_mem_object(buf);
  return err;
}
} // namespace ansel

extern "C" {

int dt_hip_iop_bilat_process(int devid, const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d, dt_hip_mem_t dev_in,
                             dt_hip_mem_t dev_out)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out || piece->channels != 4) return DT_HIP_INVALID_ARG;
  const int width = piece->roi_in.width, height = piece->roi_in.height;
  if(width <= 0 || height <= 0) return DT_HIP_SUCCESS;
  if(d->mode == DT_HIP_BILAT_LOCAL_LAPLACIAN) // bilat.c:352-357: (midtone, sigma_s, sigma_r, detail)
  {
    if((width < height ? width : height) == 2 || (width < height ? width : height) == 3)
    {
      // one pyramid level: the reference indexes its level array at -1 (locallaplacian.c:405) and, for a side of
      // 3, pads for two levels -- it reads outside its buffers (and crashes): nothing to be identical to
      set_last_error("bilat: local laplacian is undefined in the reference for a 2- or 3-pixel side (%d x %d)", width, height);
      return DT_HIP_INVALID_ARG;
    }
    return local_laplacian_launch(devid, (const float4 *)dev_in, (float4 *)dev_out, width, height, d->midtone, d->sigma_s,
                                  d->sigma_r, d->detail);
  }
  if(d->mode != DT_HIP_BILAT_BILATERAL)
  {
    set_last_error("bilat: unknown mode %d", d->mode);
    return DT_HIP_INVALID_ARG;
  }
  grid_t b;
  const int gerr = bilat_grid_of(piece, d, b);
  if(gerr != DT_HIP_SUCCESS) return gerr;
  const size_t cells = (size_t)b.size_x * b.size_y * b.size_z;
  float *buf = (float *)dt_hip_alloc_device_buffer(devid, cells * sizeof(float));
  if(!buf) return DT_HIP_SYSMEM_ALLOCATION;
  int err = bilat_splat_rows(devid, b, buf, (const float4 *)dev_in, 0, height, 0);
  if(err == DT_HIP_SUCCESS) err = bilat_blur_and_slice(devid, b, buf, d, (const float4 *)dev_in, (float4 *)dev_out, 0, height);
  dt_hip_release_mem_object(buf);
  return err;
}

// tiling_callback(), src/iop/bilat.c:252-297.  factor / maxbuf / overlap are the reference's, for the host's own
// tiling; factor_cl / maxbuf_cl are what THIS implementation holds on the device: in + out + two grids (the blur is
// out of place: the splat and its blurred copy), or in + out + the (2 + 6)-plane padded pyramid of local_laplacian_memory_use(), locallaplacian.c:566-591
void dt_hip_iop_bilat_tiling(const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d, dt_hip_tiling_t *tiling)
{
  const int width = piece->roi_in.width, height = piece->roi_in.height;
  const float basebuffer = (float)sizeof(float) * piece->channels * width * height;
  const float scale = (float)(d->iscale / piece->roi_in.scale);
  memset(tiling, 0, sizeof(*tiling));
  tiling->xalign = tiling->yalign = 1;
  if(d->mode == DT_HIP_BILAT_BILATERAL)
  {
    grid_t b;
    const float sigma_s = d->sigma_s / scale;
    grid_size(b, width, height, 100.0f, sigma_s, d->sigma_r);
    const float grid = (float)((size_t)b.size_x * b.size_y * b.size_z * sizeof(float));
    tiling->factor = 2.0f + 2.0f * grid / basebuffer; // dt_bilateral_memory_use(), bilateral.c:80-94 (OpenCL build)
    tiling->maxbuf = fmaxf(1.0f, grid / basebuffer);
    tiling->factor_cl = 2.0f + 2.0f * grid / basebuffer;
    tiling->maxbuf_cl = tiling->maxbuf;
    tiling->overlap = (unsigned)ceilf(4 * sigma_s);
  }
  else
  {
    const int m = width < height ? width : height;
    const int nl = m > 0 ? 31 - __builtin_clz((unsigned)m) : 0;
    const int num_levels = nl < 30 ? nl : 30;
    const int max_supp = num_levels > 0 ? 1 << (num_levels - 1) : 1;
    auto dl = [](int size, const int level) {
      for(int l = 0; l < level; l++) size = (size - 1) / 2 + 1;
      return size;
    };
    float mem = 0.0f;
    for(int l = 0; l < num_levels; l++)
      mem += (float)sizeof(float) * (2 + 6) * dl(width + 2 * max_supp, l) * dl(height + 2 * max_supp, l);
    const float single = (float)sizeof(float) * (width + 2 * max_supp) * (height + 2 * max_supp);
    tiling->factor = tiling->factor_cl = 2.0f + mem / basebuffer;
    tiling->maxbuf = tiling->maxbuf_cl = fmaxf(1.0f, single / basebuffer);
    const float rad = ceilf(256.0f / scale);
    tiling->overlap = (unsigned)((float)width < rad ? (float)width : rad);
  }
}

} // extern "C"
