// demosaic_rcd.hip -- ratio-corrected demosaic (RCD) of a Bayer mosaic on gfx950.
//
// Reference: rcd_demosaic(), src/iop/demosaic/rcd.c:274-564 and rcd_ppg_border(), :92-272
// (CPU path; the reference's OpenCL version, data/kernels/demosaic_rcd.cl, is 13 full-frame
// launches over 8 full-size float scratch planes and is NOT what this follows).
//
// MI355X design.  The CPU code works on 112 x 112 tiles whose inner 94 x 94 are written out,
// and its results near a tile rim depend on the tile grid (VH_Dir is zero on the 4-px rim), so
// the grid is part of the algorithm.  Here ONE WORKGROUP OWNS ONE REFERENCE TILE and keeps the
// whole working set in LDS -- 147 KiB of the CU's 160 KiB:
//
//     cfa   112 x 112 f32   normalised, clamped mosaic                        50176 B
//     vh    112 x 112 f32   VH_Dir (vertical/horizontal discrimination)       50176 B
//     g     112 x  56 f32   green at red/blue sites                           25088 B
//     x     112 x  56 f32   low-pass filter -> PQ_Dir -> colour at the        25088 B
//                           opposite-colour site, reusing one buffer as each dies
//
// so HBM sees exactly one read of the mosaic (+19 % halo) and one float4 write per pixel:
// 4 + 16 = 20 B/px, against ~250 B/px of full-frame scratch traffic in the reference's OpenCL
// decomposition.  The P/Q colour-difference high-pass planes of the CPU code are not stored:
// each PQ_Dir needs three samples of each, recomputed from cfa in LDS (17 LDS reads, same count
// as storing and re-reading them).  Half-resolution planes are indexed (row * 112 + col) / 2,
// exactly as in the reference (that indexing is visible in its results, see below).
//
// Numerics.  Same operation order as the CPU code, one IEEE operation per C operation
// (-ffp-contract=off).  The reference's gradient sums call fabs() -- double -- so they are
// accumulated in binary64 and rounded once (rcd.c:414-417, 478-481, 508-527); the kernel does the
// same in v_add_f64.  rcd_demosaic() runs with FTZ/DAZ (rcd.c:300); this file is compiled with
// -fgpu-flush-denormals-to-zero.  Scratch words the CPU code reads without having written them in
// the current tile are 0.0f (see oracle/src/demosaic_rcd.c for the two places this matters).
#include "hip_common.h"
#include "ppg_device.h"
#include <atomic>
#include <cstdlib>

using namespace ansel;

namespace
{

constexpr int TS = 112;        // RCD_TILESIZE
constexpr int HS = TS / 2;     // row length of a half-resolution plane
constexpr int RCD_BORDER = 9;
constexpr int RCD_MARGIN = 6;
constexpr int TV = TS - 2 * RCD_BORDER;
[[maybe_unused]] constexpr int W1 = TS, W2 = 2 * TS, W3 = 3 * TS, W4 = 4 * TS; [[maybe_unused]] constexpr size_t LDS_BYTES = sizeof(float) * (2 * TS * TS + 2 * TS * HS); // (W*: device code only; LDS_BYTES: the measuring build's first kernel)
constexpr int NT = 896;        // 14 waves: 112*112 / 896 = 14 and 112*56 / 896 = 7 exactly
constexpr int FULL_ITERS = TS * TS / NT;
constexpr int HALF_ITERS = TS * HS / NT;

#define EPS 1e-5f
#define EPSSQ 1e-10f

__device__ __forceinline__ int fc(const int row, const int col, const uint32_t filters)
{
  return filters >> ((((row << 1) & 14) + (col & 1)) << 1) & 3;
}
__device__ __forceinline__ float sqf(const float v) { return v * v; }
__device__ __forceinline__ float intp(const float a, const float b, const float c) { return a * (b - c) + c; }
__device__ __forceinline__ double dabs(const float v) { return fabs((double)v); }

__device__ __forceinline__ float hpf(const float *c, const int i, const int s)
{
  return sqf((c[i - 3 * s] - c[i - s] - c[i + s] + c[i + 3 * s]) - 3.0f * (c[i - 2 * s] + c[i + 2 * s]) + 6.0f * c[i]);
}

// P/Q_CDiff_Hpf word idx2 of the reference (rcd.c:444-451): computed at (row, odd col) with
// idx2 == (row*TS + col)/2; 0 where step 4.0 of this tile does not write it
__device__ __forceinline__ float pq_hpf(const float *cfa, const int idx2, const int tileRows, const int tileCols,
                                        const int s)
{
  const int row = idx2 / HS;
  const int col = 2 * (idx2 - row * HS) + 1;
  if(row < 3 || row >= tileRows - 3 || col < 3 || col >= tileCols - 3) return 0.0f;
  return hpf(cfa, row * TS + col, s);
}

struct rcd_args
{
  int width, height;
  uint32_t filters;
  float scaler, revscaler;
  int num_vertical, num_horizontal;
  // band mode (multi-GPU row bands): the launch covers tile rows tv0.. of the
  // frame's own tile grid; `in` row 0 is frame row in_row0, `out` row 0 is frame row out_row0 and
  // only frame rows [out_row0, out_row1) are written.  Whole frame: 0, 0, 0, height.
  int tv0, in_row0, out_row0, out_row1;
};

#ifdef ANSEL_HIP_MEASURING
// The first version of the tile kernel (row-major planes, one index decode per site and step): kept for A/B timing
// (ANSEL_HIP_RCD_V1, measuring builds only) -- rcd_tiles below computes the same values from de-interleaved planes.
__global__ __launch_bounds__(NT) void rcd_tiles_v1(const float *__restrict__ in, float4 *__restrict__ out,
                                                    const rcd_args a)
{
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *const cfa = lds;
  float *const vh = cfa + TS * TS;
  float *const g = vh + TS * TS;
  float *const x = g + TS * HS;

  const int tid = threadIdx.x;
  const int tile_vertical_local = blockIdx.x / a.num_horizontal;
  const int tile_horizontal = blockIdx.x - tile_vertical_local * a.num_horizontal;
  const int tile_vertical = a.tv0 + tile_vertical_local;
  const int rowStart = tile_vertical * TV, rowEnd = min(rowStart + TS, a.height);
  const int colStart = tile_horizontal * TV, colEnd = min(colStart + TS, a.width);
  const int tileRows = rowEnd - rowStart, tileCols = colEnd - colStart;
  const uint32_t filters = a.filters;
  // column parity of the red/blue sites of tile row 0 and 1 (tile origins are even)
  const int p0 = fc(0, 0, filters) & 1, p1 = fc(1, 0, filters) & 1;

  // ---- step 0 (rcd.c:345-354): cfa = fmaxf(0, in) * revscaler, zero outside a partial tile
#pragma unroll
  for(int it = 0; it < FULL_ITERS; it++)
  {
    const int idx = tid + it * NT;
    const int row = idx / TS, col = idx - row * TS;
    float v = 0.0f;
    if(row < tileRows && col < tileCols)
      v = fmaxf(0.0f, in[(size_t)(rowStart - a.in_row0 + row) * a.width + colStart + col]) * a.revscaler;
    cfa[idx] = v;
  }
  __syncthreads();

  // ---- step 1 (rcd.c:356-390) VH_Dir, 0 on the rim; step 2.1 (rcd.c:394-402) lpf at red/blue
  //      sites; rgb[1] starts as cfa (rcd.c:352)
#pragma unroll
  for(int it = 0; it < FULL_ITERS; it++)
  {
    const int idx = tid + it * NT;
    const int row = idx / TS, col = idx - row * TS;
    float v = 0.0f;
    if(row >= 4 && row < tileRows - 4 && col >= 4 && col < tileCols - 4)
    {
      const float V_Stat = fmaxf(EPSSQ, hpf(cfa, idx - W1, W1) + hpf(cfa, idx, W1) + hpf(cfa, idx + W1, W1));
      const float H_Stat = fmaxf(EPSSQ, hpf(cfa, idx - 1, 1) + hpf(cfa, idx, 1) + hpf(cfa, idx + 1, 1));
      v = V_Stat / (V_Stat + H_Stat);
    }
    vh[idx] = v;
  }
#pragma unroll
  for(int it = 0; it < HALF_ITERS; it++)
  {
    const int h = tid + it * NT;
    const int row = h / HS, col = 2 * (h - row * HS) + ((row & 1) ? p1 : p0);
    const int idx = row * TS + col;
    float lp = 0.0f;
    if(row >= 2 && row < tileRows - 2 && col >= 2 && col < tileCols - 2)
      lp = cfa[idx] + 0.5f * (cfa[idx - W1] + cfa[idx + W1] + cfa[idx - 1] + cfa[idx + 1])
           + 0.25f * (cfa[idx - W1 - 1] + cfa[idx - W1 + 1] + cfa[idx + W1 - 1] + cfa[idx + W1 + 1]);
    x[h] = lp;
    g[h] = cfa[idx];
  }
  __syncthreads();

  // ---- step 3.1 (rcd.c:406-440): green at red/blue sites
#pragma unroll
  for(int it = 0; it < HALF_ITERS; it++)
  {
    const int h = tid + it * NT;
    const int row = h / HS, col = 2 * (h - row * HS) + ((row & 1) ? p1 : p0);
    if(row < 4 || row >= tileRows - 4 || col < 4 || col >= tileCols - 4) continue;
    const int indx = row * TS + col;
    const float cfai = cfa[indx];
    const float cN1 = cfa[indx - W1], cS1 = cfa[indx + W1], cW1 = cfa[indx - 1], cE1 = cfa[indx + 1];
    const float cN2 = cfa[indx - W2], cS2 = cfa[indx + W2], cW2 = cfa[indx - 2], cE2 = cfa[indx + 2];
    const float N_Grad = (float)((double)EPS + dabs(cN1 - cS1) + dabs(cfai - cN2) + dabs(cN1 - cfa[indx - W3]) + dabs(cN2 - cfa[indx - W4]));
    const float S_Grad = (float)((double)EPS + dabs(cN1 - cS1) + dabs(cfai - cS2) + dabs(cS1 - cfa[indx + W3]) + dabs(cS2 - cfa[indx + W4]));
    const float W_Grad = (float)((double)EPS + dabs(cW1 - cE1) + dabs(cfai - cW2) + dabs(cW1 - cfa[indx - 3]) + dabs(cW2 - cfa[indx - 4]));
    const float E_Grad = (float)((double)EPS + dabs(cW1 - cE1) + dabs(cfai - cE2) + dabs(cE1 - cfa[indx + 3]) + dabs(cE2 - cfa[indx + 4]));
    const float lpfi = x[h];
    const float N_Est = cN1 * (lpfi + lpfi) / (EPS + lpfi + x[h - W1]);
    const float S_Est = cS1 * (lpfi + lpfi) / (EPS + lpfi + x[h + W1]);
    const float W_Est = cW1 * (lpfi + lpfi) / (EPS + lpfi + x[h - 1]);
    const float E_Est = cE1 * (lpfi + lpfi) / (EPS + lpfi + x[h + 1]);
    const float V_Est = (S_Grad * N_Est + N_Grad * S_Est) / (N_Grad + S_Grad);
    const float H_Est = (W_Grad * E_Est + E_Grad * W_Est) / (E_Grad + W_Grad);
    const float VH_Central_Value = vh[indx];
    const float VH_Neighbourhood_Value = 0.25f * (vh[indx - W1 - 1] + vh[indx - W1 + 1] + vh[indx + W1 - 1] + vh[indx + W1 + 1]);
    const float VH_Disc = (fabsf(0.5f - VH_Central_Value) < fabsf(0.5f - VH_Neighbourhood_Value)) ? VH_Neighbourhood_Value : VH_Central_Value;
    g[h] = intp(VH_Disc, H_Est, V_Est);
  }
  __syncthreads();

  // ---- steps 4.0 + 4.1 (rcd.c:444-463): PQ_Dir over the lpf buffer
#pragma unroll
  for(int it = 0; it < HALF_ITERS; it++)
  {
    const int h = tid + it * NT;
    const int row = h / HS, col = 2 * (h - row * HS) + ((row & 1) ? p1 : p0);
    if(row < 4 || row >= tileRows - 4 || col < 4 || col >= tileCols - 4) continue;
    const int indx = row * TS + col;
    const int indx3 = (indx - W1 - 1) / 2, indx4 = (indx + W1 - 1) / 2;
    const float P_Stat = fmaxf(EPSSQ, pq_hpf(cfa, indx3, tileRows, tileCols, W1 + 1) + pq_hpf(cfa, h, tileRows, tileCols, W1 + 1) + pq_hpf(cfa, indx4 + 1, tileRows, tileCols, W1 + 1));
    const float Q_Stat = fmaxf(EPSSQ, pq_hpf(cfa, indx3 + 1, tileRows, tileCols, W1 - 1) + pq_hpf(cfa, h, tileRows, tileCols, W1 - 1) + pq_hpf(cfa, indx4, tileRows, tileCols, W1 - 1));
    x[h] = P_Stat / (P_Stat + Q_Stat);
  }
  __syncthreads();

  // ---- step 4.2 (rcd.c:466-496): red at blue sites / blue at red sites, kept in registers
  //      until every thread has read its PQ_Dir neighbourhood, then stored over PQ_Dir
  float co[HALF_ITERS];
#pragma unroll
  for(int it = 0; it < HALF_ITERS; it++)
  {
    const int h = tid + it * NT;
    const int row = h / HS, col = 2 * (h - row * HS) + ((row & 1) ? p1 : p0);
    co[it] = 0.0f;
    if(row < 4 || row >= tileRows - 4 || col < 4 || col >= tileCols - 4) continue;
    const int indx = row * TS + col;
    const int pqindx2 = (indx - W1 - 1) / 2, pqindx3 = (indx + W1 - 1) / 2;
    const float PQ_Central_Value = x[h];
    const float PQ_Neighbourhood_Value = 0.25f * (x[pqindx2] + x[pqindx2 + 1] + x[pqindx3] + x[pqindx3 + 1]);
    const float PQ_Disc = (fabsf(0.5f - PQ_Central_Value) < fabsf(0.5f - PQ_Neighbourhood_Value)) ? PQ_Neighbourhood_Value : PQ_Central_Value;
    const float nw = cfa[indx - W1 - 1], ne = cfa[indx - W1 + 1], sw = cfa[indx + W1 - 1], se = cfa[indx + W1 + 1];
    const float g0 = g[h];
    const float NW_Grad = (float)((double)EPS + dabs(nw - se) + dabs(nw - cfa[indx - W3 - 3]) + dabs(g0 - g[(indx - W2 - 2) / 2]));
    const float NE_Grad = (float)((double)EPS + dabs(ne - sw) + dabs(ne - cfa[indx - W3 + 3]) + dabs(g0 - g[(indx - W2 + 2) / 2]));
    const float SW_Grad = (float)((double)EPS + dabs(ne - sw) + dabs(sw - cfa[indx + W3 - 3]) + dabs(g0 - g[(indx + W2 - 2) / 2]));
    const float SE_Grad = (float)((double)EPS + dabs(nw - se) + dabs(se - cfa[indx + W3 + 3]) + dabs(g0 - g[(indx + W2 + 2) / 2]));
    const float NW_Est = nw - g[(indx - W1 - 1) / 2];
    const float NE_Est = ne - g[(indx - W1 + 1) / 2];
    const float SW_Est = sw - g[(indx + W1 - 1) / 2];
    const float SE_Est = se - g[(indx + W1 + 1) / 2];
    const float P_Est = (NW_Grad * SE_Est + SE_Grad * NW_Est) / (NW_Grad + SE_Grad);
    const float Q_Est = (NE_Grad * SW_Est + SW_Grad * NE_Est) / (NE_Grad + SW_Grad);
    co[it] = g0 + intp(PQ_Disc, Q_Est, P_Est);
  }
  __syncthreads();
#pragma unroll
  for(int it = 0; it < HALF_ITERS; it++) x[tid + it * NT] = co[it];
  __syncthreads();

  // ---- output (rcd.c:539-555) with step 4.3 (rcd.c:499-536) evaluated at the green sites that
  //      are written out
  const int first_vertical = (tile_vertical == 0) ? RCD_MARGIN : RCD_BORDER;
  const int last_vertical = tileRows - ((tile_vertical == a.num_vertical - 1) ? RCD_MARGIN : RCD_BORDER);
  const int first_horizontal = (tile_horizontal == 0) ? RCD_MARGIN : RCD_BORDER;
  const int last_horizontal = tileCols - ((tile_horizontal == a.num_horizontal - 1) ? RCD_MARGIN : RCD_BORDER);
  const int orows = last_vertical - first_vertical, ocols = last_horizontal - first_horizontal;
  if(orows <= 0 || ocols <= 0) return;
  const float scaler = a.scaler;

  // One lane = one horizontally adjacent {red/blue site, green site} pair starting at an even
  // column: every lane does the same work (one cheap assembly, one step-4.3 evaluation) and a wave
  // stores 64 x 32 contiguous bytes -- full lines instead of two half-populated passes (the first
  // version wrote 1.68x the output bytes to HBM, profiles/r01_a_pmc_*).
  const int pbase = first_horizontal & ~1;
  const int npairs = ((last_horizontal - pbase) + 1) >> 1;
  const int nwork = orows * npairs;
  for(int s = tid; s < nwork; s += NT)
  {
    const int r = s / npairs, k = s - r * npairs;
    const int row = first_vertical + r;
    const int p = (row & 1) ? p1 : p0;      // column parity of the red/blue sites of this row
    const int c0 = pbase + 2 * k;
    const int col_rb = c0 + p, col_g = c0 + 1 - p;
    const bool ok_rb = col_rb >= first_horizontal && col_rb < last_horizontal;
    const bool ok_g = col_g >= first_horizontal && col_g < last_horizontal;
    float4 o_rb = make_float4(0.f, 0.f, 0.f, 0.f), o_g = make_float4(0.f, 0.f, 0.f, 0.f);
    if(ok_rb)
    {
      const int indx = row * TS + col_rb;
      const int f = fc(row, col_rb, filters); // 0 or 2
      const float native = scaler * fmaxf(0.0f, cfa[indx]);
      const float green = scaler * fmaxf(0.0f, g[indx >> 1]);
      const float other = scaler * fmaxf(0.0f, x[indx >> 1]);
      o_rb.x = (f == 0) ? native : other;
      o_rb.y = green;
      o_rb.z = (f == 0) ? other : native;
    }
    if(ok_g)
    {
      // step 4.3 (rcd.c:499-536) at this green site
      const int indx = row * TS + col_g;
      const float VH_Central_Value = vh[indx];
      const float VH_Neighbourhood_Value = 0.25f * (vh[indx - W1 - 1] + vh[indx - W1 + 1] + vh[indx + W1 - 1] + vh[indx + W1 + 1]);
      const float VH_Disc = (fabsf(0.5f - VH_Central_Value) < fabsf(0.5f - VH_Neighbourhood_Value)) ? VH_Neighbourhood_Value : VH_Central_Value;
      const float rgb1 = cfa[indx];
      const float N1 = (float)((double)EPS + dabs(rgb1 - cfa[indx - W2]));
      const float S1 = (float)((double)EPS + dabs(rgb1 - cfa[indx + W2]));
      const float W1g = (float)((double)EPS + dabs(rgb1 - cfa[indx - 2]));
      const float E1 = (float)((double)EPS + dabs(rgb1 - cfa[indx + 2]));
      const float rgb1mw1 = g[(indx - W1) >> 1], rgb1pw1 = g[(indx + W1) >> 1];
      const float rgb1m1 = g[(indx - 1) >> 1], rgb1p1 = g[(indx + 1) >> 1];
      // the row neighbours carry colour `ch` natively, the column neighbours the other one; the
      // non-native samples are step 4.2 results (x)
      const int ch = fc(row, col_g + 1, filters);
      float res[2];
#pragma unroll
      for(int ci = 0; ci < 2; ci++)
      {
        const int c = 2 * ci;
        const bool hn = (c == ch);
        const float cN1 = hn ? x[(indx - W1) >> 1] : cfa[indx - W1];
        const float cS1 = hn ? x[(indx + W1) >> 1] : cfa[indx + W1];
        const float cN3 = hn ? x[(indx - W3) >> 1] : cfa[indx - W3];
        const float cS3 = hn ? x[(indx + W3) >> 1] : cfa[indx + W3];
        const float cW1 = hn ? cfa[indx - 1] : x[(indx - 1) >> 1];
        const float cE1 = hn ? cfa[indx + 1] : x[(indx + 1) >> 1];
        const float cW3 = hn ? cfa[indx - 3] : x[(indx - 3) >> 1];
        const float cE3 = hn ? cfa[indx + 3] : x[(indx + 3) >> 1];
        const float SNabs = fabsf(cN1 - cS1);
        const float EWabs = fabsf(cW1 - cE1);
        const float N_Grad = (float)((double)(N1 + SNabs) + dabs(cN1 - cN3));
        const float S_Grad = (float)((double)(S1 + SNabs) + dabs(cS1 - cS3));
        const float W_Grad = (float)((double)(W1g + EWabs) + dabs(cW1 - cW3));
        const float E_Grad = (float)((double)(E1 + EWabs) + dabs(cE1 - cE3));
        const float N_Est = cN1 - rgb1mw1;
        const float S_Est = cS1 - rgb1pw1;
        const float W_Est = cW1 - rgb1m1;
        const float E_Est = cE1 - rgb1p1;
        const float V_Est = (N_Grad * S_Est + S_Grad * N_Est) / (N_Grad + S_Grad);
        const float H_Est = (E_Grad * W_Est + W_Grad * E_Est) / (E_Grad + W_Grad);
        res[ci] = rgb1 + intp(VH_Disc, H_Est, V_Est);
      }
      o_g.x = scaler * fmaxf(0.0f, res[0]);
      o_g.y = scaler * fmaxf(0.0f, rgb1);
      o_g.z = scaler * fmaxf(0.0f, res[1]);
    }
    const int frame_row = rowStart + row;
    float4 *dst = out + (size_t)(frame_row - a.out_row0) * a.width + colStart + c0;
    const float4 lo = p ? o_g : o_rb, hi = p ? o_rb : o_g; // column c0, column c0 + 1
    const bool in_band = frame_row >= a.out_row0 && frame_row < a.out_row1;
    const bool ok_lo = in_band && (p ? ok_g : ok_rb), ok_hi = in_band && (p ? ok_rb : ok_g);
    if(ok_lo) dst[0] = lo;
    if(ok_hi) dst[1] = hi;
  }
}
#endif // ANSEL_HIP_MEASURING

// ---- the tile kernel, second version ------------------------------------------------------------------------------
// Same steps, same arithmetic; what changed is where the samples sit and how a lane finds them.
//  * The two full-resolution arrays (cfa, VH_Dir) are stored DE-INTERLEAVED by column parity: element (row, col) at
//    [(col & 1) * PL + row * 56 + (col >> 1)].  Every step after 2.1 walks the red/blue (or green) sites of a row,
//    i.e. every other column: row-major that is a stride-2 walk = a 2-way bank conflict on every cfa / VH_Dir read
//    (SQ_LDS_BANK_CONFLICT: 37 % of the LDS cycles of v1); de-interleaved it is stride 1.  PL = 112 * 56 + 16 puts the
//    two planes half a bank row apart, so the full-resolution steps (consecutive columns = alternating planes) are
//    conflict-free too.
//  * A lane keeps ITS sites for the whole kernel: 896 = 8 * 112 = 16 * 56, so lane t walks column t % 112 of rows
//    t / 112 + 8 it in the full-resolution steps and site t % 56 of rows t / 56 + 16 it in the half-resolution ones --
//    same column, same row parity, same CFA colour every time.  Row / column decode, parity selects, rim tests on the
//    column and the four plane bases are computed once; inside the steps every neighbour is base + compile-time
//    offset.  The bases are biased by BIAS floats (and the arrays preceded by PAD) so that all offsets are >= 0 and fit
//    the 16-bit immediate of ds_read (v1 spent 24 % of its VALU instructions on integer index arithmetic).
//  * The reference's half-resolution indexing of the P/Q high-pass planes ((row * 112 + col) / 2 at odd columns only,
//    rcd.c:444-463) is resolved per lane parity q = 1 - p: the three P (Q) words a site reads were computed at
//    (row - 1, col - q), (row, col + q), (row + 1, col + 2 - q)  ((row - 1, col + 2 - q), (row, col + q), (row + 1, col - q));
//    only the word at col + 2 - q can lie outside what step 4.0 writes (then it is 0).
constexpr int PL = TS * HS + 16;
constexpr int PAD = 512;
constexpr int BIAS = 4 * HS + 4;
constexpr size_t LDS2_FLOATS = PAD + 4 * PL + 2 * TS * HS;
constexpr size_t LDS2_BYTES = sizeof(float) * LDS2_FLOATS;

// element (dy, dx) relative to the site the base pair (S = plane of the site's column parity, O = the other plane,
// shifted by the parity) belongs to; dy, dx compile-time
#define AT(S, O, dy, dx) (((dx) & 1) ? (O)[BIAS + (dy) * HS + ((dx) - 1) / 2] : (S)[BIAS + (dy) * HS + (dx) / 2])
// hpf() of the reference around (cy, cx) along (sy, sx)
#define HPF(S, O, cy, cx, sy, sx)                                                                                          \
  sqf((AT(S, O, (cy) - 3 * (sy), (cx) - 3 * (sx)) - AT(S, O, (cy) - (sy), (cx) - (sx)) - AT(S, O, (cy) + (sy), (cx) + (sx))   \
       + AT(S, O, (cy) + 3 * (sy), (cx) + 3 * (sx)))                                                                        \
      - 3.0f * (AT(S, O, (cy) - 2 * (sy), (cx) - 2 * (sx)) + AT(S, O, (cy) + 2 * (sy), (cx) + 2 * (sx))) + 6.0f * AT(S, O, cy, cx))

__global__ __launch_bounds__(NT) void rcd_tiles(const float *__restrict__ in, float4 *__restrict__ out, const rcd_args a)
{
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *const cfa = lds + PAD;
  float *const vh = cfa + 2 * PL;
  float *const g = vh + 2 * PL;
  float *const x = g + TS * HS;

  const int tid = threadIdx.x;
  const int tile_vertical_local = blockIdx.x / a.num_horizontal;
  const int tile_horizontal = blockIdx.x - tile_vertical_local * a.num_horizontal;
  const int tile_vertical = a.tv0 + tile_vertical_local;
  const int rowStart = tile_vertical * TV, rowEnd = min(rowStart + TS, a.height);
  const int colStart = tile_horizontal * TV, colEnd = min(colStart + TS, a.width);
  const int tileRows = rowEnd - rowStart, tileCols = colEnd - colStart;
  const uint32_t filters = a.filters;
  const int p0 = fc(0, 0, filters) & 1, p1 = fc(1, 0, filters) & 1;

  // the lane's column of the full-resolution walk (rows frow0 + 8 it)
  const int frow0 = tid / TS, fcol = tid - frow0 * TS;
  const int fpar = fcol & 1, fhx = fcol >> 1;
  const int fsite = frow0 * HS + fhx; // + it * 8 * HS
  // the lane's site of the half-resolution walks (rows hrow0 + 16 it): red/blue, column parity p
  const int hrow0 = tid / HS, hc = tid - hrow0 * HS;
  const int p = (hrow0 & 1) ? p1 : p0, q = 1 - p;
  const int hcol = 2 * hc + p;
  const int hsite = hrow0 * HS + hc; // = h of iteration 0; + it * 16 * HS
  const bool hcol4 = hcol >= 4 && hcol < tileCols - 4;

  // ---- step 0 (rcd.c:345-354): cfa = fmaxf(0, in) * revscaler, zero outside a partial tile.  All fourteen loads are
  //      issued before the first is used (clamped addresses, the select comes after): the workgroup is alone on its CU,
  //      nobody else hides a load's latency
  {
    const int ccol = min(fcol, tileCols - 1);
    const float *const src = in + (size_t)(rowStart - a.in_row0) * a.width + colStart + ccol;
    float *const dst = cfa + fpar * PL + fsite;
    float raw[FULL_ITERS];
#pragma unroll
    for(int it = 0; it < FULL_ITERS; it++) raw[it] = src[(size_t)min(frow0 + 8 * it, tileRows - 1) * a.width];
#pragma unroll
    for(int it = 0; it < FULL_ITERS; it++)
    {
      const int row = frow0 + 8 * it;
      const float v = fmaxf(0.0f, raw[it]) * a.revscaler;
      dst[it * 8 * HS] = (row < tileRows && fcol < tileCols) ? v : 0.0f;
    }
  }
  __syncthreads();

  // ---- step 1 (rcd.c:356-390) VH_Dir, 0 on the rim; step 2.1 (rcd.c:394-402) lpf at red/blue sites; rgb[1] starts as
  //      cfa (rcd.c:352)
  {
    const float *const cS = cfa + fpar * PL + fsite - BIAS, *const cO = cfa + (1 - fpar) * PL + fsite + fpar - BIAS;
    float *const dst = vh + fpar * PL + fsite;
    const bool col_in = fcol >= 4 && fcol < tileCols - 4;
#pragma unroll
    for(int it = 0; it < FULL_ITERS; it++)
    {
      const int row = frow0 + 8 * it;
      const float *const S = cS + it * 8 * HS, *const O = cO + it * 8 * HS;
      float v = 0.0f;
      if(col_in && row >= 4 && row < tileRows - 4)
      {
        const float V_Stat = fmaxf(EPSSQ, HPF(S, O, -1, 0, 1, 0) + HPF(S, O, 0, 0, 1, 0) + HPF(S, O, 1, 0, 1, 0));
        const float H_Stat = fmaxf(EPSSQ, HPF(S, O, 0, -1, 0, 1) + HPF(S, O, 0, 0, 0, 1) + HPF(S, O, 0, 1, 0, 1));
        v = V_Stat / (V_Stat + H_Stat);
      }
      dst[it * 8 * HS] = v;
    }
  }
  // the red/blue site's plane bases: cfa, VH_Dir
  const float *const sS = cfa + p * PL + hsite - BIAS, *const sO = cfa + q * PL + hsite + p - BIAS;
  const float *const vS = vh + p * PL + hsite - BIAS, *const vO = vh + q * PL + hsite + p - BIAS;
  {
    const bool col_in = hcol >= 2 && hcol < tileCols - 2;
#pragma unroll
    for(int it = 0; it < HALF_ITERS; it++)
    {
      const int row = hrow0 + 16 * it;
      const float *const S = sS + it * 16 * HS, *const O = sO + it * 16 * HS;
      float lp = 0.0f;
      const float c = AT(S, O, 0, 0);
      if(col_in && row >= 2 && row < tileRows - 2)
        lp = c + 0.5f * (AT(S, O, -1, 0) + AT(S, O, 1, 0) + AT(S, O, 0, -1) + AT(S, O, 0, 1))
             + 0.25f * (AT(S, O, -1, -1) + AT(S, O, -1, 1) + AT(S, O, 1, -1) + AT(S, O, 1, 1));
      x[hsite + it * 16 * HS] = lp;
      g[hsite + it * 16 * HS] = c;
    }
  }
  __syncthreads();

  // ---- step 3.1 (rcd.c:406-440): green at red/blue sites
#pragma unroll
  for(int it = 0; it < HALF_ITERS; it++)
  {
    const int row = hrow0 + 16 * it;
    if(!hcol4 || row < 4 || row >= tileRows - 4) continue;
    const float *const S = sS + it * 16 * HS, *const O = sO + it * 16 * HS;
    const float *const VS = vS + it * 16 * HS, *const VO = vO + it * 16 * HS;
    const float *const xh = x + hsite + it * 16 * HS;
    const float cfai = AT(S, O, 0, 0);
    const float cN1 = AT(S, O, -1, 0), cS1 = AT(S, O, 1, 0), cW1 = AT(S, O, 0, -1), cE1 = AT(S, O, 0, 1);
    const float cN2 = AT(S, O, -2, 0), cS2 = AT(S, O, 2, 0), cW2 = AT(S, O, 0, -2), cE2 = AT(S, O, 0, 2);
    const float N_Grad = (float)((double)EPS + dabs(cN1 - cS1) + dabs(cfai - cN2) + dabs(cN1 - AT(S, O, -3, 0)) + dabs(cN2 - AT(S, O, -4, 0)));
    const float S_Grad = (float)((double)EPS + dabs(cN1 - cS1) + dabs(cfai - cS2) + dabs(cS1 - AT(S, O, 3, 0)) + dabs(cS2 - AT(S, O, 4, 0)));
    const float W_Grad = (float)((double)EPS + dabs(cW1 - cE1) + dabs(cfai - cW2) + dabs(cW1 - AT(S, O, 0, -3)) + dabs(cW2 - AT(S, O, 0, -4)));
    const float E_Grad = (float)((double)EPS + dabs(cW1 - cE1) + dabs(cfai - cE2) + dabs(cE1 - AT(S, O, 0, 3)) + dabs(cE2 - AT(S, O, 0, 4)));
    const float lpfi = xh[0];
    const float N_Est = cN1 * (lpfi + lpfi) / (EPS + lpfi + xh[-W1]);
    const float S_Est = cS1 * (lpfi + lpfi) / (EPS + lpfi + xh[W1]);
    const float W_Est = cW1 * (lpfi + lpfi) / (EPS + lpfi + xh[-1]);
    const float E_Est = cE1 * (lpfi + lpfi) / (EPS + lpfi + xh[1]);
    const float V_Est = (S_Grad * N_Est + N_Grad * S_Est) / (N_Grad + S_Grad);
    const float H_Est = (W_Grad * E_Est + E_Grad * W_Est) / (E_Grad + W_Grad);
    const float VH_Central_Value = AT(VS, VO, 0, 0);
    const float VH_Neighbourhood_Value = 0.25f * (AT(VS, VO, -1, -1) + AT(VS, VO, -1, 1) + AT(VS, VO, 1, -1) + AT(VS, VO, 1, 1));
    const float VH_Disc = (fabsf(0.5f - VH_Central_Value) < fabsf(0.5f - VH_Neighbourhood_Value)) ? VH_Neighbourhood_Value : VH_Central_Value;
    g[hsite + it * 16 * HS] = intp(VH_Disc, H_Est, V_Est);
  }
  __syncthreads();

  // ---- steps 4.0 + 4.1 (rcd.c:444-463): PQ_Dir over the lpf buffer.  The six high-pass words sit at odd columns:
  //      site 1 = (row, hcol - q), site 2 = (row, hcol + q) -- plane 1 is theirs, plane 0 the other
  {
    const float *const aS = cfa + PL + hsite - q - BIAS, *const aO = cfa + hsite - q + 1 - BIAS;
    const float *const bS = cfa + PL + hsite - BIAS, *const bO = cfa + hsite + 1 - BIAS;
    const bool far_in = hcol + 2 - q < tileCols - 3; // the word at column hcol + 2 - q is one step 4.0 writes
#pragma unroll
    for(int it = 0; it < HALF_ITERS; it++)
    {
      const int row = hrow0 + 16 * it;
      if(!hcol4 || row < 4 || row >= tileRows - 4) continue;
      const float *const A = aS + it * 16 * HS, *const Ao = aO + it * 16 * HS;
      const float *const B = bS + it * 16 * HS, *const Bo = bO + it * 16 * HS;
      const float p_far = far_in ? HPF(A, Ao, 1, 2, 1, 1) : 0.0f, q_far = far_in ? HPF(A, Ao, -1, 2, 1, -1) : 0.0f;
      const float P_Stat = fmaxf(EPSSQ, HPF(A, Ao, -1, 0, 1, 1) + HPF(B, Bo, 0, 0, 1, 1) + p_far);
      const float Q_Stat = fmaxf(EPSSQ, q_far + HPF(B, Bo, 0, 0, 1, -1) + HPF(A, Ao, 1, 0, 1, -1));
      x[hsite + it * 16 * HS] = P_Stat / (P_Stat + Q_Stat);
    }
  }
  __syncthreads();

  // ---- step 4.2 (rcd.c:466-496): red at blue sites / blue at red sites.  The result goes into the site's own word of
  //      VH_Dir: steps 3.1 (behind two barriers) was the last to read VH_Dir at red/blue sites -- the output stage reads
  //      it at green sites and their diagonals, green too -- so nobody waits before storing (the first version kept the
  //      results in registers behind one more barrier and stored them over PQ_Dir)
  float *const codst = vh + p * PL + hsite;
#pragma unroll
  for(int it = 0; it < HALF_ITERS; it++)
  {
    const int row = hrow0 + 16 * it;
    if(!hcol4 || row < 4 || row >= tileRows - 4)
    {
      codst[it * 16 * HS] = 0.0f;
      continue;
    }
    const float *const S = sS + it * 16 * HS, *const O = sO + it * 16 * HS;
    const float *const xq = x + hsite - q + it * 16 * HS, *const gq = g + hsite - q + it * 16 * HS;
    const float *const gh = g + hsite + it * 16 * HS;
    const float PQ_Central_Value = xq[q];
    const float PQ_Neighbourhood_Value = 0.25f * (xq[-HS] + xq[-HS + 1] + xq[HS] + xq[HS + 1]);
    const float PQ_Disc = (fabsf(0.5f - PQ_Central_Value) < fabsf(0.5f - PQ_Neighbourhood_Value)) ? PQ_Neighbourhood_Value : PQ_Central_Value;
    const float nw = AT(S, O, -1, -1), ne = AT(S, O, -1, 1), sw = AT(S, O, 1, -1), se = AT(S, O, 1, 1);
    const float g0 = gh[0];
    const float NW_Grad = (float)((double)EPS + dabs(nw - se) + dabs(nw - AT(S, O, -3, -3)) + dabs(g0 - gh[-2 * HS - 1]));
    const float NE_Grad = (float)((double)EPS + dabs(ne - sw) + dabs(ne - AT(S, O, -3, 3)) + dabs(g0 - gh[-2 * HS + 1]));
    const float SW_Grad = (float)((double)EPS + dabs(ne - sw) + dabs(sw - AT(S, O, 3, -3)) + dabs(g0 - gh[2 * HS - 1]));
    const float SE_Grad = (float)((double)EPS + dabs(nw - se) + dabs(se - AT(S, O, 3, 3)) + dabs(g0 - gh[2 * HS + 1]));
    const float NW_Est = nw - gq[-HS];
    const float NE_Est = ne - gq[-HS + 1];
    const float SW_Est = sw - gq[HS];
    const float SE_Est = se - gq[HS + 1];
    const float P_Est = (NW_Grad * SE_Est + SE_Grad * NW_Est) / (NW_Grad + SE_Grad);
    const float Q_Est = (NE_Grad * SW_Est + SW_Grad * NE_Est) / (NE_Grad + SW_Grad);
    codst[it * 16 * HS] = g0 + intp(PQ_Disc, Q_Est, P_Est);
  }
  __syncthreads();

  // ---- output (rcd.c:539-555) with step 4.3 (rcd.c:499-536) evaluated at the green sites that are written out
  const int first_vertical = (tile_vertical == 0) ? RCD_MARGIN : RCD_BORDER;
  const int last_vertical = tileRows - ((tile_vertical == a.num_vertical - 1) ? RCD_MARGIN : RCD_BORDER);
  const int first_horizontal = (tile_horizontal == 0) ? RCD_MARGIN : RCD_BORDER;
  const int last_horizontal = tileCols - ((tile_horizontal == a.num_horizontal - 1) ? RCD_MARGIN : RCD_BORDER);
  const int orows = last_vertical - first_vertical, ocols = last_horizontal - first_horizontal;
  if(orows <= 0 || ocols <= 0) return;
  const float scaler = a.scaler;

  // One lane = one horizontally adjacent {red/blue site, green site} pair starting at an even column: every lane does
  // the same work (one cheap assembly, one step-4.3 evaluation) and a wave stores 64 x 32 contiguous bytes
  const int pbase = first_horizontal & ~1;
  const int npairs = ((last_horizontal - pbase) + 1) >> 1;
  const int nwork = orows * npairs;
  const unsigned magic = 0xffffffffu / (unsigned)npairs + 1u; // s / npairs == umulhi(s, magic) for s < 2^32 / npairs
  for(int s = tid; s < nwork; s += NT)
  {
    const int r = (int)__umulhi((unsigned)s, magic), k = s - r * npairs;
    const int row = first_vertical + r;
    const int pr = (row & 1) ? p1 : p0;     // column parity of the red/blue sites of this row
    const int c0 = pbase + 2 * k;
    const int col_rb = c0 + pr, col_g = c0 + 1 - pr;
    const bool ok_rb = col_rb >= first_horizontal && col_rb < last_horizontal;
    const bool ok_g = col_g >= first_horizontal && col_g < last_horizontal;
    const int hh = row * HS + (c0 >> 1);    // both sites of the pair sit at column c0 / 2 of their planes
    float4 o_rb = make_float4(0.f, 0.f, 0.f, 0.f), o_g = make_float4(0.f, 0.f, 0.f, 0.f);
    if(ok_rb)
    {
      const int f = fc(row, col_rb, filters); // 0 or 2
      const float native = scaler * fmaxf(0.0f, cfa[pr * PL + hh]);
      const float green = scaler * fmaxf(0.0f, g[hh]);
      const float other = scaler * fmaxf(0.0f, vh[pr * PL + hh]);
      o_rb.x = (f == 0) ? native : other;
      o_rb.y = green;
      o_rb.z = (f == 0) ? other : native;
    }
    if(ok_g)
    {
      // step 4.3 (rcd.c:499-536) at this green site: column parity e = 1 - pr
      const int e = 1 - pr;
      const float *const S = cfa + e * PL + hh - BIAS, *const O = cfa + pr * PL + hh + e - BIAS;
      const float *const VS = vh + e * PL + hh - BIAS, *const VO = vh + pr * PL + hh + e - BIAS;
      // the red/blue sites left and right: [-1], [0]; +-3: [-2], [1]; above and below: [-+ HS], [-+ 3 HS].  Their step-4.2
      // colour sits in VH_Dir's words of those sites: plane pr in this row, plane e in the rows above and below
      const float *const ge = g + hh + e, *const xe = vh + pr * PL + hh + e;
      const float *const gv = g + hh, *const xv = vh + e * PL + hh;
      const float VH_Central_Value = AT(VS, VO, 0, 0);
      const float VH_Neighbourhood_Value = 0.25f * (AT(VS, VO, -1, -1) + AT(VS, VO, -1, 1) + AT(VS, VO, 1, -1) + AT(VS, VO, 1, 1));
      const float VH_Disc = (fabsf(0.5f - VH_Central_Value) < fabsf(0.5f - VH_Neighbourhood_Value)) ? VH_Neighbourhood_Value : VH_Central_Value;
      const float rgb1 = AT(S, O, 0, 0);
      const float N1 = (float)((double)EPS + dabs(rgb1 - AT(S, O, -2, 0)));
      const float S1 = (float)((double)EPS + dabs(rgb1 - AT(S, O, 2, 0)));
      const float W1g = (float)((double)EPS + dabs(rgb1 - AT(S, O, 0, -2)));
      const float E1 = (float)((double)EPS + dabs(rgb1 - AT(S, O, 0, 2)));
      const float rgb1mw1 = gv[-HS], rgb1pw1 = gv[HS];
      const float rgb1m1 = ge[-1], rgb1p1 = ge[0];
      // the row neighbours carry colour `ch` natively, the column neighbours the other one; the non-native samples are
      // step 4.2 results (x).  res_h: colour ch (row neighbours from cfa), res_v: the other (column neighbours from cfa)
      const int ch = fc(row, col_g + 1, filters);
      float res_h = 0.0f, res_v = 0.0f;
#pragma unroll
      for(int hn = 1; hn >= 0; hn--)
      {
        const float cN1 = hn ? xv[-HS] : AT(S, O, -1, 0);
        const float cS1 = hn ? xv[HS] : AT(S, O, 1, 0);
        const float cN3 = hn ? xv[-3 * HS] : AT(S, O, -3, 0);
        const float cS3 = hn ? xv[3 * HS] : AT(S, O, 3, 0);
        const float cW1 = hn ? AT(S, O, 0, -1) : xe[-1];
        const float cE1 = hn ? AT(S, O, 0, 1) : xe[0];
        const float cW3 = hn ? AT(S, O, 0, -3) : xe[-2];
        const float cE3 = hn ? AT(S, O, 0, 3) : xe[1];
        const float SNabs = fabsf(cN1 - cS1);
        const float EWabs = fabsf(cW1 - cE1);
        const float N_Grad = (float)((double)(N1 + SNabs) + dabs(cN1 - cN3));
        const float S_Grad = (float)((double)(S1 + SNabs) + dabs(cS1 - cS3));
        const float W_Grad = (float)((double)(W1g + EWabs) + dabs(cW1 - cW3));
        const float E_Grad = (float)((double)(E1 + EWabs) + dabs(cE1 - cE3));
        const float N_Est = cN1 - rgb1mw1;
        const float S_Est = cS1 - rgb1pw1;
        const float W_Est = cW1 - rgb1m1;
        const float E_Est = cE1 - rgb1p1;
        const float V_Est = (N_Grad * S_Est + S_Grad * N_Est) / (N_Grad + S_Grad);
        const float H_Est = (E_Grad * W_Est + W_Grad * E_Est) / (E_Grad + W_Grad);
        const float res = rgb1 + intp(VH_Disc, H_Est, V_Est);
        if(hn) res_h = res;
        else res_v = res;
      }
      o_g.x = scaler * fmaxf(0.0f, ch == 0 ? res_h : res_v);
      o_g.y = scaler * fmaxf(0.0f, rgb1);
      o_g.z = scaler * fmaxf(0.0f, ch == 0 ? res_v : res_h);
    }
    const int frame_row = rowStart + row;
    float4 *dst = out + (size_t)(frame_row - a.out_row0) * a.width + colStart + c0;
    const float4 lo = pr ? o_g : o_rb, hi = pr ? o_rb : o_g; // column c0, column c0 + 1
    const bool in_band = frame_row >= a.out_row0 && frame_row < a.out_row1;
    const bool ok_lo = in_band && (pr ? ok_g : ok_rb), ok_hi = in_band && (pr ? ok_rb : ok_g);
    if(ok_lo) dst[0] = lo;
    if(ok_hi) dst[1] = hi;
  }
}
#undef AT
#undef HPF

// ---- border ring: rcd_ppg_border(), rcd.c:92-272, through ppg_device.h (clamped samples) ----
__global__ __launch_bounds__(256) void rcd_border(const float *__restrict__ in, float4 *__restrict__ out,
                                                   const int width, const int height, const uint32_t filters,
                                                   const int in_row0, const int in_rows, const int out_row0,
                                                   const int out_row1)
{
  // enumerate the ring of RCD_MARGIN pixels: top rows, bottom rows, then left/right columns
  const int M = RCD_MARGIN;
  const long top = (long)M * width, side = (long)(height - 2 * M) * 2 * M;
  const long total = 2 * top + (side > 0 ? side : 0);
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if(t >= total) return;
  int j, i;
  if(t < top)
  {
    j = (int)(t / width);
    i = (int)(t - (long)j * width);
  }
  else if(t < 2 * top)
  {
    const long u = t - top;
    j = height - M + (int)(u / width);
    i = (int)(u % width);
  }
  else
  {
    const long u = t - 2 * top;
    j = M + (int)(u / (2 * M));
    const int q = (int)(u % (2 * M));
    i = q < M ? q : width - 2 * M + q;
  }
  if(j < 0 || j >= height || i < 0 || i >= width) return;
  if(j < out_row0 || j >= out_row1) return;
  // band mode: the ring pixel reads frame rows j-4..j+4, all inside the band's halo (>= 9 rows) or
  // outside the frame (never dereferenced); address the band buffer as if it were the whole frame
  const float *frame = in - (ptrdiff_t)in_row0 * width;
  const ppg_ctx k = { frame, width, height, width, height, 0, 0, filters, nullptr };
  (void)in_rows;
  const float4 v = ppg_pixel<true>(k, j, i);
  float4 *const o = out + (size_t)(j - out_row0) * width + i;
  if(ring_lt(k, j, i, 3))
  {
    // first pass of rcd_ppg_border() (rcd.c:96-127): three channels stored, the fourth is the caller's
    o->x = v.x;
    o->y = v.y;
    o->z = v.z;
  }
  else
    *o = v;
}

} // namespace

// FC-based shift of the dcraw filter word: dt_rawspeed_crop_dcraw_filters()
// (src/imageio/imageio_rawspeed.cc:146-151 -> rawspeed ColorFilterArray::shiftDcrawFilter)
extern "C" uint32_t dt_hip_crop_dcraw_filters(uint32_t filters, uint32_t crop_x, uint32_t crop_y)
{
  if(!filters || filters == 9u) return filters;
  uint32_t out = 0;
  for(int r = 0; r < 8; r++)
    for(int c = 0; c < 2; c++)
    {
      const int row = r + crop_y, col = c + crop_x;
      const uint32_t colour = filters >> ((((row << 1) & 14) + (col & 1)) << 1) & 3;
      out |= colour << ((((r << 1) & 14) + (c & 1)) << 1);
    }
  return out;
}

namespace ansel
{

// band == nullptr: the whole frame.  Otherwise piece describes the whole frame, `in` holds frame rows
// [in_row0, in_row0 + in_rows) and `out` frame rows [out_row0, out_row0 + out_rows); the launch runs
// the frame's tile rows [tv0, tv1) and the part of the border ring inside the output rows.
int rcd_demosaic_launch(int devid, const dt_hip_piece_t *piece, uint32_t filters, const float *in, float4 *out,
                        const rcd_band_t *band)
{
  const int width = piece->roi_in.width, height = piece->roi_in.height;
  if(width < 16 || height < 16) return DT_HIP_SUCCESS; // rcd.c:280-284: "too small area", output untouched
  const int in_row0 = band ? band->in_row0 : 0, in_rows = band ? band->in_rows : height;
  const int out_row0 = band ? band->out_row0 : 0, out_row1 = band ? band->out_row0 + band->out_rows : height;
  // rows alternate R/G and G/B with period 2 for every Bayer filter word rawspeed produces;
  // the CPU code already relies on it by mixing image-row and tile-row FC() calls
  hipStream_t s = stream_of(devid);
  {
    const long ring = 2L * RCD_MARGIN * width + 2L * RCD_MARGIN * (height > 2 * RCD_MARGIN ? height - 2 * RCD_MARGIN : 0);
    launch_scope ls(devid, "rcd_border");
    rcd_border<<<(unsigned)((ring + 255) / 256), 256, 0, s>>>(in, out, width, height, filters, in_row0, in_rows,
                                                              out_row0, out_row1);
  }
  rcd_args a;
  a.width = width;
  a.height = height;
  a.filters = filters;
  a.scaler = fmaxf(piece->processed_maximum[0], fmaxf(piece->processed_maximum[1], piece->processed_maximum[2]));
  a.revscaler = 1.0f / a.scaler;
  a.num_vertical = 1 + (height - 2 * RCD_BORDER - 1) / TV;
  a.num_horizontal = 1 + (width - 2 * RCD_BORDER - 1) / TV;
  a.tv0 = band ? band->tv0 : 0;
  a.in_row0 = in_row0;
  a.out_row0 = out_row0;
  a.out_row1 = out_row1;
  const int tile_rows = band ? band->tv1 - band->tv0 : a.num_vertical;
  if(band && (band->tv0 < 0 || band->tv1 > a.num_vertical || tile_rows <= 0 || band->tv0 * TV < in_row0
              || min(band->tv0 * TV + (tile_rows - 1) * TV + TS, height) > in_row0 + in_rows))
  {
    set_last_error("rcd band: tile rows [%d,%d) need frame rows outside the band buffer", band->tv0, band->tv1);
    return DT_HIP_INVALID_ARG;
  }
#ifdef ANSEL_HIP_MEASURING
  static const bool v1 = getenv("ANSEL_HIP_RCD_V1") != nullptr; // the first tile kernel, for A/B timing
#endif
  // the opt-in to more than 64 KB of LDS is per device
  static std::atomic<unsigned long long> attr_set{ 0ull };
  const int hip_dev = hip_device_of(devid);
  if(hip_dev < 0 || hip_dev >= 64) return DT_HIP_INVALID_ARG;
  if(!(attr_set.load() >> hip_dev & 1ull))
  {
    ANSEL_HIP_CHECK(hipFuncSetAttribute((const void *)rcd_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS2_BYTES));
#ifdef ANSEL_HIP_MEASURING
    ANSEL_HIP_CHECK(hipFuncSetAttribute((const void *)rcd_tiles_v1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
#endif
    attr_set.fetch_or(1ull << hip_dev);
  }
  {
    launch_scope ls(devid, "rcd_tiles");
#ifdef ANSEL_HIP_MEASURING
    if(v1) rcd_tiles_v1<<<(unsigned)(tile_rows * a.num_horizontal), NT, LDS_BYTES, s>>>(in, out, a);
    else
#endif
      rcd_tiles<<<(unsigned)(tile_rows * a.num_horizontal), NT, LDS2_BYTES, s>>>(in, out, a);
  }
  return check_launch("rcd_tiles");
}

} // namespace ansel
