// nlmeans.hip -- the non-local-means core on gfx950, for denoise (non-local means) and for denoise
// (profiled) in non-local-means mode.
//
// Reference: nlmeans_denoise(), src/pixel/nlmeans_core.c:315-532 (the OpenCL variant
// nlmeans_denoise_cl() :560-808 computes the patch distances with separable box sums in another
// order; the pin here is the CPU path, the one an export without OpenCL runs).
//
// The CPU algorithm walks the frame in chunks of about 72 x 60 pixels (a pure function of the frame
// size) and, per chunk and patch offset, slides a column sum down the rows and a row sum across the
// columns, both as binary32 recurrences.  Bit parity therefore requires the same chunk grid and the
// same recurrences; the parallelism left is across chunks, across the columns of the column-sum
// recurrence, across the rows of the row-sum recurrence, and across pixels everywhere else.
//
// One workgroup (1024 threads) = one chunk.  The chunk's input window (chunk + patch radius + largest
// patch shift on every side, RGB planes) is staged in LDS once when it fits next to the two tables
// (it does for every default: 94 KB at P 2 / K 7), so all four steps below read LDS, not L2.  Per patch offset:
//   A1  every (row, column) difference term of the column-sum recurrence, all threads, from global
//   A2  the column-sum recurrence, one thread per column, down the rows of an LDS table
//   B   the row-sum ("distortion") recurrence, one thread per row, across the columns of that table
//   C   weight = 2^-f(distortion) and out += weight * shifted pixel, all threads, coalesced;
//       accumulators stay in registers for the whole chunk (patch order = the reference's)
// and at the end the normalisation / luma-chroma blend of nlmeans_core.c:490-521.
#include "hip_common.h"
#include "nlmeans_core_params.h"
#include "nlm2_body.h"
#include "ieee_inrange.h"
#include "px_bilat.h"
#include "nlm3_body.h"
#include "nlm_tail_body.h"

#include <math.h>
#include <algorithm>
#include <vector>
#include <type_traits>

using namespace ansel;

namespace
{

#define SLICE_WIDTH 72  // nlmeans_core.c:55
#define SLICE_HEIGHT 60 // nlmeans_core.c:56
#define NLM_THREADS 1024
#define MAX_PX_PER_THREAD 5 // ceil(72 * 69 / NLM_THREADS)
#define CS_LANES 128         // table columns are walked 128 lanes at a time (chk_w + 2 * 16 + 1 <= 105)

struct nlm_args
{
  int W, H;
  int chk_w, chk_h, nchx;
  int radius, npatch;
  int cs_pitch, wt_pitch;
  float sharpness, center_weight, inv_den; // inv_den unused; the division is kept as in the reference
  float cpn;                               // centre pixel norm
  float norm[3];
  float luma, chroma;
  int skip_blend;
  int reach;     // patch radius + largest |shift|: rows/columns of input around the chunk a patch can touch
  int win_pitch; // staged window: (chk_h + 2 reach) rows x win_pitch columns per colour plane
  // row bands (hip_common.h band_view_t): the launch covers chunk rows cy0.. of the FRAME's grid and
  // stores frame rows [out_row0, out_row1) only; `in` / `out` are addressed with frame row indices
  int cy0, out_row0, out_row1;
  int variant;       // nlm_chunks_v2: 0, or the A/B switches of nlm2_body.h (ANSEL_NLM2_VARIANT; timing experiments)
  // the lightness cell of every output pixel for the bilateral grid behind the module (nlm_core_params_t::cell_out), or nullptr
  float2 *zc;
  float zc_sigma_r;
  int zc_size_z;
};

// the epilogue's extra store: bilat_zcells() of the pixel just written (same operands, same function: px_bilat.h)
__device__ __forceinline__ void nlm_store_cell(const nlm_args &a, const long o, const float L)
{
  if(a.zc)
  {
    float zf;
    const int zi = bilat_axis(L, a.zc_sigma_r, a.zc_size_z, zf);
    a.zc[o] = make_float2(zf, __int_as_float(zi));
  }
}

__device__ __forceinline__ int imin(const int a, const int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(const int a, const int b) { return a > b ? a : b; }

// float -> int the way the reference's target converts (cvttss2si): out of range and NaN -> INT_MIN
__device__ __forceinline__ int cvtt(const float v)
{
  return (v >= -2147483648.0f && v < 2147483648.0f) ? (int)v : (int)0x80000000;
}

// dt_fast_mexp2f(), src/math/math.h:290-301
__device__ __forceinline__ float mexp2(const float x)
{
  const int k0 = (int)(0x3f800000u + (unsigned)cvtt(x * -8388608.0f));
  return __int_as_float(k0 >= 0x800000 ? k0 : 0);
}

// pixel_difference(), nlmeans_core.c:160-170
__device__ __forceinline__ float pixdiff(const float4 a, const float4 b, const float n0, const float n1, const float n2)
{
  const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  return dx * dx * n0 + dy * dy * n1 + dz * dz * n2;
}

// diff_of_pixels_diff(), nlmeans_core.c:172-184
__device__ __forceinline__ float pixdiff2(const float4 a, const float4 b, const float4 c, const float4 d, const float n0,
                                          const float n1, const float n2)
{
  const float ax = a.x - b.x, ay = a.y - b.y, az = a.z - b.z;
  const float cx = c.x - d.x, cy = c.y - d.y, cz = c.z - d.z;
  return (ax * ax - cx * cx) * n0 + (ay * ay - cy * cy) * n1 + (az * az - cz * cz) * n2;
}

// pixel (r, c) of the frame: from the staged LDS window or from global memory
template <bool STAGED> struct pixel_source
{
  const float4 *in;
  const float *wr, *wg, *wb; // window planes, indexable by (r - r0) * pitch + (c - c0)
  int W, r0, c0, pitch;
  __device__ __forceinline__ float4 operator()(const int r, const int c) const
  {
    if(STAGED)
    {
      const int i = (r - r0) * pitch + (c - c0);
      return make_float4(wr[i], wg[i], wb[i], 0.0f);
    }
    return in[(long)r * W + c];
  }
};

template <bool STAGED>
__global__ __launch_bounds__(NLM_THREADS) void nlm_chunks(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                          const nlm_args a, const int2 *__restrict__ patches)
{
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int cy_launch = blockIdx.x / a.nchx, cx = blockIdx.x - cy_launch * a.nchx;
  const int cy = cy_launch + a.cy0;
  const int top = cy * a.chk_h, left = cx * a.chk_w;
  const int bot = imin(top + a.chk_h, a.H), right = imin(left + a.chk_w, a.W);
  const int ch = bot - top, cw = right - left;
  const int P = a.radius, W = a.W, H = a.H;
  const int csw = a.chk_w + 2 * P + 1; // table columns: frame columns left - P - 1 .. left + chk_w + P - 1
  const int cs0 = left - P - 1;
  float *const cs = lds;                         // [chk_h][cs_pitch]
  float *const dist = lds + (a.chk_h + 16) * a.cs_pitch; // [chk_h][wt_pitch], 16 spare table rows before it
  const float n0 = a.norm[0], n1 = a.norm[1], n2 = a.norm[2];

  pixel_source<STAGED> px;
  px.in = in;
  px.W = W;
  px.r0 = top - a.reach;
  px.c0 = left - a.reach;
  px.pitch = a.win_pitch;
  if(STAGED)
  {
    float *const win = dist + a.chk_h * a.wt_pitch;
    const int wh = a.chk_h + 2 * a.reach, plane = wh * a.win_pitch;
    px.wr = win;
    px.wg = win + plane;
    px.wb = win + 2 * plane;
    for(int i = tid; i < plane; i += NLM_THREADS)
    {
      const int wy = i / a.win_pitch, wx = i - wy * a.win_pitch;
      const int r = px.r0 + wy, c = px.c0 + wx;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if(r >= 0 && r < H && c >= 0 && c < W) v = in[(long)r * W + c];
      win[i] = v.x;
      win[i + plane] = v.y;
      win[i + 2 * plane] = v.z;
    }
    __syncthreads();
  }

  // the pixels this thread accumulates for the whole chunk
  float4 acc[MAX_PX_PER_THREAD];
  int rc[MAX_PX_PER_THREAD];
#pragma unroll
  for(int k = 0; k < MAX_PX_PER_THREAD; k++)
  {
    acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int idx = tid + NLM_THREADS * k;
    const int r = idx / cw;
    rc[k] = idx < ch * cw ? ((r << 16) | (idx - r * cw)) : -1;
  }

  for(int p = 0; p < a.npatch; p++)
  {
    const int srow = patches[p].x, scol = patches[p].y;
    const int row_min = imax(top, imax(0, -srow)), row_max = imin(bot, H - imax(0, srow));
    if(row_min >= row_max) continue; // uniform
    const int row_top = imax(row_min, imax(P, P - srow));
    const int row_bot = imin(row_max, H - 1 - imax(P, P + srow));
    const int col_min = imax(left, -scol), col_max = imin(right, W - scol);
    const int pc_min = left - imin(P, imin(left, left + scol));
    const int pc_max = right + imin(P, imin(W - right, W - (right + scol)));
    const int nrows = row_max - row_min;
    const int first_end = imin(row_top, row_bot); // rows below it: "add the new bottom row" branch

    // ---- A1: terms of the column-sum recurrence.  Table row t holds, for t = 0, the from-scratch
    //      sums at row_min (init_column_sums(), :208-262) and, for t >= 1, the update that takes
    //      row_min + t - 1 to row_min + t (:437-488).
    const int ci = tid & (CS_LANES - 1);
    for(int t = tid / CS_LANES; t < nrows; t += NLM_THREADS / CS_LANES)
    {
      if(ci >= csw) continue;
      const int c = cs0 + ci;
      float v = 0.0f;
      if(c >= pc_min && c < pc_max)
      {
        if(t == 0)
        {
          const int row = row_min;
          const int rmin = row - imin(P, imin(row, row + srow));
          const int rmax = row + imin(P, imin(H - 1 - row, H - 1 - (row + srow)));
          for(int r = rmin; r <= rmax; r++) v += pixdiff(px(r, c), px(r + srow, c + scol), n0, n1, n2);
        }
        else
        {
          const int row = row_min + t - 1;
          if(row < first_end)
          {
            const int b = row + 1 + P;
            v = pixdiff(px(b, c), px(b + srow, c + scol), n0, n1, n2);
          }
          else if(row < row_bot)
          {
            const int b = row + 1 + P, tt = row - P;
            v = pixdiff2(px(b, c), px(b + srow, c + scol), px(tt, c), px(tt + srow, c + scol), n0, n1, n2);
          }
          else if(row >= row_top)
          {
            const int tt = row - P;
            v = pixdiff(px(tt, c), px(tt + srow, c + scol), n0, n1, n2); // subtracted in A2
          }
        }
      }
      cs[t * a.cs_pitch + ci] = v;
    }
    __syncthreads();

    // ---- A2: the recurrence itself, one thread per table column.  The table is read and written in
    //      batches of 16 rows with unconditional LDS accesses (the table has 16 spare rows), so the
    //      round trips overlap instead of serialising the chain; only the adds are sequential.
    if(tid < csw)
    {
      float v = cs[tid];
      for(int t0 = 1; t0 < nrows; t0 += 16)
      {
        float term[16];
        float *const col = cs + t0 * a.cs_pitch + tid;
#pragma unroll
        for(int u = 0; u < 16; u++) term[u] = col[u * a.cs_pitch];
#pragma unroll
        for(int u = 0; u < 16; u++)
        {
          const int row = row_min + t0 + u - 1;
          const bool live = t0 + u < nrows;
          const float plus = v + term[u], minus = v - term[u];
          v = (live && row < row_bot) ? plus : ((live && row >= row_top) ? minus : v);
          term[u] = v;
        }
#pragma unroll
        for(int u = 0; u < 16; u++) col[u * a.cs_pitch] = term[u];
      }
    }
    __syncthreads();

    // ---- B: sliding row sum, one thread per row (:405-415), batched like A2 (rows of the distortion
    //      table are padded by 16 columns)
    if(tid < nrows)
    {
      const float *const row_cs = cs + tid * a.cs_pitch - cs0; // indexable by frame column
      float distortion = 0.0f;
      for(int i = col_min - P; i < imin(col_min + P, col_max); i++) distortion += row_cs[i];
      float *const drow = dist + (row_min + tid - top) * a.wt_pitch - left;
      for(int c0 = col_min; c0 < col_max; c0 += 16)
      {
        float hi[16], lo[16];
#pragma unroll
        for(int u = 0; u < 16; u++)
        {
          hi[u] = row_cs[c0 + u + P];
          lo[u] = row_cs[c0 + u - P - 1];
        }
#pragma unroll
        for(int u = 0; u < 16; u++)
        {
          const float next = distortion + (hi[u] - lo[u]);
          distortion = (c0 + u < col_max) ? next : distortion;
          hi[u] = distortion;
        }
#pragma unroll
        for(int u = 0; u < 16; u++) drow[c0 + u] = hi[u];
      }
    }
    __syncthreads();

    // ---- C: weights and accumulation (:416-436), all threads
#pragma unroll
    for(int k = 0; k < MAX_PX_PER_THREAD; k++)
    {
      if(rc[k] < 0) continue;
      const int row = top + (rc[k] >> 16), col = left + (rc[k] & 0xffff);
      if(row < row_min || row >= row_max || col < col_min || col >= col_max) continue;
      const float distortion = dist[(row - top) * a.wt_pitch + (col - left)];
      const float4 q = px(row + srow, col + scol);
      float w;
      if(a.center_weight < 0)
        w = mexp2(distortion * a.sharpness);
      else
      {
        const float dis = (distortion + pixdiff(px(row, col), q, a.cpn, a.cpn, a.cpn)) / (1.0f + a.center_weight);
        w = mexp2(fmaxf(0.0f, dis * a.sharpness - 2.0f));
      }
      acc[k].x += q.x * w;
      acc[k].y += q.y * w;
      acc[k].z += q.z * w;
      acc[k].w += 1.0f * w;
    }
    __syncthreads();
  }

  // ---- normalise, blend (:490-521)
#pragma unroll
  for(int k = 0; k < MAX_PX_PER_THREAD; k++)
  {
    if(rc[k] < 0) continue;
    const int row = top + (rc[k] >> 16), col = left + (rc[k] & 0xffff);
    if(row < a.out_row0 || row >= a.out_row1) continue;
    const long o = (long)row * W + col;
    const float4 s = acc[k];
    float4 r;
    if(a.skip_blend)
      r = make_float4(s.x / s.w, s.y / s.w, s.z / s.w, s.w / s.w);
    else
    {
      const float4 ip = in[o];
      r.x = (ip.x * (1.0f - a.luma)) + (s.x / s.w * a.luma);
      r.y = (ip.y * (1.0f - a.chroma)) + (s.y / s.w * a.chroma);
      r.z = (ip.z * (1.0f - a.chroma)) + (s.z / s.w * a.chroma);
      r.w = (ip.w * 0.0f) + (s.w / s.w * 1.0f);
    }
    out[o] = r;
    nlm_store_cell(a, o, r.x);
  }
}

// ---- the pipelined kernel ------------------------------------------------------------------------
// Same four steps, software-pipelined over the patch offsets so that the two recurrences -- which
// only have one column / one row per lane to offer -- run beside the two fully parallel steps instead
// of between them.  Waves 0-3 are the "serial" group, waves 4-15 the "parallel" group; two column-sum
// tables alternate, and one iteration i of the loop is
//   phase 1   parallel: C(i-1) on table (i-1)&1          serial (waves 0-1): A2(i) on table i&1
//   phase 2   parallel: A1(i+1) into table (i+1)&1       serial (waves 2-3): B(i) in place on table i&1
//                                                        serial (waves 0-1): row 0 of A1(i+1)
// with one workgroup barrier after each phase.  Differences in bookkeeping (never in arithmetic):
//   * A1 stores SIGNED terms -- a row leaving the patch is stored negated, a row that neither enters
//     nor leaves as +0 -- so the recurrence A2 is a bare chain of additions (x - y == x + (-y), and a
//     column sum is never -0, so adding +0 is the identity);
//   * a term's entering and leaving rows are masked to 0 instead of branching on the row class
//     ((a - 0) * n == a * n and (0 - c) * n == -(c * n) exactly);
//   * B writes the row sums in place, distortion of frame column c into the slot of column c - P - 1,
//     which it has just read for the last time.
// The staged window uses a compile-time pitch and keeps the three colour planes of a row next to each
// other, so that the twelve LDS reads of a term are immediate offsets from four addresses.
#define NLP_SERIAL 256                      // threads of the serial group
#define NLP_PAR (NLM_THREADS - NLP_SERIAL)  // threads of the parallel group
#define NLP_PX 7                            // accumulators per parallel thread: ceil(72 * 69 / 768)
#define NLP_WP 96                           // window pitch (floats); a window row is 3 planes x NLP_WP
#define NLP_TP 81                           // table pitch (floats), odd: B walks the table one row per lane

typedef float f2 __attribute__((ext_vector_type(2)));

struct nlm_geom
{
  int srow, scol, row_min, row_max, row_top, row_bot, col_min, col_max, pc_min, pc_max, nrows;
};

__device__ __forceinline__ nlm_geom geom_of(const int2 sh, const int top, const int bot, const int left, const int right,
                                            const int P, const int W, const int H)
{
  nlm_geom g;
  g.srow = sh.x;
  g.scol = sh.y;
  g.row_min = imax(top, imax(0, -g.srow));
  g.row_max = imin(bot, H - imax(0, g.srow));
  g.nrows = g.row_max - g.row_min;
  g.row_top = imax(g.row_min, imax(P, P - g.srow));
  g.row_bot = imin(g.row_max, H - 1 - imax(P, P + g.srow));
  g.col_min = imax(left, -g.scol);
  g.col_max = imin(right, W - g.scol);
  g.pc_min = left - imin(P, imin(left, left + g.scol));
  g.pc_max = right + imin(P, imin(W - right, W - (right + g.scol)));
  return g;
}

// `chunk`: the workgroup's chunk in the launch's grid, `lds` its dynamic LDS (the body is shared by nlm_chunks_pipelined
// and by the border workgroups of nlm_chunks_v2)
__device__ __forceinline__ void pipelined_body(const int chunk, float *const lds, const float4 *__restrict__ in,
                                               float4 *__restrict__ out, const nlm_args &a, const int2 *__restrict__ patches)
{
  const int tid = threadIdx.x;
  const int cy_launch = chunk / a.nchx, cx = chunk - cy_launch * a.nchx;
  const int cy = cy_launch + a.cy0;
  const int top = cy * a.chk_h, left = cx * a.chk_w;
  const int bot = imin(top + a.chk_h, a.H), right = imin(left + a.chk_w, a.W);
  const int ch = bot - top, cw = right - left;
  const int P = a.radius, W = a.W, H = a.H;
  const int csw = a.chk_w + 2 * P + 1; // table columns: frame columns left - P - 1 .. left + chk_w + P - 1
  const int cs0 = left - P - 1;
  constexpr int pitch = NLP_TP;
  const int tabsz = a.chk_h * pitch;
  float *const win = lds + 2 * tabsz + 16 * pitch + 64; // 16 spare rows: the batched recurrences read whole batches
  const int r0 = top - a.reach, c0 = left - a.reach;
  const float n0 = a.norm[0], n1 = a.norm[1], n2 = a.norm[2];
  const int n = a.npatch;

  {
    const int wh = a.chk_h + 2 * a.reach;
    for(int i = tid; i < wh * NLP_WP; i += NLM_THREADS)
    {
      const int wy = i / NLP_WP, wx = i - wy * NLP_WP;
      const int r = r0 + wy, c = c0 + wx;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if(r >= 0 && r < H && c >= 0 && c < W) v = in[(long)r * W + c];
      float *const w = win + wy * 3 * NLP_WP + wx;
      w[0] = v.x;
      w[NLP_WP] = v.y;
      w[2 * NLP_WP] = v.z;
    }
  }

  const bool par = tid >= NLP_SERIAL;
  if(!par) __builtin_amdgcn_s_setprio(3); // the recurrences are latency chains: let them issue ahead of the parallel waves
  const int u = tid - NLP_SERIAL;
  // parallel group: the pixels this thread accumulates for the whole chunk (prc = chunk row << 16 | chunk
  // column, wbase = the pixel's place in the window) ...
  float4 acc[NLP_PX];
  int prc[NLP_PX];
#pragma unroll
  for(int k = 0; k < NLP_PX; k++)
  {
    acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int idx = u + NLP_PAR * k;
    const int r = idx / cw, c = idx - r * cw;
    const bool ok = par && idx < ch * cw;
    prc[k] = ok ? ((r << 16) | c) : -1;
  }
  const int wbase0 = ((top - r0) * 3) * NLP_WP + (left - c0); // window offset of the chunk's first pixel
  // ... and the A1 terms it computes for every patch offset: term j is table row t + 1, table column ci
  // (columns 1 .. csw-1; column 0, frame column left - P - 1, is outside every patch and stays 0), packed
  // as window offset of (row t, column ci) | table offset << 16
  int a1[NLP_PX];
#pragma unroll
  for(int k = 0; k < NLP_PX; k++)
  {
    const int idx = u + NLP_PAR * k;
    const int t = idx / (csw - 1), ci = 1 + idx - t * (csw - 1);
    a1[k] = (t * 3 * NLP_WP + ci) | (((t + 1) * pitch + ci) << 16);
  }
  __syncthreads();

  // Everything below runs once per patch offset.  For a chunk that no patch can carry over the frame border
  // (all but the outermost ring of chunks) the geometry of an offset is the chunk itself, so that case is
  // compiled separately: no per-offset scalar arithmetic, no masks, no range tests.
  auto run = [&](auto interior_tag) {
    constexpr bool INTERIOR = decltype(interior_tag)::value;
    // ---- A1, rows 1.. of the table: signed terms of the column-sum recurrence (:437-488), parallel group
    auto A1 = [&](const nlm_geom &g, float *const T) {
      const int dS = g.srow * 3 * NLP_WP + g.scol;
      const int dT = (2 * P + 1) * 3 * NLP_WP;
      // window offset of (row_min + 1 + P, table column 0)
      const float *const wb = win + ((g.row_min + 1 + P - r0) * 3) * NLP_WP + (cs0 - c0);
      const int tend = g.nrows * pitch; // terms of table rows 1 .. nrows - 1
      // every row both enters and leaves, every column is inside the patch range: no masks
      const bool plain = INTERIOR || (g.row_top <= g.row_min && g.row_bot >= g.row_max - 1 && g.pc_min == left - P
                         && g.pc_max == left + a.chk_w + P);
      const f2 n01 = { n0, n1 };
      if(u < g.nrows - 1) T[(u + 1) * pitch] = 0.0f; // column 0 (B leaves row sums in it)
      if(plain)
      {
  #pragma unroll
        for(int k = 0; k < NLP_PX; k++)
        {
          if(k == 4) asm volatile("" ::: "memory"); // keeps the register footprint of the hoisted LDS reads at four terms
          int pk = a1[k];
          asm volatile("" : "+v"(pk)); // unpack here, every time: hoisted out of the offset loop the fields cost 14 registers
          const int ti = (unsigned)pk >> 16;
          if(ti >= tend) continue;
          const float *const pb = wb + (pk & 0xffff);
          const float *const pt = pb - dT;
          const f2 b = { pb[0], pb[NLP_WP] }, bs = { pb[dS], pb[dS + NLP_WP] };
          const f2 l = { pt[0], pt[NLP_WP] }, ls = { pt[dS], pt[dS + NLP_WP] };
          const f2 z = { pb[2 * NLP_WP], pt[2 * NLP_WP] }, zs = { pb[dS + 2 * NLP_WP], pt[dS + 2 * NLP_WP] };
          f2 e = b - bs, q = l - ls, w = z - zs;
          e = e * e;
          q = q * q;
          w = w * w;
          const f2 d = (e - q) * n01;
          T[ti] = d.x + d.y + (w.x - w.y) * n2;
        }
      }
      else
      {
  #pragma unroll
        for(int k = 0; k < NLP_PX; k++)
        {
          if(k == 4) asm volatile("" ::: "memory");
          int pk = a1[k];
          asm volatile("" : "+v"(pk));
          const int ti = (unsigned)pk >> 16;
          if(ti >= tend) continue;
          const int lw = pk & 0xffff;
          const int t = (int)(((unsigned)lw * 58255u) >> 24); // lw / (3 * NLP_WP), exact below 2^16
          const int row = g.row_min + t, c = cs0 + (lw - t * 3 * NLP_WP);
          const float *const pb = wb + lw;
          const float *const pt = pb - dT;
          const float ax = pb[0] - pb[dS], ay = pb[NLP_WP] - pb[dS + NLP_WP], az = pb[2 * NLP_WP] - pb[dS + 2 * NLP_WP];
          const float tx = pt[0] - pt[dS], ty = pt[NLP_WP] - pt[dS + NLP_WP], tz = pt[2 * NLP_WP] - pt[dS + 2 * NLP_WP];
          const bool colok = c >= g.pc_min && c < g.pc_max;
          const bool enter = colok && row < g.row_bot, leave = colok && row >= g.row_top;
          const float ex = enter ? ax * ax : 0.f, ey = enter ? ay * ay : 0.f, ez = enter ? az * az : 0.f;
          const float lx = leave ? tx * tx : 0.f, ly = leave ? ty * ty : 0.f, lz = leave ? tz * tz : 0.f;
          T[ti] = (ex - lx) * n0 + (ey - ly) * n1 + (ez - lz) * n2;
        }
      }
    };
    // ---- row 0 of the table: the from-scratch sums at row_min (init_column_sums(), :208-262), threads 0..csw-1
    auto A1_first = [&](const nlm_geom &g, float *const T) {
      const int c = cs0 + tid;
      float v = 0.0f;
      if(c >= g.pc_min && c < g.pc_max)
      {
        const int row = g.row_min;
        const int rmin = row - imin(P, imin(row, row + g.srow));
        const int rmax = row + imin(P, imin(H - 1 - row, H - 1 - (row + g.srow)));
        const int dS = g.srow * 3 * NLP_WP + g.scol;
        const float *pr = win + ((rmin - r0) * 3) * NLP_WP + (c - c0);
        for(int r = rmin; r <= rmax; r++, pr += 3 * NLP_WP)
        {
          const float dx = pr[0] - pr[dS], dy = pr[NLP_WP] - pr[dS + NLP_WP], dz = pr[2 * NLP_WP] - pr[dS + 2 * NLP_WP];
          v += dx * dx * n0 + dy * dy * n1 + dz * dz * n2;
        }
      }
      T[tid] = v;
    };
    // ---- A2: the recurrence, one thread per table column, 16 rows of LDS traffic in flight and ONE dependent
    //      addition per row.  A lone wave issues an instruction every four to five cycles, so the instruction
    //      count of this loop IS the length of phase 1: read, add, write per row, addresses as immediates
    auto A2 = [&](const nlm_geom &g, float *const T) {
      float v = T[tid];
      int t0 = 1;
      for(; t0 + 16 <= g.nrows; t0 += 16)
      {
        float term[16];
        float *const col = T + t0 * pitch + tid;
  #pragma unroll
        for(int k = 0; k < 16; k++) term[k] = col[k * pitch];
        term[0] = v + term[0];
  #pragma unroll
        for(int k = 1; k < 16; k++) term[k] = term[k - 1] + term[k];
        v = term[15];
  #pragma unroll
        for(int k = 0; k < 16; k++) col[k * pitch] = term[k];
      }
      if(t0 < g.nrows)
      {
        float term[16];
        float *const col = T + t0 * pitch + tid;
        const int live = g.nrows - t0;
  #pragma unroll
        for(int k = 0; k < 16; k++) term[k] = col[k * pitch];
  #pragma unroll
        for(int k = 0; k < 16; k++)
        {
          const float next = v + term[k];
          v = k < live ? next : v;
          term[k] = v;
        }
  #pragma unroll
        for(int k = 0; k < 16; k++)
          if(k < live) col[k * pitch] = term[k];
      }
    };
    // ---- B: sliding row sum (:405-415), one thread per table row, in place; same shape
    auto B = [&](const nlm_geom &g, float *const T, const int rr) {
      float *const rowp = T + rr * pitch - cs0; // indexable by frame column
      float distortion = 0.0f;
      for(int i = g.col_min - P; i < imin(g.col_min + P, g.col_max); i++) distortion += rowp[i];
      int cb = g.col_min;
      for(; cb + 16 <= g.col_max; cb += 16)
      {
        float hi[16], lo[16];
  #pragma unroll
        for(int k = 0; k < 16; k++)
        {
          hi[k] = rowp[cb + k + P];
          lo[k] = rowp[cb + k - P - 1];
        }
  #pragma unroll
        for(int k = 0; k < 16; k++) hi[k] = hi[k] - lo[k];
        hi[0] = distortion + hi[0];
  #pragma unroll
        for(int k = 1; k < 16; k++) hi[k] = hi[k - 1] + hi[k];
        distortion = hi[15];
  #pragma unroll
        for(int k = 0; k < 16; k++) rowp[cb + k - P - 1] = hi[k];
      }
      if(cb < g.col_max)
      {
        float hi[16], lo[16];
        const int live = g.col_max - cb;
  #pragma unroll
        for(int k = 0; k < 16; k++)
        {
          hi[k] = rowp[cb + k + P];
          lo[k] = rowp[cb + k - P - 1];
        }
  #pragma unroll
        for(int k = 0; k < 16; k++)
        {
          const float next = distortion + (hi[k] - lo[k]);
          distortion = k < live ? next : distortion;
          hi[k] = distortion;
        }
  #pragma unroll
        for(int k = 0; k < 16; k++)
          if(k < live) rowp[cb + k - P - 1] = hi[k];
      }
    };
    // ---- C: weights and accumulation (:416-436), parallel group; all LDS reads first, then the arithmetic
    auto C = [&](const nlm_geom &g, const float *const T) {
      const int dS = g.srow * 3 * NLP_WP + g.scol;
      const int trow = (g.row_min - top) * pitch; // table row 0 is chunk row row_min - top
      const bool whole = INTERIOR || (g.row_min == top && g.row_max == bot && g.col_min == left && g.col_max == right);
      float dist[NLP_PX], qx[NLP_PX], qy[NLP_PX], qz[NLP_PX];
      bool ok[NLP_PX];
  #pragma unroll
      for(int k = 0; k < NLP_PX; k++)
      {
        int pk = prc[k];
        asm volatile("" : "+v"(pk));
        const int r = pk >> 16, c = pk & 0xffff;
        ok[k] = pk >= 0;
        if(!whole)
        {
          const int row = top + r, col = left + c;
          ok[k] = ok[k] && row >= g.row_min && row < g.row_max && col >= g.col_min && col < g.col_max;
        }
        const int ti = ok[k] ? __mul24(r, pitch) + c - trow : 0;
        dist[k] = T[ti];
        const float *const pq = win + (wbase0 + dS) + __mul24(r, 3 * NLP_WP) + c;
        qx[k] = pq[0];
        qy[k] = pq[NLP_WP];
        qz[k] = pq[2 * NLP_WP];
      }
      if(a.center_weight < 0)
      {
  #pragma unroll
        for(int k = 0; k < NLP_PX; k++)
        {
          const float w = mexp2(dist[k] * a.sharpness);
          const float sx = acc[k].x + qx[k] * w, sy = acc[k].y + qy[k] * w, sz = acc[k].z + qz[k] * w, sw = acc[k].w + 1.0f * w;
          acc[k].x = ok[k] ? sx : acc[k].x;
          acc[k].y = ok[k] ? sy : acc[k].y;
          acc[k].z = ok[k] ? sz : acc[k].z;
          acc[k].w = ok[k] ? sw : acc[k].w;
        }
      }
      else
      {
  #pragma unroll
        for(int k = 0; k < NLP_PX; k++)
        {
          int pk = prc[k];
          asm volatile("" : "+v"(pk));
          const float *const pp = win + wbase0 + __mul24(pk >> 16, 3 * NLP_WP) + (pk & 0xffff);
          const float dx = pp[0] - qx[k], dy = pp[NLP_WP] - qy[k], dz = pp[2 * NLP_WP] - qz[k];
          const float dis = (dist[k] + (dx * dx * a.cpn + dy * dy * a.cpn + dz * dz * a.cpn)) / (1.0f + a.center_weight);
          const float w = mexp2(fmaxf(0.0f, dis * a.sharpness - 2.0f));
          const float sx = acc[k].x + qx[k] * w, sy = acc[k].y + qy[k] * w, sz = acc[k].z + qz[k] * w, sw = acc[k].w + 1.0f * w;
          acc[k].x = ok[k] ? sx : acc[k].x;
          acc[k].y = ok[k] ? sy : acc[k].y;
          acc[k].z = ok[k] ? sz : acc[k].z;
          acc[k].w = ok[k] ? sw : acc[k].w;
        }
      }
    };

    // the patch shifts of offsets i - 1, i, i + 1 stay in scalar registers and the one for i + 2 is fetched an
    // iteration ahead: a scalar load at the top of every phase would sit on every wave's critical path
    auto geom = [&](const int2 sh) {
      if(!INTERIOR) return geom_of(sh, top, bot, left, right, P, W, H);
      nlm_geom g;
      g.srow = sh.x;
      g.scol = sh.y;
      g.row_min = g.row_top = top;
      g.row_max = g.row_bot = bot;
      g.nrows = ch;
      g.col_min = left;
      g.col_max = right;
      g.pc_min = left - P;
      g.pc_max = right + P;
      return g;
    };
    int2 sh_prev = make_int2(0, 0), sh_cur = patches[0], sh_next = patches[n > 1 ? 1 : 0];
    {
      const nlm_geom g = geom(sh_cur);
      if(g.nrows > 0)
      {
        if(par) A1(g, lds);
        else if(tid < csw) A1_first(g, lds);
      }
    }
    __syncthreads();
    for(int i = 0; i <= n; i++)
    {
      const int2 sh_next2 = patches[i + 2 < n ? i + 2 : n - 1];
      float *const Ti = lds + (i & 1) * tabsz;       // offset i, and i + 2
      float *const To = lds + ((i + 1) & 1) * tabsz; // offsets i - 1 and i + 1
      // phase 1
      if(par)
      {
        if(i >= 1)
        {
          const nlm_geom g = geom(sh_prev);
          if(g.nrows > 0) C(g, To);
        }
      }
      else if(tid < csw && i < n)
      {
        const nlm_geom g = geom(sh_cur);
        if(g.nrows > 0) A2(g, Ti);
      }
      __syncthreads();
      // phase 2
      if(par)
      {
        if(i + 1 < n)
        {
          const nlm_geom g = geom(sh_next);
          if(g.nrows > 0) A1(g, To);
        }
      }
      else if(tid < 128)
      {
        if(tid < csw && i + 1 < n)
        {
          const nlm_geom g = geom(sh_next);
          if(g.nrows > 0) A1_first(g, To);
        }
      }
      else if(i < n)
      {
        const nlm_geom g = geom(sh_cur);
        if(tid - 128 < g.nrows) B(g, Ti, tid - 128);
      }
      __syncthreads();
      sh_prev = sh_cur;
      sh_cur = sh_next;
      sh_next = sh_next2;
    }

  };
  const bool interior = top >= a.reach && bot + a.reach <= H && left >= a.reach && right + a.reach <= W && ch == a.chk_h
                        && cw == a.chk_w;
  if(interior)
    run(std::true_type{});
  else
    run(std::false_type{});

  // ---- normalise, blend (:490-521)
#pragma unroll
  for(int k = 0; k < NLP_PX; k++)
  {
    if(prc[k] < 0) continue;
    const int row = top + (prc[k] >> 16), col = left + (prc[k] & 0xffff);
    if(row < a.out_row0 || row >= a.out_row1) continue;
    const long o = (long)row * W + col;
    const float4 s = acc[k];
    float4 r;
    if(a.skip_blend)
      r = make_float4(s.x / s.w, s.y / s.w, s.z / s.w, s.w / s.w);
    else
    {
      const float4 ip = in[o];
      r.x = (ip.x * (1.0f - a.luma)) + (s.x / s.w * a.luma);
      r.y = (ip.y * (1.0f - a.chroma)) + (s.y / s.w * a.chroma);
      r.z = (ip.z * (1.0f - a.chroma)) + (s.z / s.w * a.chroma);
      r.w = (ip.w * 0.0f) + (s.w / s.w * 1.0f);
    }
    out[o] = r;
    nlm_store_cell(a, o, r.x);
  }
}

__global__ __launch_bounds__(NLM_THREADS) void nlm_chunks_pipelined(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                                    const nlm_args a, const int2 *__restrict__ patches)
{
  extern __shared__ float lds[];
  pipelined_body(blockIdx.x, lds, in, out, a, patches);
}

// ---- the interior-chunk kernel: nlm2_body.h (also compiled for the host: tests/native/nlm2_host.cpp) -------------
// The two statements below write M0 and say so in their clobber lists; clang warns that M0 is a register it reserves (once per
// instantiation and use: 1 388 times a build).  The compiler's own users of M0 on gfx950 -- LDS-DMA, GWS, relative register moves --
// do not occur in this file's kernels, and each would write M0 itself in front of its use.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
struct nlm2_device_env
{
  float *lds_;
  int chunk_;
  __device__ __forceinline__ int tid() const { return threadIdx.x; }
  __device__ __forceinline__ int bid() const { return chunk_; }
  __device__ __forceinline__ float *lds() const { return lds_; }
  __device__ __forceinline__ void sync() const { __syncthreads(); }
  __device__ __forceinline__ void prio_high() const { __builtin_amdgcn_s_setprio(3); }
  __device__ __forceinline__ void store_cell(const nlm_args &a, const long o, const float L) const { nlm_store_cell(a, o, L); }
  __device__ __forceinline__ void sched_fence() const { __builtin_amdgcn_sched_barrier(0); } // nothing moves across
  // ds_write_addtid_b32: LDS address = M0 + offset + 4 * lane, data from one register, no address register.  M0 is
  // written right in front of the store (the compiler does not know the instruction reads it) with the wait state the
  // hardware wants between an SALU write of M0 and an add-TID instruction
  template <int OFF> __device__ __forceinline__ void st_addtid(float *const wave_base, const int, const float v) const
  {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float *)wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:%2" : : "s"(m0v), "v"(v), "n"(OFF) : "memory", "m0");
  }
  // eight rows of the column recurrence in ONE statement: M0 written once, then add / store eight times.  The additions are the
  // v_add_f32 the compiler emits for `v + term` (no contraction to worry about: there is no product); a VGPR written by a VALU
  // instruction and read by the LDS instruction behind it is interlocked by the hardware
  template <int T0, int ROWBYTES> __device__ __forceinline__ void chain8(float *const wave_base, const int, float &v, const float *const term) const
  {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float *)wave_base);
    asm volatile("s_mov_b32 m0, %[b]\n\ts_nop 0\n\t"
                 "v_add_f32 %[v], %[v], %[t0]\n\tds_write_addtid_b32 %[v] offset:%[o0]\n\t"
                 "v_add_f32 %[v], %[v], %[t1]\n\tds_write_addtid_b32 %[v] offset:%[o1]\n\t"
                 "v_add_f32 %[v], %[v], %[t2]\n\tds_write_addtid_b32 %[v] offset:%[o2]\n\t"
                 "v_add_f32 %[v], %[v], %[t3]\n\tds_write_addtid_b32 %[v] offset:%[o3]\n\t"
                 "v_add_f32 %[v], %[v], %[t4]\n\tds_write_addtid_b32 %[v] offset:%[o4]\n\t"
                 "v_add_f32 %[v], %[v], %[t5]\n\tds_write_addtid_b32 %[v] offset:%[o5]\n\t"
                 "v_add_f32 %[v], %[v], %[t6]\n\tds_write_addtid_b32 %[v] offset:%[o6]\n\t"
                 "v_add_f32 %[v], %[v], %[t7]\n\tds_write_addtid_b32 %[v] offset:%[o7]"
                 : [v] "+v"(v)
                 : [b] "s"(m0v), [t0] "v"(term[T0]), [t1] "v"(term[T0 + 1]), [t2] "v"(term[T0 + 2]), [t3] "v"(term[T0 + 3]),
                   [t4] "v"(term[T0 + 4]), [t5] "v"(term[T0 + 5]), [t6] "v"(term[T0 + 6]), [t7] "v"(term[T0 + 7]),
                   [o0] "n"(T0 * ROWBYTES), [o1] "n"((T0 + 1) * ROWBYTES), [o2] "n"((T0 + 2) * ROWBYTES), [o3] "n"((T0 + 3) * ROWBYTES),
                   [o4] "n"((T0 + 4) * ROWBYTES), [o5] "n"((T0 + 5) * ROWBYTES), [o6] "n"((T0 + 6) * ROWBYTES), [o7] "n"((T0 + 7) * ROWBYTES)
                 : "memory", "m0");
  }
  __device__ __forceinline__ bool any(const bool c) const { return __builtin_amdgcn_ballot_w64(c) != 0; }
  // the value of the lane to the left within a row of 16 lanes (DPP row_shr:1); lane 0 of a row gets 0
  __device__ __forceinline__ float lane_shr1(const float v) const
  {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
  }
  static constexpr bool TIMED = false;
  __device__ __forceinline__ long long clock() const { return 0; }
  static __device__ __forceinline__ float int_as_float(const int v) { return __int_as_float(v); }
  static __device__ __forceinline__ float max_num(const float a, const float b) { return __builtin_fmaxf(a, b); } // v_max_f32
  static __device__ __forceinline__ float min_num(const float a, const float b) { return __builtin_fminf(a, b); } // v_min_f32
  static __device__ __forceinline__ int cvt_i32_sat(const float v)
  {
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v)); // the instruction's own semantics, not C's (undefined out of range)
    return r;
  }
  // n / d, correctly rounded, for a wave-uniform divisor d in [1, 2^20] (nlm3_body.h CENTER: 1 + center_weight; the launch checks it):
  // ieee_inrange.h div_uniform().  Numerators below 2^-103 or from 2^96 on may come out an ulp off (their residuals are subnormal /
  // the expansion would have scaled): the weight behind them is 1 or 0 either way for a sharpness in [2^-60, 2^60], which is what
  // the launch admits.  tests/test_gpu_devmath.py compares it with the host's division.
  static __device__ __forceinline__ float rcp_refined(const float d) { return ansel_ieee::rcp_refined(d); }
  static __device__ __forceinline__ float div_uniform(const float n, const float d, const float y1) { return ansel_ieee::div_uniform(n, d, y1); }
};
#pragma clang diagnostic pop

template <int P, int WP, int TP, bool DEEP, bool CENTER>
__global__ __launch_bounds__(NL2_THREADS) void nlm_chunks_v2(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                             const nlm_args a, const int2 *__restrict__ patches,
                                                             const int *__restrict__ order, const int n_border)
{
  // ONE launch for the whole grid: `order` lists the chunks whose patches can leave the frame first -- they take the
  // pipelined kernel's body with all its per-offset geometry and are the slower ones, so they start first and the
  // interior chunks fill the machine behind them (as a launch of their own the ~2.5 % border chunks of a 100 MP frame
  // cost three rounds of 256 CUs at the end: 2.6 of 46.6 ms)
  extern __shared__ float lds[];
  const int chunk = order[blockIdx.x];
  if(blockIdx.x < n_border)
  {
    pipelined_body(chunk, lds, in, out, a, patches);
    return;
  }
  nlm2_device_env env;
  env.lds_ = lds;
  env.chunk_ = chunk;
  nlm2::body<P, WP, TP, DEEP, CENTER>(env, in, out, a, patches);
}

// the measuring build (ANSEL_NLM2_TIMED, tools/nlm_phase_clocks.py): the same body with a clock read around every step
#ifdef ANSEL_HIP_MEASURING // clock reads around every step (ANSEL_NLM2_TIMED, tools/nlm_phase_clocks.py)
struct nlm2_timed_env : nlm2_device_env
{
  static constexpr bool TIMED = true;
  __device__ __forceinline__ long long clock() const { return (long long)__builtin_readcyclecounter(); }
};
__global__ __launch_bounds__(NL2_THREADS) void nlm_chunks_v2_timed(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                                   const nlm_args a, const int2 *__restrict__ patches,
                                                                   const int *__restrict__ order, const int n_border)
{
  extern __shared__ float lds[];
  if(blockIdx.x < n_border) return;
  nlm2_timed_env env;
  env.lds_ = lds;
  env.chunk_ = order[blockIdx.x];
  nlm2::body<2, NL2_WP_TIGHT, NL2_TP_TIGHT, true>(env, in, out, a, patches);
}
#endif // ANSEL_HIP_MEASURING

// the third version of the interior-chunk kernel (nlm3_body.h): offsets in rows of consecutive column shifts, patch
// radius 2, chunks of at most 56 rows; same launch shape, border chunks first with the pipelined body
// P, CENTER (round 6): patch radius 1 beside 2; the weight with the centre pixel's term (denoise (profiled)'s non-local-means mode)
template <int NPXL, int MSEG, int P = 2, bool CENTER = false>
__global__ __launch_bounds__(NL3_THREADS) void nlm_chunks_v3(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                             const nlm_args a_by_value, const int2 *__restrict__ patches,
                                                             const int *__restrict__ order, const int n_border, const int ndx)
{
  // the launch's ~30 parameters are read from the kernarg segment where they are used (hip_common.h kernarg_at(): behind the two
  // pointers), not held in scalar registers through the offsets' loop
  constexpr int at = kernarg_offset_after<nlm_args, const float4 *, float4 *>();
  static_assert(at == 16, "nlm_chunks_v3: the by-value nlm_args follows the two plane pointers");
  const nlm_args &a = kernarg_at<nlm_args>(at);
  (void)a_by_value;
  extern __shared__ float lds[];
  const int chunk = order[blockIdx.x];
  nlm2_device_env env;
  env.lds_ = lds;
  env.chunk_ = chunk;
  if(blockIdx.x < n_border)
  {
    // the outermost ring: the same body with the squared differences that are not there set to +0 (nlm3_body.h, BORDER);
    // a last row of chunks lower than ten rows keeps the first version's body
    const int cy = chunk / a.nchx + a.cy0, cx = chunk % a.nchx;
    const int cw = min(a.chk_w, a.W - cx * a.chk_w), ch = min(a.chk_h, a.H - cy * a.chk_h);
    if(a.variant & 2048 || !nlm3::border_fits(cw, ch)) pipelined_body(chunk, lds, in, out, a, patches);
    else nlm3::body<NPXL, MSEG, true, false, false, P, CENTER>(env, in, out, a, patches, ndx);
    return;
  }
  nlm3::body<NPXL, MSEG, false, false, false, P, CENTER>(env, in, out, a, patches, ndx);
}

// the fused variant of the third version (nlm3_body.h, FUSED): three tables, the row recurrence inside the C role --
// chunks of up to 64 rows (the 45 MP and 60 MP frames' grids)
// TALL (round 5): a chunk grid of 65 - 69 rows (the 24 / 42 / 150 MP frames).  The body runs a chunk's first 64 rows and exports
// the column sums behind them, per offset, to seeds[position in the launch][offset][slot] (nlm3_body.h TALL); nlm_tail below
// continues them through the rows that are left -- interior chunks and the outermost ring alike (BORDER bodies).
template <int NPXL, int MSEG, bool TALL, int P = 2, bool CENTER = false>
__global__ __launch_bounds__(NL3_THREADS) void nlm_chunks_v4(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                             const nlm_args a_by_value, const int2 *__restrict__ patches,
                                                             const int *__restrict__ order, const int n_border, const int ndx,
                                                             float *__restrict__ seeds)
{
  constexpr int at = kernarg_offset_after<nlm_args, const float4 *, float4 *>(); // (see nlm_chunks_v3)
  static_assert(at == 16, "nlm_chunks_v4: the by-value nlm_args follows the two plane pointers");
  const nlm_args &a = kernarg_at<nlm_args>(at);
  (void)a_by_value;
  extern __shared__ float lds[];
  const int chunk = order[blockIdx.x];
  nlm2_device_env env;
  env.lds_ = lds;
  env.chunk_ = chunk;
  if(blockIdx.x < n_border)
  {
    const int cy = chunk / a.nchx + a.cy0, cx = chunk % a.nchx;
    const int cw = min(a.chk_w, a.W - cx * a.chk_w), ch = min(a.chk_h, a.H - cy * a.chk_h);
    // (a tall chunk: the head's rows decide; a chunk the BORDER body refuses is lower than ten rows and has no tail)
    if(!nlm3::border_fits(cw, TALL ? min(ch, nlm3::TALL_HEAD) : ch)) pipelined_body(chunk, lds, in, out, a, patches);
    else nlm3::body<NPXL, MSEG, true, true, TALL, P, CENTER>(env, in, out, a, patches, ndx,
                                                             TALL ? seeds + (size_t)blockIdx.x * a.npatch * nlm3::TALL_SEED_PITCH : nullptr);
    return;
  }
  nlm3::body<NPXL, MSEG, false, true, TALL, P, CENTER>(env, in, out, a, patches, ndx,
                                                       TALL ? seeds + (size_t)blockIdx.x * a.npatch * nlm3::TALL_SEED_PITCH : nullptr);
}

// the rows of the chunks of a tall grid behind the 64th (nlm_tail_body.h): one workgroup of 512 threads per chunk, in the
// order (and with the export slots) of the head launch
// (two instantiations, two launches: with both bodies in one kernel the interior chunks' launch carried the ring's scalar
// registers -- 86 where 80 let four workgroups share a CU)
template <bool BORDER, int P = 2, bool CENTER = false>
__global__ __launch_bounds__(NLT_THREADS, 8) void nlm_tail(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                           const nlm_args a, const int2 *__restrict__ patches,
                                                           const int *__restrict__ order, const int first,
                                                           const float *__restrict__ seeds)
{
  extern __shared__ float lds[];
  nlm2_device_env env;
  env.lds_ = lds;
  const int pos = first + (int)blockIdx.x; // position in the head launch: its chunk and its export slot
  env.chunk_ = order[pos];
  nlmt::body<BORDER, P, CENTER>(env, in, out, a, patches, seeds + (size_t)pos * a.npatch * NLT_SEED_PITCH);
}

typedef void (*nlm2_kernel_t)(const float4 *, float4 *, nlm_args, const int2 *, const int *, int);
template <int P, bool CENTER> nlm2_kernel_t nlm2_kernel_of_(const bool tight, const bool deep)
{
  if(tight) return deep ? nlm_chunks_v2<P, NL2_WP_TIGHT, NL2_TP_TIGHT, true, CENTER> : nlm_chunks_v2<P, NL2_WP_TIGHT, NL2_TP_TIGHT, false, CENTER>;
  return deep ? nlm_chunks_v2<P, NL2_WP_LOOSE, NL2_TP_LOOSE, true, CENTER> : nlm_chunks_v2<P, NL2_WP_LOOSE, NL2_TP_LOOSE, false, CENTER>;
}
// center: the weight of denoise (profiled)'s non-local-means mode (nlm2_body.h, CENTER)
template <int P> nlm2_kernel_t nlm2_kernel_of(const bool tight, const bool deep, const bool center)
{
  return center ? nlm2_kernel_of_<P, true>(tight, deep) : nlm2_kernel_of_<P, false>(tight, deep);
}

int sgn(const int v) { return (v > 0) - (v < 0); }

// scatter(), nlmeans_core.c:95-105
int scatter(const float scale, const float scattering, const int i1, const int i2)
{
  const int a1 = abs(i1), a2 = abs(i2);
  return (int)(scale * ((a1 * a1 * a1 + 7.0 * a1 * sqrt((double)a2)) * sgn(i1) * scattering / 6.0 + i1));
}

// compute_slice_height(), nlmeans_core.c:264-295
int slice_height(const int height)
{
  if(height % SLICE_HEIGHT == 0) return SLICE_HEIGHT;
  int best = height % SLICE_HEIGHT, best_incr = 0;
  for(int incr = 1; incr < 10; incr++)
  {
    const int plus = height % (SLICE_HEIGHT + incr);
    if(plus == 0) return SLICE_HEIGHT + incr;
    if(plus > best)
    {
      best_incr = incr;
      best = plus;
    }
    const int minus = height % (SLICE_HEIGHT - incr);
    if(minus == 0) return SLICE_HEIGHT - incr;
    if(minus > best)
    {
      best_incr = -incr;
      best = minus;
    }
  }
  return SLICE_HEIGHT + best_incr;
}

// compute_slice_width(), nlmeans_core.c:297-313
int slice_width(const int width)
{
  int sl = SLICE_WIDTH;
  int rem = width % sl;
  if(rem < SLICE_WIDTH / 2 && (width % (sl - 4)) > rem)
  {
    sl -= 4;
    rem = width % sl;
    if(rem < SLICE_WIDTH / 2 && (width % (sl - 4)) > rem) sl -= 4;
  }
  return sl;
}

} // namespace

namespace ansel
{

int nlmeans_core_launch(int devid, const float4 *in, float4 *out, int width, int height, const nlm_core_params_t &p)
{
  if(p.band && (p.band->frame_h <= 0 || p.band->row0 < 0 || p.band->row1 > p.band->frame_h || p.band->row0 >= p.band->row1))
    return DT_HIP_INVALID_ARG;
  if(width <= 0 || height <= 0) return DT_HIP_SUCCESS;
  if(p.patch_radius < 0 || p.patch_radius > 16 || p.search_radius < 0 || p.search_radius > 32)
  {
    set_last_error("nlmeans: patch radius %d / search radius %d outside the device limits (16 / 32)", p.patch_radius,
                   p.search_radius);
    return DT_HIP_INVALID_ARG;
  }
  const int K = p.search_radius;
  std::vector<int2> patches;
  int max_shift = 0;
  for(int ri = -K; ri <= K; ri++)
    for(int ci = -K; ci <= K; ci++)
    {
      const int r = scatter(p.scale, p.scattering, ri, ci), c = scatter(p.scale, p.scattering, ci, ri);
      patches.push_back(make_int2(r, c));
      max_shift = std::max(max_shift, std::max(abs(r), abs(c)));
    }
  nlm_args a;
  memset(&a, 0, sizeof(a));
  // a row band (p.band != nullptr): `in` holds frame rows from band->buf_row0 on, `out` the band's own rows
  // [row0, row1); the chunk grid, the validity tests and every index are the FRAME's, the launch covers
  // the chunk rows that intersect the own rows
  const band_view_t *const bv = p.band;
  if(bv) height = bv->frame_h;
  a.W = width;
  a.H = height;
  a.chk_h = slice_height(height);
  a.chk_w = slice_width(width);
  a.nchx = (width + a.chk_w - 1) / a.chk_w;
  a.cy0 = bv ? bv->row0 / a.chk_h : 0;
  a.out_row0 = bv ? bv->row0 : 0;
  a.out_row1 = bv ? bv->row1 : height;
  const int nchy = bv ? (bv->row1 + a.chk_h - 1) / a.chk_h - a.cy0 : (height + a.chk_h - 1) / a.chk_h;
  if(bv)
  {
    in -= (size_t)bv->buf_row0 * width;
    out -= (size_t)bv->row0 * width;
  }
  a.radius = p.patch_radius;
  a.npatch = (int)patches.size();
  a.cs_pitch = (a.chk_w + 2 * a.radius + 1) | 1; // odd pitches: the row-parallel step strides whole rows
  a.wt_pitch = (a.chk_w + 16) | 1; // 16 spare columns: the batched row recurrence stores whole batches
  a.zc = p.band ? nullptr : p.cell_out; // (a band's rows are indexed with frame rows: the grid's relay has its own pass)
  a.zc_sigma_r = p.cell_sigma_r;
  a.zc_size_z = p.cell_size_z;
  a.sharpness = p.sharpness;
  a.center_weight = p.center_weight;
  a.cpn = p.center_weight * (2 * p.patch_radius + 1) * (2 * p.patch_radius + 1); // compute_center_pixel_norm()
  for(int k = 0; k < 3; k++) a.norm[k] = p.norm[k];
  a.luma = p.luma;
  a.chroma = p.chroma;
  a.skip_blend = (p.luma == 1.0 && p.chroma == 1.0);
  if(a.chk_w * a.chk_h > NLM_THREADS * MAX_PX_PER_THREAD || a.chk_h > NLM_THREADS
     || a.chk_w + 2 * a.radius + 1 > CS_LANES)
  {
    set_last_error("nlmeans: chunk %d x %d exceeds the kernel's accumulator budget", a.chk_w, a.chk_h);
    return DT_HIP_DEFAULT_ERROR;
  }
  // the window a chunk's patches can touch: init_column_sums() reads radius rows/columns beyond the
  // chunk, the recurrence one more row below, and everything once more shifted by the patch offset
  a.reach = a.radius + 1 + max_shift;
  a.win_pitch = (a.chk_w + 2 * a.reach) | 1;
  // the pipelined kernel: two tables + the planar window with its fixed pitch
  const size_t pipe_bytes = ((size_t)2 * a.chk_h * NLP_TP + 16 * NLP_TP + 64
                             + (size_t)(a.chk_h + 2 * a.reach) * 3 * NLP_WP) * sizeof(float);
  const bool pipelined = pipe_bytes <= 160 * 1024 && a.chk_w + 2 * a.reach <= NLP_WP && a.chk_h <= NLP_SERIAL / 2
                         && a.chk_w + 2 * a.radius + 1 <= NLP_TP && a.chk_w * a.chk_h <= NLP_PAR * NLP_PX;
  const size_t table_bytes = ((size_t)(a.chk_h + 16) * a.cs_pitch + (size_t)a.chk_h * a.wt_pitch + 64) * sizeof(float);
  const size_t window_bytes = (size_t)3 * (a.chk_h + 2 * a.reach) * a.win_pitch * sizeof(float);
  const bool staged = table_bytes + window_bytes <= 160 * 1024;
  const size_t lds_bytes = pipelined ? pipe_bytes : table_bytes + (staged ? window_bytes : 0);
  hipStream_t s = stream_of(devid);
  const void *const fn = pipelined ? (const void *)nlm_chunks_pipelined
                                   : (staged ? (const void *)nlm_chunks<true> : (const void *)nlm_chunks<false>);
  if(lds_bytes > 64 * 1024)
    ANSEL_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  // interior chunks (all but the outermost ring: 97.5 % of a 100 MP frame) take nlm_chunks_v2's body when the
  // configuration is one it is built for: patch radius 1..3, either weight (denoise (non-local means)'s, and -- round 5 --
  // the one with the centre-pixel term of denoise (profiled)'s non-local-means mode, whose default patch radius is 1)
  const int S2 = 2 * a.radius + 1, ncol2 = a.chk_w + 2 * a.radius;
  // layout: the tight pitches when the chunk fits them; schedule: four tables (one barrier per offset) when they fit
  // LDS beside the window, else two (nlm2_body.h)
  const bool tight = a.chk_w + 2 * a.reach <= NL2_WP_TIGHT && ncol2 + 1 <= NL2_TP_TIGHT && measuring_env("ANSEL_NLM2_LAYOUT") == nullptr;
  const int WP2 = tight ? NL2_WP_TIGHT : NL2_WP_LOOSE, TP2 = tight ? NL2_TP_TIGHT : NL2_TP_LOOSE;
  const bool deep = nlm2::lds_floats(4, a.chk_h, a.reach, a.npatch, WP2, TP2) * sizeof(float) <= 160 * 1024 && a.chk_h <= 64
                    && measuring_env("ANSEL_NLM2_DEEP") == nullptr;
  // the border workgroups of the same launch run the pipelined body: the launch's LDS is the larger of the two
  const size_t v2_bytes = std::max(nlm2::lds_floats(deep ? 4 : 2, a.chk_h, a.reach, a.npatch, WP2, TP2) * sizeof(float), pipe_bytes);
  const bool center = !(p.center_weight < 0); // denoise (profiled)'s weight: the second version's body takes it, the later ones do not
  bool v2 = pipelined && a.radius >= 1 && a.radius <= 3 && ncol2 * S2 <= NL2_PAR
            && a.chk_w * a.chk_h <= NL2_PAR * NL2_PX && a.chk_w + 2 * a.reach <= WP2 && ncol2 + 1 <= TP2 && a.npatch <= 4096
            && a.chk_h <= NL2_SERIAL / 2 && v2_bytes <= 160 * 1024 && measuring_env("ANSEL_HIP_NLM_V1") == nullptr;
  if(v2)
  {
    const int nseg = NL2_PAR / (ncol2 * S2), m0 = (a.chk_h - 2) / S2 + 1;
    v2 = (m0 + nseg - 1) / nseg <= NL2_MSEG;
  }
  // the third version where it applies (nlm3_body.h): the module's defaults on frames whose chunks have at most 56 rows
  int ndx3 = 0;
  const bool force_v2 = dispatch_override(DISPATCH_NLM_V2) || measuring_env("ANSEL_HIP_NLM_V2") != nullptr;
  // round 6: patch radius 1 (the <9, 7, 1> instantiations) beside 2, and the weight with the centre pixel's term where its division
  // has a divisor and a sharpness inside what nlm2_device_env::div_uniform() is exact / harmless for (anything else -- no preset
  // comes near -- keeps the second version's IEEE division)
  const bool p1 = a.radius == 1;
  const float cden = 1.0f + p.center_weight;
  const bool center_ok = !center || (cden >= 1.0f && cden <= 1048576.0f && p.sharpness >= 0x1p-60f && p.sharpness <= 0x1p60f);
  const bool v3 = v2 && center_ok && (p1 ? nlm3::fits<9, 7, 1>(a.chk_w, a.chk_h, a.radius, a.reach) : nlm3::fits<9, 6>(a.chk_w, a.chk_h, a.radius, a.reach))
                  && nlm3::regular_grid(patches.data(), a.npatch, &ndx3) && !force_v2;
  const size_t v3_bytes = std::max((p1 ? nlm3::lds_floats<9, 1>(a.chk_h, a.reach) : nlm3::lds_floats<9>(a.chk_h, a.reach)) * sizeof(float), pipe_bytes);
  // its fused variant for the grids the third version does not take (57 - 64 rows); ANSEL_HIP_NLM_FUSED=1: wherever it fits
  const size_t v4_bytes = std::max((p1 ? nlm3::lds_floats_fused<9, 1>(a.chk_h, a.reach) : nlm3::lds_floats_fused<9>(a.chk_h, a.reach)) * sizeof(float), pipe_bytes);
  const char *const fused_env = measuring_env("ANSEL_HIP_NLM_FUSED");
  const bool force_fused = dispatch_override(DISPATCH_NLM_FUSED) || (fused_env && atoi(fused_env) != 0);
  const bool v4 = v2 && center_ok
                  && (p1 ? nlm3::fits_fused<9, 7, 1>(a.chk_w, a.chk_h, a.radius, a.reach) : nlm3::fits_fused<9, 7>(a.chk_w, a.chk_h, a.radius, a.reach))
                  && nlm3::regular_grid(patches.data(), a.npatch, &ndx3)
                  && v4_bytes <= 160 * 1024 && !force_v2 && (force_fused || !(v3 && v3_bytes <= 160 * 1024));
  // the instantiation a launch takes: (patch radius, weight) -> kernel
  typedef void (*nlm3_kernel_t)(const float4 *, float4 *, nlm_args, const int2 *, const int *, int, int);
  typedef void (*nlm4_kernel_t)(const float4 *, float4 *, nlm_args, const int2 *, const int *, int, int, float *);
  const nlm3_kernel_t k3 = p1 ? (center ? nlm_chunks_v3<9, 7, 1, true> : nlm_chunks_v3<9, 7, 1, false>)
                              : (center ? nlm_chunks_v3<9, 6, 2, true> : nlm_chunks_v3<9, 6, 2, false>);
  const nlm4_kernel_t k4 = p1 ? (center ? nlm_chunks_v4<9, 7, false, 1, true> : nlm_chunks_v4<9, 7, false, 1, false>)
                              : (center ? nlm_chunks_v4<9, 7, false, 2, true> : nlm_chunks_v4<9, 7, false, 2, false>);
  // chunk grids of 65 - 69 rows (24 / 42 / 150 MP): the fused body on the first 64 rows of every interior chunk + nlm_tail
  static_assert(NLT_HEAD_ROWS == nlm3::TALL_HEAD && NLT_SEED_PITCH == nlm3::TALL_SEED_PITCH, "head and tail share the export's layout");
  const size_t tall_bytes = std::max((p1 ? nlm3::lds_floats_fused<9, 1>(NLT_HEAD_ROWS, a.reach) : nlm3::lds_floats_fused<9>(NLT_HEAD_ROWS, a.reach)) * sizeof(float), pipe_bytes);
  const size_t tail_bytes = nlmt::lds_floats(a.chk_h - NLT_HEAD_ROWS, a.reach, a.npatch) * sizeof(float);
  bool tall = v2 && center_ok && !v3 && !v4 && !force_v2
                    && (p1 ? nlmt::fits<1>(a.chk_w, a.chk_h, a.radius, a.reach, a.npatch) && nlm3::fits_fused<9, 7, 1>(a.chk_w, NLT_HEAD_ROWS, a.radius, a.reach)
                           : nlmt::fits<2>(a.chk_w, a.chk_h, a.radius, a.reach, a.npatch) && nlm3::fits_fused<9, 7>(a.chk_w, NLT_HEAD_ROWS, a.radius, a.reach))
                    && nlm3::regular_grid(patches.data(), a.npatch, &ndx3)
                    && tall_bytes <= 160 * 1024 && tail_bytes <= 64 * 1024;
  const nlm4_kernel_t k4t = p1 ? (center ? nlm_chunks_v4<9, 7, true, 1, true> : nlm_chunks_v4<9, 7, true, 1, false>)
                               : (center ? nlm_chunks_v4<9, 7, true, 2, true> : nlm_chunks_v4<9, 7, true, 2, false>);
  typedef void (*nlmt_kernel_t)(const float4 *, float4 *, nlm_args, const int2 *, const int *, int, const float *);
  const nlmt_kernel_t kt_ring = p1 ? (center ? nlm_tail<true, 1, true> : nlm_tail<true, 1, false>) : (center ? nlm_tail<true, 2, true> : nlm_tail<true, 2, false>);
  const nlmt_kernel_t kt_in = p1 ? (center ? nlm_tail<false, 1, true> : nlm_tail<false, 1, false>) : (center ? nlm_tail<false, 2, true> : nlm_tail<false, 2, false>);
  static_assert(NL2_SERIAL == NLP_SERIAL && NL2_THREADS == NLM_THREADS && NL3_THREADS == NLM_THREADS,
                "nlm_chunks_v2 / _v3 share the launch shape of nlm_chunks_pipelined");
  nlm2_kernel_t k2 = nullptr;
  const int nchunks = a.nchx * nchy;
  // the launch order of the chunks: those a patch can leave the frame from (the pipelined body) first
  std::vector<int> order;
  int n_border = 0;
  if(v2)
  {
    k2 = a.radius == 1 ? nlm2_kernel_of<1>(tight, deep, center)
                       : (a.radius == 2 ? nlm2_kernel_of<2>(tight, deep, center) : nlm2_kernel_of<3>(tight, deep, center));
#ifdef ANSEL_HIP_MEASURING
    if(a.radius == 2 && tight && deep && measuring_env("ANSEL_NLM2_TIMED")) k2 = nlm_chunks_v2_timed;
#endif
    if(v2_bytes > 64 * 1024)
      ANSEL_HIP_CHECK(hipFuncSetAttribute((const void *)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v2_bytes));
    const char *const var_env = measuring_env("ANSEL_NLM2_VARIANT");
    a.variant = var_env ? atoi(var_env) : 0;
    order.resize(nchunks);
    std::vector<int> inner;
    inner.reserve(nchunks);
    for(int c = 0; c < nchunks; c++)
    {
      const int cyl = c / a.nchx, cx = c - cyl * a.nchx, cy = cyl + a.cy0;
      const int top = cy * a.chk_h, left = cx * a.chk_w;
      const int bot = std::min(top + a.chk_h, a.H), right = std::min(left + a.chk_w, a.W);
      const bool interior = top >= a.reach && bot + a.reach <= a.H && left >= a.reach && right + a.reach <= a.W
                            && bot - top == a.chk_h && right - left == a.chk_w; // the test of nlm2::body()
      if(interior) inner.push_back(c);
      else order[n_border++] = c;
    }
    std::copy(inner.begin(), inner.end(), order.begin() + n_border);
  }
  // one upload: the patch shifts, then the chunk order
  const size_t patch_bytes = patches.size() * sizeof(int2), order_bytes = order.size() * sizeof(int);
  std::vector<unsigned char> host_blob(patch_bytes + order_bytes);
  memcpy(host_blob.data(), patches.data(), patch_bytes);
  if(order_bytes) memcpy(host_blob.data() + patch_bytes, order.data(), order_bytes);
  int2 *dev_patches = (int2 *)dt_hip_alloc_device_buffer(devid, host_blob.size());
  if(!dev_patches) return DT_HIP_SYSMEM_ALLOCATION;
  // (through the runtime's pinned staging ring: `host_blob` is a stack-lifetime buffer, and waiting for the stream here was a host
  // meeting in the middle of every frame)
  if(const int uerr = upload_small(devid, dev_patches, host_blob.data(), host_blob.size()); uerr != DT_HIP_SUCCESS)
  {
    dt_hip_release_mem_object(dev_patches);
    return uerr;
  }
  const int *const dev_order = (const int *)((const unsigned char *)dev_patches + patch_bytes);
  // the head's export: one column-sum row per chunk and offset (NLT_SEED_PITCH floats: ~14.5 B per pixel of the frame, written
  // once, read once -- the module's one per-call device allocation beyond in + out, dt_hip_iop_nlmeans_tiling()).  A frame
  // it does not fit beside keeps the second version's body, which needs nothing of the kind
  float *seeds = nullptr;
  if(tall)
  {
    seeds = (float *)dt_hip_alloc_device_buffer(devid, (size_t)nchunks * a.npatch * NLT_SEED_PITCH * sizeof(float));
    if(!seeds) tall = false;
  }
  // the opt-in to more than 64 KB of LDS, before anything is launched: a failure leaves nothing behind
  hipError_t attr_err = hipSuccess;
  if(tall)
    attr_err = hipFuncSetAttribute((const void *)k4t, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tall_bytes);
  else if(v4)
    attr_err = hipFuncSetAttribute((const void *)k4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v4_bytes);
  else if(v3 && v3_bytes <= 160 * 1024)
    attr_err = hipFuncSetAttribute((const void *)k3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v3_bytes);
  if(attr_err != hipSuccess)
  {
    set_last_error("%s:%d hipFuncSetAttribute(nlm_chunks): %s", __FILE__, __LINE__, hipGetErrorString(attr_err));
    if(seeds) dt_hip_release_mem_object(seeds);
    dt_hip_release_mem_object(dev_patches);
    return DT_HIP_DEFAULT_ERROR;
  }
  {
    launch_scope ls(devid, "nlm_chunks");
    const unsigned grid = (unsigned)nchunks;
    if(tall)
      k4t<<<grid, NL3_THREADS, tall_bytes, s>>>(in, out, a, dev_patches, dev_order, n_border, ndx3, seeds);
    else if(v4)
      k4<<<grid, NL3_THREADS, v4_bytes, s>>>(in, out, a, dev_patches, dev_order, n_border, ndx3, nullptr);
    else if(v3 && v3_bytes <= 160 * 1024)
      k3<<<grid, NL3_THREADS, v3_bytes, s>>>(in, out, a, dev_patches, dev_order, n_border, ndx3);
    else if(v2)
      k2<<<grid, NL2_THREADS, v2_bytes, s>>>(in, out, a, dev_patches, dev_order, n_border);
    else if(pipelined)
      nlm_chunks_pipelined<<<grid, NLM_THREADS, lds_bytes, s>>>(in, out, a, dev_patches);
    else if(staged)
      nlm_chunks<true><<<grid, NLM_THREADS, lds_bytes, s>>>(in, out, a, dev_patches);
    else
      nlm_chunks<false><<<grid, NLM_THREADS, lds_bytes, s>>>(in, out, a, dev_patches);
  }
  int err = check_launch("nlm_chunks");
  if(err == DT_HIP_SUCCESS && tall)
  {
    launch_scope ls(devid, "nlm_tail");
    if(n_border > 0) kt_ring<<<(unsigned)n_border, NLT_THREADS, tail_bytes, s>>>(in, out, a, dev_patches, dev_order, 0, seeds);
    if(nchunks > n_border)
      kt_in<<<(unsigned)(nchunks - n_border), NLT_THREADS, tail_bytes, s>>>(in, out, a, dev_patches, dev_order, n_border, seeds);
    err = check_launch("nlm_tail");
  }
  if(seeds) dt_hip_release_mem_object(seeds);
  dt_hip_release_mem_object(dev_patches);
  return err;
}

} // namespace ansel

extern "C" {

int dt_hip_iop_nlmeans_process(int devid, const dt_hip_piece_t *piece, const dt_hip_nlmeans_data_t *d,
                               dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  return nlmeans_process_band(devid, piece, d, nullptr, dev_in, dev_out);
}

// tiling_callback(), src/iop/nlmeans.c:400-414 (factor: in + out + tmp + the per-thread column sums, NUM_BUCKETS 4);
// on the device the sums live in LDS: in + out -- and, on a chunk grid of 65 - 69 rows (the 24 / 42 / 150 MP frames), the head's
// export for the tail kernel: one 80-float row of column sums per chunk and offset, at most 225 x 320 B per 72 x 65 pixels = 0.96 of
// a float4 plane (nlmeans_core_launch(): `seeds`; the module falls back to a body without it when the allocation fails)
void dt_hip_iop_nlmeans_tiling(const dt_hip_piece_t *piece, const dt_hip_nlmeans_data_t *d, dt_hip_tiling_t *tiling)
{
  const float scale = (float)fmin(piece->roi_in.scale, 2.0f);
  memset(tiling, 0, sizeof(*tiling));
  tiling->factor = 2.0f + 1.0f + 0.25f * 4;
  tiling->factor_cl = 2.0f + (slice_height(piece->roi_in.height) > NLT_HEAD_ROWS ? 0.96f : 0.0f);
  tiling->maxbuf = tiling->maxbuf_cl = 1.0f;
  tiling->overlap = (unsigned)((int)ceilf(d->radius * scale) + (int)ceilf(7 * scale));
  tiling->xalign = tiling->yalign = 1;
}

} // extern "C"

namespace ansel
{

static nlm_core_params_t nlmeans_params(const dt_hip_piece_t *piece, const dt_hip_nlmeans_data_t *d);

// rows/columns of input a chunk's patches touch beyond the chunk (a.reach of the launch) + the chunk height:
// what a row band needs from its neighbours (pipe.cpp)
int nlmeans_core_halo_rows(const int frame_h, const nlm_core_params_t &p)
{
  const int K = p.search_radius;
  int max_shift = 0;
  for(int ri = -K; ri <= K; ri++)
    for(int ci = -K; ci <= K; ci++)
      max_shift = std::max(max_shift, std::max(abs(scatter(p.scale, p.scattering, ri, ci)), abs(scatter(p.scale, p.scattering, ci, ri))));
  return p.patch_radius + 1 + max_shift + slice_height(frame_h) - 1;
}

int nlmeans_halo_rows(const dt_hip_piece_t *piece, const dt_hip_nlmeans_data_t *d)
{
  return nlmeans_core_halo_rows(piece->roi_out.height, nlmeans_params(piece, d));
}

int nlmeans_process_band(int devid, const dt_hip_piece_t *piece, const dt_hip_nlmeans_data_t *d, const band_view_t *band,
                         dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  if(piece->channels != 4) return DT_HIP_INVALID_ARG;
  nlm_core_params_t p = nlmeans_params(piece, d);
  p.band = band;
  return nlmeans_core_launch(devid, (const float4 *)dev_in, (float4 *)dev_out, piece->roi_out.width,
                             piece->roi_out.height, p);
}

// the module with the lightness cells of its output pixels written beside them (pipe.cpp: local contrast's bilateral grid follows)
int nlmeans_process_cells(int devid, const dt_hip_piece_t *piece, const dt_hip_nlmeans_data_t *d, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out,
                          dt_hip_mem_t cells, float sigma_r, int size_z)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out || !cells) return DT_HIP_INVALID_ARG;
  if(piece->channels != 4) return DT_HIP_INVALID_ARG;
  nlm_core_params_t p = nlmeans_params(piece, d);
  p.cell_out = (float2 *)cells;
  p.cell_sigma_r = sigma_r;
  p.cell_size_z = size_z;
  return nlmeans_core_launch(devid, (const float4 *)dev_in, (float4 *)dev_out, piece->roi_out.width, piece->roi_out.height, p);
}

// process_cpu(), src/iop/nlmeans.c:416-457
static nlm_core_params_t nlmeans_params(const dt_hip_piece_t *piece, const dt_hip_nlmeans_data_t *d)
{
  const float scale = (float)fmin(piece->roi_in.scale, 2.0f);
  const float max_L = 120.0f, max_C = 512.0f;
  const float nL = 1.0f / max_L, nC = 1.0f / max_C;
  nlm_core_params_t p;
  memset(&p, 0, sizeof(p));
  p.scattering = 0;
  p.scale = scale;
  p.luma = d->luma;
  p.chroma = d->chroma;
  p.center_weight = -1;
  p.sharpness = 3000.0f / (1.0f + d->strength);
  p.patch_radius = (int)ceilf(d->radius * scale);
  p.search_radius = (int)ceilf(7 * scale);
  p.norm[0] = nL * nL;
  p.norm[1] = nC * nC;
  p.norm[2] = nC * nC;
  p.norm[3] = 1.0f;
  return p;
}

} // namespace ansel
