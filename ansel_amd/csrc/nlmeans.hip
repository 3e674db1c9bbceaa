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

#include <math.h>
#include <algorithm>
#include <vector>

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
};

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
  const int cy = blockIdx.x / a.nchx, cx = blockIdx.x - cy * a.nchx;
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
  }
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
  a.W = width;
  a.H = height;
  a.chk_h = slice_height(height);
  a.chk_w = slice_width(width);
  a.nchx = (width + a.chk_w - 1) / a.chk_w;
  const int nchy = (height + a.chk_h - 1) / a.chk_h;
  a.radius = p.patch_radius;
  a.npatch = (int)patches.size();
  a.cs_pitch = (a.chk_w + 2 * a.radius + 1) | 1; // odd pitches: the row-parallel step strides whole rows
  a.wt_pitch = (a.chk_w + 16) | 1; // 16 spare columns: the batched row recurrence stores whole batches
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
  const size_t table_bytes = ((size_t)(a.chk_h + 16) * a.cs_pitch + (size_t)a.chk_h * a.wt_pitch + 64) * sizeof(float);
  // the window a chunk's patches can touch: init_column_sums() reads radius rows/columns beyond the
  // chunk, the recurrence one more row below, and everything once more shifted by the patch offset
  a.reach = a.radius + 1 + max_shift;
  a.win_pitch = (a.chk_w + 2 * a.reach) | 1;
  const size_t window_bytes = (size_t)3 * (a.chk_h + 2 * a.reach) * a.win_pitch * sizeof(float);
  const bool staged = table_bytes + window_bytes <= 160 * 1024;
  const size_t lds_bytes = table_bytes + (staged ? window_bytes : 0);
  hipStream_t s = stream_of(devid);
  int2 *dev_patches = (int2 *)dt_hip_alloc_device_buffer(devid, patches.size() * sizeof(int2));
  if(!dev_patches) return DT_HIP_SYSMEM_ALLOCATION;
  if(hipMemcpyAsync(dev_patches, patches.data(), patches.size() * sizeof(int2), hipMemcpyHostToDevice, s) != hipSuccess
     || hipStreamSynchronize(s) != hipSuccess) // `patches` is a stack-lifetime host buffer
  {
    dt_hip_release_mem_object(dev_patches);
    return DT_HIP_DEFAULT_ERROR;
  }
  if(lds_bytes > 64 * 1024)
    ANSEL_HIP_CHECK(hipFuncSetAttribute(staged ? (const void *)nlm_chunks<true> : (const void *)nlm_chunks<false>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  {
    launch_scope ls(devid, "nlm_chunks");
    if(staged)
      nlm_chunks<true><<<(unsigned)(a.nchx * nchy), NLM_THREADS, lds_bytes, s>>>(in, out, a, dev_patches);
    else
      nlm_chunks<false><<<(unsigned)(a.nchx * nchy), NLM_THREADS, lds_bytes, s>>>(in, out, a, dev_patches);
  }
  dt_hip_release_mem_object(dev_patches);
  return check_launch("nlm_chunks");
}

} // namespace ansel

extern "C" {

// process_cpu(), src/iop/nlmeans.c:416-457
int dt_hip_iop_nlmeans_process(int devid, const dt_hip_piece_t *piece, const dt_hip_nlmeans_data_t *d,
                               dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  if(piece->channels != 4) return DT_HIP_INVALID_ARG;
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
  return nlmeans_core_launch(devid, (const float4 *)dev_in, (float4 *)dev_out, piece->roi_out.width,
                             piece->roi_out.height, p);
}

} // extern "C"
