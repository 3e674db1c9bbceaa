// guided_filter.hip -- the guided filter that feathers a blend mask, on gfx950.
//
// Reference: guided_filter(), src/pixel/guided_filter.c:369-402; its per-tile body guided_filter_tiling(), :122-330; the
// compensated box means of src/pixel/box_filters.c (:408-506 rows, :577-888 columns).  The reference's own device
// version (guided_filter_cl_impl(), :560-760) uses plain running sums and no tiles, so it is NOT what its CPU path
// computes; this file follows the CPU path, the one the oracle pins:
//   * the tile grid is part of the result: targets of max(3 w, 512) pixels a side, each filtered on its own source
//     region (target grown by 2 w, clipped at the image) -- every box mean restarts at a source border;
//   * a box mean is a sliding Kahan sum over one scan line, in scan order, divided by the samples under the window: one
//     lane per scan line, lanes along the contiguous axis of the plane (the planes are transposed in front of the row
//     pass and back), 13 lines per pixel row / column (mask, 3 guide means, 3 covariances, 6 variances), then 4 (the
//     coefficients);
//   * box_filters.c's 1-wide column variant adds the sample leaving the window in its tail instead of subtracting it
//     (:630-640).  It runs on the last (9 * source width) % 4 scalar columns of the reference's interleaved variance
//     image; the same columns get the same tail here.
// Layout: per tile, 13 planes of its source size, plane-major, two sets (every pass reads one and writes the other, so a
// line's earlier samples are still there when they leave the window).  All tiles of a batch run in the same launches;
// a batch is as many tiles as fit the budget below (104 B per source pixel).
#include "hip_common.h"

#include <algorithm>
#include <vector>

using namespace ansel;

namespace
{

struct gf_tile
{
  int sl, slo, sw, sh;     // source region: left, lower, width, height
  int tl, tr, tlo, tup;    // target region
  size_t off;              // first float of the tile's planes in a plane set
};

constexpr int GF_PLANES = 13;                        // 0: mask, 1-3: guide, 4-6: covariance, 7-12: variance
constexpr size_t GF_BATCH_MAX = (size_t)8 << 30;     // both plane sets of one batch, at most (and at most half of what is free)

// Kahan_sum(), src/math/math.h:105-111
__device__ __forceinline__ float kahan(const float m, float &c, const float add)
{
  const float t1 = add - c;
  const float t2 = m + t1;
  c = (t2 - m) - t1;
  return t2;
}

// one scan line of n samples `stride` floats apart: src -> dst
__device__ __forceinline__ void box_mean_line(const float *__restrict__ src, float *__restrict__ dst, const size_t stride,
                                              const int n, const int radius, const bool tail_adds)
{
  float L = 0.0f, c = 0.0f;
  int hits = 0;
  const int head = radius < n ? radius : n;
  for(int x = 0; x < head; x++)
  {
    hits++;
    L = kahan(L, c, src[(size_t)x * stride]);
  }
  int x;
  for(x = 0; x <= radius && x + radius < n; x++)
  {
    hits++;
    L = kahan(L, c, src[(size_t)(x + radius) * stride]);
    dst[(size_t)x * stride] = L / (float)hits;
  }
  for(; x <= radius && x < n; x++) dst[(size_t)x * stride] = L / (float)hits;
  for(; x + radius < n; x++)
  {
    L = kahan(L, c, -src[(size_t)(x - radius - 1) * stride]);
    L = kahan(L, c, src[(size_t)(x + radius) * stride]);
    dst[(size_t)x * stride] = L / (float)hits;
  }
  for(; x < n; x++)
  {
    hits--;
    const float v = src[(size_t)(x - radius - 1) * stride];
    L = kahan(L, c, tail_adds ? v : -v);
    dst[(size_t)x * stride] = L / (float)hits;
  }
}

// guided_filter_tiling(), :166-192: the 13 products of a source pixel
__global__ __launch_bounds__(256) void gf_moments(const float4 *__restrict__ guide, const float *__restrict__ in,
                                                  float *__restrict__ A, const gf_tile *__restrict__ tiles, const int iw,
                                                  const float guide_weight)
{
  const gf_tile t = tiles[blockIdx.y];
  const size_t size = (size_t)t.sw * t.sh;
  for(size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < size; k += (size_t)gridDim.x * blockDim.x)
  {
    const int j = (int)(k / t.sw), i = (int)(k - (size_t)j * t.sw);
    const size_t g = (size_t)(t.slo + j) * iw + t.sl + i;
    const float4 px = guide[g];
    const float p0 = px.x * guide_weight, p1 = px.y * guide_weight, p2 = px.z * guide_weight;
    const float input = in[g];
    float *const a = A + t.off + k;
    a[0] = input;
    a[size] = p0;
    a[2 * size] = p1;
    a[3 * size] = p2;
    a[4 * size] = p0 * input;
    a[5 * size] = p1 * input;
    a[6 * size] = p2 * input;
    a[7 * size] = p0 * p0;
    a[8 * size] = p0 * p1;
    a[9 * size] = p0 * p2;
    a[10 * size] = p1 * p1;
    a[11 * size] = p1 * p2;
    a[12 * size] = p2 * p2;
  }
}

// One scan line per lane, lanes along the CONTIGUOUS axis of the plane, the scan along the other one (`stride` floats
// apart): 64 lanes read 256 contiguous bytes per step.  A first version scanned the rows of a row-major plane with
// one lane per row -- a lane stride of one image row, 64 cache lines per step -- and spent 22.7 of the filter's 25.4 ms
// (24 MP, w = 20) there, against 0.8 ms for the columns; now the planes are transposed in front of the row pass and
// back behind it (gf_transpose, at copy speed) and both passes are this kernel.
//   transposed == 0: columns of a row-major plane (box_mean_vert_1ch_Kahan() over the scalar columns of the reference's
//                    interleaved image; `interleave` = its plane count when the 1-wide variant can occur (9), else 0)
//   transposed == 1: the plane holds element (row, col) at col * sh + row; the scan runs along the image ROWS
//                    (blur_horizontal_4ch_Kahan() / blur_horizontal_Nch_Kahan(): no 1-wide variant there)
__global__ __launch_bounds__(64) void gf_scan(const float *__restrict__ src, float *__restrict__ dst,
                                              const gf_tile *__restrict__ tiles, const int planes, const int radius,
                                              const int interleave, const int transposed)
{
  const gf_tile t = tiles[blockIdx.y];
  const int lanes_n = transposed ? t.sh : t.sw; // scan lines per plane = extent of the contiguous axis
  const int n = transposed ? t.sw : t.sh;       // samples per scan line
  const int line = blockIdx.x * blockDim.x + threadIdx.x; // plane * lanes_n + position on the contiguous axis
  if(line >= planes * lanes_n) return;
  const int c = line / lanes_n, i = line - c * lanes_n;
  bool tail_adds = false;
  if(!transposed && interleave && c >= 4)
  {
    const size_t cols = (size_t)interleave * t.sw, k = (size_t)i * interleave + (c - 4);
    tail_adds = k >= (cols & ~(size_t)3);
  }
  const size_t o = t.off + (size_t)c * t.sw * t.sh + i;
  box_mean_line(src + o, dst + o, (size_t)lanes_n, n, radius, tail_adds);
}

// planes of a tile, row-major (rows x cols) -> (cols x rows), 32 x 32 elements per workgroup through LDS;
// to_transposed: rows = sh, cols = sw; else the way back
__global__ __launch_bounds__(256) void gf_transpose(const float *__restrict__ src, float *__restrict__ dst,
                                                    const gf_tile *__restrict__ tiles, const int to_transposed)
{
  __shared__ float tile[32][33];
  const gf_tile t = tiles[blockIdx.z];
  const int rows = to_transposed ? t.sh : t.sw, cols = to_transposed ? t.sw : t.sh;
  const int bpr = (cols + 31) / 32, nblocks = bpr * ((rows + 31) / 32);
  if((int)blockIdx.x >= nblocks) return;
  const int by = blockIdx.x / bpr, bx = blockIdx.x - by * bpr;
  const size_t base = t.off + (size_t)blockIdx.y * t.sw * t.sh;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
#pragma unroll
  for(int k = 0; k < 4; k++)
  {
    const int r = by * 32 + ty + 8 * k, cc = bx * 32 + tx;
    if(r < rows && cc < cols) tile[ty + 8 * k][tx] = src[base + (size_t)r * cols + cc];
  }
  __syncthreads();
#pragma unroll
  for(int k = 0; k < 4; k++)
  {
    const int cc = bx * 32 + ty + 8 * k, r = by * 32 + tx; // output row = input column
    if(cc < cols && r < rows) dst[base + (size_t)cc * rows + r] = tile[tx][ty + 8 * k];
  }
}

// the coefficients a_r, a_g, a_b, b over the means, :223-287: A (13 planes) -> B planes 0..3
__global__ __launch_bounds__(256) void gf_solve(const float *__restrict__ A, float *__restrict__ B,
                                                const gf_tile *__restrict__ tiles, const float eps)
{
  const gf_tile t = tiles[blockIdx.y];
  const size_t size = (size_t)t.sw * t.sh;
  for(size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < size; k += (size_t)gridDim.x * blockDim.x)
  {
    const float *const m = A + t.off + k;
    const float inp_mean = m[0], guide_r = m[size], guide_g = m[2 * size], guide_b = m[3 * size];
    const float S00 = m[7 * size] - (guide_r * guide_r) + eps;
    const float S01 = m[8 * size] - (guide_r * guide_g);
    const float S02 = m[9 * size] - (guide_r * guide_b);
    const float S11 = m[10 * size] - (guide_g * guide_g) + eps;
    const float S12 = m[11 * size] - (guide_g * guide_b);
    const float S22 = m[12 * size] - (guide_b * guide_b) + eps;
    const float det0 = S00 * (S11 * S22 - S12 * S12) - S01 * (S01 * S22 - S02 * S12) + S02 * (S01 * S12 - S02 * S11);
    float ar, ag, ab, b;
    if(fabsf(det0) > 4.f * 1.1920928955078125e-7f)
    {
      const float cov_r = m[4 * size] - guide_r * inp_mean;
      const float cov_g = m[5 * size] - guide_g * inp_mean;
      const float cov_b = m[6 * size] - guide_b * inp_mean;
      const float det1 = cov_r * (S11 * S22 - S12 * S12) - S01 * (cov_g * S22 - cov_b * S12) + S02 * (cov_g * S12 - cov_b * S11);
      const float det2 = S00 * (cov_g * S22 - cov_b * S12) - cov_r * (S01 * S22 - S02 * S12) + S02 * (S01 * cov_b - S02 * cov_g);
      const float det3 = S00 * (S11 * cov_b - S12 * cov_g) - S01 * (S01 * cov_b - S02 * cov_g) + cov_r * (S01 * S12 - S02 * S11);
      ar = det1 / det0;
      ag = det2 / det0;
      ab = det3 / det0;
      b = inp_mean - ar * guide_r - ag * guide_g - ab * guide_b;
    }
    else
    {
      ar = ag = ab = 0.f;
      b = inp_mean;
    }
    float *const o = B + t.off + k;
    o[0] = ar;
    o[size] = ag;
    o[2 * size] = ab;
    o[3 * size] = b;
  }
}

// :299-318: the target pixels from the averaged coefficients (planes 0..3 of P)
__global__ __launch_bounds__(256) void gf_apply(const float4 *__restrict__ guide, const float *__restrict__ P,
                                                float *__restrict__ out, const gf_tile *__restrict__ tiles, const int iw,
                                                const float guide_weight, const float minv, const float maxv)
{
  const gf_tile t = tiles[blockIdx.y];
  const int tw = t.tr - t.tl, th = t.tup - t.tlo;
  const size_t size = (size_t)t.sw * t.sh, n = (size_t)tw * th;
  for(size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x)
  {
    const int jj = (int)(k / tw), ii = (int)(k - (size_t)jj * tw);
    const int j = t.tlo + jj, i = t.tl + ii;
    const float4 px = guide[(size_t)j * iw + i];
    const float *const ab = P + t.off + (size_t)(j - t.slo) * t.sw + (i - t.sl);
    float res = guide_weight * (ab[0] * px.x + ab[size] * px.y + ab[2 * size] * px.z);
    res += ab[3 * size];
    out[(size_t)j * iw + i] = (res > maxv) ? maxv : ((res < minv) ? minv : res); // glib's CLAMP()
  }
}

} // namespace

namespace ansel
{

// `mask` (width x height floats) is filtered in place, guided by `guide` (width x height float4 pixels);
// _develop_blend_process_feather(), blend.c:603-623, calls it with sqrt_eps 1, range [0, 1]
int guided_filter_launch(int devid, const float4 *guide, float *mask, int width, int height, int w, float sqrt_eps,
                         float guide_weight, float minv, float maxv)
{
  if(width <= 0 || height <= 0 || w < 1) return DT_HIP_INVALID_ARG;
  hipStream_t s = stream_of(devid);
  const int tile = 3 * w > 512 ? 3 * w : 512;
  const float eps = sqrt_eps * sqrt_eps;
  std::vector<gf_tile> tiles;
  for(int j = 0; j < height; j += tile)
    for(int i = 0; i < width; i += tile)
    {
      gf_tile t;
      t.tl = i;
      t.tr = std::min(i + tile, width);
      t.tlo = j;
      t.tup = std::min(j + tile, height);
      t.sl = std::max(t.tl - 2 * w, 0);
      t.slo = std::max(t.tlo - 2 * w, 0);
      t.sw = std::min(t.tr + 2 * w, width) - t.sl;
      t.sh = std::min(t.tup + 2 * w, height) - t.slo;
      t.off = 0;
      tiles.push_back(t);
    }
  // the input of every tile is the mask as it was: a copy, like the reference's mask_bak
  const size_t np = (size_t)width * height;
  float *bak = (float *)dt_hip_alloc_device_buffer(devid, np * sizeof(float));
  if(!bak) return DT_HIP_SYSMEM_ALLOCATION;
  int err = DT_HIP_SUCCESS;
  if(hipMemcpyAsync(bak, mask, np * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) err = DT_HIP_DEFAULT_ERROR;
  // the batch budget follows the device: batching does not change the result, so a device with little free memory (or
  // a caller inside the host tiler's budget) runs more, smaller batches instead of failing
  const size_t GF_BATCH_BYTES = std::max<size_t>(std::min<size_t>(GF_BATCH_MAX, (size_t)dt_hip_get_device_available(devid) / 2),
                                                 (size_t)64 << 20);
  size_t first = 0;
  while(first < tiles.size() && err == DT_HIP_SUCCESS)
  {
    // as many tiles as the budget takes (at least one)
    size_t last = first, floats = 0;
    int max_sw = 0, max_sh = 0;
    size_t max_size = 0;
    while(last < tiles.size())
    {
      const size_t add = (size_t)tiles[last].sw * tiles[last].sh * GF_PLANES;
      if(last > first && (floats + add) * 2 * sizeof(float) > GF_BATCH_BYTES) break;
      tiles[last].off = floats;
      floats += add;
      max_sw = std::max(max_sw, tiles[last].sw);
      max_sh = std::max(max_sh, tiles[last].sh);
      max_size = std::max(max_size, (size_t)tiles[last].sw * tiles[last].sh);
      last++;
    }
    const unsigned nt = (unsigned)(last - first);
    float *A = (float *)dt_hip_alloc_device_buffer(devid, floats * sizeof(float));
    float *B = (float *)dt_hip_alloc_device_buffer(devid, floats * sizeof(float));
    gf_tile *dt = (gf_tile *)dt_hip_alloc_device_buffer(devid, nt * sizeof(gf_tile));
    if(!A || !B || !dt) err = DT_HIP_SYSMEM_ALLOCATION;
    else if(hipMemcpyAsync(dt, tiles.data() + first, nt * sizeof(gf_tile), hipMemcpyHostToDevice, s) != hipSuccess
            || hipStreamSynchronize(s) != hipSuccess) // `tiles` is a stack-lifetime host buffer
      err = DT_HIP_DEFAULT_ERROR;
    if(err == DT_HIP_SUCCESS)
    {
      launch_scope ls(devid, "guided_filter");
      const unsigned gpix = (unsigned)std::min<size_t>((max_size + 255) / 256, 4096);
      const int max_dim = std::max(max_sw, max_sh);
      const unsigned tblocks = (unsigned)(((max_sw + 31) / 32) * ((max_sh + 31) / 32));
      const unsigned scan13 = (unsigned)((GF_PLANES * max_dim + 63) / 64), scan4 = (unsigned)((4 * max_dim + 63) / 64);
      gf_moments<<<dim3(gpix, nt), 256, 0, s>>>(guide, bak, A, dt, width, guide_weight);
      gf_transpose<<<dim3(tblocks, GF_PLANES, nt), 256, 0, s>>>(A, B, dt, 1);
      gf_scan<<<dim3(scan13, nt), 64, 0, s>>>(B, A, dt, GF_PLANES, w, 0, 1); // rows
      gf_transpose<<<dim3(tblocks, GF_PLANES, nt), 256, 0, s>>>(A, B, dt, 0);
      gf_scan<<<dim3(scan13, nt), 64, 0, s>>>(B, A, dt, GF_PLANES, w, 9, 0); // columns
      gf_solve<<<dim3(gpix, nt), 256, 0, s>>>(A, B, dt, eps);
      gf_transpose<<<dim3(tblocks, 4, nt), 256, 0, s>>>(B, A, dt, 1);
      gf_scan<<<dim3(scan4, nt), 64, 0, s>>>(A, B, dt, 4, w, 0, 1);
      gf_transpose<<<dim3(tblocks, 4, nt), 256, 0, s>>>(B, A, dt, 0);
      gf_scan<<<dim3(scan4, nt), 64, 0, s>>>(A, B, dt, 4, w, 0, 0);
      gf_apply<<<dim3(gpix, nt), 256, 0, s>>>(guide, B, mask, dt, width, guide_weight, minv, maxv);
      err = check_launch("guided_filter");
    }
    if(A) dt_hip_release_mem_object(A);
    if(B) dt_hip_release_mem_object(B);
    if(dt) dt_hip_release_mem_object(dt);
    first = last;
  }
  dt_hip_release_mem_object(bak);
  return err;
}

} // namespace ansel
