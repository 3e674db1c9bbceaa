// pointwise.hip -- the streaming glue modules of the raw -> RGB export pipe as gfx950 kernels:
//   rawprepare   src/iop/rawprepare.c:467-520        u16|f32 CFA -> normalised f32 CFA   (6 B/px)
//   temperature  src/iop/temperature.c:487-600       CFA or RGBA times WB coefficients   (8 | 32 B/px)
//   highlights   src/iop/highlights.c:680-789 (clip) MIN(clip, in) with the <25 bypass   (8 | 32 B/px)
//   exposure     src/iop/exposure.c:501-545          (in - black) * scale                (8 | 32 B/px)
//   export u16/u8 src/imageio/imageio_core.c:706-737 RGBA f32 -> RGBA u16 / u8           (24 | 20 B/px)
//
// All are HBM-bound: one read and one write per element, 16 B per lane per access wherever
// the geometry allows, grid-stride over <= 8192 workgroups of 256 threads.  Arithmetic is one
// IEEE binary32 operation per reference operation, in the reference's order; the file is
// compiled with -ffp-contract=off so nothing is fused that the reference does not fuse.
#include "hip_common.h"

using namespace ansel;

namespace
{

__device__ __forceinline__ int fc(const int row, const int col, const uint32_t filters)
{
  // FC(), src/develop/imageop_math.h:190-193
  return filters >> ((((row << 1) & 14) + (col & 1)) << 1) & 3;
}

// ---------------------------------------------------------------------------------------
// rawprepare
// ---------------------------------------------------------------------------------------
struct rawprepare_args
{
  int width, height, in_width;
  int csx, csy;
  int cfa_x, cfa_y;
  float sub[4], inv_div[4];
};

// One thread = 4 consecutive photosites of one output row (x0 is a multiple of 4, so the
// Bayer column phase of lane k is (cfa_x + k) & 1).
template <typename in_t, bool VEC>
__global__ __launch_bounds__(256) void rawprepare_1f(const in_t *__restrict__ in, float *__restrict__ out,
                                                      const rawprepare_args a)
{
  const int quads = (a.width + 3) >> 2;
  const size_t total = (size_t)quads * a.height;
  for(size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x)
  {
    const int j = (int)(t / quads);
    const int x0 = (int)(t - (size_t)j * quads) << 2;
    const int row_phase = ((j + a.cfa_y) & 1) << 1;
    const int x_phase = a.cfa_x & 1;
    const int id0 = row_phase + x_phase, id1 = row_phase + (x_phase ^ 1);
    const float sub0 = a.sub[id0], sub1 = a.sub[id1];
    const float inv0 = a.inv_div[id0], inv1 = a.inv_div[id1];
    const size_t pin = (size_t)a.in_width * (j + a.csy) + a.csx + x0;
    const size_t pout = (size_t)j * a.width + x0;
    if(VEC)
    {
      float v0, v1, v2, v3;
      if(sizeof(in_t) == 2)
      {
        const ushort4 r = *reinterpret_cast<const ushort4 *>(in + pin);
        v0 = (float)r.x; v1 = (float)r.y; v2 = (float)r.z; v3 = (float)r.w;
      }
      else
      {
        const float4 r = *reinterpret_cast<const float4 *>(in + pin);
        v0 = r.x; v1 = r.y; v2 = r.z; v3 = r.w;
      }
      float4 o;
      o.x = (v0 - sub0) * inv0;
      o.y = (v1 - sub1) * inv1;
      o.z = (v2 - sub0) * inv0;
      o.w = (v3 - sub1) * inv1;
      nt_store(reinterpret_cast<float4 *>(out + pout), o);
    }
    else
    {
      const int n = min(4, a.width - x0);
      for(int k = 0; k < n; k++)
      {
        const float s = (k & 1) ? sub1 : sub0;
        const float iv = (k & 1) ? inv1 : inv0;
        out[pout + k] = ((float)in[pin + k] - s) * iv;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// temperature
// ---------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void temperature_1f(const float *__restrict__ in, float *__restrict__ out,
                                                       const int width, const int height, const int rx, const int ry,
                                                       const uint32_t filters, const float4 coeffs)
{
  const float cf[4] = { coeffs.x, coeffs.y, coeffs.z, coeffs.w };
  const int quads = (width + 3) >> 2;
  const size_t total = (size_t)quads * height;
  for(size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x)
  {
    const int j = (int)(t / quads);
    const int x0 = (int)(t - (size_t)j * quads) << 2;
    const float c0 = cf[fc(j + ry, x0 + rx, filters)];
    const float c1 = cf[fc(j + ry, x0 + rx + 1, filters)];
    const size_t p = (size_t)j * width + x0;
    if(VEC)
    {
      const float4 r = *reinterpret_cast<const float4 *>(in + p);
      float4 o;
      o.x = r.x * c0; o.y = r.y * c1; o.z = r.z * c0; o.w = r.w * c1;
      nt_store(reinterpret_cast<float4 *>(out + p), o);
    }
    else
    {
      const int n = min(4, width - x0);
      for(int k = 0; k < n; k++) out[p + k] = in[p + k] * ((k & 1) ? c1 : c0);
    }
  }
}

__global__ __launch_bounds__(256) void temperature_4f(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                       const size_t npixels, const float4 coeffs)
{
  for(size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < npixels; k += (size_t)gridDim.x * blockDim.x)
  {
    const float4 r = in[k];
    float4 o;
    o.x = r.x * coeffs.x; o.y = r.y * coeffs.y; o.z = r.z * coeffs.z; o.w = r.w;
    nt_store(out + k, o);
  }
}

// ---------------------------------------------------------------------------------------
// highlights (clip)
// ---------------------------------------------------------------------------------------
// The reference counts the photosites above the clip threshold first and copies the input
// through untouched when there are fewer than DT_HL_MIN_CLIPPED_PIXELS = 25 of them
// (highlights.c:266-300, :728-733).  A separate counting pass would double the HBM traffic
// of an 8 B/px module, so one pass clips, counts, and journals the first 25 clipped
// photosites {index, original value}; a one-wave epilogue restores them if the bypass
// condition turns out to hold.  Bit-exact either way.
__device__ __forceinline__ void hl_note(hl_journal *j, const bool over, const size_t index, const float value)
{
  const unsigned long long mask = __ballot(over);
  if(mask == 0ull) return;
  const int lane = threadIdx.x & 63;
  const int n = __popcll(mask);
  unsigned long long base = 0;
  const int leader = __ffsll((long long)mask) - 1;
  if(lane == leader) base = atomicAdd(&j->count, (unsigned long long)n);
  base = __shfl(base, leader);
  if(over && base < HL_MIN_CLIPPED)
  {
    const unsigned long long rank = base + __popcll(mask & ((1ull << lane) - 1ull));
    if(rank < HL_MIN_CLIPPED)
    {
      j->index[rank] = index + 1; // 0 = empty slot
      j->value[rank] = value;
    }
  }
}

// elements: CFA photosites (ch = 1) -- every element compared against one threshold
__global__ __launch_bounds__(256) void highlights_clip_1f(const float *__restrict__ in, float *__restrict__ out,
                                                           const size_t n, const float clip, const float threshold,
                                                           hl_journal *journal)
{
  const size_t nvec = n >> 2;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  // whole-wave trip count so that __ballot() sees every lane
  const size_t iters = (nvec + stride - 1) / stride;
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  // Once 25 clipped photosites are known the exact count is irrelevant (only `count < 25` is
  // ever tested), so a wave stops touching the journal for good: on a frame with 1 % blown
  // highlights every wave would otherwise queue on one L2 atomic (measured 3.2 ms at 100 MP
  // against 0.16 ms for the copy itself).
  bool settled = false;
  for(size_t it = 0; it < iters; it++, k += stride)
  {
    const bool live = k < nvec;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if(live) r = reinterpret_cast<const float4 *>(in)[k];
    const bool o0 = live && r.x > threshold, o1 = live && r.y > threshold;
    const bool o2 = live && r.z > threshold, o3 = live && r.w > threshold;
    if(!settled && __ballot(o0 | o1 | o2 | o3) != 0ull)
    {
      settled = __hip_atomic_load(&journal->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= HL_MIN_CLIPPED;
      if(!settled)
      {
        hl_note(journal, o0, 4 * k + 0, r.x);
        hl_note(journal, o1, 4 * k + 1, r.y);
        hl_note(journal, o2, 4 * k + 2, r.z);
        hl_note(journal, o3, 4 * k + 3, r.w);
      }
    }
    if(live)
    {
      float4 o; // MIN(clip, in) == ((clip) < (in) ? (clip) : (in)), highlights/clip.c:73
      o.x = clip < r.x ? clip : r.x;
      o.y = clip < r.y ? clip : r.y;
      o.z = clip < r.z ? clip : r.z;
      o.w = clip < r.w ? clip : r.w;
      nt_store(reinterpret_cast<float4 *>(out) + k, o);
    }
  }
  // tail (n % 4 elements), handled by the first wave of block 0
  if(blockIdx.x == 0 && threadIdx.x < 64)
  {
    const size_t e = (nvec << 2) + threadIdx.x;
    const bool live = e < n;
    const float v = live ? in[e] : 0.f;
    hl_note(journal, live && v > threshold, e, v);
    if(live) out[e] = clip < v ? clip : v;
  }
}

// the same for buffers whose base is not 16-byte aligned (a row band of a frame whose width is not a multiple
// of 4 starts its own rows behind 9 halo rows): one photosite per lane
__global__ __launch_bounds__(256) void highlights_clip_1f_scalar(const float *__restrict__ in, float *__restrict__ out,
                                                                  const size_t n, const float clip, const float threshold,
                                                                  hl_journal *journal)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t iters = (n + stride - 1) / stride; // whole-wave trip count so that __ballot() sees every lane
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool settled = false;
  for(size_t it = 0; it < iters; it++, k += stride)
  {
    const bool live = k < n;
    const float v = live ? in[k] : 0.f;
    const bool over = live && v > threshold;
    if(!settled && __ballot(over) != 0ull)
    {
      settled = __hip_atomic_load(&journal->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= HL_MIN_CLIPPED;
      if(!settled) hl_note(journal, over, k, v);
    }
    if(live) out[k] = clip < v ? clip : v;
  }
}

// elements: RGBA pixels (ch = 4) -- a pixel counts once if any of R,G,B is over its threshold;
// process_clip clamps all 4 channels (highlights/clip.c:78-83)
__global__ __launch_bounds__(256) void highlights_clip_4f(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                           const size_t npixels, const float clip, const float4 thr,
                                                           unsigned long long *count)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t iters = (npixels + stride - 1) / stride;
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long local = 0;
  for(size_t it = 0; it < iters; it++, k += stride)
  {
    if(k >= npixels) continue;
    const float4 r = in[k];
    local += ((r.x > thr.x) | (r.y > thr.y) | (r.z > thr.z)) ? 1 : 0;
    float4 o;
    o.x = clip < r.x ? clip : r.x;
    o.y = clip < r.y ? clip : r.y;
    o.z = clip < r.z ? clip : r.z;
    o.w = clip < r.w ? clip : r.w;
    out[k] = o;
  }
  for(int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
  if((threadIdx.x & 63) == 0 && local) atomicAdd(count, local);
}

__global__ void highlights_restore_1f(float *__restrict__ out, const hl_journal *journal)
{
  // count may be the sum over several bands: each band restores the slots it journalled itself
  if(journal->count >= HL_MIN_CLIPPED || threadIdx.x >= HL_MIN_CLIPPED) return;
  const unsigned long long slot = journal->index[threadIdx.x];
  if(slot) out[slot - 1] = journal->value[threadIdx.x];
}

// 4-channel bypass: fewer than 25 pixels over threshold -> output must equal the input.
// Rare (a frame with no highlights at all), so a plain second pass is acceptable.
__global__ __launch_bounds__(256) void highlights_copy_if_bypass(const float4 *__restrict__ in,
                                                                  float4 *__restrict__ out, const size_t npixels,
                                                                  const unsigned long long *count)
{
  if(*count >= HL_MIN_CLIPPED) return;
  for(size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < npixels; k += (size_t)gridDim.x * blockDim.x)
    out[k] = in[k];
}

// ---------------------------------------------------------------------------------------
// exposure
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void exposure_f4(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                    const size_t nvec, const float black, const float scale)
{
  for(size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < nvec; k += (size_t)gridDim.x * blockDim.x)
  {
    const float4 r = in[k];
    float4 o;
    o.x = (r.x - black) * scale;
    o.y = (r.y - black) * scale;
    o.z = (r.z - black) * scale;
    o.w = (r.w - black) * scale;
    nt_store(out + k, o);
  }
}

__global__ void exposure_tail(const float *__restrict__ in, float *__restrict__ out, const size_t begin,
                              const size_t end, const float black, const float scale)
{
  const size_t k = begin + threadIdx.x;
  if(k < end) out[k] = (in[k] - black) * scale;
}

// ---------------------------------------------------------------------------------------
// export float -> u16 / u8
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float glib_clamp(const float x, const float lo, const float hi)
{
  // CLAMP() of glib: (((x) > (high)) ? (high) : (((x) < (low)) ? (low) : (x)))
  return x > hi ? hi : (x < lo ? lo : x);
}

__device__ __forceinline__ float clampf(const float a, const float mn, const float mx)
{
  // CLAMPF(), src/math/math.h:91
  return a >= mn ? (a <= mx ? a : mx) : mn;
}

template <typename T>
__global__ __launch_bounds__(256) void pack_rows(const T *__restrict__ in, T *__restrict__ out, const size_t npixels,
                                                 const int layers)
{
  typedef T t4 __attribute__((ext_vector_type(4)));
  for(size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < npixels; k += (size_t)gridDim.x * blockDim.x)
  {
    const t4 v = reinterpret_cast<const t4 *>(in)[k];
    if(layers == 3)
    {
      out[3 * k] = v.x;
      out[3 * k + 1] = v.y;
      out[3 * k + 2] = v.z;
    }
    else
      out[k] = v.x;
  }
}

__global__ __launch_bounds__(256) void export_u16(const float4 *__restrict__ in, ushort4 *__restrict__ out,
                                                   const size_t npixels)
{
  for(size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < npixels; k += (size_t)gridDim.x * blockDim.x)
  {
    const float4 r = in[k];
    ushort4 o;
    // a NaN survives CLAMP(); x86 cvttss2si then yields 0x80000000 whose low 16 bits are 0,
    // which is also what the device conversion produces
    o.x = (unsigned short)(int)glib_clamp(roundf(r.x * 65535.f), 0.f, 65535.f);
    o.y = (unsigned short)(int)glib_clamp(roundf(r.y * 65535.f), 0.f, 65535.f);
    o.z = (unsigned short)(int)glib_clamp(roundf(r.z * 65535.f), 0.f, 65535.f);
    o.w = (unsigned short)(int)glib_clamp(roundf(r.w * 65535.f), 0.f, 65535.f);
    out[k] = o;
  }
}

__global__ __launch_bounds__(256) void export_u8(const float4 *__restrict__ in, uchar4 *__restrict__ out,
                                                  const size_t npixels)
{
  for(size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < npixels; k += (size_t)gridDim.x * blockDim.x)
  {
    const float4 r = in[k];
    uchar4 o;
    o.x = (unsigned char)(int)clampf(roundf(r.x * 255.f), 0.f, 255.f);
    o.y = (unsigned char)(int)clampf(roundf(r.y * 255.f), 0.f, 255.f);
    o.z = (unsigned char)(int)clampf(roundf(r.z * 255.f), 0.f, 255.f);
    o.w = (unsigned char)(int)clampf(roundf(r.w * 255.f), 0.f, 255.f);
    out[k] = o;
  }
}

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

} // namespace

namespace ansel
{
int highlights_resolve_launch(int devid, float *out, const hl_journal *journal)
{
  highlights_restore_1f<<<1, 64, 0, stream_of(devid)>>>(out, journal);
  return check_launch("highlights_restore_1f");
}
} // namespace ansel

extern "C" {

int dt_hip_iop_rawprepare_process(int devid, const dt_hip_piece_t *piece, const dt_hip_rawprepare_data_t *d,
                                  dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  if(!(piece->filters && piece->channels == 1))
  {
    set_last_error("rawprepare: only raw mosaic input is implemented on device");
    return DT_HIP_INVALID_ARG;
  }
  rawprepare_args a;
  a.width = piece->roi_out.width;
  a.height = piece->roi_out.height;
  a.in_width = piece->roi_in.width;
  // compute_proper_crop(), rawprepare.c:206-210
  a.csx = (int)roundf((float)((double)d->x * piece->roi_in.scale));
  a.csy = (int)roundf((float)((double)d->y * piece->roi_in.scale));
  a.cfa_x = piece->roi_out.x + d->x;
  a.cfa_y = piece->roi_out.y + d->y;
  for(int k = 0; k < 4; k++)
  {
    a.sub[k] = d->sub[k];
    a.inv_div[k] = 1.0f / d->div[k];
  }
  if(a.width <= 0 || a.height <= 0) return DT_HIP_SUCCESS;
  const bool u16 = piece->datatype == DT_HIP_TYPE_UINT16;
  const size_t esz = u16 ? 2 : 4;
  const bool vec = (a.width % 4 == 0) && (a.in_width % 4 == 0) && (a.csx % 4 == 0)
                   && (((uintptr_t)dev_in) % (4 * esz) == 0) && aligned16(dev_out);
  const size_t work = (size_t)((a.width + 3) / 4) * a.height;
  const unsigned grid = stream_grid(work, 256);
  launch_scope ls(devid, "rawprepare_1f");
  hipStream_t s = stream_of(devid);
  if(u16)
  {
    if(vec)
      rawprepare_1f<uint16_t, true><<<grid, 256, 0, s>>>((const uint16_t *)dev_in, (float *)dev_out, a);
    else
      rawprepare_1f<uint16_t, false><<<grid, 256, 0, s>>>((const uint16_t *)dev_in, (float *)dev_out, a);
  }
  else
  {
    if(vec)
      rawprepare_1f<float, true><<<grid, 256, 0, s>>>((const float *)dev_in, (float *)dev_out, a);
    else
      rawprepare_1f<float, false><<<grid, 256, 0, s>>>((const float *)dev_in, (float *)dev_out, a);
  }
  return check_launch("rawprepare_1f");
}

int dt_hip_iop_temperature_process(int devid, const dt_hip_piece_t *piece, const dt_hip_temperature_data_t *d,
                                   dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  const int width = piece->roi_out.width, height = piece->roi_out.height;
  if(width <= 0 || height <= 0) return DT_HIP_SUCCESS;
  const float4 cf = make_float4(d->coeffs[0], d->coeffs[1], d->coeffs[2], d->coeffs[3]);
  hipStream_t s = stream_of(devid);
  if(piece->filters == 9u)
  {
    set_last_error("temperature: X-Trans is out of scope");
    return DT_HIP_INVALID_ARG;
  }
  if(piece->filters)
  {
    const bool vec = (width % 4 == 0) && aligned16(dev_in) && aligned16(dev_out);
    const unsigned grid = stream_grid((size_t)((width + 3) / 4) * height, 256);
    launch_scope ls(devid, "temperature_1f");
    if(vec)
      temperature_1f<true><<<grid, 256, 0, s>>>((const float *)dev_in, (float *)dev_out, width, height,
                                                   piece->roi_out.x, piece->roi_out.y, piece->filters, cf);
    else
      temperature_1f<false><<<grid, 256, 0, s>>>((const float *)dev_in, (float *)dev_out, width, height,
                                                    piece->roi_out.x, piece->roi_out.y, piece->filters, cf);
    return check_launch("temperature_1f");
  }
  if(piece->channels != 4) return DT_HIP_INVALID_ARG;
  const size_t np = (size_t)width * height;
  launch_scope ls(devid, "temperature_4f");
  temperature_4f<<<stream_grid(np, 256), 256, 0, s>>>((const float4 *)dev_in, (float4 *)dev_out, np, cf);
  return check_launch("temperature_4f");
}

static int highlights_launch(int devid, const dt_hip_piece_t *piece, const dt_hip_highlights_data_t *d,
                             dt_hip_mem_t dev_in, dt_hip_mem_t dev_out, dt_hip_mem_t deferred);

int dt_hip_iop_highlights_process(int devid, const dt_hip_piece_t *piece, const dt_hip_highlights_data_t *d,
                                  dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  return highlights_launch(devid, piece, d, dev_in, dev_out, nullptr);
}

// Band mode (multi-GPU row bands): the bypass of highlights.c:728-733 depends on the number of
// clipped photosites of the WHOLE frame.  _deferred clips this band and journals into the caller's
// buffer (DT_HIP_HIGHLIGHTS_JOURNAL_BYTES); the caller sums the leading uint64 over all bands
// (all-reduce) and calls _resolve, which restores this band's journalled photosites if the sum < 25.
int dt_hip_iop_highlights_process_deferred(int devid, const dt_hip_piece_t *piece, const dt_hip_highlights_data_t *d,
                                           dt_hip_mem_t dev_in, dt_hip_mem_t dev_out, dt_hip_mem_t journal)
{
  if(!journal || !piece || !piece->filters)
  {
    set_last_error("highlights (deferred): needs a journal buffer and a mosaic input");
    return DT_HIP_INVALID_ARG;
  }
  return highlights_launch(devid, piece, d, dev_in, dev_out, journal);
}

static_assert(sizeof(hl_journal) <= DT_HIP_HIGHLIGHTS_JOURNAL_BYTES, "journal does not fit the ABI constant");

int dt_hip_iop_highlights_resolve(int devid, dt_hip_mem_t dev_out, dt_hip_mem_t journal)
{
  if(!valid_device(devid) || !dev_out || !journal) return DT_HIP_INVALID_ARG;
  return highlights_resolve_launch(devid, (float *)dev_out, (const hl_journal *)journal);
}

static int highlights_launch(int devid, const dt_hip_piece_t *piece, const dt_hip_highlights_data_t *d,
                             dt_hip_mem_t dev_in, dt_hip_mem_t dev_out, dt_hip_mem_t deferred)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  if(d->mode != DT_HIP_HIGHLIGHTS_CLIP)
  {
    set_last_error("highlights: only the clip mode is implemented on device");
    return DT_HIP_INVALID_ARG;
  }
  const int width = piece->roi_out.width, height = piece->roi_out.height;
  if(width <= 0 || height <= 0) return DT_HIP_SUCCESS;
  // highlights.c:717-720
  float pmax[4];
  for(int c = 0; c < 4; c++) pmax[c] = (piece->processed_maximum[c] > 0.f) ? piece->processed_maximum[c] : 1.0f;
  const float clip = d->clip * fminf(pmax[0], fminf(pmax[1], pmax[2]));
  // _hl_count_thresholds(): clip mode tests the scalar clip on every channel (highlights.c:232-255)
  const float4 thr = make_float4(clip, clip, clip, clip);
  hipStream_t s = stream_of(devid);
  hl_journal *journal = deferred ? (hl_journal *)deferred : (hl_journal *)dt_hip_alloc_device_buffer(devid, sizeof(hl_journal));
  if(!journal) return DT_HIP_SYSMEM_ALLOCATION;
  int err = DT_HIP_SUCCESS;
  if(hipMemsetAsync(journal, 0, sizeof(hl_journal), s) != hipSuccess) err = DT_HIP_DEFAULT_ERROR;
  const size_t np = (size_t)width * height;
  if(err == DT_HIP_SUCCESS && piece->filters)
  {
    const float raw_threshold = fminf(fminf(thr.x, thr.y), thr.z);
    {
      launch_scope ls(devid, "highlights_clip_1f");
      if(aligned16(dev_in) && aligned16(dev_out))
        highlights_clip_1f<<<stream_grid(np / 4 + 1, 256), 256, 0, s>>>((const float *)dev_in, (float *)dev_out, np,
                                                                          clip, raw_threshold, journal);
      else
        highlights_clip_1f_scalar<<<stream_grid(np, 256), 256, 0, s>>>((const float *)dev_in, (float *)dev_out, np,
                                                                         clip, raw_threshold, journal);
    }
    if(!deferred) highlights_restore_1f<<<1, 64, 0, s>>>((float *)dev_out, journal);
    err = check_launch("highlights_clip_1f");
  }
  else if(err == DT_HIP_SUCCESS)
  {
    if(piece->channels != 4) err = DT_HIP_INVALID_ARG;
    else
    {
      {
        launch_scope ls(devid, "highlights_clip_4f");
        highlights_clip_4f<<<stream_grid(np, 256), 256, 0, s>>>((const float4 *)dev_in, (float4 *)dev_out, np, clip,
                                                                  thr, &journal->count);
      }
      highlights_copy_if_bypass<<<stream_grid(np, 256), 256, 0, s>>>((const float4 *)dev_in, (float4 *)dev_out, np,
                                                                       &journal->count);
      err = check_launch("highlights_clip_4f");
    }
  }
  if(!deferred) dt_hip_release_mem_object(journal); // stream-ordered: reused only by later work on this stream
  return err;
}

int dt_hip_iop_exposure_process(int devid, const dt_hip_piece_t *piece, const dt_hip_exposure_data_t *d,
                                dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  const size_t n = (size_t)piece->roi_out.width * piece->roi_out.height * piece->channels;
  if(n == 0) return DT_HIP_SUCCESS;
  if(!aligned16(dev_in) || !aligned16(dev_out)) return DT_HIP_INVALID_ARG;
  hipStream_t s = stream_of(devid);
  const size_t nvec = n / 4;
  launch_scope ls(devid, "exposure");
  if(nvec)
    exposure_f4<<<stream_grid(nvec, 256), 256, 0, s>>>((const float4 *)dev_in, (float4 *)dev_out, nvec, d->black,
                                                         d->scale);
  if(n % 4) exposure_tail<<<1, 64, 0, s>>>((const float *)dev_in, (float *)dev_out, nvec * 4, n, d->black, d->scale);
  return check_launch("exposure");
}

// the row loop of the format writers: `layers` of the 4 samples of every pixel, packed
// (src/imageio/format/tiff.c:293-360 for 32 / 16 / 8 bits per sample; png.c and jpeg.c hand RGB rows to their
// libraries the same way)
int dt_hip_export_pack_rows(int devid, int width, int height, int bpp, int layers, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  if(!valid_device(devid) || !dev_in || !dev_out || (layers != 3 && layers != 1)) return DT_HIP_INVALID_ARG;
  const size_t np = (size_t)width * height;
  if(width <= 0 || height <= 0) return DT_HIP_SUCCESS;
  launch_scope ls(devid, "export_rows");
  hipStream_t s = stream_of(devid);
  const unsigned grid = stream_grid(np, 256);
  if(bpp == 32)
    pack_rows<float><<<grid, 256, 0, s>>>((const float *)dev_in, (float *)dev_out, np, layers);
  else if(bpp == 16)
    pack_rows<unsigned short><<<grid, 256, 0, s>>>((const unsigned short *)dev_in, (unsigned short *)dev_out, np, layers);
  else if(bpp == 8)
    pack_rows<unsigned char><<<grid, 256, 0, s>>>((const unsigned char *)dev_in, (unsigned char *)dev_out, np, layers);
  else
    return DT_HIP_INVALID_ARG;
  return check_launch("export_rows");
}

int dt_hip_export_convert_u16(int devid, int width, int height, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  if(!valid_device(devid) || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  const size_t np = (size_t)width * height;
  if(np == 0) return DT_HIP_SUCCESS;
  launch_scope ls(devid, "export_u16");
  export_u16<<<stream_grid(np, 256), 256, 0, stream_of(devid)>>>((const float4 *)dev_in, (ushort4 *)dev_out, np);
  return check_launch("export_u16");
}

int dt_hip_export_convert_u8(int devid, int width, int height, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  if(!valid_device(devid) || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  const size_t np = (size_t)width * height;
  if(np == 0) return DT_HIP_SUCCESS;
  launch_scope ls(devid, "export_u8");
  export_u8<<<stream_grid(np, 256), 256, 0, stream_of(devid)>>>((const float4 *)dev_in, (uchar4 *)dev_out, np);
  return check_launch("export_u8");
}

} // extern "C"

// ---- raw unpack: the wire end in front of rawprepare (include/ansel_hip.h) --------------------------------------
namespace
{
template <int ORDER>
__global__ __launch_bounds__(256) void raw_unpack(const unsigned char *__restrict__ packed, unsigned short *__restrict__ out,
                                                  const int width, const int height, const size_t row_bytes, const int bits)
{
  const size_t n = (size_t)width * height;
  for(size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x)
  {
    const int y = (int)(k / width), x = (int)(k - (size_t)y * width);
    const unsigned char *const row = packed + (size_t)y * row_bytes;
    const unsigned bit0 = (unsigned)x * (unsigned)bits;
    const unsigned byte0 = bit0 >> 3, sh = bit0 & 7u;
    // the photosite lies in at most three bytes (16 bits at a shift of up to 7); bytes past the row's last are not read
    const unsigned last = (unsigned)((size_t)width * bits + 7) / 8 - 1;
    const unsigned b0 = row[byte0], b1 = byte0 + 1 <= last ? row[byte0 + 1] : 0u, b2 = byte0 + 2 <= last ? row[byte0 + 2] : 0u;
    unsigned v;
    if(ORDER == DT_HIP_RAW_PACK_MSB)
      v = ((b0 << 16 | b1 << 8 | b2) >> (24 - sh - bits)) & ((1u << bits) - 1u);
    else
      v = ((b0 | b1 << 8 | b2 << 16) >> sh) & ((1u << bits) - 1u);
    out[k] = (unsigned short)v;
  }
}
} // namespace

extern "C" int dt_hip_raw_unpack(int devid, dt_hip_mem_t dev_packed, int width, int height, size_t row_bytes, int bits, int order,
                                 dt_hip_mem_t dev_out_u16)
{
  if(!valid_device(devid) || !dev_packed || !dev_out_u16 || width < 0 || height < 0) return DT_HIP_INVALID_ARG;
  if((bits != 8 && bits != 10 && bits != 12 && bits != 14 && bits != 16) || (order != DT_HIP_RAW_PACK_MSB && order != DT_HIP_RAW_PACK_LSB))
  {
    set_last_error("raw_unpack: %d bits in order %d is not a layout this unpacks (8, 10, 12, 14, 16 bits; MSB or LSB first)", bits, order);
    return DT_HIP_INVALID_ARG;
  }
  if(row_bytes < ((size_t)width * bits + 7) / 8)
  {
    set_last_error("raw_unpack: rows of %zu bytes cannot hold %d photosites of %d bits", row_bytes, width, bits);
    return DT_HIP_INVALID_ARG;
  }
  const size_t n = (size_t)width * height;
  if(n == 0) return DT_HIP_SUCCESS;
  hipStream_t s = stream_of(devid);
  launch_scope ls(devid, "raw_unpack");
  if(order == DT_HIP_RAW_PACK_MSB)
    raw_unpack<DT_HIP_RAW_PACK_MSB><<<stream_grid(n, 256), 256, 0, s>>>((const unsigned char *)dev_packed, (unsigned short *)dev_out_u16, width,
                                                                         height, row_bytes, bits);
  else
    raw_unpack<DT_HIP_RAW_PACK_LSB><<<stream_grid(n, 256), 256, 0, s>>>((const unsigned char *)dev_packed, (unsigned short *)dev_out_u16, width,
                                                                         height, row_bytes, bits);
  return check_launch("raw_unpack");
}
