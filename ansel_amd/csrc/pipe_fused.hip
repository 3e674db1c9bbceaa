// pipe_fused.hip -- fused stage groups of the export pipe.
//
// Why: every module after demosaic on this path is pointwise on a float4 plane (exposure, colorin,
// color calibration, filmic, colorout, the final float -> u16).  Run as separate process_cl() calls
// they move 5 x 32 + 24 = 184 B/px through HBM; fused, the pixel stays in registers from the
// demosaic output to the exported u16: 16 B in + 8 B out.  Same for the three CFA stages before
// demosaic (rawprepare, white balance, highlight clip): 6 + 8 + 8 = 22 B/px become 2 + 4.
// "Fuse elementwise work into the producing kernel, keep tensors resident": the arithmetic per
// pixel is literally the same device functions the standalone kernels call (px_*.h), in the same
// order, so results are bit-identical to the module-by-module chain (tests/test_gpu_pipe.py).
//
// The reference has no equivalent: its pixelpipe materialises every module output as a cacheline
// (src/develop/pixelpipe_hb.c:985, :1043-1047) because the GUI re-uses them; an export does not.
#include "pipe_fused.h"
#include "rgb_chain_kernel.h"

using namespace ansel;

namespace
{

// ---------------------------------------------------------------------------------------------
// CFA chain: rawprepare [-> temperature] [-> highlights clip]
// ---------------------------------------------------------------------------------------------
struct raw_args
{
  int width, height, in_width;
  int csx, csy;
  int cfa_x, cfa_y;     // rawprepare's CFA phase
  float sub[4], inv_div[4];
  int has_wb;
  int wb_x, wb_y;       // temperature's roi_out origin
  uint32_t filters;
  float coeffs[4];
  int has_clip;
  float clip, threshold;
};

__device__ __forceinline__ int fc_(const int row, const int col, const uint32_t filters)
{
  return filters >> ((((row << 1) & 14) + (col & 1)) << 1) & 3;
}

__device__ __forceinline__ void hl_note_(hl_journal *j, const bool over, const size_t index, const float value)
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

// one thread = 4 consecutive photosites of one row (width % 4 == 0, rows 8-byte aligned)
template <typename in_t>
__global__ __launch_bounds__(256) void raw_chain(const in_t *__restrict__ in, float *__restrict__ out, const raw_args a,
                                                  hl_journal *journal)
{
  const int quads = a.width >> 2;
  const size_t total = (size_t)quads * a.height;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t iters = (total + stride - 1) / stride;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool settled = !a.has_clip;
  for(size_t it = 0; it < iters; it++, t += stride)
  {
    const bool live = t < total;
    const size_t tt = live ? t : 0;
    const int j = (int)(tt / quads);
    const int x0 = (int)(tt - (size_t)j * quads) << 2;
    const int row_phase = ((j + a.cfa_y) & 1) << 1;
    const int x_phase = a.cfa_x & 1;
    const int id0 = row_phase + x_phase, id1 = row_phase + (x_phase ^ 1);
    const size_t pin = (size_t)a.in_width * (j + a.csy) + a.csx + x0;
    const size_t pout = (size_t)j * a.width + x0;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if(live)
    {
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
    }
    // rawprepare, rawprepare.c:507-511
    v0 = (v0 - a.sub[id0]) * a.inv_div[id0];
    v1 = (v1 - a.sub[id1]) * a.inv_div[id1];
    v2 = (v2 - a.sub[id0]) * a.inv_div[id0];
    v3 = (v3 - a.sub[id1]) * a.inv_div[id1];
    if(a.has_wb)
    {
      // temperature, temperature.c:543-560
      const float c0 = a.coeffs[fc_(j + a.wb_y, x0 + a.wb_x, a.filters)];
      const float c1 = a.coeffs[fc_(j + a.wb_y, x0 + a.wb_x + 1, a.filters)];
      v0 = v0 * c0; v1 = v1 * c1; v2 = v2 * c0; v3 = v3 * c1;
    }
    if(a.has_clip)
    {
      const bool o0 = live && v0 > a.threshold, o1 = live && v1 > a.threshold;
      const bool o2 = live && v2 > a.threshold, o3 = live && v3 > a.threshold;
      if(!settled && __ballot(o0 | o1 | o2 | o3) != 0ull)
      {
        settled = __hip_atomic_load(&journal->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= HL_MIN_CLIPPED;
        if(!settled)
        {
          hl_note_(journal, o0, pout + 0, v0);
          hl_note_(journal, o1, pout + 1, v1);
          hl_note_(journal, o2, pout + 2, v2);
          hl_note_(journal, o3, pout + 3, v3);
        }
      }
      v0 = a.clip < v0 ? a.clip : v0;
      v1 = a.clip < v1 ? a.clip : v1;
      v2 = a.clip < v2 ? a.clip : v2;
      v3 = a.clip < v3 ? a.clip : v3;
    }
    if(live) nt_store(reinterpret_cast<float4 *>(out + pout), make_float4(v0, v1, v2, v3));
  }
}

} // namespace

namespace ansel
{

bool raw_group_supported(const raw_group_t &g)
{
  const dt_hip_piece_t &p = g.rawprepare_piece;
  if(!(p.filters && p.filters != 9u && p.channels == 1)) return false;
  const int w = p.roi_out.width, inw = p.roi_in.width;
  const int csx = (int)roundf((float)((double)g.rawprepare.x * p.roi_in.scale));
  if(w <= 0 || (w % 4) || (inw % 4) || (csx % 4)) return false;
  if(g.has_temperature)
  {
    const dt_hip_piece_t &t = g.temperature_piece;
    if(!t.filters || t.filters == 9u || t.roi_out.width != w || t.roi_out.height != p.roi_out.height) return false;
  }
  if(g.has_highlights)
  {
    const dt_hip_piece_t &h = g.highlights_piece;
    if(!h.filters || h.filters == 9u || g.highlights.mode != DT_HIP_HIGHLIGHTS_CLIP) return false;
    if(h.roi_out.width != w || h.roi_out.height != p.roi_out.height) return false;
  }
  return true;
}

// deferred != nullptr (band mode): the caller's journal is used and the <25 bypass is left to
// highlights_resolve_launch() once the count has been summed over all bands
int raw_group_launch(int devid, const raw_group_t &g, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out, dt_hip_mem_t deferred)
{
  if(!valid_device(devid) || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  if(!raw_group_supported(g)) return DT_HIP_INVALID_ARG;
  if(((uintptr_t)dev_in & 15u) || ((uintptr_t)dev_out & 15u)) return DT_HIP_INVALID_ARG;
  const dt_hip_piece_t &p = g.rawprepare_piece;
  raw_args a;
  memset(&a, 0, sizeof(a));
  a.width = p.roi_out.width;
  a.height = p.roi_out.height;
  a.in_width = p.roi_in.width;
  a.csx = (int)roundf((float)((double)g.rawprepare.x * p.roi_in.scale));
  a.csy = (int)roundf((float)((double)g.rawprepare.y * p.roi_in.scale));
  a.cfa_x = p.roi_out.x + g.rawprepare.x;
  a.cfa_y = p.roi_out.y + g.rawprepare.y;
  for(int k = 0; k < 4; k++)
  {
    a.sub[k] = g.rawprepare.sub[k];
    a.inv_div[k] = 1.0f / g.rawprepare.div[k];
  }
  a.has_wb = g.has_temperature;
  if(g.has_temperature)
  {
    a.wb_x = g.temperature_piece.roi_out.x;
    a.wb_y = g.temperature_piece.roi_out.y;
    a.filters = g.temperature_piece.filters;
    for(int k = 0; k < 4; k++) a.coeffs[k] = g.temperature.coeffs[k];
  }
  a.has_clip = g.has_highlights;
  hipStream_t s = stream_of(devid);
  hl_journal *journal = nullptr;
  if(g.has_highlights)
  {
    float pmax[4];
    for(int c = 0; c < 4; c++)
      pmax[c] = (g.highlights_piece.processed_maximum[c] > 0.f) ? g.highlights_piece.processed_maximum[c] : 1.0f;
    a.clip = g.highlights.clip * fminf(pmax[0], fminf(pmax[1], pmax[2]));
    a.threshold = a.clip;
    journal = deferred ? (hl_journal *)deferred : (hl_journal *)dt_hip_alloc_device_buffer(devid, sizeof(hl_journal));
    if(!journal) return DT_HIP_SYSMEM_ALLOCATION;
    if(hipMemsetAsync(journal, 0, sizeof(hl_journal), s) != hipSuccess)
    {
      if(!deferred) dt_hip_release_mem_object(journal);
      return DT_HIP_DEFAULT_ERROR;
    }
  }
  const size_t work = (size_t)(a.width / 4) * a.height;
  {
    launch_scope ls(devid, "raw_chain");
    if(p.datatype == DT_HIP_TYPE_UINT16)
      raw_chain<uint16_t><<<stream_grid(work, 256), 256, 0, s>>>((const uint16_t *)dev_in, (float *)dev_out, a, journal);
    else
      raw_chain<float><<<stream_grid(work, 256), 256, 0, s>>>((const float *)dev_in, (float *)dev_out, a, journal);
  }
  if(journal && !deferred)
  {
    highlights_resolve_launch(devid, (float *)dev_out, journal);
    dt_hip_release_mem_object(journal);
  }
  return check_launch("raw_chain");
}

int rgb_group_fill_args(const rgb_group_t &g, chain_args &a, int &cm_kind, int &fm)
{
  memset(&a, 0, sizeof(a));
  cm_kind = CM_NONE;
  fm = FM_NONE;
  for(int i = 0; i < g.n_ops && i < 8; i++)
  {
    switch(g.ops[i])
    {
      case RGB_OP_EXPOSURE:
        a.has_exposure = 1;
        a.exp_black = g.exposure.black;
        a.exp_scale = g.exposure.scale;
        break;
      case RGB_OP_COLORIN:
        a.has_colorin = 1;
        conversion_fill_args(&g.colorin, a.colorin);
        break;
      case RGB_OP_COLOROUT:
        a.has_colorout = 1;
        conversion_fill_args(&g.colorout, a.colorout);
        break;
      case RGB_OP_CHANNELMIXER:
        channelmixerrgb_fill_args(&g.channelmixer, a.cm);
        cm_kind = a.cm.kind;
        a.cm_clip = a.cm.clip;
        break;
      case RGB_OP_FILMIC:
      {
        const int err = filmicrgb_fill_args(&g.filmic, a.filmic);
        if(err != DT_HIP_SUCCESS) return err;
        fm = a.filmic.mode;
        a.filmic_export = a.filmic.use_export;
        break;
      }
      default: return DT_HIP_INVALID_ARG;
    }
  }
  a.to_u16 = g.to_u16;
  a.pre_lab = g.pre_lab;
  a.post_lab = g.post_lab;
  memcpy(a.lab_pre, g.lab_pre.matrix, sizeof(a.lab_pre));
  memcpy(a.lab_post, g.lab_post.matrix, sizeof(a.lab_post));
  return DT_HIP_SUCCESS;
}

int rgb_group_launch(int devid, const rgb_group_t &g, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out, const bilat_slice_args *pre_slice)
{
  if(!valid_device(devid) || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  const size_t np = (size_t)g.width * g.height;
  if(np == 0) return DT_HIP_SUCCESS;
  chain_args a;
  int cm_kind, fm;
  const int ferr = rgb_group_fill_args(g, a, cm_kind, fm);
  if(ferr != DT_HIP_SUCCESS) return ferr;
  if(pre_slice)
  {
    a.pre_bilat = 1;
    a.bilat = *pre_slice;
  }
  hipStream_t s = stream_of(devid);
  const unsigned grid = pixel_grid(np); // one pixel per thread, see rgb_chain_kernel.h
  const float4 *in = (const float4 *)dev_in;
  launch_scope ls(devid, g.to_u16 == 2 ? "rgb_chain_rows16" : (g.to_u16 ? "rgb_chain_u16" : "rgb_chain"));
  int err;
  switch(fm)
  {
    case FM_NONE: err = rgb_chain_launch_none(cm_kind, grid, s, in, dev_out, np, a); break;
    case MODE_AGX: err = rgb_chain_launch_agx(cm_kind, grid, s, in, dev_out, np, a); break;
    case MODE_V5: err = rgb_chain_launch_v5(cm_kind, grid, s, in, dev_out, np, a); break;
    case MODE_SPLIT_V4: err = rgb_chain_launch_split_v4(cm_kind, grid, s, in, dev_out, np, a); break;
    case MODE_CHROMA_V4: err = rgb_chain_launch_chroma_v4(cm_kind, grid, s, in, dev_out, np, a); break;
    default: // the 2019-2020 colour sciences run in their own launch (the planner keeps them out of a run)
      set_last_error("rgb_chain: filmic mode %d has no fused kernel", fm);
      return DT_HIP_INVALID_ARG;
  }
  if(err != DT_HIP_SUCCESS) return err;
  return check_launch("rgb_chain");
}

} // namespace ansel
