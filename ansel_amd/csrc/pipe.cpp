// pipe.cpp -- export-pipe executor on the dt_hip_* runtime: the device-resident part of
// dt_dev_pixelpipe_process_rec() (src/develop/pixelpipe_hb.c:881-1282) + pixelpipe_process_on_GPU()
// (src/develop/pixelpipe_gpu.c:191-744) for an export, where no module output is re-used.
//
// The reference walks the node list recursively from the last node, acquires a cacheline per module
// output and calls process_cl(); here the list is walked forward, outputs come from the runtime's
// pool and go back to it as soon as the consumer is enqueued (stream-ordered, so safe), and runs
// of pointwise modules are planned into fused launches (pipe_fused.hip).
#include "hip_common.h"
#include "pipe_fused.h"
#include "amaze_stream_body.h" // amz::stream_tile_ok(): which AMaZE tiles the on-chip kernel takes (band planning)

#include <algorithm>
#include <condition_variable>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace ansel;

namespace ansel
{
int dt_hip_iop_demosaic_process_band(int devid, const dt_hip_piece_t *piece, const dt_hip_demosaic_data_t *d,
                                     const rcd_band_t *band, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);
}

namespace
{

enum op_t
{
  OP_RAWPREPARE,
  OP_TEMPERATURE,
  OP_HIGHLIGHTS,
  OP_DEMOSAIC,
  OP_DENOISEPROFILE,
  OP_EXPOSURE,
  OP_COLORIN,
  OP_CHANNELMIXERRGB,
  OP_FILMICRGB,
  OP_COLOROUT,
  OP_DIFFUSE,
  OP_RGB_TO_LAB,
  OP_NLMEANS,
  OP_BILAT,
  OP_LAB_TO_RGB,
  OP_FINALSCALE,
  OP_INITIALSCALE,
  OP_EXPORT_U16,
  OP_BLEND,
  OP_EXPORT_ROWS,
  OP_EXPORT_U8,
  OP_DETAILMASK,
  OP_UNKNOWN
};

struct op_info_t
{
  const char *name;
  size_t data_size;
  int bpp_out; // bytes per pixel of the module output
};

const op_info_t k_ops[] = {
  { "rawprepare", sizeof(dt_hip_rawprepare_data_t), 4 },
  { "temperature", sizeof(dt_hip_temperature_data_t), 0 /* = input */ },
  { "highlights", sizeof(dt_hip_highlights_data_t), 0 },
  { "demosaic", sizeof(dt_hip_demosaic_data_t), 16 },
  { "denoiseprofile", sizeof(dt_hip_denoiseprofile_data_t), 16 },
  { "exposure", sizeof(dt_hip_exposure_data_t), 0 },
  { "colorin", sizeof(dt_hip_conversion_t), 16 },
  { "channelmixerrgb", sizeof(dt_hip_channelmixerrgb_data_t), 16 },
  { "filmicrgb", sizeof(dt_hip_filmicrgb_data_t), 16 },
  { "colorout", sizeof(dt_hip_conversion_t), 16 },
  { "diffuse", sizeof(dt_hip_diffuse_data_t), 16 },
  { "rgb_to_lab", sizeof(dt_hip_lab_data_t), 16 },
  { "nlmeans", sizeof(dt_hip_nlmeans_data_t), 16 },
  { "bilat", sizeof(dt_hip_bilat_data_t), 16 },
  { "lab_to_rgb", sizeof(dt_hip_lab_data_t), 16 },
  { "finalscale", sizeof(dt_hip_finalscale_data_t), 16 },
  { "initialscale", sizeof(dt_hip_finalscale_data_t), 16 },
  { "export_u16", 0, 8 },
  { "blend", sizeof(dt_hip_blend_data_t), 0 },
  { "export_rows", sizeof(dt_hip_export_rows_t), 0 },
  { "export_u8", 0, 4 },
  { "detailmask", sizeof(dt_hip_detailmask_data_t), 16 },
};

struct node_t
{
  op_t op;
  dt_hip_piece_t piece;
  std::vector<unsigned char> data;
  template <typename T> const T *as() const { return reinterpret_cast<const T *>(data.data()); }
};

struct group_t
{
  enum kind_t { SINGLE, RAW, RGB } kind;
  int first, count; // node range
  raw_group_t raw;
  rgb_group_t rgb;
};

size_t out_bytes(const node_t &n)
{
  const size_t px = (size_t)n.piece.roi_out.width * n.piece.roi_out.height;
  switch(n.op)
  {
    case OP_RAWPREPARE: return px * 4;
    case OP_TEMPERATURE:
    case OP_HIGHLIGHTS:
    case OP_EXPOSURE: return px * 4 * n.piece.channels;
    case OP_EXPORT_U16: return px * 8;
    case OP_EXPORT_U8: return px * 4;
    case OP_EXPORT_ROWS:
      return px * (size_t)n.as<dt_hip_export_rows_t>()->layers * (size_t)(n.as<dt_hip_export_rows_t>()->bpp / 8);
    default: return px * 16;
  }
}

int run_single(int devid, const node_t &n, dt_hip_mem_t in, dt_hip_mem_t out)
{
  switch(n.op)
  {
    case OP_RAWPREPARE: return dt_hip_iop_rawprepare_process(devid, &n.piece, n.as<dt_hip_rawprepare_data_t>(), in, out);
    case OP_TEMPERATURE: return dt_hip_iop_temperature_process(devid, &n.piece, n.as<dt_hip_temperature_data_t>(), in, out);
    case OP_HIGHLIGHTS: return dt_hip_iop_highlights_process(devid, &n.piece, n.as<dt_hip_highlights_data_t>(), in, out);
    case OP_DEMOSAIC: return dt_hip_iop_demosaic_process(devid, &n.piece, n.as<dt_hip_demosaic_data_t>(), in, out);
    case OP_DENOISEPROFILE: return dt_hip_iop_denoiseprofile_process(devid, &n.piece, n.as<dt_hip_denoiseprofile_data_t>(), in, out);
    case OP_DIFFUSE: return dt_hip_iop_diffuse_process(devid, &n.piece, n.as<dt_hip_diffuse_data_t>(), in, out);
    case OP_RGB_TO_LAB: return dt_hip_transform_rgb_to_lab(devid, &n.piece, n.as<dt_hip_lab_data_t>(), in, out);
    case OP_NLMEANS: return dt_hip_iop_nlmeans_process(devid, &n.piece, n.as<dt_hip_nlmeans_data_t>(), in, out);
    case OP_BILAT: return dt_hip_iop_bilat_process(devid, &n.piece, n.as<dt_hip_bilat_data_t>(), in, out);
    case OP_LAB_TO_RGB: return dt_hip_transform_lab_to_rgb(devid, &n.piece, n.as<dt_hip_lab_data_t>(), in, out);
    case OP_EXPOSURE: return dt_hip_iop_exposure_process(devid, &n.piece, n.as<dt_hip_exposure_data_t>(), in, out);
    case OP_COLORIN: return dt_hip_iop_colorin_process(devid, &n.piece, n.as<dt_hip_conversion_t>(), in, out);
    case OP_CHANNELMIXERRGB: return dt_hip_iop_channelmixerrgb_process(devid, &n.piece, n.as<dt_hip_channelmixerrgb_data_t>(), in, out);
    case OP_FILMICRGB: return dt_hip_iop_filmicrgb_process(devid, &n.piece, n.as<dt_hip_filmicrgb_data_t>(), in, out);
    case OP_COLOROUT: return dt_hip_iop_colorout_process(devid, &n.piece, n.as<dt_hip_conversion_t>(), in, out);
    case OP_FINALSCALE: return dt_hip_iop_finalscale_process(devid, &n.piece, n.as<dt_hip_finalscale_data_t>(), in, out);
    case OP_INITIALSCALE: return dt_hip_iop_initialscale_process(devid, &n.piece, n.as<dt_hip_finalscale_data_t>(), in, out);
    case OP_DETAILMASK: return dt_hip_iop_detailmask_process(devid, &n.piece, n.as<dt_hip_detailmask_data_t>(), in, out);
    case OP_EXPORT_U16: return dt_hip_export_convert_u16(devid, n.piece.roi_out.width, n.piece.roi_out.height, in, out);
    case OP_EXPORT_U8: return dt_hip_export_convert_u8(devid, n.piece.roi_out.width, n.piece.roi_out.height, in, out);
    case OP_EXPORT_ROWS:
      return dt_hip_export_pack_rows(devid, n.piece.roi_out.width, n.piece.roi_out.height, n.as<dt_hip_export_rows_t>()->bpp,
                                     n.as<dt_hip_export_rows_t>()->layers, in, out);
    default: return DT_HIP_INVALID_ARG;
  }
}

} // namespace

struct dt_hip_pipe_t
{
  int devid;
  bool fusion;
  bool planned;
  std::vector<node_t> nodes;
  std::vector<group_t> groups;

  void plan()
  {
    groups.clear();
    const int n = (int)nodes.size();
    // a module followed by a "blend" node keeps both its input and its output as buffers: it is never fused
    auto blended = [&](const int k) { return k + 1 < n && nodes[k + 1].op == OP_BLEND; };
    int i = 0;
    while(i < n)
    {
      group_t g;
      g.kind = group_t::SINGLE;
      g.first = i;
      g.count = 1;
      if(fusion && nodes[i].op == OP_RAWPREPARE && !blended(i) && !blended(i + 1) && !blended(i + 2))
      {
        raw_group_t r;
        memset(&r, 0, sizeof(r));
        r.rawprepare_piece = nodes[i].piece;
        r.rawprepare = *nodes[i].as<dt_hip_rawprepare_data_t>();
        int j = i + 1;
        if(j < n && nodes[j].op == OP_TEMPERATURE)
        {
          r.has_temperature = true;
          r.temperature_piece = nodes[j].piece;
          r.temperature = *nodes[j].as<dt_hip_temperature_data_t>();
          j++;
        }
        if(j < n && nodes[j].op == OP_HIGHLIGHTS)
        {
          r.has_highlights = true;
          r.highlights_piece = nodes[j].piece;
          r.highlights = *nodes[j].as<dt_hip_highlights_data_t>();
          j++;
        }
        if(j - i > 1 && raw_group_supported(r))
        {
          g.kind = group_t::RAW;
          g.count = j - i;
          g.raw = r;
        }
      }
      else if(fusion && nodes[i].piece.channels == 4 && !blended(i)
              && ((nodes[i].op >= OP_EXPOSURE && nodes[i].op <= OP_COLOROUT)
                  || (nodes[i].op == OP_LAB_TO_RGB && !nodes[i].as<dt_hip_lab_data_t>()->nonlinearlut && i + 1 < n
                      && nodes[i + 1].op >= OP_EXPOSURE && nodes[i + 1].op <= OP_COLOROUT)))
      {
        rgb_group_t r;
        memset(&r, 0, sizeof(r));
        r.width = nodes[i].piece.roi_out.width;
        r.height = nodes[i].piece.roi_out.height;
        // the fused kernel applies its stages in the reference's pipe order (exposure < colorin <
        // channelmixerrgb < filmicrgb < colorout, src/develop/iop_order.c); a run is fusable as long
        // as it walks that order
        int last_op = -1;
        int j = i;
        if(nodes[i].op == OP_LAB_TO_RGB)
        {
          // the Lab -> RGB glue behind a Lab module is the first stage of the run that follows it
          r.pre_lab = 1;
          r.lab_pre = *nodes[i].as<dt_hip_lab_data_t>();
          j++;
        }
        while(j < n && r.n_ops < 8)
        {
          const node_t &nd = nodes[j];
          if(nd.op < OP_EXPOSURE || nd.op > OP_COLOROUT || (int)nd.op <= last_op || blended(j)) break;
          if(nd.piece.roi_out.width != r.width || nd.piece.roi_out.height != r.height || nd.piece.channels != 4) break;
          if(nd.op == OP_FILMICRGB)
          {
            const int v = nd.as<dt_hip_filmicrgb_data_t>()->version;
            if(v < 3 || v > 9) break;
          }
          if(nd.op == OP_CHANNELMIXERRGB && nd.as<dt_hip_channelmixerrgb_data_t>()->adaptation > DT_HIP_ADAPTATION_RGB) break;
          last_op = (int)nd.op;
          switch(nd.op)
          {
            case OP_EXPOSURE: r.ops[r.n_ops++] = RGB_OP_EXPOSURE; r.exposure = *nd.as<dt_hip_exposure_data_t>(); break;
            case OP_COLORIN: r.ops[r.n_ops++] = RGB_OP_COLORIN; r.colorin = *nd.as<dt_hip_conversion_t>(); break;
            case OP_CHANNELMIXERRGB: r.ops[r.n_ops++] = RGB_OP_CHANNELMIXER; r.channelmixer = *nd.as<dt_hip_channelmixerrgb_data_t>(); break;
            case OP_FILMICRGB: r.ops[r.n_ops++] = RGB_OP_FILMIC; r.filmic = *nd.as<dt_hip_filmicrgb_data_t>(); break;
            default: r.ops[r.n_ops++] = RGB_OP_COLOROUT; r.colorout = *nd.as<dt_hip_conversion_t>(); break;
          }
          j++;
        }
        if(j < n && nodes[j].op == OP_EXPORT_U16 && nodes[j].piece.roi_out.width == r.width
           && nodes[j].piece.roi_out.height == r.height)
        {
          r.to_u16 = 1;
          j++;
          // ... and straight into the scanlines of the format writer
          if(j < n && nodes[j].op == OP_EXPORT_ROWS && nodes[j].as<dt_hip_export_rows_t>()->bpp == 16
             && nodes[j].as<dt_hip_export_rows_t>()->layers == 3 && nodes[j].piece.roi_out.width == r.width
             && nodes[j].piece.roi_out.height == r.height)
          {
            r.to_u16 = 2;
            j++;
          }
        }
        else if(r.n_ops > 0 && j < n && nodes[j].op == OP_RGB_TO_LAB && !blended(j) && nodes[j].piece.channels == 4
                && !nodes[j].as<dt_hip_lab_data_t>()->nonlinearlut
                && nodes[j].piece.roi_out.width == r.width && nodes[j].piece.roi_out.height == r.height)
        {
          // ... and the RGB -> Lab glue in front of a Lab module its last one
          r.post_lab = 1;
          r.lab_post = *nodes[j].as<dt_hip_lab_data_t>();
          j++;
        }
        if(j - i > 1 && r.n_ops > 0)
        {
          g.kind = group_t::RGB;
          g.count = j - i;
          g.rgb = r;
        }
      }
      groups.push_back(g);
      i += g.count;
    }
    planned = true;
  }
};

extern "C" {

dt_hip_pipe_t *dt_hip_pipe_new(int devid)
{
  if(!valid_device(devid)) return nullptr;
  dt_hip_pipe_t *p = new dt_hip_pipe_t;
  p->devid = devid;
  p->fusion = true;
  p->planned = false;
  return p;
}

void dt_hip_pipe_free(dt_hip_pipe_t *pipe) { delete pipe; }

int dt_hip_pipe_add_node(dt_hip_pipe_t *pipe, const char *op, const dt_hip_piece_t *piece, const void *data,
                         size_t data_size)
{
  if(!pipe || !op || !piece) return DT_HIP_INVALID_ARG;
  op_t o = OP_UNKNOWN;
  for(int k = 0; k < (int)OP_UNKNOWN; k++)
    if(!strcmp(op, k_ops[k].name)) o = (op_t)k;
  if(o == OP_UNKNOWN)
  {
    set_last_error("dt_hip_pipe_add_node: module '%s' has no device implementation", op);
    return DT_HIP_INVALID_ARG;
  }
  if(data_size != k_ops[o].data_size || (data_size && !data))
  {
    set_last_error("dt_hip_pipe_add_node: '%s' expects %zu bytes of data, got %zu", op, k_ops[o].data_size, data_size);
    return DT_HIP_INVALID_ARG;
  }
  node_t n;
  n.op = o;
  n.piece = *piece;
  if(data_size) n.data.assign((const unsigned char *)data, (const unsigned char *)data + data_size);
  pipe->nodes.push_back(n);
  pipe->planned = false;
  return DT_HIP_SUCCESS;
}

void dt_hip_pipe_set_fusion(dt_hip_pipe_t *pipe, int enabled)
{
  if(!pipe) return;
  pipe->fusion = enabled != 0;
  pipe->planned = false;
}

int dt_hip_pipe_num_groups(dt_hip_pipe_t *pipe)
{
  if(!pipe) return 0;
  if(!pipe->planned) pipe->plan();
  return (int)pipe->groups.size();
}

int dt_hip_pipe_process(dt_hip_pipe_t *pipe, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  if(!pipe || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  if(pipe->nodes.empty()) return DT_HIP_SUCCESS;
  if(!pipe->planned) pipe->plan();
  const int devid = pipe->devid;
  dt_hip_mem_t cur = dev_in;
  bool cur_owned = false;
  dt_hip_mem_t held = NULL; // the input of a module whose output is about to be blended
  bool held_owned = false;
  // the lightness cells of the current buffer's pixels, written by the non-local-means kernels for the bilateral grid of the local
  // contrast module right behind them (round 6: bilat_zcells' pass over the frame -- 24 B/px -- folded into their epilogue)
  dt_hip_mem_t cells = NULL;
  struct cells_guard // whatever path leaves the walk: the cells go back to the pool
  {
    dt_hip_mem_t &c;
    ~cells_guard()
    {
      if(c) dt_hip_release_mem_object(c);
    }
  } cells_owner{ cells };
  const size_t ng = pipe->groups.size();
  auto is_blend = [&](const size_t k) { return k < ng && pipe->nodes[pipe->groups[k].first].op == OP_BLEND; };
  for(size_t gi = 0; gi < ng; gi++)
  {
    const group_t &g = pipe->groups[gi];
    const node_t &last = pipe->nodes[g.first + g.count - 1];
    if(last.op == OP_BLEND)
    {
      // dt_develop_blend_process() after the module's process(), pixelpipe_cpu.c:137-228: in place in the output
      int err = DT_HIP_INVALID_ARG;
      if(held) err = dt_hip_develop_blend_process(devid, &last.piece, last.as<dt_hip_blend_data_t>(), held, cur);
      else set_last_error("pipe: a blend node needs the module it blends in front of it");
      if(held_owned) dt_hip_release_mem_object(held);
      held = NULL;
      if(err != DT_HIP_SUCCESS)
      {
        if(cur_owned) dt_hip_release_mem_object(cur);
        return err;
      }
      continue;
    }
    // denoise (profiled) followed by a pointwise run: the run becomes the tail of the module's last kernel
    // (denoiseprofile.hip dn_finish_chain) when there is such a kernel for the combination
    if(pipe->fusion && g.kind == group_t::SINGLE && pipe->nodes[g.first].op == OP_DENOISEPROFILE && gi + 1 < ng
       && pipe->groups[gi + 1].kind == group_t::RGB && !is_blend(gi + 1))
    {
      const group_t &gn = pipe->groups[gi + 1];
      const node_t &tail = pipe->nodes[gn.first + gn.count - 1];
      const bool final_pair = gi + 2 == ng || (is_blend(gi + 2) && gi + 3 == ng);
      dt_hip_mem_t fout = final_pair ? dev_out : dt_hip_alloc_device_buffer(devid, out_bytes(tail));
      if(fout)
      {
        const node_t &dn = pipe->nodes[g.first];
        const int ferr = denoiseprofile_process_chain(devid, &dn.piece, dn.as<dt_hip_denoiseprofile_data_t>(), cur, fout, &gn.rgb);
        if(ferr == DT_HIP_SUCCESS)
        {
          // a blend behind the run wants the run's input, which no longer exists as a buffer: such runs are not
          // fused (is_blend(gi + 2) with a non-final pair is excluded below)
          if(cur_owned) dt_hip_release_mem_object(cur);
          cur = fout;
          cur_owned = !final_pair;
          gi++;
          continue;
        }
        if(!final_pair) dt_hip_release_mem_object(fout);
        if(ferr != DT_HIP_INVALID_ARG)
        {
          if(cur_owned) dt_hip_release_mem_object(cur);
          return ferr;
        }
      }
    }
    // local contrast (bilateral grid) followed by a pointwise run: the module's slice -- pointwise, given the blurred grid -- becomes
    // the run's first stage (bilat.hip bilat_process_chain()): the module's output plane is never written
    if(pipe->fusion && g.kind == group_t::SINGLE && pipe->nodes[g.first].op == OP_BILAT && gi + 1 < ng
       && pipe->groups[gi + 1].kind == group_t::RGB && !is_blend(gi + 1))
    {
      const group_t &gn = pipe->groups[gi + 1];
      const node_t &tail = pipe->nodes[gn.first + gn.count - 1];
      const bool final_pair = gi + 2 == ng || (is_blend(gi + 2) && gi + 3 == ng);
      // (a blend behind the run wants the run's input -- the module's output -- as a buffer: such runs are not fused)
      dt_hip_mem_t fout = is_blend(gi + 2) ? NULL : (final_pair ? dev_out : dt_hip_alloc_device_buffer(devid, out_bytes(tail)));
      if(fout)
      {
        const node_t &bl = pipe->nodes[g.first];
        const int ferr = bilat_process_chain(devid, &bl.piece, bl.as<dt_hip_bilat_data_t>(), cur, fout, &gn.rgb, cells);
        if(ferr == DT_HIP_SUCCESS)
        {
          if(cells) dt_hip_release_mem_object(cells); // stream-ordered
          cells = NULL;
          if(cur_owned) dt_hip_release_mem_object(cur);
          cur = fout;
          cur_owned = !final_pair;
          gi++;
          continue;
        }
        if(!final_pair) dt_hip_release_mem_object(fout);
        if(ferr != DT_HIP_INVALID_ARG)
        {
          if(cur_owned) dt_hip_release_mem_object(cur);
          return ferr;
        }
      }
    }
    // diffuse or sharpen followed by the RGB -> Lab glue: the conversion is the tail of the module's last kernel
    if(pipe->fusion && g.kind == group_t::SINGLE && pipe->nodes[g.first].op == OP_DIFFUSE && gi + 1 < ng
       && pipe->groups[gi + 1].kind == group_t::SINGLE && pipe->nodes[pipe->groups[gi + 1].first].op == OP_RGB_TO_LAB
       && !pipe->nodes[pipe->groups[gi + 1].first].as<dt_hip_lab_data_t>()->nonlinearlut && !is_blend(gi + 2))
    {
      const node_t &df = pipe->nodes[g.first], &lab = pipe->nodes[pipe->groups[gi + 1].first];
      const bool final_pair = gi + 2 == ng;
      dt_hip_mem_t fout = final_pair ? dev_out : dt_hip_alloc_device_buffer(devid, out_bytes(lab));
      if(fout)
      {
        const int ferr = diffuse_process_post_lab(devid, &df.piece, df.as<dt_hip_diffuse_data_t>(), cur, fout, lab.as<dt_hip_lab_data_t>());
        if(ferr == DT_HIP_SUCCESS)
        {
          if(cur_owned) dt_hip_release_mem_object(cur);
          cur = fout;
          cur_owned = !final_pair;
          gi++;
          continue;
        }
        if(!final_pair) dt_hip_release_mem_object(fout);
        if(ferr != DT_HIP_INVALID_ARG)
        {
          if(cur_owned) dt_hip_release_mem_object(cur);
          return ferr;
        }
      }
    }
    dt_hip_mem_t out = dev_out;
    bool out_owned = false;
    const bool final_out = gi + 1 == ng || (is_blend(gi + 1) && gi + 2 == ng);
    if(!final_out)
    {
      out = dt_hip_alloc_device_buffer(devid, out_bytes(last));
      if(!out)
      {
        if(cur_owned) dt_hip_release_mem_object(cur);
        return DT_HIP_SYSMEM_ALLOCATION;
      }
      out_owned = true;
    }
    int err;
    if(g.kind == group_t::RAW)
      err = raw_group_launch(devid, g.raw, cur, out);
    else if(g.kind == group_t::RGB)
      err = rgb_group_launch(devid, g.rgb, cur, out);
    else
    {
      const node_t &nd = pipe->nodes[g.first];
      err = DT_HIP_INVALID_ARG;
      // denoise (non-local means) with local contrast's bilateral grid right behind it: the cells of the grid's third axis leave the
      // non-local-means kernels with the pixels (no blend in between: the grid is splatted from the module's own output)
      if(pipe->fusion && nd.op == OP_NLMEANS && !is_blend(gi + 1) && gi + 1 < ng && pipe->groups[gi + 1].kind == group_t::SINGLE
         && pipe->nodes[pipe->groups[gi + 1].first].op == OP_BILAT)
      {
        const node_t &bl = pipe->nodes[pipe->groups[gi + 1].first];
        float sigma_r = 0.0f;
        int size_z = 0;
        if(bilat_cell_params(&bl.piece, bl.as<dt_hip_bilat_data_t>(), &sigma_r, &size_z) == DT_HIP_SUCCESS
           && bl.piece.roi_in.width == nd.piece.roi_out.width && bl.piece.roi_in.height == nd.piece.roi_out.height)
        {
          cells = dt_hip_alloc_device_buffer(devid, (size_t)nd.piece.roi_out.width * nd.piece.roi_out.height * 2 * sizeof(float));
          if(cells)
          {
            err = nlmeans_process_cells(devid, &nd.piece, nd.as<dt_hip_nlmeans_data_t>(), cur, out, cells, sigma_r, size_z);
            if(err != DT_HIP_SUCCESS)
            {
              dt_hip_release_mem_object(cells);
              cells = NULL;
            }
          }
        }
      }
      else if(nd.op == OP_BILAT && cells)
      {
        err = bilat_process_cells(devid, &nd.piece, nd.as<dt_hip_bilat_data_t>(), cur, out, cells);
        dt_hip_release_mem_object(cells);
        cells = NULL;
      }
      if(err == DT_HIP_INVALID_ARG && !cells) err = run_single(devid, nd, cur, out);
    }
    if(err == DT_HIP_SUCCESS && is_blend(gi + 1))
    {
      held = cur;
      held_owned = cur_owned;
    }
    else if(cur_owned)
      dt_hip_release_mem_object(cur); // stream-ordered: re-used only by later launches
    if(err != DT_HIP_SUCCESS)
    {
      if(out_owned) dt_hip_release_mem_object(out);
      return err;
    }
    cur = out;
    cur_owned = out_owned;
  }
  return DT_HIP_SUCCESS;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------
// row bands (include/ansel_hip.h section 3b)
// ---------------------------------------------------------------------------------------------
namespace
{
const int RCD_TV = 94, RCD_TS = 112, RCD_HALO = 9; // tile pitch, tile size, RCD_BORDER (rcd.c:70-76)
const int AMZ_TV = 128, AMZ_HALO = 16;              // AMaZE: rows a tile keeps, rows it reads beyond them (amaze.cc:181-350)

struct band_priv_t
{
  dt_hip_mem_t cfa;     // output of the CFA stages (halo layout when a demosaic follows)
  bool cfa_owned;
  dt_hip_mem_t journal; // deferred highlights journal or nullptr
  dt_hip_mem_t hl_out;  // the buffer the journal indexes
  size_t next_group;
  // dt_hip_pipe_band_finish() is resumable: where the walk stands
  bool walking;
  dt_hip_mem_t cur, cur_base; // the band's own rows of the current module input / the allocation they live in
  bool cur_owned, cur_is_halo_layout;
  int stage;                  // inside a stencil group: 0 before the halo exchange, 1 after it, 2 after the sums, 3 done
  dt_hip_mem_t out;
  bool out_own_rows, out_owned;
  dt_hip_mem_t held, held_base; // own rows of the input of a module whose output is about to be blended
  bool held_owned;
  dn_band_job_t *dn_job;
  dt_hip_mem_t relay;  // local contrast: this band's copy of the frame's bilateral grid
  size_t relay_bytes;
};

bool is_stencil_op(op_t o) { return o == OP_DENOISEPROFILE || o == OP_DIFFUSE || o == OP_NLMEANS; }

// rows of its input a stencil module's own rows depend on beyond the band (-1: the module passes this frame through)
int band_halo_rows(const node_t &n)
{
  switch(n.op)
  {
    case OP_DENOISEPROFILE: return denoiseprofile_halo_rows(&n.piece, n.as<dt_hip_denoiseprofile_data_t>());
    case OP_DIFFUSE: return diffuse_halo_rows(&n.piece, n.as<dt_hip_diffuse_data_t>());
    case OP_NLMEANS: return nlmeans_halo_rows(&n.piece, n.as<dt_hip_nlmeans_data_t>());
    default: return 0;
  }
}

bool is_cfa_op(op_t o) { return o == OP_RAWPREPARE || o == OP_TEMPERATURE || o == OP_HIGHLIGHTS; }

// the band's view of a node: same columns, rows [row0, row0 + rows) of the frame
void band_piece(dt_hip_piece_t &p, const dt_hip_band_t &b)
{
  p.roi_in.y += b.row0;
  p.roi_in.height = b.rows;
  p.roi_out.y += b.row0;
  p.roi_out.height = b.rows;
}
void band_rawprepare(dt_hip_piece_t &p, dt_hip_rawprepare_data_t &d, const dt_hip_band_t &b)
{
  // the band input starts at input row crop_y + row0: fold the crop into the CFA phase
  p.roi_out.y += d.y + b.row0;
  d.y = 0;
  p.roi_in.height = b.rows;
  p.roi_out.height = b.rows;
}
} // namespace

extern "C" {

// ---- a stream of frames through one pipe: upload / kernels / download of consecutive frames overlap ----
struct batch_slot_t
{
  dt_hip_mem_t d_in, d_out;
  hipEvent_t up, done, down;
  hipEvent_t kstart; // in front of the frame's first kernel (timed with `done` and `down`: the upload policy below)
  bool in_flight;
  // with a writer: the frame's place in the stream, its host buffer, and what became of it (guarded by the batch's mutex)
  long seq;
  void *host_out;
  bool written;
  int write_err;
};

struct dt_hip_batch_t
{
  dt_hip_pipe_t *pipe;
  size_t in_bytes, out_bytes;
  hipStream_t s_up, s_down;
  std::vector<batch_slot_t> slots;
  int next;
  // who awaits a frame's upload -- the host (true) or the compute stream: decided from the frames that have completed
  // (dt_hip_batch_submit())
  bool host_awaits_upload;
  // the fourth leg: the format's write_image() of frame n on a host thread of its own while frames n + 1 ... are on the
  // device (imageio_core.c:965 runs it after the pipe, serially).  The thread takes the slots in submission order
  dt_hip_batch_writer_t writer;
  void *writer_user;
  long submitted;
  std::thread writer_thread;
  std::mutex mtx;
  std::condition_variable cv_job, cv_done;
  std::vector<int> jobs; // slots whose download is enqueued, oldest first
  bool quit;
};

namespace
{
static void batch_writer_loop(dt_hip_batch_t *b)
{
  (void)make_current(b->pipe->devid);
  for(;;)
  {
    int k;
    {
      std::unique_lock<std::mutex> lk(b->mtx);
      b->cv_job.wait(lk, [&] { return b->quit || !b->jobs.empty(); });
      if(b->jobs.empty()) return; // quit, and nothing left to write
      k = b->jobs.front();
      b->jobs.erase(b->jobs.begin());
    }
    batch_slot_t &sl = b->slots[k];
    int err = hipEventSynchronize(sl.down) == hipSuccess ? DT_HIP_SUCCESS : DT_HIP_DEFAULT_ERROR;
    if(err == DT_HIP_SUCCESS && b->writer(b->writer_user, sl.seq, sl.host_out, b->out_bytes) != 0) err = DT_HIP_WRITER_FAILED;
    {
      std::lock_guard<std::mutex> lk(b->mtx);
      sl.write_err = err;
      sl.written = true;
    }
    b->cv_done.notify_all();
  }
}
} // namespace

dt_hip_batch_t *dt_hip_batch_new(dt_hip_pipe_t *pipe, int depth, size_t in_bytes, size_t out_bytes)
{
  if(!pipe || depth < 1 || depth > 8 || !in_bytes || !out_bytes) return nullptr;
  dt_hip_batch_t *b = new dt_hip_batch_t;
  b->pipe = pipe;
  b->in_bytes = in_bytes;
  b->out_bytes = out_bytes;
  b->next = 0;
  b->host_awaits_upload = false;
  b->writer = nullptr;
  b->writer_user = nullptr;
  b->submitted = 0;
  b->quit = false;
  b->s_up = b->s_down = nullptr;
  // streams and events belong to the device that is current when they are made: the pipe's, not whatever the
  // calling thread used last (hipEventRecord rejects an event of another device than its stream's)
  bool ok = make_current(pipe->devid) && hipStreamCreateWithFlags(&b->s_up, hipStreamNonBlocking) == hipSuccess
            && hipStreamCreateWithFlags(&b->s_down, hipStreamNonBlocking) == hipSuccess;
  for(int k = 0; k < depth && ok; k++)
  {
    batch_slot_t sl;
    memset(&sl, 0, sizeof(sl));
    sl.d_in = dt_hip_alloc_device_buffer(pipe->devid, in_bytes);
    sl.d_out = dt_hip_alloc_device_buffer(pipe->devid, out_bytes);
    ok = sl.d_in && sl.d_out && hipEventCreateWithFlags(&sl.up, hipEventDisableTiming) == hipSuccess
         && hipEventCreate(&sl.kstart) == hipSuccess && hipEventCreate(&sl.done) == hipSuccess && hipEventCreate(&sl.down) == hipSuccess;
    b->slots.push_back(sl);
  }
  if(!ok)
  {
    set_last_error("dt_hip_batch_new: could not create %d slots of %zu + %zu bytes", depth, in_bytes, out_bytes);
    dt_hip_batch_free(b);
    return nullptr;
  }
  return b;
}

void dt_hip_batch_free(dt_hip_batch_t *b)
{
  if(!b) return;
  make_current(b->pipe->devid);
  dt_hip_batch_drain(b);
  if(b->writer_thread.joinable())
  {
    {
      std::lock_guard<std::mutex> lk(b->mtx);
      b->quit = true;
    }
    b->cv_job.notify_all();
    b->writer_thread.join();
  }
  // a submit that failed half way leaves its upload (or download) enqueued without marking the slot in flight:
  // the copy streams must be idle before the slot buffers go back to the pool
  if(b->s_up) (void)hipStreamSynchronize(b->s_up);
  if(b->s_down) (void)hipStreamSynchronize(b->s_down);
  for(batch_slot_t &sl : b->slots)
  {
    if(sl.d_in) dt_hip_release_mem_object(sl.d_in);
    if(sl.d_out) dt_hip_release_mem_object(sl.d_out);
    if(sl.up) (void)hipEventDestroy(sl.up);
    if(sl.kstart) (void)hipEventDestroy(sl.kstart);
    if(sl.done) (void)hipEventDestroy(sl.done);
    if(sl.down) (void)hipEventDestroy(sl.down);
  }
  if(b->s_up) (void)hipStreamDestroy(b->s_up);
  if(b->s_down) (void)hipStreamDestroy(b->s_down);
  delete b;
}

namespace
{
// Who awaits the NEXT frames' uploads.  A frame whose kernels take longer than its two transfers (the full pipe: 67 ms against 3.6 +
// 15 at 100 MP) wants the HOST to await the upload: with a stream-wait in front of the kernels AND the download's stream-wait behind
// them, the runtime ran the downloads beside the next frame's kernels at the sum of their times (79 - 83 ms a frame; 71 with the
// host awaiting -- tools/batch_sdma_probe.py, profiles/r06_batch_probe.txt).  A frame whose transfers are the longer leg (the light
// pipe: 5.7 ms of kernels) wants everything asynchronous: the host blocked on an upload cannot enqueue the next download (24.8
// against 16.3 ms a frame).  Measured per completed frame from the slot's events.
static void batch_update_policy(dt_hip_batch_t *b, batch_slot_t &sl)
{
  float kernels_ms = 0.0f, down_ms = 0.0f;
  if(hipEventElapsedTime(&kernels_ms, sl.kstart, sl.done) != hipSuccess || hipEventElapsedTime(&down_ms, sl.done, sl.down) != hipSuccess)
  {
    (void)hipGetLastError();
    return;
  }
  const float up_ms = down_ms * (float)((double)b->in_bytes / (double)b->out_bytes);
  b->host_awaits_upload = kernels_ms > up_ms + down_ms;
}
} // namespace

int dt_hip_batch_wait(dt_hip_batch_t *b, int slot)
{
  if(!b || slot < 0 || slot >= (int)b->slots.size()) return DT_HIP_INVALID_ARG;
  batch_slot_t &sl = b->slots[slot];
  if(!sl.in_flight) return DT_HIP_SUCCESS;
  if(b->writer)
  {
    // the frame is done when its writer has returned: only then may the caller reuse host_out
    std::unique_lock<std::mutex> lk(b->mtx);
    b->cv_done.wait(lk, [&] { return sl.written; });
    sl.in_flight = false;
    if(sl.write_err == DT_HIP_SUCCESS) batch_update_policy(b, sl);
    if(sl.write_err == DT_HIP_WRITER_FAILED) set_last_error("dt_hip_batch_wait: the writer refused frame %ld", sl.seq);
    else if(sl.write_err != DT_HIP_SUCCESS)
      set_last_error("dt_hip_batch_wait: the download of frame %ld did not complete (hipEventSynchronize on the writer thread)", sl.seq);
    return sl.write_err;
  }
  ANSEL_HIP_CHECK(hipEventSynchronize(sl.down));
  sl.in_flight = false;
  batch_update_policy(b, sl);
  return DT_HIP_SUCCESS;
}

int dt_hip_batch_set_writer(dt_hip_batch_t *b, dt_hip_batch_writer_t writer, void *user)
{
  if(!b) return DT_HIP_INVALID_ARG;
  // between frames only: no slot may be in flight
  const int e = dt_hip_batch_drain(b);
  if(e != DT_HIP_SUCCESS) return e;
  b->writer = writer;
  b->writer_user = user;
  if(writer && !b->writer_thread.joinable())
  {
    // std::thread's constructor throws std::system_error when the system has no thread to give: not through a C boundary
    try
    {
      b->writer_thread = std::thread(batch_writer_loop, b);
    }
    catch(const std::exception &e)
    {
      b->writer = nullptr;
      b->writer_user = nullptr;
      set_last_error("dt_hip_batch_set_writer: cannot start the writer thread (%s)", e.what());
      return DT_HIP_DEFAULT_ERROR;
    }
  }
  return DT_HIP_SUCCESS;
}

int dt_hip_batch_drain(dt_hip_batch_t *b)
{
  if(!b) return DT_HIP_INVALID_ARG;
  int err = DT_HIP_SUCCESS;
  for(int k = 0; k < (int)b->slots.size(); k++)
  {
    const int e = dt_hip_batch_wait(b, k);
    if(e != DT_HIP_SUCCESS) err = e;
  }
  return err;
}

int dt_hip_batch_submit(dt_hip_batch_t *b, const void *host_in, void *host_out)
{
  if(!b || !host_in || !host_out) return DT_HIP_INVALID_ARG;
  const int k = b->next;
  batch_slot_t &sl = b->slots[k];
  // the slot's previous frame must have left the device before its buffers are reused
  const int w = dt_hip_batch_wait(b, k);
  if(w != DT_HIP_SUCCESS)
  {
    // the failure belongs to the frame that held this slot, NOT to the one being submitted (which is not submitted):
    // the message says so, the slot is free again, and the caller may submit the same frame once more
    const std::string prev = dt_hip_last_error();
    set_last_error("dt_hip_batch_submit: the previous frame of slot %d failed (%s); the new frame was not submitted", k, prev.c_str());
    return w;
  }
  hipStream_t compute = stream_of(b->pipe->devid);
  ANSEL_HIP_CHECK(hipMemcpyAsync(sl.d_in, host_in, b->in_bytes, hipMemcpyHostToDevice, b->s_up));
  ANSEL_HIP_CHECK(hipEventRecord(sl.up, b->s_up));
  if(b->host_awaits_upload) ANSEL_HIP_CHECK(hipEventSynchronize(sl.up)); // (batch_update_policy(): which, and why)
  else ANSEL_HIP_CHECK(hipStreamWaitEvent(compute, sl.up, 0));
  ANSEL_HIP_CHECK(hipEventRecord(sl.kstart, compute));
  const int err = dt_hip_pipe_process(b->pipe, sl.d_in, sl.d_out);
  if(err != DT_HIP_SUCCESS) return err;
  ANSEL_HIP_CHECK(hipEventRecord(sl.done, compute));
  ANSEL_HIP_CHECK(hipStreamWaitEvent(b->s_down, sl.done, 0));
  ANSEL_HIP_CHECK(hipMemcpyAsync(host_out, sl.d_out, b->out_bytes, hipMemcpyDeviceToHost, b->s_down));
  ANSEL_HIP_CHECK(hipEventRecord(sl.down, b->s_down));
  sl.in_flight = true;
  if(b->writer)
  {
    {
      std::lock_guard<std::mutex> lk(b->mtx);
      sl.seq = b->submitted;
      sl.host_out = host_out;
      sl.written = false;
      sl.write_err = DT_HIP_SUCCESS;
      b->jobs.push_back(k);
    }
    b->cv_job.notify_one();
  }
  b->submitted++;
  b->next = (k + 1) % (int)b->slots.size();
  return k;
}


int dt_hip_plan_bands(int width, int height, int demosaic_method, int n_bands, dt_hip_band_t *bands)
{
  if(width <= 0 || height <= 0 || n_bands <= 0 || !bands) return DT_HIP_INVALID_ARG;
  memset(bands, 0, sizeof(dt_hip_band_t) * (size_t)n_bands);
  if(demosaic_method == DT_HIP_DEMOSAIC_RCD)
  {
    if(width < 16 || height < 16) return DT_HIP_INVALID_ARG;
    const int num_vertical = 1 + (height - 2 * RCD_HALO - 1) / RCD_TV; // rcd.c:286
    if(num_vertical < n_bands)
    {
      set_last_error("dt_hip_plan_bands: %d rows give %d RCD tile rows, fewer than %d bands", height, num_vertical, n_bands);
      return DT_HIP_INVALID_ARG;
    }
    for(int k = 0; k < n_bands; k++)
    {
      const int tv0 = (int)((long)k * num_vertical / n_bands), tv1 = (int)((long)(k + 1) * num_vertical / n_bands);
      dt_hip_band_t &b = bands[k];
      b.tile_row0 = tv0;
      b.tile_row1 = tv1;
      b.row0 = tv0 ? tv0 * RCD_TV + RCD_HALO : 0;
      const int row1 = (tv1 < num_vertical) ? tv1 * RCD_TV + RCD_HALO : height;
      b.rows = row1 - b.row0;
      b.halo_top = b.row0 - tv0 * RCD_TV;
      const int need1 = (tv1 - 1) * RCD_TV + RCD_TS < height ? (tv1 - 1) * RCD_TV + RCD_TS : height;
      b.halo_bottom = need1 > row1 ? need1 - row1 : 0;
    }
    return DT_HIP_SUCCESS;
  }
  if(demosaic_method == DT_HIP_DEMOSAIC_AMAZE)
  {
    // AMaZE's own tiles (amaze.cc:181-350): 160 rows of the mosaic 16 above a tile row's 128 kept rows.  A band owns whole
    // tile rows, so it needs 16 mosaic rows of either neighbour; the rows a tile mirrors at the frame's bottom edge lie in
    // the last band's own rows and in what the band above it fetches of them (fewer than 16 rows are left there)
    if(width < 34 || height < 34) return DT_HIP_INVALID_ARG;
    const int tile_rows = (height + AMZ_TV - 1) / AMZ_TV;
    if(tile_rows < n_bands)
    {
      set_last_error("dt_hip_plan_bands: %d rows give %d AMaZE tile rows, fewer than %d bands", height, tile_rows, n_bands);
      return DT_HIP_INVALID_ARG;
    }
    // only the on-chip kernel walks a band (demosaic_amaze.hip): a frame that keeps tiles in the first kernel's body -- a
    // last tile column of odd width, a mirrored strip past its plane -- would be refused by the band's demosaic launch,
    // after the CFA stages and the halo copies of every band have run.  Say so here, where the caller can still take
    // the unsplit path
    for(int ty = 0; ty < tile_rows; ty++)
      for(int tx = 0; tx < (width + AMZ_HALO + AMZ_TV - 1) / AMZ_TV; tx++) // the launch's tile columns (demosaic_amaze.hip)
        if(!amz::stream_tile_ok(width, height, -AMZ_HALO + ty * AMZ_TV, -AMZ_HALO + tx * AMZ_TV))
        {
          set_last_error("dt_hip_plan_bands: the AMaZE tile at row %d, column %d of a %d x %d frame is not one the on-chip kernel takes "
                         "(odd width of the last tile column, or a mirrored strip past its plane): no band mode for this frame",
                         ty * AMZ_TV, tx * AMZ_TV, width, height);
          return DT_HIP_INVALID_ARG;
        }
    for(int k = 0; k < n_bands; k++)
    {
      const int tv0 = (int)((long)k * tile_rows / n_bands), tv1 = (int)((long)(k + 1) * tile_rows / n_bands);
      dt_hip_band_t &b = bands[k];
      b.tile_row0 = tv0;
      b.tile_row1 = tv1;
      b.row0 = tv0 * AMZ_TV;
      const int row1 = tv1 < tile_rows ? tv1 * AMZ_TV : height;
      b.rows = row1 - b.row0;
      b.halo_top = tv0 ? AMZ_HALO : 0;
      b.halo_bottom = height - row1 < AMZ_HALO ? height - row1 : AMZ_HALO;
    }
    return DT_HIP_SUCCESS;
  }
  if(demosaic_method != -1)
  {
    set_last_error("dt_hip_plan_bands: demosaic method %d has no band mode", demosaic_method);
    return DT_HIP_INVALID_ARG;
  }
  if(height / 2 < n_bands) return DT_HIP_INVALID_ARG;
  for(int k = 0; k < n_bands; k++)
  {
    const int r0 = (int)((long)k * (height / 2) / n_bands) * 2;
    const int r1 = (k + 1 == n_bands) ? height : (int)((long)(k + 1) * (height / 2) / n_bands) * 2;
    bands[k].row0 = r0;
    bands[k].rows = r1 - r0;
  }
  return DT_HIP_SUCCESS;
}

int dt_hip_pipe_band_begin(dt_hip_pipe_t *pipe, const dt_hip_band_t *band, dt_hip_mem_t dev_in_band,
                           dt_hip_band_state_t *state)
{
  if(!pipe || !band || !dev_in_band || !state || band->rows <= 0) return DT_HIP_INVALID_ARG;
  memset(state, 0, sizeof(*state));
  if(pipe->nodes.empty()) return DT_HIP_INVALID_ARG;
  if(!pipe->planned) pipe->plan();
  const int devid = pipe->devid;
  const dt_hip_band_t &b = *band;
  const int W = pipe->nodes[0].piece.roi_out.width, H = pipe->nodes[0].piece.roi_out.height;
  for(const node_t &n : pipe->nodes)
    if(n.piece.roi_out.width != W || n.piece.roi_out.height != H)
    {
      set_last_error("band mode: every node must produce the same %d x %d geometry", W, H);
      return DT_HIP_INVALID_ARG;
    }
  if(b.row0 < 0 || b.row0 + b.rows > H) return DT_HIP_INVALID_ARG;
  for(const node_t &n : pipe->nodes)
  {
    if(n.op == OP_FINALSCALE || n.op == OP_INITIALSCALE)
    {
      // finalscale / initialscale change the geometry
      set_last_error("band mode: '%s' has no row-band implementation", k_ops[n.op].name);
      return DT_HIP_INVALID_ARG;
    }
    if(n.op == OP_BILAT && !bilat_band_supported(&n.piece, n.as<dt_hip_bilat_data_t>()))
    {
      // the bilateral grid is relayed from band to band; the local laplacian's pyramid is not
      set_last_error("band mode: local contrast runs on row bands in its bilateral-grid mode only");
      return DT_HIP_INVALID_ARG;
    }
    if(n.op == OP_DETAILMASK || (n.op == OP_BLEND && blend_refines_with_detail_mask(n.as<dt_hip_blend_data_t>())))
    {
      // the raw detail mask is one plane of the frame on one device; its 9 x 9 blur reads across band borders
      set_last_error("band mode: the detail mask (the \"detailmask\" stage, a blend's details threshold) has no row-band "
                     "implementation");
      return DT_HIP_INVALID_ARG;
    }
    if(n.op == OP_BLEND && n.as<dt_hip_blend_data_t>()->feathering_radius > 0.1f)
    {
      // the guided filter works on its own 512-pixel tile grid over the whole frame
      set_last_error("band mode: a blend with mask feathering has no row-band implementation");
      return DT_HIP_INVALID_ARG;
    }
    if(n.op == OP_BLEND && n.as<dt_hip_blend_data_t>()->blur_radius > 0.0f)
    {
      // uniform and parametric masks are pointwise; the mask blur is a recursive filter down whole columns
      set_last_error("band mode: a blend with a mask blur has no row-band implementation");
      return DT_HIP_INVALID_ARG;
    }
  }
  const size_t ng = pipe->groups.size();
  // the CFA stage ends where the first non-CFA group starts
  size_t n_cfa = 0;
  while(n_cfa < ng && is_cfa_op(pipe->nodes[pipe->groups[n_cfa].first].op)) n_cfa++;
  const bool has_demosaic = n_cfa < ng && pipe->nodes[pipe->groups[n_cfa].first].op == OP_DEMOSAIC;
  if(!has_demosaic && (b.halo_top || b.halo_bottom)) return DT_HIP_INVALID_ARG;
  band_priv_t *pv = new band_priv_t;
  memset(pv, 0, sizeof(*pv));
  const size_t row_bytes = (size_t)W * 4;
  const size_t halo_rows = (size_t)b.halo_top + b.rows + b.halo_bottom;

  int err = DT_HIP_SUCCESS;
  dt_hip_mem_t cur = dev_in_band;
  bool cur_owned = false;
  for(size_t gi = 0; gi < n_cfa && err == DT_HIP_SUCCESS; gi++)
  {
    const group_t &g = pipe->groups[gi];
    const bool last_cfa = gi + 1 == n_cfa;
    dt_hip_mem_t buf = dt_hip_alloc_device_buffer(devid, last_cfa ? halo_rows * row_bytes : (size_t)b.rows * row_bytes);
    if(!buf)
    {
      err = DT_HIP_SYSMEM_ALLOCATION;
      break;
    }
    dt_hip_mem_t out = last_cfa ? (dt_hip_mem_t)((char *)buf + (size_t)b.halo_top * row_bytes) : buf;
    const node_t &first = pipe->nodes[g.first];
    if(g.kind == group_t::RAW)
    {
      raw_group_t r = g.raw;
      band_rawprepare(r.rawprepare_piece, r.rawprepare, b);
      if(r.has_temperature) band_piece(r.temperature_piece, b);
      if(r.has_highlights)
      {
        band_piece(r.highlights_piece, b);
        pv->journal = dt_hip_alloc_device_buffer(devid, DT_HIP_HIGHLIGHTS_JOURNAL_BYTES);
        pv->hl_out = out;
        if(!pv->journal) err = DT_HIP_SYSMEM_ALLOCATION;
      }
      if(err == DT_HIP_SUCCESS) err = raw_group_launch(devid, r, cur, out, pv->journal);
    }
    else
    {
      dt_hip_piece_t p = first.piece;
      if(first.op == OP_RAWPREPARE)
      {
        dt_hip_rawprepare_data_t d = *first.as<dt_hip_rawprepare_data_t>();
        band_rawprepare(p, d, b);
        err = dt_hip_iop_rawprepare_process(devid, &p, &d, cur, out);
      }
      else if(first.op == OP_TEMPERATURE)
      {
        band_piece(p, b);
        err = dt_hip_iop_temperature_process(devid, &p, first.as<dt_hip_temperature_data_t>(), cur, out);
      }
      else
      {
        band_piece(p, b);
        pv->journal = dt_hip_alloc_device_buffer(devid, DT_HIP_HIGHLIGHTS_JOURNAL_BYTES);
        pv->hl_out = out;
        if(!pv->journal) err = DT_HIP_SYSMEM_ALLOCATION;
        else err = dt_hip_iop_highlights_process_deferred(devid, &p, first.as<dt_hip_highlights_data_t>(), cur, out, pv->journal);
      }
    }
    if(cur_owned) dt_hip_release_mem_object(cur);
    cur = buf;
    cur_owned = true;
  }
  if(err == DT_HIP_SUCCESS && n_cfa == 0 && has_demosaic)
  {
    // the pipe starts at demosaic: stage the band's mosaic rows into the halo layout
    dt_hip_mem_t buf = dt_hip_alloc_device_buffer(devid, halo_rows * row_bytes);
    if(!buf) err = DT_HIP_SYSMEM_ALLOCATION;
    else
    {
      err = dt_hip_enqueue_copy_buffer_to_buffer(devid, cur, buf, 0, (size_t)b.halo_top * row_bytes, (size_t)b.rows * row_bytes);
      cur = buf;
      cur_owned = true;
    }
  }
  if(err != DT_HIP_SUCCESS)
  {
    if(cur_owned) dt_hip_release_mem_object(cur);
    if(pv->journal) dt_hip_release_mem_object(pv->journal);
    delete pv;
    return err;
  }
  pv->cfa = cur;
  pv->cfa_owned = cur_owned;
  pv->next_group = n_cfa;
  state->halo_buf = (has_demosaic && cur_owned) ? cur : nullptr;
  state->row_bytes = row_bytes;
  state->clipped_count = pv->journal;
  state->priv = pv;
  return DT_HIP_SUCCESS;
}

int dt_hip_pipe_band_resolve(dt_hip_pipe_t *pipe, const dt_hip_band_t *band, dt_hip_band_state_t *state)
{
  if(!pipe || !band || !state || !state->priv) return DT_HIP_INVALID_ARG;
  band_priv_t *pv = (band_priv_t *)state->priv;
  int err = DT_HIP_SUCCESS;
  if(pv->journal)
  {
    // must precede the halo exchange: the neighbours read these rows after the bypass decision
    err = dt_hip_iop_highlights_resolve(pipe->devid, pv->hl_out, pv->journal);
    dt_hip_release_mem_object(pv->journal);
    pv->journal = nullptr;
    state->clipped_count = nullptr;
  }
  return err;
}

// ---- one frame over the devices of ONE process (BASELINE.json config 4 from C) ----------------------------------
// The reference is a single C process (src/develop/pixelpipe_hb.c:1470): it cannot run one rank per GPU under a
// launcher, so the band walk of section 3b is also driven from inside the library -- one host thread per band (a
// module's launch code may block on ITS device, e.g. the patch table upload of the non-local means; with a thread
// per device the others keep enqueueing), the bands in lockstep at the exchange points, the collectives as peer
// copies over xGMI:
//   * the clipped count of the highlights bypass: 8 bytes per band through the host, summed in band order (integers);
//   * halo rows: each band PULLS the rows it needs from its neighbours' buffers (hipMemcpyPeerAsync on its own
//     stream), after every band has finished the step that produces them and before any band goes on;
//   * the profiled wavelets' table of partial sums: every entry is non-zero in exactly one band's table (the band
//     that owns the row), so the all-reduce is an all-gather of row segments -- one strided peer copy per
//     neighbour and band, exact by construction (x + 0 + ... + 0), no arithmetic at all.
// Bands may share a device (the single-GPU test of this path): a peer copy is then a device copy.
namespace
{
// What the bands of one walk share.  A band publishes POINTS: a hipEvent recorded on its stream behind the work the
// point stands for, and a counter the other bands' host threads watch.  Somebody who needs that work waits on the HOST
// only until the event has been recorded (the owner's thread got that far enqueueing), then makes ITS stream wait for
// the event: no stream is ever drained inside the walk, and a band only waits for the bands it reads from (its two
// neighbours at a halo stop).  Every band passes the same points in the same order (same node list).
struct band_gang_t
{
  int n = 0;
  std::mutex m;
  std::condition_variable cv;
  std::vector<int> posted;                    // points band k has published
  std::vector<std::vector<hipEvent_t>> events; // [band][point]
  std::vector<int> done;                      // the band's walk has ended (its posted count is final)
  bool failed = false;
  // the classic meeting, used once per frame for the 8-byte clipped count that travels through the host
  int waiting = 0;
  unsigned long generation = 0;
  // statistics of the last walk (dt_hip_pipe_bands_stats())
  std::atomic<unsigned long long> peer_bytes{ 0 }, peer_copies{ 0 }, host_wait_ns{ 0 };

  void meet()
  {
    std::unique_lock<std::mutex> lk(m);
    const unsigned long g = generation;
    if(++waiting == n)
    {
      waiting = 0;
      generation++;
      cv.notify_all();
    }
    else
      cv.wait(lk, [&] { return generation != g; });
  }
  // band k: "everything enqueued on `s` so far is point number posted[k]"
  bool publish(const int k, hipStream_t s)
  {
    hipEvent_t e = nullptr;
    if(hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess || hipEventRecord(e, s) != hipSuccess)
    {
      (void)hipGetLastError();
      if(e) (void)hipEventDestroy(e);
      fail();
      return false;
    }
    std::lock_guard<std::mutex> lk(m);
    events[k].push_back(e);
    posted[k]++;
    cv.notify_all();
    return true;
  }
  // make stream `s` wait for point `pt` of band j; false when the walk has failed or band j will never get there
  bool await(const int j, const int pt, hipStream_t s)
  {
    hipEvent_t e = nullptr;
    {
      const auto t0 = std::chrono::steady_clock::now();
      std::unique_lock<std::mutex> lk(m);
      cv.wait(lk, [&] { return failed || posted[j] > pt || done[j]; });
      host_wait_ns += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
      if(failed || posted[j] <= pt) return false;
      e = events[j][pt];
    }
    if(hipStreamWaitEvent(s, e, 0) != hipSuccess)
    {
      (void)hipGetLastError();
      fail();
      return false;
    }
    return true;
  }
  void fail()
  {
    std::lock_guard<std::mutex> lk(m);
    failed = true;
    cv.notify_all();
  }
  void finished(const int k)
  {
    std::lock_guard<std::mutex> lk(m);
    done[k] = 1;
    cv.notify_all();
  }
  bool has_failed()
  {
    std::lock_guard<std::mutex> lk(m);
    return failed;
  }
};

dt_hip_band_stats_t g_band_stats = { 0, 0, 0, 0, 0, 0, 0 };
std::mutex g_band_stats_mutex;

static int copy_between(band_gang_t &gang, const int dst_devid, void *dst, const int src_devid, const void *src, const size_t bytes,
                 hipStream_t s)
{
  if(!bytes) return DT_HIP_SUCCESS;
  const int dd = hip_device_of(dst_devid), sd = hip_device_of(src_devid);
  if(dd == sd) ANSEL_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
  else
  {
    ANSEL_HIP_CHECK(hipMemcpyPeerAsync(dst, dd, src, sd, bytes, s));
    gang.peer_bytes += bytes;
    gang.peer_copies++;
  }
  return DT_HIP_SUCCESS;
}
} // namespace

void dt_hip_pipe_bands_stats(dt_hip_band_stats_t *out)
{
  if(!out) return;
  std::lock_guard<std::mutex> lk(g_band_stats_mutex);
  *out = g_band_stats;
}

// Can the devices of a band walk reach each other?  Checks hipDeviceCanAccessPeer for every ordered pair, enables the
// access, and moves a small buffer device to device and back with a cross-device event in between -- the three things
// dt_hip_pipe_process_bands() relies on and a single-GPU box never executes.  0, or an error with the pair in the text.
int dt_hip_peer_selftest(const int *devids, int n)
{
  if(!devids || n < 1) return DT_HIP_INVALID_ARG;
  for(int i = 0; i < n; i++)
    if(!valid_device(devids[i])) return DT_HIP_INVALID_ARG;
  for(int i = 0; i < n; i++)
    for(int j = 0; j < n; j++)
    {
      const int di = hip_device_of(devids[i]), dj = hip_device_of(devids[j]);
      if(di == dj) continue;
      int can = 0;
      if(hipDeviceCanAccessPeer(&can, di, dj) != hipSuccess || !can)
      {
        (void)hipGetLastError();
        set_last_error("peer self-test: device %d cannot access device %d (hipDeviceCanAccessPeer): halo rows would travel "
                       "through the host", devids[i], devids[j]);
        return DT_HIP_DEFAULT_ERROR;
      }
      (void)stream_of(devids[i]); // makes device i current
      const hipError_t e = hipDeviceEnablePeerAccess(dj, 0);
      if(e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
      {
        (void)hipGetLastError();
        set_last_error("peer self-test: hipDeviceEnablePeerAccess(%d -> %d): %s", devids[i], devids[j], hipGetErrorString(e));
        return DT_HIP_DEFAULT_ERROR;
      }
      (void)hipGetLastError();
    }
  // every ORDERED pair (a, b): a's pattern travels to b behind an event of a's stream -- dt_hip_pipe_process_bands() pulls halo
  // rows from both neighbours, the bilateral grid from the last band to every band, the wavelets' sums from every band to every
  // band -- once as a linear peer copy and once as the strided hipMemcpy2DAsync(hipMemcpyDefault) the sums' all-gather uses
  const size_t N = 1 << 16;
  std::vector<unsigned> pattern(N), back(N);
  for(int pair = 0; pair < (n == 1 ? 1 : n * n); pair++)
  {
    const int ka = n == 1 ? 0 : pair / n, kb = n == 1 ? 0 : pair % n;
    if(n > 1 && ka == kb) continue;
    const int k = pair;
    const int a = devids[ka], b = devids[kb];
    for(size_t i = 0; i < N; i++) pattern[i] = (unsigned)(i * 2654435761u + (unsigned)k);
    unsigned *da = (unsigned *)dt_hip_alloc_device_buffer(a, N * 4), *db = (unsigned *)dt_hip_alloc_device_buffer(b, N * 4);
    int err = (da && db) ? DT_HIP_SUCCESS : DT_HIP_SYSMEM_ALLOCATION;
    hipEvent_t ev = nullptr;
    if(err == DT_HIP_SUCCESS)
    {
      hipStream_t sa = stream_of(a);
      if(hipMemcpyAsync(da, pattern.data(), N * 4, hipMemcpyHostToDevice, sa) != hipSuccess
         || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, sa) != hipSuccess)
        err = DT_HIP_DEFAULT_ERROR;
      hipStream_t sb = stream_of(b);
      if(err == DT_HIP_SUCCESS
         && (hipStreamWaitEvent(sb, ev, 0) != hipSuccess
             || (hip_device_of(a) == hip_device_of(b) ? hipMemcpyAsync(db, da, N * 4, hipMemcpyDeviceToDevice, sb)
                                                      : hipMemcpyPeerAsync(db, hip_device_of(b), da, hip_device_of(a), N * 4, sb))
                    != hipSuccess
             || hipMemcpyAsync(back.data(), db, N * 4, hipMemcpyDeviceToHost, sb) != hipSuccess
             || hipStreamSynchronize(sb) != hipSuccess))
        err = DT_HIP_DEFAULT_ERROR;
      if(err == DT_HIP_SUCCESS && memcmp(back.data(), pattern.data(), N * 4) != 0) err = DT_HIP_DEFAULT_ERROR;
      // the strided form: 64 rows of 256 words out of rows of 1024, kind Default (the runtime routes between the two memories)
      if(err == DT_HIP_SUCCESS
         && (hipMemsetAsync(db, 0, N * 4, sb) != hipSuccess
             || hipMemcpy2DAsync(db + 128, 1024 * 4, da + 128, 1024 * 4, 256 * 4, 64, hipMemcpyDefault, sb) != hipSuccess
             || hipMemcpyAsync(back.data(), db, N * 4, hipMemcpyDeviceToHost, sb) != hipSuccess
             || hipStreamSynchronize(sb) != hipSuccess))
        err = DT_HIP_DEFAULT_ERROR;
      for(size_t i = 0; err == DT_HIP_SUCCESS && i < N; i++)
      {
        const size_t col = i % 1024;
        if(back[i] != ((col >= 128 && col < 384) ? pattern[i] : 0u)) err = DT_HIP_DEFAULT_ERROR;
      }
      (void)hipStreamSynchronize(sa);
    }
    if(ev) (void)hipEventDestroy(ev);
    if(da) dt_hip_release_mem_object(da);
    if(db) dt_hip_release_mem_object(db);
    if(err != DT_HIP_SUCCESS)
    {
      (void)hipGetLastError();
      set_last_error("peer self-test: the copy device %d -> device %d behind a cross-device event did not arrive intact", a, b);
      return err;
    }
  }
  return DT_HIP_SUCCESS;
}

int dt_hip_pipe_process_bands(dt_hip_pipe_t *const *pipes, int n, const dt_hip_band_t *bands, const dt_hip_mem_t *dev_in,
                              const dt_hip_mem_t *dev_out)
{
  if(!pipes || n < 1 || n > 64 || !bands || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  for(int k = 0; k < n; k++)
    if(!pipes[k] || !dev_in[k] || !dev_out[k] || pipes[k]->nodes.empty() || !valid_device(pipes[k]->devid)) return DT_HIP_INVALID_ARG;
  const int W = pipes[0]->nodes[0].piece.roi_out.width, H = pipes[0]->nodes[0].piece.roi_out.height;
  for(int k = 0; k < n; k++)
  {
    bool same = pipes[k]->nodes.size() == pipes[0]->nodes.size() && pipes[k]->nodes[0].piece.roi_out.width == W
                && pipes[k]->nodes[0].piece.roi_out.height == H;
    for(size_t i = 0; same && i < pipes[k]->nodes.size(); i++) same = pipes[k]->nodes[i].op == pipes[0]->nodes[i].op;
    if(!same)
    {
      set_last_error("dt_hip_pipe_process_bands: pipe %d does not hold the node list of pipe 0", k);
      return DT_HIP_INVALID_ARG;
    }
    if(bands[k].row0 != (k ? bands[k - 1].row0 + bands[k - 1].rows : 0) || (k + 1 == n && bands[k].row0 + bands[k].rows != H))
    {
      set_last_error("dt_hip_pipe_process_bands: the bands do not tile the %d rows of the frame", H);
      return DT_HIP_INVALID_ARG;
    }
  }
  // the mosaic halo is pulled out of the neighbour's OWN rows: a band thinner than it cannot serve it
  for(int k = 0; k < n; k++)
    if((k > 0 && bands[k].halo_top > bands[k - 1].rows) || (k + 1 < n && bands[k].halo_bottom > bands[k + 1].rows))
    {
      set_last_error("dt_hip_pipe_process_bands: band %d owns fewer rows than the mosaic halo its neighbour needs: use fewer bands", k);
      return DT_HIP_INVALID_ARG;
    }
  band_gang_t gang;
  gang.n = n;
  gang.posted.assign(n, 0);
  gang.done.assign(n, 0);
  gang.events.resize(n);
  std::vector<int> rcs(n, DT_HIP_SUCCESS);
  std::vector<dt_hip_band_state_t> st(n);
  std::vector<unsigned long long> counts(n, 0ull);
  std::vector<std::string> errors(n);
  std::atomic<int> peer_missing{ 0 }, stops{ 0 };
  for(auto &x : st) memset(&x, 0, sizeof(x));

  auto worker = [&](const int k) {
    dt_hip_pipe_t *const pipe = pipes[k];
    const int devid = pipe->devid;
    const dt_hip_band_t &b = bands[k];
    hipStream_t s = stream_of(devid); // also makes the device current for this thread
    // direct loads / stores between the devices of the gang.  A pair without peer access still works (the runtime
    // stages the copies through the host) but not at xGMI speed: counted, reported by dt_hip_pipe_bands_stats()
    for(int j = 0; j < n; j++)
      if(hip_device_of(pipes[j]->devid) != hip_device_of(devid))
      {
        int can = 0;
        if(hipDeviceCanAccessPeer(&can, hip_device_of(devid), hip_device_of(pipes[j]->devid)) != hipSuccess || !can) peer_missing++;
        else
        {
          const hipError_t e = hipDeviceEnablePeerAccess(hip_device_of(pipes[j]->devid), 0);
          if(e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) peer_missing++;
        }
        (void)hipGetLastError();
      }
    bool walking = false; // the band state holds buffers that give_up() must free
    auto fail = [&](const int code, const char *what = nullptr) {
      rcs[k] = code;
      errors[k] = what ? what : dt_hip_last_error();
      gang.fail();
    };
    auto give_up = [&]() {
      // the failure is published (fail() before every give_up() that follows an error of this band) and the own stream drained
      // BEFORE the band's buffers go back to the pool: a healthy neighbour may have peer copies in flight that read them --
      // what they copy is discarded, but it must still be this band's memory -- and a fault stays attributed to this band
      (void)hipStreamSynchronize(s);
      if(walking) dt_hip_pipe_band_abort(pipe, &st[k]);
      walking = false;
      gang.finished(k);
    };
    auto await = [&](const int j, const int pt) -> bool {
      if(j == k) return true;
      if(!gang.await(j, pt, s))
      {
        if(rcs[k] >= 0) rcs[k] = DT_HIP_DEFAULT_ERROR, errors[k] = "another band failed";
        return false;
      }
      return true;
    };

    // 1. the CFA stages on the own rows
    int rc = dt_hip_pipe_band_begin(pipe, &b, dev_in[k], &st[k]);
    if(rc != DT_HIP_SUCCESS) fail(rc);
    else walking = true;
    if(rcs[k] >= 0 && st[k].clipped_count
       && (hipMemcpyAsync(&counts[k], st[k].clipped_count, sizeof(unsigned long long), hipMemcpyDeviceToHost, s) != hipSuccess
           || hipStreamSynchronize(s) != hipSuccess))
      fail(DT_HIP_DEFAULT_ERROR);
    gang.meet(); // the one meeting of the walk: eight bytes per band through the host (the CFA stages are short)
    if(gang.has_failed()) return give_up();
    // 2. the bypass of the highlight clipping is decided on the frame's count
    if(st[k].clipped_count)
    {
      unsigned long long total = 0;
      for(int j = 0; j < n; j++) total += counts[j];
      if(hipMemcpyAsync(st[k].clipped_count, &total, sizeof(total), hipMemcpyHostToDevice, s) != hipSuccess
         || hipStreamSynchronize(s) != hipSuccess) // `total` is a stack variable
        fail(DT_HIP_DEFAULT_ERROR);
    }
    if(rcs[k] >= 0 && (rc = dt_hip_pipe_band_resolve(pipe, &b, &st[k])) != DT_HIP_SUCCESS) fail(rc);
    if(rcs[k] < 0) return give_up();
    int pt = 0; // the next point this band publishes; the same number on every band at the same place of the walk
    // 3. mosaic rows the demosaic reads beyond the band: point 0 = "my CFA rows are final", point 1 = "I have pulled"
    if(st[k].halo_buf)
    {
      if(!gang.publish(k, s)) return fail(DT_HIP_DEFAULT_ERROR, "hipEventRecord"), give_up();
      char *const mine = (char *)st[k].halo_buf;
      const size_t rb = st[k].row_bytes;
      if(k > 0 && b.halo_top && st[k - 1].halo_buf)
      {
        if(!await(k - 1, pt)) return give_up();
        const dt_hip_band_t &ub = bands[k - 1];
        rc = copy_between(gang, devid, mine, pipes[k - 1]->devid,
                          (const char *)st[k - 1].halo_buf + (size_t)(ub.halo_top + ub.rows - b.halo_top) * rb,
                          (size_t)b.halo_top * rb, s);
        if(rc != DT_HIP_SUCCESS) return fail(rc), give_up();
      }
      if(k + 1 < n && b.halo_bottom && st[k + 1].halo_buf)
      {
        if(!await(k + 1, pt)) return give_up();
        const dt_hip_band_t &db = bands[k + 1];
        rc = copy_between(gang, devid, mine + (size_t)(b.halo_top + b.rows) * rb, pipes[k + 1]->devid,
                          (const char *)st[k + 1].halo_buf + (size_t)db.halo_top * rb, (size_t)b.halo_bottom * rb, s);
        if(rc != DT_HIP_SUCCESS) return fail(rc), give_up();
      }
      if(!gang.publish(k, s)) return fail(DT_HIP_DEFAULT_ERROR, "hipEventRecord"), give_up();
      // nobody frees or overwrites rows a neighbour is still pulling
      if((k > 0 && !await(k - 1, pt + 1)) || (k + 1 < n && !await(k + 1, pt + 1))) return give_up();
      pt += 2;
    }
    // 4. the walk, stopping where a stencil module needs its neighbours.  A stop is three points: "what the others
    //    read from me is written", "my turn of a relay is over", "I have pulled everything I need".
    for(;;)
    {
      rc = dt_hip_pipe_band_finish(pipe, &b, &st[k], dev_out[k]);
      if(rc < 0)
      {
        walking = false; // a finish() that failed has freed its state itself
        return fail(rc), give_up();
      }
      if(rc == DT_HIP_SUCCESS) break;
      stops++;
      if(!gang.publish(k, s)) return fail(DT_HIP_DEFAULT_ERROR, "hipEventRecord"), give_up();
      const bool everybody = st[k].relay_buf || (st[k].sum_buf && st[k].sum_planes > 0);
      if(st[k].relay_buf)
      {
        // local contrast: the bands splat their rows into the grid one after the other (the frame's pixel order), each
        // starting from the grid its predecessor left; the last band's grid is the frame's and goes to everybody
        if(k > 0)
        {
          if(!await(k - 1, pt + 1)) return give_up();
          if((rc = copy_between(gang, devid, st[k].relay_buf, pipes[k - 1]->devid, st[k - 1].relay_buf, st[k].relay_bytes, s)) != DT_HIP_SUCCESS)
            return fail(rc), give_up();
        }
        if((rc = dt_hip_pipe_band_relay(pipe, &b, &st[k])) != DT_HIP_SUCCESS) return fail(rc), give_up();
      }
      if(!gang.publish(k, s)) return fail(DT_HIP_DEFAULT_ERROR, "hipEventRecord"), give_up(); // point pt + 1
      if(st[k].relay_buf && k + 1 < n)
      {
        if(!await(n - 1, pt + 1)) return give_up();
        if((rc = copy_between(gang, devid, st[k].relay_buf, pipes[n - 1]->devid, st[n - 1].relay_buf, st[k].relay_bytes, s)) != DT_HIP_SUCCESS)
          return fail(rc), give_up();
      }
      if(st[k].sum_buf && st[k].sum_planes > 0)
      {
        const size_t plane = st[k].sum_count / (size_t)st[k].sum_planes, per_row = plane / (size_t)H;
        for(int j = 0; j < n; j++)
        {
          if(j == k) continue;
          if(!await(j, pt)) return give_up();
          const size_t off = (size_t)bands[j].row0 * per_row, len = (size_t)bands[j].rows * per_row;
          // kind Default: the runtime routes the strided copy between the two devices' memories (unified addressing)
          if(hipMemcpy2DAsync(st[k].sum_buf + off, plane * sizeof(double), st[j].sum_buf + off, plane * sizeof(double),
                              len * sizeof(double), (size_t)st[k].sum_planes, hipMemcpyDefault, s) != hipSuccess)
            return fail(DT_HIP_DEFAULT_ERROR), give_up();
          if(hip_device_of(pipes[j]->devid) != hip_device_of(devid))
          {
            gang.peer_bytes += len * sizeof(double) * (size_t)st[k].sum_planes;
            gang.peer_copies++;
          }
        }
      }
      if(st[k].halo_rows > 0 && st[k].halo_buf)
      {
        const int h = st[k].halo_rows;
        auto parts = [&](const int j, int &top, int &bottom) {
          top = std::min(h, bands[j].row0);
          bottom = std::min(h, H - bands[j].row0 - bands[j].rows);
        };
        int top, bottom;
        parts(k, top, bottom);
        char *const mine = (char *)st[k].halo_buf;
        const size_t rb = st[k].row_bytes;
        if((k > 0 && bands[k - 1].rows < top) || (k + 1 < n && bands[k + 1].rows < bottom))
          return fail(DT_HIP_INVALID_ARG, "dt_hip_pipe_process_bands: a band owns fewer rows than the halo its neighbour needs: use fewer bands"),
                 give_up();
        if(k > 0 && top)
        {
          int utop, ubot;
          parts(k - 1, utop, ubot);
          if(!await(k - 1, pt)) return give_up();
          rc = copy_between(gang, devid, mine, pipes[k - 1]->devid,
                            (const char *)st[k - 1].halo_buf + (size_t)(utop + bands[k - 1].rows - top) * rb, (size_t)top * rb, s);
          if(rc != DT_HIP_SUCCESS) return fail(rc), give_up();
        }
        if(k + 1 < n && bottom)
        {
          int dtop, dbot;
          parts(k + 1, dtop, dbot);
          if(!await(k + 1, pt)) return give_up();
          rc = copy_between(gang, devid, mine + (size_t)(top + b.rows) * rb, pipes[k + 1]->devid,
                            (const char *)st[k + 1].halo_buf + (size_t)dtop * rb, (size_t)bottom * rb, s);
          if(rc != DT_HIP_SUCCESS) return fail(rc), give_up();
        }
      }
      if(!gang.publish(k, s)) return fail(DT_HIP_DEFAULT_ERROR, "hipEventRecord"), give_up(); // point pt + 2
      // what the others pull from this band stays as it is until they have: the neighbours at a halo stop, everybody
      // where the table of sums or the grid travelled
      for(int j = 0; j < n; j++)
        if(j != k && (everybody || j == k - 1 || j == k + 1) && !await(j, pt + 2)) return give_up();
      pt += 3;
    }
    walking = false;
    if(hipStreamSynchronize(s) != hipSuccess)
    {
      (void)hipGetLastError();
      fail(DT_HIP_DEFAULT_ERROR, "the band's stream reported an error at the end of the walk");
    }
    gang.finished(k);
  };

  std::vector<std::thread> gangsters;
  gangsters.reserve(n);
  int started = 0;
  try
  {
    for(int k = 0; k < n; k++, started++) gangsters.emplace_back(worker, k);
  }
  catch(...)
  {
    // no thread for band `started`: the others must not wait for it at the meeting
    gang.fail();
    {
      std::lock_guard<std::mutex> lk(gang.m);
      gang.n = started;
      if(gang.waiting >= gang.n && gang.n > 0)
      {
        gang.waiting = 0;
        gang.generation++;
      }
      for(int k = started; k < n; k++) gang.done[k] = 1;
      gang.cv.notify_all();
    }
    for(int k = started; k < n; k++) rcs[k] = DT_HIP_DEFAULT_ERROR, errors[k] = "no host thread for this band";
  }
  for(auto &t : gangsters) t.join();
  for(auto &ev : gang.events)
    for(hipEvent_t e : ev) (void)hipEventDestroy(e);
  {
    std::lock_guard<std::mutex> lk(g_band_stats_mutex);
    g_band_stats.bands = n;
    int devs = 0;
    for(int k = 0; k < n; k++)
    {
      bool seen = false;
      for(int j = 0; j < k; j++) seen |= hip_device_of(pipes[j]->devid) == hip_device_of(pipes[k]->devid);
      devs += !seen;
    }
    g_band_stats.devices = devs;
    g_band_stats.exchange_stops = n ? stops.load() / n : 0;
    g_band_stats.peer_copies = gang.peer_copies.load();
    g_band_stats.peer_bytes = gang.peer_bytes.load();
    g_band_stats.host_wait_ns = gang.host_wait_ns.load();
    g_band_stats.pairs_without_peer_access = peer_missing.load();
  }
  for(int k = 0; k < n; k++)
    if(rcs[k] < 0 && errors[k] != "another band failed")
    {
      set_last_error("band %d of %d: %s", k, n, errors[k].c_str());
      return rcs[k];
    }
  for(int k = 0; k < n; k++)
    if(rcs[k] < 0)
    {
      set_last_error("band %d of %d: %s", k, n, errors[k].c_str());
      return rcs[k];
    }
  return DT_HIP_SUCCESS;
}

// ---- default_process_tiling_cl() for roi_in == roi_out, src/develop/tiling.c:842-1067 ----------------------------
// the tile plan of _default_process_tiling_cl_ptp(), :868-979, as a pure function of the frame, the module's
// requirements and the device's limits
int dt_hip_plan_tiles_ptp(int roi_width, int roi_height, int in_bpp, int out_bpp, const dt_hip_tiling_t *tiling,
                          unsigned filters, size_t available_bytes, size_t memalloc_bytes, int max_width, int max_height,
                          dt_hip_tile_plan_t *plan)
{
  if(!tiling || !plan || roi_width <= 0 || roi_height <= 0 || in_bpp <= 0 || out_bpp <= 0) return DT_HIP_INVALID_ARG;
  auto gcd = [](unsigned a, unsigned b) {
    while(b)
    {
      const unsigned t = b;
      b = a % b;
      a = t;
    }
    return a;
  };
  auto lcm = [&](unsigned a, unsigned b) { return (a && b) ? a / gcd(a, b) * b : 0u; };
  const int max_bpp = in_bpp > out_bpp ? in_bpp : out_bpp;
  const float available = (float)available_bytes;
  const float factor = fmaxf(tiling->factor_cl, 1.0f);
  const float singlebuffer = fminf(fmaxf((available - tiling->overhead) / factor, 0.0f), (float)memalloc_bytes);
  const float maxbuf = fmaxf(tiling->maxbuf_cl, 1.0f);
  int width = roi_width < max_width ? roi_width : max_width;
  int height = roi_height < max_height ? roi_height : max_height;
  // shrink the tile when it exceeds the per-buffer budget, :879-899
  if((float)width * height * max_bpp * maxbuf > singlebuffer)
  {
    const float scale = singlebuffer / ((float)width * height * max_bpp * maxbuf);
    if(width < height && scale >= 0.333f)
      height = (int)floorf(height * scale);
    else if(height <= width && scale >= 0.333f)
      width = (int)floorf(width * scale);
    else
    {
      width = (int)floorf(width * sqrtf(scale));
      height = (int)floorf(height * sqrtf(scale));
    }
  }
  // squares when the overlap would eat the tile, :901-907
  if(3 * tiling->overlap > (unsigned)width || 3 * tiling->overlap > (unsigned)height)
    width = height = (int)floorf(sqrtf((float)width * height));
  // alignment, :917-933 (CL_ALIGNMENT, :54: 4 unless X-Trans)
  const unsigned xyalign = lcm(tiling->xalign, tiling->yalign);
  const unsigned walign = lcm(xyalign, filters != 9u ? 4u : 1u);
  const unsigned halign = xyalign;
  if(!xyalign || !walign) return DT_HIP_INVALID_ARG;
  if(width < roi_width) width = (width / walign) * walign;
  if(height < roi_height) height = (height / halign) * halign;
  // the rounded-footprint loop, :941-950 (linear allocations are not rounded: dt_hip_dev_roundup_* are identities)
  while((float)width * height * max_bpp * maxbuf > singlebuffer)
  {
    if(width <= (int)walign && height <= (int)halign) break;
    if(width < height && height > (int)halign)
      height -= halign;
    else if(width > (int)walign)
      width -= walign;
    else
      height -= halign;
  }
  // :961-962
  auto align_down = [](int n, int a) { return n - n % a; };
  if(width < roi_width) width = std::max((int)walign, align_down(width, (int)walign));
  if(height < roi_height) height = std::max((int)halign, align_down(height, (int)halign));
  const int overlap = tiling->overlap % xyalign != 0 ? (tiling->overlap / xyalign + 1) * xyalign : tiling->overlap;
  plan->width = width;
  plan->height = height;
  plan->overlap = overlap;
  plan->tile_wd = width - 2 * overlap > 0 ? width - 2 * overlap : 1;
  plan->tile_ht = height - 2 * overlap > 0 ? height - 2 * overlap : 1;
  plan->tiles_x = width < roi_width ? (int)ceilf(roi_width / (float)plan->tile_wd) : 1;
  plan->tiles_y = height < roi_height ? (int)ceilf(roi_height / (float)plan->tile_ht) : 1;
  if((long)plan->tiles_x * plan->tiles_y > 10000) // _maximum_number_tiles(), :110-113
  {
    set_last_error("tiling: %d x %d tiles is too many", plan->tiles_x, plan->tiles_y);
    return DT_HIP_DEFAULT_ERROR;
  }
  return DT_HIP_SUCCESS;
}

// the tile loop, :981-1054: upload a tile of the host input, run the module on it with the tile's ROIs, download the
// part of its output that is not overlap.  available_bytes = 0 asks the device.
int dt_hip_default_process_tiling_ptp(int devid, const char *op, const dt_hip_piece_t *piece, const void *data,
                                      size_t data_size, const dt_hip_tiling_t *tiling, const void *host_in, void *host_out,
                                      int in_bpp, int out_bpp, size_t available_bytes)
{
  if(!valid_device(devid) || !op || !piece || !tiling || !host_in || !host_out) return DT_HIP_INVALID_ARG;
  node_t n;
  n.op = OP_UNKNOWN;
  for(int k = 0; k < (int)OP_UNKNOWN; k++)
    if(!strcmp(op, k_ops[k].name)) n.op = (op_t)k;
  if(n.op == OP_UNKNOWN || n.op == OP_BLEND || data_size != k_ops[n.op].data_size || (data_size && !data))
  {
    set_last_error("tiling: module '%s' cannot be tiled here", op);
    return DT_HIP_INVALID_ARG;
  }
  if(data_size) n.data.assign((const unsigned char *)data, (const unsigned char *)data + data_size);
  const dt_hip_roi_t &ri = piece->roi_in, &ro = piece->roi_out;
  if(ri.x != ro.x || ri.y != ro.y || ri.width != ro.width || ri.height != ro.height || ri.scale != ro.scale)
  {
    set_last_error("tiling: '%s' changes the geometry (roi_in != roi_out): only the point-to-point plan is implemented", op);
    return DT_HIP_INVALID_ARG;
  }
  int max_w = 0, max_h = 0;
  dt_hip_get_device_max_image_size(devid, &max_w, &max_h);
  dt_hip_tile_plan_t pl;
  int err = dt_hip_plan_tiles_ptp(ri.width, ri.height, in_bpp, out_bpp, tiling, piece->filters,
                                  available_bytes ? available_bytes : dt_hip_get_device_available(devid),
                                  dt_hip_get_device_memalloc(devid), max_w, max_h, &pl);
  if(err != DT_HIP_SUCCESS) return err;
  const size_t ipitch = (size_t)ri.width * in_bpp, opitch = (size_t)ro.width * out_bpp;
  hipStream_t st = stream_of(devid);
  for(int tx = 0; tx < pl.tiles_x; tx++)
    for(int ty = 0; ty < pl.tiles_y; ty++)
    {
      const int wd = tx * pl.tile_wd + pl.width > ri.width ? ri.width - tx * pl.tile_wd : pl.width;
      const int ht = ty * pl.tile_ht + pl.height > ri.height ? ri.height - ty * pl.tile_ht : pl.height;
      // end tiles that are all overlap carry nothing new, :990-991
      if((wd <= 2 * pl.overlap && tx > 0) || (ht <= 2 * pl.overlap && ty > 0)) continue;
      n.piece = *piece;
      n.piece.roi_in.x = ri.x + tx * pl.tile_wd;
      n.piece.roi_in.y = ri.y + ty * pl.tile_ht;
      n.piece.roi_in.width = n.piece.roi_out.width = wd;
      n.piece.roi_in.height = n.piece.roi_out.height = ht;
      n.piece.roi_out.x = ro.x + tx * pl.tile_wd;
      n.piece.roi_out.y = ro.y + ty * pl.tile_ht;
      const size_t ioffs = (size_t)ty * pl.tile_ht * ipitch + (size_t)tx * pl.tile_wd * in_bpp;
      size_t ooffs = (size_t)ty * pl.tile_ht * opitch + (size_t)tx * pl.tile_wd * out_bpp;
      dt_hip_mem_t input = dt_hip_alloc_device(devid, wd, ht, in_bpp), output = dt_hip_alloc_device(devid, wd, ht, out_bpp);
      err = (input && output) ? DT_HIP_SUCCESS : DT_HIP_SYSMEM_ALLOCATION;
      if(err == DT_HIP_SUCCESS)
        err = dt_hip_write_host_to_device_rowpitch(devid, (const char *)host_in + ioffs, input, wd, ht, in_bpp, ipitch, 1);
      // a module may leave part of its output to the caller (the alpha of the demosaic border ring): the tile buffer
      // comes from the pool, so give those bytes a value
      if(err == DT_HIP_SUCCESS && hipMemsetAsync(output, 0, (size_t)wd * ht * out_bpp, st) != hipSuccess) err = DT_HIP_DEFAULT_ERROR;
      if(err == DT_HIP_SUCCESS) err = run_single(devid, n, input, output);
      if(err == DT_HIP_SUCCESS)
      {
        // only the good part goes back, :1023-1040
        int ox = 0, oy = 0, rw = wd, rh = ht;
        if(tx > 0)
        {
          ox = pl.overlap;
          rw -= pl.overlap;
          ooffs += (size_t)pl.overlap * out_bpp;
        }
        if(ty > 0)
        {
          oy = pl.overlap;
          rh -= pl.overlap;
          ooffs += (size_t)pl.overlap * opitch;
        }
        const char *src = (const char *)output + ((size_t)oy * wd + ox) * out_bpp;
        if(hipMemcpy2DAsync((char *)host_out + ooffs, opitch, src, (size_t)wd * out_bpp, (size_t)rw * out_bpp, rh,
                            hipMemcpyDeviceToHost, st) != hipSuccess
           || hipStreamSynchronize(st) != hipSuccess)
        {
          set_last_error("tiling: download of tile (%d, %d) failed: %s", tx, ty, hipGetErrorString(hipGetLastError()));
          err = DT_HIP_DEFAULT_ERROR;
        }
      }
      if(input) dt_hip_release_mem_object(input);
      if(output) dt_hip_release_mem_object(output);
      if(err != DT_HIP_SUCCESS) return err;
    }
  return DT_HIP_SUCCESS;
}

// ---- default_process_tiling_cl() for roi_in != roi_out, src/develop/tiling.c:1076-1390 (_default_process_tiling_cl_roi)
// The only module of the export path whose output geometry differs from its input is finalscale, so its
// modify_roi_in() (src/iop/finalscale.c:76-107, the full-resolution pipeline of an export) is the one restated here.
namespace
{
static int ra_align_up(const int n, const int a) { return n + a - (n % a); } // tiling.c:92-95: one more step even when aligned
static int ra_align_down(const int n, const int a) { return n - (n % a); }
static int ra_align_close(const int n, const int a)
{
  const int off = n % a;
  const int shift = (off > a / 2) ? a - off : -off;
  return n + shift;
}

// finalscale modify_roi_in(), finalscale.c:76-107
static void finalscale_modify_roi_in(const dt_hip_roi_t *roi_out, dt_hip_roi_t *roi_in)
{
  *roi_in = *roi_out;
  if(roi_in->scale > 1.f)
  {
    roi_in->x = (int)roundf((float)roi_in->x / roi_out->scale);
    roi_in->y = (int)roundf((float)roi_in->y / roi_out->scale);
    roi_in->width = (int)roundf(roi_out->width / roi_out->scale);
    roi_in->height = (int)roundf(roi_out->height / roi_out->scale);
    roi_in->scale = 1.0f;
  }
  else
  {
    roi_in->width = (int)roundf(roi_out->width / roi_out->scale);
    roi_in->height = (int)roundf(roi_out->height / roi_out->scale);
    roi_in->scale = 1.0f;
    const float resample_scale = roi_out->scale / roi_in->scale;
    roi_in->x = (int)roundf(roi_in->x / resample_scale);
    roi_in->y = (int)roundf(roi_in->y / resample_scale);
  }
}

// _fit_output_to_input_roi(), tiling.c:197-237, its iterative search.  The Nelder-Mead fallback (:170-190) is for
// modules that distort; finalscale's search converges in one or two steps, so its failure is reported, not papered over
static bool fit_output_to_input_roi(const dt_hip_roi_t *iroi, dt_hip_roi_t *oroi, const int delta, int iter)
{
  dt_hip_roi_t probe = *iroi;
  finalscale_modify_roi_in(oroi, &probe);
  while((abs(probe.x - iroi->x) > delta || abs(probe.y - iroi->y) > delta || abs(probe.width - iroi->width) > delta
         || abs(probe.height - iroi->height) > delta)
        && iter > 0)
  {
    oroi->x += (iroi->x - probe.x) * oroi->scale / iroi->scale;
    oroi->y += (iroi->y - probe.y) * oroi->scale / iroi->scale;
    oroi->width += (iroi->width - probe.width) * oroi->scale / iroi->scale;
    oroi->height += (iroi->height - probe.height) * oroi->scale / iroi->scale;
    finalscale_modify_roi_in(oroi, &probe);
    iter--;
  }
  return iter > 0;
}
} // namespace

// the tile grid of :1100-1220 as a pure function of the two regions, the module's requirements and the device's limits
int dt_hip_plan_tiles_roi(const dt_hip_roi_t *roi_in, const dt_hip_roi_t *roi_out, int in_bpp, int out_bpp,
                          const dt_hip_tiling_t *tiling, unsigned filters, size_t available_bytes, size_t memalloc_bytes,
                          int max_width, int max_height, dt_hip_tile_plan_roi_t *plan)
{
  if(!roi_in || !roi_out || !tiling || !plan || roi_in->width <= 0 || roi_in->height <= 0 || roi_out->width <= 0
     || roi_out->height <= 0 || in_bpp <= 0 || out_bpp <= 0)
    return DT_HIP_INVALID_ARG;
  auto gcd = [](unsigned a, unsigned b) {
    while(b)
    {
      const unsigned t = b;
      b = a % b;
      a = t;
    }
    return a;
  };
  auto lcm = [&](unsigned a, unsigned b) { return (a && b) ? a / gcd(a, b) * b : 0u; };
  const int max_bpp = std::max(in_bpp, out_bpp);
  const float fullscale = fmaxf((float)(roi_in->scale / roi_out->scale),
                                sqrtf(((float)roi_in->width * roi_in->height) / ((float)roi_out->width * roi_out->height)));
  const int delta = (int)ceilf(fullscale);
  const int inacc = 5 * delta; // RESERVE, :59
  const float available = (float)available_bytes;
  const float factor = fmaxf(tiling->factor_cl, 1.0f);
  const float singlebuffer = fminf(fmaxf((available - tiling->overhead) / factor, 0.0f), (float)memalloc_bytes);
  const float maxbuf = fmaxf(tiling->maxbuf_cl, 1.0f);
  int width = std::min(std::max(roi_in->width, roi_out->width), max_width);
  int height = std::min(std::max(roi_in->height, roi_out->height), max_height);
  unsigned xyalign = lcm(tiling->xalign, tiling->yalign);
  xyalign = lcm(xyalign, filters != 9u ? 4u : 1u); // CL_ALIGNMENT, :54
  if(!xyalign) return DT_HIP_INVALID_ARG;
  const int al = (int)xyalign;
  if((float)width * height * max_bpp * maxbuf > singlebuffer)
  {
    const float scale = singlebuffer / ((float)width * height * max_bpp * maxbuf);
    if(width < height && scale >= 0.333f)
      height = ra_align_down((int)floorf(height * scale), al);
    else if(height <= width && scale >= 0.333f)
      width = ra_align_down((int)floorf(width * scale), al);
    else
    {
      width = ra_align_down((int)floorf(width * sqrtf(scale)), al);
      height = ra_align_down((int)floorf(height * sqrtf(scale)), al);
    }
  }
  if(3 * tiling->overlap > (unsigned)width || 3 * tiling->overlap > (unsigned)height)
    width = height = ra_align_down((int)floorf(sqrtf((float)width * height)), al);
  const int overlap_in = ra_align_up((int)tiling->overlap, al);
  const int overlap_out = (int)ceilf((float)overlap_in / fullscale);
  // the rounded-footprint loop, :1170-1179 (linear allocations are not rounded: dt_hip_dev_roundup_* are identities)
  while((float)width * height * max_bpp * maxbuf > singlebuffer)
  {
    if(width <= al && height <= al) break;
    if(width < height && height > al)
      height -= al;
    else if(width > al)
      width -= al;
    else
      height -= al;
  }
  if(width < std::max(roi_in->width, roi_out->width)) width = std::max(al, ra_align_down(width, al));
  if(height < std::max(roi_in->height, roi_out->height)) height = std::max(al, ra_align_down(height, al));
  int tiles_x = 1, tiles_y = 1;
  if(roi_in->width > roi_out->width)
    tiles_x = width < roi_in->width ? (int)ceilf((float)roi_in->width / (float)std::max(width - 2 * overlap_in - inacc, 1)) : 1;
  else
    tiles_x = width < roi_out->width ? (int)ceilf((float)roi_out->width / (float)std::max(width - 2 * overlap_out, 1)) : 1;
  if(roi_in->height > roi_out->height)
    tiles_y = height < roi_in->height ? (int)ceilf((float)roi_in->height / (float)std::max(height - 2 * overlap_in - inacc, 1)) : 1;
  else
    tiles_y = height < roi_out->height ? (int)ceilf((float)roi_out->height / (float)std::max(height - 2 * overlap_out, 1)) : 1;
  if((long)tiles_x * tiles_y > 10000)
  {
    set_last_error("tiling: %d x %d tiles is too many", tiles_x, tiles_y);
    return DT_HIP_DEFAULT_ERROR;
  }
  plan->width = width;
  plan->height = height;
  plan->tiles_x = tiles_x;
  plan->tiles_y = tiles_y;
  plan->tile_wd = ra_align_up(roi_out->width % tiles_x == 0 ? roi_out->width / tiles_x : roi_out->width / tiles_x + 1, al);
  plan->tile_ht = ra_align_up(roi_out->height % tiles_y == 0 ? roi_out->height / tiles_y : roi_out->height / tiles_y + 1, al);
  plan->overlap_in = overlap_in;
  plan->overlap_out = overlap_out;
  plan->delta = delta;
  plan->xyalign = al;
  return DT_HIP_SUCCESS;
}

// the three regions of tile (tx, ty), :1228-1300: the good part of the output, the input it is computed from (with
// overlap, alignment and `delta` of slack) and the output region that input produces
int dt_hip_tile_rois_finalscale(const dt_hip_tile_plan_roi_t *pl, const dt_hip_roi_t *roi_in, const dt_hip_roi_t *roi_out, int tx,
                                int ty, dt_hip_roi_t *iroi_full_out, dt_hip_roi_t *oroi_full_out, dt_hip_roi_t *oroi_good_out)
{
  if(!pl || !roi_in || !roi_out || tx < 0 || ty < 0 || tx >= pl->tiles_x || ty >= pl->tiles_y) return DT_HIP_INVALID_ARG;
  const int tile_wd = pl->tile_wd, tile_ht = pl->tile_ht, al = pl->xyalign, delta = pl->delta, overlap_in = pl->overlap_in;
  const int wd = (tx + 1) * tile_wd > roi_out->width ? roi_out->width - tx * tile_wd : tile_wd;
  const int ht = (ty + 1) * tile_ht > roi_out->height ? roi_out->height - ty * tile_ht : tile_ht;
  if(wd <= 0 || ht <= 0) return DT_HIP_TILE_EMPTY; // align_up() of the tile step can leave nothing for the last tile
  dt_hip_roi_t iroi_good = { roi_in->x + tx * tile_wd, roi_in->y + ty * tile_ht, wd, ht, roi_in->scale };
  dt_hip_roi_t oroi_good = { roi_out->x + tx * tile_wd, roi_out->y + ty * tile_ht, wd, ht, roi_out->scale };
  finalscale_modify_roi_in(&oroi_good, &iroi_good);
  iroi_good.x = std::max(iroi_good.x, roi_in->x);
  iroi_good.y = std::max(iroi_good.y, roi_in->y);
  iroi_good.width = std::min(iroi_good.width, roi_in->width + roi_in->x - iroi_good.x);
  iroi_good.height = std::min(iroi_good.height, roi_in->height + roi_in->y - iroi_good.y);
  const int x_in = iroi_good.x, y_in = iroi_good.y, width_in = iroi_good.width, height_in = iroi_good.height;
  const int new_x_in = std::max(ra_align_close(x_in - overlap_in - delta, al), roi_in->x);
  const int new_y_in = std::max(ra_align_close(y_in - overlap_in - delta, al), roi_in->y);
  const int new_width_in = std::min(ra_align_up(width_in + overlap_in + delta + (x_in - new_x_in), al), roi_in->width + roi_in->x - new_x_in);
  const int new_height_in = std::min(ra_align_up(height_in + overlap_in + delta + (y_in - new_y_in), al), roi_in->height + roi_in->y - new_y_in);
  dt_hip_roi_t iroi_full = { new_x_in, new_y_in, new_width_in, new_height_in, iroi_good.scale };
  dt_hip_roi_t oroi_full = oroi_good;
  if(!fit_output_to_input_roi(&iroi_full, &oroi_full, delta, 10))
  {
    set_last_error("tiling: no output region matches the input of tile (%d, %d)", tx, ty);
    return DT_HIP_DEFAULT_ERROR;
  }
  oroi_full.x = std::min(oroi_full.x, oroi_good.x);
  oroi_full.y = std::min(oroi_full.y, oroi_good.y);
  oroi_full.width = std::max(oroi_full.width, oroi_good.x + oroi_good.width - oroi_full.x);
  oroi_full.height = std::max(oroi_full.height, oroi_good.y + oroi_good.height - oroi_full.y);
  oroi_full.x = std::max(oroi_full.x, roi_out->x);
  oroi_full.y = std::max(oroi_full.y, roi_out->y);
  oroi_full.width = std::min(oroi_full.width, roi_out->width + roi_out->x - oroi_full.x);
  oroi_full.height = std::min(oroi_full.height, roi_out->height + roi_out->y - oroi_full.y);
  finalscale_modify_roi_in(&oroi_full, &iroi_full);
  iroi_full.x = std::max(iroi_full.x, roi_in->x);
  iroi_full.y = std::max(iroi_full.y, roi_in->y);
  iroi_full.width = std::min(iroi_full.width, roi_in->width + roi_in->x - iroi_full.x);
  iroi_full.height = std::min(iroi_full.height, roi_in->height + roi_in->y - iroi_full.y);
  if(iroi_full_out) *iroi_full_out = iroi_full;
  if(oroi_full_out) *oroi_full_out = oroi_full;
  if(oroi_good_out) *oroi_good_out = oroi_good;
  return DT_HIP_SUCCESS;
}

// the loop of :1222-1370: host frame -> every tile's full input region through the device -> the good part of its
// output back into the host frame.  `op` must be "finalscale" (the module whose modify_roi_in() is restated above)
int dt_hip_default_process_tiling_roi(int devid, const char *op, const dt_hip_piece_t *piece, const void *data, size_t data_size,
                                      const dt_hip_tiling_t *tiling, const void *host_in, void *host_out, int in_bpp,
                                      int out_bpp, size_t available_bytes)
{
  if(!valid_device(devid) || !op || !piece || !tiling || !host_in || !host_out) return DT_HIP_INVALID_ARG;
  if(strcmp(op, "finalscale") != 0)
  {
    set_last_error("tiling (roi_in != roi_out): '%s' has no modify_roi_in() here; finalscale is the one module of the path that "
                   "changes the geometry", op);
    return DT_HIP_INVALID_ARG;
  }
  node_t n;
  n.op = OP_FINALSCALE;
  if(data_size != k_ops[n.op].data_size || (data_size && !data)) return DT_HIP_INVALID_ARG;
  if(data_size) n.data.assign((const unsigned char *)data, (const unsigned char *)data + data_size);
  const dt_hip_roi_t &ri = piece->roi_in, &ro = piece->roi_out;
  int max_w = 0, max_h = 0;
  dt_hip_get_device_max_image_size(devid, &max_w, &max_h);
  dt_hip_tile_plan_roi_t pl;
  int err = dt_hip_plan_tiles_roi(&ri, &ro, in_bpp, out_bpp, tiling, piece->filters,
                                  available_bytes ? available_bytes : dt_hip_get_device_available(devid),
                                  dt_hip_get_device_memalloc(devid), max_w, max_h, &pl);
  if(err != DT_HIP_SUCCESS) return err;
  const size_t ipitch = (size_t)ri.width * in_bpp, opitch = (size_t)ro.width * out_bpp;
  hipStream_t st = stream_of(devid);
  for(int tx = 0; tx < pl.tiles_x; tx++)
    for(int ty = 0; ty < pl.tiles_y; ty++)
    {
      dt_hip_roi_t iroi_full, oroi_full, oroi_good;
      err = dt_hip_tile_rois_finalscale(&pl, &ri, &ro, tx, ty, &iroi_full, &oroi_full, &oroi_good);
      if(err == DT_HIP_TILE_EMPTY) continue;
      if(err != DT_HIP_SUCCESS) return err;
      const size_t ioffs = (size_t)(iroi_full.y - ri.y) * ipitch + (size_t)(iroi_full.x - ri.x) * in_bpp;
      const size_t ooffs = (size_t)(oroi_good.y - ro.y) * opitch + (size_t)(oroi_good.x - ro.x) * out_bpp;
      dt_hip_mem_t input = dt_hip_alloc_device(devid, iroi_full.width, iroi_full.height, in_bpp);
      dt_hip_mem_t output = dt_hip_alloc_device(devid, oroi_full.width, oroi_full.height, out_bpp);
      err = (input && output) ? DT_HIP_SUCCESS : DT_HIP_SYSMEM_ALLOCATION;
      if(err == DT_HIP_SUCCESS)
        err = dt_hip_write_host_to_device_rowpitch(devid, (const char *)host_in + ioffs, input, iroi_full.width, iroi_full.height,
                                                   in_bpp, ipitch, 1);
      if(err == DT_HIP_SUCCESS)
      {
        n.piece = *piece;
        n.piece.roi_in = iroi_full;
        n.piece.roi_out = oroi_full;
        err = run_single(devid, n, input, output);
      }
      if(err == DT_HIP_SUCCESS)
      {
        const char *src = (const char *)output + ((size_t)(oroi_good.y - oroi_full.y) * oroi_full.width + (oroi_good.x - oroi_full.x)) * out_bpp;
        if(hipMemcpy2DAsync((char *)host_out + ooffs, opitch, src, (size_t)oroi_full.width * out_bpp, (size_t)oroi_good.width * out_bpp,
                            oroi_good.height, hipMemcpyDeviceToHost, st) != hipSuccess
           || hipStreamSynchronize(st) != hipSuccess)
        {
          set_last_error("tiling: download of tile (%d, %d) failed: %s", tx, ty, hipGetErrorString(hipGetLastError()));
          err = DT_HIP_DEFAULT_ERROR;
        }
      }
      if(input) dt_hip_release_mem_object(input);
      if(output) dt_hip_release_mem_object(output);
      if(err != DT_HIP_SUCCESS) return err;
    }
  return DT_HIP_SUCCESS;
}

// default_tiling_callback(), src/develop/tiling.c:1423-1463, for the modules without a callback of their own
// (rawprepare, temperature, highlights, exposure, colorin, channelmixerrgb, filmicrgb, colorout, finalscale)
void dt_hip_default_tiling(const dt_hip_piece_t *piece, int before_demosaic, dt_hip_tiling_t *tiling)
{
  const float ioratio = ((float)piece->roi_out.width * (float)piece->roi_out.height)
                        / ((float)piece->roi_in.width * (float)piece->roi_in.height);
  tiling->factor = tiling->factor_cl = 1.0f + ioratio;
  tiling->maxbuf = tiling->maxbuf_cl = 1.0f;
  tiling->overhead = 0;
  tiling->overlap = 0;
  tiling->xalign = tiling->yalign = 1;
  if(before_demosaic && piece->filters) tiling->xalign = tiling->yalign = piece->filters == 9u ? 3 : 2;
}

// Give up a band between dt_hip_pipe_band_begin() and the last dt_hip_pipe_band_finish(): frees what the state holds
void dt_hip_pipe_band_abort(dt_hip_pipe_t *pipe, dt_hip_band_state_t *state)
{
  if(!pipe || !state || !state->priv) return;
  band_priv_t *pv = (band_priv_t *)state->priv;
  if(pv->walking)
  {
    if(pv->cur_owned && pv->cur_base) dt_hip_release_mem_object(pv->cur_base);
  }
  else if(pv->cfa_owned && pv->cfa)
    dt_hip_release_mem_object(pv->cfa);
  if(pv->out && pv->out_owned) dt_hip_release_mem_object(pv->out);
  if(pv->held_owned && pv->held_base) dt_hip_release_mem_object(pv->held_base);
  if(pv->journal) dt_hip_release_mem_object(pv->journal);
  if(pv->dn_job) denoiseprofile_band_abort(pv->dn_job);
  if(pv->relay) dt_hip_release_mem_object(pv->relay);
  delete pv;
  memset(state, 0, sizeof(*state));
}

int dt_hip_band_halo_rows(const char *op, const dt_hip_piece_t *piece, const void *data, size_t data_size)
{
  if(!op || !piece) return -1;
  node_t n;
  n.op = OP_UNKNOWN;
  for(int k = 0; k < (int)OP_UNKNOWN; k++)
    if(!strcmp(op, k_ops[k].name)) n.op = (op_t)k;
  if(n.op == OP_UNKNOWN || data_size != k_ops[n.op].data_size || (data_size && !data)) return -1;
  n.piece = *piece;
  if(data_size) n.data.assign((const unsigned char *)data, (const unsigned char *)data + data_size);
  return band_halo_rows(n);
}

// Resumable: returns DT_HIP_BAND_EXCHANGE in front of a stencil module (fill the halo rows of state->halo_buf)
// and in the middle of the profiled wavelets (all-reduce state->sum_buf); the caller does what the state
// asks for and calls again with the same arguments.
// The band's turn in a relay stop: its rows of the module input are accumulated on top of what relay_buf holds (the
// grid bands 0 .. k-1 left, copied in by the driver).
int dt_hip_pipe_band_relay(dt_hip_pipe_t *pipe, const dt_hip_band_t *band, dt_hip_band_state_t *state)
{
  if(!pipe || !band || !state || !state->priv) return DT_HIP_INVALID_ARG;
  band_priv_t *pv = (band_priv_t *)state->priv;
  if(!pv->walking || !pv->relay || pv->next_group >= pipe->groups.size())
  {
    set_last_error("dt_hip_pipe_band_relay: the band is not at a relay stop");
    return DT_HIP_INVALID_ARG;
  }
  const node_t &n = pipe->nodes[pipe->groups[pv->next_group].first];
  if(n.op != OP_BILAT || pv->stage != 1) return DT_HIP_INVALID_ARG;
  return bilat_band_splat(pipe->devid, &n.piece, n.as<dt_hip_bilat_data_t>(), pv->relay, pv->cur, band->row0, band->rows);
}

int dt_hip_pipe_band_finish(dt_hip_pipe_t *pipe, const dt_hip_band_t *band, dt_hip_band_state_t *state,
                            dt_hip_mem_t dev_out_band)
{
  if(!pipe || !band || !state || !state->priv || !dev_out_band) return DT_HIP_INVALID_ARG;
  band_priv_t *pv = (band_priv_t *)state->priv;
  const int devid = pipe->devid;
  const dt_hip_band_t &b = *band;
  const size_t ng = pipe->groups.size();
  const int W = pipe->nodes[0].piece.roi_out.width, H = pipe->nodes[0].piece.roi_out.height;
  const size_t rgba_row = (size_t)W * 16;
  int err = DT_HIP_SUCCESS;
  state->halo_rows = 0;
  state->sum_buf = nullptr;
  state->sum_count = 0;
  state->sum_planes = 0;
  state->relay_buf = nullptr;
  state->relay_bytes = 0;
  if(!pv->walking)
  {
    if(pv->journal) err = dt_hip_pipe_band_resolve(pipe, band, state); // caller skipped the explicit step
    pv->cur = pv->cur_base = pv->cfa;
    pv->cur_owned = pv->cfa_owned;
    pv->walking = true;
    if(pv->next_group >= ng && err == DT_HIP_SUCCESS)
    {
      // CFA-only pipe: the result is the band buffer itself
      const node_t &last = pipe->nodes.back();
      err = dt_hip_enqueue_copy_buffer_to_buffer(devid, pv->cur, dev_out_band, 0, 0,
                                                 (size_t)b.rows * last.piece.roi_out.width * 4);
    }
  }
  auto drop_cur = [&]() {
    if(pv->cur_owned && pv->cur_base && pv->cur_base != dev_out_band) dt_hip_release_mem_object(pv->cur_base);
    pv->cur = pv->cur_base = nullptr;
    pv->cur_owned = false;
  };
  auto next_is_blend = [&](const size_t gi) { return gi + 1 < ng && pipe->nodes[pipe->groups[gi + 1].first].op == OP_BLEND; };
  // the current buffer stops being the module input: free it, or keep it for the blend that follows the module
  auto retire_cur = [&](const size_t gi) {
    if(next_is_blend(gi))
    {
      pv->held = pv->cur;
      pv->held_base = pv->cur_base;
      pv->held_owned = pv->cur_owned;
      pv->cur = pv->cur_base = nullptr;
      pv->cur_owned = false;
    }
    else
      drop_cur();
  };
  // rows a stencil group takes from the neighbours, clipped at the frame
  auto halo_of = [&](const size_t gi, int &top, int &bottom) {
    const int h = band_halo_rows(pipe->nodes[pipe->groups[gi].first]);
    top = h < b.row0 ? h : b.row0;
    bottom = h < H - (b.row0 + b.rows) ? h : H - (b.row0 + b.rows);
    return h;
  };
  while(pv->next_group < ng && err == DT_HIP_SUCCESS)
  {
    const size_t gi = pv->next_group;
    const group_t &g = pipe->groups[gi];
    const node_t &first = pipe->nodes[g.first];
    const node_t &last = pipe->nodes[g.first + g.count - 1];
    const bool final_group = gi + 1 == ng || (next_is_blend(gi) && gi + 2 == ng);
    if(first.op == OP_BLEND)
    {
      // dt_develop_blend_process() after the module's process(), pixelpipe_cpu.c:137-228: in place in the output
      if(!pv->held)
      {
        set_last_error("pipe: a blend node needs the module it blends in front of it");
        err = DT_HIP_INVALID_ARG;
        break;
      }
      node_t n = first;
      band_piece(n.piece, b);
      dt_hip_blend_data_t bd = *n.as<dt_hip_blend_data_t>();
      // the host-rendered form mask is the FRAME's plane (every band's device holds it whole): the band reads its rows
      if(bd.form_mask) bd.form_mask = (dt_hip_mem_t)((float *)bd.form_mask + (size_t)b.row0 * first.piece.roi_out.width);
      err = dt_hip_develop_blend_process(devid, &n.piece, &bd, pv->held, pv->cur);
      if(pv->held_owned && pv->held_base) dt_hip_release_mem_object(pv->held_base);
      pv->held = pv->held_base = nullptr;
      pv->held_owned = false;
      pv->next_group++;
      continue;
    }
    if(g.kind == group_t::SINGLE && first.op == OP_BILAT)
    {
      // the bilateral grid: a relay stop (every band splats its rows in turn), then blur + slice of the own rows
      const dt_hip_bilat_data_t *d = first.as<dt_hip_bilat_data_t>();
      if(pv->stage == 0)
      {
        err = bilat_band_begin(devid, &first.piece, d, &pv->relay, &pv->relay_bytes);
        if(err != DT_HIP_SUCCESS) break;
        pv->stage = 1;
        state->relay_buf = pv->relay;
        state->relay_bytes = pv->relay_bytes;
        return DT_HIP_BAND_EXCHANGE;
      }
      dt_hip_mem_t out = dev_out_band;
      if(!final_group)
      {
        out = dt_hip_alloc_device_buffer(devid, (size_t)b.rows * rgba_row);
        if(!out)
        {
          err = DT_HIP_SYSMEM_ALLOCATION;
          break;
        }
      }
      err = bilat_band_finish(devid, &first.piece, d, pv->relay, pv->cur, out, b.row0, b.rows);
      dt_hip_release_mem_object(pv->relay); // stream-ordered
      pv->relay = nullptr;
      if(err != DT_HIP_SUCCESS)
      {
        if(out != dev_out_band) dt_hip_release_mem_object(out);
        break;
      }
      retire_cur(gi);
      pv->cur = pv->cur_base = out;
      pv->cur_owned = out != dev_out_band;
      pv->stage = 0;
      pv->next_group++;
      continue;
    }
    if(g.kind == group_t::SINGLE && is_stencil_op(first.op))
    {
      int top, bottom;
      const int h = halo_of(gi, top, bottom);
      const int buf_rows = top + b.rows + bottom;
      if(pv->stage == 0)
      {
        if(h < 0)
        {
          // the module does nothing on a frame this small (denoiseprofile.c:1325-1329): pass the rows through
          pv->stage = 3;
          continue;
        }
        if(!pv->cur_is_halo_layout)
        {
          // own rows into the middle of a [top][rows][bottom] buffer
          dt_hip_mem_t hb = dt_hip_alloc_device_buffer(devid, (size_t)buf_rows * rgba_row);
          if(!hb)
          {
            err = DT_HIP_SYSMEM_ALLOCATION;
            break;
          }
          err = dt_hip_enqueue_copy_buffer_to_buffer(devid, pv->cur, hb, 0, (size_t)top * rgba_row, (size_t)b.rows * rgba_row);
          drop_cur();
          pv->cur_base = hb;
          pv->cur = (char *)hb + (size_t)top * rgba_row;
          pv->cur_owned = true;
          if(err != DT_HIP_SUCCESS) break;
        }
        pv->cur_is_halo_layout = false;
        pv->stage = 1;
        if(top || bottom)
        {
          state->halo_buf = pv->cur_base;
          state->row_bytes = rgba_row;
          state->halo_rows = h;
          return DT_HIP_BAND_EXCHANGE;
        }
      }
      if(pv->stage == 1)
      {
        // the module on the buffer.  diffuse and the wavelets run on it as on a frame of its own: every row whose
        // stencils stay inside the buffer or hit a real frame border is exact, and the halo covers the rest.
        // non-local means keeps the frame's chunk grid and stores own rows only.
        state->halo_buf = nullptr;
        band_view_t v;
        v.frame_h = H;
        v.buf_row0 = b.row0 - top;
        v.row0 = b.row0;
        v.row1 = b.row0 + b.rows;
        // only diffuse runs on the whole buffer and leaves its halo rows in the output
        const bool own_rows_out = first.op != OP_DIFFUSE;
        dt_hip_mem_t out = dev_out_band;
        if(!(own_rows_out && final_group))
        {
          out = dt_hip_alloc_device_buffer(devid, (size_t)(own_rows_out ? b.rows : buf_rows) * rgba_row);
          if(!out)
          {
            err = DT_HIP_SYSMEM_ALLOCATION;
            break;
          }
        }
        pv->out = out;
        pv->out_own_rows = own_rows_out;
        pv->out_owned = out != dev_out_band;
        if(first.op == OP_NLMEANS)
          err = nlmeans_process_band(devid, &first.piece, first.as<dt_hip_nlmeans_data_t>(), &v, pv->cur_base, out);
        else if(first.op == OP_DIFFUSE)
        {
          dt_hip_piece_t p = first.piece;
          p.roi_in.height = p.roi_out.height = buf_rows;
          err = diffuse_process_rows(devid, &p, first.as<dt_hip_diffuse_data_t>(), v.buf_row0, pv->cur_base, out);
        }
        else
        {
          err = denoiseprofile_band_begin(devid, &first.piece, first.as<dt_hip_denoiseprofile_data_t>(), &v, buf_rows,
                                          pv->cur_base, out, &pv->dn_job);
          if(err == DT_HIP_SUCCESS && pv->dn_job) pv->stage = 2; // wavelets: one decomposition per step below
        }
        if(err != DT_HIP_SUCCESS)
        {
          if(out != dev_out_band) dt_hip_release_mem_object(out);
          pv->out = nullptr;
          break;
        }
        if(pv->stage != 2) pv->stage = 3;
      }
      if(pv->stage == 2)
      {
        // the profiled wavelets: a decomposition, then the neighbours' rows of its coarse plane for the next one; after
        // the last, the frame-wide sums; then thresholds and synthesis
        int rc;
        do
        {
          dt_hip_mem_t hbuf = nullptr;
          int hrows = 0;
          double *sums = nullptr;
          size_t count = 0;
          rc = denoiseprofile_band_step(pv->dn_job, &hbuf, &hrows, &sums, &count);
          if(rc < 0)
          {
            pv->dn_job = nullptr; // freed by the failing step
            err = rc;
            break;
          }
          if(rc > 0 && b.rows < H)
          {
            state->halo_buf = hbuf;
            state->halo_rows = hrows;
            state->row_bytes = rgba_row;
            state->sum_buf = sums;
            state->sum_count = count;
            // planes of [frame rows][segments][4]: each band's own rows are the only non-zero entries of its table
            state->sum_planes = count ? (int32_t)(count / ((size_t)H * ((W + 255) / 256) * 4)) : 0;
            return DT_HIP_BAND_EXCHANGE;
          }
        } while(rc > 0);
        if(err != DT_HIP_SUCCESS)
        {
          if(pv->out != dev_out_band) dt_hip_release_mem_object(pv->out);
          pv->out = nullptr;
          break;
        }
        state->halo_buf = nullptr;
        err = denoiseprofile_band_finish(pv->dn_job, pv->out);
        pv->dn_job = nullptr;
        if(err != DT_HIP_SUCCESS)
        {
          if(pv->out != dev_out_band) dt_hip_release_mem_object(pv->out);
          pv->out = nullptr;
          break;
        }
        pv->stage = 3;
      }
      // stage 3: the module's output becomes the current buffer
      if(pv->out)
      {
        retire_cur(gi);
        pv->cur_base = pv->out;
        pv->cur = pv->out_own_rows ? pv->out : (dt_hip_mem_t)((char *)pv->out + (size_t)top * rgba_row);
        pv->cur_owned = pv->out != dev_out_band;
        pv->out = nullptr;
      }
      if(final_group && pv->cur != dev_out_band)
      {
        err = dt_hip_enqueue_copy_buffer_to_buffer(devid, pv->cur_base, dev_out_band, (size_t)((char *)pv->cur - (char *)pv->cur_base),
                                                   0, (size_t)b.rows * rgba_row);
        drop_cur(); // stream-ordered; a blend that closes the pipe then works in dev_out_band
        pv->cur = pv->cur_base = dev_out_band;
      }
      pv->stage = 0;
      pv->next_group++;
      continue;
    }
    // demosaic and pointwise groups
    dt_hip_mem_t out = dev_out_band, out_base = dev_out_band;
    bool out_owned = false, out_halo_layout = false;
    if(!final_group)
    {
      node_t sized = last;
      sized.piece.roi_out.height = b.rows;
      size_t bytes = out_bytes(sized), lead = 0;
      const group_t &nx = pipe->groups[gi + 1];
      if(nx.kind == group_t::SINGLE && is_stencil_op(pipe->nodes[nx.first].op) && bytes == (size_t)b.rows * rgba_row)
      {
        // the next group is a stencil: write the own rows where its halo layout wants them
        int top, bottom;
        if(halo_of(gi + 1, top, bottom) >= 0)
        {
          bytes = (size_t)(top + b.rows + bottom) * rgba_row;
          lead = (size_t)top * rgba_row;
          out_halo_layout = true;
        }
      }
      out_base = dt_hip_alloc_device_buffer(devid, bytes);
      if(!out_base)
      {
        err = DT_HIP_SYSMEM_ALLOCATION;
        break;
      }
      out = (char *)out_base + lead;
      out_owned = true;
    }
    if(first.op == OP_DEMOSAIC)
    {
      const dt_hip_demosaic_data_t *d = first.as<dt_hip_demosaic_data_t>();
      if(d->demosaicing_method != DT_HIP_DEMOSAIC_RCD && d->demosaicing_method != DT_HIP_DEMOSAIC_AMAZE)
      {
        set_last_error("band mode: only the RCD and AMaZE demosaics run on row bands");
        err = DT_HIP_INVALID_ARG;
      }
      else
      {
        rcd_band_t rb;
        rb.tv0 = b.tile_row0;
        rb.tv1 = b.tile_row1;
        rb.in_row0 = b.row0 - b.halo_top;
        rb.in_rows = b.halo_top + b.rows + b.halo_bottom;
        rb.out_row0 = b.row0;
        rb.out_rows = b.rows;
        err = dt_hip_iop_demosaic_process_band(devid, &first.piece, d, &rb, pv->cur, out);
      }
    }
    else if(g.kind == group_t::RGB)
    {
      rgb_group_t r = g.rgb;
      r.height = b.rows;
      err = rgb_group_launch(devid, r, pv->cur, out);
    }
    else
    {
      node_t n = first;
      band_piece(n.piece, b);
      err = run_single(devid, n, pv->cur, out);
    }
    retire_cur(gi);
    pv->cur = out;
    pv->cur_base = out_base;
    pv->cur_owned = out_owned;
    pv->cur_is_halo_layout = out_halo_layout;
    pv->next_group++;
  }
  drop_cur();
  if(pv->held_owned && pv->held_base) dt_hip_release_mem_object(pv->held_base);
  if(pv->out && pv->out != dev_out_band) dt_hip_release_mem_object(pv->out);
  if(pv->dn_job) denoiseprofile_band_abort(pv->dn_job);
  delete pv;
  state->priv = nullptr;
  state->halo_buf = nullptr;
  state->clipped_count = nullptr;
  return err;
}

} // extern "C"
